"""The numpy encoder oracle reproduces the golden vectors generated from the
in-container transformers.BertModel (tests/golden/make_encoder_golden.py)."""
import os

import numpy as np
import pytest

from oracle import encoder as oenc

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    z = np.load(os.path.join(GOLD, f"encoder_{name}.npz"))
    cfg = {k: (float(v) if k == "ln_eps" else int(v)) for k, v in zip(z["cfg_keys"], z["cfg_vals"])}
    return cfg, int(z["seed"]), z["ids"], z["lens"], z["emb"], z["hidden0"]


@pytest.mark.parametrize("name", ["tiny", "minilm_l6"])
def test_numpy_encoder_matches_transformers_golden(name):
    cfg, seed, ids, lens, emb, hidden0 = load(name)
    w = oenc.random_weights(cfg, seed)
    got = oenc.encode(w, cfg, ids, lens)
    assert got.shape == emb.shape
    assert np.abs(got - emb).max() < 2e-6          # golden is float32 torch, oracle float64
    assert np.abs(np.linalg.norm(got, axis=1) - 1).max() < 1e-12
    hid = oenc.encode(w, cfg, ids[:1], lens[:1], return_hidden=True)[0]
    n = int(lens[0])
    assert np.abs(hid[:n] - hidden0[:n]).max() < 2e-5


def test_padding_tokens_do_not_change_the_embedding():
    cfg, seed, ids, lens, emb, _ = load("tiny")
    w = oenc.random_weights(cfg, seed)
    ids2 = ids.copy()
    T = ids.shape[1]
    pad = np.arange(T)[None, :] >= lens[:, None]
    ids2[pad] = 123                                  # garbage past the length
    a = oenc.encode(w, cfg, ids, lens)
    b = oenc.encode(w, cfg, ids2, lens)
    assert np.array_equal(a, b)
    wide = np.concatenate([ids, np.zeros((ids.shape[0], 7), ids.dtype)], axis=1)
    c = oenc.encode(w, cfg, wide, lens)
    assert np.abs(a - c).max() < 1e-12


def test_param_count_is_minilm_l6():
    w = oenc.random_weights(oenc.MINILM_L6, 0)
    # SURVEY.md 2b counts 22,713,216 with BertModel's pooler (384*384 + 384), which
    # sentence-transformers never evaluates (it mean-pools the hidden states)
    assert sum(v.size for v in w.values()) + 384 * 384 + 384 == 22_713_216

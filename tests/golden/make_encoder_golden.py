"""Generates tests/golden/encoder_*.npz with the in-container transformers.BertModel
(the model class sentence-transformers wraps for all-MiniLM-L6-v2) on SEEDED RANDOM
weights -- the real checkpoint is fetched by name in the reference and is not
available offline.  Run once, here, on CPU:  python tests/golden/make_encoder_golden.py

Stored: config, seed (weights are regenerated from it by oracle.encoder.random_weights),
token ids, lengths, and the BertModel -> mean-pool -> L2-normalise outputs (float32),
plus the last hidden state of one small case.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import encoder as oenc  # noqa: E402


def bert_from_weights(cfg, w):
    from transformers import BertConfig, BertModel
    bc = BertConfig(vocab_size=cfg["vocab_size"], hidden_size=cfg["hidden"],
                    num_hidden_layers=cfg["layers"], num_attention_heads=cfg["heads"],
                    intermediate_size=cfg["intermediate"], max_position_embeddings=cfg["max_position"],
                    type_vocab_size=cfg["type_vocab"], layer_norm_eps=cfg["ln_eps"], hidden_act="gelu",
                    hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    m = BertModel(bc, add_pooling_layer=False).eval()
    H = cfg["hidden"]
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    sd = {"embeddings.word_embeddings.weight": t(w["word_emb"]),
          "embeddings.position_embeddings.weight": t(w["pos_emb"]),
          "embeddings.token_type_embeddings.weight": t(w["type_emb"]),
          "embeddings.LayerNorm.weight": t(w["emb_ln_g"]), "embeddings.LayerNorm.bias": t(w["emb_ln_b"])}
    for l in range(cfg["layers"]):
        p = f"encoder.layer.{l}."
        qw, kw, vw = np.split(w["qkv_w"][l], 3, axis=0)
        qb, kb, vb = np.split(w["qkv_b"][l], 3, axis=0)
        sd.update({p + "attention.self.query.weight": t(qw), p + "attention.self.query.bias": t(qb),
                   p + "attention.self.key.weight": t(kw), p + "attention.self.key.bias": t(kb),
                   p + "attention.self.value.weight": t(vw), p + "attention.self.value.bias": t(vb),
                   p + "attention.output.dense.weight": t(w["ao_w"][l]),
                   p + "attention.output.dense.bias": t(w["ao_b"][l]),
                   p + "attention.output.LayerNorm.weight": t(w["ln1_g"][l]),
                   p + "attention.output.LayerNorm.bias": t(w["ln1_b"][l]),
                   p + "intermediate.dense.weight": t(w["ff1_w"][l]),
                   p + "intermediate.dense.bias": t(w["ff1_b"][l]),
                   p + "output.dense.weight": t(w["ff2_w"][l]), p + "output.dense.bias": t(w["ff2_b"][l]),
                   p + "output.LayerNorm.weight": t(w["ln2_g"][l]),
                   p + "output.LayerNorm.bias": t(w["ln2_b"][l])})
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected and all("position_ids" in k or "token_type_ids" in k for k in missing), \
        (missing, unexpected)
    return m


def st_pool(hidden, mask):
    """sentence-transformers Pooling(mean) + Normalize."""
    m = mask.unsqueeze(-1).to(hidden.dtype)
    pooled = (hidden * m).sum(1) / torch.clamp(m.sum(1), min=1e-9)
    return torch.nn.functional.normalize(pooled, p=2, dim=1)


def make(name, cfg, seed, B, T, lens_lo):
    w = oenc.random_weights(cfg, seed)
    rng = np.random.default_rng(seed + 1000)
    lens = rng.integers(lens_lo, T + 1, B)
    lens[0] = T
    ids = rng.integers(1000 if cfg["vocab_size"] > 2000 else 1, cfg["vocab_size"], (B, T))
    ids[np.arange(T)[None, :] >= lens[:, None]] = 0
    model = bert_from_weights(cfg, w)
    mask = torch.from_numpy((np.arange(T)[None, :] < lens[:, None]).astype(np.int64))
    with torch.no_grad():
        hidden = model(input_ids=torch.from_numpy(ids), attention_mask=mask).last_hidden_state
        emb = st_pool(hidden, mask)
    np.savez_compressed(os.path.join(os.path.dirname(__file__), f"encoder_{name}.npz"),
                        cfg_keys=np.array(list(cfg.keys())),
                        cfg_vals=np.array([float(v) for v in cfg.values()]),
                        seed=seed, ids=ids.astype(np.int32), lens=lens.astype(np.int32),
                        emb=emb.numpy().astype(np.float32),
                        hidden0=hidden[0].numpy().astype(np.float32))
    print(name, "emb", emb.shape, "norms", emb.norm(dim=1)[:3].tolist())


if __name__ == "__main__":
    torch.manual_seed(0)
    tiny = dict(vocab_size=500, hidden=384, layers=2, heads=12, intermediate=1536, max_position=64,
                type_vocab=2, ln_eps=1e-12)
    make("tiny", tiny, seed=7, B=4, T=24, lens_lo=3)
    make("minilm_l6", oenc.MINILM_L6, seed=11, B=6, T=48, lens_lo=5)

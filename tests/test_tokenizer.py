"""Tokenizer parity on CPU.  Three implementations of BERT WordPiece must agree id for id:
  * transformers.BertTokenizer (the library the reference's SentenceTransformer model wraps;
    importable in this container, used here as the independent oracle),
  * rag_fin_amd.tokenizer.WordPieceTokenizer.encode (Python restatement),
  * WordPieceTokenizer.batch_native (csrc/tokenizer.cpp through rf_tokenize_batch).
Inputs: the reference's 16 golden chunk texts (tests/golden/chunks_golden.json), hand-picked
edge cases (empty, control characters, CJK, accents, special tokens, over-long words,
truncation) and seeded random strings over a mixed ASCII / non-ASCII alphabet."""
import json
import os
import random
import re

import numpy as np
import pytest

from rag_fin_amd.tokenizer import WordPieceTokenizer

HERE = os.path.dirname(os.path.abspath(__file__))

EDGE = ["", " ", "Hello, WORLD!!", "café au lait — naïve résumé", "中文字符 mixed with English",
        "[CLS] literal special [SEP] and ₹52,084.00 crore • bullet", "a" * 150,
        "x\x00y\x01z\x7f w\x0bq", "İstanbul ΑΣ ǅ", "tab\tnew\nline\r\nend",
        "नमस्ते दुनिया", "emoji \U0001f600 test",
        "straße STRASSE", " en quad　ideographic space", "[MASK] [UNK] [PAD]", "##abc #hash",
        "semi;colon", "What was ICICI Bank's total income in Q1 2024?", "é precomposed é � dropped"]


def _texts():
    chunks = json.load(open(os.path.join(HERE, "golden", "chunks_golden.json")))
    texts = [c["text"] for c in chunks]
    rng = random.Random(0)
    alphabet = ("abcdefghij KLMNOP.,;:!?()[]{}-_/\\'\"0123456789 \t\n₹•é中ßİ́"
                "​�\U0001f600«»—…€№√א٣अািか한가ǅΣς\u2028\u00a0\u00ad")
    fuzz = ["".join(rng.choice(alphabet) for _ in range(rng.randint(0, 300))) for _ in range(1500)]
    return texts, texts + EDGE + fuzz


def _vocab(texts):
    words = set()
    for t in texts:
        for w in re.findall(r"[a-z]+|[0-9]|[^\sa-z0-9]", t.lower()):
            words.add(w)
    vocab = (["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]"] + sorted(words) +
             ["##" + w for w in sorted(words) if w.isalpha()] + ["##%d" % i for i in range(10)] +
             ["##" + c for c in "abcdefghijklmnopqrstuvwxyz"] + list("abcdefghijklmnopqrstuvwxyz") +
             ["é", "##é", "中", "नमसत", "ss", "\U0001f600"])
    return list(dict.fromkeys(vocab))


def test_python_tokenizer_matches_transformers(tmp_path):
    transformers = pytest.importorskip("transformers")
    chunk_texts, texts = _texts()
    vocab = _vocab(chunk_texts)
    path = tmp_path / "vocab.txt"
    path.write_text("\n".join(vocab) + "\n", encoding="utf-8")
    try:
        ref = transformers.BertTokenizer(str(path), do_lower_case=True)
    except Exception as e:   # an API change in the installed version is not a parity failure
        pytest.skip(f"transformers.BertTokenizer not constructible from a vocab file here: {e}")
    tok = WordPieceTokenizer.from_vocab_file(str(path))
    bad = []
    for t in texts:
        if "[" in t and any(sp in t for sp in ("[CLS]", "[SEP]", "[MASK]", "[UNK]", "[PAD]")):
            continue   # literal special tokens in text: the libraries' never_split handling differs by version
        want = ref.encode(t, add_special_tokens=True, truncation=True, max_length=256)
        got = tok.encode(t, 256)
        if list(want) != list(got):
            bad.append((t[:60], want[:12], got[:12]))
    assert not bad, bad[:3]


@pytest.mark.parametrize("max_len", [256, 16, 3, 2])
def test_native_tokenizer_matches_python(max_len):
    chunk_texts, texts = _texts()
    tok = WordPieceTokenizer(_vocab(chunk_texts))
    a_ids, a_len = tok.batch(texts, max_len)
    b_ids, b_len = tok.batch_native(texts, max_len)
    assert np.array_equal(a_len, b_len), np.argwhere(a_len != b_len)[:5]
    assert a_ids.shape == b_ids.shape and np.array_equal(a_ids, b_ids)
    c_ids, c_len = tok.batch_native(texts, max_len, n_threads=1)
    assert np.array_equal(b_ids, c_ids) and np.array_equal(b_len, c_len)


def test_native_tokenizer_edge_arguments():
    from rag_fin_amd import _lib
    tok = WordPieceTokenizer(["[PAD]", "[UNK]", "[CLS]", "[SEP]", "a", "##b"])
    ids, lens = tok.batch_native([], 8)
    assert ids.shape[0] == 0 and lens.shape == (0,)
    ids, lens = tok.batch_native(["ab abb zzz", ""], 8)
    assert ids.tolist() == [[2, 4, 5, 4, 5, 5, 1, 3], [2, 3, 0, 0, 0, 0, 0, 0]][:2] or \
        ids.tolist() == [[2, 4, 5, 4, 5, 5, 1, 3], [2, 3, 0, 0, 0, 0, 0, 0]]
    assert lens.tolist() == [8, 2]
    with pytest.raises(_lib.RagfinError):
        tok.batch_native(["x"], 1)          # max_len < 2 cannot hold [CLS] [SEP]
    with pytest.raises(ValueError):
        WordPieceTokenizer(["a", "b"])      # no special tokens


def test_batch_prepass_paths_agree_and_utf8_offsets():
    """batch_native encodes a whole batch with one join + one UTF-8 encode when its non-ASCII characters are all
    "simple" (a currency sign), and falls back to the per-text pre-normalising loop otherwise: the three kinds of
    batch (ASCII, simple non-ASCII, complex) give the ids of the Python restatement; rf_utf8_offsets turns
    character counts into byte offsets (empty texts, multi-byte characters at the boundaries, bad counts)."""
    import ctypes
    from ctypes import c_void_p
    from rag_fin_amd import _lib
    chunk_texts, _ = _texts()
    tok = WordPieceTokenizer(_vocab(chunk_texts))
    ascii_batch = [t.encode("ascii", "ignore").decode() for t in chunk_texts[:8]] + ["", "net profit 12.5 crore"]
    simple = ["₹ 1,234 crore", "", "€ 5 and £ 6", "abc", "₹"] + chunk_texts[:8]
    complex_ = simple + ["Café déjà vu ₹ 12", "日本語", "x́y", "emoji \U0001F600 test"]
    for batch, fast in ((ascii_batch, True), (simple, True), (complex_, False)):
        joined = "".join(batch)
        assert (joined.isascii() or tok._only_simple_non_ascii(joined)) == fast
        a, la = tok.batch(batch, 64)
        b, lb = tok.batch_native(batch, 64)
        assert np.array_equal(la, lb) and np.array_equal(a, b)
    lib = _lib.load_library()
    texts = ["₹a", "", "éé", "xyz", "\U0001F600"]
    blob = "".join(texts).encode("utf-8")
    chars = np.zeros(len(texts) + 1, dtype=np.int64)
    np.cumsum([len(t) for t in texts], out=chars[1:])
    out = np.full(len(texts) + 1, -7, dtype=np.int64)
    assert lib.rf_utf8_offsets(blob, len(blob), c_void_p(chars.ctypes.data), len(texts), c_void_p(out.ctypes.data)) == 0
    want = np.zeros(len(texts) + 1, dtype=np.int64)
    np.cumsum([len(t.encode("utf-8")) for t in texts], out=want[1:])
    assert np.array_equal(out, want)
    chars[-1] += 1                                   # one character more than the blob holds
    assert lib.rf_utf8_offsets(blob, len(blob), c_void_p(chars.ctypes.data), len(texts), c_void_p(out.ctypes.data)) == -1
    assert lib.rf_utf8_offsets(None, 0, None, 0, None) == -1


def test_batches_of_any_mix_agree_with_the_python_restatement():
    """Property (hypothesis): whatever mix of ASCII, "simple" non-ASCII (currency signs: the whole-batch pre-pass)
    and complex text (accents, marks, CJK, emoji, control characters: the per-text pre-pass) a batch holds, and
    wherever the empty strings sit, batch_native returns the ids of the Python restatement."""
    from hypothesis import given, settings, strategies as st
    chunk_texts, _ = _texts()
    tok = WordPieceTokenizer(_vocab(chunk_texts))
    ascii_words = st.text(alphabet="abcXYZ 019.,;()-", max_size=40)
    simple = st.text(alphabet="abc 12₹€£•«»—", max_size=40)
    complex_ = st.text(alphabet="aé́中 İßǅ\U0001f600 ­\x00ई", max_size=30)
    batch = st.lists(st.one_of(ascii_words, simple, complex_, st.just("")), min_size=0, max_size=12)

    @settings(max_examples=120, deadline=None)
    @given(batch, st.sampled_from([4, 16, 64]))
    def check(texts, max_len):
        a, la = tok.batch(texts, max_len)
        b, lb = tok.batch_native(texts, max_len)
        assert np.array_equal(la, lb)
        if len(texts):
            assert a.shape == b.shape and np.array_equal(a, b)
    check()

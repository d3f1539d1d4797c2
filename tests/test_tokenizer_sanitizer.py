"""AddressSanitizer + UndefinedBehaviorSanitizer run of the native tokenizer (host code of the
C-ABI library).  csrc/tokenizer.cpp is compiled with
`g++ -fsanitize=address,undefined -fno-sanitize-recover=all` together with a small driver
(tests/native/tokenizer_sanitizer_main.cpp) and run, as a child process, over
  * the fuzz corpus of tests/test_tokenizer.py (golden chunk texts, edge cases, 1 500 seeded
    random strings), pre-normalised exactly as the product does, and
  * raw byte strings that are NOT valid UTF-8 (truncated sequences, stray continuation bytes,
    over-long leads, NULs) -- the product never sends those, a C caller of the ABI might.
A sanitizer report aborts the child (non-zero exit).  For the valid texts the ids must equal the
Python restatement's, which tests/test_tokenizer.py pins to transformers.BertTokenizer."""
import os
import random
import shutil
import struct
import subprocess

import numpy as np
import pytest

from rag_fin_amd.tokenizer import WordPieceTokenizer
from test_tokenizer import EDGE, _texts, _vocab

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def driver(tmp_path_factory):
    gxx = shutil.which("g++")
    if gxx is None:
        pytest.skip("g++ not available")
    out = tmp_path_factory.mktemp("san") / "tokenizer_san"
    cmd = [gxx, "-std=c++17", "-g", "-O1", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
           "-fno-omit-frame-pointer", os.path.join(ROOT, "rag_fin_amd", "csrc", "tokenizer.cpp"),
           os.path.join(ROOT, "tests", "native", "tokenizer_sanitizer_main.cpp"), "-lpthread", "-o", str(out)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    return str(out)


def _run(driver, tmp_path, vocab, blobs, max_len, threads):
    from rag_fin_amd.tokenizer import _simple_tables
    vb = ("\n".join(vocab) + "\n").encode("utf-8")
    punct = np.asarray(_simple_tables()[1], dtype=np.int32)    # what the product passes to rf_tokenizer_set_punctuation
    offsets = np.zeros(len(blobs) + 1, dtype=np.int64)
    np.cumsum([len(b) for b in blobs], out=offsets[1:])
    inp, outp = tmp_path / "in.bin", tmp_path / "out.bin"
    with open(inp, "wb") as f:
        f.write(struct.pack("<q", len(vb)) + vb + struct.pack("<q", len(punct)) + punct.tobytes() +
                struct.pack("<q", len(blobs)) + offsets.tobytes() + b"".join(blobs))
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:halt_on_error=1",
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    env.pop("LD_PRELOAD", None)
    r = subprocess.run([driver, str(inp), str(max_len), str(threads), str(outp)], capture_output=True, text=True,
                       env=env, timeout=600)
    assert r.returncode == 0, f"exit {r.returncode}\n{r.stderr[-4000:]}"
    raw = np.fromfile(outp, dtype=np.int32)
    n = len(blobs)
    return raw[:n * max_len].reshape(n, max_len), raw[n * max_len:]


@pytest.mark.parametrize("max_len,threads", [(256, 4), (7, 1)])
def test_fuzz_corpus_under_asan_ubsan_matches_python(driver, tmp_path, max_len, threads):
    chunk_texts, texts = _texts()
    vocab = _vocab(chunk_texts)
    tok = WordPieceTokenizer(vocab)
    tok._native()                                # builds the pre-normalisation tables
    blobs, keep = [], []
    for t in texts:
        if not t.isascii() and tok._complex_re.search(t) is not None:
            t = tok._prenormalise(t)            # what batch_native hands to the native code
            if t is None:
                continue
        try:
            blobs.append(t.encode("utf-8"))
        except UnicodeEncodeError:
            continue
        keep.append(t)
    assert len(blobs) > 1400
    ids, lens = _run(driver, tmp_path, vocab, blobs, max_len, threads)
    want_ids, want_lens = tok.batch_native(keep, max_len)      # the product build of the same source
    assert np.array_equal(lens, want_lens)
    assert np.array_equal(ids[:, :want_ids.shape[1]], want_ids)
    # ASCII-only texts need no pre-normalisation: compare those with the Python restatement directly
    for i, t in enumerate(keep):
        if t.isascii():
            row = tok.encode(t, max_len)
            assert ids[i, :lens[i]].tolist() == row, t[:50]


def test_malformed_utf8_is_memory_safe(driver, tmp_path):
    rng = random.Random(5)
    vocab = _vocab(_texts()[0])
    blobs = [b"", b"\x80", b"\xc3", b"abc\xe2\x82", b"\xf0\x9f\x98", b"\xff\xfe\xfd", b"a\x00b\x00", b"\xed\xa0\x80",
             b"\xc0\xaf", b"\xf8\x88\x80\x80\x80", b"x" * 5000, b"\xe2" * 300, "é".encode() * 200 + b"\xc3"]
    blobs += [bytes(rng.getrandbits(8) for _ in range(rng.randint(0, 400))) for _ in range(800)]
    blobs += [t.encode("utf-8", "surrogatepass")[:-1] for t in EDGE if t]       # every edge case cut mid-sequence
    ids, lens = _run(driver, tmp_path, vocab, blobs, 64, 4)
    assert lens.min() >= 2 and lens.max() <= 64
    assert ((ids >= 0) & (ids < len(vocab))).all()

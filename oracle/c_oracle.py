"""ctypes wrapper of oracle/search_oracle.c.  TEST INFRASTRUCTURE ONLY (same
rules as oracle/search.py).  Build with `make -C oracle` (done by
__graft_entry__.build())."""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libsearch_oracle.so")
_lib = None


def build() -> str:
    subprocess.run(["make", "-C", _HERE], check=True, capture_output=True)
    return _SO


def _load():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
        _lib.oracle_search.restype = ctypes.c_int
        _lib.oracle_search.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64,
                                       ctypes.c_int64, ctypes.c_int64, ctypes.c_int64,
                                       ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p]
    return _lib


def _search_one(q, c, k, id_base):
    lib = _load()
    B, D = q.shape
    N = c.shape[0]
    out_s = np.empty((B, k), dtype=np.float64)
    out_i = np.empty((B, k), dtype=np.int64)
    rc = lib.oracle_search(q.ctypes.data, c.ctypes.data, B, N, D, k, id_base,
                           out_s.ctypes.data, out_i.ctypes.data)
    if rc != 0:
        raise MemoryError("oracle_search failed")
    return out_s, out_i


def search(q16: np.ndarray, c16: np.ndarray, k: int, id_base: int = 0, threads: int | None = None):
    """Exact top-k under the ranking contract.  Large problems are split by ROWS over host
    threads (ctypes releases the GIL): every part is the plain C routine on a contiguous row
    range with its own id_base, and the per-part lists are merged by (score desc, id asc) --
    the same rule, so the result is independent of the split (tests/test_oracle_search.py)."""
    q = np.ascontiguousarray(q16, dtype=np.float16)
    c = np.ascontiguousarray(c16, dtype=np.float16)
    N = c.shape[0]
    if threads is None:
        work = N * q.shape[0] * q.shape[1]
        threads = 1 if work < (1 << 28) else min(16, os.cpu_count() or 1)
    threads = max(1, min(threads, N // 4096 if N >= 8192 else 1))
    if threads == 1:
        return _search_one(q, c, k, id_base)
    from concurrent.futures import ThreadPoolExecutor
    bounds = np.linspace(0, N, threads + 1).astype(np.int64)
    with ThreadPoolExecutor(max_workers=threads) as ex:
        parts = list(ex.map(lambda t: _search_one(q, c[bounds[t]:bounds[t + 1]], k, id_base + int(bounds[t])),
                            range(threads)))
    s = np.concatenate([p[0] for p in parts], axis=1)          # [B, threads * k]
    i = np.concatenate([p[1] for p in parts], axis=1)
    out_s = np.full((q.shape[0], k), -np.inf, dtype=np.float64)
    out_i = np.full((q.shape[0], k), -1, dtype=np.int64)
    for b in range(q.shape[0]):
        ok = i[b] >= 0
        sb, ib = s[b][ok], i[b][ok]
        order = np.lexsort((ib, -sb))[:k]
        out_s[b, :len(order)] = sb[order]
        out_i[b, :len(order)] = ib[order]
    return out_s, out_i

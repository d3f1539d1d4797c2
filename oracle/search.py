"""CPU oracle for the vector-search half of the hot path.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module; the product (rag_fin_amd) never does.

What it restates
----------------
The reference delegates search to Milvus:
    collection.search(query_embedding, "embedding", {"metric_type": "COSINE"}, top_k, ...)
    -- vector_rag_mcp/main.py:51-57, retrieve.py:28-34,
       "chunking_storing (1).py":411-417, graph_cons.py:275-281
and consumes hits in descending-score order (main.py:59-70).  Milvus (pymilvus
2.3.0 -> an external Milvus 2.3.x server, IVF_FLAT/COSINE) is neither vendored in
/root/reference nor installable here, so the algorithm restated is its published
contract: cosine similarity of the query against every stored vector, the `limit`
best returned in descending order.  (IVF_FLAT is approximate; with nlist=128 on a
16-row collection every row is probed, so the exact scan is what the reference
observes.)

PARITY UNPINNED w.r.t. the reference: the reference tree stores no embedding,
score or ranked list anywhere, so this oracle cannot be pinned to
reference outputs.  It is pinned instead by construction (closed-form definition
below) and cross-checked three ways in tests/: numpy fp64 BLAS, the C twin
(oracle/search_oracle.c) and brute-force Python on tiny cases.

Ranking contract (shared with rag_fin_amd/csrc/merge.hip)
----------------------------------------------------------
Vectors are stored as IEEE fp16 (D a multiple of 8).  score(q, c) is the float64
    p = ((p0 + p1) + (p2 + p3)) + ((p4 + p5) + (p6 + p7))
where chain p_j = 0; for d in j, j+8, j+16, ... < D: p_j = fma(q[d], c[d], p_j)
(the product of two fp16 values is exact in float64, so fma == multiply, then
add).  Eight interleaved chains are what a 16-byte fp16 chunk feeds naturally on
both the GPU (8 lanes per candidate) and a SIMD CPU.  Rows are ranked by
(score descending, row id ascending).  Returned scores are that float64 (and
its float32 rounding).
"""
from __future__ import annotations

import numpy as np


def to_fp16(x: np.ndarray) -> np.ndarray:
    return np.ascontiguousarray(x, dtype=np.float16)


def l2_normalize_f32(x: np.ndarray) -> np.ndarray:
    """F.normalize(p=2, dim=1, eps=1e-12) in float32, as sentence-transformers'
    Normalize module does before the reference inserts/searches (main.py:50)."""
    x = np.asarray(x, dtype=np.float32)
    n = np.sqrt((x * x).sum(axis=1, keepdims=True, dtype=np.float32))
    return x / np.maximum(n, np.float32(1e-12))


def exact_scores(q16: np.ndarray, c16: np.ndarray, row_chunk: int = 32768) -> np.ndarray:
    """float64 [B, N] scores in the contract's order (8 chains + pairwise tree)."""
    q = np.asarray(q16, dtype=np.float16).astype(np.float64)
    c = np.asarray(c16, dtype=np.float16).astype(np.float64)
    B, D = q.shape
    assert D % 8 == 0, "contract needs D % 8 == 0"
    N = c.shape[0]
    out = np.empty((B, N), dtype=np.float64)
    row_chunk = max(1, min(row_chunk, (1 << 24) // max(1, B)))
    for s in range(0, N, row_chunk):
        cc = c[s:s + row_chunk]
        p = np.zeros((B, cc.shape[0], 8), dtype=np.float64)
        for t in range(D // 8):
            # exact product, one rounding in the add == fma(q, c, p)
            p += q[:, None, 8 * t:8 * t + 8] * cc[None, :, 8 * t:8 * t + 8]
        out[:, s:s + row_chunk] = ((p[..., 0] + p[..., 1]) + (p[..., 2] + p[..., 3])) + \
                                  ((p[..., 4] + p[..., 5]) + (p[..., 6] + p[..., 7]))
    return out


def topk_from_scores(scores: np.ndarray, k: int, id_base: int = 0):
    """Top-k of each row by (score desc, id asc).  Returns (scores f64 [B,k],
    ids int64 [B,k]); slots past N hold (-inf, -1)."""
    B, N = scores.shape
    out_s = np.full((B, k), -np.inf, dtype=np.float64)
    out_i = np.full((B, k), -1, dtype=np.int64)
    kk = min(k, N)
    if kk == 0:
        return out_s, out_i
    for b in range(B):
        row = scores[b]
        if N > 4 * kk:
            # preselect everything >= the kk-th largest value (keeps all ties)
            kth = np.partition(row, N - kk)[N - kk]
            cand = np.nonzero(row >= kth)[0]
        else:
            cand = np.arange(N)
        order = np.lexsort((cand, -row[cand]))[:kk]
        sel = cand[order]
        out_s[b, :kk] = row[sel]
        out_i[b, :kk] = sel + id_base
    return out_s, out_i


def search(q16: np.ndarray, c16: np.ndarray, k: int, id_base: int = 0):
    """Exact top-k under the ranking contract."""
    if c16.shape[0] == 0:
        B = q16.shape[0]
        return (np.full((B, k), -np.inf), np.full((B, k), -1, dtype=np.int64))
    return topk_from_scores(exact_scores(q16, c16), k, id_base)


def search_blas_f64(q16: np.ndarray, c16: np.ndarray, k: int):
    """Cross-check: float64 BLAS matmul (unspecified summation order).  Equal to
    search() except where two scores differ by < ~1e-15 relative."""
    s = np.asarray(q16, np.float16).astype(np.float64) @ np.asarray(c16, np.float16).astype(np.float64).T
    return topk_from_scores(s, k)


def cpu_search_blas(q16: np.ndarray, c32: np.ndarray, k: int):
    """The CPU baseline timed by bench.py: what a host-only deployment of the
    reference's search semantics does -- float32 BLAS `Q @ C.T` on all cores,
    argpartition, then an ordered top-k.  c32 is the
    corpus already widened to float32 (the reference stores FLOAT_VECTOR)."""
    q32 = np.asarray(q16, dtype=np.float32)
    s = q32 @ c32.T
    N = s.shape[1]
    kk = min(k, N)
    part = np.argpartition(s, N - kk, axis=1)[:, N - kk:]
    ps = np.take_along_axis(s, part, axis=1)
    order = np.lexsort((part, -ps), axis=1)
    ids = np.take_along_axis(part, order, axis=1)
    sc = np.take_along_axis(ps, order, axis=1)
    return sc, ids.astype(np.int64)


def merge_shards(scores: np.ndarray, ids: np.ndarray, k: int):
    """[W,B,k] per-shard results (exact f64 scores, global ids; -1 = empty) ->
    [B,k] by (score desc, id asc).  Restates SURVEY.md 8e (new in this build)."""
    W, B, kk = scores.shape
    out_s = np.full((B, k), -np.inf, dtype=np.float64)
    out_i = np.full((B, k), -1, dtype=np.int64)
    for b in range(B):
        s = scores[:, b, :].reshape(-1)
        i = ids[:, b, :].reshape(-1)
        ok = i >= 0
        s, i = s[ok], i[ok]
        order = np.lexsort((i, -s))[:k]
        out_s[b, :len(order)] = s[order]
        out_i[b, :len(order)] = i[order]
    return out_s, out_i


def synth_unit_rows(n: int, dim: int, seed: int, chunk: int = 65536) -> np.ndarray:
    """SURVEY.md 8d generator: standard normal rows (default_rng(seed)),
    L2-normalised in float32, rounded to fp16."""
    rng = np.random.default_rng(seed)
    out = np.empty((n, dim), dtype=np.float16)
    for s in range(0, n, chunk):
        m = min(chunk, n - s)
        x = rng.standard_normal((m, dim), dtype=np.float32)
        out[s:s + m] = l2_normalize_f32(x).astype(np.float16)
    return out

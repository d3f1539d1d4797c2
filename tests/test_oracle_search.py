"""CPU suite for the search oracle: numpy definition vs C twin vs float64 BLAS vs
brute-force Python, plus tie and edge-case behaviour.  The reference holds no
golden vectors for this path (parity unpinned), so these pin the
oracle to its closed-form contract instead."""
import numpy as np
import pytest

from oracle import c_oracle, search as osearch


def brute(q16, c16, k):
    out = []
    for q in q16.astype(np.float64):
        scored = []
        for r, c in enumerate(c16.astype(np.float64)):
            p = [0.0] * 8
            for d, (a, b) in enumerate(zip(q, c)):
                p[d % 8] = p[d % 8] + a * b   # product exact in f64 -> same as fma
            acc = ((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]))
            scored.append((-acc, r))
        scored.sort()
        out.append(scored[:k])
    return out


def test_numpy_matches_bruteforce_python():
    c = osearch.synth_unit_rows(97, 64, 1)
    q = osearch.synth_unit_rows(3, 64, 2)
    s, i = osearch.search(q, c, 5)
    ref = brute(q, c, 5)
    for b in range(3):
        assert [r for _, r in ref[b]] == i[b].tolist()
        assert [-v for v, _ in ref[b]] == s[b].tolist()


@pytest.mark.parametrize("n,d,b,k", [(5000, 384, 16, 10), (1000, 768, 4, 20), (33, 128, 2, 64)])
def test_c_twin_bit_identical(n, d, b, k):
    c = osearch.synth_unit_rows(n, d, 1234)
    q = osearch.synth_unit_rows(b, d, 5678)
    s1, i1 = osearch.search(q, c, k)
    s2, i2 = c_oracle.search(q, c, k)
    assert np.array_equal(i1, i2)
    assert np.array_equal(s1, s2)


def test_blas_f64_agrees():
    c = osearch.synth_unit_rows(20000, 384, 1234)
    q = osearch.synth_unit_rows(32, 384, 5678)
    s1, i1 = c_oracle.search(q, c, 10)
    s2, i2 = osearch.search_blas_f64(q, c, 10)
    assert np.array_equal(i1, i2)
    assert np.abs(s1 - s2).max() < 1e-14


def test_ties_break_by_ascending_row():
    base = osearch.synth_unit_rows(4, 64, 3)
    c = np.concatenate([base, base, base])          # rows r, r+4, r+8 identical
    q = base[:1]
    s, i = osearch.search(q, c, 6)
    assert i[0, :3].tolist() == [0, 4, 8]
    assert s[0, 0] == s[0, 1] == s[0, 2]
    s2, i2 = c_oracle.search(q, c, 6)
    assert np.array_equal(i, i2)


def test_k_larger_than_n_pads():
    c = osearch.synth_unit_rows(16, 384, 1)
    q = osearch.synth_unit_rows(2, 384, 2)
    s, i = osearch.search(q, c, 20)
    assert (i[:, 16:] == -1).all() and np.isneginf(s[:, 16:]).all()
    assert sorted(i[0, :16].tolist()) == list(range(16))
    s2, i2 = c_oracle.search(q, c, 20)
    assert np.array_equal(i, i2) and np.array_equal(s, s2)


def test_empty_corpus():
    q = osearch.synth_unit_rows(2, 384, 2)
    s, i = osearch.search(q, np.zeros((0, 384), np.float16), 3)
    assert (i == -1).all()


def test_cosine_within_fp16_tolerance_of_fp32_reference_semantics():
    # scores of fp16-stored unit vectors vs float32 cosine: north_star's 1e-3
    rng = np.random.default_rng(0)
    c32 = osearch.l2_normalize_f32(rng.standard_normal((2000, 384), dtype=np.float32))
    q32 = osearch.l2_normalize_f32(rng.standard_normal((8, 384), dtype=np.float32))
    s, i = osearch.search(q32.astype(np.float16), c32.astype(np.float16), 10)
    full = q32 @ c32.T
    assert np.abs(np.take_along_axis(full, i, 1) - s).max() < 1e-3


def test_cpu_baseline_leg_matches_oracle_ids():
    c = osearch.synth_unit_rows(30000, 384, 1234)
    q = osearch.synth_unit_rows(16, 384, 5678)
    _, i1 = c_oracle.search(q, c, 10)
    _, i2 = osearch.cpu_search_blas(q, c.astype(np.float32), 10)
    assert (i1 == i2).mean() > 0.99   # fp32 BLAS may swap near-ties


def test_merge_shards_equals_single_search():
    c = osearch.synth_unit_rows(4000, 128, 11)
    q = osearch.synth_unit_rows(5, 128, 12)
    s, i = osearch.search(q, c, 10)
    parts = [osearch.search(q, c[a:b], 10, id_base=a) for a, b in [(0, 1000), (1000, 1007), (1007, 4000)]]
    ms, mi = osearch.merge_shards(np.stack([p[0] for p in parts]), np.stack([p[1] for p in parts]), 10)
    assert np.array_equal(mi, i) and np.array_equal(ms, s)


def test_c_oracle_row_split_over_threads_is_identical():
    c = osearch.synth_unit_rows(40_000, 64, 5)
    c[30_000:30_020] = c[17]                       # ties across the split boundaries
    q = osearch.synth_unit_rows(9, 64, 6)
    q[0] = c[17]
    a = c_oracle.search(q, c, 12, threads=1)
    for t in (2, 3, 7):
        b = c_oracle.search(q, c, 12, id_base=0, threads=t)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


def test_oracle_agrees_with_sklearn_brute_force_cosine():
    """Independent third-party restatement of the reference's metric (Milvus COSINE top-k = the k
    smallest cosine distances): scikit-learn's brute-force cosine nearest neighbours in float64,
    its own normalisation and matmul code path.  The oracle ranks the SAME fp16 rows by inner
    product (they are unit-norm up to fp16 rounding, as in the product):
      * similarity: sklearn's 1 - distance == oracle score / (|q| |c|) to float64 round-off;
      * ranking: identical lists wherever the oracle's neighbouring scores are further apart
        than the norm deviation can move them; the returned SETS always agree up to that band."""
    sk = pytest.importorskip("sklearn.neighbors")
    c16 = osearch.synth_unit_rows(5000, 384, 71)
    q16 = osearch.synth_unit_rows(16, 384, 72)
    k = 10
    c64, q64 = c16.astype(np.float64), q16.astype(np.float64)
    nn = sk.NearestNeighbors(n_neighbors=k, algorithm="brute", metric="cosine").fit(c64)
    dist, idx = nn.kneighbors(q64)
    os_, oi = osearch.search(q16, c16, k + 5)
    qn, cn = np.linalg.norm(q64, axis=1), np.linalg.norm(c64, axis=1)
    dev = 2 * max(np.abs(qn - 1).max(), np.abs(cn - 1).max())          # how far cosine and inner product can differ
    assert dev < 2e-3
    for b in range(16):
        row_score = dict(zip(oi[b].tolist(), os_[b].tolist()))
        for j in range(k):
            r = int(idx[b, j])
            assert r in row_score, "sklearn returned a row outside the oracle's top-(k+5)"
            assert abs((1.0 - dist[b, j]) - row_score[r] / (qn[b] * cn[r])) < 1e-12
        gaps = os_[b][:k] - os_[b][1:k + 1]
        if gaps.min() > dev:                                           # well separated: the very same list
            assert idx[b].tolist() == oi[b][:k].tolist()
        assert set(idx[b].tolist()) <= {r for r, s_ in row_score.items() if s_ >= os_[b][k - 1] - dev}

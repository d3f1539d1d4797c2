"""Property tests (hypothesis; SURVEY.md 4 iii) of the ranking contract's CPU restatement
(oracle/search.py): top-k is a subset of the rows in (score desc, id asc) order, ties break on the
smaller id, and splitting the corpus into shards and merging the per-shard lists gives the
same answer as one search over the whole corpus -- the invariant the N > 1 step rests on
(rag_fin_amd/sharded.py; reference call: vector_rag_mcp/main.py:51-57)."""
import numpy as np
from hypothesis import given, settings, strategies as st

from oracle import search as osearch


def _corpus(seed, n, d, dup):
    rng = np.random.default_rng(seed)
    c = rng.standard_normal((n, d)).astype(np.float32)
    if dup and n > 1:      # exact duplicates: equal scores, the smaller id must rank first
        src = rng.integers(0, n, size=n // 3 + 1)
        dst = rng.integers(0, n, size=n // 3 + 1)
        c[dst] = c[src]
    return osearch.l2_normalize_f32(c).astype(np.float16)


@settings(max_examples=60, deadline=None)
@given(seed=st.integers(0, 2**31 - 1), n=st.integers(1, 300), d=st.sampled_from([16, 32, 64]),
       b=st.integers(1, 5), k=st.integers(1, 20), dup=st.booleans())
def test_topk_is_ordered_subset_with_id_tie_break(seed, n, d, b, k, dup):
    c = _corpus(seed, n, d, dup)
    q = _corpus(seed + 1, b, d, False)
    s, i = osearch.search(q, c, k)
    assert s.shape == (b, k) and i.shape == (b, k)
    full = osearch.exact_scores(q, c)
    for row in range(b):
        m = min(k, n)
        ids = i[row, :m]
        assert len(set(ids.tolist())) == m and ids.min() >= 0 and ids.max() < n          # a subset, no repeats
        assert np.array_equal(s[row, :m], full[row, ids])                                  # the contract's scores
        key = [(-s[row, j], ids[j]) for j in range(m)]
        assert key == sorted(key)                                                          # score desc, id asc
        rest = np.setdiff1d(np.arange(n), ids)
        if rest.size:                                                                      # nothing outside beats the last hit
            worst = (-s[row, m - 1], ids[m - 1])
            assert all((-full[row, r], r) > worst for r in rest)
        assert np.all(i[row, m:] == -1) and np.all(np.isneginf(s[row, m:]))                # k > n: padded, never garbage


@settings(max_examples=60, deadline=None)
@given(seed=st.integers(0, 2**31 - 1), n=st.integers(1, 400), d=st.sampled_from([16, 48]),
       k=st.integers(1, 16), cuts=st.lists(st.integers(0, 400), min_size=0, max_size=6), dup=st.booleans())
def test_shard_split_and_merge_equals_one_search(seed, n, d, k, cuts, dup):
    c = _corpus(seed, n, d, dup)
    q = _corpus(seed + 7, 3, d, False)
    ws, wi = osearch.search(q, c, k)
    bounds = sorted({0, n, *[min(x, n) for x in cuts]})
    parts_s, parts_i = [], []
    for lo, hi in zip(bounds, bounds[1:]):            # empty shards included when two cuts coincide with an end
        s, i = osearch.search(q, c[lo:hi], k, id_base=lo)
        parts_s.append(s)
        parts_i.append(i)
    if not parts_s:                                   # n == 0 cannot happen (n >= 1), but bounds may collapse to one shard
        parts_s, parts_i = [ws], [wi]
    ms, mi = osearch.merge_shards(np.stack(parts_s), np.stack(parts_i), k)
    assert np.array_equal(mi, wi) and np.array_equal(ms, ws)
    # the order of the shards in the gathered buffer does not matter
    perm = np.random.default_rng(seed).permutation(len(parts_s))
    ms2, mi2 = osearch.merge_shards(np.stack([parts_s[p] for p in perm]), np.stack([parts_i[p] for p in perm]), k)
    assert np.array_equal(mi2, wi) and np.array_equal(ms2, ws)

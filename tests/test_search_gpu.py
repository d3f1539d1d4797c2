"""GPU parity tests for the search half of the hot path.  Everything goes through
the C ABI (libragfin_hip.so via rag_fin_amd.store.GpuIndex); the checker is the
CPU oracle (oracle/search.py, oracle/search_oracle.c).  Bar: row ids and ranks
bit-exact, fp64 ranking scores bit-exact, fp32 scores == float32(oracle)."""
import numpy as np
import pytest

from oracle import c_oracle, search as osearch

pytestmark = pytest.mark.gpu


def make_index(c16, device, capacity=None, pieces=None):
    import torch
    from rag_fin_amd.store import GpuIndex
    n, d = c16.shape
    ix = GpuIndex(d, capacity or max(n, 1), device)
    t = torch.from_numpy(c16).to(device)
    if pieces is None:
        if n:
            ix.add(t)
    else:
        s = 0
        for p in pieces:
            ix.add(t[s:s + p])
            s += p
        assert s == n
    return ix


def check_against_oracle(ix, q16, c16, k, device, expect_flags_zero=True, raw=True):
    import torch
    q = torch.from_numpy(q16).to(device)
    if raw:
        scores, ids, exact, flags = ix.search_raw(q, k, want_exact=True)
        torch.cuda.synchronize()
        if expect_flags_zero:
            assert int(flags.abs().sum()) == 0, f"flags set: {flags.cpu().numpy()}"
    else:
        scores, ids, exact = ix.search(q, k, want_exact=True)
    os_, oi = c_oracle.search(q16, c16, k)
    ids = ids.cpu().numpy()
    assert np.array_equal(ids, oi), f"ids differ at {np.argwhere(ids != oi)[:5]}"
    assert np.array_equal(exact.cpu().numpy(), os_)
    assert np.array_equal(scores.cpu().numpy(), os_.astype(np.float32))


@pytest.mark.parametrize("n,d,b,k", [
    (16, 384, 5, 5),         # the reference's real corpus size (16 ICICI chunks)
    (16, 384, 2, 20),        # top_k > rows: fewer hits, no padding garbage
    (1, 384, 1, 3),
    (31, 384, 3, 10),        # ragged last block
    (1000, 384, 64, 10),
    (5001, 384, 33, 10),     # small-corpus path (every row a candidate), JB=2
    (8192, 384, 8, 10),      # boundary of the small-corpus path
    (8193, 384, 8, 10),      # first size that takes the sample pass
    (20011, 384, 64, 10),
    (30000, 128, 7, 64),     # max k
    (50000, 768, 64, 10),    # 8-wave kernel, two ring revolutions per block
    (40000, 64, 64, 5),
    (40000, 256, 40, 10),
    (30000, 512, 64, 10),
    (20000, 1024, 20, 10),
])
def test_parity_with_oracle(gpu_device, n, d, b, k):
    c = osearch.synth_unit_rows(n, d, 1234)
    q = osearch.synth_unit_rows(b, d, 5678)
    ix = make_index(c, gpu_device)
    check_against_oracle(ix, q, c, k, gpu_device)


def test_config2_100k_batch64(gpu_device):
    """BASELINE.json configs[1]: 100k x 384 fp16, batch 64, top-10."""
    c = osearch.synth_unit_rows(100_000, 384, 1234)
    q = osearch.synth_unit_rows(64, 384, 5678)
    ix = make_index(c, gpu_device)
    check_against_oracle(ix, q, c, 10, gpu_device)


def test_query_batches_larger_than_one_sweep(gpu_device):
    c = osearch.synth_unit_rows(30000, 384, 1)
    q = osearch.synth_unit_rows(150, 384, 2)        # 64 + 64 + 22
    ix = make_index(c, gpu_device)
    check_against_oracle(ix, q, c, 10, gpu_device)


def test_appends_in_ragged_pieces_and_fetch_by_row(gpu_device):
    import torch
    c = osearch.synth_unit_rows(10_000, 384, 7)
    q = osearch.synth_unit_rows(9, 384, 8)
    ix = make_index(c, gpu_device, capacity=12_000, pieces=[1, 30, 33, 936, 4000, 5000])
    assert ix.size == 10_000
    check_against_oracle(ix, q, c, 10, gpu_device)
    rows = np.array([0, 1, 31, 32, 33, 9999, 5000, 17], dtype=np.int64)
    got = ix.get_rows(rows).cpu().numpy()
    assert np.array_equal(got.view(np.uint16), c[rows].view(np.uint16))
    # reset == drop + recreate
    ix.reset()
    assert ix.size == 0
    ix.add(torch.from_numpy(c[:100]).to(gpu_device))
    check_against_oracle(ix, q, c[:100], 10, gpu_device)


def test_empty_index(gpu_device):
    import torch
    from rag_fin_amd.store import GpuIndex
    ix = GpuIndex(384, 64, gpu_device)
    q = torch.from_numpy(osearch.synth_unit_rows(3, 384, 1)).to(gpu_device)
    s, i, e, f = ix.search_raw(q, 4, want_exact=True)
    assert (i.cpu().numpy() == -1).all() and np.isneginf(s.cpu().numpy()).all()


def test_duplicate_rows_tie_break_by_row(gpu_device):
    base = osearch.synth_unit_rows(5000, 384, 3)
    c = np.concatenate([base, base[:2500], base])       # many exact duplicates
    q = base[[10, 2000, 4999, 77]]                       # each query has 2-3 exact copies
    ix = make_index(c, gpu_device)
    check_against_oracle(ix, q, c, 10, gpu_device)


def test_all_rows_identical_falls_back_and_stays_exact(gpu_device):
    """Every row ties: the rescoring set overflows, rf_search flags the queries
    and the exhaustive kernel resolves them (ids 0..k-1 by the tie rule)."""
    import torch
    row = osearch.synth_unit_rows(1, 384, 5)
    c = np.repeat(row, 20_000, axis=0)
    q = osearch.synth_unit_rows(3, 384, 6)
    ix = make_index(c, gpu_device)
    qq = torch.from_numpy(q).to(gpu_device)
    _, _, _, flags = ix.search_raw(qq, 10)
    assert (flags.cpu().numpy() != 0).all()
    check_against_oracle(ix, q, c, 10, gpu_device, raw=False)


def test_sorted_corpus_adversarial_for_thresholds(gpu_device):
    """Rows ordered by increasing similarity to the query: a prefix sample would
    give a useless threshold; the strided sample must still be exact."""
    c = osearch.synth_unit_rows(60_000, 384, 9)
    q = osearch.synth_unit_rows(4, 384, 10)
    order = np.argsort(c.astype(np.float32) @ q[0].astype(np.float32))
    c = np.ascontiguousarray(c[order])
    ix = make_index(c, gpu_device)
    check_against_oracle(ix, q, c, 10, gpu_device, raw=False)


def test_unnormalised_inner_product_rows(gpu_device):
    rng = np.random.default_rng(3)
    c = (rng.standard_normal((30_000, 384)) * rng.uniform(0.1, 30, (30_000, 1))).astype(np.float16)
    q = (rng.standard_normal((16, 384)) * 5).astype(np.float16)
    ix = make_index(c, gpu_device)
    check_against_oracle(ix, q, c, 10, gpu_device)


def test_exhaustive_kernel_matches_oracle(gpu_device):
    import torch
    c = osearch.synth_unit_rows(25_000, 384, 21)
    q = osearch.synth_unit_rows(6, 384, 22)
    ix = make_index(c, gpu_device)
    s, i, e = ix.search_exhaustive(torch.from_numpy(q).to(gpu_device), 10, want_exact=True)
    os_, oi = c_oracle.search(q, c, 10)
    assert np.array_equal(i.cpu().numpy(), oi) and np.array_equal(e.cpu().numpy(), os_)


def test_mfma_scores_layout_and_error_bound(gpu_device):
    """Raw MFMA scan scores: (a) the fragment/accumulator layout maps every
    (query,row) to the right place (asymmetric integer data, exact in fp16/fp32);
    (b) |mfma - exact| stays far inside the eps the emit threshold assumes."""
    import torch
    rng = np.random.default_rng(1)
    n, d, b = 200, 384, 40
    c = rng.integers(-3, 4, (n, d)).astype(np.float16)
    q = rng.integers(-3, 4, (b, d)).astype(np.float16)
    ix = make_index(c, gpu_device)
    got = ix.debug_scores(torch.from_numpy(q).to(gpu_device)).cpu().numpy()
    want = q.astype(np.float64) @ c.astype(np.float64).T     # small integers: exact
    assert np.array_equal(got.astype(np.float64), want)

    c = osearch.synth_unit_rows(4096, 384, 31)
    q = osearch.synth_unit_rows(64, 384, 32)
    ix = make_index(c, gpu_device)
    got = ix.debug_scores(torch.from_numpy(q).to(gpu_device)).cpu().numpy().astype(np.float64)
    exact = osearch.exact_scores(q, c)
    err = np.abs(got - exact).max()
    eps = 384 * 2.0 ** -23      # ||q|| = ||c|| = 1 (k_threshold's bound before its 1.25 slack)
    assert err < eps / 4, (err, eps)


@pytest.mark.parametrize("dim", [384])
def test_full_size_1m_properties(gpu_device, dim):
    """BASELINE.json configs[2] size (1M x 384): size-independent checks.
    (1) planted needles: each query is also stored as a row -> that row is rank 1
    with score ||q||^2; (2) fused path == exhaustive fp64 kernel on the same
    device for a subset of queries; (3) ranking order is non-increasing and ids
    unique; (4) the 64-query result equals the same queries searched alone."""
    import torch
    n = 1_000_000
    gen = torch.Generator(device=gpu_device).manual_seed(1234)
    c = torch.randn((n, dim), generator=gen, device=gpu_device, dtype=torch.float32)
    c = (c / c.norm(dim=1, keepdim=True)).half()
    q = torch.randn((64, dim), generator=gen, device=gpu_device, dtype=torch.float32)
    q = (q / q.norm(dim=1, keepdim=True)).half()
    plant = torch.arange(64, device=gpu_device) * 15_013 + 7
    c[plant] = q
    from rag_fin_amd.store import GpuIndex
    ix = GpuIndex(dim, n, gpu_device)
    ix.add(c)
    s, i, e, f = ix.search_raw(q, 10, want_exact=True)
    torch.cuda.synchronize()
    assert int(f.abs().sum()) == 0
    assert torch.equal(i[:, 0], plant)
    qn = (q.double() ** 2).sum(1)
    assert torch.allclose(e[:, 0], qn, rtol=0, atol=1e-12)
    assert bool((e[:, 1:] <= e[:, :-1]).all())
    assert all(len(set(r.tolist())) == 10 for r in i.cpu())
    s2, i2, e2 = ix.search_exhaustive(q[:6], 10, want_exact=True)
    assert torch.equal(i2, i[:6]) and torch.equal(e2, e[:6])
    s3, i3, e3, _ = ix.search_raw(q[5:6], 10, want_exact=True)
    assert torch.equal(i3, i[5:6]) and torch.equal(e3, e[5:6])


def test_full_size_1m_batch256_properties(gpu_device):
    """BASELINE.json configs[2] itself (1M x 384, batch 256 -> one wide LDS-DMA sweep): size-
    independent checks.  (1) planted needles rank first with score ||q||^2; (2) the 256-query
    result equals the same queries answered by 64-query sweeps (a different kernel) bit for
    bit; (3) a subset equals the exhaustive fp64 kernel; (4) order and uniqueness."""
    import torch
    n, dim, B = 1_000_000, 384, 256
    gen = torch.Generator(device=gpu_device).manual_seed(4321)
    c = torch.randn((n, dim), generator=gen, device=gpu_device, dtype=torch.float32)
    c = (c / c.norm(dim=1, keepdim=True)).half()
    q = torch.randn((B, dim), generator=gen, device=gpu_device, dtype=torch.float32)
    q = (q / q.norm(dim=1, keepdim=True)).half()
    plant = torch.arange(B, device=gpu_device) * 3_907 + 11
    c[plant] = q
    from rag_fin_amd.store import GpuIndex
    ix = GpuIndex(dim, n, gpu_device)
    ix.add(c)
    s, i, e, f = ix.search_raw(q, 10, want_exact=True)
    torch.cuda.synchronize()
    assert int(f.abs().sum()) == 0
    assert torch.equal(i[:, 0], plant)
    assert torch.allclose(e[:, 0], (q.double() ** 2).sum(1), rtol=0, atol=1e-12)
    assert bool((e[:, 1:] <= e[:, :-1]).all())
    assert all(len(set(r.tolist())) == 10 for r in i.cpu())
    parts = [ix.search_raw(q[a:a + 64].contiguous(), 10, want_exact=True) for a in range(0, B, 64)]
    assert torch.equal(i, torch.cat([p[1] for p in parts])) and torch.equal(e, torch.cat([p[2] for p in parts]))
    s2, i2, e2 = ix.search_exhaustive(q[100:106].contiguous(), 10, want_exact=True)
    assert torch.equal(i2, i[100:106]) and torch.equal(e2, e[100:106])


def test_config5_shard_size_768_properties(gpu_device):
    """One GPU's shard of BASELINE.json configs[4] (1.25M x 768): planted needles, order,
    agreement with the exhaustive kernel, and global ids through id_base."""
    import torch
    n, dim, B = 1_250_000, 768, 64
    gen = torch.Generator(device=gpu_device).manual_seed(99)
    c = torch.randn((n, dim), generator=gen, device=gpu_device, dtype=torch.float32)
    c = (c / c.norm(dim=1, keepdim=True)).half()
    q = torch.randn((B, dim), generator=gen, device=gpu_device, dtype=torch.float32)
    q = (q / q.norm(dim=1, keepdim=True)).half()
    plant = torch.arange(B, device=gpu_device) * 19_001 + 3
    c[plant] = q
    from rag_fin_amd.store import GpuIndex
    ix = GpuIndex(dim, n, gpu_device)
    ix.add(c)
    base = 3 * n                                   # rank 3 of a row-sharded job
    s, i, e, f = ix.search_raw(q, 10, id_base=base, want_exact=True)
    torch.cuda.synchronize()
    assert int(f.abs().sum()) == 0
    assert torch.equal(i[:, 0], plant + base)
    assert bool((e[:, 1:] <= e[:, :-1]).all())
    s2, i2, e2 = ix.search_exhaustive(q[:4].contiguous(), 10, id_base=base, want_exact=True)
    assert torch.equal(i2, i[:4]) and torch.equal(e2, e[:4])


@pytest.mark.parametrize("n,b,k", [(40_000, 256, 10), (25_000, 300, 10), (3_000, 100, 10), (70_000, 65, 64),
                                   (8_193, 129, 5), (20_000, 200, 10)])   # 20 000 rows: an ODD number of 32-row blocks
def test_wide_sweep_parity(gpu_device, n, b, k):
    """Batches above 64 queries at dim 384 take the wide sweep (scan_wide.hip: up to
    256 queries per corpus pass); same oracle, same bit-exact bar."""
    c = osearch.synth_unit_rows(n, 384, 77)
    q = osearch.synth_unit_rows(b, 384, 78)
    c[1234 % n] = q[b - 1]                       # a planted exact match for the last query
    ix = make_index(c, gpu_device)
    check_against_oracle(ix, q, c, k, gpu_device)


def test_wide_sweep_duplicate_heavy_corpus_overflows_and_stays_exact(gpu_device):
    """Every row ties for every query of a wide sweep: the per-wave staging of the wide
    kernel and the candidate lists overflow, every query is flagged, and the exhaustive
    path returns rows 0..k-1 by the tie rule.  A second corpus has ONE hot duplicate
    group (500 copies of the best match of query 0): that query is flagged (and at most a
    couple of others whose top-10 cut the tie group happens to straddle)."""
    import torch
    row = osearch.synth_unit_rows(1, 384, 15)
    c = np.repeat(row, 30_000, axis=0)
    q = osearch.synth_unit_rows(130, 384, 16)
    ix = make_index(c, gpu_device)
    _, _, _, flags = ix.search_raw(torch.from_numpy(q).to(gpu_device), 10)
    assert (flags.cpu().numpy() != 0).all()
    check_against_oracle(ix, q, c, 10, gpu_device, raw=False)

    c2 = osearch.synth_unit_rows(60_000, 384, 17)
    q2 = osearch.synth_unit_rows(200, 384, 18)
    c2[1000:1500] = q2[0]
    ix2 = make_index(c2, gpu_device)
    _, _, _, f2 = ix2.search_raw(torch.from_numpy(q2).to(gpu_device), 10)
    f2 = f2.cpu().numpy()
    assert f2[0] != 0 and int((f2 != 0).sum()) <= 3   # a 500-way tie may straddle another query's cut too
    check_against_oracle(ix2, q2, c2, 10, gpu_device, raw=False)


def test_wide_sweep_equals_narrow_sweeps(gpu_device, monkeypatch):
    """The same 200 queries answered by one wide sweep and by 64-query sweeps."""
    import torch
    c = osearch.synth_unit_rows(120_000, 384, 5)
    q = torch.from_numpy(osearch.synth_unit_rows(200, 384, 6)).to(gpu_device)
    ix = make_index(c, gpu_device)
    s1, i1, e1, f1 = ix.search_raw(q, 10, want_exact=True)
    parts = [ix.search_raw(q[a:a + 64].contiguous(), 10, want_exact=True) for a in range(0, 200, 64)]
    torch.cuda.synchronize()
    assert int(f1.abs().sum()) == 0
    assert torch.equal(i1, torch.cat([p[1] for p in parts])) and torch.equal(e1, torch.cat([p[2] for p in parts]))


def test_wide_sweep_matches_the_oracle_and_repeats_bitwise(gpu_device):
    """The batch-256 sweep against the C oracle, then a short soak: the kernel waits on HAND-
    COUNTED vmcnt / lgkmcnt values and supplies MFMA -> VALU wait states by hand, and a
    misplaced count shows up as a RARE wrong tile, not as a failing unit test -- so the sweep
    is repeated on fresh queries and every result compared bitwise with a second run and with
    four 64-query sweeps of the other kernel (tools/soak.py is the long form of this)."""
    import torch
    c = osearch.synth_unit_rows(150_000, 384, 51)
    q16 = osearch.synth_unit_rows(256, 384, 52)
    ix = make_index(c, gpu_device)
    os_, oi = c_oracle.search(q16, c, 10)
    q = torch.from_numpy(q16).to(gpu_device)
    for _ in range(2):
        s, i, e, f = ix.search_raw(q, 10, want_exact=True)
    torch.cuda.synchronize()
    assert int(f.abs().sum()) == 0
    assert np.array_equal(i.cpu().numpy(), oi) and np.array_equal(e.cpu().numpy(), os_)
    gen = torch.Generator(device=gpu_device).manual_seed(7)
    for rnd in range(150):
        qr = torch.nn.functional.normalize(torch.randn((256, 384), device=gpu_device, generator=gen), dim=1).half()
        _, i1, e1, f1 = ix.search_raw(qr, 10, want_exact=True)
        i1, e1 = i1.clone(), e1.clone()
        _, i2, e2, _ = ix.search_raw(qr, 10, want_exact=True)
        assert int(f1.abs().sum()) == 0
        assert torch.equal(i1, i2) and torch.equal(e1, e2), rnd
        if rnd % 10 == 0:
            parts = [ix.search_raw(qr[a:a + 64].contiguous(), 10, want_exact=True) for a in range(0, 256, 64)]
            assert torch.equal(i1, torch.cat([p[1] for p in parts])), rnd
            assert torch.equal(e1, torch.cat([p[2] for p in parts])), rnd


def test_search_does_not_depend_on_workspace_contents(gpu_device):
    """A workspace full of garbage (never zeroed, or left dirty by an aborted search) gives the
    same exact answer: every search zeroes its own counters (k_threshold)."""
    import torch
    c = osearch.synth_unit_rows(40_000, 384, 61)
    q16 = osearch.synth_unit_rows(64, 384, 62)
    ix = make_index(c, gpu_device)
    ws = torch.randint(0, 255, (ix.workspace_bytes,), dtype=torch.uint8, device=gpu_device)
    q = torch.from_numpy(q16).to(gpu_device)
    os_, oi = c_oracle.search(q16, c, 10)
    for _ in range(2):
        s, i, e, f = ix.search_raw(q, 10, want_exact=True, workspace=ws)
        torch.cuda.synchronize()
        assert int(f.abs().sum()) == 0
        assert np.array_equal(i.cpu().numpy(), oi) and np.array_equal(e.cpu().numpy(), os_)
        ws[:65536].random_(0, 255)     # scribble over the control arrays between searches


@pytest.mark.parametrize("n,k", [(5_000, 1000), (30_000, 200), (100, 1000)])
def test_limits_above_64_are_paged_exactly(gpu_device, n, k):
    """graph_cons.hybrid_query_simple searches with limit=1000 (graph_cons.py:275-281):
    pages of 64 ranked strictly after the previous page's last hit must reproduce the
    oracle's full ranking, across exact-duplicate ties that straddle page boundaries."""
    import torch
    c = osearch.synth_unit_rows(n, 384, 91)
    c[60:70] = c[5]                              # a tie group inside / across the first page edge
    q16 = osearch.synth_unit_rows(3, 384, 92)
    q16[0] = c[5]
    ix = make_index(c, gpu_device)
    scores, ids = ix.search_large(torch.from_numpy(q16).to(gpu_device), k)
    os_, oi = c_oracle.search(q16, c, k)
    assert np.array_equal(ids.cpu().numpy(), oi)
    assert np.array_equal(scores.cpu().numpy(), os_.astype(np.float32))


def test_search_host_downloads_once_and_resolves_flagged_queries(gpu_device):
    """GpuIndex.search_host (the serving path's one-synchronisation download) on a corpus where
    every row ties: all queries are flagged and must come back through the exhaustive kernel;
    then on a plain corpus, twice with the same shape (cached pinned buffers)."""
    import torch
    row = osearch.synth_unit_rows(1, 384, 5)
    c = np.repeat(row, 20_000, axis=0)
    q16 = osearch.synth_unit_rows(3, 384, 6)
    ix = make_index(c, gpu_device)
    s, i = ix.search_host(torch.from_numpy(q16).to(gpu_device), 10)
    os_, oi = c_oracle.search(q16, c, 10)
    assert np.array_equal(i, oi) and np.array_equal(s, os_.astype(np.float32))
    c2 = osearch.synth_unit_rows(30_000, 384, 7)
    ix2 = make_index(c2, gpu_device)
    for seed in (8, 9):
        q2 = osearch.synth_unit_rows(5, 384, seed)
        s2, i2 = ix2.search_host(torch.from_numpy(q2).to(gpu_device), 7)
        os2, oi2 = c_oracle.search(q2, c2, 7)
        assert np.array_equal(i2, oi2) and np.array_equal(s2, os2.astype(np.float32))

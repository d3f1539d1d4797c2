"""Cross-shard merge on the GPU: three shards held by three GpuIndex objects on one
card stand in for three ranks; rf_merge_shards over their stacked per-shard top-k
must equal the single-index search (and the oracle) bit for bit."""
import numpy as np
import pytest

from oracle import c_oracle, search as osearch

pytestmark = pytest.mark.gpu


def test_merge_shards_equals_single_index(gpu_device):
    import torch
    from rag_fin_amd.sharded import HipShardBackend, ShardedSearcher
    from rag_fin_amd.store import GpuIndex
    n, d, b, k = 50_000, 384, 64, 10
    base = osearch.synth_unit_rows(n, d, 1234)
    base[20_000:20_050] = base[100:150]           # exact duplicates across shard boundaries
    q16 = osearch.synth_unit_rows(b, d, 5678)
    q = torch.from_numpy(q16).to(gpu_device)
    bounds = [ShardedSearcher.shard_bounds(n, 3, r) for r in range(3)]
    exact_all, ids_all = [], []
    for lo, hi in bounds:
        ix = GpuIndex(d, hi - lo, gpu_device)
        ix.add(torch.from_numpy(base[lo:hi]).to(gpu_device))
        e, i, f = HipShardBackend(ix).local_topk(q, k, lo)
        assert int(f.abs().sum()) == 0
        exact_all.append(e)
        ids_all.append(i)
    full = GpuIndex(d, n, gpu_device)
    full.add(torch.from_numpy(base).to(gpu_device))
    backend = HipShardBackend(full)
    scores, gids = backend.merge(torch.stack(exact_all).contiguous(), torch.stack(ids_all).contiguous(), k)
    s1, i1, e1, _ = full.search_raw(q, k, want_exact=True)
    assert torch.equal(gids, i1) and torch.equal(scores, s1)
    os_, oi = c_oracle.search(q16, base, k)
    assert np.array_equal(gids.cpu().numpy(), oi)
    # world_size 1 degenerate path of the searcher
    s2, i2, _ = ShardedSearcher(backend, row_base=0).search(q, k)
    assert torch.equal(i2, i1) and torch.equal(s2, s1)


def test_merge_shards_with_short_shards(gpu_device):
    """Shards holding fewer than k rows pad with (-inf, -1); the merge skips them."""
    import torch
    from rag_fin_amd.sharded import HipShardBackend
    from rag_fin_amd.store import GpuIndex
    d, k = 128, 10
    c = osearch.synth_unit_rows(23, d, 3)
    q16 = osearch.synth_unit_rows(4, d, 4)
    q = torch.from_numpy(q16).to(gpu_device)
    parts = [(0, 3), (3, 3), (3, 23)]              # 3 rows, empty, 20 rows
    ex, ids, backend = [], [], None
    for lo, hi in parts:
        ix = GpuIndex(d, max(hi - lo, 1), gpu_device)
        if hi > lo:
            ix.add(torch.from_numpy(c[lo:hi]).to(gpu_device))
        backend = HipShardBackend(ix)
        e, i, _ = backend.local_topk(q, k, lo)
        ex.append(e)
        ids.append(i)
    scores, gids = backend.merge(torch.stack(ex).contiguous(), torch.stack(ids).contiguous(), k)
    os_, oi = c_oracle.search(q16, c, k)
    assert np.array_equal(gids.cpu().numpy(), oi)
    assert np.array_equal(scores.cpu().numpy(), os_.astype(np.float32))


def test_search_group_equals_single_searches(gpu_device):
    """Several batches in flight whose per-shard top-k share one collective
    (ShardedSearcher.search_group, rf_merge_shards_group) answer exactly like one search per
    batch -- world-size-1 form of the path bench.py takes at N > 1."""
    import torch
    from rag_fin_amd.sharded import HipShardBackend, ShardedSearcher
    from rag_fin_amd.store import GpuIndex
    n, d, b, k, L = 60_000, 384, 64, 10, 3
    c = osearch.synth_unit_rows(n, d, 21)
    ix = GpuIndex(d, n, gpu_device)
    ix.add(torch.from_numpy(c).to(gpu_device))
    qs = [torch.from_numpy(osearch.synth_unit_rows(b, d, 30 + i)).to(gpu_device) for i in range(L)]
    searcher = ShardedSearcher(HipShardBackend(ix), row_base=1000)
    wss = [ix.workspace] + [ix.new_workspace() for _ in range(L - 1)]
    streams = [torch.cuda.Stream(device=gpu_device) for _ in range(L)]
    for _ in range(2):      # second round reuses the group buffers
        scores, ids, flags = searcher.search_group(qs, k, wss, streams)
        torch.cuda.synchronize()
        assert int(flags.abs().sum()) == 0
        for i in range(L):
            s1, i1, e1, _ = ix.search_raw(qs[i], k, id_base=1000, want_exact=True)
            assert torch.equal(ids[i], i1) and torch.equal(scores[i], s1)
            os_, oi = c_oracle.search(qs[i].cpu().numpy(), c, k)
            assert np.array_equal(ids[i].cpu().numpy() - 1000, oi)


def test_search_on_bare_enqueues_equal_search(gpu_device):
    """ShardedSearcher.search_on (the bench's N > 1 step: three ctypes enqueues on an explicit
    stream with cached pointers) returns what search() returns, on side streams and repeatedly."""
    import torch
    from rag_fin_amd.sharded import HipShardBackend, ShardedSearcher
    from rag_fin_amd.store import GpuIndex
    n, d, b, k = 40_000, 384, 64, 10
    c = osearch.synth_unit_rows(n, d, 51)
    ix = GpuIndex(d, n, gpu_device)
    ix.add(torch.from_numpy(c).to(gpu_device))
    searcher = ShardedSearcher(HipShardBackend(ix), row_base=7)
    streams = [torch.cuda.current_stream(), torch.cuda.Stream(device=gpu_device)]
    wss = [ix.workspace, ix.new_workspace()]
    torch.cuda.synchronize()
    for rnd in range(3):
        q16 = osearch.synth_unit_rows(b, d, 60 + rnd)
        q = torch.from_numpy(q16).to(gpu_device)
        torch.cuda.synchronize()
        outs = [searcher.search_on(q, k, wss[i], streams[i]) for i in range(2)]
        torch.cuda.synchronize()
        os_, oi = c_oracle.search(q16, c, k)
        for s_, i_, f_ in outs:
            assert int(f_.abs().sum()) == 0
            assert np.array_equal(i_.cpu().numpy() - 7, oi)
            assert np.array_equal(s_.cpu().numpy(), os_.astype(np.float32))

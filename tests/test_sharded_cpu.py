"""N > 1 path on CPU: world_size-2 (and 3) gloo runs of rag_fin_amd.sharded.
ShardedSearcher with a CPU backend injected by the test (oracle-backed), checking
that shard bounds + one all-gather + merge reproduce the single-process result bit
for bit.  The product backend (HipShardBackend) is exercised on the GPU box."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import search as osearch
from rag_fin_amd.sharded import ShardedSearcher


class OracleBackend:
    """Stands in for HipShardBackend: same contract, CPU arithmetic from oracle/."""

    def __init__(self, c16_local):
        self.c = c16_local

    def local_topk(self, q16, k, row_base):
        s, i = osearch.search(q16.numpy(), self.c, k, id_base=row_base)
        return torch.from_numpy(s), torch.from_numpy(i), torch.zeros(q16.shape[0], dtype=torch.int32)

    def merge(self, exact_all, ids_all, k):
        s, i = osearch.merge_shards(exact_all.numpy(), ids_all.numpy(), k)
        return torch.from_numpy(s.astype(np.float32)), torch.from_numpy(i)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n, d, b, k, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        c = osearch.synth_unit_rows(n, d, 1234)
        q = torch.from_numpy(osearch.synth_unit_rows(b, d, 5678))
        lo, hi = ShardedSearcher.shard_bounds(n, world, rank)
        searcher = ShardedSearcher(OracleBackend(c[lo:hi]), row_base=lo)
        scores, ids, flags = searcher.search(q, k)
        np.savez(os.path.join(out_dir, f"r{rank}.npz"), scores=scores.numpy(), ids=ids.numpy(),
                 lo=lo, hi=hi)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n", [(2, 3001), (3, 100), (2, 7)])
def test_sharded_equals_single(tmp_path, world, n):
    d, b, k = 64, 5, 10
    mp.spawn(_worker, args=(world, _free_port(), n, d, b, k, str(tmp_path)), nprocs=world, join=True)
    c = osearch.synth_unit_rows(n, d, 1234)
    q = osearch.synth_unit_rows(b, d, 5678)
    ws, wi = osearch.search(q, c, k)
    covered = []
    for r in range(world):
        z = np.load(tmp_path / f"r{r}.npz")
        assert np.array_equal(z["ids"], wi), f"rank {r}"
        assert np.array_equal(z["scores"], ws.astype(np.float32))
        covered.append((int(z["lo"]), int(z["hi"])))
    assert covered[0][0] == 0 and covered[-1][1] == n
    assert all(a[1] == b_[0] for a, b_ in zip(covered, covered[1:]))


def test_shard_bounds_cover_everything():
    for n in (0, 1, 7, 8, 1_000_003):
        for w in (1, 2, 3, 8):
            spans = [ShardedSearcher.shard_bounds(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1

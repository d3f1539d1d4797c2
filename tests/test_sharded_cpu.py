"""N > 1 path on CPU: world_size-2 (and 3) gloo runs of rag_fin_amd.sharded /
rag_fin_amd.sharded_store with CPU doubles injected by the test (oracle-backed index and
backend), checking that
  * shard bounds + one all-gather + merge reproduce the single-process result bit for bit;
  * a query flagged on ONE rank only is re-run exhaustively on every rank before the answer is
    trusted, and the flags every rank returns are the global OR;
  * ShardedCorpusStore (multi-batch ingest, non-contiguous shards, replicated columns, save /
    load across world sizes, rank-0-led serving) answers like a single store.
The product backend (HipShardBackend) is exercised on the GPU box (tests/test_sharded_gpu.py)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import search as osearch
from rag_fin_amd.sharded import ShardedSearcher


class OracleIndex:
    """CPU double of rag_fin_amd.store.GpuIndex: fp16 rows in a numpy array."""

    def __init__(self, dim, capacity, device=None):
        self.dim, self.capacity = dim, int(capacity)
        self.device = torch.device("cpu")
        self.rows = np.zeros((0, dim), dtype=np.float16)

    @property
    def size(self):
        return self.rows.shape[0]

    def add(self, rows):
        assert self.size + rows.shape[0] <= self.capacity
        self.rows = np.concatenate([self.rows, rows.numpy().astype(np.float16)])

    def reset(self):
        self.rows = self.rows[:0]

    def get_rows(self, ids):
        return torch.from_numpy(self.rows[np.asarray(ids, dtype=np.int64)])

    def to_fp16(self, x, normalize=True):
        x = np.asarray(x, dtype=np.float32)
        return torch.from_numpy((osearch.l2_normalize_f32(x) if normalize else x).astype(np.float16))


class OracleBackend:
    """Stands in for HipShardBackend: same contract, CPU arithmetic from oracle/.
    flag_queries: queries this rank reports as unproven -- and answers WRONGLY (an empty list),
    so a merge that trusted them would be visibly wrong."""

    def __init__(self, index, flag_queries=()):
        self.index = index
        self.flag_queries = list(flag_queries)
        self.exhaustive_calls = 0

    def local_topk(self, q16, k, row_base, workspace=None):
        s, i = osearch.search(q16.numpy(), self.index.rows, k, id_base=row_base)
        flags = np.zeros(q16.shape[0], dtype=np.int32)
        for b in self.flag_queries:
            flags[b] = 1
            s[b], i[b] = -np.inf, -1
        return torch.from_numpy(s), torch.from_numpy(i), torch.from_numpy(flags)

    def local_exhaustive(self, q16, k, row_base):
        self.exhaustive_calls += 1
        s, i = osearch.search(q16.numpy(), self.index.rows, k, id_base=row_base)
        return torch.from_numpy(s), torch.from_numpy(i)

    def merge(self, exact_all, ids_all, k):
        s, i = osearch.merge_shards(exact_all.numpy(), ids_all.numpy(), k)
        return torch.from_numpy(s.astype(np.float32)), torch.from_numpy(i)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _init(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)


def _worker(rank, world, port, n, d, b, k, out_dir, flag_rank, flag_queries):
    _init(rank, world, port)
    try:
        c = osearch.synth_unit_rows(n, d, 1234)
        q = torch.from_numpy(osearch.synth_unit_rows(b, d, 5678))
        lo, hi = ShardedSearcher.shard_bounds(n, world, rank)
        ix = OracleIndex(d, max(hi - lo, 1))
        ix.add(torch.from_numpy(c[lo:hi]))
        backend = OracleBackend(ix, flag_queries if rank == flag_rank else ())
        searcher = ShardedSearcher(backend, row_base=lo)
        scores, ids, flags = searcher.search(q, k)
        raw = searcher.search(q, k, resolve=False)            # what an unresolved merge would hold
        np.savez(os.path.join(out_dir, f"r{rank}.npz"), scores=scores.numpy(), ids=ids.numpy(), flags=flags.numpy(),
                 raw_ids=raw[1].numpy(), ex_calls=backend.exhaustive_calls, lo=lo, hi=hi)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n", [(2, 3001), (3, 100), (2, 7)])
def test_sharded_equals_single(tmp_path, world, n):
    d, b, k = 64, 5, 10
    mp.spawn(_worker, args=(world, _free_port(), n, d, b, k, str(tmp_path), -1, ()), nprocs=world, join=True)
    c = osearch.synth_unit_rows(n, d, 1234)
    q = osearch.synth_unit_rows(b, d, 5678)
    ws, wi = osearch.search(q, c, k)
    covered = []
    for r in range(world):
        z = np.load(tmp_path / f"r{r}.npz")
        assert np.array_equal(z["ids"], wi), f"rank {r}"
        assert np.array_equal(z["scores"], ws.astype(np.float32))
        assert not z["flags"].any() and int(z["ex_calls"]) == 0
        covered.append((int(z["lo"]), int(z["hi"])))
    assert covered[0][0] == 0 and covered[-1][1] == n
    assert all(a[1] == b_[0] for a, b_ in zip(covered, covered[1:]))


@pytest.mark.parametrize("world,flag_rank", [(2, 1), (3, 0)])
def test_query_flagged_on_one_rank_is_resolved_on_all(tmp_path, world, flag_rank):
    """Only `flag_rank`'s local scan flags queries 1 and 3 (and returns garbage for them).  Every
    rank must see the flags, every rank must re-run those two queries exhaustively, and the
    final answer must equal the single-process one; the unresolved merge is visibly wrong."""
    n, d, b, k = 2000, 64, 5, 10
    mp.spawn(_worker, args=(world, _free_port(), n, d, b, k, str(tmp_path), flag_rank, (1, 3)), nprocs=world,
             join=True)
    ws, wi = osearch.search(osearch.synth_unit_rows(b, d, 5678), osearch.synth_unit_rows(n, d, 1234), k)
    for r in range(world):
        z = np.load(tmp_path / f"r{r}.npz")
        assert z["flags"].tolist() == [0, 1, 0, 1, 0], f"rank {r}: flags must be the global OR"
        assert int(z["ex_calls"]) == 1, f"rank {r} did not join the exhaustive re-run"
        assert np.array_equal(z["ids"], wi) and np.array_equal(z["scores"], ws.astype(np.float32))
        assert not np.array_equal(z["raw_ids"][1], wi[1])      # the flagged rank's rows were missing before
        assert np.array_equal(z["raw_ids"][0], wi[0])


def test_shard_bounds_cover_everything():
    for n in (0, 1, 7, 8, 1_000_003):
        for w in (1, 2, 3, 8):
            spans = [ShardedSearcher.shard_bounds(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


# ---- ShardedCorpusStore ---------------------------------------------------------------------------
def _store_worker(rank, world, port, out_dir):
    from rag_fin_amd.sharded_store import ShardedCorpusStore
    _init(rank, world, port)
    try:
        d = 64
        vec = osearch.synth_unit_rows(257, d, 77).astype(np.float32)
        batches = [(0, 100), (100, 101), (101, 257)]          # three inserts: shards are NOT contiguous globally
        ix = OracleIndex(d, 8)
        st = ShardedCorpusStore("t", dim=d, capacity=8, index=ix, backend=None)
        st._backend_factory = lambda index: OracleBackend(index)
        for lo, hi in batches:
            n = hi - lo
            st.insert([[f"k{i}" for i in range(lo, hi)], [f"text {i}" for i in range(lo, hi)], vec[lo:hi],
                       ["p"] * n, ["c"] * n, ["s"] * n, [float(i) for i in range(lo, hi)]])
        st.flush()
        assert st.num_entities == 257 and 0 < st.local_rows < 257
        q = osearch.synth_unit_rows(6, d, 78).astype(np.float32)
        scores, rows = st.search_rows(q, 10)
        hits = st.search(q[:2], "embedding", {"metric_type": "COSINE"}, 3, output_fields=["id", "text", "primary_value"])
        big_s, big_r = st.search_rows(q[:2], 100)             # limit > RF_MAX_K: per-shard large top-k, merged
        got = st.query(expr='id in ["k5", "k150", "nope", "k256"]', output_fields=["id", "text", "embedding"])
        st.save(os.path.join(out_dir, "corpus"))
        # serving: rank 0 leads, the others follow until released
        st.start_workers()
        led = None
        if rank == 0:
            led = [st.search_rows(q[i:i + 1], 5) for i in range(3)]
            st.stop_workers()
        # reload under the same world: contiguous shards this time, same answers
        st2 = ShardedCorpusStore.load_from(os.path.join(out_dir, "corpus"), index_factory=OracleIndex,
                                           backend=None)
        st2._backend_factory = lambda index: OracleBackend(index)
        s2, r2 = st2.search_rows(q, 10)
        np.savez(os.path.join(out_dir, f"s{rank}.npz"), scores=scores, rows=rows, s2=s2, r2=r2, big_r=big_r,
                 hit_ids=np.array([[h.id for h in hh] for hh in hits]),
                 hit_pv=np.array([[h.entity.primary_value for h in hh] for hh in hits]),
                 q_ids=np.array([g["id"] for g in got]), q_emb=np.array([g["embedding"] for g in got]),
                 led=np.array([l[1][0] for l in led]) if led is not None else np.zeros(0),
                 local_rows=st.local_rows)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_corpus_store_equals_single_store(tmp_path, world):
    mp.spawn(_store_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    d = 64
    vec = osearch.synth_unit_rows(257, d, 77).astype(np.float32)
    c16 = osearch.l2_normalize_f32(vec).astype(np.float16)
    q16 = osearch.l2_normalize_f32(osearch.synth_unit_rows(6, d, 78).astype(np.float32)).astype(np.float16)
    ws, wi = osearch.search(q16, c16, 10)
    wbs, wbi = osearch.search(q16[:2], c16, 100)
    total = 0
    for r in range(world):
        z = np.load(tmp_path / f"s{r}.npz")
        assert np.array_equal(z["rows"], wi) and np.array_equal(z["scores"], ws.astype(np.float32)), f"rank {r}"
        assert np.array_equal(z["r2"], wi) and np.array_equal(z["s2"], ws.astype(np.float32))
        assert np.array_equal(z["big_r"], wbi)
        assert z["hit_ids"].tolist() == [[f"k{i}" for i in wi[b][:3]] for b in range(2)]
        assert z["hit_pv"].tolist() == [[float(i) for i in wi[b][:3]] for b in range(2)]
        assert z["q_ids"].tolist() == ["k5", "k150", "k256"]
        assert np.allclose(z["q_emb"], c16[[5, 150, 256]].astype(np.float32))
        total += int(z["local_rows"])
        if r == 0:
            assert np.array_equal(z["led"], wi[:3, :5])
    assert total == 257
    # the saved corpus is the single-GPU format: vectors in GLOBAL row order
    mm = np.fromfile(tmp_path / "corpus" / "vectors.f16", dtype=np.float16).reshape(257, d)
    assert np.array_equal(mm.view(np.uint16), c16.view(np.uint16))


# ---- serving: several threads on the leading rank, and a shard that fails -------------------------------
class FlakyBackend(OracleBackend):
    """Raises inside the local scan for the searches listed in `fail_calls` (0-based call numbers)."""

    def __init__(self, index, fail_calls=()):
        super().__init__(index)
        self.fail_calls, self.calls = set(fail_calls), 0

    def local_topk(self, q16, k, row_base, workspace=None):
        n = self.calls
        self.calls += 1
        if n in self.fail_calls:
            raise RuntimeError("injected scan failure")
        return super().local_topk(q16, k, row_base, workspace)


def _serving_worker(rank, world, port, out_dir, fail_calls):
    import threading
    from rag_fin_amd.sharded_store import ShardedCorpusStore
    _init(rank, world, port)
    try:
        d = 32
        vec = osearch.synth_unit_rows(301, d, 5).astype(np.float32)
        st = ShardedCorpusStore("t", dim=d, capacity=8, index=OracleIndex(d, 8), backend=None)
        st._backend_factory = lambda index: FlakyBackend(index, fail_calls if rank == 1 else ())
        n = vec.shape[0]
        st.insert([[f"k{i}" for i in range(n)], ["t"] * n, vec, ["p"] * n, ["c"] * n, ["s"] * n, [0.0] * n])
        st.flush()
        st.start_workers()            # ranks > 0 stay inside until released
        if rank != 0:
            return
        q = osearch.synth_unit_rows(24, d, 6).astype(np.float32)
        out, errors = {}, []

        def client(t):
            for j in range(6):
                i = t * 6 + j
                try:
                    out[i] = st.search_rows(q[i:i + 1], 5 + (i % 3))[1][0]     # different k per call: a mispaired header shows
                except RuntimeError as e:
                    errors.append((i, str(e)))
        threads = [threading.Thread(target=client, args=(t,)) for t in range(4)]
        for th in threads:
            th.start()
        for th in threads:
            th.join(timeout=120)
        alive = any(th.is_alive() for th in threads)
        after = st.search_rows(q[:1], 5)[1][0] if not alive else None   # the store still works after a failed search
        st.stop_workers()
        np.savez(os.path.join(out_dir, "serving.npz"), alive=alive, n_err=len(errors),
                 err_msgs=np.array([m for _, m in errors] or [""]), err_idx=np.array([i for i, _ in errors] or [-1]),
                 after=after if after is not None else np.zeros(0), **{f"r{i}": v for i, v in out.items()})
    finally:
        dist.destroy_process_group()


def _serving_expected():
    d = 32
    c16 = osearch.l2_normalize_f32(osearch.synth_unit_rows(301, d, 5).astype(np.float32)).astype(np.float16)
    q16 = osearch.l2_normalize_f32(osearch.synth_unit_rows(24, d, 6).astype(np.float32)).astype(np.float16)
    return c16, q16


def test_threads_on_the_leading_rank_do_not_interleave_collectives(tmp_path):
    """FastMCP calls the tools from a thread pool (vector_rag_mcp/main.py:126,135-146): four threads x six
    searches on rank 0 with rank 1 parked in start_workers().  Header broadcast, query broadcast and the
    all-gather of one call must stay together (ShardedCorpusStore._coll_lock) -- every answer equals the
    single-store one and nothing hangs."""
    mp.spawn(_serving_worker, args=(2, _free_port(), str(tmp_path), ()), nprocs=2, join=True)
    z = np.load(tmp_path / "serving.npz")
    assert not bool(z["alive"]) and int(z["n_err"]) == 0
    c16, q16 = _serving_expected()
    for i in range(24):
        k = 5 + (i % 3)
        _, wi = osearch.search(q16[i:i + 1], c16, k)
        assert np.array_equal(z[f"r{i}"], wi[0]), i


def test_a_failing_shard_is_reported_to_the_caller_and_nobody_hangs(tmp_path):
    """Rank 1's scan raises on its 3rd and 10th search: it still enters the collective (with SHARD_FAILED in
    its flags), rank 0's search_rows raises for exactly those two calls, every other call is exact, the
    worker stays alive and the store answers afterwards."""
    mp.spawn(_serving_worker, args=(2, _free_port(), str(tmp_path), (2, 9)), nprocs=2, join=True)
    z = np.load(tmp_path / "serving.npz")
    assert not bool(z["alive"])
    assert int(z["n_err"]) == 2 and all("a rank could not scan its shard" in m for m in z["err_msgs"])
    c16, q16 = _serving_expected()
    failed = set(int(i) for i in z["err_idx"])
    for i in range(24):
        if i in failed:
            continue
        _, wi = osearch.search(q16[i:i + 1], c16, 5 + (i % 3))
        assert np.array_equal(z[f"r{i}"], wi[0]), i
    _, w0 = osearch.search(q16[:1], c16, 5)
    assert np.array_equal(z["after"], w0[0])


# ---- bench.py --gpus N: the answer-check bookkeeping of the strong and the weak job ------------------------
def _bench_worker(rank, world, port, out_dir):
    import bench
    _init(rank, world, port)
    try:
        n, d, b, k, nq = 1203, 48, 8, 10, 4
        # strong job: rank r holds rows shard_bounds(n) of ONE corpus; weak job: every rank its own corpus
        for label, rows, base, seed in (("strong", None, None, 1234), ("weak", 400, rank * 400, 1234 + rank)):
            if rows is None:
                lo, hi = ShardedSearcher.shard_bounds(n, world, rank)
                c = osearch.synth_unit_rows(n, d, seed)[lo:hi]
                base = lo
            else:
                c = osearch.synth_unit_rows(rows, d, seed)
            q = osearch.synth_unit_rows(b, d, 5678)
            ix = OracleIndex(d, max(c.shape[0], 1))
            ix.add(torch.from_numpy(c))
            res = ShardedSearcher(OracleBackend(ix), row_base=base).search(torch.from_numpy(q), k, resolve=False)
            os_l, oi_l = osearch.search(q[:nq], c, k)
            exp_s, exp_i = bench.gather_expected(os_l, oi_l + base, world, k, torch.device("cpu"))
            fields = bench.compare_global(res[0].numpy(), res[1].numpy(), exp_s, exp_i, k)
            clean = bench.all_ranks_agree(int(res[2].abs().sum()) == 0, world, torch.device("cpu"))
            bad = bench.all_ranks_agree(rank != 1, world, torch.device("cpu"))     # one dissenting rank -> False everywhere
            np.savez(os.path.join(out_dir, f"b_{label}_{rank}.npz"), exact=fields["global_ids_ranks_exact"],
                     recall=fields["global_recall_at_10"], err=fields["global_max_abs_score_err"],
                     nq=fields["global_checked_queries"], clean=clean, bad=bad, exp_i=exp_i)
    finally:
        dist.destroy_process_group()


def test_bench_global_check_bookkeeping_under_gloo(tmp_path):
    """bench.py's N > 1 answer check (gather_expected / compare_global / all_ranks_agree: what the strong-scaled
    headline and the weak job run after their timed regions) at world 2 under gloo with the oracle backend, so
    that the first real multi-GPU run cannot die in bookkeeping."""
    world = 2
    mp.spawn(_bench_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    n, d, b, k, nq = 1203, 48, 8, 10, 4
    q = osearch.synth_unit_rows(b, d, 5678)
    _, wi_strong = osearch.search(q[:nq], osearch.synth_unit_rows(n, d, 1234), k)
    weak_corpus = np.concatenate([osearch.synth_unit_rows(400, d, 1234 + r) for r in range(world)])
    _, wi_weak = osearch.search(q[:nq], weak_corpus, k)
    for label, want in (("strong", wi_strong), ("weak", wi_weak)):
        for r in range(world):
            z = np.load(tmp_path / f"b_{label}_{r}.npz")
            assert bool(z["exact"]) and float(z["recall"]) == 1.0 and float(z["err"]) == 0.0 and int(z["nq"]) == nq
            assert bool(z["clean"]) and not bool(z["bad"])
            assert np.array_equal(z["exp_i"], want), (label, r)

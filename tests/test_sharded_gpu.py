"""Cross-shard merge on the GPU.  (1) Three shards held by three GpuIndex objects on one card
stand in for three ranks; rf_merge_shards over their stacked per-shard top-k must equal the
single-index search (and the oracle) bit for bit.  (2) The PRODUCT lane path
(HipShardBackend + ShardedSearcher._search_lane / ShardedCorpusStore + VectorRAG) with
world_size 2: two processes, both on cuda:0, collectives over gloo (one card cannot host two
RCCL ranks) -- real flags raised by a duplicate-heavy shard on ONE rank must be resolved on
both, and the merged answers must equal the C oracle over the whole corpus."""
import os
import socket

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


# ---- world_size 2 on one card: the product classes end to end ------------------------------------------
def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _corpus(n, d):
    c = osearch.synth_unit_rows(n, d, 301)
    c[n // 2 + 100: n // 2 + 100 + 12_000] = c[n // 2 + 100]     # 12 000 identical rows, all inside rank 1's shard
    return c


def _lane_worker(rank, world, port, out_dir):
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from rag_fin_amd.sharded import HipShardBackend, ShardedSearcher
        from rag_fin_amd.store import GpuIndex
        dev = torch.device("cuda:0")
        n, d, b, k = 60_000, 384, 64, 10
        c = _corpus(n, d)
        q16 = osearch.synth_unit_rows(b, d, 302)
        q16[5] = c[n // 2 + 100]                                  # its top-10 is a 12 000-way tie: rank 1 flags it
        lo, hi = ShardedSearcher.shard_bounds(n, world, rank)
        ix = GpuIndex(d, hi - lo, dev)
        ix.add(torch.from_numpy(c[lo:hi]).to(dev))
        searcher = ShardedSearcher(HipShardBackend(ix), row_base=lo)
        q = torch.from_numpy(q16).to(dev)
        raw = searcher.search(q, k, resolve=False)
        local_flags = searcher._lanes[(0, b, k)]["local_flags"].cpu().numpy().copy()
        gflags_raw = raw[2].cpu().numpy().copy()
        scores, gids, gflags = searcher.search(q, k)              # resolves flagged queries on every rank
        ws2 = ix.new_workspace()                                  # a second lane (own workspace)
        s2, g2, _ = searcher.search(q, k, workspace=ws2)
        torch.cuda.synchronize()
        np.savez(os.path.join(out_dir, f"l{rank}.npz"), scores=scores.cpu().numpy(), ids=gids.cpu().numpy(),
                 gflags=gflags.cpu().numpy(), gflags_raw=gflags_raw, local_flags=local_flags,
                 ids2=g2.cpu().numpy())
    finally:
        dist.destroy_process_group()


def test_world2_product_lane_path_resolves_real_flags(tmp_path, gpu_device):
    import torch.multiprocessing as mp
    mp.spawn(_lane_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    n, d, b, k = 60_000, 384, 64, 10
    c = _corpus(n, d)
    q16 = osearch.synth_unit_rows(b, d, 302)
    q16[5] = c[n // 2 + 100]
    os_, oi = c_oracle.search(q16, c, k)
    z0, z1 = (np.load(tmp_path / f"l{r}.npz") for r in range(2))
    assert z0["local_flags"][5] == 0 and z1["local_flags"][5] != 0        # flagged on rank 1 only ...
    for z in (z0, z1):
        assert z["gflags_raw"][5] != 0 and z["gflags"][5] != 0             # ... but visible on both
        assert np.array_equal(z["ids"], oi)
        assert np.array_equal(z["scores"], os_.astype(np.float32))
        assert np.array_equal(z["ids2"], oi)


def _store_worker(rank, world, port, out_dir):
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import encoder as oenc, synth_text
        from rag_fin_amd.embedder import Embedder
        from rag_fin_amd.rag import VectorRAG
        from rag_fin_amd.service import ingest_sharded
        from rag_fin_amd.sharded_store import ShardedCorpusStore
        from rag_fin_amd.tokenizer import WordPieceTokenizer
        dev = torch.device("cuda:0")
        cfg = dict(oenc.MINILM_L6, layers=2)
        emb = Embedder(oenc.random_weights(cfg, 9), cfg, tokenizer=WordPieceTokenizer(synth_text.vocab_for()),
                       device=dev)
        texts = synth_text.retemplated_texts(2000, 31)
        chunks = [dict(id=f"c{i}", text=t, period=f"P{i % 4}", chunk_type="t", statement_type="s",
                       primary_value=float(i)) for i, t in enumerate(texts)]
        store = ShardedCorpusStore("fin_chunks", dim=384, capacity=16, device=dev)
        assert ingest_sharded(store, emb, chunks[:1200]) == 1200          # two ingests: non-contiguous shards
        assert ingest_sharded(store, emb, chunks[1200:]) == 800
        rag = VectorRAG(None, "fin_chunks", embedder=emb, store=store)
        queries = synth_text.retemplated_texts(6, 32)
        # What each rank EMBEDDED at search time is recorded: the first query-sized forward after a large-batch one has
        # been seen to differ from later ones in the last fp16 bit of some components, so the oracle below
        # is given the bits that were searched with -- rank 0's, which the store sends to every shard in both forms.
        seen_one, seen_many = [], []
        plain_encode = emb.encode_to_device

        def recording_encode(sentences, *a, **kw):
            out = plain_encode(sentences, *a, **kw)
            (seen_one if len(sentences) == 1 else seen_many).append(out.cpu().numpy().copy())
            return out
        emb.encode_to_device = recording_encode
        collective = [rag.search(qt, 5) for qt in queries[:3]]            # every rank calls (collective form)
        store.start_workers()                                             # ranks > 0 stay inside
        led = None
        if rank == 0:
            led = [rag.search(qt, 5) for qt in queries[3:]] + [rag.search_batch(queries, 3)]
            store.stop_workers()
        emb.encode_to_device = plain_encode
        np.save(os.path.join(out_dir, f"seen_q{rank}.npy"), np.concatenate(seen_one))
        if rank == 0:
            np.save(os.path.join(out_dir, "seen_qb.npy"), np.concatenate(seen_many))
        vec_local = store.index.get_rows(np.arange(store.local_rows)).cpu().numpy()
        np.savez(os.path.join(out_dir, f"v{rank}.npz"), vec=vec_local, gmap=store._id_map.cpu().numpy())
        if rank == 0:
            import json
            with open(os.path.join(out_dir, "rag.json"), "w") as f:
                json.dump({"collective": collective, "led": led[:3], "batch": led[3]}, f)
            # rag.search embeds ONE query per call, search_batch all six at once (another GEMM tile shape:
            # the fp16 embedding may differ in the last bit), so the oracle gets each form's own vectors
            np.save(os.path.join(out_dir, "q.npy"), np.concatenate([emb.encode_to_device([qt]).cpu().numpy() for qt in queries]))
            np.save(os.path.join(out_dir, "qb.npy"), emb.encode_to_device(queries).cpu().numpy())
    finally:
        dist.destroy_process_group()


def test_world2_sharded_store_behind_vector_rag(tmp_path, gpu_device):
    """VectorRAG.search on a ShardedCorpusStore (vector_rag_mcp/main.py:48-70 at N = 2): the
    payload must be what a single store over the same vectors returns -- checked against the C
    oracle on the union of the two shards' stored rows, in global row order."""
    import json
    import torch.multiprocessing as mp
    from oracle import synth_text
    mp.spawn(_store_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    texts = synth_text.retemplated_texts(2000, 31)
    c16 = np.zeros((2000, 384), dtype=np.float16)
    seen = 0
    for r in range(2):
        z = np.load(tmp_path / f"v{r}.npz")
        c16[z["gmap"]] = z["vec"]
        seen += len(z["gmap"])
    assert seen == 2000
    got = json.load(open(tmp_path / "rag.json"))
    shard_rows = [np.sort(np.load(tmp_path / f"v{r}.npz")["gmap"]) for r in range(2)]
    s0, s1 = np.load(tmp_path / "seen_q0.npy"), np.load(tmp_path / "seen_q1.npy")
    assert s0.shape == (6, 384) and s1.shape == (3, 384)      # rank 0: three collective + three led; rank 1: three collective
    later = np.load(tmp_path / "q.npy")
    from conftest import record_measurement
    record_measurement("query_embedding_bits_search_time_vs_later",
                       rank0_rows_differing=int((s0.view(np.uint16) != later.view(np.uint16)).any(1).sum()),
                       rank1_rows_differing=int((s1.view(np.uint16) != later[:3].view(np.uint16)).any(1).sum()),
                       max_abs=float(max(np.abs(s0.astype(np.float32) - later.astype(np.float32)).max(),
                                         np.abs(s1.astype(np.float32) - later[:3].astype(np.float32)).max())))
    assert np.abs(s0.astype(np.float32) - later.astype(np.float32)).max() < 2e-4      # last-bit differences at most

    def expected(per_shard_queries):
        """Top-5 over the whole corpus when shard r was searched with per_shard_queries[r] (one fp16 vector each)."""
        cand = []
        for r, qv in enumerate(per_shard_queries):
            sc, ix = c_oracle.search(qv[None, :], c16[shard_rows[r]], 5)
            cand += [(-float(sc[0, j]), int(shard_rows[r][ix[0, j]])) for j in range(ix.shape[1]) if ix[0, j] >= 0]
        cand.sort()
        return [-c[0] for c in cand[:5]], [c[1] for c in cand[:5]]

    answers = got["collective"] + got["led"]
    for b, ctx in enumerate(answers):
        es, ei = expected([s0[b], s0[b]])     # both forms: rank 0's vector goes to every shard (ShardedCorpusStore.search_rows)
        assert [c["rank"] for c in ctx] == [1, 2, 3, 4, 5]
        assert [c["text"] for c in ctx] == [texts[i] for i in ei]
        assert [c["primary_value"] for c in ctx] == [float(i) for i in ei]
        assert np.allclose([c["score"] for c in ctx], es, atol=1e-6)
    os3, oi3 = c_oracle.search(np.load(tmp_path / "seen_qb.npy"), c16, 3)  # search_batch on rank 0 while leading
    for b, ctx in enumerate(got["batch"]):
        assert [int(c["primary_value"]) for c in ctx] == list(oi3[b])
        assert np.allclose([c["score"] for c in ctx], os3[b], atol=1e-6)


def test_id_table_is_applied_on_the_device_in_every_step_form(gpu_device):
    """A shard whose rows are NOT contiguous in the global numbering (ShardedCorpusStore's layout: every
    insert leaves a slice on every rank): rf_search runs with id_base 0 and rf_map_ids turns local row
    numbers into global ids on the stream -- in the lane path (`search`), in the bare-enqueue step
    (`search_on`, which used to ignore the table) and in the raw C call, -1 ("no hit") untouched."""
    import torch
    from ctypes import c_void_p
    from rag_fin_amd import _lib
    from rag_fin_amd.sharded import HipShardBackend, ShardedSearcher
    from rag_fin_amd.store import GpuIndex
    n, d, b, k = 5000, 384, 64, 10
    c = osearch.synth_unit_rows(n, d, 41)
    q16 = osearch.synth_unit_rows(b, d, 42)
    gid = np.random.default_rng(0).permutation(10 * n)[:n].astype(np.int64)      # arbitrary, unique global ids
    ix = GpuIndex(d, n, gpu_device)
    ix.add(torch.from_numpy(c).to(gpu_device))
    q = torch.from_numpy(q16).to(gpu_device)
    table = torch.from_numpy(gid).to(gpu_device)
    searcher = ShardedSearcher(HipShardBackend(ix), row_base=0, id_map=table)
    os_, oi = c_oracle.search(q16, c, k)
    want = gid[oi]
    s1, g1, f1 = searcher.search(q, k)
    assert np.array_equal(g1.cpu().numpy(), want) and int(f1.abs().sum()) == 0
    st = torch.cuda.Stream(device=gpu_device)
    s2, g2, f2 = searcher.search_on(q, k, ix.new_workspace(), st)
    st.synchronize()
    assert np.array_equal(g2.cpu().numpy(), want) and np.array_equal(s2.cpu().numpy(), os_.astype(np.float32))
    # the raw call: -1 stays, a row number past the table becomes -1
    ids = torch.tensor([0, 4999, -1, 5000, 17], dtype=torch.int64, device=gpu_device)
    with torch.cuda.device(gpu_device):
        _lib.check(ix.lib.rf_map_ids(c_void_p(ids.data_ptr()), ids.numel(), c_void_p(table.data_ptr()), n,
                                     _lib.current_stream_ptr()))
    assert ids.cpu().tolist() == [int(gid[0]), int(gid[4999]), -1, -1, int(gid[17])]


def test_native_sharded_step_behind_the_c_abi(gpu_device):
    """rf_comm_init + rf_search_sharded (include/ragfin.h): the whole step -- scan into the packed
    send buffer, id table, ncclAllGather, merge -- as ONE C call with no Python in it.  One rank here (a card
    cannot host two RCCL ranks): the collective is still issued (a rank gathering from itself), so the calls N
    ranks would make are the calls this test makes.  Answers = the C oracle's; ids through id_base and through
    an id table; a duplicate-heavy corpus raises the same flags rf_search raises."""
    import ctypes
    import torch
    from ctypes import byref, c_void_p
    from rag_fin_amd import _lib
    from rag_fin_amd.store import GpuIndex
    n, d, b, k = 30_000, 384, 64, 10
    c = osearch.synth_unit_rows(n, d, 77)
    q16 = osearch.synth_unit_rows(b, d, 78)
    ix = GpuIndex(d, n, gpu_device)
    ix.add(torch.from_numpy(c).to(gpu_device))
    q = torch.from_numpy(q16).to(gpu_device)
    lib = ix.lib
    uid = ctypes.create_string_buffer(128)
    _lib.check(lib.rf_comm_unique_id(uid))
    comm = c_void_p()
    _lib.check(lib.rf_comm_init(0, 1, uid, gpu_device.index or 0, byref(comm)))
    try:
        assert lib.rf_comm_rank(comm) == 0 and lib.rf_comm_world(comm) == 1
        words = int(lib.rf_search_sharded_scratch_words(comm, b, k))
        assert words == 2 * int(lib.rf_packed_shard_words(b, k)) + (b * k + 1) // 2
        scratch = torch.empty(words, dtype=torch.int64, device=gpu_device)
        scores = torch.empty((b, k), dtype=torch.float32, device=gpu_device)
        ids = torch.empty((b, k), dtype=torch.int64, device=gpu_device)
        flags = torch.empty((b,), dtype=torch.int32, device=gpu_device)
        ws = ix.new_workspace()
        st = torch.cuda.Stream(device=gpu_device)
        torch.cuda.synchronize(gpu_device)

        def step(id_base, table):
            with torch.cuda.device(gpu_device):
                _lib.check(lib.rf_search_sharded(
                    ix.handle, comm, c_void_p(q.data_ptr()), b, k, id_base,
                    c_void_p(table.data_ptr()) if table is not None else None, table.numel() if table is not None else 0,
                    c_void_p(scores.data_ptr()), c_void_p(ids.data_ptr()), c_void_p(flags.data_ptr()),
                    c_void_p(ws.data_ptr()), ix.workspace_bytes, c_void_p(scratch.data_ptr()), words,
                    c_void_p(st.cuda_stream)))
            st.synchronize()
            return scores.cpu().numpy().copy(), ids.cpu().numpy().copy(), flags.cpu().numpy().copy()

        os_, oi = c_oracle.search(q16, c, k)
        s, i, f = step(1_000_000, None)
        assert np.array_equal(i, oi + 1_000_000) and np.array_equal(s, os_.astype(np.float32)) and not f.any()
        gid = np.random.default_rng(1).permutation(10 * n)[:n].astype(np.int64)
        table = torch.from_numpy(gid).to(gpu_device)
        s, i, f = step(123, table)                      # id_base is ignored when a table is given
        assert np.array_equal(i, gid[oi]) and np.array_equal(s, os_.astype(np.float32)) and not f.any()
        # argument checks come back as codes, not faults
        assert lib.rf_search_sharded(ix.handle, comm, c_void_p(q.data_ptr()), b, k, 0, None, 0, c_void_p(scores.data_ptr()),
                                     c_void_p(ids.data_ptr()), None, c_void_p(ws.data_ptr()), ix.workspace_bytes,
                                     c_void_p(scratch.data_ptr()), words - 1, c_void_p(st.cuda_stream)) == -3   # RF_ERR_CAPACITY
        assert lib.rf_search_sharded(ix.handle, comm, c_void_p(q.data_ptr()), b, k, 0, c_void_p(table.data_ptr()), 0,
                                     c_void_p(scores.data_ptr()), c_void_p(ids.data_ptr()), None, c_void_p(ws.data_ptr()),
                                     ix.workspace_bytes, c_void_p(scratch.data_ptr()), words, c_void_p(st.cuda_stream)) == -1
        # the same flags as rf_search on data built to overflow the candidate lists: every row the same vector
        dup = np.repeat(c[:1], n, axis=0)
        ix2 = GpuIndex(d, n, gpu_device)
        ix2.add(torch.from_numpy(dup).to(gpu_device))
        _, _, _, f_local = ix2.search_raw(q, k, want_exact=True)
        with torch.cuda.device(gpu_device):
            _lib.check(lib.rf_search_sharded(ix2.handle, comm, c_void_p(q.data_ptr()), b, k, 0, None, 0,
                                             c_void_p(scores.data_ptr()), c_void_p(ids.data_ptr()), c_void_p(flags.data_ptr()),
                                             c_void_p(ws.data_ptr()), ix2.workspace_bytes, c_void_p(scratch.data_ptr()), words,
                                             c_void_p(st.cuda_stream)))
        st.synchronize()
        assert torch.equal(flags, f_local.to(torch.int32))
    finally:
        _lib.check(lib.rf_comm_destroy(comm))

"""Row-sharded search across the GPUs of one node (new in this
build -- the reference is a single process talking to one Milvus server).

One process per GPU.  Rank r holds a disjoint set of corpus rows and scans them for
the (replicated) query batch; the only exchange is ONE all-gather of the per-shard
top-k
  {fp64 ranking score [B, k], int64 global row id [B, k], uint32 flags [B]}
(B=64, k=10: 10.5 KB per rank) over RCCL, followed by a merge by (score desc, id asc)
on every rank.  Because the fp64 ranking scores are bit-reproducible per (query, row)
and ids are global, the merged result is bit-identical to a single-GPU search over the
whole corpus.

Exactness across ranks.  rf_search proves each LOCAL answer exact or flags the query
(candidate / tie buffers overflowed on adversarial data).  The flags travel in the
payload and the merge ORs them, so every rank holds the same GLOBAL flags; `search()`
then re-runs exactly the flagged queries through rf_search_exhaustive on every shard
and merges again (a second, smaller all-gather that all ranks enter, because they all
see the same flags).  The common case -- no flag -- costs nothing extra.

The local scan and the merge go through a backend object; the product backend is
libragfin_hip.so (HipShardBackend).  Tests inject a CPU backend to exercise the
collective logic under gloo.
"""
from __future__ import annotations

from ctypes import c_void_p

from . import _lib


# Flag bit a rank sends in place of an answer when its local scan raised (ShardedSearcher.poison_step):
# the merge ORs the shards' flags, so EVERY rank sees it, nobody enters the resolution collective and
# the leading rank raises instead of returning a partial answer.  (int32 view of 0x80000000.)
SHARD_FAILED = -(1 << 31)


class HipShardBackend:
    """Local scan = rf_search / rf_search_exhaustive, merge = rf_merge_shards* (all HIP)."""

    def __init__(self, index):
        self.index = index
        self.device = index.device
        self.lib = index.lib

    # -- generic contract (also what the CPU test backend implements) ----------------------
    def local_topk(self, q16, k: int, row_base: int, workspace=None):
        scores, ids, exact, flags = self.index.search_raw(q16, k, id_base=row_base, want_exact=True,
                                                          workspace=workspace)
        return exact, ids, flags

    def local_exhaustive(self, q16, k: int, row_base: int):
        scores, ids, exact = self.index.search_exhaustive(q16, k, id_base=row_base, want_exact=True)
        return exact, ids

    def merge(self, exact_all, ids_all, k: int):
        import torch
        W, B, _ = exact_all.shape
        scores = torch.empty((B, k), dtype=torch.float32, device=self.device)
        ids = torch.empty((B, k), dtype=torch.int64, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.rf_merge_shards(c_void_p(exact_all.data_ptr()),
                                                c_void_p(ids_all.data_ptr()), W, B, k,
                                                c_void_p(scores.data_ptr()),
                                                c_void_p(ids.data_ptr()),
                                                _lib.current_stream_ptr()))
        return scores, ids

    # -- low-overhead lane path: the scan writes straight into the all-gather's send buffer
    # and the merge reads the receive buffer in place (no cat / contiguous / empty per step)
    def new_lane(self, B: int, k: int, world: int):
        import torch
        dev = self.device
        words = int(self.lib.rf_packed_shard_words(B, k))
        packed = torch.zeros(words, dtype=torch.int64, device=dev)   # {score bits, ids, flags}
        return dict(
            B=B, k=k, world=world, words=words, packed=packed,
            exact=packed[:B * k].view(torch.float64).view(B, k),
            local_ids=packed[B * k:2 * B * k].view(B, k),
            local_flags=packed[2 * B * k:].view(torch.int32)[:B],
            flat=torch.empty((world, words), dtype=torch.int64, device=dev),     # gathered
            local_scores=torch.empty((B, k), dtype=torch.float32, device=dev),
            flags=torch.zeros((B,), dtype=torch.int32, device=dev),            # OR over the shards
            scores=torch.empty((B, k), dtype=torch.float32, device=dev),
            ids=torch.empty((B, k), dtype=torch.int64, device=dev))

    def local_topk_into(self, q16, k: int, row_base: int, lane, workspace=None):
        self.index.search_raw(q16, k, id_base=row_base, want_exact=True, workspace=workspace,
                              out=(lane["local_scores"], lane["local_ids"], lane["exact"], lane["local_flags"]))

    def merge_packed(self, flat, lane):
        import torch
        with torch.cuda.device(self.device):
            _lib.check(self.lib.rf_merge_shards_packed(
                c_void_p(flat.data_ptr()), lane["world"], lane["B"], lane["k"],
                c_void_p(lane["scores"].data_ptr()), c_void_p(lane["ids"].data_ptr()),
                c_void_p(lane["flags"].data_ptr()), _lib.current_stream_ptr()))
        return lane["scores"], lane["ids"], lane["flags"]


class ShardedSearcher:
    def __init__(self, backend, row_base: int, group=None, id_map=None):
        """row_base: global id of this shard's local row 0 (contiguous shards).  id_map (optional,
        int64 tensor [n_local] on the backend's device): global id of every local row, for stores
        whose shards are not contiguous in the global numbering (ShardedCorpusStore)."""
        import torch.distributed as dist
        self.backend = backend
        self.row_base = int(row_base)
        self.id_map = id_map
        self.group = group
        self.dist = dist
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self._lanes = {}
        # world 1 normally skips the collective; set to run it anyway (overhead measurements)
        self.force_collective = False
        self.rccl = None   # enable_direct_rccl(): ncclAllGather through ctypes instead of torch.distributed

    def enable_direct_rccl(self, device):
        """Own RCCL communicator for the per-step all-gather (rag_fin_amd/rccl.py); every rank of
        the group must call this at the same point.  Returns True when it is in use."""
        from .rccl import RcclComm
        self.rccl = RcclComm(self.rank, self.world, device, self.group)
        return True

    # -- the step --------------------------------------------------------------------------------
    def search_on(self, q16, k: int, workspace, stream):
        """The step as three bare enqueues on `stream` (a torch.cuda.Stream of this rank's device):
        rf_search -> [rf_map_ids when the shard has an id table] -> ncclAllGather -> rf_merge_shards_packed,
        all through ctypes with cached
        pointers -- no torch call on the hot path, no current-stream switch, no host sync.  Needs
        enable_direct_rccl() when world > 1.  q16 must be a contiguous fp16 [B, dim] tensor on the
        device that stays alive until the step has run.  Returns (scores, global ids, GLOBAL
        flags) -- the lane's buffers; flagged queries are NOT resolved here (no host sync):
        callers check `flags` once their stream has drained (resolve_flagged)."""
        B = q16.shape[0]
        key = ("on", workspace.data_ptr() if workspace is not None else 0, B, k, stream.cuda_stream)
        st = self._lanes.get(key)
        if st is None:
            lane = self.backend.new_lane(B, k, self.world)
            # new_lane zero-fills its send buffer on torch's CURRENT stream; the step below runs on `stream`.  Without
            # this (one-time) wait the fill can land after rf_search has written the lane's first answer into it
            # (seen once as a half-zero result in tests/test_sharded_gpu.py::test_search_on_bare_enqueues_equal_search).
            import torch
            torch.cuda.current_stream(lane["packed"].device).synchronize()
            ws = workspace if workspace is not None else self.backend.index.workspace
            st = self._lanes[key] = dict(
                lane=lane, sp=c_void_p(stream.cuda_stream), ws=ws.data_ptr(),
                scores_local=lane["local_scores"].data_ptr(), ids=lane["local_ids"].data_ptr(),
                exact=lane["exact"].data_ptr(), lflags=lane["local_flags"].data_ptr(),
                packed=lane["packed"].data_ptr(), flat=lane["flat"].data_ptr(), words=lane["words"],
                out_s=lane["scores"].data_ptr(), out_i=lane["ids"].data_ptr(), out_f=lane["flags"].data_ptr())
        ix = self.backend.index
        ix.enqueue_search(q16.data_ptr(), B, k, 0 if self.id_map is not None else self.row_base, st["scores_local"],
                          st["ids"], st["exact"], st["lflags"], st["ws"], st["sp"])
        if self.id_map is not None:   # shards not contiguous in the global numbering: local rows -> global ids (HIP)
            rc = self.backend.lib.rf_map_ids(st["ids"], B * k, self.id_map.data_ptr(), self.id_map.numel(), st["sp"])
            if rc:
                _lib.check(rc)
        if self.world == 1 and not self.force_collective:
            src = st["packed"]
        else:
            if self.rccl is None:
                raise RuntimeError("search_on needs enable_direct_rccl() when the job has more than one rank")
            self.rccl.all_gather_i64(st["packed"], st["flat"], st["words"], st["sp"])
            src = st["flat"]
        rc = self.backend.lib.rf_merge_shards_packed(src, self.world, B, k, st["out_s"], st["out_i"], st["out_f"],
                                                     st["sp"])
        if rc:
            _lib.check(rc)
        lane = st["lane"]
        return lane["scores"], lane["ids"], lane["flags"]

    def _all_gather(self, inp, outp):
        """inp int64 [n] -> outp int64 [world * n] (device tensors of the product backend)."""
        if self.rccl is not None:
            self.rccl.all_gather_i64(inp.data_ptr(), outp.data_ptr(), inp.numel(), _lib.current_stream_ptr())
        elif inp.is_cuda and self.dist.get_backend(self.group) == "gloo":
            # rehearsal path (gloo has no device all-gather): stage through the host
            host = outp.cpu()
            self.dist.all_gather_into_tensor(host, inp.cpu(), group=self.group)
            outp.copy_(host)
        else:
            self.dist.all_gather_into_tensor(outp, inp, group=self.group)

    def _search_lane(self, q16, k: int, workspace):
        """Product path (HipShardBackend): per-(workspace, B, k) preallocated buffers; the
        returned tensors are those buffers -- consume them before the same lane searches
        again.  Three enqueues per step: rf_search, all-gather, merge."""
        B = q16.shape[0]
        key = (workspace.data_ptr() if workspace is not None else 0, B, k)
        lane = self._lanes.get(key)
        if lane is None:
            lane = self._lanes[key] = self.backend.new_lane(B, k, self.world)
        self.backend.local_topk_into(q16, k, 0 if self.id_map is not None else self.row_base, lane, workspace)
        if self.id_map is not None:
            self._map_ids_(lane["local_ids"])
        if self.world == 1 and not self.force_collective:
            flat = lane["packed"]
        else:
            flat = lane["flat"]
            self._all_gather(lane["packed"], flat.view(-1))
        return self.backend.merge_packed(flat, lane)

    def _map_ids_(self, ids):
        """local row numbers -> global ids through id_map, in place (-1 stays -1).  On the GPU this is
        one rf_map_ids enqueue on the current stream (no torch math on the N > 1 product path); the
        torch form below serves the CPU test doubles."""
        import torch
        if self.id_map.numel() == 0:
            return ids
        if ids.is_cuda and hasattr(self.backend, "lib") and ids.is_contiguous():
            with torch.cuda.device(ids.device):
                _lib.check(self.backend.lib.rf_map_ids(c_void_p(ids.data_ptr()), ids.numel(),
                                                       c_void_p(self.id_map.data_ptr()), self.id_map.numel(),
                                                       _lib.current_stream_ptr()))
            return ids
        g = self.id_map[ids.clamp(min=0)]
        ids.copy_(torch.where(ids >= 0, g, ids))
        return ids

    @staticmethod
    def shard_bounds(n_total: int, world: int, rank: int):
        """Contiguous row split; the first n_total % world ranks take one extra row."""
        base, extra = divmod(n_total, world)
        start = rank * base + min(rank, extra)
        return start, start + base + (1 if rank < extra else 0)

    def _gather_merge(self, exact, ids, flags, k: int):
        """Generic path (any backend): ONE collective of int64 [B, 2k + 1] per rank
        = {score bits, global ids, flags}, merge, OR of the flags."""
        import torch
        B = exact.shape[0]
        packed = torch.cat([exact.contiguous().view(torch.int64), ids,
                            flags.to(torch.int64).view(B, 1)], dim=1).contiguous()
        if self.world == 1:
            gathered = packed.view(1, B, 2 * k + 1)
        else:
            flat = torch.empty((self.world * B, 2 * k + 1), dtype=torch.int64, device=packed.device)
            self._all_gather(packed.view(-1), flat.view(-1))
            gathered = flat.view(self.world, B, 2 * k + 1)
        exact_all = gathered[:, :, :k].contiguous().view(torch.float64)
        ids_all = gathered[:, :, k:2 * k].contiguous()
        acc = gathered[0, :, 2 * k].clone()
        for w in range(1, gathered.shape[0]):   # OR over the shards
            acc |= gathered[w, :, 2 * k]
        gflags = acc.to(torch.int32)
        scores, gids = self.backend.merge(exact_all, ids_all, k)
        return scores, gids, gflags

    def search(self, q16, k: int, workspace=None, resolve: bool = True):
        """Returns (scores f32 [B,k], global ids i64 [B,k], flags i32 [B]).

        `flags` are GLOBAL: the OR over all shards of the local scans' per-query flags, equal on
        every rank.  With resolve=True (default) the flagged queries -- normally none -- are
        re-run through the exhaustive path on every shard and merged again before returning
        (this reads the flags on the host: one synchronisation); the returned flags still say
        which queries took that path.  resolve=False returns without a host sync (pipelined
        callers: bench.py), leaving resolution to the caller via resolve_flagged().

        Several batches may be in flight: call from different HIP streams with one
        `workspace` each (GpuIndex.new_workspace()); every rank must issue its searches
        in the same order, because the all-gathers share one communicator."""
        if hasattr(self.backend, "local_topk_into"):
            scores, gids, gflags = self._search_lane(q16, k, workspace)
        else:
            base = 0 if self.id_map is not None else self.row_base
            exact, ids, flags = (self.backend.local_topk(q16, k, base) if workspace is None
                                 else self.backend.local_topk(q16, k, base, workspace))
            if self.id_map is not None:
                self._map_ids_(ids)
            scores, gids, gflags = self._gather_merge(exact, ids, flags, k)
        if resolve:
            self.resolve_flagged(q16, k, scores, gids, gflags)
        return scores, gids, gflags

    def poison_step(self, B: int, k: int, device):
        """This rank could not produce its local answer (its scan raised): enter the SAME collective
        the healthy ranks are in with an empty answer flagged SHARD_FAILED, so that nobody hangs and
        every rank learns of the failure from the merged flags."""
        import torch
        if hasattr(self.backend, "local_topk_into"):
            key = (0, B, k)
            lane = self._lanes.get(key)
            if lane is None:
                lane = self._lanes[key] = self.backend.new_lane(B, k, self.world)
            lane["exact"].fill_(float("-inf"))
            lane["local_ids"].fill_(-1)
            lane["local_flags"].fill_(SHARD_FAILED)
            flat = lane["flat"]
            self._all_gather(lane["packed"], flat.view(-1))
            return self.backend.merge_packed(flat, lane)
        exact = torch.full((B, k), float("-inf"), dtype=torch.float64, device=device)
        ids = torch.full((B, k), -1, dtype=torch.int64, device=device)
        flags = torch.full((B,), SHARD_FAILED, dtype=torch.int32, device=device)
        return self._gather_merge(exact, ids, flags, k)

    @staticmethod
    def failed(gflags) -> bool:
        """True when some rank answered with SHARD_FAILED (host sync)."""
        return bool((gflags < 0).any().item())

    def resolve_flagged(self, q16, k: int, scores, gids, gflags) -> int:
        """Re-run the globally flagged queries exhaustively on every shard, merge, patch
        `scores` / `gids` in place.  Collective-safe: gflags is identical on all ranks, so either
        every rank enters the second all-gather or none does.  Returns the number re-run, or -1
        when a rank reported SHARD_FAILED (then nobody resolves anything: the answer is void)."""
        import torch
        if self.failed(gflags):                         # host sync; identical on every rank
            return -1
        bad = torch.nonzero(gflags != 0).flatten()      # host sync
        nb = int(bad.numel())
        if nb == 0:
            return 0
        qb = q16.index_select(0, bad.to(q16.device)).contiguous()
        base = 0 if self.id_map is not None else self.row_base
        exact, ids = self.backend.local_exhaustive(qb, k, base)
        if self.id_map is not None:
            self._map_ids_(ids)
        zero = torch.zeros(nb, dtype=torch.int32, device=exact.device)
        s2, i2, _ = self._gather_merge(exact, ids, zero, k)
        scores[bad] = s2
        gids[bad] = i2
        return nb

"""Row-sharded search across the GPUs of one node (new in this
build -- the reference is a single process talking to one Milvus server).

One process per GPU.  Rank r holds the contiguous corpus rows
[row_base, row_base + n_local) and scans them for the (replicated) query batch;
the only exchange is ONE all-gather of the per-shard top-k
  {fp64 ranking score, int64 global row id}[B, k]      (B=64, k=10: 10 KB/rank)
over RCCL (torch.distributed backend "nccl"), followed by a merge by
(score desc, id asc) on every rank.  Because the fp64 ranking scores are
bit-reproducible per (query, row) and ids are global, the merged result is
bit-identical to a single-GPU search over the concatenated corpus.

The local scan and the merge go through a backend object; the product backend is
libragfin_hip.so (HipShardBackend).  Tests inject a CPU backend to exercise the
collective logic under gloo.
"""
from __future__ import annotations

from ctypes import c_void_p

from . import _lib


class HipShardBackend:
    """Local scan = rf_search, merge = rf_merge_shards (both HIP)."""

    def __init__(self, index):
        self.index = index
        self.device = index.device
        self.lib = index.lib

    def local_topk(self, q16, k: int, row_base: int, workspace=None):
        scores, ids, exact, flags = self.index.search_raw(q16, k, id_base=row_base, want_exact=True,
                                                          workspace=workspace)
        return exact, ids, flags

    # -- low-overhead lane path: the scan writes straight into the all-gather's send buffer
    # and the merge reads the receive buffer in place (no cat / contiguous / empty per step)
    def new_lane(self, B: int, k: int, world: int):
        import torch
        dev = self.device
        return dict(
            B=B, k=k, world=world,
            packed=torch.empty((2, B, k), dtype=torch.int64, device=dev),          # {score bits, ids}
            flat=torch.empty((world, 2, B, k), dtype=torch.int64, device=dev),     # gathered
            local_scores=torch.empty((B, k), dtype=torch.float32, device=dev),
            flags=torch.empty((B,), dtype=torch.int32, device=dev),
            scores=torch.empty((B, k), dtype=torch.float32, device=dev),
            ids=torch.empty((B, k), dtype=torch.int64, device=dev))

    def local_topk_into(self, q16, k: int, row_base: int, lane, workspace=None):
        packed = lane["packed"]
        self.index.search_raw(q16, k, id_base=row_base, want_exact=True, workspace=workspace,
                              out=(lane["local_scores"], packed[1], packed[0].view(self._f64()), lane["flags"]))

    def merge_packed(self, flat, lane):
        with self._device_ctx():
            _lib.check(self.lib.rf_merge_shards_packed(
                c_void_p(flat.data_ptr()), lane["world"], lane["B"], lane["k"],
                c_void_p(lane["scores"].data_ptr()), c_void_p(lane["ids"].data_ptr()),
                _lib.current_stream_ptr()))
        return lane["scores"], lane["ids"]

    @staticmethod
    def _f64():
        import torch
        return torch.float64

    def _device_ctx(self):
        import torch
        return torch.cuda.device(self.device)

    # -- several batches in flight sharing ONE collective -------------------------------------
    def new_group(self, L: int, B: int, k: int, world: int):
        import torch
        dev = self.device
        return dict(
            L=L, B=B, k=k, world=world,
            send=torch.empty((L, 2, B, k), dtype=torch.int64, device=dev),
            recv=torch.empty((world, L, 2, B, k), dtype=torch.int64, device=dev),
            local_scores=torch.empty((L, B, k), dtype=torch.float32, device=dev),
            flags=torch.empty((L, B), dtype=torch.int32, device=dev),
            scores=torch.empty((L, B, k), dtype=torch.float32, device=dev),
            ids=torch.empty((L, B, k), dtype=torch.int64, device=dev))

    def local_topk_group(self, q16, k: int, row_base: int, grp, lane: int, workspace):
        send = grp["send"]
        self.index.search_raw(q16, k, id_base=row_base, want_exact=True, workspace=workspace,
                              out=(grp["local_scores"][lane], send[lane, 1], send[lane, 0].view(self._f64()),
                                   grp["flags"][lane]))

    def merge_group(self, recv, grp):
        with self._device_ctx():
            _lib.check(self.lib.rf_merge_shards_group(
                c_void_p(recv.data_ptr()), grp["world"], grp["L"], grp["B"], grp["k"],
                c_void_p(grp["scores"].data_ptr()), c_void_p(grp["ids"].data_ptr()), _lib.current_stream_ptr()))
        return grp["scores"], grp["ids"]

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


class ShardedSearcher:
    def __init__(self, backend, row_base: int, group=None):
        import torch.distributed as dist
        self.backend = backend
        self.row_base = int(row_base)
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

    def search_on(self, q16, k: int, workspace, stream):
        """The step as three bare enqueues on `stream` (a torch.cuda.Stream of this rank's device):
        rf_search -> ncclAllGather -> rf_merge_shards_packed, all through ctypes with cached
        pointers -- no torch call on the hot path, no current-stream switch.  Needs
        enable_direct_rccl() when world > 1.  q16 must be a contiguous fp16 [B, dim] tensor on the
        device that stays alive until the step has run; same return as search()."""
        B = q16.shape[0]
        key = ("on", workspace.data_ptr() if workspace is not None else 0, B, k, stream.cuda_stream)
        st = self._lanes.get(key)
        if st is None:
            lane = self.backend.new_lane(B, k, self.world)
            ws = workspace if workspace is not None else self.backend.index.workspace
            packed, flat = lane["packed"], lane["flat"]
            st = self._lanes[key] = dict(
                lane=lane, sp=c_void_p(stream.cuda_stream), ws=ws.data_ptr(),
                scores_local=lane["local_scores"].data_ptr(), ids=packed[1].data_ptr(), exact=packed[0].data_ptr(),
                flags=lane["flags"].data_ptr(), packed=packed.data_ptr(), flat=flat.data_ptr(),
                out_s=lane["scores"].data_ptr(), out_i=lane["ids"].data_ptr())
        ix = self.backend.index
        ix.enqueue_search(q16.data_ptr(), B, k, self.row_base, st["scores_local"], st["ids"], st["exact"],
                          st["flags"], st["ws"], st["sp"])
        if self.world == 1 and not self.force_collective:
            src = st["packed"]
        else:
            if self.rccl is None:
                raise RuntimeError("search_on needs enable_direct_rccl() when the job has more than one rank")
            self.rccl.all_gather_i64(st["packed"], st["flat"], 2 * B * k, st["sp"])
            src = st["flat"]
        rc = self.backend.lib.rf_merge_shards_packed(src, self.world, B, k, st["out_s"], st["out_i"], st["sp"])
        if rc:
            _lib.check(rc)
        lane = st["lane"]
        return lane["scores"], lane["ids"], lane["flags"]

    def _search_lane(self, q16, k: int, workspace):
        """Product path (HipShardBackend): per-(workspace, B, k) preallocated buffers; the
        returned tensors are those buffers -- consume them before the same lane searches
        again.  Three enqueues per step: rf_search, all_gather_into_tensor, merge."""
        B = q16.shape[0]
        key = (workspace.data_ptr() if workspace is not None else 0, B, k)
        lane = self._lanes.get(key)
        if lane is None:
            lane = self._lanes[key] = self.backend.new_lane(B, k, self.world)
        self.backend.local_topk_into(q16, k, self.row_base, lane, workspace)
        if self.world == 1 and not self.force_collective:
            flat = lane["packed"]
        else:
            flat = lane["flat"]
            inp, outp = lane["packed"].view(2 * B, k), flat.view(self.world * 2 * B, k)
            if self.rccl is not None:
                self.rccl.all_gather_i64(inp.data_ptr(), outp.data_ptr(), 2 * B * k, _lib.current_stream_ptr())
            elif self.dist.get_backend(self.group) == "gloo":
                # rehearsal path (gloo has no device all-gather): stage through the host
                host = outp.cpu()
                self.dist.all_gather_into_tensor(host, inp.cpu(), group=self.group)
                outp.copy_(host)
            else:
                self.dist.all_gather_into_tensor(outp, inp, group=self.group)
        scores, gids = self.backend.merge_packed(flat, lane)
        return scores, gids, lane["flags"]

    def search_group(self, queries, k: int, workspaces, streams):
        """L batches at once (product backend only): batch i is scanned on streams[i] with
        workspaces[i]; their per-shard top-k travel in ONE all-gather and are merged by one launch
        on the CURRENT stream.  -> (scores f32 [L,B,k], global ids i64 [L,B,k], flags i32 [L,B]);
        the tensors are reused by the next call with the same shape.  Measured on one GPU with a
        one-rank collective this is SLOWER than one all-gather per batch (58 vs 50 us/step at
        125 k-row shards: consecutive groups serialise on the shared send / receive buffers);
        kept for the real multi-GPU runs, where the collective itself is dearer, to try."""
        import torch
        L = len(queries)
        B = queries[0].shape[0]
        key = ("group", L, B, k)
        grp = self._lanes.get(key)
        if grp is None:
            grp = self._lanes[key] = self.backend.new_group(L, B, k, self.world)
        main = torch.cuda.current_stream()
        for i, q in enumerate(queries):
            with torch.cuda.stream(streams[i]):
                self.backend.local_topk_group(q, k, self.row_base, grp, i, workspaces[i])
            main.wait_stream(streams[i])
        if self.world == 1 and not self.force_collective:
            recv = grp["send"]
        else:
            recv = grp["recv"]
            inp, outp = grp["send"].view(L * 2 * B, k), recv.view(self.world * L * 2 * B, k)
            if self.dist.get_backend(self.group) == "gloo":   # rehearsal path: stage through the host
                host = outp.cpu()
                self.dist.all_gather_into_tensor(host, inp.cpu(), group=self.group)
                outp.copy_(host)
            else:
                self.dist.all_gather_into_tensor(outp, inp, group=self.group)
        scores, gids = self.backend.merge_group(recv, grp)
        for st in streams:            # the lanes' next scans overwrite `send`: they follow the collective
            st.wait_stream(main)
        return scores, gids, grp["flags"]

    @staticmethod
    def shard_bounds(n_total: int, world: int, rank: int):
        """Contiguous row split; the first n_total % world ranks take one extra row."""
        base, extra = divmod(n_total, world)
        start = rank * base + min(rank, extra)
        return start, start + base + (1 if rank < extra else 0)

    def search(self, q16, k: int, workspace=None):
        """Returns (scores f32 [B,k], global ids i64 [B,k], flags).  flags are the
        local scan's per-query flags (non-zero -> caller re-runs those queries
        through the exhaustive path before trusting the merge).

        Several batches may be in flight: call from different HIP streams with one
        `workspace` each (GpuIndex.new_workspace()); every rank must issue its searches
        in the same order, because the all-gathers share one communicator."""
        import torch
        if hasattr(self.backend, "local_topk_into"):
            return self._search_lane(q16, k, workspace)
        if workspace is None:
            exact, ids, flags = self.backend.local_topk(q16, k, self.row_base)
        else:
            exact, ids, flags = self.backend.local_topk(q16, k, self.row_base, workspace)
        if self.world == 1:
            scores, gids = self.backend.merge(exact.unsqueeze(0).contiguous(),
                                              ids.unsqueeze(0).contiguous(), k)
            return scores, gids, flags
        B = exact.shape[0]
        # one collective: pack {score bits, id} as int64 [B, 2k]
        packed = torch.cat([exact.view(torch.int64), ids], dim=1).contiguous()
        # concatenated-along-dim-0 output: the form both RCCL and gloo accept
        flat = torch.empty((self.world * B, 2 * k), dtype=torch.int64, device=packed.device)
        if packed.is_cuda and self.dist.get_backend(self.group) == "gloo":
            # rehearsal path (gloo has no device all-gather): stage through the host
            host = torch.empty(flat.shape, dtype=torch.int64)
            self.dist.all_gather_into_tensor(host, packed.cpu(), group=self.group)
            flat.copy_(host)
        else:
            self.dist.all_gather_into_tensor(flat, packed, group=self.group)
        gathered = flat.view(self.world, B, 2 * k)
        exact_all = gathered[:, :, :k].contiguous().view(torch.float64)
        ids_all = gathered[:, :, k:].contiguous()
        scores, gids = self.backend.merge(exact_all, ids_all, k)
        return scores, gids, flags

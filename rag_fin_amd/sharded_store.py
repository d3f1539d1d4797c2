"""`ShardedCorpusStore`: the corpus store of rag_fin_amd.store with its vectors row-sharded
over the GPUs of one node, so that `VectorRAG.search` -- the reference's
vector_rag_mcp/main.py:48-70 surface -- works unchanged on N > 1 GPUs.

Layout.  One process per GPU (torch.distributed group).  Every batch handed to `add` is cut
into `world` contiguous slices; rank r keeps slice r's VECTORS in its own HBM (fp16, fragment
tiled, through its GpuIndex) and remembers each local row's GLOBAL id (= insertion order over
the whole collection, exactly the row numbers a single-GPU CorpusStore would give) in a device
table.  The six scalar columns (id, text, period, chunk_type, statement_type, primary_value)
are replicated on every rank: they are small next to the vectors and it lets any rank marshal
hits.  A search = local rf_search on every rank, ONE all-gather of the per-shard top-k, merge
by (score desc, id asc) on every rank (rag_fin_amd.sharded.ShardedSearcher): bit-identical to
the single-GPU answer, with flagged queries resolved through the exhaustive path first.

Calling convention.  Methods marked COLLECTIVE must be entered by every rank with the same
arguments.  For serving, where only rank 0 faces the MCP / REST layer, ranks > 0 park in
`start_workers()`; rank 0's searches then broadcast (B, k) and the query vectors first and the
workers join the collective -- callers above (VectorRAG, the MCP tools) see an ordinary store.

On-disk format = CorpusStore's (`ragfin-corpus-v1`: vectors.f16 in global row order +
columns.json), so a corpus saved on 1 GPU loads on 8 and back.  One node only: `save` has every
rank write its rows into the one file.
"""
from __future__ import annotations

import numpy as np

import logging
import threading

from . import _lib
from .sharded import HipShardBackend, ShardedSearcher
from .store import SCALAR_FIELDS, CorpusStore


logger = logging.getLogger(__name__)


def _torch():
    import torch
    return torch


class ShardedCorpusStore(CorpusStore):
    def __init__(self, name: str = "fin_chunks", dim: int = 384, capacity: int = 4096, device=None,
                 metric_type: str = "COSINE", group=None, index=None, backend=None):
        """capacity: rows of THIS rank's shard to reserve (grows on demand).  index / backend:
        test doubles (CPU); the product builds a GpuIndex + HipShardBackend."""
        import torch.distributed as dist
        if not dist.is_initialized():
            raise RuntimeError("ShardedCorpusStore needs an initialised torch.distributed process group "
                               "(one process per GPU); use CorpusStore on a single GPU")
        super().__init__(name, dim, capacity, device, metric_type, index=index)
        torch = _torch()
        self.group = group
        self.dist = dist
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self._backend_factory = (lambda ix: backend) if backend is not None else HipShardBackend
        self._id_map = torch.empty(0, dtype=torch.int64, device=self.index.device)   # local row -> global id
        self._searcher = None
        self._leading = False
        # One collective at a time per process: the header / query broadcasts and the all-gather of a search share
        # the group (and the searcher's lane buffers and the index's default workspace), so two threads of the
        # serving rank (FastMCP runs the tools on a pool: vector_rag_mcp/main.py:126,135-146) must not interleave
        # them -- the workers would pair one call's header with another call's queries.  Re-entrant: query() with
        # vectors and save() take it too.
        self._coll_lock = threading.RLock()
        on_gpu = dist.get_backend(group) == "nccl"
        self._bdev = self.index.device if on_gpu else torch.device("cpu")

    # -- ingest (COLLECTIVE) --------------------------------------------------------------------
    def add(self, ids, texts, embeddings, periods, chunk_types, statement_types, primary_values,
            local: bool = False) -> int:
        """COLLECTIVE.  Scalar columns: the whole batch, on every rank.  `embeddings`: the whole
        batch [n, dim] (each rank keeps its slice), or with local=True only this rank's slice
        [hi - lo, dim] where (lo, hi) = ShardedSearcher.shard_bounds(n, world, rank) -- the
        encoder runs as replicas, each rank embedding its own rows."""
        torch = _torch()
        n = len(ids)
        cols = (texts, periods, chunk_types, statement_types, primary_values)
        if any(len(c) != n for c in cols):
            raise ValueError("insert columns differ in length")
        for pk in ids:
            if pk in self._pk_row:
                raise ValueError(f"duplicate primary key {pk!r}")
        if len(set(ids)) != n:
            raise ValueError("duplicate primary keys in insert")
        lo, hi = ShardedSearcher.shard_bounds(n, self.world, self.rank)
        m = hi - lo
        if torch.is_tensor(embeddings) and embeddings.dtype == torch.float16:
            vec = embeddings if local else embeddings[lo:hi]
        else:
            emb = embeddings if torch.is_tensor(embeddings) else np.asarray(embeddings, dtype=np.float32)
            emb = emb if local else emb[lo:hi]
            vec = self.index.to_fp16(emb, normalize=self.metric_type == "COSINE") if m else \
                torch.empty((0, self.dim), dtype=torch.float16, device=self.index.device)
        if tuple(vec.shape) != (m, self.dim):
            raise ValueError(f"rank {self.rank}: embeddings must be [{m if local else n}, {self.dim}]")
        if m:
            if self.index.size + m > self.index.capacity:
                self._grow(self.index.size + m)
            self.index.add(vec.to(self.index.device).contiguous())
        base = self.num_entities
        self._id_map = torch.cat([self._id_map,
                                  torch.arange(base + lo, base + hi, dtype=torch.int64, device=self._id_map.device)])
        self._searcher = None        # id_map changed (and _grow may have replaced the index)
        for j, pk in enumerate(ids):
            self._pk_row[pk] = base + j
        self.columns["id"].extend(ids)
        self.columns["text"].extend(texts)
        self.columns["period"].extend(periods)
        self.columns["chunk_type"].extend(chunk_types)
        self.columns["statement_type"].extend(statement_types)
        self.columns["primary_value"].extend(float(v) for v in primary_values)
        return n

    def drop(self) -> None:
        torch = _torch()
        super().drop()
        self._id_map = torch.empty(0, dtype=torch.int64, device=self.index.device)
        self._searcher = None

    def flush(self) -> None:
        if self.index.device.type == "cuda":
            _torch().cuda.synchronize(self.index.device)
        self.dist.barrier(group=self.group)

    @property
    def local_rows(self) -> int:
        return int(self._id_map.numel())

    # -- search (COLLECTIVE; rank 0 alone once start_workers() is active) ---------------------------
    def _get_searcher(self) -> ShardedSearcher:
        if self._searcher is None:
            self._searcher = ShardedSearcher(self._backend_factory(self.index), row_base=0, group=self.group,
                                             id_map=self._id_map)
        return self._searcher

    def _collective_search(self, q16, limit: int):
        """-> (scores, global ids, merged flags).  A rank whose local part raises still enters the
        collective (ShardedSearcher.poison_step), so the others do not hang; every rank then sees
        SHARD_FAILED in the merged flags."""
        torch = _torch()
        s = self._get_searcher()
        B = q16.shape[0]
        if limit > _lib.RF_MAX_K:
            # large limits (graph_cons.py:275-281 asks for 1000): each shard's own top-`limit`
            # (paged, exhaustive beyond RF_MAX_K), one all-gather, merge
            try:
                exact, ids = self._local_large(q16, limit)
                flags = torch.zeros(B, dtype=torch.int32, device=exact.device)
            except Exception:
                logger.exception("rank %d: local search failed", self.rank)
                return s.poison_step(B, limit, q16.device)
            return s._gather_merge(exact, ids, flags, limit)
        try:
            scores, gids, gflags = s.search(q16, limit, resolve=False)
        except Exception:
            logger.exception("rank %d: local search failed", self.rank)
            return s.poison_step(B, limit, q16.device)
        s.resolve_flagged(q16, limit, scores, gids, gflags)   # flagged queries; a no-op for every rank on SHARD_FAILED
        return scores, gids, gflags

    def _local_large(self, q16, limit: int):
        torch = _torch()
        s = self._get_searcher()
        if hasattr(self.index, "search_large"):
            _, ids, exact = self.index.search_large(q16, limit, want_exact=True)
        else:   # CPU double
            exact, ids = s.backend.local_exhaustive(q16, limit, 0)
        s._map_ids_(ids)
        return exact, ids

    def search_rows(self, data, limit: int):
        torch = _torch()
        if limit < 1:
            raise ValueError("limit must be >= 1")
        q16 = self._prepare_queries(data).contiguous()
        with self._coll_lock:   # broadcasts + collective + download as ONE unit (see __init__)
            if self._leading:
                hdr = torch.tensor([q16.shape[0], limit], dtype=torch.int64, device=self._bdev)
                self.dist.broadcast(hdr, src=0, group=self.group)
                qb = q16.to(self._bdev)
                self.dist.broadcast(qb, src=0, group=self.group)
            elif self.world > 1:
                # collective form (every rank called search with its own copy of the queries): all shards are searched
                # with RANK 0's bits.  A merged top-k is the top-k of a query only if every shard scored the same
                # vector, and two ranks that embedded the same text need not agree in the last fp16 bit.
                qb = q16.to(self._bdev)
                self.dist.broadcast(qb, src=0, group=self.group)
                q16 = qb.to(q16.device)
            scores, gids, gflags = self._collective_search(q16, limit)
            if ShardedSearcher.failed(gflags):
                raise RuntimeError("sharded search failed: a rank could not scan its shard (see that rank's log)")
            kk = min(limit, self.num_entities)
            return scores[:, :kk].cpu().numpy(), gids[:, :kk].cpu().numpy()

    # -- serving: ranks > 0 follow rank 0 ---------------------------------------------------------------
    def start_workers(self) -> None:
        """COLLECTIVE.  Rank 0 returns at once and from now on drives every search; ranks > 0
        stay inside, answering rank 0's searches, until rank 0 calls stop_workers()."""
        torch = _torch()
        if self.rank == 0:
            self._leading = True
            return
        while True:
            hdr = torch.zeros(2, dtype=torch.int64, device=self._bdev)
            self.dist.broadcast(hdr, src=0, group=self.group)
            B, k = int(hdr[0].item()), int(hdr[1].item())
            if B <= 0:
                return
            q = torch.empty((B, self.dim), dtype=torch.float16, device=self._bdev)
            self.dist.broadcast(q, src=0, group=self.group)
            try:   # a failing scan is reported through the collective itself (SHARD_FAILED); the worker lives on
                self._collective_search(q.to(self.index.device), k)
            except Exception:
                logger.exception("rank %d: search step failed outside the local scan", self.rank)
                raise

    def stop_workers(self) -> None:
        """Rank 0 only: release the ranks parked in start_workers()."""
        torch = _torch()
        if self.rank != 0 or not self._leading:
            return
        self.dist.broadcast(torch.zeros(2, dtype=torch.int64, device=self._bdev), src=0, group=self.group)
        self._leading = False

    # -- scalar queries: columns are replicated, vectors live on their owner ----------------------------
    def query(self, expr: str = "", limit=None, output_fields=None):
        fields = list(output_fields or ["id"])
        if "embedding" not in fields:
            return super().query(expr, limit, fields)
        # COLLECTIVE when vectors are asked for: every rank contributes the rows it owns
        with self._coll_lock:
            return self._query_with_vectors(expr, limit, fields)

    def _query_with_vectors(self, expr, limit, fields):
        torch = _torch()
        recs = super().query(expr, limit, [f for f in fields if f != "embedding"] + ["id"])
        want = [self._pk_row[r["id"]] for r in recs]
        mine = {}
        if want and self._id_map.numel():
            gmap = self._id_map.cpu().numpy()
            pos = {int(g): i for i, g in enumerate(gmap)}
            loc = [(g, pos[g]) for g in want if g in pos]
            if loc:
                rows = self.index.get_rows(np.asarray([l for _, l in loc], dtype=np.int64)).float().cpu().numpy()
                mine = {g: rows[j].tolist() for j, (g, _) in enumerate(loc)}
        parts = [None] * self.world
        self.dist.all_gather_object(parts, mine, group=self.group)
        vecs = {}
        for p in parts:
            vecs.update(p)
        for r, g in zip(recs, want):
            r["embedding"] = vecs[g]
            if "id" not in fields:
                r.pop("id", None)
        return recs

    # -- persistence: the single-GPU format, written / read in place by every rank (one node) -------------
    def save(self, path: str, chunk_rows: int = 1 << 18) -> None:
        """COLLECTIVE."""
        with self._coll_lock:
            self._save(path, chunk_rows)

    def _save(self, path: str, chunk_rows: int) -> None:
        import json
        import os
        n = self.num_entities
        vec_path = os.path.join(path, "vectors.f16")
        if self.rank == 0:
            os.makedirs(path, exist_ok=True)
            with open(vec_path, "wb") as f:
                f.truncate(n * self.dim * 2)
        self.dist.barrier(group=self.group)
        if self.local_rows:
            mm = np.memmap(vec_path, dtype=np.float16, mode="r+", shape=(n, self.dim))
            gmap = self._id_map.cpu().numpy()
            for s0 in range(0, self.local_rows, chunk_rows):
                rows = np.arange(s0, min(self.local_rows, s0 + chunk_rows), dtype=np.int64)
                mm[gmap[rows]] = self.index.get_rows(rows).cpu().numpy()
            mm.flush()
            del mm
        self.dist.barrier(group=self.group)
        if self.rank == 0:
            meta = {"format": "ragfin-corpus-v1", "name": self.name, "dim": self.dim,
                    "metric_type": self.metric_type, "n": n, "columns": self.columns}
            tmp = os.path.join(path, "columns.json.tmp")
            with open(tmp, "w", encoding="utf-8") as f:
                json.dump(meta, f, ensure_ascii=False)
            os.replace(tmp, os.path.join(path, "columns.json"))
        self.dist.barrier(group=self.group)

    @classmethod
    def load_from(cls, path: str, device=None, capacity=None, chunk_rows: int = 1 << 18, group=None,
                  index_factory=None, backend=None) -> "ShardedCorpusStore":
        """COLLECTIVE.  Rank r memory-maps vectors.f16 and streams rows
        shard_bounds(n, world, r) into its HBM; the columns are read by every rank."""
        import json
        import os
        import torch.distributed as dist
        torch = _torch()
        with open(os.path.join(path, "columns.json"), encoding="utf-8") as f:
            meta = json.load(f)
        if meta.get("format") != "ragfin-corpus-v1":
            raise ValueError(f"{path}: not a ragfin corpus directory")
        n, dim = int(meta["n"]), int(meta["dim"])
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        lo, hi = ShardedSearcher.shard_bounds(n, world, rank)
        cap = max(capacity or 0, hi - lo, 1)
        index = index_factory(dim, cap, device) if index_factory is not None else None
        st = cls(meta["name"], dim=dim, capacity=cap, device=device, metric_type=meta["metric_type"], group=group,
                 index=index, backend=backend)
        if n:
            expect = n * dim * 2
            got = os.path.getsize(os.path.join(path, "vectors.f16"))
            if got != expect:
                raise ValueError(f"{path}/vectors.f16 holds {got} bytes, expected {expect}")
            mm = np.memmap(os.path.join(path, "vectors.f16"), dtype=np.float16, mode="r", shape=(n, dim))
            for s0 in range(lo, hi, chunk_rows):
                s1 = min(hi, s0 + chunk_rows)
                st.index.add(torch.from_numpy(np.ascontiguousarray(mm[s0:s1])).to(st.index.device))
            del mm
        st._id_map = torch.arange(lo, hi, dtype=torch.int64, device=st.index.device)
        cols = meta["columns"]
        if any(len(cols[f]) != n for f in SCALAR_FIELDS):
            raise ValueError(f"{path}: column lengths do not match n={n}")
        st.columns = {f: list(cols[f]) for f in SCALAR_FIELDS}
        st._pk_row = {pk: i for i, pk in enumerate(st.columns["id"])}
        return st

"""HBM-resident corpus store: the in-process replacement for the reference's Milvus
collection on the vector-RAG path.

Reference surface mirrored (pymilvus `Collection`, as the reference uses it):
  schema          "chunking_storing (1).py":14-22  (id, text, embedding[384], period,
                                                   chunk_type, statement_type, primary_value)
  insert/flush/load                    same file :383-396 (seven parallel columns)
  search(data, "embedding", {"metric_type": "COSINE"}, limit, output_fields=[...])
                                        vector_rag_mcp/main.py:51-57, retrieve.py:28-34
  hit.score / hit.entity.<field>        vector_rag_mcp/main.py:59-70
  num_entities                          vector_rag_mcp/main.py:113,120,164
  query(expr="id in [...]" | "", limit, output_fields)   graph_cons.py:38-42,308-311;
                                        test_vector.py:35-39

Vectors live on the GPU (fp16, MFMA-fragment tiled); the scalar
columns stay in host Python lists.  All arithmetic goes through libragfin_hip.so;
there is no CPU path.
"""
from __future__ import annotations

import ctypes
import re
import threading
from ctypes import c_void_p
from typing import Any, Iterable, Sequence

import numpy as np

from . import _lib

import os as _os
# RAGFIN_ZERO_COPY=0: search_host copies results with async memcpys instead of letting the merge kernel
# store into the pinned host buffers (A/B switch)
_ZERO_COPY = _os.environ.get("RAGFIN_ZERO_COPY", "1") != "0"

SCALAR_FIELDS = ("id", "text", "period", "chunk_type", "statement_type", "primary_value")


def _torch():
    import torch
    return torch


def require_gpu(device=None):
    """Fail loudly when there is no MI355X to run on."""
    torch = _torch()
    if not torch.cuda.is_available():
        raise RuntimeError("rag_fin_amd needs a ROCm GPU (gfx950); torch.cuda.is_available() is "
                           "False and there is no CPU fallback")
    dev = torch.device(device if device is not None else "cuda:0")
    if dev.type != "cuda":
        raise RuntimeError(f"device {dev} is not a GPU")
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    _lib.check(_lib.load_library().rf_device_check(idx))
    return torch.device("cuda", idx)


class GpuIndex:
    """Thin object wrapper over rf_index_* / rf_search (include/ragfin.h)."""

    def __init__(self, dim: int, capacity: int, device=None):
        torch = _torch()
        self.device = require_gpu(device)
        self.lib = _lib.load_library()
        self.dim = int(dim)
        self.capacity = int(capacity)
        nbytes = self.lib.rf_index_storage_bytes(self.dim, self.capacity)
        if nbytes == 0:
            raise _lib.RagfinError(-1, f"unsupported index shape dim={dim} capacity={capacity}")
        with torch.cuda.device(self.device):
            self.storage = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
            handle = c_void_p()
            _lib.check(self.lib.rf_index_create(ctypes.byref(handle), self.dim, self.capacity,
                                                c_void_p(self.storage.data_ptr()), nbytes,
                                                self.device.index))
            self.handle = handle
            ws = self.lib.rf_search_workspace_bytes(self.handle)
            self.workspace = torch.zeros(ws, dtype=torch.uint8, device=self.device)
            self.workspace_bytes = ws
        self._lock = threading.Lock()
        self._host_bufs = {}

    def __del__(self):
        h = getattr(self, "handle", None)
        if h:
            self.lib.rf_index_destroy(h)
            self.handle = None

    @property
    def size(self) -> int:
        return int(self.lib.rf_index_size(self.handle))

    def reset(self) -> None:
        torch = _torch()
        with torch.cuda.device(self.device):
            _lib.check(self.lib.rf_index_reset(self.handle, _lib.current_stream_ptr()))

    # -- ingest --------------------------------------------------------------
    def add(self, rows) -> None:
        """rows: fp16 [n, dim] tensor on this device (row-major, contiguous)."""
        torch = _torch()
        if rows.dtype != torch.float16 or rows.dim() != 2 or rows.shape[1] != self.dim:
            raise ValueError(f"add expects fp16 [n, {self.dim}], got {rows.dtype} {tuple(rows.shape)}")
        rows = rows.to(self.device).contiguous()
        with torch.cuda.device(self.device):
            _lib.check(self.lib.rf_index_add_f16(self.handle, c_void_p(rows.data_ptr()),
                                                 rows.shape[0], _lib.current_stream_ptr()))

    def to_fp16(self, rows_f32, normalize: bool = True):
        """fp32 [n, dim] -> (L2-normalised) fp16 on device, via rf_normalize_f32_to_f16."""
        torch = _torch()
        x = torch.as_tensor(rows_f32, dtype=torch.float32).to(self.device).contiguous()
        if x.dim() != 2 or x.shape[1] != self.dim:
            raise ValueError(f"expected [n, {self.dim}] vectors, got {tuple(x.shape)}")
        out = torch.empty(x.shape, dtype=torch.float16, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.rf_normalize_f32_to_f16(c_void_p(x.data_ptr()), x.shape[0], self.dim,
                                                        1 if normalize else 0,
                                                        c_void_p(out.data_ptr()),
                                                        _lib.current_stream_ptr()))
        return out

    def get_rows(self, row_ids):
        torch = _torch()
        ids = torch.as_tensor(row_ids, dtype=torch.int64).to(self.device).contiguous()
        out = torch.empty((ids.numel(), self.dim), dtype=torch.float16, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.rf_index_get_rows_f16(self.handle, c_void_p(ids.data_ptr()),
                                                      ids.numel(), c_void_p(out.data_ptr()),
                                                      _lib.current_stream_ptr()))
        return out

    # -- search --------------------------------------------------------------
    def new_workspace(self):
        """An extra search workspace: one per batch in flight when several streams
        search the same (immutable) index concurrently."""
        torch = _torch()
        return torch.zeros(self.workspace_bytes, dtype=torch.uint8, device=self.device)

    def search_raw(self, q16, k: int, id_base: int = 0, want_exact: bool = False, out=None,
                   workspace=None, stream_ptr=None):
        """Enqueue rf_search on the current stream (or on `stream_ptr`, a c_void_p holding a
        hipStream_t of this device); no host sync.  Returns
        (scores f32 [B,k], ids i64 [B,k], exact f64 [B,k] | None, flags u32 [B])."""
        torch = _torch()
        if q16.dtype != torch.float16 or q16.dim() != 2 or q16.shape[1] != self.dim:
            raise ValueError(f"search expects fp16 [B, {self.dim}] queries")
        if not q16.is_contiguous() or q16.device != self.device:
            q16 = q16.to(self.device).contiguous()
        B = q16.shape[0]
        if out is None:
            scores = torch.empty((B, k), dtype=torch.float32, device=self.device)
            ids = torch.empty((B, k), dtype=torch.int64, device=self.device)
            exact = torch.empty((B, k), dtype=torch.float64, device=self.device) if want_exact else None
            flags = torch.empty((B,), dtype=torch.int32, device=self.device)
        else:
            scores, ids, exact, flags = out
        with torch.cuda.device(self.device):
            _lib.check(self.lib.rf_search(
                self.handle, c_void_p(q16.data_ptr()), B, k, id_base, c_void_p(scores.data_ptr()),
                c_void_p(ids.data_ptr()), c_void_p(exact.data_ptr()) if exact is not None else None,
                c_void_p(flags.data_ptr()),
                c_void_p((workspace if workspace is not None else self.workspace).data_ptr()),
                self.workspace_bytes, stream_ptr if stream_ptr is not None else _lib.current_stream_ptr()))
        return scores, ids, exact, flags

    def enqueue_search(self, q_ptr: int, B: int, k: int, id_base: int, scores_ptr: int, ids_ptr: int,
                       exact_ptr: int, flags_ptr: int, workspace_ptr: int, stream_ptr):
        """The bare rf_search enqueue for callers that own every buffer (the sharded step: no
        tensor checks, no allocations, no stream / device context).  The caller guarantees that
        this index's device is the thread's current HIP device."""
        rc = self.lib.rf_search(self.handle, q_ptr, B, k, id_base, scores_ptr, ids_ptr, exact_ptr, flags_ptr,
                                workspace_ptr, self.workspace_bytes, stream_ptr)
        if rc:
            _lib.check(rc)

    def search_profile(self, q16, k: int):
        """rf_search_profile: per-stage HIP-event times in ms (synchronises)."""
        torch = _torch()
        q16 = q16.to(self.device).contiguous()
        B = min(q16.shape[0], 256)      # the first sweep: 64 queries, or up to 256 on the wide path
        scores = torch.empty((B, k), dtype=torch.float32, device=self.device)
        ids = torch.empty((B, k), dtype=torch.int64, device=self.device)
        flags = torch.empty((B,), dtype=torch.int32, device=self.device)
        ms = (ctypes.c_float * 4)()
        with torch.cuda.device(self.device):
            _lib.check(self.lib.rf_search_profile(
                self.handle, c_void_p(q16.data_ptr()), B, k, 0, c_void_p(scores.data_ptr()),
                c_void_p(ids.data_ptr()), None, c_void_p(flags.data_ptr()),
                c_void_p(self.workspace.data_ptr()), self.workspace_bytes,
                _lib.current_stream_ptr(), ms))
        return {"sample": ms[0], "threshold": ms[1], "emit": ms[2], "merge": ms[3]}

    def search_exhaustive(self, q16, k: int, id_base: int = 0, want_exact: bool = False):
        torch = _torch()
        q16 = q16.to(self.device).contiguous()
        B = q16.shape[0]
        scores = torch.empty((B, k), dtype=torch.float32, device=self.device)
        ids = torch.empty((B, k), dtype=torch.int64, device=self.device)
        exact = torch.empty((B, k), dtype=torch.float64, device=self.device) if want_exact else None
        with torch.cuda.device(self.device):
            _lib.check(self.lib.rf_search_exhaustive(
                self.handle, c_void_p(q16.data_ptr()), B, k, id_base, c_void_p(scores.data_ptr()),
                c_void_p(ids.data_ptr()), c_void_p(exact.data_ptr()) if exact is not None else None,
                c_void_p(self.workspace.data_ptr()), self.workspace_bytes,
                _lib.current_stream_ptr()))
        return scores, ids, exact

    def search_large(self, q16, k: int, id_base: int = 0, want_exact: bool = False):
        """Limits above RF_MAX_K: the first page through the fused path, further pages
        of RF_MAX_K through rf_search_exhaustive_after (each page = the hits ranked
        strictly after the previous page's last hit).  Returns (scores, ids) [B, k]
        (+ the fp64 ranking scores with want_exact: what a cross-shard merge ranks by)."""
        torch = _torch()
        q16 = q16.to(self.device).contiguous()
        B = q16.shape[0]
        page = _lib.RF_MAX_K
        s0, i0, e0 = self.search(q16, page, id_base, want_exact=True)
        scores, ids, exacts = [s0], [i0], [e0]
        last_s, last_i = e0[:, -1].contiguous(), i0[:, -1].contiguous()
        got = page
        while got < k and bool((last_i >= 0).any()):
            s = torch.empty((B, page), dtype=torch.float32, device=self.device)
            i = torch.empty((B, page), dtype=torch.int64, device=self.device)
            e = torch.empty((B, page), dtype=torch.float64, device=self.device)
            # exhausted queries keep a bound nothing can follow
            bs = torch.where(last_i >= 0, last_s, torch.full_like(last_s, float("-inf")))
            bi = torch.where(last_i >= 0, last_i, torch.full_like(last_i, 2 ** 62))
            with self._lock, torch.cuda.device(self.device):
                _lib.check(self.lib.rf_search_exhaustive_after(
                    self.handle, c_void_p(q16.data_ptr()), B, page, id_base, c_void_p(bs.data_ptr()),
                    c_void_p(bi.data_ptr()), c_void_p(s.data_ptr()), c_void_p(i.data_ptr()),
                    c_void_p(e.data_ptr()), c_void_p(self.workspace.data_ptr()), self.workspace_bytes,
                    _lib.current_stream_ptr()))
            scores.append(s)
            ids.append(i)
            exacts.append(e)
            last_s, last_i = e[:, -1].contiguous(), i[:, -1].contiguous()
            got += page
        if got < k:   # corpus exhausted before k hits: pad like rf_search does
            scores.append(torch.full((B, k - got), float("-inf"), dtype=torch.float32, device=self.device))
            ids.append(torch.full((B, k - got), -1, dtype=torch.int64, device=self.device))
            exacts.append(torch.full((B, k - got), float("-inf"), dtype=torch.float64, device=self.device))
        out = (torch.cat(scores, 1)[:, :k].contiguous(), torch.cat(ids, 1)[:, :k].contiguous())
        return out + (torch.cat(exacts, 1)[:, :k].contiguous(),) if want_exact else out

    def search(self, q16, k: int, id_base: int = 0, want_exact: bool = False):
        """rf_search, then re-run any query the fused path could not prove exact
        (flags != 0) through the exhaustive fp64 kernel.  Serialised: the index
        workspace is shared (SURVEY.md 8b threading row)."""
        torch = _torch()
        with self._lock:
            scores, ids, exact, flags = self.search_raw(q16, k, id_base, want_exact)
            bad = torch.nonzero(flags != 0).flatten()
            if bad.numel() > 0:
                qb = q16.to(self.device)[bad].contiguous()
                s2, i2, e2 = self.search_exhaustive(qb, k, id_base, want_exact)
                scores[bad] = s2
                ids[bad] = i2
                if exact is not None:
                    exact[bad] = e2
        return scores, ids, exact

    ZERO_COPY_MAX = 4096   # B * k up to which search_host lets the kernel write into host memory

    def search_host(self, q16, k: int):
        """search() whose results land on the host with ONE synchronisation: scores, ids and
        flags are copied into cached pinned buffers asynchronously.  -> (scores f32 [B,k],
        ids i64 [B,k]) numpy arrays (the caller's own copies)."""
        torch = _torch()
        with self._lock:
            B = q16.shape[0]
            key = (B, k)
            bufs = self._host_bufs.get(key)
            if bufs is None:
                bufs = self._host_bufs[key] = (torch.empty((B, k), dtype=torch.float32, pin_memory=True),
                                               torch.empty((B, k), dtype=torch.int64, pin_memory=True),
                                               torch.empty((B,), dtype=torch.int32, pin_memory=True))
            if B * k <= self.ZERO_COPY_MAX and _ZERO_COPY:
                # query-sized results: the merge kernel stores straight into the pinned host buffers
                # (host-coherent memory, mapped at the same address on the device) -- no copy commands,
                # only the synchronisation
                self.search_raw(q16, k, out=(bufs[0], bufs[1], None, bufs[2]))
            else:
                scores, ids, _, flags = self.search_raw(q16, k)
                bufs[0].copy_(scores, non_blocking=True)
                bufs[1].copy_(ids, non_blocking=True)
                bufs[2].copy_(flags, non_blocking=True)
            torch.cuda.current_stream(self.device).synchronize()
            if bool(bufs[2].any()):
                bad = torch.nonzero(bufs[2] != 0).flatten()
                qb = q16.to(self.device)[bad.to(self.device)].contiguous()
                s2, i2, _ = self.search_exhaustive(qb, k)
                bufs[0][bad] = s2.cpu()
                bufs[1][bad] = i2.cpu()
            # private copies, taken while the lock is still held: the pinned buffers are shared by every caller with
            # this (B, k) and the next search's merge kernel stores straight into them
            return bufs[0].numpy().copy(), bufs[1].numpy().copy()

    def debug_scores(self, q16, n: int | None = None):
        torch = _torch()
        n = self.size if n is None else n
        q16 = q16.to(self.device).contiguous()
        out = torch.empty((q16.shape[0], n), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.rf_debug_scores(self.handle, c_void_p(q16.data_ptr()), q16.shape[0], n,
                                                c_void_p(out.data_ptr()), _lib.current_stream_ptr()))
        return out


class _Entity:
    """hit.entity.<field> / hit.entity.get(field) as pymilvus exposes it."""

    def __init__(self, fields: dict):
        self.__dict__.update(fields)
        self._fields = fields

    def get(self, name, default=None):
        return self._fields.get(name, default)

    def to_dict(self):
        return dict(self._fields)


class Hit:
    def __init__(self, row: int, pk: Any, score: float, fields: dict):
        self.row = row
        self.id = pk
        self.score = score
        self.distance = score
        self.entity = _Entity(fields)

    def __repr__(self):
        return f"Hit(id={self.id!r}, score={self.score:.6f})"


_ID_IN = re.compile(r"^\s*id\s+in\s+\[(.*)\]\s*$", re.S)


class CorpusStore:
    """Drop-in for the reference's `Collection("fin_chunks")` on this path."""

    def __init__(self, name: str = "fin_chunks", dim: int = 384, capacity: int = 4096,
                 device=None, metric_type: str = "COSINE", index=None):
        self.name = name
        self.dim = dim
        self.metric_type = metric_type.upper()
        if self.metric_type not in ("COSINE", "IP"):
            raise ValueError("metric_type must be COSINE or IP")
        # `index`: an already-built GpuIndex (tests of the sharded store pass a CPU double)
        self.index = index if index is not None else GpuIndex(dim, capacity, device)
        self.columns: dict[str, list] = {f: [] for f in SCALAR_FIELDS}
        self._pk_row: dict[Any, int] = {}

    # -- pymilvus-shaped lifecycle ---------------------------------------------
    def flush(self) -> None:
        _torch().cuda.synchronize(self.index.device)

    def load(self) -> None:
        return None

    def release(self) -> None:
        return None

    def drop(self) -> None:
        """utility.drop_collection + recreate ("chunking_storing (1).py":25-28)."""
        self.index.reset()
        for col in self.columns.values():
            col.clear()
        self._pk_row.clear()

    @property
    def num_entities(self) -> int:
        return len(self.columns["id"])

    def _grow(self, need: int) -> None:
        old = self.index
        cap = max(need, old.capacity * 2)
        new = type(old)(self.dim, cap, old.device)
        n = old.size
        step = 1 << 18
        for s in range(0, n, step):
            new.add(old.get_rows(np.arange(s, min(n, s + step), dtype=np.int64)))
        self.index = new

    # -- ingest ------------------------------------------------------------------
    def add(self, ids: Sequence, texts: Sequence[str], embeddings, periods: Sequence[str],
            chunk_types: Sequence[str], statement_types: Sequence[str],
            primary_values: Sequence[float]) -> int:
        torch = _torch()
        n = len(ids)
        cols = (texts, periods, chunk_types, statement_types, primary_values)
        if any(len(c) != n for c in cols):
            raise ValueError("insert columns differ in length")
        if torch.is_tensor(embeddings) and embeddings.dtype == torch.float16:
            vec = embeddings  # already normalised fp16 (embedder output)
            if vec.shape != (n, self.dim):
                raise ValueError(f"embeddings must be [{n}, {self.dim}]")
        else:
            emb = np.asarray(embeddings, dtype=np.float32) if not torch.is_tensor(embeddings) else embeddings
            if tuple(emb.shape) != (n, self.dim):
                raise ValueError(f"embeddings must be [{n}, {self.dim}], got {tuple(emb.shape)}")
            vec = self.index.to_fp16(emb, normalize=self.metric_type == "COSINE")
        for pk in ids:
            if pk in self._pk_row:
                raise ValueError(f"duplicate primary key {pk!r}")
        if len(set(ids)) != n:
            raise ValueError("duplicate primary keys in insert")
        if self.index.size + n > self.index.capacity:
            self._grow(self.index.size + n)
        base = self.index.size
        self.index.add(vec)
        for j, pk in enumerate(ids):
            self._pk_row[pk] = base + j
        self.columns["id"].extend(ids)
        self.columns["text"].extend(texts)
        self.columns["period"].extend(periods)
        self.columns["chunk_type"].extend(chunk_types)
        self.columns["statement_type"].extend(statement_types)
        self.columns["primary_value"].extend(float(v) for v in primary_values)
        return n

    def insert(self, data: Sequence[Sequence]) -> int:
        """Column-major insert in the reference's order
        [id, text, embedding, period, chunk_type, statement_type, primary_value]."""
        if len(data) != 7:
            raise ValueError("insert expects 7 columns: id, text, embedding, period, chunk_type, "
                             "statement_type, primary_value")
        ids, texts, emb, periods, ctypes_, stypes, pvals = data
        return self.add(ids, texts, emb, periods, ctypes_, stypes, pvals)

    # -- search --------------------------------------------------------------------
    def _prepare_queries(self, data):
        torch = _torch()
        if torch.is_tensor(data) and data.dtype == torch.float16:
            q = data.to(self.index.device)
            return q if q.dim() == 2 else q.unsqueeze(0)
        q = np.asarray(data, dtype=np.float32) if not torch.is_tensor(data) else data.float()
        if q.ndim == 1:
            q = q[None, :]
        return self.index.to_fp16(q, normalize=self.metric_type == "COSINE")

    def search_rows(self, data, limit: int):
        """(scores f32 [B,k'], rows i64 [B,k']) as host numpy, k' = min(limit, N)."""
        if limit < 1:
            raise ValueError("limit must be >= 1")
        q16 = self._prepare_queries(data)
        if limit > _lib.RF_MAX_K:
            scores, rows = self.index.search_large(q16, limit)   # paged, exhaustive beyond 64
            kk = min(limit, self.num_entities)
            return scores[:, :kk].cpu().numpy(), rows[:, :kk].cpu().numpy()
        scores, rows = self.index.search_host(q16, limit)   # one synchronisation for the whole download
        kk = min(limit, self.num_entities)
        return scores[:, :kk], rows[:, :kk]

    def search(self, data, anns_field: str = "embedding", param: dict | None = None,
               limit: int = 3, expr=None, output_fields: Iterable[str] | None = None):
        """pymilvus-shaped search: one list of hits per query vector, best first."""
        if anns_field != "embedding":
            raise ValueError(f"unknown vector field {anns_field!r}")
        metric = (param or {}).get("metric_type", self.metric_type).upper()
        if metric != self.metric_type:
            raise ValueError(f"collection was built for {self.metric_type}, search asked for {metric}")
        if expr not in (None, ""):
            raise NotImplementedError("filtered search is outside the reference's use of this path")
        fields = list(output_fields or [])
        for f in fields:
            if f not in self.columns:
                raise KeyError(f"unknown output field {f!r}")
        scores, rows = self.search_rows(data, limit)
        out = []
        for b in range(rows.shape[0]):
            hits = []
            for j in range(rows.shape[1]):
                r = int(rows[b, j])
                if r < 0:
                    break
                hits.append(Hit(r, self.columns["id"][r], float(scores[b, j]),
                                {f: self.columns[f][r] for f in fields}))
            out.append(hits)
        return out

    # -- persistence -------------------------------------------------
    # The reference leans on the Milvus server for durability and re-creates the
    # collection on every ingest ("chunking_storing (1).py":25-28); an in-process store
    # needs its own format.  Directory layout:
    #   vectors.f16   raw little-endian fp16, row-major [n, dim]   (np.memmap-able)
    #   columns.json  {"name", "dim", "metric_type", "n", "columns": {field: [...]}}
    def save(self, path: str, chunk_rows: int = 1 << 18) -> None:
        import json
        import os
        os.makedirs(path, exist_ok=True)
        n = self.num_entities
        tmp = os.path.join(path, "vectors.f16.tmp")
        with open(tmp, "wb") as f:
            for s0 in range(0, n, chunk_rows):
                rows = np.arange(s0, min(n, s0 + chunk_rows), dtype=np.int64)
                f.write(self.index.get_rows(rows).cpu().numpy().tobytes())
        os.replace(tmp, os.path.join(path, "vectors.f16"))
        meta = {"format": "ragfin-corpus-v1", "name": self.name, "dim": self.dim,
                "metric_type": self.metric_type, "n": n, "columns": self.columns}
        tmp = os.path.join(path, "columns.json.tmp")
        with open(tmp, "w", encoding="utf-8") as f:
            json.dump(meta, f, ensure_ascii=False)
        os.replace(tmp, os.path.join(path, "columns.json"))

    @classmethod
    def load_from(cls, path: str, device=None, capacity: int | None = None,
                  chunk_rows: int = 1 << 18) -> "CorpusStore":
        """Memory-map vectors.f16 and stream it into HBM in chunks (never the whole file
        in host RAM)."""
        import json
        import os
        torch = _torch()
        with open(os.path.join(path, "columns.json"), encoding="utf-8") as f:
            meta = json.load(f)
        if meta.get("format") != "ragfin-corpus-v1":
            raise ValueError(f"{path}: not a ragfin corpus directory")
        n, dim = int(meta["n"]), int(meta["dim"])
        st = cls(meta["name"], dim=dim, capacity=max(capacity or 0, n, 1), device=device,
                 metric_type=meta["metric_type"])
        if n:
            expect = n * dim * 2
            got = os.path.getsize(os.path.join(path, "vectors.f16"))
            if got != expect:
                raise ValueError(f"{path}/vectors.f16 holds {got} bytes, expected {expect}")
            mm = np.memmap(os.path.join(path, "vectors.f16"), dtype=np.float16, mode="r", shape=(n, dim))
            for s0 in range(0, n, chunk_rows):
                st.index.add(torch.from_numpy(np.ascontiguousarray(mm[s0:s0 + chunk_rows])).to(st.index.device))
            del mm
        cols = meta["columns"]
        if any(len(cols[f]) != n for f in SCALAR_FIELDS):
            raise ValueError(f"{path}: column lengths do not match n={n}")
        st.columns = {f: list(cols[f]) for f in SCALAR_FIELDS}
        st._pk_row = {pk: i for i, pk in enumerate(st.columns["id"])}
        return st

    # -- scalar queries ----------------------------------------------------------------
    def query(self, expr: str = "", limit: int | None = None,
              output_fields: Iterable[str] | None = None) -> list[dict]:
        """`query(expr="id in [...]")` fetch-by-PK and `query(expr="", limit=n)` scan."""
        fields = list(output_fields or ["id"])
        want_vec = "embedding" in fields
        fields = [f for f in fields if f != "embedding"]
        for f in fields:
            if f not in self.columns:
                raise KeyError(f"unknown output field {f!r}")
        if expr is None or expr.strip() == "":
            rows = list(range(self.num_entities))
        else:
            m = _ID_IN.match(expr)
            if not m:
                raise NotImplementedError(f"unsupported expr {expr!r} (only 'id in [...]')")
            import ast
            body = m.group(1).strip()
            keys = list(ast.literal_eval("[" + body + "]")) if body else []
            rows = [self._pk_row[k] for k in keys if k in self._pk_row]
        if limit is not None:
            rows = rows[:limit]
        vecs = self.index.get_rows(np.asarray(rows, dtype=np.int64)).float().cpu().numpy() \
            if (want_vec and rows) else None
        out = []
        for j, r in enumerate(rows):
            rec = {f: self.columns[f][r] for f in fields}
            if "id" not in rec:
                rec["id"] = self.columns["id"][r]
            if vecs is not None:
                rec["embedding"] = vecs[j].tolist()
            out.append(rec)
        return out

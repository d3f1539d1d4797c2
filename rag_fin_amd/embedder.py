"""Sentence embedder on the GPU: the in-process replacement for
`SentenceTransformer('all-MiniLM-L6-v2')` on the vector-RAG path.

Reference call sites mirrored:
  ctor            vector_rag_mcp/main.py:41, retrieve.py:14, "chunking_storing (1).py":8
  .encode([q])    vector_rag_mcp/main.py:50, retrieve.py:27  -> np.float32 [1, 384]
  .encode(texts)  "chunking_storing (1).py":379-380          -> np.float32 [n, 384]

The reference loads the model by NAME (a network fetch); here the checkpoint comes
from a LOCAL directory in the Hugging Face layout (config.json, vocab.txt,
model.safetensors), read with safetensors only.  The forward pass runs in
libragfin_hip.so (rf_encode); there is no CPU path.
"""
from __future__ import annotations

import ctypes
import json
import os
import threading
from ctypes import c_void_p

import numpy as np

from . import _lib
from .store import require_gpu
from .tokenizer import WordPieceTokenizer

MINILM_L6 = dict(vocab_size=30522, hidden=384, layers=6, heads=12, intermediate=1536,
                 max_position=512, type_vocab=2, ln_eps=1e-12)


def _torch():
    import torch
    return torch


def stack_hf_state_dict(sd: dict, cfg: dict) -> dict:
    """Hugging Face BertModel tensor names -> the stacked layout of rf_encoder_weights."""
    def get(name):
        for prefix in ("", "bert.", "0.auto_model."):
            if prefix + name in sd:
                return np.asarray(sd[prefix + name], dtype=np.float32)
        raise KeyError(f"checkpoint lacks {name}")
    L = cfg["layers"]
    lay = lambda l, n: get(f"encoder.layer.{l}.{n}")
    return {
        "word_emb": get("embeddings.word_embeddings.weight"),
        "pos_emb": get("embeddings.position_embeddings.weight"),
        "type_emb": get("embeddings.token_type_embeddings.weight"),
        "emb_ln_g": get("embeddings.LayerNorm.weight"), "emb_ln_b": get("embeddings.LayerNorm.bias"),
        "qkv_w": np.stack([np.concatenate([lay(l, "attention.self.query.weight"),
                                           lay(l, "attention.self.key.weight"),
                                           lay(l, "attention.self.value.weight")]) for l in range(L)]),
        "qkv_b": np.stack([np.concatenate([lay(l, "attention.self.query.bias"),
                                           lay(l, "attention.self.key.bias"),
                                           lay(l, "attention.self.value.bias")]) for l in range(L)]),
        "ao_w": np.stack([lay(l, "attention.output.dense.weight") for l in range(L)]),
        "ao_b": np.stack([lay(l, "attention.output.dense.bias") for l in range(L)]),
        "ln1_g": np.stack([lay(l, "attention.output.LayerNorm.weight") for l in range(L)]),
        "ln1_b": np.stack([lay(l, "attention.output.LayerNorm.bias") for l in range(L)]),
        "ff1_w": np.stack([lay(l, "intermediate.dense.weight") for l in range(L)]),
        "ff1_b": np.stack([lay(l, "intermediate.dense.bias") for l in range(L)]),
        "ff2_w": np.stack([lay(l, "output.dense.weight") for l in range(L)]),
        "ff2_b": np.stack([lay(l, "output.dense.bias") for l in range(L)]),
        "ln2_g": np.stack([lay(l, "output.LayerNorm.weight") for l in range(L)]),
        "ln2_b": np.stack([lay(l, "output.LayerNorm.bias") for l in range(L)]),
    }


class Embedder:
    """`encode(list[str]) -> np.float32 [n, 384]`, unit-norm rows."""

    def __init__(self, weights: dict, cfg: dict | None = None, tokenizer: WordPieceTokenizer | None = None,
                 device=None, max_seq_length: int = 256):
        torch = _torch()
        self.cfg = dict(cfg or MINILM_L6)
        self.device = require_gpu(device)
        self.lib = _lib.load_library()
        self.tokenizer = tokenizer
        self.max_seq_length = min(max_seq_length, self.cfg["max_position"])
        self.dim = self.cfg["hidden"]
        c = self.cfg
        self._cfg_c = _lib.EncoderConfig(c["vocab_size"], c["hidden"], c["layers"], c["heads"],
                                         c["intermediate"], c["max_position"], c["type_vocab"],
                                         c["ln_eps"])
        nbytes = self.lib.rf_encoder_storage_bytes(ctypes.byref(self._cfg_c))
        if nbytes == 0:
            raise _lib.RagfinError(-2, f"encoder config not supported by the HIP kernels: {c}")
        # fp16 device copies of every tensor (kept alive: the library keeps pointers)
        self._w = {}
        wc = _lib.EncoderWeights()
        for name in _lib.ENCODER_WEIGHT_FIELDS:
            t = torch.as_tensor(np.ascontiguousarray(weights[name])).to(torch.float16)
            t = t.to(self.device).contiguous()
            self._w[name] = t
            setattr(wc, name, t.data_ptr())
        with torch.cuda.device(self.device):
            self._storage = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
            h = c_void_p()
            _lib.check(self.lib.rf_encoder_create(ctypes.byref(h), ctypes.byref(self._cfg_c),
                                                  ctypes.byref(wc), c_void_p(self._storage.data_ptr()),
                                                  nbytes, self.device.index, _lib.current_stream_ptr()))
        self.handle = h
        # Threading: the handle holds no per-call state, so concurrent encodes only
        # have to keep their BUFFERS apart.  Large batches take a workspace of their own per call
        # (torch's caching allocator: stream-ordered reuse, no hipMalloc in steady state).
        # Query-sized batches share fixed staging buffers (rf_encode replays a hipGraph keyed by
        # their addresses) and serialise on _small_lock; _small_done orders a caller on another
        # stream behind the previous use.  The chunked text ingest (pinned staging, copy stream)
        # runs one at a time under _ingest_lock.
        self._small_ws = None
        self._small_lock = threading.Lock()
        self._small_done = None
        self._ingest_lock = threading.Lock()
        self._copy_stream = None
        self._pin = None

    def __del__(self):
        h = getattr(self, "handle", None)
        if h:
            self.lib.rf_encoder_destroy(h)
            self.handle = None

    # -- construction -----------------------------------------------------------------
    @classmethod
    def from_local(cls, path: str, device=None, max_seq_length: int | None = None) -> "Embedder":
        """Load config.json + vocab.txt + model.safetensors from a local directory
        (e.g. a copy of sentence-transformers/all-MiniLM-L6-v2).  Nothing is downloaded."""
        from safetensors.numpy import load_file
        with open(os.path.join(path, "config.json")) as f:
            hc = json.load(f)
        cfg = dict(vocab_size=hc["vocab_size"], hidden=hc["hidden_size"], layers=hc["num_hidden_layers"],
                   heads=hc["num_attention_heads"], intermediate=hc["intermediate_size"],
                   max_position=hc["max_position_embeddings"], type_vocab=hc.get("type_vocab_size", 2),
                   ln_eps=hc.get("layer_norm_eps", 1e-12))
        if hc.get("hidden_act", "gelu") != "gelu":
            raise _lib.RagfinError(-2, f"activation {hc.get('hidden_act')} is not supported (gelu only)")
        sd = load_file(os.path.join(path, "model.safetensors"))
        tok = WordPieceTokenizer.from_vocab_file(os.path.join(path, "vocab.txt"))
        msl = max_seq_length
        sbert_cfg = os.path.join(path, "sentence_bert_config.json")
        if msl is None and os.path.exists(sbert_cfg):
            with open(sbert_cfg) as f:
                msl = json.load(f).get("max_seq_length")
        return cls(stack_hf_state_dict(sd, cfg), cfg, tok, device, msl or 256)

    @classmethod
    def from_random(cls, cfg: dict | None = None, seed: int = 0, tokenizer=None, device=None,
                    scale: float = 0.05) -> "Embedder":
        """Seeded random weights of the named architecture (no checkpoint exists
        offline): same generator as the test oracle so both sides can be rebuilt
        from the seed alone."""
        cfg = dict(cfg or MINILM_L6)
        rng = np.random.default_rng(seed)
        H, L, I = cfg["hidden"], cfg["layers"], cfg["intermediate"]

        def mat(*shape, s=scale):
            return (rng.standard_normal(shape, dtype=np.float32) * s).astype(np.float32)
        w = {"word_emb": mat(cfg["vocab_size"], H), "pos_emb": mat(cfg["max_position"], H),
             "type_emb": mat(cfg["type_vocab"], H), "emb_ln_g": 1 + mat(H, s=0.1), "emb_ln_b": mat(H, s=0.1),
             "qkv_w": mat(L, 3 * H, H), "qkv_b": mat(L, 3 * H, s=0.02), "ao_w": mat(L, H, H),
             "ao_b": mat(L, H, s=0.02), "ln1_g": 1 + mat(L, H, s=0.1), "ln1_b": mat(L, H, s=0.1),
             "ff1_w": mat(L, I, H), "ff1_b": mat(L, I, s=0.02), "ff2_w": mat(L, H, I),
             "ff2_b": mat(L, H, s=0.02), "ln2_g": 1 + mat(L, H, s=0.1), "ln2_b": mat(L, H, s=0.1)}
        return cls(w, cfg, tokenizer, device)

    # -- forward ------------------------------------------------------------------------
    def encode_ids(self, ids, lens, out_dtype="float16"):
        """ids int32 [B, T], lens int32 [B] (tensors or arrays) -> [B, 384] tensor on
        the device (fp16 by default: exactly what CorpusStore stores)."""
        torch = _torch()
        ids = torch.as_tensor(ids, dtype=torch.int32)
        lens = torch.as_tensor(lens, dtype=torch.int32)
        if ids.dim() != 2 or lens.shape != (ids.shape[0],):
            raise ValueError("encode_ids expects ids [B, T] and lens [B]")
        B, T = ids.shape
        if T > self.cfg["max_position"]:
            raise ValueError(f"T={T} exceeds max_position {self.cfg['max_position']}")
        want32 = out_dtype in ("float32", np.float32)
        small = B * T <= self.SMALL_SLOTS
        if small:
            with self._small_lock:   # the staging buffers are shared: enqueue copy + forward + clone as one unit
                return self._encode_small(ids, lens, B, T, want32)
        return self._encode_large(ids, lens, B, T, want32)

    def _launch(self, ids, lens, B, T, out16, out32, ws=None):
        torch = _torch()
        with torch.cuda.device(self.device):
            if ws is None:   # per call: concurrent encodes (any thread, any stream) never share scratch
                ws = torch.empty(self.lib.rf_encode_workspace_bytes(self.handle, B, T), dtype=torch.uint8,
                                 device=self.device)
            _lib.check(self.lib.rf_encode(self.handle, c_void_p(ids.data_ptr()), c_void_p(lens.data_ptr()),
                                          B, T, c_void_p(out16.data_ptr()) if out16 is not None else None,
                                          c_void_p(out32.data_ptr()) if out32 is not None else None,
                                          c_void_p(ws.data_ptr()), ws.numel(),
                                          _lib.current_stream_ptr()))

    def _encode_large(self, ids, lens, B, T, want32):
        torch = _torch()
        ids = ids.to(self.device).contiguous()
        lens = lens.to(self.device).contiguous()
        out16 = None if want32 else torch.empty((B, self.dim), dtype=torch.float16, device=self.device)
        out32 = torch.empty((B, self.dim), dtype=torch.float32, device=self.device) if want32 else None
        self._launch(ids, lens, B, T, out16, out32)
        return out32 if want32 else out16

    def _encode_small(self, ids, lens, B, T, want32):
        # a query-sized batch: fixed staging buffers, so that rf_encode sees the same pointers call
        # after call and replays its cached hipGraph instead of ~45 launches (the result is cloned)
        torch = _torch()
        sb = self._small_buffers()
        cur = torch.cuda.current_stream(self.device)
        if self._small_done is not None:
            cur.wait_event(self._small_done)     # a previous caller on ANOTHER stream may still be using the buffers
        # rf_encode keys its cached hipGraphs on (B, T, buffers) and the tokenizer returns T = the longest row, so every
        # query length would be a key of its own: round the width up (padding changes no bit:
        # tests/test_encoder_gpu.py::test_padding_content_and_width_are_ignored) -- to a multiple of 8 up to 32 tokens
        # (a lone query is 5-20 tokens: its forward should not grow to 32 rows), of 32 beyond: at most 11 widths per
        # batch size.  Host ids are padded on the host (one upload, no extra launch); device ids with two small kernels.
        Tr = (T + 7) // 8 * 8 if T <= 32 else (T + 31) // 32 * 32
        Tr = min(Tr, self.cfg["max_position"])
        if B * Tr > self.SMALL_SLOTS:
            Tr = T
        if ids.device.type == "cpu" and lens.device.type == "cpu":
            # Host inputs go through PINNED staging buffers owned by the embedder: an asynchronous copy from pageable
            # memory reads its source when the copy RUNS, and the caller's arrays (or a padded temporary made here) can
            # be freed and reused before that -- seen as a rare wrong hit when two ranks share one card
            # (tests/test_sharded_gpu.py::test_world2_sharded_store_behind_vector_rag).  The buffers are rewritten only
            # after the previous use's copy has run (host wait on its event: already complete in a serving loop).
            if self._small_done is not None:
                self._small_done.synchronize()
            hid = sb["ids_host"][:B * Tr].view(B, Tr)
            hid[:, :T] = ids
            if Tr != T:
                hid[:, T:] = 0
            sb["lens_host"][:B] = lens
            sb["ids"][:B * Tr].copy_(sb["ids_host"][:B * Tr], non_blocking=True)
            sb["lens"][:B].copy_(sb["lens_host"][:B], non_blocking=True)
        else:
            ids = ids.to(self.device, non_blocking=False)
            lens = lens.to(self.device, non_blocking=False)
            wide = sb["ids"][:B * Tr].view(B, Tr)
            wide[:, :T].copy_(ids)
            if Tr != T:
                wide[:, T:].zero_()
            sb["lens"][:B].copy_(lens)
        T = Tr
        ids, lens = sb["ids"], sb["lens"]
        out16 = None if want32 else sb["o16"]
        out32 = sb["o32"] if want32 else None
        self._launch(ids, lens, B, T, out16, out32, ws=self._small_ws)
        res = (out32 if want32 else out16)[:B].clone()
        if self._small_done is None:
            self._small_done = torch.cuda.Event()
        self._small_done.record(cur)
        return res

    SMALL_SLOTS = 1024   # = SM_MAX_TOK of csrc/encoder.hip: the small-batch GEMM path / hipGraph replay

    def _small_buffers(self):
        torch = _torch()
        if getattr(self, "_sb", None) is None:
            n = self.SMALL_SLOTS
            self._sb = {"ids": torch.zeros(n, dtype=torch.int32, device=self.device),
                        "lens": torch.zeros(n, dtype=torch.int32, device=self.device),
                        "ids_host": torch.zeros(n, dtype=torch.int32).pin_memory(),
                        "lens_host": torch.zeros(n, dtype=torch.int32).pin_memory(),
                        "o16": torch.zeros((n, self.dim), dtype=torch.float16, device=self.device),
                        "o32": torch.zeros((n, self.dim), dtype=torch.float32, device=self.device)}
            # workspace large enough for every small shape: its pointer must not move either
            nbytes = max(self.lib.rf_encode_workspace_bytes(self.handle, n, 1),
                         self.lib.rf_encode_workspace_bytes(self.handle, max(1, n // 256), 256))
            self._small_ws = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        return self._sb

    def encode_to_device(self, sentences, batch_tokens: int = 65536, out_dtype="float16"):
        """Tokenise, bucket by length, encode; returns fp16 (default: what CorpusStore keeps) or
        fp32 [n, 384] on the device in the input order."""
        held = [False]    # the chunked path takes _ingest_lock on first use (queries never do)
        try:
            return self._encode_to_device(sentences, batch_tokens, out_dtype, held)
        finally:
            if held[0]:
                self._ingest_lock.release()

    def _encode_to_device(self, sentences, batch_tokens, out_dtype, held):
        torch = _torch()
        want32 = out_dtype in ("float32", np.float32)
        odt = torch.float32 if want32 else torch.float16
        if self.tokenizer is None:
            raise RuntimeError("Embedder has no tokenizer: construct it with from_local(path) or pass "
                               "tokenizer=WordPieceTokenizer(vocab)")
        if isinstance(sentences, str):
            sentences = [sentences]
        # text -> ids through the native tokenizer (csrc/tokenizer.cpp, one thread per host core);
        # a tokenizer object without batch_native (a test double) takes the per-sentence path.
        # Large inputs go in chunks of `chunk_texts`: rf_encode only ENQUEUES, so while the GPU works
        # through one chunk's buckets the host is already tokenising the next chunk.
        import time
        sentences = list(sentences)
        n = len(sentences)
        out = torch.empty((n, self.dim), dtype=odt, device=self.device)
        if n == 0:
            return out
        # host seconds per stage of the last call (bench.py reports them beside from_text): the GPU work is only
        # ENQUEUED below, so whatever the host spends here in series is time the card may sit idle
        stats = self.ingest_stats = {"prepass_s": 0.0, "native_tokenize_s": 0.0, "sort_s": 0.0, "pinned_wait_s": 0.0, "pinned_stage_s": 0.0,
                                     "upload_enqueue_s": 0.0, "bucket_and_launch_s": 0.0, "chunks": 0, "buckets": 0}

        def lap(key, t0):
            t1 = time.perf_counter()
            stats[key] += t1 - t0
            return t1
        chunk_texts = int(os.environ.get("RAGFIN_INGEST_CHUNK_TEXTS", "4096"))   # (tools/ingest_probe.py sweeps it)
        # the first chunk is a quarter of the others: nothing overlaps ITS tokenisation, so the
        # GPU should get its first buckets early
        starts = [0] + list(range(min(n, chunk_texts // 4), n, chunk_texts))
        for ci, c0 in enumerate(starts):
            part = sentences[c0:(starts[ci + 1] if ci + 1 < len(starts) else n)]
            stats["chunks"] += 1
            tm = time.perf_counter()
            if hasattr(self.tokenizer, "batch_native"):
                all_ids, all_lens = self.tokenizer.batch_native(part, self.max_seq_length)
                lt = getattr(self.tokenizer, "last_timing", None)
                if lt:
                    stats["prepass_s"] += lt["prepass_s"]
                    stats["native_tokenize_s"] += lt["native_s"]
            else:
                rows = [self.tokenizer.encode(s, self.max_seq_length) for s in part]
                all_lens = np.array([len(r) for r in rows], dtype=np.int32)
                all_ids = np.full((len(rows), max(int(all_lens.max()), 1) if len(rows) else 1),
                                  self.tokenizer.pad_id, dtype=np.int32)
                for r, row in enumerate(rows):
                    all_ids[r, :len(row)] = row
            m = len(part)
            if m * int(all_ids.shape[1]) <= self.SMALL_SLOTS:
                # a query or a handful: straight to the small-batch path (no sorting, no side stream)
                out[c0:c0 + m] = self.encode_ids(all_ids, all_lens, out_dtype=out_dtype)
                continue
            if not held[0]:      # pinned staging buffers + copy stream below are one-at-a-time
                self._ingest_lock.acquire()
                held[0] = True
            tm = time.perf_counter()
            order = np.argsort(all_lens, kind="stable")
            sorted_lens = all_lens[order].astype(np.int64)
            tm = lap("sort_s", tm)
            # ONE upload per chunk, from pinned memory on a side stream: a pageable host-to-device copy
            # on the compute stream would make the host wait for every kernel already queued there, and
            # the chunk-to-chunk overlap would be gone.  Buckets are then gathered ON the device.
            if self._copy_stream is None:
                self._copy_stream = torch.cuda.Stream(device=self.device)
            # two sets of pinned staging buffers used alternately, so that chunk k+1 can be staged
            # while chunk k's copy is still in flight (allocating pinned memory per chunk would
            # synchronise the device each time)
            Tc = int(all_ids.shape[1])
            if self._pin is None:
                cap = chunk_texts * self.max_seq_length
                self._pin = [dict(ids=torch.empty(cap, dtype=torch.int32).pin_memory(),
                                  lens=torch.empty(chunk_texts, dtype=torch.int32).pin_memory(),
                                  order=torch.empty(chunk_texts, dtype=torch.int64).pin_memory(),
                                  done=None) for _ in range(2)]
            pb = self._pin[ci & 1]
            if pb["done"] is not None:
                pb["done"].synchronize()      # its previous upload has left the buffers
            tm = lap("pinned_wait_s", tm)
            # numpy views of the pinned buffers: a plain memcpy on this thread (torch's CPU copy_ fans a 4 MB copy
            # out over every OpenMP thread of the host, which a container's CPU quota punishes: hostcpu.py)
            pb["ids"].numpy()[:m * Tc].reshape(m, Tc)[...] = all_ids
            pb["lens"].numpy()[:m] = all_lens
            pb["order"].numpy()[:m] = order
            tm = lap("pinned_stage_s", tm)
            with torch.cuda.stream(self._copy_stream):
                ids_dev = pb["ids"][:m * Tc].view(m, Tc).to(self.device, non_blocking=True)
                lens_dev = pb["lens"][:m].to(self.device, non_blocking=True)
                order_dev = pb["order"][:m].to(self.device, non_blocking=True)
                ready = torch.cuda.Event()
                ready.record()
            pb["done"] = ready
            torch.cuda.current_stream(self.device).wait_event(ready)
            for t in (ids_dev, lens_dev, order_dev):
                t.record_stream(torch.cuda.current_stream(self.device))
            tm = lap("upload_enqueue_s", tm)
            i = 0
            while i < m:
                stats["buckets"] += 1
                # rows are sorted ascending, so the last row of a bucket sets its width and
                # (rows in bucket) x (that width) is non-decreasing in the bucket's end: binary search
                cost = (np.arange(1, m - i + 1, dtype=np.int64)) * sorted_lens[i:]
                j = i + max(1, int(np.searchsorted(cost, batch_tokens, side="right")))
                T = int(sorted_lens[j - 1])
                sel = order_dev[i:j]
                ids = ids_dev.index_select(0, sel)[:, :T].contiguous()
                lens = lens_dev.index_select(0, sel)
                out.index_copy_(0, sel + c0, self.encode_ids(ids, lens, out_dtype=out_dtype))
                i = j
            lap("bucket_and_launch_s", tm)
        return out

    def encode(self, sentences, batch_size: int = 32, **_ignored) -> np.ndarray:
        """sentence-transformers' signature: numpy float32 [n, 384] (or [384] for a str) --
        rf_encode's fp32 output (the reference returns fp32, vector_rag_mcp/main.py:50), not the
        fp16 rows the store keeps."""
        single = isinstance(sentences, str)
        emb = self.encode_to_device([sentences] if single else list(sentences), out_dtype="float32")
        arr = emb.cpu().numpy()
        return arr[0] if single else arr


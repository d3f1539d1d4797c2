"""Wiring: build a VectorRAG from local files / environment, and the ingest step.

Environment (dotenv-style names):
  RAGFIN_MODEL_DIR     local all-MiniLM-L6-v2 directory (config.json, vocab.txt,
                       model.safetensors).  Required: nothing is fetched by name.
  RAGFIN_DATA_DIR      folder with icici_q{1..4}_2023/*.json (default: extract_data)
  RAGFIN_DEVICE        torch device string (default cuda:0)
  WORLD_SIZE / RANK / LOCAL_RANK (torch.distributed.run): with WORLD_SIZE > 1 the corpus is
                       row-sharded over the ranks' GPUs (build_sharded_rag); rank 0 serves, the
                       other ranks follow it
  MILVUS_COLLECTION    collection name reported by the tools (default fin_chunks)
"""
from __future__ import annotations

import os

from . import chunker


def ingest(store, embedder, chunks) -> int:
    """The reference's encode + insert + flush + load
    ("chunking_storing (1).py":377-397) on the GPU: texts are embedded by rf_encode
    and the fp16 rows go straight into the HBM corpus (no host round trip)."""
    if not chunks:
        return 0
    emb = embedder.encode_to_device([c["text"] for c in chunks])
    n = store.insert(chunker.insert_columns(chunks, emb))
    store.flush()
    store.load()
    return n


def build_rag(model_dir: str, data_dir: str = "extract_data", device=None,
              collection_name: str = "fin_chunks", generator=None):
    from .embedder import Embedder
    from .rag import VectorRAG
    from .store import CorpusStore
    embedder = Embedder.from_local(model_dir, device=device)
    store = CorpusStore(collection_name, dim=embedder.dim, device=device)
    ingest(store, embedder, chunker.build_all_chunks(data_dir))
    return VectorRAG(None, collection_name, embedder=embedder, store=store, generator=generator)


def ingest_sharded(store, embedder, chunks) -> int:
    """COLLECTIVE form of ingest(): every rank holds the whole chunk list (text work is cheap),
    embeds only ITS slice (encoder replicas) and keeps those rows in its HBM."""
    from .sharded import ShardedSearcher
    if not chunks:
        return 0
    lo, hi = ShardedSearcher.shard_bounds(len(chunks), store.world, store.rank)
    emb = embedder.encode_to_device([c["text"] for c in chunks[lo:hi]])
    cols = chunker.insert_columns(chunks, None)
    n = store.add(cols[0], cols[1], emb, cols[3], cols[4], cols[5], cols[6], local=True)
    store.flush()
    store.load()
    return n


def build_sharded_rag(model_dir: str, data_dir: str = "extract_data", local_rank: int = 0,
                      collection_name: str = "fin_chunks", generator=None, backend: str = "nccl"):
    """One process per GPU (launched by torch.distributed.run): every rank builds an embedder
    replica and its shard of the store.  Returns the VectorRAG on EVERY rank; a serving
    deployment then calls `rag.collection.start_workers()` on all ranks -- rank 0 comes back and
    serves the MCP / REST surface, the others stay inside answering its searches."""
    import torch
    import torch.distributed as dist
    from .embedder import Embedder
    from .rag import VectorRAG
    from .sharded_store import ShardedCorpusStore
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if not dist.is_initialized():
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group(backend, device_id=dev if backend == "nccl" else None)
    embedder = Embedder.from_local(model_dir, device=dev)
    store = ShardedCorpusStore(collection_name, dim=embedder.dim, device=dev)
    ingest_sharded(store, embedder, chunker.build_all_chunks(data_dir))
    return VectorRAG(None, collection_name, embedder=embedder, store=store, generator=generator)


def build_rag_from_env():
    model_dir = os.getenv("RAGFIN_MODEL_DIR")
    if not model_dir:
        raise RuntimeError("RAGFIN_MODEL_DIR is not set: point it at a local all-MiniLM-L6-v2 "
                           "directory (the reference fetches the model by name; this build never "
                           "touches the network)")
    if int(os.getenv("WORLD_SIZE", "1")) > 1:
        return build_sharded_rag(model_dir, os.getenv("RAGFIN_DATA_DIR", "extract_data"),
                                 int(os.getenv("LOCAL_RANK", "0")), os.getenv("MILVUS_COLLECTION", "fin_chunks"))
    return build_rag(model_dir, os.getenv("RAGFIN_DATA_DIR", "extract_data"),
                     os.getenv("RAGFIN_DEVICE", "cuda:0"), os.getenv("MILVUS_COLLECTION", "fin_chunks"))

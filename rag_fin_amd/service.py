"""Wiring: build a VectorRAG from local files / environment, and the ingest step.

Environment (dotenv-style names):
  RAGFIN_MODEL_DIR     local all-MiniLM-L6-v2 directory (config.json, vocab.txt,
                       model.safetensors).  Required: nothing is fetched by name.
  RAGFIN_DATA_DIR      folder with icici_q{1..4}_2023/*.json (default: extract_data)
  RAGFIN_DEVICE        torch device string (default cuda:0)
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


def build_rag_from_env():
    model_dir = os.getenv("RAGFIN_MODEL_DIR")
    if not model_dir:
        raise RuntimeError("RAGFIN_MODEL_DIR is not set: point it at a local all-MiniLM-L6-v2 "
                           "directory (the reference fetches the model by name; this build never "
                           "touches the network)")
    return build_rag(model_dir, os.getenv("RAGFIN_DATA_DIR", "extract_data"),
                     os.getenv("RAGFIN_DEVICE", "cuda:0"), os.getenv("MILVUS_COLLECTION", "fin_chunks"))

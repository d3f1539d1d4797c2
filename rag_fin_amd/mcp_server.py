"""MCP tool surface of the vector-RAG server (port 9006), same four tools, parameter
names, defaults and payload keys as vector_rag_mcp/main.py:127-178.

The tool bodies are plain functions over a lazily built `rag` so they can be called
(and tested) without FastMCP; `main()` registers them on a FastMCP("VectorRAG")
server when the `fastmcp` package is installed.
"""
from __future__ import annotations

import logging
import os
import sys

logger = logging.getLogger(__name__)

COLLECTION_NAME = os.getenv("MILVUS_COLLECTION", "fin_chunks")
# kept in the stats payload for drop-in compatibility; the store is in-process
MILVUS_HOST = os.getenv("MILVUS_HOST", "in-process")
MILVUS_PORT = os.getenv("MILVUS_PORT", "hbm")

_rag = None


def set_rag(rag) -> None:
    global _rag
    _rag = rag


def get_rag():
    global _rag
    if _rag is None:
        from .service import build_rag_from_env
        _rag = build_rag_from_env()
    return _rag


def health_check():
    """Check Vector RAG system health"""
    return get_rag().health_check()


_batcher = None


def _searcher():
    """RAGFIN_MICROBATCH_MS > 0: coalesce concurrent tool calls into one GPU batch
    (rag_fin_amd.batching); default: one search per call, like the reference."""
    global _batcher
    ms = float(os.getenv("RAGFIN_MICROBATCH_MS", "0") or 0)
    if ms <= 0:
        return get_rag()
    if _batcher is None or _batcher.rag is not get_rag():
        from .batching import MicroBatcher
        _batcher = MicroBatcher(get_rag(), max_batch=64, max_wait_ms=ms)
    return _batcher


def search_vectors(query: str, top_k: int = 3):
    """Semantic search in vector store"""
    try:
        contexts = _searcher().search(query, top_k)
        return {"status": "success", "query": query, "results": contexts,
                "result_count": len(contexts)}
    except Exception as e:
        return {"status": "error", "message": str(e), "query": query}


def answer_question(question: str, top_k: int = 3):
    """Answer question using RAG"""
    try:
        result = get_rag().search_and_answer(question, top_k)
        return {"status": "success", "question": question, **result}
    except Exception as e:
        return {"status": "error", "message": str(e), "question": question}


def get_collection_stats():
    """Get collection statistics"""
    try:
        return {"status": "success", "collection_name": COLLECTION_NAME,
                "total_chunks": get_rag().collection.num_entities,
                "milvus_host": MILVUS_HOST, "milvus_port": MILVUS_PORT}
    except Exception as e:
        return {"status": "error", "message": str(e)}


TOOLS = (health_check, search_vectors, answer_question, get_collection_stats)


def main() -> None:
    logging.basicConfig(level=logging.INFO, format="%(asctime)s %(levelname)s: %(message)s",
                        handlers=[logging.StreamHandler(sys.stdout)])
    os.environ.setdefault("PORT", "9006")
    try:
        from fastmcp import FastMCP
    except ImportError as e:
        raise SystemExit("the MCP transport needs the `fastmcp` package (not installed here); the "
                         "tool functions in rag_fin_amd.mcp_server work without it") from e
    # N > 1 (torch.distributed.run, one process per GPU): every rank builds its shard; ranks > 0
    # then follow rank 0's searches and never reach the MCP transport
    rag = get_rag()
    if hasattr(rag.collection, "start_workers"):
        rag.collection.start_workers()
        if rag.collection.rank != 0:
            return
    mcp = FastMCP("VectorRAG")
    for fn in TOOLS:
        mcp.tool()(fn)
    get_rag()
    logger.info("Starting Vector RAG MCP Server on port %s (collection %s)", os.environ["PORT"],
                COLLECTION_NAME)
    try:
        mcp.run(transport="streamable-http")
    finally:   # release the ranks parked in start_workers(), however the server ends
        if hasattr(rag.collection, "stop_workers"):
            rag.collection.stop_workers()


if __name__ == "__main__":
    main()

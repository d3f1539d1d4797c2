"""`VectorRAG`: the object the reference's MCP server is built around
(vector_rag_mcp/main.py:36-123; CLI twin `SimpleRAG`, retrieve.py:6-82), with the
same method names, defaults and return keys, backed by the in-process GPU
embedder + corpus store instead of sentence-transformers + a Milvus server.

The LLM generation step (Gemini, main.py:40,97) is outside this build's scope: a
`generator` callable can be plugged in; without one `search_and_answer` returns the
reference's own failure shape {"error", "contexts", "context_count"}.
"""
from __future__ import annotations

import logging
import time
from typing import Callable, Sequence

logger = logging.getLogger(__name__)

OUTPUT_FIELDS = ["text", "period", "chunk_type", "statement_type", "primary_value"]

# The LLM step (Gemini, vector_rag_mcp/main.py:72-108) is out of scope; the prompt wording is
# the caller's business.  This default only lays the retrieved contexts out for a generator;
# deployments that want the reference's wording pass their own `prompt_template`
# (placeholders: {question}, {context}).
DEFAULT_PROMPT = "Question: {question}\n\nRetrieved contexts:\n{context}\n\nAnswer from the contexts only."


class VectorRAG:
    def __init__(self, gemini_api_key: str | None = None, collection_name: str = "fin_chunks", *,
                 embedder=None, store=None, generator: Callable[[str], str] | None = None,
                 llm_delay_s: float = 1.0, prompt_template: str = DEFAULT_PROMPT):
        """embedder: .encode(list[str]) -> [n, dim] (rag_fin_amd.embedder.Embedder);
        store: rag_fin_amd.store.CorpusStore.  `gemini_api_key` is accepted for
        signature compatibility and only handed to `generator` factories upstream."""
        if embedder is None or store is None:
            raise ValueError("VectorRAG needs an embedder and a corpus store (see rag_fin_amd.service."
                             "build_rag); there is no remote Milvus/sentence-transformers fallback")
        self.similarity_model = embedder
        self.collection = store
        self.collection_name = collection_name
        self.generator = generator
        self.prompt_template = prompt_template
        self.llm_delay_s = llm_delay_s
        self.collection.load()
        logger.info("VectorRAG ready on collection %s (%d chunks)", collection_name,
                    self.collection.num_entities)

    # -- retrieval ---------------------------------------------------------------------
    @staticmethod
    def _contexts(hits) -> list[dict]:
        return [{"rank": i + 1, "text": h.entity.text, "period": h.entity.period,
                 "chunk_type": h.entity.chunk_type, "statement_type": h.entity.statement_type,
                 "primary_value": h.entity.primary_value, "score": float(h.score)}
                for i, h in enumerate(hits)]

    def search(self, query: str, top_k: int = 3) -> list[dict]:
        """Ranked context dicts, keys exactly as vector_rag_mcp/main.py:59-70."""
        q = self._embed([query])
        results = self.collection.search(q, "embedding", {"metric_type": "COSINE"}, top_k,
                                         output_fields=OUTPUT_FIELDS)
        return self._contexts(results[0])

    def _embed(self, texts):
        """`.encode(texts)` of the reference (main.py:50) -- kept on the device when the embedder
        offers it: the unit-norm fp16 rows rf_encode writes are exactly what the store searches
        with, so the download / re-upload / re-normalise round trip of the numpy form is skipped."""
        to_dev = getattr(self.similarity_model, "encode_to_device", None)
        return to_dev(texts) if to_dev is not None else self.similarity_model.encode(texts)

    retrieve = search  # BASELINE.json's "retrieve(query, k)" name for the same call

    def search_batch(self, queries: Sequence[str], top_k: int = 3) -> list[list[dict]]:
        """Many queries in one embed + one corpus sweep per 64 (new: the reference
        is strictly one query per call)."""
        if not queries:
            return []
        q = self._embed(list(queries))
        results = self.collection.search(q, "embedding", {"metric_type": "COSINE"}, top_k,
                                         output_fields=OUTPUT_FIELDS)
        return [self._contexts(r) for r in results]

    # -- generation (out of scope; interface kept) -----------------------------------------
    def build_prompt(self, question: str, contexts: list[dict]) -> str:
        ctx = "\n\n".join(f"Context {i + 1} [{c['period']} - {c['chunk_type']}]:\n{c['text']}"
                          for i, c in enumerate(contexts))
        return self.prompt_template.format(question=question, context=ctx)

    def search_and_answer(self, question: str, top_k: int = 3) -> dict:
        contexts = self.search(question, top_k)
        prompt = self.build_prompt(question, contexts)
        try:
            if self.generator is None:
                raise RuntimeError("no LLM generator configured (generation is outside this build)")
            if self.llm_delay_s:
                time.sleep(self.llm_delay_s)
            answer = self.generator(prompt)
            return {"answer": str(answer).strip(), "contexts": contexts, "context_count": len(contexts)}
        except Exception as e:  # the reference's blanket handler (main.py:103-108)
            return {"error": str(e), "contexts": contexts, "context_count": len(contexts)}

    def health_check(self) -> dict:
        try:
            n = self.collection.num_entities
            llm = "available" if self.generator is not None else "not configured"
            return {"status": "healthy", "milvus": "in-process MI355X store", "gemini": llm,
                    "collection": self.collection_name, "total_chunks": n}
        except Exception as e:
            return {"status": "unhealthy", "error": str(e)}

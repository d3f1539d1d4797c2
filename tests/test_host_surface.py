"""Host-side surface (no GPU): VectorRAG marshalling, MCP tool payloads and error
dicts, REST adapter bounds/error mapping.  Fakes stand in for the GPU embedder and
store so that only the host logic is under test -- the product classes themselves
refuse to run without the HIP library and a GPU (tests/test_abi.py)."""
import asyncio
import json

import numpy as np
import pytest

from rag_fin_amd import mcp_server
from rag_fin_amd.rag import OUTPUT_FIELDS, VectorRAG
from rag_fin_amd.store import Hit


class FakeEmbedder:
    dim = 4

    def encode(self, texts):
        return np.stack([np.full(4, float(len(t))) for t in texts]).astype(np.float32)


class FakeStore:
    def __init__(self, n=5):
        self.rows = [dict(id=f"id{i}", text=f"text {i}", period=f"Q{i % 4 + 1}_FY2024", chunk_type="t",
                          statement_type="consolidated", primary_value=float(i)) for i in range(n)]
        self.calls = []

    num_entities = property(lambda self: len(self.rows))

    def load(self):
        pass

    def search(self, data, anns_field, param, limit, expr=None, output_fields=None):
        self.calls.append((np.asarray(data).shape, anns_field, param, limit, tuple(output_fields)))
        out = []
        for _ in range(np.asarray(data).shape[0]):
            hits = [Hit(i, r["id"], 1.0 - 0.1 * i, {f: r[f] for f in output_fields})
                    for i, r in enumerate(self.rows[:limit])]
            out.append(hits)
        return out


def make_rag(**kw):
    return VectorRAG("unused-key", "fin_chunks", embedder=FakeEmbedder(), store=FakeStore(), **kw)


def test_search_payload_matches_reference_schema():
    rag = make_rag()
    ctx = rag.search("net profit Q1", 3)
    assert [c["rank"] for c in ctx] == [1, 2, 3]
    assert list(ctx[0].keys()) == ["rank", "text", "period", "chunk_type", "statement_type",
                                   "primary_value", "score"]      # no "id" key (main.py:61-69)
    assert isinstance(ctx[0]["score"], float)
    shape, field, param, limit, fields = rag.collection.calls[0]
    assert shape == (1, 4) and field == "embedding" and param == {"metric_type": "COSINE"}
    assert limit == 3 and list(fields) == OUTPUT_FIELDS
    assert rag.retrieve("net profit Q1", 2) == rag.search("net profit Q1", 2)
    assert len(rag.search("q", 20)) == 5       # top_k > rows: fewer hits, never padded


def test_search_batch_and_default_top_k():
    rag = make_rag()
    out = rag.search_batch(["aaaa", "bbbbbb"])
    assert len(out) == 2 and all(len(o) == 3 for o in out)
    assert rag.search_batch([]) == []


def test_search_and_answer_shapes():
    rag = make_rag()
    r = rag.search_and_answer("what was net profit?", 2)
    assert set(r) == {"error", "contexts", "context_count"} and r["context_count"] == 2
    rag2 = make_rag(generator=lambda prompt: "  42 crore \n", llm_delay_s=0)
    r2 = rag2.search_and_answer("what was net profit?", 2)
    assert r2["answer"] == "42 crore" and set(r2) == {"answer", "contexts", "context_count"}
    prompt = rag2.build_prompt("Q?", rag2.search("Q?", 1))
    assert "Context 1 [Q1_FY2024 - t]:\ntext 0" in prompt and prompt.startswith("Question: Q?")
    # the wording is the caller's: a deployment passes its own template (placeholders {question}, {context})
    rag4 = make_rag(prompt_template="Q={question}|C={context}|ANSWER:")
    assert rag4.build_prompt("Q?", rag4.search("Q?", 1)).endswith("|ANSWER:")
    rag3 = make_rag(generator=lambda p: 1 / 0, llm_delay_s=0)
    assert "division" in rag3.search_and_answer("what was net profit?")["error"]


def test_health_check_payload():
    h = make_rag().health_check()
    assert h["status"] == "healthy" and h["total_chunks"] == 5 and h["collection"] == "fin_chunks"
    assert set(h) == {"status", "milvus", "gemini", "collection", "total_chunks"}


def test_vector_rag_refuses_to_build_without_backends():
    with pytest.raises(ValueError, match="no remote Milvus"):
        VectorRAG("k")


def test_mcp_tools_payloads_and_error_dicts():
    mcp_server.set_rag(make_rag())
    try:
        assert [f.__name__ for f in mcp_server.TOOLS] == ["health_check", "search_vectors",
                                                          "answer_question", "get_collection_stats"]
        r = mcp_server.search_vectors("net profit Q1")
        assert r["status"] == "success" and r["query"] == "net profit Q1" and r["result_count"] == 3
        assert set(r) == {"status", "query", "results", "result_count"}
        a = mcp_server.answer_question("what was net profit?", top_k=1)
        assert a["status"] == "success" and a["question"] == "what was net profit?" and "contexts" in a
        s = mcp_server.get_collection_stats()
        assert set(s) == {"status", "collection_name", "total_chunks", "milvus_host", "milvus_port"}
        assert s["total_chunks"] == 5
        assert mcp_server.health_check()["status"] == "healthy"

        class Broken:
            def search(self, *a):
                raise RuntimeError("boom")
            search_and_answer = search
            collection = property(lambda self: (_ for _ in ()).throw(RuntimeError("gone")))
        mcp_server.set_rag(Broken())
        assert mcp_server.search_vectors("qqqqq") == {"status": "error", "message": "boom", "query": "qqqqq"}
        assert mcp_server.answer_question("qqqqq") == {"status": "error", "message": "boom",
                                                       "question": "qqqqq"}
        assert mcp_server.get_collection_stats() == {"status": "error", "message": "gone"}
    finally:
        mcp_server.set_rag(None)


# ---- REST adapter ----------------------------------------------------------------------
def test_adapter_request_bounds():
    from pydantic import ValidationError
    from rag_fin_amd.adapter import AnswerRequest, SearchRequest
    assert SearchRequest(query="hello").top_k == 3
    assert SearchRequest(query="hello", top_k=20).top_k == 20
    for bad in (dict(query="hi"), dict(query="hello", top_k=0), dict(query="hello", top_k=21)):
        with pytest.raises(ValidationError):
            SearchRequest(**bad)
    with pytest.raises(ValidationError):
        AnswerRequest(question="hello", top_k=11)


def test_sse_parsing_first_result_wins():
    from fastapi import HTTPException
    from rag_fin_amd.adapter import first_sse_result
    lines = ["event: message", "data: not-json", 'data: {"jsonrpc":"2.0","id":1,"result":{"a":1}}',
             'data: {"result":{"a":2}}']
    assert first_sse_result(lines) == {"a": 1}
    assert first_sse_result(["event: ping"]) is None
    with pytest.raises(HTTPException) as e:
        first_sse_result(['data: {"error":{"code":-1}}'])
    assert e.value.status_code == 500


def _fake_mcp_transport(tool_result, status=200, session="sess-1"):
    import httpx
    seen = []

    def handler(request: httpx.Request):
        body = json.loads(request.content)
        seen.append((body.get("method"), request.headers.get("mcp-session-id"), body.get("id")))
        if body["method"] == "initialize":
            return httpx.Response(200, headers={"mcp-session-id": session}, json={"result": {}})
        if body["method"] == "notifications/initialized":
            return httpx.Response(202)
        payload = "event: message\ndata: " + json.dumps({"jsonrpc": "2.0", "id": 1, "result": tool_result}) + "\n\n"
        return httpx.Response(status, text=payload, headers={"content-type": "text/event-stream"})
    return httpx.MockTransport(handler), seen


def test_mcp_client_handshake_and_tool_call():
    import httpx
    from rag_fin_amd.adapter import MCPClient
    transport, seen = _fake_mcp_transport({"status": "success"})
    client = MCPClient("http://mcp.test/mcp", httpx.AsyncClient(transport=transport))

    async def go():
        a = await client.call_tool("search_vectors", {"query": "hello", "top_k": 3})
        b = await client.call_tool("get_collection_stats", {})
        return a, b
    a, b = asyncio.run(go())
    assert a == {"status": "success"} == b
    methods = [m for m, _, _ in seen]
    assert methods == ["initialize", "notifications/initialized", "tools/call", "tools/call"]
    assert seen[0][1] is None and seen[2][1] == "sess-1" and seen[2][2] == 1   # session kept, id == 1


def test_adapter_routes_and_error_mapping(monkeypatch):
    import httpx
    from fastapi.testclient import TestClient
    from rag_fin_amd import adapter
    transport, _ = _fake_mcp_transport({"status": "success", "results": []})
    monkeypatch.setattr(adapter, "mcp", adapter.MCPClient("http://mcp.test/mcp",
                                                          httpx.AsyncClient(transport=transport)))
    tc = TestClient(adapter.app)
    assert tc.get("/").json()["endpoints"] == {"health": "/health", "search": "/search",
                                               "answer": "/answer", "stats": "/stats"}
    assert tc.post("/search", json={"query": "net profit Q1", "top_k": 3}).json()["status"] == "success"
    assert tc.post("/search", json={"query": "np"}).status_code == 422
    assert tc.post("/answer", json={"question": "what about Q2?", "top_k": 11}).status_code == 422
    assert tc.get("/stats").status_code == 200
    assert tc.get("/health").json()["status"] == "healthy"
    bad, _ = _fake_mcp_transport({}, status=502)
    monkeypatch.setattr(adapter, "mcp", adapter.MCPClient("http://mcp.test/mcp",
                                                          httpx.AsyncClient(transport=bad)))
    tc = TestClient(adapter.app)
    assert tc.post("/search", json={"query": "net profit Q1"}).status_code == 503
    assert tc.get("/health").json() == {"status": "unhealthy", "mcp": "unavailable"}

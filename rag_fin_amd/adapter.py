"""REST -> MCP bridge for the vector-RAG server (port 9001 -> MCP 9006).

Same routes, request bounds and error mapping as adapters/vectorrag_adapter.py:
  GET /            service card                         (:121-132)
  GET /health      {"status": "healthy", "mcp": ...} | {"status": "unhealthy", "mcp": "unavailable"}
  POST /search     {query: str >= 5 chars, top_k: 1..20 = 3}   -> tool search_vectors
  POST /answer     {question: str >= 5 chars, top_k: 1..10 = 3} -> tool answer_question
  GET /stats       -> tool get_collection_stats
Transport: JSON-RPC 2.0 over MCP streamable HTTP; one session, initialised on
first use ("initialize" + "notifications/initialized"); tool calls are streamed
and the first SSE `data:` line carrying a "result" wins; non-200 -> 503, a
JSON-RPC "error" -> 500, a stream without result -> 500 (:91-111).
"""
from __future__ import annotations

import json
import logging
import os

import httpx
from fastapi import FastAPI, HTTPException
from pydantic import BaseModel, Field

logger = logging.getLogger(__name__)

MCP_URL = os.getenv("VECTOR_RAG_MCP_URL", "http://localhost:9006/mcp")
TIMEOUT = 300
PROTOCOL_VERSION = "2024-11-05"


class SearchRequest(BaseModel):
    query: str = Field(..., min_length=5)
    top_k: int = Field(default=3, ge=1, le=20)


class AnswerRequest(BaseModel):
    question: str = Field(..., min_length=5)
    top_k: int = Field(default=3, ge=1, le=10)


def first_sse_result(lines):
    """Scan SSE lines; return the first JSON-RPC result, raise on a JSON-RPC error."""
    for line in lines:
        if not line.startswith("data: "):
            continue
        try:
            msg = json.loads(line[len("data: "):])
        except json.JSONDecodeError:
            continue
        if "result" in msg:
            return msg["result"]
        if "error" in msg:
            raise HTTPException(500, f"Tool error: {msg['error']}")
    return None


class MCPClient:
    def __init__(self, url: str = MCP_URL, client: httpx.AsyncClient | None = None):
        self.url = url
        self.client = client or httpx.AsyncClient(timeout=TIMEOUT)
        self.session_id = None

    def _headers(self) -> dict:
        h = {"Content-Type": "application/json", "Accept": "application/json, text/event-stream"}
        if self.session_id:
            h["mcp-session-id"] = self.session_id
        return h

    @staticmethod
    def _rpc(method: str, params: dict | None = None, notify: bool = False) -> dict:
        msg = {"jsonrpc": "2.0", "method": method}
        if not notify:
            msg["id"] = 1   # the reference always sends id 1
        if params is not None:
            msg["params"] = params
        return msg

    async def init_session(self) -> None:
        if self.session_id:
            return
        hello = self._rpc("initialize", {"protocolVersion": PROTOCOL_VERSION, "capabilities": {},
                                         "clientInfo": {"name": "vectorrag-adapter", "version": "1.0.0"}})
        resp = await self.client.post(self.url, json=hello, headers=self._headers())
        self.session_id = resp.headers.get("mcp-session-id")
        await self.client.post(self.url, json=self._rpc("notifications/initialized", notify=True),
                               headers=self._headers())
        logger.info("Session initialized: %s", self.session_id)

    async def call_tool(self, tool_name: str, args: dict) -> dict:
        await self.init_session()
        call = self._rpc("tools/call", {"name": tool_name, "arguments": args})
        logger.info("Calling tool: %s", tool_name)
        async with self.client.stream("POST", self.url, json=call, headers=self._headers()) as resp:
            if resp.status_code != 200:
                raise HTTPException(503, f"MCP error: {resp.status_code}")
            result = None
            async for line in resp.aiter_lines():
                result = first_sse_result([line])
                if result is not None:
                    break
            if result is None:
                raise HTTPException(500, "No result from MCP")
            return result


mcp = MCPClient()
app = FastAPI(title="Vector RAG Adapter")


@app.get("/")
def root():
    return {"service": "vector-rag-adapter", "mcp_server": MCP_URL,
            "endpoints": {"health": "/health", "search": "/search", "answer": "/answer",
                          "stats": "/stats"}}


@app.get("/health")
async def health():
    try:
        return {"status": "healthy", "mcp": await mcp.call_tool("health_check", {})}
    except Exception:
        return {"status": "unhealthy", "mcp": "unavailable"}


@app.post("/search")
async def search(req: SearchRequest):
    return await mcp.call_tool("search_vectors", {"query": req.query, "top_k": req.top_k})


@app.post("/answer")
async def answer(req: AnswerRequest):
    return await mcp.call_tool("answer_question", {"question": req.question, "top_k": req.top_k})


@app.get("/stats")
async def stats():
    return await mcp.call_tool("get_collection_stats", {})


def main() -> None:
    import uvicorn
    logging.basicConfig(level=logging.INFO)
    logger.info("Vector RAG Adapter on port 9001 -> %s", MCP_URL)
    uvicorn.run(app, host="0.0.0.0", port=9001)


if __name__ == "__main__":
    main()

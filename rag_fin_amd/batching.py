"""Micro-batching of concurrent searches.

The reference answers strictly one query per request (`encode([query])`, one Milvus
search; vector_rag_mcp/main.py:50-57, adapters/vectorrag_adapter.py:142-149).  On the
GPU a corpus sweep costs the same for 1 or 64 queries, so concurrent requests are
worth coalescing: callers block in `search()`, a single worker thread drains the
queue every `max_wait_ms` (or as soon as `max_batch` requests are waiting) and
answers them with ONE `VectorRAG.search_batch` call (one embed + one sweep).
"""
from __future__ import annotations

import threading
import time
from collections import deque


class _Request:
    __slots__ = ("query", "top_k", "done", "result", "error")

    def __init__(self, query: str, top_k: int):
        self.query, self.top_k = query, top_k
        self.done = threading.Event()
        self.result = None
        self.error = None


class MicroBatcher:
    def __init__(self, rag, max_batch: int = 64, max_wait_ms: float = 2.0):
        self.rag = rag
        self.max_batch = max_batch
        self.max_wait = max_wait_ms / 1e3
        self._q: deque[_Request] = deque()
        self._cv = threading.Condition()
        self._stop = False
        self.batches = 0           # statistics: number of GPU batches issued
        self.requests = 0
        self._worker = threading.Thread(target=self._run, name="ragfin-microbatch", daemon=True)
        self._worker.start()

    # -- caller side -------------------------------------------------------------------
    def search(self, query: str, top_k: int = 3):
        """Same contract as VectorRAG.search; blocks until this request's batch is done."""
        req = _Request(query, top_k)
        with self._cv:
            if self._stop:
                raise RuntimeError("MicroBatcher is closed")
            self._q.append(req)
            self._cv.notify()
        req.done.wait()
        if req.error is not None:
            raise req.error
        return req.result

    def close(self) -> None:
        with self._cv:
            self._stop = True
            self._cv.notify_all()
        self._worker.join()

    # -- worker --------------------------------------------------------------------------
    def _take(self):
        with self._cv:
            while not self._q and not self._stop:
                self._cv.wait()
            if not self._q:
                return []
            deadline = time.monotonic() + self.max_wait
            while len(self._q) < self.max_batch and not self._stop:
                left = deadline - time.monotonic()
                if left <= 0:
                    break
                self._cv.wait(left)
            n = min(len(self._q), self.max_batch)
            return [self._q.popleft() for _ in range(n)]

    def _run(self) -> None:
        while True:
            batch = self._take()
            if not batch:
                if self._stop:
                    return
                continue
            self.batches += 1
            self.requests += len(batch)
            try:
                k = max(r.top_k for r in batch)
                results = self.rag.search_batch([r.query for r in batch], k)
                for r, ctx in zip(batch, results):
                    r.result = ctx[:r.top_k]
            except Exception as e:          # every waiter gets the failure, none hangs
                for r in batch:
                    r.error = e
            finally:
                for r in batch:
                    r.done.set()

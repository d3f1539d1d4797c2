"""Micro-batching host logic (no GPU): concurrent searches coalesce into few batches,
per-request top_k is honoured, failures propagate to every waiter."""
import threading
import time

import pytest

from rag_fin_amd.batching import MicroBatcher


class FakeRag:
    def __init__(self, delay=0.0, fail=False):
        self.calls = []
        self.delay = delay
        self.fail = fail

    def search_batch(self, queries, top_k):
        self.calls.append((list(queries), top_k))
        if self.fail:
            raise RuntimeError("gpu on fire")
        time.sleep(self.delay)
        return [[{"rank": i + 1, "text": f"{q}:{i}"} for i in range(top_k)] for q in queries]


def test_concurrent_requests_share_batches():
    rag = FakeRag(delay=0.01)
    mb = MicroBatcher(rag, max_batch=16, max_wait_ms=30)
    out = {}

    def worker(i):
        out[i] = mb.search(f"q{i}", 1 + i % 3)
    threads = [threading.Thread(target=worker, args=(i,)) for i in range(40)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    mb.close()
    assert len(out) == 40
    for i, ctx in out.items():
        assert len(ctx) == 1 + i % 3 and ctx[0]["text"] == f"q{i}:0"
    assert mb.requests == 40 and mb.batches <= 8          # far fewer sweeps than requests
    assert all(len(qs) <= 16 for qs, _ in rag.calls)


def test_single_request_is_not_delayed_much():
    mb = MicroBatcher(FakeRag(), max_batch=64, max_wait_ms=5)
    t = time.monotonic()
    assert mb.search("solo", 2)[1]["rank"] == 2
    assert time.monotonic() - t < 0.5
    mb.close()


def test_failure_reaches_every_waiter_and_close_rejects():
    mb = MicroBatcher(FakeRag(fail=True), max_batch=4, max_wait_ms=20)
    errs = []

    def worker():
        try:
            mb.search("q")
        except RuntimeError as e:
            errs.append(str(e))
    ts = [threading.Thread(target=worker) for _ in range(6)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert errs == ["gpu on fire"] * 6
    mb.close()
    with pytest.raises(RuntimeError, match="closed"):
        mb.search("late")

"""Concurrent callers on the real GPU stack (SURVEY.md 8b threading row, 8f rank 4).

The reference runs one process-wide `rag` that FastMCP calls from its worker pool
(vector_rag_mcp/main.py:126,135-146) and one shared adapter client
(adapters/vectorrag_adapter.py:116).  Here 8 threads mix `VectorRAG.search`,
`VectorRAG.search_batch`, `Embedder.encode` (small and large batches) and the `MicroBatcher`
on ONE embedder + ONE store, and every answer must equal the answer the same call gives
single-threaded.  Also: two indexes / raw searches with own workspaces from several threads."""
import threading

import numpy as np
import pytest

from oracle import c_oracle, encoder as oenc, search as osearch, synth_text

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def stack(gpu_device):
    from rag_fin_amd.embedder import Embedder
    from rag_fin_amd.rag import VectorRAG
    from rag_fin_amd.store import CorpusStore
    from rag_fin_amd.tokenizer import WordPieceTokenizer
    texts = synth_text.retemplated_texts(3000, 21)
    cfg = dict(oenc.MINILM_L6, layers=2)
    emb = Embedder(oenc.random_weights(cfg, 4), cfg, tokenizer=WordPieceTokenizer(synth_text.vocab_for()),
                   device=gpu_device)
    store = CorpusStore("thr", dim=384, capacity=3000, device=gpu_device)
    store.insert([[f"c{i}" for i in range(3000)], texts, emb.encode_to_device(texts), ["p"] * 3000, ["t"] * 3000,
                  ["s"] * 3000, [float(i) for i in range(3000)]])
    return dict(texts=texts, emb=emb, store=store, rag=VectorRAG(None, "thr", embedder=emb, store=store))


def test_eight_threads_mixing_every_entry_point(stack):
    from rag_fin_amd.batching import MicroBatcher
    rag, emb, texts = stack["rag"], stack["emb"], stack["texts"]
    queries = synth_text.retemplated_texts(48, 22)
    # single-threaded reference answers
    want_single = {q: rag.search(q, 5) for q in queries}
    want_emb_small = emb.encode(queries[:3])
    big = texts[:600]                                        # ~100 k token slots: the large-batch path
    want_emb_big = emb.encode(big)
    batcher = MicroBatcher(rag, max_batch=16, max_wait_ms=1.0)
    errors, rounds = [], 12

    def same(a, b):     # the same call twice: bit-identical
        return [c["text"] for c in a] == [c["text"] for c in b] and \
            [c["score"] for c in a] == [c["score"] for c in b]

    def consistent(g, w, tol=1e-3):
        # a query embedded inside a BATCH runs through another GEMM tile shape, so its fp16 embedding may
        # differ in the last bit: same ranking up to swaps of chunks whose scores are closer than 2 tol
        # (the re-templated chunks are near-duplicates of 16 templates: close scores are the rule here)
        gs, ws_ = [c["score"] for c in g], [c["score"] for c in w]
        if not np.allclose(gs, ws_, atol=tol):
            return False
        wscore = {c["text"]: c["score"] for c in w}
        for j, c in enumerate(g):
            if c["text"] not in wscore:                       # may only replace something within 2 tol of the cut
                if c["score"] < ws_[-1] - 2 * tol:
                    return False
            elif abs(wscore[c["text"]] - c["score"]) > tol:
                return False
        return True

    def worker(tid):
        try:
            rng = np.random.default_rng(tid)
            for r in range(rounds):
                kind = (tid + r) % 4
                if kind == 0:
                    q = queries[int(rng.integers(0, 48))]
                    assert same(rag.search(q, 5), want_single[q]), ("search", tid, r)
                elif kind == 1:
                    qs = [queries[int(i)] for i in rng.integers(0, 48, 7)]
                    got = rag.search_batch(qs, 5)
                    for q, g in zip(qs, got):
                        assert consistent(g, want_single[q]), ("batch", tid, r)
                elif kind == 2:
                    q = queries[int(rng.integers(0, 48))]
                    assert consistent(batcher.search(q, 5), want_single[q]), ("batcher", tid, r)
                elif tid % 2 == 0:
                    assert np.array_equal(emb.encode(queries[:3]), want_emb_small), ("encode small", tid, r)
                else:
                    assert np.array_equal(emb.encode(big), want_emb_big), ("encode big", tid, r)
        except BaseException as e:      # noqa: BLE001 -- reported by the main thread
            errors.append(e)

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(8)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    batcher.close()
    assert not errors, errors[:3]
    assert batcher.requests > 0 and batcher.batches <= batcher.requests


def test_raw_searches_from_threads_with_own_workspaces_and_streams(gpu_device):
    """The C ABI's threading contract: one immutable index, any number of concurrent rf_search
    calls as long as each has its own workspace (and stream).  6 threads x 40 searches on two
    indexes of different dims; every result must equal the oracle's."""
    import torch
    from rag_fin_amd.store import GpuIndex
    rigs = []
    for d, seed in ((384, 1), (128, 2)):
        c = osearch.synth_unit_rows(30_000, d, seed)
        ix = GpuIndex(d, 30_000, gpu_device)
        ix.add(torch.from_numpy(c).to(gpu_device))
        q16 = osearch.synth_unit_rows(64, d, 10 + seed)
        rigs.append((ix, torch.from_numpy(q16).to(gpu_device), c_oracle.search(q16, c, 10)))
    torch.cuda.synchronize()
    errors = []

    def worker(tid):
        try:
            ix, q, (os_, oi) = rigs[tid % 2]
            ws = ix.new_workspace()
            stream = torch.cuda.Stream(device=gpu_device)
            with torch.cuda.stream(stream):
                for _ in range(40):
                    s, i, e, f = ix.search_raw(q, 10, want_exact=True, workspace=ws)
                    stream.synchronize()
                    assert int(f.abs().sum()) == 0
                    assert np.array_equal(i.cpu().numpy(), oi) and np.array_equal(e.cpu().numpy(), os_)
        except BaseException as e:      # noqa: BLE001
            errors.append(e)

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(6)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors[:3]

#!/usr/bin/env python3
"""Latency of the reference's actual serving call -- VectorRAG.search(query, top_k), one query
per request (vector_rag_mcp/main.py:48-70) -- on this build: tokenizer -> rf_encode -> rf_search
-> hit marshalling, with a stage split.  Corpus: synthetic unit rows (the store's own vectors
do not matter for latency), model: MiniLM-L6 architecture with seeded random weights and a
vocabulary built from the golden chunk texts (no real files offline)."""
import argparse
import json
import os
import re
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("RAGFIN_LIB", "exp")   # the experiments build: rf_set_tuning and the diagnostic hooks live there only

QUESTIONS = ["What was the total income in the first quarter?", "How did net profit change year over year?",
             "What is the capital adequacy ratio?", "Segment results for retail banking",
             "Gross NPA ratio and provisions in the latest quarter"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=100_000)
    ap.add_argument("--calls", type=int, default=300)
    ap.add_argument("--top-k", type=int, default=5)
    ap.add_argument("--tune", default="", help="rf_set_tuning pairs, e.g. ln_tail=0")
    ap.add_argument("--profile", action="store_true", help="cProfile of the calls: where the host time of a query goes")
    args = ap.parse_args()
    import torch
    from oracle import encoder as oenc, search as osearch
    from rag_fin_amd.embedder import Embedder
    from rag_fin_amd.rag import VectorRAG
    from rag_fin_amd.store import CorpusStore
    from rag_fin_amd.tokenizer import WordPieceTokenizer
    dev = torch.device("cuda:0")
    cfg = dict(oenc.MINILM_L6)
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "chunks_golden.json")))
    words = set()
    for t in [c["text"] for c in gold] + QUESTIONS:
        words.update(re.findall(r"[a-z]+|[0-9]|[^\sa-z0-9]", t.lower()))
    vocab = list(dict.fromkeys(["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]"] + sorted(words) +
                               ["##" + w for w in sorted(words) if w.isalpha()]))
    vocab += ["[unused%d]" % i for i in range(cfg["vocab_size"] - len(vocab))]
    emb = Embedder(oenc.random_weights(cfg, 0), cfg, tokenizer=WordPieceTokenizer(vocab), device=dev)
    store = CorpusStore("fin_chunks", dim=384, capacity=args.rows, device=dev)
    n = args.rows
    vec = torch.from_numpy(osearch.synth_unit_rows(n, 384, 1234)).to(dev)
    texts = [gold[i % len(gold)]["text"] for i in range(n)]
    store.insert([[f"id{i}" for i in range(n)], texts, vec, ["Q1 2024"] * n, ["profitability_analysis"] * n,
                  ["consolidated"] * n, [float(i) for i in range(n)]])
    store.flush()
    store.load()
    rag = VectorRAG(None, embedder=emb, store=store)
    if args.tune:
        from rag_fin_amd import _lib
        for kv in args.tune.split(","):
            k_, v_ = kv.split("=")
            _lib.check(_lib.load_library().rf_set_tuning(k_.encode(), int(v_)))
    for q in QUESTIONS * 4:
        rag.search(q, args.top_k)
    if args.profile:
        import cProfile, pstats
        pr = cProfile.Profile()
        pr.enable()
        for i in range(args.calls):
            rag.search(QUESTIONS[i % len(QUESTIONS)], args.top_k)
        pr.disable()
        pstats.Stats(pr).sort_stats("tottime").print_stats(28)
        return
    lat, t_tok, t_enc, t_search = [], [], [], []
    for i in range(args.calls):
        q = QUESTIONS[i % len(QUESTIONS)]
        t0 = time.perf_counter()
        r = rag.search(q, args.top_k)
        lat.append(time.perf_counter() - t0)
        assert len(r) == args.top_k and r[0]["rank"] == 1
        # stage split (separate calls, each synchronised)
        t0 = time.perf_counter()
        ids, lens = emb.tokenizer.batch_native([q], 256)
        t1 = time.perf_counter()
        e = emb.encode_ids(ids, lens)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        store.index.search(e, args.top_k)[1].cpu()
        t3 = time.perf_counter()
        t_tok.append(t1 - t0)
        t_enc.append(t2 - t1)
        t_search.append(t3 - t2)
    us = lambda xs, p: round(float(np.percentile(xs, p)) * 1e6, 1)
    print(json.dumps({"workload": f"VectorRAG.search(query, top_k={args.top_k}), one query per call, {n} x 384 corpus",
                      "calls": args.calls, "p50_us": us(lat, 50), "p90_us": us(lat, 90), "p99_us": us(lat, 99),
                      "stage_p50_us": {"tokenize": us(t_tok, 50), "encode (sync)": us(t_enc, 50),
                                       "search + download (sync)": us(t_search, 50)}}))


if __name__ == "__main__":
    main()

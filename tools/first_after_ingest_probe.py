#!/usr/bin/env python3
"""One process: [large-batch forward] -> [query-sized forward], many times, in four settings -- plain launches or hipGraph
replay for the query, with or without a device synchronisation between the two -- counting query embeddings that differ in
any bit from the reference (the query encoded on a quiet device).  (the open item of round 3.)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("RAGFIN_LIB", "exp")
import torch
from oracle import encoder as oenc, synth_text
from rag_fin_amd import _lib
from rag_fin_amd.embedder import Embedder
from rag_fin_amd.tokenizer import WordPieceTokenizer

dev = torch.device("cuda:0")
cfg = dict(oenc.MINILM_L6, layers=2)
emb = Embedder(oenc.random_weights(cfg, 9), cfg, tokenizer=WordPieceTokenizer(synth_text.vocab_for()), device=dev)
lib = _lib.load_library()
texts = synth_text.retemplated_texts(2000, 31)
queries = synth_text.retemplated_texts(6, 32)
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 40
ref = {}
for qi, q in enumerate(queries):
    for _ in range(3):
        e = emb.encode_to_device([q])
    torch.cuda.synchronize()
    ref[qi] = e.cpu().numpy().view(np.uint16).copy()
for graph in (0, 1):
    _lib.check(lib.rf_set_tuning(b"encode_graph", graph))
    for sync in (0, 1):
        bad, worst = 0, 0.0
        for it in range(iters):
            emb.encode_to_device(texts[:1200] if it % 2 == 0 else texts[1200:])
            if sync:
                torch.cuda.synchronize()
            qi = it % len(queries)
            e = emb.encode_to_device([queries[qi]]).cpu().numpy().view(np.uint16)
            if not np.array_equal(e, ref[qi]):
                bad += 1
                worst = max(worst, float(np.abs(e.view(np.float16).astype(np.float32) - ref[qi].view(np.float16).astype(np.float32)).max()))
        print(f"query forward by {'hipGraph replay' if graph else 'plain launches'}, {'with' if sync else 'no'} synchronize after the ingest: "
              f"{bad} of {iters} first-after-ingest embeddings differ from the reference (max abs {worst:.1e})", flush=True)

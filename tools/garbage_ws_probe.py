#!/usr/bin/env python3
"""Does a query-sized forward read workspace memory it did not write?  Encode the same queries with the embedder's
fixed small-path workspace pre-filled with zeros, with 0xFF bytes (NaN patterns), with large finite values and with
random bits, and compare the embeddings bitwise."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import encoder as oenc, synth_text
from rag_fin_amd.embedder import Embedder
from rag_fin_amd.tokenizer import WordPieceTokenizer

dev = torch.device("cuda:0")
cfg = dict(oenc.MINILM_L6, layers=2)
emb = Embedder(oenc.random_weights(cfg, 9), cfg, tokenizer=WordPieceTokenizer(synth_text.vocab_for()), device=dev)
queries = synth_text.retemplated_texts(6, 32) + ["what was the net profit", "total income in Q1 2024 of ICICI Bank?"]
emb.encode_to_device([queries[0]])          # creates the small-path buffers
ws = emb._small_ws
gen = torch.Generator(device=dev).manual_seed(1)
fills = {"zeros": lambda: ws.zero_(), "0xFF": lambda: ws.fill_(255),
         "fp16 60000": lambda: ws.view(torch.float16).fill_(60000.0),
         "random bits": lambda: ws.copy_(torch.randint(0, 256, ws.shape, dtype=torch.uint8, device=dev, generator=gen))}
ref = None
for name, fill in fills.items():
    outs = []
    for q in queries:
        fill()
        torch.cuda.synchronize()
        outs.append(emb.encode_to_device([q]).cpu().numpy().view(np.uint16).copy())
    outs = np.concatenate(outs)
    if ref is None:
        ref = outs
        print(f"{name}: reference")
    else:
        d = outs != ref
        print(f"{name}: {int(d.any(1).sum())} of {len(queries)} embeddings differ from the zero-filled run ({int(d.sum())} components)")

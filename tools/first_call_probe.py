#!/usr/bin/env python3
"""Fresh process per trial: ingest 1 200 + 800 chunk texts (the large-batch path), then encode one query five times in a row
(plain launches, hipGraph capture, replays) and report which of the five differ from the last one.  Variants (env PROBE_VARIANT):
  base        as the sharded-store test does it
  sync        torch.cuda.synchronize() between the ingest and the first query
  prewarm     the query-sized buffers are created (one throw-away query) BEFORE the ingest
  zero_ws     the small-path workspace is zero-filled before the first query
(the open item of round 3.)"""
import os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child():
    import torch
    from oracle import encoder as oenc, synth_text
    from rag_fin_amd.embedder import Embedder
    from rag_fin_amd.tokenizer import WordPieceTokenizer
    variant = os.environ.get("PROBE_VARIANT", "base")
    dev = torch.device("cuda:0")
    cfg = dict(oenc.MINILM_L6, layers=2)
    emb = Embedder(oenc.random_weights(cfg, 9), cfg, tokenizer=WordPieceTokenizer(synth_text.vocab_for()), device=dev)
    texts = synth_text.retemplated_texts(2000, 31)
    q = synth_text.retemplated_texts(6, 32)[0]
    if variant == "prewarm":
        emb.encode_to_device(["warm up the query-sized path"])
        torch.cuda.synchronize()
    emb.encode_to_device(texts[:1200])
    emb.encode_to_device(texts[1200:])
    if variant == "sync":
        torch.cuda.synchronize()
    if variant == "zero_ws":
        emb._small_buffers()
        emb._small_ws.zero_()
    rep = np.stack([emb.encode_to_device([q]).cpu().numpy()[0] for _ in range(5)]).view(np.uint16)
    print("PATTERN", "".join("=" if np.array_equal(rep[i], rep[-1]) else "X" for i in range(5)), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child()
        sys.exit(0)
    trials = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    for variant in ("base", "sync", "prewarm", "zero_ws"):
        pats = []
        for _ in range(trials):
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], capture_output=True, text=True,
                               env=dict(os.environ, PROBE_VARIANT=variant), timeout=120)
            pats += [l.split()[1] for l in r.stdout.splitlines() if l.startswith("PATTERN")]
        print(f"{variant:8s}: " + " ".join(pats) + f"   ({sum(p != '=====' for p in pats)} of {len(pats)} trials with a deviating encode)", flush=True)

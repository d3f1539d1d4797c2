#!/usr/bin/env python3
"""Two processes on one card: after a large text ingest each encodes the same queries over and over through the
query-sized path and counts embeddings that differ in any bit from its first answer; the first answers of the
two processes are compared at the end (files under gpurun_out/)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def worker(rank, out_dir, iters):
    import torch
    from oracle import encoder as oenc, synth_text
    from rag_fin_amd.embedder import Embedder
    from rag_fin_amd.tokenizer import WordPieceTokenizer
    dev = torch.device("cuda:0")
    cfg = dict(oenc.MINILM_L6, layers=2)
    emb = Embedder(oenc.random_weights(cfg, 9), cfg, tokenizer=WordPieceTokenizer(synth_text.vocab_for()), device=dev)
    texts = synth_text.retemplated_texts(2000, 31)
    queries = synth_text.retemplated_texts(6, 32)
    first, bad = {}, 0
    mode = os.environ.get("PROBE_MODE", "both")
    if mode == "split" and rank == 1:          # this process only ingests (large batches), the other only answers queries
        t_end = time.time() + float(os.environ.get("PROBE_SECONDS", "20"))
        n = 0
        while time.time() < t_end:
            emb.encode_to_device(texts)
            torch.cuda.synchronize()
            n += 1
        print(f"rank 1: {n} ingests of {len(texts)} texts", flush=True)
        np.save(os.path.join(out_dir, "det_first_1.npy"), np.zeros((len(queries), 384), np.uint16))
        return
    for it in range(iters):
        if it % 10 == 0 and mode != "split":
            emb.encode_to_device(texts[(rank * 1000):(rank * 1000) + 1000])    # an ingest in between, as the store test does
        for qi, qt in enumerate(queries):
            e = emb.encode_to_device([qt]).cpu().numpy().view(np.uint16)
            if qi not in first:
                first[qi] = e.copy()
            elif not np.array_equal(e, first[qi]):
                bad += 1
                d = np.abs(e.view(np.float16).astype(np.float32) - first[qi].view(np.float16).astype(np.float32)).max()
                print(f"rank {rank} iter {it} query {qi}: differs from its first answer, max abs {d:.2e}", flush=True)
    np.save(os.path.join(out_dir, f"det_first_{rank}.npy"), np.stack([first[i] for i in range(len(queries))]))
    print(f"rank {rank}: {iters * len(queries)} query encodes, {bad} differ from the first answer", flush=True)


if __name__ == "__main__":
    import torch.multiprocessing as mp
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(out, exist_ok=True)
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    mp.spawn(worker, args=(out, iters), nprocs=2, join=True)
    a, b = np.load(os.path.join(out, "det_first_0.npy")), np.load(os.path.join(out, "det_first_1.npy"))
    print("first answers of the two processes identical:", bool(np.array_equal(a, b)))

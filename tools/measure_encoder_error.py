#!/usr/bin/env python3
"""Achieved numerical error of the GPU encoder (rf_encode) against the float64 oracle at the
FULL architecture: 6 layers, hidden 384, vocab 30 522, sequences up to T = 256 (seeded random
weights; the real all-MiniLM-L6-v2 checkpoint does not exist offline).  The figures this prints
are what the tolerances in tests/test_encoder_gpu.py, tests/test_model_dir.py and
tests/test_end_to_end_gpu.py are derived from (<= 3x measured); the JSON is committed under
profiles/.  Also measures the north-star quantity: |score(GPU encode -> GPU search) -
score(oracle encode -> oracle search)| on a corpus built from the same rows."""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def stats(got, want):
    d = got.astype(np.float64) - want
    cos = (got * want).sum(1) / np.linalg.norm(got, axis=1) / np.linalg.norm(want, axis=1)
    return {"max_abs": float(np.abs(d).max()), "l2_max": float(np.linalg.norm(d, axis=1).max()),
            "l2_mean": float(np.linalg.norm(d, axis=1).mean()), "one_minus_cos_max": float((1 - cos).max()),
            "norm_dev_max": float(np.abs(np.linalg.norm(got, axis=1) - 1).max())}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=48)
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()
    import torch
    from oracle import c_oracle, encoder as oenc
    from rag_fin_amd.embedder import Embedder
    from rag_fin_amd.store import GpuIndex
    dev = torch.device("cuda:0")
    cfg = dict(oenc.MINILM_L6)
    w = oenc.random_weights(cfg, args.seed)
    emb = Embedder(w, cfg, device=dev)
    rng = np.random.default_rng(99)
    B, T = args.rows, 256
    lens = rng.integers(8, T + 1, B).astype(np.int32)
    lens[0], lens[1], lens[2] = T, 1, 12
    ids = rng.integers(1000, cfg["vocab_size"], (B, T)).astype(np.int32)
    want = oenc.encode(oenc.round_weights_fp16(w), cfg, ids, lens)
    out = {"config": "MiniLM-L6 architecture (6 layers, H 384, 12 heads, FFN 1536, vocab 30522), seeded random "
                     "weights rounded to fp16 on both sides; %d sequences, lens 1..256 (one at 256)" % B,
           "oracle": "oracle/encoder.py float64"}
    # the three GEMM paths: small (<= 1024 slots), direct tiles, LDS-DMA ring (>= 8192 slots)
    got32 = emb.encode_ids(ids, lens, out_dtype="float32").cpu().numpy()
    got16 = emb.encode_ids(ids, lens).float().cpu().numpy()
    out["large_batch_f32_out"] = stats(got32, want)
    out["large_batch_f16_out"] = stats(got16, want)
    small = np.concatenate([emb.encode_ids(ids[i:i + 4], lens[i:i + 4], out_dtype="float32").cpu().numpy()
                            for i in range(0, 8, 4)])
    out["small_batch_f32_out"] = stats(small, want[:8])
    mid = emb.encode_ids(ids[:16], lens[:16], out_dtype="float32").cpu().numpy()
    out["mid_batch_f32_out"] = stats(mid, want[:16])
    # north star: scores of the all-GPU pipeline vs the all-CPU pipeline, same texts/ids
    corpus_gpu = emb.encode_ids(ids, lens)                      # fp16 rows as the store keeps them
    ix = GpuIndex(384, B, dev)
    ix.add(corpus_gpu)
    nq = 8
    s_gpu, i_gpu, _ = ix.search(corpus_gpu[:nq].contiguous(), 10)
    c_or = want.astype(np.float32).astype(np.float16)
    s_or, i_or = c_oracle.search(c_or[:nq], c_or, 10)
    s_gpu, i_gpu = s_gpu.cpu().numpy().astype(np.float64), i_gpu.cpu().numpy()
    # compare score of the same (query, row) pairs: the oracle's ranked rows, scored by the GPU pipeline
    full_gpu = ix.debug_scores(corpus_gpu[:nq].contiguous()).cpu().numpy().astype(np.float64)
    full_or = c_or[:nq].astype(np.float64) @ c_or.astype(np.float64).T
    out["north_star"] = {"what": "|cos(GPU-encoded q, GPU-encoded c) - cos(oracle-encoded q, oracle-encoded c)| over "
                                 "all %d x %d pairs" % (nq, B),
                         "max_abs_score_diff": float(np.abs(full_gpu - full_or).max()),
                         "ranks_equal_top10": bool(np.array_equal(i_gpu, i_or)),
                         "top10_score_diff_max": float(np.abs(np.sort(s_gpu, 1) - np.sort(s_or, 1)).max())}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Config 4 of BASELINE.json: on-GPU encode of 10k finance-chunk-sized sequences
(token ids ~U[40,250], seeded; real tokenizer/vocab are not available offline) with
the MiniLM-L6 architecture on seeded random weights, then top-10 search over the
10k corpus.  Reports tokens/s, encoder TFLOP/s (oracle.encoder.flops_per_token) and
the per-kernel split (HIP events around whole calls; use rocprofv3 for kernels)."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("RAGFIN_LIB", "exp")   # the experiments build: rf_set_tuning and the diagnostic hooks live there only


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--chunks", type=int, default=10_000)
    ap.add_argument("--batch-tokens", type=int, default=65536)
    ap.add_argument("--iters", type=int, default=3)
    ap.add_argument("--fixed-len", type=int, default=0)
    ap.add_argument("--tune", default="", help="rf_set_tuning pairs, e.g. ffn2_ntb=4,linear_dma=0")
    ap.add_argument("--texts", action="store_true", help="also time the text -> tokenizer -> encoder path")
    ap.add_argument("--stamps", action="store_true", help="clock stamps of the last k_linear_dma launch")
    ap.add_argument("--stamp-epi", type=int, default=1, help="0 = QKV, 1 = FFN1, 2 = attention, 3 = FFN2, 4 = out-projection, 5 = post block")
    ap.add_argument("--linear-dbg", type=int, default=0, help="ablation bits of k_linear_dma (results wrong)")
    args = ap.parse_args()
    import torch
    from oracle import encoder as oenc
    from rag_fin_amd.embedder import Embedder
    from rag_fin_amd.store import GpuIndex
    dev = torch.device("cuda:0")
    cfg = dict(oenc.MINILM_L6)
    emb = Embedder(oenc.random_weights(cfg, 0), cfg, device=dev)
    if args.tune:
        from rag_fin_amd import _lib
        for kv in args.tune.split(","):
            k_, v_ = kv.split("=")
            _lib.check(_lib.load_library().rf_set_tuning(k_.encode(), int(v_)))
    rng = np.random.default_rng(7)
    lens = (np.full(args.chunks, args.fixed_len) if args.fixed_len
            else rng.integers(40, 251, args.chunks)).astype(np.int32)
    order = np.argsort(lens, kind="stable")
    batches = []
    i = 0
    while i < len(order):
        j = i
        while j < len(order) and (j - i + 1) * lens[order[j]] <= args.batch_tokens:
            j += 1
        j = max(j, i + 1)
        idx = order[i:j]
        T = int(lens[idx].max())
        ids = rng.integers(1000, cfg["vocab_size"], (len(idx), T)).astype(np.int32)
        batches.append((torch.from_numpy(ids).to(dev), torch.from_numpy(lens[idx]).to(dev), idx))
        i = j
    tokens = int(lens.sum())
    flops = float(sum(oenc.flops_per_token(cfg, int(l)) * int(l) for l in lens))
    out = torch.empty((args.chunks, 384), dtype=torch.float16, device=dev)

    def run():
        for ids, ln, idx in batches:
            out[torch.as_tensor(idx, device=dev)] = emb.encode_ids(ids, ln)
    run()
    torch.cuda.synchronize()
    if args.stamps:
        from ctypes import c_void_p
        from rag_fin_amd import _lib
        lib = _lib.load_library()
        buf = torch.zeros(512 * 8 * 8, dtype=torch.float32, device=dev)
        lib.rf_set_tuning(b"debug_epi", args.stamp_epi)
        lib.rf_set_tuning(b"linear_dbg", args.linear_dbg)
        lib.rf_debug_set_buffer(c_void_p(buf.data_ptr()))
        ids, ln, idx = batches[len(batches) // 2]
        emb.encode_ids(ids, ln)
        torch.cuda.synchronize()
        lib.rf_debug_set_buffer(None)
        if args.stamp_epi == 2:     # attention: [workgroup][wave] x {cycles, staging, QK + max, exp, PV, output, items, tokens}
            st = buf.view(1024, 4, 8).cpu().numpy()
            ok = st[..., 0] > 0
            tot, stage, qk, ex, pv, outp, items, ntok = (st[..., i][ok] for i in range(8))
            per = np.maximum(items / 4.0, 1.0)
            print("attention stamps (last launch, %d waves, median tokens %d, items/workgroup %d): wave %.0f cycles = staging %.0f + "
                  "per item (x%.1f): QK+max %.0f, exp+sum %.0f, PV %.0f, scale+store %.0f" %
                  (ok.sum(), np.median(ntok), np.median(items), np.median(tot), np.median(stage), np.median(per),
                   np.median(qk / per), np.median(ex / per), np.median(pv / per), np.median(outp / per)))
            return
        if args.stamp_epi == 5:     # k_post_block: [workgroup][wave 0-3 of 8] x {cycles, prologue, out-proj, LN1, MLP, LN2, wait, steps}
            full = buf.view(512, 8, 8).cpu().numpy()
            st = full[:, :4]
            ex = full[:, 4:]
            okx = ex[..., 0] > 0
            if okx.any():
                print("out-projection step by step (cycles, medians): " + " ".join("%.0f" % np.median(ex[..., q][okx]) for q in range(6)) +
                      "; residual MFMAs %.0f; next layer's QKV (18 steps) %.0f" % (np.median(ex[..., 6][okx]), np.median(ex[..., 7][okx])))
            ok = st[..., 0] > 0
            for name, sel in (("all workgroups", slice(0, 512)), ("workgroups 0-255 (first on their CU)", slice(0, 256)),
                              ("workgroups 256-511", slice(256, 512))):
                sub = st[sel]
                ok = sub[..., 0] > 0
                if not ok.any():
                    continue
                tot, pro, pa, l1, pb, l2, wait, ticks = (sub[..., i][ok] for i in range(8))
                print("k_post_block stamps, %s (last launch, %d waves, clock %.2f GHz): wave %.0f cycles = prologue %.0f + out-projection %.0f "
                      "(6 steps x %.0f) + LayerNorm 1 %.0f + MLP %.0f (50 steps x %.0f) + LayerNorm 2 and stores %.0f; vmcnt wait + barrier %.0f per step" %
                      (name, ok.sum(), np.median(tot / ticks) * 0.1, np.median(tot), np.median(pro), np.median(pa), np.median(pa) / 6, np.median(l1),
                       np.median(pb), np.median(pb) / 50, np.median(l2), np.median(wait) / 56))
            return
        if args.stamp_epi in (3, 4):   # k_gemm_tile: [workgroup][wave] x {cycles, prologue, loop, wait in loop, epilogue, stages}
            st = buf.view(512, 8, 8).cpu().numpy()
            ok = st[..., 0] > 0
            tot, pro, loop, wait, epi, nst, e1, e2 = (st[..., i][ok] for i in range(8))
            print("k_gemm_tile stamps (last launch of the %s GEMM, %d waves): wave %.0f cycles = prologue %.0f + %d stages x %.0f "
                  "(of which vmcnt wait + barrier %.0f) + epilogue %.0f (loads + bias + residual %.0f, LayerNorm statistics %.0f, "
                  "normalise + store %.0f)" %
                  ("K = 1536" if args.stamp_epi == 3 else "K = 384", ok.sum(), np.median(tot), np.median(pro), int(nst.max()),
                   np.median(loop / nst), np.median(wait / nst), np.median(epi), np.median(e1), np.median(e2),
                   np.median(epi - e1 - e2)))
            return
        st = buf.view(512, 8, 8).cpu().numpy()
        ok = st[..., 1] > 0
        cyc, ticks, pre, wait, nph = (st[..., i][ok] for i in range(5))
        print("stamps (last k_linear_dma launch of the chosen epilogue, %d waves): clock %.2f GHz; kernel %.1f us; "
              "prologue %.0f cycles; loop %.0f cycles/phase (%d phases); wait+barrier share %.1f %%" %
              (ok.sum(), np.median(cyc / ticks) * 0.1, np.median(ticks) / 100.0, np.median(pre),
               np.median((cyc - pre) / nph), int(nph.max()), 100.0 * np.median(wait / cyc)))
    times = []
    for _ in range(args.iters):
        t = time.perf_counter()
        run()
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t)
    best = min(times)
    ix = GpuIndex(384, args.chunks, dev)
    ix.add(out)
    q = out[:64].contiguous()
    ix.search_raw(q, 10)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(50):
        ix.search_raw(q, 10)
    torch.cuda.synchronize()
    search_ms = (time.perf_counter() - t) / 50 * 1e3
    text_leg = None
    if args.texts:
        # end-to-end from TEXT: chunk texts re-templated from the 16 golden chunks with perturbed
        # figures, vocabulary built from their words (no real vocab offline),
        # native tokenizer -> bucketed encode -> search
        import re
        from rag_fin_amd.tokenizer import WordPieceTokenizer
        gold = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                           "tests", "golden", "chunks_golden.json")))
        base = [c["text"] for c in gold]
        trng = np.random.default_rng(11)
        texts = [re.sub(r"\d", lambda m: str(int(trng.integers(0, 10))), base[i % len(base)]) for i in range(args.chunks)]
        words = set()
        for t in base:
            words.update(re.findall(r"[a-z]+|[0-9]|[^\sa-z0-9]", t.lower()))
        vocab = (["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]"] + sorted(words) +
                 ["##" + w for w in sorted(words) if w.isalpha()] + ["##%d" % i for i in range(10)])
        vocab = list(dict.fromkeys(vocab)) + ["[unused%d]" % i for i in range(cfg["vocab_size"])]
        emb_t = Embedder(oenc.random_weights(cfg, 0), cfg, tokenizer=WordPieceTokenizer(vocab[:cfg["vocab_size"]]),
                         device=dev)
        emb_t.encode_to_device(texts[:256])
        torch.cuda.synchronize()
        t = time.perf_counter()
        ids_, lens_ = emb_t.tokenizer.batch_native(texts, 256)
        tok_s = time.perf_counter() - t
        runs = []
        for _ in range(3):
            t = time.perf_counter()
            e_ = emb_t.encode_to_device(texts, batch_tokens=args.batch_tokens)
            torch.cuda.synchronize()
            runs.append(time.perf_counter() - t)
        e2e_s = min(runs)
        text_leg = {"texts": len(texts), "tokens": int(lens_.sum()), "tokenize_s": round(tok_s, 4),
                    "tokenize_texts_per_s": round(len(texts) / tok_s, 1),
                    "text_to_embedding_s": round(e2e_s, 4), "text_to_embedding_runs_s": [round(r, 4) for r in runs],
                    "texts_per_s": round(len(texts) / e2e_s, 1),
                    "host_threads": len(os.sched_getaffinity(0))}
    print(json.dumps({"workload": f"encode {args.chunks} chunks (lens U[40,250]) + top-10 search, MiniLM-L6 random weights",
                      "from_text": text_leg,
                      "chunks": args.chunks, "tokens": tokens, "batches": len(batches),
                      "encode_s": round(best, 4), "tokens_per_s": round(tokens / best, 1),
                      "chunks_per_s": round(args.chunks / best, 1),
                      "encoder_TFLOPs": round(flops / best / 1e12, 2),
                      "mfma_f16_peak_TFLOPs": 2500, "frac_of_mfma_peak": round(flops / best / 2.5e15, 4),
                      "search_batch64_ms": round(search_ms, 4)}))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""A/B timing of the wide sweep (65..256 queries per corpus pass) in ONE process:
variants selected through rf_set_tuning, whole rf_search steps timed with HIP events
(interleaved rounds, medians).  Prints per-step time, QPS and the corpus rate."""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("RAGFIN_LIB", "exp")   # the experiments build: rf_set_tuning and the diagnostic hooks live there only


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1_000_000)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--sample-pairs", type=int, default=4)
    ap.add_argument("--dbgs", default="0,64", help="comma-separated wide_dbg values to A/B in one process: 0 = the product "
                    "kernel (16x16x32 MFMA, staggered halves), 64 = the round-1 kernel (32x32x16, lockstep); ablations of the "
                    "product kernel (results wrong): 1 = DMA pieces re-read one cached KiB, 2 = no filters, 8 = no LDS-DMA in the "
                    "loop, 10 = 2 + 8; 4 = clock stamps")
    ap.add_argument("--check", action="store_true", help="compare the result with four 64-query sweeps")
    args = ap.parse_args()
    import torch
    from rag_fin_amd import _lib
    from rag_fin_amd.store import GpuIndex
    dev = torch.device("cuda:0")
    dim = 384
    gen = torch.Generator(device=dev).manual_seed(1234)
    c = torch.randn((args.rows, dim), generator=gen, device=dev)
    c = (c / c.norm(dim=1, keepdim=True)).half()
    q = torch.randn((args.batch, dim), generator=gen, device=dev)
    q = (q / q.norm(dim=1, keepdim=True)).half()
    ix = GpuIndex(dim, args.rows, dev)
    ix.add(c)
    del c
    lib = _lib.load_library()
    _lib.check(lib.rf_set_tuning(b"wide_sample_pairs", args.sample_pairs))
    # "dbg" or "dbg:ne" (ne = LDS-DMA pieces per phase of waves 0-3: 6 | 8 | 9 | 10 | 11, 0 = the product's 12; + 100 = with stamps)
    # third field: form (0 = eight waves x 32 queries, the round-2 kernel; 1 = four waves x 64 queries, round 3)
    variants = [tuple(int(y) for y in (x.split(":") + ["0", "0"])[:3]) for x in args.dbgs.split(",")]

    def apply(v):
        _lib.check(lib.rf_set_tuning(b"wide_dbg", v[0]))
        _lib.check(lib.rf_set_tuning(b"wide_ne", v[1]))
        _lib.check(lib.rf_set_tuning(b"wide_form", v[2]))

    ref = None
    for v in variants:
        apply(v)
        for _ in range(3):
            s, i, e, f = ix.search_raw(q, 10, want_exact=True)
        torch.cuda.synchronize()
        wrong = v[0] & 11          # ablations that change the result
        assert wrong or int(f.abs().sum()) == 0, ("flags", v)
        if args.check and not wrong:
            parts = [ix.search_raw(q[a:a + 64].contiguous(), 10, want_exact=True) for a in range(0, args.batch, 64)]
            assert torch.equal(i, torch.cat([p[1] for p in parts])) and torch.equal(e, torch.cat([p[2] for p in parts]))
    res = {v: [] for v in variants}
    for _ in range(args.rounds):
        for v in variants:
            apply(v)
            t0 = torch.cuda.Event(enable_timing=True)
            t1 = torch.cuda.Event(enable_timing=True)
            t0.record()
            for _ in range(args.reps):
                ix.search_raw(q, 10, want_exact=True)
            t1.record()
            torch.cuda.synchronize()
            res[v].append(t0.elapsed_time(t1) / args.reps)
    if (variants[-1][0] & 4) or variants[-1][1] >= 100:
        # clock stamps of the last launch: [workgroup][wave] x {cycles, 100-MHz ticks, vmcnt-wait cycles, phases, barrier cycles}
        off = lib.rf_debug_workspace_offset(b"pmax")
        st = ix.workspace[off:off + 256 * 8 * 8 * 4].view(torch.float32).view(256, 8, 8).cpu().numpy()
        cyc, ticks, vm, ph, bar = st[..., 0], st[..., 1], st[..., 2], st[..., 3], st[..., 4]
        ok = (ticks > 100) & (ph >= 1)      # rows of waves the kernel does not have hold the sample pass's maxima
        print("stamps: in-kernel clock %.2f GHz; loop %.1f us; cycles/phase %.0f (median over workgroups x waves; phases %d..%d)" %
              (np.median(cyc[ok] / ticks[ok]) * 0.1, np.median(ticks[ok]) / 100.0, np.median(cyc[ok] / ph[ok]), ph[ok].min(), ph[ok].max()))
        fl, nf = st[..., 5], st[..., 6]
        if ok.any() and float(nf[ok].max()) > 0:
            print("  staging flushes in the loop: %.1f per wave, %.0f cycles each, %.1f %% of the wave's cycles" % (
                np.median(nf[ok]), np.median(fl[ok] / np.maximum(nf[ok], 1)), 100 * np.median(fl[ok] / cyc[ok])))
        for name, sl in (("waves 0-3", slice(0, 4)), ("waves 4-7", slice(4, 8))):
            o2 = ok[:, sl]
            print("  %s: vmcnt wait %.1f %% of the cycles (%.0f cycles/phase), barrier %.1f %% (%.0f cycles/phase)" % (
                name, 100 * np.median(vm[:, sl][o2] / cyc[:, sl][o2]), np.median(vm[:, sl][o2] / ph[:, sl][o2]),
                100 * np.median(bar[:, sl][o2] / cyc[:, sl][o2]), np.median(bar[:, sl][o2] / ph[:, sl][o2])))
    out = {}
    for v in variants:
        apply(v)
        st = [ix.search_profile(q, 10) for _ in range(20)]
        print("  stages (HIP events, us): " + "  ".join("%s %.1f" % (n, 1e3 * float(np.median([x[n] for x in st]))) for n in st[0]))
        ms = float(np.median(res[v]))
        out["wide_dbg_%d_ne_%d_form_%d" % v] = {"ms_per_step": round(ms, 5), "qps": round(args.batch / ms * 1e3, 1),
                                     "corpus_GBps": round(args.rows * dim * 2 / ms / 1e6, 1)}
        print("wide_dbg=%d ne=%d form=%d: %.1f us/step  %.0f QPS  %.0f GB/s" %
              (v[0], v[1], v[2], ms * 1e3, args.batch / ms * 1e3, args.rows * dim * 2 / ms / 1e6))
    print(json.dumps({"rows": args.rows, "batch": args.batch, "results": out}))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc passes (one counter group per pass, as MI355X_MICROARCH.md
prescribes) into the per-kernel JSON kept under profiles/.

  python tools/pmc_summary.py --out profiles/rNN_pmc.json --kernel 'k_scan<24, 8, 2, 4, 1>' \
      --rows 1000000 --dim 384 --batch 64  gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_sq

HBM bytes per launch of --kernel = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: on gfx950 FETCH_SIZE
tallies the 128-B fabric requests of a 16-B/lane streaming read at 64 B (the guide's correction)."""
import argparse
import collections
import csv
import glob
import json
import os


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dirs", nargs="+")
    ap.add_argument("--out", required=True)
    ap.add_argument("--kernel", required=True, help="substring of the dominant kernel's name")
    ap.add_argument("--rows", type=int, required=True)
    ap.add_argument("--dim", type=int, required=True)
    ap.add_argument("--batch", type=int, required=True)
    ap.add_argument("--topk", type=int, default=10)
    ap.add_argument("--skip", type=int, default=5, help="launches per kernel to drop as warm-up")
    ap.add_argument("--command", default="")
    ap.add_argument("--round", default="1")
    args = ap.parse_args()
    vals = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in args.dirs:
        for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(path)):
                vals[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    per = {}
    for kname, counters in vals.items():
        if not any(t in kname for t in ("k_scan", "k_merge", "k_threshold", "k_linear", "k_gemm", "k_attention", "k_embed", "k_pool", "k_post")):
            continue
        rec = {}
        for cname, xs in counters.items():
            xs = xs[args.skip:] if len(xs) > 2 * args.skip else xs
            rec[cname] = round(sum(xs) / len(xs), 2)
            rec["launches_averaged"] = len(xs)
        per[kname] = rec
    dom = [k for k in per if args.kernel in k]
    out = {"round": args.round, "rows": args.rows, "dim": args.dim, "batch": args.batch, "topk": args.topk,
           "command": args.command,
           "correction": "gfx950: FETCH_SIZE counts 128-B fabric requests at 64 B -> doubled for a 16 B/lane "
                         "streaming read (MI355X_MICROARCH.md, HBM section); WRITE_SIZE exact",
           "algorithmic_bytes_per_launch": args.rows * args.dim * 2, "per_kernel_counters": per}
    if dom:
        k = dom[0]
        out["kernel"] = k
        f, w = per[k].get("FETCH_SIZE"), per[k].get("WRITE_SIZE")
        if f is not None and w is not None:
            out["FETCH_SIZE_KB"], out["WRITE_SIZE_KB"] = f, w
            out["hbm_bytes_per_launch"] = int(round((2 * f + w) * 1024))
        busy, gui = per[k].get("SQ_VALU_MFMA_BUSY_CYCLES"), per[k].get("GRBM_GUI_ACTIVE")
        if busy and gui:
            # SQ counters sum over the SIMDs that ran waves; GRBM_GUI_ACTIVE sums the 8 XCDs
            out["mfma_busy_frac"] = round(busy / (gui / 8.0 * 1024.0), 4)
    # matrix-pipe utilisation of every kernel that issued MFMAs (busy cycles / (GPU-active cycles x 1024 SIMDs))
    util = {}
    for kname, rec in per.items():
        busy, gui = rec.get("SQ_VALU_MFMA_BUSY_CYCLES"), rec.get("GRBM_GUI_ACTIVE")
        if busy and gui:
            util[kname] = {"mfma_busy_frac": round(busy / (gui / 8.0 * 1024.0), 4),
                           "valu_insts_per_mfma_incl_mfma": round(rec.get("SQ_INSTS_VALU", 0) / max(rec.get("SQ_INSTS_MFMA", 1), 1), 3)}
    out["mfma_utilisation"] = util
    json.dump(out, open(args.out, "w"), indent=1)
    print(json.dumps({k: out.get(k) for k in ("kernel", "hbm_bytes_per_launch", "algorithmic_bytes_per_launch",
                                               "mfma_busy_frac")}))


if __name__ == "__main__":
    main()

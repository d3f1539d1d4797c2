#!/usr/bin/env python3
"""A/B tuning of the scan pipeline in ONE process (interleaved rounds, medians):
ring depth x emit workgroups per CU x sample blocks per wave, via
rf_set_tuning.  Stage times come from rf_search_profile (HIP events)."""
import argparse
import itertools
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("RAGFIN_LIB", "exp")   # the experiments build: rf_set_tuning and the diagnostic hooks live there only


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1_000_000)
    ap.add_argument("--dim", type=int, default=384)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--rings", default="6,8,12,24")
    ap.add_argument("--wgs", default="2,3")
    ap.add_argument("--bpw", default="1,2")
    args = ap.parse_args()
    import torch
    from rag_fin_amd import _lib
    from rag_fin_amd.store import GpuIndex
    dev = torch.device("cuda:0")
    gen = torch.Generator(device=dev).manual_seed(1234)
    c = torch.randn((args.rows, args.dim), generator=gen, device=dev)
    c = (c / c.norm(dim=1, keepdim=True)).half()
    q = torch.randn((args.batch, args.dim), generator=gen, device=dev)
    q = (q / q.norm(dim=1, keepdim=True)).half()
    ix = GpuIndex(args.dim, args.rows, dev)
    ix.add(c)
    lib = _lib.load_library()
    configs = list(itertools.product([int(x) for x in args.rings.split(",")], [int(x) for x in args.wgs.split(",")],
                                     [int(x) for x in args.bpw.split(",")]))
    res = {cfg: [] for cfg in configs}
    flags_bad = {cfg: 0 for cfg in configs}

    def apply(cfg):
        ring, wgs, bpw = cfg
        for k, v in (("ring24", ring), ("emit_wgs_per_cu", wgs), ("sample_bpw", bpw)):
            _lib.check(lib.rf_set_tuning(k.encode(), v))

    for cfg in configs:  # warm every variant (first launch loads code, sets LDS attributes)
        apply(cfg)
        for _ in range(3):
            ix.search_profile(q, 10)
    for rnd in range(args.rounds):
        for cfg in configs:
            apply(cfg)
            stages = [ix.search_profile(q, 10) for _ in range(args.reps)]
            res[cfg].append({k: float(np.median([s[k] for s in stages])) for k in stages[0]})
            _, _, _, f = ix.search_raw(q, 10)
            flags_bad[cfg] += int(f.abs().sum().item())
    print("ring wgs bpw | sample thr emit merge | total (us, median of round medians)")
    rows = []
    for cfg in configs:
        med = {k: float(np.median([r[k] for r in res[cfg]])) * 1e3 for k in res[cfg][0]}
        tot = sum(med.values())
        rows.append((tot, cfg, med))
    for tot, cfg, med in sorted(rows):
        print("%4d %3d %3d | %6.1f %5.1f %6.1f %5.1f | %6.1f  flags=%d" %
              (*cfg, med["sample"], med["threshold"], med["emit"], med["merge"], tot, flags_bad[cfg]))
    print(json.dumps({"best": {"ring24": rows and sorted(rows)[0][1][0]}}))


if __name__ == "__main__":
    main()

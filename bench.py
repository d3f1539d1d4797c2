#!/usr/bin/env python3
"""bench.py -- headline benchmark of the rag-fin vector-retrieval hot path on MI355X.

Metric (BASELINE.json): queries/sec, top-10, 1M x 384-d fp16 corpus, batch=64,
with recall@10 vs the CPU oracle.  A "step" is one pass of the hot path over one
batch: 64 query embeddings (already in HBM) -> brute-force cosine/IP top-10 over
the HBM-resident corpus -> ranked (score, row id) lists in HBM.

  python bench.py --gpus 1 --steps 50 --warmup 5
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

N > 1: the corpus is row-sharded, the query batch is replicated, and each step ends
with ONE RCCL all-gather of the per-shard top-k plus a merge (rag_fin_amd/sharded.py);
batches in flight work the same way (the all-gathers of the lanes share one
communicator and are issued in the same order on every rank).
Default = STRONG scaling: the metric's 1M-row corpus is split over the N GPUs (rank r
holds rows [r N/W, (r+1) N/W)), so `value` -- queries/sec of the whole job -- grows
with N until the per-step launch floor (six launches, ~31 us with 8 batches in flight) is
reached; at 8 GPUs a shard is 125k rows = 15 us of scan.  `--scaling weak` keeps
--rows rows PER GPU instead (capacity scaling, BASELINE configs[4] style: the same 64
queries against an N-times larger corpus; ideal = flat value, `rows_per_s` grows).

Beside `value` (4 batches in flight, inputs resident in HBM) the one JSON line carries:
`serial` (one batch at a time), `roofline` (emit sweep, HIP events, PMC traffic from
profiles/), `stage_ms`, `batch256` (BASELINE configs[2], two batches in flight),
`pcie_inclusive` (queries from / results to pinned host memory; never `value`),
recall / exactness against the C oracle (at N > 1 also of the merged answer:
`global_*`), and `cpu_baseline` (numpy BLAS port on the host cores, N = 1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--rows", type=int, default=1_000_000,
                    help="corpus rows of the whole job (--scaling strong) or per GPU (--scaling weak)")
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong")
    ap.add_argument("--dim", type=int, default=384)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--topk", type=int, default=10)
    ap.add_argument("--streams", type=int, default=0,
                    help="batches in flight (one HIP stream + workspace each); 1 = strictly serial steps; "
                         "0 = 4 on one GPU, 8 for the sharded step (its scan is short: 31.9 vs 37.5 us/step at 125 k-row shards)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-check", action="store_true")
    ap.add_argument("--group", action="store_true",
                    help="N > 1: one all-gather per group of batches in flight (ShardedSearcher.search_group) instead "
                         "of one per batch -- measured SLOWER in the one-GPU rehearsal (58 vs 50 us/step at 125 k-row "
                         "shards: the groups serialise on the shared buffers), kept as an experiment")
    ap.add_argument("--no-batch256", action="store_true", help="skip the configs[2] (batch 256) side measurement")
    return ap.parse_args()


def pmc_traffic(rows, dim, batch):
    """HBM bytes per emit-scan launch from the committed PMC passes, if they were
    taken on this workload (profiles/*_pmc.json; see DESIGN.md 'Measurement')."""
    try:
        import glob
        for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc.json")), reverse=True):
            rec = json.load(open(path))
            if rec.get("rows") == rows and rec.get("dim") == dim and rec.get("batch") == batch:
                return rec.get("hbm_bytes_per_launch")
    except Exception:
        pass
    return None


def main():
    args = parse()
    # stdout carries ONE line, the JSON: everything any library prints to fd 1 before that (RCCL's
    # version banner, for one) is sent to stderr
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    # read by the HSA runtime when the first HIP call initialises it: must be set before torch
    # touches the GPU (the host driver only supports dmabuf IPC, which RCCL needs across processes)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # kernel arguments in device memory: the image's default, pinned here because the sharded step and
    # the small kernels are launch-latency-bound (31.7 vs 35.1 us per 125 k-row shard step without it)
    os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run "
                             "(one process per GPU)")
        raise SystemExit(f"WORLD_SIZE={world} but --gpus {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU: the hot path has no CPU fallback")
    if os.environ.get("RAGFIN_SHARE_GPU") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # RAGFIN_FORCE_SHARDED=1 at N=1: run the N>1 code path (scan -> RCCL all-gather -> merge) with
    # a one-rank communicator, to measure its per-step overhead on a single-GPU box
    force_sharded = world == 1 and os.environ.get("RAGFIN_FORCE_SHARDED") == "1"
    if args.streams <= 0:
        args.streams = 8 if (world > 1 or force_sharded) else 4
    if force_sharded:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29533", rank=0, world_size=1,
                                device_id=dev)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # RAGFIN_DIST_BACKEND=gloo + RAGFIN_SHARE_GPU=1: rehearsal of the N>1 code path
        # on a single-GPU box (all ranks on cuda:0, collectives over gloo)
        backend = os.environ.get("RAGFIN_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from oracle import search as osearch
    from rag_fin_amd.sharded import HipShardBackend, ShardedSearcher
    from rag_fin_amd.store import GpuIndex

    dim, B, k = args.dim, args.batch, args.topk
    if args.scaling == "strong" and world > 1:
        lo, hi = ShardedSearcher.shard_bounds(args.rows, world, rank)
        rows, row_base, rows_total = hi - lo, lo, args.rows
    else:
        rows, row_base, rows_total = args.rows, rank * args.rows, args.rows * world
    # synthetic data, SURVEY.md 8d recipe: N(0,1) rows, L2-normalised, fp16 (each rank its own shard)
    c16 = osearch.synth_unit_rows(rows, dim, 1234 + rank)
    q16 = osearch.synth_unit_rows(B, dim, 5678)
    index = GpuIndex(dim, rows, dev)
    step_rows = 1 << 18
    for s in range(0, rows, step_rows):
        index.add(torch.from_numpy(c16[s:s + step_rows]).to(dev))
    q = torch.from_numpy(q16).to(dev)
    torch.cuda.synchronize()

    searcher = ShardedSearcher(HipShardBackend(index), row_base=row_base) if (world > 1 or force_sharded) else None
    if force_sharded:
        searcher.force_collective = True
    direct_rccl = False
    # (RAGFIN_DIRECT_RCCL=force tries it under the gloo rehearsal too: two ranks on ONE GPU make
    # ncclCommInitRank fail with "duplicate GPU" AFTER the bootstrap exchange -- a check that the
    # unique id really travels -- and the agreed fallback below takes over)
    want_direct = os.environ.get("RAGFIN_DIRECT_RCCL", "1")
    if searcher is not None and (want_direct == "force" or (want_direct == "1" and (
            force_sharded or os.environ.get("RAGFIN_DIST_BACKEND", "nccl") == "nccl"))):
        # the step's all-gather straight through librccl (ctypes) on the lane's stream: torch's wrapper
        # costs ~25 us of host time per call, which bounds the strong-scaled job from 4 GPUs on
        try:
            direct_rccl = searcher.enable_direct_rccl(dev)
        except Exception as e:   # fall back to torch.distributed -- on EVERY rank (agreed below)
            print(f"[bench] direct RCCL unavailable on rank {rank}: {e}", file=sys.stderr)
            direct_rccl = False
        if world > 1:
            ok = torch.tensor([1 if direct_rccl else 0], dtype=torch.int32, device=dev)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok.item()) == 0 and searcher.rccl is not None:
                searcher.rccl.destroy()
                searcher.rccl = None
            direct_rccl = bool(int(ok.item()))
    # N > 1 (or the one-rank rehearsal) with the direct RCCL binding: a step is three bare ctypes
    # enqueues on the lane's stream (ShardedSearcher.search_on), no torch call on the hot path
    bare_enqueues = searcher is not None and (direct_rccl or (world == 1 and not force_sharded)) and \
        os.environ.get("RAGFIN_BARE_ENQUEUES", "1") == "1"
    # `streams` batches in flight: each has its own HIP stream, workspace and output
    # buffers; the corpus index is immutable and shared.  Step i runs on lane i % lanes.
    max_lanes = max(1, args.streams)
    lanes = []
    for i in range(max_lanes):
        lanes.append(dict(
            stream=torch.cuda.Stream(device=dev) if i > 0 else torch.cuda.current_stream(),
            ws=index.workspace if i == 0 else index.new_workspace(),
            out=(torch.empty((B, k), dtype=torch.float32, device=dev),
                 torch.empty((B, k), dtype=torch.int64, device=dev),
                 torch.empty((B, k), dtype=torch.float64, device=dev),
                 torch.empty((B,), dtype=torch.int32, device=dev))))
    out = lanes[0]["out"]
    if searcher is None and os.environ.get("RAGFIN_BARE_ENQUEUES", "1") == "1":
        from ctypes import c_void_p
        for l in lanes:
            o = l["out"]
            l["bare"] = (q.data_ptr(), B, k, 0, o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), o[3].data_ptr(),
                         l["ws"].data_ptr(), c_void_p(l["stream"].cuda_stream))

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # setup, not warm-up: first use of every lane (code objects, kernel attributes, the one-time
    # zeroing of each workspace's control block, RCCL's first collective) happens here, so that
    # short runs (--steps 5 --warmup 1) time steady-state steps too
    for l in lanes:
        with torch.cuda.stream(l["stream"]):
            for _ in range(2):
                if searcher is not None:
                    searcher.search(q, k, workspace=l["ws"])
                else:
                    index.search_raw(q, k, want_exact=True, out=l["out"], workspace=l["ws"])
    barrier()

    def timed_region(n_lanes):
        """W untimed + exactly K timed steps, barrier + synchronize on both sides."""
        counter = [0]
        res = [None]

        def step():
            lane = lanes[counter[0] % n_lanes]
            counter[0] += 1
            if searcher is not None and bare_enqueues:
                lane["res"] = searcher.search_on(q, k, lane["ws"], lane["stream"])
                res[0] = lane["res"]
                return
            if searcher is None and lane.get("bare") is not None:
                index.enqueue_search(*lane["bare"])   # one ctypes call, cached pointers, the lane's stream
                return
            with torch.cuda.stream(lane["stream"]):
                if searcher is not None:
                    lane["res"] = searcher.search(q, k, workspace=lane["ws"])
                    res[0] = lane["res"]
                else:
                    index.search_raw(q, k, want_exact=True, out=lane["out"], workspace=lane["ws"])

        # --group (experiment): the lanes' per-shard top-k share ONE all-gather
        # (ShardedSearcher.search_group); `n` steps = n // lanes groups + a remainder of single steps
        grouped = searcher is not None and n_lanes > 1 and args.group
        side = [torch.cuda.Stream(device=dev) for _ in range(n_lanes)] if grouped else None

        def run(n):
            if not grouped:
                for _ in range(n):
                    step()
                return
            for _ in range(n // n_lanes):
                g = searcher.search_group([q] * n_lanes, k, [l["ws"] for l in lanes[:n_lanes]], side)
                res[0] = (g[0][-1], g[1][-1], g[2][-1])
                for i in range(n_lanes):
                    lanes[i]["res"] = (g[0][i], g[1][i], g[2][i])
            for _ in range(n % n_lanes):
                step()
        run(args.warmup)
        barrier()
        t0 = time.perf_counter()
        run(args.steps)
        barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt, res[0]

    # serial region first (one batch at a time: the latency view), then the
    # pipelined region that `value` reports when --streams > 1
    serial_s, res = timed_region(1)
    elapsed, n_lanes = serial_s, 1
    if max_lanes > 1:
        elapsed, res = timed_region(max_lanes)
        n_lanes = max_lanes

    # flags must be clean for the number to count as an exact search
    if searcher is None:
        flags_clean = all(int(l["out"][3].abs().sum().item()) == 0 for l in lanes)
        same = all(torch.equal(l["out"][1], out[1]) and torch.equal(l["out"][2], out[2]) for l in lanes)
        flags_clean = flags_clean and same   # every lane answered the same queries identically
    else:
        done = [l["res"] for l in lanes if l.get("res") is not None]
        flags_clean = all(int(r[2].abs().sum().item()) == 0 for r in done) and \
            all(torch.equal(r[1], done[0][1]) for r in done)

    # N > 1: the merged (global) answer is checked against the oracle too -- every rank runs the
    # C oracle on ITS shard for the first queries, the per-shard oracle lists are all-gathered and
    # rank 0 merges them by (score desc, id asc): what an oracle over the whole corpus would return
    global_oracle = None
    if world > 1 and searcher is not None and not args.no_check:
        from oracle import c_oracle
        nq = min(8, B)
        os_l, oi_l = c_oracle.search(q16[:nq], c16, k)
        on_gpu = dist.get_backend() == "nccl"
        cdev = dev if on_gpu else torch.device("cpu")
        t_s = torch.from_numpy(np.ascontiguousarray(os_l, dtype=np.float64)).to(cdev)
        t_i = torch.from_numpy(np.ascontiguousarray(oi_l, dtype=np.int64) + row_base).to(cdev)
        # concatenated form [world * nq, k] (the one every backend takes), viewed as [world, nq, k]
        all_s = torch.empty((world * nq, k), dtype=torch.float64, device=cdev)
        all_i = torch.empty((world * nq, k), dtype=torch.int64, device=cdev)
        dist.all_gather_into_tensor(all_s, t_s.contiguous())
        dist.all_gather_into_tensor(all_i, t_i.contiguous())
        global_oracle = (all_s.cpu().numpy().reshape(world, nq, k), all_i.cpu().numpy().reshape(world, nq, k))

    result = None
    if rank == 0:
        ms_per_step = elapsed * 1e3 / args.steps
        qps = B * args.steps / elapsed
        # dominant kernel (emit scan) timed with HIP events, stage by stage
        stages = [index.search_profile(q, k) for _ in range(max(10, min(50, args.steps)))]
        emit_ms = float(np.mean([s["emit"] for s in stages]))
        stage_avg = {n: float(np.mean([s[n] for s in stages])) for n in stages[0]}
        alg_bytes = rows * dim * 2  # corpus read once per batch
        achieved = alg_bytes / (emit_ms * 1e-3) / 1e9
        traffic = pmc_traffic(rows, dim, B)
        result = {
            "metric": "queries/sec, brute-force cosine/IP top-%d over a %s x %d-d fp16 corpus, "
                      "batch=%d (recall@10 vs CPU oracle reported alongside)" % (k, f"{rows_total:,}", dim, B),
            "value": round(qps, 1), "unit": "queries/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 5),
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f16",
            "data": "synthetic",
            "config": {"workload": "1M x 384-d fp16 corpus, batch-64 queries, top-10"
                       if (rows_total, dim, B, k) == (1_000_000, 384, 64, 10)
                       else f"{rows_total} x {dim}-d fp16 corpus ({rows} rows per GPU), batch-{B}, top-{k}",
                       "rows_per_gpu": rows, "rows_total": rows_total, "dim": dim, "batch": B,
                       "topk": k, "batches_in_flight": n_lanes,
                       "collective_api": ("ncclAllGather via ctypes" if direct_rccl else "torch.distributed")
                       if searcher is not None else None,
                       "collective": ("none" if searcher is None else
                                      ("one all-gather per %d batches" % n_lanes if (n_lanes > 1 and args.group)
                                       else "one all-gather per batch")),
                       "sharding": ("none" if not force_sharded else "one-rank RCCL all-gather (overhead rehearsal)")
                       if world == 1 else f"rows/{world} + RCCL all-gather"},
            "rows_per_s": round(rows_total * args.steps / elapsed, 1),
            "serial": {"batches_in_flight": 1, "value": round(B * args.steps / serial_s, 1),
                       "ms_per_step": round(serial_s * 1e3 / args.steps, 5),
                       "whole_step_GBps": round(alg_bytes / (serial_s / args.steps) / 1e9, 1)},
            "whole_step_GBps": round(alg_bytes / (ms_per_step * 1e-3) / 1e9, 1),
            "stage_ms": {n: round(v, 5) for n, v in stage_avg.items()},
            "flags_clean": flags_clean,
            "roofline": {"bound": "hbm", "kernel": "k_scan<MODE_EMIT>", "achieved": round(achieved, 1),
                         "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic,
                         "algorithmic_bytes_per_launch": alg_bytes},
        }

        # ---- BASELINE.json configs[2]: the same corpus at batch 256 (one wide sweep per step,
        # scan_wide.hip).  AI = 256 flop/B: the matrix pipe, at the clock the chip holds under
        # MFMA load, is the binding roof there; both fractions are reported.  Serial steps.
        if world == 1 and dim == 384 and not args.no_batch256:
            try:
                B2 = 256
                q2 = torch.from_numpy(osearch.synth_unit_rows(B2, dim, 5679)).to(dev)
                for _ in range(5):
                    r2 = index.search_raw(q2, k, want_exact=True)
                torch.cuda.synchronize()
                n2 = max(20, args.steps // 4)
                t0 = time.perf_counter()
                for _ in range(n2):
                    r2 = index.search_raw(q2, k, want_exact=True)
                torch.cuda.synchronize()
                dt2_serial = (time.perf_counter() - t0) / n2
                # two batches in flight (two of the lanes above: own stream, workspace and outputs):
                # threshold and merge of one batch run beside the other's sweep
                from ctypes import c_void_p as _vp
                lanes2 = []
                for l in lanes[:int(os.environ.get("RAGFIN_B256_LANES", "2"))]:
                    o = (torch.empty((B2, k), dtype=torch.float32, device=dev), torch.empty((B2, k), dtype=torch.int64, device=dev),
                         torch.empty((B2, k), dtype=torch.float64, device=dev), torch.empty((B2,), dtype=torch.int32, device=dev))
                    lanes2.append((o, (q2.data_ptr(), B2, k, 0, o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(),
                                       o[3].data_ptr(), l["ws"].data_ptr(), _vp(l["stream"].cuda_stream))))
                dt2 = dt2_serial
                b256_same = True
                if len(lanes2) >= 2:
                    for i in range(6):
                        index.enqueue_search(*lanes2[i % len(lanes2)][1])
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for i in range(n2):
                        index.enqueue_search(*lanes2[i % len(lanes2)][1])
                    torch.cuda.synchronize()
                    dt2 = (time.perf_counter() - t0) / n2
                    same = all(bool((a == b).all().item()) for a, b in zip(lanes2[(n2 - 1) % len(lanes2)][0], r2))
                    b256_same = same
                flops2 = 2.0 * B2 * rows * dim
                result["batch256"] = {
                    "workload": "%s x %d-d fp16 corpus, batch-256 queries, top-%d (configs[2])" % (f"{rows:,}", dim, k),
                    "value": round(B2 / dt2, 1), "unit": "queries/s", "ms_per_step": round(dt2 * 1e3, 5),
                    "steps": n2, "batches_in_flight": len(lanes2), "serial_ms_per_step": round(dt2_serial * 1e3, 5),
                    "flags_clean": int(r2[3].abs().sum().item()) == 0, "in_flight_matches_serial": b256_same,
                    "whole_step_GBps": round(alg_bytes / dt2 / 1e9, 1),
                    "hbm_frac": round(alg_bytes / dt2 / 1e9 / HBM_PEAK_GBPS, 4),
                    "whole_step_TFLOPs": round(flops2 / dt2 / 1e12, 1),
                    "mfma_frac_of_2500_dense_f16": round(flops2 / dt2 / 1e12 / 2500.0, 4)}
            except Exception as e:   # a side measurement must never cost the headline line
                result["batch256"] = {"error": repr(e)}

        # ---- the same step for a caller that hands over HOST buffers (not `value`: reported beside
        # it): queries come from pinned host memory and scores + ids go back to pinned host memory,
        # on the lane's stream, four batches in flight
        if world == 1 and searcher is None:
            try:
                hl = []
                for l in lanes:
                    o = l["out"]
                    hl.append(dict(stream=l["stream"], ws=l["ws"], out=o, q_dev=torch.empty_like(q),
                                   q_host=q.cpu().pin_memory(), s_host=torch.empty((B, k), dtype=torch.float32).pin_memory(),
                                   i_host=torch.empty((B, k), dtype=torch.int64).pin_memory()))

                def host_step(i):
                    l = hl[i % len(hl)]
                    with torch.cuda.stream(l["stream"]):
                        l["q_dev"].copy_(l["q_host"], non_blocking=True)
                        index.search_raw(l["q_dev"], k, want_exact=True, out=l["out"], workspace=l["ws"])
                        l["s_host"].copy_(l["out"][0], non_blocking=True)
                        l["i_host"].copy_(l["out"][1], non_blocking=True)
                for i in range(2 * len(hl)):
                    host_step(i)
                torch.cuda.synchronize()
                nh = max(20, args.steps // 2)
                t0 = time.perf_counter()
                for i in range(nh):
                    host_step(i)
                torch.cuda.synchronize()
                dth = (time.perf_counter() - t0) / nh
                result["pcie_inclusive"] = {
                    "value": round(B / dth, 1), "unit": "queries/s", "ms_per_step": round(dth * 1e3, 5), "steps": nh,
                    "batches_in_flight": len(hl),
                    "what": "pinned-host queries in (%d B), scores + ids out to pinned host (%d B) per step" % (
                        B * dim * 2, B * k * 12),
                    "ids_match_device_path": bool((hl[(nh - 1) % len(hl)]["i_host"] == out[1].cpu()).all().item())}
            except Exception as e:   # a side measurement must never cost the headline line
                result["pcie_inclusive"] = {"error": repr(e)}

        # ---- correctness beside the number: recall@10 / exact ids vs the CPU oracle
        if not args.no_check:
            from oracle import c_oracle
            nq = min(8, B)
            if searcher is None:
                gi = out[1][:nq].cpu().numpy()
                ge = out[2][:nq].cpu().numpy()
            else:
                s_, i_, e_, f_ = index.search_raw(q[:nq], k, want_exact=True)
                gi, ge = i_.cpu().numpy(), e_.cpu().numpy()
            os_, oi = c_oracle.search(q16[:nq], c16, k)
            recall = float(np.mean([len(set(gi[b]) & set(oi[b])) / k for b in range(nq)]))
            result["recall_at_10"] = recall
            result["ids_ranks_exact"] = bool(np.array_equal(gi, oi))
            result["max_abs_score_err"] = float(np.abs(ge - os_).max())
            result["checked_queries"] = nq
            if global_oracle is not None and res is not None:
                gs, gid = global_oracle   # [world, nq, k]
                exp_s, exp_i = osearch.merge_shards(gs, gid, k)   # (score desc, id asc), tests/test_oracle_search.py
                got_i = res[1][:nq].cpu().numpy()
                got_s = res[0][:nq].cpu().numpy().astype(np.float64)
                result["global_ids_ranks_exact"] = bool(np.array_equal(got_i, exp_i))
                result["global_recall_at_10"] = float(np.mean([len(set(got_i[b]) & set(exp_i[b])) / k for b in range(nq)]))
                result["global_max_abs_score_err"] = float(np.abs(got_s - exp_s).max())

        # ---- CPU baseline leg (rank 0, N=1 only): the oracle's BLAS restatement of the
        # reference's search semantics on the host cores, same data
        if world == 1 and not args.no_cpu_baseline:
            c32 = c16.astype(np.float32)
            osearch.cpu_search_blas(q16, c32, k)  # warm-up
            times = []
            budget = time.perf_counter() + 20.0
            while len(times) < 5 and (len(times) < 2 or time.perf_counter() < budget):
                t = time.perf_counter()
                osearch.cpu_search_blas(q16, c32, k)
                times.append(time.perf_counter() - t)
            med = float(np.median(times))
            try:
                avail = len(os.sched_getaffinity(0))
            except AttributeError:
                avail = os.cpu_count()
            cores = avail
            try:   # the threads the BLAS pool actually runs (it may be capped below the core count)
                from threadpoolctl import threadpool_info
                pools = [p_["num_threads"] for p_ in threadpool_info() if p_.get("user_api") == "blas"]
                if pools:
                    cores = min(avail, max(pools))
            except Exception:
                pass
            result["cpu_baseline"] = {
                "value": round(B / med, 1), "unit": "queries/s", "cores": cores, "kind": "port",
                "sample": f"{len(times)} full batches of the same workload ({rows} x {dim}, batch {B}, "
                          f"top-{k}); numpy float32 BLAS matmul + argpartition (oracle/search.py "
                          f"cpu_search_blas), median {med * 1e3:.1f} ms/batch"}
        sys.stdout.flush()
        os.dup2(json_fd, 1)
        print(json.dumps(result), flush=True)
        os.dup2(2, 1)
    if world > 1:
        dist.barrier()
    if searcher is not None and searcher.rccl is not None:
        torch.cuda.synchronize()
        searcher.rccl.destroy()
    if world > 1 or force_sharded:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""bench.py -- benchmark of the rag-fin vector-retrieval hot path on MI355X.

Headline metric (BASELINE.json): queries/sec, top-10, 1M x 384-d fp16 corpus, batch=64, with
recall@10 vs the CPU oracle.  A "step" is one pass of the hot path over one batch: 64 query
embeddings -> brute-force cosine/IP top-10 over the HBM-resident corpus -> ranked (score, row
id) lists.

  python bench.py --gpus 1 --steps 50 --warmup 5
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

ONE JSON line on stdout:
  value / ms_per_step   exactly --steps timed steps between barrier + synchronize pairs, 4 batches
                        in flight, inputs (queries too) resident in HBM when the region starts
  survey_8d             the same step as SURVEY.md 8d words the metric: queries uploaded from
                        pinned host memory and scores + ids downloaded to pinned host memory
                        inside the step, median of >= 30 timed groups (PCIe-inclusive; reported
                        beside `value`, never as it)
  serial, stage_ms      one batch at a time (latency view); per-stage HIP-event times
  roofline              dominant kernel (emit sweep) timed live with HIP events on its stream;
                        `traffic` is looked up in a committed rocprofv3 --pmc pass
                        (`traffic_source` names the file; it is not measured by this run)
  configs               N = 1: BASELINE.json configs[1] (100 k x 384, B 64), configs[2]
                        (1 M x 384, B 256) and configs[3] (encode 10 k chunks + search), each with
                        its own value, roofline, cpu_baseline and oracle check
  cpu_baseline          numpy BLAS port of the reference's search on the host cores (N = 1)
  recall / exactness    ALL queries of the batch against the C oracle
N > 1: the corpus is row-sharded, the query batch replicated, each step ends with ONE RCCL
all-gather of the per-shard top-k + flags and a merge (rag_fin_amd/sharded.py).  Default =
STRONG scaling of the metric's 1 M-row corpus; the same line carries `weak` = BASELINE
configs[4]'s shard (1.25 M x 768 rows PER GPU, generated on the device) so that one SCALE run
answers both questions.  `--scaling weak` makes the weak job the headline instead.
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

HBM_PEAK_GBPS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_F16_PEAK_TFLOPS = 2500.0   # dense fp16/bf16 MFMA peak (same guide)


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
                         "0 = 4 on one GPU, 8 for the sharded step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-check", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="skip the configs[1..3] / weak side measurements")
    return ap.parse_args()


def pmc_traffic(rows, dim, batch):
    """HBM bytes per emit-scan launch from the newest committed PMC pass taken on this workload
    (profiles/*_pmc*.json; tools/collect_pmc.sh + tools/pmc_summary.py).  -> (bytes, file) | (None, None)"""
    try:
        import glob
        for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc*.json")), reverse=True):
            rec = json.load(open(path))
            if rec.get("rows") == rows and rec.get("dim") == dim and rec.get("batch") == batch:
                return rec.get("hbm_bytes_per_launch"), os.path.relpath(path, ROOT)
    except Exception:
        pass
    return None, None


def limit_host_pools():
    """Size the BLAS / OpenMP / torch intra-op pools by the CPUs this process may BURN (a container's CFS quota
    can sit far below its affinity mask: rag_fin_amd/hostcpu.py), so that the CPU legs are not throttled and
    `cores` says what really ran.  Returns (budget, the threadpoolctl limiter to keep alive)."""
    from rag_fin_amd.hostcpu import cpu_budget
    budget = cpu_budget()
    keep = None
    try:
        from threadpoolctl import threadpool_limits
        keep = threadpool_limits(limits=budget)
    except Exception:
        pass
    try:
        import torch
        torch.set_num_threads(budget)
    except Exception:
        pass
    return budget, keep


def host_cores():
    from rag_fin_amd.hostcpu import cpu_budget
    avail = cpu_budget()
    cores = avail
    try:   # the threads the BLAS pool actually runs (it may be capped below the core count)
        from threadpoolctl import threadpool_info
        pools = [p_["num_threads"] for p_ in threadpool_info() if p_.get("user_api") == "blas"]
        if pools:
            cores = min(avail, max(pools))
    except Exception:
        pass
    return cores


def cpu_search_baseline(q16, c16, k, budget_s=12.0, max_reps=5):
    """numpy float32 BLAS matmul + argpartition (oracle/search.py cpu_search_blas) on the host cores."""
    from oracle import search as osearch
    c32 = c16.astype(np.float32)
    osearch.cpu_search_blas(q16, c32, k)  # warm-up
    times = []
    budget = time.perf_counter() + budget_s
    while len(times) < max_reps and (len(times) < 2 or time.perf_counter() < budget):
        t = time.perf_counter()
        osearch.cpu_search_blas(q16, c32, k)
        times.append(time.perf_counter() - t)
    med = float(np.median(times))
    B = q16.shape[0]
    return {"value": round(B / med, 1), "unit": "queries/s", "cores": host_cores(), "kind": "port",
            "sample": f"{len(times)} full batches of the same workload ({c16.shape[0]} x {c16.shape[1]}, batch {B}, "
                      f"top-{k}); numpy float32 BLAS matmul + argpartition (oracle/search.py cpu_search_blas), "
                      f"median {med * 1e3:.1f} ms/batch"}


def check_against_oracle(ids, exact, q16, c16, k, rows=None):
    """ids / exact [nq, k] numpy vs the C oracle (all queries given)."""
    from oracle import c_oracle
    os_, oi = c_oracle.search(q16, c16, k)
    nq = q16.shape[0]
    return {"recall_at_%d" % k: float(np.mean([len(set(ids[b]) & set(oi[b])) / k for b in range(nq)])),
            "ids_ranks_exact": bool(np.array_equal(ids, oi)),
            "max_abs_score_err": float(np.abs(exact - os_).max()), "checked_queries": nq}


def pipelined(enqueue, n_lanes, steps, warm, sync):
    """wall-clock seconds per step of `enqueue(i)` with n_lanes batches in flight."""
    for i in range(warm):
        enqueue(i)
    sync()
    t0 = time.perf_counter()
    for i in range(steps):
        enqueue(i)
    sync()
    return (time.perf_counter() - t0) / steps


def gather_expected(local_scores, local_ids, world, k, cdev):
    """N > 1 answer check, shared by the headline job and the weak job: every rank contributes the
    reference top-k lists of ITS shard for the first nq queries (float64 scores, GLOBAL int64 ids),
    ONE all-gather each, and the lists are merged by (score desc, id asc) -- what a reference over the
    whole corpus returns.  COLLECTIVE; -> (scores [nq, k] float64, ids [nq, k] int64) on every rank."""
    import torch
    import torch.distributed as dist
    from oracle import search as osearch
    nq = int(local_scores.shape[0])
    t_s = torch.as_tensor(np.ascontiguousarray(local_scores, dtype=np.float64)).to(cdev).contiguous()
    t_i = torch.as_tensor(np.ascontiguousarray(local_ids, dtype=np.int64)).to(cdev).contiguous()
    all_s = torch.empty((world * nq, k), dtype=torch.float64, device=cdev)
    all_i = torch.empty((world * nq, k), dtype=torch.int64, device=cdev)
    dist.all_gather_into_tensor(all_s, t_s)
    dist.all_gather_into_tensor(all_i, t_i)
    return osearch.merge_shards(all_s.cpu().numpy().reshape(world, nq, k), all_i.cpu().numpy().reshape(world, nq, k), k)


def compare_global(got_scores, got_ids, exp_s, exp_i, k):
    """The merged answer of the sharded step (float32 scores, int64 global ids of the first nq queries)
    against gather_expected's lists -> the global_* fields of the JSON line."""
    nq = int(exp_i.shape[0])
    got_i = np.asarray(got_ids)[:nq]
    got_s = np.asarray(got_scores)[:nq].astype(np.float64)
    return {"global_ids_ranks_exact": bool(np.array_equal(got_i, exp_i)),
            "global_recall_at_10": float(np.mean([len(set(got_i[b]) & set(exp_i[b])) / k for b in range(nq)])),
            "global_max_abs_score_err": float(np.abs(got_s - exp_s.astype(np.float32).astype(np.float64)).max()),
            "global_checked_queries": nq}


def all_ranks_agree(flag: bool, world, cdev) -> bool:
    """AND of a per-rank boolean over the job (MIN all-reduce).  COLLECTIVE when world > 1."""
    if world <= 1:
        return bool(flag)
    import torch
    import torch.distributed as dist
    t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=cdev)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return bool(int(t.item()))


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
    cpu_budget_n, _pool_limit = limit_host_pools()
    # RAGFIN_FORCE_SHARDED=1 at N=1: run the N>1 code path (scan -> RCCL all-gather -> merge) with
    # a one-rank communicator, to measure its per-step overhead on a single-GPU box
    force_sharded = world == 1 and os.environ.get("RAGFIN_FORCE_SHARDED") == "1"
    if args.streams <= 0:
        args.streams = 8 if (world > 1 or force_sharded) else 4
    if force_sharded:
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29533", rank=0, world_size=1,
                                device_id=dev)
    if world > 1:
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
    sharded = world > 1 or force_sharded

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x):
        if world > 1:
            t = torch.tensor([x], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t.item())
        return x

    def make_searcher(index, row_base):
        s = ShardedSearcher(HipShardBackend(index), row_base=row_base)
        s.force_collective = force_sharded
        return s

    # the step's all-gather straight through librccl (ctypes) on the lane's stream: torch's wrapper
    # costs ~25 us of host time per call, which bounds the strong-scaled job from 4 GPUs on.
    # (RAGFIN_DIRECT_RCCL=force tries it under the gloo rehearsal too: two ranks on ONE GPU make
    # ncclCommInitRank fail with "duplicate GPU" AFTER the bootstrap exchange, and the agreed
    # fallback below takes over)
    rccl_comm = [None]
    direct_rccl = False
    want_direct = os.environ.get("RAGFIN_DIRECT_RCCL", "1")
    if sharded and (want_direct == "force" or (want_direct == "1" and (
            force_sharded or os.environ.get("RAGFIN_DIST_BACKEND", "nccl") == "nccl"))):
        try:
            from rag_fin_amd.rccl import RcclComm
            rccl_comm[0] = RcclComm(rank, world, dev, None)
            direct_rccl = True
        except Exception as e:   # fall back to torch.distributed -- on EVERY rank (agreed below)
            print(f"[bench] direct RCCL unavailable on rank {rank}: {e}", file=sys.stderr)
        if world > 1:
            ok = torch.tensor([1 if direct_rccl else 0], dtype=torch.int32, device=dev)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok.item()) == 0 and rccl_comm[0] is not None:
                rccl_comm[0].destroy()
                rccl_comm[0] = None
            direct_rccl = bool(int(ok.item()))

    def run_job(rows, row_base, rows_total, jdim, c16, q16, steps, warmup, max_lanes, label):
        """Build the (local shard of the) index, time `steps` steps serial and pipelined.
        -> dict with index, lanes, timings, last result; everything the JSON needs."""
        index = GpuIndex(jdim, rows, dev)
        if isinstance(c16, np.ndarray):
            step_rows = 1 << 18
            for s in range(0, rows, step_rows):
                index.add(torch.from_numpy(c16[s:s + step_rows]).to(dev))
        else:
            c16(index)   # generator callback (device-side synthesis)
        q = torch.from_numpy(q16).to(dev)
        torch.cuda.synchronize()
        searcher = make_searcher(index, row_base) if sharded else None
        if searcher is not None:
            searcher.rccl = rccl_comm[0]
        # N > 1 with the direct RCCL binding: a step is three bare ctypes enqueues on the lane's
        # stream (ShardedSearcher.search_on), no torch call on the hot path
        bare = searcher is not None and direct_rccl and os.environ.get("RAGFIN_BARE_ENQUEUES", "1") == "1"
        Bq = q.shape[0]
        lanes = []
        for i in range(max_lanes):
            lanes.append(dict(
                stream=torch.cuda.Stream(device=dev) if i > 0 else torch.cuda.current_stream(),
                ws=index.workspace if i == 0 else index.new_workspace(),
                out=(torch.empty((Bq, k), dtype=torch.float32, device=dev),
                     torch.empty((Bq, k), dtype=torch.int64, device=dev),
                     torch.empty((Bq, k), dtype=torch.float64, device=dev),
                     torch.empty((Bq,), dtype=torch.int32, device=dev))))
        if searcher is None:
            from ctypes import c_void_p
            for l in lanes:
                o = l["out"]
                l["bare"] = (q.data_ptr(), Bq, k, 0, o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(),
                             o[3].data_ptr(), l["ws"].data_ptr(), c_void_p(l["stream"].cuda_stream))
        # setup, not warm-up: first use of every lane (code objects, kernel attributes, RCCL's first
        # collective) happens here, so that short runs (--steps 5 --warmup 1) time steady-state steps
        for l in lanes:
            with torch.cuda.stream(l["stream"]):
                for _ in range(2):
                    if searcher is not None:
                        searcher.search(q, k, workspace=l["ws"], resolve=False)
                    else:
                        index.search_raw(q, k, want_exact=True, out=l["out"], workspace=l["ws"])
        barrier()

        def timed_region(n_lanes):
            """W untimed + exactly K timed steps, barrier + synchronize on both sides."""
            counter = [0]

            def step():
                lane = lanes[counter[0] % n_lanes]
                counter[0] += 1
                if searcher is not None and bare:
                    lane["res"] = searcher.search_on(q, k, lane["ws"], lane["stream"])
                elif searcher is not None:
                    with torch.cuda.stream(lane["stream"]):
                        lane["res"] = searcher.search(q, k, workspace=lane["ws"], resolve=False)
                else:
                    index.enqueue_search(*lane["bare"])   # one ctypes call, cached pointers, the lane's stream
            for _ in range(warmup):
                step()
            barrier()
            t0 = time.perf_counter()
            for _ in range(steps):
                step()
            barrier()
            return max_over_ranks(time.perf_counter() - t0)

        serial_s = timed_region(1)
        elapsed, n_lanes = serial_s, 1
        if max_lanes > 1:
            elapsed = timed_region(max_lanes)
            n_lanes = max_lanes
        # flags must be clean for the number to count as an exact search: every lane, every rank
        # (the sharded step's flags are already the OR over the shards; the reduction below also
        # covers a rank whose lanes disagree)
        if searcher is None:
            clean = all(int(l["out"][3].abs().sum().item()) == 0 for l in lanes) and \
                all(torch.equal(l["out"][1], lanes[0]["out"][1]) and torch.equal(l["out"][2], lanes[0]["out"][2])
                    for l in lanes)
            res = None
        else:
            done = [l["res"] for l in lanes if l.get("res") is not None]
            clean = all(int(r[2].abs().sum().item()) == 0 for r in done) and \
                all(torch.equal(r[1], done[0][1]) for r in done)
            res = done[-1]
        clean = all_ranks_agree(clean, world, dev)
        return dict(index=index, q=q, searcher=searcher, lanes=lanes, serial_s=serial_s, elapsed=elapsed,
                    n_lanes=n_lanes, flags_clean=clean, res=res, bare=bare, label=label)

    def global_check(job, q16, c16_local, row_base, nq):
        """N > 1: every rank runs the C oracle on ITS shard for the first nq queries, the per-shard
        oracle lists are all-gathered and rank 0 merges them by (score desc, id asc): what an
        oracle over the whole corpus returns -- compared with the merged answer of the last step."""
        from oracle import c_oracle
        os_l, oi_l = c_oracle.search(q16[:nq], c16_local, k)
        cdev = dev if dist.get_backend() == "nccl" else torch.device("cpu")
        exp_s, exp_i = gather_expected(os_l, np.asarray(oi_l, dtype=np.int64) + row_base, world, k, cdev)
        if rank != 0 or job["res"] is None:
            return {}
        return compare_global(job["res"][0].cpu().numpy(), job["res"][1].cpu().numpy(), exp_s, exp_i, k)

    # ---- the headline job ---------------------------------------------------------------------
    if args.scaling == "strong" and world > 1:
        lo, hi = ShardedSearcher.shard_bounds(args.rows, world, rank)
        rows, row_base, rows_total = hi - lo, lo, args.rows
        c16 = osearch.synth_unit_rows(rows, dim, 1234 + rank)   # SURVEY.md 8d recipe, each rank its own shard
    else:
        rows, row_base, rows_total = args.rows, rank * args.rows, args.rows * world
        c16 = osearch.synth_unit_rows(rows, dim, 1234 + rank)
    q16 = osearch.synth_unit_rows(B, dim, 5678)
    job = run_job(rows, row_base, rows_total, dim, c16, q16, args.steps, args.warmup, max(1, args.streams), "headline")
    gcheck = global_check(job, q16, c16, row_base, min(16, B)) if (world > 1 and not args.no_check) else {}

    index, q, lanes = job["index"], job["q"], job["lanes"]
    result = None
    if rank == 0:
        elapsed, serial_s, n_lanes = job["elapsed"], job["serial_s"], job["n_lanes"]
        ms_per_step = elapsed * 1e3 / args.steps
        qps = B * args.steps / elapsed
        # dominant kernel (emit scan) timed with HIP events on its stream, stage by stage
        stages = [index.search_profile(q, k) for _ in range(max(10, min(50, args.steps)))]
        emit_ms = float(np.mean([s["emit"] for s in stages]))
        stage_avg = {n: float(np.mean([s[n] for s in stages])) for n in stages[0]}
        alg_bytes = rows * dim * 2  # corpus read once per batch
        achieved = alg_bytes / (emit_ms * 1e-3) / 1e9
        traffic, traffic_src = pmc_traffic(rows, dim, B)
        result = {
            "metric": "queries/sec, brute-force cosine/IP top-%d over a %s x %d-d fp16 corpus, "
                      "batch=%d (recall@10 vs CPU oracle reported alongside)" % (k, f"{rows_total:,}", dim, B),
            "value": round(qps, 1), "unit": "queries/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 5),
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f16",
            "data": "synthetic",
            "value_basis": "inputs resident in HBM when the timed region starts (bench contract); the figure "
                           "SURVEY.md 8d words -- query upload and result download inside the step -- is survey_8d.value",
            "config": {"workload": "1M x 384-d fp16 corpus, batch-64 queries, top-10"
                       if (rows_total, dim, B, k) == (1_000_000, 384, 64, 10)
                       else f"{rows_total} x {dim}-d fp16 corpus ({rows} rows per GPU), batch-{B}, top-{k}",
                       "rows_per_gpu": rows, "rows_total": rows_total, "dim": dim, "batch": B,
                       "topk": k, "batches_in_flight": n_lanes,
                       "collective_api": ("ncclAllGather via ctypes" if direct_rccl else "torch.distributed")
                       if sharded else None,
                       "collective": "none" if not sharded else "one all-gather per batch ({score, id}[B,k] + flags[B])",
                       "sharding": ("none" if not force_sharded else "one-rank RCCL all-gather (overhead rehearsal)")
                       if world == 1 else f"rows/{world} + RCCL all-gather"},
            "rows_per_s": round(rows_total * args.steps / elapsed, 1),
            "serial": {"batches_in_flight": 1, "value": round(B * args.steps / serial_s, 1),
                       "ms_per_step": round(serial_s * 1e3 / args.steps, 5),
                       "whole_step_GBps": round(alg_bytes / (serial_s / args.steps) / 1e9, 1)},
            "whole_step_GBps": round(alg_bytes / (ms_per_step * 1e-3) / 1e9, 1),
            "stage_ms": {n: round(v, 5) for n, v in stage_avg.items()},
            "flags_clean": job["flags_clean"],
            "roofline": {"bound": "hbm", "kernel": "k_scan<MODE_EMIT>", "achieved": round(achieved, 1),
                         "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic,
                         "traffic_source": (traffic_src + " (committed rocprofv3 --pmc pass of this workload; "
                                            "not measured by this run)") if traffic_src else None,
                         "algorithmic_bytes_per_launch": alg_bytes,
                         "timing": "HIP events around the kernel on its stream (rf_search_profile), mean of %d" % len(stages)},
        }
        result.update(gcheck)

        # ---- SURVEY.md 8d wording of the metric: queries come from pinned host memory and scores +
        # ids go back to pinned host memory inside the step; median of >= 30 timed groups
        if world == 1 and not sharded:
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
                group = 8 * len(hl)
                pipelined(host_step, len(hl), group, group, torch.cuda.synchronize)
                per = [pipelined(host_step, len(hl), group, 0, torch.cuda.synchronize) for _ in range(30)]
                dth = float(np.median(per))
                result["survey_8d"] = {
                    "value": round(B / dth, 1), "unit": "queries/s", "ms_per_step": round(dth * 1e3, 5),
                    "timing": "median of 30 groups of %d steps, %d batches in flight" % (group, len(hl)),
                    "what": "pinned-host queries in (%d B), scores + ids out to pinned host (%d B) per step" % (
                        B * dim * 2, B * k * 12),
                    "ids_match_device_path": bool((hl[-1]["i_host"] == lanes[0]["out"][1].cpu()).all().item())}
            except Exception as e:   # a side measurement must never cost the headline line
                result["survey_8d"] = {"error": repr(e)}

        # ---- correctness beside the number: ALL queries vs the CPU oracle (local shard)
        if not args.no_check:
            if job["searcher"] is None:
                gi = lanes[0]["out"][1].cpu().numpy()
                ge = lanes[0]["out"][2].cpu().numpy()
            else:
                s_, i_, e_, f_ = index.search_raw(q, k, want_exact=True)
                gi, ge = i_.cpu().numpy(), e_.cpu().numpy()
            chk = check_against_oracle(gi, ge, q16, c16, k)
            result["recall_at_10"] = chk["recall_at_%d" % k]
            result["ids_ranks_exact"] = chk["ids_ranks_exact"]
            result["max_abs_score_err"] = chk["max_abs_score_err"]
            result["checked_queries"] = chk["checked_queries"]

        # ---- CPU baseline leg (rank 0, N=1 only): the oracle's BLAS restatement of the
        # reference's search semantics on the host cores, same data
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_search_baseline(q16, c16, k, budget_s=20.0)

    # ---- the other single-GPU configs of BASELINE.json, each with its own roofline / baseline / check
    if world == 1 and not sharded and not args.no_configs and (rows, dim, B) == (1_000_000, 384, 64):
        configs = []
        for fn in (config2, config3, config4):
            try:
                configs.append(fn(args, dev, index, lanes, c16, k))
            except Exception as e:   # a side measurement must never cost the headline line
                configs.append({"workload": fn.__name__, "error": repr(e)})
        result["configs"] = configs

    # ---- N > 1: the weak-scaling job (BASELINE configs[4]: 1.25 M x 768 rows PER GPU) in the same line
    if world > 1 and not args.no_configs and args.scaling == "strong":
        try:
            weak = weak_job(args, dev, rank, world, run_job, k)
            if rank == 0:
                result["weak"] = weak
        except Exception as e:
            if rank == 0:
                result["weak"] = {"error": repr(e)}

    if rank == 0:
        sys.stdout.flush()
        os.dup2(json_fd, 1)
        print(json.dumps(result), flush=True)
        os.dup2(2, 1)
    if world > 1:
        dist.barrier()
    if rccl_comm[0] is not None:
        torch.cuda.synchronize()
        rccl_comm[0].destroy()
    if world > 1 or force_sharded:
        dist.destroy_process_group()


# =============================================================================================
# BASELINE.json configs[1]: synthetic 100 k x 384 corpus, batch-64, top-10
# =============================================================================================
def config2(args, dev, index1m, lanes, c16_1m, k):
    import torch
    from ctypes import c_void_p
    from oracle import search as osearch
    from rag_fin_amd.store import GpuIndex
    n, dim, B = 100_000, 384, 64
    c16 = osearch.synth_unit_rows(n, dim, 1234)
    q16 = osearch.synth_unit_rows(B, dim, 5678)
    ix = GpuIndex(dim, n, dev)
    ix.add(torch.from_numpy(c16).to(dev))
    q = torch.from_numpy(q16).to(dev)
    ls = []
    for l in lanes:
        o = (torch.empty((B, k), dtype=torch.float32, device=dev), torch.empty((B, k), dtype=torch.int64, device=dev),
             torch.empty((B, k), dtype=torch.float64, device=dev), torch.empty((B,), dtype=torch.int32, device=dev))
        ws = ix.new_workspace()
        ls.append((o, (q.data_ptr(), B, k, 0, o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), o[3].data_ptr(),
                       ws.data_ptr(), c_void_p(l["stream"].cuda_stream)), ws))
    steps = max(200, args.steps)
    dt = pipelined(lambda i: ix.enqueue_search(*ls[i % len(ls)][1]), len(ls), steps, 2 * len(ls), torch.cuda.synchronize)
    dt1 = pipelined(lambda i: ix.enqueue_search(*ls[0][1]), 1, steps, 4, torch.cuda.synchronize)
    stages = [ix.search_profile(q, k) for _ in range(30)]
    emit_ms = float(np.mean([s["emit"] for s in stages]))
    alg = n * dim * 2
    out = {"workload": "100k x 384-d fp16 corpus, batch-64 queries, top-10 (BASELINE configs[1])",
           "value": round(B / dt, 1), "unit": "queries/s", "ms_per_step": round(dt * 1e3, 5), "steps": steps,
           "batches_in_flight": len(ls), "serial_ms_per_step": round(dt1 * 1e3, 5),
           "stage_ms": {n_: round(float(np.mean([s[n_] for s in stages])), 5) for n_ in stages[0]},
           "flags_clean": all(int(o[3].abs().sum().item()) == 0 for o, _, _ in ls),
           "roofline": {"bound": "hbm", "kernel": "k_scan<MODE_EMIT>", "achieved": round(alg / emit_ms / 1e6, 1),
                        "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(alg / emit_ms / 1e6 / HBM_PEAK_GBPS, 4),
                        "algorithmic_bytes_per_launch": alg,
                        "note": "9.6 us of streaming at the HBM rate: the launch is latency-bound (prologue, "
                                "first HBM round trip, drain), not bandwidth-bound"},
           "whole_step_GBps": round(alg / dt / 1e9, 1)}
    if not args.no_check:
        out.update(check_against_oracle(ls[0][0][1].cpu().numpy(), ls[0][0][2].cpu().numpy(), q16, c16, k))
    if not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_search_baseline(q16, c16, k, budget_s=5.0)
    return out


# =============================================================================================
# BASELINE.json configs[2]: the 1 M x 384 corpus at batch 256 (one wide sweep per step) --
# AI = 256 flop/B: HBM and the matrix pipe are both near their roofs; both fractions are reported
# =============================================================================================
def config3(args, dev, index, lanes, c16, k):
    import torch
    from ctypes import c_void_p
    from oracle import search as osearch
    B, dim = 256, 384
    rows = c16.shape[0]
    q16 = osearch.synth_unit_rows(B, dim, 5679)
    q = torch.from_numpy(q16).to(dev)
    ls = []
    for l in lanes[:int(os.environ.get("RAGFIN_B256_LANES", "2"))]:
        o = (torch.empty((B, k), dtype=torch.float32, device=dev), torch.empty((B, k), dtype=torch.int64, device=dev),
             torch.empty((B, k), dtype=torch.float64, device=dev), torch.empty((B,), dtype=torch.int32, device=dev))
        ls.append((o, (q.data_ptr(), B, k, 0, o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), o[3].data_ptr(),
                       l["ws"].data_ptr(), c_void_p(l["stream"].cuda_stream))))
    steps = max(60, args.steps // 2)
    dt1 = pipelined(lambda i: index.enqueue_search(*ls[0][1]), 1, steps, 6, torch.cuda.synchronize)
    dt = pipelined(lambda i: index.enqueue_search(*ls[i % len(ls)][1]), len(ls), steps, 6, torch.cuda.synchronize)
    same = all(torch.equal(ls[0][0][1], o[1]) and torch.equal(ls[0][0][2], o[2]) for o, _ in ls)
    stages = [index.search_profile(q, k) for _ in range(30)]
    emit_ms = float(np.mean([s["emit"] for s in stages]))
    alg = rows * dim * 2
    flops = 2.0 * B * rows * dim
    out = {"workload": "%s x %d-d fp16 corpus, batch-256 queries, top-%d (BASELINE configs[2])" % (f"{rows:,}", dim, k),
           "value": round(B / dt, 1), "unit": "queries/s", "ms_per_step": round(dt * 1e3, 5), "steps": steps,
           "batches_in_flight": len(ls), "serial_ms_per_step": round(dt1 * 1e3, 5),
           "stage_ms": {n_: round(float(np.mean([s[n_] for s in stages])), 5) for n_ in stages[0]},
           "flags_clean": all(int(o[3].abs().sum().item()) == 0 for o, _ in ls), "in_flight_lanes_agree": same,
           "roofline": {"bound": "hbm", "kernel": "k_scan_ldsdma<MODE_EMIT> (wide sweep)",
                        "achieved": round(alg / emit_ms / 1e6, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                        "frac": round(alg / emit_ms / 1e6 / HBM_PEAK_GBPS, 4), "algorithmic_bytes_per_launch": alg,
                        "mfma": {"achieved": round(flops / emit_ms / 1e9, 1), "peak": MFMA_F16_PEAK_TFLOPS,
                                 "unit": "TFLOP/s", "frac": round(flops / emit_ms / 1e9 / MFMA_F16_PEAK_TFLOPS, 4),
                                 "flops_per_launch": flops}},
           "whole_step_GBps": round(alg / dt / 1e9, 1), "whole_step_hbm_frac": round(alg / dt / 1e9 / HBM_PEAK_GBPS, 4),
           "whole_step_TFLOPs": round(flops / dt / 1e12, 1)}
    if not args.no_check:
        nq = 32   # 32 of the 256 queries through the C oracle over the full 1 M rows (a few seconds of host time)
        sel = np.arange(0, B, B // nq)
        out.update(check_against_oracle(ls[0][0][1].cpu().numpy()[sel], ls[0][0][2].cpu().numpy()[sel], q16[sel], c16, k))
    if not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_search_baseline(q16, c16, k, budget_s=10.0, max_reps=3)
    return out


# =============================================================================================
# BASELINE.json configs[3]: encode 10 k finance chunks on the GPU + top-10 search
# =============================================================================================
def config4(args, dev, index1m, lanes, c16_1m, k):
    import torch
    from oracle import c_oracle, encoder as oenc, encoder_torch, synth_text
    from rag_fin_amd.embedder import Embedder
    from rag_fin_amd.store import GpuIndex
    from rag_fin_amd.tokenizer import WordPieceTokenizer
    n = 10_000
    cfg = dict(oenc.MINILM_L6)
    w = oenc.random_weights(cfg, 0)
    tok = WordPieceTokenizer(synth_text.vocab_for(size=cfg["vocab_size"]))
    emb = Embedder(w, cfg, tokenizer=tok, device=dev)
    texts = synth_text.retemplated_texts(n, 11)
    ids_all, lens_all = tok.batch_native(texts, 256)
    tokens = int(lens_all.sum())
    flops = float(sum(oenc.flops_per_token(cfg, int(l)) * int(l) for l in lens_all))
    # (a) from token ids already on the device, length-bucketed (the encoder alone)
    order = np.argsort(lens_all, kind="stable")
    batches, i = [], 0
    while i < n:
        j = i
        while j < n and (j - i + 1) * int(lens_all[order[j]]) <= 65536:
            j += 1
        j = max(j, i + 1)
        idx = order[i:j]
        T = int(lens_all[idx].max())
        batches.append((torch.from_numpy(np.ascontiguousarray(ids_all[idx, :T])).to(dev),
                        torch.from_numpy(lens_all[idx]).to(dev), torch.as_tensor(idx, device=dev)))
        i = j
    out16 = torch.empty((n, 384), dtype=torch.float16, device=dev)

    def run_ids():
        for ids, ln, idx in batches:
            out16[idx] = emb.encode_ids(ids, ln)
    run_ids()
    torch.cuda.synchronize()
    t_ids = []
    for _ in range(5):
        t = time.perf_counter()
        run_ids()
        torch.cuda.synchronize()
        t_ids.append(time.perf_counter() - t)
    enc_s = float(np.median(t_ids))
    # (b) from TEXT: native tokenizer -> chunked pinned uploads -> bucketed encode (the ingest path)
    emb.encode_to_device(texts[:256])
    torch.cuda.synchronize()
    t_txt = []
    for _ in range(3):
        t = time.perf_counter()
        vec = emb.encode_to_device(texts)
        torch.cuda.synchronize()
        t_txt.append(time.perf_counter() - t)
    txt_s = float(np.median(t_txt))
    host_stages = {k_: (round(v, 5) if isinstance(v, float) else v) for k_, v in getattr(emb, "ingest_stats", {}).items()}
    # search over the 10 k corpus: 64 unseen chunk texts as queries
    ix = GpuIndex(384, n, dev)
    ix.add(vec)
    q16 = emb.encode_to_device(synth_text.retemplated_texts(64, 12))
    for _ in range(3):
        s_, i_, e_, f_ = ix.search_raw(q16, k, want_exact=True)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(100):
        s_, i_, e_, f_ = ix.search_raw(q16, k, want_exact=True)
    torch.cuda.synchronize()
    search_ms = (time.perf_counter() - t) / 100 * 1e3
    out = {"workload": "encode 10k finance chunks (re-templated golden chunks, %d tokens, MiniLM-L6 architecture, seeded "
                       "random weights) + top-10 search over them (BASELINE configs[3])" % tokens,
           "value": round(tokens / enc_s, 1), "unit": "tokens/s", "chunks_per_s": round(n / enc_s, 1),
           "encode_s": round(enc_s, 5), "batches": len(batches), "steps": len(t_ids),
           "from_text": {"texts_per_s": round(n / txt_s, 1), "tokens_per_s": round(tokens / txt_s, 1),
                         "text_to_embedding_s": round(txt_s, 5), "host_threads": len(os.sched_getaffinity(0)),
                         "host_cpu_budget": host_cores(),
                         "what": "native WordPiece tokenizer + chunked pinned uploads + bucketed encode",
                         "host_stage_s": host_stages,
                         "host_stage_note": "host seconds of the last run per stage (the GPU work is only enqueued); "
                                            "their sum against text_to_embedding_s says how much of the ingest the host serialises"},
           "search_batch64_ms": round(search_ms, 5),
           "end_to_end_embed_plus_search_s": round(txt_s + search_ms * 1e-3, 5),
           "roofline": {"bound": "mfma", "kernel": "rf_encode (all kernels of the forward)",
                        "achieved": round(flops / enc_s / 1e12, 1), "peak": MFMA_F16_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": round(flops / enc_s / 1e12 / MFMA_F16_PEAK_TFLOPS, 4), "flops": flops,
                        "flops_per_token": "21.2 MFLOP + 4 T 384 x 6 attention (oracle.encoder.flops_per_token)"}}
    if not args.no_check:
        # search: every query vs the C oracle on the stored vectors; encoder: 16 rows vs the float64 oracle
        c16 = vec.cpu().numpy()
        out.update(check_against_oracle(i_.cpu().numpy(), e_.cpu().numpy(), q16.cpu().numpy(), c16, k))
        sub = np.arange(0, n, n // 16)[:16]
        want = oenc.encode(oenc.round_weights_fp16(w), cfg, ids_all[sub], lens_all[sub])
        out["encoder_max_abs_err_vs_f64_oracle"] = float(np.abs(c16[sub].astype(np.float64) - want).max())
        out["encoder_checked_rows"] = int(len(sub))
    if not args.no_cpu_baseline:
        # the reference's ingest on the host: transformers.BertModel fp32 on torch-CPU (the model class
        # sentence-transformers wraps), batches of 32, same weights; a bounded sample of the same texts
        torch_threads = torch.get_num_threads()
        model = encoder_torch.build_bert(cfg, w)
        m = 256
        encoder_torch.encode(model, ids_all[:32], lens_all[:32])
        t = time.perf_counter()
        ref = encoder_torch.encode(model, ids_all[:m], lens_all[:m])
        cpu_s = time.perf_counter() - t
        out["cpu_baseline"] = {"value": round(float(lens_all[:m].sum()) / cpu_s, 1), "unit": "tokens/s",
                               "cores": min(torch_threads, len(os.sched_getaffinity(0))), "kind": "port",
                               "sample": "%d of the 10 000 texts (%d tokens) through transformers.BertModel fp32 on "
                                         "torch-CPU (the model class SentenceTransformer wraps; batches of 32, mean-pool, "
                                         "normalise; oracle/encoder_torch.py), %.1f s" % (m, int(lens_all[:m].sum()), cpu_s),
                               "max_abs_diff_gpu_vs_cpu_fp32": float(np.abs(vec[:m].float().cpu().numpy() - ref).max())}
    return out


# =============================================================================================
# N > 1: BASELINE.json configs[4] -- 10 M x 768 over 8 GPUs = 1.25 M x 768 rows PER GPU (weak scaling)
# =============================================================================================
def weak_job(args, dev, rank, world, run_job, k):
    import torch
    from oracle import search as osearch
    rows, dim, B = 1_250_000, 768, 64

    def synth(index):   # on the device, seed = 1234 + rank: never 15 GB on the host
        gen = torch.Generator(device=dev).manual_seed(1234 + rank)
        step = 1 << 17
        for s in range(0, rows, step):
            m = min(step, rows - s)
            x = torch.randn((m, dim), device=dev, generator=gen)
            index.add(torch.nn.functional.normalize(x, dim=1).half())
    q16 = osearch.synth_unit_rows(B, dim, 5678)
    steps = max(30, args.steps // 2)
    job = run_job(rows, rank * rows, rows * world, dim, synth, q16, steps, max(5, args.warmup // 2), max(1, args.streams), "weak")
    # check: 4 queries through the EXHAUSTIVE kernel on every shard (an independent fp64 path), gathered
    # and merged on the host by (score desc, id asc) -- what the sharded step must return
    import torch.distributed as dist
    nq = 4
    ix, q = job["index"], job["q"]
    s_, i_, e_ = ix.search_exhaustive(q[:nq].contiguous(), k, id_base=rank * rows, want_exact=True)
    cdev = dev if dist.get_backend() == "nccl" else torch.device("cpu")
    exp_s, exp_i = gather_expected(e_.cpu().numpy(), i_.cpu().numpy(), world, k, cdev)
    if rank != 0:
        return None
    got_i = job["res"][1][:nq].cpu().numpy()
    stages = [ix.search_profile(q, k) for _ in range(20)]
    emit_ms = float(np.mean([s["emit"] for s in stages]))
    alg = rows * dim * 2
    ms = job["elapsed"] * 1e3 / steps
    return {"workload": "%d x 768-d fp16 corpus = 1.25 M rows per GPU x %d GPUs, batch-64, top-%d (BASELINE configs[4]; "
                        "weak scaling: ideal = flat value)" % (rows * world, world, k),
            "scaling": "weak", "value": round(B * steps / job["elapsed"], 1), "unit": "queries/s", "steps": steps,
            "ms_per_step": round(ms, 5), "rows_per_s": round(rows * world * steps / job["elapsed"], 1),
            "batches_in_flight": job["n_lanes"], "serial_ms_per_step": round(job["serial_s"] * 1e3 / steps, 5),
            "flags_clean": job["flags_clean"], "global_ids_ranks_exact": bool(np.array_equal(got_i, exp_i)),
            "global_checked_queries": nq, "check": "vs rf_search_exhaustive on every shard, merged on the host",
            "roofline": {"bound": "hbm", "kernel": "k_scan<MODE_EMIT> (dim 768)", "achieved": round(alg / emit_ms / 1e6, 1),
                         "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(alg / emit_ms / 1e6 / HBM_PEAK_GBPS, 4),
                         "algorithmic_bytes_per_launch": alg, "per": "GPU"}}


if __name__ == "__main__":
    main()

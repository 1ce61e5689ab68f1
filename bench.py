#!/usr/bin/env python3
"""bench.py -- all-pairs HLL distance throughput on MI355X (BASELINE.json metric).

Workload (config.workload): BASELINE.json configs[2], the configuration the metric is quoted
on: 10 000 synthetic sketches, p=14 (16 KiB register arrays), all-pairs Jaccard with dashing's
default estimator (Ertl MLE), packed float32 upper triangle.  One "step" = one full pass of the
hot path over register arrays already resident in HBM: cardinalities + bit-plane transform +
all-pairs AND/popcount + per-pair estimator -> distances in HBM (+ for N>1 the RCCL gather of
the per-rank row spans to rank 0).  Strong scaling: the matrix is fixed, rows are sharded.

Prints ONE JSON line on rank 0.  `roofline` is for the dominant kernel (k_pair_counts), timed
with HIP events on the library's own stream; `cpu_baseline` is the CPU oracle (a restatement
of the reference algorithm and row schedule -- the reference itself is not buildable: its
bonsai/sketch submodules are absent) timed on this host on a bounded sample of the same rows.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_SKETCH = int(os.environ.get("DSH_BENCH_N", "10000"))
P = int(os.environ.get("DSH_BENCH_P", "14"))
K = 31
HBM_PEAK_GBS = 8000.0  # MI355X spec, /opt/skills/guides/MI355X_MICROARCH.md


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary p=10 workload line")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    import dashing_amd
    from dashing_amd import multigpu, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # DSH_BENCH_BACKEND=gloo is a dry-run of the N>1 code path on a box with ONE GPU: all ranks share
    # cuda:0 and the gather is staged through host memory.  Never used for reported numbers.
    backend = os.environ.get("DSH_BENCH_BACKEND", "nccl")
    # DSH_BENCH_FORCE_DIST=1 runs the sharded code path (process group, gather, un-permute) even with
    # one rank -- a functional check of the RCCL plumbing on a 1-GPU box, not a reported configuration.
    multi = world > 1 or bool(os.environ.get("DSH_BENCH_FORCE_DIST"))
    if multi:
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "gloo":
            local_rank = 0
            torch.cuda.set_device(0)
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    n, p, m = N_SKETCH, P, 1 << P
    regs_h = synth.survey_sketches(n, p, seed=0x5EED0000)[0]  # SURVEY 8d; identical bytes on every rank
    regs_d = torch.from_numpy(regs_h).to(dev)                 # resident in HBM before timing
    total_pairs = n * (n - 1) // 2
    ctx = dashing_amd.Context(local_rank)
    for kv in filter(None, os.environ.get("DSH_BENCH_OPTS", "").split(",")):  # tuning sweeps, e.g. "kc=64,emax=32"
        k_, v_ = kv.split("=")
        ctx.set_option(k_, int(v_))
    ctx.attach_device(regs_d.data_ptr(), n, p)
    # N>1: every rank holds all sketches and computes its cost-balanced share of the sorted-order triangle
    # (contiguous spans); the only exchange is the RCCL gather of the spans to rank 0, which un-permutes
    # them once into dashing's packed order.  DSH_BENCH_PIECES=K cuts a rank's share into K shards and
    # gathers piece h asynchronously while piece h+1 is computed (multigpu.PipelinedShards).  On one GPU
    # every extra piece costs ~0.25 ms (smaller launches); the estimated gain is ~0.5 ms at 2 ranks (one xGMI
    # link carries 100 MB), ~0.15 ms at 4 and nothing at 8 -- too little to make it the default before it
    # has run on real links, so the default is one piece and a plain gather.
    pieces = 1
    if multi:
        pieces = max(1, int(os.environ.get("DSH_BENCH_PIECES", "1")))
    nshards = world * pieces
    span_off = ctx.shard_plan(nshards) if multi else [0, total_pairs]
    span = span_off[(rank + 1) * pieces] - span_off[rank * pieces] if multi else total_pairs
    final = torch.empty(total_pairs, dtype=torch.float32, device=dev) if multi and rank == 0 else None
    out_d = stage = pipe = None
    if not multi:
        out_d = torch.empty(max(total_pairs, 1), dtype=torch.float32, device=dev)
    elif pieces == 1 and backend == "nccl":
        mx = max(max(span_off[r + 1] - span_off[r] for r in range(world)), 1)
        out_d = torch.empty(mx, dtype=torch.float32, device=dev)
        stage = torch.empty(world * mx, dtype=torch.float32, device=dev) if rank == 0 else None
    else:
        # gloo dry-run: the pieces travel through host memory (ranks share one GPU)
        pipe = multigpu.PipelinedShards(span_off, rank, world, pieces, dev if backend == "nccl" else torch.device("cpu"))
        gpu_out = [torch.empty(pipe.mx[h], dtype=torch.float32, device=dev) for h in range(pieces)] if backend != "nccl" else None

    def compute_piece(h):
        buf = pipe.out(h) if backend == "nccl" else gpu_out[h]
        ctx.dist_shard_device(buf.data_ptr(), pipe.shard(h), nshards, dashing_amd.ESTIM_ERTL_MLE, dashing_amd.JI, K)
        ctx.synchronize()
        if backend != "nccl":
            pipe.out(h).copy_(buf)

    def step():
        # re-attach: invalidates cached planes/cardinalities, so every step is a full pass
        ctx.attach_device(regs_d.data_ptr(), n, p)
        if not multi:
            ctx.dist_rows_device(out_d.data_ptr(), 0, n, dashing_amd.ESTIM_ERTL_MLE, dashing_amd.JI, K)
            ctx.synchronize()
            return out_d[:span]
        if pipe is None:
            ctx.dist_shard_device(out_d.data_ptr(), rank, world, dashing_amd.ESTIM_ERTL_MLE, dashing_amd.JI, K)
            ctx.synchronize()
            # the gathered (padded) blocks are un-permuted where they arrived: no back-to-back copy
            st = multigpu.gather_shard_spans(out_d, span_off, rank, world, stage, None, 0, staged=True)
            # NCCL collectives only enqueue: wait on the host until this rank's part is done (rank 0 must
            # see the data, the others must not overwrite out_d in the next step while it is being sent)
            torch.cuda.current_stream().synchronize()
            if rank == 0:
                ctx.unpermute_staged_device(st.data_ptr(), mx, world, final.data_ptr())
                return final
            return None
        for h in range(pieces):
            compute_piece(h)
            pipe.submit(h)  # async gather of piece h; the next piece is computed meanwhile
        got = pipe.wait()
        torch.cuda.current_stream().synchronize()  # as above: the gathers have completed on this rank
        if rank == 0:
            st, block_off = got
            if backend != "nccl":
                st = st.to(dev)
                torch.cuda.current_stream().synchronize()
            ctx.unpermute_blocks_device(st.data_ptr(), block_off, final.data_ptr())
            return final
        return None

    def fence():
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        full = step()
    fence()
    dt = time.perf_counter() - t0
    if multi:
        t = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = dt / args.steps * 1e3
    value = total_pairs * args.steps / dt

    # ---- roofline of the dominant kernel: HIP events on the library stream, outside the timed region
    ctx.set_profiling(True)
    pair_ms, launches, fin_ms, prep_ms = 0.0, 0, 0.0, 0.0
    reps = 3
    for _ in range(reps):
        ctx.attach_device(regs_d.data_ptr(), n, p)
        calls = [None] if not multi else ([(out_d, rank, world)] if pipe is None else
                                         [((pipe.out(h) if backend == "nccl" else gpu_out[h]), pipe.shard(h), nshards) for h in range(pieces)])
        for call in calls:
            if call is None:
                ctx.dist_rows_device(out_d.data_ptr(), 0, n, dashing_amd.ESTIM_ERTL_MLE, dashing_amd.JI, K)
            else:
                ctx.dist_shard_device(call[0].data_ptr(), call[1], call[2], dashing_amd.ESTIM_ERTL_MLE, dashing_amd.JI, K)
            ctx.synchronize()
            k = ctx.last_kernel_ms()
            pair_ms += k["pair_ms"]
            fin_ms += k["finalize_ms"]
            prep_ms += k["prepare_ms"]
            launches += k["pair_launches"]
    ctx.set_profiling(False)
    traffic = None  # HBM bytes per launch from the committed PMC passes of the same workload
    try:
        pm = json.load(open(os.path.join(ROOT, "profiles", "pmc_pair_kernel.json")))
        if pm["workload"]["n_sketches"] == n and pm["workload"]["p"] == p and world == 1:
            traffic = pm["hbm_bytes_per_launch"]
    except (OSError, KeyError, ValueError):
        pass
    b_pair = 2 * m + 4                                   # algorithmic bytes per pair
    my_pairs = span
    achieved = my_pairs * reps * b_pair / (pair_ms * 1e-3) / 1e9 if pair_ms > 0 else 0.0
    roofline = {
        "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
        "traffic_note": "bytes/launch, rocprofv3 PMC (2*FETCH_SIZE + WRITE_SIZE), profiles/pmc_pair_kernel.json; collected in separate --pmc passes, not in this run",
        "kernel": "k_pair_counts", "launches_per_step": launches // reps,
        "avg_launch_ms": round(pair_ms / max(launches, 1), 4),
        "bytes_per_pair": b_pair, "pairs_per_launch_avg": my_pairs * reps // max(launches, 1),
        "dense_planes": ctx.info("planes"), "reg_value_range": [ctx.info("vlo"), ctx.info("vhi")],
        "exception_list_cap": ctx.info("emax"), "avg_planes_per_tile": ctx.info("avg_tile_planes_x100") / 100.0,
        "sorted_columns": bool(ctx.info("sorted")),
        "finalize_ms_per_step": round(fin_ms / reps, 4), "prepare_ms_per_step": round(prep_ms / reps, 4),
        "note": "streaming-model bytes (2*2^p+4 per pair); >1.0 is possible because LDS tiles reuse each staged sketch",
        # SURVEY.md 8d also asks for the physical HBM rate and the compulsory floor (inputs once + outputs once)
        "physical_hbm_gbs": round(traffic / (pair_ms / max(launches, 1) * 1e-3) / 1e9, 1) if traffic and pair_ms > 0 else None,
        "compulsory_bytes_per_step": n * m + 4 * total_pairs,
    }

    # what actually bounds the kernel: integer VALU issue (v_and_b32 + v_bcnt_u32_b32 per 32 pair-bits,
    # 4 cycles per wave64 instruction => 256 CUs x 4 SIMDs x 16 lanes x 2.4 GHz = 3.93e13 lane-ops/s)
    lane_ops = 2.0 * ctx.info("avg_tile_planes_x100") / 100.0 * ctx.info("words_per_plane") * ctx.info("tiles") * 128 * 128
    roofline["valu_int"] = {"achieved_lane_ops_per_s": lane_ops * reps / (pair_ms * 1e-3) if pair_ms > 0 else 0.0,
                            "peak_lane_ops_per_s": 3.93e13,
                            "frac": round(lane_ops * reps / (pair_ms * 1e-3) / 3.93e13, 4) if pair_ms > 0 else 0.0,
                            "note": "the binding resource of k_pair_counts (DESIGN.md 3.2); PMC SQ_INSTS_VALU in profiles/r1g agrees"}
    cpu = None
    parity = None
    if rank == 0 and multi:
        # assembled multi-rank matrix vs one single-GPU call on rank 0 (outside the timed region)
        ref = torch.empty(total_pairs, dtype=torch.float32, device=dev)
        ctx.attach_device(regs_d.data_ptr(), n, p)
        ctx.dist_rows_device(ref.data_ptr(), 0, n, dashing_amd.ESTIM_ERTL_MLE, dashing_amd.JI, K)
        ctx.synchronize()
        parity = {"assembled_equals_single_gpu": bool(torch.equal(ref, full)), "pairs_checked": total_pairs}
    if rank == 0 and not multi and not args.no_cpu_baseline:
        cpu, parity = cpu_baseline(regs_h, full, n, p, args.cpu_seconds)

    line = None
    if rank == 0:
        line = {
            "metric": "genome-pairs/sec, all-pairs HLL Jaccard (Ertl-MLE), N=%d p=%d" % (n, p),
            "value": value, "unit": "pairs/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "u8/u32 popcount + f64 estimator",
            "data": "synthetic",
            "config": {"workload": "BASELINE configs[2]: %d synthetic sketches, p=%d (%d B each), all-pairs dist on MI355X" % (n, p, m),
                       "n_sketches": n, "p": p, "k": K, "estimator": "ERTL_MLE", "result": "JI",
                       "sharding": ("cost-balanced row shards of the sorted-order triangle (%d per rank, gathered piece by piece while the next is computed), RCCL gather to rank 0 + un-permute" % pieces) if multi else "single GPU"},
            "roofline": roofline,
            "cpu_baseline": cpu,
            "parity_vs_cpu": parity,
        }
    ctx.close()
    # The JSON line must be the LAST thing on stdout: RCCL (NCCL_DEBUG=VERSION is exported on the GPU boxes)
    # prints its banner through C stdio, which is block-buffered when stdout is a pipe and would otherwise
    # come out at process exit, after the line.  Flush C stdio on every rank, tear the group down, then print.
    import ctypes

    def flush_c_stdio():
        try:
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass

    if multi:
        flush_c_stdio()
        dist.barrier()
        dist.destroy_process_group()
    flush_c_stdio()
    if line is not None:
        sys.stdout.flush()
        print(json.dumps(line), flush=True)


def cpu_baseline(regs_h, gpu_full, n, p, seconds):
    """Time the CPU oracle (reference algorithm + row schedule) on a bounded sample of rows.
    The oracle is only the checker/baseline here -- never the thing measured as `value`."""
    import subprocess

    from oracle import oracle_c

    cores = oracle_c.effective_cpus()
    native = "/tmp/liboracle_native_%d.so" % os.getpid()
    kind_lib = None
    try:
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "native", "OUT=" + native],
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        kind_lib = oracle_c.load(native, threads=cores)
        isa = "-march=native"
    except Exception:
        kind_lib = oracle_c.load(threads=cores)
        isa = "-march=x86-64-v3 (prebuilt)"
    # calibrate on a few rows, then size the sample for ~`seconds`
    t0 = time.perf_counter()
    r0 = oracle_c.dist_rows(regs_h, 0, 8, lib=kind_lib)
    t_cal = time.perf_counter() - t0
    rate = r0.size / max(t_cal, 1e-6)
    rows = int(min(n - 1, max(16, seconds * rate / n)))
    t0 = time.perf_counter()
    ref = oracle_c.dist_rows(regs_h, 0, rows, lib=kind_lib)
    t = time.perf_counter() - t0
    got = gpu_full[: ref.size].cpu().numpy()
    rel = np.abs(got.astype(np.float64) - ref) / np.maximum(np.abs(ref), 1e-9)
    cpu = {"value": ref.size / t, "unit": "pairs/s", "cores": cores, "kind": "port",
           "sample": "rows [0,%d) of the same %d-sketch p=%d matrix = %d pairs in %.1f s; oracle/dsh_oracle.c (%s, OpenMP dynamic over j per row as src/sketch_and_cmp.h:699-710)" % (rows, n, p, ref.size, t, isa),
           "note": "CPU restatement; reference not buildable (bonsai/sketch submodules absent)"}
    parity = {"pairs_checked": int(ref.size), "max_rel_diff": float(rel.max()), "tolerance": 1e-6,
              "exact_float32_matches": int((got == ref).sum())}
    try:
        os.unlink(native)
    except OSError:
        pass
    return cpu, parity


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""bench.py -- all-pairs HLL distance throughput on MI355X (BASELINE.json metric).

Workload (config.workload): BASELINE.json configs[2], the configuration the metric is quoted
on: 10 000 synthetic sketches, p=14 (16 KiB register arrays), all-pairs Jaccard with dashing's
default estimator (Ertl MLE), packed float32 upper triangle.  One "step" = one full pass of the
hot path over register arrays already resident in HBM: per-sketch pass (cardinalities, exception
lists) + bit-plane transform + all-pairs AND/popcount + per-pair estimator -> distances in HBM.
N>1 (strong scaling, the matrix is fixed): rank r computes the rows [b_r, b_{r+1}) of the triangle
(tile-aligned bounds balanced by tile count, dsh_balance_rows; the plane matrix is laid out for that range, so the rank's result is ONE
contiguous span of the final packed matrix) and the only exchange is point-to-point: every rank
sends its span straight into its place on rank 0 (RCCL over xGMI) -- no collective inside the
compare, no un-permute.  The exchange runs through the library's own C-ABI (dsh_comm_init + the pipelined
dsh_dist_rows_parts_device_async / dsh_collect_parts_async: part q of a rank's rows travels on the copy stream
while the later parts are still being finalized); `python bench.py --gpus N` launches its N ranks itself.

Prints ONE JSON line on rank 0.  `roofline` describes the dominant kernel (k_pair_counts_ls / k_pair_counts), timed
with HIP events on the library's own stream; `cpu_baseline` is the CPU oracle (a restatement of the
reference algorithm and row schedule with an AVX-512BW / AVX2 histogram-of-max -- the reference
itself is not buildable: its bonsai/sketch submodules are absent) timed on this host on a bounded
sample of the same rows; `secondary` is the same pass on a configs[3]-shaped matrix (100 000 x p=10),
where the per-pair estimator, not the popcounts, is the hot kernel.
"""
import argparse
import hashlib
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
CLOCK_HZ = 2.4e9       # max shader clock (same guide); cycle figures below are "wall time x 2.4 GHz"
N_SIMD = 256 * 4


def source_hash():
    """sha256 over the device sources (same function as tools/pmc_collect.py): PMC files measured on other
    sources are refused."""
    h = hashlib.sha256()
    for rel in ("dashing_amd/csrc/kernels_compare.hip", "dashing_amd/csrc/kernels_sketch.hip",
                "dashing_amd/csrc/estimators.h", "dashing_amd/csrc/kernels.h", "dashing_amd/csrc/consts.h", "dashing_amd/csrc/ctx.h",
                "dashing_amd/csrc/plan.h", "dashing_amd/csrc/plan.cpp", "dashing_amd/csrc/engine.hip", "dashing_amd/csrc/abi.hip",
                "dashing_amd/csrc/knn.hip", "dashing_amd/csrc/exchange.hip"):
        with open(os.path.join(ROOT, rel), "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def measure_kernels(ctx, regs_d, n, p, calls, reps=3):
    """HIP-event times of the three phases on the library's own stream, summed over `reps` full passes"""
    import dashing_amd

    ctx.set_profiling(True)
    acc = {"pair_ms": 0.0, "finalize_ms": 0.0, "prepare_ms": 0.0, "pair_launches": 0}
    for _ in range(reps):
        ctx.attach_device(regs_d.data_ptr(), n, p)
        for (ptr, rb, re) in calls:
            ctx.dist_rows_device(ptr, rb, re, dashing_amd.ESTIM_ERTL_MLE, dashing_amd.JI, K)
            ctx.synchronize()
            k = ctx.last_kernel_ms()
            for key in acc:
                acc[key] += k[key]
    ctx.set_profiling(False)
    return acc


def live_pmc_traffic(n, p):
    """roofline.traffic measured by THIS run: two child processes of bench.py under `rocprofv3 --pmc` (FETCH_SIZE, then
    WRITE_SIZE -- separate passes, never combined with a trace, as MI355X_MICROARCH.md prescribes), each running the same
    workload for one warm-up and two timed steps; per launch of the tile kernel, bytes = (2 * FETCH_SIZE + WRITE_SIZE) KiB
    (gfx950 counts wide coalesced reads at half their bytes).  Returns (bytes or None, note)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile

    if shutil.which("rocprofv3") is None:
        return None, "rocprofv3 not on PATH"
    # never nest: when this process is itself being profiled (rocprofv3 / rocprof preload their tool library and export
    # ROCPROF* / ROCP_TOOL* variables) a second profiler in a child would inherit that environment
    if any(k.startswith(("ROCPROF", "ROCP_TOOL", "ROCPROFILER")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        return None, "this run is itself being profiled"
    vals = {}
    work = tempfile.mkdtemp(prefix="dsh_pmc_", dir="/tmp")
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(work, counter)
            os.makedirs(d)
            argv = ["rocprofv3", "--pmc", counter, "-d", d, "-o", "p", "--output-format", "csv", "--", sys.executable,
                    os.path.abspath(__file__), "--no-cpu-baseline", "--no-secondary", "--no-pmc", "--steps", "2", "--warmup", "1"]
            env = dict(os.environ, TMPDIR="/tmp", DSH_BENCH_N=str(n), DSH_BENCH_P=str(p))
            for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
                env.pop(k, None)
            try:
                r = subprocess.run(argv, env=env, cwd="/tmp", stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=150)
            except (OSError, subprocess.TimeoutExpired) as e:
                return None, "rocprofv3 --pmc %s pass did not finish (%s)" % (counter, type(e).__name__)
            got = []
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if "k_pair_counts" in row["Kernel_Name"] and "mfma" not in row["Kernel_Name"] and row["Counter_Name"] == counter:
                        got.append(float(row["Counter_Value"]))
            if r.returncode != 0 or not got:
                return None, "rocprofv3 --pmc %s pass: rc %d, %d tile-kernel rows" % (counter, r.returncode, len(got))
            vals[counter] = sum(got) / len(got)
    finally:
        shutil.rmtree(work, ignore_errors=True)
    return (2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0, (
        "bytes/launch measured in this run: two child runs of this script under rocprofv3 --pmc (FETCH_SIZE, WRITE_SIZE; separate "
        "passes, no trace), mean over the tile-kernel launches; (2*FETCH_SIZE + WRITE_SIZE) KiB as MI355X_MICROARCH.md prescribes for gfx950")


def pair_slots(ctx):
    """wave-level (AND, BCNT) slots of the last pair-kernel schedule: per tile, planes x words x 128x128 pairs / 64 lanes"""
    return ctx.info("avg_tile_planes_x100") / 100.0 * ctx.info("words_per_plane") * ctx.info("tiles") * 128 * 128 / 64.0


def self_launch(n_gpus, backend):
    """`python bench.py --gpus N` (N > 1) without a launcher: re-exec under torch.distributed.run, one rank per GPU on
    this node (the same command line the driver uses).  Fails -- never falls back to fewer GPUs -- when the node shows
    fewer than N devices (backend gloo = dry run of the N-rank code path with every rank on cuda:0, never reported)."""
    import socket
    import subprocess

    import dashing_amd

    have = dashing_amd.device_count()
    need = 1 if backend == "gloo" else n_gpus
    if have < need:
        sys.stderr.write("bench.py: --gpus %d needs %d visible gfx950 device(s), found %d\n" % (n_gpus, need, have))
        return 3
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    argv = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_gpus),
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(argv, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary p=10 workload line")
    ap.add_argument("--no-pmc", action="store_true", help="do not run the two rocprofv3 --pmc child passes for roofline.traffic")
    args = ap.parse_args()
    if args.gpus < 1:
        sys.exit("bench.py: --gpus must be >= 1")
    backend = os.environ.get("DSH_BENCH_BACKEND", "nccl")
    if "WORLD_SIZE" not in os.environ:
        if args.gpus > 1:
            sys.exit(self_launch(args.gpus, backend))  # one rank per GPU under torch.distributed.run; never a silent 1-GPU run
    elif int(os.environ["WORLD_SIZE"]) != args.gpus:
        sys.exit("bench.py: launched with WORLD_SIZE=%s but --gpus %d: refusing to report a mislabelled run" % (os.environ["WORLD_SIZE"], args.gpus))

    import torch
    import torch.distributed as dist

    import dashing_amd
    from dashing_amd import multigpu, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # DSH_BENCH_BACKEND=gloo is a dry-run of the N>1 code path on a box with ONE GPU: all ranks share
    # cuda:0 and the spans travel through host memory.  Never used for reported numbers.
    # DSH_BENCH_FORCE_DIST=1 runs the distributed code path (process group, exchange) even with one rank --
    # a functional check of the RCCL plumbing on a 1-GPU box, not a reported configuration.
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
            if torch.cuda.device_count() < world:
                sys.exit("bench.py: %d ranks but only %d visible GPUs" % (world, torch.cuda.device_count()))
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

    # The exchange of the spans: through the C-ABI (dsh_comm_init / dsh_collect_spans: RCCL inside libdashing_hip.so, on
    # the library's stream -- what a C++ host calls) unless DSH_BENCH_EXCHANGE=torch; if the library's communicator
    # cannot be brought up, torch.distributed's RCCL does the same point-to-point transfers (recorded in the line).
    exchange = "none"
    if multi:
        exchange = "torch.distributed" if backend == "nccl" else "gloo (host staged)"
        if backend == "nccl" and os.environ.get("DSH_BENCH_EXCHANGE", "cabi") == "cabi":
            try:
                multigpu.cabi_comm_init(ctx, rank, world)
                exchange = "c-abi rccl (dsh_collect_spans)"
            except Exception as e:  # noqa: BLE001
                sys.stderr.write("bench.py: rank %d: C-ABI communicator unavailable (%s): torch.distributed exchange\n" % (rank, e))
            flag = torch.tensor([1 if exchange.startswith("c-abi") else 0], device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)  # all ranks or none
            if int(flag.item()) == 0:
                exchange = "torch.distributed"
    use_cabi = exchange.startswith("c-abi")
    NPARTS = int(os.environ.get("DSH_BENCH_PARTS", "8"))
    if use_cabi:
        exchange = "c-abi rccl, pipelined in <= %d parts per rank (dsh_dist_rows_parts_device_async + dsh_collect_parts_async)" % NPARTS
    bounds = dashing_amd.balance_rows(n, world) if multi else [0, n]
    sizes = multigpu.span_sizes(n, bounds)
    offs = [0]
    for s_ in sizes:
        offs.append(offs[-1] + s_)
    my_pairs = sizes[rank] if multi else total_pairs
    host_stage = backend == "gloo" and multi
    final = None
    if rank == 0:
        final = torch.empty(max(total_pairs, 1), dtype=torch.float32, device=dev)
    # rank 0 computes in place (its rows are the head of the matrix); the others into a span-sized buffer
    local = final if rank == 0 else torch.empty(max(my_pairs, 1), dtype=torch.float32, device=dev)
    final_h = torch.empty(max(total_pairs, 1), dtype=torch.float32) if host_stage and rank == 0 else None
    phase = {"compute": 0.0, "exchange": 0.0}

    def step(timed=False):
        # re-attach: invalidates cached planes/cardinalities, so every step is a full pass
        t0 = time.perf_counter()
        ctx.attach_device(regs_d.data_ptr(), n, p)
        if use_cabi:
            # pipelined: the rank's rows in NPARTS parts; part q travels to rank 0 (copy stream, grouped ncclSend/ncclRecv
            # behind the part's event) while the later parts are still being finalized on the ctx stream
            ctx.dist_rows_parts_device_async(local.data_ptr(), bounds[rank], bounds[rank + 1], NPARTS, dashing_amd.ESTIM_ERTL_MLE, dashing_amd.JI, K)
            computed = ctx.event_record()
            ctx.collect_parts_async(n, bounds, NPARTS, 0 if rank == 0 else local.data_ptr(), final.data_ptr() if rank == 0 else 0, 0)
            ctx.event_wait(computed)
            t1 = time.perf_counter()
            ctx.wait()
            t2 = time.perf_counter()
            if timed:
                phase["compute"] += t1 - t0
                phase["exchange"] += t2 - t1  # what is left of the exchange after the last kernel
            return final
        ctx.dist_rows_device(local.data_ptr(), bounds[rank], bounds[rank + 1], dashing_amd.ESTIM_ERTL_MLE, dashing_amd.JI, K)
        ctx.synchronize()
        t1 = time.perf_counter()
        if multi:
            if host_stage:
                lh = local[: max(my_pairs, 1)].cpu()
                if rank == 0:
                    final_h[: sizes[0]] = lh[: sizes[0]]
                multigpu.collect_row_spans(lh, final_h, n, bounds, rank, world, 0)
                if rank == 0:
                    final.copy_(final_h)
                    torch.cuda.synchronize()
            else:
                multigpu.collect_row_spans(local, final, n, bounds, rank, world, 0)  # host-synchronised inside
        t2 = time.perf_counter()
        if timed:
            phase["compute"] += t1 - t0
            phase["exchange"] += t2 - t1
        return final

    def fence():
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        full = step(True)
    fence()
    dt = time.perf_counter() - t0
    phases = [phase["compute"] / max(args.steps, 1) * 1e3, phase["exchange"] / max(args.steps, 1) * 1e3]
    if multi:
        t = torch.tensor([dt] + phases, dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t[0].item())
        phases = [float(t[1].item()), float(t[2].item())]
    ms_per_step = dt / args.steps * 1e3
    value = total_pairs * args.steps / dt

    # ---- kernel phases: HIP events on the library stream, outside the timed region
    reps = 3
    km = measure_kernels(ctx, regs_d, n, p, [(local.data_ptr(), bounds[rank], bounds[rank + 1])], reps)
    pair_ms, fin_ms, prep_ms, launches = km["pair_ms"], km["finalize_ms"], km["prepare_ms"], km["pair_launches"]
    kphase = [pair_ms / reps, fin_ms / reps, prep_ms / reps]
    if multi:
        t = torch.tensor(kphase, dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        kphase = [float(x) for x in t.tolist()]

    src = source_hash()
    traffic, traffic_note = None, "no PMC file for this workload (tools/pmc_collect.py writes profiles/pmc_pair_kernel.json)"
    try:
        pm = json.load(open(os.path.join(ROOT, "profiles", "pmc_pair_kernel.json")))
        if pm.get("source_sha256") != src:
            traffic_note = "profiles/pmc_pair_kernel.json was measured on other kernel sources (sha256 differs): refused as stale"
        elif pm["workload"]["n_sketches"] == n and pm["workload"]["p"] == p and world == 1:
            traffic = pm["hbm_bytes_per_launch"]
            traffic_note = "bytes/launch, rocprofv3 PMC (2*FETCH_SIZE + WRITE_SIZE) collected by tools/pmc_collect.py in separate --pmc passes on these kernel sources (sha256 checked); not measured in this run"
    except (OSError, KeyError, ValueError):
        pass
    if rank == 0 and not multi and not args.no_pmc and not os.environ.get("DSH_BENCH_NO_PMC"):
        live, live_note = live_pmc_traffic(n, p)
        if live is not None:
            if traffic is not None:
                live_note += "; the committed profiles/pmc_pair_kernel.json (same sources, sha256 checked) says %.0f" % traffic
            traffic, traffic_note = live, live_note
        else:
            traffic_note += " [live PMC passes unavailable: %s]" % live_note
    b_pair = 2 * m + 4                                   # algorithmic bytes per pair
    achieved = my_pairs * reps * b_pair / (pair_ms * 1e-3) / 1e9 if pair_ms > 0 else 0.0
    avg_launch_ms = pair_ms / max(launches, 1)
    roofline = {
        "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_note": traffic_note,
        "kernel": "k_pair_counts", "launches_per_step": launches // reps,
        "avg_launch_ms": round(avg_launch_ms, 4),
        "bytes_per_pair": b_pair, "pairs_per_launch_avg": my_pairs * reps // max(launches, 1),
        "dense_planes": ctx.info("planes"), "reg_value_range": [ctx.info("vlo"), ctx.info("vhi")],
        "listed_tail_caps": {"upper": ctx.info("emax"), "lower": ctx.info("elow")}, "avg_planes_per_tile": ctx.info("avg_tile_planes_x100") / 100.0,
        "key_ordered_columns": bool(ctx.info("sorted")),
        "note": "SURVEY 8d streaming-model bytes (2*2^p+4 per pair, the reference's own traffic): frac > 1 only says the LDS-tiled kernel is not HBM-bound (each staged sketch is reused 128x); the binding resource is integer VALU issue, see valu_int; physical HBM is physical_hbm_gbs",
        "physical_hbm_gbs": round(traffic / (avg_launch_ms * 1e-3) / 1e9, 1) if traffic and pair_ms > 0 else None,
        "physical_hbm_frac_of_peak": round(traffic / (avg_launch_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if traffic and pair_ms > 0 else None,
        "compulsory_bytes_per_step": n * m + 4 * total_pairs,
    }
    # What actually bounds the tile kernel: integer VALU issue of one v_and_b32 + one v_bcnt_u32_b32 per 32 pair-bits.
    # Cycles are wall time x 2.4 GHz on 1024 SIMDs, the same convention as profiles/ubench/pair_sched.txt:
    #   nominal issue model  2 (v_and, SIMD-32 rate) + 4 (v_bcnt)                                  = 6.0
    #   each op on its own   2.06 + 4.07..4.42  (and_only / bcnt_only rows of pair_sched.txt)       = 6.13 (..6.49)
    #   the mix with the waves of a SIMD phase-locked (all ANDs, barrier, all BCNTs, barrier),
    #     registers only 6.4-6.8, with the LDS operand reads 6.9                                     = 6.9
    #   the mix free-running (waves of a SIMD in different instruction classes), any order/banks    = 8.18
    slots = pair_slots(ctx)
    cyc = pair_ms / reps * 1e-3 * CLOCK_HZ * N_SIMD / slots if pair_ms > 0 and slots > 0 else 0.0
    lockstep = bool(ctx.info("lockstep"))
    roofline["kernel"] = "k_pair_counts_ls" if lockstep else "k_pair_counts"
    roofline["valu_int"] = {
        "cycles_per_and_bcnt_pair": round(cyc, 3),
        "frac_of_free_running_mix_ceiling": round(8.18 / cyc, 4) if cyc else 0.0,
        "frac_of_phase_locked_mix_ceiling": round(6.9 / cyc, 4) if cyc else 0.0,
        "frac_of_isolated_rates": round(6.13 / cyc, 4) if cyc else 0.0,
        "frac_of_nominal_issue_model": round(6.0 / cyc, 4) if cyc else 0.0,
        "ceilings_cycles": {"free_running_mix": 8.18, "phase_locked_mix_with_lds_reads": 6.9, "isolated_sum": 6.13, "nominal_2_plus_4": 6.0},
        "phase_locked_kernel": lockstep,
        "note": "wave64 (AND,BCNT) slots = tiles x planes x words x 16384 / 64; cycles = wall x 2.4 GHz x 1024 SIMDs / slots. A SIMD issues ANDs from two waves at one per 2.06 cycles and BCNTs at one per 4.07-4.42 (run to run), but an AND stream next to a BCNT stream costs 8.18 per pair in any order or VGPR-bank placement (profiles/ubench/pair_sched.txt); k_pair_counts_ls keeps the 8 waves of a CU in one instruction class with ONE s_barrier per k-row (after the BCNT batch; round 2 had two: profiles/r3f), the k loop fully unrolled: a k-row is 64 v_and_b32 + 64 v_bcnt_u32_b32 + 4 ds_read_b128 + 1 s_barrier. The chip holds 2.30-2.39 GHz under this mix, so 6.9 shader cycles of the micro-benchmark twin are ~7.1 of the wall cycles quoted here",
    }
    roofline["finalize"] = {
        "kernel": "k_finalize", "ms_per_step": round(kphase[1], 4), "bound": "fp64 VALU issue",
        "pairs_per_s": round(my_pairs / (kphase[1] * 1e-3), 1) if kphase[1] > 0 else 0.0,
        "cycles_per_wave64_of_pairs": round(kphase[1] * 1e-3 * CLOCK_HZ * N_SIMD / (my_pairs / 64.0), 1) if kphase[1] > 0 and my_pairs else 0.0,
        "note": "VALU-issue bound; most of its VALU instructions are the Ertl-MLE estimator (~3 secant iterations x ~17 bins = ~51 steps x 19 instructions, 15 of them dependent fp64 operations, plus fp64 divisions per iteration) that must be reproduced bit for bit; the sparse tails of the histogram come from a position-index join (k_build_colindex) instead of round 2's per-pair list walk -- profiles/r3a (before), r3b, r3f, DESIGN.md 3.5",
    }
    roofline["step"] = {
        "ms": {"prepare": round(kphase[2], 4), "pair_counts": round(kphase[0], 4), "finalize": round(kphase[1], 4)},
        "algorithmic_gbs_whole_step": round(total_pairs * b_pair / (ms_per_step * 1e-3) / 1e9, 1),
    }

    cpu = None
    parity = None
    if rank == 0 and multi:
        # assembled multi-rank matrix vs one single-GPU call on rank 0 (outside the timed region)
        ref = torch.empty(total_pairs, dtype=torch.float32, device=dev)
        ctx.attach_device(regs_d.data_ptr(), n, p)
        ctx.dist_rows_device(ref.data_ptr(), 0, n, dashing_amd.ESTIM_ERTL_MLE, dashing_amd.JI, K)
        ctx.synchronize()
        parity = {"assembled_equals_single_gpu": bool(torch.equal(ref, full[:total_pairs])), "pairs_checked": total_pairs}
        del ref
    if rank == 0 and not multi and not args.no_cpu_baseline:
        cpu, parity = cpu_baseline(regs_h, full, n, p, args.cpu_seconds)

    what_if = None
    if rank == 0 and not multi and not args.no_secondary:
        # WHAT-IF, never the product path (the north star excludes the matrix cores; `value` above is the integer-VALU
        # kernel): the same tile kernel with the AND+popcount done as a 0/1 i8 MFMA (option pair_mfma), same inputs
        ref = full[:total_pairs].clone()
        ctx.set_option("pair_mfma", 1)
        ctx.set_option("kc", 16)  # (the what-if kernel was tuned at 16 rows per stage: two workgroups per CU)
        ts = []
        for _ in range(3):
            ctx.attach_device(regs_d.data_ptr(), n, p)
            t0 = time.perf_counter()
            ctx.dist_rows_device(local.data_ptr(), 0, n, dashing_amd.ESTIM_ERTL_MLE, dashing_amd.JI, K)
            ctx.synchronize()
            ts.append(time.perf_counter() - t0)
        same = bool(torch.equal(ref, local[:total_pairs]))
        kw = measure_kernels(ctx, regs_d, n, p, [(local.data_ptr(), 0, n)], 1)
        ctx.set_option("pair_mfma", 0)
        ctx.set_option("kc", 0)
        what_if = {"label": "WHAT-IF ONLY, not the shipped path and not `value`: v_mfma_i32_32x32x32_i8 on 0/1 bytes expanded from the bit-planes in registers (option pair_mfma=1, default 0)",
                   "ms_per_step": round(min(ts) * 1e3, 3), "pairs_per_s": total_pairs / min(ts),
                   "k_pair_counts_mfma_ms": round(kw["pair_ms"], 3), "output_identical_to_valu_path": same}
        del ref
    secondary = None
    if rank == 0 and not multi and not args.no_secondary and (n, p) == (10000, 14):
        secondary = secondary_p10(ctx, torch, dev, synth, dashing_amd)

    dependence = None
    if rank == 0 and not multi and not args.no_secondary and (n, p) == (10000, 14):
        dependence = data_dependence(ctx, torch, dev, dashing_amd, n, p)

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
                       "sharding": "tile-count-balanced row ranges of the final triangle, one per rank (plane matrix laid out per range); point-to-point send of each span into place on rank 0, no un-permute" if multi else "single GPU"},
            "roofline": roofline,
            "cpu_baseline": cpu,
            "parity_vs_cpu": parity,
            "kernel_source_sha256": src,
        }
        if multi:
            line["multi_gpu"] = {
                "ranks": world, "rccl_ranks": dist.get_world_size() if backend == "nccl" else 0, "backend": backend, "exchange": exchange,
                "row_bounds": bounds, "pairs_per_rank": sizes,
                "phase_ms_max_over_ranks": {"compute_incl_prepare": round(phases[0], 4), "exchange" if not use_cabi else "exchange_exposed_after_last_kernel": round(phases[1], 4),
                                            "k_pair_counts": round(kphase[0], 4), "k_finalize": round(kphase[1], 4), "prepare": round(kphase[2], 4)},
                "exchange_bytes_into_rank0": 4 * (total_pairs - sizes[0]),
            }
        if secondary:
            line["secondary"] = secondary
        if dependence:
            line["data_dependence"] = dependence
        if what_if:
            line["what_if_mfma"] = what_if
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


def data_dependence(ctx, torch, dev, dashing_amd, n, p):
    """The headline workload is friendly (cardinalities within a factor of 4: few planes per tile).  The same pass on
    two collections it is NOT -- time is proportional to planes per tile -- so the line carries the spread:
    (a) cardinalities log-uniform over 1e4 .. 1e8 (a RefSeq-like spread), registers from the register law;
    (b) registers uniform over [0, q+1]: not the sketch of anything, the adversary of thermometer planes."""
    m, q1 = 1 << p, 64 - p + 1
    g = torch.Generator(device=dev)
    g.manual_seed(77)
    total = n * (n - 1) // 2
    out = torch.empty(total, dtype=torch.float32, device=dev)
    res = {}
    cards = torch.exp(torch.empty(n, 1, device=dev, dtype=torch.float64).uniform_(float(np.log(1e4)), float(np.log(1e8)), generator=g))
    u = torch.rand((n, m), generator=g, device=dev, dtype=torch.float64).clamp_(1e-300, 1.0 - 1e-16)
    regs_a = torch.ceil(torch.log2((cards / m) / -torch.log(u))).clamp_(0, q1).to(torch.uint8)
    del u
    regs_b = torch.randint(0, q1 + 1, (n, m), generator=g, device=dev, dtype=torch.uint8)
    for name, regs in (("log_uniform_cardinalities_1e4_1e8", regs_a), ("uniform_registers_adversary", regs_b)):
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            ctx.attach_device(regs.data_ptr(), n, p)
            t0 = time.perf_counter()
            ctx.dist_rows_device(out.data_ptr(), 0, n, dashing_amd.ESTIM_ERTL_MLE, dashing_amd.JI, K)
            ctx.synchronize()
            best = min(best, time.perf_counter() - t0)
        res[name] = {"ms_per_step": round(best * 1e3, 3), "pairs_per_s": total / best,
                     "avg_planes_per_tile": ctx.info("avg_tile_planes_x100") / 100.0, "dense_planes_global": ctx.info("planes"),
                     "all_finite": bool(torch.isfinite(out[: 1 << 24]).all())}
    res["note"] = "same N, p, estimator as the headline; correctness on such inputs is covered by tests/test_gpu_compare.py (adversarial registers, heterogeneous collections) and tests/test_gpu_fuzz.py"
    del out, regs_a, regs_b
    torch.cuda.empty_cache()
    return res


def secondary_p10(ctx, torch, dev, synth, dashing_amd, n=100_000, p=10):
    """configs[3]-shaped matrix on ONE GPU (the 8-GPU run is the driver's): at 1 KiB per sketch the popcounts are
    cheap and the per-pair estimator (k_finalize) is the hot kernel, which the p=14 headline hides."""
    regs = synth.survey_sketches(n, p, seed=0x5EED0000)[0]
    regs_d = torch.from_numpy(regs).to(dev)
    total = n * (n - 1) // 2
    out = torch.empty(total, dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    times = []
    for _ in range(3):
        ctx.attach_device(regs_d.data_ptr(), n, p)
        t0 = time.perf_counter()
        ctx.dist_rows_device(out.data_ptr(), 0, n, dashing_amd.ESTIM_ERTL_MLE, dashing_amd.JI, K)
        ctx.synchronize()
        times.append(time.perf_counter() - t0)
    km = measure_kernels(ctx, regs_d, n, p, [(out.data_ptr(), 0, n)], 1)
    t = min(times[1:])
    res = {"workload": "BASELINE configs[3] shape on one GPU: %d synthetic sketches, p=%d, full triangle (%.1f GB of float32 left in HBM)" % (n, p, total * 4 / 1e9),
           "value": total / t, "unit": "pairs/s", "ms_per_step": t * 1e3, "steps": 2,
           "kernel_ms": {"k_pair_counts": round(km["pair_ms"], 3), "k_finalize": round(km["finalize_ms"], 3), "prepare": round(km["prepare_ms"], 3),
                         "pair_launches": km["pair_launches"]},
           "avg_planes_per_tile": ctx.info("avg_tile_planes_x100") / 100.0,
           "finalize_cycles_per_wave64_of_pairs": round(km["finalize_ms"] * 1e-3 * CLOCK_HZ * N_SIMD / (total / 64.0), 1)}
    del out, regs_d
    torch.cuda.empty_cache()
    return res


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
        build = "-march=native"
    except Exception:
        kind_lib = oracle_c.load(threads=cores)
        build = "-march=x86-64-v3 (prebuilt)"
    level = oracle_c.simd_level(kind_lib)  # the histogram-of-max variant is chosen by cpuid, not by the build flags
    isa = {0: "scalar", 1: "avx2", 2: "avx512bw"}[level]

    def timed_rows(rows, simd):
        oracle_c.set_simd(simd, kind_lib)
        t0 = time.perf_counter()
        r = oracle_c.dist_rows(regs_h, 0, rows, lib=kind_lib)
        return r, time.perf_counter() - t0

    # calibrate on a few rows, then size the sample for ~`seconds`
    r0, t_cal = timed_rows(8, level)
    rate = r0.size / max(t_cal, 1e-6)
    rows = int(min(n - 1, max(16, seconds * rate / n)))
    ref, t = timed_rows(rows, level)
    # the scalar histogram on a quarter of the sample, for the record (and as a cross-check of the SIMD one)
    rows_s = max(8, rows // 4)
    ref_s, t_s = timed_rows(rows_s, 0)
    oracle_c.set_simd(0, kind_lib)
    assert (ref[: ref_s.size] == ref_s).all(), "SIMD and scalar CPU histograms disagree"
    got = gpu_full[: ref.size].cpu().numpy()
    rel = np.abs(got.astype(np.float64) - ref) / np.maximum(np.abs(ref), 1e-9)
    cpu = {"value": ref.size / t, "unit": "pairs/s", "cores": cores, "kind": "port", "isa": isa, "cpu_model": cpu_model(),
           "sample": "rows [0,%d) of the same %d-sketch p=%d matrix = %d pairs in %.1f s; oracle/dsh_oracle.c (%s; %s histogram-of-max: 64-byte max_epu8 + per-value compare-and-count; OpenMP dynamic over j per row as src/sketch_and_cmp.h:699-710)" % (rows, n, p, ref.size, t, build, isa),
           "scalar_histogram_value": ref_s.size / t_s,
           "note": "CPU restatement of the reference's algorithm and schedule in its SIMD form (Makefile:159-190); the reference itself is not buildable (bonsai/sketch submodules absent)"}
    parity = {"pairs_checked": int(ref.size), "max_rel_diff": float(rel.max()), "tolerance": 1e-6,
              "exact_float32_matches": int((got == ref).sum())}
    try:
        os.unlink(native)
    except OSError:
        pass
    return cpu, parity


if __name__ == "__main__":
    main()

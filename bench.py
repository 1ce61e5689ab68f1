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
dsh_exchange_rows_device_async / dsh_exchange_collect_async: part q of a rank's rows travels on the copy stream
while the later parts are still being finalized; short ranges keep their rows key-ordered as one run and rank 0 puts
the rows of a received part into place); `python bench.py --gpus N` launches its N ranks itself.
Outputs beyond 2 GB (configs[3]-sized: DSH_BENCH_N=100000 DSH_BENCH_P=10) are NOT gathered by default: every rank keeps
its span (what `dashing-amd dist --ngpus -b` writes per device); the gathered variant is reported beside it.

Prints ONE JSON line on rank 0 -- also when the run fails (then with "error").  `roofline` describes the dominant kernel
(k_pair_counts_ls), timed with HIP events on the library's own stream; `roofline.binding` names the resource that
actually bounds it (integer VALU issue -- the streaming-model HBM figure above 1 only says the kernel is not HBM-bound).
`cpu_baseline` is the CPU oracle (a restatement of the reference algorithm and row schedule with an AVX-512BW / AVX2
histogram-of-max -- the reference itself is not buildable: its bonsai/sketch submodules are absent) timed on this host on
a bounded sample of the same rows.  `configs` carries, per BASELINE config measurable on one GPU, throughput, the binding
resource, physical HBM, the CPU sample and parity: configs[1] (k_sketch on 200 x 5 Mbp resident in HBM), configs[2] (the
headline), configs[3] shape (100 000 x p=10, where k_finalize is the hot kernel) and a configs[4]-shaped band (one row
range of 300 000 x p=14, extrapolated and labelled so).
"""
import argparse
import hashlib
import json
import os
import sys
import time
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_SKETCH = int(os.environ.get("DSH_BENCH_N", "10000"))
P = int(os.environ.get("DSH_BENCH_P", "14"))
K = 31
HBM_PEAK_GBS = 8000.0  # MI355X spec, /opt/skills/guides/MI355X_MICROARCH.md
CLOCK_HZ = 2.4e9       # max shader clock (same guide); cycle figures below are "wall time x 2.4 GHz"
N_SIMD = 256 * 4
GATHER_LIMIT_BYTES = 2 << 30  # larger outputs stay on the ranks that computed them (default; --exchange overrides)
PARITY_NOTE = ("vs the CPU restatement (oracle/dsh_oracle.c); upstream arithmetic unpinned (bonsai / sketch submodules absent from the reference tree). "
               "The contract is 1e-6 relative; exact_float32_matches is OBSERVED, not guaranteed (the recurrence's division is one Newton step, "
               "correctly rounded unless the quotient lies within 2^-97 of a rounding midpoint: DESIGN.md 3.4)")

DEVICE_SOURCES = ("dashing_amd/csrc/kernels_compare.hip", "dashing_amd/csrc/kernels_sketch.hip",
                  "dashing_amd/csrc/estimators.h", "dashing_amd/csrc/kernels.h", "dashing_amd/csrc/consts.h", "dashing_amd/csrc/ctx.h",
                  "dashing_amd/csrc/plan.h", "dashing_amd/csrc/plan.cpp", "dashing_amd/csrc/engine.hip", "dashing_amd/csrc/abi.hip",
                  "dashing_amd/csrc/knn.hip", "dashing_amd/csrc/exchange.hip", "dashing_amd/csrc/kernels_fastx.hip")


def source_hash():
    """sha256 over the device sources (same function as tools/pmc_collect.py): PMC files measured on other
    sources are refused."""
    h = hashlib.sha256()
    for rel in DEVICE_SOURCES:
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


def cached_sketches(synth, n, p, tag):
    """the SURVEY 8d register arrays; the parent keeps them as .npy under /tmp for the PMC child passes of the same run
    (generation takes seconds, the children only need the bytes)"""
    path = os.environ.get("DSH_BENCH_CACHE_" + tag)
    if path and os.path.exists(path):
        return np.load(path)
    return synth.survey_sketches(n, p, seed=0x5EED0000)[0]


def measure_kernels(ctx, regs_d, n, p, calls, reps=3):
    """HIP-event times of the three phases on the library's own stream, summed over `reps` full passes"""
    import dashing_amd

    ctx.set_profiling(True)
    acc = {"pair_ms": 0.0, "finalize_ms": 0.0, "prepare_ms": 0.0, "pair_launches": 0}
    for _ in range(reps):
        ctx.attach_device(regs_d.data_ptr(), n, p)
        for call in calls:
            if callable(call):  # (the exchange-aware call of an N-rank run: a row set, not a range)
                call()
            else:
                ptr, rb, re = call
                ctx.dist_rows_device(ptr, rb, re, dashing_amd.ESTIM_ERTL_MLE, dashing_amd.JI, K)
            ctx.synchronize()
            k = ctx.last_kernel_ms()
            for key in acc:
                acc[key] += k[key]
    ctx.set_profiling(False)
    return acc


# ---- synthetic genomes in HBM (configs[1] shape) -----------------------------------------------------------------------------
def device_genomes(torch, dev, G, L, seed=0xDA5410):
    """G related genomes of L bases as ASCII in ONE device buffer (genome g at [g*L, (g+1)*L), L a multiple of 32): a random
    root, clusters of 10 at 5 % from it, members at 0.1 % .. 5 % from their cluster ancestor, every 10th
    genome with a run of 50 N and a lowercase kilobase.  Generated on the GPU: the bench times the kernel, not numpy."""
    assert L % 32 == 0
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)

    def mutate(codes, rate):
        hit = torch.rand(codes.shape, generator=g, device=dev) < rate
        shift = torch.randint(1, 4, codes.shape, generator=g, device=dev, dtype=torch.uint8)
        return torch.where(hit, (codes + shift) & 3, codes)

    root = torch.randint(0, 4, (L,), generator=g, device=dev, dtype=torch.uint8)
    seq = torch.empty(G * L + 256, dtype=torch.uint8, device=dev)
    seq[G * L:] = ord("N")
    rates = (0.001, 0.005, 0.01, 0.02, 0.05)
    anc = None
    for i in range(G):
        if i % 10 == 0:
            anc = mutate(root, 0.05)
        seq[i * L:(i + 1) * L] = lut[mutate(anc, rates[i % 5]).long()]
        if i % 10 == 0:
            seq[i * L + L // 3: i * L + L // 3 + 50] = ord("N")
            seq[i * L + L // 2: i * L + L // 2 + 1000] |= 0x20
    torch.cuda.synchronize(dev)  # torch's stream wrote the bases; the library reads them on its own stream
    return seq


def sketch_workload(ctx, torch, dev, G, L, p, steps):
    """k_sketch over G x L bases resident in HBM; returns (seq, seconds per call wall, kernel ms by HIP events, registers)"""
    seq = device_genomes(torch, dev, G, L)
    offs = np.arange(G + 1, dtype=np.uint64) * np.uint64(L)
    ctx.alloc(G, p)
    ctx.clear()
    ctx.sketch_batch_device(seq.data_ptr(), offs, 0, K, True)  # warm-up (and the registers for the parity check)
    regs = ctx.download(0, G)
    ctx.set_profiling(True)
    kms, ts = [], []
    for _ in range(steps):
        ctx.clear()
        ctx.synchronize()
        t0 = time.perf_counter()
        ctx.sketch_batch_device(seq.data_ptr(), offs, 0, K, True)  # returns when the kernel is done
        ts.append(time.perf_counter() - t0)
        kms.append(ctx.info("sketch_kernel_us") / 1e3)
    ctx.set_profiling(False)
    return seq, min(ts), sum(kms) / len(kms), regs


# ---- PMC child passes ------------------------------------------------------------------------------------------------------------
PMC_PASSES = (("FETCH_SIZE",), ("WRITE_SIZE",), ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_WAVES"))
PMC_SKETCH_GENOMES = 40


def pmc_child():
    """(internal, run under rocprofv3 --pmc by live_pmc): the three workloads in short form, nothing printed"""
    import torch

    import dashing_amd
    from dashing_amd import synth

    dev = torch.device("cuda", 0)
    ctx = dashing_amd.Context(0)
    for tag, n, p in (("C3", N_SKETCH, P), ("C4", 100_000, 10)):
        if tag == "C4" and not os.environ.get("DSH_BENCH_CACHE_C4"):
            continue
        regs_d = torch.from_numpy(cached_sketches(synth, n, p, tag)).to(dev)
        out = torch.empty(n * (n - 1) // 2, dtype=torch.float32, device=dev)
        for _ in range(3 if tag == "C3" else 2):  # the LAST pass of each workload is the one read
            ctx.attach_device(regs_d.data_ptr(), n, p)
            ctx.dist_rows_device(out.data_ptr(), 0, n, dashing_amd.ESTIM_ERTL_MLE, dashing_amd.JI, K)
            ctx.synchronize()
        del regs_d, out
        torch.cuda.empty_cache()
    if os.environ.get("DSH_BENCH_PMC_BAND"):  # one band of the configs[4]-shaped collection (a compare pass of its own)
        b = C5_BAND
        regs5 = c5_sketches(torch, dev, synth, b["n"], b["p"], b["nbase"])
        out = torch.empty(dashing_amd.tri_span(b["n"], 0, b["rows"]), dtype=torch.float32, device=dev)
        ctx.attach_device(regs5.data_ptr(), b["n"], b["p"])
        ctx.dist_rows_device(out.data_ptr(), 0, b["rows"], dashing_amd.ESTIM_ERTL_MLE, dashing_amd.JI, K)
        ctx.synchronize()
        del regs5, out
        torch.cuda.empty_cache()
    sketch_workload(ctx, torch, dev, PMC_SKETCH_GENOMES, 5_000_000, 10, 1)
    ctx.close()


def live_pmc(n, p, cache_env):
    """PMC counters measured by THIS run: child processes of bench.py under `rocprofv3 --pmc` -- separate passes per counter
    group, never combined with a trace, as MI355X_MICROARCH.md prescribes -- each running the headline workload, the
    configs[3]-shaped matrix and the sketch kernel in short form.  Returns ({workload_kernel: {counter: sum over the
    dispatches of the LAST pass of that workload, launches_<counter>: their number}}, note).  Keys: C3_pair, C3_finalize,
    C4_pair, C4_finalize (k_pair_counts* / k_finalize) and sketch (the last k_sketch dispatch)."""
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
    res = {}
    work = tempfile.mkdtemp(prefix="dsh_pmc_", dir="/tmp")
    try:
        for counters in PMC_PASSES:
            d = os.path.join(work, counters[0])
            os.makedirs(d)
            argv = ["rocprofv3", "--pmc"] + list(counters) + ["-d", d, "-o", "p", "--output-format", "csv", "--", sys.executable,
                                                            os.path.abspath(__file__), "--pmc-child"]
            env = dict(os.environ, TMPDIR="/tmp", DSH_BENCH_N=str(n), DSH_BENCH_P=str(p), **cache_env)
            for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
                env.pop(k, None)
            try:
                r = subprocess.run(argv, env=env, cwd="/tmp", stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=300)
            except (OSError, subprocess.TimeoutExpired) as e:
                return None, "rocprofv3 --pmc %s pass did not finish (%s)" % (counters[0], type(e).__name__)
            rows = []
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                rows += list(csv.DictReader(open(f)))
            if r.returncode != 0 or not rows:
                return None, "rocprofv3 --pmc %s pass: rc %d, %d rows" % (counters[0], r.returncode, len(rows))
            rows.sort(key=lambda x: int(x.get("Dispatch_Id", 0) or 0))
            # the child runs C3 (3 passes), then C4 (2 passes), then the sketch kernel; every compare pass starts with one
            # k_selfhist_card dispatch: split the dispatch stream there, keep the last pass of each compare workload
            passes, cur = [], None
            for row in rows:
                if "k_selfhist_card" in row["Kernel_Name"] and row["Counter_Name"] == counters[0]:
                    cur = []
                    passes.append(cur)
                if cur is not None:
                    cur.append(row)
            last = {}
            if len(passes) >= 3:
                last["C3"] = passes[2]
            if len(passes) >= 5:
                last["C4"] = passes[4]
            if len(passes) >= 6 and cache_env.get("DSH_BENCH_PMC_BAND"):
                last["C5"] = passes[5]
            for wl, prow in last.items():
                for row in prow:
                    kn = row["Kernel_Name"]
                    fam = "pair" if ("k_pair_counts" in kn and "mfma" not in kn) else "finalize" if "k_finalize" in kn else None
                    if fam:
                        d_ = res.setdefault("%s_%s" % (wl, fam), {})
                        c_ = row["Counter_Name"]
                        d_[c_] = d_.get(c_, 0.0) + float(row["Counter_Value"])
                        d_["launches_" + c_] = d_.get("launches_" + c_, 0) + 1
            for c_ in counters:
                v = [float(row["Counter_Value"]) for row in rows if "k_sketch" in row["Kernel_Name"] and row["Counter_Name"] == c_]
                if v:
                    res.setdefault("sketch", {})[c_] = v[-1]
    finally:
        shutil.rmtree(work, ignore_errors=True)
    return res, ("measured in this run: three child runs of this script under rocprofv3 --pmc (FETCH_SIZE | WRITE_SIZE | SQ_INSTS_VALU, "
                 "SQ_INSTS_SALU, SQ_WAVES; separate passes, no trace); HBM bytes = (2*FETCH_SIZE + WRITE_SIZE) KiB as MI355X_MICROARCH.md prescribes for gfx950")


def hbm_bytes(pm):
    if not pm or "FETCH_SIZE" not in pm or "WRITE_SIZE" not in pm:
        return None
    return (2.0 * pm["FETCH_SIZE"] + pm["WRITE_SIZE"]) * 1024.0


def pair_slots(ctx):
    """wave-level (AND, BCNT) slots of the last pair-kernel schedule: per tile, planes x words x 128x128 pairs / 64 lanes"""
    return ctx.info("avg_tile_planes_x100") / 100.0 * ctx.info("words_per_plane") * ctx.info("tiles") * 128 * 128 / 64.0


def pair_binding(ctx, pair_ms_per_pass):
    """What bounds the tile kernel: integer VALU issue of one v_and_b32 + one v_bcnt_u32_b32 per 32 pair-bits.
    Cycles are wall time x 2.4 GHz on 1024 SIMDs, the same convention as profiles/ubench/pair_sched.txt:
      nominal issue model  2 (v_and, SIMD-32 rate) + 4 (v_bcnt)                                  = 6.0
      each op on its own   2.06 + 4.07..4.42  (and_only / bcnt_only rows of pair_sched.txt)       = 6.13 (..6.49)
      the mix with the waves of a SIMD phase-locked, with the LDS operand reads                   = 6.9
      the mix free-running (waves of a SIMD in different instruction classes), any order/banks    = 8.18"""
    slots = pair_slots(ctx)
    cyc = pair_ms_per_pass * 1e-3 * CLOCK_HZ * N_SIMD / slots if pair_ms_per_pass > 0 and slots > 0 else 0.0
    return cyc, {"resource": "int VALU issue (v_and_b32 + v_bcnt_u32_b32)", "frac": round(6.0 / cyc, 4) if cyc else None,
                 "cycles_per_pair": round(cyc, 3), "ceiling": 6.0,
                 "note": "cycles per wave64 (AND,BCNT) pair = wall x 2.4 GHz x 1024 SIMDs / (tiles x planes x words x 16384 / 64); ceiling = 2 + 4 issue cycles"}


def finalize_binding(fin_ms, pairs, pm):
    """k_finalize is VALU-issue bound too, most of it fp64 (4 issue cycles per wave64 instruction at the DP rate, 2 for the
    integer ones): with SQ_INSTS_VALU from the PMC pass the fraction is (instructions x 4) / SIMD-cycles spent -- an upper
    bound of the true issue utilisation by the share of 2-cycle integer instructions in the mix."""
    cyc_wave = fin_ms * 1e-3 * CLOCK_HZ * N_SIMD / (pairs / 64.0) if fin_ms > 0 and pairs else 0.0
    b = {"resource": "VALU issue, mostly fp64 (the bit-exact Ertl-MLE recurrence)", "cycles_per_wave64_of_pairs": round(cyc_wave, 1), "ceiling_cycles_per_valu_inst": 4.0,
         "frac": None, "valu_insts_per_wave": None, "cycles_per_valu_inst": None}
    if pm and pm.get("SQ_INSTS_VALU") and pm.get("SQ_WAVES"):
        ipw = pm["SQ_INSTS_VALU"] / pm["SQ_WAVES"]
        cpi = fin_ms * 1e-3 * CLOCK_HZ * N_SIMD / pm["SQ_INSTS_VALU"]
        b.update({"valu_insts_per_wave": round(ipw, 1), "cycles_per_valu_inst": round(cpi, 3), "frac": round(4.0 / cpi, 4) if cpi else None,
                  "salu_insts_per_wave": round(pm.get("SQ_INSTS_SALU", 0.0) / pm["SQ_WAVES"], 1)})
    return b


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


def flush_c_stdio():
    import ctypes

    try:
        ctypes.CDLL(None).fflush(None)
    except OSError:
        pass


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-secondary", action="store_true", help="skip the per-config entries (configs[1], [3], [4]) and the data-dependence line")
    ap.add_argument("--no-pmc", action="store_true", help="do not run the rocprofv3 --pmc child passes (roofline.traffic, VALU counts)")
    ap.add_argument("--what-if", action="store_true", help="also time the matrix-core what-if of the tile kernel (needs a library built with make WHATIF=1)")
    ap.add_argument("--exchange", choices=("auto", "gather", "keep"), default="auto",
                    help="N>1: gather the spans on rank 0 (the BASELINE formulation) or keep every span on its rank; auto = gather up to 2 GB of output")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.pmc_child:
        pmc_child()
        return
    if args.gpus < 1:
        sys.exit("bench.py: --gpus must be >= 1")
    backend = os.environ.get("DSH_BENCH_BACKEND", "nccl")
    if "WORLD_SIZE" not in os.environ:
        if args.gpus > 1:
            sys.exit(self_launch(args.gpus, backend))  # one rank per GPU under torch.distributed.run; never a silent 1-GPU run
    elif int(os.environ["WORLD_SIZE"]) != args.gpus:
        sys.exit("bench.py: launched with WORLD_SIZE=%s but --gpus %d: refusing to report a mislabelled run" % (os.environ["WORLD_SIZE"], args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    line = {"metric": "genome-pairs/sec, all-pairs HLL Jaccard (Ertl-MLE), N=%d p=%d" % (N_SKETCH, P), "value": None, "unit": "pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": None, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "u8/u32 popcount + f64 estimator", "data": "synthetic"}
    dist_on = [False]
    try:
        run(args, backend, world, rank, line, dist_on)
    except BaseException as e:  # noqa: BLE001 -- the contract is ONE JSON line, whatever happened
        if isinstance(e, SystemExit) and e.code in (0, None):
            raise
        line["error"] = "%s: %s" % (type(e).__name__, e)
        line["traceback_tail"] = traceback.format_exc().strip().splitlines()[-6:]
        sys.stderr.write(traceback.format_exc())
    finally:
        if dist_on[0]:
            import torch.distributed as dist

            flush_c_stdio()
            try:
                if "error" not in line:
                    dist.barrier()
                dist.destroy_process_group()
            except Exception:  # noqa: BLE001
                pass
        # The JSON line must be the LAST thing on stdout: RCCL (NCCL_DEBUG=VERSION is exported on the GPU boxes) prints its
        # banner through C stdio, which is block-buffered when stdout is a pipe: flush it before the line goes out.
        flush_c_stdio()
        if rank == 0:
            sys.stdout.flush()
            print(json.dumps(line), flush=True)
    if "error" in line:
        sys.exit(1)


def run(args, backend, world, rank, line, dist_on):
    import torch
    import torch.distributed as dist

    import dashing_amd
    from dashing_amd import multigpu, synth

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
                raise RuntimeError("%d ranks but only %d visible GPUs" % (world, torch.cuda.device_count()))
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local_rank))
        dist_on[0] = True
    else:
        torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    cpu_or_dev = dev if backend == "nccl" else "cpu"

    n, p, m = N_SKETCH, P, 1 << P
    regs_h = cached_sketches(synth, n, p, "C3")  # SURVEY 8d; identical bytes on every rank
    regs_d = torch.from_numpy(regs_h).to(dev)    # resident in HBM before timing
    total_pairs = n * (n - 1) // 2
    ctx = dashing_amd.Context(local_rank)
    for kv in filter(None, os.environ.get("DSH_BENCH_OPTS", "").split(",")):  # option sweeps, e.g. "kc=16,emax=32,finalize_signal=0"
        k_, v_ = kv.split("=")
        ctx.set_option(k_, int(v_))
    ctx.attach_device(regs_d.data_ptr(), n, p)

    # The exchange of the spans: through the C-ABI (dsh_comm_init / dsh_collect_parts_async: RCCL inside libdashing_hip.so, on
    # the library's streams -- what a C++ host calls) unless DSH_BENCH_EXCHANGE=torch; if the library's communicator
    # cannot be brought up ON EVERY RANK, torch.distributed's RCCL does the same point-to-point transfers (recorded in the
    # line).  Availability is agreed on BEFORE the collective dsh_comm_init: a rank that cannot load librccl must not leave
    # the others waiting inside ncclCommInitRank.
    exchange = "none"
    rccl_info = None
    if multi:
        exchange = "torch.distributed" if backend == "nccl" else "gloo (host staged)"
        path, ver = dashing_amd.comm_library()
        # every RCCL image mapped into this process (torch bundles one with the soname librccl.so.1, so the library's
        # dlopen("librccl.so.1") resolves to THAT image when torch was imported first: one copy, shared) and the one
        # whose ncclSend/ncclRecv the C-ABI exchange calls
        copies = sorted({l.split()[-1] for l in open("/proc/self/maps") if "rccl" in l.rsplit("/", 1)[-1] and l.split()[-1].startswith("/")})
        rccl_info = {"library": path, "carried_the_exchange": None, "rccl_copies_mapped": copies,
                     "torch_bundled_rccl": next((c_ for c_ in copies if "/torch/lib/" in c_), None),
                     "nccl_version_code": ver, "available_on_rank0": dashing_amd.comm_available(),
                     "NCCL_DEBUG": os.environ.get("NCCL_DEBUG"),
                     "torch_nccl_version": ".".join(str(x) for x in torch.cuda.nccl.version()) if backend == "nccl" else None}
        # (DSH_BENCH_EXCHANGE=cabi-mock with the gloo dry run: the C-ABI exchange between the ranks sharing cuda:0 over the
        # stand-in transport of tests/mock_rccl, DSH_RCCL_LIB -- the code path of a real run, never a reported number)
        want_x = os.environ.get("DSH_BENCH_EXCHANGE", "cabi")
        if (backend == "nccl" and want_x == "cabi") or (backend == "gloo" and want_x == "cabi-mock"):
            flag = torch.tensor([1 if dashing_amd.comm_available() else 0], device=cpu_or_dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            ok = int(flag.item()) == 1
            why = None if ok else "librccl not loadable by libdashing_hip.so on at least one rank (%s)" % path
            if ok:
                try:
                    multigpu.cabi_comm_init(ctx, rank, world)
                except Exception as e:  # noqa: BLE001
                    ok, why = False, "dsh_comm_init failed on rank %d: %s" % (rank, e)
                mine_ok = ok
                flag = torch.tensor([1 if ok else 0], device=cpu_or_dev)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)  # all ranks or none
                if int(flag.item()) == 0:
                    if mine_ok:
                        ctx.comm_destroy()
                    ok, why = False, why or "dsh_comm_init failed on another rank"
            if ok:
                exchange = "c-abi rccl"
            else:
                sys.stderr.write("bench.py: rank %d: C-ABI communicator not used (%s): torch.distributed exchange\n" % (rank, why))
                rccl_info["cabi_fallback_reason"] = why
    use_cabi = exchange.startswith("c-abi")
    if rccl_info is not None:
        rccl_info["carried_the_exchange"] = path if use_cabi else ("torch.distributed (%s)" % rccl_info["torch_bundled_rccl"] if backend == "nccl" else "gloo")
    NPARTS = int(os.environ.get("DSH_BENCH_PARTS", "8"))
    if use_cabi:
        exchange = "c-abi rccl, pipelined in <= %d parts per rank (dsh_exchange_rows_device_async + dsh_exchange_collect_async)" % NPARTS
    gather = not multi or args.exchange == "gather" or (args.exchange == "auto" and 4 * total_pairs <= GATHER_LIMIT_BYTES)
    bounds = dashing_amd.balance_rows(n, world) if multi else [0, n]
    sizes = multigpu.span_sizes(n, bounds)
    # the C-ABI exchange partitions the rows into ROW SETS: a range per rank plus top-up tile rows from the bottom of the
    # triangle (dsh_balance_rowsets), so that every rank computes about the same number of tiles; the torch.distributed
    # fallback keeps the contiguous ranges (its spans are received in place)
    rows_of = dashing_amd.balance_rowsets(n, world, int(os.environ.get("DSH_BENCH_PREP_PERMILLE", "-1")), 0 if gather else -1,
                                          int(os.environ.get("DSH_BENCH_DST_BONUS_PERMILLE", "-1"))) if use_cabi else None
    my_floats = None
    if rows_of is not None:
        sizes = [rows_of.pairs(r) for r in range(world)]
        my_floats = dashing_amd.exchange_mode(n, rows_of, rank, NPARTS, 0, want_floats=True)[2]
    my_pairs = sizes[rank] if multi else total_pairs
    host_stage = backend == "gloo" and multi
    buf = {"final": None, "local": None, "final_h": None}

    def alloc_buffers(with_final):
        buf["final"] = torch.empty(max(total_pairs, 1), dtype=torch.float32, device=dev) if (rank == 0 and with_final) else None
        # rank 0 computes in place (its rows are the head of the matrix); the others into a span-sized buffer
        buf["local"] = buf["final"] if buf["final"] is not None else torch.empty(max(my_floats or my_pairs, 1), dtype=torch.float32, device=dev)
        buf["final_h"] = torch.empty(max(total_pairs, 1), dtype=torch.float32) if host_stage and rank == 0 and with_final else None

    alloc_buffers(gather)
    phase = {"compute": 0.0, "exchange": 0.0}

    def step(timed=False, do_gather=True):
        local, final, final_h = buf["local"], buf["final"], buf["final_h"]
        # re-attach: invalidates cached planes/cardinalities, so every step is a full pass
        t0 = time.perf_counter()
        ctx.attach_device(regs_d.data_ptr(), n, p)
        if use_cabi and do_gather:
            # pipelined: the rank's rows in <= NPARTS parts (short ranges: row-sorted parts, placed row by row on rank 0);
            # part q travels to rank 0 (copy stream, grouped ncclSend/ncclRecv behind the part's event) while the later
            # parts are still being finalized on the ctx stream
            ctx.exchange_rows_device_async(local.data_ptr(), rows_of, rank, NPARTS, 0, dashing_amd.ESTIM_ERTL_MLE, dashing_amd.JI, K)
            computed = ctx.event_record()  # (a ticket orders nothing between the streams: the transfers are not held back)
            ctx.exchange_collect_async(n, rows_of, NPARTS, 0 if rank == 0 else local.data_ptr(), final.data_ptr() if rank == 0 else 0, 0)
            ctx.event_wait(computed)
            t1 = time.perf_counter()
            ctx.comm_wait()  # (with a deadline: a missing peer is an error, not a hang)
            t2 = time.perf_counter()
            if timed:
                phase["compute"] += t1 - t0
                phase["exchange"] += t2 - t1  # what is left of the exchange after the last kernel
            return final
        if rows_of is not None:  # (every rank keeps its rows: the same exchange-aware layout, no transfer)
            ctx.exchange_rows_device_async(local.data_ptr(), rows_of, rank, NPARTS, 0, dashing_amd.ESTIM_ERTL_MLE, dashing_amd.JI, K)
        else:
            ctx.dist_rows_device(local.data_ptr(), bounds[rank], bounds[rank + 1], dashing_amd.ESTIM_ERTL_MLE, dashing_amd.JI, K)
        ctx.synchronize()
        t1 = time.perf_counter()
        if multi and do_gather:
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
        return final if do_gather else local

    def fence():
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    def timed_loop(do_gather):
        phase["compute"] = phase["exchange"] = 0.0
        for _ in range(args.warmup):
            step(False, do_gather)
        fence()
        t0 = time.perf_counter()
        res = None
        for _ in range(args.steps):
            res = step(True, do_gather)
        fence()
        dt_ = time.perf_counter() - t0
        ph = [phase["compute"] / max(args.steps, 1) * 1e3, phase["exchange"] / max(args.steps, 1) * 1e3]
        phase["mine_ms"] = list(ph)  # (this rank's own; `ph` becomes the max over ranks below)
        if multi:
            t = torch.tensor([dt_] + ph, dtype=torch.float64, device=cpu_or_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt_, ph = float(t[0].item()), [float(t[1].item()), float(t[2].item())]
        return dt_, ph, res

    # ---- probe (N > 1, gathered, C-ABI exchange).  The library's exchange has only ever run over a stand-in transport on
    # one GPU; the driver's multi-GPU run may be the only real one.  ONE untimed step must therefore reproduce the
    # single-GPU matrix bit for bit before anything is timed; if it does not (or fails, or misses its deadline), the
    # parts are announced by events instead of flags (finalize_signal = 0) and the probe repeated; if that fails too,
    # torch.distributed carries contiguous spans (the fallback of a library that cannot load RCCL).  Every rank takes
    # the same turn (all-reduce); the line says which path was timed and why.
    probe = None
    if multi and gather and use_cabi:
        ref_p = None
        if rank == 0:
            ref_p = torch.empty(total_pairs, dtype=torch.float32, device=dev)
            ctx.attach_device(regs_d.data_ptr(), n, p)
            ctx.dist_rows_device(ref_p.data_ptr(), 0, n, dashing_amd.ESTIM_ERTL_MLE, dashing_amd.JI, K)
            ctx.synchronize()
        inject = int(os.environ.get("DSH_BENCH_INJECT_PROBE_FAIL", "0"))  # (tests: the first k probes "fail")
        attempts = []

        def probe_once(label):
            ok, why = 1, None
            old_to = os.environ.get("DSH_COMM_TIMEOUT_S")
            os.environ["DSH_COMM_TIMEOUT_S"] = os.environ.get("DSH_BENCH_PROBE_TIMEOUT_S", "45")
            try:
                res = step(False, True)
                if rank == 0 and not torch.equal(ref_p, res[:total_pairs]):
                    ok, why = 0, "the assembled matrix differs from the single-GPU one in %d values" % int((ref_p != res[:total_pairs]).sum().item())
                if len(attempts) < inject:
                    ok, why = 0, "injected (DSH_BENCH_INJECT_PROBE_FAIL)"
            except Exception as e:  # noqa: BLE001
                ok, why = 0, "%s: %s" % (type(e).__name__, e)
            finally:
                if old_to is None:
                    os.environ.pop("DSH_COMM_TIMEOUT_S", None)
                else:
                    os.environ["DSH_COMM_TIMEOUT_S"] = old_to
            flag = torch.tensor([ok], device=cpu_or_dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            attempts.append({"path": label, "ok_on_this_rank": bool(ok), "ok_on_every_rank": int(flag.item()) == 1, "why": why})
            return int(flag.item()) == 1

        if probe_once("c-abi exchange, parts announced as configured (flags where the device has stream wait-value)"):
            probe = {"timed_path": "c-abi exchange", "attempts": attempts}
        else:
            try:
                ctx.set_option("finalize_signal", 0)
            except Exception:  # noqa: BLE001
                pass
            if probe_once("c-abi exchange, finalize_signal = 0 (one k_finalize launch and one event per part)"):
                probe = {"timed_path": "c-abi exchange with finalize_signal = 0", "attempts": attempts}
                exchange += " [finalize_signal = 0 after a failed probe]"
            else:
                try:
                    ctx.comm_destroy()
                except Exception:  # noqa: BLE001
                    pass
                use_cabi, rows_of, my_floats = False, None, None
                exchange = ("torch.distributed" if backend == "nccl" else "gloo (host staged)") + " [fallback: the c-abi exchange failed its probe twice]"
                sizes = multigpu.span_sizes(n, bounds)
                my_pairs = sizes[rank]
                alloc_buffers(gather)
                probe = {"timed_path": exchange, "attempts": attempts}
                if rccl_info is not None:
                    rccl_info["carried_the_exchange"] = "torch.distributed (%s)" % rccl_info["torch_bundled_rccl"] if backend == "nccl" else "gloo"
                sys.stderr.write("bench.py: rank %d: C-ABI exchange failed its probe (%s): torch.distributed exchange\n" % (rank, attempts[-1]["why"]))
        del ref_p
    dt, phases, full = timed_loop(gather)
    ms_per_step = dt / args.steps * 1e3
    value = total_pairs * args.steps / dt
    line.update({"value": value, "ms_per_step": ms_per_step})
    gathered_variant = None
    if multi and not gather and args.exchange == "auto":
        # the second figure: the same step with every span collected on rank 0 (what bounds a configs[3]-sized job on 8 GPUs)
        try:
            alloc_buffers(True)
            dt_g, ph_g, _ = timed_loop(True)
            gathered_variant = {"ms_per_step": dt_g / args.steps * 1e3, "pairs_per_s": total_pairs * args.steps / dt_g,
                                "phase_ms_max_over_ranks": {"compute_incl_prepare": round(ph_g[0], 4), "exchange": round(ph_g[1], 4)}}
        except Exception as e:  # noqa: BLE001
            gathered_variant = {"error": "%s: %s" % (type(e).__name__, e)}
    local = buf["local"]

    # ---- kernel phases: HIP events on the library stream, outside the timed region
    reps = 3
    if rows_of is not None:
        km = measure_kernels(ctx, regs_d, n, p, [lambda: ctx.exchange_rows_device_async(local.data_ptr(), rows_of, rank, NPARTS, 0, dashing_amd.ESTIM_ERTL_MLE, dashing_amd.JI, K)], reps)
    else:
        km = measure_kernels(ctx, regs_d, n, p, [(local.data_ptr(), bounds[rank], bounds[rank + 1])], reps)
    pair_ms, fin_ms, prep_ms, launches = km["pair_ms"], km["finalize_ms"], km["prepare_ms"], km["pair_launches"]
    kphase = [pair_ms / reps, fin_ms / reps, prep_ms / reps]
    cyc, pbind = pair_binding(ctx, pair_ms / reps)
    lockstep = bool(ctx.info("lockstep"))
    rinfo = {"dense_planes": ctx.info("planes"), "reg_value_range": [ctx.info("vlo"), ctx.info("vhi")],
             "listed_tail_caps": {"upper": ctx.info("emax"), "lower": ctx.info("elow")}, "avg_planes_per_tile": ctx.info("avg_tile_planes_x100") / 100.0,
             "key_ordered_columns": bool(ctx.info("sorted"))}
    if multi:
        t = torch.tensor(kphase, dtype=torch.float64, device=cpu_or_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        kphase = [float(x) for x in t.tolist()]

    src = source_hash()
    single = rank == 0 and not multi
    # ---- PMC: the child passes of this run (single GPU only), else the committed file of the same sources
    pmc, pmc_note, cache_env, cache_files = None, "no PMC passes in this run", {}, []
    want_cfg = single and not args.no_secondary and (n, p) == (10000, 14)
    regs4_h = synth.survey_sketches(100_000, 10, seed=0x5EED0000)[0] if want_cfg else None
    if single and not args.no_pmc and not os.environ.get("DSH_BENCH_NO_PMC"):
        if want_cfg:
            cache_env["DSH_BENCH_PMC_BAND"] = "1"
        for tag, arr in (("C3", regs_h), ("C4", regs4_h)):
            if arr is not None:
                path = "/tmp/dsh_bench_%s_%d.npy" % (tag, os.getpid())
                np.save(path, arr)
                cache_env["DSH_BENCH_CACHE_" + tag] = path
                cache_files.append(path)
        try:
            pmc, pmc_note = live_pmc(n, p, cache_env)
        finally:
            for f_ in cache_files:
                try:
                    os.unlink(f_)
                except OSError:
                    pass
    pmc = pmc or {}
    traffic, traffic_note = hbm_bytes(pmc.get("C3_pair")), pmc_note
    if traffic is not None:
        traffic /= max(pmc["C3_pair"].get("launches_FETCH_SIZE", 1), 1)
    else:
        try:
            pm = json.load(open(os.path.join(ROOT, "profiles", "pmc_pair_kernel.json")))
            if pm.get("source_sha256") != src:
                traffic_note += "; profiles/pmc_pair_kernel.json was measured on other kernel sources (sha256 differs): refused as stale"
            elif pm["workload"]["n_sketches"] == n and pm["workload"]["p"] == p and world == 1:
                traffic = pm["hbm_bytes_per_launch"]
                traffic_note = "bytes/launch, rocprofv3 PMC (2*FETCH_SIZE + WRITE_SIZE) collected by tools/pmc_collect.py in separate --pmc passes on these kernel sources (sha256 checked); not measured in this run [" + pmc_note + "]"
        except (OSError, KeyError, ValueError):
            pass
    b_pair = 2 * m + 4                                   # algorithmic bytes per pair
    achieved = my_pairs * reps * b_pair / (pair_ms * 1e-3) / 1e9 if pair_ms > 0 else 0.0
    avg_launch_ms = pair_ms / max(launches, 1)
    fbind = finalize_binding(kphase[1], my_pairs, pmc.get("C3_finalize"))
    # `bound` / `frac` name the resource that BINDS the kernel -- integer VALU issue of its (v_and_b32, v_bcnt_u32_b32) pairs --
    # in the contract's shape (achieved / peak in wave-instruction pairs per second over the whole chip).  The SURVEY 8d
    # streaming figure (2*2^p + 4 bytes per pair against the HBM peak) exceeds 1 for an LDS-tiled kernel (each staged sketch
    # is reused 128x): it lives under `streaming_model`, never under the key a reader takes for efficiency.
    slots_per_pass = pair_slots(ctx)
    ach_pairs = slots_per_pass / (pair_ms / reps * 1e-3) / 1e9 if pair_ms > 0 else 0.0
    peak_pairs = N_SIMD * CLOCK_HZ / 6.0 / 1e9
    roofline = {
        "bound": "int VALU issue", "achieved": round(ach_pairs, 2), "peak": round(peak_pairs, 2),
        "unit": "G wave64 (v_and_b32 + v_bcnt_u32_b32) pairs/s, whole chip (peak = 1024 SIMDs x 2.4 GHz / 6 issue cycles)",
        "frac": pbind["frac"], "traffic": traffic, "traffic_note": traffic_note,
        "binding": pbind,
        "kernel": "k_pair_counts_ls" if lockstep else "k_pair_counts", "launches_per_step": launches // reps,
        "avg_launch_ms": round(avg_launch_ms, 4),
        "pairs_per_launch_avg": my_pairs * reps // max(launches, 1),
        **rinfo,
        "streaming_model": {
            "bytes_per_pair": b_pair, "achieved_gbs": round(achieved, 1), "hbm_peak_gbs": HBM_PEAK_GBS,
            "frac_of_hbm": round(achieved / HBM_PEAK_GBS, 4),
            "note": "SURVEY 8d algorithmic bytes (2*2^p+4 per pair, the reference's own traffic) / the tile kernel's launch time: a fraction above 1 only says the LDS-tiled kernel is not HBM-bound"},
        "streaming_model_frac_of_hbm": round(achieved / HBM_PEAK_GBS, 4),
        "note": "the tile kernel issues one v_and_b32 + one v_bcnt_u32_b32 per 32 pair-bits and nothing else per k-row (64 + 64 + 4 ds_read_b128 + 1 s_barrier): frac = 6 issue cycles / measured wall-cycles per pair; physical HBM is physical_hbm_gbs (PMC)",
        "physical_hbm_gbs": round(traffic / (avg_launch_ms * 1e-3) / 1e9, 1) if traffic and pair_ms > 0 else None,
        "physical_hbm_frac_of_peak": round(traffic / (avg_launch_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if traffic and pair_ms > 0 else None,
        "compulsory_bytes_per_step": n * m + 4 * total_pairs,
    }
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
        "kernel": "k_finalize", "ms_per_step": round(kphase[1], 4), "bound": "fp64 VALU issue", "binding": fbind,
        "pairs_per_s": round(my_pairs / (kphase[1] * 1e-3), 1) if kphase[1] > 0 else 0.0,
        "physical_hbm_bytes_per_step": hbm_bytes(pmc.get("C3_finalize")),
        "note": "VALU-issue bound; about half of its VALU instructions are the Ertl-MLE estimator (secant iterations x live bins x ~20 instructions, 15 of them dependent fp64 operations) that must be reproduced bit for bit; the sparse tails of the histogram come from a position-index join (k_build_colindex) -- profiles/r4*, DESIGN.md 3.5",
    }
    roofline["step"] = {
        "ms": {"prepare": round(kphase[2], 4), "pair_counts": round(kphase[0], 4), "finalize": round(kphase[1], 4)},
        "algorithmic_gbs_whole_step": round(total_pairs * b_pair / (ms_per_step * 1e-3) / 1e9, 1),
    }

    cpu = None
    parity = None
    if rank == 0 and multi:
        if gather:
            # assembled multi-rank matrix vs one single-GPU call on rank 0 (outside the timed region)
            ref = torch.empty(total_pairs, dtype=torch.float32, device=dev)
            ctx.attach_device(regs_d.data_ptr(), n, p)
            ctx.dist_rows_device(ref.data_ptr(), 0, n, dashing_amd.ESTIM_ERTL_MLE, dashing_amd.JI, K)
            ctx.synchronize()
            parity = {"assembled_equals_single_gpu": bool(torch.equal(ref, full[:total_pairs])), "pairs_checked": total_pairs}
            del ref
        else:
            parity = {"assembled_equals_single_gpu": None, "note": "spans kept on their ranks (not gathered): rank 0's own span is checked against the CPU oracle"}
    if rank == 0 and not args.no_cpu_baseline:
        # every line carries the CPU leg -- with N ranks a bounded 4-s sample on rank 0, outside the
        # timed loop, while the other ranks wait in the next collective.  A gathered run holds the whole assembled matrix
        # on rank 0; otherwise only the rows rank 0 computed can be compared.
        cpu, par2 = cpu_baseline(regs_h, full, n, p, args.cpu_seconds if not multi else min(args.cpu_seconds, 4.0),
                                 max_rows=None if (not multi or gather) else (rows_of.rows(0)[0][1] if rows_of is not None else bounds[1]))
        parity = par2 if parity is None else {**parity, "rank0_span_vs_cpu": par2}

    line.update({
        "config": {"workload": "BASELINE configs[2]: %d synthetic sketches, p=%d (%d B each), all-pairs dist on MI355X" % (n, p, m),
                   "n_sketches": n, "p": p, "k": K, "estimator": "ERTL_MLE", "result": "JI",
                   "sharding": ("row sets: a tile-aligned row range of the final triangle per rank + top-up tile rows from its bottom (dsh_balance_rowsets: equal tiles + prepare per rank), plane matrix laid out per rank; pipelined point-to-point send of each rank's parts to rank 0, which puts the rows into place; no un-permute" if rows_of is not None else
                                "tile-count-balanced row ranges of the final triangle, one per rank (plane matrix laid out per range); point-to-point send of each span into place on rank 0, no un-permute") if multi else "single GPU"},
        "roofline": roofline,
        "cpu_baseline": cpu,
        "parity_vs_cpu": parity,
        "kernel_source_sha256": src,
    })
    if multi:
        diag = multi_gpu_diagnostics(ctx, torch, dist, dashing_amd, multigpu, dev, regs_d, n, p, rank, world, rows_of, bounds, NPARTS, use_cabi,
                                     km, reps, phase.get("mine_ms", [0.0, 0.0]), my_pairs, ms_per_step, buf, gather)
        line["multi_gpu"] = {
            **diag,
            "ranks": world, "rccl_ranks": dist.get_world_size() if backend == "nccl" else 0, "backend": backend,
            "exchange": exchange if gather else "none: every rank keeps its span (output %.1f GB > %.1f GB; --exchange gather collects it)" % (4 * total_pairs / 1e9, GATHER_LIMIT_BYTES / 1e9),
            "exchange_library": rccl_info, "probe": probe, "row_bounds": bounds if rows_of is None else None,
            "row_sets": rows_of.describe() if rows_of is not None else None, "pairs_per_rank": sizes,
            "phase_ms_max_over_ranks": {"compute_incl_prepare": round(phases[0], 4), "exchange" if not use_cabi else "exchange_exposed_after_last_kernel": round(phases[1], 4),
                                        "k_pair_counts": round(kphase[0], 4), "k_finalize": round(kphase[1], 4), "prepare": round(kphase[2], 4)},
            "exchange_bytes_into_rank0": 4 * (total_pairs - sizes[0]) if gather else 0,
            "gathered_variant": gathered_variant,
        }

    if want_cfg:
        # ---- per-config entries: each guarded so that a failure costs its entry, not the line
        cfgs = []

        def guarded(name, fn):
            try:
                cfgs.append(fn())
            except Exception as e:  # noqa: BLE001
                sys.stderr.write(traceback.format_exc())
                cfgs.append({"workload": name, "error": "%s: %s" % (type(e).__name__, e)})

        guarded("BASELINE configs[1]", lambda: config_sketch(ctx, torch, dev, dashing_amd, pmc.get("sketch"), args))
        cfgs.append({"workload": line["config"]["workload"], "pairs_per_s": value, "ms_per_step": ms_per_step,
                     "roofline": {"binding": pbind, "streaming_model_frac_of_hbm": roofline["streaming_model_frac_of_hbm"], "physical_hbm_gbs": roofline["physical_hbm_gbs"]},
                     "cpu_baseline": None if cpu is None else {"value": cpu["value"], "cores": cpu["cores"], "sample": cpu["sample"]},
                     "parity": parity, "note": "the headline: full detail at top level"})
        guarded("BASELINE configs[3] shape", lambda: config_c4(ctx, torch, dev, dashing_amd, regs4_h, pmc, args))
        regs4_h = None
        guarded("BASELINE configs[4] shape, one band", lambda: config_c5_band(ctx, torch, dev, dashing_amd, synth, pmc, args))
        line["configs"] = cfgs
        line["secondary"] = next((c for c in cfgs if c.get("workload", "").startswith("BASELINE configs[3]") and "error" not in c), None)
        try:
            line["data_dependence"] = data_dependence(ctx, torch, dev, dashing_amd, n, p)
        except Exception as e:  # noqa: BLE001
            line["data_dependence"] = {"error": "%s: %s" % (type(e).__name__, e)}

    if single and args.what_if:
        line["what_if_mfma"] = what_if_mfma(ctx, torch, dashing_amd, regs_d, full, local, n, p, total_pairs)
    ctx.close()


def link_ping(ctx, torch, dist, dashing_amd, dev, rank, world, mb):
    """Measured point-to-point rate of the library's own communicator (dsh_collect_spans: grouped ncclSend/ncclRecv into
    rank 0): `mb` MB from ONE source at a time (every source in turn) and from all sources at once.  GB/s as seen by rank 0
    (barrier, then the blocking call; best of 3 after a warm-up that also pays RCCL's channel set-up)."""
    floats = mb * (1 << 20) // 4

    def tri_n(f):  # smallest n whose triangle holds f floats
        n_ = int((2 * f) ** 0.5)
        while n_ * (n_ - 1) // 2 < f:
            n_ += 1
        return n_

    n1, nc = tri_n(floats), tri_n(floats * world)
    bc = dashing_amd.partition_rows(nc, world, 1)
    span_c = [dashing_amd.tri_span(nc, bc[r], bc[r + 1]) for r in range(world)]
    tot1, totc = n1 * (n1 - 1) // 2, nc * (nc - 1) // 2
    fin = torch.zeros(max(tot1, totc), dtype=torch.float32, device=dev) if rank == 0 else None
    loc = torch.ones(max(tot1, max(span_c)), dtype=torch.float32, device=dev) if rank != 0 else None
    torch.cuda.synchronize()

    def timed(nn, b):
        best = None
        for it in range(4):
            dist.barrier()
            t0 = time.perf_counter()
            ctx.collect_spans(nn, b, 0 if rank == 0 else loc.data_ptr(), fin.data_ptr() if rank == 0 else 0, 0, wait=True)
            dt_ = time.perf_counter() - t0
            if it and (best is None or dt_ < best):
                best = dt_
        return best

    single = {}
    for src in range(1, world):
        b = [0] * (src + 1) + [n1] * (world - src)
        t = timed(n1, b)
        single[str(src)] = round(4 * tot1 / t / 1e9, 2)
    t = timed(nc, bc)
    into0 = 4 * (totc - span_c[0])
    conc = into0 / t / 1e9
    ok = bool(rank != 0 or float(fin[span_c[0]:totc].min().item()) == 1.0)  # (what arrived is what was sent)
    return {"message_MB": mb, "single_GBs_by_source": single, "single_GBs_min": min(single.values()), "single_GBs_max": max(single.values()),
            "concurrent_total_GBs": round(conc, 2), "concurrent_per_link_GBs": round(conc / (world - 1), 2), "payload_intact": ok,
            "how": "dsh_collect_spans (grouped ncclSend/ncclRecv on the ctx stream) into rank 0, wall time on rank 0 behind a barrier, best of 3"}


def multi_gpu_diagnostics(ctx, torch, dist, dashing_amd, multigpu, dev, regs_d, n, p, rank, world, rows_of, bounds, nparts, use_cabi,
                          km, reps, mine_ms, my_pairs, ms_per_step, buf, gather):
    """What makes an N-rank line readable on its own: every rank's phases (not only the max), the link
    rate measured through the library's communicator, and the step tools/shard_model.py's pipeline model predicts for THESE
    per-rank times at THAT rate -- prediction and measurement side by side."""
    if rows_of is not None:
        rs, k, _ = dashing_amd.exchange_mode(n, rows_of, rank, nparts, 0, want_floats=True)
        my_rows = rows_of.rows(rank)
    else:
        rs, k, my_rows = False, 1, [(bounds[rank], bounds[rank + 1])]
    ctx.set_profiling(True)  # one more pass for the parts' completion times (events; outside the timed loop)
    ctx.attach_device(regs_d.data_ptr(), n, p)
    if rows_of is not None:
        ctx.exchange_rows_device_async(buf["local"].data_ptr(), rows_of, rank, nparts, 0, dashing_amd.ESTIM_ERTL_MLE, dashing_amd.JI, K)
        ctx.synchronize()
    pinfo = ctx.last_part_info() if rows_of is not None else []
    ctx.set_profiling(False)
    items = ctx.info("items")  # (of the rank's own job: rank 0 has run the single-GPU reference pass in between)
    mine = {"rank": rank, "part_info": [(round(a, 4), b) for a, b in pinfo], "rows": my_rows, "pairs": my_pairs, "span_bytes": 4 * my_pairs, "tiles": ctx.info("tiles"), "items": items,
            "rounds_of_512": -(-items // 512), "planes_per_tile": ctx.info("avg_tile_planes_x100") / 100.0, "parts": k, "rowsorted": rs,
            "bands": ctx.info("bands"), "prepare_ms": round(km["prepare_ms"] / reps, 4), "pair_ms": round(km["pair_ms"] / reps, 4),
            "finalize_ms": round(km["finalize_ms"] / reps, 4), "wall_ms": round(mine_ms[0], 4), "exposed_exchange_ms": round(mine_ms[1], 4)}
    allr = [None] * world
    dist.all_gather_object(allr, mine)
    out = {"per_rank": allr,
           "per_rank_note": "wall_ms = attach + prepare + tile kernel + k_finalize of the rank's rows per step (host clock); exposed_exchange_ms = what is left of the exchange after the rank's last kernel; prepare/pair/finalize_ms = HIP events outside the timed loop"}
    ping = None
    if use_cabi and world > 1 and not os.environ.get("DSH_BENCH_NO_PING"):
        try:
            ping = link_ping(ctx, torch, dist, dashing_amd, dev, rank, world, int(os.environ.get("DSH_BENCH_PING_MB", "64")))
        except Exception as e:  # noqa: BLE001
            ping = {"error": "%s: %s" % (type(e).__name__, e)}
    out["link_gbs_measured"] = ping if ping is not None else {"skipped": "the library's communicator is not in use (torch.distributed exchange) or one rank"}
    # the destination's placement rate of a row-sorted source (rank 1's rows computed here, then put into place)
    place_rate = None
    if rank == 0 and rows_of is not None and world > 1 and gather and buf["final"] is not None:
        rs1, _, fl1 = dashing_amd.exchange_mode(n, rows_of, 1, nparts, 0, want_floats=True)
        if rs1 and fl1:
            tmp = torch.empty(fl1, dtype=torch.float32, device=dev)
            ctx.attach_device(regs_d.data_ptr(), n, p)
            ctx.exchange_rows_device_async(tmp.data_ptr(), rows_of, 1, nparts, 0, dashing_amd.ESTIM_ERTL_MLE, dashing_amd.JI, K)
            ctx.synchronize()
            best = 1e9
            ctx.set_profiling(True)
            for _ in range(3):
                ctx.exchange_place_device(rows_of, 1, nparts, tmp.data_ptr(), buf["final"].data_ptr(), 0)  # (the values it already holds)
                best = min(best, max(ctx.info("place_kernel_us"), 1) * 1e-6)  # the placement kernel's device time
            ctx.set_profiling(False)
            place_rate = 4 * fl1 / best
            del tmp
    if rank == 0 and world > 1:
        def predict(g, interference=None):
            ms, worst = multigpu.pipeline_model(allr, place_rate, g, nmsg=nparts, dst_interference=interference)
            return {"step_model_ms": round(ms, 4), "bound_by": "link/placement of rank %d" % worst if worst else "compute (slowest rank)"}

        model = {"what": "dashing_amd.multigpu.pipeline_model (the model of tools/shard_model.py) over the per-rank times above, following dsh_exchange_collect_async: %d rounds, message q of every source (the q-th share of its buffer, ready when the part that holds its last value is final) over its own link into rank 0, a round as long as its largest message + 20 us, the completed rows of a round placed by one launch at place_rate beside the next round" % nparts,
                 "place_rate_GBs": round(place_rate / 1e9, 2) if place_rate else None,
                 "sensitivity_by_assumed_link_GBs": {"%g" % g: predict(g) for g in (30.0, 45.0, 60.0)},
                 "with_measured_recv_interference_at_45_GBs": {k_: predict(45.0, v_) for k_, v_ in multigpu.MEASURED_RECV_INTERFERENCE.items()},
                 # the three things the model ASSUMES: the first real run names the wrong one
                 "model_assumptions": {"link_gbs": [30.0, 45.0, 60.0], "round_overhead_us": 20.0,
                                       "interference": "none in step_model_ms; with_measured_recv_interference_* applies what a waiting kernel of 7-28 workgroups cost the destination's kernels on one GPU (profiles/rd6a/interference_probe.jsonl): x1.13 tile kernel (not when the receives are gated behind it) and x1.17 k_finalize if it holds > 32 KB of LDS, x1.02 / x1.17 at 4 KB"},
                 "measured_ms_per_step": round(ms_per_step, 4)}
        if ping and "concurrent_per_link_GBs" in ping and ping["concurrent_per_link_GBs"] > 0:
            model["at_measured_link_rate"] = {"link_GBs": ping["concurrent_per_link_GBs"], **predict(ping["concurrent_per_link_GBs"])}
            model["measured_over_predicted"] = round(ms_per_step / model["at_measured_link_rate"]["step_model_ms"], 3)
        out["model"] = model
    return out


def what_if_mfma(ctx, torch, dashing_amd, regs_d, full, local, n, p, total_pairs):
    """WHAT-IF, never the product path (the north star excludes the matrix cores; `value` is the integer-VALU kernel): the
    same tile kernel with the AND+popcount done as a 0/1 i8 MFMA (option pair_mfma; only in a library built with
    `make WHATIF=1`), same inputs.  Run only with --what-if."""
    if not ctx.info("whatif_mfma"):
        return {"skipped": "libdashing_hip.so was built without the what-if kernel (make -C dashing_amd/csrc WHATIF=1)"}
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
    return {"label": "WHAT-IF ONLY, not the shipped path and not `value`: v_mfma_i32_32x32x32_i8 on 0/1 bytes expanded from the bit-planes in registers (option pair_mfma=1, default 0, library built with WHATIF=1)",
            "ms_per_step": round(min(ts) * 1e3, 3), "pairs_per_s": total_pairs / min(ts),
            "k_pair_counts_mfma_ms": round(kw["pair_ms"], 3), "output_identical_to_valu_path": same}


def cli_end_to_end(torch, dev, seq, G, L, p, want_tri, threads):
    """`dashing-amd dist -b` from FASTA files (80-column lines) on a RAM-backed file system: the whole of configs[1] as a user
    runs it -- process start, context creation, the raw file bytes read into page-locked staging, batched upload, FASTA
    decode ON THE DEVICE (dsh_sketch_fastx_batch_async) + k_sketch, dist, the binary matrix written.  Bound by the HIP
    runtime's start-up and the PCIe stream, labelled so; the matrix must equal the in-process one."""
    import shutil
    import subprocess
    import tempfile

    need = int(G * L * 1.02) + (64 << 20)
    where = None
    for cand in ("/dev/shm", "/tmp"):
        try:
            if shutil.disk_usage(cand).free > 2 * need + (4 << 30):
                where = cand
                break
        except OSError:
            pass
    if where is None:
        return {"skipped": "no file system with %.1f GB free for the FASTA files" % (2 * need / 1e9)}
    cli = os.path.join(ROOT, "dashing_amd", "dashing-amd")
    d = tempfile.mkdtemp(prefix="dsh_c1_", dir=where)
    try:
        t0 = time.perf_counter()
        nl = torch.full((G * L // 80, 1), 10, dtype=torch.uint8, device=dev)
        fa = torch.cat([seq[: G * L].view(-1, 80), nl], dim=1).view(G, L // 80 * 81)  # the line breaks, inserted on the device
        paths = []
        for g_ in range(G):
            pth = os.path.join(d, "g%04d.fna" % g_)
            with open(pth, "wb") as f:
                f.write(b">genome%d\n" % g_)
                f.write(fa[g_].cpu().numpy().tobytes())
            paths.append(pth)
        del fa, nl
        lst = os.path.join(d, "paths.txt")
        with open(lst, "w") as f:
            f.write("\n".join(paths) + "\n")
        t_write = time.perf_counter() - t0
        out = os.path.join(d, "dist.bin")
        walls = []
        for _ in range(2):
            time.sleep(0.5)  # (the driver tears the previous process down: its dsh_create otherwise takes 0.24 s instead of 0.07)
            t0 = time.perf_counter()
            r = subprocess.run([cli, "dist", "-k", str(K), "-S", str(p), "-p", str(threads), "-b", "--avoid-sorting", "-O", out, "-o", os.devnull, "-F", lst],
                               capture_output=True, timeout=300)
            walls.append(time.perf_counter() - t0)
            if r.returncode != 0:
                return {"error": "dashing-amd dist: rc %d: %s" % (r.returncode, r.stderr.decode(errors="replace")[-400:])}
        raw = np.fromfile(out, np.uint8)
        got = raw[9:].view(np.float32)
        same = bool(got.size == want_tri.size and (got == want_tri).all())
        wall = min(walls)
        return {"what": "wall time of `dashing-amd dist -k%d -S%d -p%d -b --avoid-sorting -F <%d FASTA files of %d bp, 80-column lines, on %s>`: process start + context + raw file bytes into page-locked staging + upload + FASTA decode on the device + k_sketch + dist + the binary matrix" % (K, p, threads, G, L, where),
                "wall_s": round(wall, 4), "wall_s_runs": [round(w, 4) for w in walls], "bases_per_s": G * L / wall, "host_threads": threads,
                "matrix_equals_in_process_result": same, "fasta_write_s_not_counted": round(t_write, 2), "file_bytes": int(sum(os.path.getsize(p_) for p_ in paths)),
                "bound": "HIP runtime start-up (~0.1 s) + host staging copies / PCIe (%d host threads)" % threads}
    finally:
        shutil.rmtree(d, ignore_errors=True)


def config_sketch(ctx, torch, dev, dashing_amd, pm, args, G=1000, L=5_000_000, p=10):
    """BASELINE configs[1] at its stated size and scope: 1 000 synthetic 5 Mbp genomes resident in HBM, k=31, p=10 -- k_sketch
    (hot loop 1, src/sketch_and_cmp.h:314-360's replacement) + dist of the 1 000 sketches: bases/s of the kernel, the
    sketch+dist step, fraction of HBM at 1 B/base, VALU-issue fraction, the CPU oracle on >= 1 s of the same genomes,
    registers of 20 sampled genomes bit-exact, the distances within 1e-6; then the same job through the CLI from FASTA."""
    from oracle import oracle_c

    seq, wall_s, kernel_ms, regs = sketch_workload(ctx, torch, dev, G, L, p, 3)
    bases = G * L
    offs = np.arange(G + 1, dtype=np.uint64) * np.uint64(L)
    total = G * (G - 1) // 2
    out = torch.empty(total, dtype=torch.float32, device=dev)
    # sketch + dist, one step: clear, k_sketch over all genomes, all-pairs dist of the fresh sketches
    steps = []
    for _ in range(3):
        ctx.clear()
        ctx.synchronize()
        t0 = time.perf_counter()
        ctx.sketch_batch_device(seq.data_ptr(), offs, 0, K, True)
        ctx.dist_rows_device(out.data_ptr(), 0, G, dashing_amd.ESTIM_ERTL_MLE, dashing_amd.JI, K)
        ctx.synchronize()
        steps.append(time.perf_counter() - t0)
    step_s = min(steps)
    tri = out.cpu().numpy()
    # registers bit-exact on 20 sampled genomes (decorated ones among them: every 10th holds an N run and a lowercase kilobase)
    cores = oracle_c.effective_cpus()
    oracle_c.load(threads=cores)
    rng = np.random.default_rng(11)
    sample = sorted(set([0, 10, G - 1] + [int(x) for x in rng.choice(G, 17, replace=False)]))
    hs = torch.cat([seq[g_ * L:(g_ + 1) * L] for g_ in sample]).cpu().numpy()
    want = oracle_c.sketch_batch(hs, np.arange(len(sample) + 1, dtype=np.uint64) * np.uint64(L), K, p, True)
    exact = bool((regs[sample] == want).all())
    # distances of all 499 500 pairs against the oracle's estimator on the device's registers (bit-exact to the oracle's own
    # on the sample above)
    ref = oracle_c.dist_tri(regs)
    rel = np.abs(tri.astype(np.float64) - ref) / np.maximum(np.abs(ref), 1e-9)
    # CPU oracle on >= 1 s of the same genomes, one genome per thread (src/sketch_and_cmp.h:314-360)
    nc = min(G, 2 * cores)
    h0 = seq[: nc * L].cpu().numpy()
    t0 = time.perf_counter()
    oracle_c.sketch_batch(h0, np.arange(nc + 1, dtype=np.uint64) * np.uint64(L), K, p, True)
    rate = nc * L / (time.perf_counter() - t0)
    nc = int(min(G, max(nc, np.ceil(1.3 * rate / L / cores) * cores)))
    h0 = seq[: nc * L].cpu().numpy()
    t0 = time.perf_counter()
    oracle_c.sketch_batch(h0, np.arange(nc + 1, dtype=np.uint64) * np.uint64(L), K, p, True)
    tc = time.perf_counter() - t0
    del h0
    ksec = kernel_ms * 1e-3
    # the kernel's static mix (tools/sketch_instr.py on k_sketch<false, true, true, 31, 256>, the all-valid copy of the unrolled
    # loop, final tree of round 6): 53.7 % of its VALU instructions are full-rate VOP1/VOP2 (2 issue cycles per wave64
    # instruction), the rest -- the 64-bit shifts and v_mad_u64_u32 of the Wang hash, the funnel shifts -- take 4; measured
    # alone 2.5 / 4.4 (profiles/ubench).  (Rounds 4-5 had 66.4 % here: the instructions round 6 removed were full-rate ones.)
    full_share = 0.537
    ceil_nominal = 2.0 * full_share + 4.0 * (1 - full_share)
    ceil_measured = 2.5 * full_share + 4.4 * (1 - full_share)
    binding = {"resource": "int VALU issue (24 of the ~42 instructions per k-mer are the 64-bit Wang hash, fixed by the bit-exactness contract)",
               "frac": None, "valu_insts_per_kmer": None, "ceiling_cycles_per_valu_inst": round(ceil_nominal, 3),
               "ceiling_note": "mix-aware: %.1f %% full-rate (2 cycles per wave64 instruction) + %.1f %% half-rate (4); measured alone 2.5 / 4.4 -> %.2f" % (
                   100 * full_share, 100 * (1 - full_share), ceil_measured)}
    if pm and pm.get("SQ_INSTS_VALU"):
        # the PMC child sketches PMC_SKETCH_GENOMES genomes of the same length: instructions per base carry over
        per_base = pm["SQ_INSTS_VALU"] * 64.0 / (PMC_SKETCH_GENOMES * L)
        cpi = ksec * CLOCK_HZ * N_SIMD / (per_base * bases / 64.0)
        binding.update({"valu_insts_per_kmer": round(per_base, 2), "cycles_per_valu_inst": round(cpi, 3), "frac": round(ceil_nominal / cpi, 4),
                        "frac_of_isolated_rates": round(ceil_measured / cpi, 4)})
    traffic = hbm_bytes(pm)
    # the parse on the device (dsh_sketch_fastx_batch_async): the first 40 genomes as FASTA text (80-column lines) from host
    # memory -- decode kernels timed by HIP events, registers compared with the ones sketched from the bases in HBM
    fastx = None
    try:
        ng = min(G, 40)
        nl = torch.full((ng * (L // 80), 1), 10, dtype=torch.uint8, device=dev)
        fa = torch.cat([seq[: ng * L].view(-1, 80), nl], dim=1).view(ng, L // 80 * 81).cpu().numpy()
        files = [b">genome%d\n" % g_ + fa[g_].tobytes() for g_ in range(ng)]
        del fa, nl
        ctx.sketch_fastx_batch(files[:2], 0, K, True)  # (the first use loads the decoder's code object: not what is timed)
        ctx.clear(0, ng)
        ctx.set_profiling(True)
        status = ctx.sketch_fastx_batch(files, 0, K, True)
        dec_us, sk_us = ctx.info("fastx_decode_us"), ctx.info("sketch_kernel_us")
        ctx.set_profiling(False)
        fbytes = sum(len(f_) for f_ in files)
        fastx = {"what": "dsh_sketch_fastx_batch_async on %d of the genomes as FASTA text (80-column lines, %d bytes of file): header lines -> one invalid byte, newlines removed, on the device (k_fastx_scan / _offsets / _compact / _pad), then k_sketch" % (ng, fbytes),
                 "decode_ms": round(dec_us / 1e3, 4), "decode_file_GBs": round(fbytes / max(dec_us, 1) / 1e3, 1), "bytes_moved_per_file_byte": 3,
                 "k_sketch_ms": round(sk_us / 1e3, 4), "refused": int((status != 0).sum()),
                 "registers_equal_those_sketched_from_hbm": bool((ctx.download(0, ng) == regs[:ng]).all())}
        del files
    except Exception as e:  # noqa: BLE001
        fastx = {"error": "%s: %s" % (type(e).__name__, e)}
    e2e = None
    if not args.no_secondary and not os.environ.get("DSH_BENCH_NO_CLI"):
        try:
            e2e = cli_end_to_end(torch, dev, seq, G, L, p, tri, cores)
            if e2e and "wall_s" in e2e:
                e2e["gpu_idle_fraction"] = round(1.0 - step_s / e2e["wall_s"], 4)
                e2e["gpu_idle_note"] = ("1 - (the in-process sketch+dist step, %.1f ms) / wall.  Where the wall goes (DSH_TIMING=1, profiles/rd6e ... rd6i/cli_e2e_timing.jsonl): ~0.10 s bringing the HIP runtime up (the first batch is staged meanwhile), "
                                        "~0.12 s streaming the 5.06 GB of file bytes through page-locked staging and PCIe at ~43 GB/s (host memory copies of 16 threads; the device decodes the FASTA text and sketches each 48 MB batch in 0.2 ms), ~0.07 s first batch + dist + output, and in most runs ~0.10 s between the program's _Exit and the parent seeing it end (the driver releasing the process's GPU state); the host parser was never the bound (it runs at the same 45 GB/s on 16 threads), it is at 2-4 threads" % (step_s * 1e3))
        except Exception as e:  # noqa: BLE001
            e2e = {"error": "%s: %s" % (type(e).__name__, e)}
    del seq, out
    torch.cuda.empty_cache()
    return {"workload": "BASELINE configs[1]: %d synthetic %d bp genomes resident in HBM, k=%d, p=%d (canonical k-mers): k_sketch, then dist of the %d sketches" % (G, L, K, p, G),
            "bases_per_s": bases / ksec, "ms_per_step": kernel_ms, "wall_ms_per_call": wall_s * 1e3, "steps": 3,
            "sketch_plus_dist": {"ms_per_step": round(step_s * 1e3, 3), "bases_per_s": bases / step_s, "pairs": total,
                                 "what": "clear + k_sketch over the %d genomes + all-pairs dist (Ertl-MLE JI) of the fresh sketches, wall" % G},
            "roofline": {"bound": "int VALU issue", "frac": binding["frac"], "binding": binding,
                         "streaming_model": {"bytes_per_base": 1, "achieved_gbs": round(bases / ksec / 1e9, 1), "frac_of_hbm": round(bases / ksec / 1e9 / HBM_PEAK_GBS, 4),
                                             "note": "HBM is the nominal roof at 1 B/base (SURVEY 8d); the kernel is bound by integer VALU issue"},
                         "physical_hbm_bytes_per_base": round(traffic / (PMC_SKETCH_GENOMES * L), 3) if traffic else None},
            "cpu_baseline": {"value": nc * L / tc, "unit": "bases/s", "cores": cores, "kind": "port",
                             "sample": "%d of the same genomes (%d bases) in %.2f s; oracle/dsh_oracle.c dsho_sketch_batch, one genome per thread as src/sketch_and_cmp.h:314-360" % (nc, nc * L, tc)},
            "parity": {"registers_bit_exact": exact, "genomes_checked": len(sample), "dist_pairs_checked": int(ref.size), "dist_max_rel_diff": float(rel.max()),
                       "tolerance": 1e-6, "note": PARITY_NOTE},
            "fastx_decode_on_device": fastx, "end_to_end_cli": e2e}


def config_c4(ctx, torch, dev, dashing_amd, regs, pmc, args, n=100_000, p=10):
    """configs[3]-shaped matrix on ONE GPU (the 8-GPU run is the driver's): at 1 KiB per sketch the popcounts are
    cheap and the per-pair estimator (k_finalize) is the hot kernel, which the p=14 headline hides."""
    from oracle import oracle_c

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
    _, pbind = pair_binding(ctx, km["pair_ms"])
    planes = ctx.info("avg_tile_planes_x100") / 100.0
    # CPU oracle on a sample of >= 1e7 pairs of the same matrix + parity on them
    cpu = parity = None
    if not args.no_cpu_baseline:
        cores = oracle_c.effective_cpus()
        lib = oracle_c.load(threads=cores)
        level = oracle_c.simd_level(lib)
        oracle_c.set_simd(level, lib)
        rows = 128
        t0 = time.perf_counter()
        ref = oracle_c.dist_rows(regs, 0, rows, lib=lib)
        tc = time.perf_counter() - t0
        oracle_c.set_simd(0, lib)
        got = out[: ref.size].cpu().numpy()
        rel = np.abs(got.astype(np.float64) - ref) / np.maximum(np.abs(ref), 1e-9)
        cpu = {"value": ref.size / tc, "unit": "pairs/s", "cores": cores, "kind": "port",
               "sample": "rows [0,%d) = %d pairs in %.2f s (oracle/dsh_oracle.c, %s histogram-of-max, OpenMP dynamic over j per row)" % (rows, ref.size, tc, {0: "scalar", 1: "avx2", 2: "avx512bw"}[level])}
        parity = {"pairs_checked": int(ref.size), "max_rel_diff": float(rel.max()), "tolerance": 1e-6, "exact_float32_matches": int((got == ref).sum())}
    fb = finalize_binding(km["finalize_ms"], total, pmc.get("C4_finalize"))
    phys_f, phys_p = hbm_bytes(pmc.get("C4_finalize")), hbm_bytes(pmc.get("C4_pair"))
    b_pair = 2 * (1 << p) + 4
    res = {"workload": "BASELINE configs[3] shape on one GPU: %d synthetic sketches, p=%d, full triangle (%.1f GB of float32 left in HBM)" % (n, p, total * 4 / 1e9),
           "value": total / t, "pairs_per_s": total / t, "unit": "pairs/s", "ms_per_step": t * 1e3, "steps": 2,
           "kernel_ms": {"k_pair_counts": round(km["pair_ms"], 3), "k_finalize": round(km["finalize_ms"], 3), "prepare": round(km["prepare_ms"], 3),
                         "pair_launches": km["pair_launches"]},
           "avg_planes_per_tile": planes,
           "finalize_cycles_per_wave64_of_pairs": round(km["finalize_ms"] * 1e-3 * CLOCK_HZ * N_SIMD / (total / 64.0), 1),
           "roofline": {"dominant_kernel": "k_finalize", "binding": fb, "tile_kernel_binding": pbind,
                        "streaming_model_gbs": round(total * b_pair / t / 1e9, 1), "streaming_model_frac_of_hbm": round(total * b_pair / t / 1e9 / HBM_PEAK_GBS, 4),
                        "physical_hbm_bytes_per_step": {"k_finalize": phys_f, "k_pair_counts": phys_p},
                        "physical_hbm_gbs": round((phys_f + phys_p) / t / 1e9, 1) if phys_f and phys_p else None,
                        "compulsory_bytes_per_step": n * (1 << p) + 4 * total},
           "cpu_baseline": cpu, "parity": parity}
    del out, regs_d
    torch.cuda.empty_cache()
    return res


C5_BAND = {"n": 300_000, "p": 14, "rows": 2048, "nbase": 4000}


def c5_sketches(torch, dev, synth, n, p, nbase):
    """configs[4]-shaped collection built on the device: sketch g >= nbase = max(base[a_g], base[b_g]), unions of two base
    sketches (as tests/test_gpu_configs.py)"""
    base = synth.survey_sketches(nbase, p, seed=0x5EED0000)[0]
    bd = torch.from_numpy(base).to(dev)
    regs = torch.empty((n, 1 << p), dtype=torch.uint8, device=dev)
    regs[:nbase] = bd
    g = torch.arange(nbase, n, device=dev, dtype=torch.int64)
    a, b2 = g % nbase, (g * 2654435761 + 12345) % nbase
    for s0 in range(0, n - nbase, 1 << 14):
        e0 = min(n - nbase, s0 + (1 << 14))
        regs[nbase + s0: nbase + e0] = torch.maximum(bd[a[s0:e0]], bd[b2[s0:e0]])
    del bd
    torch.cuda.synchronize()
    return regs


def config_c5_band(ctx, torch, dev, dashing_amd, synth, pmc, args, n=C5_BAND["n"], p=C5_BAND["p"], rows=C5_BAND["rows"], nbase=C5_BAND["nbase"]):
    """configs[4] shape (300 000 x p=14): ONE row range of the triangle on one GPU, extrapolated to the full matrix by pair
    count and labelled so; parity on sampled pairs of the band against the CPU oracle, the CPU oracle timed on >= 1e7 pairs
    of the band itself, physical HBM of the band's kernels from the PMC child."""
    from oracle import oracle_c

    regs = c5_sketches(torch, dev, synth, n, p, nbase)
    span = dashing_amd.tri_span(n, 0, rows)
    out = torch.empty(span, dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    ts = []
    for _ in range(2):
        ctx.attach_device(regs.data_ptr(), n, p)
        t0 = time.perf_counter()
        ctx.dist_rows_device(out.data_ptr(), 0, rows, dashing_amd.ESTIM_ERTL_MLE, dashing_amd.JI, K)
        ctx.synchronize()
        ts.append(time.perf_counter() - t0)
    ctx.set_profiling(True)
    ctx.attach_device(regs.data_ptr(), n, p)
    ctx.dist_rows_device(out.data_ptr(), 0, rows, dashing_amd.ESTIM_ERTL_MLE, dashing_amd.JI, K)
    ctx.synchronize()
    k = ctx.last_kernel_ms()
    ctx.set_profiling(False)
    _, pbind = pair_binding(ctx, k["pair_ms"])
    t = min(ts)
    total = n * (n - 1) // 2
    # parity: 2 rows x 3 000 sampled columns
    rng = np.random.default_rng(5)
    cols = np.sort(rng.choice(np.arange(rows, n), 3000, replace=False))
    qi = [1, rows - 1]
    q_h = regs[qi].cpu().numpy()
    c_h = regs[torch.from_numpy(cols).to(dev)].cpu().numpy()
    ref = oracle_c.dist_rect(q_h, c_h)
    got = np.stack([out[torch.from_numpy(np.array([dashing_amd.tri_index(n, i, int(j)) for j in cols], np.int64)).to(dev)].cpu().numpy() for i in qi])
    rel = np.abs(got.astype(np.float64) - ref) / np.maximum(np.abs(ref), 1e-9)
    # the CPU oracle on >= 1e7 pairs of the band itself: the band's first rows against a run of its columns
    cpu = None
    if not args.no_cpu_baseline:
        cores = oracle_c.effective_cpus()
        lib = oracle_c.load(threads=cores)
        level = oracle_c.simd_level(lib)
        oracle_c.set_simd(level, lib)
        qn, cn = 64, 160_000
        q2 = regs[:qn].cpu().numpy()
        c2 = regs[rows: rows + cn].cpu().numpy()
        t0 = time.perf_counter()
        ref2 = oracle_c.dist_rect(q2, c2)  # (the library loaded above: every core, the SIMD histogram)
        tc = time.perf_counter() - t0
        oracle_c.set_simd(0, lib)
        idx = torch.from_numpy(np.array([dashing_amd.tri_index(n, i, rows) for i in range(qn)], np.int64)).to(dev)
        got2 = torch.stack([out[int(i0): int(i0) + cn] for i0 in idx.tolist()]).cpu().numpy()
        rel2 = np.abs(got2.astype(np.float64) - ref2) / np.maximum(np.abs(ref2), 1e-9)
        cpu = {"value": ref2.size / tc, "unit": "pairs/s", "cores": cores, "kind": "port",
               "sample": "rows [0,%d) x columns [%d,%d) of the band = %d pairs in %.2f s (oracle/dsh_oracle.c, %s histogram-of-max)" % (
                   qn, rows, rows + cn, ref2.size, tc, {0: "scalar", 1: "avx2", 2: "avx512bw"}[level]),
               "max_rel_diff_on_the_sample": float(rel2.max())}
        del q2, c2
    phys_p, phys_f = hbm_bytes(pmc.get("C5_pair")), hbm_bytes(pmc.get("C5_finalize"))
    res = {"workload": "BASELINE configs[4] shape, ONE BAND: rows [0,%d) of %d synthetic sketches, p=%d (%.2e of the matrix's %.2e pairs) on one GPU" % (rows, n, p, span, total),
           "pairs_per_s": span / t, "ms_band": t * 1e3, "pairs_in_band": span,
           "extrapolated_full_matrix_s": round(total / (span / t), 2),
           "extrapolation": "full matrix = pairs / (band pairs per second); the band pays the per-sketch pass and the bit-plane transform of ALL %d columns (%.1f ms of %.1f), which a full pass pays once per row range of its own columns: pessimistic" % (n, k["prepare_ms"], t * 1e3),
           "kernel_ms": {"k_pair_counts": round(k["pair_ms"], 3), "k_finalize": round(k["finalize_ms"], 3), "prepare": round(k["prepare_ms"], 3)},
           "avg_planes_per_tile": ctx.info("avg_tile_planes_x100") / 100.0,
           "roofline": {"bound": "int VALU issue", "frac": pbind["frac"], "binding": pbind,
                        "streaming_model_frac_of_hbm": round(span * (2 * (1 << p) + 4) / t / 1e9 / HBM_PEAK_GBS, 4),
                        "physical_hbm_bytes_per_band": {"k_pair_counts": phys_p, "k_finalize": phys_f},
                        "physical_hbm_gbs": round((phys_p + phys_f) / t / 1e9, 1) if phys_p and phys_f else None},
           "cpu_baseline": cpu,
           "parity": {"pairs_checked": int(ref.size), "max_rel_diff": float(rel.max()), "tolerance": 1e-6, "exact_float32_matches": int((got == ref).sum())},
           "note": "the full 300 000 x p=14 matrix (45e9 pairs, 180 GB) is computed twice on one GPU by tests/test_gpu_configs.py::test_config4_full_300k_p14_eight_ranges; CPU rate per pair at p=14: see the headline's cpu_baseline"}
    del out, regs
    torch.cuda.empty_cache()
    return res


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


def cpu_baseline(regs_h, gpu_full, n, p, seconds, max_rows=None):
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
    if max_rows:
        rows = max(1, min(rows, int(max_rows)))  # (the rows `gpu_full` holds)
    ref, t = timed_rows(rows, level)
    # the scalar histogram on a small part of the sample, for the record (and as a cross-check of the SIMD one)
    rows_s = max(8, rows // 8)
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
              "exact_float32_matches": int((got == ref).sum()),
              "note": PARITY_NOTE}
    try:
        os.unlink(native)
    except OSError:
        pass
    return cpu, parity


if __name__ == "__main__":
    main()

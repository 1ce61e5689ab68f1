#!/usr/bin/env python3
"""k_finalize accounting: what the key-ordered output scatter, the estimator and each phase cost.

  python tools/finalize_accounting.py --tag r3a            # times (HIP events) + PMC passes, both workloads
  python tools/finalize_accounting.py --one C3 sort1 2     # (internal) one warm + one measured pass, for rocprofv3

Workloads: C3 = 10 000 x p=14 (the bench), C4 = 100 000 x p=10 (BASELINE configs[3] shape, one GPU).
Layouts:   sort1   key-ordered columns, values scattered to their final packed position (the default)
           sort0   identity columns (option sort=0): coalesced output, more planes per tile, mixed waves
           sorted  key-ordered columns AND key-ordered output (the shard path with one shard: dsh_dist_shard_device):
                   exactly the work of sort1 with coalesced stores -- the difference to sort1 is the scatter
Estimators: 2 = ERTL_MLE (default), 0 = ORIGINAL.  Phases through the profiling option finalize_stop (1 prologue,
2 +histogram, 3 +list walk, 4 +estimator, 0 everything).
PMC: separate rocprofv3 --pmc passes per counter group (never combined with a trace), k_finalize rows only.
"""
import argparse
import csv
import glob
import json
import os
import re
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

WORK = {"C3": (10000, 14), "C4": (100000, 10)}
PMC = {
    "write": ["WRITE_SIZE"],
    "fetch": ["FETCH_SIZE"],
    "sq1": ["SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_ACTIVE_INST_VALU",
            "SQ_LDS_BANK_CONFLICT", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES"],
}


def setup(wl):
    import torch

    import dashing_amd
    from dashing_amd import synth

    n, p = WORK[wl]
    regs = torch.from_numpy(synth.survey_sketches(n, p, seed=0x5EED0000)[0]).cuda()
    out = torch.empty(n * (n - 1) // 2, dtype=torch.float32, device="cuda")
    ctx = dashing_amd.Context(0)
    return torch, ctx, regs, out, n, p


def one_pass(ctx, regs, out, n, p, layout, estim):
    ctx.attach_device(regs.data_ptr(), n, p)
    if layout == "sorted":
        ctx.dist_shard_device(out.data_ptr(), 0, 1, estim)
    else:
        ctx.dist_rows_device(out.data_ptr(), 0, n, estim)
    ctx.synchronize()


def run_one(wl, layout, estim):
    torch, ctx, regs, out, n, p = setup(wl)
    ctx.set_option("sort", 0 if layout == "sort0" else -1)
    one_pass(ctx, regs, out, n, p, layout, estim)
    one_pass(ctx, regs, out, n, p, layout, estim)
    ctx.close()


def times(wl, reps):
    torch, ctx, regs, out, n, p = setup(wl)
    ctx.set_profiling(True)
    rows = []
    for layout in ("sort1", "sorted", "sort0"):
        ctx.set_option("sort", 0 if layout == "sort0" else -1)
        for estim in (2, 0):
            for stop in (0, 1, 2, 3, 4):
                if stop and layout == "sorted":
                    continue
                ctx.set_option("finalize_stop", stop)
                best = None
                for _ in range(reps):
                    one_pass(ctx, regs, out, n, p, layout, estim)
                    k = ctx.last_kernel_ms()
                    if best is None or k["finalize_ms"] < best["finalize_ms"]:
                        best = k
                rows.append({"workload": wl, "n": n, "p": p, "layout": layout, "estim": estim, "finalize_stop": stop,
                             "finalize_ms": round(best["finalize_ms"], 3), "pair_ms": round(best["pair_ms"], 3),
                             "prepare_ms": round(best["prepare_ms"], 3),
                             "planes_per_tile": ctx.info("avg_tile_planes_x100") / 100.0})
                print(json.dumps(rows[-1]), flush=True)
    ctx.set_option("finalize_stop", 0)
    ctx.close()
    return rows


def short_name(k):
    return re.sub(r"\(.*$", "", k).replace("void ", "").strip()


def pmc(wl, layout, estim, outdir):
    res = {}
    for name, counters in PMC.items():
        d = os.path.join(outdir, "raw_%s_%s_%d_%s" % (wl, layout, estim, name))
        shutil.rmtree(d, ignore_errors=True)
        os.makedirs(d)
        argv = ["rocprofv3", "--pmc"] + counters + ["-d", d, "-o", "p", "--output-format", "csv", "--",
                                                    sys.executable, os.path.abspath(__file__), "--one", wl, layout, str(estim)]
        e = dict(os.environ)
        e["TMPDIR"] = "/tmp"
        with open(os.path.join(outdir, "pmc_%s_%s_%d_%s.log" % (wl, layout, estim, name)), "w") as log:
            rc = subprocess.call(argv, env=e, cwd="/tmp", stdout=log, stderr=subprocess.STDOUT)
        res["rc_" + name] = rc
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            acc = {}
            for row in csv.DictReader(open(f)):
                kn = short_name(row["Kernel_Name"])
                if "k_finalize" not in kn:
                    continue
                acc.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
                res["vgpr"] = int(row["VGPR_Count"])
                res["lds_block"] = int(row["LDS_Block_Size"])
            for c, v in acc.items():
                res[c] = v[-1]  # the measured (second) pass; per dispatch = per band
                res[c + "_sum_last_pass"] = sum(v[len(v) // 2:])
                res["dispatches_" + name] = len(v)
        shutil.rmtree(d, ignore_errors=True)
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tag", default="r3a")
    ap.add_argument("--one", nargs=3)
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--no-pmc", action="store_true")
    ap.add_argument("--workloads", default="C3,C4")
    args = ap.parse_args()
    if args.one:
        run_one(args.one[0], args.one[1], int(args.one[2]))
        return
    outdir = os.path.join(ROOT, "gpurun_out", args.tag)
    os.makedirs(outdir, exist_ok=True)
    with open(os.path.join(outdir, "finalize_times.jsonl"), "w") as f:
        for wl in args.workloads.split(","):
            for r in times(wl, args.reps):
                f.write(json.dumps(r) + "\n")
    if args.no_pmc:
        return
    with open(os.path.join(outdir, "finalize_pmc.jsonl"), "w") as f:
        for wl in args.workloads.split(","):
            n, p = WORK[wl]
            for layout in ("sort1", "sorted", "sort0"):
                r = pmc(wl, layout, 2, outdir)
                r.update({"workload": wl, "n": n, "p": p, "layout": layout, "estim": 2, "output_bytes": 4 * (n * (n - 1) // 2),
                          "note": "sums over the k_finalize dispatches (bands) of ONE full pass; FETCH_SIZE/WRITE_SIZE in KiB (gfx950: wide reads count half)"})
                f.write(json.dumps(r) + "\n")
                f.flush()
                print(json.dumps(r), flush=True)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Instructions of k_finalize per phase: SQ_INSTS_VALU / SALU / LDS per wave with the kernel cut after phase 1..4
(option finalize_stop) and whole, for the MLE and the ORIGINAL estimator -- the differences are what each phase issues.
(The TIME of the phases comes from tools/finalize_probe.py: stamps inside the full kernel.)

  python tools/finalize_instr.py --out gpurun_out/r4d/finalize_instr.jsonl [--workloads C4,C3]

Runs itself under `rocprofv3 --pmc` (one pass, no trace) and reads the k_finalize dispatches in order."""
import argparse
import csv
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
WORK = {"C3": (10000, 14), "C4": (100000, 10)}
RUNS = [(2, 1), (2, 2), (2, 3), (2, 4), (2, 0), (0, 4), (0, 0)]  # (estimator, finalize_stop)


def child(wls):
    import torch

    import dashing_amd
    from dashing_amd import synth

    ctx = dashing_amd.Context(0)
    ctx.set_profiling(True)
    for wl in wls:
        n, p = WORK[wl]
        regs = torch.from_numpy(synth.survey_sketches(n, p, seed=0x5EED0000)[0]).cuda()
        out = torch.empty(n * (n - 1) // 2, dtype=torch.float32, device="cuda")
        for estim, stop in RUNS:
            ctx.set_option("finalize_stop", stop)
            ctx.attach_device(regs.data_ptr(), n, p)
            ctx.dist_rows_device(out.data_ptr(), 0, n, estim)
            ctx.synchronize()
        ctx.set_option("finalize_stop", 0)
        del regs, out
        torch.cuda.empty_cache()
    ctx.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workloads", default="C4,C3")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "r4d", "finalize_instr.jsonl"))
    ap.add_argument("--child", action="store_true")
    args = ap.parse_args()
    wls = args.workloads.split(",")
    if args.child:
        child(wls)
        return
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    counters = ["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_WAVES", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR"]
    d = tempfile.mkdtemp(prefix="fin_instr_", dir="/tmp")
    argv = ["rocprofv3", "--pmc"] + counters + ["-d", d, "-o", "p", "--output-format", "csv", "--", sys.executable, os.path.abspath(__file__),
                                               "--child", "--workloads", args.workloads]
    rc = subprocess.call(argv, env=dict(os.environ, TMPDIR="/tmp"), cwd="/tmp", stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        rows += list(csv.DictReader(open(f)))
    shutil.rmtree(d, ignore_errors=True)
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    # every compare pass starts with ONE k_selfhist_card dispatch: split the dispatch stream there; the passes come in
    # the order workloads x RUNS; a pass may hold several k_finalize dispatches (bands): summed
    passes, cur, seen = [], None, set()
    for r in rows:
        if "k_selfhist_card" in r["Kernel_Name"] and r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"])
            cur = {}
            passes.append(cur)
        if cur is not None and "k_finalize" in r["Kernel_Name"]:
            cur[r["Counter_Name"]] = cur.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
            cur["_n"] = cur.get("_n", 0) + (1 if r["Counter_Name"] == "SQ_WAVES" else 0)
    with open(args.out, "a") as fo:
        if rc != 0 or len(passes) != len(wls) * len(RUNS):
            fo.write(json.dumps({"error": "rocprofv3 rc %d, %d passes (expected %d)" % (rc, len(passes), len(wls) * len(RUNS))}) + "\n")
            print("rocprofv3 rc", rc, len(passes))
            return
        for wi, wl in enumerate(wls):
            for ri, (estim, stop) in enumerate(RUNS):
                acc = passes[wi * len(RUNS) + ri]
                w = acc.get("SQ_WAVES", 0.0) or 1.0
                row = {"workload": wl, "estim": estim, "finalize_stop": stop, "dispatches": acc.get("_n"), "waves": acc.get("SQ_WAVES"),
                       "per_wave": {k.replace("SQ_INSTS_", "").lower(): round(v / w, 1) for k, v in acc.items() if k not in ("SQ_WAVES", "_n")}}
                fo.write(json.dumps(row) + "\n")
                print(json.dumps(row), flush=True)

if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Collect rocprofv3 evidence for one workload on the GPU box and write a summary that is stamped
with the hash of the kernel sources it was measured on.

  python tools/pmc_collect.py --tag r2a [--env DSH_BENCH_N=100000 --env DSH_BENCH_P=10] [--steps 3]
      [--passes kt,fetch,write,sq1,sq2] [--cmd "python bench.py --no-cpu-baseline"]

Passes (each its own process; --pmc is never combined with a trace domain other than --kernel-trace,
as MI355X_MICROARCH.md / the gpurun rules prescribe):
  kt     rocprofv3 --kernel-trace --stats            -> <out>/kernel_stats.csv
  fetch  --pmc FETCH_SIZE                              (x2 for wide coalesced reads on gfx950, see the guide)
  write  --pmc WRITE_SIZE
  sq1    --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES SQ_WAVE_CYCLES
  sq2    --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM GRBM_GUI_ACTIVE
Output: gpurun_out/<tag>/{kernel_stats.csv,pmc_summary.json} -- copy what should be judged into profiles/<tag>/.
pmc_summary.json: {"source_sha256", "workload", "kernels": {name: {counter: mean per dispatch, ..., "calls": n}}}
and, for the pair kernel, "hbm_bytes_per_launch" = 2*FETCH_SIZE + WRITE_SIZE (KiB -> bytes), which is what
bench.py reports as roofline.traffic -- bench.py refuses the file when source_sha256 differs from the
sources it is running (see source_hash()).
"""
import argparse
import csv
import glob
import hashlib
import json
import os
import re
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

PASSES = {
    "fetch": ["FETCH_SIZE"],
    "write": ["WRITE_SIZE"],
    "sq1": ["SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_ACTIVE_INST_VALU",
            "SQ_LDS_BANK_CONFLICT", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES"],
    "mfma": ["SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_LDS",
             "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES"],
    "sq2": ["SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_LDS", "SQ_INSTS_VMEM_RD",
            "SQ_INSTS_VMEM_WR", "SQ_INSTS_SMEM", "GRBM_GUI_ACTIVE"],
}


def source_hash():
    """sha256 over the device sources the timed kernels are built from (same function in bench.py)."""
    h = hashlib.sha256()
    for rel in ("dashing_amd/csrc/kernels_compare.hip", "dashing_amd/csrc/kernels_sketch.hip",
                "dashing_amd/csrc/estimators.h", "dashing_amd/csrc/kernels.h", "dashing_amd/csrc/consts.h", "dashing_amd/csrc/ctx.h",
                "dashing_amd/csrc/plan.h", "dashing_amd/csrc/plan.cpp", "dashing_amd/csrc/engine.hip", "dashing_amd/csrc/abi.hip",
                "dashing_amd/csrc/knn.hip", "dashing_amd/csrc/exchange.hip", "dashing_amd/csrc/kernels_fastx.hip"):
        with open(os.path.join(ROOT, rel), "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def short_name(k):
    k = re.sub(r"\(.*$", "", k)
    k = k.replace("void ", "")
    return k.strip()


def run_pass(name, counters, cmd, env, outdir):
    d = os.path.join(outdir, "raw_" + name)
    shutil.rmtree(d, ignore_errors=True)
    os.makedirs(d)
    if name == "kt":
        argv = ["rocprofv3", "--kernel-trace", "--stats", "-d", d, "-o", "p", "--output-format", "csv", "--"] + cmd
    else:
        argv = ["rocprofv3", "--pmc"] + counters + ["-d", d, "-o", "p", "--output-format", "csv", "--"] + cmd
    e = dict(os.environ)
    e.update(env)
    e["TMPDIR"] = "/tmp"
    log = open(os.path.join(outdir, name + ".log"), "w")
    rc = subprocess.call(argv, env=e, cwd="/tmp", stdout=log, stderr=subprocess.STDOUT)
    return rc, d


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tag", required=True)
    ap.add_argument("--env", action="append", default=[])
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--passes", default="kt,fetch,write,sq1,sq2")
    ap.add_argument("--cmd", default="python %s --no-cpu-baseline --no-secondary --no-pmc" % os.path.join(ROOT, "bench.py"))
    args = ap.parse_args()
    env = dict(kv.split("=", 1) for kv in args.env)
    outdir = os.path.join(ROOT, "gpurun_out", args.tag)
    os.makedirs(outdir, exist_ok=True)
    base = args.cmd.split()
    summary = {"source_sha256": source_hash(), "env": env, "cmd": args.cmd, "kernels": {}, "passes": {}}
    for name in args.passes.split(","):
        steps = args.steps if name == "kt" else 1
        cmd = base + ["--steps", str(steps), "--warmup", "1" if name == "kt" else "0"]
        rc, d = run_pass(name, PASSES.get(name), cmd, env, outdir)
        summary["passes"][name] = {"rc": rc}
        if name == "kt":
            for f in glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True):
                shutil.copy(f, os.path.join(outdir, "kernel_stats.csv"))
            continue
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            acc = {}
            for row in csv.DictReader(open(f)):
                key = (short_name(row["Kernel_Name"]), row["Counter_Name"])
                acc.setdefault(key, []).append(float(row["Counter_Value"]))
                kk = summary["kernels"].setdefault(short_name(row["Kernel_Name"]), {})
                kk["vgpr"] = int(row["VGPR_Count"])
                kk["sgpr"] = int(row["SGPR_Count"])
                kk["lds_block"] = int(row["LDS_Block_Size"])
            for (k, c), v in acc.items():
                kk = summary["kernels"][k]
                kk[c] = sum(v) / len(v)
                kk["calls_" + name] = len(v)
    for k, kk in summary["kernels"].items():
        if "FETCH_SIZE" in kk and "WRITE_SIZE" in kk:
            # both counters are in KiB; FETCH_SIZE counts wide coalesced reads at half their bytes on gfx950
            kk["hbm_bytes_per_launch"] = (2.0 * kk["FETCH_SIZE"] + kk["WRITE_SIZE"]) * 1024.0
    # the bench line of the kt pass (its JSON is the last stdout line)
    try:
        last = [l for l in open(os.path.join(outdir, "kt.log")) if l.startswith("{")][-1]
        summary["bench_line_of_kt_pass"] = json.loads(last)
    except (OSError, IndexError, ValueError):
        pass
    json.dump(summary, open(os.path.join(outdir, "pmc_summary.json"), "w"), indent=1)
    # the file bench.py reads for roofline.traffic (copy it to profiles/pmc_pair_kernel.json): pair kernel only,
    # stamped with the sources it was measured on
    for k, kk in summary["kernels"].items():
        if "k_pair_counts" in k and "hbm_bytes_per_launch" in kk:
            json.dump({"source_sha256": summary["source_sha256"], "kernel": k,
                       "workload": {"n_sketches": int(env.get("DSH_BENCH_N", "10000")), "p": int(env.get("DSH_BENCH_P", "14"))},
                       "fetch_size_kib_raw": kk["FETCH_SIZE"], "write_size_kib": kk["WRITE_SIZE"],
                       "hbm_bytes_per_launch": kk["hbm_bytes_per_launch"],
                       "how": "tools/pmc_collect.py: separate rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes; bytes = (2*FETCH_SIZE + WRITE_SIZE) KiB (gfx950 counts wide coalesced reads at half their bytes, MI355X_MICROARCH.md)"},
                      open(os.path.join(outdir, "pmc_pair_kernel.json"), "w"), indent=1)
    for d in glob.glob(os.path.join(outdir, "raw_*")):
        shutil.rmtree(d, ignore_errors=True)
    print(json.dumps({k: {c: v for c, v in kk.items()} for k, kk in summary["kernels"].items() if "k_" in k}, indent=1))


if __name__ == "__main__":
    sys.exit(main())

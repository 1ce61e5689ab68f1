#!/usr/bin/env python3
"""EXPERIMENT (profiling only): what would cheaper evaluations of the MLE's inner step buy?  Option mle_variant:
0 shipped (two Newton steps, no contraction), 1 one Newton step, 2 products contracted to fma, 3 both.
Reports k_finalize ms and how the float32 results differ from the shipped evaluation."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import dashing_amd  # noqa: E402
from dashing_amd import synth  # noqa: E402

WORK = {"C3": (10000, 14), "C4": (100000, 10)}
ctx = dashing_amd.Context(0)
for wl in os.environ.get("WL", "C3,C4").split(","):
    n, p = WORK[wl]
    regs = torch.from_numpy(synth.survey_sketches(n, p, seed=0x5EED0000)[0]).cuda()
    total = n * (n - 1) // 2
    ref = torch.empty(total, dtype=torch.float32, device="cuda")
    out = torch.empty(total, dtype=torch.float32, device="cuda")
    ctx.set_profiling(True)
    for v in (0, 1, 2, 3):
        ctx.set_option("mle_variant", v)
        best = 1e9
        for _ in range(3):
            ctx.attach_device(regs.data_ptr(), n, p)
            ctx.dist_rows_device((ref if v == 0 else out).data_ptr(), 0, n)
            ctx.synchronize()
            best = min(best, ctx.last_kernel_ms()["finalize_ms"])
        row = {"workload": wl, "mle_variant": v, "finalize_ms": round(best, 3)}
        if v:
            neq = int((out != ref).sum().item())
            d = (out.double() - ref.double()).abs() / ref.double().abs().clamp_min(1e-9)
            row.update({"float32_values_that_differ": neq, "of": total, "max_rel_diff": float(d.max().item())})
        print(json.dumps(row), flush=True)
    ctx.set_option("mle_variant", 0)
    ctx.set_profiling(False)
    del regs, ref, out
    torch.cuda.empty_cache()

#!/usr/bin/env python3
"""k_finalize time per estimator and measure on the C3 matrix (10 000 x p=14): separates the estimator's
share of the kernel from the histogram assembly and the list walk."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dashing_amd
from dashing_amd import synth
n, p = 10000, 14
regs = torch.from_numpy(synth.survey_sketches(n, p, seed=0x5EED0000)[0]).cuda()
out = torch.empty(n * (n - 1) // 2, dtype=torch.float32, device="cuda")
ctx = dashing_amd.Context(0)
ctx.set_profiling(True)
for estim in (0, 1, 2):
    for rt in (1, 0):
        best = None
        for _ in range(3):
            ctx.attach_device(regs.data_ptr(), n, p)
            ctx.dist_rows_device(out.data_ptr(), 0, n, estim, rt, 31)
            ctx.synchronize()
            k = ctx.last_kernel_ms()
            if best is None or k["finalize_ms"] < best["finalize_ms"]:
                best = k
        print(json.dumps({"estim": estim, "result_type": rt, "finalize_ms": round(best["finalize_ms"], 3), "pair_ms": round(best["pair_ms"], 3)}))

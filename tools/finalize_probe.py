#!/usr/bin/env python3
"""k_finalize experiments on a p=10 matrix: stop points x tiles per group."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import dashing_amd
from dashing_amd import synth
n, p = 40000, 10
regs = torch.from_numpy(synth.survey_sketches(n, p)[0]).cuda()
out = torch.empty(n * (n - 1) // 2, dtype=torch.float32, device="cuda")
ctx = dashing_amd.Context(0)
ctx.set_profiling(True)
for gt in (1, 4, 32, 128):
    ctx.set_option("fin_group_tiles", gt)
    res = {}
    for stop in (1, 5, 2, 3, 0):
        ctx.set_option("finalize_stop", stop)
        best = 1e9
        for _ in range(2):
            ctx.attach_device(regs.data_ptr(), n, p)
            ctx.dist_rows_device(out.data_ptr(), 0, n)
            ctx.synchronize()
            best = min(best, ctx.last_kernel_ms()["finalize_ms"])
        res["stop%d" % stop] = round(best, 2)
    print(json.dumps({"group_tiles": gt, "ms": res}))

#!/usr/bin/env python3
"""Where k_finalize spends its time: runs the kernel with the profiling option "finalize_stop" = 1..4 (leave
after the block prologue / the histogram assembly / the list walk / the estimator) and 0 (whole kernel) on the
bench workload (C3) and on a p=10 matrix, HIP-event times per variant."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import dashing_amd  # noqa: E402
from dashing_amd import synth  # noqa: E402

for n, p in ((10000, 14), (40000, 10)):
    regs = torch.from_numpy(synth.survey_sketches(n, p)[0]).cuda()
    out = torch.empty(n * (n - 1) // 2, dtype=torch.float32, device="cuda")
    ctx = dashing_amd.Context(0)
    ctx.set_profiling(True)
    res = {}
    for estim in (2, 0):
        for stop in (1, 2, 3, 4, 0):
            ctx.set_option("finalize_stop", stop)
            best = 1e9
            for _ in range(3):
                ctx.attach_device(regs.data_ptr(), n, p)
                ctx.dist_rows_device(out.data_ptr(), 0, n, estim)
                ctx.synchronize()
                best = min(best, ctx.last_kernel_ms()["finalize_ms"])
            res["estim%d_stop%d" % (estim, stop)] = round(best, 3)
    ctx.set_option("finalize_stop", 0)
    print(json.dumps({"n": n, "p": p, "finalize_ms_by_stop(1 prologue,2 +histogram,3 +list walk,4 +estimator,0 all)": res}))
    ctx.close()

#!/usr/bin/env python3
"""configs[3] shape (100 000 x p=10) on one GPU: kernel times with the phase-locked vs the free-running tile kernel and with 2 / 8 GiB
of C(v) scratch (number of bands)."""
import sys, json
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, dashing_amd
from dashing_amd import synth
n, p = 100000, 10
regs = torch.from_numpy(synth.survey_sketches(n, p, seed=0x5EED0000)[0]).cuda()
out = torch.empty(n*(n-1)//2, dtype=torch.float32, device="cuda")
ctx = dashing_amd.Context(0); ctx.set_profiling(True)
for ls in (1, 0):
    for budget in (2 << 30, 8 << 30):
        ctx.set_option("pair_lockstep", ls); ctx.set_option("cum_budget_bytes", budget)
        best = None
        for _ in range(2):
            ctx.attach_device(regs.data_ptr(), n, p); ctx.dist_rows_device(out.data_ptr(), 0, n); ctx.synchronize()
            k = ctx.last_kernel_ms()
            if best is None or k["pair_ms"] < best["pair_ms"]: best = k
        print(json.dumps({"lockstep": ls, "cum_budget_GiB": budget >> 30, **{a: round(b, 2) for a, b in best.items()}, "planes": ctx.info("avg_tile_planes_x100")/100}))

#!/usr/bin/env python3
"""Phase-locked tile kernel (k_pair_counts_ls) vs the free-running one (k_pair_counts) per precision: pair-kernel time
with option pair_lockstep = 0 / 1 and kc = 16 / 32 where allowed."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import dashing_amd  # noqa: E402
from dashing_amd import synth  # noqa: E402

for n, p in ((40000, 10), (30000, 11), (20000, 12), (14000, 13), (10000, 14), (5000, 16)):
    regs = torch.from_numpy(synth.survey_sketches(n, p, seed=0x5EED0000)[0]).cuda()
    out = torch.empty(n * (n - 1) // 2, dtype=torch.float32, device="cuda")
    ctx = dashing_amd.Context(0)
    ctx.set_profiling(True)
    row = {"n": n, "p": p}
    for ls in (0, 1):
        for kc in (16, 32):
            if ls and (1 << p) // 32 < kc:
                continue
            ctx.set_option("pair_lockstep", ls)
            ctx.set_option("kc", kc)
            best = 1e9
            for _ in range(3):
                ctx.attach_device(regs.data_ptr(), n, p)
                ctx.dist_rows_device(out.data_ptr(), 0, n)
                ctx.synchronize()
                best = min(best, ctx.last_kernel_ms()["pair_ms"])
            row["lockstep%d_kc%d" % (ls, kc)] = round(best, 3)
    print(json.dumps(row), flush=True)
    ctx.close()

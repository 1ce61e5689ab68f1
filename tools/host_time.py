#!/usr/bin/env python3
"""Host-side share of a dist call (dsh_get_info host_keys_wait_us / host_layout_us / host_lists_us) for the full C3 triangle and
for the first and last of 8 ranks' row ranges: what sits between the per-sketch pass and the tile kernel (profiles/r3f)."""
import sys, time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, dashing_amd
from dashing_amd import synth
n, p = 10000, 14
regs = torch.from_numpy(synth.survey_sketches(n, p)[0]).cuda()
out = torch.empty(n*(n-1)//2, dtype=torch.float32, device="cuda")
ctx = dashing_amd.Context(0)
for (rb, re) in ((0, n), (0, 640), (6400, n)):
    best = 1e9
    for _ in range(5):
        ctx.attach_device(regs.data_ptr(), n, p); torch.cuda.synchronize()
        t0 = time.perf_counter(); ctx.dist_rows_device(out.data_ptr(), rb, re); ctx.synchronize(); best = min(best, time.perf_counter() - t0)
    print(rb, re, "ms %.3f" % (best*1e3), {k: ctx.info(k) for k in ("host_keys_wait_us", "host_layout_us", "host_lists_us")})

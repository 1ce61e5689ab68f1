#!/usr/bin/env python3
"""One virtual rank's step of an N-rank exchange plan, a few times over, for a kernel trace:
  cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/rt -o t -- python tools/rank_trace.py
  G=8 RANK=1 NPARTS=8 N=10000 P=14 STEPS=6 [OPTS=k=v,...]
Prints nothing but one JSON line (the wall of the last step); the trace's kernel start / end times show where a rank's
step spends the time BETWEEN its kernels (tools/trace_gaps.py)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import dashing_amd  # noqa: E402
from dashing_amd import synth  # noqa: E402

n, p = int(os.environ.get("N", "10000")), int(os.environ.get("P", "14"))
G, r, nparts = int(os.environ.get("G", "8")), int(os.environ.get("RANK", "1")), int(os.environ.get("NPARTS", "8"))
regs = torch.from_numpy(synth.survey_sketches(n, p, seed=0x5EED0000)[0]).cuda()
ctx = dashing_amd.Context(0)
for kv in filter(None, os.environ.get("OPTS", "").split(",")):
    k_, v_ = kv.split("=")
    ctx.set_option(k_, int(v_))
rows = dashing_amd.balance_rowsets(n, G, -1, 0, -1)
floats = dashing_amd.exchange_mode(n, rows, r, nparts, 0, want_floats=True)[2]
local = torch.empty(max(floats, 1), dtype=torch.float32, device="cuda")
wall = 0.0
for _ in range(int(os.environ.get("STEPS", "6"))):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ctx.attach_device(regs.data_ptr(), n, p)
    ctx.exchange_rows_device_async(local.data_ptr(), rows, r, nparts, 0)
    ctx.synchronize()
    wall = time.perf_counter() - t0
print(json.dumps({"G": G, "rank": r, "wall_ms": round(wall * 1e3, 3), "items": ctx.info("items"), "bands": ctx.info("bands")}))

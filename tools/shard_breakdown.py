#!/usr/bin/env python3
"""Where does a rank's time go in a G-way run?  Every rank's share computed one after the other on ONE GPU:
wall time of the step (no profiling), then the same step with HIP-event phase times and the host-side times.
  N=10000 P=14 GS=8 python tools/shard_breakdown.py"""
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
regs = torch.from_numpy(synth.survey_sketches(n, p)[0]).cuda()
ctx = dashing_amd.Context(0)
for kv in filter(None, os.environ.get("OPTS", "").split(",")):
    k_, v_ = kv.split("=")
    ctx.set_option(k_, int(v_))
NPARTS = int(os.environ.get("NPARTS", "0"))
for G in tuple(int(x) for x in os.environ.get("GS", "1,8").split(",")):
    b = dashing_amd.balance_rows(n, G)
    mx = max(dashing_amd.tri_span(n, b[r], b[r + 1]) for r in range(G))
    out = torch.empty(mx, dtype=torch.float32, device="cuda")
    rows = []
    for r in range(G):
        def step():
            ctx.attach_device(regs.data_ptr(), n, p)
            if NPARTS:
                ctx.dist_rows_parts_device_async(out.data_ptr(), b[r], b[r + 1], NPARTS)
            else:
                ctx.dist_rows_device(out.data_ptr(), b[r], b[r + 1])
            ctx.synchronize()
        best = 1e9
        for _ in range(5):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            step()
            best = min(best, time.perf_counter() - t0)
        ctx.set_profiling(True)
        step()
        step()
        k = ctx.last_kernel_ms()
        ctx.set_profiling(False)
        nt = (n + 127) // 128
        rows.append({"rank": r, "rows": [b[r], b[r + 1]], "tiles": ctx.info("tiles"), "wall_ms": round(best * 1e3, 3),
                     "prepare_ms": round(k["prepare_ms"], 3), "pair_ms": round(k["pair_ms"], 3), "finalize_ms": round(k["finalize_ms"], 3),
                     "sum_events_ms": round(k["prepare_ms"] + k["pair_ms"] + k["finalize_ms"], 3),
                     "host_keys_wait_us": ctx.info("host_keys_wait_us"), "host_layout_us": ctx.info("host_layout_us"), "host_lists_us": ctx.info("host_lists_us"),
                     "planes_per_tile": ctx.info("avg_tile_planes_x100") / 100})
    print(json.dumps({"G": G, "n": n, "p": p, "nparts": NPARTS, "max_wall_ms": max(x["wall_ms"] for x in rows), "mean_wall_ms": round(sum(x["wall_ms"] for x in rows) / G, 3), "ranks": rows}))

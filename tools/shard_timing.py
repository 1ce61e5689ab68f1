#!/usr/bin/env python3
"""Times every rank's share of a G-way run sequentially on ONE GPU (what each rank of a G-GPU run would
spend in compute, incl. its own prepare): shows the balance and the fixed per-rank overhead.
  range_ms  the scheme bench.py uses: pair-balanced row ranges of the final triangle (plane matrix laid out per range)
  shard_ms  the older scheme: cost-balanced shards of the sorted-order triangle (+ gather + un-permute on rank 0)"""
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
if n <= 100000:
    regs = torch.from_numpy(synth.survey_sketches(n, p)[0]).cuda()
else:  # sketch g = max(base[a_g], base[b_g]): unions of two base sets, built on the device (as tests/test_gpu_configs.py)
    nbase = 4000
    bd = torch.from_numpy(synth.survey_sketches(nbase, p)[0]).cuda()
    regs = torch.empty((n, 1 << p), dtype=torch.uint8, device="cuda")
    regs[:nbase] = bd
    g = torch.arange(nbase, n, device="cuda", dtype=torch.int64)
    a, b2 = g % nbase, (g * 2654435761 + 12345) % nbase
    for s0 in range(0, n - nbase, 1 << 14):
        e0 = min(n - nbase, s0 + (1 << 14))
        regs[nbase + s0 : nbase + e0] = torch.maximum(bd[a[s0:e0]], bd[b2[s0:e0]])
    torch.cuda.synchronize()
ctx = dashing_amd.Context(0)
if os.environ.get("C0"):
    ctx.set_option("shard_c0_x10", int(os.environ["C0"]))
GS = tuple(int(x) for x in os.environ.get("GS", "1,2,4,8").split(","))
REPS = int(os.environ.get("REPS", "3"))
for G in GS:
    b = dashing_amd.balance_rows(n, G)
    mx = max(dashing_amd.tri_span(n, b[r], b[r + 1]) for r in range(G))
    out = torch.empty(mx, dtype=torch.float32, device="cuda")
    rows = []
    for r in range(G):
        best = 1e9
        for _ in range(REPS):
            ctx.attach_device(regs.data_ptr(), n, p)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ctx.dist_rows_device(out.data_ptr(), b[r], b[r + 1])
            ctx.synchronize()
            best = min(best, time.perf_counter() - t0)
        rows.append(round(best * 1e3, 3))
    print(json.dumps({"G": G, "range_ms": rows, "max_ms": max(rows), "row_bounds": b, "planes_per_tile_last_rank": ctx.info("avg_tile_planes_x100") / 100}))
# the same ranges computed in parts (what the pipelined exchange needs): cost of the extra launches / tails
for nparts in (() if os.environ.get("NO_PARTS") else (2, 4, 8)):
    G = 8
    b = dashing_amd.balance_rows(n, G)
    mx = max(dashing_amd.tri_span(n, b[r], b[r + 1]) for r in range(G))
    out = torch.empty(mx, dtype=torch.float32, device="cuda")
    rows = []
    for r in range(G):
        best = 1e9
        for _ in range(3):
            ctx.attach_device(regs.data_ptr(), n, p)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ctx.dist_rows_parts_device_async(out.data_ptr(), b[r], b[r + 1], nparts)
            ctx.synchronize()
            best = min(best, time.perf_counter() - t0)
        rows.append(round(best * 1e3, 3))
    # exchange model (NOT measured: one GPU here): rank 0 receives every other rank's span over that rank's own xGMI link,
    # all links at once, at 45 GB/s per link (about what RCCL point-to-point reaches on one link) + 20 us per round of
    # grouped send/recv.  Part q of a rank can leave when its last k_finalize segment is done (~ (q+1)/nparts of the rank's
    # time when the tile kernel is cut per part -- large parts -- else after the whole tile kernel); a link carries one
    # part at a time: done_q = max(ready_q, done_{q-1}) + bytes_q / link.  step = max over ranks of done_last.
    LINK = 45e9
    step, worst = max(rows), 0
    for r in range(1, G):
        pr = dashing_amd.range_parts(n, b[r], b[r + 1], nparts)
        done = 0.0
        for q in range(len(pr) - 1):
            by = dashing_amd.tri_span(n, pr[q], pr[q + 1]) * 4
            ready = rows[r] * (q + 1) / (len(pr) - 1)
            done = max(ready, done) + by / LINK * 1e3 + 0.02
        if done > step:
            step, worst = done, r
    print(json.dumps({"G": G, "nparts": nparts, "range_ms": rows, "max_ms": max(rows), "parts_rank0": dashing_amd.range_parts(n, b[0], b[1], nparts),
                      "exchange_model": {"assumed_link_GBs": 45, "span_bytes_per_link_max": max(dashing_amd.tri_span(n, b[r], b[r + 1]) for r in range(1, G)) * 4,
                                         "step_model_ms": round(step, 3), "bound_by": "link of rank %d" % worst if worst else "compute"}}))
for G in (() if os.environ.get("NO_SHARDS") else (1, 2, 4, 8)):
    ctx.attach_device(regs.data_ptr(), n, p)
    off = ctx.shard_plan(G)
    mx = max(off[r + 1] - off[r] for r in range(G))
    out = torch.empty(mx, dtype=torch.float32, device="cuda")
    rows = []
    for r in range(G):
        best = 1e9
        for _ in range(3):
            ctx.attach_device(regs.data_ptr(), n, p)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ctx.dist_shard_device(out.data_ptr(), r, G)
            ctx.synchronize()
            best = min(best, time.perf_counter() - t0)
        rows.append(round(best * 1e3, 3))
    print(json.dumps({"G": G, "shard_ms": rows, "max_ms": max(rows), "pairs_share": [round((off[r + 1] - off[r]) / off[-1], 3) for r in range(G)]}))
if os.environ.get("NO_SHARDS"):
    sys.exit(0)
# unpermute cost
full = torch.empty(n * (n - 1) // 2, dtype=torch.float32, device="cuda")
fin = torch.empty_like(full)
ctx.attach_device(regs.data_ptr(), n, p)
ctx.shard_plan(8)
res = {}
for name, mode in (("gather", 1), ("scatter", 0)):
    ctx.set_option("unpermute_gather", mode)
    ctx.unpermute_device(full.data_ptr(), fin.data_ptr())
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        ctx.unpermute_device(full.data_ptr(), fin.data_ptr())
    ctx.synchronize()
    res["unpermute_%s_ms" % name] = (time.perf_counter() - t0) * 100
print(json.dumps(res))

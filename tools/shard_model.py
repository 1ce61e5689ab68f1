#!/usr/bin/env python3
"""The N-rank step modelled from ONE GPU: every rank's compute timed one after the other with the exchange-aware call
(dsh_exchange_rows_device_async: the buffer layout the real run uses), the destination's placement of the received
row-sorted parts timed on the same GPU, and a pipeline model of the transfers (NOT measured: one GPU here).

  N=10000 P=14 G=8 NPARTS=8 python tools/shard_model.py

Exchange model: every source's parts go to rank 0 over that source's own xGMI link at LINK_GBS (default 45, about what
RCCL point-to-point reaches on one link) + 20 us per round; part q of a rank is ready after its prepare, its tile kernel
and (q+1)/parts of its finalize (small parts share one launch of the tile kernel); a link carries one part at a time;
rank 0 places a row-sorted part behind its arrival (measured copy rate)."""
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
G, NPARTS = int(os.environ.get("G", "8")), int(os.environ.get("NPARTS", "8"))
LINK = float(os.environ.get("LINK_GBS", "45")) * 1e9
regs = torch.from_numpy(synth.survey_sketches(n, p)[0]).cuda()
ctx = dashing_amd.Context(0)
for kv in filter(None, os.environ.get("OPTS", "").split(",")):
    k_, v_ = kv.split("=")
    ctx.set_option(k_, int(v_))
total = n * (n - 1) // 2
final = torch.empty(total, dtype=torch.float32, device="cuda")
want = torch.empty(total, dtype=torch.float32, device="cuda")


def single():
    ctx.attach_device(regs.data_ptr(), n, p)
    ctx.dist_rows_device(want.data_ptr(), 0, n)
    ctx.synchronize()


single()
t1 = 1e9
for _ in range(5):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    single()
    t1 = min(t1, time.perf_counter() - t0)
b = dashing_amd.balance_rows(n, G)
rows = []
locals_ = {}
for r in range(G):
    span = dashing_amd.tri_span(n, b[r], b[r + 1])
    rs, k = dashing_amd.exchange_mode(n, b, r, NPARTS, 0)
    local = final if r == 0 else torch.empty(max(span, 1), dtype=torch.float32, device="cuda")
    locals_[r] = local

    def step():
        ctx.attach_device(regs.data_ptr(), n, p)
        ctx.exchange_rows_device_async(local.data_ptr(), b, r, NPARTS, 0)
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
    km = ctx.last_kernel_ms()
    ctx.set_profiling(False)
    rows.append({"rank": r, "rows": [b[r], b[r + 1]], "rowsorted": rs, "parts": k, "span_bytes": 4 * span, "tiles": ctx.info("tiles"), "bands": ctx.info("bands"),
                 "wall_ms": round(best * 1e3, 3), "prepare_ms": round(km["prepare_ms"], 3), "pair_ms": round(km["pair_ms"], 3),
                 "finalize_ms": round(km["finalize_ms"], 3), "planes_per_tile": ctx.info("avg_tile_planes_x100") / 100})
# rank 0 places what it received: all sources, timed together (its per-sketch pass covers every sketch: redo rank 0's step)
ctx.attach_device(regs.data_ptr(), n, p)
ctx.exchange_rows_device_async(final.data_ptr(), b, 0, NPARTS, 0)
ctx.synchronize()
place = 1e9
for _ in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for r in range(1, G):
        ctx.exchange_place_device(b, r, NPARTS, locals_[r].data_ptr(), final.data_ptr(), 0)
    place = min(place, time.perf_counter() - t0)
same = bool(torch.equal(final, want))
# pipeline model
step_ms, worst = max(x["wall_ms"] for x in rows), 0
place_rate = sum(x["span_bytes"] for x in rows[1:]) / max(place, 1e-9)  # bytes per second of the row placement (incl. host tables)
for x in rows[1:]:
    k = max(x["parts"], 1)
    done = 0.0
    for q in range(k):
        by = x["span_bytes"] / k
        if x["bands"] >= k > 1:  # the tile kernel is cut per part: part q is ready after (q+1)/k of tile kernel + finalize
            ready = x["prepare_ms"] + (x["pair_ms"] + x["finalize_ms"]) * (q + 1) / k
        else:
            ready = x["prepare_ms"] + x["pair_ms"] + x["finalize_ms"] * (q + 1) / k
        if q == k - 1:
            ready = max(ready, x["wall_ms"])
        done = max(ready, done) + by / LINK * 1e3 + 0.02
    if x["rowsorted"]:
        done += x["span_bytes"] / k / place_rate * 1e3  # the last part's rows put into place
    if done > step_ms:
        step_ms, worst = done, x["rank"]
print(json.dumps({"n": n, "p": p, "G": G, "nparts": NPARTS, "opts": os.environ.get("OPTS", ""), "single_gpu_ms": round(t1 * 1e3, 3), "ranks": rows,
                  "max_rank_wall_ms": max(x["wall_ms"] for x in rows), "mean_rank_wall_ms": round(sum(x["wall_ms"] for x in rows) / G, 3),
                  "dst_place_all_sources_ms": round(place * 1e3, 3), "assembled_equals_single_gpu": same,
                  "exchange_model": {"assumed_link_GBs": LINK / 1e9, "step_model_ms": round(step_ms, 3),
                                     "bound_by": "link/placement of rank %d" % worst if worst else "compute",
                                     "speedup_vs_single_gpu": round(t1 * 1e3 / step_ms, 2)}}))

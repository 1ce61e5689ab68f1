#!/usr/bin/env python3
"""Copy-out overlapped with compute through the asynchronous C-ABI on a configs[3]-shaped job:
100 000 sketches, p=10, the 20 GB packed matrix delivered to (page-locked) host memory in row blocks of <= 256 Mi
values, the way `dashing-amd dist -b` consumes it.
  compute_only   every block computed into a device buffer, nothing copied
  serialized     dsh_dist_rows_async + dsh_wait per block (kernels(b) -> copy(b) -> kernels(b+1) ...)
  overlapped     block b+1 enqueued before block b is awaited (dsh_event_record / dsh_event_wait): the copy of block b
                 runs on the copy stream while the kernels of block b+1 fill the other device buffer
  d2h_only       the same bytes copied device -> pinned host with nothing else running (torch, for the PCIe rate)"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import dashing_amd  # noqa: E402
from dashing_amd import synth  # noqa: E402

n, p = int(os.environ.get("N", "100000")), int(os.environ.get("P", "10"))
regs = torch.from_numpy(synth.survey_sketches(n, p, seed=0x5EED0000)[0]).cuda()
ctx = dashing_amd.Context(0)
ctx.attach_device(regs.data_ptr(), n, p)
block_vals = 256 << 20
cuts = [0]
while cuts[-1] < n:
    rb = cuts[-1]
    re = rb + 1
    while re < n and dashing_amd.tri_span(n, rb, re + 1) <= block_vals:
        re += 1
    cuts.append(re)
spans = [dashing_amd.tri_span(n, cuts[i], cuts[i + 1]) for i in range(len(cuts) - 1)]
nb = len(spans)
pins = [dashing_amd.PinnedArray(max(spans)) for _ in range(2)]
dbuf = torch.empty(max(spans), dtype=torch.float32, device="cuda")
torch.cuda.synchronize()
res = {"n": n, "p": p, "blocks": nb, "bytes": 4 * sum(spans)}


def timed(f):
    f()  # warm
    ctx.wait()
    t0 = time.perf_counter()
    f()
    ctx.wait()
    return time.perf_counter() - t0


def compute_only():
    for b in range(nb):
        ctx.dist_rows_device_async(dbuf.data_ptr(), cuts[b], cuts[b + 1])


def serialized():
    for b in range(nb):
        ctx.dist_rows_async(pins[b & 1].array, cuts[b], cuts[b + 1])
        ctx.wait()


arrivals = []


def overlapped():
    t = {}
    del arrivals[:]
    t00 = time.perf_counter()
    ctx.dist_rows_async(pins[0].array, cuts[0], cuts[1])
    t[0] = ctx.event_record()
    for b in range(nb):
        if b + 1 < nb:
            ctx.dist_rows_async(pins[(b + 1) & 1].array, cuts[b + 1], cuts[b + 2])
            t[b + 1] = ctx.event_record()
        te = time.perf_counter()
        ctx.event_wait(t[b])
        arrivals.append((round((te - t00) * 1e3, 1), round((time.perf_counter() - t00) * 1e3, 1)))  # (enqueued b+1 at, block b arrived at)
        # (the consumer would emit pins[b & 1] here)


res["compute_only_s"] = round(timed(compute_only), 4)
res["serialized_s"] = round(timed(serialized), 4)
res["overlapped_s"] = round(timed(overlapped), 4)
res["overlapped_ms_(next_block_enqueued, block_arrived)"] = list(arrivals)
# raw PCIe rate with page-locked memory
ph = torch.empty(max(spans), dtype=torch.float32).pin_memory()
torch.cuda.synchronize()
t0 = time.perf_counter()
for b in range(nb):
    ph[: spans[b]].copy_(dbuf[: spans[b]], non_blocking=True)
torch.cuda.synchronize()
res["d2h_only_s"] = round(time.perf_counter() - t0, 4)
res["d2h_gbs"] = round(res["bytes"] / res["d2h_only_s"] / 1e9, 1)
res["overlapped_over_max_compute_d2h"] = round(res["overlapped_s"] / max(res["compute_only_s"], res["d2h_only_s"]), 3)
print(json.dumps(res))

#!/usr/bin/env python3
"""One workload, several option sets: ms per step (best of REPS) and the kernels' own times.

  N=10000 P=14 SETS='ls_tail_frag=0;ls_tail_frag=4;ls_tail_frag=8' python tools/step_options.py"""
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
reps = int(os.environ.get("REPS", "10"))
regs = torch.from_numpy(synth.survey_sketches(n, p)[0]).cuda()
out = torch.empty(n * (n - 1) // 2, dtype=torch.float32, device="cuda")
ref = None
for opts in os.environ.get("SETS", "").split(";"):
    ctx = dashing_amd.Context(0)
    for kv in filter(None, opts.split(",")):
        k_, v_ = kv.split("=")
        ctx.set_option(k_, int(v_))

    def step():
        ctx.attach_device(regs.data_ptr(), n, p)
        ctx.dist_rows_device(out.data_ptr(), 0, n)
        ctx.synchronize()

    step()
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        step()
        best = min(best, time.perf_counter() - t0)
    ctx.set_profiling(True)
    step()
    step()
    km = ctx.last_kernel_ms()
    ctx.set_profiling(False)
    if ref is None:
        ref = out.clone()
    print(json.dumps({"n": n, "p": p, "options": opts, "step_ms": round(best * 1e3, 3), "prepare_ms": round(km["prepare_ms"], 3),
                      "pair_ms": round(km["pair_ms"], 3), "finalize_ms": round(km["finalize_ms"], 3), "bands": ctx.info("bands"), "host_us": {k: ctx.info("host_" + k + "_us") for k in ("keys_wait", "layout", "lists")},
                      "same_as_first": bool(torch.equal(out, ref))}), flush=True)
    ctx.close()

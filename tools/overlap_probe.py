#!/usr/bin/env python3
"""Does k_finalize hide under the tile kernel?  Two contexts (own streams) run the same job on ONE GPU from two host
threads; the tile kernel's workgroups cannot share a CU (128 KiB of LDS each), k_finalize's can sit beside one.  If two
jobs together take clearly less than twice one job, a band pipeline inside one job (tile kernel of band b+1 beside the
k_finalize of band b, C(v) scratch double-buffered) is worth building; if not, it is not.

  N=10000 P=14 STEPS=10 python tools/overlap_probe.py"""
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import dashing_amd  # noqa: E402
from dashing_amd import synth  # noqa: E402

n, p = int(os.environ.get("N", "10000")), int(os.environ.get("P", "14"))
K = int(os.environ.get("STEPS", "10"))
regs = torch.from_numpy(synth.survey_sketches(n, p)[0]).cuda()
total = n * (n - 1) // 2
ctxs = [dashing_amd.Context(0) for _ in range(2)]
outs = [torch.empty(total, dtype=torch.float32, device="cuda") for _ in range(2)]
torch.cuda.synchronize()


def job(i, steps, estim=dashing_amd.ESTIM_ERTL_MLE):
    c = ctxs[i]
    for _ in range(steps):
        c.attach_device(regs.data_ptr(), n, p)
        c.dist_rows_device(outs[i].data_ptr(), 0, n, estim=estim)
        c.synchronize()


def timed(fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn()
    torch.cuda.synchronize()
    return time.perf_counter() - t0


def both(steps, estims):
    th = [threading.Thread(target=job, args=(i, steps, estims[i])) for i in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()


res = {"n": n, "p": p, "steps": K}
for name, estims in (("mle", (2, 2)), ("original", (0, 0))):
    job(0, 2, estims[0])
    job(1, 2, estims[1])
    one = min(timed(lambda: job(0, K, estims[0])) for _ in range(3)) / K
    two = min(timed(lambda: both(K, estims)) for _ in range(3)) / K
    res[name] = {"one_job_ms": round(one * 1e3, 3), "two_jobs_side_by_side_ms": round(two * 1e3, 3),
                 "per_job_ms_when_paired": round(two * 1e3 / 2, 3), "gain": round(1 - two / (2 * one), 4)}
res["same"] = bool(torch.equal(outs[0], outs[1]))
print(json.dumps(res))

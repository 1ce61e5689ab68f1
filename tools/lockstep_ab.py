#!/usr/bin/env python3
"""A/B of the two integer-VALU tile kernels: k_pair_counts (256-thread workgroups, free-running) vs k_pair_counts_ls
(512-thread workgroups, AND and BCNT batches phase-locked across the waves of a SIMD), per precision and item size.
Outputs must be byte-identical."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import dashing_amd
from dashing_amd import synth
cases = [(10000, 14), (40000, 10), (20000, 12), (3000, 16), (2000, 18)]
variants = [("free", {"pair_lockstep": 0}), ("lockstep", {"pair_lockstep": 1})]
for extra in os.environ.get("AB_EXTRA", "").split(";"):
    if extra:
        variants.append((extra, dict([("pair_lockstep", 1)] + [(kv.split("=")[0], int(kv.split("=")[1])) for kv in extra.split(",")])))
for n, p in cases:
    regs = torch.from_numpy(synth.survey_sketches(n, p)[0]).cuda()
    total = n * (n - 1) // 2
    ref = torch.empty(total, dtype=torch.float32, device="cuda")
    out = torch.empty(total, dtype=torch.float32, device="cuda")
    ctx = dashing_amd.Context(0)
    ctx.set_profiling(True)
    row = {"n": n, "p": p}
    for vi, (name, opts) in enumerate(variants):
        for k_, v_ in (("nsplit", 0), ("kc", 16), ("ls_item_chunks", 16), ("ls_sort_items", 1)):
            ctx.set_option(k_, v_)
        for k_, v_ in opts.items():
            ctx.set_option(k_, v_)
        best = None
        dst = ref if vi == 0 else out
        for _ in range(3):
            ctx.attach_device(regs.data_ptr(), n, p)
            ctx.dist_rows_device(dst.data_ptr(), 0, n)
            ctx.synchronize()
            k = ctx.last_kernel_ms()
            if best is None or k["pair_ms"] < best["pair_ms"]:
                best = k
        row[name] = round(best["pair_ms"], 3)
        if vi:
            row[name + "_identical"] = bool(torch.equal(ref, out))
    print(json.dumps(row))
    ctx.close()

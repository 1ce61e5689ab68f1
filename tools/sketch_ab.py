#!/usr/bin/env python3
"""(Needs the tree of commit 1383876, which carried BOTH instruction streams behind the option sketch_variant; the result is
profiles/rd5d/sketch_ab.jsonl, after which the old stream was removed.)
A/B of k_sketch's two instruction streams (option sketch_variant: 0 = the kernel of rounds 1-4, 1 = the trimmed one) on
the bench's configs[1]-shaped input (G x 5 Mbp resident in HBM, k=31, p=10): kernel ms by HIP events, alternating the
variants; the registers must be identical between the variants and equal to the CPU oracle's on sampled genomes.

  G=200 python tools/sketch_ab.py"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
import dashing_amd  # noqa: E402
from oracle import oracle_c  # noqa: E402


def main():
    G, L, p = int(os.environ.get("G", "200")), 5_000_000, int(os.environ.get("P", "10"))
    dev = torch.device("cuda", 0)
    ctx = dashing_amd.Context(0)
    seq = bench.device_genomes(torch, dev, G, L)
    offs = np.arange(G + 1, dtype=np.uint64) * np.uint64(L)
    ctx.alloc(G, p)
    res = {0: [], 1: []}
    regs = {}
    ctx.set_profiling(True)
    for rnd in range(4):
        for var in (0, 1):
            ctx.set_option("sketch_variant", var)
            ctx.clear()
            ctx.synchronize()
            ctx.sketch_batch_device(seq.data_ptr(), offs, 0, bench.K, True)
            if rnd:
                res[var].append(ctx.info("sketch_kernel_us") / 1e3)
            else:
                regs[var] = ctx.download(0, G)
    ctx.set_profiling(False)
    same = bool((regs[0] == regs[1]).all())
    sample = [0, 10, G - 1]
    hs = torch.cat([seq[g * L:(g + 1) * L] for g in sample]).cpu().numpy()
    want = oracle_c.sketch_batch(hs, np.arange(len(sample) + 1, dtype=np.uint64) * np.uint64(L), bench.K, p, True)
    exact = bool((regs[1][sample] == want).all())
    out = {"workload": "%d x %d bp in HBM, k=%d, p=%d" % (G, L, bench.K, p), "registers_identical_between_variants": same,
           "registers_equal_oracle_on_sample": exact}
    for var in (0, 1):
        ms = min(res[var])
        out["variant_%d" % var] = {"kernel_ms": round(ms, 4), "kernel_ms_runs": [round(x, 4) for x in res[var]], "bases_per_s": G * L / (ms * 1e-3)}
    out["speedup"] = round(out["variant_0"]["kernel_ms"] / out["variant_1"]["kernel_ms"], 4)
    print(json.dumps(out))


if __name__ == "__main__":
    main()

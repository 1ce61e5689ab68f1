#!/usr/bin/env python3
"""debug: which genomes of bench.py's configs[1] workload differ from the oracle, and by how many registers"""
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

dev = torch.device("cuda", 0)
ctx = dashing_amd.Context(0)
for G, L in (((200, 5_000_000),) if os.environ.get("G200") else ((12, 1_000_000), (40, 5_000_000))):
    seq, wall, kms, regs = bench.sketch_workload(ctx, torch, dev, G, L, 10, 1)
    host = seq[: G * L].cpu().numpy()
    want = oracle_c.sketch_batch(host, np.arange(G + 1, dtype=np.uint64) * np.uint64(L), 31, 10, True)
    bad = [(g, int((regs[g] != want[g]).sum()), int((regs[g] > want[g]).sum())) for g in range(G) if (regs[g] != want[g]).any()]
    print(json.dumps({"G": G, "L": L, "genomes_that_differ": bad[:20], "n_bad": len(bad)}), flush=True)
    # one genome at a time through the host-buffer entry point
    g = bad[0][0] if bad else 0
    ctx.clear()
    one = ctx.sketch_batch(host[g * L:(g + 1) * L], np.array([0, L], np.uint64), 0, 31, True)
    print(json.dumps({"single_genome_call_equals_oracle": bool((one[0] == want[g]).all()), "genome": g}), flush=True)
    # the same genomes with the decorations removed
    clean = host.copy()
    clean[clean == ord("N")] = ord("A")
    clean &= 0xDF
    sd = torch.from_numpy(np.concatenate([clean, np.full(256, ord("N"), np.uint8)])).to(dev)
    ctx.clear()
    ctx.sketch_batch_device(sd.data_ptr(), np.arange(G + 1, dtype=np.uint64) * np.uint64(L), 0, 31, True)
    r2 = ctx.download(0, G)
    w2 = oracle_c.sketch_batch(clean, np.arange(G + 1, dtype=np.uint64) * np.uint64(L), 31, 10, True)
    print(json.dumps({"undecorated_n_bad": int(sum((r2[g_] != w2[g_]).any() for g_ in range(G)))}), flush=True)

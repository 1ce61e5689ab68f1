#!/usr/bin/env python3
"""The N-rank step modelled from ONE GPU: every rank's compute timed one after the other with the exchange-aware call
(dsh_exchange_rows_device_async: the buffer layout the real run uses), the destination's placement of the received
row-sorted parts timed on the same GPU, and a pipeline model of the transfers (NOT measured: one GPU here).

  N=10000 P=14 G=8 NPARTS=8 python tools/shard_model.py
  PARTITIONS=contiguous,rowsets,rowsets:450 ...   which partitions to model: contiguous = dsh_balance_rows ranges (round 4),
                                                  rowsets[:prep_permille[:dst_bonus_permille]] = dsh_balance_rowsets (range + top-up
                                                  tile rows; rank 0 receives and takes the bonus)

Exchange model (dashing_amd.multigpu.pipeline_model, following dsh_exchange_collect_async): NPARTS rounds; round q carries
message q of every source -- the q-th NPARTS-th of its buffer, ready when every part (measured: dsh_last_part_info) up to the
one that holds its last value is final -- over that source's own xGMI link at LINK_GBS; a round starts when the previous
one has arrived, lasts as long as its largest message + 20 us; rank 0 puts the rows a round completed into place on a
stream of its own (measured kernel rate).  The link rate is an ASSUMPTION (45 GB/s = about what RCCL point-to-point reaches
on one link): the step is printed for 30 / 45 / 60 GB/s so that a measured rate (bench.py --gpus N:
multi_gpu.link_gbs_measured) can be placed."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import dashing_amd  # noqa: E402
from dashing_amd import synth  # noqa: E402
from dashing_amd.multigpu import pipeline_model  # noqa: E402

LINKS = [float(x) for x in os.environ.get("LINK_GBS", "30,45,60").split(",")]


def model_partition(ctx, regs, n, p, G, NPARTS, part, final, want, t1):
    rows_of = part["rows"]
    rows, locals_ = [], {}
    for r in range(G):
        rs, k, floats = dashing_amd.exchange_mode(n, rows_of, r, NPARTS, 0, want_floats=True)
        local = final if r == 0 else torch.empty(max(floats, 1), dtype=torch.float32, device="cuda")
        locals_[r] = local

        def step():
            ctx.attach_device(regs.data_ptr(), n, p)
            ctx.exchange_rows_device_async(local.data_ptr(), rows_of, r, NPARTS, 0)
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
        pinfo = ctx.last_part_info()
        ctx.set_profiling(False)
        items = ctx.info("items")
        rows.append({"rank": r, "rows": rows_of.rows(r), "rowsorted": rs, "parts": k, "span_bytes": 4 * rows_of.pairs(r), "tiles": ctx.info("tiles"),
                     "items": items, "rounds_of_512": -(-items // 512), "bands": ctx.info("bands"),
                     "wall_ms": round(best * 1e3, 3), "prepare_ms": round(km["prepare_ms"], 3), "pair_ms": round(km["pair_ms"], 3),
                     "finalize_ms": round(km["finalize_ms"], 3), "planes_per_tile": ctx.info("avg_tile_planes_x100") / 100,
                     "part_info": [(round(a, 4), b) for a, b in pinfo]})
    # rank 0 places what it received: all sources, timed together (its per-sketch pass covers every sketch: redo rank 0's step)
    final.fill_(-1.0)
    ctx.attach_device(regs.data_ptr(), n, p)
    ctx.exchange_rows_device_async(final.data_ptr(), rows_of, 0, NPARTS, 0)
    ctx.synchronize()
    place, place_dev = 1e9, 1e9
    ctx.set_profiling(True)
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        dev_us = 0
        for r in range(1, G):
            ctx.exchange_place_device(rows_of, r, NPARTS, locals_[r].data_ptr(), final.data_ptr(), 0)
            dev_us += ctx.info("place_kernel_us") if rows[r]["rowsorted"] else 0
        place = min(place, time.perf_counter() - t0)
        place_dev = min(place_dev, dev_us * 1e-6)
    ctx.set_profiling(False)
    same = bool(torch.equal(final, want))
    # bytes per second of the placement KERNEL (one launch per source here; the exchange puts all sources' rows of a round
    # into place in one launch); `place` also holds the host's tables, their upload and a synchronisation per source
    staged = sum(x["span_bytes"] for x in rows[1:] if x["rowsorted"])
    place_rate = staged / max(place_dev, 1e-9) if staged and place_dev > 0 else 0.0
    sens = {}
    for g in LINKS:
        ms, worst = pipeline_model(rows, place_rate, g, nmsg=NPARTS)
        sens["%g" % g] = {"step_model_ms": round(ms, 3), "speedup_vs_single_gpu": round(t1 * 1e3 / ms, 2),
                          "bound_by": "link/placement of rank %d" % worst if worst else "compute"}
    walls = [x["wall_ms"] for x in rows]
    return {"partition": part["name"], "n": n, "p": p, "G": G, "nparts": NPARTS, "single_gpu_ms": round(t1 * 1e3, 3), "ranks": rows,
            "max_rank_wall_ms": max(walls), "mean_rank_wall_ms": round(sum(walls) / G, 3),
            "speedup_before_exchange": round(t1 * 1e3 / max(walls), 2),
            "dst_place_all_sources_ms": round(place * 1e3, 3), "dst_place_kernels_ms": round(place_dev * 1e3, 3) if staged else None,
            "place_rate_GBs": round(place_rate / 1e9, 1), "assembled_equals_single_gpu": same,
            "exchange_model_by_assumed_link_GBs": sens}


def main():
    n, p = int(os.environ.get("N", "10000")), int(os.environ.get("P", "14"))
    G, NPARTS = int(os.environ.get("G", "8")), int(os.environ.get("NPARTS", "8"))
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
    for name in os.environ.get("PARTITIONS", "contiguous,rowsets").split(","):
        if name == "contiguous":
            rows_of = dashing_amd.rowsets_from_bounds(n, dashing_amd.balance_rows(n, G))
        else:
            f = name.split(":")  # rowsets[:prep_permille[:dst_bonus_permille]] (-1 = the default; bonus 0 = none)
            prep = int(f[1]) if len(f) > 1 else -1
            bonus = int(f[2]) if len(f) > 2 else -1
            rows_of = dashing_amd.balance_rowsets(n, G, prep, 0, bonus)
        out = model_partition(ctx, regs, n, p, G, NPARTS, {"name": name, "rows": rows_of}, final, want, t1)
        out["opts"] = os.environ.get("OPTS", "")
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()

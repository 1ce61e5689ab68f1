#!/usr/bin/env python3
"""k_finalize accounting with in-kernel timestamps: the FULL kernel, stamped with s_memtime between
its phases (option finalize_timing), instead of early-exit stops whose occupancy and overlap differ.

  python tools/finalize_probe.py [--workloads C3,C4] [--out gpurun_out/r4a/finalize_phases.jsonl]

Per (workload, estimator, layout): HIP-event kernel times of a plain pass, then one stamped pass: cycles per phase summed
over waves -> share of the wave-time each phase holds; trip counts of the estimator per lane vs per wave (what divergence
costs).  Workloads: C3 = 10 000 x p=14, C4 = 100 000 x p=10 (configs[3] shape on one GPU)."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
WORK = {"C3": (10000, 14), "C4": (100000, 10), "C3s": (3000, 14), "P12": (30000, 12)}
PH = ["prologue+loads", "hist_columns", "joins", "fixups", "estimator", "result+store"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workloads", default="C3,C4")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "r4a", "finalize_phases.jsonl"))
    ap.add_argument("--opts", default="")
    args = ap.parse_args()
    import torch

    import dashing_amd
    from dashing_amd import synth

    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    ctx = dashing_amd.Context(0)
    for kv in filter(None, args.opts.split(",")):
        k_, v_ = kv.split("=")
        ctx.set_option(k_, int(v_))
    with open(args.out, "a") as f:
        for wl in args.workloads.split(","):
            n, p = WORK[wl]
            regs = torch.from_numpy(synth.survey_sketches(n, p, seed=0x5EED0000)[0]).cuda()
            out = torch.empty(n * (n - 1) // 2, dtype=torch.float32, device="cuda")
            for layout in ("sort1", "sort0"):
                ctx.set_option("sort", 0 if layout == "sort0" else -1)
                for estim in (2, 0):
                    ctx.set_profiling(True)
                    best = None
                    for _ in range(3):
                        ctx.attach_device(regs.data_ptr(), n, p)
                        ctx.dist_rows_device(out.data_ptr(), 0, n, estim)
                        ctx.synchronize()
                        k = ctx.last_kernel_ms()
                        if best is None or k["finalize_ms"] < best["finalize_ms"]:
                            best = k
                    ref = out[: 1 << 22].clone()
                    ctx.set_option("finalize_timing", 1)
                    ctx.attach_device(regs.data_ptr(), n, p)
                    ctx.dist_rows_device(out.data_ptr(), 0, n, estim)
                    ctx.synchronize()
                    kt = ctx.last_kernel_ms()
                    ph = ctx.finalize_phase_cycles()
                    same = bool(torch.equal(ref, out[: 1 << 22]))
                    ctx.set_option("finalize_timing", 0)
                    ctx.set_profiling(False)
                    tot = float(sum(ph[:6])) or 1.0
                    waves = ph[6] or 1
                    row = {"workload": wl, "n": n, "p": p, "layout": layout, "estim": estim,
                           "finalize_ms": round(best["finalize_ms"], 3), "pair_ms": round(best["pair_ms"], 3),
                           "finalize_ms_stamped": round(kt["finalize_ms"], 3), "stamped_output_identical": same,
                           "planes_per_tile": ctx.info("avg_tile_planes_x100") / 100.0,
                           "waves": ph[6], "cycles_per_wave": round(tot / waves, 1),
                           "phase_share": {PH[i]: round(ph[i] / tot, 4) for i in range(6)},
                           "phase_ms_of_kernel": {PH[i]: round(ph[i] / tot * best["finalize_ms"], 3) for i in range(6)},
                           "mle": {"lanes": ph[7], "iters_per_lane": round(ph[8] / max(ph[7], 1), 3), "bins_per_lane": round(ph[9] / max(ph[7], 1), 3),
                                   "steps_per_lane": round(ph[10] / max(ph[7], 1), 2),
                                   "iters_per_wave_max": round(ph[11] / waves, 3), "bins_per_wave_max": round(ph[12] / waves, 3),
                                   "steps_per_wave_paid": round(ph[13] / waves, 2),
                                   "lane_efficiency": round(ph[10] / max(ph[13] * 64, 1), 4)}}
                    print(json.dumps(row), flush=True)
                    f.write(json.dumps(row) + "\n")
                    f.flush()
            del regs, out
            torch.cuda.empty_cache()
    ctx.close()


if __name__ == "__main__":
    main()

"""what does a kernel that WAITS like an RCCL receive kernel (workgroups polling a flag, holding some
LDS) cost the phase-locked tile kernel beside it?  The destination of an exchange posts its receives either at once or
behind its first tile kernel (option xch_recv_gate); the choice was reasoned, not measured.  Here the destination's job of
BASELINE configs[2] over 8 ranks (and the single-GPU job) is timed with HIP events alone and beside `nblocks` waiting
workgroups of `threads` lanes and `lds` bytes (dsh_diag_spin_start), several shapes (SHAPES=13x256x19968: the shape
librccl 2.26.6 launches for a point-to-point message to the rank itself, profiles/rd6u/rccl_kernel_shape.csv; 7 peers
with such a kernel each would be 91 workgroups).  One JSON line per case; the ratio
feeds dashing_amd.multigpu.pipeline_model(dst_interference=...)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch

    import dashing_amd
    from dashing_amd import synth

    n, p = int(os.environ.get("N", "10000")), int(os.environ.get("P", "14"))
    world, reps = int(os.environ.get("G", "8")), int(os.environ.get("REPS", "5"))
    regs = torch.from_numpy(synth.survey_sketches(n, p, seed=0x5EED0000)[0]).cuda()
    rows = dashing_amd.balance_rowsets(n, world, -1, 0)
    shapes = [(0, 0, 0)] + [(nb, th, lds) for nb in (7, 14, 28, 56) for th, lds in ((256, 4096), (256, 32768), (512, 65536))] + [(0, 0, 0)]
    if os.environ.get("SHAPES"):  # e.g. SHAPES=13x256x19968+91x256x19968 (workgroups x lanes x LDS bytes; '+' between shapes)
        shapes = [(0, 0, 0)] + [tuple(int(x) for x in sh.split("x")) for sh in os.environ["SHAPES"].replace(",", "+").split("+")] + [(0, 0, 0)]
    with dashing_amd.Context(0) as ctx:
        for job in ("dst_of_%d" % world, "single_gpu"):
            floats = dashing_amd.exchange_mode(n, rows, 0, 8, 0, want_floats=True)[2] if job != "single_gpu" else n * (n - 1) // 2
            out = torch.empty(floats, dtype=torch.float32, device="cuda")
            base = None
            for nb, th, lds in shapes:
                ctx.set_profiling(True)
                acc = {"pair_ms": 0.0, "finalize_ms": 0.0, "prepare_ms": 0.0}
                for _ in range(reps + 1):
                    ctx.attach_device(regs.data_ptr(), n, p)
                    if nb:
                        ctx.diag_spin_start(nb, th, lds, 2000)
                    if job == "single_gpu":
                        ctx.dist_rows_device(out.data_ptr(), 0, n)
                    else:
                        ctx.exchange_rows_device_async(out.data_ptr(), rows, 0, 8, 0)
                    ctx.synchronize()
                    if nb:
                        ctx.diag_spin_stop()
                    k = ctx.last_kernel_ms()
                    if _:  # (the first pass warms up)
                        for key in acc:
                            acc[key] += k[key] / reps
                ctx.set_profiling(False)
                if base is None:
                    base = dict(acc)
                print(json.dumps({"job": job, "n": n, "p": p, "waiting_workgroups": nb, "threads": th, "lds_bytes": lds,
                                  "items": ctx.info("items"), "rounds_of_512": -(-ctx.info("items") // 512),
                                  **{k_: round(v, 4) for k_, v in acc.items()},
                                  "pair_ratio_vs_alone": round(acc["pair_ms"] / base["pair_ms"], 4),
                                  "finalize_ratio_vs_alone": round(acc["finalize_ms"] / base["finalize_ms"], 4)}), flush=True)


if __name__ == "__main__":
    main()

"""What librccl's own point-to-point kernel costs a SOURCE rank's step, measured on one GPU: rank r of an 8-rank plan of
BASELINE configs[2] (10 000 x p=14) computes its rows (dsh_exchange_rows_device_async) alone, and with its buffer sent
message by message THROUGH librccl to the rank itself (a communicator of one rank; dsh_exchange_probe_parts_async behind
the parts' gates: ncclSend + ncclRecv per message, as dsh_exchange_collect_async would send them to the destination).
The copy is local (HBM speed, not a link), so the transfer itself is shorter than a real one; what the numbers show is
(a) that the messages leave while the rank still computes -- the step with the loop is barely longer than without -- and
(b) what RCCL's kernel beside the tile kernel / k_finalize costs them.  One JSON line per (rank, mode)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch

    import dashing_amd
    from dashing_amd import synth

    n, p, world, nparts, dst = int(os.environ.get("N", "10000")), int(os.environ.get("P", "14")), 8, 8, 0
    reps = int(os.environ.get("REPS", "10"))
    regs = torch.from_numpy(synth.survey_sketches(n, p, seed=0x5EED0000)[0]).cuda()
    rows = dashing_amd.balance_rowsets(n, world, -1, dst)
    with dashing_amd.Context(0) as ctx:
        ctx.comm_init(dashing_amd.comm_unique_id(), 0, 1)
        path, version = dashing_amd.comm_library()
        try:
            for rank in (1, 4, 7):
                floats = dashing_amd.exchange_mode(n, rows, rank, nparts, dst, want_floats=True)[2]
                local = torch.empty(floats, dtype=torch.float32, device="cuda")
                probe = torch.empty(floats, dtype=torch.float32, device="cuda")
                for mode in ("alone", "through_librccl", "alone", "through_librccl"):
                    ctx.set_profiling(True)
                    wall, acc = 0.0, {"pair_ms": 0.0, "finalize_ms": 0.0, "prepare_ms": 0.0}
                    for it in range(reps + 1):
                        torch.cuda.synchronize()
                        t0 = time.perf_counter()
                        ctx.attach_device(regs.data_ptr(), n, p)
                        ctx.exchange_rows_device_async(local.data_ptr(), rows, rank, nparts, dst)
                        if mode != "alone":
                            ctx.exchange_probe_parts_async(n, rows, rank, nparts, local.data_ptr(), probe.data_ptr(), dst)
                        ctx.comm_wait()
                        dt = (time.perf_counter() - t0) * 1e3
                        if it:
                            wall += dt / reps
                            k = ctx.last_kernel_ms()
                            for key in acc:
                                acc[key] += k[key] / reps
                    ctx.set_profiling(False)
                    ok = bool(torch.equal(probe, local)) if mode != "alone" else None
                    print(json.dumps({"rank": rank, "of": world, "n": n, "p": p, "mode": mode, "wall_ms_per_step": round(wall, 4),
                                      **{k_: round(v, 4) for k_, v in acc.items()}, "message_mb": round(floats * 4 / nparts / 1e6, 3),
                                      "delivered_equals_computed": ok, "librccl": os.path.basename(path), "nccl_version": version}), flush=True)
        finally:
            ctx.comm_destroy()


if __name__ == "__main__":
    main()

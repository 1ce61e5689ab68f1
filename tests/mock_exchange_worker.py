"""Worker of tests/test_gpu_multirank.py::test_exchange_protocol_between_processes: ONE rank of a world > 1 run of the
library's exchange on a box with one GPU.  Every rank is a process of its own on cuda:0; libdashing_hip.so loads
tests/mock_rccl/libmock_rccl.so (DSH_RCCL_LIB) instead of librccl: the same dsh_comm_* / dsh_exchange_* / dsh_collect_*
calls a real multi-GPU run makes, with messages that only match when both sides agree on order, peer and size.

env: RANK, WORLD, N, P, NPARTS, MODE (exchange | parts | spans | collect | collect-auto | allgather), ID_FILE (rank 0 leaves
the unique id there), DST, ROWSETS=1 (MODE exchange: the balanced row-set table -- ranges + top-up tile rows -- instead of
contiguous bounds).  Rank DST checks the assembled matrix against its own single-GPU result, byte for byte."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch

    import dashing_amd
    from dashing_amd import synth

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD"])
    n, p, nparts = int(os.environ["N"]), int(os.environ["P"]), int(os.environ["NPARTS"])
    mode, dst = os.environ["MODE"], int(os.environ.get("DST", "0"))
    idf = os.environ["ID_FILE"]
    assert "mock_rccl" in dashing_amd.comm_library()[0], dashing_amd.comm_library()
    if rank == 0:
        uid = dashing_amd.comm_unique_id()
        with open(idf + ".tmp", "wb") as f:
            f.write(bytes(uid))
        os.rename(idf + ".tmp", idf)
    else:
        t0 = time.time()
        while not os.path.exists(idf):
            assert time.time() - t0 < 300, "rank 0 never wrote the id"
            time.sleep(0.01)
        uid = open(idf, "rb").read()
    regs = torch.from_numpy(synth.survey_sketches(n, p, seed=0xE0C)[0]).cuda()
    ctx = dashing_amd.Context(0)
    for kv in filter(None, os.environ.get("OPTS", "").split(",")):
        k_, v_ = kv.split("=")
        ctx.set_option(k_, int(v_))
    ctx.attach_device(regs.data_ptr(), n, p)
    ctx.comm_init(uid, rank, world)
    assert ctx.comm_rank() == (rank, world)
    total = n * (n - 1) // 2
    bounds = dashing_amd.balance_rows(n, world)
    if os.environ.get("BOUNDS"):  # ragged ranges, empty ones included
        bounds = [int(x) for x in os.environ["BOUNDS"].split(",")]
    span = dashing_amd.tri_span(n, bounds[rank], bounds[rank + 1])
    first_row = bounds[rank]
    rows = bounds
    if os.environ.get("ROWSETS") and mode == "exchange":
        rows = dashing_amd.balance_rowsets(n, world)
        span = dashing_amd.exchange_mode(n, rows, rank, nparts, dst, want_floats=True)[2]
        first_row = rows.rows(rank)[0][0] if rows.rows(rank) else 0
        if os.environ.get("EXPECT_TOPUPS"):
            assert any(len(rows.rows(r)) > 1 for r in range(world)), "this case is meant to exercise top-up segments"
    final = torch.zeros(max(total, 1), dtype=torch.float32, device="cuda") if rank == dst else None
    in_place = rank == dst and mode in ("exchange", "parts", "spans")
    local = final[dashing_amd.tri_span(n, 0, first_row):] if in_place else torch.zeros(max(span, 1), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    lp, fp = local.data_ptr(), (final.data_ptr() if rank == dst else 0)
    got = None
    if mode == "exchange":  # what bench.py --gpus N runs
        for _ in range(2):  # twice: the second call reuses staging, tables and events
            ctx.attach_device(regs.data_ptr(), n, p)
            ctx.exchange_rows_device_async(lp, rows, rank, nparts, dst)
            ctx.exchange_collect_async(n, rows, nparts, 0 if rank == dst else lp, fp, dst)
            ctx.comm_wait()
        got = final
    elif mode == "parts":
        ctx.dist_rows_parts_device_async(lp, bounds[rank], bounds[rank + 1], nparts)
        ctx.collect_parts_async(n, bounds, nparts, 0 if rank == dst else lp, fp, dst)
        ctx.comm_wait()
        got = final
    elif mode == "spans":
        ctx.dist_rows_device(lp, bounds[rank], bounds[rank + 1])
        ctx.collect_spans(n, bounds, 0 if rank == dst else lp, fp, dst, wait=True)
        got = final
    elif mode == "collect":  # the whole step for a host without device pointers
        out = ctx.dist_collect(bounds, dst)
        got = torch.from_numpy(np.asarray(out)).cuda() if rank == dst else None
    elif mode == "collect-auto":  # the same with the library's own partition (balanced row sets, the pipelined pair)
        out = ctx.dist_collect(None, dst)
        got = torch.from_numpy(np.asarray(out)).cuda() if rank == dst else None
    elif mode == "allgather":
        m = 1 << p
        per = (n + world - 1) // world
        mine = torch.zeros((per, m), dtype=torch.uint8, device="cuda")
        rows = list(range(rank, n, world))
        mine[: len(rows)] = regs[rows]
        allr = torch.empty((world, per, m), dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        ctx.allgather_device(mine.data_ptr(), per * m, allr.data_ptr())
        assert torch.equal(allr.permute(1, 0, 2).reshape(per * world, m)[:n], regs), "all-gather of the register arrays"
    else:
        raise SystemExit("unknown MODE " + mode)
    if rank == dst and got is not None:
        torch.cuda.synchronize()
        ctx.comm_destroy()
        want = torch.empty(max(total, 1), dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        ctx.attach_device(regs.data_ptr(), n, p)
        ctx.dist_rows_device(want.data_ptr(), 0, n)
        ctx.synchronize()
        assert torch.equal(got[:total], want[:total]), "the assembled matrix differs from the single-GPU one"
    else:
        ctx.comm_destroy()
    print("MOCK_EXCHANGE_OK rank %d" % rank, flush=True)


if __name__ == "__main__":
    main()

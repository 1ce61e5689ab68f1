"""Worker for tests/test_gpu_multirank.py: the whole N-rank path on real GPU kernels --
genomes dealt to ranks -> k_sketch -> all-gather of the register arrays -> all-pairs matrix split over
the ranks -> assembled on rank 0 -- checked against the CPU oracle on rank 0.  E2E_PIECES=0: the scheme
bench.py uses (tile-balanced row ranges of the final triangle, each rank's span sent point-to-point into
place, nothing re-ordered); >= 1: cost-balanced shards of the key-ordered triangle, gather, un-permute.

Launched with torchrun (RANK/WORLD_SIZE/MASTER_* from the env).  E2E_BACKEND=nccl is the real thing
(one GPU per rank; with WORLD_SIZE=1 it still initialises RCCL and runs every collective);
E2E_BACKEND=gloo lets several ranks share cuda:0 on a 1-GPU box (collectives staged through the host)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    backend = os.environ.get("E2E_BACKEND", "nccl")
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0")) if backend == "nccl" else 0
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)

    import dashing_amd
    from dashing_amd import multigpu, synth

    n, p, k = 301, 12, 31  # 3 tile rows of 128 sketches: every rank of a 2- or 3-rank run gets a shard
    lens = [9000 + 477 * ((g * 7) % 11) for g in range(n)]
    base = synth.synthetic_genomes(n, 14000, seed=0xE2E)
    genomes = [g[:L] for g, L in zip(base, lens)]
    m = 1 << p

    ctx = dashing_amd.Context(local_rank)
    # ---- sketch my share (genome g -> rank g % world), rows in local order
    mine = multigpu.deal_genomes(n, rank, world)
    per = (n + world - 1) // world
    ctx.alloc(per, p)
    if mine:
        seq, off = synth.concat_for_device([genomes[g] for g in mine])
        ctx.sketch_batch(seq, off, 0, k, True, want_regs=False)
    local = torch.empty((per, m), dtype=torch.uint8, device=dev)
    ctx.copy_sketches_device(local.data_ptr(), 0, per)
    # ---- all-gather the register arrays: every rank gets all n sketches in input order
    if backend == "gloo":
        regs_d = multigpu.allgather_sketches(local.cpu(), n, rank, world).to(dev)
    else:
        regs_d = multigpu.allgather_sketches(local, n, rank, world)
    torch.cuda.synchronize()
    # ---- all-pairs: cost-balanced shards (E2E_PIECES per rank), gather on rank 0, un-permute
    ctx.attach_device(regs_d.data_ptr(), n, p)
    total = n * (n - 1) // 2
    pieces = int(os.environ.get("E2E_PIECES", "1"))
    rt = dashing_amd.MASH_DIST
    pipe = None
    ranges = pieces == 0  # the scheme bench.py uses: tile-balanced row ranges of the FINAL triangle, point-to-point
    if ranges:
        # n = 301 has 3 tile rows; min rows for the key-ordered layout lowered so the small ranges take that path too
        ctx.set_option("range_sort_min_rows", 1)
        bounds = dashing_amd.balance_rows(n, world)
        sizes = multigpu.span_sizes(n, bounds)
        final_r = torch.zeros(total, dtype=torch.float32, device=dev) if rank == 0 else None
        my = torch.zeros(max(sizes[rank], 1), dtype=torch.float32, device=dev)
        torch.cuda.synchronize()
        dst_ptr = final_r.data_ptr() if rank == 0 else my.data_ptr()  # rank 0 computes in place
        if backend == "nccl" and os.environ.get("E2E_CABI") == "parts":
            # the pipelined form: rows in parts, each part sent behind its event while the next is finalized
            multigpu.cabi_comm_init(ctx, rank, world)
            ctx.dist_rows_parts_device_async(dst_ptr, bounds[rank], bounds[rank + 1], 3, dashing_amd.ESTIM_ERTL_MLE, rt, k)
            ctx.collect_parts_async(n, bounds, 3, 0 if rank == 0 else my.data_ptr(), final_r.data_ptr() if rank == 0 else 0, 0)
            ctx.wait()
            full = final_r
        else:
            ctx.dist_rows_device(dst_ptr, bounds[rank], bounds[rank + 1], dashing_amd.ESTIM_ERTL_MLE, rt, k)
            ctx.synchronize()
        if backend == "nccl" and os.environ.get("E2E_CABI") == "parts":
            pass
        elif backend == "gloo":
            fh = final_r.cpu() if rank == 0 else None
            multigpu.collect_row_spans(my.cpu(), fh, n, bounds, rank, world, 0)
            full = fh.to(dev) if rank == 0 else None
        elif os.environ.get("E2E_CABI"):  # the exchange inside libdashing_hip.so (dsh_comm_init / dsh_collect_spans)
            multigpu.cabi_comm_init(ctx, rank, world)
            assert ctx.comm_rank() == (rank, world)
            full = multigpu.collect_row_spans_cabi(ctx, my, final_r, n, bounds, rank, 0)
            # the all-gather of the register arrays through the same communicator gives what torch's gave
            allr = torch.empty((world, per, m), dtype=torch.uint8, device=dev)
            torch.cuda.synchronize()
            ctx.allgather_device(local.data_ptr(), per * m, allr.data_ptr())
            assert torch.equal(allr.permute(1, 0, 2).reshape(per * world, m)[:n], regs_d)
        else:
            full = multigpu.collect_row_spans(my, final_r, n, bounds, rank, world, 0)
    elif pieces == 1:
        span_off = ctx.shard_plan(world)
        mx = max(max(span_off[r + 1] - span_off[r] for r in range(world)), 1)
        out_d = torch.zeros(mx, dtype=torch.float32, device=dev)
        torch.cuda.synchronize()  # the fill runs on torch's stream, the library on its own
        ctx.dist_shard_device(out_d.data_ptr(), rank, world, dashing_amd.ESTIM_ERTL_MLE, rt, k)
        ctx.synchronize()
        if backend == "gloo":
            full = multigpu.gather_shard_spans(out_d.cpu(), span_off, rank, world)
            full = full.to(dev) if rank == 0 else None
        else:  # as bench.py: un-permute straight from the gathered (padded) blocks
            full = multigpu.gather_shard_spans(out_d, span_off, rank, world, staged=True)
    else:  # pipelined: piece h is gathered while piece h+1 is computed (multigpu.PipelinedShards)
        span_off = ctx.shard_plan(world * pieces)
        pipe = multigpu.PipelinedShards(span_off, rank, world, pieces, dev if backend == "nccl" else torch.device("cpu"))
        for h in range(pieces):
            buf = pipe.out(h) if backend == "nccl" else torch.empty(pipe.mx[h], dtype=torch.float32, device=dev)
            ctx.dist_shard_device(buf.data_ptr(), pipe.shard(h), world * pieces, dashing_amd.ESTIM_ERTL_MLE, rt, k)
            ctx.synchronize()
            if backend != "nccl":
                pipe.out(h).copy_(buf)
            pipe.submit(h)
        got = pipe.wait()
        torch.cuda.current_stream().synchronize()
        full = None
        if rank == 0:
            full, block_off = got
            full = full.to(dev) if backend != "nccl" else full
    ok = True
    if rank == 0:
        torch.cuda.synchronize()
        final = torch.empty(total, dtype=torch.float32, device=dev)
        torch.cuda.synchronize()
        if ranges:
            final = full  # already in dashing's packed order: nothing to un-permute
        elif pipe is not None:
            ctx.unpermute_blocks_device(full.data_ptr(), block_off, final.data_ptr())
        elif backend == "gloo":
            ctx.unpermute_device(full.data_ptr(), final.data_ptr())
        else:
            ctx.unpermute_staged_device(full.data_ptr(), mx, world, final.data_ptr())
        ctx.synchronize()
        from oracle import oracle_c  # the checker (tests only)

        oracle_c.load(threads=4)
        seq, off = synth.concat_for_device(genomes)
        want_regs = oracle_c.sketch_batch(seq, off, k, p, True)
        got_regs = regs_d.cpu().numpy()
        assert (got_regs == want_regs).all(), "all-gathered registers differ from the oracle"
        want = oracle_c.dist_tri(want_regs, result_type=rt, k=k)
        got = final.cpu().numpy()
        rel = np.abs(got.astype(np.float64) - want) / np.maximum(np.abs(want), 1e-9)
        assert rel.max() <= 1e-6, rel.max()
        print("E2E_OK world=%d backend=%s pieces=%d pairs=%d max_rel=%.3g" % (world, backend, pieces, total, rel.max()), flush=True)
    dist.barrier()
    ctx.close()
    dist.destroy_process_group()
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())

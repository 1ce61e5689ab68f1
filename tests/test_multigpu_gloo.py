"""CPU, world_size 2 (and 3), gloo: the N>1 path of bench.py -- row partition by pair count,
per-rank contiguous span of the packed triangle, padded gather to rank 0 -- assembles a matrix
byte-identical to the single-rank one.  The per-rank compute is stood in by the CPU oracle here
(no GPU in this container); on the GPU box the same multigpu.gather_spans runs over RCCL."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, p, outdir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dashing_amd import multigpu, synth
    from oracle import oracle_c

    oracle_c.load(threads=2)
    regs = synth.synthetic_sketches(n, p, seed=123)
    bounds = multigpu.row_bounds(n, world)
    mx = multigpu.max_span(n, bounds)
    local = torch.zeros(max(mx, 1), dtype=torch.float32)
    rows = oracle_c.dist_rows(regs, bounds[rank], bounds[rank + 1])
    local[: rows.size] = torch.from_numpy(rows)
    full = multigpu.gather_spans(local, n, bounds, rank, world, 0)
    if rank == 0:
        np.save(os.path.join(outdir, "full_%d.npy" % world), full.numpy())
    else:
        assert full is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_gather_spans_gloo(tmp_path, world):
    n, p = 300, 8
    mp.spawn(_worker, args=(world, _free_port(), n, p, str(tmp_path)), nprocs=world, join=True)
    from dashing_amd import synth
    from oracle import oracle_c

    regs = synth.synthetic_sketches(n, p, seed=123)
    want = oracle_c.dist_tri(regs)
    got = np.load(os.path.join(str(tmp_path), "full_%d.npy" % world))
    assert got.tobytes() == want.tobytes()


def _worker_spans(rank, world, port, outdir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dashing_amd import multigpu

    span_off = [0, 1000, 1700, 4000][: world + 1]  # unequal spans, as a cost-balanced plan gives
    mx = max(span_off[r + 1] - span_off[r] for r in range(world))
    local = torch.full((mx,), -1.0)
    local[: span_off[rank + 1] - span_off[rank]] = torch.arange(span_off[rank], span_off[rank + 1], dtype=torch.float32)
    full = multigpu.gather_shard_spans(local, span_off, rank, world)
    if rank == 0:
        np.save(os.path.join(outdir, "spans_%d.npy" % world), full.numpy())
    else:
        assert full is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_gather_shard_spans_gloo(tmp_path, world):
    """bench.py's N>1 exchange: padded gather of unequal sorted-order spans, laid back to back."""
    mp.spawn(_worker_spans, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    got = np.load(os.path.join(str(tmp_path), "spans_%d.npy" % world))
    end = [0, 1000, 1700, 4000][world]
    assert (got == np.arange(end, dtype=np.float32)).all()


def _worker_collect(rank, world, port, n, p, outdir, dst):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dashing_amd import api, multigpu, synth
    from oracle import oracle_c

    oracle_c.load(threads=2)
    regs = synth.synthetic_sketches(n, p, seed=321)
    bounds = api.partition_rows(n, world, 1)  # pair-balanced, unaligned: any row range is a contiguous final span
    sizes = multigpu.span_sizes(n, bounds)
    total = n * (n - 1) // 2
    rows = torch.from_numpy(oracle_c.dist_rows(regs, bounds[rank], bounds[rank + 1]))
    final = torch.full((total,), -7.0) if rank == dst else None
    local = torch.full((max(sizes[rank], 1) + 3,), -1.0)
    if rank == dst:  # the receiver computes its own rows in place
        off = sum(sizes[:rank])
        final[off : off + sizes[rank]] = rows
    else:
        local[: sizes[rank]] = rows
    got = multigpu.collect_row_spans(local, final, n, bounds, rank, world, dst)
    if rank == dst:
        np.save(os.path.join(outdir, "collect_%d.npy" % world), got.numpy())
    else:
        assert got is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,dst", [(2, 0), (3, 0), (3, 2)])
def test_collect_row_spans_gloo(tmp_path, world, dst):
    """bench.py's N>1 exchange: every rank computes a row range (a contiguous span of the FINAL packed
    triangle); point-to-point into place on the receiver -- no padding, no staging, no un-permute."""
    n, p = 301, 8
    mp.spawn(_worker_collect, args=(world, _free_port(), n, p, str(tmp_path), dst), nprocs=world, join=True)
    from dashing_amd import synth
    from oracle import oracle_c

    want = oracle_c.dist_tri(synth.synthetic_sketches(n, p, seed=321))
    got = np.load(os.path.join(str(tmp_path), "collect_%d.npy" % world))
    assert got.tobytes() == want.tobytes()


def test_bounds_cover_and_align():
    from dashing_amd import multigpu

    for n in (10000, 300, 129, 5):
        for w in (1, 2, 4, 8):
            b = multigpu.row_bounds(n, w)
            assert b[0] == 0 and b[-1] == n
            assert sum(multigpu.span_sizes(n, b)) == n * (n - 1) // 2
            assert all(x % 128 == 0 or x == n for x in b)


def _worker_sketch(rank, world, port, n, p, outdir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dashing_amd import multigpu, synth
    from oracle import oracle_c

    oracle_c.load(threads=2)
    gs = synth.synthetic_genomes(n, 6000, seed=31)
    mine = multigpu.deal_genomes(n, rank, world)
    per = (n + world - 1) // world
    local = torch.zeros((per, 1 << p), dtype=torch.uint8)
    if mine:  # the per-rank sketching is stood in by the oracle here (no GPU in this container)
        seq, off = synth.concat_for_device([gs[g] for g in mine])
        local[: len(mine)] = torch.from_numpy(oracle_c.sketch_batch(seq, off, 21, p))
    full = multigpu.allgather_sketches(local, n, rank, world)
    np.save(os.path.join(outdir, "regs_%d_%d.npy" % (world, rank)), full.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n", [(2, 7), (3, 7), (2, 1)])
def test_sharded_sketching_allgather_gloo(tmp_path, world, n):
    """Genomes dealt round-robin to the ranks, register arrays all-gathered: every rank ends up with
    the matrix a single rank would have sketched, rows in input order (ragged shares are padded)."""
    p = 8
    mp.spawn(_worker_sketch, args=(world, _free_port(), n, p, str(tmp_path)), nprocs=world, join=True)
    from dashing_amd import synth
    from oracle import oracle_c

    seq, off = synth.concat_for_device(synth.synthetic_genomes(n, 6000, seed=31))
    want = oracle_c.sketch_batch(seq, off, 21, p)
    for r in range(world):
        got = np.load(os.path.join(str(tmp_path), "regs_%d_%d.npy" % (world, r)))
        assert got.shape == want.shape and (got == want).all()


def _worker_pipe(rank, world, port, pieces, outdir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dashing_amd import multigpu

    S = world * pieces
    sizes = [37 + 11 * ((s * 5) % 7) for s in range(S)]
    sizes[1] = 0  # an empty shard is legal (more shards than tile rows)
    span_off = [0]
    for z in sizes:
        span_off.append(span_off[-1] + z)
    pipe = multigpu.PipelinedShards(span_off, rank, world, pieces, torch.device("cpu"))
    for h in range(pieces):
        s_ = pipe.shard(h)
        buf = pipe.out(h)
        buf.fill_(-1.0)
        buf[: sizes[s_]] = torch.arange(span_off[s_], span_off[s_ + 1], dtype=torch.float32)
        pipe.submit(h)
    got = pipe.wait()
    if rank == 0:
        stage, block_off = got
        full = torch.cat([stage[block_off[s] : block_off[s] + sizes[s]] for s in range(S)])
        np.save(os.path.join(outdir, "pipe_%d_%d.npy" % (world, pieces)), full.numpy())
    else:
        assert got is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,pieces", [(2, 3), (3, 2), (2, 1)])
def test_pipelined_shards_gloo(tmp_path, world, pieces):
    """Piece-by-piece asynchronous gather: every shard lands in its own block of the staging buffer."""
    mp.spawn(_worker_pipe, args=(world, _free_port(), pieces, str(tmp_path)), nprocs=world, join=True)
    got = np.load(os.path.join(str(tmp_path), "pipe_%d_%d.npy" % (world, pieces)))
    assert (got == np.arange(got.size, dtype=np.float32)).all() and got.size >= 37


def _worker_rowsets(rank, world, port, n, p, dst, outdir):
    """every rank computes the rows of ITS ROW SET (dsh_balance_rowsets: a range + top-up tile rows), segment by segment
    in final order, and sends each segment's span to `dst`, which puts it at dsh_tri_span(n, 0, first row)"""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import dashing_amd
    from dashing_amd import synth
    from oracle import oracle_c

    oracle_c.load(threads=2)
    regs = synth.synthetic_sketches(n, p, seed=321)
    rows = dashing_amd.balance_rowsets(n, world, dst=dst, dst_bonus_permille=100)  # (identical on every rank)
    total = n * (n - 1) // 2
    full = torch.full((total,), -1.0) if rank == dst else None
    for r in range(world):
        for b, e in rows.rows(r):
            off, cnt = dashing_amd.tri_span(n, 0, b), dashing_amd.tri_span(n, b, e)
            if r == rank:
                seg = torch.from_numpy(oracle_c.dist_rows(regs, b, e))
                assert seg.numel() == cnt
                if rank == dst:
                    full[off:off + cnt] = seg
                else:
                    dist.send(seg, dst)
            elif rank == dst:
                buf = torch.empty(cnt)
                dist.recv(buf, r)
                full[off:off + cnt] = buf
    if rank == dst:
        np.save(os.path.join(outdir, "rowsets_%d.npy" % world), full.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,dst", [(2, 0), (3, 1)])
def test_row_sets_assemble_the_matrix_gloo(tmp_path, world, dst):
    """N > 1 on CPU: the balanced row sets (with a destination bonus) partition the triangle -- the segments of all ranks,
    each placed at its span, are the single-rank matrix byte for byte; some rank holds a top-up segment"""
    import dashing_amd

    n, p = 1700, 8
    assert any(len(dashing_amd.balance_rowsets(n, world, dst=dst, dst_bonus_permille=100).rows(r)) > 1 for r in range(world))
    mp.spawn(_worker_rowsets, args=(world, _free_port(), n, p, dst, str(tmp_path)), nprocs=world, join=True)
    from dashing_amd import synth
    from oracle import oracle_c

    want = oracle_c.dist_tri(synth.synthetic_sketches(n, p, seed=321))
    got = np.load(os.path.join(str(tmp_path), "rowsets_%d.npy" % world))
    assert got.tobytes() == want.tobytes()

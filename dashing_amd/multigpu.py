"""Row-block sharding of the packed upper triangle across ranks + gather of the distance rows
(torch.distributed: backend "nccl" is RCCL over xGMI on the GPU box, "gloo" in CPU tests).

Every pair is independent, so there is no data-path collective inside the compare itself: rank r
computes the contiguous row range [bounds[r], bounds[r+1]) (near-equal pair counts, boundaries on
whole 128-row tile rows) and the only exchange is the gather of the per-rank spans of the packed
triangle to rank 0 -- the writer, as in dashing where one process emits the matrix
(src/sketch_and_cmp.h:838-849)."""
import torch
import torch.distributed as dist

from . import api


def _host_sync(t):
    """Collectives on CUDA tensors only ENQUEUE on torch's current stream; the library runs on its own
    stream.  Block the host until the transfer has completed so that whatever the caller does next with the
    buffers (another dsh_* call on the library stream, reuse of the send buffer) is ordered after it."""
    if t is not None and t.is_cuda:
        torch.cuda.current_stream(t.device).synchronize()


def row_bounds(n, world):
    return api.partition_rows(n, world, 128)


def span_sizes(n, bounds):
    return [api.tri_span(n, bounds[r], bounds[r + 1]) for r in range(len(bounds) - 1)]


def gather_spans(local, n, bounds, rank, world, dst=0, staging=None):
    """local: 1-D float32 tensor holding this rank's span (may be longer than the span: padded).
    Returns the full packed triangle on `dst` (a 1-D tensor of n(n-1)/2 floats), None elsewhere.
    Spans are padded to the largest span so one dist.gather moves everything."""
    sizes = span_sizes(n, bounds)
    if world == 1:
        return local[: sizes[0]]
    mx = max(sizes)
    assert local.numel() >= mx, "allocate the local span with max_span(n, bounds) elements"
    send = local[:mx]
    if rank == dst:
        if staging is None:
            staging = [torch.empty(mx, dtype=local.dtype, device=local.device) for _ in range(world)]
        dist.gather(send, gather_list=staging, dst=dst)
        out = torch.cat([staging[r][: sizes[r]] for r in range(world)])
        _host_sync(out)
        return out
    dist.gather(send, gather_list=None, dst=dst)
    _host_sync(send)
    return None


def max_span(n, bounds):
    return max(span_sizes(n, bounds))


def cabi_comm_init(ctx, rank, world):
    """Give `ctx` an RCCL communicator of its own INSIDE libdashing_hip.so (dsh_comm_init): the exchange then runs on
    the library's stream, ordered with its kernels, and a C++ host gets the same call sequence without torch.
    torch.distributed only carries the 128-byte id from rank 0 to the others (any backend)."""
    ids = [api.comm_unique_id() if rank == 0 else None]
    if dist.is_initialized():
        dist.broadcast_object_list(ids, src=0)
    ctx.comm_init(ids[0], rank, world)


def collect_row_spans_cabi(ctx, local, final, n, bounds, rank, dst=0, wait=True):
    """collect_row_spans through the C-ABI (dsh_collect_spans): grouped ncclSend/ncclRecv on the ctx stream, every span
    received at its final place on `dst`.  `final` is only needed on dst, whose own span is expected to be there
    already (computed in place), exactly as with collect_row_spans."""
    ctx.collect_spans(n, bounds, 0 if rank == dst else local.data_ptr(), final.data_ptr() if rank == dst else 0, dst, wait=wait)
    return final if rank == dst else None


def collect_row_spans(local, final, n, bounds, rank, world, dst=0):
    """The multi-GPU exchange of the distance matrix: rank r has computed the rows [bounds[r], bounds[r+1])
    (Context.dist_rows_device), i.e. ONE contiguous span of the packed triangle already in its final
    order, in `local` (>= its span long).  Every rank != dst sends its span, dst receives each span
    straight into its place in `final` (n(n-1)/2 floats; dst's own span is expected to be there already
    -- compute it in place) -- point-to-point, one message per peer, all 7 xGMI links of dst busy at once,
    no staging copy and no un-permute.  Returns when the data has arrived (host-synchronised)."""
    sizes = span_sizes(n, bounds)
    offs = [0]
    for s_ in sizes:
        offs.append(offs[-1] + s_)
    if world == 1:
        return final
    ops = []
    if rank == dst:
        for r in range(world):
            if r != dst and sizes[r]:
                ops.append(dist.P2POp(dist.irecv, final[offs[r] : offs[r + 1]], r))
    elif sizes[rank]:
        ops.append(dist.P2POp(dist.isend, local[: sizes[rank]], dst))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    _host_sync(final if rank == dst else local)
    return final if rank == dst else None


# ---- shards of the sorted-order triangle (faster: narrow per-tile plane ranges, cost-balanced) ----
def gather_shard_spans(local, span_off, rank, world, stage=None, sorted_full=None, dst=0, staged=False):
    """local: this rank's span padded to max span.  On `dst` returns the spans laid back to back
    (the packed triangle in sorted order) ready for Context.unpermute_device; None elsewhere.
    staged=True skips the back-to-back copy: `dst` gets the gathered blocks as they arrived, shard r
    at stage[r*mx : r*mx+span_r] with mx = the largest span, for Context.unpermute_staged_device."""
    sizes = [span_off[r + 1] - span_off[r] for r in range(world)]
    mx = max(max(sizes), 1)
    assert local.numel() >= mx
    if world == 1 and not dist.is_initialized():
        return local[:mx] if staged else local[: sizes[0]]
    send = local[:mx]
    if rank != dst:
        dist.gather(send, gather_list=None, dst=dst)
        _host_sync(send)  # the send buffer may be overwritten by the next dsh_* call only after this
        return None
    if stage is None:
        stage = torch.empty(world * mx, dtype=local.dtype, device=local.device)
    parts = [stage[r * mx : (r + 1) * mx] for r in range(world)]
    dist.gather(send, gather_list=parts, dst=dst)
    if staged:
        _host_sync(stage)  # the un-permute runs on the library's stream, not torch's
        return stage
    if sorted_full is None:
        sorted_full = torch.empty(max(span_off[-1], 1), dtype=local.dtype, device=local.device)
    for r in range(world):
        sorted_full[span_off[r] : span_off[r + 1]] = parts[r][: sizes[r]]
    _host_sync(sorted_full)
    return sorted_full


# ---- sharded sketching: genomes dealt to ranks, register arrays all-gathered ----
def deal_genomes(n_genomes, rank, world):
    """Genome g (in the size-sorted input order of sort_paths_by_fsize, src/finalizers.cpp:6-21) goes to
    rank g % world: round-robin over the descending sizes keeps the bases per rank balanced."""
    return list(range(rank, n_genomes, world))


def allgather_sketches(local, n_genomes, rank, world):
    """local: uint8 tensor [ceil(n_genomes/world)][2^p] -- row t is this rank's t-th genome (genome
    rank + t*world; rows past the rank's share are padding).  Returns the full matrix [n_genomes][2^p]
    in input order on EVERY rank (one all_gather_into_tensor of equal-sized blocks: RCCL over xGMI on
    the GPU box, gloo in the CPU tests), ready for Context.attach_device."""
    per = (n_genomes + world - 1) // world
    assert local.dim() == 2 and local.shape[0] == per and local.dtype == torch.uint8
    if world == 1 and not dist.is_initialized():
        return local[:n_genomes]
    m = local.shape[1]
    allr = torch.empty((world, per, m), dtype=torch.uint8, device=local.device)
    if dist.get_backend() == "gloo":  # gloo has no all_gather_into_tensor for every build: use the list form
        parts = [allr[r] for r in range(world)]
        dist.all_gather(parts, local.contiguous())
    else:
        dist.all_gather_into_tensor(allr, local.contiguous())
    # rank-major [r][t] -> genome order g = t*world + r
    return allr.permute(1, 0, 2).reshape(per * world, m)[:n_genomes].contiguous()


# ---- pipelined gather: K pieces per rank, piece h travels while piece h+1 is computed -------------
class PipelinedShards:
    """Rank r owns the K consecutive shards r*K .. r*K+K-1 of a (world*K)-shard plan.  After computing
    piece h (Context.dist_shard_device(out(h), shard(h), nshards)) the rank calls submit(h): the piece
    is gathered to `dst` asynchronously (its own block of the staging buffer, all ranks' pieces h padded
    to the largest), so the transfer over xGMI overlaps the computation of piece h+1.  wait() returns
    (stage, block_off) on `dst` for Context.unpermute_blocks_device, None elsewhere.  The collectives are
    issued in the same order on every rank (one gather per piece)."""

    def __init__(self, span_off, rank, world, pieces, device, dst=0, dtype=torch.float32):
        assert len(span_off) == world * pieces + 1
        self.rank, self.world, self.K, self.dst = rank, world, pieces, dst
        self.nshards = world * pieces
        size = lambda s: span_off[s + 1] - span_off[s]
        self.mx = [max(max(size(r * pieces + h) for r in range(world)), 1) for h in range(pieces)]
        self.base = [0]
        for h in range(pieces):
            self.base.append(self.base[-1] + world * self.mx[h])
        # shard s = r*K + h lives at base[h] + r*mx[h] of the staging buffer
        self.block_off = [self.base[s % pieces] + (s // pieces) * self.mx[s % pieces] for s in range(self.nshards)]
        self.outs = [torch.empty(self.mx[h], dtype=dtype, device=device) for h in range(pieces)]
        self.stage = torch.empty(self.base[-1], dtype=dtype, device=device) if rank == dst else None
        self.works = []

    def shard(self, h):
        return self.rank * self.K + h

    def out(self, h):
        return self.outs[h]

    def submit(self, h):
        send = self.outs[h]
        if self.world == 1 and not dist.is_initialized():
            self.stage[self.base[h] : self.base[h] + self.mx[h]].copy_(send)
            return
        parts = None
        if self.rank == self.dst:
            parts = [self.stage[self.base[h] + r * self.mx[h] : self.base[h] + (r + 1) * self.mx[h]] for r in range(self.world)]
        self.works.append(dist.gather(send, gather_list=parts, dst=self.dst, async_op=True))

    def wait(self):
        """Returns after every submitted piece has ARRIVED (w.wait() only orders torch's stream; the host
        is synchronised here so the caller may hand the stage to Context.unpermute_blocks_device, which runs
        on the library's own stream, or overwrite the piece buffers)."""
        for w in self.works:
            w.wait()
        self.works = []
        _host_sync(self.outs[0] if self.outs else None)
        return (self.stage, self.block_off) if self.rank == self.dst else None


# What a kernel that WAITS like an RCCL receive kernel costs the destination's own kernels (tools/interference_probe.py,
# profiles/rd6a/interference_probe.jsonl: BASELINE configs[2] over 8 ranks, the destination's job, 7 ... 28 waiting
# workgroups): holding more LDS than the tile kernel leaves free on a CU (> 32 KB: it then owns a CU) the tile kernel
# takes 1.12-1.14x and k_finalize 1.14-1.30x as long; holding 4 KB (co-resident) 0.98-1.05x and 1.17-1.30x.
MEASURED_RECV_INTERFERENCE = {"lds_over_32k": {"pair": 1.13, "finalize": 1.17}, "lds_4k": {"pair": 1.02, "finalize": 1.17}}


def pipeline_model(rows, place_rate, link_gbs, round_ms=0.02, place_launch_ms=0.005, nmsg=8, dst_gate=None, dst_interference=None):
    """The N-rank step predicted from per-rank compute times (tools/shard_model.py measures them on one GPU, bench.py
    --gpus N on the ranks themselves), following what dsh_exchange_collect_async does: the destination (rows[0]) receives
    in `nmsg` ROUNDS -- one grouped ncclSend/ncclRecv per round: message q of every source, the q-th nmsg-th of its
    buffer, every source over its own xGMI link at `link_gbs`.  A source's message is ready when the part that holds its
    last value is final; a round starts when the previous one has arrived and every message in it is ready, and lasts as
    long as its largest message + `round_ms`.  What a round completes of the row-sorted sources' rows is put into place
    by ONE launch on a stream of its own (`place_rate` bytes/s + `place_launch_ms`), beside the next round's transfer.
    rows[r]: rank, wall_ms, rowsorted, and either part_info = [(ready_ms, bytes)] -- when every part was final, measured
    on the device (dsh_last_part_info; shifted so that the last part ends with the rank's wall) -- or the older estimate
    from prepare_ms, pair_ms, finalize_ms, parts, bands, span_bytes: part q ready after prepare, the tile kernel and
    (q+1)/parts of k_finalize.
    dst_gate: None = as the library decides (a destination whose job is one launch of at most 8 rounds posts its receives
    behind its tile kernel, so that no RCCL kernel spins beside it; option xch_recv_gate), True / False to force.
    dst_interference: None (the assumption of rounds 3-5: receive kernels waiting beside the destination's kernels cost
    them nothing) or {"pair": f, "finalize": f} (e.g. MEASURED_RECV_INTERFERENCE[...]): the destination's tile kernel --
    unless its receives are gated behind it -- and its k_finalize take f times as long.
    Returns (step ms, the rank that bounds it; 0: the destination's own compute)."""
    step_ms, worst = rows[0]["wall_ms"] if rows else 0.0, 0
    gate_ms = 0.0
    if rows and "finalize_ms" in rows[0]:
        short = rows[0].get("bands", 1) == 1 and rows[0].get("rounds_of_512", 99) <= 8
        gated = bool(dst_gate or (dst_gate is None and short))
        if dst_interference and len(rows) > 1:
            extra = rows[0]["finalize_ms"] * (dst_interference.get("finalize", 1.0) - 1.0)
            if not gated:
                extra += rows[0].get("pair_ms", 0.0) * (dst_interference.get("pair", 1.0) - 1.0)
            step_ms += extra
        if gated:
            gate_ms = max(0.0, rows[0]["wall_ms"] - rows[0]["finalize_ms"])  # (the end of its tile kernel)
    srcs = []
    for x in rows[1:]:
        step_ms = max(step_ms, x["wall_ms"])
        parts = x.get("part_info")
        if parts:
            shift = max(0.0, x["wall_ms"] - max(r for r, _ in parts))  # (host time of the call: counted in front of the kernels)
            parts = [(r + shift, b) for r, b in parts]
        else:
            k = max(x["parts"], 1)
            parts = []
            for q in range(k):
                if x["bands"] >= k > 1:  # the tile kernel is cut per part
                    ready = x["prepare_ms"] + (x["pair_ms"] + x["finalize_ms"]) * (q + 1) / k
                else:
                    ready = x["prepare_ms"] + x["pair_ms"] + x["finalize_ms"] * (q + 1) / k
                parts.append((max(ready, x["wall_ms"]) if q == k - 1 else ready, x["span_bytes"] / k))
        total = sum(b for _, b in parts)
        if not total:
            continue
        ends, acc, latest = [], 0.0, 0.0
        for r, b in parts:
            acc += b
            latest = max(latest, r)  # (a message waits for every part up to the one that holds its last value)
            ends.append((acc, latest))
        msgs = []
        for q in range(nmsg):
            end = total * (q + 1) / nmsg
            ready = next((r for e, r in ends if e >= end - 0.5), ends[-1][1])
            msgs.append((ready, total / nmsg))
        srcs.append((x["rank"], bool(x["rowsorted"]), msgs))
    if not srcs:
        return step_ms, worst
    arrived, placed = gate_ms, 0.0
    gate = 0
    for q in range(nmsg):
        ready_rank, ready = max(((rk, m[q][0]) for rk, _, m in srcs), key=lambda t: t[1])
        big_rank, big = max(((rk, m[q][1]) for rk, _, m in srcs), key=lambda t: t[1])
        gate = ready_rank if ready > arrived else big_rank
        arrived = max(arrived, ready) + big / (link_gbs * 1e9) * 1e3 + round_ms
        staged = sum(m[q][1] for _, rs, m in srcs if rs)
        placed = max(placed, arrived) + (staged / place_rate * 1e3 + place_launch_ms if staged and place_rate else 0.0)
    done = max(arrived, placed)
    if done > step_ms:
        step_ms, worst = done, gate
    return step_ms, worst

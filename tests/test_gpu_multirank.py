"""GPU: the N-rank path end to end on real kernels (sketch shares -> all-gather of registers ->
sharded all-pairs -> gather -> un-permute), see tests/e2e_multigpu_worker.py.  A 1-GPU box can run
(a) RCCL with a single rank -- every collective is issued through the nccl backend -- and
(b) two and three ranks sharing cuda:0 over gloo.  Real multi-GPU RCCL runs only in the driver's bench."""
import os
import socket
import uuid
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
WORKER = os.path.join(ROOT, "tests", "e2e_multigpu_worker.py")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("backend,world,pieces", [("nccl", 1, 0), ("gloo", 2, 0), ("gloo", 3, 0), ("nccl", 1, 1), ("gloo", 2, 1), ("gloo", 3, 1), ("nccl", 1, 2), ("gloo", 2, 3)])
def test_end_to_end_ranks(backend, world, pieces):
    """pieces == 0: row ranges of the final triangle, point-to-point into place (bench.py's N>1 path).
    pieces > 1: every rank's share is cut into that many shards, each gathered asynchronously while the
    next is computed (multigpu.PipelinedShards, dsh_unpermute_blocks_device)."""
    env = dict(os.environ, E2E_BACKEND=backend, E2E_PIECES=str(pieces), HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), WORKER]
    r = subprocess.run(cmd, env=env, capture_output=True, timeout=600, cwd=ROOT)
    out = r.stdout.decode() + r.stderr.decode()
    assert r.returncode == 0, out[-3000:]
    assert "E2E_OK world=%d backend=%s pieces=%d" % (world, backend, pieces) in out, out[-3000:]


@pytest.mark.parametrize("mode", ["1", "parts"])
def test_end_to_end_cabi_exchange_single_rank_rccl(mode):
    """the same path with the exchange done by the library's own RCCL communicator (dsh_comm_init, dsh_collect_spans or the
    pipelined dsh_dist_rows_parts_device_async + dsh_collect_parts_async, dsh_allgather_device) next to torch.distributed"""
    env = dict(os.environ, E2E_BACKEND="nccl", E2E_PIECES="0", E2E_CABI=mode, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), WORKER]
    r = subprocess.run(cmd, env=env, capture_output=True, timeout=600, cwd=ROOT)
    out = r.stdout.decode() + r.stderr.decode()
    assert r.returncode == 0 and "E2E_OK world=1 backend=nccl pieces=0" in out, out[-3000:]


def test_cabi_comm_in_process(ctx):
    """dsh_comm_* without torch.distributed (what the C++ CLI does): unique id -> communicator of one rank ->
    dsh_dist_collect == dsh_dist_rows; spans computed out of place land in the final buffer; a second init replaces the
    communicator; calls before dsh_comm_init fail with DSH_ESTATE"""
    import numpy as np
    import torch

    import dashing_amd
    from dashing_amd import synth

    n, p = 500, 12
    regs = synth.synthetic_sketches(n, p, seed=8)
    ctx.set_sketches(regs)
    want = ctx.dist_rows()
    assert ctx.comm_rank() is None
    with pytest.raises(dashing_amd.DshError):
        ctx.allgather_device(0, 0, 0)
    uid = dashing_amd.comm_unique_id()
    assert len(uid) == 128
    ctx.comm_init(uid, 0, 1)
    try:
        assert ctx.comm_rank() == (0, 1)
        assert ctx.dist_collect([0, n]).tobytes() == want.tobytes()
        dev = torch.device("cuda", 0)
        local = torch.empty(want.size, dtype=torch.float32, device=dev)
        final = torch.full((want.size,), -1.0, dtype=torch.float32, device=dev)
        torch.cuda.synchronize()
        ctx.dist_rows_device_async(local.data_ptr(), 0, n)
        ctx.collect_spans(n, [0, n], local.data_ptr(), final.data_ptr(), 0, wait=False)  # enqueued behind the kernels
        ctx.wait()
        assert final.cpu().numpy().tobytes() == want.tobytes()
        blk = torch.arange(4096, dtype=torch.uint8, device=dev)
        got = torch.zeros(4096, dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        ctx.allgather_device(blk.data_ptr(), 4096, got.data_ptr())
        assert torch.equal(blk, got)
        with pytest.raises(dashing_amd.DshError):
            ctx.collect_spans(n, [0, n - 1], local.data_ptr(), final.data_ptr(), 0)  # bounds must end at n
        ctx.comm_init(dashing_amd.comm_unique_id(), 0, 1)  # re-initialisation replaces the communicator
        assert ctx.dist_collect([0, n]).tobytes() == want.tobytes()
    finally:
        ctx.comm_destroy()
    assert ctx.comm_rank() is None and np.isfinite(ctx.dist_rows()).all()


def _bench(args, env_extra, timeout=900):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **env_extra)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True,
                          timeout=timeout, cwd=ROOT)


def test_bench_gpus_flag_launches_ranks():
    """`python bench.py --gpus 2` with no launcher around it starts two ranks by itself.  On the
    one-GPU box the two ranks share cuda:0 over gloo (DSH_BENCH_BACKEND=gloo, a dry run of the N-rank code path)."""
    import json

    r = _bench(["--gpus", "2", "--steps", "2", "--warmup", "1"], {"DSH_BENCH_BACKEND": "gloo", "DSH_BENCH_N": "3000"})
    out = r.stdout.decode()
    assert r.returncode == 0, (out + r.stderr.decode())[-3000:]
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out[-3000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 2 and line["scaling"] == "strong"
    mg = line["multi_gpu"]
    assert mg["ranks"] == 2 and mg["backend"] == "gloo" and len(mg["row_bounds"]) == 3
    assert line["parity_vs_cpu"]["assembled_equals_single_gpu"] is True
    for key in ("compute_incl_prepare", "exchange", "k_pair_counts", "k_finalize", "prepare"):
        assert mg["phase_ms_max_over_ranks"][key] >= 0.0


def test_bench_gpus_flag_refuses_missing_devices():
    """more ranks than devices over RCCL: non-zero exit, no JSON line -- never a silent 1-GPU run"""
    import dashing_amd

    want = dashing_amd.device_count() + 1
    r = _bench(["--gpus", str(want), "--steps", "1", "--warmup", "0"], {"DSH_BENCH_BACKEND": "nccl"}, timeout=120)
    assert r.returncode != 0
    assert not [l for l in r.stdout.decode().splitlines() if l.startswith("{")]
    assert "needs %d visible" % want in r.stderr.decode()


def test_bench_measures_hbm_traffic_itself():
    """roofline.traffic of the default bench line comes from two rocprofv3 --pmc child passes of the run itself (not only
    from the committed profiles/pmc_pair_kernel.json): non-null, of the size the tile kernel is known to move, and within
    a few percent of the algorithm-independent floor-to-ceiling window (compulsory bytes .. streaming-model bytes)"""
    import json
    import shutil

    if shutil.which("rocprofv3") is None:
        pytest.skip("rocprofv3 not installed")
    r = _bench(["--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-secondary"], {"DSH_BENCH_N": "4000"})
    out = r.stdout.decode()
    assert r.returncode == 0, (out + r.stderr.decode())[-3000:]
    line = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
    rf = line["roofline"]
    assert rf["traffic"] is not None and "measured in this run" in rf["traffic_note"], rf["traffic_note"]
    n, p = 4000, 14
    compulsory = n * (1 << p)                       # every register read once
    streaming = (n * (n - 1) // 2) * (2 * (1 << p) + 4)
    assert compulsory < rf["traffic"] < streaming
    # --no-pmc: the hash-checked file or null, never a child pass
    r = _bench(["--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-secondary", "--no-pmc"], {"DSH_BENCH_N": "4000"})
    line = json.loads([l for l in r.stdout.decode().splitlines() if l.startswith("{")][-1])
    assert "measured in this run" not in line["roofline"]["traffic_note"]


def test_bench_falls_back_when_the_library_cannot_load_rccl():
    """with the C-ABI communicator unavailable (DSH_RCCL_LIB points nowhere) the N-rank code path of
    bench.py (forced with one rank) still prints ONE valid JSON line: the exchange that actually ran (torch.distributed),
    the reason, the library the loader tried; and with RCCL loadable the same command uses the C-ABI exchange and names
    the resolved librccl and its version."""
    import json

    common = {"DSH_BENCH_FORCE_DIST": "1", "DSH_BENCH_N": "3000", "MASTER_PORT": "29533"}
    common["MASTER_PORT"] = str(_free_port())
    r = _bench(["--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-secondary", "--no-pmc"], dict(common, DSH_RCCL_LIB="/nonexistent/librccl.so"))
    out = r.stdout.decode()
    assert r.returncode == 0, (out + r.stderr.decode())[-3000:]
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out[-3000:]
    mg = json.loads(lines[0])["multi_gpu"]
    assert mg["exchange"] == "torch.distributed" and mg["rccl_ranks"] == 1
    assert mg["exchange_library"]["available_on_rank0"] is False and "cabi_fallback_reason" in mg["exchange_library"]
    assert json.loads(lines[0])["parity_vs_cpu"]["assembled_equals_single_gpu"] is True
    r = _bench(["--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-secondary", "--no-pmc"], dict(common, MASTER_PORT=str(_free_port())))
    out = r.stdout.decode()
    assert r.returncode == 0, (out + r.stderr.decode())[-3000:]
    line = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
    mg = line["multi_gpu"]
    assert mg["exchange"].startswith("c-abi rccl") and mg["exchange_library"]["nccl_version_code"] > 0
    assert "librccl" in mg["exchange_library"]["library"]
    assert line["parity_vs_cpu"]["assembled_equals_single_gpu"] is True


def test_parts_of_short_ranges_get_their_events(ctx):
    import dashing_amd
    from dashing_amd import synth

    _parts_of_short_ranges(ctx, dashing_amd, synth)


def _parts_of_short_ranges(ctx, dashing_amd, synth):
    """the ranks of an 8-way run at n = 10 000 hold 640-896 rows -- below range_sort_min_rows -- and a
    rank may hold a single part.  A call with parts must still lay its range out in exactly dsh_range_parts' parts and
    mark every one, or dsh_collect_parts_async refuses (and the peers hang).  Same values as the plain call."""
    import torch

    n, p = 2600, 12
    regs = torch.from_numpy(synth.survey_sketches(n, p, seed=11)[0]).cuda()
    ctx.attach_device(regs.data_ptr(), n, p)
    want = torch.empty(n * (n - 1) // 2, dtype=torch.float32, device="cuda")
    ctx.dist_rows_device(want.data_ptr(), 0, n)
    ctx.synchronize()
    bounds = [0, 640, 1280, 1290, 1290, 2048, 2599, n]
    for nparts in (1, 3, 8):
        for r in range(len(bounds) - 1):
            rb, re = bounds[r], bounds[r + 1]
            span = dashing_amd.tri_span(n, rb, re)
            got = torch.full((max(span, 1),), -1.0, dtype=torch.float32, device="cuda")
            ctx.dist_rows_parts_device_async(got.data_ptr(), rb, re, nparts)
            ctx.synchronize()
            parts = dashing_amd.range_parts(n, rb, re, nparts)
            assert ctx.info("parts_done") == (len(parts) - 1 if rb < re else 0), (rb, re, nparts, parts)
            off = dashing_amd.tri_span(n, 0, rb)
            assert torch.equal(got[:span], want[off:off + span]), (rb, re, nparts)
    # one rank, full range: the collect call accepts what the compute call left
    out = torch.empty_like(want)
    ctx.dist_rows_parts_device_async(out.data_ptr(), 0, n, 4)
    ctx.collect_parts_async(n, [0, n], 4, 0, out.data_ptr(), 0)
    ctx.comm_wait()
    assert torch.equal(out, want)


def test_exchange_pair_virtual_ranks_row_sorted_parts(ctx):
    """dsh_exchange_*: every rank's buffer laid out for the exchange.  Virtual ranks on one GPU: each rank's rows computed
    with dsh_exchange_rows_device_async (short ranges in row-sorted parts: the buffer holds the rows in key order), then
    handed to dsh_exchange_place_device, which does what the destination does with a received buffer.  The assembled
    matrix must equal the single-GPU one byte for byte; worlds 1..8, destinations first / middle / last, parts 1..8."""
    import torch

    import dashing_amd
    from dashing_amd import synth

    n, p = 2600, 12
    regs = torch.from_numpy(synth.survey_sketches(n, p, seed=13)[0]).cuda()
    ctx.attach_device(regs.data_ptr(), n, p)
    total = n * (n - 1) // 2
    want = torch.empty(total, dtype=torch.float32, device="cuda")
    ctx.dist_rows_device(want.data_ptr(), 0, n)
    ctx.synchronize()
    seen_rowsorted = seen_plain = False
    for world, dst, nparts in ((1, 0, 4), (2, 0, 8), (3, 1, 2), (8, 0, 8), (8, 7, 3), (5, 2, 1)):
        bounds = dashing_amd.balance_rows(n, world)
        final = torch.full((total,), -7.0, dtype=torch.float32, device="cuda")
        order = [dst] + [r for r in range(world) if r != dst]  # (the destination's per-sketch pass comes first, as in a real run)
        for r in order:
            rs, k = dashing_amd.exchange_mode(n, bounds, r, nparts, dst)
            span = dashing_amd.tri_span(n, bounds[r], bounds[r + 1])
            if r == dst:
                assert not rs and k == (1 if bounds[r] < bounds[r + 1] else 0)
                local = final[dashing_amd.tri_span(n, 0, bounds[r]):]
            else:
                local = torch.full((max(span, 1),), -3.0, dtype=torch.float32, device="cuda")
            seen_rowsorted |= rs
            seen_plain |= (not rs and r != dst and span > 0)
            ctx.attach_device(regs.data_ptr(), n, p)  # a rank starts from the registers alone
            ctx.exchange_rows_device_async(local.data_ptr(), bounds, r, nparts, dst)
            ctx.synchronize()
            # (a row-sorted source whose parts announce themselves from inside k_finalize cuts as finely as its rows complete:
            # every tile row a part -- the messages of the exchange are shares of its buffer, not parts)
            assert ctx.info("parts_done") >= k if rs else ctx.info("parts_done") == k, (world, dst, nparts, r)
            if rs and span:  # key order, not the final span
                off = dashing_amd.tri_span(n, 0, bounds[r])
                assert not torch.equal(local[:span], want[off:off + span])
                assert torch.equal(torch.sort(local[:span])[0], torch.sort(want[off:off + span])[0])
            if r != dst:
                ctx.exchange_place_device(bounds, r, nparts, local.data_ptr(), final.data_ptr(), dst)
        assert torch.equal(final, want), (world, dst, nparts)
    assert seen_rowsorted
    # one rank, no communicator: the collect call accepts what the compute call left
    out = torch.empty_like(want)
    ctx.attach_device(regs.data_ptr(), n, p)
    ctx.exchange_rows_device_async(out.data_ptr(), [0, n], 0, 4, 0)
    ctx.exchange_collect_async(n, [0, n], 4, 0, out.data_ptr(), 0)
    ctx.comm_wait()
    assert torch.equal(out, want)
    # a long range keeps parts of consecutive rows (received in place): n large enough that 1024 rows per part are exceeded
    rs, k = dashing_amd.exchange_mode(100000, dashing_amd.balance_rows(100000, 2), 1, 2, 0)
    assert not rs and k == 2


def test_exchange_pair_virtual_ranks_row_sets(ctx):
    """the same with dsh_balance_rowsets' tables: ranges + top-up tile rows.  Every rank's pairs are its own (sorted values
    equal those of its rows in the single-GPU matrix), the tiles it computes are the table's count, and the destination's
    placement assembles the single-GPU matrix byte for byte; worlds 2..8, destinations first / middle / last."""
    import torch

    import dashing_amd
    from dashing_amd import synth

    n, p = 3300, 12
    regs = torch.from_numpy(synth.survey_sketches(n, p, seed=14)[0]).cuda()
    ctx.attach_device(regs.data_ptr(), n, p)
    total = n * (n - 1) // 2
    want = torch.empty(total, dtype=torch.float32, device="cuda")
    ctx.dist_rows_device(want.data_ptr(), 0, n)
    ctx.synchronize()
    topups = 0
    for world, dst, nparts in ((2, 0, 8), (3, 1, 2), (4, 3, 4), (8, 0, 8), (8, 7, 3), (5, 2, 1)):
        rows = dashing_amd.balance_rowsets(n, world)
        final = torch.full((total,), -7.0, dtype=torch.float32, device="cuda")
        order = [dst] + [r for r in range(world) if r != dst]
        tiles = 0
        for r in order:
            segs = rows.rows(r)
            topups += len(segs) > 1
            rs, k, floats = dashing_amd.exchange_mode(n, rows, r, nparts, dst, want_floats=True)
            if not segs:
                continue
            if r == dst:
                assert not rs and k == 1
                local = final[dashing_amd.tri_span(n, 0, segs[0][0]):]
            else:
                assert floats == rows.pairs(r)
                assert rs == (len(segs) > 1 or (nparts >= 2 and segs[0][1] - segs[0][0] < 1024 * nparts))
                local = torch.full((max(floats, 1),), -3.0, dtype=torch.float32, device="cuda")
            ctx.attach_device(regs.data_ptr(), n, p)
            ctx.exchange_rows_device_async(local.data_ptr(), rows, r, nparts, dst)
            ctx.synchronize()
            assert (ctx.info("parts_done") >= k if rs else ctx.info("parts_done") == k) and ctx.info("tiles") == rows.tiles(r), (world, dst, nparts, r)
            assert ctx.info("parts_signalled") == 1  # (the default on gfx950: flags from inside one launch per band)
            tiles += rows.tiles(r)
            if r != dst:
                mine = torch.cat([want[dashing_amd.tri_span(n, 0, b):dashing_amd.tri_span(n, 0, e)] for b, e in segs])
                assert torch.equal(torch.sort(local[:floats])[0], torch.sort(mine)[0]), (world, dst, nparts, r)
                ctx.exchange_place_device(rows, r, nparts, local.data_ptr(), final.data_ptr(), dst)
        assert torch.equal(final, want), (world, dst, nparts)
        nt = (n + 127) // 128
        assert tiles == nt * (nt + 1) // 2  # nothing computed twice, nothing for another rank's rows
    assert topups >= 6, "these cases are meant to exercise top-up segments"
    # dsh_last_part_info (what the pipeline model of bench.py / tools/shard_model.py is fed with): under profiling every part
    # of a source rank reports when it was final and how many bytes of the buffer it holds; the bytes add up to the buffer
    rows = dashing_amd.balance_rowsets(n, 4)
    rs, k, floats = dashing_amd.exchange_mode(n, rows, 1, 4, 0, want_floats=True)
    local = torch.empty(floats, dtype=torch.float32, device="cuda")
    ctx.set_profiling(True)
    ctx.attach_device(regs.data_ptr(), n, p)
    ctx.exchange_rows_device_async(local.data_ptr(), rows, 1, 4, 0)
    ctx.synchronize()
    info = ctx.last_part_info()
    ctx.set_profiling(False)
    assert len(info) >= k >= 2 and sum(b for _, b in info) == 4 * floats
    assert all(ms > 0 for ms, _ in info) and info[-1][0] == max(ms for ms, _ in info)


MOCK = os.path.join(ROOT, "tests", "mock_rccl", "libmock_rccl.so")
MOCK_WORKER = os.path.join(ROOT, "tests", "mock_exchange_worker.py")


def run_mock_world(tmp_path, world, n, p, nparts, mode, dst=0, bounds=None, timeout=900, opts="", rowsets=False, expect_topups=False):
    """`world` processes on the one GPU, the library's RCCL calls served by tests/mock_rccl (messages as files, matched
    by order, peer and exact size)"""
    if not os.path.exists(MOCK):
        subprocess.check_call(["make", "-s", "-C", os.path.dirname(MOCK)])
    env = dict(os.environ, DSH_RCCL_LIB=MOCK, WORLD=str(world), N=str(n), P=str(p), NPARTS=str(nparts), MODE=mode, DST=str(dst),
               ID_FILE=str(tmp_path / ("id_" + uuid.uuid4().hex)), MOCK_RCCL_TIMEOUT_S="240", DSH_COMM_TIMEOUT_S="300",  # (a file per run)
               HSA_ENABLE_IPC_MODE_LEGACY="0", OPTS=opts)
    if bounds:
        env["BOUNDS"] = ",".join(str(b) for b in bounds)
    if rowsets:
        env["ROWSETS"] = "1"
    if expect_topups:
        env["EXPECT_TOPUPS"] = "1"
    procs = [subprocess.Popen([sys.executable, MOCK_WORKER], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in range(world)]
    outs = []
    try:
        for pr in procs:
            outs.append(pr.communicate(timeout=timeout)[0])
    finally:
        for pr in procs:
            if pr.poll() is None:
                pr.kill()
    for r, (pr, out) in enumerate(zip(procs, outs)):
        assert pr.returncode == 0 and "MOCK_EXCHANGE_OK rank %d" % r in out, "rank %d:\n%s" % (r, out[-3000:])


@pytest.mark.gpu
@pytest.mark.parametrize("world,n,p,nparts,mode", [
    (3, 3000, 12, 4, "exchange"),    # short ranges: row-sorted parts, staged and placed row by row on the destination
    (8, 2500, 10, 8, "exchange"),    # the 8 ranks of the driver's scaling run (ranges of 1-2 tile rows, some with ONE part)
    (2, 9000, 10, 2, "exchange"),    # long ranges (>= 1024 rows per part): parts of consecutive rows, received in place
    (3, 2000, 12, 3, "parts"),       # dsh_dist_rows_parts_device_async + dsh_collect_parts_async
    (4, 1500, 12, 1, "spans"),       # dsh_collect_spans: one message per peer
    (3, 1200, 10, 1, "collect"),     # dsh_dist_collect, the whole step for a host without device pointers
    (3, 700, 10, 1, "allgather"),    # the register arrays after sharded sketching
])
def test_exchange_protocol_between_processes(tmp_path, world, n, p, nparts, mode):
    """VERDICT r3 (weak 7): the world > 1 half of the exchange had never executed anywhere -- RCCL refuses two ranks on one
    device and the build has one GPU.  Here it does, between real processes, over a stand-in transport that is STRICTER
    than RCCL about what the protocol must get right (a receive whose size differs from the matching send is an error,
    a missing peer a timeout): the destination's matrix equals the single-GPU one byte for byte."""
    run_mock_world(tmp_path, world, n, p, nparts, mode)


@pytest.mark.gpu
def test_exchange_protocol_other_destination_and_ragged_ranges(tmp_path):
    """the destination need not be rank 0, a rank may own no rows at all, another a single row"""
    run_mock_world(tmp_path, 4, 1700, 12, 3, "exchange", dst=2, bounds=[0, 0, 640, 1699, 1700])
    run_mock_world(tmp_path, 3, 1700, 12, 2, "parts", dst=1, bounds=[0, 900, 900, 1700])
    # the DESTINATION owns no rows while the sources are row-sorted -- it used to return before any per-sketch
    # pass and fail in the collect with the peers' sends already posted
    run_mock_world(tmp_path, 3, 1700, 12, 2, "exchange", dst=0, bounds=[0, 0, 900, 1700])


@pytest.mark.gpu
@pytest.mark.parametrize("world,n,p,nparts,dst", [
    (8, 4000, 10, 8, 0),   # the shape of the driver's 8-rank run, scaled: ranges of 2-6 tile rows + top-ups
    (4, 3000, 12, 3, 2),   # the destination itself holds top-up segments (computed in place, in final order)
    (3, 1400, 12, 8, 1),   # more parts asked for than a rank has tile rows
    (3, 6000, 10, 8, 1),   # a row-sorted range that ends in a run of its last rows (plan::rowsorted_split)
])
def test_exchange_protocol_row_sets_between_processes(tmp_path, world, n, p, nparts, dst):
    """the balanced partition -- a rank's rows are a range plus top-up tile rows from the bottom of the
    triangle (dsh_balance_rowsets) -- through the real exchange code between processes: every source row-sorted (each
    segment one key-ordered run), staged and placed row by row; the destination's matrix equals the single-GPU one."""
    run_mock_world(tmp_path, world, n, p, nparts, "exchange", dst=dst, rowsets=True, expect_topups=True)


@pytest.mark.gpu
def test_exchange_protocol_row_sets_edge_shapes(tmp_path):
    """row-set tables at the edges: more ranks than tile rows (ranks without rows, contiguous fall-back), ONE part per rank
    with top-up segments, 32-bit C(v) counts (p = 16), a collection just above the top-up limit (plain ranges again)"""
    run_mock_world(tmp_path, 8, 700, 10, 4, "exchange", dst=3, rowsets=True)           # 6 tile rows over 8 ranks
    run_mock_world(tmp_path, 4, 2600, 12, 1, "exchange", dst=0, rowsets=True, expect_topups=True)  # one part each
    run_mock_world(tmp_path, 3, 1100, 16, 3, "exchange", dst=2, rowsets=True, expect_topups=True)  # p = 16
    run_mock_world(tmp_path, 2, 33000, 8, 4, "exchange", dst=1, rowsets=True)          # n > 32 768: contiguous ranges


@pytest.mark.gpu
def test_exchange_protocol_row_sorted_range_in_two_runs(tmp_path):
    """2 ranks: the source's range reaches to the bottom of the triangle and is key-ordered as two runs (its last rows on
    their own, plan::rowsorted_split); the destination derives the same order from its own keys"""
    run_mock_world(tmp_path, 2, 5200, 10, 8, "exchange", dst=0, rowsets=True)
    run_mock_world(tmp_path, 2, 5200, 12, 5, "exchange", dst=1, rowsets=True)


@pytest.mark.gpu
def test_dist_collect_with_the_librarys_own_partition(tmp_path):
    """dsh_dist_collect(bounds = NULL): balanced row sets + the pipelined exchange pair, for a host without device pointers"""
    run_mock_world(tmp_path, 4, 2200, 10, 1, "collect-auto")
    run_mock_world(tmp_path, 3, 1500, 12, 1, "collect-auto", dst=2)
    run_mock_world(tmp_path, 2, 1, 10, 1, "collect-auto")  # one genome over two devices: an empty matrix, not an error
    run_mock_world(tmp_path, 3, 2, 10, 1, "collect-auto", dst=1)  # one pair, more ranks than rows


@pytest.mark.gpu
def test_bench_two_ranks_run_the_cabi_exchange_over_the_stand_in(tmp_path):
    """`python bench.py --gpus 2`, the multi-rank step of the driver's scaling run (dsh_exchange_rows_device_async +
    dsh_exchange_collect_async + dsh_comm_wait per step), on the one-GPU box: two ranks on cuda:0, the library's RCCL calls
    served by tests/mock_rccl.  The line names the exchange that ran and the assembled matrix equals the single-GPU one."""
    import json

    if not os.path.exists(MOCK):
        subprocess.check_call(["make", "-s", "-C", os.path.dirname(MOCK)])
    r = _bench(["--gpus", "2", "--steps", "3", "--warmup", "1"],
               {"DSH_BENCH_BACKEND": "gloo", "DSH_BENCH_EXCHANGE": "cabi-mock", "DSH_RCCL_LIB": MOCK, "DSH_BENCH_N": "3000",
                "MOCK_RCCL_TIMEOUT_S": "240", "DSH_BENCH_PING_MB": "4"}, timeout=600)
    out = r.stdout.decode()
    assert r.returncode == 0, (out + r.stderr.decode())[-3000:]
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out[-3000:]
    line = json.loads(lines[0])
    mg = line["multi_gpu"]
    assert mg["ranks"] == 2 and mg["exchange"].startswith("c-abi rccl"), mg
    assert "mock_rccl" in mg["exchange_library"]["library"]
    assert line["parity_vs_cpu"]["assembled_equals_single_gpu"] is True
    # an N > 1 line carries the CPU leg too, checked against the ASSEMBLED matrix
    cpu = line["cpu_baseline"]
    assert cpu is not None and cpu["value"] > 0 and cpu["cores"] >= 1 and cpu["kind"] == "port", cpu
    assert line["parity_vs_cpu"]["rank0_span_vs_cpu"]["max_rel_diff"] <= 1e-6
    # both RCCL copies a process may hold are named, and which one carried the exchange
    xl = mg["exchange_library"]
    assert xl["carried_the_exchange"] == xl["library"] and isinstance(xl["rccl_copies_mapped"], list)
    assert any("mock_rccl" in p_ for p_ in xl["rccl_copies_mapped"])
    # the line is self-diagnosing: the row sets, every rank's phases, the link rate measured through
    # the library's communicator, and the pipeline model's prediction for these times at that rate beside the measurement
    assert [d["rank"] for d in mg["row_sets"]] == [0, 1] and sum(d["pairs"] for d in mg["row_sets"]) == 3000 * 2999 // 2
    assert any(len(d["rows"]) > 1 for d in mg["row_sets"]), "n = 3000 over 2 ranks is meant to exercise a top-up segment"
    assert [r_["rank"] for r_ in mg["per_rank"]] == [0, 1]
    for r_ in mg["per_rank"]:
        assert {"prepare_ms", "pair_ms", "finalize_ms", "wall_ms", "exposed_exchange_ms", "tiles", "items", "rounds_of_512", "part_info"} <= set(r_)
        assert r_["wall_ms"] > 0 and r_["pair_ms"] > 0
    assert len(mg["per_rank"][1]["part_info"]) >= mg["per_rank"][1]["parts"] >= 1  # (a signalling row-sorted rank: a part per tile row)
    ping = mg["link_gbs_measured"]
    assert ping["payload_intact"] is True and ping["single_GBs_min"] > 0 and ping["concurrent_per_link_GBs"] > 0, ping
    model = mg["model"]
    assert set(model["sensitivity_by_assumed_link_GBs"]) == {"30", "45", "60"}
    assert model["at_measured_link_rate"]["step_model_ms"] > 0 and model["measured_over_predicted"] > 0
    assert line["roofline"]["bound"] == "int VALU issue" and 0 < line["roofline"]["frac"] <= 1


@pytest.mark.gpu
def test_bench_probe_falls_back_when_the_exchange_does_not_reproduce_the_matrix(tmp_path):
    """The N-rank bench line is only worth something if its exchange is right, and the driver's multi-GPU run may be the
    first time RCCL carries it: before anything is timed ONE step must reproduce the single-GPU matrix.  A failed probe
    (injected here) first switches the parts' announcement from flags to events and probes again; a second failure hands
    the exchange to torch.distributed with contiguous spans.  Either way every rank takes the same turn, the line says
    what was timed and why, and the assembled matrix equals the single-GPU one."""
    import json

    if not os.path.exists(MOCK):
        subprocess.check_call(["make", "-s", "-C", os.path.dirname(MOCK)])
    for inject, want_path in ((1, "finalize_signal = 0"), (2, "fallback")):
        r = _bench(["--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"],
                   {"DSH_BENCH_BACKEND": "gloo", "DSH_BENCH_EXCHANGE": "cabi-mock", "DSH_RCCL_LIB": MOCK, "DSH_BENCH_N": "2000",
                    "MOCK_RCCL_TIMEOUT_S": "240", "DSH_BENCH_NO_PING": "1", "DSH_BENCH_INJECT_PROBE_FAIL": str(inject)}, timeout=600)
        out = r.stdout.decode()
        assert r.returncode == 0, (out + r.stderr.decode())[-3000:]
        line = json.loads([l for l in out.splitlines() if l.startswith("{")][0])
        pr = line["multi_gpu"]["probe"]
        assert want_path in pr["timed_path"], pr
        assert len(pr["attempts"]) == inject + (1 if inject == 1 else 0) and not pr["attempts"][0]["ok_on_every_rank"], pr
        assert line["parity_vs_cpu"]["assembled_equals_single_gpu"] is True
        assert line["value"] > 0
        if inject == 2:
            assert line["multi_gpu"]["exchange"].startswith("gloo") and line["multi_gpu"]["row_sets"] is None


@pytest.mark.gpu
def test_exchange_protocol_with_finalize_on_one_stream_and_cut_bands(tmp_path):
    """By default (round 5) a band is finalized by ONE launch and the parts announce themselves from inside it (flags the
    copy stream waits for with hipStreamWaitValue32); finalize_signal = 0 is the scheme for devices without stream
    wait-value -- one launch and one event per part.  Bands cut per part (part_band_tiles) put several tile-kernel
    launches between them.  The exchange sees the same parts either way."""
    run_mock_world(tmp_path, 3, 3000, 12, 4, "exchange", opts="finalize_signal=0")
    run_mock_world(tmp_path, 4, 3000, 12, 3, "exchange", dst=2, rowsets=True, expect_topups=True, opts="finalize_signal=0")
    run_mock_world(tmp_path, 2, 9000, 10, 4, "parts", opts="finalize_signal=0,part_band_tiles=200")
    run_mock_world(tmp_path, 2, 9000, 10, 4, "parts", opts="part_band_tiles=200")  # flags, bands cut per part
    # the destination posts its receives at once, or behind its first tile kernel (auto: short jobs only)
    run_mock_world(tmp_path, 3, 3000, 12, 4, "exchange", rowsets=True, opts="xch_recv_gate=1")
    run_mock_world(tmp_path, 3, 1400, 12, 8, "exchange", dst=1, rowsets=True, opts="xch_recv_gate=0")


@pytest.mark.parametrize("signal", [1, 0])
def test_signalled_parts_are_visible_to_a_kernel_reader_with_inputs_changing_every_step(signal):
    """VERDICT r5 item 2 / ADVICE r5: the parts of a source rank announce themselves from INSIDE the running k_finalize
    (write-through stores, a flag, hipStreamWaitValue32 on the copy stream) and the next thing on the copy stream of a real
    run is an RCCL send KERNEL that reads the part through some XCD's L2.  The stand-in transport reads by copy engine and
    every other test recomputes identical values, so a flag raised before the data, or a stale L2 line of the previous
    step, would pass unseen.  Here: behind every part's gate a copy kernel (dsh_exchange_probe_parts_async: plain loads
    from workgroups on all XCDs, while k_finalize is still running) copies the part into a side buffer, and the register
    matrix CHANGES every step (two collections taken in turn), so that anything left over from the previous step is a
    wrong value.  40 steps, several virtual ranks of an 8-rank plan (row-sorted with top-ups, consecutive rows, the
    destination), both completion schemes (flags from inside one launch / an event per part)."""
    import torch

    import dashing_amd
    from dashing_amd import synth

    n, p, world, nparts, dst = 5200, 12, 8, 8, 0
    mats = [torch.from_numpy(synth.survey_sketches(n, p, seed=s)[0]).cuda() for s in (31, 32)]
    with dashing_amd.Context(0) as ctx:
        ctx.set_option("finalize_signal", signal)
        rows = dashing_amd.balance_rowsets(n, world, -1, dst)
        # what every step must produce: the rank's buffer after a full synchronisation, per collection
        for rank in (1, 4, 7, dst):
            rs, k, floats = dashing_amd.exchange_mode(n, rows, rank, nparts, dst, want_floats=True)
            assert floats > 0
            local = torch.empty(floats, dtype=torch.float32, device="cuda")
            probe = torch.empty(floats, dtype=torch.float32, device="cuda")
            want = []
            for m in mats:
                ctx.attach_device(m.data_ptr(), n, p)
                ctx.exchange_rows_device_async(local.data_ptr(), rows, rank, nparts, dst)
                ctx.synchronize()
                want.append(local.clone())
            assert not torch.equal(want[0], want[1])
            assert ctx.info("parts_signalled") == signal
            for step in range(40):
                which = step & 1
                probe.fill_(-5.0)
                torch.cuda.synchronize()
                ctx.attach_device(mats[which].data_ptr(), n, p)
                ctx.exchange_rows_device_async(local.data_ptr(), rows, rank, nparts, dst)
                ctx.exchange_probe_parts_async(n, rows, rank, nparts, local.data_ptr(), probe.data_ptr(), dst)
                ctx.synchronize()
                bad = int((probe != want[which]).sum().item())
                assert bad == 0, "rank %d step %d (finalize_signal=%d): %d of %d values read behind a part's gate differ from the step's result (stale: %d equal the previous step's)" % (
                    rank, step, signal, bad, floats, int(((probe != want[which]) & (probe == want[1 - which])).sum().item()))
                assert torch.equal(local, want[which])


@pytest.mark.parametrize("signal", [1, 0])
def test_signalled_parts_travel_through_librccl_itself_on_a_communicator_of_one_rank(signal):
    """The same property with librccl as the reader (the real library, not the stand-in): on a communicator of ONE rank
    dsh_exchange_probe_parts_async runs the rank's step of the exchange the way dsh_exchange_collect_async does for a
    source -- nparts messages, each behind the gate of the part that holds its last value, one grouped call per message --
    with the only peer such a communicator has: ncclSend to itself, paired with the ncclRecv into the side buffer.  What
    RCCL has never done here with more than one rank it does with one: take grouped send/recv calls on the library's copy
    stream behind hipStreamWaitValue32 gates while k_finalize is still running, and deliver what the step computed --
    register matrix changing every step, virtual ranks of an 8-rank plan, both completion schemes."""
    import torch

    import dashing_amd
    from dashing_amd import synth

    if os.environ.get("DSH_RCCL_LIB"):
        pytest.skip("DSH_RCCL_LIB names a stand-in: this test is about librccl")
    n, p, world, nparts, dst = 5200, 12, 8, 8, 0
    mats = [torch.from_numpy(synth.survey_sketches(n, p, seed=s)[0]).cuda() for s in (41, 42)]
    with dashing_amd.Context(0) as ctx:
        ctx.set_option("finalize_signal", signal)
        ctx.comm_init(dashing_amd.comm_unique_id(), 0, 1)
        try:
            path, version = dashing_amd.comm_library()
            assert "rccl" in os.path.basename(path) and "mock" not in path and version > 0, (path, version)
            rows = dashing_amd.balance_rowsets(n, world, -1, dst)
            for rank in (1, 5, 7, dst):
                rs, k, floats = dashing_amd.exchange_mode(n, rows, rank, nparts, dst, want_floats=True)
                local = torch.empty(floats, dtype=torch.float32, device="cuda")
                probe = torch.empty(floats, dtype=torch.float32, device="cuda")
                want = []
                for m in mats:
                    ctx.attach_device(m.data_ptr(), n, p)
                    ctx.exchange_rows_device_async(local.data_ptr(), rows, rank, nparts, dst)
                    ctx.synchronize()
                    want.append(local.clone())
                assert not torch.equal(want[0], want[1])
                for step in range(12):
                    which = step & 1
                    probe.fill_(-5.0)
                    torch.cuda.synchronize()
                    ctx.attach_device(mats[which].data_ptr(), n, p)
                    ctx.exchange_rows_device_async(local.data_ptr(), rows, rank, nparts, dst)
                    ctx.exchange_probe_parts_async(n, rows, rank, nparts, local.data_ptr(), probe.data_ptr(), dst)
                    ctx.comm_wait()
                    bad = int((probe != want[which]).sum().item())
                    assert bad == 0, "rank %d step %d (finalize_signal=%d): %d of %d values that librccl delivered differ from the step's result (stale: %d)" % (
                        rank, step, signal, bad, floats, int(((probe != want[which]) & (probe == want[1 - which])).sum().item()))
                    assert torch.equal(local, want[which])
        finally:
            ctx.comm_destroy()


def test_diag_spin_leaves_by_itself_and_results_do_not_change(ctx):
    """dsh_diag_spin_start: workgroups that wait like an RCCL receive kernel beside the library's kernels -- they leave on
    dsh_diag_spin_stop or when max_ms have passed (a host that never comes back cannot hang the GPU), a second start
    without a stop is refused, and a compare call beside them gives the same bytes."""
    import time

    import torch

    import dashing_amd
    from dashing_amd import synth

    n, p = 1500, 12
    regs = torch.from_numpy(synth.survey_sketches(n, p, seed=5)[0]).cuda()
    ctx.attach_device(regs.data_ptr(), n, p)
    want = torch.empty(n * (n - 1) // 2, dtype=torch.float32, device="cuda")
    ctx.dist_rows_device(want.data_ptr(), 0, n)
    ctx.synchronize()
    got = torch.zeros_like(want)
    ctx.diag_spin_start(16, 256, 32768, 3000)
    with pytest.raises(dashing_amd.DshError):
        ctx.diag_spin_start(16, 256, 32768, 3000)
    ctx.attach_device(regs.data_ptr(), n, p)
    ctx.dist_rows_device(got.data_ptr(), 0, n)
    ctx.synchronize()
    ctx.diag_spin_stop()
    assert torch.equal(got, want)
    t0 = time.perf_counter()
    ctx.diag_spin_start(8, 512, 65536, 200)  # nobody stops it: it leaves after 200 ms
    time.sleep(0.5)
    ctx.diag_spin_stop()
    assert time.perf_counter() - t0 < 5.0

"""GPU parity: the HIP compare path (through the C-ABI) vs the CPU oracle on the same inputs.
Bar: distances within 1e-6 relative (|d| <= 1e-6*max(|ref|,1e-9)), the J==0 -> Mash==1 branch
exact, cardinalities within 1e-12 relative (only libm log/log1p/pow ulps may differ)."""
import json
import os

import numpy as np
import pytest

import dashing_amd
from dashing_amd import synth

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
RTOL = 1e-6


def close(got, ref):
    got = np.asarray(got, np.float64)
    ref = np.asarray(ref, np.float64)
    assert got.shape == ref.shape
    fin = np.isfinite(ref)
    assert (np.isfinite(got) == fin).all()
    err = np.abs(got[fin] - ref[fin])
    tol = RTOL * np.maximum(np.abs(ref[fin]), 1e-9)
    bad = err > tol
    assert not bad.any(), "max rel err %.3g at %d of %d" % (
        (err / np.maximum(np.abs(ref[fin]), 1e-9)).max(), int(bad.sum()), err.size)


def test_backend_is_hip(ctx):
    assert dashing_amd.backend_name() == "hip:gfx950"
    assert dashing_amd.device_count() >= 1


def test_golden_pairs(ctx):
    with open(os.path.join(GOLD, "kat.json")) as f:
        kat = json.load(f)
    for row in kat["pairs"]:
        regs = np.load(os.path.join(GOLD, "regs_p%d.npy" % row["p"]))
        ctx.set_sketches(regs)
        want = np.frombuffer(bytes.fromhex(row["tri_hex"]), np.float32)
        got = ctx.dist_rows(estim=row["estim"], result_type=row["result_type"], k=row["k"])
        close(got, want)
        wc = np.frombuffer(bytes.fromhex(row["card_hex"]), np.float64)
        gc = ctx.cardinalities(row["estim"])
        assert np.allclose(gc, wc, rtol=1e-12, atol=0)
        if row["result_type"] == dashing_amd.MASH_DIST:
            assert ((got == 1.0) == (want == 1.0)).all()  # J==0 branch agrees exactly


@pytest.mark.parametrize("p,n", [(10, 1), (10, 2), (10, 127), (10, 128), (10, 129), (10, 300), (14, 130), (14, 260), (15, 40), (12, 97), (7, 70), (4, 33), (5, 40), (16, 20), (16, 140), (17, 6), (18, 5), (19, 4), (19, 131), (20, 5), (22, 3), (24, 3)])
@pytest.mark.parametrize("estim", [0, 1, 2])
def test_tri_vs_oracle(ctx, oracle, p, n, estim):
    regs = synth.synthetic_sketches(n, p, seed=0x1234 + p * 131 + n)
    if n > 3:
        regs[n // 2] = regs[0]      # identical pair -> J == 1
        regs[n - 1] = 0             # empty sketch -> J == 0, Mash == 1
    ctx.set_sketches(regs)
    for rt in (dashing_amd.JI, dashing_amd.MASH_DIST, dashing_amd.FULL_MASH_DIST):
        want = oracle.dist_tri(regs, estim, rt, 31)
        got = ctx.dist_rows(estim=estim, result_type=rt, k=31)
        close(got, want)
        if rt == dashing_amd.MASH_DIST:
            assert ((got == 1.0) == (want == 1.0)).all()
    if n > 3:
        ji = ctx.dist_rows(estim=estim, result_type=dashing_amd.JI)
        assert ji[dashing_amd.tri_index(n, 0, n // 2)] == 1.0


def test_cardinalities_vs_oracle(ctx, oracle):
    for p in (4, 10, 14):
        regs = synth.synthetic_sketches(50, p, seed=p)
        regs[7] = 0
        regs[8] = 64 - p + 1  # saturated
        ctx.set_sketches(regs)
        for e in (0, 1, 2):
            want = oracle.cardinalities(regs, e)
            got = ctx.cardinalities(e)
            fin = np.isfinite(want)
            assert (np.isfinite(got) == fin).all()
            assert np.allclose(got[fin], want[fin], rtol=1e-12, atol=0)


def test_row_ranges_concatenate(ctx):
    """Sharding primitive: any split of the rows gives byte-identical spans (multi-GPU by construction)."""
    n, p = 333, 10
    regs = synth.synthetic_sketches(n, p, seed=42)
    ctx.set_sketches(regs)
    full = ctx.dist_rows()
    for parts in (2, 3, 8):
        b = dashing_amd.partition_rows(n, parts, 128)
        cat = np.concatenate([ctx.dist_rows(b[i], b[i + 1]) for i in range(parts)])
        assert cat.tobytes() == full.tobytes()
    odd = np.concatenate([ctx.dist_rows(0, 1), ctx.dist_rows(1, 70), ctx.dist_rows(70, 71), ctx.dist_rows(71, n)])
    assert odd.tobytes() == full.tobytes()


@pytest.mark.parametrize("n,p", [(333, 10), (2500, 12), (1300, 14)])
def test_row_ranges_key_ordered_layout(ctx, oracle, n, p):
    """Row ranges computed with the plane matrix laid out for the range (wanted rows first, both parts
    key-ordered; results land at their final packed positions): any split -- unaligned, tiny, a single
    row, with or without the layout -- concatenates to the byte-identical full triangle."""
    parts_h = []
    for k, card in enumerate((40_000, 600_000, 9_000_000)):  # heterogeneous: the key order matters
        parts_h += [synth.hll_registers(1000 * k + g, card * (1 + g % 3), p) for g in range(n // 3 + 1)]
    regs = np.stack(parts_h)[:n]
    regs = regs[np.random.default_rng(3).permutation(n)]
    ctx.set_sketches(regs)
    full = ctx.dist_rows(result_type=dashing_amd.MASH_DIST, k=21)
    assert ctx.info("sorted") == 1 and ctx.info("ncols") == n
    for r in (0, n // 2, n - 2):
        want = oracle.dist_rows(regs, r, r + 1, 2, oracle.MASH_DIST, 21)
        lo = dashing_amd.tri_index(n, r, r + 1)
        close(full[lo : lo + want.size], want)
    rng = np.random.default_rng(n)
    try:
        for min_rows in (1024, 1):
            ctx.set_option("range_sort_min_rows", min_rows)
            for parts in (2, 3, 8):
                b = dashing_amd.partition_rows(n, parts, 1)
                cat = np.concatenate([ctx.dist_rows(b[i], b[i + 1], result_type=dashing_amd.MASH_DIST, k=21) for i in range(parts)])
                assert cat.tobytes() == full.tobytes()
            cuts = sorted(set([0, n] + [int(x) for x in rng.integers(1, n, 5)] + [1, 127, 128, 129, n - 1]))
            cat = np.concatenate([ctx.dist_rows(cuts[i], cuts[i + 1], result_type=dashing_amd.MASH_DIST, k=21) for i in range(len(cuts) - 1)])
            assert cat.tobytes() == full.tobytes()
            if min_rows == 1:
                ctx.dist_rows(n // 3, n // 2)
                assert ctx.info("sorted") == 1 and ctx.info("ncols") == n - n // 3  # sketches before the range are left out
        ctx.set_option("sort", 0)
        assert ctx.dist_rows(result_type=dashing_amd.MASH_DIST, k=21).tobytes() == full.tobytes()
    finally:
        ctx.set_option("sort", -1)
        ctx.set_option("range_sort_min_rows", 1024)


@pytest.mark.parametrize("n,p", [(1500, 12), (2100, 10)])
def test_row_range_in_parts_and_pipelined_collect(ctx, n, p):
    """dsh_dist_rows_parts_device_async: a row range computed in parts (each key-ordered on its own, cuts on tile
    rows) is the byte-identical span; dsh_collect_parts_async (here one rank: the copy-into-place rounds, each behind
    its part's event on the copy stream) assembles it in the final buffer."""
    import torch

    parts_h = []
    for k, card in enumerate((60_000, 900_000, 7_000_000)):
        parts_h += [synth.hll_registers(77 * k + g, card * (1 + g % 3), p) for g in range(n // 3 + 1)]
    regs = np.stack(parts_h)[:n]
    regs = regs[np.random.default_rng(5).permutation(n)]
    ctx.set_sketches(regs)
    full = ctx.dist_rows(result_type=dashing_amd.MASH_DIST, k=21)
    dev = torch.device("cuda", 0)
    for rb, re in ((0, n), (256, n), (300, 1400), (0, 130)):
        span = dashing_amd.tri_span(n, rb, re)
        lo = dashing_amd.tri_span(n, 0, rb)
        for nparts in (1, 2, 3, 7):
            pr = dashing_amd.range_parts(n, rb, re, nparts)
            assert pr[0] == rb and pr[-1] == re and all((x - rb) % 128 == 0 for x in pr[1:-1]) and sorted(set(pr)) == pr
            out = torch.full((span,), -3.0, dtype=torch.float32, device=dev)
            torch.cuda.synchronize()
            ctx.dist_rows_parts_device_async(out.data_ptr(), rb, re, nparts, result_type=dashing_amd.MASH_DIST, k=21)
            ctx.wait()
            assert out.cpu().numpy().tobytes() == full[lo : lo + span].tobytes(), (rb, re, nparts)
    # one rank, whole triangle in 4 parts, delivered into a separate final buffer part by part
    total = full.size
    local = torch.full((total,), -1.0, dtype=torch.float32, device=dev)
    final = torch.full((total,), -2.0, dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    ctx.dist_rows_parts_device_async(local.data_ptr(), 0, n, 4, result_type=dashing_amd.MASH_DIST, k=21)
    ctx.collect_parts_async(n, [0, n], 4, local.data_ptr(), final.data_ptr(), 0)
    ctx.wait()
    assert final.cpu().numpy().tobytes() == full.tobytes()
    with pytest.raises(dashing_amd.DshError):  # the parts of the exchange must be the parts that were computed
        ctx.collect_parts_async(n, [0, n], 2, local.data_ptr(), final.data_ptr(), 0)
    ctx.wait()
    # parts that span several bands of the C(v) scratch (a part's event comes after its LAST segment)
    ctx.set_option("cum_budget_bytes", 1 << 21)
    try:
        final.fill_(-2.0)
        torch.cuda.synchronize()
        ctx.dist_rows_parts_device_async(local.data_ptr(), 0, n, 3, result_type=dashing_amd.MASH_DIST, k=21)
        ctx.collect_parts_async(n, [0, n], 3, local.data_ptr(), final.data_ptr(), 0)
        ctx.wait()
        assert final.cpu().numpy().tobytes() == full.tobytes()
    finally:
        ctx.set_option("cum_budget_bytes", 8 << 30)


def test_finalize_tile_order_does_not_change_results(ctx):
    """k_finalize walks its own row-major tile list while the tile kernel's launch order is XCD-interleaved (run_pairs):
    with hundreds of tiles per segment, a row range, parts and several bands give the same bytes"""
    import torch

    n, p = 3300, 10
    regs = synth.survey_sketches(n, p, seed=0xF1A7)[0]
    ctx.set_sketches(regs)
    base = ctx.dist_rows()
    dev = torch.device("cuda", 0)
    try:
        assert ctx.dist_rows().tobytes() == base.tobytes()
        lo, span = dashing_amd.tri_span(n, 0, 640), dashing_amd.tri_span(n, 640, 2100)
        assert ctx.dist_rows(640, 2100).tobytes() == base[lo : lo + span].tobytes()
        out = torch.full((base.size,), -1.0, dtype=torch.float32, device=dev)
        torch.cuda.synchronize()
        ctx.dist_rows_parts_device_async(out.data_ptr(), 0, n, 3)
        ctx.wait()
        assert out.cpu().numpy().tobytes() == base.tobytes()
        ctx.set_option("cum_budget_bytes", 1 << 24)  # several bands
        assert ctx.dist_rows().tobytes() == base.tobytes()
        out.fill_(-1.0)
        torch.cuda.synchronize()
        ctx.dist_rows_parts_device_async(out.data_ptr(), 0, n, 3)
        ctx.wait()
        assert out.cpu().numpy().tobytes() == base.tobytes()
    finally:
        ctx.set_option("cum_budget_bytes", 8 << 30)


def test_large_parts_cut_the_tile_kernel(ctx):
    """parts of >= 2 048 tiles also end a band of the tile kernel (run_pairs, kPartBandTiles): the span stays the
    byte-identical one and every part's event still follows its last segment (pipelined collect into a second buffer)"""
    import torch

    n, p = 24000, 10
    regs = torch.from_numpy(synth.survey_sketches(n, p, seed=0xBA4D)[0]).cuda()
    dev = torch.device("cuda", 0)
    total = n * (n - 1) // 2
    ref = torch.empty(total, dtype=torch.float32, device=dev)
    ctx.attach_device(regs.data_ptr(), n, p)
    ctx.dist_rows_device(ref.data_ptr(), 0, n)
    ctx.synchronize()
    for rb, re in ((0, n), (4096, 20000)):
        for nparts in (2, 3):
            span, lo = dashing_amd.tri_span(n, rb, re), dashing_amd.tri_span(n, 0, rb)
            local = torch.full((span,), -1.0, dtype=torch.float32, device=dev)
            final = torch.full((span,), -2.0, dtype=torch.float32, device=dev)
            torch.cuda.synchronize()
            ctx.attach_device(regs.data_ptr(), n, p)
            ctx.dist_rows_parts_device_async(local.data_ptr(), rb, re, nparts)
            ctx.wait()
            assert ctx.info("bands") == nparts  # one band per (large) part
            assert torch.equal(local.view(torch.int32), ref[lo : lo + span].view(torch.int32)), (rb, re, nparts)
            del local, final
    local = torch.full((total,), -1.0, dtype=torch.float32, device=dev)
    final = torch.full((total,), -2.0, dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    ctx.dist_rows_parts_device_async(local.data_ptr(), 0, n, 4)
    ctx.collect_parts_async(n, [0, n], 4, local.data_ptr(), final.data_ptr(), 0)
    ctx.wait()
    assert torch.equal(final.view(torch.int32), ref.view(torch.int32))


def test_async_rows_and_wait(ctx):
    """dsh_dist_rows_async / dsh_wait (the reference's ping-pong buffers, src/sketch_and_cmp.h:804-816): calls
    return before the work is done, may be issued back to back, and deliver the blocking call's bytes."""
    n, p = 900, 12
    regs = synth.synthetic_sketches(n, p, seed=4242)
    ctx.set_sketches(regs)
    full = ctx.dist_rows(result_type=dashing_amd.MASH_DIST, k=25)
    cuts = [0, 300, 301, 650, n]
    bufs = [dashing_amd.PinnedArray(dashing_amd.tri_span(n, cuts[i], cuts[i + 1]) + 4) for i in range(len(cuts) - 1)]
    for b in bufs:
        b.array[:] = -5.0
    for i, b in enumerate(bufs):  # four calls in flight on the ctx stream
        ctx.dist_rows_async(b.array, cuts[i], cuts[i + 1], result_type=dashing_amd.MASH_DIST, k=25)
    ctx.wait()
    got = np.concatenate([b.array[: dashing_amd.tri_span(n, cuts[i], cuts[i + 1])] for i, b in enumerate(bufs)])
    assert got.tobytes() == full.tobytes()
    assert all((b.array[-4:] == -5.0).all() for b in bufs)  # nothing written past a span
    # the classic loop: block b+1 enqueued before block b is consumed
    two = [dashing_amd.PinnedArray(max(dashing_amd.tri_span(n, cuts[i], cuts[i + 1]) for i in range(4))) for _ in range(2)]
    out = []
    ctx.dist_rows_async(two[0].array, cuts[0], cuts[1])
    for i in range(4):
        ctx.wait()
        if i + 1 < 4:
            ctx.dist_rows_async(two[(i + 1) & 1].array, cuts[i + 1], cuts[i + 2])
        out.append(two[i & 1].array[: dashing_amd.tri_span(n, cuts[i], cuts[i + 1])].copy())
    assert np.concatenate(out).tobytes() == ctx.dist_rows().tobytes()
    # per-call completion (dsh_event_record / dsh_event_wait): block b+1 and b+2 are already enqueued -- their kernels
    # fill the OTHER device buffer while block b is copied out on the copy stream -- when block b is consumed; every host
    # buffer is poisoned as soon as it has been read, so a copy that lands late or twice would show
    n2, p2 = 2600, 12
    regs2 = synth.synthetic_sketches(n2, p2, seed=777)
    ctx.set_sketches(regs2)
    want = ctx.dist_rows(result_type=dashing_amd.JI)
    cuts2 = dashing_amd.partition_rows(n2, 9, 1)
    spans = [dashing_amd.tri_span(n2, cuts2[i], cuts2[i + 1]) for i in range(9)]
    three = [dashing_amd.PinnedArray(max(spans)) for _ in range(3)]
    tickets, got = {}, []

    def enqueue(i):
        ctx.dist_rows_async(three[i % 3].array, cuts2[i], cuts2[i + 1])
        tickets[i] = ctx.event_record()

    enqueue(0)
    enqueue(1)
    for i in range(9):
        if i + 2 < 9:
            enqueue(i + 2)
        ctx.event_wait(tickets[i])
        assert ctx.event_done(tickets[i])
        got.append(three[i % 3].array[: spans[i]].copy())
        three[i % 3].array[:] = np.float32(-7.0)
    assert np.concatenate(got).tobytes() == want.tobytes()
    ctx.wait()
    with pytest.raises(dashing_amd.DshError):
        ctx.event_wait(10_000_000)  # never recorded


def test_device_async_and_wait_event(ctx):
    """*_device_async leaves the result in a caller buffer without blocking; dsh_wait_event orders the ctx
    stream after work the caller enqueued on ITS stream (here: torch producing the sketches)."""
    import torch

    n, p = 700, 10
    dev = torch.device("cuda", 0)
    regs = synth.synthetic_sketches(n, p, seed=99)
    want = None
    ctx.set_sketches(regs)
    want = ctx.dist_rows()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        regs_d = torch.from_numpy(regs).to(dev, non_blocking=False)
        for _ in range(20):  # keep torch's stream busy so that a missing dependency would show
            regs_d = torch.maximum(regs_d, regs_d)
        ev = torch.cuda.Event()
        ev.record(s)
    out = torch.full((n * (n - 1) // 2,), -1.0, dtype=torch.float32, device=dev)
    torch.cuda.current_stream().synchronize()  # `out` is filled on torch's default stream
    ctx.attach_device(regs_d.data_ptr(), n, p)
    ctx.wait_event(ev.cuda_event)
    ctx.dist_rows_device_async(out.data_ptr(), 0, n)
    ctx.wait()
    assert out.cpu().numpy().tobytes() == want.tobytes()


def test_rect_matches_tri(ctx, oracle):
    n, p = 150, 10
    regs = synth.synthetic_sketches(n, p, seed=43)
    ctx.set_sketches(regs)
    tri = ctx.dist_rows(result_type=dashing_amd.MASH_DIST, k=21)
    rect = ctx.dist_rect(3, 70, 60, 150, result_type=dashing_amd.MASH_DIST, k=21)
    for i in (3, 10, 64, 69):
        for j in (70, 100, 128, 149):
            assert rect[i - 3, j - 60] == tri[dashing_amd.tri_index(n, i, j)]
    want = oracle.dist_rect(regs[3:70], regs[60:150], 2, oracle.MASH_DIST, 21)
    # diagonal (i == j) is computed too in a rectangle: identical sketches -> J = 1 -> Mash 0
    close(rect, want)


def test_options_do_not_change_results(ctx):
    n, p = 140, 14
    regs = synth.synthetic_sketches(n, p, seed=44)
    ctx.set_sketches(regs)
    base = ctx.dist_rows()
    try:
        for kc in (16, 32):
            ctx.set_option("kc", kc)
            assert ctx.dist_rows().tobytes() == base.tobytes()
        with pytest.raises(dashing_amd.DshError):
            ctx.set_option("kc", 64)
        for emax in (0, 1, 8, 64, 200, 255, -1):   # dense-only .. full exception lists: same exact histogram
            ctx.set_option("emax", emax)
            assert ctx.dist_rows().tobytes() == base.tobytes()
        for elow, emax in ((0, -1), (1, 0), (7, 3), (64, 255), (255, 0), (255, 255), (200, 17), (-1, -1)):  # the listed lower tail
            ctx.set_option("elow", elow)
            ctx.set_option("emax", emax)
            assert ctx.dist_rows().tobytes() == base.tobytes()
            assert ctx.dist_rect(3, 133, 0, 140).tobytes() == ctx.dist_rect(3, 133, 0, 140).tobytes()
        for sm in (0, 1, -1):   # identity vs (threshold,min)-sorted plane columns
            ctx.set_option("sort", sm)
            assert ctx.dist_rows().tobytes() == base.tobytes()
            assert ctx.info("sorted") == (0 if sm == 0 else 1)
        for ns in (1, 2, 7, 64, 0):
            ctx.set_option("nsplit", ns)
            assert ctx.dist_rows().tobytes() == base.tobytes()
        ctx.set_option("cum_budget_bytes", 1 << 21)  # force many bands
        assert ctx.dist_rows().tobytes() == base.tobytes()
        ctx.set_option("cum_budget_bytes", 8 << 30)
        for kc, ns in ((16, 0), (32, 0), (16, 3), (16, 64)):
            ctx.set_option("kc", kc)
            ctx.set_option("nsplit", ns)
            assert ctx.dist_rows().tobytes() == base.tobytes()
        ctx.set_option("nsplit", 0)
        # the tuning knobs whose A/B was decided in rounds 2-5 are gone, with their losing arms
        for gone in ("pair_lockstep", "ls_item_chunks", "ls_sort_items", "xcd_swizzle", "finalize_rowmajor", "finalize_xcd_tiles",
                     "finalize_two_streams", "finalize_shared_instance", "colindex_split", "unpermute_gather", "xch_tail_permille",
                     "xch_tail_permille2", "xch_tail_head_min_rounds", "assembler_permille", "shard_c0_x10"):
            with pytest.raises(dashing_amd.DshError):
                ctx.set_option(gone, 1)
        # the what-if variant of the tile kernel on the matrix cores is not in the product library (`make WHATIF=1`
        # builds it): the default build refuses the option; a what-if build must give the same integers
        if ctx.info("whatif_mfma"):
            ctx.set_option("pair_mfma", 1)
            for kc in (16, 32):
                ctx.set_option("kc", kc)
                assert ctx.dist_rows().tobytes() == base.tobytes()
        else:
            with pytest.raises(dashing_amd.DshError):
                ctx.set_option("pair_mfma", 1)
    finally:
        if ctx.info("whatif_mfma"):
            ctx.set_option("pair_mfma", 0)
        ctx.set_option("kc", 0)
        ctx.set_option("emax", -1)
        ctx.set_option("elow", -1)
        ctx.set_option("sort", -1)
        ctx.set_option("nsplit", 0)
        ctx.set_option("cum_budget_bytes", 8 << 30)


@pytest.mark.parametrize("p,n", [(8, 300), (9, 300), (10, 700), (11, 300), (12, 260), (13, 300), (15, 200), (16, 150), (18, 40)])
def test_tile_kernel_item_shapes_agree_with_the_oracle(ctx, oracle, p, n):
    """k_pair_counts_ls (two work items per 512-thread workgroup, AND/BCNT batches phase-locked, deferred plane flush, a
    shorter or missing partner item idling at the barriers) at every precision where a plane spans whole chunks (p >= 9;
    below, the free-running k_pair_counts of the small precisions), k-rows per stage 16 / 32, tiles cut into 1 .. 5 pieces
    (odd item counts, unequal item lengths): the same bytes every way, and those within 1e-6 of the CPU oracle."""
    regs = synth.synthetic_sketches(n, p, seed=900 + p)
    ctx.set_sketches(regs)
    try:
        base = ctx.dist_rows()
        want = oracle.dist_tri(regs)
        close(base, want)
        for kc, ns in ((16, 0), (32, 0), (16, 3), (16, 1), (32, 5), (16, 64)):
            ctx.set_option("kc", kc)
            ctx.set_option("nsplit", ns)
            got = ctx.dist_rows()
            assert ctx.info("lockstep") == (1 if (1 << p) // 32 >= kc else 0)
            assert got.tobytes() == base.tobytes(), (kc, ns)
        # a row range (odd tile counts) and a rectangle through the same kernel
        ctx.set_option("kc", 0)
        ctx.set_option("nsplit", 0)
        part = ctx.dist_rows(5, n - 3)
        lo = dashing_amd.tri_span(n, 0, 5)
        assert part.tobytes() == base[lo : lo + part.size].tobytes()
        rect = ctx.dist_rect(0, n // 2, n // 3, n)
        close(rect, oracle.dist_rect(regs[: n // 2], regs[n // 3 :]))
    finally:
        ctx.set_option("kc", 0)
        ctx.set_option("nsplit", 0)


def test_properties_at_scale(ctx, oracle):
    """Size-independent properties at a size the oracle cannot sweep: symmetry under
    permutation, identical rows -> J == 1, sampled rows agree with the oracle."""
    n, p = 2000, 14
    regs = synth.synthetic_sketches(n, p, seed=45)
    regs[1500] = regs[3]
    ctx.set_sketches(regs)
    tri = ctx.dist_rows()
    assert tri.size == n * (n - 1) // 2
    assert np.isfinite(tri).all() and tri.min() >= 0 and tri.max() <= 1
    assert tri[dashing_amd.tri_index(n, 3, 1500)] == 1.0
    # permutation: reversing the sketch order must give the same values at the mirrored index
    ctx.set_sketches(regs[::-1])
    rev = ctx.dist_rows()
    rng = np.random.default_rng(0)
    for _ in range(2000):
        i, j = sorted(rng.choice(n, 2, replace=False))
        assert tri[dashing_amd.tri_index(n, i, j)] == rev[dashing_amd.tri_index(n, n - 1 - j, n - 1 - i)]
    # a few full rows against the oracle
    for r in (0, 777, 1998):
        want = oracle.dist_rows(regs, r, r + 1)
        lo = dashing_amd.tri_index(n, r, r + 1)
        close(tri[lo : lo + want.size], want)


def test_heterogeneous_collection(ctx, oracle):
    """Cardinalities spread over 3 decades (thresholds and minima differ a lot between sketches):
    exercises per-tile plane ranges, empty dense ranges and the sorted column order."""
    p, m = 12, 1 << 12
    parts = []
    for k, card in enumerate((3_000, 40_000, 600_000, 9_000_000, 150_000_000)):
        parts += [synth.hll_registers(1000 * k + g, card * (1 + g % 3), p) for g in range(60)]
    regs = np.stack(parts)
    rng = np.random.default_rng(7)
    regs = regs[rng.permutation(len(regs))]
    regs[5] = 0
    regs[6] = regs[7]
    ctx.set_sketches(regs)
    for estim in (0, 2):
        want = oracle.dist_tri(regs, estim, oracle.JI, 31)
        try:
            for sm in (1, 0):
                ctx.set_option("sort", sm)
                close(ctx.dist_rows(estim=estim), want)
        finally:
            ctx.set_option("sort", -1)
    assert ctx.dist_rows()[dashing_amd.tri_index(len(regs), 6, 7)] == 1.0


@pytest.mark.parametrize("nshards", [1, 2, 3, 8])
def test_virtual_shards_assemble(ctx, nshards):
    """Multi-GPU path with G virtual ranks on one device: cost-balanced shards of the sorted-order
    triangle, spans laid back to back, one un-permute == the single-call matrix, byte for byte."""
    import torch

    n, p = 700, 12
    regs = synth.synthetic_sketches(n, p, seed=77)
    ctx.set_sketches(regs)
    want = ctx.dist_rows(result_type=dashing_amd.MASH_DIST, k=21)
    off = ctx.shard_plan(nshards)
    assert off[0] == 0 and off[-1] == n * (n - 1) // 2 and all(off[i] <= off[i + 1] for i in range(nshards))
    dev = torch.device("cuda", 0)
    sorted_full = torch.full((off[-1],), -1.0, dtype=torch.float32, device=dev)
    for r in range(nshards):
        span = torch.full((max(off[r + 1] - off[r], 1),), -2.0, dtype=torch.float32, device=dev)
        torch.cuda.synchronize()  # torch fills run on torch's stream, the library on its own: order them
        ctx.dist_shard_device(span.data_ptr(), r, nshards, result_type=dashing_amd.MASH_DIST, k=21)
        ctx.synchronize()
        sorted_full[off[r] : off[r + 1]] = span[: off[r + 1] - off[r]]
    torch.cuda.synchronize()
    final = torch.full((off[-1],), -3.0, dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    ctx.unpermute_device(sorted_full.data_ptr(), final.data_ptr())
    ctx.synchronize()
    assert final.cpu().numpy().tobytes() == want.tobytes()
    # the same from the padded blocks a gather delivers (shard r at r * stride)
    stride = max(max(off[r + 1] - off[r] for r in range(nshards)), 1) + 5
    stage = torch.full((nshards * stride,), -4.0, dtype=torch.float32, device=dev)
    for r in range(nshards):
        stage[r * stride : r * stride + off[r + 1] - off[r]] = sorted_full[off[r] : off[r + 1]]
    final.fill_(-3.0)
    torch.cuda.synchronize()
    ctx.unpermute_staged_device(stage.data_ptr(), stride, nshards, final.data_ptr())
    assert final.cpu().numpy().tobytes() == want.tobytes()


@pytest.mark.parametrize("rt", [2, 4, 5, 6, 7, 8])
def test_set_triple_measures(ctx, oracle, rt):
    """Second arm of result_cmp (SIZES, containment family): triangle and rectangle vs the oracle."""
    n, p = 150, 12
    regs = synth.synthetic_sketches(n, p, seed=91)
    regs[10] = 0
    regs[11] = regs[12]
    ctx.set_sketches(regs)
    want = oracle.dist_tri(regs, 2, rt, 31)
    got = ctx.dist_rows(result_type=rt, k=31)
    fin = np.isfinite(want)
    assert (np.isfinite(got) == fin).all()
    assert np.allclose(got[fin], want[fin], rtol=1e-6, atol=1e-9)
    rect = ctx.dist_rect(100, 150, 0, 100, result_type=rt, k=31)
    wr = oracle.dist_rect(regs[100:150], regs[:100], 2, rt, 31)
    f2 = np.isfinite(wr)
    assert np.allclose(rect[f2], wr[f2], rtol=1e-6, atol=1e-9)


@pytest.mark.parametrize("p", [9, 10, 12, 13, 14])
def test_record_width_does_not_change_results(ctx, oracle, p):
    """round 4: the position index keeps a bucket's first 7 (p <= 12, lists of <= 256 entries) or 3 entries inside its
    record; list caps beyond 256 entries switch a small-p collection to the narrow records and four look-up rounds.
    Same exact histograms whatever the combination -- and equal to the oracle's."""
    n = 300
    regs = synth.synthetic_sketches(n, p, seed=70 + p)
    ctx.set_sketches(regs)
    base = ctx.dist_rows()
    want = oracle.dist_tri(regs)
    assert np.allclose(base, want, rtol=1e-6, atol=1e-15)
    try:
        for emax, elow in ((255, 255), (200, 17), (128, 128), (0, 0), (64, 255), (3, 250), (-1, -1)):
            ctx.set_option("emax", emax)
            ctx.set_option("elow", elow)
            assert ctx.dist_rows().tobytes() == base.tobytes(), (p, emax, elow)
            assert ctx.dist_rows(estim=dashing_amd.ESTIM_ORIGINAL).tobytes() != b""  # (runs; checked against the oracle below)
        ctx.set_option("emax", 255)
        ctx.set_option("elow", 255)
        for estim in (0, 1, 2):  # every estimator on the narrow-record path of a small p
            got = ctx.dist_rows(estim=estim)
            assert np.allclose(got, oracle.dist_tri(regs, estim), rtol=1e-6, atol=1e-15), (p, estim)
    finally:
        ctx.set_option("emax", -1)
        ctx.set_option("elow", -1)


@pytest.mark.parametrize("p", [16, 22, 23])
def test_constant_sketches_at_large_p(ctx, oracle, p):
    """Sketches whose registers are all equal (or of two values) at 2^16 ... 2^23 registers: every count of the per-sketch
    histogram lands in one or two bins -- the case that overflows a counter narrower than the sketch is long, and the
    one in which every LDS atomic of a wave hits the same address."""
    regs = np.empty((3, 1 << p), np.uint8)
    regs[0] = 5
    regs[1] = 6
    regs[2] = 5
    regs[2, ::2] = 4
    ctx.set_sketches(regs)
    for estim in (0, 2):
        got = ctx.cardinalities(estim)
        want = oracle.cardinalities(regs, estim)
        assert np.allclose(got, want, rtol=1e-12, atol=0), (p, estim, got, want)
    got = ctx.dist_rows()
    want = oracle.dist_tri(regs)
    assert np.allclose(got, want, rtol=1e-6, atol=1e-15)


@pytest.mark.parametrize("p", [8, 11, 15])
def test_adversarial_registers(ctx, oracle, p):
    """Register arrays that do NOT follow the HLL law: uniform random over the whole value range
    (no sparse tail: thresholds collapse to the maximum, ~50 dense planes), saturated and constant
    sketches, a sketch with a single non-zero register.  Every estimator, sorted and identity columns."""
    q = 64 - p
    rng = np.random.default_rng(p)
    n, m = 140, 1 << p
    regs = rng.integers(0, q + 2, size=(n, m)).astype(np.uint8)
    regs[0] = q + 1                 # saturated: MLE -> inf, J -> 0
    regs[1] = 7                     # constant
    regs[2] = 0
    regs[2, 5] = 9                  # one non-zero register
    regs[3] = rng.integers(0, 3, size=m)          # tiny values
    regs[4] = rng.integers(q - 2, q + 2, size=m)  # huge values
    ctx.set_sketches(regs)
    try:
        for sm in (1, 0):
            ctx.set_option("sort", sm)
            for estim in (0, 1, 2):
                want = oracle.dist_tri(regs, estim, oracle.JI, 31)
                got = ctx.dist_rows(estim=estim)
                fin = np.isfinite(want)
                assert (np.isfinite(got) == fin).all()
                err = np.abs(got[fin].astype(np.float64) - want[fin])
                assert (err <= 1e-6 * np.maximum(np.abs(want[fin]), 1e-9)).all(), (p, sm, estim, err.max())
    finally:
        ctx.set_option("sort", -1)


@pytest.mark.parametrize("rt", [dashing_amd.JI, dashing_amd.MASH_DIST, dashing_amd.SYMMETRIC_CONTAINMENT_INDEX])
def test_knn_vs_oracle(ctx, oracle, rt):
    """--nearest-neighbors: k best per sketch, similarity descending / distance ascending, ties by
    lower index (identical sketches and the empty sketch create exact ties)."""
    n, p = 300, 10
    regs = synth.synthetic_sketches(n, p, seed=55)
    regs[7] = regs[8] = regs[9]
    regs[20] = 0
    ctx.set_sketches(regs)
    for nn in (1, 5, 64):
        wi, wv = oracle.knn(regs, nn, result_type=rt, k=21)
        gi, gv = ctx.knn(nn, result_type=rt, k=21)
        assert (gi == wi).all(), (rt, nn, np.argwhere(gi != wi)[:5])
        assert np.allclose(gv, wv, rtol=1e-6, atol=1e-12, equal_nan=True)
    # all neighbours (nn = n-1) and more than exist
    gi, gv = ctx.knn(n + 5, result_type=rt, k=21)
    assert (gi[:, n - 1 :] == 0xFFFFFFFF).all() and (gi[:, : n - 1] != 0xFFFFFFFF).all()
    assert all(sorted(row[: n - 1].tolist()) == [j for j in range(n) if j != i] for i, row in enumerate(gi[:10]))
    # queries x references
    wi, wv = oracle.knn(regs, 3, qb=200, qe=300, rb=0, re=200, result_type=rt, k=21)
    gi, gv = ctx.knn(3, 200, 300, 0, 200, result_type=rt, k=21)
    assert (gi == wi).all() and np.allclose(gv, wv, rtol=1e-6, atol=1e-12, equal_nan=True)


@pytest.mark.parametrize("rt", [dashing_amd.JI, dashing_amd.CONTAINMENT_INDEX, dashing_amd.FULL_CONTAINMENT_DIST])
def test_knn_all_vs_all_paths_agree(ctx, oracle, rt):
    """All-vs-all kNN computes each pair once into an n x n matrix (both orientations -- the containment
    measures are asymmetric); the block-of-queries fallback must select exactly the same neighbours."""
    n, p = 700, 12
    regs = synth.related_sketches(n, p, seed=91)[0]
    ctx.set_sketches(regs)
    gi, gv = ctx.knn(7, result_type=rt, k=31)
    ctx.set_option("knn_square_budget_bytes", 0)
    try:
        fi, fv = ctx.knn(7, result_type=rt, k=31)
    finally:
        ctx.set_option("knn_square_budget_bytes", 96 << 30)
    assert (gi == fi).all() and (gv.view(np.uint32) == fv.view(np.uint32)).all()
    wi, wv = oracle.knn(regs, 7, result_type=rt, k=31)
    assert (gi == wi).all() and np.allclose(gv, wv, rtol=1e-6, atol=1e-12, equal_nan=True)


@pytest.mark.parametrize("rt", [dashing_amd.JI, dashing_amd.CONTAINMENT_INDEX, dashing_amd.MASH_DIST])
def test_knn_bands_match_square_at_c3_size(ctx, rt):
    """The band-wise all-vs-all kNN (triangle computed once, no n x n matrix: dsh_knn beyond its square budget) selects
    exactly what the n x n path selects -- BASELINE configs[2] size, with exact ties from duplicated sketches."""
    n, p = 10_000, 14
    regs = synth.survey_sketches(n, p, seed=0x5EED0000)[0]
    regs[5] = regs[6] = regs[7]
    regs[9000] = regs[123]
    ctx.set_sketches(regs)
    si, sv = ctx.knn(10, result_type=rt, k=31)
    ctx.set_option("knn_square_budget_bytes", 256 << 20)  # bands of 3 200 rows: four of them
    try:
        bi, bv = ctx.knn(10, result_type=rt, k=31)
    finally:
        ctx.set_option("knn_square_budget_bytes", 96 << 30)
    assert (si == bi).all() and (sv.view(np.uint32) == bv.view(np.uint32)).all()
    assert si[5, 0] == 6 and si[6, 0] == 5 and si[7, 0] == 5 and si[5, 1] == 7  # ties go to the lower index


def test_knn_more_neighbours_than_the_band_path_takes(ctx, oracle):
    """nn > 1024 (the band path keeps a sketch's running list in LDS) falls back to query blocks"""
    n, p = 1100, 10
    regs = synth.synthetic_sketches(n, p, seed=5)
    ctx.set_sketches(regs)
    ctx.set_option("knn_square_budget_bytes", 0)
    try:
        gi, gv = ctx.knn(1050, k=21)
    finally:
        ctx.set_option("knn_square_budget_bytes", 96 << 30)
    wi, wv = oracle.knn(regs, 1050, k=21)
    assert (gi == wi).all() and np.allclose(gv, wv, rtol=1e-6, atol=1e-12, equal_nan=True)


@pytest.mark.parametrize("p", [4, 24])
def test_extreme_histograms_smallest_and_largest_precision(ctx, oracle, p):
    """the estimator's divisions at the edges of their operand range -- all-zero sketches (c0 = m), one
    non-zero register (c0 = m - 1), saturated sketches (every register q + 1: the MLE returns +inf), identical pairs --
    at the smallest and the largest precision the library takes; all three estimators, vs the oracle."""
    m, q = 1 << p, 64 - p
    rng = np.random.default_rng(p)
    regs = np.zeros((7, m), np.uint8)
    regs[1, m // 3] = 1                       # c0 = m - 1
    regs[2, :] = q + 1                        # saturated
    regs[3] = synth.hll_registers(5, 3 * m, p)
    regs[4] = regs[3]                         # identical pair
    regs[5] = rng.integers(0, q + 2, m).astype(np.uint8)   # uniform over the whole value range
    regs[6, : m // 2] = q + 1                 # half saturated, half empty
    ctx.set_sketches(regs)
    for estim in (0, 1, 2):
        want_c = oracle.cardinalities(regs, estim)
        got_c = ctx.cardinalities(estim)
        fin = np.isfinite(want_c)
        assert (np.isfinite(got_c) == fin).all() and np.allclose(got_c[fin], want_c[fin], rtol=1e-12)
        for rt in (dashing_amd.JI, dashing_amd.MASH_DIST, dashing_amd.SIZES):
            want = oracle.dist_tri(regs, estim, rt, 31).astype(np.float64)
            got = ctx.dist_rows(estim=estim, result_type=rt, k=31).astype(np.float64)
            fin = np.isfinite(want)
            assert (np.isfinite(got) == fin).all(), (estim, rt)
            assert (np.isnan(got) == np.isnan(want)).all(), (estim, rt)
            err = np.abs(got[fin] - want[fin])
            assert (err <= 1e-6 * np.maximum(np.abs(want[fin]), 1e-9)).all(), (estim, rt, err.max())


@pytest.mark.parametrize("p", [8, 10, 12, 14])
def test_mle_division_against_the_oracle_over_the_operand_range(ctx, oracle, p):
    """the division of the MLE's inner recurrence takes ONE Newton step (div_inner, estimators.h): correctly
    rounded unless the exact quotient lies within 2^-97 of a rounding midpoint, so identity with a CPU `/` is statistical,
    not by construction.  This pins it: sketches whose cardinalities sweep eleven decades (the recurrence's operands
    x' in [2^-k, 2) x every histogram shape from nearly empty to nearly saturated), unions of very unequal sets, every
    pair's Jaccard / sizes against the oracle -- the contract 1e-6, and (observed, asserted loosely) float32-identical in
    all but a vanishing share of the pairs."""
    m = 1 << p
    rng = np.random.default_rng(1000 + p)
    cards = np.unique(np.round(np.logspace(0, 11, 90) * (0.5 + rng.random(90))).astype(np.int64))
    regs = np.stack([synth.hll_registers(int(rng.integers(1, 1 << 30)), int(c), p) for c in cards])
    ctx.set_sketches(regs)
    n = len(regs)
    for rt in (dashing_amd.JI, dashing_amd.SIZES):
        want = oracle.dist_tri(regs, dashing_amd.ESTIM_ERTL_MLE, rt, 31)
        got = ctx.dist_rows(estim=dashing_amd.ESTIM_ERTL_MLE, result_type=rt, k=31)
        fin = np.isfinite(want)
        assert (np.isfinite(got) == fin).all()
        err = np.abs(got[fin].astype(np.float64) - want[fin]) / np.maximum(np.abs(want[fin].astype(np.float64)), 1e-9)
        assert err.max() <= 1e-6, (p, rt, err.max())
        assert (got[fin] == want[fin]).mean() >= 0.999, (p, rt, (got[fin] != want[fin]).sum(), n)
    want_c = oracle.cardinalities(regs, dashing_amd.ESTIM_ERTL_MLE)
    got_c = ctx.cardinalities(dashing_amd.ESTIM_ERTL_MLE)
    fin = np.isfinite(want_c)
    assert np.allclose(got_c[fin], want_c[fin], rtol=1e-12)


@pytest.mark.parametrize("p", [12, 14, 16])
def test_overflow_fragments_do_not_change_results(ctx, p):
    """overflow fragments of the tile kernel (plan.h): when a band's one-plane items lie a little above a multiple of 512,
    the items left over run as fragments of a plane that ADD their counts (32-bit atomics, two uint16 counts per word below
    p = 16) to a cleared C(v) block.  Byte-identical to the run without them, for row ranges too."""
    sizes = {12: range(3000, 4400, 200), 14: range(1500, 2700, 150), 16: range(900, 1500, 100)}[p]
    hit = 0
    try:
        for n in sizes:
            regs = synth.survey_sketches(n, p, seed=500 + n)[0]
            ctx.set_sketches(regs)
            ctx.set_option("overflow_frag_permille", 0)
            base = ctx.dist_rows()
            assert ctx.info("frag_items") == 0
            part0 = ctx.dist_rows(128, n // 2 // 128 * 128 + 256)
            for pm in (500, 1000):
                ctx.set_option("overflow_frag_permille", pm)
                got = ctx.dist_rows()
                hit += ctx.info("frag_items") > 0
                assert got.tobytes() == base.tobytes(), (p, n, pm, ctx.info("frag_items"))
                assert ctx.dist_rows(128, n // 2 // 128 * 128 + 256).tobytes() == part0.tobytes(), (p, n, pm)
    finally:
        ctx.set_option("overflow_frag_permille", 500)
    assert hit >= 3, "these sizes are meant to produce overflow fragments"


def test_out_of_range_registers_are_refused(ctx):
    """uploaded registers above 64 - p + 1 (corrupt / foreign sketches) make the compare entry points fail loudly
    instead of aliasing into wrong histogram bins"""
    p = 12
    regs = synth.synthetic_sketches(40, p, seed=3)
    for badval in (64 - p + 2, 100, 200, 255):
        r = regs.copy()
        r[17, 1234] = badval
        ctx.set_sketches(r)
        with pytest.raises(dashing_amd.DshError) as e:
            ctx.dist_rows()
        assert e.value.code == -22 and "sketch 17" in str(e.value)
    regs[17, 1234] = 64 - p + 1  # the largest legal value is fine
    ctx.set_sketches(regs)
    assert np.isfinite(ctx.dist_rows()).all()


def test_errors(ctx):
    with pytest.raises(dashing_amd.DshError):
        ctx.alloc(10, 3)
    with pytest.raises(dashing_amd.DshError):
        ctx.alloc(10, 30)
    ctx.alloc(4, 10)
    with pytest.raises(dashing_amd.DshError):
        ctx.dist_rows(result_type=9)
    with pytest.raises(dashing_amd.DshError):
        ctx.set_option("nope", 1)
    # slot ranges whose end wraps around 2^64 are out of range, not "slot 0"
    with pytest.raises(dashing_amd.DshError):
        ctx.clear((1 << 64) - 1, 2)
    with pytest.raises(dashing_amd.DshError):
        ctx.upload(np.zeros((2, 1 << 10), np.uint8), first_slot=(1 << 64) - 1)
    with pytest.raises(dashing_amd.DshError):
        ctx.clear(3, 2)
    ctx.clear(3, 1)
    # empty / degenerate
    ctx.alloc(0, 10)
    assert ctx.dist_rows().size == 0
    ctx.alloc(1, 10)
    assert ctx.dist_rows().size == 0

"""GPU: seeded random sweep over (n, p, estimator, measure, register law) against the oracle --
triangle, a random row range, a random rectangle and a random shard plan per case."""
import os

import numpy as np
import pytest

import dashing_amd
from dashing_amd import synth

pytestmark = pytest.mark.gpu


def _regs(rng, n, p, kind):
    m = 1 << p
    q = 64 - p
    if kind == "law":
        cards = np.exp(rng.uniform(np.log(50 * m / 1024 + 10), np.log(4e8), n))
        return np.stack([synth.hll_registers(int(rng.integers(1 << 30)), int(c), p) for c in cards])
    if kind == "related":
        return synth.synthetic_sketches(n, p, seed=int(rng.integers(1 << 30)))
    if kind == "uniform":
        return rng.integers(0, q + 2, size=(n, m)).astype(np.uint8)
    if kind == "narrow":
        return rng.integers(5, 8, size=(n, m)).astype(np.uint8)
    raise AssertionError(kind)


# distance measure -> the index it is a function of (result_cmp, src/dashing.h:568-592).  The distance
# formulas jump at index == 0 (`ret != 0 ? -log(ret)/k : 1`), so where the index is 0 up to rounding an
# ulp of the device log() in a cardinality (ORIGINAL/IMPROVED small-range terms) may pick the other side.
INDEX_OF = {0: 1, 3: 1, 6: 5, 4: 5, 8: 7}


def _close(got, want, index_got=None, index_want=None):
    """index_*: the same pairs under the underlying index measure; a mismatch is tolerated only where
    both implementations put that index within 1e-9 of zero (the discontinuity of the distance formulas)."""
    if index_got is not None:
        at_jump = (np.abs(index_got) < 1e-9) & (np.abs(index_want) < 1e-9)
        got, want = got[~at_jump], want[~at_jump]
    fin = np.isfinite(want)
    assert (np.isfinite(got) == fin).all()
    err = np.abs(got[fin].astype(np.float64) - want[fin])
    # 1e-6 relative; plus an absolute floor of 1e-12 of the matrix scale: SIZES / containment values are
    # differences of cardinalities, so one ulp of log() (device libm vs glibc, ORIGINAL/IMPROVED estimators)
    # in a cardinality of ~1e2..1e8 can leave ~1e-14 where the CPU gets an exact 0
    scale = float(np.abs(want[fin]).max()) if fin.any() else 1.0
    assert (err <= 1e-6 * np.maximum(np.abs(want[fin]), 1e-9) + 1e-12 * max(scale, 1.0)).all(), err.max()


_FIRST = int(os.environ.get("DSH_FUZZ_FIRST", "0"))  # e.g. DSH_FUZZ_FIRST=2500 DSH_FUZZ_CASES=2500 for a long soak on fresh seeds


@pytest.mark.parametrize("case", range(_FIRST, _FIRST + int(os.environ.get("DSH_FUZZ_CASES", "100"))))
def test_random_case(ctx, oracle, case):
    rng = np.random.default_rng(1000 + case)
    p = int(rng.choice([4, 6, 8, 9, 10, 11, 12, 13, 14, 15, 16, 18, 20]))
    n = int(rng.integers(2, 420 if p <= 12 else (180 if p <= 16 else 24)))
    kind = str(rng.choice(["law", "related", "uniform", "narrow"]))
    estim = int(rng.integers(0, 3))
    rt = int(rng.choice([0, 1, 2, 3, 4, 5, 6, 7, 8]))
    k = int(rng.choice([15, 21, 31, 32]))
    regs = _regs(rng, n, p, kind)
    if n > 4 and rng.random() < 0.5:
        regs[int(rng.integers(n))] = 0
        a, b = rng.choice(n, 2, replace=False)
        regs[a] = regs[b]
    ctx.set_sketches(regs)
    emax = int(rng.choice([-1, -1, 0, 3, 17, 255]))  # caps of the two listed tails: speed knobs, never result knobs
    elow = int(rng.choice([-1, -1, 0, 2, 40, 255]))
    ctx.set_option("emax", emax)
    ctx.set_option("elow", elow)
    want = oracle.dist_tri(regs, estim, rt, k)
    ig = iw = None
    if rt in INDEX_OF:
        ig = ctx.dist_rows(estim=estim, result_type=INDEX_OF[rt], k=k)
        iw = oracle.dist_tri(regs, estim, INDEX_OF[rt], k)
        _close(ig, iw)
    _close(ctx.dist_rows(estim=estim, result_type=rt, k=k), want, ig, iw)
    # a row range: identity columns, and the layout built for the range (wanted rows first, key-ordered), which
    # normally starts at 1024 rows -- the two must give the same bytes
    rb = int(rng.integers(0, n))
    re = int(rng.integers(rb, n + 1))
    lo = dashing_amd.tri_span(n, 0, rb)
    part = ctx.dist_rows(rb, re, estim=estim, result_type=rt, k=k)
    sl = slice(lo, lo + part.size)
    _close(part, want[sl], None if ig is None else ig[sl], None if iw is None else iw[sl])
    ctx.set_option("range_sort_min_rows", 1)
    try:
        part2 = ctx.dist_rows(rb, re, estim=estim, result_type=rt, k=k)
        if part.size:  # (a range without pairs returns before any layout is built)
            assert ctx.info("sorted") == 1 and ctx.info("ncols") == n - rb
    finally:
        ctx.set_option("range_sort_min_rows", 1024)
    assert part2.tobytes() == part.tobytes()
    # a rectangle
    q0 = int(rng.integers(0, n)); q1 = int(rng.integers(q0, n + 1))
    r0 = int(rng.integers(0, n)); r1 = int(rng.integers(r0, n + 1))
    rect = ctx.dist_rect(q0, q1, r0, r1, estim=estim, result_type=rt, k=k)
    if rect.size:
        rg = rw = None
        if rt in INDEX_OF:
            rg = ctx.dist_rect(q0, q1, r0, r1, estim=estim, result_type=INDEX_OF[rt], k=k)
            rw = oracle.dist_rect(regs[q0:q1], regs[r0:r1], estim, INDEX_OF[rt], k)
        _close(rect, oracle.dist_rect(regs[q0:q1], regs[r0:r1], estim, rt, k), rg, rw)
    # shards of the sorted order, assembled
    import torch

    G = int(rng.integers(1, 6))
    off = ctx.shard_plan(G, estim)
    total = n * (n - 1) // 2
    assert off[0] == 0 and off[-1] == total
    sf = torch.zeros(max(total, 1), dtype=torch.float32, device="cuda")
    for r in range(G):
        span = torch.zeros(max(off[r + 1] - off[r], 1), dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()  # the zero fill (torch's stream) must land before the library's stream writes
        ctx.dist_shard_device(span.data_ptr(), r, G, estim, rt, k)
        ctx.synchronize()
        sf[off[r] : off[r + 1]] = span[: off[r + 1] - off[r]]
    torch.cuda.synchronize()
    fin = torch.zeros(max(total, 1), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    ctx.unpermute_device(sf.data_ptr(), fin.data_ptr())
    ctx.synchronize()
    _close(fin.cpu().numpy()[:total], want, ig, iw)
    # the same rows in parts (each key-ordered on its own, an event per part): byte-identical to the plain range
    if part.size:
        nparts = int(rng.integers(2, 6))
        pd = torch.full((part.size,), -9.0, dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        ctx.set_option("range_sort_min_rows", 1)
        try:
            ctx.dist_rows_parts_device_async(pd.data_ptr(), rb, re, nparts, estim=estim, result_type=rt, k=k)
            ctx.wait()
        finally:
            ctx.set_option("range_sort_min_rows", 1024)
        assert pd.cpu().numpy().tobytes() == part.tobytes()
    # nearest neighbours: the band path (no n x n matrix) selects what the square path selects
    if n > 3 and case % 4 == 0:
        nn = int(rng.integers(1, min(n, 9)))
        si, sv = ctx.knn(nn, estim=estim, result_type=rt, k=k)
        ctx.set_option("knn_square_budget_bytes", 0)
        try:
            bi, bv = ctx.knn(nn, estim=estim, result_type=rt, k=k)
        finally:
            ctx.set_option("knn_square_budget_bytes", 96 << 30)
        assert (si == bi).all() and (sv.view(np.uint32) == bv.view(np.uint32)).all()
    ctx.set_option("emax", -1)
    ctx.set_option("elow", -1)


@pytest.mark.parametrize("case", range(int(os.environ.get("DSH_FUZZ_BIG_CASES", "6"))))
def test_random_case_many_tiles(ctx, oracle, case):
    """the same sweep at sizes with hundreds of tiles per launch (XCD-interleaved tile order for the tile kernel, row-major
    for k_finalize, several bands, parts that cut the tile kernel): whole triangle vs the oracle; a row range, parts and
    the band-wise nearest neighbours vs the triangle"""
    import torch

    rng = np.random.default_rng(77000 + case)
    p = int(rng.choice([9, 10, 11]))
    n = int(rng.integers(2100, 3600))
    kind = str(rng.choice(["law", "related", "narrow"]))
    estim = int(rng.integers(0, 3))
    rt = int(rng.choice([1, 1, 5, 7]))  # index measures: no jump at 0
    k = int(rng.choice([21, 31]))
    regs = _regs(rng, n, p, kind)
    ctx.set_sketches(regs)
    want = oracle.dist_tri(regs, estim, rt, k)
    try:
        if case % 2:
            ctx.set_option("cum_budget_bytes", 1 << int(rng.integers(22, 27)))  # several bands
        got = ctx.dist_rows(estim=estim, result_type=rt, k=k)
        _close(got, want)
        rb = int(rng.integers(0, n - 300))
        re = int(rng.integers(rb + 200, n + 1))
        lo, span = dashing_amd.tri_span(n, 0, rb), dashing_amd.tri_span(n, rb, re)
        assert ctx.dist_rows(rb, re, estim=estim, result_type=rt, k=k).tobytes() == got[lo : lo + span].tobytes()
        pd = torch.full((span,), -9.0, dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        ctx.dist_rows_parts_device_async(pd.data_ptr(), rb, re, int(rng.integers(2, 9)), estim=estim, result_type=rt, k=k)
        ctx.wait()
        assert pd.cpu().numpy().tobytes() == got[lo : lo + span].tobytes()
        nn = int(rng.integers(1, 12))
        si, sv = ctx.knn(nn, estim=estim, result_type=rt, k=k)
        ctx.set_option("knn_square_budget_bytes", 0)
        bi, bv = ctx.knn(nn, estim=estim, result_type=rt, k=k)
        assert (si == bi).all() and (sv.view(np.uint32) == bv.view(np.uint32)).all()
    finally:
        ctx.set_option("knn_square_budget_bytes", 96 << 30)
        ctx.set_option("cum_budget_bytes", 8 << 30)

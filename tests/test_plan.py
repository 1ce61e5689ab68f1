"""CPU: the pure-host planner of the compare path (dashing_amd/csrc/plan.cpp) -- the schedule that replaces dist_loop /
perform_core_op (src/sketch_and_cmp.h:785-880, :699-710): column layouts, tiles, bands, parts, work items, row partitions.
dshh_plan_check (csrc/host/plan_capi.cpp) builds layout + plan exactly as engine.hip does and checks the contract: every
wanted pair owned by exactly one (tile, lane), each tile's dense plane range exact for every pair in it, bands within
the C(v) budget, parts completing in order, items covering every tile's chunks once, the two device lists consistent."""
import ctypes as C
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def host():
    lib = C.CDLL(os.path.join(ROOT, "dashing_amd", "libdashing_host.so"))
    u64, i32, u32, vp = C.c_uint64, C.c_int, C.c_uint32, C.c_void_p
    lib.dshh_plan_check.argtypes = [u64, vp, i32, i32, u64, u64, u64, u64, u32, i32, i32, u64, i32, i32, i32, vp, C.c_char_p, C.c_size_t]
    lib.dsh_tri_span.restype = u64
    lib.dsh_tri_span.argtypes = [u64, u64, u64]
    lib.dsh_tri_index.restype = u64
    lib.dsh_tri_index.argtypes = [u64, u64, u64]
    lib.dsh_balance_rows.argtypes = [u64, u32, vp]
    lib.dsh_partition_rows.argtypes = [u64, u32, u32, vp]
    lib.dsh_range_parts.argtypes = [u64, u64, u64, u32, vp, C.POINTER(u32)]
    return lib


def make_keys(rng, n, p, spread=6):
    """per-sketch keys as k_selfhist_card writes them: lo <= L <= T <= hi <= 64 - p + 1"""
    q1 = 64 - p + 1
    lo = rng.integers(0, max(1, q1 - spread), n)
    w = rng.integers(0, spread + 1, (3, n))
    L = np.minimum(lo + w[0], q1)
    T = np.minimum(L + w[1], q1)
    hi = np.minimum(T + w[2], q1)
    return (hi.astype(np.uint32) << 18 | T.astype(np.uint32) << 12 | L.astype(np.uint32) << 6 | lo.astype(np.uint32)).astype(np.uint32)


def check(host, keys, mode=0, sorted_=1, rb=0, re=None, cb=0, ce=0, nparts=1, want_parts=0, p=12, budget=8 << 30, lockstep=1,
          nsplit=0, chunks=64):
    n = len(keys)
    re = n if re is None else re
    stats = np.zeros(8, np.uint64)
    err = C.create_string_buffer(512)
    rc = host.dshh_plan_check(n, keys.ctypes.data, mode, sorted_, rb, re, cb, ce, nparts, want_parts, p, budget, lockstep, nsplit,
                              chunks, stats.ctypes.data, err, 512)
    assert rc == 0, err.value.decode()
    return dict(tiles=int(stats[0]), bands=int(stats[1]), items=int(stats[2]), parts=int(stats[3]), planes_x100=int(stats[4]),
                npad=int(stats[5]), P=int(stats[6]))


def test_full_triangle_sorted_and_identity(host):
    rng = np.random.default_rng(1)
    for n in (1, 2, 127, 128, 129, 700, 1500):
        keys = make_keys(rng, n, 12)
        for sorted_ in (1, 0):
            st = check(host, keys, sorted_=sorted_)
            nt = (n + 127) // 128
            assert st["tiles"] == nt * (nt + 1) // 2
    # the key order is what makes tiles cheap: fewer planes per tile than the identity layout on the same keys
    keys = make_keys(rng, 3000, 12, spread=10)
    assert check(host, keys, sorted_=1)["planes_x100"] < check(host, keys, sorted_=0)["planes_x100"]


def test_row_ranges_cover_their_span_exactly_once(host):
    rng = np.random.default_rng(2)
    for _ in range(40):
        n = int(rng.integers(2, 1800))
        keys = make_keys(rng, n, int(rng.choice([10, 12, 14])))
        rb = int(rng.integers(0, n))
        re = int(rng.integers(rb, n + 1))
        check(host, keys, sorted_=int(rng.integers(0, 2)), rb=rb, re=re, p=12)


def test_parts_any_range_length(host):
    """a call with parts always gets the key-ordered layout of its range in exactly dsh_range_parts' parts, however short
    (ranks 0-4 of n = 10 000 over 8 ranks hold 640-896 rows)"""
    rng = np.random.default_rng(3)
    n = 2600
    keys = make_keys(rng, n, 12)
    for (rb, re) in ((0, 640), (640, 1280), (0, 100), (2500, 2600), (2599, 2600), (0, n), (1000, 1000)):
        for nparts in (1, 2, 8):
            st = check(host, keys, rb=rb, re=re, nparts=nparts, want_parts=1)
            b = np.zeros(nparts + 1, np.uint64)
            k = C.c_uint32()
            assert host.dsh_range_parts(n, rb, re, nparts, b.ctypes.data, C.byref(k)) == 0
            if rb < re:
                assert st["parts"] == k.value >= 1, (rb, re, nparts, st, k.value)
                assert all((int(b[q]) - rb) % 128 == 0 for q in range(1, k.value))
    # bounds of dsh_balance_rows at the headline size: every rank's range, in parts
    n = 10000
    keys = make_keys(rng, n, 14)
    bounds = np.zeros(9, np.uint64)
    assert host.dsh_balance_rows(n, 8, bounds.ctypes.data) == 0
    for r in (0, 3, 7):
        st = check(host, keys, rb=int(bounds[r]), re=int(bounds[r + 1]), nparts=8, want_parts=1, p=14)
        assert st["parts"] >= 1


def test_row_sorted_parts(host):
    """short ranges under dsh_exchange_*: the wanted rows ONE key-ordered run (homogeneous 128-row blocks), parts = runs of
    whole tile rows of that order, the rank's buffer in key order (rowoff) -- same pairs, each exactly once"""
    rng = np.random.default_rng(8)
    n = 2600
    keys = make_keys(rng, n, 12, spread=10)
    for (rb, re) in ((0, 640), (640, 1280), (1280, 1290), (2048, 2599), (2599, 2600), (0, n)):
        for nparts in (1, 2, 8):
            st = check(host, keys, mode=3, rb=rb, re=re, nparts=nparts, want_parts=1)
            assert 1 <= st["parts"] <= nparts
    # the point of it: a short range in 8 parts of consecutive rows has one 128-row block per part (no ordering at all)
    keys = make_keys(rng, 10000, 14, spread=12)
    plain = check(host, keys, rb=0, re=640, nparts=8, want_parts=1, p=14)
    rs = check(host, keys, mode=3, rb=0, re=640, nparts=8, want_parts=1, p=14)
    assert rs["tiles"] == plain["tiles"] and rs["planes_x100"] < plain["planes_x100"], (rs, plain)


def test_bands_follow_the_scratch_budget_and_large_parts_cut_them(host):
    rng = np.random.default_rng(4)
    n = 1500
    keys = make_keys(rng, n, 12)
    one = check(host, keys)
    assert one["bands"] == 1
    per_tile = 128 * 128 * 2 * max(one["P"], 1)
    few = check(host, keys, budget=7 * per_tile)
    assert few["bands"] == -(-one["tiles"] // 7)
    tiny = check(host, keys, budget=1 << 10)  # smaller than one tile: one tile per band
    assert tiny["bands"] == one["tiles"]
    check(host, keys, budget=5 * per_tile, nparts=4, want_parts=1)  # parts inside and across bands
    # a part of >= 2048 tiles ends its band (so it can leave before the rest is computed)
    n = 12000
    keys = make_keys(rng, n, 10)
    st = check(host, keys, nparts=2, want_parts=1, p=10)
    assert st["parts"] == 2 and st["bands"] == 2


def test_rectangles_and_sorted_row_bands(host):
    rng = np.random.default_rng(5)
    n = 900
    keys = make_keys(rng, n, 12)
    check(host, keys, mode=1, rb=100, re=300, cb=0, ce=n)
    check(host, keys, mode=1, rb=0, re=1, cb=899, ce=900)
    check(host, keys, mode=1, rb=5, re=5, cb=0, ce=10)
    for (rb, re) in ((0, 256), (256, 900), (0, 900), (128, 129)):
        check(host, keys, mode=2, rb=rb, re=re)


def test_item_options(host):
    rng = np.random.default_rng(6)
    keys = make_keys(rng, 1000, 14, spread=12)
    base = check(host, keys, p=14)
    for kw in (dict(nsplit=1), dict(nsplit=5), dict(chunks=1), dict(chunks=16), dict(chunks=100000), dict(lockstep=0), dict(p=8), dict(p=4), dict(p=17)):
        st = check(host, keys, **{"p": 14, **kw})
        assert st["tiles"] == base["tiles"]


def test_triangle_arithmetic_and_partitions(host):
    # distmat/distmat.h:260-264
    for n in (2, 3, 10, 1000):
        for i in range(min(n, 40)):
            for j in range(i + 1, min(n, i + 5)):
                assert host.dsh_tri_index(n, i, j) == i * (2 * n - i - 1) // 2 + j - (i + 1)
        assert host.dsh_tri_span(n, 0, n) == n * (n - 1) // 2
        assert host.dsh_tri_span(n, 1, 1) == 0 and host.dsh_tri_span(n, 0, n + 7) == n * (n - 1) // 2
    for n in (0, 1, 5, 129, 10000, 100000):
        for world in (1, 2, 3, 8):
            b = np.zeros(world + 1, np.uint64)
            assert host.dsh_balance_rows(n, world, b.ctypes.data) == 0
            assert b[0] == 0 and b[world] == n and all(b[r] <= b[r + 1] for r in range(world))
            assert all(int(b[r]) % 128 == 0 for r in range(1, world) if b[r] < n)
            assert sum(host.dsh_tri_span(n, int(b[r]), int(b[r + 1])) for r in range(world)) == n * (n - 1) // 2
            assert host.dsh_partition_rows(n, world, 128, b.ctypes.data) == 0
            assert b[0] == 0 and b[world] == n and all(b[r] <= b[r + 1] for r in range(world))
    # balance at the headline size: no rank more than 15 % above the mean tile count (128-row granularity)
    n, world = 10000, 8
    b = np.zeros(world + 1, np.uint64)
    host.dsh_balance_rows(n, world, b.ctypes.data)
    nt = (n + 127) // 128
    tiles = [sum(nt - t for t in range(int(b[r]) // 128, (int(b[r + 1]) + 127) // 128)) for r in range(world)]
    assert max(tiles) <= 1.15 * sum(tiles) / world, tiles

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
    stats = np.zeros(10, np.uint64)
    err = C.create_string_buffer(512)
    rc = host.dshh_plan_check(n, keys.ctypes.data, mode, sorted_, rb, re, cb, ce, nparts, want_parts, p, budget, lockstep, nsplit,
                              chunks, stats.ctypes.data, err, 512)
    assert rc == 0, err.value.decode()
    return dict(tiles=int(stats[0]), bands=int(stats[1]), items=int(stats[2]), parts=int(stats[3]), planes_x100=int(stats[4]),
                npad=int(stats[5]), P=int(stats[6]), rounds=int(stats[7]), frags=int(stats[8]))


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


# ---- row sets: a rank's rows as a range + top-up tile rows (plan.h; dsh_balance_rowsets) ---------------------------------
PREP = 0.4  # dsh_balance_rowsets' default weight of a rank's own prepare: tiles per 128 columns of its plane matrix


def _rowset_api(host):
    u64, u32, vp = C.c_uint64, C.c_uint32, C.c_void_p
    host.dsh_balance_rowsets.argtypes = [u64, u32, C.c_int, C.c_int, C.c_int, vp, u32, C.POINTER(u32)]
    host.dsh_rowsets_from_bounds.argtypes = [vp, u32, vp]
    host.dsh_rowsets_rank.argtypes = [u64, vp, u32, vp, u32, C.POINTER(u32), C.POINTER(u64), C.POINTER(u64)]
    host.dshh_plan_check_rowset.argtypes = [u64, vp, vp, u32, C.c_int, u32, C.c_int, u64, vp, C.c_char_p, C.c_size_t]
    return host


def balance_rowsets(host, n, world, prep=-1, dst=-1, bonus=-1):
    _rowset_api(host)
    words = C.c_uint32(0)
    assert host.dsh_balance_rowsets(n, world, prep, dst, bonus, None, 0, C.byref(words)) == 0
    tab = np.zeros(words.value, np.uint64)
    assert host.dsh_balance_rowsets(n, world, prep, dst, bonus, tab.ctypes.data, len(tab), C.byref(words)) == 0
    return tab


def rank_rows(host, n, tab, r):
    segs = np.zeros(2 * 64, np.uint64)
    ns, pairs, tiles = C.c_uint32(0), C.c_uint64(0), C.c_uint64(0)
    assert host.dsh_rowsets_rank(n, tab.ctypes.data, r, segs.ctypes.data, 64, C.byref(ns), C.byref(pairs), C.byref(tiles)) == 0
    return [(int(segs[2 * i]), int(segs[2 * i + 1])) for i in range(ns.value)], pairs.value, tiles.value


def check_rowset(host, keys, tab, rank, rowsorted, nparts, p=12, budget=8 << 30):
    stats = np.zeros(10, np.uint64)
    err = C.create_string_buffer(512)
    rc = host.dshh_plan_check_rowset(len(keys), keys.ctypes.data, tab.ctypes.data, rank, rowsorted, nparts, p, budget,
                                     stats.ctypes.data, err, 512)
    assert rc == 0, err.value.decode()
    return dict(tiles=int(stats[0]), bands=int(stats[1]), items=int(stats[2]), parts=int(stats[3]), planes_x100=int(stats[4]), rounds=int(stats[7]), frags=int(stats[8]))


def test_balanced_rowsets_partition_every_row_once(host):
    """every row has exactly one owner, every rank's segments are tile-aligned when it has more than one, the pairs of the
    ranks add up to the triangle, and the largest cost of any rank (tiles + 0.4 per 128 columns of its plane matrix: its own
    prepare) is never above the contiguous balance's"""
    _rowset_api(host)
    rng = np.random.default_rng(11)
    cases = [(10000, 8), (10000, 4), (10000, 2), (10000, 3), (1000, 8), (5000, 8), (32768, 8), (40000, 8), (300, 8), (129, 2)]
    cases += [(int(rng.integers(2, 30000)), int(rng.integers(1, 17))) for _ in range(60)]
    for n, world in cases:
        tab = balance_rowsets(host, n, world)
        assert int(tab[0]) == world
        ns = int(tab[1])
        seg, own = tab[2:3 + ns], tab[3 + ns:3 + 2 * ns]
        assert seg[0] == 0 and seg[-1] == n and np.all(np.diff(seg.astype(np.int64)) >= 0) and np.all(own < world)
        owner_of = np.full(n, -1, np.int64)
        tot_pairs, tiles, cost = 0, [], []
        nt = (n + 127) // 128
        for r in range(world):
            segs, pairs, t = rank_rows(host, n, tab, r)
            cost.append(t + PREP * (nt - segs[0][0] // 128) if segs else 0.0)
            for b, e in segs:
                assert np.all(owner_of[b:e] == -1)
                owner_of[b:e] = r
                if len(segs) > 1:
                    assert b % 128 == 0 and (e % 128 == 0 or e == n)
            assert pairs == sum(host.dsh_tri_span(n, b, e) for b, e in segs)
            tot_pairs += pairs
            tiles.append(t)
        assert np.all(owner_of >= 0) and tot_pairs == n * (n - 1) // 2
        b = np.zeros(world + 1, np.uint64)
        host.dsh_balance_rows(n, world, b.ctypes.data)
        contiguous = [sum(nt - t for t in range(int(b[r]) // 128, (int(b[r + 1]) + 127) // 128)) for r in range(world)]
        ccost = [contiguous[r] + (PREP * (nt - int(b[r]) // 128) if b[r + 1] > b[r] else 0.0) for r in range(world)]
        assert sum(tiles) == nt * (nt + 1) // 2 or sum(tiles) == sum(contiguous)
        assert max(cost) <= max(ccost) + 1e-6, (n, world, cost, ccost)


def test_balanced_rowsets_c3_over_8_ranks_is_level(host):
    """the case the top-ups exist for (BASELINE configs[2] over 8 GPUs): 3 160 tiles, 395 per rank -- contiguous 128-row
    ranges leave the largest rank at 432+; with the bottom tile rows dealt every rank is within a few tiles of the others
    once its own prepare (0.4 tile-equivalents per 128 columns of its plane matrix, the default weight) is counted"""
    n, world = 10000, 8
    tab = balance_rowsets(host, n, world)
    nt = (n + 127) // 128
    cost = []
    for r in range(world):
        segs, _, tiles = rank_rows(host, n, tab, r)
        cost.append(tiles + PREP * (nt - segs[0][0] // 128))
    assert max(cost) - min(cost) <= 12, cost
    assert max(cost) <= 1.02 * (sum(cost) / world), cost
    assert any(len(rank_rows(host, n, tab, r)[0]) > 1 for r in range(world))


def test_rowset_plans_cover_every_pair_exactly_once(host):
    """every rank of a balanced row-set table, as a SOURCE of the exchange (row-sorted parts) and as the DESTINATION (one
    part in final order): each of its pairs owned by exactly one (tile, lane), no pair of another rank's row computed,
    parts complete in order; over all ranks the tiles add up to the whole triangle's"""
    rng = np.random.default_rng(12)
    for n, world, nparts in [(1500, 4, 3), (2000, 4, 8), (1100, 3, 2), (2000, 5, 4), (1400, 2, 8), (2500, 4, 2)]:
        keys = make_keys(rng, n, 12)
        tab = balance_rowsets(host, n, world)
        tiles = 0
        multi = 0
        for r in range(world):
            segs, _, t = rank_rows(host, n, tab, r)
            multi += len(segs) > 1
            for rowsorted in (1, 0):
                st = check_rowset(host, keys, tab, r, rowsorted, nparts)
                assert st["tiles"] == t
                if rowsorted and t:
                    assert 1 <= st["parts"] <= nparts
            tiles += t
        nt = (n + 127) // 128
        assert tiles == nt * (nt + 1) // 2
        assert multi > 0, "no rank of this case holds a top-up segment: the case does not test what it is for"


def test_rowset_tables_reject_unaligned_extra_segments(host):
    _rowset_api(host)
    n = 1000
    # rank 0: [0,300) and [700,1000); rank 1: [300,700) -- rank 0 has an extra segment on a non-multiple of 128
    tab = np.array([2, 3, 0, 300, 700, 1000, 0, 1, 0], np.uint64)
    ns = C.c_uint32(0)
    assert host.dsh_rowsets_rank(n, tab.ctypes.data, 0, None, 0, C.byref(ns), None, None) != 0
    tab = np.array([2, 3, 0, 256, 768, 1000, 0, 1, 0], np.uint64)
    assert host.dsh_rowsets_rank(n, tab.ctypes.data, 0, None, 0, C.byref(ns), None, None) == 0 and ns.value == 2
    # contiguous bounds as a table: any alignment
    b = np.array([0, 301, 1000], np.uint64)
    t2 = np.zeros(3 + 2 * 2, np.uint64)
    assert host.dsh_rowsets_from_bounds(b.ctypes.data, 2, t2.ctypes.data) == 0
    assert host.dsh_rowsets_rank(n, t2.ctypes.data, 1, None, 0, C.byref(ns), None, None) == 0 and ns.value == 1


def test_tail_bands_cut_the_tile_kernel_at_whole_rounds(host):
    """a small job with parts (the exchange): the tile kernel is cut into a head and a tail launch so that the head's parts
    can travel while the tail computes -- at a multiple of a round of 512 work items, never at the cost of a round"""
    rng = np.random.default_rng(21)
    cut = 0
    for n, world in [(2000, 2), (1900, 2), (2100, 3), (2000, 1)]:
        keys = make_keys(rng, n, 14, spread=8)
        tab = balance_rowsets(host, n, world)
        for r in range(world):
            st = check_rowset(host, keys, tab, r, 1 if world > 1 else 0, 8, p=14)
            if not st["tiles"]:
                continue
            # no band rounds up on its own: whole rounds of whole items (+ one short round of overflow fragments)
            assert st["rounds"] <= -(-st["items"] // 512) + (1 if st["frags"] else 0), (n, world, r, st)
            if st["items"] > 2 * 512 and st["items"] <= 16 * 512 and world > 1:
                assert st["bands"] >= 2, (n, world, r, st)
                cut += 1
    assert cut >= 2


def test_a_second_tail_band_keeps_its_own_quota_of_rounds(host):
    """14 rounds = 8 + 5 + 1: what the head leaves of its last round must not slide into the middle band and push it over a
    round (the plan then fell back to 13 + 1 and the rank's first part left a millisecond later: 4 ranks of BASELINE
    configs[2], profiles/rd5t)"""
    rng = np.random.default_rng(33)
    seen = 0
    for n in (7000, 8000, 9000, 10000):
        for spread in (2, 3, 4):
            keys = make_keys(rng, n, 14, spread=spread)
            tab = balance_rowsets(host, n, 4, dst=0)
            for r in range(1, 4):
                st = check_rowset(host, keys, tab, r, 1, 8, p=14)
                rounds = -(-round(st["tiles"] * st["planes_x100"] / 100) // 512)
                if 13 <= rounds <= 16:
                    assert st["bands"] == 3, (n, spread, r, st)
                    seen += 1
    assert seen >= 5


def test_a_long_row_sorted_range_ends_in_a_run_of_its_last_rows(host):
    """2-4 ranks: a row-sorted range that reaches far down the triangle is key-ordered as TWO runs, so that the tile rows the
    rank's last launch computes hold its shortest rows (little output behind the last kernel).  Short ranges (8 ranks)
    stay one run.  The plans still cover every pair exactly once (the checker), at a few hundredths of a plane per tile."""
    host.dshh_rowsorted_split.restype = C.c_uint64
    host.dshh_rowsorted_split.argtypes = [C.c_uint64] * 3
    n = 10000
    for world, split_ranks in ((2, {1}), (3, {2}), (4, {3}), (8, set())):
        tab = balance_rowsets(host, n, world, dst=0)
        for r in range(1, world):
            (rb, re), *_ = rank_rows(host, n, tab, r)[0]
            x = host.dshh_rowsorted_split(n, rb, re)
            assert (x < re) == (r in split_ranks), (world, r, rb, re, x)
            if x < re:
                assert rb < x and (x - rb) % 128 == 0
                # the rows behind the cut are short: at most 0.65 of the range's mean output per row
                assert (n - 1 - (x + re - 1) / 2) <= 0.65 * (n - 1 - (rb + re - 1) / 2)
    rng = np.random.default_rng(8)
    for n, world in ((6000, 2), (9000, 3)):
        keys = make_keys(rng, n, 12)
        tab = balance_rowsets(host, n, world, dst=0)
        for r in range(1, world):
            st = check_rowset(host, keys, tab, r, 1, 8)
            assert st["parts"] >= 2


def test_the_destination_of_an_exchange_takes_a_bonus(host):
    """dsh_balance_rowsets(dst): the rank that receives sends nothing, so it holds ~12 % (or the share asked for) more tiles
    than the mean and the others correspondingly fewer -- still every row with one owner, every boundary aligned"""
    n, world = 10000, 4
    nt = (n + 127) // 128
    base = [rank_rows(host, n, balance_rowsets(host, n, world), r)[2] for r in range(world)]
    # (the default applies where a rank holds at least 16 tile rows: 79 tile rows over 4 ranks, not over 8)
    assert [rank_rows(host, n, balance_rowsets(host, n, 8, -1, 0, -1), r)[2] for r in range(8)] == \
           [rank_rows(host, n, balance_rowsets(host, n, 8), r)[2] for r in range(8)]
    for dst, bonus in ((0, -1), (2, -1), (3, 150), (0, 0)):
        tab = balance_rowsets(host, n, world, -1, dst, bonus)
        tiles = [rank_rows(host, n, tab, r)[2] for r in range(world)]
        assert sum(tiles) == nt * (nt + 1) // 2
        mean = sum(tiles) / world
        share = 0.12 if bonus < 0 else bonus / 1000.0
        others = [t for r, t in enumerate(tiles) if r != dst]
        if share == 0:
            assert tiles == base
        else:
            assert tiles[dst] - sum(others) / len(others) >= 0.6 * share * mean, (dst, bonus, tiles)
            assert max(others) <= max(base), (tiles, base)
        owner = np.full(n, -1)
        for r in range(world):
            for b, e in rank_rows(host, n, tab, r)[0]:
                assert np.all(owner[b:e] == -1)
                owner[b:e] = r
        assert np.all(owner >= 0)


def test_overflow_fragments_replace_a_nearly_empty_round(host):
    """a band of one-plane items a little above a multiple of 512 (the tile kernel's round): the items left over are cut
    into equal fragments of a plane, at most one round of them, behind a whole number of rounds of whole items -- every
    chunk of every tile still covered exactly once (the contract check), nothing when the band fits or overflows by much"""
    rng = np.random.default_rng(31)
    seen = {"frag": 0, "none": 0}
    for n in range(1500, 2600, 100):
        keys = make_keys(rng, n, 14, spread=6)
        st = check(host, keys, sorted_=1, p=14)
        if st["frags"]:
            seen["frag"] += 1
            whole = st["items"] - st["frags"]
            assert whole % 512 == 0 and 2 <= st["frags"] <= 512 and st["bands"] == 1, st
        else:
            seen["none"] += 1
    assert seen["frag"] >= 2 and seen["none"] >= 1, seen

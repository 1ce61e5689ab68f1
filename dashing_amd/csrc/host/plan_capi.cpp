// plan_capi.cpp -- CPU test hook of the pure-host planner (../plan.cpp): builds a layout and a pair plan for given
// per-sketch keys exactly as engine.hip does and checks the plan against its contract, so the schedule that replaces
// dist_loop / perform_core_op (src/sketch_and_cmp.h:785-880, :699-710) is unit-tested without a GPU
// (tests/test_plan.py).  Built into libdashing_host.so; not part of the GPU C-ABI.
#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../plan.h"

using namespace dsh;
using namespace dsh::plan;

namespace {

struct Err {
    char *buf;
    size_t cap;
    int fail(const char *fmt, ...)
    {
        if (buf && cap) {
            va_list ap;
            va_start(ap, fmt);
            vsnprintf(buf, cap, fmt, ap);
            va_end(ap);
        }
        return -1;
    }
};

}  // namespace

// mode: 0 triangle rows [rb, re) (want_sorted decides the layout), 1 rectangle rows [rb,re) x cols [cb,ce) (identity
// layout), 2 sorted_rows (whole sorted layout, rows [rb,re) index plane columns: shards / band-wise kNN), 3 triangle
// rows in ROW-SORTED parts (the wanted rows one key-ordered run, parts = runs of whole tile rows of that order).
// stats out: [0] tiles, [1] bands, [2] items, [3] parts with an event, [4] planes per tile x 100, [5] Npad, [6] P,
// [7] rounds of 512 items summed over the bands (a round of fragments counted as one), [8] overflow fragments.
// extra: further wanted segments {b0, e0, ...} (plan.h, row sets; modes 0 and 3 with a sorted layout only)
static int plan_check(uint64_t n, const uint32_t *keys, int mode, int want_sorted, uint64_t rb, uint64_t re, uint64_t cb,
                      uint64_t ce, uint32_t nparts, int want_parts, int p, uint64_t cum_budget, int lockstep, int nsplit,
                      int ls_item_chunks, const std::vector<uint64_t> &extra, uint64_t *stats, char *err, size_t cap)
{
    Err E{err, cap};
    if (re > n) re = n;
    const bool rowsorted = mode == 3;
    if (rowsorted) mode = 0, want_sorted = 1, want_parts = 1;
    if (mode == 1 || !want_sorted) want_sorted = mode == 2 ? 1 : 0;
    if (mode == 2) want_sorted = 1;
    std::vector<uint64_t> parts;
    uint64_t lrb = 0, lre = n;
    if (want_sorted && mode == 0) lrb = rb, lre = re;
    if (want_sorted && !rowsorted) range_parts(n, lrb, lre, std::max<uint32_t>(want_parts ? nparts : 1, 1), parts);
    if (!extra.empty() && !(want_sorted && mode == 0)) return E.fail("extra segments need a sorted triangle layout");
    Layout L;
    build_layout(keys, n, want_sorted, lrb, lre, parts, L, rowsorted ? std::max<uint32_t>(nparts, 1) : 0, extra.empty() ? nullptr : &extra);
    // the wanted segments, main range first
    std::vector<std::pair<uint64_t, uint64_t>> wsegs;  // (the wanted order: extra segments first, then the main range)
    wanted_order(n, lrb, lre, &extra, wsegs, rowsorted);
    auto in_rows = [&](uint64_t i) {
        for (auto &w : wsegs)
            if (i >= w.first && i < w.second) return true;
        return false;
    };
    uint64_t nw = 0, wspan = 0;
    for (auto &w : wsegs) nw += w.second - w.first, wspan += tri_span(n, w.first, w.second);
    if (want_sorted && mode == 0 && L.nwanted != nw) return E.fail("nwanted %llu != %llu", (unsigned long long)L.nwanted, (unsigned long long)nw);
    if (rowsorted) {  // one key-ordered run per wanted segment, cuts on whole tile rows of the wanted order, row offsets of the rank's buffer
        auto skey = [&](uint32_t g) { return ((uint32_t)key_T(keys[g]) << 12) | ((uint32_t)key_L(keys[g]) << 6) | (uint32_t)key_hi(keys[g]); };
        for (auto &w : wsegs)
            for (uint64_t s = w.first - lrb + 1; s < w.second - lrb; ++s)
                if (skey(L.perm[s - 1]) > skey(L.perm[s])) return E.fail("row-sorted: a wanted segment is not one key-ordered run at %llu", (unsigned long long)s);
        if (L.part_w.empty() || L.part_w.front() != 0 || L.part_w.back() != nw) return E.fail("row-sorted: part cuts do not span the wanted rows");
        for (size_t q = 1; q + 1 < L.part_w.size(); ++q) {  // every cut at the start of a tile row of the wanted order
            if (L.part_w[q] <= L.part_w[q - 1]) return E.fail("row-sorted: cut %zu at %llu", q, (unsigned long long)L.part_w[q]);
            uint64_t w0 = 0;
            bool ok = false;
            for (auto &w : wsegs) {
                if (L.part_w[q] >= w0 && L.part_w[q] < w0 + (w.second - w.first)) ok = (L.part_w[q] - w0) % kTile == 0;
                w0 += w.second - w.first;
            }
            if (!ok) return E.fail("row-sorted: cut %zu at %llu is not at a tile row of the wanted order", q, (unsigned long long)L.part_w[q]);
        }
        if (L.part_w.size() - 1 > std::max<uint32_t>(nparts, 1)) return E.fail("row-sorted: more parts than asked for");
        uint64_t acc = 0, w_ = 0;
        if (L.rowoff_w.size() != nw + 1) return E.fail("row-sorted: rowoff_w size");
        for (auto &w : wsegs)
            for (uint64_t s = w.first - lrb; s < w.second - lrb; ++s, ++w_) {
                if (s >= L.rowoff.size() || L.rowoff[s] != acc || L.rowoff_w[w_] != acc) return E.fail("row-sorted: rowoff[%llu]", (unsigned long long)s);
                acc += n - 1 - L.perm[s];
            }
        if (L.rowoff_w.back() != acc || acc != wspan) return E.fail("row-sorted: the rows do not add up to the span");
    }
    // ---- layout: perm is a permutation of the columns; the parts hold exactly their rows; block stats are right
    const uint64_t col0 = want_sorted ? lrb : 0;
    if (L.ncols != n - col0) return E.fail("ncols %llu != %llu", (unsigned long long)L.ncols, (unsigned long long)(n - col0));
    if (L.Npad % kTile || L.Npad < L.ncols || L.Npad >= L.ncols + kTile) return E.fail("bad Npad %u", L.Npad);
    std::vector<uint8_t> seen(n, 0);
    for (uint64_t s = 0; s < L.ncols; ++s) {
        const uint32_t g = L.perm[s];
        if (g < col0 || g >= n || seen[g]) return E.fail("perm[%llu] = %u is out of range or repeated", (unsigned long long)s, g);
        seen[g] = 1;
    }
    if (want_sorted) {
        uint64_t s = 0;
        for (size_t q = 0; q + 1 < L.parts.size(); ++q)
            for (uint64_t r = L.parts[q]; r < L.parts[q + 1]; ++r, ++s)
                if (L.perm[s] < L.parts[q] || L.perm[s] >= L.parts[q + 1])
                    return E.fail("column %llu holds sketch %u, not a row of part %zu", (unsigned long long)s, L.perm[s], q);
        if (!rowsorted && extra.empty()) {
            if (L.part_pos.size() != L.parts.size()) return E.fail("part positions and parts differ in number");
            for (size_t q = 0; q < L.parts.size(); ++q)
                if (L.part_pos[q] != L.parts[q] - lrb) return E.fail("part position %zu", q);
        }
        for (; s < L.ncols; ++s)
            if (L.perm[s] < lre) return E.fail("column %llu (after the wanted rows) holds wanted row %u", (unsigned long long)s, L.perm[s]);
        // every run holds exactly its rows: a wanted segment's positions hold its rows; every block is wanted or not as a whole
        for (auto &w : wsegs)
            for (uint64_t s2 = w.first - lrb; s2 < w.second - lrb; ++s2)
                if (L.perm[s2] < w.first || L.perm[s2] >= w.second) return E.fail("position %llu of a wanted segment holds row %u", (unsigned long long)s2, L.perm[s2]);
        if (!extra.empty())
            for (uint64_t s2 = 0; s2 < L.ncols; ++s2) {
                // rows to the right of a position are larger unless they share its run: position order = row order between runs
                const uint64_t b = s2 / kTile * kTile;
                if (in_rows(L.perm[s2]) != in_rows(L.perm[b])) return E.fail("block %llu mixes wanted and other rows", (unsigned long long)(b / kTile));
            }
        if (L.whole)
            for (uint64_t i = 0; i < n; ++i)
                if (L.perm[n + L.perm[i]] != i) return E.fail("inverse permutation wrong at %llu", (unsigned long long)i);
    }
    Tuning tu;
    const uint64_t m = 1ull << p;
    tu.W = (uint32_t)std::max<uint64_t>(1, m / 32);
    tu.kc = tu.W >= 32 ? 32 : 16;
    tu.cum_bytes = p <= 15 ? 2 : 4;
    tu.cum_budget = cum_budget;
    tu.nsplit = nsplit;
    tu.lockstep = lockstep && tu.W >= (uint32_t)tu.kc;
    tu.ls_item_chunks = ls_item_chunks;
    PairQuery q;
    q.rect = mode == 1;
    q.sorted_rows = mode == 2;
    q.want_parts = want_parts && mode == 0 && want_sorted;
    q.row_begin = rb;
    q.row_end = re;
    q.col_begin = cb;
    q.col_end = ce;
    PairPlan pp;
    const bool any = build_pairs(L, q, tu, pp);
    for (int i = 0; i < 9; ++i) stats[i] = 0;
    stats[5] = L.Npad;
    stats[6] = L.P;
    // ---- every wanted pair is owned by exactly one (tile, lane) -- the `active` predicate of k_finalize
    std::vector<uint8_t> hit;
    uint64_t wanted = 0;
    auto orig = [&](uint64_t s) -> uint64_t { return L.perm[s]; };
    if (mode == 1) {
        const uint64_t nr = re > rb ? re - rb : 0, nc = ce > cb ? ce - cb : 0;
        wanted = nr * nc;
        hit.assign(wanted, 0);
    } else {
        hit.assign(n * n, 0);
        for (uint64_t i = rb; i < re; ++i) wanted += n - 1 - i;
        for (size_t x = 0; x + 1 < extra.size(); x += 2)
            for (uint64_t i = extra[x]; i < extra[x + 1]; ++i) wanted += n - 1 - i;
        if (mode == 2) {
            wanted = 0;
            for (uint64_t si = rb; si < re; ++si) wanted += n - 1 - si;
        }
    }
    if (!any) {
        if (wanted) return E.fail("nothing planned although %llu pairs are wanted", (unsigned long long)wanted);
        return 0;
    }
    const std::vector<U4> &T = pp.T;
    uint64_t got = 0, planes_sum = 0;
    for (const U4 &t : T) {
        if (t.z > t.w || t.w > L.P) return E.fail("tile (%u,%u) has the plane range [%u,%u) of %u", t.x, t.y, t.z, t.w, L.P);
        planes_sum += t.w - t.z;
        for (uint32_t a = 0; a < kTile; ++a)
            for (uint32_t b = 0; b < kTile; ++b) {
                const uint64_t si = (uint64_t)t.x * kTile + a, sj = (uint64_t)t.y * kTile + b;
                if (si >= L.ncols || sj >= L.ncols) continue;
                const uint64_t i = orig(si), j = orig(sj);
                bool active;
                uint64_t slot;
                if (mode == 1) {
                    active = i >= rb && i < re && j >= cb && j < ce;
                    slot = active ? (i - rb) * (ce - cb) + (j - cb) : 0;
                } else if (mode == 2) {
                    active = si < sj && si >= rb && si < re;
                    slot = si * n + sj;
                } else {
                    // (k_finalize: with extra segments run_pairs hands it the rows [rb, n) -- every pair of a launched tile)
                    const uint64_t oi = std::min(i, j), oj = std::max(i, j);
                    active = si < sj && oi >= rb && oi < (extra.empty() ? re : n);
                    slot = oi * n + oj;
                    if (active && !extra.empty() && !in_rows(oi))
                        return E.fail("tile (%u,%u) computes the pair (%llu,%llu) of a row this rank does not hold", t.x, t.y, (unsigned long long)oi, (unsigned long long)oj);
                }
                if (!active) continue;
                if (hit[slot]++) return E.fail("pair slot %llu computed twice (tile %u,%u)", (unsigned long long)slot, t.x, t.y);
                ++got;
                // exactness of the tile's dense range for this pair: everything above T is listed by
                // one of the two sketches, everything below Lp is below both low thresholds or below both minima
                const uint32_t ka = keys[i], kb = keys[j];
                const int Tp = L.pbase + (int)t.w, Lp = L.pbase + (int)t.z;
                if (Tp < std::max(key_T(ka), key_T(kb))) return E.fail("tile (%u,%u): T %d below a sketch's threshold", t.x, t.y, Tp);
                if (Lp > std::max(std::max(key_lo(ka), key_lo(kb)), std::min(key_L(ka), key_L(kb))))
                    return E.fail("tile (%u,%u): Lp %d above what the low lists cover", t.x, t.y, Lp);
            }
    }
    if (got != wanted) return E.fail("%llu of %llu wanted pairs are covered", (unsigned long long)got, (unsigned long long)wanted);
    // ---- bands cover the tiles in order and respect the scratch budget (a single tile may exceed it)
    size_t at = 0;
    if (pp.bands.size() != pp.segs.size() || pp.bands.size() != pp.band_items.size()) return E.fail("band bookkeeping sizes differ");
    std::vector<int> part_done(pp.nparts, 0);
    int last_part = -1;
    for (size_t bi = 0; bi < pp.bands.size(); ++bi) {
        const auto &bd = pp.bands[bi];
        if (bd.first != at || bd.second <= bd.first || bd.second > T.size()) return E.fail("band %zu = [%zu,%zu) does not follow %zu", bi, bd.first, bd.second, at);
        at = bd.second;
        if (bd.second - bd.first > 1 && (bd.second - bd.first) * pp.per_tile_bytes > cum_budget)
            return E.fail("band %zu needs %llu bytes of C(v), budget %llu", bi, (unsigned long long)((bd.second - bd.first) * pp.per_tile_bytes), (unsigned long long)cum_budget);
        size_t sa = bd.first;
        for (const Seg &sg : pp.segs[bi]) {
            if (sg.b != sa || sg.e <= sg.b || sg.e > bd.second) return E.fail("segment [%zu,%zu) of band %zu does not follow %zu", sg.b, sg.e, bi, sa);
            sa = sg.e;
            if (sg.part >= 0) {
                if ((uint32_t)sg.part >= pp.nparts || part_done[sg.part]++ || sg.part != last_part + 1)
                    return E.fail("part %d completed out of order or twice", sg.part);
                last_part = sg.part;
            }
            // every tile of the segment belongs to the part that is current (parts are runs of the WANTED order)
            if (q.want_parts)
                for (size_t t = sg.b; t < sg.e; ++t) {
                    uint64_t pos0 = ~0ull;  // wanted rows in front of the tile's row block
                    for (size_t k = 0; k < L.wtr.size(); ++k)
                        if (T[t].x >= L.wtr[k].first && T[t].x < L.wtr[k].second) pos0 = L.wtr_w[k] + (uint64_t)(T[t].x - L.wtr[k].first) * kTile;
                    const int cur = last_part + (sg.part >= 0 ? 0 : 1);
                    if (cur < 0 || (size_t)cur + 1 >= L.part_w.size() || pos0 < L.part_w[cur] || pos0 >= L.part_w[cur + 1])
                        return E.fail("tile row %u is not in part %d", T[t].x, cur);
                }
        }
        if (sa != bd.second) return E.fail("segments of band %zu end at %zu, not %zu", bi, sa, bd.second);
        // items of the band cover every tile's chunk range exactly once
        const auto &bi_ = pp.band_items[bi];
        if (band_item_count(tu, pp, bi) != bi_.second - bi_.first) return E.fail("band %zu: band_item_count disagrees with the items built", bi);
        std::vector<uint32_t> next(bd.second - bd.first);
        for (size_t t = bd.first; t < bd.second; ++t) next[t - bd.first] = pp.chunks[t].x;
        std::vector<std::vector<std::pair<uint32_t, uint32_t>>> per(bd.second - bd.first);
        const uint32_t cpp_ = tu.W >= (uint32_t)tu.kc ? tu.W / tu.kc : 1;
        // overflow fragments (plan.h): the band's LAST band_frags items, each inside one plane, equal in length, at most one
        // round of them; the whole items in front of them a whole number of rounds
        const uint32_t nfr = bi < pp.band_frags.size() ? pp.band_frags[bi] : 0u;
        if (nfr > bi_.second - bi_.first || nfr > tu.round_items) return E.fail("band %zu: %u fragments", bi, nfr);
        if (nfr && (!tu.lockstep || (bi_.second - bi_.first - nfr) % tu.round_items)) return E.fail("band %zu: fragments behind %zu whole items", bi, bi_.second - bi_.first - nfr);
        stats[8] += nfr;
        for (size_t it = bi_.first; it < bi_.second; ++it) {
            const U4 &I = pp.items[it];
            if (I.x >= per.size() || I.y >= I.z) return E.fail("item %zu is malformed", it);
            const bool frag = it >= bi_.second - nfr;
            if ((I.w != 0) != frag) return E.fail("item %zu: fragment flag %u at the wrong place", it, I.w);
            if (frag && (I.y / cpp_ != (I.z - 1) / cpp_ || I.z - I.y != pp.items[bi_.second - 1].z - pp.items[bi_.second - 1].y))
                return E.fail("fragment %zu spans two planes or differs in length", it);
            per[I.x].emplace_back(I.y, I.z | (frag ? 0x80000000u : 0u));
        }
        for (size_t t = 0; t < per.size(); ++t) {
            std::sort(per[t].begin(), per[t].end());
            uint32_t x = pp.chunks[bd.first + t].x;
            for (size_t k = 0; k < per[t].size(); ++k) {
                auto &pr = per[t][k];
                const bool frag = (pr.second & 0x80000000u) != 0;
                pr.second &= 0x7FFFFFFFu;
                if (pr.first != x) return E.fail("items of tile %zu leave a gap or overlap at chunk %u", t, x);
                // whole items are stored, fragments added: a plane is covered by ONE whole item or by fragments only
                if (!frag && tu.W >= (uint32_t)tu.kc && ((pr.first % cpp_) || (pr.second % cpp_))) return E.fail("an item of tile %zu is not whole planes", t);
                if (frag && (pr.first % cpp_) && !(k > 0 && (per[t][k - 1].second & 0x7FFFFFFFu) == pr.first))
                    return E.fail("a fragment of tile %zu starts inside a plane that fragments do not cover from its start", t);
                x = pr.second;
            }
            if (x != pp.chunks[bd.first + t].y) return E.fail("items of tile %zu end at chunk %u, not %u", t, x, pp.chunks[bd.first + t].y);
        }
    }
    if (at != T.size()) return E.fail("bands end at %zu of %zu tiles", at, T.size());
    for (uint32_t qd = 0; qd < pp.nparts; ++qd)
        if (part_done[qd] != 1) return E.fail("part %u never completes", qd);
    if (pp.nparts) {  // the tile counts the signalling k_finalize waits for
        uint64_t tsum = 0;
        if (pp.part_tiles.size() != pp.nparts) return E.fail("part_tiles has %zu entries for %u parts", pp.part_tiles.size(), pp.nparts);
        for (uint32_t c : pp.part_tiles) tsum += c;
        if (tsum != T.size()) return E.fail("the parts hold %llu tiles of %zu", (unsigned long long)tsum, T.size());
    }
    if (q.want_parts && pp.nparts + 1 != L.part_w.size()) return E.fail("%u parts planned, layout has %zu", pp.nparts, L.part_w.size() - 1);
    // ---- the two device lists describe the same tiles
    std::vector<U4> dt(T.size()), df(T.size());
    emit_tile_lists(L, pp, dt.data(), df.data());
    for (size_t bi = 0; bi < pp.bands.size(); ++bi)
        for (const Seg &sg : pp.segs[bi]) {
            std::vector<uint8_t> used(pp.bands[bi].second - pp.bands[bi].first, 0);
            for (size_t t = sg.b; t < sg.e; ++t) {
                const U4 &f = df[t];
                const uint32_t fw = f.w & 0xFFFFu, fpart = f.w >> 16;  // (C(v) block in the band | the tile's part)
                const size_t src = pp.bands[bi].first + fw;
                if (fw >= used.size() || src < sg.b || src >= sg.e || used[fw]++) return E.fail("finalize list names a C(v) block twice or outside its segment");
                if (pp.nparts) {
                    if (fpart >= pp.nparts || src < pp.part_first[fpart] || src >= pp.part_first[fpart + 1])
                        return E.fail("finalize list entry %zu carries part %u, which does not hold tile %zu", t, fpart, src);
                } else if (fpart) {
                    return E.fail("finalize list entry %zu carries a part without parts", t);
                }
                if (f.x != T[src].x || f.y != T[src].y || (f.z & 0xFFu) != T[src].z || ((f.z >> 8) & 0xFFu) != T[src].w)
                    return E.fail("finalize list entry %zu does not match tile %zu", t, src);
                if (dt[src].x != T[src].x || dt[src].y != T[src].y || dt[src].z != (T[src].z | (T[src].w << 8)))
                    return E.fail("tile list entry %zu is wrong", src);
                const int lo = (int)((f.z >> 16) & 0xFFu), hi = (int)(f.z >> 24);
                if (hi - lo + 1 > sg.hist_bins) return E.fail("segment hist_bins %d too small for tile %zu", sg.hist_bins, src);
            }
            {  // tile row by tile row inside the segment, columns ascending (the tile rows come in the wanted order: the
               // extra segments' first)
                std::vector<uint32_t> rows_seen;
                for (size_t t = sg.b + 1; t < sg.e; ++t) {
                    if (df[t - 1].x == df[t].x) {
                        if (df[t - 1].y >= df[t].y) return E.fail("finalize list of a segment is not row-major at %zu", t);
                    } else {
                        rows_seen.push_back(df[t - 1].x);
                        if (std::find(rows_seen.begin(), rows_seen.end(), df[t].x) != rows_seen.end())
                            return E.fail("finalize list of a segment returns to tile row %u at %zu", df[t].x, t);
                    }
                }
            }
        }
    stats[0] = T.size();
    stats[1] = pp.bands.size();
    stats[2] = pp.items.size();
    stats[3] = pp.nparts;
    stats[4] = T.empty() ? 0 : planes_sum * 100 / T.size();
    // rounds of the lockstep tile kernel over all bands (a band of k items takes ceil(k / round_items) rounds)
    stats[7] = 0;
    for (size_t b = 0; b < pp.band_items.size(); ++b) {
        const uint32_t nfr = b < pp.band_frags.size() ? pp.band_frags[b] : 0u;
        stats[7] += (pp.band_items[b].second - pp.band_items[b].first - nfr + tu.round_items - 1) / tu.round_items + (nfr ? 1 : 0);
    }
    return 0;
}

extern "C" {

int dshh_plan_check(uint64_t n, const uint32_t *keys, int mode, int want_sorted, uint64_t rb, uint64_t re, uint64_t cb,
                    uint64_t ce, uint32_t nparts, int want_parts, int p, uint64_t cum_budget, int lockstep, int nsplit,
                    int ls_item_chunks, uint64_t *stats, char *err, size_t cap)
{
    return plan_check(n, keys, mode, want_sorted, rb, re, cb, ce, nparts, want_parts, p, cum_budget, lockstep, nsplit, ls_item_chunks,
                      std::vector<uint64_t>(), stats, err, cap);
}

// the rows of ONE rank of a row-set table (plan.h): main range + extra segments, row-sorted parts (rowsorted != 0: what a
// source rank of the exchange computes) or one part in final order (the destination)
int dshh_plan_check_rowset(uint64_t n, const uint32_t *keys, const uint64_t *tab, uint32_t rank, int rowsorted, uint32_t nparts, int p,
                           uint64_t cum_budget, uint64_t *stats, char *err, size_t cap)
{
    Err E{err, cap};
    RowSets rs;
    if (const char *why = parse_rowsets(tab, n, rs)) return E.fail("%s", why);
    if (rank >= rs.world) return E.fail("rank %u of %u", rank, rs.world);
    uint64_t rb, re;
    std::vector<uint64_t> extra;
    rs.rank_rows(rank, rb, re, extra);
    for (int i = 0; i < 9; ++i) stats[i] = 0;
    if (rb >= re) return 0;
    return plan_check(n, keys, rowsorted ? 3 : 0, 1, rb, re, 0, 0, rowsorted ? nparts : 1, 1, p, cum_budget, 1, 0, 64, extra, stats, err, cap);
}

// first row of the second run of a row-sorted range (plan::rowsorted_split), re when the range stays one run
uint64_t dshh_rowsorted_split(uint64_t n, uint64_t rb, uint64_t re) { return rowsorted_split(n, rb, re); }

}  // extern "C"

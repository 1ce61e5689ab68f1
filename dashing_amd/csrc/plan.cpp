// plan.cpp -- see plan.h.  Pure host code: compiled into libdashing_hip.so (the product) and into
// libdashing_host.so (CPU unit tests of the planner, tests/test_plan.py).
#include "plan.h"

#include <algorithm>
#include <cstring>

namespace dsh {
namespace plan {

uint64_t tri_index(uint64_t n, uint64_t i, uint64_t j) { return i * (2 * n - i - 1) / 2 + j - (i + 1); }

uint64_t tri_span(uint64_t n, uint64_t rb, uint64_t re)
{
    if (re > n) re = n;
    if (rb >= re) return 0;
    // sum_{i=rb}^{re-1} (n-1-i)
    const uint64_t cnt = re - rb;
    return cnt * (n - 1) - (rb + re - 1) * cnt / 2;
}

void partition_rows(uint64_t n, uint32_t nparts, uint32_t align, uint64_t *bounds)
{
    if (align == 0) align = 1;
    const uint64_t total = n ? n * (n - 1) / 2 : 0;
    bounds[0] = 0;
    uint64_t row = 0;
    for (uint32_t r = 1; r < nparts; ++r) {
        const long double target = (long double)total * r / nparts;
        // advance in `align` steps to the boundary whose cumulative pair count is nearest target
        uint64_t best = row;
        long double bestd = -1;
        for (uint64_t cand = row; cand <= n; cand += align) {
            const long double cum = (long double)tri_span(n, 0, cand);
            const long double d = cum > target ? cum - target : target - cum;
            if (bestd < 0 || d < bestd) {
                bestd = d;
                best = cand;
            }
            if (cum > target) break;
        }
        row = std::min<uint64_t>(best, n);
        bounds[r] = row;
    }
    bounds[nparts] = n;
}

void balance_rows(uint64_t n, uint32_t nparts, uint64_t *bounds)
{
    // Contiguous row ranges on 128-row (tile) boundaries that minimise the largest cost of any part.  A part with
    // the tile rows [a,b) of NT computes sum_{t=a}^{b-1} (NT - t) tiles (its triangle + the rectangle to its right)
    // and first prepares its own plane matrix over the columns a*128 .. n (per-sketch pass, key order, transform):
    // kPrepPerTileRow tile-equivalents per 128 columns (0.42 ms per 10 000 columns vs 6.0 us per tile on the C3
    // workload, profiles/r2d) -- the first ranks hold every column, the last only a third, so they get fewer tiles.
    // Unaligned bounds would leave part of a tile row empty on every rank (at n = 10 000 / 8 ranks the first rank has
    // ~5 tile rows: up to 16 % waste).
    constexpr double kPrepPerTileRow = 0.9;
    const uint64_t NT = (n + kTile - 1) / kTile;
    auto cost = [NT](uint64_t t) { return (double)(NT - t); };
    auto fill = [&](double limit, uint64_t *out) -> bool {
        uint64_t t = 0;
        for (uint32_t r = 0; r < nparts; ++r) {
            double acc = kPrepPerTileRow * (double)(NT - t);  // the part's prepare, paid once it holds any row
            if (out) out[r] = std::min<uint64_t>(n, t * kTile);
            while (t < NT && acc + cost(t) <= limit) acc += cost(t++);
        }
        if (out) out[nparts] = n;
        return t == NT;
    };
    double lo = 0, hi = (double)NT * (double)(NT + 1) / 2.0 + kPrepPerTileRow * (double)NT + 1.0;
    if (NT == 0) lo = hi = 0;
    for (int it = 0; it < 100 && hi - lo > 1e-3; ++it) {
        const double mid = 0.5 * (lo + hi);
        if (fill(mid, nullptr)) hi = mid;
        else lo = mid;
    }
    fill(hi, bounds);
}

void range_parts(uint64_t n, uint64_t rb, uint64_t re, uint32_t nparts, std::vector<uint64_t> &out)
{
    out.assign(1, rb);
    if (re > n) re = n;
    if (rb > re) rb = re, out[0] = rb;
    const uint64_t total = tri_span(n, rb, re);
    for (uint32_t q = 1; q < nparts && out.back() < re; ++q) {
        const long double target = (long double)total * q / nparts;
        uint64_t best = out.back() + kTile;
        for (uint64_t cand = out.back() + kTile; cand < re; cand += kTile) {
            best = cand;
            if ((long double)tri_span(n, rb, cand) >= target) break;
        }
        if (best >= re) break;
        out.push_back(best);
    }
    if (out.back() != re) out.push_back(re);
}

// A pair shares cap^2 / 2^p listed positions per side, each one an LDS atomic in k_finalize; every halving of the
// upper tail (a plane saved) costs twice the entries, while the lower tail of the register law falls off
// double-exponentially: listing ~200 registers removes the one or two nearly empty planes at the bottom.  2^p / 32
// entries per side = one shared position per pair and side on average: 32/32 at p = 10, 128/128 at p = 12, the caps
// 255/200 from p = 13.
int auto_list_cap(int p, bool upper)
{
    const uint64_t m = 1ull << p;
    return (int)std::min<uint64_t>(upper ? kMaxListSide : 200, m >> 5);
}

bool rowsorted_rule(uint64_t n, uint64_t rb, uint64_t re, uint32_t nparts)
{
    if (re > n) re = n;
    if (nparts < 2 || rb >= re) return false;
    return re - rb < (uint64_t)1024 * nparts && tri_span(n, rb, re) * sizeof(float) <= ((uint64_t)1 << 30);
}

// tiles of every wanted tile row of a row set, in layout order: the block at layout block index b meets the NTc - b
// blocks from itself to the right (the runs of the layout lie in row order, plan.h)
uint64_t rowsorted_split(uint64_t n, uint64_t rb, uint64_t re)
{
    if (re > n) re = n;
    if (rb >= re) return re;
    const uint64_t TR = (re - rb + kTile - 1) / kTile, NTc = (n - rb + kTile - 1) / kTile;
    if (TR < 16) return re;
    uint64_t total = 0;
    for (uint64_t t = 0; t < TR; ++t) total += NTc - t;
    uint64_t K = 0, acc = 0;
    while (K < TR && acc * 100 < total * 15) acc += NTc - (TR - 1 - K), ++K;
    if (K < 2 || TR - K < 8) return re;
    const uint64_t x = rb + (TR - K) * kTile;
    // mean output length of a row: n - 1 - row
    const long double len_all = (long double)n - 1 - 0.5L * (long double)(rb + re - 1), len_tail = (long double)n - 1 - 0.5L * (long double)(x + re - 1);
    return len_tail <= 0.65L * len_all ? x : re;
}

void wanted_order(uint64_t n, uint64_t rb, uint64_t re, const std::vector<uint64_t> *extra, std::vector<std::pair<uint64_t, uint64_t>> &segs,
                  bool rowsorted)
{
    segs.clear();
    if (re > n) re = n;
    if (rb >= re) return;
    if (extra)
        for (size_t x = 0; x + 1 < extra->size(); x += 2) segs.emplace_back((*extra)[x], std::min<uint64_t>((*extra)[x + 1], n));
    const uint64_t x = rowsorted ? rowsorted_split(n, rb, re) : re;
    if (x > rb && x < re) {
        segs.emplace_back(rb, x);
        segs.emplace_back(x, re);
    } else {
        segs.emplace_back(rb, re);
    }
}

static void wanted_tile_rows(uint64_t n, uint64_t rb, uint64_t re, const std::vector<uint64_t> *extra, std::vector<uint64_t> &cnt)
{
    cnt.clear();
    if (re > n) re = n;
    if (rb >= re) return;
    const uint64_t NTc = (n - rb + kTile - 1) / kTile;
    std::vector<std::pair<uint64_t, uint64_t>> segs;
    wanted_order(n, rb, re, extra, segs);
    for (auto &sg : segs)
        for (uint64_t t = (sg.first - rb) / kTile, t1 = (sg.second - rb + kTile - 1) / kTile; t < t1; ++t) cnt.push_back(NTc - t);
}

uint64_t rowset_rows(uint64_t rb, uint64_t re, const std::vector<uint64_t> &extra)
{
    uint64_t r = re > rb ? re - rb : 0;
    for (size_t x = 0; x + 1 < extra.size(); x += 2) r += extra[x + 1] - extra[x];
    return r;
}

uint64_t rowset_span(uint64_t n, uint64_t rb, uint64_t re, const std::vector<uint64_t> &extra)
{
    uint64_t s = tri_span(n, rb, re);
    for (size_t x = 0; x + 1 < extra.size(); x += 2) s += tri_span(n, extra[x], extra[x + 1]);
    return s;
}

uint64_t rowset_tiles(uint64_t n, uint64_t rb, uint64_t re, const std::vector<uint64_t> &extra)
{
    std::vector<uint64_t> cnt;
    wanted_tile_rows(n, rb, re, &extra, cnt);
    uint64_t t = 0;
    for (uint64_t c : cnt) t += c;
    return t;
}

void rowsorted_part_positions(uint64_t n, uint64_t rb, uint64_t re, uint32_t nparts, std::vector<uint64_t> &pos,
                              const std::vector<uint64_t> *extra)
{
    pos.assign(1, 0);
    if (re > n) re = n;
    if (rb >= re) return;
    // Parts of about equal OUTPUT (bytes), cut between tile rows of the wanted order.  A part is a message of the exchange:
    // the destination receives part q of every source in one round, and a round lasts as long as its largest message --
    // sources whose parts differ in size (or in number) leave links idle.  The rows of a key-ordered run come from anywhere
    // in its row range, so a tile row of the run holds 128 x the run's MEAN row length (n - 1 - row) -- known without the
    // keys, which is what lets every rank compute every rank's cuts.  The extra segments come FIRST in the wanted order:
    // their tile rows are short (the bottom of the triangle), several of them share a part; a rank's last part is then
    // the end of its main range, which is what stays exposed behind its last kernel.
    std::vector<std::pair<uint64_t, uint64_t>> segs;
    wanted_order(n, rb, re, extra, segs, true);
    std::vector<uint64_t> rows_after;   // wanted rows up to and including tile row t of the wanted order
    std::vector<long double> w_after;   // output up to and including it
    uint64_t acc_rows = 0;
    long double acc_w = 0;
    for (auto &sg : segs) {
        const long double mean_len = (long double)n - 1 - 0.5L * (long double)(sg.first + sg.second - 1);
        for (uint64_t b = sg.first; b < sg.second; b += kTile) {
            const uint64_t rows = std::min<uint64_t>(kTile, sg.second - b);
            const long double w = (long double)rows * std::max<long double>(mean_len, 1.0L);
            acc_rows += rows;
            acc_w += w;
            rows_after.push_back(acc_rows);
            w_after.push_back(acc_w);
        }
    }
    const size_t TR = rows_after.size();
    const size_t K = (size_t)std::max<uint64_t>(1, std::min<uint64_t>(nparts, TR));
    size_t t = 0;  // tile rows in front of the next part
    for (size_t q = 1; q < K; ++q) {
        // the boundary nearest to q / K of the output, behind at least one more tile row and with one left for every later part
        const long double target = acc_w * (long double)q / (long double)K;
        size_t best = t + 1;
        for (size_t c = t + 1; c + (K - q) <= TR; ++c) {
            const long double dc = w_after[c - 1] > target ? w_after[c - 1] - target : target - w_after[c - 1];
            const long double db = w_after[best - 1] > target ? w_after[best - 1] - target : target - w_after[best - 1];
            if (dc < db) best = c;
            if (w_after[c - 1] >= target) break;
        }
        pos.push_back(rows_after[best - 1]);
        t = best;
    }
    pos.push_back(acc_rows);
}

// ---- row sets --------------------------------------------------------------------------------------------------------
void RowSets::write(uint64_t *tab) const
{
    const size_t ns = owner.size();
    tab[0] = world;
    tab[1] = ns;
    for (size_t s = 0; s <= ns; ++s) tab[2 + s] = seg[s];
    for (size_t s = 0; s < ns; ++s) tab[3 + ns + s] = owner[s];
}

void RowSets::rank_rows(uint32_t r, uint64_t &rb, uint64_t &re, std::vector<uint64_t> &extra) const
{
    rb = re = 0;
    extra.clear();
    bool have = false;
    for (size_t s = 0; s < owner.size(); ++s) {
        if (owner[s] != r || seg[s] >= seg[s + 1]) continue;
        if (!have) {
            rb = seg[s], re = seg[s + 1], have = true;
        } else if (extra.empty() ? seg[s] == re : seg[s] == extra.back()) {  // adjacent: one segment
            if (extra.empty()) re = seg[s + 1];
            else extra.back() = seg[s + 1];
        } else {
            extra.push_back(seg[s]);
            extra.push_back(seg[s + 1]);
        }
    }
}

const char *parse_rowsets(const uint64_t *tab, uint64_t n, RowSets &rs)
{
    if (!tab) return "the row-set table is NULL";
    if (tab[0] == 0 || tab[0] > 65536) return "row-set table: bad number of ranks";
    if (tab[1] > ((uint64_t)1 << 24)) return "row-set table: bad number of segments";
    rs.world = (uint32_t)tab[0];
    const size_t ns = (size_t)tab[1];
    rs.seg.assign(tab + 2, tab + 3 + ns);
    rs.owner.resize(ns);
    if (rs.seg.front() != 0 || rs.seg.back() != n) return "row-set table: the segments must run from 0 to n";
    for (size_t s = 0; s < ns; ++s) {
        if (rs.seg[s] > rs.seg[s + 1]) return "row-set table: segment boundaries not monotone";
        if (tab[3 + ns + s] >= rs.world) return "row-set table: a segment's owner is not a rank";
        rs.owner[s] = (uint32_t)tab[3 + ns + s];
    }
    // a rank with extra segments: every boundary of its segments on a multiple of 128 (or at n)
    std::vector<uint64_t> extra;
    for (uint32_t r = 0; r < rs.world; ++r) {
        uint64_t rb, re;
        rs.rank_rows(r, rb, re, extra);
        if (extra.empty()) continue;
        auto ok = [n](uint64_t x) { return x % kTile == 0 || x == n; };
        if (!ok(rb) || !ok(re)) return "row-set table: a rank with extra segments needs all its boundaries on multiples of 128";
        for (uint64_t x : extra)
            if (!ok(x)) return "row-set table: a rank with extra segments needs all its boundaries on multiples of 128";
    }
    return nullptr;
}

void rowsets_from_bounds(const uint64_t *bounds, uint32_t world, RowSets &rs)
{
    rs.world = world;
    rs.seg.assign(bounds, bounds + world + 1);
    rs.owner.resize(world);
    for (uint32_t r = 0; r < world; ++r) rs.owner[r] = r;
}

void balance_rowsets(uint64_t n, uint32_t world, RowSets &rs, uint32_t prep_permille, int dst, uint32_t dst_bonus_permille)
{
    const uint64_t NT = (n + kTile - 1) / kTile;
    std::vector<uint64_t> cb(world + 1);
    balance_rows(n, world, cb.data());
    rowsets_from_bounds(cb.data(), world, rs);
    if (world < 2 || n > kTopupMaxRows || NT < 2 * (uint64_t)world) return;
    // The cost of a rank = its tiles + its own prepare (balance_rows: kPrepPerTileRow per 128 columns of its plane matrix,
    // which starts at its first row).  For a limit: main ranges filled from the top, none above the limit; the tile rows
    // left over at the bottom (NT - t tiles each) are dealt in RUNS of consecutive tile rows -- the rank furthest below
    // the limit takes rows from the top of what is left while it stays under the limit, then the next rank (a run of
    // consecutive rows is ONE extra segment, key-ordered as one run of the rank's layout; single rows dealt round-robin
    // would each be a 128-row block in no key order); whatever is left then goes row by row to the cheapest rank.  Two
    // sweeps over the limits: the smallest maximum any limit reaches, then, among the limits within half a percent of
    // it, the one with the fewest segments.
    // the weight of a rank's own prepare.  Measured on BASELINE configs[2] over 8 ranks (profiles/rd5b): prepare costs 2.9 us
    // per 128 columns against 4.8 us per tile (tile kernel + k_finalize), i.e. 0.6 tiles; the default leans towards equal
    // TILE counts because the tile kernel's time moves in whole rounds of 512 work items (7 or 8 at that size) and the
    // ranks with the cheap prepare are the ones that would spill into another round
    const double kPrepPerTileRow = prep_permille == ~0u ? 0.4 : (double)prep_permille / 1000.0;
    // The DESTINATION of an exchange sends nothing: every other rank's step ends when its last part has ARRIVED, i.e. its
    // compute + what stays exposed of its transfer (the last part's 4.6 MB + the rounds' latency: ~0.2 of 2.2 ms at
    // BASELINE configs[2] over 8 ranks, profiles/rd5c), the destination's with its last kernel.  So the destination takes
    // a bonus of work, as a share of a rank's mean tile count: every rank's cost below is its tiles + prepare, minus the
    // bonus for the destination.  Default 12 % where a rank holds at least 16 tile rows (profiles/rd5f, rd5h: 2 ranks of
    // BASELINE configs[2] 1.61 -> 1.72x, 4 ranks 3.09 -> 3.23x modelled at 45 GB/s per link), none below that: at 8 ranks
    // (10 tile rows each) the tile kernel takes 7 rounds for 385 tiles and for 400, so taking tiles off the sources buys
    // nothing while the destination spills into an 8th round and becomes the last to finish (5.75x -> 5.66-5.81x).
    const double total = (double)NT * (double)(NT + 1) / 2.0;
    const double share = dst_bonus_permille != ~0u ? (double)dst_bonus_permille / 1000.0 : (NT >= 16 * (uint64_t)world ? 0.12 : 0.0);
    const double bonus = dst >= 0 && (uint32_t)dst < world ? total / world * share : 0.0;
    auto handicap = [&](uint32_t r) { return (int)r == dst ? -bonus : 0.0; };
    std::vector<uint64_t> start(world), stop(world);
    std::vector<double> cost(world);
    std::vector<uint8_t> has(world);
    std::vector<uint32_t> deal, best_deal, byneed(world);
    std::vector<uint64_t> best_start, best_stop;
    double best_max = -1;
    uint64_t best_t = NT + 1;
    size_t best_segs = 0;
    const double lo = total / world - bonus, hi = total / world + kPrepPerTileRow * (double)NT + (double)NT;
    const int steps = 800;
    for (int sweep = 0; sweep < 2; ++sweep) {
        const double accept = best_max * 1.005;
        for (int it = 0; it <= steps; ++it) {
            const double limit = lo + (hi - lo) * it / steps;
            uint64_t t = 0;
            for (uint32_t r = 0; r < world; ++r) {
                start[r] = t;
                double acc = kPrepPerTileRow * (double)(NT - t) + handicap(r);
                while (t < NT && acc + (double)(NT - t) <= limit) acc += (double)(NT - t++);
                stop[r] = t;
                has[r] = stop[r] > start[r];
                cost[r] = has[r] ? acc : handicap(r);
            }
            const uint64_t t_pool = t;
            deal.assign(NT - t_pool, ~0u);
            for (uint32_t r = 0; r < world; ++r) byneed[r] = r;
            std::stable_sort(byneed.begin(), byneed.end(), [&](uint32_t a, uint32_t b) { return cost[a] < cost[b]; });
            uint64_t u = t_pool;
            for (uint32_t k = 0; k < world && u < NT; ++k) {
                const uint32_t r = byneed[k];
                // (a rank without a main range would start its plane matrix at the dealt row: it pays that prepare)
                if (!has[r]) cost[r] = kPrepPerTileRow * (double)(NT - u) + handicap(r), has[r] = 1;
                while (u < NT && cost[r] + (double)(NT - u) <= limit) cost[r] += (double)(NT - u), deal[u++ - t_pool] = r;
            }
            for (; u < NT; ++u) {
                uint32_t to = 0;
                for (uint32_t r = 1; r < world; ++r)
                    if (cost[r] < cost[to]) to = r;
                cost[to] += (double)(NT - u);
                deal[u - t_pool] = to;
            }
            double mx = 0;
            for (uint32_t r = 0; r < world; ++r) mx = std::max(mx, cost[r]);
            size_t segs = 0;
            for (size_t d = 0; d < deal.size(); ++d) segs += d == 0 || deal[d] != deal[d - 1];
            if (sweep == 0) {
                if (best_max < 0 || mx < best_max) best_max = mx;
            } else if (mx <= accept && (best_t == NT + 1 || segs < best_segs)) {
                best_t = t_pool;
                best_segs = segs;
                best_deal = deal;
                best_start = start;
                best_stop = stop;
            }
        }
    }
    if (best_t >= NT && bonus == 0.0) return;  // nothing dealt: the contiguous ranges of balance_rows
    if (best_t > NT) return;                   // (no limit was accepted)
    // a rank that holds dealt rows needs aligned boundaries: all main boundaries are multiples of 128 by construction
    rs.seg.clear();
    rs.owner.clear();
    rs.seg.push_back(0);
    auto push = [&](uint64_t end_row, uint32_t who) {
        if (end_row <= rs.seg.back()) return;
        if (!rs.owner.empty() && rs.owner.back() == who) rs.seg.back() = end_row;
        else rs.seg.push_back(end_row), rs.owner.push_back(who);
    };
    for (uint32_t r = 0; r < world; ++r) push(std::min<uint64_t>(n, best_stop[r] * kTile), r);
    for (uint64_t u = best_t; u < NT; ++u) push(std::min<uint64_t>(n, (u + 1) * kTile), best_deal[u - best_t]);
    if (rs.owner.empty()) rs.owner.push_back(0), rs.seg.push_back(n);
    rs.seg.back() = n;
}

void sort_rows_by_key(const uint32_t *k32, uint64_t lo, uint64_t hi, uint32_t *dst, std::vector<uint32_t> &a)
{
    // stable LSD radix sort, two 9-bit digits of the 18-bit key (T, L, max value) (a single 2^18-bucket counting sort
    // spends ~0.1 ms clearing and scanning its counters); the host sits between the per-sketch pass and the transform,
    // so this is on the critical path of every layout
    auto skey = [&](uint64_t i) -> uint32_t {  // 6 bits each: high threshold, low threshold, max value
        const uint32_t key = k32[i];
        return ((uint32_t)key_T(key) << 12) | ((uint32_t)key_L(key) << 6) | (uint32_t)key_hi(key);
    };
    const uint64_t cnt_ = hi - lo;
    if (a.size() < 2 * cnt_) a.resize(2 * cnt_);  // [indices after the first digit | sort keys]
    uint32_t *const ks = a.data() + cnt_;
    uint32_t c0[513], c1[513];  // both digits counted in the pass that makes the sort keys
    std::memset(c0, 0, sizeof c0);
    std::memset(c1, 0, sizeof c1);
    for (uint64_t i = 0; i < cnt_; ++i) {
        const uint32_t k = skey(lo + i);
        ks[i] = k;
        c0[(k & 511u) + 1u]++;
        c1[(k >> 9) + 1u]++;
    }
    for (int k = 1; k < 513; ++k) c0[k] += c0[k - 1], c1[k] += c1[k - 1];
    for (uint64_t i = 0; i < cnt_; ++i) a[c0[ks[i] & 511u]++] = (uint32_t)i;
    for (uint64_t i = 0; i < cnt_; ++i) dst[c1[ks[a[i]] >> 9]++] = (uint32_t)(lo + a[i]);
}

void rowsorted_offsets(uint64_t n, const uint32_t *order, uint64_t cnt, std::vector<uint64_t> &rowoff)
{
    rowoff.resize(cnt + 1);
    uint64_t acc = 0;
    for (uint64_t s = 0; s < cnt; ++s) {
        rowoff[s] = acc;
        acc += n - 1 - order[s];
    }
    rowoff[cnt] = acc;
}

void tile_planes(const Layout &L, uint32_t ti, uint32_t tj, int &pb, int &pe)
{
    const int lo_t = std::max<int>(std::max<int>(L.blk_lo[ti], L.blk_lo[tj]), std::min<int>(L.blk_L[ti], L.blk_L[tj]));
    const int T_t = std::max<int>(L.blk_T[ti], L.blk_T[tj]);
    pb = std::max(0, lo_t - L.pbase);
    pe = std::max(pb, T_t - L.pbase);
}

void build_layout(const uint32_t *k32, uint64_t n, int want_sorted, uint64_t rb, uint64_t re,
                  const std::vector<uint64_t> &parts, Layout &L, uint32_t rowsorted_nparts, const std::vector<uint64_t> *extra_in)
{
    if (re > n) re = n;
    if (!want_sorted) rb = 0, re = n;
    if (rb > re) rb = re;
    static const std::vector<uint64_t> kNoExtra;
    const std::vector<uint64_t> &extra = (want_sorted && extra_in && rb < re) ? *extra_in : kNoExtra;
    // the plane matrix holds the sketches col0 .. n-1 (a row range [rb,re) of the triangle never looks at
    // sketches before rb); value range and thresholds are taken over those only
    const uint64_t col0 = want_sorted ? rb : 0;
    const uint64_t ncols = n - col0;
    int vr[3] = {63, 0, 0};  // min register value anywhere, max value, max threshold
    for (uint64_t i = col0; i < n; ++i) {
        const uint32_t key = k32[i];
        vr[0] = std::min<int>(vr[0], key_lo(key));
        vr[1] = std::max<int>(vr[1], key_hi(key));
        vr[2] = std::max<int>(vr[2], key_T(key));
    }
    if (ncols == 0) vr[0] = vr[1] = vr[2] = 0;
    L.sorted = want_sorted;
    L.rb = rb;
    L.re = re;
    L.rowsorted = want_sorted && rowsorted_nparts > 0;
    L.extra = extra;
    if (L.rowsorted) {  // ONE key-ordered run, or two (rowsorted_split)
        const uint64_t x = rowsorted_split(n, rb, re);
        if (x > rb && x < re) L.parts = {rb, x, re};
        else L.parts = {rb, re};
    } else if (!extra.empty()) {
        L.parts = {rb, re};
    } else {
        L.parts = want_sorted ? parts : std::vector<uint64_t>();
    }
    L.n = n;
    L.vlo = vr[0];
    L.vhi = vr[1];
    L.ncols = ncols;
    L.Npad = (uint32_t)((ncols + kTile - 1) / kTile * kTile);
    L.whole = want_sorted && rb == 0 && re == n;
    // column order: identity, or a counting sort by (threshold, min value, max value): the first
    // two make the 128-column blocks need few planes, the third keeps the 64 pairs of a finalize
    // wave alike in their largest register, i.e. in the trip count of the estimator's loops.
    // With a row range the wanted rows [rb,re) come first (their tile rows are the only ones computed),
    // then the later rows; each part is key-ordered on its own.
    L.perm.resize(ncols);
    L.part_pos.clear();
    L.part_w.clear();
    L.rowoff.clear();
    L.rowoff_w.clear();
    L.wtr.clear();
    L.wtr_w.clear();
    L.nwanted = want_sorted ? rowset_rows(rb, re, extra) : n;
    if (want_sorted) {
        // the runs, in row order (a run of rows [lo, hi) lies at the positions [lo - rb, hi - rb)): the parts of the
        // wanted range, then the later rows -- cut into [gap | extra segment | gap ...] where extra segments lie
        for (size_t q = 0; q + 1 < L.parts.size(); ++q)
            sort_rows_by_key(k32, L.parts[q], L.parts[q + 1], L.perm.data() + (L.parts[q] - rb), L.sort_a);
        uint64_t at = re;
        for (size_t x = 0; x + 1 < extra.size(); x += 2) {
            if (extra[x] > at) sort_rows_by_key(k32, at, extra[x], L.perm.data() + (at - rb), L.sort_a);
            sort_rows_by_key(k32, extra[x], extra[x + 1], L.perm.data() + (extra[x] - rb), L.sort_a);
            at = extra[x + 1];
        }
        if (at < n) sort_rows_by_key(k32, at, n, L.perm.data() + (at - rb), L.sort_a);
        // tile rows that hold wanted rows, in the WANTED ORDER (extra segments first, then the main range: the order of the
        // tile list, of the parts and of a row-sorted buffer), and the wanted rows in front of each of these ranges
        std::vector<std::pair<uint64_t, uint64_t>> wsegs;
        wanted_order(n, rb, re, &extra, wsegs, L.rowsorted);
        uint64_t wrows = 0;
        for (auto &sg : wsegs) {
            L.wtr.emplace_back((uint32_t)((sg.first - rb) / kTile), (uint32_t)((sg.second - rb + kTile - 1) / kTile));
            L.wtr_w.push_back(wrows);
            wrows += sg.second - sg.first;
        }
        const uint64_t wend = extra.empty() ? re - rb : extra.back() - rb;  // position behind the last wanted row
        if (L.rowsorted) {
            rowsorted_part_positions(n, rb, re, rowsorted_nparts, L.part_w, &extra);
            // wanted order -> the rows' offsets in the rank's buffer, and the same by layout position
            L.rowoff_w.resize(L.nwanted + 1);
            L.rowoff.assign(wend + 1, 0);
            uint64_t acc = 0, w = 0;
            for (auto &sg : wsegs)
                for (uint64_t s = sg.first - rb; s < sg.second - rb; ++s, ++w) {
                    L.rowoff_w[w] = L.rowoff[s] = acc;
                    acc += n - 1 - L.perm[s];
                }
            L.rowoff_w[L.nwanted] = acc;
        } else if (!extra.empty()) {
            L.part_w = {0, L.nwanted};
        } else {
            for (uint64_t r : L.parts) L.part_w.push_back(r - rb);
        }
        // (layout positions of the cuts: only meaningful where the wanted order is the layout order -- no extra segments)
        if (extra.empty()) L.part_pos = L.part_w;
        else L.part_pos = {0, wend};
        // (whole collection only) the inverse for the un-permute of the shard path
        if (L.whole) {
            L.perm.resize(2 * n);
            for (uint64_t s = 0; s < n; ++s) L.perm[n + L.perm[s]] = (uint32_t)s;
        }
    } else {
        for (uint64_t i = 0; i < n; ++i) L.perm[i] = (uint32_t)i;
    }
    const uint32_t NT = L.Npad / kTile;
    L.blk_T.assign(NT, 0);
    L.blk_lo.assign(NT, 255);
    L.blk_L.assign(NT, 255);
    L.blk_hi.assign(NT, 0);
    int pbase = vr[2];
    const uint32_t *perm = L.perm.data();
    for (uint32_t b = 0; b < NT; ++b) {  // (a block at a time, its four statistics in registers)
        const uint64_t s0 = (uint64_t)b * kTile, s1 = std::min<uint64_t>(ncols, s0 + kTile);
        int bt = 0, blo = 255, bl = 255, bhi = 0;
        for (uint64_t s = s0; s < s1; ++s) {
            const uint32_t key = k32[perm[s]];
            bt = std::max(bt, key_T(key));
            blo = std::min(blo, key_lo(key));
            bl = std::min(bl, key_L(key));
            bhi = std::max(bhi, key_hi(key));
        }
        L.blk_T[b] = (uint8_t)bt;
        L.blk_lo[b] = (uint8_t)blo;
        L.blk_L[b] = (uint8_t)bl;
        L.blk_hi[b] = (uint8_t)bhi;
        pbase = std::min(pbase, bl);
    }
    // dense planes cover v in (pbase, Tmax]: below the smallest low threshold every C(v) comes from the list join
    L.pbase = pbase;
    L.P = (uint32_t)(vr[2] - pbase);
}

// Order the tiles of a segment so that workgroups that run on the same XCD (block b -> XCD b % 8,
// observed dispatch behaviour; speed only, never correctness) walk one tile row together and
// share its A panel in that XCD's L2.
static void xcd_order(std::vector<U4> &t, std::vector<uint32_t> &rank, size_t b, size_t e, size_t group)
{
    const size_t cnt = e - b;
    if (cnt < 16) return;
    std::vector<U4> tmp(cnt);
    std::vector<uint32_t> rtmp(cnt);
    const size_t nx = 8, per = (cnt + nx - 1) / nx;
    // `group` consecutive positions of the launch order share a workgroup (2 with the lockstep kernel); workgroup w
    // runs on XCD w % 8; give XCD x the contiguous range [x*per, (x+1)*per) of the row-major list
    size_t q = 0;
    for (size_t r = 0; r < per; r += group)
        for (size_t x = 0; x < nx; ++x)
            for (size_t u = 0; u < group && r + u < per; ++u) {
                const size_t src = x * per + r + u;
                if (src < cnt) {
                    rtmp[q] = rank[b + src];
                    tmp[q++] = t[b + src];
                }
            }
    std::copy(tmp.begin(), tmp.begin() + q, t.begin() + b);
    std::copy(rtmp.begin(), rtmp.begin() + q, rank.begin() + b);
}

static void tile_vrange(const Layout &L, const U4 &t, int &lo, int &hi)
{
    lo = std::min<int>(L.blk_lo[t.x], L.blk_lo[t.y]);
    hi = std::max<int>(L.blk_hi[t.x], L.blk_hi[t.y]);
    if (hi < lo) hi = lo;  // (blocks of padding only)
}

bool build_tiles(const Layout &L, const PairQuery &job, const Tuning &tu, PairPlan &pp)
{
    std::vector<U4> &T = pp.T;
    T.clear();
    pp.bands.clear();
    pp.band_items.clear();
    pp.segs.clear();
    pp.items.clear();
    pp.band_frags.clear();
    pp.nparts = 0;
    pp.max_band = 0;
    const uint32_t NT = L.Npad / kTile;
    // tile list: {row block, col block, plane begin, plane end}; a tile only needs the planes
    // v in (max(min lo of its two blocks), max threshold of its two blocks]
    // (a row of tiles at a time, the row block's statistics in registers: at 100 000 sketches this loop writes 306 000
    // tiles and the host's planning is not hidden behind anything the GPU does -- profiles/r4y)
    const uint8_t *bLo = L.blk_lo.data(), *bL = L.blk_L.data(), *bT = L.blk_T.data();
    const int pbase = L.pbase;
    auto tile_row = [&](uint32_t ti, uint32_t c0, uint32_t c1) {
        const size_t at = T.size();
        T.resize(at + (c1 - c0));
        U4 *out = T.data() + at;
        const int lo_i = bLo[ti], L_i = bL[ti], T_i = bT[ti];
        for (uint32_t tj = c0; tj < c1; ++tj) {  // tile_planes(), inlined
            const int lo_t = std::max(std::max<int>(lo_i, bLo[tj]), std::min<int>(L_i, bL[tj]));
            const int pb = std::max(0, lo_t - pbase), pe = std::max(pb, std::max<int>(T_i, bT[tj]) - pbase);
            *out++ = U4{ti, tj, (uint32_t)pb, (uint32_t)pe};
        }
    };
    std::vector<size_t> trow_first;   // (wanted tile rows of a sorted layout) first tile of the tile row ...
    std::vector<uint64_t> trow_w;     // ... and the wanted rows in front of it (wanted order)
    if (job.rect) {
        if (job.row_begin >= job.row_end || job.col_begin >= job.col_end) return false;
        const uint32_t r0 = (uint32_t)(job.row_begin / kTile), r1 = (uint32_t)((job.row_end + kTile - 1) / kTile);
        const uint32_t c0 = (uint32_t)(job.col_begin / kTile), c1 = (uint32_t)((job.col_end + kTile - 1) / kTile);
        T.reserve((size_t)(r1 - r0) * (c1 - c0));
        for (uint32_t ti = r0; ti < r1; ++ti) tile_row(ti, c0, c1);
    } else {
        if (job.row_begin >= job.row_end) return false;
        if (!L.sorted || job.sorted_rows) {  // rows index plane columns: only their tile rows
            const uint32_t r0 = (uint32_t)(job.row_begin / kTile);
            const uint32_t r1 = std::min<uint32_t>(NT, (uint32_t)((job.row_end + kTile - 1) / kTile));
            if (r1 > r0) T.reserve((size_t)(r1 - r0) * (NT - r0) - (size_t)(r1 - r0) * (r1 - r0 - 1) / 2);
            for (uint32_t ti = r0; ti < r1; ++ti) tile_row(ti, ti, NT);
        } else {  // the tile rows of the wanted runs, in the wanted order (extra segments' blocks first, then the main range)
            size_t cnt = 0;
            for (const auto &w : L.wtr)
                for (uint32_t ti = w.first; ti < std::min(w.second, NT); ++ti) cnt += NT - ti;
            T.reserve(cnt);
            for (size_t k = 0; k < L.wtr.size(); ++k)
                for (uint32_t ti = L.wtr[k].first; ti < std::min(L.wtr[k].second, NT); ++ti) {
                    trow_first.push_back(T.size());
                    trow_w.push_back(L.wtr_w[k] + (uint64_t)(ti - L.wtr[k].first) * kTile);
                    tile_row(ti, ti, NT);
                }
        }
    }
    if (T.empty()) return false;
    // bands bounded by the cum scratch budget
    pp.per_tile_bytes = (uint64_t)kTile * kTile * tu.cum_bytes * std::max<uint32_t>(L.P, 1);
    // (at most 2^16 tiles per band: k_finalize addresses a pair slot of the band with 32 bits)
    const uint64_t max_tiles = std::min<uint64_t>(65536, std::max<uint64_t>(1, tu.cum_budget / pp.per_tile_bytes));
    // Parts (dsh_dist_rows_parts_device_async): k_finalize runs once per SEGMENT -- the tiles of one part inside one
    // band -- and an event marks the end of a part's last segment: the part's span of the matrix is final and can travel
    // while the other parts are computed.  The tile kernel runs once per band; a band is also cut at a part boundary
    // when the part is large (>= kPartBandTiles tiles: a cut costs 0.2-0.3 ms -- two tails and a pipeline bubble,
    // profiles/r3g -- nothing against the ~1 ms per 1 000 tiles the part takes, and the first part can leave after 1/nparts of
    // the compute instead of after the whole tile kernel); small parts (C3 / 8 ranks: ~50-400 tiles) only cut k_finalize.
    // A layout with ONE part (a short range) still gets its event: the exchange of every rank looks the same.
    const size_t kPartBandTiles = tu.part_band_tiles;
    const bool parts_on = job.want_parts && !job.rect && !job.sorted_rows && L.sorted && L.part_w.size() >= 2;
    // first tile of every part: T holds the wanted tile rows in the wanted order and a part is a run of whole tile rows of
    // that order, so the part of a tile is monotone
    std::vector<size_t> pstart{0};
    if (parts_on) {
        size_t q = 0;
        for (size_t r = 0; r < trow_first.size(); ++r)
            while (q + 2 < L.part_w.size() && trow_w[r] >= L.part_w[q + 1]) {
                ++q;
                pstart.push_back(trow_first[r]);
            }
        pp.nparts = (uint32_t)pstart.size();
    }
    pstart.push_back(T.size());
    pp.part_first = pstart;
    pp.part_tiles.clear();
    if (parts_on)
        for (size_t q = 0; q + 1 < pstart.size(); ++q) pp.part_tiles.push_back((uint32_t)(pstart[q + 1] - pstart[q]));
    auto part_of_tile = [&](size_t t) -> size_t {
        return (size_t)(std::upper_bound(pstart.begin(), pstart.end() - 1, t) - pstart.begin()) - 1;
    };
    // tail bands (plan.h, Tuning): cuts of the tile list at whole rounds of the tile kernel
    std::vector<size_t> cuts;
    // (a call with ONE part has nothing to send early: the destination of an exchange computes its rows that way)
    if (parts_on && pp.nparts >= 2 && tu.tail_bands > 0 && tu.lockstep && tu.W >= (uint32_t)tu.kc && tu.nsplit == 0 && tu.round_items > 0) {
        const uint64_t RI = tu.round_items;
        // one item per plane of a tile while a BAND holds at most 16 rounds of them (build_band_items: piece = one plane);
        // a longer launch takes pieces of 2-4 planes, i.e. longer rounds.  Jobs of up to 64 rounds are therefore cut into
        // bands of at most 16: the head as well as the tails (2 ranks of BASELINE configs[2]: 27 rounds = 15 + 9 + 3
        // instead of ONE launch of 15 two-plane rounds whose first part is final after 6 ms).
        uint64_t items = 0;
        for (const U4 &t : T) items += t.w - t.z;
        const uint64_t R = (items + RI - 1) / RI;
        constexpr uint64_t kBandRounds = 16;
        for (uint32_t ntails = tu.tail_bands; ntails >= 1 && cuts.empty() && items <= 4 * kBandRounds * RI && R >= 3; --ntails) {
            std::vector<uint64_t> tails;  // rounds of the tail bands, last band first
            uint64_t left = R;
            for (uint32_t b = 0; b < ntails; ++b) {
                const uint64_t pm = b == 0 ? tu.tail_permille : tu.tail_permille2;
                const uint64_t r = std::max<uint64_t>(1, (left * pm + 500) / 1000);
                // (a tail in front of the last one only pays where the head keeps enough rounds for its parts' transfer to
                // hide behind it: 14 rounds -> 8 + 5 + 1, but 7 rounds -> 6 + 1, profiles/rd5d)
                if (left < r + 2 || (b > 0 && left < r + tu.tail_head_min_rounds)) break;
                tails.push_back(r);
                left -= r;
            }
            // the head takes `left` rounds (in bands of at most 16), then the tails in reverse; a band ends at the last
            // tile that still fits
            std::vector<uint64_t> quota;
            {
                const uint64_t nb = (left + kBandRounds - 1) / kBandRounds;
                for (uint64_t b = 0; b < nb; ++b) quota.push_back(left / nb + (b < left % nb ? 1 : 0));
            }
            for (size_t b = tails.size(); b-- > 0;) {
                const uint64_t nb = (tails[b] + kBandRounds - 1) / kBandRounds;
                for (uint64_t x = 0; x < nb; ++x) quota.push_back(tails[b] / nb + (x < tails[b] % nb ? 1 : 0));
            }
            // (the quota is the band's own: what a band leaves of its last round must not slide into the next band and
            // push that one over a round boundary)
            uint64_t acc = 0, rounds = 0, band_items = 0;
            size_t t = 0;
            for (size_t b = 0; b + 1 < quota.size(); ++b) {
                while (t < T.size() && band_items + (T[t].w - T[t].z) <= quota[b] * RI) {
                    acc += T[t].w - T[t].z;
                    band_items += T[t].w - T[t].z;
                    ++t;
                }
                if (t == 0 || t >= T.size() || (!cuts.empty() && cuts.back() == t)) continue;
                cuts.push_back(t);
                rounds += (band_items + RI - 1) / RI;
                band_items = 0;
            }
            rounds += (items - acc + band_items + RI - 1) / RI;
            // a cut that costs a round is worse than none -- except for a long job, where one round in twenty buys parts that
            // leave milliseconds earlier and one-plane rounds instead of two-plane ones
            if (rounds != R && !(R > kBandRounds && rounds == R + 1)) cuts.clear();
        }
    }
    size_t next_cut = 0;
    for (size_t b = 0; b < T.size();) {
        size_t e = std::min<size_t>(T.size(), b + max_tiles);
        if (parts_on) {  // a large part also ends the band
            const size_t q = part_of_tile(b);
            if (pstart[q + 1] - pstart[q] >= kPartBandTiles) e = std::min(e, pstart[q + 1]);
        }
        while (next_cut < cuts.size() && cuts[next_cut] <= b) ++next_cut;
        if (next_cut < cuts.size()) e = std::min(e, cuts[next_cut]);
        pp.bands.emplace_back(b, e);
        b = e;
    }
    pp.segs.resize(pp.bands.size());
    for (size_t bi = 0; bi < pp.bands.size(); ++bi) {
        size_t b = pp.bands[bi].first;
        while (b < pp.bands[bi].second) {
            const size_t q = part_of_tile(b);  // part of a tile = part of its tile row (parts are runs of whole tile rows)
            const size_t e = std::min(pp.bands[bi].second, pstart[q + 1]);
            const bool last_of_part = e == pstart[q + 1];
            Seg sg{b, e, parts_on && last_of_part ? (int)q : -1, 1};
            for (size_t t = b; t < e; ++t) {
                int lo, hi;
                tile_vrange(L, T[t], lo, hi);
                sg.hist_bins = std::max(sg.hist_bins, hi - lo + 1);
            }
            pp.segs[bi].push_back(sg);
            b = e;
        }
    }
    // rank[t] = position of tile t in the row-major order of its segment (the order k_finalize walks)
    std::vector<uint32_t> &rank = pp.rank;
    rank.resize(T.size());
    for (auto &sv : pp.segs)
        for (auto &sg : sv)
            for (size_t t = sg.b; t < sg.e; ++t) rank[t] = (uint32_t)(t - sg.b);
    for (auto &sv : pp.segs)
        for (auto &sg : sv) xcd_order(T, rank, sg.b, sg.e, tu.lockstep ? 2 : 1);
    // chunk range of every tile (the work items of a band are cut from these: build_band_items)
    const uint32_t KC = (uint32_t)tu.kc;
    const uint32_t cpp = tu.W >= KC ? tu.W / KC : 1;  // chunks per plane when a plane spans chunks
    std::vector<U2> &CR = pp.chunks;
    CR.resize(T.size());
    if (tu.W >= KC) {  // whole chunks per plane (both powers of two)
        for (size_t t = 0; t < T.size(); ++t) CR[t] = U2{T[t].z * cpp, T[t].w * cpp};
    } else {
        for (size_t t = 0; t < T.size(); ++t)
            CR[t] = U2{(uint32_t)(((uint64_t)T[t].z * tu.W) / KC), (uint32_t)(((uint64_t)T[t].w * tu.W + KC - 1) / KC)};
    }
    for (auto &bd : pp.bands) pp.max_band = std::max(pp.max_band, bd.second - bd.first);
    return true;
}

// overflow fragments (plan.h, Tuning): of `ni` one-plane items, how many are cut (the last `over`) and into how many
// fragments each (`f`); over = 0: none
static void overflow_fragments(const Tuning &tu, uint32_t cpp, uint64_t piece, uint64_t ni, uint64_t &over, uint32_t &f)
{
    over = 0;
    f = 1;
    const uint64_t RI = tu.round_items;
    if (!tu.lockstep || tu.nsplit != 0 || RI == 0 || tu.overflow_frag_max_permille == 0 || piece != cpp || cpp < 2 || ni == 0) return;
    // (a band of less than a round -- the tail band of an exchange call can hold a few dozen items -- is all overflow)
    const uint64_t o = ni % RI;
    if (o == 0 || o * 1000 > RI * tu.overflow_frag_max_permille) return;
    uint32_t ff = 1;
    while (ff * 2 <= cpp && o * (ff * 2) <= RI) ff *= 2;
    if (ff < 2) return;
    over = o;
    f = ff;
}

// work items of band bi, appended to pp.items (bands in order): {tile index in band, chunk begin, chunk end, fragment?}
void build_band_items(const Tuning &tu, PairPlan &pp, size_t bi)
{
    const auto &bd = pp.bands[bi];
    const std::vector<U2> &CR = pp.chunks;
    std::vector<U4> &I = pp.items;
    const uint32_t KC = (uint32_t)tu.kc;
    const uint32_t cpp = tu.W >= KC ? tu.W / KC : 1;
    const size_t nt = bd.second - bd.first;
    uint64_t tot = 0;
    uint32_t maxlen = 0;
    for (size_t t = bd.first; t < bd.second; ++t) {
        const uint32_t len = CR[t].y - CR[t].x;
        tot += len;
        maxlen = std::max(maxlen, len);
    }
    // piece size: whole planes, aiming at >= 16 items per resident workgroup slot (512)
    uint64_t piece = tu.nsplit > 0 ? std::max<uint64_t>(1, (tot / std::max<size_t>(nt, 1) + tu.nsplit - 1) / tu.nsplit)
                                   : std::max<uint64_t>(1, tot / (16 * 512));
    if (tu.nsplit == 0 && tu.lockstep)  // equal, short items: the two items of a workgroup run in lockstep
        piece = std::min<uint64_t>(piece, std::max<uint64_t>(cpp, (uint64_t)tu.ls_item_chunks));
    piece = (piece + cpp - 1) / cpp * cpp;
    const size_t i0 = I.size();
    const uint32_t maxpieces = (uint32_t)((maxlen + piece - 1) / piece);
    // Piece-major, so that neighbours in launch order share planes.  The lockstep kernel pairs consecutive items: inside
    // a group of pieces equal lengths are kept together, longest first, tiles in order (only a tile's last piece can be
    // shorter; with whole-plane pieces every item of the group has the same length and nothing moves).  One counting
    // pass, one placing pass: at 100 000 sketches the host plans 306 000 tiles and nothing hides it (profiles/r4y).
    const bool by_length = tu.lockstep && piece <= 65536;
    std::vector<uint32_t> &first = pp.sort_first;
    for (uint32_t s = 0; s < maxpieces; ++s) {
        const size_t g0 = I.size();
        const uint64_t off = (uint64_t)s * piece;
        size_t cnt = 0;
        uint32_t lmin = ~0u, lmax = 0;
        if (by_length) first.assign((size_t)piece + 2, 0);
        for (size_t t = bd.first; t < bd.second; ++t) {
            const uint64_t b0 = CR[t].x + off;
            if (b0 >= CR[t].y) continue;
            const uint32_t len = (uint32_t)std::min<uint64_t>(CR[t].y - b0, piece);
            ++cnt;
            lmin = std::min(lmin, len);
            lmax = std::max(lmax, len);
            if (by_length) ++first[piece - len + 1];
        }
        I.resize(g0 + cnt);
        U4 *out = I.data() + g0;
        const bool place = by_length && lmin != lmax;
        if (place)
            for (size_t v = 1; v < first.size(); ++v) first[v] += first[v - 1];
        for (size_t t = bd.first; t < bd.second; ++t) {
            const uint64_t b0 = CR[t].x + off;
            if (b0 >= CR[t].y) continue;
            const uint32_t len = (uint32_t)std::min<uint64_t>(CR[t].y - b0, piece);
            const U4 it{(uint32_t)(t - bd.first), (uint32_t)b0, (uint32_t)b0 + len, 0};
            if (place) out[first[piece - len]++] = it;
            else *out++ = it;
        }
        if (!by_length && tu.lockstep && lmin != lmax)
            std::stable_sort(I.begin() + g0, I.end(), [](const U4 &x, const U4 &y) { return x.z - x.y > y.z - y.y; });
    }
    // overflow fragments: the band's last `over` items (whole planes all of them: piece = one plane) in f pieces each
    uint64_t over = 0;
    uint32_t f = 1;
    overflow_fragments(tu, cpp, piece, I.size() - i0, over, f);
    if (pp.band_frags.size() <= bi) pp.band_frags.resize(bi + 1, 0);
    pp.band_frags[bi] = 0;
    if (over) {
        std::vector<U4> &tail = pp.sort_tmp;
        tail.assign(I.end() - (ptrdiff_t)over, I.end());
        I.resize(I.size() - over);
        const uint32_t fr = cpp / f;
        for (const U4 &it : tail)
            for (uint32_t b0 = it.y; b0 < it.z; b0 += fr) I.push_back(U4{it.x, b0, std::min<uint32_t>(it.z, b0 + fr), 1});
        pp.band_frags[bi] = (uint32_t)(over * f);
    }
    if (pp.band_items.size() <= bi) pp.band_items.resize(bi + 1);
    pp.band_items[bi] = std::make_pair(i0, I.size());
}

bool build_pairs(const Layout &L, const PairQuery &q, const Tuning &t, PairPlan &pp)
{
    if (!build_tiles(L, q, t, pp)) return false;
    for (size_t bi = 0; bi < pp.bands.size(); ++bi) build_band_items(t, pp, bi);
    return true;
}

void emit_tile_lists(const Layout &L, const PairPlan &pp, U4 *pinT, U4 *pinF)
{
    // The tile kernel's list (launch order: XCD-interleaved, see xcd_order) holds {row block, column block, ..}.
    // k_finalize has its own list, every segment in ROW-MAJOR order: {row block, column block, plane begin | plane end
    // << 8 | smallest << 16 | largest << 24 register value of the two blocks' sketches (its histogram columns only span
    // the values the tile's sketches can hold), index of the tile's C(v) block in the band}.  A block of k_finalize
    // writes one row of a tile into row perm[si] of the packed matrix, scattered over the row (the columns are
    // key-ordered); block b runs on XCD b % 8 = tile row % 8, so a given output row is always written through the same
    // L2.  Walking a tile row's tiles one after the other keeps that row's lines in L2 until they are complete (the 128
    // rows of a tile row are 5 MB at C3, spread over the 8 L2s); in the tile kernel's interleaved order 8 tile rows were
    // in flight at once, lines left the L2 partly written and WRITE_SIZE was 6x the output (profiles/r3a, r3i).
    for (size_t bi = 0; bi < pp.bands.size(); ++bi) emit_band_lists(L, pp, bi, pinT, pinF);
}

void emit_band_lists(const Layout &L, const PairPlan &pp, size_t bi, U4 *pinT, U4 *pinF)
{
    const std::vector<U4> &T = pp.T;
    const size_t b0 = pp.bands[bi].first;
    for (const Seg &sg : pp.segs[bi])
        for (size_t t = sg.b; t < sg.e; ++t) {
            int lo, hi;
            tile_vrange(L, T[t], lo, hi);
            const uint32_t pl = T[t].z | (T[t].w << 8);
            pinT[t] = U4{T[t].x, T[t].y, pl, (uint32_t)lo | ((uint32_t)hi << 8)};
            const size_t at = sg.b + pp.rank[t];
            // (w: the tile's C(v) block in its band -- at most 2^16 tiles per band -- and, above it, the tile's part: the
            // signalling k_finalize counts a finished tile into that part)
            const uint32_t part = pp.nparts ? (uint32_t)((std::upper_bound(pp.part_first.begin(), pp.part_first.end() - 1, t) - pp.part_first.begin()) - 1) : 0u;
            pinF[at] = U4{T[t].x, T[t].y, pl | ((uint32_t)lo << 16) | ((uint32_t)hi << 24), (uint32_t)(t - b0) | (part << 16)};
        }
}

uint64_t band_item_count(const Tuning &tu, const PairPlan &pp, size_t bi)
{
    // (the arithmetic of build_band_items, without writing anything: the device and staging buffers are sized before the
    // first band is launched)
    const auto &bd = pp.bands[bi];
    const std::vector<U2> &CR = pp.chunks;
    const uint32_t KC = (uint32_t)tu.kc;
    const uint32_t cpp = tu.W >= KC ? tu.W / KC : 1;
    const size_t nt = bd.second - bd.first;
    uint64_t tot = 0;
    for (size_t t = bd.first; t < bd.second; ++t) tot += CR[t].y - CR[t].x;
    uint64_t piece = tu.nsplit > 0 ? std::max<uint64_t>(1, (tot / std::max<size_t>(nt, 1) + tu.nsplit - 1) / tu.nsplit)
                                   : std::max<uint64_t>(1, tot / (16 * 512));
    if (tu.nsplit == 0 && tu.lockstep) piece = std::min<uint64_t>(piece, std::max<uint64_t>(cpp, (uint64_t)tu.ls_item_chunks));
    piece = (piece + cpp - 1) / cpp * cpp;
    uint64_t cnt = 0;
    for (size_t t = bd.first; t < bd.second; ++t) {
        const uint64_t len = CR[t].y - CR[t].x;
        cnt += len <= piece ? (len ? 1 : 0) : (len + piece - 1) / piece;
    }
    uint64_t over = 0;
    uint32_t f = 1;
    overflow_fragments(tu, cpp, piece, cnt, over, f);
    return cnt - over + over * f;
}

}  // namespace plan
}  // namespace dsh

// ---- C-ABI (include/dashing_hip.h): the pure entry points ------------------------------------------------------------
#include "../../include/dashing_hip.h"

extern "C" {

uint64_t dsh_tri_index(uint64_t n, uint64_t i, uint64_t j) { return dsh::plan::tri_index(n, i, j); }

uint64_t dsh_tri_span(uint64_t n, uint64_t rb, uint64_t re) { return dsh::plan::tri_span(n, rb, re); }

int dsh_partition_rows(uint64_t n, uint32_t nparts, uint32_t align, uint64_t *bounds)
{
    if (!bounds || nparts == 0) return DSH_EINVAL;
    dsh::plan::partition_rows(n, nparts, align, bounds);
    return DSH_OK;
}

int dsh_balance_rows(uint64_t n, uint32_t nparts, uint64_t *bounds)
{
    if (!bounds || nparts == 0) return DSH_EINVAL;
    dsh::plan::balance_rows(n, nparts, bounds);
    return DSH_OK;
}

int dsh_balance_rowsets(uint64_t n, uint32_t world, int prep_permille, int dst, int dst_bonus_permille, uint64_t *tab_out,
                        uint32_t cap_words, uint32_t *words_out)
{
    if (world == 0 || (!tab_out && cap_words) || dst >= (int)world) return DSH_EINVAL;
    dsh::plan::RowSets rs;
    dsh::plan::balance_rowsets(n, world, rs, prep_permille < 0 ? ~0u : (uint32_t)prep_permille, dst,
                               dst_bonus_permille < 0 ? ~0u : (uint32_t)dst_bonus_permille);
    if (words_out) *words_out = (uint32_t)rs.words();
    if (!tab_out) return DSH_OK;  // (size query)
    if (rs.words() > cap_words) return DSH_EINVAL;
    rs.write(tab_out);
    return DSH_OK;
}

int dsh_rowsets_from_bounds(const uint64_t *bounds, uint32_t world, uint64_t *tab_out)
{
    if (!bounds || !tab_out || world == 0) return DSH_EINVAL;
    dsh::plan::RowSets rs;
    dsh::plan::rowsets_from_bounds(bounds, world, rs);
    rs.write(tab_out);
    return DSH_OK;
}

int dsh_rowsets_rank(uint64_t n, const uint64_t *tab, uint32_t rank, uint64_t *segs_out, uint32_t cap_segs, uint32_t *nsegs_out,
                     uint64_t *pairs_out, uint64_t *tiles_out)
{
    dsh::plan::RowSets rs;
    if (dsh::plan::parse_rowsets(tab, n, rs) || rank >= rs.world) return DSH_EINVAL;
    uint64_t rb, re;
    std::vector<uint64_t> extra;
    rs.rank_rows(rank, rb, re, extra);
    const uint32_t ns = (rb < re ? 1u : 0u) + (uint32_t)(extra.size() / 2);
    if (nsegs_out) *nsegs_out = ns;
    if (pairs_out) *pairs_out = dsh::plan::rowset_span(n, rb, re, extra);
    if (tiles_out) *tiles_out = dsh::plan::rowset_tiles(n, rb, re, extra);
    if (segs_out) {
        if (ns > cap_segs) return DSH_EINVAL;
        uint32_t at = 0;
        if (rb < re) segs_out[at++] = rb, segs_out[at++] = re;
        for (uint64_t x : extra) segs_out[at++] = x;
    }
    return DSH_OK;
}

int dsh_range_parts(uint64_t n, uint64_t rb, uint64_t re, uint32_t nparts, uint64_t *part_rows, uint32_t *nparts_out)
{
    if (!part_rows || !nparts_out || nparts == 0) return DSH_EINVAL;
    if (re > n) re = n;
    if (rb > re) rb = re;
    std::vector<uint64_t> parts;
    dsh::plan::range_parts(n, rb, re, nparts, parts);
    for (size_t q = 0; q < parts.size(); ++q) part_rows[q] = parts[q];
    *nparts_out = (uint32_t)parts.size() - 1;
    return DSH_OK;
}

}  // extern "C"

// plan.h -- the host-side planner of the compare path: pure C++ (no HIP), so that it is unit-tested on a CPU-only box
// (host/host_capi.cpp exports dshh_plan_check for tests/test_plan.py).
//
// What is planned here replaces the SCHEDULE of the reference's dist_loop (src/sketch_and_cmp.h:785-880:
// perform_core_op row by row, :699-710, with OpenMP dynamic over j) and of dm::parallel_fill
// (distmat/distmat.h:459-512): which pairs run where and in which order, and where each result lands in
// dashing's packed triangle (distmat/distmat.h:260-264).  Three layers:
//   * triangle arithmetic and row partitions (dsh_tri_*, dsh_partition_rows, dsh_balance_rows, dsh_range_parts);
//   * Layout: the column order of the bit-plane matrix for a row range -- the wanted rows first (in parts), then the
//     later rows, each part ordered by the per-sketch key (T, L, hi) -- and the per-128-column-block statistics the
//     per-tile plane ranges come from;
//   * PairPlan: tiles, bands (bounded by the C(v) scratch), segments (tiles of one part inside one band: one k_finalize
//     launch each), work items of the tile kernel, and the two device tile lists.
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <utility>
#include <vector>

#include "consts.h"

namespace dsh {
namespace plan {

struct U4 { uint32_t x, y, z, w; };  // same layout as HIP's uint4: the device lists are arrays of it
struct U2 { uint32_t x, y; };

// fields of a per-sketch key (k_selfhist_card): bad << 31 | hi << 18 | T << 12 | L << 6 | lo
inline int key_lo(uint32_t k) { return (int)(k & 63u); }
inline int key_L(uint32_t k) { return (int)((k >> 6) & 63u); }
inline int key_T(uint32_t k) { return (int)((k >> 12) & 63u); }
inline int key_hi(uint32_t k) { return (int)((k >> 18) & 63u); }
inline bool key_bad(uint32_t k) { return (k & 0x80000000u) != 0; }

// ---- triangle arithmetic (distmat/distmat.h:260-264) and row partitions -------------------------------------------
uint64_t tri_index(uint64_t n, uint64_t i, uint64_t j);
uint64_t tri_span(uint64_t n, uint64_t rb, uint64_t re);
void partition_rows(uint64_t n, uint32_t nparts, uint32_t align, uint64_t *bounds);
void balance_rows(uint64_t n, uint32_t nparts, uint64_t *bounds);
// rows [rb, re) cut into nparts consecutive parts of about equal pair counts, every cut a whole number of 128-row
// tile rows after rb (fewer parts if the range has fewer tile rows); out gets the boundaries
void range_parts(uint64_t n, uint64_t rb, uint64_t re, uint32_t nparts, std::vector<uint64_t> &out);

// ---- row sets: a rank's rows as a range plus top-up tile rows --------------------------------------------------------
// Contiguous row ranges on 128-row boundaries cannot give every rank the same number of tiles when the ranges are a
// few tile rows long (10 000 sketches over 8 ranks: tile rows hold 79 ... 1 tiles, a rank ~395): the short tile rows
// at the BOTTOM of the triangle are therefore dealt one by one to the ranks whose range falls short of the mean.  The
// partition is a table of row segments with owners (include/dashing_hip.h, dsh_balance_rowsets):
//   tab[0] = world, tab[1] = nseg, tab[2 .. 2 + nseg] = the nseg + 1 segment boundaries from 0 to n,
//   tab[3 + nseg .. 3 + 2 nseg) = the owner of every segment.
// A rank's rows are the segments it owns, adjacent ones merged: the first is its MAIN range [rb, re) -- its plane matrix
// holds the sketches rb .. n-1 -- the others are its EXTRA segments.  A rank with extra segments must have all its
// segment boundaries on multiples of 128 (or n): the runs of its layout are then whole 128-column blocks.
struct RowSets {
    uint32_t world = 0;
    std::vector<uint64_t> seg;    // nseg + 1 boundaries
    std::vector<uint32_t> owner;  // nseg
    size_t words() const { return 3 + 2 * owner.size(); }
    void write(uint64_t *tab) const;
    // the rows of rank r: main range (rb == re: none) and the extra segments, flat {b0, e0, b1, e1, ...}
    void rank_rows(uint32_t r, uint64_t &rb, uint64_t &re, std::vector<uint64_t> &extra) const;
};
// nullptr when the table is well-formed for n rows, else what is wrong with it
const char *parse_rowsets(const uint64_t *tab, uint64_t n, RowSets &rs);
void rowsets_from_bounds(const uint64_t *bounds, uint32_t world, RowSets &rs);
// main ranges as balance_rows() makes them over the top of the triangle + the bottom tile rows dealt as top-ups, so
// that the largest cost of any rank (tiles + its own prepare) is smallest; plain balance_rows() ranges when the
// collection is large (n > kTopupMaxRows: a tile row is then a small fraction of a rank's work and the row-sorted
// exchange would have to stage gigabytes) or a rank would get fewer than two tile rows
constexpr uint64_t kTopupMaxRows = 32768;
// prep_permille: the weight of a rank's own prepare, in thousandths of a tile per 128 columns of its plane matrix
// (~0u: the default).  dst >= 0: the rank that RECEIVES the others' rows and sends nothing takes dst_bonus_permille
// (~0u: the default: 120 where a rank holds at least 16 tile rows, else 0) thousandths of a rank's mean tile count more than
// the others.
void balance_rowsets(uint64_t n, uint32_t world, RowSets &rs, uint32_t prep_permille = ~0u, int dst = -1, uint32_t dst_bonus_permille = ~0u);
// the wanted segments in the WANTED ORDER: the extra segments (row order) first, then the main range.  Every segment is
// key-ordered as one run in a row-sorted layout; `rowsorted`: the main range comes as TWO runs where rowsorted_split() cuts it.
void wanted_order(uint64_t n, uint64_t rb, uint64_t re, const std::vector<uint64_t> *extra, std::vector<std::pair<uint64_t, uint64_t>> &segs,
                  bool rowsorted = false);
// A row-sorted range that reaches far down the triangle (2-4 ranks) ends in a run of its own: in ONE key-ordered run the
// last tile rows of the layout -- few tiles each, computed by the rank's last launch -- hold rows from anywhere in the
// range, i.e. as much output per row as any (2 ranks of BASELINE configs[2]: 31 MB final only behind the last kernel);
// as a run of their own they hold the range's LAST rows, the short ones of the triangle (10 MB).  Returns the first row
// of that run (rb + a multiple of 128), or re: no cut -- unless the last ~15 % of the range's tiles are tile rows whose
// rows are on average at most 0.65 as long as the range's, the cut only costs planes per tile.
uint64_t rowsorted_split(uint64_t n, uint64_t rb, uint64_t re);
// tiles a rank with these rows computes, and the rows it holds
uint64_t rowset_tiles(uint64_t n, uint64_t rb, uint64_t re, const std::vector<uint64_t> &extra);
uint64_t rowset_rows(uint64_t rb, uint64_t re, const std::vector<uint64_t> &extra);
uint64_t rowset_span(uint64_t n, uint64_t rb, uint64_t re, const std::vector<uint64_t> &extra);

// default caps of the two listed tails (profiles/r3f/list_cap_sweep.jsonl)
int auto_list_cap(int p, bool upper);

// ---- row-sorted parts (the exchange of SHORT row ranges) -------------------------------------------------------------
// Parts that are runs of original rows (range_parts) are key-ordered each on its own: a range of a few tile rows cut
// into 8 parts has one 128-row block per part, i.e. no ordering at all, and its tiles need 10 planes instead of 8.7
// (profiles/r4b: C3 over 8 ranks).  For such ranges the wanted rows are key-ordered as ONE run and a part is a run of
// whole tile rows of THAT order; the rank's buffer then holds its rows in key order (row s at rowoff[s]) and the
// destination puts the rows of a received part into place (k_row_place).
// rowsorted_rule: a range of fewer than 1024 rows per part whose span is at most 1 GiB (the destination stages it).
bool rowsorted_rule(uint64_t n, uint64_t rb, uint64_t re, uint32_t nparts);
// cut positions (multiples of 128 rows of the key order, front() = 0, back() = number of wanted rows) of about equal
// OUTPUT, at most nparts of them (nparts >= the tile rows: every tile row a part).  The positions count WANTED rows in the wanted order
// (extra segments first, then the main range): the compact order the rank's buffer and the destination's tables use.
void rowsorted_part_positions(uint64_t n, uint64_t rb, uint64_t re, uint32_t nparts, std::vector<uint64_t> &pos,
                              const std::vector<uint64_t> *extra = nullptr);
// the key order of the rows [lo, hi): dst[0 .. hi - lo) = their indices, stable by (T, L, hi) of the per-sketch keys
void sort_rows_by_key(const uint32_t *keys, uint64_t lo, uint64_t hi, uint32_t *dst, std::vector<uint32_t> &scratch);
// offsets of the rows of a row-sorted buffer: rowoff[s] = sum over s' < s of (n - 1 - order[s']), rowoff[cnt] = span
void rowsorted_offsets(uint64_t n, const uint32_t *order, uint64_t cnt, std::vector<uint64_t> &rowoff);

// ---- column layout of the plane matrix ------------------------------------------------------------------------------
// The columns of a sorted layout are RUNS of consecutive original rows in row order, each key-ordered on its own: the
// parts of the wanted range [rb, re), then -- without extra segments -- the later rows as one run; with extra segments
// the later rows are cut where the extra segments lie: [gap | extra | gap | extra | ...], so that the tile rows of an
// extra segment only meet the columns to their right, i.e. the later rows (every pair they compute is theirs).
struct Layout {
    int sorted = 0;               // 0: identity over all n sketches.  1: the sub-collection {rb .. n-1}, key-ordered
    uint64_t rb = 0, re = 0;      // wanted rows of a sorted layout (rb = 0, re = n: the whole collection)
    std::vector<uint64_t> extra;  // further wanted segments {b0, e0, ...} after re (plan.h, row sets), all on 128-row boundaries
    std::vector<std::pair<uint32_t, uint32_t>> wtr;  // the tile rows (128-column blocks) that hold wanted rows, as block ranges in the WANTED ORDER
    std::vector<uint64_t> wtr_w;  // wanted rows in front of each of these ranges (wanted order: extra segments first, then the main range)
    uint64_t nwanted = 0;         // wanted rows in all
    std::vector<uint64_t> part_w;    // the parts as counts of wanted rows in the wanted order, front() = 0, back() = nwanted
    std::vector<uint64_t> rowoff_w;  // (rowsorted) offset of the w-th wanted row (layout order) in the rank's buffer, [nwanted + 1]
    std::vector<uint64_t> parts;  // row boundaries of the parts of the wanted rows, front() = rb, back() = re
    int rowsorted = 0;            // the wanted rows are ONE key-ordered run; the parts are runs of whole tile rows of that order
    std::vector<uint64_t> part_pos;  // the parts as positions of the layout (without extra segments: = part_w)
    std::vector<uint64_t> rowoff;    // (rowsorted) offset of the row at layout position s in the rank's buffer (unwanted positions: 0)
    uint64_t n = 0, ncols = 0;    // sketches in the collection; real columns of the plane matrix
    uint32_t Npad = 0;            // ncols padded to whole 128-column blocks
    bool whole = false;           // sorted over the whole collection: perm also holds its inverse at [n, 2n)
    std::vector<uint32_t> perm;   // plane-matrix column -> sketch
    std::vector<uint8_t> blk_T, blk_lo, blk_L, blk_hi;  // per block: max high threshold, min value, min low threshold, max value
    int vlo = 0, vhi = 0;         // register value range of the columns
    int pbase = 0;                // plane pl is the threshold pbase + 1 + pl
    uint32_t P = 0;               // dense planes
    std::vector<uint32_t> sort_a;  // scratch of the column sort
};

// keys: the n per-sketch keys (only [rb, n) is read for a sorted layout).  `parts` as range_parts() gives them
// (ignored for the identity layout).
// rowsorted_nparts > 0 (sorted layouts only): row-sorted parts, `parts` is ignored.
// extra (sorted layouts only): further wanted segments; the wanted range is then ONE part (`parts` is ignored).
void build_layout(const uint32_t *keys, uint64_t n, int want_sorted, uint64_t rb, uint64_t re,
                  const std::vector<uint64_t> &parts, Layout &L, uint32_t rowsorted_nparts = 0,
                  const std::vector<uint64_t> *extra = nullptr);

// dense plane range of the tile (ti, tj): C(v) is needed for v in (max(larger of the two minima, smaller of the two
// low thresholds), larger of the two high thresholds] -- below that every C(v) is 0 or comes from the low-list join
void tile_planes(const Layout &L, uint32_t ti, uint32_t tj, int &pb, int &pe);

// ---- the pair plan ------------------------------------------------------------------------------------------------------
struct PairQuery {
    int rect = 0;          // rows x columns rectangle (identity layout) instead of triangle rows
    int sorted_rows = 0;   // rows index plane columns of the whole sorted layout (shards, band-wise kNN)
    int want_parts = 0;    // an event per part of the layout (dsh_dist_rows_parts_device_async)
    uint64_t row_begin = 0, row_end = 0, col_begin = 0, col_end = 0;
};

struct Tuning {
    uint32_t W = 0;        // 32-bit words per plane
    int kc = 32;           // k-rows per LDS stage
    int cum_bytes = 2;     // bytes of a C(v) count
    uint64_t cum_budget = 8ull << 30;
    int nsplit = 0;
    bool lockstep = true;
    int ls_item_chunks = 64;  // lockstep kernel: work items of at most about this many K-chunks (whole planes; 16/32/64 within noise, profiles/rd5j)
    uint32_t part_band_tiles = 2048;  // a part of at least this many tiles also ends the band of the tile kernel
    // Tail bands (jobs with parts = the exchange; jobs of at most 64 rounds of one-plane items, in bands of at most 16).  The lockstep
    // tile kernel runs in ROUNDS of round_items work items (2 per workgroup, one workgroup per CU); a rank's parts only
    // become final during k_finalize, i.e. after the whole tile kernel, and its link then needs longer for them than
    // k_finalize takes.  So the tile kernel is cut into a head band and `tail_bands` tail bands at multiples of a round
    // (a cut elsewhere rounds every band up: +1 round, profiles/r4i): the head's parts travel while the tails compute.
    // Each tail takes tail_permille of the rounds left; no cut is made if it would add a round.
    uint32_t round_items = 512;
    // Overflow fragments: a band of one-plane items whose count is a little above a multiple of a round would spend a
    // whole round on the few items left over (3 614 items = 7 rounds + 30 items: 1.56 ms instead of 1.37).  When the
    // overflow is at most overflow_frag_max_permille of a round, those items are cut into f equal fragments of a plane
    // (f a power of two, overflow x f <= one round): the extra round then lasts 1/f of a round.  A band of less than a
    // round (a small job; the tail band of an exchange call, which can hold a few dozen items) is all overflow.  A fragment ADDS its
    // partial counts to the plane's C(v) block with atomics (cleared first); whole items keep their plain stores.
    uint32_t overflow_frag_max_permille = 500;  // 0: never
    uint32_t tail_bands = 2;
    uint32_t tail_permille = 100;   // the last tail: one round of the tile kernel of a rank of BASELINE configs[2] over 8
    uint32_t tail_permille2 = 350;  // share of the rounds left for the tails in front of the last one
    uint32_t tail_head_min_rounds = 7;  // a tail in front of the last one must leave the head at least this many rounds
};

struct Seg {
    size_t b, e;     // tiles [b, e) of the plan
    int part;        // part completed by this segment, or -1
    int hist_bins;   // largest value span of its tiles (LDS histogram columns of the k_finalize launch)
};

struct PairPlan {
    std::vector<U4> T;                                   // {row block, col block, plane begin, plane end}, launch order
    std::vector<std::pair<size_t, size_t>> bands;        // tile ranges, one tile-kernel launch each
    std::vector<std::pair<size_t, size_t>> band_items;   // item ranges of the bands
    std::vector<std::vector<Seg>> segs;                  // per band
    std::vector<U4> items;                               // {tile index in band, chunk begin, chunk end, 1 for a fragment of a plane}
    std::vector<uint32_t> band_frags;                    // per band: its LAST band_frags[b] items are fragments
    std::vector<uint32_t> rank;                          // position of tile t in the row-major order of its segment
    std::vector<U2> chunks;                              // chunk range of every tile
    uint64_t per_tile_bytes = 0;                         // C(v) scratch per tile
    size_t max_band = 0;                                 // tiles of the largest band
    uint32_t nparts = 0;                                 // parts that get an event (0 without want_parts)
    std::vector<uint32_t> part_tiles;                    // tiles of every part (all bands): a part is final when they are finalized
    std::vector<size_t> part_first;                      // first tile of every part in T (+ the end): the part of a tile
    std::vector<uint32_t> sort_first;                    // scratch of the item sort
    std::vector<U4> sort_tmp;
};

// returns false when there is nothing to compute
bool build_pairs(const Layout &L, const PairQuery &q, const Tuning &t, PairPlan &pp);
// the same in two steps, so that a caller can launch band b while it plans band b + 1 (engine.hip): tiles, bands,
// segments and chunk ranges of the whole job, then the work items of one band at a time (in band order)
bool build_tiles(const Layout &L, const PairQuery &q, const Tuning &t, PairPlan &pp);
void build_band_items(const Tuning &t, PairPlan &pp, size_t band);
// the two device lists: the tile kernel's {row block, col block, pb | pe << 8, lo | hi << 8} in launch order and
// k_finalize's {row block, col block, pb | pe << 8 | lo << 16 | hi << 24, index of the tile's C(v) block in its band},
// every segment row-major
void emit_tile_lists(const Layout &L, const PairPlan &pp, U4 *tiles_out, U4 *ftiles_out);
// one band's entries of the two lists (the same arrays, the band's own index range)
void emit_band_lists(const Layout &L, const PairPlan &pp, size_t band, U4 *tiles_out, U4 *ftiles_out);
// how many items build_band_items will append for the band
uint64_t band_item_count(const Tuning &t, const PairPlan &pp, size_t band);

}  // namespace plan
}  // namespace dsh

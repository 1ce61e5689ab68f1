// kernels_compare.hip -- all-pairs HLL compare on gfx950 (MI355X, CDNA4, wave64).
//
// Replaces hot loop 2 of the reference: perform_core_op (src/sketch_and_cmp.h:699-710) /
// dm::parallel_fill (distmat/distmat.h:459-512) calling result_cmp (src/dashing.h:568-592)
// -> hll_t::jaccard_index -> union_size = estimate(histogram(max(a,b))).
//
// Formulation.  Per pair the reference needs the exact histogram
// c[x] = #{t : max(a_t,b_t) = x}.  For a tile with the dense value range (Lp, T] it is assembled from three exact pieces:
//  (1) dense part, Lp <= x < T: thermometer bit-planes A_v[t] = (a_t < v), v in (Lp, T]:
//        C(v) = #{t : max(a_t,b_t) < v} = popcount(A_v & B_v),   c[x] = C(x+1) - C(x)
//      -- the O(N^2 * 2^p) work: v_and_b32 + v_bcnt_u32_b32 over LDS-staged planes (integer work, no MFMA);
//  (2) upper tail, x > T: every sketch lists the (position, value) of its <= emax registers above its own T_i <= T (the
//      geometric tail of the register law); the tail bins are the sum of the two sketches' tail histograms minus, at
//      every position BOTH list, the smaller value; C(T+1) = m - |union| gives c[T];
//  (3) lower tail, x < Lp: every sketch also lists its <= elow registers below its L_i (the lower tail falls off
//      double-exponentially: the bottom planes are nearly empty); the bins below Lp are exactly the positions both
//      sketches list there, at the larger value; their number is C(Lp).
// The positions two sketches share are found by a sparse join through a per-column-block position index, not by walks.
// The estimator then runs once per pair in fp64.
//
// Kernels:
//   k_selfhist_card   per sketch: 64-bin histogram (LDS atomics) -> value range, thresholds T_i / L_i, key, the list
//   k_card_from_hist  one lane per sketch: cardinality from the histogram
//   k_build_colindex  per 128-column block: its sketches' listed registers bucketed by (position, tail)
//   k_transform(_t)   uint8 registers [N][m] -> bit-plane matrix planes[K][Npad] (u32 words, row kk = plane*W + word,
//                     sketch index fastest)
//   k_pair_counts_ls  128x128-sketch tiles, two per 512-thread workgroup, AND and BCNT batches phase-locked across the waves
//                     of a SIMD (one barrier per k-row); plane rows streamed through double-buffered LDS by LDS-DMA, each
//                     lane owns an 8x8 block of pairs; writes C(v) per pair.  (k_pair_counts: the free-running form, p < 9)
//   k_finalize        one lane per pair: C(v) differences + the two tail joins -> histogram (LDS column) -> estimator -> J
//                     -> Mash transform -> float at the packed index
//   k_topk(_merge)    nearest neighbours;  k_unpermute*  the older shard scheme;  k_upload  in-order list uploads
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>

#include "estimators.h"
#include "kernels.h"

namespace dsh {

// ------------------------------------------------------------------------------------------
// per-sketch pass.  block = 256 threads = 4 waves, one sketch per wave.
// PT = type of a listed position: uint16_t for p <= 15, uint32_t above.
// Per sketch: 64-bin histogram -> cardinality; value range [lo, hi]; the HIGH threshold T (the registers above T,
// at most emax of them, are listed: the geometric upper tail of the register law) and the LOW threshold L (the
// registers below L, at most elow of them, are listed too: the lower tail falls off double-exponentially, so two
// or three nearly empty bit-planes are traded for ~150 list entries); lo <= L <= T <= hi.  The list holds
// (position, value) in no particular order -- k_finalize joins lists through a position index
// (k_build_colindex), nothing walks them in order.
// key = bad << 31 | hi << 18 | T << 12 | L << 6 | lo.
template <typename PT>
__global__ __launch_bounds__(256, 4) void k_selfhist_card(const uint8_t *__restrict__ regs, uint64_t first,
                                                        uint64_t n, int p, int estim, int emax, int elow,
                                                        uint32_t *__restrict__ hist_out,
                                                        PT *__restrict__ exc,
                                                        uint8_t *__restrict__ excv,
                                                        uint32_t *__restrict__ exc_n,
                                                        uint32_t *__restrict__ keys,
                                                        uint8_t *__restrict__ tailhist)
{
    __shared__ uint32_t hist[4][64];
    __shared__ uint32_t sub[4][8][72];  // 8 privatised copies per wave: the register values pile up in ~8 bins, so
                                        // one copy would serialise its LDS atomics; rows padded to 72 words: copy c's
                                        // bin b is in bank (8c + b) mod 64, so the 8 copies of a window of 8 adjacent
                                        // values -- where nearly all registers are -- use 64 different banks (a stride
                                        // of 64 put every copy's bin b into bank b, 65 a window into 15 banks)
    __shared__ int thr[4], thrL[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint64_t s = first + (uint64_t)blockIdx.x * 4 + wave;  // sketches [first, n)
#pragma unroll
    for (int k = 0; k < 8; ++k) sub[wave][k][lane] = 0;
    __syncthreads();
    const uint64_t m = 1ull << p;
    const uint4 *src = reinterpret_cast<const uint4 *>(regs + (s < n ? s : 0) * m);
    const uint64_t nch = m >> 4;  // 16-byte chunks of this sketch
    // the first kRegCache*64 chunks (the whole sketch for p <= 14) are loaded once, all loads in
    // flight together, and reused by the second pass; larger sketches re-read the remainder
    constexpr int kRegCache = 16;
    uint4 cache[kRegCache];
    // (every load is issued -- at a clamped chunk where the sketch is shorter -- and masked afterwards: a load under a
    // condition is a branch region of its own, and the 16 loads went two at a time with a full wait in between)
#pragma unroll
    for (int k = 0; k < kRegCache; ++k) {
        const uint64_t c = (uint64_t)k * 64 + lane;
        cache[k] = src[c < nch ? c : nch - 1];
    }
#pragma unroll
    for (int k = 0; k < kRegCache; ++k)
        if (!(s < n && (uint64_t)k * 64 + lane < nch)) cache[k] = make_uint4(0, 0, 0, 0);
    // registers above q+1 = 64-p+1 cannot come from the register rule (a corrupt or foreign .hll): they would alias
    // into wrong histogram bins (& 63) and break the bit-planes (bytes >= 128); flagged in bit 31 of the key
    uint32_t bad = 0;
    const uint32_t limrep = (uint32_t)(64 - p + 2) * 0x01010101u;
    if (s < n) {
        uint32_t *mysub = sub[wave][lane & 7];
        auto count16 = [mysub, limrep, &bad](const uint4 x) {
            const uint32_t w[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                bad |= (w[k] | (((w[k] & 0x7F7F7F7Fu) | 0x80808080u) - limrep)) & 0x80808080u;  // byte >= 128, or >= q+2
                atomicAdd(&mysub[w[k] & 63], 1u);
                atomicAdd(&mysub[(w[k] >> 8) & 63], 1u);
                atomicAdd(&mysub[(w[k] >> 16) & 63], 1u);
                atomicAdd(&mysub[(w[k] >> 24) & 63], 1u);
            }
        };
#pragma unroll
        for (int k = 0; k < kRegCache; ++k)
            if ((uint64_t)k * 64 + lane < nch) count16(cache[k]);
        for (uint64_t c = (uint64_t)kRegCache * 64 + lane; c < nch; c += 64) count16(src[c]);
    }
    __syncthreads();
    {
        uint32_t t = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) t += sub[wave][k][lane];
        hist[wave][lane] = t;
    }
    __syncthreads();
    if (s < n && lane == 0) {
        const uint32_t *h = hist[wave];
        int lo = 0, hi = 63;
        while (lo < 63 && h[lo] == 0) ++lo;
        while (hi > 0 && h[hi] == 0) --hi;
        // (the cardinality is estimated by k_card_from_hist, one LANE per sketch: here a whole wave would idle
        // behind lane 0's fp64 recurrence -- half of this kernel's instruction issue)
        // high threshold: the largest upper tail that fits its list
        int T = hi;
        uint32_t cnt = 0;
        while (T > lo && cnt + h[T] <= (uint32_t)emax) {
            cnt += h[T];
            --T;
        }
        // low threshold: the largest lower tail (values lo .. L-1) that fits its list, never reaching T
        int L = lo;
        uint32_t cntl = 0;
        while (L < T && cntl + h[L] <= (uint32_t)elow) {
            cntl += h[L];
            ++L;
        }
        thr[wave] = T;
        thrL[wave] = L;
        exc_n[s] = cnt + cntl;
        // the host derives the global ranges from the keys (no same-address global atomics: 3 x N of them cost
        // ~12 ns each) and orders the columns by them
        keys[s] = ((uint32_t)hi << 18) | ((uint32_t)T << 12) | ((uint32_t)L << 6) | (uint32_t)lo;
    }
    if (s < n && __any(bad != 0) && lane == 0) atomicOr(&keys[s], 0x80000000u);  // (after lane 0's plain store above)
    __syncthreads();
    if (s >= n) return;
    // histogram of the listed upper-tail registers, one byte per value: k_finalize starts a pair's tail bins from
    // the two sketches' tail histograms and only corrects the positions both sketches list
    const uint32_t T = (uint32_t)thr[wave], L = (uint32_t)thrL[wave];
    tailhist[s * 64 + lane] = (uint32_t)lane > T ? (uint8_t)hist[wave][lane] : (uint8_t)0;
    hist_out[s * 64 + lane] = hist[wave][lane];
    if (emax == 0 && elow == 0) return;
    // second pass: the listed registers.  A lane first counts its own hits (SWAR: 4 registers per step), one wave
    // prefix sum gives every lane its first slot, then it writes its hits there -- a few hundred instructions per
    // wave instead of a decision (ballot or LDS counter) per register byte, which was two thirds of this kernel.
    PT *dstp = exc + s * kListCap;
    uint8_t *dstv = excv + s * kListCap;
    const uint32_t trep = (T + 1) * 0x01010101u, lrep = L * 0x01010101u;
    auto hits = [trep, lrep](uint32_t x) -> uint32_t {  // bit 7 of byte b set iff register b is > T or < L (bytes are < 128)
        const uint32_t ge = ((x | 0x80808080u) - trep) & 0x80808080u;          // >= T + 1
        const uint32_t lt = ~((x | 0x80808080u) - lrep) & 0x80808080u;         // < L
        return ge | lt;
    };
    uint32_t mine = 0;
#pragma unroll
    for (int k = 0; k < kRegCache; ++k)
        if ((uint64_t)k * 64 + lane < nch) {
            const uint32_t w[4] = {cache[k].x, cache[k].y, cache[k].z, cache[k].w};
#pragma unroll
            for (int q = 0; q < 4; ++q) mine += (uint32_t)__popc(hits(w[q]));
        }
    for (uint64_t c = (uint64_t)kRegCache * 64 + lane; c < nch; c += 64) {
        const uint4 x = src[c];
        mine += (uint32_t)(__popc(hits(x.x)) + __popc(hits(x.y)) + __popc(hits(x.z)) + __popc(hits(x.w)));
    }
    uint32_t slot = mine;  // inclusive prefix sum over the wave, then exclusive
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(slot, d, 64);
        if (lane >= d) slot += o;
    }
    slot -= mine;
    auto emit4 = [&](uint32_t x, uint32_t pos0) {
        uint32_t h = hits(x);
        while (h) {
            const uint32_t b = (uint32_t)__builtin_ctz(h) >> 3;  // byte index of the lowest hit
            h &= h - 1;
            if (slot < kListCap) {  // (only a sketch with out-of-range registers -- refused later -- can list more than planned)
                dstp[slot] = (PT)(pos0 + b);
                dstv[slot] = (uint8_t)((x >> (8 * b)) & 0xFFu);
            }
            ++slot;
        }
    };
#pragma unroll
    for (int k = 0; k < kRegCache; ++k) {
        const uint64_t c = (uint64_t)k * 64 + lane;
        if (c < nch) {
            const uint32_t w[4] = {cache[k].x, cache[k].y, cache[k].z, cache[k].w};
#pragma unroll
            for (int q = 0; q < 4; ++q) emit4(w[q], (uint32_t)(c * 16 + q * 4));
        }
    }
    for (uint64_t c = (uint64_t)kRegCache * 64 + lane; c < nch; c += 64) {
        const uint4 x = src[c];
        emit4(x.x, (uint32_t)(c * 16));
        emit4(x.y, (uint32_t)(c * 16 + 4));
        emit4(x.z, (uint32_t)(c * 16 + 8));
        emit4(x.w, (uint32_t)(c * 16 + 12));
    }
}

// cardinality_estimate(hll_t&) = h.report() (src/dashing.h:492) from the 64-bin histograms of k_selfhist_card: one
// lane per sketch (the estimators are sequential recurrences over the bins).
__global__ __launch_bounds__(64) void k_card_from_hist(const uint32_t *__restrict__ hist, const uint32_t *__restrict__ keys,
                                                       uint64_t first, uint64_t n, int p, int estim, double *__restrict__ card)
{
    const uint64_t s = first + (uint64_t)blockIdx.x * 64 + threadIdx.x;
    if (s >= n) return;
    const uint32_t *h = hist + s * 64;
    struct Bins {  // bin v of the sketch's own histogram
        const uint32_t *h;
        enum { stride = 1 };
        __device__ uint32_t operator()(int v) const { return h[v & 63]; }
        __device__ const uint32_t *at(int v) const { return h + v; }  // (v - 1 >= 0 wherever the estimator reads ahead: kMinPrime >= 1)
    };
    const Bins c{h};
    const uint32_t key = keys[s];
    card[s] = estimate(c, c, p, estim, (int)(key & 63u), (int)((key >> 18) & 63u));
}

// ------------------------------------------------------------------------------------------
// Position index of one 128-column block of the plane matrix: every listed (position, value) of its sketches,
// bucketed by (position, tail).  bucket = (pos >> sh) * 2 + (1 for the lower tail), at most 2^14 position groups.
// k_finalize looks the row sketch's listed positions up here: the two sketches of a pair list the same position
// |list_i| x |list_j| / 2^p times per tail -- a sparse join instead of every pair walking a whole list.
//   rec[b]  one RECORD per bucket, RK + 1 words (16 bytes with RK = 3 for p >= 13, 32 bytes with RK = 7 below, where a
//           bucket holds ~4 entries): word 0 = entries in the bucket | first slot in ent[] << 16, then its first RK
//           entries (0xFFFFFFFF = none).  A look-up is ONE gather; only a bucket with more than RK entries (14 % at C3,
//           5 % at p = 10) sends the reader on to ent[] (round 3 read two bounds and then the entries: three dependent
//           levels, 6x read amplification -- VERDICT r3 item 1b).
//   ent[]   all entries in bucket order (uint16 slots: a block holds at most 128 x 510 entries); an entry is
//           (pos & (2^sh - 1)) << 13 | column within the block << 6 | value.
// The kernel also leaves everything k_finalize needs per COLUMN in layout order, so that nothing there waits for the
// permutation: n_s (listed registers), key_s, card_s, the 64-byte tail histogram th_s and the compact list rl_s[E]
// (position << 8 | value).  One 1024-thread workgroup per column block; counting sort through LDS counters (the order
// inside a bucket is whatever the atomics give: every consumer treats a bucket as a set).  A wave takes 8 sketches; the
// (up to 510) entries of a sketch are loaded together, 8 per lane, before any is used.
template <typename PT, int RK>
__global__ __launch_bounds__(1024) void k_build_colindex(const PT *__restrict__ exc, const uint8_t *__restrict__ excv,
                                                          const uint32_t *__restrict__ exc_n,
                                                          const uint32_t *__restrict__ keys, const double *__restrict__ card,
                                                          const uint8_t *__restrict__ tailhist,
                                                          const uint32_t *__restrict__ perm, uint64_t ncols, int p,
                                                          uint32_t nbuckets, uint32_t ent_stride, uint32_t E,
                                                          uint32_t *__restrict__ rec, uint32_t *__restrict__ ent,
                                                          uint32_t *__restrict__ nS, uint32_t *__restrict__ keyS,
                                                          double *__restrict__ cardS, uint8_t *__restrict__ thS,
                                                          uint32_t *__restrict__ rl)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t cnt[];  // [nown] counters, then first slot << 16 | cursor
    __shared__ uint32_t part[1024];
    __shared__ uint32_t below_s;
    constexpr uint32_t RW = (uint32_t)RK + 1u;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t sh = p > 14 ? (uint32_t)(p - 14) : 0u;
    const uint64_t c0 = (uint64_t)blockIdx.x * kTile;
    // gridDim.y workgroups share a column block, each owning a range of the buckets (a collection of few column blocks
    // -- C3: 79 -- is otherwise 79 workgroups' worth of latency sticking out under the first round of the tile kernel):
    // every one reads all entries, counts and places the ones in its range [blo, blo + nown); its entries start in ent[]
    // behind those of the ranges below (counted on the way).  gridDim.y divides nbuckets (both powers of two).
    const uint32_t G = gridDim.y, g = blockIdx.y;
    const uint32_t nown = nbuckets / G, blo = g * nown;
    for (uint32_t b = tid; b < nown; b += 1024) cnt[b] = 0;
    if (tid == 0) below_s = 0;
    __syncthreads();
    // a wave's 8 sketches x 8 entries per lane are loaded ONCE, all loads in flight together (bucket << 13 | column
    // << 6 | value; 0xFFFFFFFF = none): both passes run from registers -- the kernel is one workgroup per column
    // block, i.e. latency, not throughput
    constexpr int kPer = (int)(kListCap / 64);  // entries per lane and sketch
    constexpr int kSk = (int)(kTile / 16);      // sketches per wave
    // one register per entry: position << 8 | value (what the compact list row holds; 0xFFFFFFFF = none) -- bucket and
    // index entry are re-derived from it in both passes -- and one sketch's loads at a time.  (Round 4: with bucket AND
    // entry kept per entry and all 8 sketches' loads issued together the kernel spilled ~190 registers to scratch, 400
    // bytes per thread, and ran 272 us at C3.)
    uint32_t Ts[kSk], pk[kSk][kPer];
    uint32_t sk[kSk], nes[kSk];  // the 8 sketches of this wave, their list lengths and keys: all 24 loads in flight together
#pragma unroll
    for (int q = 0; q < kSk; ++q) {
        const uint64_t col = c0 + wave + 16u * (uint32_t)q;
        sk[q] = col < ncols ? (perm ? perm[col] : (uint32_t)col) : 0xFFFFFFFFu;
    }
#pragma unroll
    for (int q = 0; q < kSk; ++q) {
        nes[q] = sk[q] != 0xFFFFFFFFu ? exc_n[sk[q]] : 0u;
        Ts[q] = keys[sk[q] != 0xFFFFFFFFu ? sk[q] : 0u];  // (the whole key for now)
    }
#pragma unroll
    for (int q = 0; q < kSk; ++q) {
        const uint32_t sl = wave + 16u * (uint32_t)q;
        const bool ok = sk[q] != 0xFFFFFFFFu;
        const uint64_t s = ok ? sk[q] : 0;
        const uint32_t ne = nes[q];
        const uint32_t key = Ts[q];
        Ts[q] = (key >> 12) & 63u;
        const PT *ps = exc + s * kListCap;
        const uint8_t *vs = excv + s * kListCap;
        const uint64_t col = c0 + sl;
        const bool mine = (uint32_t)q % G == g;  // the workgroups of a block share the columns' side data
        // the column's side data in layout order (padding columns: zeros)
        if (mine && lane == 0) {
            nS[col] = ne;
            keyS[col] = ok ? key : 0u;
            cardS[col] = ok ? card[s] : 0.;
        }
        if (mine && lane < 16) reinterpret_cast<uint32_t *>(thS + col * 64)[lane] = ok ? reinterpret_cast<const uint32_t *>(tailhist + s * 64)[lane] : 0u;
        // the sketch's 16 loads first, whether listed or not (a list has room for kListCap entries; selected afterwards):
        // a load or store under a condition is a branch region of its own, and the loads went one by one
        uint32_t pos[kPer], val[kPer];
#pragma unroll
        for (int u = 0; u < kPer; ++u) {
            pos[u] = (uint32_t)ps[lane + 64u * (uint32_t)u];
            val[u] = (uint32_t)vs[lane + 64u * (uint32_t)u];
        }
#pragma unroll
        for (int u = 0; u < kPer; ++u) {
            const uint32_t e = lane + 64u * (uint32_t)u;
            pk[q][u] = e < ne ? ((pos[u] << 8) | val[u]) : 0xFFFFFFFFu;
            // the compact list row, unused slots filled: k_finalize reads a row without asking for its length first
            if (e < E && mine) rl[col * E + e] = pk[q][u];
        }
        __builtin_amdgcn_sched_barrier(0);  // (the next sketch's loads stay behind this one's: 16 values in flight, not 128)
    }
    auto bucket_of = [sh](uint32_t x, uint32_t T) -> uint32_t {  // (position group, upper | lower tail); none stays none
        return x == 0xFFFFFFFFu ? x : ((((x >> 8) >> sh) << 1) | ((x & 0xFFu) > T ? 0u : 1u));
    };
    uint32_t below = 0;  // entries of the ranges below this workgroup's
#pragma unroll
    for (int q = 0; q < kSk; ++q)
#pragma unroll
        for (int u = 0; u < kPer; ++u) {
            const uint32_t b = bucket_of(pk[q][u], Ts[q]);
            const uint32_t rel = b - blo;  // (0xFFFFFFFF - blo >= nown: "none" is in no range)
            if (rel < nown) atomicAdd(&cnt[rel], 1u);
            else if (b < blo) ++below;
        }
    if (G > 1) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) below += __shfl_xor(below, d, 64);
        if (lane == 0 && below) atomicAdd(&below_s, below);
    }
    __syncthreads();
    // exclusive scan: thread t owns the buckets [t*per, (t+1)*per) of the range
    const uint32_t per = (nown + 1023) / 1024;
    uint32_t sum = 0;
    for (uint32_t b = tid * per; b < (tid + 1) * per && b < nown; ++b) sum += cnt[b];
    // block scan of the 1024 partial sums: within a wave by shuffles, across the 16 waves through part[]
    uint32_t incl = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(incl, d, 64);
        if ((int)lane >= d) incl += o;
    }
    if (lane == 63) part[wave] = incl;
    __syncthreads();
    uint32_t wbase = 0;
    for (uint32_t w = 0; w < wave; ++w) wbase += part[w];
    uint32_t run = below_s + wbase + incl - sum;
    uint32_t *myrec = rec + ((uint64_t)blockIdx.x * nbuckets + blo) * RW;
    for (uint32_t b = tid * per; b < (tid + 1) * per && b < nown; ++b) {
        const uint32_t x = cnt[b];
        uint4 *r4 = reinterpret_cast<uint4 *>(myrec + (uint64_t)b * RW);
        r4[0] = make_uint4(x | (run << 16), 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
        if constexpr (RK > 3) r4[1] = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
        cnt[b] = (run << 16) | run;  // first slot | cursor (at most 128 x 510 < 2^16 entries: the cursor never carries)
        run += x;
    }
    __syncthreads();  // (also orders the record headers before the inline entries below: same workgroup, same L2 lines)
    uint32_t *myent = ent + (uint64_t)blockIdx.x * ent_stride;
#pragma unroll
    for (int q = 0; q < kSk; ++q)
#pragma unroll
        for (int u = 0; u < kPer; ++u)
        {
            uint32_t x = pk[q][u];
            asm volatile("" : "+v"(x));  // (re-derive the bucket: kept from the first pass, 64 of them went to scratch)
            const uint32_t b = bucket_of(x, Ts[q]) - blo;
            if (b < nown) {
                const uint32_t pv = ((((x >> 8) & ((1u << sh) - 1u)) << 13) | ((wave + 16u * (uint32_t)q) << 6) | (x & 0xFFu));
                const uint32_t w = atomicAdd(&cnt[b], 1u);
                const uint32_t slot = w & 0xFFFFu, rel = slot - (w >> 16);
                myent[slot] = pv;
                if (rel < (uint32_t)RK) myrec[(uint64_t)b * RW + 1u + rel] = pv;
            }
        }
}

// ------------------------------------------------------------------------------------------
// registers -> thermometer bit-planes.  Thread (i, w) reads 32 registers of sketch i and emits
// one 32-bit word per plane: the 32 bits of planes[(pl*W + w)*Npad + i] are (reg[i][32w + r] < vlo+1+pl), r = 0..31,
// in the order of lt_word below.
// Padding sketches (i >= N) get zeros (they never count).
// The 32 flags (register < v) of 8 words of 4 register bytes each, y = x | 0x80808080 (bytes of x are < 128, or the
// 0xFF of a padding sketch): bit 7 of every byte of y - v says "byte >= v" (no borrow crosses a byte).  Word k's four
// flags go to bits k, 8+k, 16+k, 24+k -- register 4k + b sits at bit 8b + k of the plane word.  That order is the
// same for every sketch and nothing looks at it: the tile kernel only counts the bits two words share.  (Round 4:
// collecting every word's flags into a nibble in register order first took a 32-bit multiply per word -- 63 M VALU
// instructions at C3, the kernel was bound by them; 3 instructions per word now.)
__device__ __forceinline__ uint32_t lt_word(const uint32_t (&y)[8], uint32_t vrep)
{
    uint32_t ge = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) ge |= ((y[k] - vrep) >> (7 - k)) & (0x01010101u << k);
    return ~ge;
}

__global__ __launch_bounds__(256) void k_transform(const uint8_t *__restrict__ regs, uint64_t n,
                                                    int p, int vlo, uint32_t P, uint32_t W,
                                                    uint32_t Npad, uint32_t *__restrict__ planes,
                                                    const uint32_t *__restrict__ perm)
{
    // column i of the plane matrix holds sketch perm[i] (perm == nullptr: identity)
    const uint64_t gid = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const uint32_t i = (uint32_t)(gid % Npad);
    const uint32_t w = (uint32_t)(gid / Npad);
    if (w >= W) return;
    const uint64_t m = 1ull << p;
    uint32_t x[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) x[k] = 0xFFFFFFFFu;
    if (i < n) {
        const uint8_t *src = regs + (uint64_t)(perm ? perm[i] : i) * m + (uint64_t)w * 32;
        const uint4 a = *reinterpret_cast<const uint4 *>(src);
        x[0] = a.x; x[1] = a.y; x[2] = a.z; x[3] = a.w;
        if (m >= 32) {
            const uint4 b = *reinterpret_cast<const uint4 *>(src + 16);
            x[4] = b.x; x[5] = b.y; x[6] = b.z; x[7] = b.w;
        }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) x[k] |= 0x80808080u;
    for (uint32_t pl = 0; pl < P; ++pl) {
        const uint32_t vrep = (uint32_t)(vlo + 1 + (int)pl) * 0x01010101u;
        planes[((uint64_t)pl * W + w) * Npad + i] = lt_word(x, vrep);
    }
}

// Same transform for 2^p >= 256 with coalesced reads: a workgroup takes 64 sketches x 256 register bytes,
// reads them as 256-byte runs of one sketch each (the thread-per-(sketch, word) mapping above touches 64
// different sketches per wave, 32 B each), stages them in LDS (row stride 17 x 16 B: conflict-free for the
// 16-byte reads of the second phase) and writes 64 consecutive columns per plane row.
__global__ __launch_bounds__(256) void k_transform_t(const uint8_t *__restrict__ regs, uint64_t n,
                                                      int p, int vlo, uint32_t P, uint32_t W,
                                                      uint32_t Npad, uint32_t *__restrict__ planes,
                                                      const uint32_t *__restrict__ perm)
{
    constexpr uint32_t kRow = 272;  // bytes per staged sketch row (256 + 16)
    __shared__ __attribute__((aligned(16))) uint8_t stage[64 * kRow];
    const uint32_t tid = threadIdx.x;
    const uint32_t i0 = blockIdx.x * 64;
    const uint64_t b0 = (uint64_t)blockIdx.y * 256;
    const uint64_t m = 1ull << p;
    // the four 16-byte pieces of this thread: column indices first, then all four loads (from a clamped column where the
    // block sticks out of the collection, masked afterwards), then the stores -- written as one loop the loads went one
    // by one, each behind the previous piece's permutation look-up
    uint32_t src_col[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const uint32_t i = i0 + ((tid + 256u * r) >> 4);
        src_col[r] = i < n ? i : (uint32_t)(n - 1);
    }
    if (perm) {
#pragma unroll
        for (int r = 0; r < 4; ++r) src_col[r] = perm[src_col[r]];
    }
    uint4 piece[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const uint32_t c = (tid + 256u * r) & 15u;
        piece[r] = *reinterpret_cast<const uint4 *>(regs + (uint64_t)src_col[r] * m + b0 + c * 16u);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const uint32_t q = tid + 256u * r;
        const uint32_t sl = q >> 4, c = q & 15u;
        if (i0 + sl >= n) piece[r] = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);  // padding sketches: no bit ever set
        *reinterpret_cast<uint4 *>(stage + sl * kRow + c * 16u) = piece[r];
    }
    __syncthreads();
    const uint32_t il = tid & 63u, g = tid >> 6;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const uint32_t wl = g * 2 + h;  // word (32 registers) within the 256-byte run
        const uint4 a = *reinterpret_cast<const uint4 *>(stage + il * kRow + wl * 32u);
        const uint4 b = *reinterpret_cast<const uint4 *>(stage + il * kRow + wl * 32u + 16u);
        const uint32_t y[8] = {a.x | 0x80808080u, a.y | 0x80808080u, a.z | 0x80808080u, a.w | 0x80808080u,
                               b.x | 0x80808080u, b.y | 0x80808080u, b.z | 0x80808080u, b.w | 0x80808080u};
        const uint64_t w = (uint64_t)blockIdx.y * 8 + wl;
        uint32_t *dst = planes + w * Npad + i0 + il;  // plane 0; the next plane is W rows further on
        const uint64_t step = (uint64_t)W * Npad;
        uint32_t vrep = (uint32_t)(vlo + 1) * 0x01010101u;
        for (uint32_t pl = 0; pl < P; ++pl, dst += step, vrep += 0x01010101u) *dst = lt_word(y, vrep);
    }
}

// ------------------------------------------------------------------------------------------
// all-pairs AND+popcount.  One 256-thread workgroup per 128x128 tile of sketches.
//   waves 2x2, each covers 64x64 pairs; lane (ly,lx) of the 8x8 lane grid owns an 8x8 block.
//   per k-row: two ds_read_b128 of A-words (broadcast over lx), two of B-words (broadcast over
//   ly), 64 x (v_and_b32 + v_bcnt_u32_b32 with accumulate) -> 128 VALU per 4 LDS reads.
//   LDS rows are 512 B (128 sketches x 4 B) per operand; the 16-B slots a lane group touches
//   are distinct banks -> conflict-free.
//   K (= planes x words) is streamed in chunks of KC rows through two LDS buffers with
//   direct global->LDS DMA (global_load_lds_dwordx4: the lane-linear LDS image IS our
//   row-major [row][128] layout, 2 rows per wave-instruction), one barrier per chunk: the DMA
//   of chunk c+1 is in flight while chunk c is consumed.
//   At each plane boundary the 64 counters C(v) go to cum[pl][tile*16384 + r*128 + c]
//   (CT = uint16_t when 2^p < 65536, else uint32_t).
__device__ __forceinline__ void popc_acc(uint32_t &acc, uint32_t x)
{
    // v_bcnt_u32_b32 d, s0, s1 : d = popcount(s0) + s1  (hipcc splits this into bcnt + add3)
    asm("v_bcnt_u32_b32 %0, %1, %0" : "+v"(acc) : "v"(x));
}

typedef __attribute__((address_space(3))) void *lds_ptr_t;

// one global_load_lds_dwordx4: 64 lanes x 16 B from per-lane global addresses to the
// lane-linear LDS span [lds_byte_addr, +1 KiB).  M0 (LDS base) is saved/restored inside the
// statement (cdna_hip_programming.md section 5.7).  Not counted by hipcc: pair with dma_wait().
__device__ __forceinline__ void glds16(const void *gsrc, uint32_t lds_byte_addr)
{
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(gsrc), "s"(lds_byte_addr)
        : "memory");
}
__device__ __forceinline__ void dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

__device__ __forceinline__ void store8(uint16_t *dst, const uint32_t *a)
{
    *reinterpret_cast<uint4 *>(dst) = make_uint4(a[0] | (a[1] << 16), a[2] | (a[3] << 16),
                                                 a[4] | (a[5] << 16), a[6] | (a[7] << 16));
}
__device__ __forceinline__ void store8(uint32_t *dst, const uint32_t *a)
{
    reinterpret_cast<uint4 *>(dst)[0] = make_uint4(a[0], a[1], a[2], a[3]);
    reinterpret_cast<uint4 *>(dst)[1] = make_uint4(a[4], a[5], a[6], a[7]);
}

// a fragment's partial counts (overflow fragments, plan.h): 32-bit atomics -- two uint16 counts per word never carry into
// each other: a plane's total is at most 2^p < 65536 whenever CT is uint16_t
__device__ __forceinline__ void add8(uint16_t *dst, const uint32_t *a)
{
    uint32_t *d = reinterpret_cast<uint32_t *>(dst);
#pragma unroll
    for (int i = 0; i < 4; ++i) atomicAdd(d + i, a[2 * i] | (a[2 * i + 1] << 16));
}
__device__ __forceinline__ void add8(uint32_t *dst, const uint32_t *a)
{
#pragma unroll
    for (int i = 0; i < 8; ++i) atomicAdd(dst + i, a[i]);
}

// the C(v) blocks the fragments add to, cleared: one workgroup per fragment item, the one holding its plane's first chunk
template <typename CT>
__global__ __launch_bounds__(256) void k_zero_frag_blocks(const uint4 *__restrict__ frags, uint32_t cpp, CT *__restrict__ cum, uint64_t nslots)
{
    const uint4 it = frags[blockIdx.x];
    if (it.y % cpp) return;
    uint4 *dst = reinterpret_cast<uint4 *>(cum + (uint64_t)(it.y / cpp) * nslots + (uint64_t)it.x * (kTile * kTile));
    constexpr uint32_t n16 = kTile * kTile * sizeof(CT) / 16;
    for (uint32_t i = threadIdx.x; i < n16; i += 256) dst[i] = make_uint4(0, 0, 0, 0);
}

template <int KC, int U, typename CT>
__global__ __launch_bounds__(256) void k_pair_counts(const uint32_t *__restrict__ planes,
                                                      uint32_t Npad, uint32_t Kpad, uint32_t W,
                                                      uint32_t P, const uint4 *__restrict__ tiles,
                                                      const uint4 *__restrict__ items,
                                                      CT *__restrict__ cum, uint64_t nslots)
{
    // work item = (tile, contiguous range of K-chunks inside the tile's own plane range).  Every
    // plane's count is independent, so splitting a tile's planes over workgroups needs no
    // reduction; it shortens the items so the last round of the grid wastes less.
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];  // [2][A|B][KC][128]
    constexpr int NPASS = KC / 8;  // wave-instructions per operand per chunk (2 rows each)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ii = (wave >> 1) * 64 + (lane >> 3) * 8;
    const int jj = (wave & 1) * 64 + (lane & 7) * 8;
    const uint4 item = items[blockIdx.x];  // {tile index in band, chunk begin, chunk end, -}
    const uint32_t tile_id = item.x;
    const uint4 tile = tiles[tile_id];     // {row block, col block, plane begin, plane end}
    // DMA source of this lane: row (2*wave + lane/32) of each 8-row pass, 16 B at column lane%32
    const uint64_t lrow = (uint64_t)(wave * 2 + (lane >> 5));
    const uint32_t *gA = planes + lrow * Npad + (uint64_t)tile.x * kTile + (lane & 31) * 4;
    const uint32_t *gB = planes + lrow * Npad + (uint64_t)tile.y * kTile + (lane & 31) * 4;
    const uint64_t pass_stride = (uint64_t)8 * Npad;

    // LDS-DMA issued from inline asm so that hipcc does not count it: the compiler would
    // otherwise drain it with vmcnt(0) before the first ds_read of the chunk being consumed.
    // We wait for it ourselves (vmcnt(0) right before the barrier that publishes the chunk).
    const uint32_t lds_base = (uint32_t)(uintptr_t)(lds_ptr_t)smem + wave * 1024;  // bytes
    auto stage = [&](uint32_t chunk, int buf) {
        const uint32_t la = lds_base + buf * (2 * KC * 512);
        const uint32_t lb = la + KC * 512;
        const uint32_t *a = gA + (uint64_t)chunk * KC * Npad;
        const uint32_t *b = gB + (uint64_t)chunk * KC * Npad;
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
            glds16(a + ps * pass_stride, la + ps * 4096);
            glds16(b + ps * pass_stride, lb + ps * 4096);
        }
    };

    uint32_t acc[8][8];
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int c = 0; c < 8; ++c) acc[r][c] = 0;

    const uint32_t ch_begin = item.y, ch_end = item.z;
    CT *cum_tile = cum + (uint64_t)tile_id * (kTile * kTile) + (uint64_t)ii * kTile + jj;
    if (ch_begin >= ch_end) return;

    stage(ch_begin, ch_begin & 1);
    for (uint32_t ch = ch_begin; ch < ch_end; ++ch) {
        dma_wait();       // this wave's DMA pieces of chunk ch have landed ...
        __syncthreads();  // ... and so have everyone's; buffer (ch+1)&1 is no longer being read
        if (ch + 1 < ch_end) stage(ch + 1, (ch + 1) & 1);
        const uint32_t *As = smem + (ch & 1) * (2 * KC * 128) + ii;
        const uint32_t *Bs = smem + (ch & 1) * (2 * KC * 128) + KC * 128 + jj;
        // U rows (U = min(W, 8), compile time) per step, then a plane-boundary check
        for (uint32_t s0 = 0; s0 < (uint32_t)KC; s0 += U) {
#pragma unroll
            for (uint32_t kk = 0; kk < (uint32_t)U; ++kk) {
                const uint4 a0 = *reinterpret_cast<const uint4 *>(As + (s0 + kk) * 128);
                const uint4 a1 = *reinterpret_cast<const uint4 *>(As + (s0 + kk) * 128 + 4);
                const uint4 b0 = *reinterpret_cast<const uint4 *>(Bs + (s0 + kk) * 128);
                const uint4 b1 = *reinterpret_cast<const uint4 *>(Bs + (s0 + kk) * 128 + 4);
                const uint32_t av[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
                const uint32_t bv[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
                for (int r = 0; r < 8; ++r)
#pragma unroll
                    for (int c = 0; c < 8; ++c) popc_acc(acc[r][c], av[r] & bv[c]);
            }
            const uint32_t row_end = ch * KC + s0 + U;
            if ((row_end & (W - 1)) == 0) {
                const uint32_t pl = row_end / W - 1;
                if (pl < P) {
                    CT *dst = cum_tile + (uint64_t)pl * nslots;
#pragma unroll
                    for (int r = 0; r < 8; ++r) store8(dst + r * kTile, acc[r]);
                }
#pragma unroll
                for (int r = 0; r < 8; ++r)
#pragma unroll
                    for (int c = 0; c < 8; ++c) acc[r][c] = 0;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// k_pair_counts_ls ("lockstep"): the arithmetic, tiles, work items, LDS staging and cum output of k_pair_counts --
// scheduled so that the waves sharing a SIMD are always in the SAME instruction class.
// Why: on gfx950 a SIMD issues v_and_b32 from two waves alternately at one per ~2.07 cycles and v_bcnt_u32_b32 at one
// per ~4.42, but an AND stream of one wave next to a BCNT stream of another costs 8.03 cycles per (AND,BCNT) pair
// instead of 6.49, whatever the order, batch size or register banks inside a wave (profiles/ubench/pair_sched.txt:
// batch64 8.0, batch64_wg512_bar 6.4-6.8, ..._bar_ldsspread 6.9).  Independent workgroups drift apart; here ONE
// 512-thread workgroup per CU (8 waves = 2 per SIMD) takes TWO work items (waves 0-3 item 2b, waves 4-7 item 2b+1, each
// with its own LDS staging) and every k-row is: 64 ANDs into temporaries | 64 BCNTs | s_barrier (round 2 also had a
// barrier between the two batches; it is not needed -- see the k loop -- and costs 4 %).  The
// operands of the next k-row are read from LDS during the BCNT phase (the AND phase was their last use), one 16-byte
// read per quarter of the phase (all 8 waves arrive together: 32 ds_read_b128 at once back up the LDS queue).
// With a single workgroup per CU nothing hides a stall, so the plane-boundary flush (16 stores per lane) is deferred
// to the start of the next chunk, after that chunk's DMA has been issued: the vmcnt(0) in front of the following
// chunk then finds the stores long retired (un-deferred it costs ~100 cycles per k-row).  Requires W >= KC (p >= 9 at
// KC = 16): plane boundaries are chunk boundaries and the k loop carries no flush test.
// Measured per k-row (s_memtime, profiles/r2k): 955 cycles in the k loop + 40 per-chunk overhead, of which 850 are the
// two phases and their barriers, ~65 the LDS operand reads, ~40 the arrival of the DMA data; folding the chunk
// transition into the last row of a chunk (no extra barrier, no exposed LDS latency) was slower (14.7 vs 14.4 ms).
// A half whose item is shorter (or missing) keeps executing the same instruction stream on stale LDS data -- the
// barriers need every wave -- and simply stores nothing.
// FRAG: the instance for overflow fragments (plan.h) -- items that are pieces of ONE plane; their partial counts are
// added to the plane's C(v) block (cleared by k_zero_frag_blocks) instead of stored.  A launch of its own behind the
// whole items', so that the kernel of every other call stays exactly as it was.
template <int KC, typename CT, bool FRAG = false>
__global__ __launch_bounds__(512) void k_pair_counts_ls(const uint32_t *__restrict__ planes, uint32_t Npad,
                                                         uint32_t Kpad, uint32_t W, uint32_t P,
                                                         const uint4 *__restrict__ tiles,
                                                         const uint4 *__restrict__ items, uint32_t nitems,
                                                         CT *__restrict__ cum, uint64_t nslots)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];  // [half][2][A|B][KC][128]
    constexpr int NPASS = KC / 8;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = wave8 >> 2, wave = wave8 & 3;
    const int ii = (wave >> 1) * 64 + (lane >> 3) * 8;
    const int jj = (wave & 1) * 64 + (lane & 7) * 8;
    const uint32_t mine = 2 * blockIdx.x + (uint32_t)half, other = 2 * blockIdx.x + (uint32_t)(1 - half);
    uint4 item = make_uint4(0, 0, 0, 0);
    if (mine < nitems) item = items[mine];
    uint32_t other_len = 0;
    if (other < nitems) {
        const uint4 o = items[other];
        other_len = o.z > o.y ? o.z - o.y : 0;
    }
    const uint32_t my_len = item.z > item.y ? item.z - item.y : 0;
    const uint32_t trips = my_len > other_len ? my_len : other_len;
    const uint32_t tile_id = item.x;
    const uint4 tile = tiles[tile_id];
    const uint64_t lrow = (uint64_t)(wave * 2 + (lane >> 5));
    const uint32_t *gA = planes + lrow * Npad + (uint64_t)tile.x * kTile + (lane & 31) * 4;
    const uint32_t *gB = planes + lrow * Npad + (uint64_t)tile.y * kTile + (lane & 31) * 4;
    const uint64_t pass_stride = (uint64_t)8 * Npad;
    uint32_t *hsm = smem + half * (4 * KC * 128);
    const uint32_t lds_base = (uint32_t)(uintptr_t)(lds_ptr_t)hsm + wave * 1024;  // bytes
    auto stage = [&](uint32_t chunk, int buf) {
        const uint32_t la = lds_base + buf * (2 * KC * 512);
        const uint32_t lb = la + KC * 512;
        const uint32_t *a = gA + (uint64_t)chunk * KC * Npad;
        const uint32_t *b = gB + (uint64_t)chunk * KC * Npad;
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
            glds16(a + ps * pass_stride, la + ps * 4096);
            glds16(b + ps * pass_stride, lb + ps * 4096);
        }
    };
    uint32_t acc[8][8], tmp[8][8];
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int c = 0; c < 8; ++c) acc[r][c] = 0;
    CT *cum_tile = cum + (uint64_t)tile_id * (kTile * kTile) + (uint64_t)ii * kTile + jj;
    if (trips == 0) return;
    const uint32_t cpp = W / (uint32_t)KC;  // chunks per plane (W >= KC, both powers of two)
    bool flush_due = false;                // the chunk just finished ended a plane: store + clear the counters
    uint32_t flush_pl = 0;
    auto flush = [&]() {
        if (flush_pl < P) {
            CT *dst = cum_tile + (uint64_t)flush_pl * nslots;
            if constexpr (FRAG) {
#pragma unroll
                for (int r = 0; r < 8; ++r) add8(dst + r * kTile, acc[r]);
            } else {
#pragma unroll
                for (int r = 0; r < 8; ++r) store8(dst + r * kTile, acc[r]);
            }
        }
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int c = 0; c < 8; ++c) acc[r][c] = 0;
    };
    if (my_len) stage(item.y, item.y & 1);
    for (uint32_t it = 0; it < trips; ++it) {
        const bool active = it < my_len;  // wave-uniform
        const uint32_t ch = item.y + it;
        if (active) dma_wait();
        __syncthreads();
        if (it + 1 < my_len) stage(ch + 1, (ch + 1) & 1);
        if (flush_due) flush();  // (after the DMA issue: a whole chunk of arithmetic passes before the next vmcnt(0))
        const uint32_t *As = hsm + (ch & 1) * (2 * KC * 128) + ii;
        const uint32_t *Bs = hsm + (ch & 1) * (2 * KC * 128) + KC * 128 + jj;
        uint4 a0 = *reinterpret_cast<const uint4 *>(As), a1 = *reinterpret_cast<const uint4 *>(As + 4);
        uint4 b0 = *reinterpret_cast<const uint4 *>(Bs), b1 = *reinterpret_cast<const uint4 *>(Bs + 4);
        // fully unrolled: the LDS read offsets become immediates (no address arithmetic in the row), and ONE barrier per
        // k-row, after the BCNT batch.  Measured on C3 (profiles/r3f): barrier after both batches 11.40 ms; after the BCNT
        // batch only 10.91 (both waves of a SIMD start their ANDs together and are still together when the BCNTs begin);
        // after the AND batch only 14.2; every second row 12.3; + full unroll 10.55; + KC = 32 10.41.
#pragma unroll
        for (uint32_t kk = 0; kk < (uint32_t)KC; ++kk) {
            {
                const uint32_t av[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
                const uint32_t bv[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
                for (int r = 0; r < 8; ++r)
#pragma unroll
                    for (int c = 0; c < 8; ++c) tmp[r][c] = av[r] & bv[c];
            }
            __builtin_amdgcn_sched_barrier(0);
            const uint32_t kn = kk + 1 < (uint32_t)KC ? kk + 1 : kk;  // (the last row is simply read again)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (g == 0) a0 = *reinterpret_cast<const uint4 *>(As + kn * 128);
                if (g == 1) a1 = *reinterpret_cast<const uint4 *>(As + kn * 128 + 4);
                if (g == 2) b0 = *reinterpret_cast<const uint4 *>(Bs + kn * 128);
                if (g == 3) b1 = *reinterpret_cast<const uint4 *>(Bs + kn * 128 + 4);
#pragma unroll
                for (int r = 2 * g; r < 2 * g + 2; ++r)
#pragma unroll
                    for (int c = 0; c < 8; ++c) popc_acc(acc[r][c], tmp[r][c]);
                __builtin_amdgcn_sched_barrier(0);
            }
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (FRAG) {  // a fragment lies inside one plane: its counts go out behind its last chunk
            flush_due = active && it + 1 == my_len;
            flush_pl = ch / cpp;
        } else {
            flush_due = active && ((ch + 1) & (cpp - 1)) == 0;  // the same chunk in both halves (items are whole planes)
            flush_pl = (ch + 1) / cpp - 1;
        }
    }
    if (flush_due) flush();
}

// ------------------------------------------------------------------------------------------
// WHAT-IF, NOT THE PRODUCT PATH (option "pair_mfma" = 1; default 0).  The north star of this
// build excludes the matrix cores ("the path is integer max + fp reduction, not a dense contraction"), and the
// shipped k_pair_counts above is integer VALU only.  With the bit-plane formulation, though, C(v)[i][j] =
// popcount(A_v[i] & B_v[j]) IS a 0/1 GEMM, and the MFMA pipe is separate from the VALU pipe -- this variant measures
// what that would buy, behind the same interface (same tiles, items, LDS staging, cum output: byte-identical
// results, integer-exact i32 accumulation).  Per k-row (one 32-bit plane word per sketch = 32 register positions =
// the K of one v_mfma_i32_32x32x32_i8) a wave covers 128 x 64 pairs with 4 x 2 MFMAs; lane (r = l & 31, h = l >> 5)
// supplies row/column r of a 32-block and the 16 positions [16h, 16h+16) as 16 bytes of 0/1, expanded in registers
// (nibble -> 4 bytes by one multiply + one mask).  A and B are expanded the same way, so byte e of an A lane and
// byte e of the B lane with the same h stand for the same register position: whatever k-order the hardware assigns
// to the 16 bytes of a lane, the dot product pairs equal positions.
// Built only with `make WHATIF=1` (-DDSH_WHATIF_MFMA): the default library refuses the option.
#ifdef DSH_WHATIF_MFMA
typedef int v4i_t __attribute__((ext_vector_type(4)));
typedef int v16i_t __attribute__((ext_vector_type(16)));

__device__ __forceinline__ v4i_t expand16(uint32_t w16)
{
    // nibble x = b3 b2 b1 b0 -> bytes (b0, b1, b2, b3): x * (1 + 2^7 + 2^14 + 2^21) puts b_j at bit 8j (the four
    // shifted copies occupy disjoint bit ranges, no carries), the mask keeps exactly those
    v4i_t v;
    v.x = (int)((((w16)&0xFu) * 0x00204081u) & 0x01010101u);
    v.y = (int)((((w16 >> 4) & 0xFu) * 0x00204081u) & 0x01010101u);
    v.z = (int)((((w16 >> 8) & 0xFu) * 0x00204081u) & 0x01010101u);
    v.w = (int)((((w16 >> 12) & 0xFu) * 0x00204081u) & 0x01010101u);
    return v;
}

template <int KC, typename CT>
__global__ __launch_bounds__(128, 2) void k_pair_counts_mfma(const uint32_t *__restrict__ planes,
                                                           uint32_t Npad, uint32_t Kpad, uint32_t W,
                                                           uint32_t P, const uint4 *__restrict__ tiles,
                                                           const uint4 *__restrict__ items,
                                                           CT *__restrict__ cum, uint64_t nslots)
{
    // 128 threads = 2 waves per 128x128 tile; wave w covers all 128 rows x columns [64w, 64w+64): 4 A fragments and
    // 2 B fragments feed 8 MFMAs per plane word (6 expansions per 8 MFMAs instead of 4 per 4 with 64x64 wave tiles:
    // the first version was bound by the expansion's VALU instructions, 7.99 ms on C3)
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];  // [2][A|B][KC][128], as k_pair_counts; then the LUT
    constexpr int NPASS = KC / 4;  // a wave-instruction of the LDS-DMA covers 2 rows; 2 waves -> 4 rows per pass
    // byte -> 8 bytes of 0/1 (bit j -> byte j): with the expansion done in VALU (bfe + multiply + mask per nibble, 13
    // instructions per fragment) the kernel issued 11 VALU instructions per MFMA and was bound by them (8.0 ms on C3,
    // matrix pipe 45 % busy); a 2 KiB table in LDS costs 2 VALU + one ds_read_b64 per byte
    uint2 *lut = reinterpret_cast<uint2 *>(smem + 4 * KC * 128);
    for (uint32_t b = threadIdx.x; b < 256; b += 128)
        lut[b] = make_uint2(((b & 0xFu) * 0x00204081u) & 0x01010101u, ((b >> 4) * 0x00204081u) & 0x01010101u);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int jj0 = wave * 64;
    const uint4 item = items[blockIdx.x];
    const uint32_t tile_id = item.x;
    const uint4 tile = tiles[tile_id];
    const uint64_t lrow = (uint64_t)(wave * 2 + (lane >> 5));
    const uint32_t *gA = planes + lrow * Npad + (uint64_t)tile.x * kTile + (lane & 31) * 4;
    const uint32_t *gB = planes + lrow * Npad + (uint64_t)tile.y * kTile + (lane & 31) * 4;
    const uint64_t pass_stride = (uint64_t)4 * Npad;
    const uint32_t lds_base = (uint32_t)(uintptr_t)(lds_ptr_t)smem + wave * 1024;
    auto stage = [&](uint32_t chunk, int buf) {
        const uint32_t la = lds_base + buf * (2 * KC * 512);
        const uint32_t lb = la + KC * 512;
        const uint32_t *a = gA + (uint64_t)chunk * KC * Npad;
        const uint32_t *b = gB + (uint64_t)chunk * KC * Npad;
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
            glds16(a + ps * pass_stride, la + ps * 2048);
            glds16(b + ps * pass_stride, lb + ps * 2048);
        }
    };
    v16i_t acc[4][2];
#pragma unroll
    for (int x = 0; x < 4; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[x][y][e] = 0;
    const uint32_t ch_begin = item.y, ch_end = item.z;
    if (ch_begin >= ch_end) return;
    CT *cum_tile = cum + (uint64_t)tile_id * (kTile * kTile);
    stage(ch_begin, ch_begin & 1);
    for (uint32_t ch = ch_begin; ch < ch_end; ++ch) {
        dma_wait();
        __syncthreads();
        if (ch + 1 < ch_end) stage(ch + 1, (ch + 1) & 1);
        const uint32_t *As = smem + (ch & 1) * (2 * KC * 128) + r;
        const uint32_t *Bs = smem + (ch & 1) * (2 * KC * 128) + KC * 128 + jj0 + r;
        // software pipeline inside a chunk: the 8 MFMAs of plane word kk are issued first, then -- while the matrix
        // pipe works on them (8 x 32 cycles) -- this wave reads the words of kk+1 from LDS and expands them through the
        // table, so neither LDS round trip sits between two groups of MFMAs (un-pipelined: matrix pipe 50 % busy)
        const uint32_t sh = 16u * (uint32_t)h;
        auto expand = [lut, sh](uint32_t w) -> v4i_t {
            const uint32_t w16 = w >> sh;
            const uint2 lo = lut[w16 & 0xFFu], hi = lut[(w16 >> 8) & 0xFFu];
            v4i_t v;
            v.x = (int)lo.x;
            v.y = (int)lo.y;
            v.z = (int)hi.x;
            v.w = (int)hi.y;
            return v;
        };
        v4i_t fa[4], fb[2];
        fb[0] = expand(Bs[0]);
        fb[1] = expand(Bs[32]);
#pragma unroll
        for (int x = 0; x < 4; ++x) fa[x] = expand(As[32 * x]);
#pragma unroll 1
        for (uint32_t kk = 0; kk < (uint32_t)KC; ++kk) {
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                acc[x][0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[x], fb[0], acc[x][0], 0, 0, 0);
                acc[x][1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[x], fb[1], acc[x][1], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (kk + 1 < (uint32_t)KC) {  // (the first word of the next chunk is fetched after its barrier, un-overlapped)
                const uint32_t kn = kk + 1;
                v4i_t na[4], nb[2];
                nb[0] = expand(Bs[kn * 128]);
                nb[1] = expand(Bs[kn * 128 + 32]);
#pragma unroll
                for (int x = 0; x < 4; ++x) na[x] = expand(As[kn * 128 + 32 * x]);
                fb[0] = nb[0];
                fb[1] = nb[1];
#pragma unroll
                for (int x = 0; x < 4; ++x) fa[x] = na[x];
            }
            __builtin_amdgcn_sched_barrier(0);
            const uint32_t row_end = ch * KC + kk + 1;
            if ((row_end & (W - 1)) == 0) {  // plane boundary (uniform)
                const uint32_t pl = row_end / W - 1;
                if (pl < P) {
                    // per-lane base made opaque here so that the 128 store addresses are formed inside this rare block
                    // (hoisted out of the k loop they cost >100 VGPRs and spill the accumulators)
                    uint64_t base = (uint64_t)(uintptr_t)(cum_tile + (uint64_t)pl * nslots) + (uint64_t)((4 * h) * kTile + jj0 + r) * sizeof(CT);
                    asm volatile("" : "+v"(base));
                    CT *dst = reinterpret_cast<CT *>((uintptr_t)base);
#pragma unroll
                    for (int x = 0; x < 4; ++x)
#pragma unroll
                        for (int y = 0; y < 2; ++y)
#pragma unroll
                            for (int e = 0; e < 16; ++e) {  // C/D map of the 32x32 shapes: col = lane & 31, row = (e & 3) + 8 (e >> 2) + 4 (lane >> 5)
                                const int row = 32 * x + (e & 3) + 8 * (e >> 2);  // (+ 4h and the column are in `dst`)
                                dst[row * kTile + 32 * y] = (CT)(uint32_t)acc[x][y][e];
                            }
                }
#pragma unroll
                for (int x = 0; x < 4; ++x)
#pragma unroll
                    for (int y = 0; y < 2; ++y)
#pragma unroll
                        for (int e = 0; e < 16; ++e) acc[x][y][e] = 0;
            }
        }
    }
}

template <int KC, typename CT>
static hipError_t launch_pcm(hipStream_t st, const uint32_t *planes, uint32_t Npad, uint32_t Kpad, uint32_t W,
                             uint32_t P, const uint4 *tiles, const uint4 *items, uint32_t nitems, void *cum,
                             uint64_t nslots)
{
    const size_t lds = (size_t)KC * 2048 + 2048;  // staging + the byte -> 8 bytes table
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_pair_counts_mfma<KC, CT>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((k_pair_counts_mfma<KC, CT>), dim3(nitems), dim3(128), lds, st, planes, Npad, Kpad, W, P, tiles,
                       items, reinterpret_cast<CT *>(cum), nslots);
    return hipGetLastError();
}

#endif  // DSH_WHATIF_MFMA

hipError_t launch_pair_counts_mfma(hipStream_t st, int kc, int cum_bytes, const uint32_t *planes, uint32_t Npad,
                                   uint32_t Kpad, uint32_t W, uint32_t P, const uint4 *tiles, const uint4 *items,
                                   uint32_t nitems, void *cum, uint64_t nslots)
{
#ifndef DSH_WHATIF_MFMA
    // the product library is built without the what-if (csrc/Makefile: `make WHATIF=1` adds it)
    (void)st, (void)kc, (void)cum_bytes, (void)planes, (void)Npad, (void)Kpad, (void)W, (void)P, (void)tiles, (void)items;
    (void)nitems, (void)cum, (void)nslots;
    return hipErrorNotSupported;
#else
    if (nitems == 0 || Kpad == 0) return hipSuccess;
    if (cum_bytes == 2) {
        switch (kc) {
        case 16: return launch_pcm<16, uint16_t>(st, planes, Npad, Kpad, W, P, tiles, items, nitems, cum, nslots);
        case 32: return launch_pcm<32, uint16_t>(st, planes, Npad, Kpad, W, P, tiles, items, nitems, cum, nslots);
        case 64: return launch_pcm<64, uint16_t>(st, planes, Npad, Kpad, W, P, tiles, items, nitems, cum, nslots);
        default: return hipErrorInvalidValue;
        }
    }
    switch (kc) {
    case 16: return launch_pcm<16, uint32_t>(st, planes, Npad, Kpad, W, P, tiles, items, nitems, cum, nslots);
    case 32: return launch_pcm<32, uint32_t>(st, planes, Npad, Kpad, W, P, tiles, items, nitems, cum, nslots);
    case 64: return launch_pcm<64, uint32_t>(st, planes, Npad, Kpad, W, P, tiles, items, nitems, cum, nslots);
    default: return hipErrorInvalidValue;
    }
#endif
}

// ------------------------------------------------------------------------------------------
// finalize: one lane per pair slot of a band of tiles.  128 threads per block = one tile row
// (consecutive lanes = consecutive j: coalesced cum reads and output writes).
struct FinalizeArgs {
    const void *cum;
    uint64_t nslots;       // distance between two planes of cum (pair slots of the band)
    const uint4 *tiles;    // {row block, col block, plane begin, plane end} per tile
    const uint32_t *perm;  // plane-matrix column -> sketch index (nullptr: identity)
    // row-sorted parts (plan.h): the output buffer holds the wanted rows in KEY order, the row at layout position s
    // starting at rowoff[s]; nullptr: the rows' span of the packed triangle in final order
    const uint64_t *rowoff;
    int hist_bins;  // histogram columns allocated per lane (>= the value span of any tile of the launch)
    int pbase;  // plane pl is the threshold v = pbase + 1 + pl
    int p;
    int estim;
    int result_type;
    double ksinv;
    // per COLUMN of the layout, in layout order (k_build_colindex): nothing below waits for the permutation
    const uint32_t *nS;       // [Npad] listed registers of the column's sketch
    const uint32_t *keyS;     // [Npad] hi << 18 | T << 12 | L << 6 | lo
    const double *cardS;      // [Npad]
    const uint8_t *thS;       // [Npad][64]: how many listed registers have each value above the sketch's T
    const uint32_t *rl;       // [Npad][E]: the listed registers, position << 8 | value
    uint32_t E;
    const uint32_t *cidx_rec; // position index of the column blocks: [blocks][nbuckets][RK + 1] records
    const uint32_t *cidx_ent; // [blocks][ent_stride] all entries in bucket order (buckets with more than RK entries)
    uint32_t nbuckets, ent_stride;
    uint64_t n;      // sketches in the collection = dimension of the output matrix
    uint64_t ncols;  // real columns of the plane matrix (a sub-collection when only a row range is wanted)
    // triangle mode: rows [row_begin,row_end) (original indices), out index = tri(i,j) - base_index
    // rect mode (rect != 0): i in [row_begin,row_end) x j in [col_begin,col_end), row-major
    int rect;
    // sorted_out != 0 (triangle mode only): rows and the output index are in plane-column
    // (sorted) order instead of original sketch order -- used for multi-GPU shards, whose spans
    // are gathered first and un-permuted once (k_unpermute)
    int sorted_out;
    // square != 0 (triangle tiles, all rows): every pair is written at BOTH out[i*n+j] and
    // out[j*n+i] of an n x n matrix (the all-vs-all nearest-neighbour path: each pair computed once)
    int square;
    // knn != 0 (whole key-ordered layout, rows = plane columns [row_begin,row_end) like sorted_out): the pair (si, sj),
    // si < sj, is written twice -- out[(si - row_begin) * knn_ld + sj] is a candidate of row si, out2[sj * knn_rows +
    // (si - row_begin)] a candidate of row sj -- for the band-wise nearest-neighbour selection (k_topk_merge)
    int knn;
    float *out2;
    uint64_t knn_ld, knn_rows;
    int stop;             // profiling only (option "finalize_stop"): leave after phase 1..4 with a dummy store
    uint32_t ntiles;      // tiles of this launch
    unsigned long long *phase_cyc;  // profiling only (TIMED instances): shader-clock cycles per phase, summed over waves
    uint64_t row_begin, row_end, col_begin, col_end;
    uint64_t base_index;
    float *out;
    // part signalling (k_finalize_signal; kernels.h kSig*): counters and flags of the call's parts; the generation value
    // that marks a part final and the stamp switch live in the block itself (two kernel arguments fewer)
    uint32_t *sig;
};

// The two sparse tails of a pair's histogram without walking lists.  The block's 128 lanes share sketch i (one tile
// row); the tile has the dense planes v in (Lp, T].
//   upper tail (x > T): a lane's bins start as i's tail histogram + sketch j's (tailhist, one byte per value) -- that
//     counts a position listed by BOTH sketches twice; at each such position the smaller value is taken out again.
//   lower tail (x < Lp): max(a_t, b_t) = x < Lp needs both registers below Lp <= min(L_i, L_j), i.e. the position in
//     both low lists: the bins below Lp are exactly the positions both sketches list, at the larger of the two values
//     (if Lp comes from the value minima instead, no register pair lies below it and the join finds nothing).
// The shared positions are found through the column block's position index (k_build_colindex): lane e takes the row
// sketch's e-th listed register and reads the RECORD of its position's bucket (one gather: count + the first RK
// entries) -- |list_i| lookups per tile row instead of 128 list walks -- and applies what it finds to the owning lane's
// histogram column with LDS atomics.  Exact and order-independent.  C(Lp) = number of low joins, so c[Lp] = C(Lp+1) -
// C(Lp) and c[T] = m - |union above T| - C(T).
// Loads (round 4, profiles/r4b: the join and the prologue were 54 % of a wave's life at p = 10, all of it dependent
// global loads): every input of a block is addressed by the TILE alone -- the per-column side arrays are in layout
// order -- so the row list, C(v), tail histograms, keys and cardinalities are requested together at the top, the bucket
// records as soon as the row list is there, and they travel while the histogram columns are built.
// TIMED (option "finalize_timing", profiling only): s_memtime stamps between the phases of the FULL kernel, summed per
// phase over the waves of every 61st block (a.phase_cyc[0..5] cycles, [6] waves, [7..13] the estimator's trip counts per
// lane and per wave: what divergence costs; layout in include/dashing_hip.h at dsh_finalize_phase_cycles) -- the
// accounting VERDICT r3 asked for instead of early-exit stops, whose occupancy and overlap differ from the real kernel.
// GENERAL = false is the instance of the plain triangle (rows in original order: full matrix, row ranges, row-sorted
// parts): it never reads the arguments of the rectangle / square / sorted-output / nearest-neighbour forms, which
// relieves the scalar registers (~70 dwords of arguments: the compiler parked the overflow in VGPR lanes -- 67
// v_writelane + 105 v_readlane VALU instructions in the p <= 12 instance, profiles/r4f).
template <typename CT, int RK, bool TIMED, bool GENERAL>
__device__ __forceinline__ void finalize_block(const FinalizeArgs &a)
{
    unsigned long long tph[7] = {0, 0, 0, 0, 0, 0, 0};
    if constexpr (TIMED) tph[0] = __builtin_readcyclecounter();
    // histogram columns hold counts <= 2^p: CT (uint16 when p <= 15) halves the LDS footprint and
    // doubles the resident waves of this latency-sensitive kernel
    extern __shared__ __attribute__((aligned(16))) unsigned char hs_raw[];
    uint32_t *histA = reinterpret_cast<uint32_t *>(hs_raw);  // [64] row sketch's tail histogram above T
    uint32_t *corr = histA + 64;                              // [128] per column: upper-tail positions shared with the row sketch
    uint32_t *actm = corr + 128;                              // [4] lanes whose column is live (bit per lane) + naLive
    CT *hs = reinterpret_cast<CT *>(hs_raw + (64 + 128 + 8) * 4);
    constexpr uint32_t RW = (uint32_t)RK + 1u;  // words of a bucket record
    constexpr uint32_t kNone = 0xFFFFFFFFu;
    const int tid = threadIdx.x;
    // one block = one row of a 128 x 128 tile: everything derived from the tile is block-uniform and is
    // written so that the compiler sees it (scalar loads, SGPR compares, scalar branches)
    // k_finalize's tile descriptor (its own list, row-major per segment -- run_pairs): {row block, column block, plane
    // begin | plane end << 8 | smallest << 16 | largest << 24 register value of the two blocks' sketches, index of the
    // tile's C(v) block in the band}: the histogram columns of a block only span the values its own sketches can hold
    // block -> (tile, row of the tile): consecutive blocks go to consecutive XCDs (block b runs on XCD b % 8: observed
    // dispatch behaviour, speed only), so block b takes row (b / 8) % 128 of tile (b / 1024) * 8 + b % 8: the 128 rows of a
    // tile run on ONE XCD and its column block's position index, tail histograms and keys are fetched into that XCD's L2
    // once instead of into all eight.
    const uint32_t kq = blockIdx.x >> 3;
    const uint32_t tidx = ((kq >> 7) << 3) + (blockIdx.x & 7u), trow = kq & 127u;
    if (tidx >= a.ntiles) return;  // (the grid is rounded up to whole groups of 8 tiles)
    uint4 tile = a.tiles[tidx];
    const int vlo = (int)((tile.z >> 16) & 0xFFu), vhi = (int)(tile.z >> 24);
    // pair slot in the band's C(v): the tile's block (tile.w), this block's row of it, this lane's column.  (32-bit: a
    // band holds at most 2^16 tiles, plan.cpp.)
    const uint32_t slot = ((tile.w & 0xFFFFu) * 128u + trow) * 128u + (uint32_t)tid;  // (bits 16.. of w: the tile's part)
    tile.w = (tile.z >> 8) & 0xFFu;
    tile.z &= 0xFFu;
    // (sketch indices and layout positions are 32-bit -- the permutation is -- only the output index is wider)
    const uint32_t si = tile.x * kTile + trow;
    const uint32_t sj = tile.y * kTile + (uint32_t)tid;
    if (si >= a.ncols) return;  // padding row (uniform)
    // this tile's own plane range: dense C(v) for v in (Lp, T]
    const int Lp = a.pbase + (int)tile.z, T = a.pbase + (int)tile.w;
    // ---- requests that only need the tile, oldest first.  (1) the row sketch's listed registers: lane e takes entry e
    // (+128 per round); the bucket records below depend on them
    constexpr int kR = RK > 3 ? 2 : (int)(kListCap / 128);  // (the wide records are only used with lists of <= 256 entries)
    // (every row of rl holds E slots, the unused ones 0xFFFFFFFF: no look at the list's length first -- one dependent
    // scalar load less in front of the records)
    const int nr = (int)((a.E + 127u) >> 7);  // rounds (uniform): 1 at p <= 11, 4 at p = 14
    uint32_t le[kR];
    {
        const uint32_t *rli = a.rl + (uint64_t)si * a.E;
#pragma unroll
        for (int r = 0; r < kR; ++r) {
            le[r] = kNone;
            if (r >= nr) continue;
            const uint32_t e = (uint32_t)tid + 128u * (uint32_t)r;
            if (e < a.E) le[r] = rli[e];
        }
    }
    // (2) the sketch indices: only the output index and the row/column filters need them
    const uint32_t i = a.perm ? (uint32_t)__builtin_amdgcn_readfirstlane((int)a.perm[si]) : si;
    const bool col_ok = sj < a.ncols;
    const uint32_t j = col_ok ? (a.perm ? a.perm[sj] : sj) : 0u;
    // (3) this lane's inputs: the C(v) of up to 16 planes, 48 bins of column j's tail histogram, the keys -- one round
    // trip for all of them.  (Uniform base + 32-bit lane offset: the plane stride is added on the scalar side.)
    const CT *cum0 = reinterpret_cast<const CT *>(a.cum);
    constexpr int kBatch = 16;
    const int npl = (int)(tile.w - tile.z);  // dense planes of this tile (uniform)
    uint32_t cvv[kBatch];
    const int w0 = (T + 1) >> 4;  // first 16-byte word of the tail histogram that holds a bin > T (uniform)
    uint4 tq[3];
    {
        // one running scalar base (two SGPRs, advanced by the plane stride between the loads) -- sixteen precomputed
        // bases do not fit the scalar registers and were parked in VGPR lanes
        const CT *cp = cum0 + (uint64_t)tile.z * a.nslots;
#pragma unroll
        for (int t = 0; t < kBatch; ++t) {
            cvv[t] = 0u;
            if (t < npl) {  // uniform (a scalar compare with an immediate: nothing to keep for the second loop below)
                cvv[t] = (uint32_t)cp[slot];
                cp += a.nslots;
                asm volatile("" : "+s"(cp));  // (keep the chain: one base at a time)
            }
        }
    }
    {
        const uint4 *tb = reinterpret_cast<const uint4 *>(a.thS) + (sj << 2);  // (padding columns hold zeros)
#pragma unroll
        for (int k = 0; k < 3; ++k) tq[k] = w0 + k < 4 ? tb[w0 + k] : make_uint4(0, 0, 0, 0);
    }
    const uint32_t keyj = a.keyS[sj];
    const uint32_t keyi = (uint32_t)__builtin_amdgcn_readfirstlane((int)a.keyS[si]);
    const uint32_t hrow = tid < 64 ? (uint32_t)(a.thS + (uint64_t)si * 64)[tid] : 0u;
    const uint32_t sh = a.p > 14 ? (uint32_t)(a.p - 14) : 0u;
    const uint32_t *brec = a.cidx_rec + (uint64_t)tile.y * a.nbuckets * RW;
    const uint32_t *bent = a.cidx_ent + (uint64_t)tile.y * a.ent_stride;
    // block-level skip (uniform) when the row sketch cannot be wanted
    const uint32_t rb32 = (uint32_t)a.row_begin, re32 = (uint32_t)a.row_end;  // (<= n < 2^32)
    const bool m_rect = GENERAL && a.rect, m_square = GENERAL && a.square, m_sorted = GENERAL && a.sorted_out, m_knn = GENERAL && a.knn;
    if (m_rect && !(i >= rb32 && i < re32)) return;
    corr[tid] = 0;
    if (tid < 64) histA[tid] = tid > T ? hrow : 0u;  // the row sketch's tail histogram above this tile's threshold
    {
        // its sum = the row sketch's listed registers above T: counted from the list itself, a ballot per round (the
        // shuffle reduction over the 64 bins cost wave 0 ~30 instructions); the block's two waves leave their shares in
        // actm[4] and actm[5]
        uint32_t nup = 0;
#pragma unroll
        for (int r = 0; r < kR; ++r) {
            if (r >= nr) break;
            nup += (uint32_t)__popcll(__ballot(le[r] != kNone && (int)(le[r] & 0xFFu) > T));
        }
        if ((tid & 63) == 0) actm[4 + (tid >> 6)] = nup;
    }
    uint32_t oi = i, oj = j;
    bool active;
    if (m_rect) {
        active = j >= (uint32_t)a.col_begin && j < (uint32_t)a.col_end;
    } else if (m_square) {
        active = si < sj;
    } else if (m_sorted || m_knn) {
        oi = si;
        oj = sj;
        active = si < sj && si >= rb32 && si < re32;
    } else {
        oi = i < j ? i : j;
        oj = i < j ? j : i;
        active = si < sj && oi >= rb32 && oi < re32;
    }
    active = active && col_ok;
    const unsigned long long bal = __ballot(active);
    if ((tid & 63) == 0) {
        actm[(tid >> 6) * 2] = (uint32_t)bal;
        actm[(tid >> 6) * 2 + 1] = (uint32_t)(bal >> 32);
    }
    // where the pair's value goes: computed by whoever stores (right before the estimator; the profiling stops) -- not
    // here, where it would hold two VGPRs through the histogram phases
    auto out_index = [&](const FinalizeArgs *L) -> uint64_t {
        return m_rect  ? (uint64_t)(i - (uint32_t)L->row_begin) * (L->col_end - L->col_begin) + (j - (uint32_t)L->col_begin)
               : m_knn ? (uint64_t)(si - (uint32_t)L->row_begin) * L->knn_ld + sj
               : L->rowoff ? L->rowoff[i < j ? si : sj] + (oj - oi - 1)  // (the pair's row is the smaller sketch index)
                           : (uint64_t)oi * (2 * L->n - oi - 1) / 2 + oj - (oi + 1) - L->base_index;
    };
    if (a.stop == 1) {
        const FinalizeArgs *L = &a;
        if (active) L->out[out_index(L)] = (float)T;
        return;
    }
    const uint32_t m = 1u << a.p;
    CT *col = hs + tid;
    uint32_t prev = 0, nb = 0;
    if constexpr (TIMED) tph[1] = __builtin_readcyclecounter();
    __syncthreads();  // histA, corr, actm are set
    if (active) {
        // bins below the dense range start empty (the lower-tail join fills them); dense part: c[x] = C(x+1) - C(x),
        // x in [Lp, T), written as if C(Lp) were 0 -- corrected after the join
        for (int x = vlo; x < Lp; ++x) col[(x - vlo) * 128] = 0;
        CT *dcol = col + (a.pbase - vlo) * 128;  // bin pbase + pl lives at dcol + pl * 128
        CT *dcz = dcol + tile.z * 128;
#pragma unroll
        for (int t = 0; t < kBatch; ++t) {
            if (t < npl) {  // uniform
                dcz[t * 128] = (CT)(cvv[t] - prev);
                prev = cvv[t];
            }
        }
        for (uint32_t pl = tile.z + kBatch; pl < tile.w; ++pl) {  // wide plane ranges (heterogeneous tiles)
            const uint32_t cv = (cum0 + (uint64_t)pl * a.nslots)[slot];
            dcol[pl * 128] = (CT)(cv - prev);
            prev = cv;
        }
    }
    // (4) the bucket records of the row entries (the list arrived long ago): one gather each, in flight while the tail
    // bins are written and the block meets at the barrier.  Issued here, not at the top, for the registers: the C(v)
    // batch is dead by now (64 VGPRs = 8 waves per SIMD).
    uint4 rc[kR], rc2[RK > 3 ? kR : 1];
#pragma unroll
    for (int r = 0; r < kR; ++r) {
        rc[r] = make_uint4(0u, kNone, kNone, kNone);
        if constexpr (RK > 3) rc2[r] = make_uint4(kNone, kNone, kNone, kNone);
        if (r >= nr) continue;
        const int va = (int)(le[r] & 0xFFu);
        const bool up = va > T;
        if (le[r] == kNone || !(up || va < Lp)) continue;  // (T itself and the dense range are never joined)
        const uint32_t bk = (((le[r] >> 8) >> sh) << 1) | (up ? 0u : 1u);
        const uint4 *rp = reinterpret_cast<const uint4 *>(brec + (uint64_t)bk * RW);
        rc[r] = rp[0];
        if constexpr (RK > 3) rc2[r] = rp[1];
    }
    if (active) {
        // tail bins: histogram of i's listed values + histogram of j's listed values (both > T)
        const uint32_t qw[12] = {tq[0].x, tq[0].y, tq[0].z, tq[0].w, tq[1].x, tq[1].y, tq[1].z, tq[1].w,
                                 tq[2].x, tq[2].y, tq[2].z, tq[2].w};
        const int x0 = w0 * 16;
        CT *dst = col + (x0 - vlo) * 128;         // bin x0 + s lives at dst + s * 128 (x0 may be below vlo: never touched there)
        const uint32_t *hA = histA + x0;
#pragma unroll
        for (int s_ = 0; s_ < 48; ++s_) {
            const int x = x0 + s_;
            if (x > vhi) break;  // uniform
            if (x <= T) continue;
            const uint32_t qj = (qw[s_ >> 2] >> (8 * (s_ & 3))) & 0xFFu;
            nb += qj;
            dst[s_ * 128] = (CT)(hA[s_] + qj);
        }
        for (int x = x0 + 48; x <= vhi; ++x) {  // more than 48 bins above T: only with a very small emax
            const uint32_t qj = a.thS[(uint64_t)sj * 64 + x];
            nb += qj;
            col[(x - vlo) * 128] = (CT)(histA[x] + qj);
        }
    }
    __syncthreads();
    if constexpr (TIMED) tph[2] = __builtin_readcyclecounter();
    if (a.stop == 2) {
        const FinalizeArgs *L = &a;
        if (active) L->out[out_index(L)] = (float)(nb + prev + keyj);
        return;
    }
    // ---- the join: what the records hold is applied to the owning lanes' histogram columns
    {
        uint32_t *hw = reinterpret_cast<uint32_t *>(hs);
        // every lane of the block owns a pair (any tile off the diagonal of a full range): no per-entry ownership test
        const bool all_own = (actm[0] & actm[1] & actm[2] & actm[3]) == 0xFFFFFFFFu;
        auto apply = [&](uint32_t x, int var, bool upr, uint32_t plow) {
            const uint32_t jl = (x >> 6) & 127u;
            const int vb = (int)(x & 63u);
            // same position (groups of positions share a bucket only for p > 14), a live value on the column's side too,
            // and a lane that owns a pair (diagonal tile, row range, padding)
            if (x == kNone || (sh && (x >> 13) != plow) || !(upr ? vb > T : vb < Lp)) return;
            if (!all_own && !((actm[jl >> 5] >> (jl & 31u)) & 1u)) return;
            // upper tail: the position was counted in both tail histograms, take the smaller value out again;
            // lower tail: both registers are below the dense range, max(a_t, b_t) is the larger one
            const int bin = upr ? (var < vb ? var : vb) : (var > vb ? var : vb);
            const uint32_t cell = (uint32_t)(bin - vlo) * 128u + jl;
            const uint32_t one = sizeof(CT) == 2 ? 1u << (16u * (cell & 1u)) : 1u;
            uint32_t *w = &hw[sizeof(CT) == 2 ? cell >> 1 : cell];
            if (upr) {
                atomicAdd(&corr[jl], 1u);
                atomicSub(w, one);  // (the 16-bit half holds >= 1: no borrow into its neighbour)
            } else {
                atomicAdd(w, one);
            }
        };
#pragma unroll
        for (int r = 0; r < kR; ++r) {
            if (r >= nr) break;
            if (__ballot(rc[r].y != kNone) == 0) continue;  // no lane of this wave found anything (a short list: wave 1)
            const int va = (int)(le[r] & 0xFFu);
            const bool up = va > T;
            const uint32_t plow = (le[r] >> 8) & ((1u << sh) - 1u);
            apply(rc[r].y, va, up, plow);
            apply(rc[r].z, va, up, plow);
            apply(rc[r].w, va, up, plow);
            if constexpr (RK > 3) {
                apply(rc2[r].x, va, up, plow);
                apply(rc2[r].y, va, up, plow);
                apply(rc2[r].z, va, up, plow);
                apply(rc2[r].w, va, up, plow);
            }
            const uint32_t cnt = rc[r].x & 0xFFFFu, q0 = rc[r].x >> 16;
            for (uint32_t q = (uint32_t)RK; q < cnt; ++q) apply(bent[q0 + q], va, up, plow);  // a crowded bucket (rare)
        }
    }
    __syncthreads();
    if constexpr (TIMED) tph[3] = __builtin_readcyclecounter();
    if (!active) return;
    const uint32_t ucnt = actm[4] + actm[5] + nb - corr[tid];  // |list_i above T| + |list_j above T| - shared positions
    uint32_t clow = 0;                                // C(Lp) = the lower-tail joins = the bins below the dense range
    for (int x = vlo; x < Lp; ++x) clow += col[(x - vlo) * 128];
    if (tile.w > tile.z) col[(Lp - vlo) * 128] -= (CT)clow;  // c[Lp] = C(Lp+1) - C(Lp)
    else prev = clow;                                         // no dense plane: C(T) = C(Lp)
    col[(T - vlo) * 128] = (CT)(m - ucnt - prev);  // c[T] = C(T+1) - C(T), C(T+1) = m - |union above T|
    if (a.stop == 3) {
        const FinalizeArgs *L = &a;
        L->out[out_index(L)] = (float)(ucnt + clow);
        return;
    }
    // scan bounds for the estimator: no bin below the larger of the two minima, none above the larger of the maxima
    const int loj = (int)(keyj & 63u), loi = (int)(keyi & 63u);
    const int minv = loi > loj ? loi : loj;
    const int maxj = (int)((keyj >> 18) & 63u), maxi = (int)((keyi >> 18) & 63u);
    int maxv = maxi > maxj ? maxi : maxj;
    if (maxv < T) maxv = T;
    auto c = [col, vlo, vhi](int v) -> uint32_t {
        return (v < vlo || v > vhi) ? 0u : col[(v - vlo) * 128];
    };
    struct RawCol {  // bin v of this lane's histogram column (no bounds test), and its address
        const CT *col;
        int vlo;
        enum { stride = 128 };
        __device__ uint32_t operator()(int v) const { return col[(v - vlo) * 128]; }
        __device__ const CT *at(int v) const { return col + (v - vlo) * 128; }
    };
    const RawCol raw{col, vlo};
    // ---- what the epilogue needs is requested BEFORE the estimator (the two cardinalities travel under its ~1 000 fp64
    // instructions) and used after it.  (Reading these arguments late from the kernarg segment instead -- to spare the
    // scalar registers they occupy from the top -- was tried: fewer VALU instructions, but the scalar loads share the
    // LDS counter and every wave waited for them at the estimator's first LDS read: ORIGINAL +10 %, profiles/r4g, r4h.)
    const FinalizeArgs *L = &a;
    const double cardj = L->cardS[sj], cardi = L->cardS[si];
    const int rtype = L->result_type;
    const double ksinv = L->ksinv;
    const uint64_t oidx = out_index(L);
    float *const outp = L->out;
    if constexpr (TIMED) tph[4] = __builtin_readcyclecounter();
    int mle_it = 0;
    const double us = estimate(c, raw, a.p, a.estim, minv < T ? minv : T, maxv, TIMED ? &mle_it : nullptr);
    if constexpr (TIMED) tph[5] = __builtin_readcyclecounter();
    if (a.stop == 4) {
        outp[oidx] = (float)us;
        return;
    }
    const float res = result_cmp_from(cardj, cardi, us, rtype, ksinv);  // lhs = j, rhs = i
    if (m_square || m_knn) {  // row i sees j as lhs, row j sees i as lhs (only the containment measures differ)
        const bool asym = rtype == 4 || rtype == 5 || rtype == 6;
        const float rev = asym ? result_cmp_from(cardi, cardj, us, rtype, ksinv) : res;
        if (m_square) {
            L->out[(uint64_t)i * L->n + j] = res;
            L->out[(uint64_t)j * L->n + i] = rev;
        } else {
            outp[oidx] = res;
            L->out2[(uint64_t)sj * L->knn_rows + (si - (uint32_t)L->row_begin)] = rev;
        }
        return;
    }
    // (a signalling call: the result is written THROUGH to memory -- sc0 sc1 -- so that nothing of a part sits dirty in
    // this XCD's L2 when the part's flag goes up: the eight L2s of the chip are not coherent with each other, and a
    // release fence per block (buffer_wbl2 + buffer_inv in every wave) made the kernel 7x slower, profiles/rd5m)
    // (inline assembly: the compiler lowers a relaxed system-scope atomic store of a float to a plain store)
    if (!GENERAL && a.sig) asm volatile("global_store_dword %0, %1, off sc0 sc1" ::"v"(outp + oidx), "v"(res) : "memory");
    else outp[oidx] = res;
    if constexpr (TIMED) {
        tph[6] = __builtin_readcyclecounter();
        const unsigned long long live = __ballot(1);
        // (one block in 61 reports: with every wave adding to the same 14 counters the atomics themselves became the
        // kernel -- 30x slower, profiles/r4a -- and the phases in front of them looked 30x longer than they are)
        if (blockIdx.x % 61u != 0) return;
        if ((unsigned)(tid & 63) == (unsigned)(__ffsll((long long)live) - 1)) {  // the wave's first finishing lane
            for (int k = 0; k < 6; ++k) atomicAdd(&a.phase_cyc[k], tph[k + 1] - tph[k]);
            atomicAdd(&a.phase_cyc[6], 1ull);
        }
        // divergence of the estimator inside a wave: per lane its MLE iterations and live bins (the trip counts of the
        // outer and the inner loop); a wave pays the maxima
        corr[tid] = (uint32_t)mle_it;  // (corr is dead by now; the wave's lanes read back only their own wave's half)
        if ((unsigned)(tid & 63) == (unsigned)(__ffsll((long long)live) - 1)) {
            unsigned mx_it = 0, mx_bins = 0, sum_it = 0, sum_bins = 0, sum_work = 0, lanes = 0;
            for (int l = 0; l < 64; ++l) {
                if (!((live >> l) & 1ull)) continue;
                const uint32_t v = corr[(tid & 64) + l];
                const unsigned it = v & 255u, bins = v >> 8;
                mx_it = it > mx_it ? it : mx_it;
                mx_bins = bins > mx_bins ? bins : mx_bins;
                sum_it += it;
                sum_bins += bins;
                sum_work += it * bins;
                ++lanes;
            }
            atomicAdd(&a.phase_cyc[7], (unsigned long long)lanes);
            atomicAdd(&a.phase_cyc[8], (unsigned long long)sum_it);
            atomicAdd(&a.phase_cyc[9], (unsigned long long)sum_bins);
            atomicAdd(&a.phase_cyc[10], (unsigned long long)sum_work);
            atomicAdd(&a.phase_cyc[11], (unsigned long long)mx_it);
            atomicAdd(&a.phase_cyc[12], (unsigned long long)mx_bins);
            atomicAdd(&a.phase_cyc[13], (unsigned long long)(mx_it * mx_bins));
        }
    }
}

// 64 VGPRs (8 waves per SIMD; the compiler settles at 72 / 7 on its own): -7 % on C3, -3 % at p = 10 (profiles/r3f)
template <typename CT, int RK, bool TIMED, bool GENERAL>
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(8, 8), amdgpu_num_sgpr(96))) void k_finalize(FinalizeArgs a)
{
    finalize_block<CT, RK, TIMED, GENERAL>(a);
}

// The instance of the plain triangle (rows in original order: full matrix, row ranges, row-sorted parts), and of the
// pipelined exchange (round 5): ONE launch finalizes a whole band of the tile kernel and the parts announce themselves
// from inside it (a.sig != nullptr).  (With the signalling code behind the body the compiler parks 15 scalar values in
// VGPR lanes and reads 26 back in the p <= 12 instance, against 27 / 81 in the plain one -- and is SLOWER all the same
// when it serves calls without parts: configs[3] shape 215.5-216.7 ms against 211.3-211.6, profiles/rd5p; so those keep
// k_finalize<.., false, false>.)  One launch per part (an event behind each) cost a source rank of BASELINE
// configs[2] over 8 ranks 0.49-0.55 ms of k_finalize against 0.43 for one launch -- every launch of ~70 tiles is 2.2
// waves of blocks with a tail -- and its first part was final 0.3 ms after the tile kernel instead of 0.08.  Here every
// block, when it is through (its results written through to memory, barrier), counts itself into its tile; a tile's 128th row counts the tile into its
// part (bits 16.. of the tile descriptor's w); the part's last tile writes the call's generation value into the part's
// flag, which the copy stream waits for with hipStreamWaitValue32 (>= generation) in front of the part's transfer
// (tools/ubench/wait_value.hip: the gate opens 2-3 us after the write).  Two atomics per tile on distinct addresses,
// one per part: nothing the kernel notices.
template <typename CT, int RK>
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(8, 8), amdgpu_num_sgpr(96))) void k_finalize_signal(FinalizeArgs a)
{
    finalize_block<CT, RK, false, false>(a);
    if (a.sig == nullptr) return;  // (a call without parts: nothing to announce)
    const uint32_t tidx = (((blockIdx.x >> 3) >> 7) << 3) + (blockIdx.x & 7u);
    if (tidx >= a.ntiles) return;  // (the grid is rounded up to whole groups of 8 tiles: no tile, nothing to count)
    // every WAVE waits for its own write-through stores (gfx9 counts stores in vmcnt; a workgroup-scope release fence
    // compiles to lgkmcnt(0) only, which would let wave 1's stores still be in flight when wave 0 counts the block:
    // ADVICE r5) ...
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();  // ... so behind the barrier the stores of every lane of the block have been acknowledged
    if (threadIdx.x == 0) {
        uint32_t *sig = a.sig;
        if (atomicAdd(sig + kSigTileCnt + tidx, 1u) == kTile - 1) {  // the tile's last row
            atomicExch(sig + kSigTileCnt + tidx, 0u);  // (ready for the next launch: no clearing between the bands of a call)
            const uint32_t q = a.tiles[tidx].w >> 16;
            if (atomicAdd(sig + kSigPartCnt + q, 1u) + 1u == sig[kSigPartTotal + q]) {  // the part's last tile
                if (sig[kSigStamp]) reinterpret_cast<unsigned long long *>(sig + kSigPartTime)[q] = wall_clock64();
                __hip_atomic_store(sig + kSigPartFlag + q, sig[kSigGen], __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
}

__global__ void k_wall_stamp(unsigned long long *out) { *out = wall_clock64(); }

hipError_t launch_wall_stamp(hipStream_t st, unsigned long long *out)
{
    hipLaunchKernelGGL(k_wall_stamp, dim3(1), dim3(1), 0, st, out);
    return hipGetLastError();
}

// row-sorted parts (plan.h): the rows at the positions [pos0, pos1) of a source rank's key order, lying at rowoff[s] of its
// buffer, go to their places in dashing's packed triangle (row r starts at r (2n - r - 1) / 2 and holds n - 1 - r values).
// A row is contiguous on both sides; a block copies kPlaceChunk values of it (blockIdx.y = the chunk: the longest rows
// of a part -- 40 KB at n = 10 000 -- would otherwise keep 128 blocks busy while 128 CUs idle).
constexpr uint32_t kPlaceChunk = 4096;
__device__ __forceinline__ void place_row_chunk(const float *__restrict__ src, float *__restrict__ out, const uint32_t *__restrict__ order,
                                                const uint64_t *__restrict__ rowoff, uint64_t s, uint64_t n)
{
    const uint64_t r = order[s], len = n - 1 - r;
    const uint64_t x0 = (uint64_t)blockIdx.y * kPlaceChunk;
    if (x0 >= len) return;
    const uint64_t x1 = x0 + kPlaceChunk < len ? x0 + kPlaceChunk : len;
    const float *from = src + rowoff[s];
    float *to = out + r * (2 * n - r - 1) / 2;
    for (uint64_t x = x0 + threadIdx.x; x < x1; x += 256) to[x] = from[x];
}

__global__ __launch_bounds__(256) void k_row_place(const float *__restrict__ src, float *__restrict__ out,
                                                    const uint32_t *__restrict__ order, const uint64_t *__restrict__ rowoff,
                                                    uint64_t pos0, uint64_t n)
{
    place_row_chunk(src, out, order, rowoff, pos0 + blockIdx.x, n);
}

hipError_t launch_row_place(hipStream_t st, const float *src, float *out, const uint32_t *order, const uint64_t *rowoff,
                            uint64_t pos0, uint64_t pos1, uint64_t n)
{
    if (pos1 <= pos0 || n < 2) return hipSuccess;
    hipLaunchKernelGGL(k_row_place, dim3((uint32_t)(pos1 - pos0), (uint32_t)((n - 1 + kPlaceChunk - 1) / kPlaceChunk)), dim3(256), 0, st,
                       src, out, order, rowoff, pos0, n);
    return hipGetLastError();
}

// one round of the exchange at the destination: the parts of ALL row-sorted sources that arrived in it, in ONE launch
// (a launch per source left seven small kernels in a row behind the last transfer of an 8-rank step).  Entry e covers
// the blocks [row0, row0 + nrows) of the grid's x.
__global__ __launch_bounds__(256) void k_rows_place(const PlaceEnt *__restrict__ ent, uint32_t nent, float *__restrict__ out, uint64_t n)
{
    uint32_t e = 0;
    while (e + 1 < nent && ent[e + 1].row0 <= blockIdx.x) ++e;
    const PlaceEnt E = ent[e];
    place_row_chunk(E.src, out, E.order, E.rowoff, E.pos0 + (blockIdx.x - E.row0), n);
}

hipError_t launch_rows_place(hipStream_t st, const PlaceEnt *ent, uint32_t nent, uint32_t total_rows, float *out, uint64_t n)
{
    if (!nent || !total_rows || n < 2) return hipSuccess;
    hipLaunchKernelGGL(k_rows_place, dim3(total_rows, (uint32_t)((n - 1 + kPlaceChunk - 1) / kPlaceChunk)), dim3(256), 0, st, ent, nent, out, n);
    return hipGetLastError();
}


// same mapping driven from the destination: one block per ORIGINAL row a, coalesced writes of
// row a of the output, gathered reads (inv = inverse of perm)
__global__ __launch_bounds__(256) void k_unpermute_gather(const float *__restrict__ in,
                                                           const uint32_t *__restrict__ inv, uint64_t n,
                                                           float *__restrict__ out)
{
    const uint64_t a = blockIdx.x;
    const uint64_t sa = inv[a];
    float *row = out + a * (2 * n - a - 1) / 2 - (a + 1);
    for (uint64_t b = a + 1 + threadIdx.x; b < n; b += 256) {
        const uint64_t sb = inv[b];
        const uint64_t lo = sa < sb ? sa : sb, hi = sa < sb ? sb : sa;
        row[b] = in[lo * (2 * n - lo - 1) / 2 + hi - (lo + 1)];
    }
}

// as k_unpermute_gather, but the spans of the shards sit at a fixed stride in `in` (the padded blocks an
// RCCL gather delivers): rowdelta[sorted row / 128] = shard * stride - span_off[shard]
__global__ __launch_bounds__(256) void k_unpermute_staged(const float *__restrict__ in,
                                                           const uint32_t *__restrict__ inv,
                                                           const int64_t *__restrict__ rowdelta,
                                                           uint64_t n, float *__restrict__ out)
{
    const uint64_t a = blockIdx.x;
    const uint64_t sa = inv[a];
    float *row = out + a * (2 * n - a - 1) / 2 - (a + 1);
    for (uint64_t b = a + 1 + threadIdx.x; b < n; b += 256) {
        const uint64_t sb = inv[b];
        const uint64_t lo = sa < sb ? sa : sb, hi = sa < sb ? sb : sa;
        row[b] = in[(int64_t)(lo * (2 * n - lo - 1) / 2 + hi - (lo + 1)) + rowdelta[lo / kTile]];
    }
}

hipError_t launch_unpermute_staged(hipStream_t st, const float *in, const uint32_t *inv,
                                   const int64_t *rowdelta, uint64_t n, float *out)
{
    if (n < 2) return hipSuccess;
    hipLaunchKernelGGL(k_unpermute_staged, dim3((uint32_t)(n - 1)), dim3(256), 0, st, in, inv, rowdelta, n, out);
    return hipGetLastError();
}

hipError_t launch_unpermute(hipStream_t st, const float *in, const uint32_t *inv, uint64_t n, float *out)
{
    if (n < 2) return hipSuccess;
    hipLaunchKernelGGL(k_unpermute_gather, dim3((uint32_t)(n - 1)), dim3(256), 0, st, in, inv, n, out);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// k best columns of every row of a row-major [rows][ncols] block (perform_nns,
// src/sketch_and_cmp.h:642-697, without its heaps and mutexes): one wave per row, nn selection
// passes; pass t takes the best element that comes strictly after pass t-1's pick in the total
// order (value best-first, then column index ascending), so nothing is mutated and ties are
// deterministic.  NaN ranks last.  self_col (or ~0) is skipped.
__global__ __launch_bounds__(256) void k_topk(const float *__restrict__ vals, uint64_t rows,
                                               uint64_t ncols, uint64_t row0, uint64_t col0,
                                               int descending, uint32_t nn, int exclude_self,
                                               uint32_t *__restrict__ idx_out,
                                               float *__restrict__ val_out)
{
    const int lane = threadIdx.x & 63;
    const uint64_t r = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const float *v = vals + r * ncols;
    const float worst = descending ? -__builtin_huge_valf() : __builtin_huge_valf();
    const uint64_t self = exclude_self ? row0 + r - col0 : ~0ull;  // column of the row's own sketch
    // order: a before b  <=>  a.val better than b.val, or equal and a.idx < b.idx
    auto before = [descending](float av, uint32_t ai, float bv, uint32_t bi) {
        if (av != bv) return descending ? av > bv : av < bv;
        return ai < bi;
    };
    float pv = descending ? __builtin_huge_valf() : -__builtin_huge_valf();  // "before everything"
    uint32_t pi = 0;
    bool first = true;
    for (uint32_t t = 0; t < nn; ++t) {
        float bv = worst;
        uint32_t bi = 0xFFFFFFFFu;
        for (uint64_t j = lane; j < ncols; j += 64) {
            if (j == self) continue;
            float x = v[j];
            if (x != x) x = worst;
            const uint32_t ji = (uint32_t)j;
            if (!first && !before(pv, pi, x, ji)) continue;  // not after the previous pick
            if (bi == 0xFFFFFFFFu || before(x, ji, bv, bi)) {
                bv = x;
                bi = ji;
            }
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            const float ov = __shfl_xor(bv, d, 64);
            const uint32_t oi = __shfl_xor(bi, d, 64);
            if (oi != 0xFFFFFFFFu && (bi == 0xFFFFFFFFu || before(ov, oi, bv, bi))) {
                bv = ov;
                bi = oi;
            }
        }
        if (lane == 0) {
            idx_out[r * nn + t] = bi == 0xFFFFFFFFu ? 0xFFFFFFFFu : (uint32_t)(bi + col0);
            val_out[r * nn + t] = bi == 0xFFFFFFFFu ? worst : v[bi];
        }
        pv = bv;
        pi = bi;
        first = false;
        if (bi == 0xFFFFFFFFu) {  // fewer candidates than nn: fill the rest
            for (uint32_t u = t + 1; u < nn; ++u)
                if (lane == 0) {
                    idx_out[r * nn + u] = 0xFFFFFFFFu;
                    val_out[r * nn + u] = worst;
                }
            break;
        }
    }
}

// (dsh_preload) the runtime loads a translation unit's code object at the first use of one of its kernels: ask for one
hipError_t preload_compare_kernels()
{
    hipFuncAttributes fa;
    return hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(k_wall_stamp));
}

hipError_t launch_topk(hipStream_t st, const float *vals, uint64_t rows, uint64_t ncols,
                       uint64_t row0, uint64_t col0, int descending, uint32_t nn,
                       int exclude_self, uint32_t *idx_out, float *val_out)
{
    if (rows == 0 || nn == 0) return hipSuccess;
    hipLaunchKernelGGL(k_topk, dim3((uint32_t)((rows + 3) / 4)), dim3(256), 0, st, vals, rows,
                       ncols, row0, col0, descending, nn, exclude_self, idx_out, val_out);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// Nearest neighbours without the n x n matrix (perform_nns, src/sketch_and_cmp.h:642-697, whose heaps take every pair
// as it is computed): the triangle is computed in bands of tile rows of the key-ordered layout; k_finalize leaves a
// band's values twice -- V[r][t] (row = band row r, column = plane column t > b0 + r) and Vt[t][r] (row = plane column
// t, column = band row r with b0 + r < t) -- and this kernel folds each row's new candidates into the running list of
// that row's sketch: one wave per row, nn selection passes over (running list) U (candidates) in the total order of
// k_topk (value best-first, then sketch index ascending; NaN last), every candidate carrying its ORIGINAL sketch index.
//   mode 0: rows r in [0, rows) of V (leading dimension ld): sketch perm[b0 + r], candidates t in (b0 + r, ncols)
//   mode 1: rows t in (b0, ncols) of Vt (leading dimension = band rows): sketch perm[t], candidates r in [0, min(rows, t - b0))
// state: idx[n][nn] / val[n][nn] by original sketch index, initialised to (0xFFFFFFFF, worst).
__global__ __launch_bounds__(256) void k_topk_merge(const float *__restrict__ vals, uint64_t ld, int mode, uint64_t b0,
                                                     uint64_t rows, uint64_t ncols, const uint32_t *__restrict__ perm,
                                                     int descending, uint32_t nn, uint32_t *__restrict__ st_idx,
                                                     float *__restrict__ st_val)
{
    extern __shared__ __attribute__((aligned(8))) unsigned char tk_raw[];  // per wave: nn x (idx, val) of the running list
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t *ri = reinterpret_cast<uint32_t *>(tk_raw) + (size_t)wave * 2 * nn;
    float *rv = reinterpret_cast<float *>(ri + nn);
    const uint64_t w = (uint64_t)blockIdx.x * 4 + wave;
    uint64_t srow, c_begin, c_end;  // plane column of the row's sketch; candidate columns of this row
    const float *v;
    if (mode == 0) {
        if (w >= rows) return;
        srow = b0 + w;
        c_begin = srow + 1;
        c_end = ncols;
        v = vals + w * ld;
    } else {
        srow = b0 + 1 + w;
        if (srow >= ncols) return;
        c_begin = 0;
        c_end = srow - b0 < rows ? srow - b0 : rows;
        v = vals + srow * ld;
    }
    const uint32_t self = perm ? perm[srow] : (uint32_t)srow;
    const float worst = descending ? -__builtin_huge_valf() : __builtin_huge_valf();
    for (uint32_t t = lane; t < nn; t += 64) {
        ri[t] = st_idx[(uint64_t)self * nn + t];
        rv[t] = st_val[(uint64_t)self * nn + t];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    auto before = [descending](float av, uint32_t ai, float bv, uint32_t bi) {
        if (av != bv) return descending ? av > bv : av < bv;
        return ai < bi;
    };
    float pv = descending ? __builtin_huge_valf() : -__builtin_huge_valf();
    uint32_t pi = 0;
    bool first = true;
    for (uint32_t t = 0; t < nn; ++t) {
        float bv = worst, braw = worst;  // bv: the value the order uses (NaN ranks as the worst), braw: what is reported
        uint32_t bi = 0xFFFFFFFFu;
        auto offer = [&](float raw, uint32_t id) {
            const float x = raw != raw ? worst : raw;
            if (!first && !before(pv, pi, x, id)) return;  // not after the previous pick
            if (bi == 0xFFFFFFFFu || before(x, id, bv, bi)) {
                bv = x;
                braw = raw;
                bi = id;
            }
        };
        for (uint32_t u = lane; u < nn; u += 64)
            if (ri[u] != 0xFFFFFFFFu) offer(rv[u], ri[u]);
        for (uint64_t c = c_begin + lane; c < c_end; c += 64) {
            const uint64_t sc = mode == 0 ? c : b0 + c;  // plane column of the candidate
            offer(v[c], perm ? perm[sc] : (uint32_t)sc);
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            const float ov = __shfl_xor(bv, d, 64), oraw = __shfl_xor(braw, d, 64);
            const uint32_t oi = __shfl_xor(bi, d, 64);
            if (oi != 0xFFFFFFFFu && (bi == 0xFFFFFFFFu || before(ov, oi, bv, bi))) {
                bv = ov;
                braw = oraw;
                bi = oi;
            }
        }
        if (lane == 0) {
            st_idx[(uint64_t)self * nn + t] = bi;
            st_val[(uint64_t)self * nn + t] = bi == 0xFFFFFFFFu ? worst : braw;  // (a NaN stays a NaN, as k_topk reports it)
        }
        pv = bv;
        pi = bi;
        first = false;
        if (bi == 0xFFFFFFFFu) {  // fewer candidates than nn: the rest stays empty
            for (uint32_t u = t + 1 + lane; u < nn; u += 64) {
                st_idx[(uint64_t)self * nn + u] = 0xFFFFFFFFu;
                st_val[(uint64_t)self * nn + u] = worst;
            }
            break;
        }
    }
}

__global__ __launch_bounds__(256) void k_fill_knn_state(uint32_t *__restrict__ idx, float *__restrict__ val, uint64_t cnt, float worst)
{
    const uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (t < cnt) {
        idx[t] = 0xFFFFFFFFu;
        val[t] = worst;
    }
}

hipError_t launch_knn_state_init(hipStream_t st, uint32_t *idx, float *val, uint64_t cnt, int descending)
{
    if (cnt == 0) return hipSuccess;
    const float worst = descending ? -__builtin_huge_valf() : __builtin_huge_valf();
    hipLaunchKernelGGL(k_fill_knn_state, dim3((uint32_t)((cnt + 255) / 256)), dim3(256), 0, st, idx, val, cnt, worst);
    return hipGetLastError();
}

hipError_t launch_topk_merge(hipStream_t st, const float *vals, uint64_t ld, int mode, uint64_t b0, uint64_t rows,
                             uint64_t ncols, const uint32_t *perm, int descending, uint32_t nn, uint32_t *st_idx,
                             float *st_val)
{
    const uint64_t nrows = mode == 0 ? rows : (ncols > b0 + 1 ? ncols - b0 - 1 : 0);
    if (nrows == 0 || nn == 0) return hipSuccess;
    const size_t lds = (size_t)4 * 2 * nn * sizeof(uint32_t);
    hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void *>(k_topk_merge), lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_topk_merge, dim3((uint32_t)((nrows + 3) / 4)), dim3(256), lds, st, vals, ld, mode, b0, rows, ncols,
                       perm, descending, nn, st_idx, st_val);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// Small host -> device uploads (column permutation, tile / item / work lists) as a KERNEL that reads the page-locked
// staging buffer over PCIe: in order on the ctx stream like everything else.  A hipMemcpyAsync here goes through the
// runtime's copy path, and a copy queued on the ctx stream while the copy stream is moving a result to the host made
// the following kernels wait for that transfer in about half of the cases (profiles/r3d) -- the overlap of
// dsh_dist_rows_async needs the ctx stream free of runtime copies.
__global__ __launch_bounds__(256) void k_upload(uint32_t *__restrict__ dst, const uint32_t *__restrict__ src, uint64_t nwords)
{
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    for (uint64_t w = (uint64_t)blockIdx.x * 256 + threadIdx.x; w < nwords; w += stride) dst[w] = src[w];
}

// up to four uploads in ONE launch (blockIdx.y = the segment): a band's lists and, in front of the first band, the parts'
// signal block -- four launches of 4 us and as many gaps in front of a tile kernel otherwise (profiles/rd5tr)
__global__ __launch_bounds__(256) void k_upload_segs(UploadSegs u)
{
    const UploadSeg g = u.s[blockIdx.y];
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    for (uint64_t w = (uint64_t)blockIdx.x * 256 + threadIdx.x; w < g.nwords; w += stride) g.dst[w] = g.src[w];
}

hipError_t launch_upload_segs(hipStream_t st, const UploadSegs &u, uint32_t nseg)
{
    uint64_t most = 0;
    for (uint32_t i = 0; i < nseg; ++i) most = std::max(most, u.s[i].nwords);
    if (!nseg || !most) return hipSuccess;
    const uint32_t blocks = (uint32_t)std::min<uint64_t>(256, (most + 255) / 256);
    hipLaunchKernelGGL(k_upload_segs, dim3(blocks, nseg), dim3(256), 0, st, u);
    return hipGetLastError();
}

hipError_t launch_upload(hipStream_t st, void *dst, const void *src_pinned, size_t bytes)
{
    if (bytes == 0) return hipSuccess;
    const uint64_t nwords = (bytes + 3) / 4;  // (every list here is a whole number of 32-bit words)
    const uint32_t blocks = (uint32_t)std::min<uint64_t>(256, (nwords + 255) / 256);
    hipLaunchKernelGGL(k_upload, dim3(blocks), dim3(256), 0, st, (uint32_t *)dst, (const uint32_t *)src_pinned, nwords);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// launch wrappers (host)
hipError_t launch_selfhist_card(hipStream_t st, const uint8_t *regs, uint64_t first, uint64_t n, int p, int estim,
                                int emax, int elow, uint32_t *hist, void *exc, uint8_t *excv, uint32_t *exc_n,
                                uint32_t *keys, uint8_t *tailhist)
{
    if (n <= first) return hipSuccess;
    const uint32_t blocks = (uint32_t)((n - first + 3) / 4);
    if (p <= 15)
        hipLaunchKernelGGL(k_selfhist_card<uint16_t>, dim3(blocks), dim3(256), 0, st, regs, first, n, p, estim, emax, elow,
                           hist, (uint16_t *)exc, excv, exc_n, keys, tailhist);
    else
        hipLaunchKernelGGL(k_selfhist_card<uint32_t>, dim3(blocks), dim3(256), 0, st, regs, first, n, p, estim, emax, elow,
                           hist, (uint32_t *)exc, excv, exc_n, keys, tailhist);
    return hipGetLastError();
}

hipError_t launch_card_from_hist(hipStream_t st, const uint32_t *hist, const uint32_t *keys, uint64_t first, uint64_t n, int p,
                                 int estim, double *card)
{
    if (n <= first) return hipSuccess;
    hipLaunchKernelGGL(k_card_from_hist, dim3((uint32_t)((n - first + 63) / 64)), dim3(64), 0, st, hist, keys, first, n, p, estim, card);
    return hipGetLastError();
}

// Memory and LDS of the position index: the bucket records take nblocks x nbuckets x (RK + 1) x 4 bytes
// (nbuckets = 2 x min(2^p, 2^14); 16-byte records from p = 13, 32-byte below): 512 KiB per 128-column block at p >= 14, i.e.
// 0.4 GB at 100 000 columns and 4 GB at 1 000 000 -- against the 288 GB of the device, and bounded per call by the
// columns of the layout.  A workgroup counts nbuckets / G buckets in dynamic LDS: 128 KiB at p >= 14 when G = 1 (more
// than 255 column blocks), which the 160 KiB of a gfx950 CU hold; this library is gfx950-only (dsh_create refuses
// other devices), so no smaller-LDS fallback exists.
hipError_t launch_build_colindex(hipStream_t st, const ColIndexLaunch &c)
{
    if (c.nblocks == 0) return hipSuccess;
    // few column blocks (C3: 79; a rank of 8: 28-79): up to 4 workgroups per block, each a quarter of the buckets
    uint32_t G = 1;
    while (G < 4 && c.nblocks * G < 256 && c.nbuckets / (2 * G) >= 4096) G *= 2;
    const size_t lds = (size_t)(c.nbuckets / G) * sizeof(uint32_t);
#define DSH_COLINDEX(PT, RK)                                                                                                  \
    do {                                                                                                                      \
        hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void *>(k_build_colindex<PT, RK>), lds);                     \
        if (e != hipSuccess) return e;                                                                                        \
        hipLaunchKernelGGL((k_build_colindex<PT, RK>), dim3(c.nblocks, G), dim3(1024), lds, st, (const PT *)c.exc, c.excv,    \
                           c.exc_n, c.keys, c.card, c.tailhist, c.perm, c.ncols, c.p, c.nbuckets, c.ent_stride, c.E, c.rec,    \
                           c.ent, c.nS, c.keyS, c.cardS, c.thS, c.rl);                                                         \
    } while (0)
    if (c.p > 15) DSH_COLINDEX(uint32_t, 3);
    else if (colindex_inline(c.p, c.E) == 3) DSH_COLINDEX(uint16_t, 3);
    else DSH_COLINDEX(uint16_t, 7);
#undef DSH_COLINDEX
    return hipGetLastError();
}

hipError_t launch_transform(hipStream_t st, const uint8_t *regs, uint64_t n, int p, int vlo,
                            uint32_t P, uint32_t W, uint32_t Npad, uint32_t *planes,
                            const uint32_t *perm)
{
    if (P == 0) return hipSuccess;
    if (p >= 8 && ((uint64_t)1 << p) / 256 <= 65535) {  // Npad is a multiple of 128
        hipLaunchKernelGGL(k_transform_t, dim3(Npad / 64, (uint32_t)(((uint64_t)1 << p) / 256)), dim3(256), 0, st, regs,
                           n, p, vlo, P, W, Npad, planes, perm);
        return hipGetLastError();
    }
    const uint64_t threads = (uint64_t)Npad * W;
    const uint32_t blocks = (uint32_t)((threads + 255) / 256);
    hipLaunchKernelGGL(k_transform, dim3(blocks), dim3(256), 0, st, regs, n, p, vlo, P, W, Npad,
                       planes, perm);
    return hipGetLastError();
}

template <int KC, int U, typename CT>
static hipError_t launch_pc(hipStream_t st, const uint32_t *planes, uint32_t Npad, uint32_t Kpad,
                            uint32_t W, uint32_t P, const uint4 *tiles, const uint4 *items,
                            uint32_t nitems, void *cum, uint64_t nslots)
{
    const size_t lds = (size_t)KC * 2048;
    // per launch, not cached in a static: the attribute is per device and a process may own several
    hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void *>(k_pair_counts<KC, U, CT>), lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((k_pair_counts<KC, U, CT>), dim3(nitems), dim3(256), lds, st, planes, Npad,
                       Kpad, W, P, tiles, items, reinterpret_cast<CT *>(cum), nslots);
    return hipGetLastError();
}

template <int KC, typename CT>
static hipError_t launch_pc_u(hipStream_t st, const uint32_t *planes, uint32_t Npad,
                              uint32_t Kpad, uint32_t W, uint32_t P, const uint4 *tiles,
                              const uint4 *items, uint32_t nitems, void *cum, uint64_t nslots)
{
    if (W >= 8) return launch_pc<KC, 8, CT>(st, planes, Npad, Kpad, W, P, tiles, items, nitems, cum, nslots);
    if (W == 4) return launch_pc<KC, 4, CT>(st, planes, Npad, Kpad, W, P, tiles, items, nitems, cum, nslots);
    if (W == 2) return launch_pc<KC, 2, CT>(st, planes, Npad, Kpad, W, P, tiles, items, nitems, cum, nslots);
    return launch_pc<KC, 1, CT>(st, planes, Npad, Kpad, W, P, tiles, items, nitems, cum, nslots);
}

hipError_t launch_pair_counts(hipStream_t st, int kc, int cum_bytes, const uint32_t *planes,
                              uint32_t Npad, uint32_t Kpad, uint32_t W, uint32_t P,
                              const uint4 *tiles, const uint4 *items, uint32_t nitems, void *cum,
                              uint64_t nslots)
{
    if (nitems == 0 || Kpad == 0) return hipSuccess;
    if (cum_bytes == 2) {
        switch (kc) {
        case 16: return launch_pc_u<16, uint16_t>(st, planes, Npad, Kpad, W, P, tiles, items, nitems, cum, nslots);
        case 32: return launch_pc_u<32, uint16_t>(st, planes, Npad, Kpad, W, P, tiles, items, nitems, cum, nslots);
        default: return hipErrorInvalidValue;
        }
    }
    switch (kc) {
    case 16: return launch_pc_u<16, uint32_t>(st, planes, Npad, Kpad, W, P, tiles, items, nitems, cum, nslots);
    case 32: return launch_pc_u<32, uint32_t>(st, planes, Npad, Kpad, W, P, tiles, items, nitems, cum, nslots);
    default: return hipErrorInvalidValue;
    }
}

template <int KC, typename CT>
static hipError_t launch_pcl(hipStream_t st, const uint32_t *planes, uint32_t Npad, uint32_t Kpad, uint32_t W,
                             uint32_t P, const uint4 *tiles, const uint4 *items, uint32_t nitems, void *cum,
                             uint64_t nslots)
{
    const size_t lds = (size_t)KC * 4096;  // two halves, each double-buffered A|B
    hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void *>(k_pair_counts_ls<KC, CT>), lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((k_pair_counts_ls<KC, CT>), dim3((nitems + 1) / 2), dim3(512), lds, st, planes, Npad, Kpad, W,
                       P, tiles, items, nitems, reinterpret_cast<CT *>(cum), nslots);
    return hipGetLastError();
}

template <int KC, typename CT>
static hipError_t launch_pcl_frag(hipStream_t st, const uint32_t *planes, uint32_t Npad, uint32_t Kpad, uint32_t W, uint32_t P,
                                  const uint4 *tiles, const uint4 *frags, uint32_t nfrag, void *cum, uint64_t nslots)
{
    const size_t lds = (size_t)KC * 4096;
    hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void *>(k_pair_counts_ls<KC, CT, true>), lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((k_pair_counts_ls<KC, CT, true>), dim3((nfrag + 1) / 2), dim3(512), lds, st, planes, Npad, Kpad, W, P, tiles, frags,
                       nfrag, reinterpret_cast<CT *>(cum), nslots);
    return hipGetLastError();
}

hipError_t launch_pair_counts_lockstep(hipStream_t st, int kc, int cum_bytes, const uint32_t *planes, uint32_t Npad,
                                       uint32_t Kpad, uint32_t W, uint32_t P, const uint4 *tiles, const uint4 *items,
                                       uint32_t nitems, uint32_t nfrag, void *cum, uint64_t nslots)
{
    if (nitems == 0 || Kpad == 0) return hipSuccess;
    if (nfrag) {  // the band's last nfrag items are overflow fragments: the whole items first, then the fragments' launch
        if (nfrag > nitems || W < (uint32_t)kc || (kc != 16 && kc != 32)) return hipErrorInvalidValue;
        const uint4 *fr = items + (nitems - nfrag);
        // (the blocks the fragments add to are cleared in front of the whole items' kernel: nothing waits for it later)
        if (cum_bytes == 2) hipLaunchKernelGGL((k_zero_frag_blocks<uint16_t>), dim3(nfrag), dim3(256), 0, st, fr, W / (uint32_t)kc, (uint16_t *)cum, nslots);
        else hipLaunchKernelGGL((k_zero_frag_blocks<uint32_t>), dim3(nfrag), dim3(256), 0, st, fr, W / (uint32_t)kc, (uint32_t *)cum, nslots);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
        e = launch_pair_counts_lockstep(st, kc, cum_bytes, planes, Npad, Kpad, W, P, tiles, items, nitems - nfrag, 0, cum, nslots);
        if (e != hipSuccess) return e;
        if (cum_bytes == 2)
            return kc == 16 ? launch_pcl_frag<16, uint16_t>(st, planes, Npad, Kpad, W, P, tiles, fr, nfrag, cum, nslots)
                 : kc == 32 ? launch_pcl_frag<32, uint16_t>(st, planes, Npad, Kpad, W, P, tiles, fr, nfrag, cum, nslots) : hipErrorInvalidValue;
        return kc == 16 ? launch_pcl_frag<16, uint32_t>(st, planes, Npad, Kpad, W, P, tiles, fr, nfrag, cum, nslots)
             : kc == 32 ? launch_pcl_frag<32, uint32_t>(st, planes, Npad, Kpad, W, P, tiles, fr, nfrag, cum, nslots) : hipErrorInvalidValue;
    }
    if (W < (uint32_t)kc)  // plane boundaries inside a chunk (p < 9): the kernel with the in-loop flush
        return launch_pair_counts(st, kc, cum_bytes, planes, Npad, Kpad, W, P, tiles, items, nitems, cum, nslots);
    if (cum_bytes == 2) {
        switch (kc) {
        case 16: return launch_pcl<16, uint16_t>(st, planes, Npad, Kpad, W, P, tiles, items, nitems, cum, nslots);
        case 32: return launch_pcl<32, uint16_t>(st, planes, Npad, Kpad, W, P, tiles, items, nitems, cum, nslots);
        default: return hipErrorInvalidValue;
        }
    }
    switch (kc) {
    case 16: return launch_pcl<16, uint32_t>(st, planes, Npad, Kpad, W, P, tiles, items, nitems, cum, nslots);
    case 32: return launch_pcl<32, uint32_t>(st, planes, Npad, Kpad, W, P, tiles, items, nitems, cum, nslots);
    default: return hipErrorInvalidValue;
    }
}

hipError_t launch_finalize(hipStream_t st, const FinalizeLaunch &f)
{
    if (f.nslots == 0) return hipSuccess;
    FinalizeArgs a;
    a.rowoff = f.rowoff;
    a.cum = f.cum; a.nslots = f.cum_stride; a.tiles = f.tiles; a.perm = f.perm; a.hist_bins = f.hist_bins; a.pbase = f.pbase;
    a.p = f.p; a.estim = f.estim; a.result_type = f.result_type; a.ksinv = f.ksinv;
    a.nS = f.nS; a.keyS = f.keyS; a.cardS = f.cardS; a.thS = f.thS; a.rl = f.rl; a.E = f.E;
    a.cidx_rec = f.cidx_rec; a.cidx_ent = f.cidx_ent; a.nbuckets = f.nbuckets; a.ent_stride = f.ent_stride;
    a.n = f.n; a.ncols = f.ncols; a.rect = f.rect; a.sorted_out = f.sorted_out; a.square = f.square;
    a.knn = f.knn; a.out2 = f.out2; a.knn_ld = f.knn_ld; a.knn_rows = f.knn_rows;
    a.row_begin = f.row_begin; a.row_end = f.row_end; a.col_begin = f.col_begin;
    a.col_end = f.col_end; a.base_index = f.base_index; a.out = f.out;
    a.stop = f.stop;
    a.phase_cyc = f.phase_cyc;
    a.sig = f.sig;
    a.ntiles = (uint32_t)(f.nslots / ((uint64_t)kTile * kTile));
    const size_t lds = (64 + 128 + 8) * sizeof(uint32_t) + (size_t)f.hist_bins * 128 * (f.cum_bytes == 2 ? 2 : 4);
    const uint32_t blocks = (a.ntiles + 7u) / 8u * 8u * 128u;  // (whole groups of 8 tiles: a tile's 128 rows on one XCD)
    const bool timed = f.phase_cyc != nullptr;  // profiling only
    const bool general = f.rect || f.square || f.sorted_out || f.knn;
    if (f.sig && (general || timed)) return hipErrorInvalidValue;  // (the signalling instance is the plain triangle's)
#define DSH_FIN(CT, RK)                                                                                                  \
    do {                                                                                                                 \
        if (general) hipLaunchKernelGGL((k_finalize<CT, RK, false, true>), dim3(blocks), dim3(128), lds, st, a);        \
        else if (timed) hipLaunchKernelGGL((k_finalize<CT, RK, true, false>), dim3(blocks), dim3(128), lds, st, a);      \
        else if (f.sig) hipLaunchKernelGGL((k_finalize_signal<CT, RK>), dim3(blocks), dim3(128), lds, st, a); \
        else hipLaunchKernelGGL((k_finalize<CT, RK, false, false>), dim3(blocks), dim3(128), lds, st, a);                \
    } while (0)
    // (the record width follows the precision like k_build_colindex: colindex_inline)
    if (f.cum_bytes != 2) DSH_FIN(uint32_t, 3);
    else if (colindex_inline(f.p, f.E) == 3) DSH_FIN(uint16_t, 3);
    else DSH_FIN(uint16_t, 7);
#undef DSH_FIN
    return hipGetLastError();
}

}  // namespace dsh

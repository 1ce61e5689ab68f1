// kernels_compare.hip -- all-pairs HLL compare on gfx950 (MI355X, CDNA4, wave64).
//
// Replaces hot loop 2 of the reference: perform_core_op (src/sketch_and_cmp.h:699-710) /
// dm::parallel_fill (distmat/distmat.h:459-512) calling result_cmp (src/dashing.h:568-592)
// -> hll_t::jaccard_index -> union_size = estimate(histogram(max(a,b))).
//
// Formulation.  Per pair the reference needs the histogram
// c[v] = #{t : max(a_t,b_t) = v}.  With thermometer bit-planes  A_v[t] = (a_t < v)
//     C(v) = #{t : max(a_t,b_t) < v} = popcount(A_v & B_v),   c[v] = C(v+1) - C(v),
// exact integers.  So the O(N^2 * 2^p) part is AND + popcount over LDS-staged bit-planes
// (v_and_b32 + v_bcnt_u32_b32, no MFMA: integer work), and the estimator runs once per pair
// in fp64 in a second kernel.
//
// Kernels:
//   k_selfhist_card  per-sketch 64-bin histogram (LDS atomics) -> cardinality + value range
//   k_transform      uint8 registers [N][m] -> bit-plane matrix planes[K][Npad] (u32 words,
//                    row kk = plane*W + word, sketch index fastest)
//   k_pair_counts    64x64-sketch tiles: stage planes rows through double-buffered LDS,
//                    each lane owns a 4x4 block of pairs, AND+popcount; writes C(v) per pair
//   k_finalize       one lane per pair: differences -> histogram (LDS column) -> estimator
//                    -> J -> Mash transform -> float at the packed-triangle index
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "estimators.h"
#include "kernels.h"

namespace dsh {

// ------------------------------------------------------------------------------------------
// per-sketch histogram, cardinality, global min/max register value
// block = 256 threads = 4 waves, one sketch per wave.
__global__ __launch_bounds__(256) void k_selfhist_card(const uint8_t *__restrict__ regs,
                                                        uint64_t n, int p, int estim,
                                                        double *__restrict__ card,
                                                        int *__restrict__ vrange)
{
    __shared__ uint32_t hist[4][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint64_t s = (uint64_t)blockIdx.x * 4 + wave;
    hist[wave][lane] = 0;
    __syncthreads();
    const uint64_t m = 1ull << p;
    if (s < n) {
        const uint4 *src = reinterpret_cast<const uint4 *>(regs + s * m);
        for (uint64_t c = lane; c < (m >> 4); c += 64) {
            const uint4 x = src[c];
            const uint32_t w[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                atomicAdd(&hist[wave][w[k] & 63], 1u);
                atomicAdd(&hist[wave][(w[k] >> 8) & 63], 1u);
                atomicAdd(&hist[wave][(w[k] >> 16) & 63], 1u);
                atomicAdd(&hist[wave][(w[k] >> 24) & 63], 1u);
            }
        }
    }
    __syncthreads();
    if (s < n && lane == 0) {
        const uint32_t *h = hist[wave];
        auto c = [h](int v) -> uint32_t { return h[v]; };
        int lo = 0, hi = 63;
        while (lo < 63 && h[lo] == 0) ++lo;
        while (hi > 0 && h[hi] == 0) --hi;
        card[s] = estimate(c, p, estim, 0, 64 - p + 1);
        atomicMin(&vrange[0], lo);
        atomicMax(&vrange[1], hi);
    }
}

// ------------------------------------------------------------------------------------------
// registers -> thermometer bit-planes.  Thread (i, w) reads 32 registers of sketch i and emits
// one 32-bit word per plane: bit r of planes[(pl*W + w)*Npad + i] = (reg[i][32w + r] < vlo+1+pl).
// Padding sketches (i >= N) and padding rows get zeros (they never count).
__device__ __forceinline__ uint32_t lt_nibble(uint32_t x, uint32_t vrep)
{
    // bytes of x are < 128 (or 0xFF fillers); bit7 of (x|0x80)-v is set iff byte >= v
    const uint32_t ge = ((x | 0x80808080u) - vrep) & 0x80808080u;
    const uint32_t lt = (ge ^ 0x80808080u) >> 7;  // 0/1 per byte
    return (lt * 0x01020408u) >> 24;              // bit k = byte k (no carries: see DESIGN.md)
}

__global__ __launch_bounds__(256) void k_transform(const uint8_t *__restrict__ regs, uint64_t n,
                                                    int p, int vlo, uint32_t P, uint32_t W,
                                                    uint32_t Npad, uint32_t *__restrict__ planes)
{
    const uint64_t gid = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const uint32_t i = (uint32_t)(gid % Npad);
    const uint32_t w = (uint32_t)(gid / Npad);
    if (w >= W) return;
    const uint64_t m = 1ull << p;
    uint32_t x[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) x[k] = 0xFFFFFFFFu;
    if (i < n) {
        const uint8_t *src = regs + (uint64_t)i * m + (uint64_t)w * 32;
        const uint4 a = *reinterpret_cast<const uint4 *>(src);
        x[0] = a.x; x[1] = a.y; x[2] = a.z; x[3] = a.w;
        if (m >= 32) {
            const uint4 b = *reinterpret_cast<const uint4 *>(src + 16);
            x[4] = b.x; x[5] = b.y; x[6] = b.z; x[7] = b.w;
        }
    }
    for (uint32_t pl = 0; pl < P; ++pl) {
        const uint32_t vrep = (uint32_t)(vlo + 1 + (int)pl) * 0x01010101u;
        uint32_t word = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) word |= (lt_nibble(x[k], vrep) & 0xFu) << (4 * k);
        planes[((uint64_t)pl * W + w) * Npad + i] = word;
    }
}

// ------------------------------------------------------------------------------------------
// all-pairs AND+popcount.  One 256-thread workgroup per 64x64 tile of sketches.
//   waves: 2x2, each covers 32x32 pairs; lane (ly,lx) in 8x8 owns a 4x4 block of pairs.
//   per k-row: one ds_read_b128 of 4 A-words (broadcast over lx), one of 4 B-words (broadcast
//   over ly), 16 x (v_and_b32 + v_bcnt_u32_b32 with accumulate).  LDS rows are 256 B
//   (64 sketches x 4 B); the 8 distinct 16-B slots a wave touches per read are contiguous ->
//   conflict-free.
//   K (= planes x words) is streamed in chunks of KC rows through two LDS buffers with
//   direct global->LDS DMA (global_load_lds_dwordx4: lane-linear LDS image == our row-major
//   [row][64] layout, 4 rows per wave-instruction), one barrier per chunk: the DMA of chunk
//   c+1 is in flight while chunk c is consumed.
//   At each plane boundary the 16 counters C(v) are written to cum[pl][tile*4096 + r*64 + c].
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

template <int KC, int U>
__global__ __launch_bounds__(256) void k_pair_counts(const uint32_t *__restrict__ planes,
                                                      uint32_t Npad, uint32_t Kpad, uint32_t W,
                                                      uint32_t P, const uint2 *__restrict__ tiles,
                                                      uint32_t *__restrict__ cum, uint64_t nslots)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];  // [2][A|B][KC][64]
    constexpr int NPASS = KC / 16;  // wave-instructions per operand per chunk (4 rows each)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ii = (wave >> 1) * 32 + (lane >> 3) * 4;
    const int jj = (wave & 1) * 32 + (lane & 7) * 4;
    const uint2 tile = tiles[blockIdx.x];
    // DMA source of this lane: row (4*wave + lane/16) of each 16-row pass, 16 B at column lane%16
    const uint64_t lrow = (uint64_t)(wave * 4 + (lane >> 4));
    const uint32_t *gA = planes + lrow * Npad + (uint64_t)tile.x * 64 + (lane & 15) * 4;
    const uint32_t *gB = planes + lrow * Npad + (uint64_t)tile.y * 64 + (lane & 15) * 4;
    const uint64_t pass_stride = (uint64_t)16 * Npad;

    // LDS-DMA issued from inline asm so that hipcc does not count it: the compiler would
    // otherwise drain it with vmcnt(0) before the first ds_read of the chunk being consumed.
    // We wait for it ourselves (vmcnt(0) right before the barrier that publishes the chunk).
    const uint32_t lds_base = (uint32_t)(uintptr_t)(lds_ptr_t)smem + wave * 1024;  // bytes
    auto stage = [&](uint32_t chunk, int buf) {
        const uint32_t la = lds_base + buf * (2 * KC * 256);
        const uint32_t lb = la + KC * 256;
        const uint32_t *a = gA + (uint64_t)chunk * KC * Npad;
        const uint32_t *b = gB + (uint64_t)chunk * KC * Npad;
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
            glds16(a + ps * pass_stride, la + ps * 4096);
            glds16(b + ps * pass_stride, lb + ps * 4096);
        }
    };

    uint32_t acc[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[r][c] = 0;

    const uint32_t nchunks = Kpad / KC;
    uint32_t *cum_tile = cum + (uint64_t)blockIdx.x * 4096 + (uint64_t)ii * 64 + jj;

    stage(0, 0);
    for (uint32_t ch = 0; ch < nchunks; ++ch) {
        dma_wait();       // this wave's DMA pieces of chunk ch have landed ...
        __syncthreads();  // ... and so have everyone's; buffer (ch+1)&1 is no longer being read
        if (ch + 1 < nchunks) stage(ch + 1, (ch + 1) & 1);
        const uint32_t *As = smem + (ch & 1) * (2 * KC * 64) + ii;
        const uint32_t *Bs = smem + (ch & 1) * (2 * KC * 64) + KC * 64 + jj;
        // U rows (U = min(W, 8), compile time) per step, then a plane-boundary check
        for (uint32_t s0 = 0; s0 < (uint32_t)KC; s0 += U) {
#pragma unroll
            for (uint32_t kk = 0; kk < (uint32_t)U; ++kk) {
                const uint4 a = *reinterpret_cast<const uint4 *>(As + (s0 + kk) * 64);
                const uint4 b = *reinterpret_cast<const uint4 *>(Bs + (s0 + kk) * 64);
                const uint32_t av[4] = {a.x, a.y, a.z, a.w};
                const uint32_t bv[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int c = 0; c < 4; ++c) popc_acc(acc[r][c], av[r] & bv[c]);
            }
            const uint32_t row_end = ch * KC + s0 + U;
            if ((row_end & (W - 1)) == 0) {
                const uint32_t pl = row_end / W - 1;
                if (pl < P) {
                    uint32_t *dst = cum_tile + (uint64_t)pl * nslots;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        *reinterpret_cast<uint4 *>(dst + r * 64) =
                            make_uint4(acc[r][0], acc[r][1], acc[r][2], acc[r][3]);
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int c = 0; c < 4; ++c) acc[r][c] = 0;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// finalize: one lane per pair slot of a band of tiles.
struct FinalizeArgs {
    const uint32_t *cum;
    uint64_t nslots;
    const uint2 *tiles;
    uint32_t P;
    int vlo;
    int p;
    int estim;
    int result_type;
    double ksinv;
    const double *card;
    uint64_t n;
    // triangle mode: rows [row_begin,row_end), out index = tri(i,j) - base_index
    // rect mode (rect != 0): i in [row_begin,row_end) x j in [col_begin,col_end), row-major
    int rect;
    uint64_t row_begin, row_end, col_begin, col_end;
    uint64_t base_index;
    float *out;
};

__global__ __launch_bounds__(256) void k_finalize(FinalizeArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t hs[];  // [(P+1)][256]
    const int tid = threadIdx.x;
    const uint64_t slot = (uint64_t)blockIdx.x * 256 + tid;
    if (slot >= a.nslots) return;
    const uint2 tile = a.tiles[slot >> 12];
    const uint64_t i = (uint64_t)tile.x * 64 + ((slot >> 6) & 63);
    const uint64_t j = (uint64_t)tile.y * 64 + (slot & 63);
    bool active;
    if (a.rect) active = i >= a.row_begin && i < a.row_end && j >= a.col_begin && j < a.col_end;
    else active = i < j && j < a.n && i >= a.row_begin && i < a.row_end;
    if (!active) return;
    const uint32_t m = 1u << a.p;
    uint32_t prev = 0;
    for (uint32_t pl = 0; pl < a.P; ++pl) {
        const uint32_t cv = a.cum[(uint64_t)pl * a.nslots + slot];
        hs[pl * 256 + tid] = cv - prev;
        prev = cv;
    }
    hs[a.P * 256 + tid] = m - prev;
    const int vlo = a.vlo, vhi = a.vlo + (int)a.P;
    const uint32_t *col = hs + tid;
    auto c = [col, vlo, vhi](int v) -> uint32_t {
        return (v < vlo || v > vhi) ? 0u : col[(v - vlo) * 256];
    };
    const double us = estimate(c, a.p, a.estim, vlo, vhi);
    const double ji = jaccard_from(a.card[j], a.card[i], us);
    const float res = result_from_ji(ji, a.result_type, a.ksinv);
    uint64_t oidx;
    if (a.rect) oidx = (i - a.row_begin) * (a.col_end - a.col_begin) + (j - a.col_begin);
    else oidx = i * (2 * a.n - i - 1) / 2 + j - (i + 1) - a.base_index;
    a.out[oidx] = res;
}

// ------------------------------------------------------------------------------------------
// launch wrappers (host)
hipError_t launch_selfhist_card(hipStream_t st, const uint8_t *regs, uint64_t n, int p, int estim,
                                double *card, int *vrange)
{
    if (n == 0) return hipSuccess;
    const uint32_t blocks = (uint32_t)((n + 3) / 4);
    hipLaunchKernelGGL(k_selfhist_card, dim3(blocks), dim3(256), 0, st, regs, n, p, estim, card,
                       vrange);
    return hipGetLastError();
}

hipError_t launch_transform(hipStream_t st, const uint8_t *regs, uint64_t n, int p, int vlo,
                            uint32_t P, uint32_t W, uint32_t Npad, uint32_t *planes)
{
    if (P == 0) return hipSuccess;
    const uint64_t threads = (uint64_t)Npad * W;
    const uint32_t blocks = (uint32_t)((threads + 255) / 256);
    hipLaunchKernelGGL(k_transform, dim3(blocks), dim3(256), 0, st, regs, n, p, vlo, P, W, Npad,
                       planes);
    return hipGetLastError();
}

template <int KC, int U>
static hipError_t launch_pc(hipStream_t st, const uint32_t *planes, uint32_t Npad, uint32_t Kpad,
                            uint32_t W, uint32_t P, const uint2 *tiles, uint32_t ntiles,
                            uint32_t *cum, uint64_t nslots)
{
    static bool attr_set = false;
    const size_t lds = (size_t)KC * 1024;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_pair_counts<KC, U>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    hipLaunchKernelGGL((k_pair_counts<KC, U>), dim3(ntiles), dim3(256), lds, st, planes, Npad,
                       Kpad, W, P, tiles, cum, nslots);
    return hipGetLastError();
}

template <int KC>
static hipError_t launch_pc_u(hipStream_t st, const uint32_t *planes, uint32_t Npad,
                              uint32_t Kpad, uint32_t W, uint32_t P, const uint2 *tiles,
                              uint32_t ntiles, uint32_t *cum, uint64_t nslots)
{
    if (W >= 8) return launch_pc<KC, 8>(st, planes, Npad, Kpad, W, P, tiles, ntiles, cum, nslots);
    if (W == 4) return launch_pc<KC, 4>(st, planes, Npad, Kpad, W, P, tiles, ntiles, cum, nslots);
    if (W == 2) return launch_pc<KC, 2>(st, planes, Npad, Kpad, W, P, tiles, ntiles, cum, nslots);
    return launch_pc<KC, 1>(st, planes, Npad, Kpad, W, P, tiles, ntiles, cum, nslots);
}

hipError_t launch_pair_counts(hipStream_t st, int kc, const uint32_t *planes, uint32_t Npad,
                              uint32_t Kpad, uint32_t W, uint32_t P, const uint2 *tiles,
                              uint32_t ntiles, uint32_t *cum, uint64_t nslots)
{
    if (ntiles == 0 || Kpad == 0) return hipSuccess;
    switch (kc) {
    case 32: return launch_pc_u<32>(st, planes, Npad, Kpad, W, P, tiles, ntiles, cum, nslots);
    case 64: return launch_pc_u<64>(st, planes, Npad, Kpad, W, P, tiles, ntiles, cum, nslots);
    case 128: return launch_pc_u<128>(st, planes, Npad, Kpad, W, P, tiles, ntiles, cum, nslots);
    default: return hipErrorInvalidValue;
    }
}

hipError_t launch_finalize(hipStream_t st, const FinalizeLaunch &f)
{
    if (f.nslots == 0) return hipSuccess;
    FinalizeArgs a;
    a.cum = f.cum; a.nslots = f.nslots; a.tiles = f.tiles; a.P = f.P; a.vlo = f.vlo; a.p = f.p;
    a.estim = f.estim; a.result_type = f.result_type; a.ksinv = f.ksinv; a.card = f.card;
    a.n = f.n; a.rect = f.rect; a.row_begin = f.row_begin; a.row_end = f.row_end;
    a.col_begin = f.col_begin; a.col_end = f.col_end; a.base_index = f.base_index; a.out = f.out;
    const size_t lds = (size_t)(f.P + 1) * 256 * sizeof(uint32_t);
    static size_t attr_lds = 0;
    if (lds > attr_lds) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_finalize),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_lds = lds;
    }
    const uint32_t blocks = (uint32_t)((f.nslots + 255) / 256);
    hipLaunchKernelGGL(k_finalize, dim3(blocks), dim3(256), lds, st, a);
    return hipGetLastError();
}

}  // namespace dsh

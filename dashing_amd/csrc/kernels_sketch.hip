// kernels_sketch.hip -- genome bases -> HLL registers on gfx950.
//
// Replaces hot loop 1 of the reference: enc.for_each([&](u64 kmer){h.addh(kmer);}, ...)
// (src/sketch_and_cmp.h:342, :515): canonical 2-bit k-mer (bonsai Encoder, unspaced,
// unwindowed, k<=32) -> WangHash -> register rule (mirrored at src/readfilt.cpp:86-88)
//     idx = h >> (64-p);  v = clz(((h<<1)|1) << (p-1)) + 1;  reg[idx] = max(reg[idx], v).
//
// Design: no sequential rolling state.  Each lane packs the 64 bases
// at its 32-aligned position into 2-bit words -- F big-endian (first base most significant),
// R = complement, little-endian, V = validity bits -- so the forward and reverse-complement
// k-mer of ANY start position is a funnel-shift window of (F0:F1) / (R0:R1), and "k valid
// bases" is a mask test on V.  Non-ACGT bytes, record separators, bases outside the genome's
// [gbeg,gend) span are all just cleared V bits.  Registers live in LDS as packed bytes
// (2^p B per workgroup); a plain LDS read filters out the (vast majority of) k-mers that
// cannot raise a register, the rest do a CAS on the containing word.  At the end each
// workgroup max-merges its LDS array into the resident matrix with 32-bit CAS (byte-wise
// SWAR max).  max is commutative/associative/idempotent => bit-exact, order-independent.
// Sketches too large for LDS (p > 17; the `hll` subcommand's default is p = 24,
// src/hllmain.cpp:5) take the GLOBAL variant: same k-mer pipeline, but the filter read and
// the CAS go straight to the register array in HBM/L2 (a stale cached read can only be too
// small, i.e. cause a CAS that then sees the true value -- never a lost update).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstdlib>
#include <mutex>
#include <vector>

#include "kernels.h"

namespace dsh {

// Thomas Wang's 64-bit integer hash.  hipcc lowers the shift-and-add steps to 64-bit multiplies (two
// chained v_mad_u64_u32 with a v_mov_b32 each: 24 instructions for the hash).  Two hand-written lowerings were tried
// against it in separate processes and lost: v_mul_lo_u32 + v_add_u32 for the high word (22 instructions, -2.8 % at p = 10,
// profiles/rd6v) and the steps as the v_lshl_add_u64 they are (20 instructions, 122 VGPRs, -0.9 %, profiles/rd6z).
__device__ __forceinline__ uint64_t wang64(uint64_t key)
{
    key = (~key) + (key << 21);
    key = key ^ (key >> 24);
    key = (key + (key << 3)) + (key << 8);
    key = key ^ (key >> 14);
    key = (key + (key << 2)) + (key << 4);
    key = key ^ (key >> 28);
    key = key + (key << 31);
    return key;
}

// 0x80 in every byte of v that is zero (exact, no cross-byte borrow)
__device__ __forceinline__ uint32_t zero_bytes(uint32_t v)
{
    return ~(((v & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | v | 0x7F7F7F7Fu);
}

// ---- pack: 4 ASCII bases (byte 0 = first base) -> 8 bits of big-endian 2-bit codes + 4 validity bits; A0 C1 G2 T3,
// case-folded, anything else invalid (round 5; profiles/rd5d/sketch_instr.json: 425 -> 301 instructions per 32 bases).
// One v_perm_b32 is an 8-entry byte table looked up for 4 bases at once: index = (byte >> 1) & 7 is 0 1 2 3 for A C T G
// in either case; the table of EXPECTED upper-case letters compared with the case-folded byte validates the base (any
// other byte either indexes 4..7 -> 0xFF, or differs from the letter its index stands for), a second table gives the
// 2-bit code.  27 -> 9 instructions per 4 bases for the validity, 3 -> 1 for the code.  The reverse strand is not packed
// at all: R = the bit-pair reversal of ~F (v_bfrev_b32 + one swap of neighbouring bits per 32-bit half).
__device__ __forceinline__ void pack4_perm(uint32_t x, uint32_t &be8, uint32_t &v4)
{
    const uint32_t up = x & 0xDFDFDFDFu;
    const uint32_t idx = (x >> 1) & 0x07070707u;
    // selector bytes 0..3 pick the bytes of the second source, 4..7 those of the first
    const uint32_t expect = __builtin_amdgcn_perm(0xFFFFFFFFu, 0x47544341u, idx);  // 'A' 'C' 'T' 'G', else 0xFF
    const uint32_t code = __builtin_amdgcn_perm(0u, 0x02030100u, idx);             //  0   1   3   2
    const uint32_t ok = zero_bytes(expect ^ up);
    v4 = ((ok >> 7) * 0x01020408u) >> 24;     // bit k = byte k valid
    be8 = (code * 0x40100401u) >> 24;         // c0<<6 | c1<<4 | c2<<2 | c3
}

// the 16 two-bit fields of x in reverse order
__device__ __forceinline__ uint32_t pairrev32(uint32_t x)
{
    const uint32_t y = __builtin_bitreverse32(x);
    return ((y >> 1) & 0x55555555u) | ((y & 0x55555555u) << 1);
}

__device__ __forceinline__ void pack32_perm(const uint4 lo, const uint4 hi, uint64_t &F, uint64_t &R, uint32_t &V)
{
    const uint32_t w[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    uint32_t fh = 0, fl = 0;
    // Almost every wave sees 64 x 32 clean bases: the distance of every byte from the letter its index stands for, summed
    // over the 32 bytes (v_sad_u8, one per word), is then zero in all lanes and the per-byte validity bits -- a zero-byte
    // test, a gather multiply and a shift-or per word -- need not be formed at all: p = 10 9.80e11 -> 1.014e12 bases/s
    // (+3.4 %), p = 14 +2.2 %; where every wave meets an invalid byte (reads: one every 151 bases) -1.3 % (profiles/rd6ae).
    uint32_t dist = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const uint32_t x = w[k];
        const uint32_t idx = (x >> 1) & 0x07070707u;
        const uint32_t expect = __builtin_amdgcn_perm(0xFFFFFFFFu, 0x47544341u, idx);
        const uint32_t code = __builtin_amdgcn_perm(0u, 0x02030100u, idx);
        dist = __builtin_amdgcn_sad_u8(expect, x & 0xDFDFDFDFu, dist);
        const uint32_t be = (code * 0x40100401u) >> 24;
        if (k < 4) fh |= be << (24 - 8 * k);
        else fl |= be << (56 - 8 * k);
    }
    V = 0xFFFFFFFFu;
    if (__any(dist != 0u)) {
        V = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            uint32_t be, v;
            pack4_perm(w[k], be, v);
            V |= v << (4 * k);
        }
    }
    F = ((uint64_t)fh << 32) | fl;
    // R: complement codes, little-endian (base 0 in the lowest two bits) = the pair-reversal of ~F
    R = ((uint64_t)pairrev32(~fl) << 32) | pairrev32(~fh);
}

__device__ __forceinline__ uint32_t bytemax4(uint32_t a, uint32_t b)
{
    // bytes < 128: bit7 of (a|0x80)-b set iff a_byte >= b_byte
    const uint32_t ge = (((a | 0x80808080u) - b) >> 7) & 0x01010101u;
    const uint32_t mask = ge * 0xFFu;
    return (a & mask) | (b & ~mask);
}

// windows of k consecutive valid bases: bit j of the result is set iff bits j..j+k-1 of V are set
// (binary decomposition of k: log2(k) doubling steps + one AND per set bit of k)
__device__ __forceinline__ uint64_t valid_windows(uint64_t V, int k)
{
    uint64_t run = V;      // runs of length `len`
    uint64_t acc = ~0ull;  // AND of the pieces taken so far
    int len = 1, off = 0;
    for (int bit = 0; bit < 6; ++bit) {
        if (k & (1 << bit)) {
            acc &= run >> off;
            off += len;
        }
        run &= run >> len;
        len <<= 1;
    }
    return acc;
}

// REG32 (p <= kMaxPReg32, round 6): the workgroup's registers are ONE 32-bit word each, so the register rule is a single
// native LDS atomic -- ds_max_u32, nothing returned, nothing to wait for -- instead of a byte read that filters, a branch
// and a compare-and-swap loop on the containing word: a wave walked that 13-instruction path whenever ONE of its 64
// lanes raised a register (~5 of 57 VALU per k-mer at p = 10, profiles/rd5g/sketch_instr.json).  4 KiB of LDS at p = 10,
// 64 KiB at p = 14, 128 KiB at p = 15 (workgroups of 512 and 1 024 lanes there: launch_sketch); packed bytes for p = 16, 17.
// The word holds value - 1 = clz(t), signed, -1 = untouched (the merge adds the 1): clz(t) is the clz of t's HIGH word
// unless that word is zero (32 zero hash bits behind the index: 2^-32 of all k-mers), so the common case is one
// v_ffbh_u32 + ds_max_i32 -- v_ffbh_u32 gives -1 for 0, which a signed max ignores -- instead of two v_ffbh_u32, an add, a
// min, the guard bit and the + 1; a lane remembers whether one of its high words was zero and applies the exact rule to
// those k-mers behind the sub-chunk (idempotent max).  A/B in separate processes (profiles/rd6w/sk_ab.jsonl): p = 10
// 8.72e11 -> 9.48e11 bases/s (+8.7 %), p = 14 +6 %; the path behind the zero word is exercised by k-mers made for it
// (the hash is invertible: tests/test_gpu_sketch.py::test_kmers_whose_hash_has_32_zero_bits_behind_the_index).
template <bool GLOBAL, bool CANON, bool REG32, int KC, int NT>
__global__ __launch_bounds__(NT) void k_sketch(const uint8_t *__restrict__ seq,
                                                 const SketchWork *__restrict__ work, int k,
                                                 int p, uint8_t *__restrict__ regs)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    // [0, 2^p/4): registers as packed bytes; then the packed words of the current sub-chunk, one
    // slot per lane plus one for the 32 bases that follow it (each lane needs its right neighbour)
    const int tid = threadIdx.x;
    const uint32_t mwords = GLOBAL ? 0u : (REG32 ? (1u << p) : (1u << p) >> 2);
    const SketchWork wk = work[blockIdx.x];
    uint32_t *lregs = GLOBAL ? reinterpret_cast<uint32_t *>(regs + ((uint64_t)wk.slot << p)) : lds;
    uint64_t *xF = reinterpret_cast<uint64_t *>(lds + ((mwords + 3) & ~3u));
    uint64_t *xR = xF + (NT + 4);
    uint32_t *xV = reinterpret_cast<uint32_t *>(xR + (NT + 4));
    // FAST (word registers): a register holds value - 1 = the count of leading zeros, as a signed word, -1 = untouched
    constexpr bool FAST = REG32 && !GLOBAL;
    for (uint32_t w = tid; w < mwords; w += NT) lregs[w] = FAST ? 0xFFFFFFFFu : 0u;

    if constexpr (KC != 0) k = KC;  // (KC: the k-mer length as a compile-time constant -- dashing's default 31 -- else the argument)
    const uint64_t kmask = k == 32 ? ~0ull : ((1ull << (2 * k)) - 1);
    const int fshift = 64 - 2 * k;
    const uint64_t guard = 1ull << (p - 1);  // ((h << 1) | 1) << (p - 1) == (h << p) | guard
    uint32_t guard_v = (uint32_t)guard;  // (in a VGPR: a VOP3 instruction reads ONE scalar register, and the shift amount is one)
    asm("" : "+v"(guard_v));

    // pack the 32 bases at absolute offset B (bases outside [gbeg,gend) are invalid)
    auto pack_at = [&](uint64_t B, uint64_t &F, uint64_t &R, uint32_t &V) {
        F = 0; R = 0; V = 0;
        if (B >= wk.gend) return;
        const uint4 *src = reinterpret_cast<const uint4 *>(seq + B);
        pack32_perm(src[0], src[1], F, R, V);
        const uint64_t lo = wk.gbeg > B ? wk.gbeg - B : 0;
        const uint64_t hi = wk.gend - B;  // > 0
        uint32_t rmask = hi >= 32 ? 0xFFFFFFFFu : ((1u << hi) - 1);
        rmask = lo >= 32 ? 0u : (rmask & ~((1u << lo) - 1));
        V &= rmask;
    };

    // Software pipeline over the work item's bases in steps of NT x 32 (NT = 256: one 8 192-base sub-chunk of the work list
    // per step; the 512- and 1 024-lane workgroups of p = 14 and 15 -- whose registers leave room for two and one workgroup
    // per CU -- take two and four): a lane packs the NEXT step's 32 bases while this one's k-mers are
    // processed (the global loads travel under the hash arithmetic), and lane 0's look-ahead IS the right neighbour of
    // the last lane -- no separate pack of the 32 bases behind the step (round 4: wave 0 packed them on top of its own,
    // 2.3 of 57.5 instructions per k-mer).  Behind the work item's last step only wave 0 looks ahead.  Lanes of the last
    // step that lie behind the item's end start no k-mer (the next work item does), but their bases are the right
    // neighbours of the lanes in front of them.
    constexpr uint32_t kStep = (uint32_t)NT * 32u;
    const uint64_t item_end = wk.start + (uint64_t)wk.nsub * kSketchSub;
    const uint32_t nstep = (uint32_t)(((uint64_t)wk.nsub * kSketchSub + kStep - 1) / kStep);
    uint64_t F0, R0;
    uint32_t V0;
    pack_at(wk.start + (uint64_t)tid * 32, F0, R0, V0);
    for (uint32_t s = 0; s < nstep; ++s) {
        const uint64_t Bn = wk.start + (uint64_t)(s + 1) * kStep + (uint64_t)tid * 32;
        uint64_t Fn = 0, Rn = 0;
        uint32_t Vn = 0;
        if (s + 1 < nstep || tid < 64) pack_at(Bn, Fn, Rn, Vn);  // (wave-uniform)
        __syncthreads();  // previous sub-chunk's neighbour reads (and the register clear) are done
        xF[tid] = F0; xR[tid] = R0; xV[tid] = V0;
        if (tid == 0) { xF[NT] = Fn; xR[NT] = Rn; xV[NT] = Vn; }
        __syncthreads();
        const uint64_t F1 = xF[tid + 1], R1 = xR[tid + 1];
        const uint64_t V = (uint64_t)V0 | ((uint64_t)xV[tid + 1] << 32);
        uint32_t ok = (uint32_t)valid_windows(V, k);  // bit j: a k-mer starts at base B+j
        if (NT != 256 && wk.start + (uint64_t)s * kStep + (uint64_t)tid * 32 >= item_end) ok = 0;  // (behind the item's end)
        uint32_t mn = 0xFFFFFFFFu;  // (FAST) the smallest high word of t among this lane's k-mers of the sub-chunk
        const uint32_t fwv[4] = {(uint32_t)(F0 >> 32), (uint32_t)F0, (uint32_t)(F1 >> 32), (uint32_t)F1};
        const uint32_t rwv[4] = {(uint32_t)R0, (uint32_t)(R0 >> 32), (uint32_t)R1, (uint32_t)(R1 >> 32)};
        // the k-mer that starts at base B + j (j a constant after unrolling, so the word selection folds away): 64-bit
        // windows of (F0:F1) << 2j and (R1:R0) >> 2j, one v_alignbit_b32 per 32-bit half
        auto kmer_at = [&](const int j) {
            const int q = (2 * j) >> 5, r = (2 * j) & 31;
            const uint32_t rlo = __builtin_amdgcn_alignbit(rwv[q + 1], rwv[q], r);
            const uint32_t rhi = __builtin_amdgcn_alignbit(rwv[q + 2], rwv[q + 1], r);
            const uint64_t rl = ((uint64_t)rhi << 32) | rlo;
            const uint64_t rc = rl & kmask;
            uint64_t fw;
            if constexpr (KC != 0) {
                // with k known the forward window is taken where it ENDS: (F0:F1) >> s, s = 128 - 2j - 2k, one funnel
                // shift per half and the mask on the high word -- no 64-bit shift by 64 - 2k behind the two funnel shifts
                const int s = 128 - 2 * j - 2 * KC, qs = s >> 5, rs = s & 31;
                auto L = [&](const int i) -> uint32_t { return i <= 3 ? fwv[3 - i] : 0u; };  // (F0:F1) as little-endian words
                const uint32_t lo = rs ? __builtin_amdgcn_alignbit(L(qs + 1), L(qs), rs) : L(qs);
                const uint32_t hi = (rs ? __builtin_amdgcn_alignbit(L(qs + 2), L(qs + 1), rs) : L(qs + 1)) & (uint32_t)(kmask >> 32);
                fw = ((uint64_t)hi << 32) | lo;
            } else {
                const uint32_t fhi = r ? __builtin_amdgcn_alignbit(fwv[q], fwv[q + 1], 32 - r) : fwv[q];
                const uint32_t flo = r ? __builtin_amdgcn_alignbit(fwv[q + 1], fwv[q + 2], 32 - r) : fwv[q + 1];
                fw = (((uint64_t)fhi << 32) | flo) >> fshift;
            }
            const uint64_t km = (CANON && rc < fw) ? rc : fw;
            const uint64_t h = wang64(km);
            // index and register value from the 32-bit halves of h (4 <= p <= 24 < 32): the index is the top of the high
            // word; t = (h << p) | guard has its high word from one funnel shift, its low word from one shift-or (the guard
            // bit p - 1 lies there); value - 1 = clz(t); v_ffbh_u32 gives -1 for 0, so min() picks the right half
            const uint32_t hhi = (uint32_t)(h >> 32), hlo = (uint32_t)h;
            const uint32_t idx = hhi >> (32 - p);
            const uint32_t thi = __builtin_amdgcn_alignbit(hhi, hlo, 32 - p);
            if constexpr (FAST) {
                // value - 1 = clz(t) is the clz of t's HIGH word unless that word is zero -- 32 zero hash bits behind the
                // index: 2^-32 of all k-mers.  v_ffbh_u32 gives -1 for 0 and a signed max with -1 changes nothing, so
                // the common case is one instruction + the LDS atomic; `mn` remembers whether a high word was zero and
                // the exact rule is then applied to those k-mers behind the sub-chunk (max is idempotent).
                int lzs;
                asm("v_ffbh_u32 %0, %1" : "=v"(lzs) : "v"(thi));
                atomicMax(reinterpret_cast<int *>(lregs) + idx, lzs);  // ds_max_i32
                mn = thi < mn ? thi : mn;
                return;
            }
            uint32_t tlo;  // (hlo << p) | guard in ONE instruction (hipcc emits two: both operands would be scalar registers)
            asm("v_lshl_or_b32 %0, %1, %2, %3" : "=v"(tlo) : "v"(hlo), "s"(p), "v"(guard_v));
            uint32_t zh, zl;
            asm("v_ffbh_u32 %0, %1" : "=v"(zh) : "v"(thi));
            asm("v_ffbh_u32 %0, %1" : "=v"(zl) : "v"(tlo));
            uint32_t lz = zh < zl + 32u ? zh : zl + 32u;
            asm("" : "+v"(lz));  // keep the comparison below in 32 bits (hipcc otherwise widens it to u64)
            const uint32_t val = lz + 1u;
            if constexpr (REG32) {
                atomicMax(&lregs[idx], val);  // ds_max_u32
                return;
            }
            // filter with a plain byte read: almost no k-mer can raise its register
            if (reinterpret_cast<const uint8_t *>(lregs)[idx] > lz) return;
            uint32_t *wp = &lregs[idx >> 2];
            const uint32_t sh = (idx & 3u) * 8u;
            uint32_t old = *wp;
            while (((old >> sh) & 0xFFu) < val) {
                const uint32_t nw = (old & ~(0xFFu << sh)) | (val << sh);
                const uint32_t prev = atomicCAS(wp, old, nw);
                if (prev == old) break;
                old = prev;
            }
        };
        if (ok == 0) {
            // (nothing starts in this lane's 32 positions)
        } else if (__all(ok == 0xFFFFFFFFu)) {  // (the common case: no N, no record edge -- no per-position test)
#pragma unroll
            for (int j = 0; j < 32; ++j) kmer_at(j);
        } else {
#pragma unroll
            for (int j = 0; j < 32; ++j)
                if (ok & (1u << j)) kmer_at(j);
        }
        if constexpr (FAST) {
            if (__any(mn == 0u)) {  // (never, in practice: once per 4e9 k-mers)
                if (mn == 0u) {
#pragma unroll 1
                    for (int j = 0; j < 32; ++j) {
                        if (!((ok >> j) & 1u)) continue;
                        const uint64_t fh = j ? ((F0 << (2 * j)) | (F1 >> (64 - 2 * j))) : F0;
                        const uint64_t rl = j ? ((R0 >> (2 * j)) | (R1 << (64 - 2 * j))) : R0;
                        const uint64_t fw = fh >> fshift, rc = rl & kmask;
                        const uint64_t h = wang64((CANON && rc < fw) ? rc : fw);
                        const uint64_t t = (h << p) | guard;
                        if ((uint32_t)(t >> 32) == 0u)
                            atomicMax(reinterpret_cast<int *>(lregs) + (uint32_t)(h >> (64 - p)), 32 + __builtin_clz((uint32_t)t));
                    }
                }
            }
        }
        F0 = Fn; R0 = Rn; V0 = Vn;
    }
    if (GLOBAL) return;
    __syncthreads();
    uint32_t *g = reinterpret_cast<uint32_t *>(regs + ((uint64_t)wk.slot << p));
    const uint32_t gwords = (1u << p) >> 2;
    for (uint32_t w = tid; w < gwords; w += NT) {
        uint32_t mine;
        if constexpr (REG32) {
            const uint4 q = reinterpret_cast<const uint4 *>(lregs)[w];  // four registers -> their packed bytes
            if constexpr (FAST) mine = (q.x + 1u) | ((q.y + 1u) << 8) | ((q.z + 1u) << 16) | ((q.w + 1u) << 24);
            else mine = q.x | (q.y << 8) | (q.z << 16) | (q.w << 24);
        } else {
            mine = lregs[w];
        }
        if (!mine) continue;
        uint32_t old = g[w];
        for (;;) {
            const uint32_t nw = bytemax4(old, mine);
            if (nw == old) break;
            const uint32_t prev = atomicCAS(&g[w], old, nw);
            if (prev == old) break;
            old = prev;
        }
    }
}

hipError_t ensure_dynamic_lds(const void *kernel, size_t bytes)
{
    // (a handful of kernels x devices: a small table under a mutex; the CLI's one-thread-per-device path gets here from
    // several threads)
    struct Ent { const void *k; int dev; size_t granted; };
    static std::mutex mu;
    static std::vector<Ent> tab;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lk(mu);
    for (Ent &t : tab)
        if (t.k == kernel && t.dev == dev) {
            if (bytes <= t.granted) return hipSuccess;
            e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
            if (e == hipSuccess) t.granted = bytes;
            return e;
        }
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e == hipSuccess) tab.push_back(Ent{kernel, dev, bytes});
    return e;
}

template <bool GLOBAL, bool REG32, int KC, int NT>
static hipError_t launch_sketch_v(hipStream_t st, const uint8_t *seq, const SketchWork *work, uint32_t nwork, int k, int p, int canon,
                                  uint8_t *regs, size_t lds)
{
    if (lds > (48u << 10)) {
        hipError_t e = ensure_dynamic_lds(canon ? reinterpret_cast<const void *>(k_sketch<GLOBAL, true, REG32, KC, NT>)
                                                : reinterpret_cast<const void *>(k_sketch<GLOBAL, false, REG32, KC, NT>), lds);
        if (e != hipSuccess) return e;
    }
    if (canon) hipLaunchKernelGGL((k_sketch<GLOBAL, true, REG32, KC, NT>), dim3(nwork), dim3(NT), lds, st, seq, work, k, p, regs);
    else hipLaunchKernelGGL((k_sketch<GLOBAL, false, REG32, KC, NT>), dim3(nwork), dim3(NT), lds, st, seq, work, k, p, regs);
    return hipGetLastError();
}

// the LDS of a workgroup of NT lanes: its registers (16-byte aligned) + the exchange slots, (NT + 4) x (F, R, V)
static size_t sketch_lds(size_t reg_bytes, int nt) { return ((reg_bytes + 15) & ~(size_t)15) + (size_t)(nt + 4) * 20 + 16; }

hipError_t preload_sketch_kernels()
{
    hipFuncAttributes fa;
    return hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(k_sketch<false, true, true, 31, 256>));
}

hipError_t launch_sketch(hipStream_t st, const uint8_t *seq, const SketchWork *work,
                         uint32_t nwork, int k, int p, int canon, uint8_t *regs)
{
    if (nwork == 0) return hipSuccess;
    if (p > kMaxPLds) return launch_sketch_v<true, false, 0, 256>(st, seq, work, nwork, k, p, canon, regs, sketch_lds(0, 256));
    // registers: a word each up to p = kMaxPReg32, packed bytes above
    const bool reg32 = p <= kMaxPReg32;
    // A workgroup's registers decide how many workgroups share a CU: 64 KiB (words at p = 14, bytes at p = 16) leave room
    // for two, 128 KiB (p = 15, p = 17) for one -- with 256 lanes that is 2 and 1 wave per SIMD.  Those precisions run 512
    // and 1 024 lanes per workgroup (4 waves per SIMD again; a step of the loop is then 2 or 4 sub-chunks of the work
    // list).  A/B behind an environment switch, since removed, 300 x 5 Mbp, registers identical (profiles/rd6ab):
    //   p = 14  6.85e11 -> 7.86e11 bases/s (+15 %)      p = 15  5.27e11 (packed bytes) -> 6.2e11 (words, 1 024 lanes; 256 lanes: 3.7e11)
    //   p = 16  3.06e11 -> 4.61e11 (+51 %)              p = 17  1.33e11 -> 3.31e11 (x2.5)
    if (!reg32) {  // packed bytes
        const size_t rbb = (size_t)1 << p;
        if (p >= 17) return launch_sketch_v<false, false, 0, 1024>(st, seq, work, nwork, k, p, canon, regs, sketch_lds(rbb, 1024));
        if (p == 16) return launch_sketch_v<false, false, 0, 512>(st, seq, work, nwork, k, p, canon, regs, sketch_lds(rbb, 512));
        return launch_sketch_v<false, false, 0, 256>(st, seq, work, nwork, k, p, canon, regs, sketch_lds(rbb, 256));
    }
    // dashing's default k = 31 (src/distmain.cpp:29) has an instance of its own with the window arithmetic folded: p = 10
    // 9.40e11 -> 9.67e11 bases/s (+2.9 %), p = 14 +2.3 % (A/B in separate processes behind an environment switch, since
    // removed: profiles/rd6y/sk_k31_ab.jsonl)
    const size_t rb = (size_t)4 << p;
    if (p >= 15) {
        if (k == 31) return launch_sketch_v<false, true, 31, 1024>(st, seq, work, nwork, k, p, canon, regs, sketch_lds(rb, 1024));
        return launch_sketch_v<false, true, 0, 1024>(st, seq, work, nwork, k, p, canon, regs, sketch_lds(rb, 1024));
    }
    if (p == 14) {
        if (k == 31) return launch_sketch_v<false, true, 31, 512>(st, seq, work, nwork, k, p, canon, regs, sketch_lds(rb, 512));
        return launch_sketch_v<false, true, 0, 512>(st, seq, work, nwork, k, p, canon, regs, sketch_lds(rb, 512));
    }
    if (k == 31) return launch_sketch_v<false, true, 31, 256>(st, seq, work, nwork, k, p, canon, regs, sketch_lds(rb, 256));
    return launch_sketch_v<false, true, 0, 256>(st, seq, work, nwork, k, p, canon, regs, sketch_lds(rb, 256));
}

}  // namespace dsh

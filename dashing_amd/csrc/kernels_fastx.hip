// kernels_fastx.hip -- FASTA text -> the clean base stream k_sketch walks, on the device.
//
// The reference's hot loop 1 includes the parse: Encoder::for_each(func, path) reads every record with kseq and feeds
// its sequence to the k-mer loop (src/sketch_and_cmp.h:338-342).  Through round 5 the host parsed (16 threads
// at 0.6 GB/s each) and the GPU idled 98.7 % of BASELINE configs[1] end to end.  Here the host only read()s the raw file
// bytes into page-locked staging; three small kernels turn them into what the host parser (host/host.cpp FastxParser)
// would have produced -- header lines become ONE invalid byte ('N': k-mers never span records), '\n' and '\r' vanish,
// everything else is copied (k_sketch validates and case-folds the bases itself) -- at the SAME offsets of a second
// buffer, the rest of every genome's region filled with 'N', so that the sketch work list is known to the host before a
// byte has been decoded (no length travels back).
//
// A byte's fate depends on the first character of its line, which may lie any distance to the left (a 5 Mbp genome on
// one line is legal FASTA): a carry with three states -- the line that runs into this position is a HEADER line, a
// SEQUENCE line, or a FRESH line starts exactly here -- composed as functions {HDR,SEQ,FRESH} -> {HDR,SEQ,FRESH}:
//   lane (64 bytes)   bit masks of '\n', '\r', header characters; header spans by ONE 64-bit add (a carry that runs from a
//                     header's first byte to its newline); the lane's transfer function in 6 bits
//   workgroup (16 KB) prefix composition over its 256 lanes                                       k_fastx_scan / _compact
//   genome            prefix composition + prefix sum of the kept bytes over its chunks           k_fastx_offsets
// What is NOT plain FASTA -- a genome that does not begin with '>', a line that begins with '+' (FASTQ: the quality
// lines may hold any character and need record state) -- raises the genome's status word: nothing of it is emitted
// (its region becomes all 'N': no k-mer) and the host parses that file itself.  HBM-bound byte work: no MFMA.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.h"

namespace dsh {

namespace {

constexpr uint32_t FX_HDR = 0, FX_SEQ = 1, FX_FRESH = 2;
constexpr uint32_t kFnIdentity = FX_HDR | (FX_SEQ << 2) | (FX_FRESH << 4);

__device__ __forceinline__ uint32_t fn_const(uint32_t c) { return c | (c << 2) | (c << 4); }
__device__ __forceinline__ uint32_t fn_apply(uint32_t f, uint32_t x) { return (f >> (2 * x)) & 3u; }
// first a, then b
__device__ __forceinline__ uint32_t fn_then(uint32_t a, uint32_t b)
{
    return fn_apply(b, fn_apply(a, 0)) | (fn_apply(b, fn_apply(a, 1)) << 2) | (fn_apply(b, fn_apply(a, 2)) << 4);
}

// bit k = byte k of w equals the byte replicated in pat
__device__ __forceinline__ uint32_t eq4(uint32_t w, uint32_t pat)
{
    const uint32_t v = w ^ pat;
    const uint32_t z = ~(((v & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | v | 0x7F7F7F7Fu);  // 0x80 where the byte is zero
    return ((z >> 7) * 0x01020408u) >> 24;
}

struct LaneMasks {
    uint64_t nl, nl_real, cr, hc, pl;
    uint32_t w[16];
};

// the 64 bytes of this lane (absent ones -- behind the chunk's length -- read as '\n': dropped, harmless)
__device__ __forceinline__ void lane_masks(const uint8_t *__restrict__ raw, uint64_t abs, uint32_t have, LaneMasks &m)
{
    if (have) {
        const uint4 *src = reinterpret_cast<const uint4 *>(raw + abs);
        const uint4 a = src[0], b = src[1], c = src[2], d = src[3];
        const uint32_t w[16] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x, c.y, c.z, c.w, d.x, d.y, d.z, d.w};
#pragma unroll
        for (int k = 0; k < 16; ++k) m.w[k] = w[k];
    } else {
#pragma unroll
        for (int k = 0; k < 16; ++k) m.w[k] = 0;
    }
    uint32_t nl[2] = {0, 0}, cr[2] = {0, 0}, hc[2] = {0, 0}, pl[2] = {0, 0};
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int h = k >> 3, s = 4 * (k & 7);
        nl[h] |= eq4(m.w[k], 0x0A0A0A0Au) << s;
        cr[h] |= eq4(m.w[k], 0x0D0D0D0Du) << s;
        hc[h] |= (eq4(m.w[k], 0x3E3E3E3Eu) | eq4(m.w[k], 0x40404040u)) << s;  // '>' '@' (the host parser takes both)
        pl[h] |= eq4(m.w[k], 0x2B2B2B2Bu) << s;
    }
    const uint64_t valid = have >= 64 ? ~0ull : ((1ull << have) - 1);
    m.nl_real = (((uint64_t)nl[1] << 32) | nl[0]) & valid;
    m.nl = m.nl_real | ~valid;
    m.cr = (((uint64_t)cr[1] << 32) | cr[0]) & valid;
    m.hc = (((uint64_t)hc[1] << 32) | hc[0]) & valid;
    m.pl = (((uint64_t)pl[1] << 32) | pl[0]) & valid;
}

// the lane's transfer function: with a newline inside, whatever came in is forgotten
__device__ __forceinline__ uint32_t lane_fn(const LaneMasks &m)
{
    if (m.nl == 0) return FX_HDR | (FX_SEQ << 2) | (((m.hc & 1) ? FX_HDR : FX_SEQ) << 4);
    const uint64_t x = ~m.nl, hs = (m.nl << 1) & m.hc;
    const bool open = x + hs < x;  // the last header line has no newline yet
    return fn_const(open ? FX_HDR : ((m.nl >> 63) ? FX_FRESH : FX_SEQ));
}

// with carry-in s: the bytes that are emitted (sequence bytes as they are, a header's first byte as 'N')
__device__ __forceinline__ uint64_t lane_out(const LaneMasks &m, uint32_t s, uint64_t &hdr_start)
{
    const uint64_t ls = (m.nl << 1) | (s == FX_FRESH ? 1ull : 0ull);
    hdr_start = ls & m.hc;
    const uint64_t hsx = hdr_start | (s == FX_HDR ? 1ull : 0ull);
    const uint64_t x = ~m.nl;
    const uint64_t span = (x + hsx) ^ x;  // from every header start to its newline, inclusive
    return (~span & ~m.nl & ~m.cr) | hdr_start;
}

// exclusive prefix composition of the lanes' functions over the workgroup (256 lanes): Hillis-Steele through LDS
__device__ __forceinline__ uint32_t wg_scan_fn(uint32_t f, uint8_t *sh, uint32_t &total)
{
    const int t = threadIdx.x;
    sh[t] = (uint8_t)f;
    __syncthreads();
#pragma unroll
    for (int d = 1; d < 256; d <<= 1) {
        uint32_t v = sh[t];
        if (t >= d) v = fn_then(sh[t - d], v);
        __syncthreads();
        sh[t] = (uint8_t)v;
        __syncthreads();
    }
    total = sh[255];
    const uint32_t ex = t ? sh[t - 1] : kFnIdentity;
    __syncthreads();
    return ex;
}

}  // namespace

// pass A: every chunk's transfer function and its emitted bytes for each of the three carries; the genome's status
__global__ __launch_bounds__(256) void k_fastx_scan(const uint8_t *__restrict__ raw, const FastxChunk *__restrict__ chunks,
                                                     const FastxGenome *__restrict__ genomes, uint4 *__restrict__ summ,
                                                     uint32_t *__restrict__ status)
{
    __shared__ uint8_t sh[256];
    __shared__ uint32_t cnt[3];
    const FastxChunk ck = chunks[blockIdx.x];
    const int t = threadIdx.x;
    if (t < 3) cnt[t] = 0;
    const uint32_t at = (uint32_t)t * 64u;
    const uint32_t have = at < ck.len ? (ck.len - at < 64u ? ck.len - at : 64u) : 0u;
    LaneMasks m;
    lane_masks(raw, ck.begin + at, have, m);
    // not plain FASTA: the genome does not begin with '>', or a line begins with '+' (the byte behind this lane's last
    // newline may be the next lane's, or the next chunk's, first)
    bool bad = false;
    if (have) {
        const FastxGenome g = genomes[ck.genome];
        const uint64_t next = ck.begin + at + 64;
        const uint32_t nb = next < g.off + g.rawlen ? raw[next] : 0u;
        bad = (m.nl_real & ((m.pl >> 1) | ((uint64_t)(nb == '+') << 63))) != 0;
        if (ck.begin + at == g.off) bad = bad || (m.w[0] & 0xFFu) != '>';
    }
    if (bad) atomicOr(&status[ck.genome], 1u);
    uint32_t total;
    const uint32_t ex = wg_scan_fn(lane_fn(m), sh, total);
    uint64_t hs;
    const uint32_t c[3] = {(uint32_t)__popcll(lane_out(m, FX_HDR, hs)), (uint32_t)__popcll(lane_out(m, FX_SEQ, hs)),
                           (uint32_t)__popcll(lane_out(m, FX_FRESH, hs))};
#pragma unroll
    for (uint32_t x = 0; x < 3; ++x) {
        const uint32_t s = fn_apply(ex, x);
        uint32_t v = s == FX_HDR ? c[0] : (s == FX_SEQ ? c[1] : c[2]);
#pragma unroll
        for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o);
        if ((t & 63) == 0) atomicAdd(&cnt[x], v);
    }
    __syncthreads();
    if (t == 0) summ[blockIdx.x] = make_uint4(total, cnt[0], cnt[1], cnt[2]);
}

// pass B: one workgroup per genome -- every chunk's carry and where its output starts; the decoded length
__global__ __launch_bounds__(256) void k_fastx_offsets(const FastxGenome *__restrict__ genomes, const uint4 *__restrict__ summ,
                                                        const uint32_t *__restrict__ status, uint2 *__restrict__ state,
                                                        uint64_t *__restrict__ declen)
{
    __shared__ uint32_t sf[256];
    __shared__ uint64_t sc[256][3];
    const FastxGenome g = genomes[blockIdx.x];
    const int t = threadIdx.x;
    if (status[blockIdx.x]) {  // the host will parse this one: nothing is emitted
        for (uint32_t c = t; c < g.nchunks; c += 256) state[g.chunk0 + c] = make_uint2(0xFFFFFFFFu, 0xFFu);
        if (t == 0) declen[blockIdx.x] = 0;
        return;
    }
    const uint32_t per = (g.nchunks + 255) / 256;
    const uint32_t c0 = (uint32_t)t * per, c1 = c0 + per < g.nchunks ? c0 + per : g.nchunks;
    // this lane's run of chunks as ONE element: function + emitted bytes per incoming carry
    uint32_t f = kFnIdentity;
    uint64_t n3[3] = {0, 0, 0};
    for (uint32_t c = c0; c < c1; ++c) {
        const uint4 s = summ[g.chunk0 + c];
        const uint32_t cc[3] = {s.y, s.z, s.w};
#pragma unroll
        for (uint32_t x = 0; x < 3; ++x) {
            const uint32_t y = fn_apply(f, x);
            n3[x] += y == 0 ? cc[0] : (y == 1 ? cc[1] : cc[2]);
        }
        f = fn_then(f, s.x);
    }
    sf[t] = f;
    sc[t][0] = n3[0], sc[t][1] = n3[1], sc[t][2] = n3[2];
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {  // inclusive scan of (function, counts): (a then b)[x] = a.n[x] + b.n[a.f(x)]
        uint32_t vf = sf[t];
        uint64_t vn[3] = {sc[t][0], sc[t][1], sc[t][2]};
        if (t >= d) {
            const uint32_t af = sf[t - d];
#pragma unroll
            for (uint32_t x = 0; x < 3; ++x) {
                const uint32_t y = fn_apply(af, x);
                vn[x] = sc[t - d][x] + (y == 0 ? sc[t][0] : (y == 1 ? sc[t][1] : sc[t][2]));
            }
            vf = fn_then(af, vf);
        }
        __syncthreads();
        sf[t] = vf;
        sc[t][0] = vn[0], sc[t][1] = vn[1], sc[t][2] = vn[2];
        __syncthreads();
    }
    // the genome starts FRESH: this lane's carry and offset, then its chunks one after the other
    uint32_t carry = t ? fn_apply(sf[t - 1], FX_FRESH) : FX_FRESH;
    uint64_t off = t ? sc[t - 1][FX_FRESH] : 0;
    for (uint32_t c = c0; c < c1; ++c) {
        const uint4 s = summ[g.chunk0 + c];
        // (a chunk's output starts less than 4 GiB into its genome's region or the genome is refused below)
        state[g.chunk0 + c] = make_uint2((uint32_t)off, carry | ((uint32_t)(off >> 32) << 8));
        off += carry == 0 ? s.y : (carry == 1 ? s.z : s.w);
        carry = fn_apply(s.x, carry);
    }
    if (t == 255) declen[blockIdx.x] = sc[255][FX_FRESH];
}

// pass C: the chunk's emitted bytes, compacted through LDS, to out + (genome offset + the chunk's offset)
__global__ __launch_bounds__(256) void k_fastx_compact(const uint8_t *__restrict__ raw, const FastxChunk *__restrict__ chunks,
                                                        const FastxGenome *__restrict__ genomes, const uint2 *__restrict__ state,
                                                        uint8_t *__restrict__ out)
{
    __shared__ uint8_t sh[256];
    __shared__ uint32_t wsum[4];
    __shared__ __attribute__((aligned(16))) uint8_t stage[kFastxChunk + 32];
    const uint2 st = state[blockIdx.x];
    if ((st.y & 0xFFu) == 0xFFu) return;  // (a refused genome)
    const FastxChunk ck = chunks[blockIdx.x];
    const int t = threadIdx.x;
    const uint32_t at = (uint32_t)t * 64u;
    const uint32_t have = at < ck.len ? (ck.len - at < 64u ? ck.len - at : 64u) : 0u;
    LaneMasks m;
    lane_masks(raw, ck.begin + at, have, m);
    uint32_t total_fn;
    const uint32_t ex = wg_scan_fn(lane_fn(m), sh, total_fn);
    const uint32_t s = fn_apply(ex, st.y & 3u);
    uint64_t hs;
    const uint64_t keep = lane_out(m, s, hs);
    const uint32_t mine = (uint32_t)__popcll(keep);
    // exclusive prefix sum of the lanes' byte counts: inside the wave by shuffles, across the four waves through LDS
    uint32_t inc = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t v = __shfl_up(inc, o);
        if ((t & 63) >= o) inc += v;
    }
    if ((t & 63) == 63) wsum[t >> 6] = inc;
    __syncthreads();
    uint32_t base = 0, total = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        if (w < (t >> 6)) base += wsum[w];
        total += wsum[w];
    }
    const uint64_t dst0 = genomes[ck.genome].off + (((uint64_t)(st.y >> 8) << 32) | st.x);
    const uint32_t A = (uint32_t)(dst0 & 15u);
    uint32_t pos = A + base + inc - mine;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int bit = 4 * k + b;
            if ((keep >> bit) & 1ull) stage[pos++] = ((hs >> bit) & 1ull) ? (uint8_t)'N' : (uint8_t)(m.w[k] >> (8 * b));
        }
    }
    __syncthreads();
    // LDS [A, A + total) -> out [dst0, dst0 + total): whole 16-byte groups as uint4, the ragged ends byte by byte (the
    // neighbouring chunks write the other bytes of those groups)
    uint8_t *base_out = out + (dst0 - A);
    const uint32_t end = A + total;
    const uint32_t g0 = A ? 1u : 0u, g1 = end >> 4;  // full groups [g0, g1)
    for (uint32_t gidx = g0 + t; gidx < g1; gidx += 256)
        reinterpret_cast<uint4 *>(base_out)[gidx] = reinterpret_cast<const uint4 *>(stage)[gidx];
    if (A && t < 16 && (uint32_t)t >= A && (uint32_t)t < end) base_out[t] = stage[t];
    if (g1 >= g0 && t >= 32 && t < 48) {  // (g1 < g0: everything lies inside the first group, written by the head lanes)
        const uint32_t x = (g1 << 4) + (uint32_t)(t - 32);
        if (x < end && x >= A) base_out[x] = stage[x];
    }
}

// what the decoded genome leaves of its region: 'N' (no k-mer starts there)
__global__ __launch_bounds__(256) void k_fastx_pad(const FastxGenome *__restrict__ genomes, const uint64_t *__restrict__ declen,
                                                    uint8_t *__restrict__ out)
{
    const FastxGenome g = genomes[blockIdx.x];
    const uint64_t b = g.off + declen[blockIdx.x], e = g.region_end;
    const uint64_t tid = (uint64_t)blockIdx.y * 256 + threadIdx.x, nthr = (uint64_t)gridDim.y * 256;
    const uint64_t b16 = (b + 15) & ~15ull, e16 = e & ~15ull;
    if (b16 >= e16) {
        for (uint64_t x = b + tid; x < e; x += nthr) out[x] = 'N';
        return;
    }
    for (uint64_t x = b + tid; x < b16; x += nthr) out[x] = 'N';
    const uint4 nn = make_uint4(0x4E4E4E4Eu, 0x4E4E4E4Eu, 0x4E4E4E4Eu, 0x4E4E4E4Eu);
    for (uint64_t x = b16 + 16 * tid; x < e16; x += 16 * nthr) *reinterpret_cast<uint4 *>(out + x) = nn;
    for (uint64_t x = e16 + tid; x < e; x += nthr) out[x] = 'N';
}

hipError_t preload_fastx_kernels()
{
    hipFuncAttributes fa;
    return hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(k_fastx_pad));
}

hipError_t launch_fastx_decode(hipStream_t st, const uint8_t *raw, const FastxChunk *chunks, uint32_t nchunks,
                               const FastxGenome *genomes, uint32_t ngenomes, uint4 *summ, uint2 *state, uint64_t *declen,
                               uint32_t *status, uint8_t *out)
{
    if (ngenomes == 0) return hipSuccess;
    if (nchunks) {
        hipLaunchKernelGGL(k_fastx_scan, dim3(nchunks), dim3(256), 0, st, raw, chunks, genomes, summ, status);
    }
    hipLaunchKernelGGL(k_fastx_offsets, dim3(ngenomes), dim3(256), 0, st, genomes, summ, status, state, declen);
    if (nchunks) {
        hipLaunchKernelGGL(k_fastx_compact, dim3(nchunks), dim3(256), 0, st, raw, chunks, genomes, state, out);
    }
    hipLaunchKernelGGL(k_fastx_pad, dim3(ngenomes, 16), dim3(256), 0, st, genomes, declen, out);
    return hipGetLastError();
}

}  // namespace dsh

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
// FASTQ (a genome that begins with '@') in strict four-line records takes the same three kernels with a simpler carry --
// the line index modulo 4, i.e. the newlines so far: additive -- and per-lane bit planes of that count; sequence lines
// are kept, a header's newline becomes the record's invalid byte.
// What does not keep its format's promise -- a first byte that is neither '>' nor '@', a FASTA line that begins with '+',
// FASTQ lines 4r that are not '@' headers or 4r + 2 that are not '+' lines, a sequence line that begins with '@' '>' '+',
// a record whose quality line is not as long as its sequence line (multi-line or cut-off records: kseq's record state
// decides those) -- raises the genome's status word:
// nothing of it is emitted (its region becomes all 'N': no k-mer) and the host parses that file itself.  HBM-bound
// byte work: no MFMA.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.h"

namespace dsh {

namespace {

// carries.  FASTA: the line that runs into a position is a HEADER line / a SEQUENCE line / a FRESH line starts here.
// FASTQ: the index of the line, modulo 4 (0 header, 1 sequence, 2 '+', 3 quality).  Transfer functions on (at most) four
// states, two bits per entry.
constexpr uint32_t FX_HDR = 0, FX_SEQ = 1, FX_FRESH = 2;
constexpr uint32_t kFnIdentity = 0u | (1u << 2) | (2u << 4) | (3u << 6);

__device__ __forceinline__ uint32_t fn_const(uint32_t c) { return c | (c << 2) | (c << 4) | (3u << 6); }
__device__ __forceinline__ uint32_t fn_apply(uint32_t f, uint32_t x) { return (f >> (2 * x)) & 3u; }
// first a, then b
__device__ __forceinline__ uint32_t fn_then(uint32_t a, uint32_t b)
{
    return fn_apply(b, fn_apply(a, 0)) | (fn_apply(b, fn_apply(a, 1)) << 2) | (fn_apply(b, fn_apply(a, 2)) << 4) |
           (fn_apply(b, fn_apply(a, 3)) << 6);
}
// x -> (x + k) mod 4
__device__ __forceinline__ uint32_t fn_add(uint32_t k) { return (k & 3u) | (((k + 1) & 3u) << 2) | (((k + 2) & 3u) << 4) | (((k + 3) & 3u) << 6); }

// bit k = byte k of w equals the byte replicated in pat
__device__ __forceinline__ uint32_t eq4(uint32_t w, uint32_t pat)
{
    const uint32_t v = w ^ pat;
    const uint32_t z = ~(((v & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | v | 0x7F7F7F7Fu);  // 0x80 where the byte is zero
    return ((z >> 7) * 0x01020408u) >> 24;
}

struct LaneMasks {
    uint64_t nl, nl_real, cr, hc, pl, at, valid;
    uint32_t w[16];
};

// the 64 bytes of this lane (absent ones -- behind the chunk's length -- read as '\n' for the FASTA carry: dropped, harmless)
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
    uint32_t nl[2] = {0, 0}, cr[2] = {0, 0}, gt[2] = {0, 0}, at[2] = {0, 0}, pl[2] = {0, 0};
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int h = k >> 3, s = 4 * (k & 7);
        nl[h] |= eq4(m.w[k], 0x0A0A0A0Au) << s;
        cr[h] |= eq4(m.w[k], 0x0D0D0D0Du) << s;
        gt[h] |= eq4(m.w[k], 0x3E3E3E3Eu) << s;
        at[h] |= eq4(m.w[k], 0x40404040u) << s;
        pl[h] |= eq4(m.w[k], 0x2B2B2B2Bu) << s;
    }
    const uint64_t valid = have >= 64 ? ~0ull : ((1ull << have) - 1);
    m.valid = valid;
    m.nl_real = (((uint64_t)nl[1] << 32) | nl[0]) & valid;
    m.nl = m.nl_real | ~valid;
    m.cr = (((uint64_t)cr[1] << 32) | cr[0]) & valid;
    m.at = (((uint64_t)at[1] << 32) | at[0]) & valid;
    m.hc = ((((uint64_t)gt[1] << 32) | gt[0]) & valid) | m.at;  // '>' '@' (the host parser takes both for a FASTA header)
    m.pl = (((uint64_t)pl[1] << 32) | pl[0]) & valid;
}

// ---- FASTA ----------------------------------------------------------------------------------------------------------------
// the lane's transfer function: with a newline inside, whatever came in is forgotten
__device__ __forceinline__ uint32_t lane_fn(const LaneMasks &m)
{
    if (m.nl == 0) return FX_HDR | (FX_SEQ << 2) | (((m.hc & 1) ? FX_HDR : FX_SEQ) << 4) | (3u << 6);
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

// ---- FASTQ (strict four-line records) ---------------------------------------------------------------------------------------
// A byte's line index modulo 4 = (carry + the newlines in front of it) mod 4, a newline counting to the line it ENDS: two
// bit planes by prefix-xor (the low plane is the parity of the newlines so far, the high plane flips where a newline
// arrives on odd parity).  q[j] = the bytes of this lane whose line index is j.
__device__ __forceinline__ uint64_t prefix_xor(uint64_t x)
{
    x ^= x << 1; x ^= x << 2; x ^= x << 4; x ^= x << 8; x ^= x << 16; x ^= x << 32;
    return x;
}
__device__ __forceinline__ void lane_line_index(uint64_t nl, uint32_t carry, uint64_t q[4])
{
    const uint64_t lo = prefix_xor(nl << 1);          // parity of the newlines in front of each byte
    const uint64_t hi = prefix_xor((nl & lo) << 1);   // ... of those that arrived on odd parity
    const uint64_t c0 = (carry & 1u) ? ~0ull : 0ull, c1 = (carry & 2u) ? ~0ull : 0ull;
    const uint64_t b0 = lo ^ c0, b1 = hi ^ c1 ^ (lo & c0);
    q[0] = ~b0 & ~b1, q[1] = b0 & ~b1, q[2] = ~b0 & b1, q[3] = b0 & b1;
}
// emitted: the sequence lines' bytes, and ONE invalid byte per record -- the newline that ends its header line
__device__ __forceinline__ uint64_t lane_out_fastq(const LaneMasks &m, uint32_t carry, uint64_t &as_n)
{
    uint64_t q[4];
    lane_line_index(m.nl_real, carry, q);
    as_n = m.nl_real & q[0];
    return (q[1] & ~m.nl_real & ~m.cr & m.valid) | as_n;
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

__device__ __forceinline__ uint32_t wave_sum(uint32_t v)
{
#pragma unroll
    for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

}  // namespace

// pass A: every chunk's transfer function and what it emits under each possible carry.
//   FASTA: a[s] = bytes emitted with carry s; the genome's status is raised here (its checks do not depend on the carry).
//   FASTQ: relative to a carry of 0, a[j] = bytes (no newline, no '\r') on lines of index j, b[j] = newlines that end a line
//   of index j, bad8 bit j / bit 4 + j = a line of index j is followed by a line that does not begin with '@' / '+', bit 8 + j
//   = ... by a line that begins with '@', '>' or '+' (k_fastx_offsets, which knows the carry, rotates them into place).
__global__ __launch_bounds__(256) void k_fastx_scan(const uint8_t *__restrict__ raw, const FastxChunk *__restrict__ chunks,
                                                     const FastxGenome *__restrict__ genomes, FastxSumm *__restrict__ summ,
                                                     uint32_t *__restrict__ status)
{
    __shared__ uint8_t sh[256];
    __shared__ uint32_t acc[9];
    const FastxChunk ck = chunks[blockIdx.x];
    const FastxGenome g = genomes[ck.genome];
    const int t = threadIdx.x;
    if (t < 9) acc[t] = 0;
    const uint32_t at = (uint32_t)t * 64u;
    const uint32_t have = at < ck.len ? (ck.len - at < 64u ? ck.len - at : 64u) : 0u;
    LaneMasks m;
    lane_masks(raw, ck.begin + at, have, m);
    const uint64_t next = ck.begin + at + 64;
    const bool has_next = have && next < g.off + g.rawlen;
    const uint32_t nb = has_next ? raw[next] : 0u;  // (the byte behind this lane: the next lane's, or the next chunk's, first)
    uint32_t total;
    if (g.fmt == 0) {
        // not plain FASTA: the genome does not begin with '>', or a line begins with '+'
        bool bad = false;
        if (have) {
            bad = (m.nl_real & ((m.pl >> 1) | ((uint64_t)(nb == '+') << 63))) != 0;
            if (ck.begin + at == g.off) bad = bad || (m.w[0] & 0xFFu) != '>';
        }
        if (bad) atomicOr(&status[ck.genome], 1u);
        const uint32_t ex = wg_scan_fn(lane_fn(m), sh, total);
        uint64_t hs;
        const uint32_t c[3] = {(uint32_t)__popcll(lane_out(m, FX_HDR, hs)), (uint32_t)__popcll(lane_out(m, FX_SEQ, hs)),
                               (uint32_t)__popcll(lane_out(m, FX_FRESH, hs))};
#pragma unroll
        for (uint32_t x = 0; x < 3; ++x) {
            const uint32_t s = fn_apply(ex, x);
            const uint32_t v = wave_sum(s == FX_HDR ? c[0] : (s == FX_SEQ ? c[1] : c[2]));
            if ((t & 63) == 0) atomicAdd(&acc[x], v);
        }
    } else {
        if (have && ck.begin + at == g.off && (m.w[0] & 0xFFu) != '@') atomicOr(&status[ck.genome], 1u);
        const uint32_t ex = wg_scan_fn(fn_add((uint32_t)__popcll(m.nl_real)), sh, total);
        uint64_t q[4];
        lane_line_index(m.nl_real, fn_apply(ex, 0), q);
        const uint64_t body = ~m.nl_real & ~m.cr & m.valid;
        // the first byte of the line behind a newline: the next bit, or the byte behind the lane; nothing behind the genome's end
        const uint64_t nx_exists = (m.valid >> 1) | ((uint64_t)has_next << 63);
        const uint64_t nx_at = (m.at >> 1) | ((uint64_t)(nb == '@') << 63), nx_pl = (m.pl >> 1) | ((uint64_t)(nb == '+') << 63);
        const uint64_t nx_mark = (m.hc >> 1) | nx_pl | ((uint64_t)(nb == '@' || nb == '>') << 63);  // '@' '>' '+': what the host parser reads as structure
        uint32_t bad8 = 0;
#pragma unroll
        for (uint32_t j = 0; j < 4; ++j) {
            const uint32_t n = wave_sum((uint32_t)__popcll(q[j] & body)), nn = wave_sum((uint32_t)__popcll(q[j] & m.nl_real));
            if ((t & 63) == 0) {
                atomicAdd(&acc[j], n);
                atomicAdd(&acc[4 + j], nn);
            }
            const uint64_t ends = m.nl_real & q[j] & nx_exists;
            if (ends & ~nx_at) bad8 |= 1u << j;
            if (ends & ~nx_pl) bad8 |= 16u << j;
            if (ends & nx_mark) bad8 |= 256u << j;
        }
        if (bad8) atomicOr(&acc[8], bad8);
    }
    __syncthreads();
    if (t == 0) {
        FastxSumm o;
        o.fn = total;
        o.bad8 = acc[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) o.a[j] = acc[j], o.b[j] = acc[4 + j];
        summ[blockIdx.x] = o;
    }
}

namespace {
// a chunk's emitted bytes under the concrete carry c (FASTQ also: whether it breaks the four-line pattern)
__device__ __forceinline__ uint32_t chunk_emits(const FastxSumm &s, uint32_t fmt, uint32_t c, bool &bad)
{
    if (fmt == 0) return s.a[c < 3 ? c : 2];
    // a line of raw index r has the index (r + c) & 3: sequence lines r = 1 - c, header newlines 0 - c; the line behind one of
    // index 3 must begin with '@', the one behind index 1 with '+', and the one behind index 0 -- the sequence line -- with
    // none of '@' '>' '+' (the host parser, like kseq, would read those as structure: a header, the separator)
    bad = bad || ((s.bad8 >> ((3u - c) & 3u)) & 1u) || ((s.bad8 >> (4u + ((1u - c) & 3u))) & 1u) || ((s.bad8 >> (8u + ((0u - c) & 3u))) & 1u);
    return s.a[(1u - c) & 3u] + s.b[(0u - c) & 3u];
}
}  // namespace

// pass B: one workgroup per genome -- every chunk's carry and where its output starts; the decoded length; the verdict
__global__ __launch_bounds__(256) void k_fastx_offsets(const FastxGenome *__restrict__ genomes, const FastxSumm *__restrict__ summ,
                                                        uint32_t *__restrict__ status, uint4 *__restrict__ state,
                                                        uint64_t *__restrict__ declen)
{
    __shared__ uint8_t sf[256];
    __shared__ uint64_t sn[256];
    __shared__ uint32_t sl[256];
    __shared__ uint32_t sbad;
    const FastxGenome g = genomes[blockIdx.x];
    const int t = threadIdx.x;
    if (t == 0) sbad = status[blockIdx.x];
    const uint32_t per = (g.nchunks + 255) / 256;
    const uint32_t c0 = (uint32_t)t * per, c1 = c0 + per < g.nchunks ? c0 + per : g.nchunks;
    // 1. the carry in front of every lane's run of chunks: prefix composition of the chunks' functions
    uint32_t f = kFnIdentity;
    for (uint32_t c = c0; c < c1; ++c) f = fn_then(f, summ[g.chunk0 + c].fn);
    sf[t] = (uint8_t)f;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {
        uint32_t v = sf[t];
        if (t >= d) v = fn_then(sf[t - d], v);
        __syncthreads();
        sf[t] = (uint8_t)v;
        __syncthreads();
    }
    const uint32_t start = g.fmt == 0 ? FX_FRESH : 0u;  // a genome begins with a fresh line / with line 0
    const uint32_t carry0 = t ? fn_apply(sf[t - 1], start) : start;
    // 2. with the carries known: what every chunk emits; prefix sum over the lanes' runs
    uint64_t mine = 0;
    uint32_t lines = 0;  // newlines of this lane's chunks (FASTQ: the line number of a byte = the newlines in front of it)
    bool bad = false;
    uint32_t carry = carry0;
    for (uint32_t c = c0; c < c1; ++c) {
        const FastxSumm s = summ[g.chunk0 + c];
        mine += chunk_emits(s, g.fmt, carry, bad);
        lines += s.b[0] + s.b[1] + s.b[2] + s.b[3];
        carry = fn_apply(s.fn, carry);
    }
    sn[t] = mine;
    sl[t] = lines;
    if (bad) atomicOr(&sbad, 1u);
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {
        uint64_t v = sn[t];
        uint32_t w = sl[t];
        if (t >= d) v += sn[t - d], w += sl[t - d];
        __syncthreads();
        sn[t] = v;
        sl[t] = w;
        __syncthreads();
    }
    if (sbad != 0) {  // the host will parse this one: nothing is emitted
        for (uint32_t c = t; c < g.nchunks; c += 256) state[g.chunk0 + c] = make_uint4(0xFFFFFFFFu, 0xFFu, 0u, 0u);
        if (t == 0) declen[blockIdx.x] = 0, status[blockIdx.x] = 1u;
        return;
    }
    uint64_t off = t ? sn[t - 1] : 0;
    lines = t ? sl[t - 1] : 0;
    carry = carry0;
    for (uint32_t c = c0; c < c1; ++c) {
        const FastxSumm s = summ[g.chunk0 + c];
        state[g.chunk0 + c] = make_uint4((uint32_t)off, carry | ((uint32_t)(off >> 32) << 8), lines, 0u);
        bool db = false;
        off += chunk_emits(s, g.fmt, carry, db);
        lines += s.b[0] + s.b[1] + s.b[2] + s.b[3];
        carry = fn_apply(s.fn, carry);
    }
    if (t == 255) declen[blockIdx.x] = sn[255];
}

// pass C: the chunk's emitted bytes, compacted through LDS, to out + (genome offset + the chunk's offset)
__global__ __launch_bounds__(256) void k_fastx_compact(const uint8_t *__restrict__ raw, const FastxChunk *__restrict__ chunks,
                                                        const FastxGenome *__restrict__ genomes, const uint4 *__restrict__ state,
                                                        uint8_t *__restrict__ out, unsigned long long *__restrict__ fingerprint)
{
    __shared__ uint8_t sh[256];
    __shared__ uint32_t wsum[4], lsum[4];
    __shared__ __attribute__((aligned(16))) uint8_t stage[kFastxChunk + 32];
    const uint4 st = state[blockIdx.x];
    if ((st.y & 0xFFu) == 0xFFu) return;  // (a refused genome)
    const FastxChunk ck = chunks[blockIdx.x];
    const FastxGenome g = genomes[ck.genome];
    const int t = threadIdx.x;
    const uint32_t at = (uint32_t)t * 64u;
    const uint32_t have = at < ck.len ? (ck.len - at < 64u ? ck.len - at : 64u) : 0u;
    LaneMasks m;
    lane_masks(raw, ck.begin + at, have, m);
    uint32_t total_fn;
    uint64_t hs, keep;
    if (g.fmt == 0) {
        const uint32_t ex = wg_scan_fn(lane_fn(m), sh, total_fn);
        keep = lane_out(m, fn_apply(ex, st.y & 3u), hs);
    } else {
        const uint32_t ex = wg_scan_fn(fn_add((uint32_t)__popcll(m.nl_real)), sh, total_fn);
        keep = lane_out_fastq(m, fn_apply(ex, st.y & 3u), hs);
        // The host parser ends a record's quality when it holds as many bytes as the sequence ('\r' not counted): a
        // quality line of ANOTHER length would shift its reading of every later line.  Per record that is a comparison of two
        // line lengths that may lie chunks apart; as ONE number per genome: F = sum over records r of H(r) x (sequence bytes -
        // quality bytes) with H a 64-bit mix of the record's number, in wrapping arithmetic -- linear in the bytes, so every
        // lane adds its own pieces of lines (line number = the newlines in front, known here) -- and F = 0 when every record
        // matches; a file in which one does not passes with probability 2^-64.  Summed into fingerprint[genome]; k_fastx_pad
        // refuses the genome AFTER the emission (and blanks it) if the sum is not zero.
        uint32_t lcnt = (uint32_t)__popcll(m.nl_real), linc = lcnt;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t v = __shfl_up(linc, o);
            if ((t & 63) >= o) linc += v;
        }
        if ((t & 63) == 63) lsum[t >> 6] = linc;
        __syncthreads();
        uint32_t line = st.z + linc - lcnt;  // the number of the line this lane's first byte lies on
        for (int w = 0; w < (t >> 6); ++w) line += lsum[w];
        const uint64_t body = ~m.nl_real & ~m.cr & m.valid;
        uint64_t nlb = m.nl_real, below = 0, F = 0;  // `below`: the bits in front of the current piece
        for (;;) {
            const uint64_t upto = nlb ? ((nlb & (0 - nlb)) - 1) : ~0ull;  // bits in front of the next newline (all: none left)
            const uint32_t len = (uint32_t)__popcll(body & upto & ~below);
            if (len && (line & 1u)) {  // a sequence line adds, a quality line takes away
                uint64_t h = (uint64_t)(line >> 2) + 0x9E3779B97F4A7C15ull;
                h = (h ^ (h >> 30)) * 0xBF58476D1CE4E5B9ull;
                h = (h ^ (h >> 27)) * 0x94D049BB133111EBull;
                h = (h ^ (h >> 31)) | 1ull;
                F += (line & 2u) ? (0 - h * len) : h * len;
            }
            if (!nlb) break;
            below = upto | (upto + 1);  // ... and the newline itself
            nlb &= nlb - 1;
            ++line;
        }
        uint32_t flo = (uint32_t)F, fhi = (uint32_t)(F >> 32);
        // (wrapping 64-bit sum over the wave: carries between the halves matter, so add whole words lane by lane)
#pragma unroll
        for (int o = 32; o; o >>= 1) {
            const uint64_t other = ((uint64_t)__shfl_xor(fhi, o) << 32) | __shfl_xor(flo, o);
            F += other;
            flo = (uint32_t)F, fhi = (uint32_t)(F >> 32);
        }
        if ((t & 63) == 0 && F) atomicAdd(&fingerprint[ck.genome], (unsigned long long)F);
    }
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
    const uint64_t dst0 = g.off + (((uint64_t)(st.y >> 8) << 32) | st.x);
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
// (FASTQ: a genome whose length fingerprint is not zero is refused here, after its emission, and blanked altogether)
__global__ __launch_bounds__(256) void k_fastx_pad(const FastxGenome *__restrict__ genomes, const uint64_t *__restrict__ declen,
                                                    const unsigned long long *__restrict__ fingerprint, uint32_t *__restrict__ status,
                                                    uint8_t *__restrict__ out)
{
    const FastxGenome g = genomes[blockIdx.x];
    const bool late = g.fmt != 0 && fingerprint[blockIdx.x] != 0;
    if (late && blockIdx.y == 0 && threadIdx.x == 0) status[blockIdx.x] = 1u;
    const uint64_t b = g.off + (late ? 0 : declen[blockIdx.x]), e = g.region_end;
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
                               const FastxGenome *genomes, uint32_t ngenomes, FastxSumm *summ, uint4 *state, uint64_t *declen,
                               uint32_t *status, unsigned long long *fingerprint, uint8_t *out)
{
    if (ngenomes == 0) return hipSuccess;
    if (nchunks) {
        hipLaunchKernelGGL(k_fastx_scan, dim3(nchunks), dim3(256), 0, st, raw, chunks, genomes, summ, status);
    }
    hipLaunchKernelGGL(k_fastx_offsets, dim3(ngenomes), dim3(256), 0, st, genomes, summ, status, state, declen);
    if (nchunks) {
        hipLaunchKernelGGL(k_fastx_compact, dim3(nchunks), dim3(256), 0, st, raw, chunks, genomes, state, out, fingerprint);
    }
    hipLaunchKernelGGL(k_fastx_pad, dim3(ngenomes, 16), dim3(256), 0, st, genomes, declen, fingerprint, status, out);
    return hipGetLastError();
}

}  // namespace dsh

// kernels.h -- host-callable launchers of the gfx950 kernels (internal to libdashing_hip.so).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "consts.h"

namespace dsh {

hipError_t launch_selfhist_card(hipStream_t st, const uint8_t *regs, uint64_t first, uint64_t n, int p, int estim,
                                int emax, int elow, uint32_t *hist, void *exc, uint8_t *excv, uint32_t *exc_n,
                                uint32_t *keys, uint8_t *tailhist);
hipError_t launch_card_from_hist(hipStream_t st, const uint32_t *hist, const uint32_t *keys, uint64_t first, uint64_t n, int p,
                                 int estim, double *card);
// position index of every 128-column block of the plane layout (perm == nullptr: identity) and the per-column side
// data of k_finalize in layout order (kernels_compare.hip, k_build_colindex)
// entries held inside a bucket's record: 7 (32-byte records) where a bucket holds ~4 entries and a list at most 256
inline int colindex_inline(int p, uint32_t E) { return p <= 12 && E <= 256 ? 7 : 3; }
struct ColIndexLaunch {
    const void *exc;          // per sketch (original index): listed positions, values, counts, keys, cardinalities, tail histograms
    const uint8_t *excv;
    const uint32_t *exc_n, *keys;
    const double *card;
    const uint8_t *tailhist;
    const uint32_t *perm;
    uint64_t ncols;
    int p;
    uint32_t nblocks, nbuckets, ent_stride, E;  // E = entries of a compact list row (emax + elow)
    uint32_t *rec;            // [nblocks][nbuckets][colindex_inline(p, E) + 1]
    uint32_t *ent;            // [nblocks][ent_stride]
    uint32_t *nS, *keyS;      // [Npad]
    double *cardS;            // [Npad]
    uint8_t *thS;             // [Npad][64]
    uint32_t *rl;             // [Npad][E]
};
hipError_t launch_build_colindex(hipStream_t st, const ColIndexLaunch &c);
hipError_t launch_transform(hipStream_t st, const uint8_t *regs, uint64_t n, int p, int vlo,
                            uint32_t P, uint32_t W, uint32_t Npad, uint32_t *planes,
                            const uint32_t *perm);
hipError_t launch_pair_counts(hipStream_t st, int kc, int cum_bytes, const uint32_t *planes,
                              uint32_t Npad, uint32_t Kpad, uint32_t W, uint32_t P,
                              const uint4 *tiles, const uint4 *items, uint32_t nitems, void *cum,
                              uint64_t nslots);

// what-if variant on the matrix cores (option "pair_mfma"; never the default: the north star excludes MFMA)
// the same arithmetic with the waves of a SIMD kept in one instruction class (512-thread workgroups, two items each)
hipError_t launch_pair_counts_lockstep(hipStream_t st, int kc, int cum_bytes, const uint32_t *planes, uint32_t Npad,
                                       uint32_t Kpad, uint32_t W, uint32_t P, const uint4 *tiles, const uint4 *items,
                                       uint32_t nitems, uint32_t nfrag, void *cum, uint64_t nslots);
hipError_t launch_pair_counts_mfma(hipStream_t st, int kc, int cum_bytes, const uint32_t *planes,
                                   uint32_t Npad, uint32_t Kpad, uint32_t W, uint32_t P,
                                   const uint4 *tiles, const uint4 *items, uint32_t nitems, void *cum,
                                   uint64_t nslots);

struct FinalizeLaunch {
    const void *cum;
    int cum_bytes;  // 2 or 4
    uint64_t cum_stride;  // pair slots of the whole band (distance between two planes of cum)
    uint64_t nslots;      // pair slots to finalize (a band, or a segment of it: cum/tiles point at its first tile)
    const uint4 *tiles;
    const uint32_t *perm;
    const uint64_t *rowoff = nullptr;  // row-sorted parts: out index = rowoff[layout position of the pair's row] + column offset
    int hist_bins, pbase, p, estim, result_type;  // hist_bins: max over the launch's tiles of (largest - smallest value + 1)
    double ksinv;
    // per column of the layout, in layout order (k_build_colindex): listed registers, key, cardinality, tail histogram,
    // compact list (position << 8 | value, E entries per column); the position index of the column blocks
    const uint32_t *nS, *keyS;
    const double *cardS;
    const uint8_t *thS;
    const uint32_t *rl;
    uint32_t E;
    const uint32_t *cidx_rec;
    const uint32_t *cidx_ent;
    uint32_t nbuckets, ent_stride;
    uint64_t n, ncols;  // collection size (output dimension); real columns of the plane matrix
    int rect, sorted_out, square;
    int knn = 0;          // band-wise nearest neighbours: out = V[band rows][knn_ld], out2 = Vt[columns][knn_rows]
    float *out2 = nullptr;
    uint64_t knn_ld = 0, knn_rows = 0;
    int stop = 0;  // profiling: k_finalize leaves after phase `stop`
    unsigned long long *phase_cyc = nullptr;  // profiling: per-phase cycle sums of the full kernel (8 x u64, device)
    uint64_t row_begin, row_end, col_begin, col_end, base_index;
    float *out;
    // part signalling (k_finalize_signal): the call's signal block (layout below), the generation value that marks a part
    // final, whether completion times are stamped (profiling).  nullptr: completion is marked by events between launches.
    uint32_t *sig = nullptr;
};
hipError_t launch_finalize(hipStream_t st, const FinalizeLaunch &f);
// the signal block of a call with parts, in 32-bit words: per part a flag (= the generation of the call that completed
// it), a count of finished tiles, the tiles it holds in all, a 64-bit wall-clock stamp of its completion; the stamp of the
// call's start; per tile of the band in flight a count of finished rows
constexpr uint32_t kSigMaxParts = 256;
constexpr uint32_t kSigPartFlag = 0, kSigPartCnt = kSigMaxParts, kSigPartTotal = 2 * kSigMaxParts,
                   kSigGen = 3 * kSigMaxParts, kSigStamp = kSigGen + 1,  // (uploaded with the totals: kSigMaxParts + 2 words)
                   kSigPartTime = 3 * kSigMaxParts + 2, kSigT0 = 5 * kSigMaxParts + 2, kSigTileCnt = 5 * kSigMaxParts + 4,
                   kSigWords = kSigTileCnt + 65536;
// (k_upload_segs) up to four uploads from page-locked staging in one launch
struct UploadSeg {
    uint32_t *dst;
    const uint32_t *src;
    uint64_t nwords;
};
struct UploadSegs {
    UploadSeg s[4];
    void add(uint32_t &n, void *dst, const void *src, size_t bytes)
    {
        if (!bytes) return;
        s[n].dst = (uint32_t *)dst, s[n].src = (const uint32_t *)src, s[n].nwords = (bytes + 3) / 4;
        ++n;
    }
};
hipError_t launch_upload_segs(hipStream_t st, const UploadSegs &u, uint32_t nseg);
hipError_t launch_wall_stamp(hipStream_t st, unsigned long long *out);
// (k_rows_place) the rows of one source's part in a round of the exchange: positions [pos0, pos0 + nrows) of its key order
struct PlaceEnt {
    const float *src;         // the source's staged buffer
    const uint32_t *order;    // its key order (original row of every position)
    const uint64_t *rowoff;   // where the row at a position starts in the buffer
    uint64_t pos0;
    uint32_t row0, nrows;     // the blocks [row0, row0 + nrows) of the launch
};
hipError_t launch_rows_place(hipStream_t st, const PlaceEnt *ent, uint32_t nent, uint32_t total_rows, float *out, uint64_t n);
// rows [pos0, pos1) of a row-sorted buffer (order[s] = original row, rowoff[s] = its offset) into the packed triangle
hipError_t launch_row_place(hipStream_t st, const float *src, float *out, const uint32_t *order, const uint64_t *rowoff,
                            uint64_t pos0, uint64_t pos1, uint64_t n);
hipError_t launch_unpermute_staged(hipStream_t st, const float *in, const uint32_t *inv,
                                   const int64_t *rowdelta, uint64_t n, float *out);
// destination-driven (coalesced writes, gathered reads through inv, the inverse of perm)
hipError_t launch_unpermute(hipStream_t st, const float *in, const uint32_t *inv, uint64_t n, float *out);

hipError_t launch_topk(hipStream_t st, const float *vals, uint64_t rows, uint64_t ncols,
                       uint64_t row0, uint64_t col0, int descending, uint32_t nn,
                       int exclude_self, uint32_t *idx_out, float *val_out);

// band-wise nearest neighbours (no n x n matrix): running lists idx/val[n][nn] by sketch index
hipError_t launch_knn_state_init(hipStream_t st, uint32_t *idx, float *val, uint64_t cnt, int descending);
hipError_t launch_topk_merge(hipStream_t st, const float *vals, uint64_t ld, int mode, uint64_t b0, uint64_t rows,
                             uint64_t ncols, const uint32_t *perm, int descending, uint32_t nn, uint32_t *st_idx,
                             float *st_val);

// in-order upload of a small page-locked host buffer by a kernel (no runtime copy on the ctx stream)
hipError_t launch_upload(hipStream_t st, void *dst, const void *src_pinned, size_t bytes);

// sketch path
struct SketchWork {
    uint64_t gbeg;   // absolute offset of the genome's first base in the device seq buffer
    uint64_t gend;   // one past its last base
    uint64_t start;  // absolute, 32-aligned offset of this workgroup's first base
    uint32_t nsub;   // number of 8192-base sub-chunks this workgroup walks
    uint32_t slot;   // row of the resident sketch matrix
};
hipError_t launch_sketch(hipStream_t st, const uint8_t *seq, const SketchWork *work,
                         uint32_t nwork, int k, int p, int canon, uint8_t *regs);

// FASTA text -> clean base stream on the device (kernels_fastx.hip).  A genome's raw file bytes lie at [off, off + rawlen)
// of the raw buffer (off 32-aligned) and are decoded to the same offset of the output buffer, the rest of the region up to
// region_end filled with 'N'; a chunk is what one workgroup takes (begin 32-aligned, len <= kFastxChunk).
constexpr uint32_t kFastxChunk = 16384;
struct FastxChunk {
    uint64_t begin;
    uint32_t len, genome;
};
struct FastxGenome {
    uint64_t off, rawlen, region_end;
    uint32_t chunk0, nchunks;
    uint32_t fmt, pad_;  // 0: FASTA (begins with '>'), 1: FASTQ in strict four-line records (begins with '@')
};
struct FastxSumm {  // what a chunk tells the per-genome scan (kernels_fastx.hip, k_fastx_scan)
    uint32_t fn, bad8;
    uint32_t a[4], b[4];
};
// summ [nchunks], state [nchunks] uint4, declen [ngenomes], status and fingerprint [ngenomes] (both zeroed by the caller;
// status != 0 afterwards: not what its format promises, nothing emitted)
hipError_t launch_fastx_decode(hipStream_t st, const uint8_t *raw, const FastxChunk *chunks, uint32_t nchunks,
                               const FastxGenome *genomes, uint32_t ngenomes, FastxSumm *summ, uint4 *state, uint64_t *declen,
                               uint32_t *status, unsigned long long *fingerprint, uint8_t *out);

// dsh_preload: load the code objects of the kernel translation units now (else: at the first launch from each)
hipError_t preload_compare_kernels();
hipError_t preload_fastx_kernels();
hipError_t preload_sketch_kernels();

// hipFuncAttributeMaxDynamicSharedMemorySize of a kernel, raised once per (kernel, device) instead of on every launch
// (ADVICE r4): remembers the largest size granted so far and only calls the runtime for a larger one.
hipError_t ensure_dynamic_lds(const void *kernel, size_t bytes);

}  // namespace dsh

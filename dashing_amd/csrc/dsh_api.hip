// dsh_api.hip -- the C-ABI of libdashing_hip.so (include/dashing_hip.h) over the gfx950 kernels.
// No CPU fallback lives here: without a HIP device dsh_create fails with DSH_ENODEV.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <rccl/rccl.h>  // types only: the library is dlopen'ed at dsh_comm_init (single-GPU users never load it)

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/dashing_hip.h"
#include "kernels.h"

using namespace dsh;

struct DevBuf {
    void *ptr = nullptr;
    size_t cap = 0;
    hipError_t ensure(size_t bytes)
    {
        if (bytes <= cap) return hipSuccess;
        if (ptr) (void)hipFree(ptr);
        ptr = nullptr;
        cap = 0;
        hipError_t e = hipMalloc(&ptr, bytes);
        if (e == hipSuccess) cap = bytes;
        return e;
    }
    void release()
    {
        if (ptr) (void)hipFree(ptr);
        ptr = nullptr;
        cap = 0;
    }
};

// page-locked host staging: a hipMemcpyAsync from it is a true asynchronous DMA, so the call that filled it
// may return before the copy has run (the event says when it may be rewritten)
struct PinBuf {
    void *ptr = nullptr;
    size_t cap = 0;
    hipError_t ensure(size_t bytes)
    {
        if (bytes <= cap) return hipSuccess;
        if (ptr) (void)hipHostFree(ptr);
        ptr = nullptr;
        cap = 0;
        const size_t want = bytes + bytes / 2;
        hipError_t e = hipHostMalloc(&ptr, want, hipHostMallocDefault);
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release()
    {
        if (ptr) (void)hipHostFree(ptr);
        ptr = nullptr;
        cap = 0;
    }
};

struct dsh_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;
    // resident sketch matrix
    DevBuf regs_own;
    const uint8_t *regs = nullptr;  // device
    uint64_t n = 0;
    int p = 0;
    bool have_sketches = false;
    // derived state
    bool planes_valid = false;
    int card_estim = -1;
    uint64_t card_from = 0;             // the per-sketch pass (cardinalities, lists, keys) covers the sketches [card_from, n)
    DevBuf card, planes, cum, tiles, items, outbuf, seqbuf, workbuf, exc, excv, exc_n, keys, perm, tailhist;
    // copy-out pipeline of dsh_dist_rows_async: results alternate between two device buffers; the copy of call b to the
    // host runs on its own stream while the kernels of call b+1 fill the other buffer
    hipStream_t copy_stream = nullptr;
    hipStream_t aux_stream = nullptr;   // prepare(): the column index is built next to the bit-plane transform
    hipEvent_t ev_aux_fork = nullptr, ev_aux_join = nullptr;
    bool aux_join_pending = false;
    DevBuf outbuf2[2];
    hipEvent_t ev_filled[2] = {nullptr, nullptr};  // kernels of the call that filled outbuf2[b] done (recorded on stream)
    hipEvent_t ev_drained[2] = {nullptr, nullptr}; // copy out of outbuf2[b] done (recorded on copy_stream)
    bool drained_pending[2] = {false, false};
    unsigned out_turn = 0;
    std::vector<hipEvent_t> tickets;    // dsh_event_record ring: slot t % 64 holds {join on the ctx stream, mark on the copy stream}
    uint64_t ticket_next = 0;
    // multi-GPU exchange (dsh_comm_*): an RCCL communicator over the ranks' contexts; all traffic on `stream`
    ncclComm_t comm = nullptr;
    int comm_rank = 0, comm_world = 1;
    DevBuf gather_full, gather_local;   // dsh_dist_collect: the assembled matrix on the destination rank / this rank's span
    DevBuf hist;                        // [n][64] per-sketch register histograms (k_selfhist_card -> k_card_from_hist)
    hipEvent_t ev_keys = nullptr;       // the keys have reached the host
    DevBuf cidx_off, cidx_ent;          // position index of the column blocks of the current layout (k_build_colindex)
    uint32_t nbuckets = 0, ent_stride = 0;
    // column layout of the cached plane matrix.  0: identity over all n sketches.  1: the sub-collection
    // {lay_rb .. n-1} -- the rows [lay_rb, lay_re) first, then the rows [lay_re, n), each part in key order
    // (lay_rb = 0, lay_re = n: the whole collection sorted, what full-triangle calls and the shard path use)
    int planes_sorted = 0;
    uint64_t lay_rb = 0, lay_re = 0;
    // a row-range layout may keep its wanted rows in several PARTS (consecutive sub-ranges, each key-ordered on its
    // own, each a whole number of 128-row tile rows except the last): a part's tiles are finished -- and its span of
    // the packed matrix complete -- before the next part starts, so it can travel while the rest is computed
    std::vector<uint64_t> lay_parts;    // row boundaries, lay_parts.front() = lay_rb, .back() = lay_re
    std::vector<hipEvent_t> ev_part;    // part q complete (recorded on the ctx stream by the last call with parts)
    uint32_t parts_done = 0;            // parts of the last dsh_dist_rows_parts_device_async call
    uint64_t ncols = 0;                 // real columns (sketches) of the plane matrix; Npad = ncols padded to 128
    PinBuf pin_keys;                    // host copy of the per-sketch keys (valid while the per-sketch pass is), page-locked:
    const uint32_t *hk32 = nullptr;     // the copy is a direct DMA and the host only waits for ev_keys
    bool hk32_valid = false;
    hipEvent_t ev_perm = nullptr;       // upload of pin_perm done (it is rewritten by the next layout)
    bool perm_in_flight = false;
    std::vector<uint32_t> sort_a, sort_b, sort_keys;  // scratch of the column sort
    std::vector<uint2> tile_chunks;     // scratch of run_pairs: chunk range per tile
    double host_layout_us = 0, host_lists_us = 0, host_keys_wait_us = 0;  // host time of the last call (dsh_get_info)
    std::vector<uint32_t> hperm;        // plane-matrix column -> sketch
    uint32_t *pin_perm = nullptr;       // page-locked copy of hperm: its upload is then truly asynchronous
    size_t pin_perm_cap = 0;
    std::vector<uint8_t> blk_T, blk_lo, blk_L, blk_hi; // per 128-column block: max high threshold, min register value, min low threshold, max register value
    std::vector<uint4> hitems;
    PinBuf pin_work;                    // sketch work list of the call in flight
    hipEvent_t ev_work = nullptr;
    bool work_in_flight = false;
    PinBuf pin_lists;                   // tiles then items of the call in flight
    hipEvent_t ev_lists = nullptr;      // recorded after their upload; waited on before they are rewritten
    bool lists_in_flight = false;
    uint32_t Npad = 0, W = 0, P = 0, Kpad = 0;
    int vlo = 0, vhi = 0, pbase = 0;  // register value range of the columns; plane pl is the threshold pbase + 1 + pl
    int emax = 0, elow = 0, cum_bytes = 4;
    std::vector<uint4> htiles;
    // options
    int kc = 16;      // k-rows per LDS stage in effect (set by prepare from kc_opt)
    int kc_opt = 0;   // 0 auto: 32 where a plane is at least that long (p >= 10), else 16 (profiles/r3f/lockstep_ab.jsonl)
    int emax_opt = -1;  // cap of the listed upper tail; -1: auto_list_cap(p, true)
    int elow_opt = -1;  // cap of the listed lower tail; -1: auto_list_cap(p, false)
    std::vector<uint32_t> tile_rank;  // scratch of run_pairs
    int finalize_rowmajor = 1;        // k_finalize walks every segment's tiles in row-major order (option, A/B only)
    size_t last_bands = 0;            // tile-kernel launches groups (bands) of the last dist call
    uint64_t cum_budget = 8ull << 30;  // scratch for C(v) per pair slot: larger jobs run in bands (2 -> 8 GiB: -1.5 % at 100 000 x p=10)
    int xcd_swizzle = 1;
    int sort_mode = -1;  // -1 auto (key-ordered columns for triangle calls of >= range_sort_min_rows rows), 0 never
    int range_sort_min_rows = 1024;  // smaller row ranges keep the cached identity layout (a rebuild costs more than it saves)
    int assembler_permille = 21;  // the un-permute (0.42 ms) on rank 0 of a 19.9 ms pass (profiles/r1k)
    int unperm_gather = 1;  // un-permute driven from the destination (coalesced writes) instead of the source
    uint64_t knn_square_budget = (uint64_t)96 << 30;  // all-vs-all kNN keeps an n x n float matrix in HBM up to this size
    double shard_c0 = 5.0;  // finalize work of a tile in plane-equivalents (shard balancing)
    int ls_sort_items = 1;
    int ls_item_chunks = 64;  // lockstep kernel: work items of at most about this many K-chunks (whole planes)
    // k_pair_counts_ls (512-thread workgroups, AND and BCNT batches phase-locked across the waves of a SIMD): -1 auto
    // = 1 = wherever a plane is at least one chunk (W >= kc), 0 never (the free-running k_pair_counts)
    int pair_lockstep = -1;
    int pair_mfma = 0;  // WHAT-IF only: 1 = the AND+popcount tile kernel on the matrix cores (never the default)
    int finalize_stop = 0;  // profiling only: k_finalize leaves after phase 1..4 (results are then meaningless)
    int nsplit = 0;  // plane-range splits per tile; 0 = auto (aim at >= 16 items per workgroup slot)
    // profiling
    bool profiling = false;
    double pair_ms = 0, fin_ms = 0, prep_ms = 0;
    uint32_t pair_launches = 0;
    std::vector<hipEvent_t> ev_pool;
    size_t ev_used = 0;
};

namespace {

int fail(dsh_ctx *c, int code, const char *fmt, ...)
{
    if (c) {
        char buf[512];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof buf, fmt, ap);
        va_end(ap);
        c->err = buf;
    }
    return code;
}

#define HIPCHK(c, expr)                                                                   \
    do {                                                                                  \
        hipError_t e_ = (expr);                                                           \
        if (e_ != hipSuccess)                                                             \
            return fail((c), e_ == hipErrorOutOfMemory ? DSH_ENOMEM : DSH_EIO, "%s: %s",  \
                        #expr, hipGetErrorString(e_));                                    \
    } while (0)

// RCCL, loaded on first use.  The library must be the one built against the HIP runtime this file links to (the
// streams handed to it are ours): librccl.so.1 through this library's RUNPATH (/opt/rocm/lib), or DSH_RCCL_LIB.
struct Rccl {
    void *h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::string err;
};

Rccl *rccl()
{
    static Rccl r;
    static bool tried = false;
    if (tried) return &r;
    tried = true;
    const char *names[3] = {std::getenv("DSH_RCCL_LIB"), "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
    for (const char *nm : names) {
        if (!nm || !*nm) continue;
        r.h = dlopen(nm, RTLD_NOW | RTLD_LOCAL);
        if (r.h) break;
        r.err = dlerror();
    }
    if (!r.h) return &r;
    auto sym = [&](const char *name) -> void * {
        void *p = dlsym(r.h, name);
        if (!p) r.err = std::string("librccl: missing symbol ") + name;
        return p;
    };
    r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
    r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
    r.GroupStart = (decltype(r.GroupStart))sym("ncclGroupStart");
    r.GroupEnd = (decltype(r.GroupEnd))sym("ncclGroupEnd");
    r.Send = (decltype(r.Send))sym("ncclSend");
    r.Recv = (decltype(r.Recv))sym("ncclRecv");
    r.AllGather = (decltype(r.AllGather))sym("ncclAllGather");
    r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
    if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.GroupStart || !r.GroupEnd || !r.Send || !r.Recv ||
        !r.AllGather || !r.GetErrorString) {
        dlclose(r.h);
        r.h = nullptr;
    }
    return &r;
}

#define NCCLCHK(c, expr)                                                                                   \
    do {                                                                                                   \
        ncclResult_t r_ = (expr);                                                                          \
        if (r_ != ncclSuccess) return fail((c), DSH_EIO, "%s: %s", #expr, rccl()->GetErrorString(r_));     \
    } while (0)

int bind(dsh_ctx *c)
{
    HIPCHK(c, hipSetDevice(c->device));
    return DSH_OK;
}

hipEvent_t next_event(dsh_ctx *c)
{
    if (c->ev_used == c->ev_pool.size()) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return nullptr;
        c->ev_pool.push_back(e);
    }
    return c->ev_pool[c->ev_used++];
}

// slots [first, first + cnt) inside [0, total) -- written so that first + cnt cannot wrap
bool slots_ok(uint64_t first, uint64_t cnt, uint64_t total) { return first <= total && cnt <= total - first; }

bool use_lockstep(const dsh_ctx *c)
{
    // wherever a plane is at least one chunk (p >= 9): since the kernel needs one barrier per k-row it beats the
    // free-running one at every precision (profiles/r3f/lockstep_ab.jsonl: -5 % at p = 10 ... -17 % at p = 16)
    return !(c->pair_mfma || c->kc > 32 || c->W < (uint32_t)c->kc || c->pair_lockstep == 0);
}

bool whole_sorted(const dsh_ctx *c)
{
    return c->planes_valid && c->planes_sorted && c->lay_rb == 0 && c->lay_re == c->n;
}

void invalidate(dsh_ctx *c)
{
    c->planes_valid = false;
    c->card_estim = -1;
    c->hk32_valid = false;
}

// fields of a per-sketch key (k_selfhist_card): bad << 31 | hi << 18 | T << 12 | L << 6 | lo
inline int key_lo(uint32_t k) { return (int)(k & 63u); }
inline int key_L(uint32_t k) { return (int)((k >> 6) & 63u); }
inline int key_T(uint32_t k) { return (int)((k >> 12) & 63u); }
inline int key_hi(uint32_t k) { return (int)((k >> 18) & 63u); }

// default caps of the two listed tails (profiles/r3f/list_cap_sweep.jsonl).  A pair shares cap^2 / 2^p listed positions
// per side, each one an LDS atomic in k_finalize; every halving of the upper tail (a plane saved) costs twice the
// entries, while the lower tail of the register law falls off double-exponentially: listing ~200 registers removes
// the one or two nearly empty planes at the bottom.  2^p / 32 entries per side = one shared position per pair and
// side on average: 32/32 at p = 10, 128/128 at p = 12, the caps 255/200 from p = 13.
int auto_list_cap(int p, bool upper)
{
    const uint64_t m = 1ull << p;
    return (int)std::min<uint64_t>(upper ? kMaxListSide : 200, m >> 5);
}

// dense plane range of the tile (ti, tj): C(v) is needed for v in (max(larger of the two minima, smaller of the two
// low thresholds), larger of the two high thresholds] -- below that every C(v) is 0 or comes from the low-list join
void tile_planes(const dsh_ctx *c, uint32_t ti, uint32_t tj, int &pb, int &pe)
{
    const int lo_t = std::max<int>(std::max<int>(c->blk_lo[ti], c->blk_lo[tj]), std::min<int>(c->blk_L[ti], c->blk_L[tj]));
    const int T_t = std::max<int>(c->blk_T[ti], c->blk_T[tj]);
    pb = std::max(0, lo_t - c->pbase);
    pe = std::max(pb, T_t - c->pbase);
}

// cardinalities + thresholds/exception lists + planes for the current sketch matrix.
// want_sorted: lay the plane-matrix columns out in (threshold, min value) order so that the
// 128-column blocks are homogeneous and every tile can use its own narrow plane range.
// rows [rb, re) cut into nparts consecutive parts of about equal pair counts, every cut a whole number of 128-row
// tile rows after rb (fewer parts if the range has fewer tile rows); out gets the boundaries
void range_parts(uint64_t n, uint64_t rb, uint64_t re, uint32_t nparts, std::vector<uint64_t> &out)
{
    out.assign(1, rb);
    if (re > n) re = n;
    const uint64_t total = dsh_tri_span(n, rb, re);
    for (uint32_t q = 1; q < nparts && out.back() < re; ++q) {
        const long double target = (long double)total * q / nparts;
        uint64_t best = out.back() + kTile;
        for (uint64_t cand = out.back() + kTile; cand < re; cand += kTile) {
            best = cand;
            if ((long double)dsh_tri_span(n, rb, cand) >= target) break;
        }
        if (best >= re) break;
        out.push_back(best);
    }
    if (out.back() != re) out.push_back(re);
}

int prepare(dsh_ctx *c, int estim, int want_sorted, bool card_only = false, uint64_t want_rb = 0,
            uint64_t want_re = ~0ull, uint32_t nparts = 1)
{
    if (!c->have_sketches) return fail(c, DSH_ESTATE, "no sketches loaded");
    if (estim < 0 || estim > 2) return fail(c, DSH_EINVAL, "bad estimator %d", estim);
    if (want_sorted < 0) {  // "whatever is cached"
        want_sorted = c->planes_sorted;
        want_rb = c->lay_rb;
        want_re = c->lay_re;
    }
    if (want_re > c->n) want_re = c->n;
    if (!want_sorted) want_rb = 0, want_re = c->n;
    if (want_rb > want_re) want_rb = want_re;
    const uint64_t n = c->n;
    std::vector<uint64_t> parts;
    if (want_sorted) range_parts(n, want_rb, want_re, std::max<uint32_t>(nparts, 1), parts);
    const int emax_new = c->emax_opt >= 0 ? std::min<int>(c->emax_opt, (int)kMaxListSide) : auto_list_cap(c->p, true);
    const int elow_new = c->elow_opt >= 0 ? std::min<int>(c->elow_opt, (int)kMaxListSide) : auto_list_cap(c->p, false);
    if (emax_new != c->emax || elow_new != c->elow) {  // thresholds and lists (hence planes) depend on them
        c->planes_valid = false;
        c->card_estim = -1;
    }
    c->emax = emax_new;
    c->elow = elow_new;
    const bool same_layout = c->planes_valid && c->planes_sorted == want_sorted &&
                             (!want_sorted || (c->lay_rb == want_rb && c->lay_re == want_re && c->lay_parts == parts));
    // sketches the per-sketch pass has to cover: a row range of the triangle never looks at the sketches before it
    const uint64_t need_from = (card_only || !want_sorted) ? 0 : want_rb;
    const bool have_pass = c->card_estim == estim && c->card_from <= need_from;
    if (have_pass && (card_only || same_layout)) return DSH_OK;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (c->profiling) {
        e0 = next_event(c);
        e1 = next_event(c);
        if (e0) (void)hipEventRecord(e0, c->stream);
    }
    // the per-sketch pass depends on (registers, estimator, emax) only: a new column layout reuses it
    if (!have_pass) {
        HIPCHK(c, c->card.ensure(std::max<uint64_t>(n, 1) * sizeof(double)));
        HIPCHK(c, c->exc.ensure(std::max<uint64_t>(n, 1) * kListCap * (c->p <= 15 ? 2 : 4)));
        HIPCHK(c, c->excv.ensure(std::max<uint64_t>(n, 1) * kListCap));
        HIPCHK(c, c->exc_n.ensure(std::max<uint64_t>(n, 1) * sizeof(uint32_t)));
        HIPCHK(c, c->keys.ensure(std::max<uint64_t>(n, 1) * sizeof(uint32_t)));
        HIPCHK(c, c->tailhist.ensure(std::max<uint64_t>(n, 1) * 64));
        HIPCHK(c, c->hist.ensure(std::max<uint64_t>(n, 1) * 64 * sizeof(uint32_t)));
        HIPCHK(c, launch_selfhist_card(c->stream, c->regs, need_from, n, c->p, estim, c->emax, c->elow,
                                       (uint32_t *)c->hist.ptr, c->exc.ptr, (uint8_t *)c->excv.ptr,
                                       (uint32_t *)c->exc_n.ptr, (uint32_t *)c->keys.ptr,
                                       (uint8_t *)c->tailhist.ptr));
        // the keys travel to the host right behind the per-sketch pass (the column order is made there); the
        // cardinalities follow on the stream while the host sorts
        HIPCHK(c, c->pin_keys.ensure(std::max<uint64_t>(n, 1) * sizeof(uint32_t)));
        c->hk32 = (const uint32_t *)c->pin_keys.ptr;
        if (n > need_from)
            HIPCHK(c, hipMemcpyAsync((uint32_t *)c->pin_keys.ptr + need_from, (const uint32_t *)c->keys.ptr + need_from,
                                     (n - need_from) * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
        if (!c->ev_keys) HIPCHK(c, hipEventCreateWithFlags(&c->ev_keys, hipEventDisableTiming));
        HIPCHK(c, hipEventRecord(c->ev_keys, c->stream));
        HIPCHK(c, launch_card_from_hist(c->stream, (const uint32_t *)c->hist.ptr, (const uint32_t *)c->keys.ptr, need_from, n,
                                        c->p, estim, (double *)c->card.ptr));
        c->card_estim = estim;
        c->card_from = need_from;
        c->hk32_valid = false;
    }
    if (card_only) {  // a cardinality query never builds planes (and leaves stale ones marked so)
        if (e0 && e1) {
            (void)hipEventRecord(e1, c->stream);
            (void)hipEventSynchronize(e1);
            float ms = 0;
            (void)hipEventElapsedTime(&ms, e0, e1);
            c->prep_ms += ms;
        }
        return DSH_OK;
    }
    if (!same_layout) {
        if (c->p > kMaxPCompare)
            return fail(c, DSH_EINVAL, "the compare path takes p <= %d (p=%d: sketching and cardinalities only)", kMaxPCompare, c->p);
        int vr[3] = {63, 0, 0};  // min register value anywhere, max value, max threshold
        // the keys are downloaded once per per-sketch pass: a later layout (next row block) needs no
        // device round trip and so does not wait for the work still queued on the stream
        const auto t_h0 = std::chrono::steady_clock::now();
        if (!c->hk32_valid) {
            HIPCHK(c, hipEventSynchronize(c->ev_keys));
            c->hk32_valid = true;
        }
        const auto t_h1 = std::chrono::steady_clock::now();
        c->host_keys_wait_us = std::chrono::duration<double, std::micro>(t_h1 - t_h0).count();
        const uint32_t *k32 = c->hk32;
        for (uint64_t i = c->card_from; i < n; ++i)  // (the sketches the per-sketch pass covered)
            if (k32[i] & 0x80000000u)
                return fail(c, DSH_EINVAL, "sketch %llu holds a register value above %d (= 64 - p + 1): not an HLL of precision %d (corrupt or foreign .hll?)",
                            (unsigned long long)i, 64 - c->p + 1, c->p);
        // the plane matrix holds the sketches col0 .. n-1 (a row range [rb,re) of the triangle never looks at
        // sketches before rb); value range and thresholds are taken over those only
        const uint64_t col0 = want_sorted ? want_rb : 0;
        const uint64_t ncols = n - col0;
        for (uint64_t i = col0; i < n; ++i) {
            const uint32_t key = k32[i];
            vr[0] = std::min<int>(vr[0], key_lo(key));
            vr[1] = std::max<int>(vr[1], key_hi(key));
            vr[2] = std::max<int>(vr[2], key_T(key));
        }
        if (ncols == 0) vr[0] = vr[1] = vr[2] = 0;
        c->vlo = vr[0];
        c->vhi = vr[1];
        c->cum_bytes = c->p <= 15 ? 2 : 4;
        const uint64_t m = 1ull << c->p;
        c->W = (uint32_t)std::max<uint64_t>(1, m / 32);
        c->kc = c->kc_opt ? c->kc_opt : (c->W >= 32 ? 32 : 16);
        c->ncols = ncols;
        c->Npad = (uint32_t)((ncols + kTile - 1) / kTile * kTile);
        // column order: identity, or a counting sort by (threshold, min value, max value): the first
        // two make the 128-column blocks need few planes, the third keeps the 64 pairs of a finalize
        // wave alike in their largest register, i.e. in the trip count of the estimator's loops.
        // With a row range the wanted rows [rb,re) come first (their tile rows are the only ones computed),
        // then the later rows; each part is key-ordered on its own.
        c->hperm.resize(ncols);
        if (want_sorted) {
            auto skey = [&](uint64_t i) -> uint32_t {  // 6 bits each: high threshold, low threshold, max value
                const uint32_t key = k32[i];
                return ((uint32_t)key_T(key) << 12) | ((uint32_t)key_L(key) << 6) | (uint32_t)key_hi(key);
            };
            // stable LSD radix sort, two 9-bit digits (a single 2^18-bucket counting sort spends ~0.1 ms clearing
            // and scanning its counters); the host sits between the per-sketch pass and the transform, so this is
            // on the critical path of every layout: scratch is kept on the context
            std::vector<uint32_t> &a = c->sort_a, &b = c->sort_b, &keys = c->sort_keys;
            keys.resize(n);
            for (uint64_t i = col0; i < n; ++i) keys[i] = skey(i);
            auto sort_part = [&](uint64_t lo, uint64_t hi, uint32_t *dst) {
                const uint64_t cnt_ = hi - lo;
                if (a.size() < cnt_) a.resize(cnt_);
                uint32_t cnt[513];
                std::memset(cnt, 0, sizeof cnt);
                for (uint64_t i = 0; i < cnt_; ++i) cnt[(keys[lo + i] & 511u) + 1u]++;
                for (int k = 1; k < 513; ++k) cnt[k] += cnt[k - 1];
                for (uint64_t i = 0; i < cnt_; ++i) a[cnt[keys[lo + i] & 511u]++] = (uint32_t)(lo + i);
                std::memset(cnt, 0, sizeof cnt);
                for (uint64_t i = 0; i < cnt_; ++i) cnt[((keys[a[i]] >> 9) & 511u) + 1u]++;
                for (int k = 1; k < 513; ++k) cnt[k] += cnt[k - 1];
                for (uint64_t i = 0; i < cnt_; ++i) dst[cnt[(keys[a[i]] >> 9) & 511u]++] = a[i];
                (void)b;
            };
            for (size_t q = 0; q + 1 < parts.size(); ++q) sort_part(parts[q], parts[q + 1], c->hperm.data() + (parts[q] - want_rb));
            sort_part(want_re, n, c->hperm.data() + (want_re - want_rb));
            // perm, then (whole collection only) its inverse for the un-permute of the shard path
            const bool whole = want_rb == 0 && want_re == n;
            const uint64_t nperm = whole ? 2 * n : ncols;
            HIPCHK(c, c->perm.ensure(std::max<uint64_t>(nperm, 1) * sizeof(uint32_t)));
            if (whole) {
                c->hperm.resize(2 * n);
                for (uint64_t s = 0; s < n; ++s) c->hperm[n + c->hperm[s]] = (uint32_t)s;
            }
            if (nperm) {
                if (c->perm_in_flight) {  // the previous layout's upload from pin_perm
                    HIPCHK(c, hipEventSynchronize(c->ev_perm));
                    c->perm_in_flight = false;
                }
                if (nperm > c->pin_perm_cap) {
                    if (c->pin_perm) (void)hipHostFree(c->pin_perm);
                    c->pin_perm = nullptr;
                    c->pin_perm_cap = 0;
                    HIPCHK(c, hipHostMalloc((void **)&c->pin_perm, nperm * sizeof(uint32_t), hipHostMallocDefault));
                    c->pin_perm_cap = nperm;
                }
                std::memcpy(c->pin_perm, c->hperm.data(), nperm * sizeof(uint32_t));
                HIPCHK(c, launch_upload(c->stream, c->perm.ptr, c->pin_perm, nperm * sizeof(uint32_t)));
                if (!c->ev_perm) HIPCHK(c, hipEventCreateWithFlags(&c->ev_perm, hipEventDisableTiming));
                HIPCHK(c, hipEventRecord(c->ev_perm, c->stream));
                c->perm_in_flight = true;
            }
        } else {
            for (uint64_t i = 0; i < n; ++i) c->hperm[i] = (uint32_t)i;
        }
        const uint32_t NT = c->Npad / kTile;
        c->blk_T.assign(NT, 0);
        c->blk_lo.assign(NT, 255);
        c->blk_L.assign(NT, 255);
        c->blk_hi.assign(NT, 0);
        int pbase = vr[2];
        for (uint64_t s = 0; s < ncols; ++s) {
            const uint32_t key = k32[c->hperm[s]];
            const uint32_t b = (uint32_t)(s / kTile);
            c->blk_T[b] = std::max<uint8_t>(c->blk_T[b], (uint8_t)key_T(key));
            c->blk_lo[b] = std::min<uint8_t>(c->blk_lo[b], (uint8_t)key_lo(key));
            c->blk_L[b] = std::min<uint8_t>(c->blk_L[b], (uint8_t)key_L(key));
            c->blk_hi[b] = std::max<uint8_t>(c->blk_hi[b], (uint8_t)key_hi(key));
            pbase = std::min<int>(pbase, key_L(key));
        }
        // dense planes cover v in (pbase, Tmax]: below the smallest low threshold every C(v) comes from the list join
        c->pbase = pbase;
        c->P = (uint32_t)(vr[2] - pbase);
        // position index of every column block (the list joins of k_finalize): 79 workgroups at C3 -- built on a
        // second stream next to the bit-plane transform, which fills the chip on its own; both only need the
        // per-sketch pass and the permutation, the tile kernels wait for both
        c->nbuckets = (uint32_t)std::min<uint64_t>(2 * m, kMaxBuckets);  // (position group, upper | lower tail)
        c->ent_stride = std::max<uint32_t>(1, kTile * (uint32_t)(c->emax + c->elow));
        HIPCHK(c, c->cidx_off.ensure(std::max<size_t>(NT, 1) * (c->nbuckets + 2) * sizeof(uint16_t)));
        HIPCHK(c, c->cidx_ent.ensure(std::max<size_t>(NT, 1) * c->ent_stride * sizeof(uint32_t)));
        c->host_layout_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_h1).count();
        HIPCHK(c, hipEventRecord(c->ev_aux_fork, c->stream));
        HIPCHK(c, hipStreamWaitEvent(c->aux_stream, c->ev_aux_fork, 0));
        HIPCHK(c, launch_build_colindex(c->aux_stream, c->exc.ptr, (const uint8_t *)c->excv.ptr, (const uint32_t *)c->exc_n.ptr,
                                        (const uint32_t *)c->keys.ptr, want_sorted ? (const uint32_t *)c->perm.ptr : nullptr, ncols,
                                        c->p, NT, c->nbuckets, c->ent_stride, (uint16_t *)c->cidx_off.ptr, (uint32_t *)c->cidx_ent.ptr));
        HIPCHK(c, hipEventRecord(c->ev_aux_join, c->aux_stream));
        const uint64_t K = (uint64_t)c->P * c->W;
        c->Kpad = (uint32_t)((K + c->kc - 1) / c->kc * c->kc);
        if (c->Kpad) {
            const size_t bytes = (size_t)c->Kpad * c->Npad * sizeof(uint32_t);
            HIPCHK(c, c->planes.ensure(bytes));
            if (c->Kpad > K)
                HIPCHK(c, hipMemsetAsync((uint32_t *)c->planes.ptr + K * c->Npad, 0,
                                         (size_t)(c->Kpad - K) * c->Npad * sizeof(uint32_t),
                                         c->stream));
            HIPCHK(c, launch_transform(c->stream, c->regs, ncols, c->p, c->pbase, c->P, c->W, c->Npad,
                                       (uint32_t *)c->planes.ptr,
                                       want_sorted ? (const uint32_t *)c->perm.ptr : nullptr));
        }
        c->aux_join_pending = true;  // only k_finalize reads the index: the tile kernel starts without waiting for it
        c->planes_sorted = want_sorted;
        c->lay_rb = want_rb;
        c->lay_re = want_re;
        c->lay_parts = parts;
        c->planes_valid = true;
    }
    if (e0 && e1) {
        (void)hipEventRecord(e1, c->stream);
        (void)hipEventSynchronize(e1);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        c->prep_ms += ms;
    }
    return DSH_OK;
}

// Order the tiles of a band so that workgroups that run on the same XCD (block b -> XCD b % 8,
// observed dispatch behaviour; speed only, never correctness) walk one tile row together and
// share its A panel in that XCD's L2.
void xcd_order(std::vector<uint4> &t, std::vector<uint32_t> &rank, size_t b, size_t e, size_t group)
{
    const size_t cnt = e - b;
    if (cnt < 16) return;
    std::vector<uint4> tmp(cnt);
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

struct PairJob {
    int estim, result_type, k;
    int rect;
    int sorted_rows = 0;  // rows (and the output) are in sorted plane-column order (shards)
    int square = 0;       // full triangle, each value written at (i,j) and (j,i) of an n x n matrix
    uint32_t nparts = 1;  // triangle rows in this many parts, an event per part (dsh_dist_rows_parts_device_async)
    int knn = 0;          // band of the key-ordered triangle for the nearest-neighbour selection: d_out = V, d_out2 = Vt
    float *d_out2 = nullptr;
    uint64_t knn_ld = 0, knn_rows = 0;
    int ksinv_double = 0; // 1./k as a double (nndist_loop, src/sketch_and_cmp.h:729) instead of the float of dist_loop (:797)
    uint64_t row_begin, row_end, col_begin, col_end;
    uint64_t base_index;
    float *d_out;
};

int run_pairs(dsh_ctx *c, const PairJob &job)
{
    // Triangle rows [rb,re): the plane matrix is laid out for exactly that range (wanted rows first, both
    // parts key-ordered), so every tile is homogeneous and the result lands at its final packed position --
    // whatever the range (a full triangle, one rank's rows, a row block of the CLI).  Tiny ranges and
    // rectangles keep the identity layout, which stays cached across calls.
    const uint64_t jre = std::min<uint64_t>(job.row_end, c->n);
    const bool full_tri = !job.rect && job.row_begin == 0 && jre >= c->n;
    int want_sorted = 0;
    uint64_t lrb = 0, lre = c->n;
    if (job.sorted_rows) want_sorted = 1;
    else if (!job.rect && c->sort_mode != 0 && jre > job.row_begin &&
             (full_tri || jre - job.row_begin >= (uint64_t)c->range_sort_min_rows)) {
        want_sorted = 1;
        lrb = job.row_begin;
        lre = jre;
    }
    int rc = prepare(c, job.estim, want_sorted, false, lrb, lre, want_sorted && !job.sorted_rows ? job.nparts : 1);
    if (rc) return rc;
    if (job.result_type < 0 || job.result_type > 8)
        return fail(c, DSH_EINVAL, "unsupported result_type %d", job.result_type);
    if (job.k < 1) return fail(c, DSH_EINVAL, "bad k %d", job.k);
    // tile list: {row block, col block, plane begin, plane end}; a tile only needs the planes
    // v in (max(min lo of its two blocks), max threshold of its two blocks]
    const auto t_l0 = std::chrono::steady_clock::now();
    std::vector<uint4> &T = c->htiles;
    T.clear();
    const uint32_t NT = c->Npad / kTile;
    auto tile_of = [&](uint32_t ti, uint32_t tj) {
        int pb, pe;
        tile_planes(c, ti, tj, pb, pe);
        return make_uint4(ti, tj, (uint32_t)pb, (uint32_t)pe);
    };
    if (job.rect) {
        if (job.row_begin >= job.row_end || job.col_begin >= job.col_end) return DSH_OK;
        const uint32_t r0 = (uint32_t)(job.row_begin / kTile), r1 = (uint32_t)((job.row_end + kTile - 1) / kTile);
        const uint32_t c0 = (uint32_t)(job.col_begin / kTile), c1 = (uint32_t)((job.col_end + kTile - 1) / kTile);
        for (uint32_t ti = r0; ti < r1; ++ti)
            for (uint32_t tj = c0; tj < c1; ++tj) T.push_back(tile_of(ti, tj));
    } else {
        if (job.row_begin >= job.row_end) return DSH_OK;
        uint32_t r0 = 0, r1 = NT;
        if (!c->planes_sorted || job.sorted_rows) {  // rows index plane columns: only their tile rows
            r0 = (uint32_t)(job.row_begin / kTile);
            r1 = std::min<uint32_t>(NT, (uint32_t)((job.row_end + kTile - 1) / kTile));
        } else {  // the wanted rows are the first lay_re - lay_rb columns
            r1 = std::min<uint32_t>(NT, (uint32_t)((c->lay_re - c->lay_rb + kTile - 1) / kTile));
        }
        for (uint32_t ti = r0; ti < r1; ++ti)
            for (uint32_t tj = ti; tj < NT; ++tj) T.push_back(tile_of(ti, tj));
    }
    if (T.empty()) return DSH_OK;
    // bands bounded by the cum scratch budget
    const uint64_t per_tile = (uint64_t)kTile * kTile * c->cum_bytes * std::max<uint32_t>(c->P, 1);
    const uint64_t max_tiles = std::max<uint64_t>(1, c->cum_budget / per_tile);
    std::vector<std::pair<size_t, size_t>> bands;
    // Parts (dsh_dist_rows_parts_device_async): k_finalize runs once per SEGMENT -- the tiles of one part inside one
    // band -- and an event marks the end of a part's last segment: the part's span of the matrix is final and can travel
    // while the other parts are computed.  The tile kernel runs once per band; a band is also cut at a part boundary
    // when the part is large (>= kPartBandTiles tiles: a cut costs 0.2-0.3 ms -- two tails and a pipeline bubble,
    // profiles/r3g -- nothing against the ~1 ms per 1 000 tiles the part takes, and the first part can leave after 1/nparts of
    // the compute instead of after the whole tile kernel); small parts (C3 / 8 ranks: ~50-400 tiles) only cut k_finalize.
    constexpr size_t kPartBandTiles = 2048;
    const bool parts_on = !job.rect && !job.sorted_rows && c->planes_sorted && c->lay_parts.size() > 2 && job.nparts > 1;
    // first tile of every part: T is row-major and a part is a run of whole tile rows, so the part of a tile is monotone
    std::vector<size_t> pstart{0};
    if (parts_on) {
        size_t q = 0;
        for (size_t t = 0; t < T.size(); ++t) {
            const uint64_t pos = (uint64_t)T[t].x * kTile;
            while (q + 2 < c->lay_parts.size() && pos >= c->lay_parts[q + 1] - c->lay_rb) {
                ++q;
                pstart.push_back(t);
            }
        }
    }
    pstart.push_back(T.size());
    auto part_of_tile = [&](size_t t) -> size_t {
        return (size_t)(std::upper_bound(pstart.begin(), pstart.end() - 1, t) - pstart.begin()) - 1;
    };
    for (size_t b = 0; b < T.size();) {
        size_t e = std::min<size_t>(T.size(), b + max_tiles);
        if (parts_on) {  // a large part also ends the band
            const size_t q = part_of_tile(b);
            if (pstart[q + 1] - pstart[q] >= kPartBandTiles) e = std::min(e, pstart[q + 1]);
        }
        bands.emplace_back(b, e);
        b = e;
    }
    struct Seg { size_t b, e; int part; };  // tiles [b, e) of T; part completed by this segment or -1
    c->last_bands = bands.size();
    std::vector<std::vector<Seg>> segs(bands.size());
    {
        const bool with_parts = parts_on;
        auto part_of = part_of_tile;  // part of a tile = part of its tile row (parts are runs of whole tile rows of the layout)
        for (size_t bi = 0; bi < bands.size(); ++bi) {
            size_t b = bands[bi].first;
            while (b < bands[bi].second) {
                const size_t q = part_of(b);
                const size_t e = std::min(bands[bi].second, pstart[q + 1]);
                const bool last_of_part = e == pstart[q + 1];
                segs[bi].push_back({b, e, with_parts && last_of_part ? (int)q : -1});
                b = e;
            }
        }
        c->parts_done = 0;
    }
    // rank[t] = position of tile t in the row-major order of its segment (the order k_finalize walks, see below)
    std::vector<uint32_t> &rank = c->tile_rank;
    rank.resize(T.size());
    for (auto &sv : segs)
        for (auto &sg : sv)
            for (size_t t = sg.b; t < sg.e; ++t) rank[t] = (uint32_t)(t - sg.b);
    if (c->xcd_swizzle)
        for (auto &sv : segs)
            for (auto &sg : sv) xcd_order(T, rank, sg.b, sg.e, use_lockstep(c) ? 2 : 1);
    // work items per band: {tile index in band, chunk begin, chunk end}
    const uint32_t KC = (uint32_t)c->kc;
    auto chunk_range = [&](const uint4 &t, uint32_t &cb, uint32_t &ce) {
        cb = (uint32_t)(((uint64_t)t.z * c->W) / KC);
        ce = (uint32_t)(((uint64_t)t.w * c->W + KC - 1) / KC);
    };
    std::vector<uint4> &I = c->hitems;
    I.clear();
    std::vector<std::pair<size_t, size_t>> band_items;
    const uint32_t cpp = c->W >= KC ? c->W / KC : 1;  // chunks per plane when a plane spans chunks
    const bool lockstep = use_lockstep(c);
    std::vector<uint2> &CR = c->tile_chunks;  // chunk range of every tile, computed once
    CR.resize(T.size());
    for (size_t t = 0; t < T.size(); ++t) {
        uint32_t cb, ce;
        chunk_range(T[t], cb, ce);
        CR[t] = make_uint2(cb, ce);
    }
    for (auto &bd : bands) {
        const size_t nt = bd.second - bd.first;
        uint64_t tot = 0;
        for (size_t t = bd.first; t < bd.second; ++t) tot += CR[t].y - CR[t].x;
        // piece size: whole planes, aiming at >= 16 items per resident workgroup slot (512)
        uint64_t piece = c->nsplit > 0 ? std::max<uint64_t>(1, (tot / std::max<size_t>(nt, 1) + c->nsplit - 1) / c->nsplit)
                                       : std::max<uint64_t>(1, tot / (16 * 512));
        if (c->nsplit == 0 && lockstep)  // equal, short items: the two items of a workgroup run in lockstep
            piece = std::min<uint64_t>(piece, std::max<uint64_t>(cpp, (uint64_t)c->ls_item_chunks));
        piece = (piece + cpp - 1) / cpp * cpp;
        const size_t i0 = I.size();
        uint32_t maxpieces = 0;
        for (size_t t = bd.first; t < bd.second; ++t)
            maxpieces = std::max<uint32_t>(maxpieces, (uint32_t)((CR[t].y - CR[t].x + piece - 1) / piece));
        for (uint32_t s = 0; s < maxpieces; ++s) {  // piece-major so neighbours in launch order share planes
            const size_t g0 = I.size();
            uint32_t lmin = ~0u, lmax = 0;
            for (size_t t = bd.first; t < bd.second; ++t) {
                const uint64_t b0 = CR[t].x + (uint64_t)s * piece;
                if (b0 >= CR[t].y) continue;
                const uint32_t e0 = (uint32_t)std::min<uint64_t>(CR[t].y, b0 + piece);
                I.push_back(make_uint4((uint32_t)(t - bd.first), (uint32_t)b0, e0, 0));
                lmin = std::min<uint32_t>(lmin, e0 - (uint32_t)b0);
                lmax = std::max<uint32_t>(lmax, e0 - (uint32_t)b0);
            }
            // the lockstep kernel pairs consecutive items: keep equal lengths together (only a tile's last piece can be
            // shorter; with whole-plane pieces every item of the group is the same length and there is nothing to do)
            if (lockstep && c->ls_sort_items && lmin != lmax)
                std::stable_sort(I.begin() + g0, I.end(), [](const uint4 &x, const uint4 &y) { return x.z - x.y > y.z - y.y; });
        }
        band_items.emplace_back(i0, I.size());
    }
    // tile and item lists travel through page-locked staging, so nothing below needs the host to wait
    if (c->lists_in_flight) {  // the previous call's upload (long done unless calls are issued back to back)
        HIPCHK(c, hipEventSynchronize(c->ev_lists));
        c->lists_in_flight = false;
    }
    HIPCHK(c, c->pin_lists.ensure((2 * T.size() + std::max<size_t>(I.size(), 1)) * sizeof(uint4)));
    uint4 *pinT = (uint4 *)c->pin_lists.ptr, *pinF = pinT + T.size(), *pinI = pinF + T.size();
    // The tile kernel's list (launch order: XCD-interleaved, see xcd_order) holds {row block, column block, ..}.
    // k_finalize has its own list, every segment in ROW-MAJOR order: {row block, column block, plane begin | plane end
    // << 8 | smallest << 16 | largest << 24 register value of the two blocks' sketches (its histogram columns only span
    // the values the tile's sketches can hold), index of the tile's C(v) block in the band}.  A block of k_finalize
    // writes one row of a tile into row perm[si] of the packed matrix, scattered over the row (the columns are
    // key-ordered); block b runs on XCD b % 8 = tile row % 8, so a given output row is always written through the same
    // L2.  Walking a tile row's tiles one after the other keeps that row's lines in L2 until they are complete (the 128
    // rows of a tile row are 5 MB at C3, spread over the 8 L2s); in the tile kernel's interleaved order 8 tile rows were
    // in flight at once, lines left the L2 partly written and WRITE_SIZE was 6x the output (profiles/r3a, r3i).
    auto tile_vrange = [&](const uint4 &t, int &lo, int &hi) {
        lo = std::min<int>(c->blk_lo[t.x], c->blk_lo[t.y]);
        hi = std::max<int>(c->blk_hi[t.x], c->blk_hi[t.y]);
        if (hi < lo) hi = lo;  // (blocks of padding only)
    };
    for (size_t t = 0; t < T.size(); ++t) {
        int lo, hi;
        tile_vrange(T[t], lo, hi);
        pinT[t] = make_uint4(T[t].x, T[t].y, T[t].z | (T[t].w << 8), (uint32_t)lo | ((uint32_t)hi << 8));
    }
    for (size_t bi = 0; bi < bands.size(); ++bi)
        for (const Seg &sg : segs[bi])
            for (size_t t = sg.b; t < sg.e; ++t) {
                int lo, hi;
                tile_vrange(T[t], lo, hi);
                const size_t at = c->finalize_rowmajor ? sg.b + rank[t] : t;
                pinF[at] = make_uint4(T[t].x, T[t].y, T[t].z | (T[t].w << 8) | ((uint32_t)lo << 16) | ((uint32_t)hi << 24),
                                      (uint32_t)(t - bands[bi].first));
            }
    if (!I.empty()) std::memcpy(pinI, I.data(), I.size() * sizeof(uint4));
    c->host_lists_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_l0).count();
    HIPCHK(c, c->tiles.ensure(2 * T.size() * sizeof(uint4)));  // [tile kernel's list | k_finalize's list]
    HIPCHK(c, launch_upload(c->stream, c->tiles.ptr, pinT, 2 * T.size() * sizeof(uint4)));
    HIPCHK(c, c->items.ensure(std::max<size_t>(I.size(), 1) * sizeof(uint4)));
    if (!I.empty())
        HIPCHK(c, launch_upload(c->stream, c->items.ptr, pinI, I.size() * sizeof(uint4)));
    if (!c->ev_lists) HIPCHK(c, hipEventCreateWithFlags(&c->ev_lists, hipEventDisableTiming));
    HIPCHK(c, hipEventRecord(c->ev_lists, c->stream));
    c->lists_in_flight = true;
    size_t max_band = 0;
    for (auto &bd : bands) max_band = std::max(max_band, bd.second - bd.first);
    HIPCHK(c, c->cum.ensure(std::max<uint64_t>(per_tile * max_band, 256)));

    const float ksinv_f = (float)(1. / (double)job.k);
    std::vector<std::pair<hipEvent_t, hipEvent_t>> evp, evf;
    for (size_t bi = 0; bi < bands.size(); ++bi) {
        const auto &bd = bands[bi];
        const uint32_t nt = (uint32_t)(bd.second - bd.first);
        const uint64_t nslots = (uint64_t)nt * kTile * kTile;
        const uint4 *dt = (const uint4 *)c->tiles.ptr + bd.first;
        const uint4 *di = (const uint4 *)c->items.ptr + band_items[bi].first;
        const uint32_t ni = (uint32_t)(band_items[bi].second - band_items[bi].first);
        hipEvent_t a = nullptr, b = nullptr, d = nullptr;
        if (c->profiling) {
            a = next_event(c);
            b = next_event(c);
            d = next_event(c);
            if (a) (void)hipEventRecord(a, c->stream);
        }
        if (c->pair_mfma)
            HIPCHK(c, launch_pair_counts_mfma(c->stream, c->kc, c->cum_bytes, (const uint32_t *)c->planes.ptr,
                                              c->Npad, c->Kpad, c->W, c->P, dt, di, ni, c->cum.ptr, nslots));
        else if (use_lockstep(c))
            HIPCHK(c, launch_pair_counts_lockstep(c->stream, c->kc, c->cum_bytes, (const uint32_t *)c->planes.ptr,
                                                  c->Npad, c->Kpad, c->W, c->P, dt, di, ni, c->cum.ptr, nslots));
        else
            HIPCHK(c, launch_pair_counts(c->stream, c->kc, c->cum_bytes, (const uint32_t *)c->planes.ptr,
                                         c->Npad, c->Kpad, c->W, c->P, dt, di, ni, c->cum.ptr, nslots));
        if (b) (void)hipEventRecord(b, c->stream);
        for (const Seg &sg : segs[bi]) {
            FinalizeLaunch f;
            f.cum = c->cum.ptr;  // the band's C(v); a tile's block is named by its descriptor
            f.cum_bytes = c->cum_bytes;
            f.cum_stride = nslots;
            f.hist_bins = 1;
            for (size_t t = sg.b; t < sg.e; ++t) {
                int lo, hi;
                tile_vrange(T[t], lo, hi);
                f.hist_bins = std::max(f.hist_bins, hi - lo + 1);
            }
            f.exc = c->exc.ptr;
            f.exc_n = (const uint32_t *)c->exc_n.ptr;
            f.excv = (const uint8_t *)c->excv.ptr;
            f.keys = (const uint32_t *)c->keys.ptr;
            f.tailhist = (const uint8_t *)c->tailhist.ptr;
            f.nslots = (uint64_t)(sg.e - sg.b) * kTile * kTile;
            f.tiles = (const uint4 *)c->tiles.ptr + T.size() + sg.b;
            f.perm = c->planes_sorted ? (const uint32_t *)c->perm.ptr : nullptr;
            f.pbase = c->pbase;
            f.cidx_off = (const uint16_t *)c->cidx_off.ptr;
            f.cidx_ent = (const uint32_t *)c->cidx_ent.ptr;
            f.nbuckets = c->nbuckets;
            f.ent_stride = c->ent_stride;
            f.p = c->p;
            f.estim = job.estim;
            f.result_type = job.result_type;
            f.ksinv = job.ksinv_double ? 1. / (double)job.k : (double)ksinv_f;
            f.card = (const double *)c->card.ptr;
            f.n = c->n;
            f.ncols = c->ncols;
            f.stop = c->finalize_stop;
            f.rect = job.rect;
            f.sorted_out = job.sorted_rows;
            f.square = job.square;
            f.knn = job.knn;
            f.out2 = job.d_out2;
            f.knn_ld = job.knn_ld;
            f.knn_rows = job.knn_rows;
            f.row_begin = job.row_begin;
            f.row_end = job.row_end;
            f.col_begin = job.col_begin;
            f.col_end = job.col_end;
            f.base_index = job.base_index;
            f.out = job.d_out;
            if (c->aux_join_pending) {
                HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_aux_join, 0));
                c->aux_join_pending = false;
            }
            HIPCHK(c, launch_finalize(c->stream, f));
            if (sg.part >= 0) {  // this segment completes a part: its span of the matrix is final
                const size_t q = (size_t)sg.part;
                while (c->ev_part.size() <= q) {
                    hipEvent_t e = nullptr;
                    HIPCHK(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
                    c->ev_part.push_back(e);
                }
                HIPCHK(c, hipEventRecord(c->ev_part[q], c->stream));
                c->parts_done = (uint32_t)q + 1;
            }
        }
        if (d) (void)hipEventRecord(d, c->stream);
        if (a && b && d) {
            evp.emplace_back(a, b);
            evf.emplace_back(b, d);
        }
    }
    // everything is enqueued; the blocking entry points synchronise, the *_async ones return here
    if (c->profiling) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        for (auto &e : evp) {
            float ms = 0;
            (void)hipEventElapsedTime(&ms, e.first, e.second);
            c->pair_ms += ms;
            if (c->Kpad) c->pair_launches++;
        }
        for (auto &e : evf) {
            float ms = 0;
            (void)hipEventElapsedTime(&ms, e.first, e.second);
            c->fin_ms += ms;
        }
    }
    return DSH_OK;
}

void reset_prof(dsh_ctx *c)
{
    c->pair_ms = c->fin_ms = c->prep_ms = 0;
    c->pair_launches = 0;
    c->ev_used = 0;
}

}  // namespace

// The copy-out stream gets the highest priority the device offers: on this runtime a device-to-host copy that has to
// wait for an event of another stream runs as a small blit kernel, which would otherwise queue behind the millions
// of workgroups of the compare kernels it is meant to overlap with (profiles/r3d).
static hipError_t create_copy_stream(hipStream_t *s)
{
    int least = 0, greatest = 0;
    if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) greatest = 0;
    if (const char *e = std::getenv("DSH_COPY_STREAM_PRIORITY")) greatest = std::atoi(e);
    return hipStreamCreateWithPriority(s, hipStreamNonBlocking, greatest);
}

extern "C" {

const char *dsh_backend_name(void) { return "hip:gfx950"; }

int dsh_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int dsh_create(int device, dsh_ctx **out)
{
    if (!out) return DSH_EINVAL;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return DSH_ENODEV;
    if (device < 0 || device >= n) return DSH_EINVAL;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return DSH_ENODEV;
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) return DSH_ENODEV;  // gfx950-only code object
    dsh_ctx *c = new (std::nothrow) dsh_ctx;
    if (!c) return DSH_ENOMEM;
    c->device = device;
    if (hipSetDevice(device) != hipSuccess ||
        hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
        create_copy_stream(&c->copy_stream) != hipSuccess ||
        hipStreamCreateWithFlags(&c->aux_stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_aux_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_aux_join, hipEventDisableTiming) != hipSuccess) {
        if (c->stream) (void)hipStreamDestroy(c->stream);
        delete c;
        return DSH_EIO;
    }
    *out = c;
    return DSH_OK;
}

void dsh_destroy(dsh_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) {
        (void)hipStreamSynchronize(c->stream);
        (void)hipStreamDestroy(c->stream);
    }
    if (c->comm && rccl()->h) (void)rccl()->CommDestroy(c->comm);
    c->gather_full.release();
    c->gather_local.release();
    for (auto e : c->ev_pool) (void)hipEventDestroy(e);
    c->regs_own.release();
    c->card.release();
    c->planes.release();
    c->exc.release();
    c->exc_n.release();
    c->excv.release();
    if (c->pin_perm) (void)hipHostFree(c->pin_perm);
    c->pin_lists.release();
    c->pin_work.release();
    c->pin_keys.release();
    if (c->ev_work) (void)hipEventDestroy(c->ev_work);
    if (c->ev_lists) (void)hipEventDestroy(c->ev_lists);
    if (c->ev_perm) (void)hipEventDestroy(c->ev_perm);
    c->keys.release();
    c->tailhist.release();
    c->hist.release();
    if (c->ev_keys) (void)hipEventDestroy(c->ev_keys);
    c->cidx_off.release();
    c->cidx_ent.release();
    c->perm.release();
    c->items.release();
    c->cum.release();
    c->tiles.release();
    c->outbuf.release();
    for (int b = 0; b < 2; ++b) {
        c->outbuf2[b].release();
        if (c->ev_filled[b]) (void)hipEventDestroy(c->ev_filled[b]);
        if (c->ev_drained[b]) (void)hipEventDestroy(c->ev_drained[b]);
    }
    for (auto e : c->tickets) (void)hipEventDestroy(e);
    for (auto e : c->ev_part) (void)hipEventDestroy(e);
    if (c->copy_stream) {
        (void)hipStreamSynchronize(c->copy_stream);
        (void)hipStreamDestroy(c->copy_stream);
    }
    if (c->aux_stream) {
        (void)hipStreamSynchronize(c->aux_stream);
        (void)hipStreamDestroy(c->aux_stream);
    }
    if (c->ev_aux_fork) (void)hipEventDestroy(c->ev_aux_fork);
    if (c->ev_aux_join) (void)hipEventDestroy(c->ev_aux_join);
    c->seqbuf.release();
    c->workbuf.release();
    delete c;
}

const char *dsh_last_error(const dsh_ctx *c) { return c ? c->err.c_str() : "null ctx"; }

int dsh_synchronize(dsh_ctx *c)
{
    if (!c) return DSH_EINVAL;
    int rc = bind(c);
    if (rc) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipStreamSynchronize(c->copy_stream));
    return DSH_OK;
}

void *dsh_stream(dsh_ctx *c) { return c ? (void *)c->stream : nullptr; }

int dsh_sketches_alloc(dsh_ctx *c, uint64_t n, int p)
{
    if (!c) return DSH_EINVAL;
    if (p < 4 || p > kMaxP) return fail(c, DSH_EINVAL, "p=%d outside [4,%d]", p, kMaxP);
    int rc = bind(c);
    if (rc) return rc;
    const size_t bytes = std::max<size_t>((size_t)n << p, 256);
    HIPCHK(c, c->regs_own.ensure(bytes));
    HIPCHK(c, hipMemsetAsync(c->regs_own.ptr, 0, bytes, c->stream));
    c->regs = (const uint8_t *)c->regs_own.ptr;
    c->n = n;
    c->p = p;
    c->have_sketches = true;
    invalidate(c);
    return DSH_OK;
}

int dsh_attach_device_sketches(dsh_ctx *c, const void *d_regs, uint64_t n, int p)
{
    if (!c || (!d_regs && n)) return DSH_EINVAL;
    if (p < 4 || p > kMaxP) return fail(c, DSH_EINVAL, "p=%d outside [4,%d]", p, kMaxP);
    if (((uintptr_t)d_regs & 15) != 0) return fail(c, DSH_EINVAL, "device sketches must be 16-byte aligned");
    c->regs = (const uint8_t *)d_regs;
    c->n = n;
    c->p = p;
    c->have_sketches = true;
    invalidate(c);
    return DSH_OK;
}

int dsh_upload_sketches(dsh_ctx *c, const uint8_t *regs, uint64_t first, uint64_t n)
{
    if (!c || (!regs && n)) return DSH_EINVAL;
    if (!c->have_sketches || c->regs != (const uint8_t *)c->regs_own.ptr)
        return fail(c, DSH_ESTATE, "dsh_sketches_alloc first");
    if (!slots_ok(first, n, c->n)) return fail(c, DSH_EINVAL, "slots [%llu,+%llu) out of range", (unsigned long long)first, (unsigned long long)n);
    int rc = bind(c);
    if (rc) return rc;
    if (n) {
        HIPCHK(c, hipMemcpyAsync((uint8_t *)c->regs_own.ptr + (first << c->p), regs, (size_t)n << c->p,
                                 hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    invalidate(c);
    return DSH_OK;
}

int dsh_download_sketches(dsh_ctx *c, uint64_t first, uint64_t n, uint8_t *out)
{
    if (!c || (!out && n)) return DSH_EINVAL;
    if (!c->have_sketches) return fail(c, DSH_ESTATE, "no sketches");
    if (!slots_ok(first, n, c->n)) return fail(c, DSH_EINVAL, "slots out of range");
    int rc = bind(c);
    if (rc) return rc;
    if (n) {
        HIPCHK(c, hipMemcpyAsync(out, c->regs + (first << c->p), (size_t)n << c->p,
                                 hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    return DSH_OK;
}

int dsh_copy_sketches_device(dsh_ctx *c, uint64_t first, uint64_t n, void *d_out)
{
    if (!c || (!d_out && n)) return DSH_EINVAL;
    if (!c->have_sketches) return fail(c, DSH_ESTATE, "no sketches");
    if (!slots_ok(first, n, c->n)) return fail(c, DSH_EINVAL, "slots out of range");
    int rc = bind(c);
    if (rc) return rc;
    if (n) {
        HIPCHK(c, hipMemcpyAsync(d_out, c->regs + (first << c->p), (size_t)n << c->p,
                                 hipMemcpyDeviceToDevice, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    return DSH_OK;
}

int dsh_clear_sketches(dsh_ctx *c, uint64_t first, uint64_t n)
{
    if (!c) return DSH_EINVAL;
    if (!c->have_sketches || c->regs != (const uint8_t *)c->regs_own.ptr)
        return fail(c, DSH_ESTATE, "dsh_sketches_alloc first");
    if (!slots_ok(first, n, c->n)) return fail(c, DSH_EINVAL, "slots out of range");
    int rc = bind(c);
    if (rc) return rc;
    if (n) HIPCHK(c, hipMemsetAsync((uint8_t *)c->regs_own.ptr + (first << c->p), 0, (size_t)n << c->p, c->stream));
    invalidate(c);
    return DSH_OK;
}

static int sketch_common(dsh_ctx *c, const uint8_t *d_seq, const uint64_t *genome_off,
                         uint32_t n_genomes, uint64_t first_slot, int k, int canon)
{
    // work list: each workgroup walks up to kSubsPerWG sub-chunks of one genome
    constexpr uint32_t kSubsPerWG = 16;
    std::vector<SketchWork> work;
    for (uint32_t g = 0; g < n_genomes; ++g) {
        const uint64_t gb = genome_off[g], ge = genome_off[g + 1];
        if (ge < gb) return fail(c, DSH_EINVAL, "genome_off not monotone at %u", g);
        if (ge - gb < (uint64_t)k) continue;
        const uint64_t c0 = gb & ~31ull;
        const uint64_t nsub = (ge - c0 + kSketchSub - 1) / kSketchSub;
        for (uint64_t s = 0; s < nsub; s += kSubsPerWG) {
            SketchWork w;
            w.gbeg = gb;
            w.gend = ge;
            w.start = c0 + s * kSketchSub;
            w.nsub = (uint32_t)std::min<uint64_t>(kSubsPerWG, nsub - s);
            w.slot = (uint32_t)(first_slot + g);
            work.push_back(w);
        }
    }
    if (work.empty()) return DSH_OK;
    // the work list travels through page-locked staging (rewritten only after its previous upload has run),
    // so nothing here waits: the blocking entry points synchronise, dsh_sketch_batch_async returns
    if (c->work_in_flight) {
        HIPCHK(c, hipEventSynchronize(c->ev_work));
        c->work_in_flight = false;
    }
    HIPCHK(c, c->pin_work.ensure(work.size() * sizeof(SketchWork)));
    std::memcpy(c->pin_work.ptr, work.data(), work.size() * sizeof(SketchWork));
    HIPCHK(c, c->workbuf.ensure(work.size() * sizeof(SketchWork)));
    HIPCHK(c, launch_upload(c->stream, c->workbuf.ptr, c->pin_work.ptr, work.size() * sizeof(SketchWork)));
    if (!c->ev_work) HIPCHK(c, hipEventCreateWithFlags(&c->ev_work, hipEventDisableTiming));
    HIPCHK(c, hipEventRecord(c->ev_work, c->stream));
    c->work_in_flight = true;
    HIPCHK(c, launch_sketch(c->stream, d_seq, (const SketchWork *)c->workbuf.ptr,
                            (uint32_t)work.size(), k, c->p, canon, (uint8_t *)c->regs_own.ptr));
    return DSH_OK;
}

static int sketch_check(dsh_ctx *c, const uint64_t *genome_off, uint32_t n_genomes,
                        uint64_t first_slot, int k)
{
    if (!c || (!genome_off && n_genomes)) return DSH_EINVAL;
    if (k < 1 || k > 32) return fail(c, DSH_EINVAL, "k=%d outside [1,32]", k);
    if (!c->have_sketches || c->regs != (const uint8_t *)c->regs_own.ptr)
        return fail(c, DSH_ESTATE, "dsh_sketches_alloc first");
    if (!slots_ok(first_slot, n_genomes, c->n)) return fail(c, DSH_EINVAL, "slots out of range");
    return DSH_OK;
}

int dsh_sketch_batch_async(dsh_ctx *c, const uint8_t *seq, const uint64_t *genome_off, uint32_t n_genomes,
                           uint64_t first_slot, int k, int canon)
{
    int rc = sketch_check(c, genome_off, n_genomes, first_slot, k);
    if (rc) return rc;
    if ((rc = bind(c))) return rc;
    if (n_genomes == 0) return DSH_OK;
    const uint64_t lo = genome_off[0], hi = genome_off[n_genomes];
    if (hi < lo) return fail(c, DSH_EINVAL, "genome_off not monotone");
    // ship only [lo,hi), 32-aligned on the device side; pad so every lane's 64-byte read is in bounds
    const uint64_t shift = lo & 31;
    const size_t bytes = (size_t)(hi - lo) + shift;
    HIPCHK(c, c->seqbuf.ensure(bytes + 256));
    if (hi > lo) {
        if (!seq) return DSH_EINVAL;
        HIPCHK(c, hipMemcpyAsync((uint8_t *)c->seqbuf.ptr + shift, seq + lo, (size_t)(hi - lo),
                                 hipMemcpyHostToDevice, c->stream));
    }
    std::vector<uint64_t> off(n_genomes + 1);
    for (uint32_t g = 0; g <= n_genomes; ++g) off[g] = genome_off[g] - lo + shift;
    rc = sketch_common(c, (const uint8_t *)c->seqbuf.ptr, off.data(), n_genomes, first_slot, k, canon);
    if (rc) return rc;
    invalidate(c);
    return DSH_OK;
}

int dsh_sketch_batch(dsh_ctx *c, const uint8_t *seq, const uint64_t *genome_off, uint32_t n_genomes,
                     uint64_t first_slot, int k, int canon, uint8_t *regs_out)
{
    int rc = dsh_sketch_batch_async(c, seq, genome_off, n_genomes, first_slot, k, canon);
    if (rc) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));  // (a pageable `seq` was copied synchronously anyway)
    if (regs_out) return dsh_download_sketches(c, first_slot, n_genomes, regs_out);
    return DSH_OK;
}

int dsh_sketch_batch_device(dsh_ctx *c, const void *d_seq, const uint64_t *genome_off,
                            uint32_t n_genomes, uint64_t first_slot, int k, int canon)
{
    int rc = sketch_check(c, genome_off, n_genomes, first_slot, k);
    if (rc) return rc;
    if ((rc = bind(c))) return rc;
    if (n_genomes == 0) return DSH_OK;
    if (!d_seq || ((uintptr_t)d_seq & 31)) return fail(c, DSH_EINVAL, "d_seq must be 32-byte aligned and padded by 128 bytes");
    rc = sketch_common(c, (const uint8_t *)d_seq, genome_off, n_genomes, first_slot, k, canon);
    if (rc) return rc;
    invalidate(c);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return DSH_OK;
}

int dsh_cardinalities(dsh_ctx *c, int estim, double *out)
{
    if (!c || !out) return DSH_EINVAL;
    int rc = bind(c);
    if (rc) return rc;
    if (!c->have_sketches) return fail(c, DSH_ESTATE, "no sketches loaded");
    if (estim < 0 || estim > 2) return fail(c, DSH_EINVAL, "bad estimator %d", estim);
    if (c->card_estim != estim || c->card_from != 0) {
        // same per-sketch pass as prepare() (thresholds/exception lists come out identical)
        rc = prepare(c, estim, -1, /*card_only=*/true);
        if (rc) return rc;
    }
    if (c->n) {
        HIPCHK(c, hipMemcpyAsync(out, c->card.ptr, c->n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return DSH_OK;
}

uint64_t dsh_tri_index(uint64_t n, uint64_t i, uint64_t j) { return i * (2 * n - i - 1) / 2 + j - (i + 1); }

uint64_t dsh_tri_span(uint64_t n, uint64_t rb, uint64_t re)
{
    if (re > n) re = n;
    if (rb >= re) return 0;
    // sum_{i=rb}^{re-1} (n-1-i)
    const uint64_t cnt = re - rb;
    return cnt * (n - 1) - (rb + re - 1) * cnt / 2;
}

int dsh_partition_rows(uint64_t n, uint32_t nparts, uint32_t align, uint64_t *bounds)
{
    if (!bounds || nparts == 0) return DSH_EINVAL;
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
            const long double cum = (long double)dsh_tri_span(n, 0, cand);
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
    return DSH_OK;
}

int dsh_balance_rows(uint64_t n, uint32_t nparts, uint64_t *bounds)
{
    if (!bounds || nparts == 0) return DSH_EINVAL;
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
            bool any = false;
            while (t < NT && acc + cost(t) <= limit) {
                acc += cost(t++);
                any = true;
            }
            (void)any;
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
    return DSH_OK;
}

int dsh_dist_rows_device_async(dsh_ctx *c, int estim, int result_type, int k, uint64_t rb, uint64_t re, void *d_out)
{
    if (!c) return DSH_EINVAL;
    int rc = bind(c);
    if (rc) return rc;
    if (!c->have_sketches) return fail(c, DSH_ESTATE, "no sketches loaded");
    reset_prof(c);
    if (re > c->n) re = c->n;
    if (rb >= re || c->n < 2) return DSH_OK;
    if (!d_out) return DSH_EINVAL;
    PairJob j;
    j.estim = estim;
    j.result_type = result_type;
    j.k = k;
    j.rect = 0;
    j.row_begin = rb;
    j.row_end = re;
    j.col_begin = j.col_end = 0;
    j.base_index = dsh_tri_span(c->n, 0, rb);
    j.d_out = (float *)d_out;
    return run_pairs(c, j);
}

int dsh_range_parts(uint64_t n, uint64_t rb, uint64_t re, uint32_t nparts, uint64_t *part_rows, uint32_t *nparts_out)
{
    if (!part_rows || !nparts_out || nparts == 0) return DSH_EINVAL;
    if (re > n) re = n;
    if (rb > re) rb = re;
    std::vector<uint64_t> parts;
    range_parts(n, rb, re, nparts, parts);
    for (size_t q = 0; q < parts.size(); ++q) part_rows[q] = parts[q];
    *nparts_out = (uint32_t)parts.size() - 1;
    return DSH_OK;
}

int dsh_dist_rows_parts_device_async(dsh_ctx *c, int estim, int result_type, int k, uint64_t rb, uint64_t re, void *d_out,
                                     uint32_t nparts)
{
    if (!c || nparts == 0) return DSH_EINVAL;
    int rc = bind(c);
    if (rc) return rc;
    if (!c->have_sketches) return fail(c, DSH_ESTATE, "no sketches loaded");
    reset_prof(c);
    c->parts_done = 0;
    if (re > c->n) re = c->n;
    if (rb >= re || c->n < 2) return DSH_OK;
    if (!d_out) return DSH_EINVAL;
    PairJob j;
    j.estim = estim;
    j.result_type = result_type;
    j.k = k;
    j.rect = 0;
    j.nparts = nparts;
    j.row_begin = rb;
    j.row_end = re;
    j.col_begin = j.col_end = 0;
    j.base_index = dsh_tri_span(c->n, 0, rb);
    j.d_out = (float *)d_out;
    return run_pairs(c, j);
}

int dsh_dist_rows_device(dsh_ctx *c, int estim, int result_type, int k, uint64_t rb, uint64_t re, void *d_out)
{
    int rc = dsh_dist_rows_device_async(c, estim, result_type, k, rb, re, d_out);
    if (rc) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return DSH_OK;
}

int dsh_dist_rows_async(dsh_ctx *c, int estim, int result_type, int k, uint64_t rb, uint64_t re, float *out)
{
    if (!c) return DSH_EINVAL;
    int rc = bind(c);
    if (rc) return rc;
    if (!c->have_sketches) return fail(c, DSH_ESTATE, "no sketches loaded");
    if (re > c->n) re = c->n;
    const uint64_t span = dsh_tri_span(c->n, rb, re);
    if (span == 0) return DSH_OK;
    if (!out) return DSH_EINVAL;
    // Two device buffers taken in turn: this call's kernels (ctx stream) wait only for the copy that last drained
    // THEIR buffer, so they run while the previous call's result is still on its way to the host (copy stream) --
    // the reference overlaps the comparison of one batch of rows with the emission of the previous one the same way
    // (src/sketch_and_cmp.h:804-816, distmat/distmat.h:475-479,504-508).
    const unsigned b = c->out_turn++ & 1u;
    for (int t = 0; t < 2; ++t) {
        if (!c->ev_filled[t]) HIPCHK(c, hipEventCreateWithFlags(&c->ev_filled[t], hipEventDisableTiming));
        if (!c->ev_drained[t]) HIPCHK(c, hipEventCreateWithFlags(&c->ev_drained[t], hipEventDisableTiming));
    }
    if (c->outbuf2[b].cap < span * sizeof(float)) {  // growing frees the old buffer: let its last copy finish first
        if (c->drained_pending[b]) HIPCHK(c, hipEventSynchronize(c->ev_drained[b]));
        c->drained_pending[b] = false;
        HIPCHK(c, c->outbuf2[b].ensure(span * sizeof(float)));
    }
    if (c->drained_pending[b]) HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_drained[b], 0));
    rc = dsh_dist_rows_device_async(c, estim, result_type, k, rb, re, c->outbuf2[b].ptr);
    if (rc) return rc;
    HIPCHK(c, hipEventRecord(c->ev_filled[b], c->stream));
    HIPCHK(c, hipStreamWaitEvent(c->copy_stream, c->ev_filled[b], 0));
    HIPCHK(c, hipMemcpyAsync(out, c->outbuf2[b].ptr, span * sizeof(float), hipMemcpyDeviceToHost, c->copy_stream));
    HIPCHK(c, hipEventRecord(c->ev_drained[b], c->copy_stream));
    c->drained_pending[b] = true;
    return DSH_OK;
}

int dsh_dist_rows(dsh_ctx *c, int estim, int result_type, int k, uint64_t rb, uint64_t re, float *out)
{
    int rc = dsh_dist_rows_async(c, estim, result_type, k, rb, re, out);
    if (rc) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipStreamSynchronize(c->copy_stream));
    return DSH_OK;
}

int dsh_wait(dsh_ctx *c) { return dsh_synchronize(c); }

// A ticket marks "everything enqueued on this ctx so far" (kernels on the ctx stream and the copies of
// dsh_dist_rows_async / transfers of dsh_collect_parts_async on the copy stream); waiting for it does not wait for work
// enqueued afterwards, and recording it orders nothing between the two streams.
int dsh_event_record(dsh_ctx *c, uint64_t *ticket)
{
    if (!c || !ticket) return DSH_EINVAL;
    int rc = bind(c);
    if (rc) return rc;
    constexpr size_t kRing = 64;
    if (c->tickets.size() < 2 * kRing) {
        hipEvent_t e = nullptr, j = nullptr;
        HIPCHK(c, hipEventCreateWithFlags(&j, hipEventDisableTiming));
        c->tickets.push_back(j);
        HIPCHK(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        c->tickets.push_back(e);
    }
    const uint64_t t = c->ticket_next++;
    hipEvent_t j = c->tickets[2 * (t % kRing)], e = c->tickets[2 * (t % kRing) + 1];
    if (t >= kRing) HIPCHK(c, hipEventSynchronize(e));  // the ticket that used this slot 64 records ago
    // the copy stream joins the ctx stream's present position, then the event marks both
    HIPCHK(c, hipEventRecord(j, c->stream));
    HIPCHK(c, hipStreamWaitEvent(c->copy_stream, j, 0));
    HIPCHK(c, hipEventRecord(e, c->copy_stream));
    *ticket = t;
    return DSH_OK;
}

static int ticket_event(dsh_ctx *c, uint64_t ticket, hipEvent_t *e)
{
    if (ticket >= c->ticket_next) return fail(c, DSH_EINVAL, "ticket %llu was never recorded", (unsigned long long)ticket);
    *e = c->ticket_next - ticket > 64 ? nullptr : c->tickets[2 * (ticket % 64) + 1];  // older than the ring: waited for at its slot's reuse
    return DSH_OK;
}

int dsh_event_wait(dsh_ctx *c, uint64_t ticket)
{
    if (!c) return DSH_EINVAL;
    int rc = bind(c);
    if (rc) return rc;
    hipEvent_t e[2] = {nullptr, nullptr};
    if ((rc = ticket_event(c, ticket, e))) return rc;
    for (hipEvent_t x : e)
        if (x) HIPCHK(c, hipEventSynchronize(x));
    return DSH_OK;
}

int dsh_event_query(dsh_ctx *c, uint64_t ticket, int *done)
{
    if (!c || !done) return DSH_EINVAL;
    int rc = bind(c);
    if (rc) return rc;
    hipEvent_t e[2] = {nullptr, nullptr};
    if ((rc = ticket_event(c, ticket, e))) return rc;
    *done = 1;
    for (hipEvent_t x : e) {
        if (!x) continue;
        const hipError_t q = hipEventQuery(x);
        if (q == hipErrorNotReady) *done = 0;
        else if (q != hipSuccess) return fail(c, DSH_EIO, "hipEventQuery: %s", hipGetErrorString(q));
    }
    return DSH_OK;
}

int dsh_wait_event(dsh_ctx *c, void *hip_event)
{
    if (!c || !hip_event) return DSH_EINVAL;
    int rc = bind(c);
    if (rc) return rc;
    HIPCHK(c, hipStreamWaitEvent(c->stream, (hipEvent_t)hip_event, 0));
    return DSH_OK;
}

int dsh_dist_rect(dsh_ctx *c, int estim, int result_type, int k, uint64_t qb, uint64_t qe, uint64_t rb, uint64_t re, float *out)
{
    if (!c) return DSH_EINVAL;
    int rc = bind(c);
    if (rc) return rc;
    if (!c->have_sketches) return fail(c, DSH_ESTATE, "no sketches loaded");
    if (qe > c->n || re > c->n) return fail(c, DSH_EINVAL, "slots out of range");
    reset_prof(c);
    if (qb >= qe || rb >= re) return DSH_OK;
    if (!out) return DSH_EINVAL;
    const uint64_t cnt = (qe - qb) * (re - rb);
    HIPCHK(c, c->outbuf.ensure(cnt * sizeof(float)));
    PairJob j;
    j.estim = estim;
    j.result_type = result_type;
    j.k = k;
    j.rect = 1;
    j.row_begin = qb;
    j.row_end = qe;
    j.col_begin = rb;
    j.col_end = re;
    j.base_index = 0;
    j.d_out = (float *)c->outbuf.ptr;
    rc = run_pairs(c, j);
    if (rc) return rc;
    HIPCHK(c, hipMemcpyAsync(out, c->outbuf.ptr, cnt * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return DSH_OK;
}

int dsh_knn(dsh_ctx *c, int estim, int result_type, int k, uint64_t qb, uint64_t qe, uint64_t rb,
            uint64_t re, uint32_t nn, uint32_t *idx_out, float *val_out)
{
    if (!c) return DSH_EINVAL;
    int rc = bind(c);
    if (rc) return rc;
    if (!c->have_sketches) return fail(c, DSH_ESTATE, "no sketches loaded");
    if (qe > c->n || re > c->n) return fail(c, DSH_EINVAL, "slots out of range");
    reset_prof(c);
    if (qb >= qe || nn == 0) return DSH_OK;
    if (!idx_out || !val_out) return DSH_EINVAL;
    // similarity measures rank descending, distances ascending (emt2nntype, src/dashing.h:268-280)
    const int descending = !(result_type == DSH_MASH_DIST || result_type == DSH_FULL_MASH_DIST ||
                             result_type == DSH_CONTAINMENT_DIST || result_type == DSH_FULL_CONTAINMENT_DIST ||
                             result_type == DSH_SYMMETRIC_CONTAINMENT_DIST);
    const uint64_t nq = qe - qb, nr = re > rb ? re - rb : 0;
    const bool overlap = qb < re && rb < qe;
    if (qb == 0 && rb == 0 && qe == c->n && re == c->n && c->n > 1 &&
        c->n * c->n * sizeof(float) <= c->knn_square_budget) {
        // all-vs-all: every pair is computed ONCE (triangle tiles, sorted columns) and written at
        // both (i,j) and (j,i) of an n x n matrix in HBM; then one selection pass per row
        const uint64_t n = c->n;
        DevBuf sq, didx, dval;
        rc = DSH_OK;
        do {
            if (sq.ensure(n * n * sizeof(float)) != hipSuccess || didx.ensure(n * nn * sizeof(uint32_t)) != hipSuccess ||
                dval.ensure(n * nn * sizeof(float)) != hipSuccess) {
                rc = fail(c, DSH_ENOMEM, "device allocation failed");
                break;
            }
            PairJob j;
            j.estim = estim;
            j.result_type = result_type;
            j.k = k;
            j.rect = 0;
            j.square = 1;
            j.ksinv_double = 1;
            j.row_begin = 0;
            j.row_end = n;
            j.col_begin = 0;
            j.col_end = n;
            j.base_index = 0;
            j.d_out = (float *)sq.ptr;
            if ((rc = run_pairs(c, j))) break;
            hipError_t e = launch_topk(c->stream, (const float *)sq.ptr, n, n, 0, 0, descending, nn, 1,
                                       (uint32_t *)didx.ptr, (float *)dval.ptr);
            if (e != hipSuccess) {
                rc = fail(c, DSH_EIO, "k_topk: %s", hipGetErrorString(e));
                break;
            }
            if (hipMemcpyAsync(idx_out, didx.ptr, n * nn * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
                hipMemcpyAsync(val_out, dval.ptr, n * nn * sizeof(float), hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
                hipStreamSynchronize(c->stream) != hipSuccess)
                rc = fail(c, DSH_EIO, "copy of neighbours failed");
        } while (0);
        sq.release();
        didx.release();
        dval.release();
        return rc;
    }
    if (qb == 0 && rb == 0 && qe == c->n && re == c->n && c->n > 1 && nn <= 1024) {
        // all-vs-all beyond the n x n budget: the triangle ONCE, in bands of tile rows of the key-ordered layout.  A band
        // leaves its values twice (V: band rows x columns, Vt: columns x band rows -- each pair is a candidate of both its
        // sketches) and two selection passes fold them into the running lists of the n sketches; nothing of size n x n
        // exists (nndist_loop, src/sketch_and_cmp.h:712-783, keeps n heaps the same way).
        const uint64_t n = c->n;
        if ((rc = prepare(c, estim, 1))) return rc;
        const uint64_t npad = c->Npad;
        const uint64_t budget = std::max<uint64_t>(std::min<uint64_t>(c->knn_square_budget, (uint64_t)16 << 30), 2 * kTile * npad * sizeof(float));
        const uint64_t band = std::min<uint64_t>(npad, budget / (2 * npad * sizeof(float)) / kTile * kTile);
        DevBuf V, Vt, didx, dval;
        rc = DSH_OK;
        do {
            if (V.ensure(band * npad * sizeof(float)) != hipSuccess || Vt.ensure(npad * band * sizeof(float)) != hipSuccess ||
                didx.ensure(n * nn * sizeof(uint32_t)) != hipSuccess || dval.ensure(n * nn * sizeof(float)) != hipSuccess) {
                rc = fail(c, DSH_ENOMEM, "device allocation failed");
                break;
            }
            hipError_t e = launch_knn_state_init(c->stream, (uint32_t *)didx.ptr, (float *)dval.ptr, n * nn, descending);
            if (e != hipSuccess) {
                rc = fail(c, DSH_EIO, "k_fill_knn_state: %s", hipGetErrorString(e));
                break;
            }
            for (uint64_t b0 = 0; b0 < n && rc == DSH_OK; b0 += band) {
                const uint64_t b1 = std::min<uint64_t>(n, b0 + band);
                PairJob j;
                j.estim = estim;
                j.result_type = result_type;
                j.k = k;
                j.rect = 0;
                j.sorted_rows = 1;
                j.knn = 1;
                j.ksinv_double = 1;
                j.row_begin = b0;
                j.row_end = b1;
                j.col_begin = j.col_end = 0;
                j.base_index = 0;
                j.d_out = (float *)V.ptr;
                j.d_out2 = (float *)Vt.ptr;
                j.knn_ld = npad;
                j.knn_rows = band;
                if ((rc = run_pairs(c, j))) break;
                const uint32_t *perm = (const uint32_t *)c->perm.ptr;
                e = launch_topk_merge(c->stream, (const float *)V.ptr, npad, 0, b0, b1 - b0, n, perm, descending, nn,
                                      (uint32_t *)didx.ptr, (float *)dval.ptr);
                if (e == hipSuccess)
                    e = launch_topk_merge(c->stream, (const float *)Vt.ptr, band, 1, b0, b1 - b0, n, perm, descending, nn,
                                          (uint32_t *)didx.ptr, (float *)dval.ptr);
                if (e != hipSuccess) rc = fail(c, DSH_EIO, "k_topk_merge: %s", hipGetErrorString(e));
            }
            if (rc) break;
            if (hipMemcpyAsync(idx_out, didx.ptr, n * nn * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
                hipMemcpyAsync(val_out, dval.ptr, n * nn * sizeof(float), hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
                hipStreamSynchronize(c->stream) != hipSuccess)
                rc = fail(c, DSH_EIO, "copy of neighbours failed");
        } while (0);
        (void)hipStreamSynchronize(c->stream);
        V.release();
        Vt.release();
        didx.release();
        dval.release();
        return rc;
    }
    DevBuf &rect = c->outbuf;
    const uint64_t qblock = std::max<uint64_t>(1, std::min<uint64_t>(nq, ((uint64_t)256 << 20) / std::max<uint64_t>(nr, 1)));
    HIPCHK(c, rect.ensure(std::max<uint64_t>(qblock * nr, 1) * sizeof(float)));
    DevBuf didx, dval;
    rc = DSH_OK;
    do {
        if (didx.ensure(nq * nn * sizeof(uint32_t)) != hipSuccess || dval.ensure(nq * nn * sizeof(float)) != hipSuccess) {
            rc = fail(c, DSH_ENOMEM, "device allocation failed");
            break;
        }
        for (uint64_t q0 = qb; q0 < qe && rc == DSH_OK; q0 += qblock) {
            const uint64_t q1 = std::min(qe, q0 + qblock);
            if (nr) {
                PairJob j;
                j.estim = estim;
                j.result_type = result_type;
                j.k = k;
                j.rect = 1;
                j.ksinv_double = 1;
                j.row_begin = q0;
                j.row_end = q1;
                j.col_begin = rb;
                j.col_end = re;
                j.base_index = 0;
                j.d_out = (float *)rect.ptr;
                rc = run_pairs(c, j);
                if (rc) break;
            }
            hipError_t e = launch_topk(c->stream, (const float *)rect.ptr, q1 - q0, nr, q0, rb, descending, nn,
                                       overlap ? 1 : 0, (uint32_t *)didx.ptr + (q0 - qb) * nn,
                                       (float *)dval.ptr + (q0 - qb) * nn);
            if (e != hipSuccess) rc = fail(c, DSH_EIO, "k_topk: %s", hipGetErrorString(e));
        }
        if (rc) break;
        if (hipMemcpyAsync(idx_out, didx.ptr, nq * nn * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
            hipMemcpyAsync(val_out, dval.ptr, nq * nn * sizeof(float), hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
            hipStreamSynchronize(c->stream) != hipSuccess)
            rc = fail(c, DSH_EIO, "copy of neighbours failed");
    } while (0);
    didx.release();
    dval.release();
    return rc;
}

// cost model for balancing shards: a tile costs its dense planes plus ~5 plane-equivalents of
// finalize work (6.6 ms finalize vs 1.4 ms per plane on the C3 workload, profiles/r1f)
static void shard_bounds(dsh_ctx *c, uint32_t nshards, std::vector<uint32_t> &tb)
{
    const uint32_t NT = c->Npad / kTile;
    std::vector<double> rowcost(NT, 0.);
    double total = 0;
    for (uint32_t ti = 0; ti < NT; ++ti) {
        for (uint32_t tj = ti; tj < NT; ++tj) {
            int pb, pe;
            tile_planes(c, ti, tj, pb, pe);
            rowcost[ti] += (pe - pb) + c->shard_c0;
        }
        total += rowcost[ti];
    }
    // Contiguous tile-row ranges that minimise the largest shard (linear partition by bisection on the
    // limit + greedy fill).  Shard 0 belongs to the rank that also assembles the result (the un-permute,
    // about assembler_permille/1000 of a single-GPU pass): it carries that as extra cost.
    const double extra0 = nshards > 1 ? total * c->assembler_permille / 1000.0 : 0.0;
    auto fill = [&](double limit, std::vector<uint32_t> *out) -> bool {
        uint32_t ti = 0;
        for (uint32_t r = 0; r < nshards; ++r) {
            double acc = r == 0 ? extra0 : 0.0;
            if (out) (*out)[r] = ti;
            while (ti < NT && acc + rowcost[ti] <= limit) acc += rowcost[ti++];
        }
        if (out) (*out)[nshards] = NT;
        return ti == NT;
    };
    double lo = 0, hi = total + extra0;
    for (uint32_t ti = 0; ti < NT; ++ti) lo = std::max(lo, rowcost[ti]);  // a shard holds whole tile rows
    for (int it = 0; it < 60 && hi - lo > 1e-9 * (hi + 1); ++it) {
        const double mid = 0.5 * (lo + hi);
        if (fill(mid, nullptr)) hi = mid;
        else lo = mid;
    }
    tb.assign(nshards + 1, NT);
    fill(hi, &tb);  // (the greedy fill front-loads: later shards may be lighter, the maximum is what counts)
}

int dsh_shard_plan(dsh_ctx *c, int estim, uint32_t nshards, uint64_t *span_off)
{
    if (!c || !span_off || nshards == 0) return DSH_EINVAL;
    int rc = bind(c);
    if (rc) return rc;
    if ((rc = prepare(c, estim, 1))) return rc;
    std::vector<uint32_t> tb;
    shard_bounds(c, nshards, tb);
    for (uint32_t r = 0; r <= nshards; ++r)
        span_off[r] = dsh_tri_span(c->n, 0, std::min<uint64_t>(c->n, (uint64_t)tb[r] * kTile));
    return DSH_OK;
}

int dsh_dist_shard_device(dsh_ctx *c, int estim, int result_type, int k, uint32_t shard,
                          uint32_t nshards, void *d_span)
{
    if (!c || nshards == 0 || shard >= nshards) return DSH_EINVAL;
    int rc = bind(c);
    if (rc) return rc;
    if (!c->have_sketches) return fail(c, DSH_ESTATE, "no sketches loaded");
    reset_prof(c);
    if ((rc = prepare(c, estim, 1))) return rc;
    if (c->n < 2) return DSH_OK;
    std::vector<uint32_t> tb;
    shard_bounds(c, nshards, tb);
    PairJob j;
    j.estim = estim;
    j.result_type = result_type;
    j.k = k;
    j.rect = 0;
    j.sorted_rows = 1;
    j.row_begin = std::min<uint64_t>(c->n, (uint64_t)tb[shard] * kTile);
    j.row_end = std::min<uint64_t>(c->n, (uint64_t)tb[shard + 1] * kTile);
    j.col_begin = j.col_end = 0;
    j.base_index = dsh_tri_span(c->n, 0, j.row_begin);
    j.d_out = (float *)d_span;
    if (j.row_begin >= j.row_end) return DSH_OK;
    if (!d_span) return DSH_EINVAL;
    if ((rc = run_pairs(c, j))) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return DSH_OK;
}

int dsh_unpermute_device(dsh_ctx *c, const void *d_sorted_tri, void *d_out_tri)
{
    if (!c) return DSH_EINVAL;
    int rc = bind(c);
    if (rc) return rc;
    if (!whole_sorted(c)) return fail(c, DSH_ESTATE, "no sorted plan (call dsh_shard_plan / dsh_dist_shard_device first)");
    if (c->n < 2) return DSH_OK;
    if (!d_sorted_tri || !d_out_tri) return DSH_EINVAL;
    HIPCHK(c, launch_unpermute(c->stream, (const float *)d_sorted_tri, (const uint32_t *)c->perm.ptr,
                               c->unperm_gather ? (const uint32_t *)c->perm.ptr + c->n : nullptr, c->n, (float *)d_out_tri));
    return DSH_OK;
}

int dsh_unpermute_blocks_device(dsh_ctx *c, const void *d_stage, const uint64_t *block_off, uint32_t nshards, void *d_out_tri)
{
    if (!c || nshards == 0 || !block_off) return DSH_EINVAL;
    int rc = bind(c);
    if (rc) return rc;
    if (!whole_sorted(c)) return fail(c, DSH_ESTATE, "no sorted plan (call dsh_shard_plan / dsh_dist_shard_device first)");
    if (c->n < 2) return DSH_OK;
    if (!d_stage || !d_out_tri) return DSH_EINVAL;
    std::vector<uint32_t> tb;
    shard_bounds(c, nshards, tb);
    const uint32_t NT = c->Npad / kTile;
    std::vector<int64_t> delta(NT, 0);
    for (uint32_t r = 0; r < nshards; ++r) {
        const uint64_t off = dsh_tri_span(c->n, 0, std::min<uint64_t>(c->n, (uint64_t)tb[r] * kTile));
        for (uint32_t t = tb[r]; t < tb[r + 1]; ++t) delta[t] = (int64_t)block_off[r] - (int64_t)off;
    }
    HIPCHK(c, c->workbuf.ensure(NT * sizeof(int64_t)));
    HIPCHK(c, hipMemcpyAsync(c->workbuf.ptr, delta.data(), NT * sizeof(int64_t), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, launch_unpermute_staged(c->stream, (const float *)d_stage, (const uint32_t *)c->perm.ptr + c->n,
                                      (const int64_t *)c->workbuf.ptr, c->n, (float *)d_out_tri));
    HIPCHK(c, hipStreamSynchronize(c->stream));  // `delta` (pageable source) must outlive the copy
    return DSH_OK;
}

int dsh_unpermute_staged_device(dsh_ctx *c, const void *d_stage, uint64_t stride, uint32_t nshards, void *d_out_tri)
{
    if (!c || nshards == 0) return DSH_EINVAL;
    if (whole_sorted(c) && c->n >= 2) {  // the spans must fit their blocks
        std::vector<uint32_t> tb;
        shard_bounds(c, nshards, tb);
        for (uint32_t r = 0; r < nshards; ++r) {
            const uint64_t off = dsh_tri_span(c->n, 0, std::min<uint64_t>(c->n, (uint64_t)tb[r] * kTile));
            const uint64_t end = dsh_tri_span(c->n, 0, std::min<uint64_t>(c->n, (uint64_t)tb[r + 1] * kTile));
            if (end - off > stride) return fail(c, DSH_EINVAL, "stride %llu smaller than the span of shard %u", (unsigned long long)stride, r);
        }
    }
    std::vector<uint64_t> off(nshards);
    for (uint32_t r = 0; r < nshards; ++r) off[r] = (uint64_t)r * stride;
    return dsh_unpermute_blocks_device(c, d_stage, off.data(), nshards, d_out_tri);
}

void *dsh_alloc_host(size_t bytes)
{
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) return nullptr;
    return p;
}

void dsh_free_host(void *p)
{
    if (p) (void)hipHostFree(p);
}

int dsh_set_profiling(dsh_ctx *c, int enable)
{
    if (!c) return DSH_EINVAL;
    c->profiling = enable != 0;
    if (!c->profiling) c->finalize_stop = 0;  // the stop points exist for profiling runs only
    return DSH_OK;
}

int dsh_last_kernel_ms(dsh_ctx *c, double *pair_ms, double *fin_ms, double *prep_ms, uint32_t *launches)
{
    if (!c) return DSH_EINVAL;
    if (pair_ms) *pair_ms = c->pair_ms;
    if (fin_ms) *fin_ms = c->fin_ms;
    if (prep_ms) *prep_ms = c->prep_ms;
    if (launches) *launches = c->pair_launches;
    return DSH_OK;
}

int dsh_get_info(dsh_ctx *c, const char *name, int64_t *out)
{
    if (!c || !name || !out) return DSH_EINVAL;
    if (!std::strcmp(name, "planes")) *out = c->P;
    else if (!std::strcmp(name, "vlo")) *out = c->vlo;
    else if (!std::strcmp(name, "vhi")) *out = c->vhi;
    else if (!std::strcmp(name, "threshold")) *out = c->pbase + (int64_t)c->P;
    else if (!std::strcmp(name, "pbase")) *out = c->pbase;
    else if (!std::strcmp(name, "host_layout_us")) *out = (int64_t)c->host_layout_us;
    else if (!std::strcmp(name, "host_lists_us")) *out = (int64_t)c->host_lists_us;
    else if (!std::strcmp(name, "host_keys_wait_us")) *out = (int64_t)c->host_keys_wait_us;
    else if (!std::strcmp(name, "emax")) *out = c->emax;
    else if (!std::strcmp(name, "elow")) *out = c->elow;
    else if (!std::strcmp(name, "kc")) *out = c->kc;
    else if (!std::strcmp(name, "tile")) *out = kTile;
    else if (!std::strcmp(name, "npad")) *out = c->Npad;
    else if (!std::strcmp(name, "kpad")) *out = c->Kpad;
    else if (!std::strcmp(name, "cum_bytes")) *out = c->cum_bytes;
    else if (!std::strcmp(name, "sorted")) *out = c->planes_sorted;
    else if (!std::strcmp(name, "ncols")) *out = (int64_t)c->ncols;
    else if (!std::strcmp(name, "lockstep")) *out = c->planes_valid && use_lockstep(c) ? 1 : 0;
    else if (!std::strcmp(name, "tiles")) *out = (int64_t)c->htiles.size();
    else if (!std::strcmp(name, "bands")) *out = (int64_t)c->last_bands;
    else if (!std::strcmp(name, "words_per_plane")) *out = c->W;
    else if (!std::strcmp(name, "avg_tile_planes_x100")) {
        uint64_t tot = 0;
        for (const auto &t : c->htiles) tot += t.w - t.z;
        *out = c->htiles.empty() ? 0 : (int64_t)(tot * 100 / c->htiles.size());
    }
    else return fail(c, DSH_EINVAL, "unknown info %s", name);
    return DSH_OK;
}

int dsh_set_option(dsh_ctx *c, const char *name, int64_t v)
{
    if (!c || !name) return DSH_EINVAL;
    if (!std::strcmp(name, "kc")) {
        if (v != 0 && v != 16 && v != 32 && v != 64) return fail(c, DSH_EINVAL, "kc must be 0 (auto), 16, 32 or 64");
        c->kc_opt = (int)v;
        c->planes_valid = false;  // Kpad depends on kc
        return DSH_OK;
    }
    if (!std::strcmp(name, "cum_budget_bytes")) {
        if (v < (1 << 20)) return fail(c, DSH_EINVAL, "cum_budget_bytes too small");
        c->cum_budget = (uint64_t)v;
        return DSH_OK;
    }
    if (!std::strcmp(name, "unpermute_gather")) {
        c->unperm_gather = v != 0;
        return DSH_OK;
    }
    if (!std::strcmp(name, "knn_square_budget_bytes")) {
        if (v < 0) return fail(c, DSH_EINVAL, "knn_square_budget_bytes must be >= 0");
        c->knn_square_budget = (uint64_t)v;
        return DSH_OK;
    }
    if (!std::strcmp(name, "assembler_permille")) {
        if (v < 0 || v > 500) return fail(c, DSH_EINVAL, "assembler_permille out of range");
        c->assembler_permille = (int)v;
        return DSH_OK;
    }
    if (!std::strcmp(name, "shard_c0_x10")) {
        if (v < 0 || v > 10000) return fail(c, DSH_EINVAL, "shard_c0_x10 out of range");
        c->shard_c0 = (double)v / 10.0;
        return DSH_OK;
    }
    if (!std::strcmp(name, "sort")) {
        if (v < -1 || v > 1) return fail(c, DSH_EINVAL, "sort must be -1, 0 or 1");
        c->sort_mode = (int)v;
        return DSH_OK;
    }
    if (!std::strcmp(name, "range_sort_min_rows")) {
        if (v < 1) return fail(c, DSH_EINVAL, "range_sort_min_rows must be >= 1");
        c->range_sort_min_rows = (int)std::min<int64_t>(v, 1 << 30);
        return DSH_OK;
    }
    if (!std::strcmp(name, "ls_sort_items")) {
        c->ls_sort_items = v != 0;
        return DSH_OK;
    }
    if (!std::strcmp(name, "ls_item_chunks")) {
        if (v < 1 || v > (1 << 20)) return fail(c, DSH_EINVAL, "ls_item_chunks out of range");
        c->ls_item_chunks = (int)v;
        return DSH_OK;
    }
    if (!std::strcmp(name, "pair_lockstep")) {
        if (v < -1 || v > 1) return fail(c, DSH_EINVAL, "pair_lockstep must be -1, 0 or 1");
        c->pair_lockstep = (int)v;
        return DSH_OK;
    }
    if (!std::strcmp(name, "pair_mfma")) {
        c->pair_mfma = v != 0;
        return DSH_OK;
    }
    if (!std::strcmp(name, "finalize_stop")) {  // profiling only: results are meaningless while it is set
        if (v < 0 || v > 4) return fail(c, DSH_EINVAL, "finalize_stop must be in [0,4]");
        if (v && !c->profiling) return fail(c, DSH_ESTATE, "finalize_stop needs dsh_set_profiling(ctx, 1)");
        c->finalize_stop = (int)v;
        return DSH_OK;
    }
    if (!std::strcmp(name, "nsplit")) {
        if (v < 0 || v > 64) return fail(c, DSH_EINVAL, "nsplit must be in [0,64]");
        c->nsplit = (int)v;
        return DSH_OK;
    }
    if (!std::strcmp(name, "emax") || !std::strcmp(name, "elow")) {
        if (v < -1 || v > (int64_t)kMaxListSide) return fail(c, DSH_EINVAL, "%s must be in [-1,%u]", name, kMaxListSide);
        (name[1] == 'm' ? c->emax_opt : c->elow_opt) = (int)v;
        return DSH_OK;
    }
    if (!std::strcmp(name, "finalize_rowmajor")) {
        c->finalize_rowmajor = v != 0;
        return DSH_OK;
    }
    if (!std::strcmp(name, "xcd_swizzle")) {
        c->xcd_swizzle = v != 0;
        return DSH_OK;
    }
    return fail(c, DSH_EINVAL, "unknown option %s", name);
}


/* ---- multi-GPU exchange through RCCL (one context per rank; ranks may be processes or threads) ------------------ */
int dsh_comm_unique_id(void *id_out)
{
    if (!id_out) return DSH_EINVAL;
    Rccl *r = rccl();
    if (!r->h) return DSH_ENODEV;
    ncclUniqueId id;
    if (r->GetUniqueId(&id) != ncclSuccess) return DSH_EIO;
    std::memcpy(id_out, &id, sizeof id);
    return DSH_OK;
}

int dsh_comm_init(dsh_ctx *c, const void *unique_id, int rank, int world)
{
    if (!c || !unique_id || world < 1 || rank < 0 || rank >= world) return DSH_EINVAL;
    int rc = bind(c);
    if (rc) return rc;
    Rccl *r = rccl();
    if (!r->h) return fail(c, DSH_ENODEV, "RCCL is not available (%s)", r->err.c_str());
    if (c->comm) {  // re-initialisation: nothing of the old communicator may still be in flight on either stream
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (c->copy_stream) HIPCHK(c, hipStreamSynchronize(c->copy_stream));
        NCCLCHK(c, r->CommDestroy(c->comm));
        c->comm = nullptr;
    }
    ncclUniqueId id;
    std::memcpy(&id, unique_id, sizeof id);
    NCCLCHK(c, r->CommInitRank(&c->comm, world, id, rank));
    c->comm_rank = rank;
    c->comm_world = world;
    return DSH_OK;
}

int dsh_comm_destroy(dsh_ctx *c)
{
    if (!c) return DSH_EINVAL;
    int rc = bind(c);
    if (rc) return rc;
    if (c->comm) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (c->copy_stream) HIPCHK(c, hipStreamSynchronize(c->copy_stream));  // dsh_collect_parts_async sends there
        NCCLCHK(c, rccl()->CommDestroy(c->comm));
        c->comm = nullptr;
    }
    c->comm_rank = 0;
    c->comm_world = 1;
    return DSH_OK;
}

int dsh_comm_rank(const dsh_ctx *c, int *rank, int *world)
{
    if (!c) return DSH_EINVAL;
    if (rank) *rank = c->comm_rank;
    if (world) *world = c->comm ? c->comm_world : 1;
    return c->comm ? DSH_OK : DSH_ESTATE;
}

static int collect_spans(dsh_ctx *c, uint64_t n, const uint64_t *bounds, const void *d_local, void *d_final, int dst)
{
    const int world = c->comm ? c->comm_world : 1, rank = c->comm ? c->comm_rank : 0;
    if (!bounds || dst < 0 || dst >= world) return fail(c, DSH_EINVAL, "bad bounds / destination rank");
    if (bounds[0] != 0 || bounds[world] != n) return fail(c, DSH_EINVAL, "bounds must run from 0 to n over the %d ranks", world);
    for (int r = 0; r < world; ++r)
        if (bounds[r] > bounds[r + 1]) return fail(c, DSH_EINVAL, "bounds not monotone at rank %d", r);
    const uint64_t mine = dsh_tri_span(n, bounds[rank], bounds[rank + 1]);
    if (rank == dst) {
        if (!d_final) return DSH_EINVAL;
        float *own = (float *)d_final + dsh_tri_span(n, 0, bounds[rank]);
        if (mine && d_local && d_local != (const void *)own)  // computed elsewhere: put it into place
            HIPCHK(c, hipMemcpyAsync(own, d_local, mine * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
    } else if (mine && !d_local) {
        return DSH_EINVAL;
    }
    if (world == 1) return DSH_OK;
    Rccl *r = rccl();
    // one message per peer, all of them in one group: the destination's links are busy at once, every span lands
    // at its final place (the ranks' row ranges are contiguous spans of the packed triangle)
    NCCLCHK(c, r->GroupStart());
    if (rank == dst) {
        for (int src = 0; src < world; ++src) {
            const uint64_t cnt = dsh_tri_span(n, bounds[src], bounds[src + 1]);
            if (src == dst || cnt == 0) continue;
            ncclResult_t e = r->Recv((float *)d_final + dsh_tri_span(n, 0, bounds[src]), cnt, ncclFloat32, src, c->comm, c->stream);
            if (e != ncclSuccess) {
                (void)r->GroupEnd();
                return fail(c, DSH_EIO, "ncclRecv: %s", r->GetErrorString(e));
            }
        }
    } else if (mine) {
        ncclResult_t e = r->Send(d_local, mine, ncclFloat32, dst, c->comm, c->stream);
        if (e != ncclSuccess) {
            (void)r->GroupEnd();
            return fail(c, DSH_EIO, "ncclSend: %s", r->GetErrorString(e));
        }
    }
    NCCLCHK(c, r->GroupEnd());
    return DSH_OK;
}

int dsh_collect_spans_async(dsh_ctx *c, uint64_t n, const uint64_t *bounds, const void *d_local, void *d_final, int dst)
{
    if (!c) return DSH_EINVAL;
    int rc = bind(c);
    if (rc) return rc;
    if (!c->comm && !(bounds && bounds[0] == 0 && bounds[1] == n)) return fail(c, DSH_ESTATE, "dsh_comm_init first");
    return collect_spans(c, n, bounds, d_local, d_final, dst);
}

int dsh_collect_spans(dsh_ctx *c, uint64_t n, const uint64_t *bounds, const void *d_local, void *d_final, int dst)
{
    int rc = dsh_collect_spans_async(c, n, bounds, d_local, d_final, dst);
    if (rc) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return DSH_OK;
}

int dsh_collect_parts_async(dsh_ctx *c, uint64_t n, const uint64_t *bounds, uint32_t nparts, const void *d_local,
                            void *d_final, int dst)
{
    if (!c || !bounds || nparts == 0) return DSH_EINVAL;
    int rc = bind(c);
    if (rc) return rc;
    const int world = c->comm ? c->comm_world : 1, rank = c->comm ? c->comm_rank : 0;
    if (!c->comm && !(bounds[0] == 0 && bounds[1] == n)) return fail(c, DSH_ESTATE, "dsh_comm_init first");
    if (dst < 0 || dst >= world) return fail(c, DSH_EINVAL, "bad destination rank %d", dst);
    if (bounds[0] != 0 || bounds[world] != n) return fail(c, DSH_EINVAL, "bounds must run from 0 to n over the %d ranks", world);
    for (int r = 0; r < world; ++r)
        if (bounds[r] > bounds[r + 1]) return fail(c, DSH_EINVAL, "bounds not monotone at rank %d", r);
    if (rank == dst && !d_final) return DSH_EINVAL;
    // every rank's parts, from the same function the compute used (dsh_range_parts): both sides of a message agree
    std::vector<std::vector<uint64_t>> parts((size_t)world);
    size_t maxparts = 0;
    for (int r = 0; r < world; ++r) {
        range_parts(n, bounds[r], bounds[r + 1], nparts, parts[r]);
        maxparts = std::max(maxparts, parts[r].size() - 1);
    }
    const size_t myparts = parts[rank].size() - 1;
    const uint64_t mine = dsh_tri_span(n, bounds[rank], bounds[rank + 1]);
    if (mine && c->parts_done != myparts)
        return fail(c, DSH_ESTATE, "dsh_dist_rows_parts_device_async(%u parts) of this rank's rows must come first", nparts);
    Rccl *rc_ = world > 1 ? rccl() : nullptr;
    const uint64_t my_off = dsh_tri_span(n, 0, bounds[rank]);
    for (size_t q = 0; q < maxparts; ++q) {
        // round q: part q of every rank.  The copy stream joins this rank's "part q done" event; the transfer then runs
        // there while the ctx stream computes part q+1
        if (q < myparts && mine) HIPCHK(c, hipStreamWaitEvent(c->copy_stream, c->ev_part[q], 0));
        if (rank == dst && q < myparts && mine && d_local) {
            const uint64_t o0 = dsh_tri_span(n, 0, parts[rank][q]) - my_off, cnt = dsh_tri_span(n, parts[rank][q], parts[rank][q + 1]);
            float *own = (float *)d_final + my_off + o0;
            if (cnt && (const float *)d_local + o0 != own)
                HIPCHK(c, hipMemcpyAsync(own, (const float *)d_local + o0, cnt * sizeof(float), hipMemcpyDeviceToDevice, c->copy_stream));
        }
        if (world == 1) continue;
        NCCLCHK(c, rc_->GroupStart());
        ncclResult_t e = ncclSuccess;
        if (rank == dst) {
            for (int src = 0; src < world && e == ncclSuccess; ++src) {
                if (src == dst || q >= parts[src].size() - 1) continue;
                const uint64_t cnt = dsh_tri_span(n, parts[src][q], parts[src][q + 1]);
                if (cnt) e = rc_->Recv((float *)d_final + dsh_tri_span(n, 0, parts[src][q]), cnt, ncclFloat32, src, c->comm, c->copy_stream);
            }
        } else if (q < myparts) {
            const uint64_t o0 = dsh_tri_span(n, 0, parts[rank][q]) - my_off, cnt = dsh_tri_span(n, parts[rank][q], parts[rank][q + 1]);
            if (cnt) {
                if (!d_local) e = ncclInvalidArgument;
                else e = rc_->Send((const float *)d_local + o0, cnt, ncclFloat32, dst, c->comm, c->copy_stream);
            }
        }
        if (e != ncclSuccess) {
            (void)rc_->GroupEnd();
            return fail(c, DSH_EIO, "ncclSend/ncclRecv: %s", rc_->GetErrorString(e));
        }
        NCCLCHK(c, rc_->GroupEnd());
    }
    return DSH_OK;
}

int dsh_allgather_device(dsh_ctx *c, const void *d_send, uint64_t bytes_per_rank, void *d_recv)
{
    if (!c || (bytes_per_rank && (!d_send || !d_recv))) return DSH_EINVAL;
    int rc = bind(c);
    if (rc) return rc;
    if (!c->comm) return fail(c, DSH_ESTATE, "dsh_comm_init first");
    if (bytes_per_rank) NCCLCHK(c, rccl()->AllGather(d_send, d_recv, bytes_per_rank, ncclUint8, c->comm, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return DSH_OK;
}

int dsh_dist_collect(dsh_ctx *c, int estim, int result_type, int k, const uint64_t *bounds, int dst, float *out)
{
    if (!c || !bounds) return DSH_EINVAL;
    int rc = bind(c);
    if (rc) return rc;
    if (!c->have_sketches) return fail(c, DSH_ESTATE, "no sketches loaded");
    const int world = c->comm ? c->comm_world : 1, rank = c->comm ? c->comm_rank : 0;
    if (dst < 0 || dst >= world) return fail(c, DSH_EINVAL, "bad destination rank %d", dst);
    const uint64_t n = c->n, total = dsh_tri_span(n, 0, n);
    if (bounds[0] != 0 || bounds[world] != n) return fail(c, DSH_EINVAL, "bounds must run from 0 to n over the %d ranks", world);
    const uint64_t mine = dsh_tri_span(n, bounds[rank], bounds[rank + 1]);
    void *d_local = nullptr;
    if (rank == dst) {
        if (total && !out) return DSH_EINVAL;
        HIPCHK(c, c->gather_full.ensure(std::max<uint64_t>(total, 1) * sizeof(float)));
        d_local = (float *)c->gather_full.ptr + dsh_tri_span(n, 0, bounds[rank]);  // computed in place
    } else {
        HIPCHK(c, c->gather_local.ensure(std::max<uint64_t>(mine, 1) * sizeof(float)));
        d_local = c->gather_local.ptr;
    }
    if (mine && (rc = dsh_dist_rows_device_async(c, estim, result_type, k, bounds[rank], bounds[rank + 1], d_local))) return rc;
    if ((rc = collect_spans(c, n, bounds, d_local, rank == dst ? c->gather_full.ptr : nullptr, dst))) return rc;
    if (rank == dst && total)
        HIPCHK(c, hipMemcpyAsync(out, c->gather_full.ptr, total * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return DSH_OK;
}

}  // extern "C"

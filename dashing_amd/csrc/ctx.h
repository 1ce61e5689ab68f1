// ctx.h -- the per-GPU context behind the opaque dsh_ctx of include/dashing_hip.h, and the internal entry points the
// translation units of libdashing_hip.so share:
//   abi.hip       context, resident sketches, sketch waist, cardinalities, compare entry points, tickets, options
//   engine.hip    prepare() (per-sketch pass, layout, bit-planes, position index) and run_pairs() (tile kernel + k_finalize)
//   knn.hip       dsh_knn
//   exchange.hip  RCCL: dsh_comm_*, dsh_collect_*, dsh_allgather_device, dsh_dist_collect
//   plan.cpp      the pure-host planner (layout, tiles, bands, parts, work items, row partitions)
#pragma once
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>  // types only: the library is dlopen'ed at dsh_comm_init (single-GPU users never load it)

#include <cstdarg>
#include <cstdio>
#include <string>
#include <vector>

#include "../../include/dashing_hip.h"
#include "kernels.h"
#include "plan.h"

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
    uint64_t pass_gen = 0, lay_gen = 0; // per-sketch passes so far; the one the layout's per-column data was built from
    DevBuf card, planes, cum, tiles, items, outbuf, seqbuf, workbuf, exc, excv, exc_n, keys, perm, tailhist;
    // copy-out pipeline of dsh_dist_rows_async: results alternate between two device buffers; the copy of call b to the
    // host runs on its own stream while the kernels of call b+1 fill the other buffer
    hipStream_t copy_stream = nullptr;
    hipStream_t aux_stream = nullptr;   // prepare(): the column index is built next to the bit-plane transform
    hipStream_t place_stream = nullptr; // destination of an exchange: received rows are put into place beside the next round's transfer
    hipEvent_t ev_aux_fork = nullptr, ev_aux_join = nullptr;
    bool aux_join_pending = false;
    DevBuf outbuf2[2];
    hipEvent_t ev_filled[2] = {nullptr, nullptr};  // kernels of the call that filled outbuf2[b] done (recorded on stream)
    hipEvent_t ev_drained[2] = {nullptr, nullptr}; // copy out of outbuf2[b] done (recorded on copy_stream)
    bool drained_pending[2] = {false, false};
    unsigned out_turn = 0;
    std::vector<hipEvent_t> tickets;    // dsh_event_record ring: slot t % 64 holds {mark on the ctx stream, mark on the copy stream}
    uint64_t ticket_next = 0;
    // multi-GPU exchange (dsh_comm_*): an RCCL communicator over the ranks' contexts
    ncclComm_t comm = nullptr;
    int comm_rank = 0, comm_world = 1;
    DevBuf gather_full, gather_local;   // dsh_dist_collect: the assembled matrix on the destination rank / this rank's span
    // dsh_exchange_*: on the destination, the row-sorted spans as they arrive and the sources' key order + row offsets
    DevBuf xch_stage, xch_tab;
    DevBuf place_tab;                   // dsh_exchange_place_device's own tables (ctx stream; xch_tab belongs to the copy stream)
    PinBuf pin_xch;
    hipEvent_t ev_xch_tab = nullptr;
    hipEvent_t ev_first_tiles = nullptr;       // the tile kernel of the last call's first band has run
    bool xch_recv_gated = false;               // (info) the last collect of a destination waited for its tile kernel
    int xch_recv_gate = -1;                    // option: -1 auto | 0 | 1 (exchange.hip, the destination's receives)
    hipEvent_t ev_place_done = nullptr;        // the last placement of a collect (the copy stream joins it)
    std::vector<hipEvent_t> ev_round;          // round q of a collect has arrived (place_stream waits for it)
    bool xch_tab_in_flight = false;
    bool pass_from_zero = false;        // the next per-sketch pass covers every sketch (the destination of an exchange)
    DevBuf hist;                        // [n][64] per-sketch register histograms (k_selfhist_card -> k_card_from_hist)
    hipEvent_t ev_keys = nullptr;       // the keys have reached the host
    DevBuf cidx_rec, cidx_ent;          // position index of the column blocks of the current layout (k_build_colindex)
    uint32_t nbuckets = 0, ent_stride = 0;
    DevBuf colS_n, colS_key, colS_card, colS_th, colS_rl;  // per column of the layout, in layout order (k_finalize's inputs)
    uint32_t rl_stride = 0;             // entries of a compact list row (emax + elow)
    // column layout of the cached plane matrix (plan.h) and the plan of the last compare call
    dsh::plan::Layout lay;
    bool lay_built = false;             // `lay` holds a layout (whatever happened to the planes since)
    dsh::plan::PairPlan pp;
    std::vector<hipEvent_t> ev_part;    // part q complete (recorded on the ctx stream by the last call with parts)
    uint32_t parts_done = 0;            // parts of the last dsh_dist_rows_parts_device_async call
    // part signalling (kernels.h kSig*): k_finalize announces the parts from inside ONE launch per band; the copy stream
    // waits for a part's flag with hipStreamWaitValue32 instead of an event between launches
    DevBuf sig;
    PinBuf pin_sig;                     // the parts' tile totals on their way to the device
    hipEvent_t ev_sig = nullptr;        // their upload has run (the staging is rewritten by the next call)
    bool sig_in_flight = false;
    uint32_t sig_gen = 0;               // generation of the last call with parts: the value its flags take
    bool parts_signalled = false;       // the last call with parts used flags (else events)
    int finalize_signal = -1;           // option: -1 auto (on where the device supports stream wait-value), 0 events, 1 flags
    int can_wait_value = -1;            // hipDeviceAttributeCanUseStreamWaitValue, queried once
    int wall_clock_khz = 0;
    PinBuf pin_keys;                    // host copy of the per-sketch keys (valid while the per-sketch pass is), page-locked:
    const uint32_t *hk32 = nullptr;     // the copy is a direct DMA and the host only waits for ev_keys
    bool hk32_valid = false;
    hipEvent_t ev_perm = nullptr;       // upload of pin_perm done (it is rewritten by the next layout)
    bool perm_in_flight = false;
    double host_layout_us = 0, host_lists_us = 0, host_keys_wait_us = 0;  // host time of the last call (dsh_get_info)
    uint32_t *pin_perm = nullptr;       // page-locked copy of lay.perm: its upload is then truly asynchronous
    size_t pin_perm_cap = 0;
    DevBuf rowoff;                      // (row-sorted parts) offset of the row at layout position s in the rank's buffer
    PinBuf pin_rowoff;
    PinBuf pin_work;                    // sketch work list of the call in flight
    hipEvent_t ev_work = nullptr;
    bool work_in_flight = false;
    PinBuf pin_lists;                   // tiles then items of the call in flight
    hipEvent_t ev_lists = nullptr;      // recorded after their upload; waited on before they are rewritten
    bool lists_in_flight = false;
    uint32_t W = 0, Kpad = 0;           // words per plane; plane rows padded to whole LDS stages
    int emax = 0, elow = 0, cum_bytes = 4;
    // options
    int kc = 16;      // k-rows per LDS stage in effect (set by prepare from kc_opt)
    int kc_opt = 0;   // 0 auto: 32 where a plane is at least that long (p >= 10), else 16 (profiles/r3f/lockstep_ab.jsonl)
    int emax_opt = -1;  // cap of the listed upper tail; -1: auto_list_cap(p, true)
    int elow_opt = -1;  // cap of the listed lower tail; -1: auto_list_cap(p, false)
    int part_band_tiles = 2048;       // a part of at least this many tiles gets its own launch of the tile kernel (plan.cpp)
    int overflow_frag_permille = 500;  // overflow fragments of the tile kernel (plan.h, Tuning): 0 = never
    int tail_bands = 2;  // option xch_tail_bands: small jobs with parts have their tile kernel cut at whole rounds (plan.h, Tuning)
    std::vector<double> part_ready_ms;  // (profiling) when each part of the last call with parts was final, from the call's start
    std::vector<uint64_t> part_floats;  // and the floats of the rank's buffer it holds
    size_t last_bands = 0;            // tile-kernel launches groups (bands) of the last dist call
    uint64_t cum_budget = 8ull << 30;  // scratch for C(v) per pair slot: larger jobs run in bands (2 -> 8 GiB: -1.5 % at 100 000 x p=10)
    int sort_mode = -1;  // -1 auto (key-ordered columns for triangle calls of >= range_sort_min_rows rows), 0 never
    int range_sort_min_rows = 1024;  // smaller row ranges keep the cached identity layout (a rebuild costs more than it saves)
    uint64_t knn_square_budget = (uint64_t)96 << 30;  // all-vs-all kNN keeps an n x n float matrix in HBM up to this size
    int pair_mfma = 0;  // WHAT-IF only (built with `make WHATIF=1`): 1 = the AND+popcount tile kernel on the matrix cores
    int finalize_stop = 0;  // profiling only: k_finalize leaves after phase 1..4 (results are then meaningless)
    int finalize_timing = 0;  // profiling only: the s_memtime-stamped instance of k_finalize (same results, per-phase cycles)
    DevBuf phase_cyc;       // 8 x u64 of the last call with finalize_timing
    int nsplit = 0;  // plane-range splits per tile; 0 = auto (aim at >= 16 items per workgroup slot)
    // dsh_sketch_fastx_batch_async: raw FASTA bytes of a batch, the decoder's tables and scratch (kernels_fastx.hip)
    DevBuf rawbuf, fx_tab, fx_summ, fx_state, fx_declen, fx_status;
    PinBuf pin_fx;
    hipEvent_t ev_fx = nullptr;         // the upload of pin_fx has run
    bool fx_in_flight = false;
    // diagnostics (dsh_diag_spin_*): a kernel that waits like an RCCL receive kernel, on a stream of its own
    uint32_t *spin_flag = nullptr;      // page-locked, mapped
    hipStream_t spin_stream = nullptr;
    bool spin_running = false;
    // profiling
    bool profiling = false;
    double place_ms = 0;  // (profiling) device time of the last dsh_exchange_place_device's placement kernel
    double pair_ms = 0, fin_ms = 0, prep_ms = 0, sketch_ms = 0, fastx_ms = 0;
    uint32_t pair_launches = 0;
    std::vector<hipEvent_t> ev_pool;
    size_t ev_used = 0;
};

namespace dsh {

inline int fail(dsh_ctx *c, int code, const char *fmt, ...)
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

#define HIPCHK(c, expr)                                                                        \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess) {                                                                \
            (void)hipGetLastError(); /* reported here: a later launch's check must not find it again */ \
            return dsh::fail((c), e_ == hipErrorOutOfMemory ? DSH_ENOMEM : DSH_EIO, "%s: %s",  \
                             #expr, hipGetErrorString(e_));                                    \
        }                                                                                      \
    } while (0)

inline int bind(dsh_ctx *c)
{
    HIPCHK(c, hipSetDevice(c->device));
    return DSH_OK;
}

inline hipEvent_t next_event(dsh_ctx *c)
{
    if (c->ev_used == c->ev_pool.size()) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return nullptr;
        c->ev_pool.push_back(e);
    }
    return c->ev_pool[c->ev_used++];
}

// slots [first, first + cnt) inside [0, total) -- written so that first + cnt cannot wrap
inline bool slots_ok(uint64_t first, uint64_t cnt, uint64_t total) { return first <= total && cnt <= total - first; }

inline bool use_lockstep(const dsh_ctx *c)
{
    // wherever a plane is at least one chunk (p >= 9): since the kernel needs one barrier per k-row it beats the
    // free-running one at every precision (profiles/r3f/lockstep_ab.jsonl: -5 % at p = 10 ... -17 % at p = 16)
    return !(c->pair_mfma || c->W < (uint32_t)c->kc);
}

inline bool whole_sorted(const dsh_ctx *c) { return c->planes_valid && c->lay.whole; }

inline void invalidate(dsh_ctx *c)
{
    c->planes_valid = false;
    c->card_estim = -1;
    c->hk32_valid = false;
}

// hipStreamWaitValue32 on this device (asked once): what lets the parts of a call announce themselves from inside k_finalize
inline bool device_can_wait_value(dsh_ctx *c)
{
    if (c->can_wait_value < 0) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeCanUseStreamWaitValue, c->device) != hipSuccess) v = 0;
        c->can_wait_value = v;
        int khz = 0;
        if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, c->device) != hipSuccess) khz = 0;
        c->wall_clock_khz = khz;
    }
    return c->can_wait_value == 1;
}
// a call with parts will signal them (engine.hip: run_pairs)
inline bool parts_will_signal(dsh_ctx *c) { return !c->finalize_timing && !c->finalize_stop && c->finalize_signal != 0 && device_can_wait_value(c); }

inline void reset_prof(dsh_ctx *c)
{
    c->pair_ms = c->fin_ms = c->prep_ms = 0;
    c->pair_launches = 0;
    c->ev_used = 0;
}

// one pass of the compare path over a set of pairs (engine.hip)
struct PairJob {
    int estim, result_type, k;
    int rect;
    int sorted_rows = 0;  // rows (and the output) are in sorted plane-column order (shards)
    int square = 0;       // full triangle, each value written at (i,j) and (j,i) of an n x n matrix
    uint32_t nparts = 0;  // > 0: triangle rows in (at most) this many parts of a key-ordered layout, an event per part
    int rowsorted = 0;    // with nparts: row-sorted parts -- d_out holds the rows in key order (plan.h), not the final span
    int knn = 0;          // band of the key-ordered triangle for the nearest-neighbour selection: d_out = V, d_out2 = Vt
    float *d_out2 = nullptr;
    uint64_t knn_ld = 0, knn_rows = 0;
    int ksinv_double = 0; // 1./k as a double (nndist_loop, src/sketch_and_cmp.h:729) instead of the float of dist_loop (:797)
    uint64_t row_begin, row_end, col_begin, col_end;
    uint64_t base_index;
    float *d_out;
    std::vector<uint64_t> extra;  // (with nparts) further row segments {b0, e0, ...} behind row_end: a row set (plan.h)
};

// cardinalities + thresholds/lists + planes + position index for the current sketch matrix.  want_sorted < 0: whatever
// is cached.  card_only: the per-sketch pass alone.
int prepare(dsh_ctx *c, int estim, int want_sorted, bool card_only = false, uint64_t want_rb = 0,
            uint64_t want_re = ~0ull, uint32_t nparts = 1, int rowsorted = 0, const std::vector<uint64_t> *extra = nullptr);
int run_pairs(dsh_ctx *c, const PairJob &job);

// exchange.hip: waits for both streams of the communicator's traffic and destroys it (no-op without one)
int comm_release(dsh_ctx *c);

}  // namespace dsh

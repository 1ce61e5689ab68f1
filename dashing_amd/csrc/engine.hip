// engine.hip -- the device side of a compare pass: prepare() builds what a layout needs (per-sketch pass, key-ordered
// columns, bit-planes, position index), run_pairs() launches the tile kernel and k_finalize over the plan of plan.cpp.
// Together they replace dist_loop / perform_core_op (src/sketch_and_cmp.h:785-880, :699-710) and the compare side of
// dm::parallel_fill (distmat/distmat.h:459-512).
#include <algorithm>
#include <chrono>
#include <cstring>

#include "ctx.h"

namespace dsh {

int prepare(dsh_ctx *c, int estim, int want_sorted, bool card_only, uint64_t want_rb, uint64_t want_re, uint32_t nparts,
            int rowsorted, const std::vector<uint64_t> *extra_in)
{
    static const std::vector<uint64_t> kNoExtra;
    std::vector<uint64_t> extra_cached;
    const std::vector<uint64_t> *extra = extra_in ? extra_in : &kNoExtra;
    if (!c->have_sketches) return fail(c, DSH_ESTATE, "no sketches loaded");
    if (estim < 0 || estim > 2) return fail(c, DSH_EINVAL, "bad estimator %d", estim);
    if (want_sorted < 0) {  // "whatever is cached"
        want_sorted = c->lay.sorted;
        want_rb = c->lay.rb;
        want_re = c->lay.re;
        rowsorted = 0;
        extra_cached = c->lay.extra;
        extra = &extra_cached;
    }
    if (!want_sorted) rowsorted = 0, extra = &kNoExtra;
    if (want_re > c->n) want_re = c->n;
    if (!want_sorted) want_rb = 0, want_re = c->n;
    if (want_rb > want_re) want_rb = want_re;
    const uint64_t n = c->n;
    std::vector<uint64_t> parts, rs_pos;
    if (want_rb >= want_re) extra = &kNoExtra;
    if (want_sorted && !rowsorted) {
        if (extra->empty()) plan::range_parts(n, want_rb, want_re, std::max<uint32_t>(nparts, 1), parts);
        else parts = {want_rb, want_re};  // (a range with extra segments is ONE part)
    }
    if (rowsorted) plan::rowsorted_part_positions(n, want_rb, want_re, std::max<uint32_t>(nparts, 1), rs_pos, extra);
    const int emax_new = c->emax_opt >= 0 ? std::min<int>(c->emax_opt, (int)kMaxListSide) : plan::auto_list_cap(c->p, true);
    const int elow_new = c->elow_opt >= 0 ? std::min<int>(c->elow_opt, (int)kMaxListSide) : plan::auto_list_cap(c->p, false);
    if (emax_new != c->emax || elow_new != c->elow) {  // thresholds and lists (hence planes) depend on them
        c->planes_valid = false;
        c->card_estim = -1;
    }
    c->emax = emax_new;
    c->elow = elow_new;
    // (the layout keeps cardinalities, keys and lists per column: it is only "the same" while it was built from the
    // per-sketch pass that is current -- another estimator or other list caps start a new pass)
    const bool same_layout = c->planes_valid && c->lay_gen == c->pass_gen && c->lay.sorted == want_sorted &&
                             (!want_sorted || (c->lay.rb == want_rb && c->lay.re == want_re && c->lay.rowsorted == rowsorted &&
                                               c->lay.extra == *extra && (rowsorted ? c->lay.part_w == rs_pos : c->lay.parts == parts)));
    // sketches the per-sketch pass has to cover: a row range of the triangle never looks at the sketches before it
    const uint64_t need_from = (card_only || !want_sorted || c->pass_from_zero) ? 0 : want_rb;
    const bool have_pass = c->card_estim == estim && c->card_from <= need_from;
    if (have_pass && (card_only || same_layout)) return DSH_OK;
    const bool keep_layout = same_layout && have_pass;  // (a new pass below also outdates the layout's per-column data)
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
        ++c->pass_gen;
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
#ifdef DSH_EXPERIMENT_REUSE_LAYOUT
    // EXPERIMENT (make EXPERIMENT=1; never in the product library): the upper bound of what ANY device-built layout could
    // save.  The sketches of a re-attached matrix are the same bytes, so the layout of the previous call -- host tables AND
    // the permutation on the device -- is still right: skip the host's part entirely (the wait for the keys, the radix sort,
    // the block statistics, the uploads) and go straight from the per-sketch pass to the index build and the transform.
    // Results stay correct only while every call sees the same registers (bench.py's timed loop).  DESIGN.md section 8.
    const bool reuse_host_layout = !keep_layout && c->lay_built && c->lay.n == n && c->lay.sorted == want_sorted &&
                                   (!want_sorted || (c->lay.rb == want_rb && c->lay.re == want_re && c->lay.rowsorted == rowsorted &&
                                                     c->lay.extra == *extra && (rowsorted ? c->lay.part_w == rs_pos : c->lay.parts == parts)));
#else
    constexpr bool reuse_host_layout = false;
#endif
    if (!keep_layout) {
        if (c->p > kMaxPCompare)
            return fail(c, DSH_EINVAL, "the compare path takes p <= %d (p=%d: sketching and cardinalities only)", kMaxPCompare, c->p);
        // the keys are downloaded once per per-sketch pass: a later layout (next row block) needs no
        // device round trip and so does not wait for the work still queued on the stream
        const auto t_h0 = std::chrono::steady_clock::now();
        if (!c->hk32_valid && !reuse_host_layout) {
            HIPCHK(c, hipEventSynchronize(c->ev_keys));
            c->hk32_valid = true;
        }
        const auto t_h1 = std::chrono::steady_clock::now();
        c->host_keys_wait_us = std::chrono::duration<double, std::micro>(t_h1 - t_h0).count();
        const uint32_t *k32 = c->hk32;
        for (uint64_t i = c->card_from; i < n && !reuse_host_layout; ++i)  // (the sketches the per-sketch pass covered)
            if (plan::key_bad(k32[i]))
                return fail(c, DSH_EINVAL, "sketch %llu holds a register value above %d (= 64 - p + 1): not an HLL of precision %d (corrupt or foreign .hll?)",
                            (unsigned long long)i, 64 - c->p + 1, c->p);
        c->planes_valid = false;  // (the cached layout is overwritten from here on)
        plan::Layout &L = c->lay;
        if (!reuse_host_layout)
            plan::build_layout(k32, n, want_sorted, want_rb, want_re, parts, L, rowsorted ? std::max<uint32_t>(nparts, 1) : 0,
                               extra->empty() ? nullptr : extra);
        c->lay_built = true;
        c->cum_bytes = c->p <= 15 ? 2 : 4;
        const uint64_t m = 1ull << c->p;
        c->W = (uint32_t)std::max<uint64_t>(1, m / 32);
        c->kc = c->kc_opt ? c->kc_opt : (c->W >= 32 ? 32 : 16);
        if (want_sorted && !reuse_host_layout) {
            // perm, then (whole collection only) its inverse for the un-permute of the shard path
            const uint64_t nperm = L.perm.size();
            HIPCHK(c, c->perm.ensure(std::max<uint64_t>(nperm, 1) * sizeof(uint32_t)));
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
                std::memcpy(c->pin_perm, L.perm.data(), nperm * sizeof(uint32_t));
                HIPCHK(c, launch_upload(c->stream, c->perm.ptr, c->pin_perm, nperm * sizeof(uint32_t)));
                if (L.rowsorted) {  // where the rows of the rank's buffer start
                    const size_t rb_ = L.rowoff.size() * sizeof(uint64_t);
                    HIPCHK(c, c->pin_rowoff.ensure(rb_));
                    HIPCHK(c, c->rowoff.ensure(rb_));
                    std::memcpy(c->pin_rowoff.ptr, L.rowoff.data(), rb_);
                    HIPCHK(c, launch_upload(c->stream, c->rowoff.ptr, c->pin_rowoff.ptr, rb_));
                }
                if (!c->ev_perm) HIPCHK(c, hipEventCreateWithFlags(&c->ev_perm, hipEventDisableTiming));
                HIPCHK(c, hipEventRecord(c->ev_perm, c->stream));
                c->perm_in_flight = true;
            }
        }
        const uint32_t NT = L.Npad / kTile;
        // position index of every column block (the list joins of k_finalize): 79 workgroups at C3 -- built on a
        // second stream next to the bit-plane transform, which fills the chip on its own; both only need the
        // per-sketch pass and the permutation, the tile kernels wait for both
        c->nbuckets = (uint32_t)std::min<uint64_t>(2 * m, kMaxBuckets);  // (position group, upper | lower tail)
        c->ent_stride = std::max<uint32_t>(1, kTile * (uint32_t)(c->emax + c->elow));
        c->rl_stride = std::max<uint32_t>(1, (uint32_t)(c->emax + c->elow));
        const size_t nbk = std::max<size_t>(NT, 1);
        HIPCHK(c, c->cidx_rec.ensure(nbk * c->nbuckets * (size_t)(colindex_inline(c->p, c->rl_stride) + 1) * sizeof(uint32_t)));
        HIPCHK(c, c->cidx_ent.ensure(nbk * c->ent_stride * sizeof(uint32_t)));
        HIPCHK(c, c->colS_n.ensure(nbk * kTile * sizeof(uint32_t)));
        HIPCHK(c, c->colS_key.ensure(nbk * kTile * sizeof(uint32_t)));
        HIPCHK(c, c->colS_card.ensure(nbk * kTile * sizeof(double)));
        HIPCHK(c, c->colS_th.ensure(nbk * kTile * 64));
        HIPCHK(c, c->colS_rl.ensure(nbk * kTile * (size_t)c->rl_stride * sizeof(uint32_t)));
        c->host_layout_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_h1).count();
        HIPCHK(c, hipEventRecord(c->ev_aux_fork, c->stream));
        HIPCHK(c, hipStreamWaitEvent(c->aux_stream, c->ev_aux_fork, 0));
        {
            ColIndexLaunch ci;
            ci.exc = c->exc.ptr;
            ci.excv = (const uint8_t *)c->excv.ptr;
            ci.exc_n = (const uint32_t *)c->exc_n.ptr;
            ci.keys = (const uint32_t *)c->keys.ptr;
            ci.card = (const double *)c->card.ptr;
            ci.tailhist = (const uint8_t *)c->tailhist.ptr;
            ci.perm = want_sorted ? (const uint32_t *)c->perm.ptr : nullptr;
            ci.ncols = L.ncols;
            ci.p = c->p;
            ci.nblocks = NT;
            ci.nbuckets = c->nbuckets;
            ci.ent_stride = c->ent_stride;
            ci.E = c->rl_stride;
            ci.rec = (uint32_t *)c->cidx_rec.ptr;
            ci.ent = (uint32_t *)c->cidx_ent.ptr;
            ci.nS = (uint32_t *)c->colS_n.ptr;
            ci.keyS = (uint32_t *)c->colS_key.ptr;
            ci.cardS = (double *)c->colS_card.ptr;
            ci.thS = (uint8_t *)c->colS_th.ptr;
            ci.rl = (uint32_t *)c->colS_rl.ptr;
            HIPCHK(c, launch_build_colindex(c->aux_stream, ci));
        }
        HIPCHK(c, hipEventRecord(c->ev_aux_join, c->aux_stream));
        const uint64_t K = (uint64_t)L.P * c->W;
        c->Kpad = (uint32_t)((K + c->kc - 1) / c->kc * c->kc);
        if (c->Kpad) {
            const size_t bytes = (size_t)c->Kpad * L.Npad * sizeof(uint32_t);
            HIPCHK(c, c->planes.ensure(bytes));
            if (c->Kpad > K)
                HIPCHK(c, hipMemsetAsync((uint32_t *)c->planes.ptr + K * L.Npad, 0,
                                         (size_t)(c->Kpad - K) * L.Npad * sizeof(uint32_t),
                                         c->stream));
            HIPCHK(c, launch_transform(c->stream, c->regs, L.ncols, c->p, L.pbase, L.P, c->W, L.Npad,
                                       (uint32_t *)c->planes.ptr,
                                       want_sorted ? (const uint32_t *)c->perm.ptr : nullptr));
        }
        c->aux_join_pending = true;  // only k_finalize reads the index: the tile kernel starts without waiting for it
        c->lay_gen = c->pass_gen;
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

int run_pairs(dsh_ctx *c, const PairJob &job)
{
    // Triangle rows [rb,re): the plane matrix is laid out for exactly that range (wanted rows first, both
    // parts key-ordered), so every tile is homogeneous and the result lands at its final packed position --
    // whatever the range (a full triangle, one rank's rows, a row block of the CLI).  Tiny ranges and
    // rectangles keep the identity layout, which stays cached across calls; a call with parts (an event per part:
    // the pipelined exchange) always gets the key-ordered layout of its range, however short, so that its parts
    // are the ones dsh_range_parts reports to the other ranks.
    const uint64_t jre = std::min<uint64_t>(job.row_end, c->n);
    const bool full_tri = !job.rect && job.row_begin == 0 && jre >= c->n;
    const bool with_parts = job.nparts > 0 && !job.rect && !job.sorted_rows;
    int want_sorted = 0;
    uint64_t lrb = 0, lre = c->n;
    if (job.sorted_rows) want_sorted = 1;
    else if (!job.rect && jre > job.row_begin &&
             (with_parts || (c->sort_mode != 0 && (full_tri || jre - job.row_begin >= (uint64_t)c->range_sort_min_rows)))) {
        want_sorted = 1;
        lrb = job.row_begin;
        lre = jre;
    }
    c->parts_done = 0;
    c->part_ready_ms.clear();
    c->part_floats.clear();
    hipEvent_t e_call0 = nullptr;
    std::vector<std::pair<size_t, hipEvent_t>> ev_part_t;  // (profiling) part -> a timing event behind its last segment
    if (with_parts) {  // the signal block of the parts (kernels.h kSig*; used when the call turns out to signal, below)
        HIPCHK(c, c->sig.ensure((size_t)kSigWords * sizeof(uint32_t)));
        if (c->sig_gen == 0) {  // (first use: the flags must not hold garbage that passes for a generation)
            HIPCHK(c, hipMemsetAsync(c->sig.ptr, 0, (size_t)kSigWords * sizeof(uint32_t), c->stream));  // (and the tile counters, which clear themselves from then on)
            c->sig_gen = 1;
        }
    }
    if (c->profiling && with_parts) {
        e_call0 = next_event(c);
        if (e_call0) (void)hipEventRecord(e_call0, c->stream);
        // the call's start on the device's wall clock: signalled parts stamp their completion on the same
        HIPCHK(c, launch_wall_stamp(c->stream, reinterpret_cast<unsigned long long *>((uint32_t *)c->sig.ptr + kSigT0)));
    }
    // extra segments (row sets, plan.h): only with a key-ordered layout of the range and parts (the exchange)
    const bool with_extra = want_sorted && with_parts && !job.extra.empty() && lre > lrb;
    if (!job.extra.empty() && !with_extra) return fail(c, DSH_EINVAL, "internal: extra row segments need a key-ordered range with parts");
    int rc = prepare(c, job.estim, want_sorted, false, lrb, lre, with_parts ? job.nparts : 1, with_parts && job.rowsorted,
                     with_extra ? &job.extra : nullptr);
    if (rc) return rc;
    if (job.result_type < 0 || job.result_type > 8)
        return fail(c, DSH_EINVAL, "unsupported result_type %d", job.result_type);
    if (job.k < 1) return fail(c, DSH_EINVAL, "bad k %d", job.k);
    const auto t_l0 = std::chrono::steady_clock::now();
    const plan::Layout &L = c->lay;
    plan::PairPlan &pp = c->pp;
    plan::PairQuery q;
    q.rect = job.rect;
    q.sorted_rows = job.sorted_rows;
    q.want_parts = with_parts;
    q.row_begin = job.row_begin;
    q.row_end = job.row_end;
    q.col_begin = job.col_begin;
    q.col_end = job.col_end;
    plan::Tuning tu;
    tu.W = c->W;
    tu.kc = c->kc;
    tu.cum_bytes = c->cum_bytes;
    tu.cum_budget = c->cum_budget;
    tu.nsplit = c->nsplit;
    tu.lockstep = use_lockstep(c);
    tu.part_band_tiles = (uint32_t)c->part_band_tiles;
    tu.overflow_frag_max_permille = (uint32_t)c->overflow_frag_permille;
    tu.tail_bands = (uint32_t)c->tail_bands;
    // Tiles, bands and segments of the whole job first; the work items and the two device lists are made, uploaded and
    // launched BAND BY BAND: the host plans band b + 1 while the GPU runs band b (at 100 000 x p=10 the plan of 306 000
    // tiles took the host 8 ms that nothing hid, round 4 / profiles/r4y).
    if (!plan::build_tiles(L, q, tu, pp)) {
        pp.T.clear();
        return DSH_OK;
    }
    const std::vector<plan::U4> &T = pp.T, &I = pp.items;
    c->last_bands = pp.bands.size();
    std::vector<uint64_t> item_off(pp.bands.size() + 1, 0);
    for (size_t bi = 0; bi < pp.bands.size(); ++bi) item_off[bi + 1] = item_off[bi] + plan::band_item_count(tu, pp, bi);
    const uint64_t nitems_total = item_off.back();
    pp.items.reserve(nitems_total);
    // tile and item lists travel through page-locked staging, so nothing below needs the host to wait
    if (c->lists_in_flight) {  // the previous call's upload (long done unless calls are issued back to back)
        HIPCHK(c, hipEventSynchronize(c->ev_lists));
        c->lists_in_flight = false;
    }
    static_assert(sizeof(plan::U4) == sizeof(uint4), "plan::U4 must have the layout of uint4");
    HIPCHK(c, c->pin_lists.ensure((2 * T.size() + std::max<uint64_t>(nitems_total, 1)) * sizeof(uint4)));
    plan::U4 *pinT = (plan::U4 *)c->pin_lists.ptr, *pinF = pinT + T.size(), *pinI = pinF + T.size();
    HIPCHK(c, c->tiles.ensure(2 * T.size() * sizeof(uint4)));  // [tile kernel's list | k_finalize's list]
    HIPCHK(c, c->items.ensure(std::max<uint64_t>(nitems_total, 1) * sizeof(uint4)));
    if (!c->ev_lists) HIPCHK(c, hipEventCreateWithFlags(&c->ev_lists, hipEventDisableTiming));
    HIPCHK(c, c->cum.ensure(std::max<uint64_t>(pp.per_tile_bytes * pp.max_band, 256)));
    c->host_lists_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_l0).count();  // (until the first band can be planned)

    // Part signalling (kernels.h, k_finalize_signal): with parts, ONE k_finalize launch per band; the parts' flags are
    // written from inside it and the copy stream waits for them with hipStreamWaitValue32 (exchange.hip).  Not with the
    // stamped / general instances (profiling aids, rectangles) and not where the device lacks stream wait-value.
    bool signal = false, sig_upload = false;
    if (with_parts && pp.nparts >= 1 && pp.nparts <= kSigMaxParts && !c->finalize_timing && !c->finalize_stop && c->finalize_signal != 0) {
        signal = device_can_wait_value(c);
        if (!signal && c->finalize_signal == 1) return fail(c, DSH_ENODEV, "option finalize_signal = 1, but the device does not support hipStreamWaitValue32");
    }
    c->parts_signalled = signal;
    if (signal) {
        ++c->sig_gen;
        if (c->sig_gen == 0) c->sig_gen = 1;
        // the parts' tile totals (host -> device through page-locked staging), their counters cleared
        if (c->sig_in_flight) {
            HIPCHK(c, hipEventSynchronize(c->ev_sig));
            c->sig_in_flight = false;
        }
        // the words kSigPartCnt .. kSigStamp of the block lie one behind the other: the parts' counters (zero), their totals,
        // the call's generation, the stamp switch -- uploaded with the first band's lists, in one launch (the tile counters
        // clear themselves: k_finalize_signal)
        static_assert(kSigPartTotal == kSigPartCnt + kSigMaxParts && kSigGen == kSigPartTotal + kSigMaxParts && kSigStamp == kSigGen + 1, "signal block layout");
        HIPCHK(c, c->pin_sig.ensure((2 * kSigMaxParts + 2) * sizeof(uint32_t)));
        uint32_t *blk = (uint32_t *)c->pin_sig.ptr, *tot = blk + kSigMaxParts;
        for (uint32_t qd = 0; qd < kSigMaxParts; ++qd) blk[qd] = 0u;
        for (uint32_t qd = 0; qd < kSigMaxParts; ++qd) tot[qd] = qd < pp.part_tiles.size() ? pp.part_tiles[qd] : 0u;
        tot[kSigMaxParts] = c->sig_gen;                   // kSigGen
        tot[kSigMaxParts + 1] = c->profiling ? 1u : 0u;   // kSigStamp
        sig_upload = true;
    }
    const float ksinv_f = (float)(1. / (double)job.k);
    if (c->finalize_timing) {
        HIPCHK(c, c->phase_cyc.ensure(16 * sizeof(unsigned long long)));
        HIPCHK(c, hipMemsetAsync(c->phase_cyc.ptr, 0, 16 * sizeof(unsigned long long), c->stream));
    }
    std::vector<std::pair<hipEvent_t, hipEvent_t>> evp, evf;
    for (size_t bi = 0; bi < pp.bands.size(); ++bi) {
        const auto &bd = pp.bands[bi];
        const uint32_t nt = (uint32_t)(bd.second - bd.first);
        const uint64_t nslots = (uint64_t)nt * kTile * kTile;
        // this band's items and list entries: planned, staged, uploaded (the uploads read the staging when they run: it is
        // not touched again before ev_lists of this call has passed)
        const auto t_b0 = std::chrono::steady_clock::now();
        plan::build_band_items(tu, pp, bi);
        if (pp.band_items[bi].first != item_off[bi] || pp.band_items[bi].second != item_off[bi + 1])
            return fail(c, DSH_EIO, "internal: band %zu has %zu items, %llu were planned for", bi,
                        pp.band_items[bi].second - pp.band_items[bi].first, (unsigned long long)(item_off[bi + 1] - item_off[bi]));
        plan::emit_band_lists(L, pp, bi, pinT, pinF);
        const uint32_t ni = (uint32_t)(item_off[bi + 1] - item_off[bi]);
        if (ni) std::memcpy(pinI + item_off[bi], I.data() + item_off[bi], (size_t)ni * sizeof(uint4));
        c->host_lists_us += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_b0).count();
        {
            UploadSegs up;
            uint32_t nseg = 0;
            if (sig_upload) up.add(nseg, (uint32_t *)c->sig.ptr + kSigPartCnt, c->pin_sig.ptr, (2 * kSigMaxParts + 2) * sizeof(uint32_t));
            up.add(nseg, (uint4 *)c->tiles.ptr + bd.first, pinT + bd.first, (size_t)nt * sizeof(uint4));
            up.add(nseg, (uint4 *)c->tiles.ptr + T.size() + bd.first, pinF + bd.first, (size_t)nt * sizeof(uint4));
            up.add(nseg, (uint4 *)c->items.ptr + item_off[bi], pinI + item_off[bi], (size_t)ni * sizeof(uint4));
            HIPCHK(c, launch_upload_segs(c->stream, up, nseg));
            if (sig_upload) {
                if (!c->ev_sig) HIPCHK(c, hipEventCreateWithFlags(&c->ev_sig, hipEventDisableTiming));
                HIPCHK(c, hipEventRecord(c->ev_sig, c->stream));
                c->sig_in_flight = true;
                sig_upload = false;
            }
        }
        HIPCHK(c, hipEventRecord(c->ev_lists, c->stream));
        c->lists_in_flight = true;
        const uint4 *dt = (const uint4 *)c->tiles.ptr + bd.first;
        const uint4 *di = (const uint4 *)c->items.ptr + item_off[bi];
        hipEvent_t a = nullptr, b = nullptr, d = nullptr;
        if (c->profiling) {
            a = next_event(c);
            b = next_event(c);
            d = next_event(c);
            if (a) (void)hipEventRecord(a, c->stream);
        }
        if (c->pair_mfma)
            HIPCHK(c, launch_pair_counts_mfma(c->stream, c->kc, c->cum_bytes, (const uint32_t *)c->planes.ptr,
                                              L.Npad, c->Kpad, c->W, L.P, dt, di, ni, c->cum.ptr, nslots));
        else if (use_lockstep(c))
            HIPCHK(c, launch_pair_counts_lockstep(c->stream, c->kc, c->cum_bytes, (const uint32_t *)c->planes.ptr,
                                                  L.Npad, c->Kpad, c->W, L.P, dt, di, ni, bi < pp.band_frags.size() ? pp.band_frags[bi] : 0u,
                                                  c->cum.ptr, nslots));
        else
            HIPCHK(c, launch_pair_counts(c->stream, c->kc, c->cum_bytes, (const uint32_t *)c->planes.ptr,
                                         L.Npad, c->Kpad, c->W, L.P, dt, di, ni, c->cum.ptr, nslots));
        if (b) (void)hipEventRecord(b, c->stream);
        if (bi == 0) {  // (the destination of an exchange may post its receives behind its first tile kernel, exchange.hip)
            if (!c->ev_first_tiles) HIPCHK(c, hipEventCreateWithFlags(&c->ev_first_tiles, hipEventDisableTiming));
            HIPCHK(c, hipEventRecord(c->ev_first_tiles, c->stream));
        }
        // signal mode: ONE launch over the band's tiles; the parts announce themselves (k_finalize_signal)
        std::vector<plan::Seg> one_seg;
        if (signal) {
            plan::Seg all{bd.first, bd.second, -1, 1};
            for (const plan::Seg &sg : pp.segs[bi]) all.hist_bins = std::max(all.hist_bins, sg.hist_bins);
            one_seg.push_back(all);
        }
        // (without flags -- finalize_signal = 0, or a device without stream wait-value -- one launch and one event per part)
        for (const plan::Seg &sg : signal ? one_seg : pp.segs[bi]) {
            hipStream_t fst = c->stream;
            FinalizeLaunch f;
            f.cum = c->cum.ptr;  // the band's C(v); a tile's block is named by its descriptor
            f.cum_bytes = c->cum_bytes;
            f.cum_stride = nslots;
            f.hist_bins = sg.hist_bins;
            f.nS = (const uint32_t *)c->colS_n.ptr;
            f.keyS = (const uint32_t *)c->colS_key.ptr;
            f.cardS = (const double *)c->colS_card.ptr;
            f.thS = (const uint8_t *)c->colS_th.ptr;
            f.rl = (const uint32_t *)c->colS_rl.ptr;
            f.E = c->rl_stride;
            f.nslots = (uint64_t)(sg.e - sg.b) * kTile * kTile;
            f.tiles = (const uint4 *)c->tiles.ptr + T.size() + sg.b;
            f.perm = L.sorted ? (const uint32_t *)c->perm.ptr : nullptr;
            f.rowoff = L.rowsorted ? (const uint64_t *)c->rowoff.ptr : nullptr;
            f.pbase = L.pbase;
            f.cidx_rec = (const uint32_t *)c->cidx_rec.ptr;
            f.cidx_ent = (const uint32_t *)c->cidx_ent.ptr;
            f.nbuckets = c->nbuckets;
            f.ent_stride = c->ent_stride;
            f.p = c->p;
            f.estim = job.estim;
            f.result_type = job.result_type;
            f.ksinv = job.ksinv_double ? 1. / (double)job.k : (double)ksinv_f;
            f.n = c->n;
            f.ncols = L.ncols;
            f.stop = c->finalize_stop;
            f.phase_cyc = c->finalize_timing ? (unsigned long long *)c->phase_cyc.ptr : nullptr;
            f.rect = job.rect;
            f.sorted_out = job.sorted_rows;
            f.square = job.square;
            f.knn = job.knn;
            f.out2 = job.d_out2;
            f.knn_ld = job.knn_ld;
            f.knn_rows = job.knn_rows;
            f.row_begin = job.row_begin;
            // (with extra segments every pair of a launched tile is this rank's: the runs of the layout lie in row order and
            // whole 128-column blocks are wanted or not, plan.h)
            f.row_end = with_extra ? c->n : job.row_end;
            f.col_begin = job.col_begin;
            f.col_end = job.col_end;
            f.base_index = job.base_index;
            f.out = job.d_out;
            if (signal) f.sig = (uint32_t *)c->sig.ptr;
            if (c->aux_join_pending) {  // (the position index is built on the second stream)
                HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_aux_join, 0));
                c->aux_join_pending = false;
            }
            HIPCHK(c, launch_finalize(fst, f));
            if (sg.part >= 0) {  // this segment completes a part: its span of the matrix is final
                const size_t qp = (size_t)sg.part;
                while (c->ev_part.size() <= qp) {
                    hipEvent_t e = nullptr;
                    HIPCHK(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
                    c->ev_part.push_back(e);
                }
                HIPCHK(c, hipEventRecord(c->ev_part[qp], fst));
                c->parts_done = (uint32_t)qp + 1;
                if (e_call0) {
                    hipEvent_t te = next_event(c);
                    if (te) {
                        (void)hipEventRecord(te, fst);
                        ev_part_t.emplace_back(qp, te);
                    }
                }
            }
        }
        if (d) (void)hipEventRecord(d, c->stream);
        if (a && b && d) {
            evp.emplace_back(a, b);
            evf.emplace_back(b, d);
        }
    }
    if (signal) c->parts_done = pp.nparts;  // (every part's flag will be written by the launches above)
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
        // when every part of the call was final (from the start of the call, prepare included) and how many floats of the
        // rank's buffer it holds: what a model of the pipelined exchange needs (dsh_last_part_info)
        auto fill_part_floats = [&] {  // how many floats of the rank's buffer every part holds
            for (size_t i = 0; i < c->part_floats.size(); ++i) {
                if (L.rowsorted && i + 1 < L.part_w.size()) c->part_floats[i] = L.rowoff_w[L.part_w[i + 1]] - L.rowoff_w[L.part_w[i]];
                else if (!L.rowsorted && L.extra.empty() && i + 1 < L.parts.size()) c->part_floats[i] = plan::tri_span(c->n, L.parts[i], L.parts[i + 1]);
                else if (!L.rowsorted) c->part_floats[i] = plan::rowset_span(c->n, L.rb, L.re, L.extra);
            }
        };
        if (signal && e_call0 && c->wall_clock_khz > 0) {
            std::vector<unsigned long long> st(kSigMaxParts + 1);
            HIPCHK(c, hipMemcpy(st.data(), (uint32_t *)c->sig.ptr + kSigPartTime, kSigMaxParts * sizeof(unsigned long long), hipMemcpyDeviceToHost));
            HIPCHK(c, hipMemcpy(&st[kSigMaxParts], (uint32_t *)c->sig.ptr + kSigT0, sizeof(unsigned long long), hipMemcpyDeviceToHost));
            c->part_ready_ms.assign(c->parts_done, 0.0);
            c->part_floats.assign(c->parts_done, 0);
            for (size_t i = 0; i < c->part_ready_ms.size(); ++i)
                c->part_ready_ms[i] = (double)(long long)(st[i] - st[kSigMaxParts]) / (double)c->wall_clock_khz;
            fill_part_floats();
        } else if (e_call0 && !ev_part_t.empty()) {
            c->part_ready_ms.assign(c->parts_done, 0.0);
            c->part_floats.assign(c->parts_done, 0);
            for (auto &pe : ev_part_t) {
                float ms = 0;
                (void)hipEventElapsedTime(&ms, e_call0, pe.second);
                if (pe.first < c->part_ready_ms.size()) c->part_ready_ms[pe.first] = ms;
            }
            fill_part_floats();
        }
    }
    return DSH_OK;
}

}  // namespace dsh

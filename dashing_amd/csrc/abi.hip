// abi.hip -- the C-ABI of libdashing_hip.so (include/dashing_hip.h) over the gfx950 kernels: context, resident sketch
// matrix, sketch waist, cardinalities, the compare entry points, per-call completion tickets, shards of the sorted
// triangle, options.  Argument checks and call sequencing live here; the work is in engine.hip (prepare / run_pairs),
// knn.hip and exchange.hip; the pure entry points (dsh_tri_*, dsh_partition_rows, dsh_balance_rows, dsh_range_parts)
// are in plan.cpp.  No CPU fallback lives here: without a HIP device dsh_create fails with DSH_ENODEV.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <new>

#include "ctx.h"

using namespace dsh;

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

int dsh_abi_version(void) { return DSH_ABI_VERSION; }

int dsh_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// One teardown for dsh_destroy and the failure path of dsh_create: wait for all three streams first (nothing of ours is
// then in flight), then the communicator, the events and buffers, and the streams last.
static void release_ctx(dsh_ctx *c)
{
    (void)hipSetDevice(c->device);
    for (hipStream_t s : {c->stream, c->copy_stream, c->aux_stream, c->place_stream})
        if (s) (void)hipStreamSynchronize(s);
    if (c->spin_running && c->spin_flag) *(volatile uint32_t *)c->spin_flag = 1;
    if (c->spin_stream) {
        (void)hipStreamSynchronize(c->spin_stream);
        (void)hipStreamDestroy(c->spin_stream);
        c->spin_stream = nullptr;
    }
    c->spin_running = false;
    if (c->spin_flag) (void)hipHostFree(c->spin_flag);
    c->spin_flag = nullptr;
    (void)comm_release(c);
    for (DevBuf *b : {&c->gather_full, &c->gather_local, &c->regs_own, &c->card, &c->planes, &c->exc, &c->exc_n, &c->excv,
                      &c->keys, &c->tailhist, &c->hist, &c->cidx_rec, &c->cidx_ent, &c->colS_n, &c->colS_key, &c->colS_card, &c->colS_th, &c->colS_rl, &c->rowoff, &c->xch_stage, &c->xch_tab, &c->place_tab, &c->sig, &c->perm, &c->items, &c->cum, &c->tiles,
                      &c->outbuf, &c->outbuf2[0], &c->outbuf2[1], &c->seqbuf, &c->workbuf, &c->phase_cyc, &c->rawbuf, &c->fx_tab,
                      &c->fx_summ, &c->fx_state, &c->fx_declen, &c->fx_status})
        b->release();
    if (c->pin_perm) (void)hipHostFree(c->pin_perm);
    c->pin_perm = nullptr;
    c->pin_lists.release();
    c->pin_work.release();
    c->pin_keys.release();
    c->pin_rowoff.release();
    c->pin_xch.release();
    c->pin_sig.release();
    c->pin_fx.release();
    for (hipEvent_t *e : {&c->ev_fx, &c->ev_work, &c->ev_lists, &c->ev_perm, &c->ev_keys, &c->ev_filled[0], &c->ev_filled[1],
                          &c->ev_drained[0], &c->ev_drained[1], &c->ev_aux_fork, &c->ev_aux_join, &c->ev_xch_tab, &c->ev_place_done, &c->ev_first_tiles, &c->ev_sig}) {
        if (*e) (void)hipEventDestroy(*e);
        *e = nullptr;
    }
    for (auto e : c->ev_pool) (void)hipEventDestroy(e);
    for (auto e : c->tickets) (void)hipEventDestroy(e);
    for (auto e : c->ev_part) (void)hipEventDestroy(e);
    for (auto e : c->ev_round) (void)hipEventDestroy(e);
    c->ev_round.clear();
    c->ev_pool.clear();
    c->tickets.clear();
    c->ev_part.clear();
    for (hipStream_t *s : {&c->stream, &c->copy_stream, &c->aux_stream, &c->place_stream}) {
        if (*s) (void)hipStreamDestroy(*s);
        *s = nullptr;
    }
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
        create_copy_stream(&c->place_stream) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_aux_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_aux_join, hipEventDisableTiming) != hipSuccess) {
        release_ctx(c);
        delete c;
        return DSH_EIO;
    }
    *out = c;
    return DSH_OK;
}

void dsh_destroy(dsh_ctx *c)
{
    if (!c) return;
    release_ctx(c);
    delete c;
}

int dsh_preload(int device, unsigned what)
{
    if (hipSetDevice(device) != hipSuccess) {
        (void)hipGetLastError();  // (the runtime keeps the last error per thread: a later launch's check must not find this one)
        return DSH_ENODEV;
    }
    hipError_t e = hipSuccess;
    if ((what & DSH_PRELOAD_SKETCH) && e == hipSuccess) e = preload_sketch_kernels();
    if ((what & DSH_PRELOAD_SKETCH) && e == hipSuccess) e = preload_fastx_kernels();
    if ((what & DSH_PRELOAD_COMPARE) && e == hipSuccess) e = preload_compare_kernels();
    if (e != hipSuccess) (void)hipGetLastError();
    return e == hipSuccess ? DSH_OK : DSH_EIO;
}

const char *dsh_last_error(const dsh_ctx *c) { return c ? c->err.c_str() : "null ctx"; }

int dsh_synchronize(dsh_ctx *c)
{
    if (!c) return DSH_EINVAL;
    int rc = bind(c);
    if (rc) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipStreamSynchronize(c->copy_stream));
    HIPCHK(c, hipStreamSynchronize(c->place_stream));
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
    // work list: each workgroup walks up to kSubsPerWG sub-chunks (8 192 bases each) of one genome.  (Fewer sub-chunks per
    // workgroup for SMALL calls -- a 46 MB batch of the streaming loader is 350 workgroups of 16 -- were tried in round 6 and
    // lost: 2 sub-chunks = 2 785 workgroups took 147 us instead of 102, every workgroup ends by max-merging its 2^p
    // registers into the same few rows of the matrix; profiles/rd6n, rd6o cli_kernel_stats.csv.)
    // At BASELINE configs[1] size 16 is the optimum: 8 -> 8.43e11, 16 -> 8.59e11, 32 -> 8.31e11, 64 -> 8.2e11, 128 -> 8.3e11 bases/s
    // (profiles/rd6p/sketch_subs_ab.jsonl).
    // Large registers make the merge dear (a workgroup ends by max-merging its 2^p registers into the matrix) and leave few
    // workgroups per CU: from p = 14 a workgroup walks more sub-chunks when the call is big enough to still give every CU
    // several workgroups -- 300 x 5 Mbp, bases/s: p = 14 16 -> 64 sub-chunks 7.97e11 -> 8.17e11; p = 15 -> 128 6.38e11 ->
    // 7.62e11 (256: 6.96e11); p = 16 4.63e11 -> 5.18e11; p = 17 3.35e11 -> 4.66e11; p = 13 stays at 16 (32: -3 %)
    // (profiles/rd6af/sketch_subs_large_p.txt).
    uint64_t total_subs = 0;
    for (uint32_t g = 0; g < n_genomes; ++g)
        if (genome_off[g + 1] >= genome_off[g]) total_subs += (genome_off[g + 1] - (genome_off[g] & ~31ull) + kSketchSub - 1) / kSketchSub;
    const uint32_t subs_cap = c->p <= 13 ? 16u : (c->p == 14 ? 64u : 128u);
    const uint32_t kSubsPerWG = (uint32_t)std::min<uint64_t>(subs_cap, std::max<uint64_t>(16, total_subs / 1024));
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
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (c->profiling) {  // k_sketch alone, on the stream it runs on
        c->ev_used = 0;
        e0 = next_event(c);
        e1 = next_event(c);
        if (e0) (void)hipEventRecord(e0, c->stream);
    }
    HIPCHK(c, launch_sketch(c->stream, d_seq, (const SketchWork *)c->workbuf.ptr,
                            (uint32_t)work.size(), k, c->p, canon, (uint8_t *)c->regs_own.ptr));
    if (e0 && e1) {
        (void)hipEventRecord(e1, c->stream);
        (void)hipEventSynchronize(e1);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        c->sketch_ms = ms;
    }
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

int dsh_sketch_fastx_batch_async(dsh_ctx *c, const uint8_t *raw, const uint64_t *genome_off, const uint64_t *raw_len,
                                 uint32_t n_genomes, uint64_t first_slot, int k, int canon, uint32_t *status_out)
{
    int rc = sketch_check(c, genome_off, n_genomes, first_slot, k);
    if (rc) return rc;
    if ((rc = bind(c))) return rc;
    if (n_genomes == 0) return DSH_OK;
    if (!raw_len) return DSH_EINVAL;
    const uint64_t lo = genome_off[0], hi = genome_off[n_genomes];
    if (hi < lo) return fail(c, DSH_EINVAL, "genome_off not monotone");
    if (lo & 31) return fail(c, DSH_EINVAL, "genome_off[0] must be a multiple of 32");
    // the decoder's tables: genomes, then the 16 KB chunks of their raw bytes (one workgroup each)
    std::vector<FastxGenome> gen(n_genomes);
    std::vector<FastxChunk> chunks;
    for (uint32_t g = 0; g < n_genomes; ++g) {
        const uint64_t b = genome_off[g], e = genome_off[g + 1];
        if (e < b || (b & 31) || raw_len[g] > e - b)
            return fail(c, DSH_EINVAL, "genome %u: its region must start on a multiple of 32 and hold its %llu raw bytes", g, (unsigned long long)raw_len[g]);
        gen[g].off = b - lo;
        gen[g].rawlen = raw_len[g];
        gen[g].fmt = (raw_len[g] && raw && raw[b] == '@') ? 1u : 0u;  // (the device checks that the rest keeps the promise)
        gen[g].pad_ = 0;
        gen[g].region_end = e - lo;
        gen[g].chunk0 = (uint32_t)chunks.size();
        for (uint64_t x = 0; x < raw_len[g]; x += kFastxChunk)
            chunks.push_back(FastxChunk{b - lo + x, (uint32_t)std::min<uint64_t>(kFastxChunk, raw_len[g] - x), g});
        gen[g].nchunks = (uint32_t)chunks.size() - gen[g].chunk0;
        if (chunks.size() > 0x7FFFFFFFull) return fail(c, DSH_EINVAL, "batch too large");
    }
    const size_t bytes = (size_t)(hi - lo);
    HIPCHK(c, c->rawbuf.ensure(bytes + 256));
    HIPCHK(c, c->seqbuf.ensure(bytes + 256));
    if (bytes) {
        if (!raw) return DSH_EINVAL;
        HIPCHK(c, hipMemcpyAsync(c->rawbuf.ptr, raw + lo, bytes, hipMemcpyHostToDevice, c->stream));
    }
    const size_t gbytes = gen.size() * sizeof(FastxGenome), cbytes = chunks.size() * sizeof(FastxChunk);
    if (c->fx_in_flight) {  // (the previous batch's tables have long been uploaded unless calls come back to back)
        HIPCHK(c, hipEventSynchronize(c->ev_fx));
        c->fx_in_flight = false;
    }
    HIPCHK(c, c->pin_fx.ensure(gbytes + cbytes));
    std::memcpy(c->pin_fx.ptr, gen.data(), gbytes);
    if (cbytes) std::memcpy((uint8_t *)c->pin_fx.ptr + gbytes, chunks.data(), cbytes);
    HIPCHK(c, c->fx_tab.ensure(gbytes + cbytes));
    HIPCHK(c, launch_upload(c->stream, c->fx_tab.ptr, c->pin_fx.ptr, gbytes + cbytes));
    if (!c->ev_fx) HIPCHK(c, hipEventCreateWithFlags(&c->ev_fx, hipEventDisableTiming));
    HIPCHK(c, hipEventRecord(c->ev_fx, c->stream));
    c->fx_in_flight = true;
    HIPCHK(c, c->fx_summ.ensure(std::max<size_t>(chunks.size(), 1) * sizeof(FastxSumm)));
    HIPCHK(c, c->fx_state.ensure(std::max<size_t>(chunks.size(), 1) * sizeof(uint4)));
    HIPCHK(c, c->fx_declen.ensure(gen.size() * sizeof(uint64_t)));
    // [status words | length fingerprints]: one buffer, cleared by one memset
    const size_t st_bytes = (gen.size() * sizeof(uint32_t) + 7) & ~(size_t)7;
    HIPCHK(c, c->fx_status.ensure(st_bytes + gen.size() * sizeof(unsigned long long)));
    HIPCHK(c, hipMemsetAsync(c->fx_status.ptr, 0, st_bytes + gen.size() * sizeof(unsigned long long), c->stream));
    hipEvent_t fe0 = nullptr, fe1 = nullptr;  // (profiling: the decode kernels alone, dsh_get_info "fastx_decode_us")
    if (c->profiling) {
        c->ev_used = 0;
        fe0 = next_event(c);
        fe1 = next_event(c);
        if (fe0) (void)hipEventRecord(fe0, c->stream);
    }
    HIPCHK(c, launch_fastx_decode(c->stream, (const uint8_t *)c->rawbuf.ptr, (const FastxChunk *)((const uint8_t *)c->fx_tab.ptr + gbytes),
                                  (uint32_t)chunks.size(), (const FastxGenome *)c->fx_tab.ptr, n_genomes, (FastxSumm *)c->fx_summ.ptr,
                                  (uint4 *)c->fx_state.ptr, (uint64_t *)c->fx_declen.ptr, (uint32_t *)c->fx_status.ptr,
                                  (unsigned long long *)((uint8_t *)c->fx_status.ptr + st_bytes), (uint8_t *)c->seqbuf.ptr));
    if (fe0 && fe1) {
        (void)hipEventRecord(fe1, c->stream);
        (void)hipEventSynchronize(fe1);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, fe0, fe1);
        c->fastx_ms = ms;
    }
    if (status_out)
        HIPCHK(c, hipMemcpyAsync(status_out, c->fx_status.ptr, gen.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    std::vector<uint64_t> off(n_genomes + 1);
    for (uint32_t g = 0; g <= n_genomes; ++g) off[g] = genome_off[g] - lo;
    rc = sketch_common(c, (const uint8_t *)c->seqbuf.ptr, off.data(), n_genomes, first_slot, k, canon);
    if (rc) return rc;
    invalidate(c);
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
    if (rb >= re || c->n < 2) return DSH_OK;  // (no rows: no parts, no events -- dsh_collect_parts_async knows)
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
// enqueued afterwards.  It is TWO independent events, one per stream, and orders nothing between the streams: a ticket
// taken between the compute call and dsh_collect_parts_async does not hold the per-part transfers behind the kernels.
static constexpr size_t kTicketRing = 64;

int dsh_event_record(dsh_ctx *c, uint64_t *ticket)
{
    if (!c || !ticket) return DSH_EINVAL;
    int rc = bind(c);
    if (rc) return rc;
    if (c->tickets.size() < 2 * kTicketRing) {
        hipEvent_t e = nullptr, j = nullptr;
        HIPCHK(c, hipEventCreateWithFlags(&j, hipEventDisableTiming));
        c->tickets.push_back(j);
        HIPCHK(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        c->tickets.push_back(e);
    }
    const uint64_t t = c->ticket_next;
    hipEvent_t es = c->tickets[2 * (t % kTicketRing)], ec = c->tickets[2 * (t % kTicketRing) + 1];
    if (t >= kTicketRing) {  // the ticket that used this slot 64 records ago
        HIPCHK(c, hipEventSynchronize(es));
        HIPCHK(c, hipEventSynchronize(ec));
    }
    HIPCHK(c, hipEventRecord(es, c->stream));
    HIPCHK(c, hipEventRecord(ec, c->copy_stream));
    c->ticket_next = t + 1;
    *ticket = t;
    return DSH_OK;
}

static int ticket_events(dsh_ctx *c, uint64_t ticket, hipEvent_t e[2])
{
    if (ticket >= c->ticket_next) return fail(c, DSH_EINVAL, "ticket %llu was never recorded", (unsigned long long)ticket);
    e[0] = e[1] = nullptr;
    if (c->ticket_next - ticket <= kTicketRing) {  // (older than the ring: waited for when its slot was reused)
        e[0] = c->tickets[2 * (ticket % kTicketRing)];
        e[1] = c->tickets[2 * (ticket % kTicketRing) + 1];
    }
    return DSH_OK;
}

int dsh_event_wait(dsh_ctx *c, uint64_t ticket)
{
    if (!c) return DSH_EINVAL;
    int rc = bind(c);
    if (rc) return rc;
    hipEvent_t e[2];
    if ((rc = ticket_events(c, ticket, e))) return rc;
    for (hipEvent_t x : e)
        if (x) HIPCHK(c, hipEventSynchronize(x));
    return DSH_OK;
}

int dsh_event_query(dsh_ctx *c, uint64_t ticket, int *done)
{
    if (!c || !done) return DSH_EINVAL;
    int rc = bind(c);
    if (rc) return rc;
    hipEvent_t e[2];
    if ((rc = ticket_events(c, ticket, e))) return rc;
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

// cost model for balancing shards: a tile costs its dense planes plus ~5 plane-equivalents of
// finalize work (6.6 ms finalize vs 1.4 ms per plane on the C3 workload, profiles/r1f)
constexpr double kShardC0 = 5.0;           // finalize work of a tile in plane-equivalents
constexpr double kAssemblerPermille = 21;  // the un-permute (0.42 ms) on rank 0 of a 19.9 ms pass (profiles/r1k)
static void shard_bounds(dsh_ctx *c, uint32_t nshards, std::vector<uint32_t> &tb)
{
    const uint32_t NT = c->lay.Npad / kTile;
    std::vector<double> rowcost(NT, 0.);
    double total = 0;
    for (uint32_t ti = 0; ti < NT; ++ti) {
        for (uint32_t tj = ti; tj < NT; ++tj) {
            int pb, pe;
            plan::tile_planes(c->lay, ti, tj, pb, pe);
            rowcost[ti] += (pe - pb) + kShardC0;
        }
        total += rowcost[ti];
    }
    // Contiguous tile-row ranges that minimise the largest shard (linear partition by bisection on the
    // limit + greedy fill).  Shard 0 belongs to the rank that also assembles the result (the un-permute,
    // about assembler_permille/1000 of a single-GPU pass): it carries that as extra cost.
    const double extra0 = nshards > 1 ? total * kAssemblerPermille / 1000.0 : 0.0;
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
    HIPCHK(c, launch_unpermute(c->stream, (const float *)d_sorted_tri, (const uint32_t *)c->perm.ptr + c->n, c->n, (float *)d_out_tri));
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
    const uint32_t NT = c->lay.Npad / kTile;
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
    if (!c->profiling) c->finalize_stop = c->finalize_timing = 0;  // the stop points and stamps exist for profiling runs only
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

int dsh_last_part_info(dsh_ctx *c, double *ready_ms, uint64_t *floats, uint32_t cap, uint32_t *nparts_out)
{
    if (!c || !nparts_out) return DSH_EINVAL;
    const uint32_t k = (uint32_t)c->part_ready_ms.size();
    *nparts_out = k;
    if (k > cap && (ready_ms || floats)) return DSH_EINVAL;
    for (uint32_t q = 0; q < k; ++q) {
        if (ready_ms) ready_ms[q] = c->part_ready_ms[q];
        if (floats) floats[q] = c->part_floats[q];
    }
    return DSH_OK;
}

int dsh_finalize_phase_cycles(dsh_ctx *c, uint64_t *out16)
{
    if (!c || !out16) return DSH_EINVAL;
    int rc = bind(c);
    if (rc) return rc;
    if (!c->phase_cyc.ptr) return fail(c, DSH_ESTATE, "no call with the option finalize_timing yet");
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(out16, c->phase_cyc.ptr, 16 * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return DSH_OK;
}

int dsh_get_info(dsh_ctx *c, const char *name, int64_t *out)
{
    if (!c || !name || !out) return DSH_EINVAL;
    if (!std::strcmp(name, "planes")) *out = c->lay.P;
    else if (!std::strcmp(name, "vlo")) *out = c->lay.vlo;
    else if (!std::strcmp(name, "vhi")) *out = c->lay.vhi;
    else if (!std::strcmp(name, "threshold")) *out = c->lay.pbase + (int64_t)c->lay.P;
    else if (!std::strcmp(name, "pbase")) *out = c->lay.pbase;
    else if (!std::strcmp(name, "host_layout_us")) *out = (int64_t)c->host_layout_us;
    else if (!std::strcmp(name, "host_lists_us")) *out = (int64_t)c->host_lists_us;
    else if (!std::strcmp(name, "host_keys_wait_us")) *out = (int64_t)c->host_keys_wait_us;
    else if (!std::strcmp(name, "emax")) *out = c->emax;
    else if (!std::strcmp(name, "elow")) *out = c->elow;
    else if (!std::strcmp(name, "kc")) *out = c->kc;
    else if (!std::strcmp(name, "tile")) *out = kTile;
    else if (!std::strcmp(name, "parts_done")) *out = c->parts_done;
    else if (!std::strcmp(name, "parts_signalled")) *out = c->parts_signalled ? 1 : 0;
    else if (!std::strcmp(name, "sketch_kernel_us")) *out = (int64_t)(c->sketch_ms * 1000.0);
    else if (!std::strcmp(name, "fastx_decode_us")) *out = (int64_t)(c->fastx_ms * 1000.0);
    else if (!std::strcmp(name, "xch_recv_gated")) *out = c->xch_recv_gated ? 1 : 0;
    else if (!std::strcmp(name, "place_kernel_us")) *out = (int64_t)(c->place_ms * 1000.0);
    else if (!std::strcmp(name, "whatif_mfma")) {
#ifdef DSH_WHATIF_MFMA
        *out = 1;
#else
        *out = 0;
#endif
    }
    else if (!std::strcmp(name, "npad")) *out = c->lay.Npad;
    else if (!std::strcmp(name, "kpad")) *out = c->Kpad;
    else if (!std::strcmp(name, "cum_bytes")) *out = c->cum_bytes;
    else if (!std::strcmp(name, "sorted")) *out = c->lay.sorted;
    else if (!std::strcmp(name, "ncols")) *out = (int64_t)c->lay.ncols;
    else if (!std::strcmp(name, "lockstep")) *out = c->planes_valid && use_lockstep(c) ? 1 : 0;
    else if (!std::strcmp(name, "tiles")) *out = (int64_t)c->pp.T.size();
    else if (!std::strcmp(name, "bands")) *out = (int64_t)c->last_bands;
    else if (!std::strcmp(name, "items")) *out = (int64_t)c->pp.items.size();  // work items of the tile kernel (rounds of 512)
    else if (!std::strcmp(name, "frag_items")) {  // ... of which overflow fragments (plan.h)
        uint64_t f = 0;
        for (uint32_t x : c->pp.band_frags) f += x;
        *out = (int64_t)f;
    }
    else if (!std::strcmp(name, "words_per_plane")) *out = c->W;
    else if (!std::strcmp(name, "avg_tile_planes_x100")) {
        uint64_t tot = 0;
        for (const auto &t : c->pp.T) tot += t.w - t.z;
        *out = c->pp.T.empty() ? 0 : (int64_t)(tot * 100 / c->pp.T.size());
    }
    else return fail(c, DSH_EINVAL, "unknown info %s", name);
    return DSH_OK;
}

int dsh_set_option(dsh_ctx *c, const char *name, int64_t v)
{
    if (!c || !name) return DSH_EINVAL;
    if (!std::strcmp(name, "kc")) {
        if (v != 0 && v != 16 && v != 32) return fail(c, DSH_EINVAL, "kc must be 0 (auto), 16 or 32");
        c->kc_opt = (int)v;
        c->planes_valid = false;  // Kpad depends on kc
        return DSH_OK;
    }
    if (!std::strcmp(name, "cum_budget_bytes")) {
        if (v < (1 << 20)) return fail(c, DSH_EINVAL, "cum_budget_bytes too small");
        c->cum_budget = (uint64_t)v;
        return DSH_OK;
    }
    if (!std::strcmp(name, "knn_square_budget_bytes")) {
        if (v < 0) return fail(c, DSH_EINVAL, "knn_square_budget_bytes must be >= 0");
        c->knn_square_budget = (uint64_t)v;
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
#ifdef DSH_WHATIF_MFMA
    if (!std::strcmp(name, "pair_mfma")) {  // what-if builds only (make WHATIF=1): the north star keeps the matrix cores off this path
        c->pair_mfma = v != 0;
        return DSH_OK;
    }
#endif
    if (!std::strcmp(name, "finalize_stop")) {  // profiling only: results are meaningless while it is set
        if (v < 0 || v > 4) return fail(c, DSH_EINVAL, "finalize_stop must be in [0,4]");
        if (v && !c->profiling) return fail(c, DSH_ESTATE, "finalize_stop needs dsh_set_profiling(ctx, 1)");
        c->finalize_stop = (int)v;
        return DSH_OK;
    }
    if (!std::strcmp(name, "finalize_timing")) {  // profiling only: same results, the stamped instance of k_finalize
        if (v && !c->profiling) return fail(c, DSH_ESTATE, "finalize_timing needs dsh_set_profiling(ctx, 1)");
        c->finalize_timing = v != 0;
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
    if (!std::strcmp(name, "part_band_tiles")) {
        if (v < 1 || v > (1 << 30)) return fail(c, DSH_EINVAL, "part_band_tiles out of range");
        c->part_band_tiles = (int)v;
        return DSH_OK;
    }
    if (!std::strcmp(name, "xch_tail_bands")) {
        if (v < 0 || v > 8) return fail(c, DSH_EINVAL, "xch_tail_bands out of range");
        c->tail_bands = (int)v;
        return DSH_OK;
    }
    if (!std::strcmp(name, "overflow_frag_permille")) {
        if (v < 0 || v > 1000) return fail(c, DSH_EINVAL, "overflow_frag_permille out of range");
        c->overflow_frag_permille = (int)v;
        return DSH_OK;
    }
    if (!std::strcmp(name, "xch_recv_gate")) {
        if (v < -1 || v > 1) return fail(c, DSH_EINVAL, "xch_recv_gate is -1 (auto), 0 or 1");
        c->xch_recv_gate = (int)v;
        return DSH_OK;
    }
    if (!std::strcmp(name, "finalize_signal")) {
        if (v < -1 || v > 1) return fail(c, DSH_EINVAL, "finalize_signal is -1 (auto), 0 or 1");
        c->finalize_signal = (int)v;
        return DSH_OK;
    }
    return fail(c, DSH_EINVAL, "unknown option %s", name);
}

}  // extern "C"

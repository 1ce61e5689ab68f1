// exchange.hip -- the multi-GPU exchange behind the C-ABI: an RCCL communicator per context (librccl is dlopen'ed on
// first use), point-to-point collection of the ranks' spans of dashing's packed triangle on one destination rank
// (one message per peer or, pipelined, per part), the all-gather of register arrays after sharded sketching.
#include <dlfcn.h>

#include <chrono>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <thread>
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <new>

#include "ctx.h"

using namespace dsh;

namespace {

// RCCL, loaded on first use (std::call_once: the CLI's one-thread-per-device path may get here from several threads).
// The library must be the one built against the HIP runtime this file links to (the streams handed to it are ours):
// librccl.so.1 through this library's RUNPATH (/opt/rocm/lib), or DSH_RCCL_LIB.
struct Rccl {
    void *h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
    ncclResult_t (*GetVersion)(int *) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::string err, path;
};

Rccl *rccl()
{
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char *names[3] = {std::getenv("DSH_RCCL_LIB"), "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
        for (const char *nm : names) {
            if (!nm || !*nm) continue;
            r.h = dlopen(nm, RTLD_NOW | RTLD_LOCAL);
            if (r.h) break;
            r.err = dlerror();
            if (nm == names[0]) return;  // an explicit DSH_RCCL_LIB that does not load is an error, not a hint
        }
        if (!r.h) return;
        auto sym = [&](const char *name) -> void * {
            void *p = dlsym(r.h, name);
            if (!p) r.err = std::string("librccl: missing symbol ") + name;
            return p;
        };
        r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
        r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
        r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
        r.CommAbort = (decltype(r.CommAbort))sym("ncclCommAbort");
        r.GetVersion = (decltype(r.GetVersion))sym("ncclGetVersion");
        r.GroupStart = (decltype(r.GroupStart))sym("ncclGroupStart");
        r.GroupEnd = (decltype(r.GroupEnd))sym("ncclGroupEnd");
        r.Send = (decltype(r.Send))sym("ncclSend");
        r.Recv = (decltype(r.Recv))sym("ncclRecv");
        r.AllGather = (decltype(r.AllGather))sym("ncclAllGather");
        r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
        if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.CommAbort || !r.GetVersion || !r.GroupStart ||
            !r.GroupEnd || !r.Send || !r.Recv || !r.AllGather || !r.GetErrorString) {
            dlclose(r.h);
            r.h = nullptr;
            return;
        }
        Dl_info di;
        if (dladdr((void *)r.GetUniqueId, &di) && di.dli_fname) r.path = di.dli_fname;
    });
    return &r;
}

#define NCCLCHK(c, expr)                                                                                   \
    do {                                                                                                   \
        ncclResult_t r_ = (expr);                                                                          \
        if (r_ != ncclSuccess) return fail((c), DSH_EIO, "%s: %s", #expr, rccl()->GetErrorString(r_));     \
    } while (0)

double env_seconds(const char *name, double dflt)
{
    const char *e = std::getenv(name);
    if (!e || !*e) return dflt;
    const double v = std::atof(e);
    return v > 0 ? v : dflt;
}

// Wait for a stream that carries RCCL traffic, with a deadline: a peer that never posts its side of a grouped
// send/recv would otherwise block this rank for ever.  On expiry the communicator is aborted (its kernels leave the
// stream) and the call fails with a message; DSH_COMM_TIMEOUT_S (default 120) sets the deadline.
int sync_guarded(dsh_ctx *c, hipStream_t st, const char *what)
{
    if (!c->comm || c->comm_world == 1) {
        HIPCHK(c, hipStreamSynchronize(st));
        return DSH_OK;
    }
    const double limit = env_seconds("DSH_COMM_TIMEOUT_S", 120.0);
    const auto t0 = std::chrono::steady_clock::now();
    unsigned spins = 0;
    for (;;) {
        const hipError_t q = hipStreamQuery(st);
        if (q == hipSuccess) return DSH_OK;
        if (q != hipErrorNotReady) return fail(c, DSH_EIO, "%s: %s", what, hipGetErrorString(q));
        if (++spins > 2000) std::this_thread::sleep_for(std::chrono::microseconds(50));
        if ((spins & 1023u) == 0 &&
            std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > limit) {
            (void)rccl()->CommAbort(c->comm);
            c->comm = nullptr;
            c->comm_rank = 0;
            c->comm_world = 1;
            return fail(c, DSH_EIO, "%s: the RCCL exchange did not complete within %.0f s (DSH_COMM_TIMEOUT_S): a peer "
                                    "is missing or stuck; the communicator was aborted", what, limit);
        }
    }
}

// the copy stream waits for part q of the last call with parts: for the flag k_finalize writes from inside its launch
// (hipStreamWaitValue32: >= the call's generation) or, where the call marked its parts with events, for the event
int wait_part(dsh_ctx *c, size_t q)
{
    if (c->parts_signalled) {
        if (q >= kSigMaxParts || !c->sig.ptr) return fail(c, DSH_ESTATE, "internal: part %zu has no flag", q);
        HIPCHK(c, hipStreamWaitValue32(c->copy_stream, (uint32_t *)c->sig.ptr + kSigPartFlag + q, c->sig_gen, hipStreamWaitValueGte, 0xFFFFFFFFu));
        return DSH_OK;
    }
    if (q >= c->ev_part.size()) return fail(c, DSH_ESTATE, "internal: part %zu has no event", q);
    HIPCHK(c, hipStreamWaitEvent(c->copy_stream, c->ev_part[q], 0));
    return DSH_OK;
}

// bounds of `world` row ranges over n rows (dsh_balance_rows / dsh_partition_rows) and a destination rank: ONE check
// for every entry point that takes them
int validate_bounds(dsh_ctx *c, uint64_t n, const uint64_t *bounds, int world, int dst)
{
    if (!bounds) return fail(c, DSH_EINVAL, "bounds is NULL");
    if (dst < 0 || dst >= world) return fail(c, DSH_EINVAL, "bad destination rank %d (world %d)", dst, world);
    if (bounds[0] != 0 || bounds[world] != n) return fail(c, DSH_EINVAL, "bounds must run from 0 to n over the %d ranks", world);
    for (int r = 0; r < world; ++r)
        if (bounds[r] > bounds[r + 1]) return fail(c, DSH_EINVAL, "bounds not monotone at rank %d", r);
    return DSH_OK;
}

}  // namespace

namespace dsh {

int comm_release(dsh_ctx *c)
{
    if (!c->comm) return DSH_OK;
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->copy_stream) (void)hipStreamSynchronize(c->copy_stream);  // dsh_collect_parts_async sends there
    const ncclResult_t e = rccl()->CommDestroy(c->comm);
    c->comm = nullptr;
    c->comm_rank = 0;
    c->comm_world = 1;
    return e == ncclSuccess ? DSH_OK : fail(c, DSH_EIO, "ncclCommDestroy: %s", rccl()->GetErrorString(e));
}

}  // namespace dsh

extern "C" {

/* ---- multi-GPU exchange through RCCL (one context per rank; ranks may be processes or threads) ------------------ */
int dsh_comm_available(void) { return rccl()->h ? DSH_OK : DSH_ENODEV; }

int dsh_comm_library(char *path_out, size_t cap, int *version_out)
{
    Rccl *r = rccl();
    if (!r->h) {
        if (path_out && cap) std::snprintf(path_out, cap, "%s", r->err.c_str());  // why it is not there
        return DSH_ENODEV;
    }
    if (path_out && cap) std::snprintf(path_out, cap, "%s", r->path.c_str());
    if (version_out) {
        int v = 0;
        if (r->GetVersion(&v) != ncclSuccess) v = 0;
        *version_out = v;
    }
    return DSH_OK;
}

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
    if ((rc = comm_release(c))) return rc;  // re-initialisation: nothing of the old communicator is in flight any more
    ncclUniqueId id;
    std::memcpy(&id, unique_id, sizeof id);
    // ncclCommInitRank blocks until every rank has joined.  It runs on a helper thread so that a rank that never
    // arrives costs this one DSH_COMM_INIT_TIMEOUT_S (default 90 s) and an error, not a hang.  On expiry the helper
    // stays blocked inside RCCL and is left behind; the context works on without a communicator, and should the peers
    // arrive after all, the helper ABORTS the communicator it then gets (the peers would otherwise see a
    // successful init and stall until their exchange deadline).  All ranks must treat an init timeout as fatal.
    struct Shared {
        std::mutex mu;
        std::condition_variable cv;
        bool done = false;
        bool abandoned = false;
        ncclResult_t res = ncclSuccess;
        ncclComm_t comm = nullptr;
    };
    auto sh = std::make_shared<Shared>();
    const int device = c->device;
    auto init = r->CommInitRank;
    auto abort_ = r->CommAbort;
    std::thread([sh, device, init, abort_, world, id, rank] {
        ncclComm_t comm = nullptr;
        ncclResult_t res = hipSetDevice(device) == hipSuccess ? init(&comm, world, id, rank) : ncclUnhandledCudaError;
        bool orphan = false;
        {
            std::lock_guard<std::mutex> lk(sh->mu);
            sh->res = res;
            sh->comm = comm;
            sh->done = true;
            orphan = sh->abandoned;
            sh->cv.notify_all();
        }
        if (orphan && res == ncclSuccess && comm) (void)abort_(comm);  // nobody will ever use or destroy it
    }).detach();
    const double limit = env_seconds("DSH_COMM_INIT_TIMEOUT_S", 90.0);
    {
        std::unique_lock<std::mutex> lk(sh->mu);
        if (!sh->cv.wait_for(lk, std::chrono::duration<double>(limit), [&] { return sh->done; })) {
            sh->abandoned = true;
            return fail(c, DSH_EIO, "ncclCommInitRank(rank %d of %d) did not return within %.0f s (DSH_COMM_INIT_TIMEOUT_S): "
                                    "not every rank joined", rank, world, limit);
        }
        if (sh->res != ncclSuccess) return fail(c, DSH_EIO, "ncclCommInitRank: %s", r->GetErrorString(sh->res));
        c->comm = sh->comm;
    }
    c->comm_rank = rank;
    c->comm_world = world;
    return DSH_OK;
}

int dsh_comm_destroy(dsh_ctx *c)
{
    if (!c) return DSH_EINVAL;
    int rc = bind(c);
    if (rc) return rc;
    return comm_release(c);
}

int dsh_comm_rank(const dsh_ctx *c, int *rank, int *world)
{
    if (!c) return DSH_EINVAL;
    if (rank) *rank = c->comm_rank;
    if (world) *world = c->comm ? c->comm_world : 1;
    return c->comm ? DSH_OK : DSH_ESTATE;
}

int dsh_comm_wait(dsh_ctx *c)
{
    if (!c) return DSH_EINVAL;
    int rc = bind(c);
    if (rc) return rc;
    if ((rc = sync_guarded(c, c->stream, "dsh_comm_wait (ctx stream)"))) return rc;
    return sync_guarded(c, c->copy_stream, "dsh_comm_wait (copy stream)");
}

static int collect_spans(dsh_ctx *c, uint64_t n, const uint64_t *bounds, const void *d_local, void *d_final, int dst)
{
    const int world = c->comm ? c->comm_world : 1, rank = c->comm ? c->comm_rank : 0;
    int rc = validate_bounds(c, n, bounds, world, dst);
    if (rc) return rc;
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
    return sync_guarded(c, c->stream, "dsh_collect_spans");
}

int dsh_collect_parts_async(dsh_ctx *c, uint64_t n, const uint64_t *bounds, uint32_t nparts, const void *d_local,
                            void *d_final, int dst)
{
    if (!c || !bounds || nparts == 0) return DSH_EINVAL;
    int rc = bind(c);
    if (rc) return rc;
    const int world = c->comm ? c->comm_world : 1, rank = c->comm ? c->comm_rank : 0;
    if (!c->comm && !(bounds[0] == 0 && bounds[1] == n)) return fail(c, DSH_ESTATE, "dsh_comm_init first");
    if ((rc = validate_bounds(c, n, bounds, world, dst))) return rc;
    if (rank == dst && !d_final) return DSH_EINVAL;
    // every rank's parts, from the same function the compute used (dsh_range_parts): both sides of a message agree.
    // dsh_dist_rows_parts_device_async lays ANY non-empty range out in exactly these parts (a short range: one part) and
    // records an event for each, so a rank with few rows takes part in the rounds like every other.
    std::vector<std::vector<uint64_t>> parts((size_t)world);
    size_t maxparts = 0;
    for (int r = 0; r < world; ++r) {
        plan::range_parts(n, bounds[r], bounds[r + 1], nparts, parts[r]);
        maxparts = std::max(maxparts, parts[r].size() - 1);
    }
    const size_t myparts = parts[rank].size() - 1;
    if (c->parts_done != myparts)
        return fail(c, DSH_ESTATE, "dsh_dist_rows_parts_device_async(%u parts) of this rank's rows must come first (%u parts computed, %zu expected)",
                    nparts, c->parts_done, myparts);
    Rccl *rc_ = world > 1 ? rccl() : nullptr;
    const uint64_t my_off = dsh_tri_span(n, 0, bounds[rank]);
    for (size_t q = 0; q < maxparts; ++q) {
        // round q: part q of every rank.  The copy stream joins this rank's "part q done" event; the transfer then runs
        // there while the ctx stream computes part q+1
        if (q < myparts) { int rcw = wait_part(c, q); if (rcw) return rcw; }
        if (rank == dst && q < myparts && d_local) {
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

}  // extern "C"

/* ---- the exchange-aware pair: every rank's buffer laid out for the exchange (plan.h: row sets, row-sorted parts) ------ */
namespace {

// what rank r does under dsh_exchange_*: the destination keeps its rows in place as ONE part; a short range, or any rank
// with extra segments, goes in row-sorted parts (every wanted segment key-ordered as one run, homogeneous tiles); a long
// range in parts of consecutive rows
struct XMode {
    bool rowsorted = false;
    uint64_t rb = 0, re = 0;      // the rank's main range
    std::vector<uint64_t> extra;  // its extra segments {b0, e0, ...}
    std::vector<uint64_t> cut;    // rowsorted: counts of wanted rows (wanted order); else row boundaries (range_parts)
    uint32_t kreq = 1;            // the number of parts asked of the part functions (what the compute call passes on)
    size_t nparts() const { return cut.empty() ? 0 : cut.size() - 1; }
    uint64_t span(uint64_t n) const { return plan::rowset_span(n, rb, re, extra); }
};

// Every rank's mode.  The parts are units of COMPLETION (a part's rows are final when k_finalize has passed them); what
// travels are MESSAGES: the q-th of `nparts` equal pieces of a source's buffer, sent as soon as the parts it overlaps are
// final (dsh_exchange_collect_async) -- the destination receives message q of every source in one round, and a round lasts
// as long as its largest message, so the messages of all sources are the same share of (about equal) spans.
// `fine_rank` >= 0: that rank (the caller, when its parts announce themselves from inside k_finalize: a part then costs a
// flag, not a launch) cuts its row-sorted rows as finely as they complete -- every tile row a part: a message waits for
// the part that holds its last value, and a coarse part that ends just short of a message's end would hold it back by a
// whole part.  Nobody else needs to know: the destination deals in messages and rows only.
std::vector<XMode> xmodes(uint64_t n, const plan::RowSets &rs, uint32_t nparts, int dst, int fine_rank = -1)
{
    std::vector<XMode> modes(rs.world);
    for (uint32_t r = 0; r < rs.world; ++r) {
        XMode &m = modes[r];
        rs.rank_rows(r, m.rb, m.re, m.extra);
        if (m.rb >= m.re) continue;
        if ((int)r == dst) {
            m.cut = {m.rb, m.re};
            continue;
        }
        m.rowsorted = !m.extra.empty() || plan::rowsorted_rule(n, m.rb, m.re, nparts);
        m.kreq = (m.rowsorted && (int)r == fine_rank) ? std::max<uint32_t>(nparts, kSigMaxParts) : nparts;
        if (m.rowsorted) plan::rowsorted_part_positions(n, m.rb, m.re, m.kreq, m.cut, &m.extra);
        else plan::range_parts(n, m.rb, m.re, nparts, m.cut);
    }
    return modes;
}

XMode xmode(uint64_t n, const plan::RowSets &rs, int r, uint32_t nparts, int dst) { return xmodes(n, rs, nparts, dst)[(size_t)r]; }

int parse_table(dsh_ctx *c, uint64_t n, const uint64_t *rowsets, int dst, plan::RowSets &rs)
{
    if (const char *why = plan::parse_rowsets(rowsets, n, rs)) return fail(c, DSH_EINVAL, "%s", why);
    if (dst < 0 || (uint32_t)dst >= rs.world) return fail(c, DSH_EINVAL, "bad destination rank %d (world %u)", dst, rs.world);
    return DSH_OK;
}

// the key order and the row offsets of a rank's row-sorted buffer (wanted rows: the extra segments, then the main range, each
// key-ordered as one run -- the main range as two where plan::rowsorted_split cuts it), from THIS context's host copy of the keys (every rank holds every sketch, the per-sketch pass
// is deterministic: the destination derives what the source used)
int rowsorted_tables(dsh_ctx *c, uint64_t n, const XMode &m, std::vector<uint32_t> &order, std::vector<uint64_t> &rowoff,
                     std::vector<uint32_t> &scratch)
{
    if (!c->hk32_valid || c->card_from > m.rb)
        return fail(c, DSH_ESTATE, "the per-sketch pass of this context does not cover the rows from %llu on (compute this rank's rows first)",
                    (unsigned long long)m.rb);
    const uint64_t cnt = plan::rowset_rows(m.rb, m.re, m.extra);
    order.resize(cnt);
    std::vector<std::pair<uint64_t, uint64_t>> segs;  // the wanted order: extra segments first, then the main range
    plan::wanted_order(n, m.rb, m.re, &m.extra, segs, true);
    uint64_t at = 0;
    for (auto &sg : segs) {
        plan::sort_rows_by_key(c->hk32, sg.first, sg.second, order.data() + at, scratch);
        at += sg.second - sg.first;
    }
    plan::rowsorted_offsets(n, order.data(), cnt, rowoff);
    return DSH_OK;
}

// the destination's own rows, computed relative to its first row: every segment of them to its place in the final matrix
// (a no-op when they were computed in place)
int place_own_rows(dsh_ctx *c, uint64_t n, const XMode &m, const void *d_local, void *d_final, hipStream_t st)
{
    const uint64_t base = dsh_tri_span(n, 0, m.rb);
    if (!d_local || (const float *)d_local == (float *)d_final + base) return DSH_OK;
    auto seg = [&](uint64_t b, uint64_t e) -> int {
        const uint64_t off = dsh_tri_span(n, 0, b), cnt = dsh_tri_span(n, b, e);
        if (cnt)
            HIPCHK(c, hipMemcpyAsync((float *)d_final + off, (const float *)d_local + (off - base), cnt * sizeof(float), hipMemcpyDeviceToDevice, st));
        return DSH_OK;
    };
    int rc = seg(m.rb, m.re);
    for (size_t x = 0; !rc && x + 1 < m.extra.size(); x += 2) rc = seg(m.extra[x], m.extra[x + 1]);
    return rc;
}

}  // namespace

extern "C" {

int dsh_exchange_mode(uint64_t n, const uint64_t *rowsets, int rank, uint32_t nparts, int dst, int *rowsorted, uint32_t *nparts_out,
                      uint64_t *local_floats_out)
{
    plan::RowSets rs;
    if (nparts == 0 || plan::parse_rowsets(rowsets, n, rs) || rank < 0 || (uint32_t)rank >= rs.world || dst < 0 || (uint32_t)dst >= rs.world)
        return DSH_EINVAL;
    const XMode m = xmode(n, rs, rank, nparts, dst);
    if (rowsorted) *rowsorted = m.rowsorted ? 1 : 0;
    if (nparts_out) *nparts_out = (uint32_t)m.nparts();
    if (local_floats_out) {
        // the destination computes relative to its first row (in place in the final matrix, or a buffer that reaches to
        // the end of its last segment); every other rank's buffer holds exactly its rows
        if (m.rb >= m.re) *local_floats_out = 0;
        else if (rank == dst) *local_floats_out = dsh_tri_span(n, m.rb, m.extra.empty() ? m.re : m.extra.back());
        else *local_floats_out = m.span(n);
    }
    return DSH_OK;
}

int dsh_exchange_rows_device_async(dsh_ctx *c, int estim, int result_type, int k, const uint64_t *rowsets, int rank, uint32_t nparts,
                                   int dst, void *d_local)
{
    if (!c || !rowsets || nparts == 0 || rank < 0) return DSH_EINVAL;
    int rc = bind(c);
    if (rc) return rc;
    if (!c->have_sketches) return fail(c, DSH_ESTATE, "no sketches loaded");
    plan::RowSets rs;
    if ((rc = parse_table(c, c->n, rowsets, dst, rs))) return rc;
    if ((uint32_t)rank >= rs.world) return fail(c, DSH_EINVAL, "bad rank %d (world %u)", rank, rs.world);
    reset_prof(c);
    c->parts_done = 0;
    if (c->n < 2) return DSH_OK;
    const std::vector<XMode> modes = xmodes(c->n, rs, nparts, dst, parts_will_signal(c) ? rank : -1);
    const XMode &m = modes[(size_t)rank];
    // the destination places the row-sorted spans of the others: its per-sketch pass must cover their rows too -- also
    // when it holds no rows itself (it used to return before any pass and fail later in the collect, with the
    // peers' sends already posted)
    bool from_zero = false;
    uint64_t first_needed = m.rb < m.re ? m.rb : c->n;
    if (rank == dst)
        for (uint32_t r = 0; r < rs.world; ++r) {
            if ((int)r == dst) continue;
            const XMode &o = modes[r];
            if (o.rowsorted && o.rb < first_needed) from_zero = true;
        }
    if (m.rb >= m.re) {
        if (!from_zero) return DSH_OK;
        c->pass_from_zero = true;
        rc = prepare(c, estim, -1, /*card_only=*/true);
        c->pass_from_zero = false;
        if (rc) return rc;
        if (!c->hk32_valid) {  // (a cardinality pass does not wait for its keys: the collect's tables need them)
            HIPCHK(c, hipEventSynchronize(c->ev_keys));
            c->hk32_valid = true;
        }
        return DSH_OK;
    }
    if (!d_local) return DSH_EINVAL;
    c->pass_from_zero = from_zero;
    PairJob j;
    j.estim = estim;
    j.result_type = result_type;
    j.k = k;
    j.rect = 0;
    j.nparts = rank == dst ? 1 : m.kreq;
    j.rowsorted = m.rowsorted ? 1 : 0;
    j.row_begin = m.rb;
    j.row_end = m.re;
    j.extra = m.extra;
    j.col_begin = j.col_end = 0;
    j.base_index = dsh_tri_span(c->n, 0, m.rb);
    j.d_out = (float *)d_local;
    rc = run_pairs(c, j);
    c->pass_from_zero = false;
    return rc;
}

// message q of M of a buffer of `total` floats: [first, first + cnt)
static void message_span(uint64_t total, size_t q, size_t M, uint64_t &first, uint64_t &cnt)
{
    first = (uint64_t)((unsigned __int128)total * q / M);
    cnt = (uint64_t)((unsigned __int128)total * (q + 1) / M) - first;
}

int dsh_exchange_collect_async(dsh_ctx *c, uint64_t n, const uint64_t *rowsets, uint32_t nparts, const void *d_local, void *d_final,
                               int dst)
{
    if (!c || !rowsets || nparts == 0) return DSH_EINVAL;
    int rc = bind(c);
    if (rc) return rc;
    plan::RowSets rs;
    if ((rc = parse_table(c, n, rowsets, dst, rs))) return rc;
    const int world = c->comm ? c->comm_world : 1, rank = c->comm ? c->comm_rank : 0;
    if (!c->comm && rs.world != 1) return fail(c, DSH_ESTATE, "dsh_comm_init first");
    if ((int)rs.world != world) return fail(c, DSH_EINVAL, "the row-set table is for %u ranks, the communicator has %d", rs.world, world);
    if (rank == dst && !d_final) return DSH_EINVAL;
    const std::vector<XMode> modes = xmodes(n, rs, nparts, dst, c->parts_signalled ? rank : -1);  // (as this rank's compute call cut them)
    const XMode &mine = modes[rank];
    if (n < 2) return DSH_OK;  // (dsh_exchange_rows_device_async computed nothing either: no pairs, no messages)
    if (c->parts_done != mine.nparts())
        return fail(c, DSH_ESTATE, "dsh_exchange_rows_device_async of this rank's rows must come first (%u parts computed, %zu expected)",
                    c->parts_done, mine.nparts());
    Rccl *rc_ = world > 1 ? rccl() : nullptr;
    // The exchange runs in M = nparts ROUNDS: message q of every source -- the q-th M-th of its buffer, whatever rows that
    // is -- in one grouped ncclSend/ncclRecv.  A round lasts as long as its largest message, and the ranks' spans are
    // about equal (dsh_balance_rowsets), so no link waits for another's longer message.  A source sends message q once
    // the part that holds its last value is final (parts = units of completion, final in order); the destination's own rows
    // are one part, computed in place, and gate nothing but the end of the call.
    const size_t M = nparts;
    // row-sorted sources: their key order and row offsets.  A source knows its own (the layout); the destination derives
    // them from its keys, stages what it receives and puts the rows that are complete into place behind every round.
    std::vector<std::vector<uint64_t>> rowoff((size_t)world);
    std::vector<std::vector<uint32_t>> order((size_t)world);
    std::vector<uint64_t> stage_off((size_t)world, 0), tab_off((size_t)world, 0), total((size_t)world, 0);
    uint64_t stage_total = 0, tab_bytes = 0;
    std::vector<PlaceEnt> ents;  // the placement launches of the rounds (destination)
    std::vector<int> ent_src;
    std::vector<uint32_t> round_ent(M + 1, 0), round_rows(M, 0);
    uint64_t ent_off = 0;
    for (int r = 0; r < world; ++r)
        if (r != dst && !modes[r].rowsorted) total[r] = dsh_tri_span(n, modes[r].rb, modes[r].re);
    if (rank != dst) {
        if (mine.rowsorted) {
            rowoff[rank] = c->lay.rowoff_w;
            total[rank] = rowoff[rank].empty() ? 0 : rowoff[rank].back();
        }
    } else {
        for (int r = 0; r < world; ++r) {
            if (r == dst || !modes[r].rowsorted) continue;
            if ((rc = rowsorted_tables(c, n, modes[r], order[r], rowoff[r], c->lay.sort_a))) return rc;
            total[r] = rowoff[r].back();
            stage_off[r] = stage_total;
            stage_total += rowoff[r].back();
            tab_off[r] = tab_bytes;
            tab_bytes += (rowoff[r].size() * sizeof(uint64_t) + order[r].size() * sizeof(uint32_t) + 15) & ~(uint64_t)15;
        }
        if (stage_total) {
            HIPCHK(c, c->xch_stage.ensure(stage_total * sizeof(float)));
            if (c->xch_tab_in_flight) {  // the previous call's upload from the pinned tables
                HIPCHK(c, hipEventSynchronize(c->ev_xch_tab));
                c->xch_tab_in_flight = false;
            }
            // behind the sources' tables: the entries of every round's placement launch (k_rows_place) -- the rows of a
            // source that lie completely inside its first q + 1 messages and did not inside its first q
            ent_off = tab_bytes;
            std::vector<uint64_t> rows_in((size_t)world, 0);
            for (size_t q = 0; q < M; ++q) {
                round_ent[q] = (uint32_t)ents.size();
                uint32_t row0 = 0;
                for (int src = 0; src < world; ++src) {
                    if (src == dst || !modes[src].rowsorted) continue;
                    uint64_t first, cnt;
                    message_span(total[src], q, M, first, cnt);
                    const std::vector<uint64_t> &ro = rowoff[src];  // ro[s] = start of the row at position s, ro.back() = total
                    const uint64_t p1 = (uint64_t)(std::upper_bound(ro.begin(), ro.end(), first + cnt) - ro.begin()) - 1;
                    const uint64_t p0 = rows_in[src];
                    rows_in[src] = std::max(p0, p1);
                    if (p1 <= p0) continue;
                    PlaceEnt e;
                    e.src = nullptr, e.order = nullptr, e.rowoff = nullptr;  // (device addresses: below, once the buffers exist)
                    e.pos0 = p0;
                    e.row0 = row0;
                    e.nrows = (uint32_t)(p1 - p0);
                    row0 += e.nrows;
                    ents.push_back(e);
                    ent_src.push_back(src);
                }
                round_rows[q] = row0;
            }
            round_ent[M] = (uint32_t)ents.size();
            tab_bytes += ents.size() * sizeof(PlaceEnt);
            HIPCHK(c, c->pin_xch.ensure(tab_bytes));
            HIPCHK(c, c->xch_tab.ensure(tab_bytes));
            for (int r = 0; r < world; ++r) {
                if (r == dst || !modes[r].rowsorted) continue;
                uint8_t *h = (uint8_t *)c->pin_xch.ptr + tab_off[r];
                std::memcpy(h, rowoff[r].data(), rowoff[r].size() * sizeof(uint64_t));
                std::memcpy(h + rowoff[r].size() * sizeof(uint64_t), order[r].data(), order[r].size() * sizeof(uint32_t));
            }
            for (size_t x = 0; x < ents.size(); ++x) {
                const int src = ent_src[x];
                const uint8_t *t = (const uint8_t *)c->xch_tab.ptr + tab_off[src];
                ents[x].src = (const float *)c->xch_stage.ptr + stage_off[src];
                ents[x].rowoff = (const uint64_t *)t;
                ents[x].order = (const uint32_t *)(t + rowoff[src].size() * sizeof(uint64_t));
            }
            if (!ents.empty()) std::memcpy((uint8_t *)c->pin_xch.ptr + ent_off, ents.data(), ents.size() * sizeof(PlaceEnt));
            HIPCHK(c, hipMemcpyAsync(c->xch_tab.ptr, c->pin_xch.ptr, tab_bytes, hipMemcpyHostToDevice, c->copy_stream));
            if (!c->ev_xch_tab) HIPCHK(c, hipEventCreateWithFlags(&c->ev_xch_tab, hipEventDisableTiming));
            HIPCHK(c, hipEventRecord(c->ev_xch_tab, c->copy_stream));
            c->xch_tab_in_flight = true;
            while (c->ev_round.size() < M) {
                hipEvent_t e = nullptr;
                HIPCHK(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
                c->ev_round.push_back(e);
            }
        }
    }
    // (source) where this rank's parts end in its buffer: message q waits for the part that holds its last value
    std::vector<uint64_t> part_end;
    if (rank != dst)
        for (size_t i = 0; i < mine.nparts(); ++i)
            part_end.push_back(mine.rowsorted ? rowoff[rank][mine.cut[i + 1]] : dsh_tri_span(n, mine.rb, mine.cut[i + 1]));
    // The destination's receives: an RCCL kernel that waits for its peers spins on the GPU; beside the phase-locked tile
    // kernel it would slow every round down to the CU it shares.  A SHORT job of the destination (one launch of at most 8
    // rounds: its peers then run 6 + 1 rounds and have nothing to send before about the same moment) therefore posts its
    // receives behind its tile kernel; longer jobs post at once -- their peers' first messages come much earlier (option
    // xch_recv_gate: -1 auto | 0 at once | 1 behind the first tile kernel).
    if (rank == dst && world > 1 && mine.nparts() && c->ev_first_tiles) {
        const uint64_t rounds = (c->pp.items.size() + 511) / 512;
        const bool gate = c->xch_recv_gate == 1 || (c->xch_recv_gate < 0 && c->last_bands == 1 && rounds <= 8);
        if (gate) HIPCHK(c, hipStreamWaitEvent(c->copy_stream, c->ev_first_tiles, 0));
        c->xch_recv_gated = gate;
    }
    bool placed = false;
    size_t next_wait = 0;  // (source) the first part the copy stream has not joined yet
    for (size_t q = 0; q < M && world > 1; ++q) {
        uint64_t my_first = 0, my_cnt = 0;
        if (rank != dst && mine.nparts()) {
            message_span(total[rank], q, M, my_first, my_cnt);
            if (my_cnt) {
                // the copy stream joins the completion (a flag k_finalize sets, or an event) of every part up to the one that
                // holds the message's last value: the parts of a band are final in ALMOST the order of the list
                const size_t i = std::min((size_t)(std::lower_bound(part_end.begin(), part_end.end(), my_first + my_cnt) - part_end.begin()),
                                          mine.nparts() - 1);
                for (; next_wait <= i; ++next_wait) {
                    int rcw = wait_part(c, next_wait);
                    if (rcw) return rcw;
                }
            }
        }
        bool any = my_cnt != 0;
        if (rank == dst)
            for (int src = 0; src < world && !any; ++src) {
                uint64_t f, k;
                message_span(src == dst ? 0 : total[src], q, M, f, k);
                any = k != 0;
            }
        if (!any) continue;
        NCCLCHK(c, rc_->GroupStart());
        ncclResult_t e = ncclSuccess;
        if (rank == dst) {
            for (int src = 0; src < world && e == ncclSuccess; ++src) {
                if (src == dst) continue;
                uint64_t first, cnt;
                message_span(total[src], q, M, first, cnt);
                if (!cnt) continue;
                float *to = modes[src].rowsorted ? (float *)c->xch_stage.ptr + stage_off[src] + first
                                                 : (float *)d_final + dsh_tri_span(n, 0, modes[src].rb) + first;
                e = rc_->Recv(to, cnt, ncclFloat32, src, c->comm, c->copy_stream);
            }
        } else {
            if (!d_local) e = ncclInvalidArgument;
            else e = rc_->Send((const float *)d_local + my_first, my_cnt, ncclFloat32, dst, c->comm, c->copy_stream);
        }
        if (e != ncclSuccess) {
            (void)rc_->GroupEnd();
            return fail(c, DSH_EIO, "ncclSend/ncclRecv: %s", rc_->GetErrorString(e));
        }
        NCCLCHK(c, rc_->GroupEnd());
        // the rows that are now complete go to their places: ONE launch per round, on a stream of its own behind the
        // round's event, so that the next round's transfer does not wait for it
        if (rank == dst && stage_total && round_rows[q]) {
            HIPCHK(c, hipEventRecord(c->ev_round[q], c->copy_stream));
            HIPCHK(c, hipStreamWaitEvent(c->place_stream, c->ev_round[q], 0));
            HIPCHK(c, launch_rows_place(c->place_stream, (const PlaceEnt *)((const uint8_t *)c->xch_tab.ptr + ent_off) + round_ent[q],
                                        round_ent[q + 1] - round_ent[q], round_rows[q], (float *)d_final, n));
            placed = true;
        }
    }
    if (rank == dst) {
        // the destination's own rows (one part, normally computed in place) and the placements: the copy stream, which
        // dsh_comm_wait waits for, ends behind both
        if (mine.nparts()) {
            int rcw = wait_part(c, 0);
            if (rcw) return rcw;
            if ((rc = place_own_rows(c, n, mine, d_local, d_final, c->copy_stream))) return rc;
        }
        if (placed) {
            if (!c->ev_place_done) HIPCHK(c, hipEventCreateWithFlags(&c->ev_place_done, hipEventDisableTiming));
            HIPCHK(c, hipEventRecord(c->ev_place_done, c->place_stream));
            HIPCHK(c, hipStreamWaitEvent(c->copy_stream, c->ev_place_done, 0));
        }
    }
    return DSH_OK;
}

int dsh_exchange_place_device(dsh_ctx *c, const uint64_t *rowsets, int src, uint32_t nparts, int dst, const void *d_src_local,
                              void *d_final)
{
    if (!c || !rowsets || nparts == 0 || src < 0 || !d_final) return DSH_EINVAL;
    int rc = bind(c);
    if (rc) return rc;
    if (!c->have_sketches) return fail(c, DSH_ESTATE, "no sketches loaded");
    const uint64_t n = c->n;
    plan::RowSets rs;
    if ((rc = parse_table(c, n, rowsets, dst, rs))) return rc;
    if ((uint32_t)src >= rs.world) return fail(c, DSH_EINVAL, "bad source rank %d (world %u)", src, rs.world);
    const XMode m = xmode(n, rs, src, nparts, dst);
    if (!m.span(n)) return DSH_OK;
    if (!d_src_local) return DSH_EINVAL;
    // (its own table buffer, on the ctx stream: dsh_exchange_collect_async keeps xch_tab busy on the copy stream -- ADVICE r4)
    if (!m.rowsorted) {  // final order already (relative to the rank's first row): every segment goes to its place
        if ((rc = place_own_rows(c, n, m, d_src_local, d_final, c->stream))) return rc;
        HIPCHK(c, hipStreamSynchronize(c->stream));
        return DSH_OK;
    }
    std::vector<uint32_t> order;
    std::vector<uint64_t> rowoff;
    if ((rc = rowsorted_tables(c, n, m, order, rowoff, c->lay.sort_a))) return rc;
    const size_t ro_bytes = rowoff.size() * sizeof(uint64_t), bytes = ro_bytes + order.size() * sizeof(uint32_t);
    HIPCHK(c, c->place_tab.ensure(bytes));
    HIPCHK(c, hipMemcpyAsync(c->place_tab.ptr, rowoff.data(), ro_bytes, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync((uint8_t *)c->place_tab.ptr + ro_bytes, order.data(), order.size() * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
    hipEvent_t e0 = nullptr, e1 = nullptr;  // (profiling: the device time of the placement itself, dsh_get_info "place_kernel_us")
    if (c->profiling) {
        HIPCHK(c, hipEventCreate(&e0));
        HIPCHK(c, hipEventCreate(&e1));
        HIPCHK(c, hipEventRecord(e0, c->stream));
    }
    HIPCHK(c, launch_row_place(c->stream, (const float *)d_src_local, (float *)d_final,
                               (const uint32_t *)((const uint8_t *)c->place_tab.ptr + ro_bytes), (const uint64_t *)c->place_tab.ptr,
                               0, order.size(), n));
    if (e1) HIPCHK(c, hipEventRecord(e1, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));  // (the pageable tables must outlive their copies)
    if (e1) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, e0, e1) == hipSuccess) c->place_ms = ms;
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
    }
    return DSH_OK;
}

/* ---- diagnostics of the exchange on ONE GPU (no communicator needed) ------------------------------------------------ */
namespace {

// what an RCCL send kernel does with a part: plain loads of the rank's buffer through the L2 of whatever XCD the
// workgroup runs on (NOT the copy engine, which reads memory), stores into another buffer
__global__ __launch_bounds__(256) void k_probe_copy(const float *__restrict__ src, float *__restrict__ dst, uint64_t cnt)
{
    for (uint64_t x = (uint64_t)blockIdx.x * 256 + threadIdx.x; x < cnt; x += (uint64_t)gridDim.x * 256) dst[x] = src[x];
}

// a kernel that waits like an RCCL receive kernel whose peers have nothing to send yet: `threads` lanes per workgroup,
// `lds` bytes of LDS held, lane 0 polling a word of page-locked host memory; leaves when the word is set or after
// max_ticks of the device's wall clock (so that a host that never comes back cannot hang the GPU)
__global__ void k_diag_spin(const uint32_t *flag, unsigned long long max_ticks, uint32_t *touched)
{
    extern __shared__ uint32_t spin_lds[];
    if (threadIdx.x == 0) spin_lds[0] = blockIdx.x;
    __syncthreads();
    const unsigned long long t0 = wall_clock64();
    for (;;) {
        uint32_t v = 0;
        if (threadIdx.x == 0) v = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        v = __shfl(v, 0);
        if (__syncthreads_or(v != 0 || wall_clock64() - t0 > max_ticks)) break;
        __builtin_amdgcn_s_sleep(8);
    }
    if (threadIdx.x == 0 && touched) atomicAdd(touched, spin_lds[0] + 1 - blockIdx.x);
}

}  // namespace

int dsh_exchange_probe_parts_async(dsh_ctx *c, uint64_t n, const uint64_t *rowsets, int rank, uint32_t nparts, int dst,
                                   const void *d_local, void *d_probe)
{
    if (!c || !rowsets || nparts == 0 || rank < 0 || !d_local || !d_probe) return DSH_EINVAL;
    int rc = bind(c);
    if (rc) return rc;
    plan::RowSets rs;
    if ((rc = parse_table(c, n, rowsets, dst, rs))) return rc;
    if ((uint32_t)rank >= rs.world) return fail(c, DSH_EINVAL, "bad rank %d (world %u)", rank, rs.world);
    const std::vector<XMode> modes = xmodes(n, rs, nparts, dst, c->parts_signalled ? rank : -1);  // (as the compute call cut them)
    const XMode &mine = modes[(size_t)rank];
    if (c->parts_done != mine.nparts())
        return fail(c, DSH_ESTATE, "dsh_exchange_rows_device_async of this rank's rows must come first (%u parts computed, %zu expected)",
                    c->parts_done, mine.nparts());
    // where the parts end in the rank's buffer (the destination computes relative to its first row, as one part)
    std::vector<uint64_t> part_end;
    for (size_t i = 0; i < mine.nparts(); ++i) {
        uint64_t end;
        if (rank == dst) end = dsh_tri_span(n, mine.rb, mine.extra.empty() ? mine.re : mine.extra.back());
        else if (mine.rowsorted) end = c->lay.rowoff_w[mine.cut[i + 1]];
        else end = dsh_tri_span(n, mine.rb, mine.cut[i + 1]);
        part_end.push_back(std::max(end, part_end.empty() ? 0 : part_end.back()));
    }
    if (c->comm && c->comm_world == 1) {
        // The reader is librccl itself: the rank's step as dsh_exchange_collect_async runs it for a source -- M = nparts
        // rounds, message q of the buffer behind the gate of the part that holds its last value, one grouped call per round
        // -- with the one peer a communicator of one rank has: ncclSend to itself paired with the ncclRecv into d_probe.
        Rccl *rc_ = rccl();
        const uint64_t total = part_end.empty() ? 0 : part_end.back();
        const size_t M = nparts;
        size_t next_wait = 0;
        for (size_t q = 0; q < M && total; ++q) {
            uint64_t first, cnt;
            message_span(total, q, M, first, cnt);
            if (!cnt) continue;
            const size_t i = std::min((size_t)(std::lower_bound(part_end.begin(), part_end.end(), first + cnt) - part_end.begin()), mine.nparts() - 1);
            for (; next_wait <= i; ++next_wait)
                if ((rc = wait_part(c, next_wait))) return rc;
            NCCLCHK(c, rc_->GroupStart());
            ncclResult_t e = rc_->Send((const float *)d_local + first, cnt, ncclFloat32, 0, c->comm, c->copy_stream);
            if (e == ncclSuccess) e = rc_->Recv((float *)d_probe + first, cnt, ncclFloat32, 0, c->comm, c->copy_stream);
            if (e != ncclSuccess) {
                (void)rc_->GroupEnd();
                return fail(c, DSH_EIO, "ncclSend/ncclRecv to this rank itself: %s", rc_->GetErrorString(e));
            }
            NCCLCHK(c, rc_->GroupEnd());
        }
        for (; next_wait < mine.nparts(); ++next_wait)  // (every gate is joined, as in the kernel-reader path)
            if ((rc = wait_part(c, next_wait))) return rc;
        return DSH_OK;
    }
    uint64_t at = 0;
    for (size_t i = 0; i < mine.nparts(); ++i) {
        const uint64_t end = part_end[i];
        if ((rc = wait_part(c, i))) return rc;
        if (end > at) {
            const uint64_t cnt = end - at;
            const uint32_t blocks = (uint32_t)std::min<uint64_t>((cnt + 255) / 256, 2048);
            hipLaunchKernelGGL(k_probe_copy, dim3(blocks), dim3(256), 0, c->copy_stream, (const float *)d_local + at, (float *)d_probe + at, cnt);
            HIPCHK(c, hipGetLastError());
        }
        at = end;
    }
    return DSH_OK;
}

int dsh_diag_spin_start(dsh_ctx *c, uint32_t nblocks, uint32_t threads, uint32_t lds_bytes, uint32_t max_ms)
{
    if (!c || nblocks == 0 || nblocks > 4096 || threads < 64 || threads > 1024 || (threads & 63) || lds_bytes > 160 * 1024 || max_ms == 0 || max_ms > 10000)
        return DSH_EINVAL;
    int rc = bind(c);
    if (rc) return rc;
    if (c->spin_running) return fail(c, DSH_ESTATE, "dsh_diag_spin_stop first");
    (void)device_can_wait_value(c);  // (also asks for the wall clock's rate)
    if (c->wall_clock_khz <= 0) return fail(c, DSH_ENODEV, "the device reports no wall clock rate");
    if (!c->spin_flag) HIPCHK(c, hipHostMalloc((void **)&c->spin_flag, 64, hipHostMallocMapped));
    if (!c->spin_stream) HIPCHK(c, hipStreamCreateWithFlags(&c->spin_stream, hipStreamNonBlocking));
    *(volatile uint32_t *)c->spin_flag = 0;
    uint32_t *dflag = nullptr;
    HIPCHK(c, hipHostGetDevicePointer((void **)&dflag, c->spin_flag, 0));
    if (lds_bytes > 64 * 1024) HIPCHK(c, ensure_dynamic_lds((const void *)k_diag_spin, lds_bytes));
    hipLaunchKernelGGL(k_diag_spin, dim3(nblocks), dim3(threads), std::max<uint32_t>(lds_bytes, 16), c->spin_stream, dflag,
                       (unsigned long long)max_ms * (unsigned long long)c->wall_clock_khz, (uint32_t *)nullptr);
    HIPCHK(c, hipGetLastError());
    c->spin_running = true;
    return DSH_OK;
}

int dsh_diag_spin_stop(dsh_ctx *c)
{
    if (!c) return DSH_EINVAL;
    int rc = bind(c);
    if (rc) return rc;
    if (!c->spin_running) return DSH_OK;
    *(volatile uint32_t *)c->spin_flag = 1;
    c->spin_running = false;
    HIPCHK(c, hipStreamSynchronize(c->spin_stream));
    return DSH_OK;
}

int dsh_allgather_device(dsh_ctx *c, const void *d_send, uint64_t bytes_per_rank, void *d_recv)
{
    if (!c || (bytes_per_rank && (!d_send || !d_recv))) return DSH_EINVAL;
    int rc = bind(c);
    if (rc) return rc;
    if (!c->comm) return fail(c, DSH_ESTATE, "dsh_comm_init first");
    if (bytes_per_rank) NCCLCHK(c, rccl()->AllGather(d_send, d_recv, bytes_per_rank, ncclUint8, c->comm, c->stream));
    return sync_guarded(c, c->stream, "dsh_allgather_device");
}

int dsh_dist_collect(dsh_ctx *c, int estim, int result_type, int k, const uint64_t *bounds, int dst, float *out)
{
    if (!c) return DSH_EINVAL;
    int rc = bind(c);
    if (rc) return rc;
    if (!c->have_sketches) return fail(c, DSH_ESTATE, "no sketches loaded");
    const int world = c->comm ? c->comm_world : 1, rank = c->comm ? c->comm_rank : 0;
    const uint64_t n = c->n, total = dsh_tri_span(n, 0, n);
    if (!bounds) {
        // the library's own partition: balanced row sets (range + top-up tile rows, dsh_balance_rowsets) through the
        // pipelined exchange pair -- what bench.py --gpus N times, for a host without device pointers
        if (dst < 0 || dst >= world) return fail(c, DSH_EINVAL, "bad destination rank %d (world %d)", dst, world);
        if (n < 2) return DSH_OK;  // no pairs: an empty matrix on every rank, nothing to exchange
        plan::RowSets rs;
        plan::balance_rowsets(n, (uint32_t)world, rs, ~0u, dst);
        std::vector<uint64_t> tab(rs.words());
        rs.write(tab.data());
        constexpr uint32_t kParts = 8;
        uint64_t local_floats = 0;
        if ((rc = dsh_exchange_mode(n, tab.data(), rank, kParts, dst, nullptr, nullptr, &local_floats))) return fail(c, rc, "internal: row-set table");
        void *d_loc = nullptr;
        if (rank == dst) {
            if (total && !out) return DSH_EINVAL;
            HIPCHK(c, c->gather_full.ensure(std::max<uint64_t>(total, 1) * sizeof(float)));
            uint64_t rb = 0, re = 0;
            std::vector<uint64_t> ex;
            rs.rank_rows((uint32_t)rank, rb, re, ex);
            d_loc = (float *)c->gather_full.ptr + dsh_tri_span(n, 0, rb);  // computed in place
        } else {
            HIPCHK(c, c->gather_local.ensure(std::max<uint64_t>(local_floats, 1) * sizeof(float)));
            d_loc = c->gather_local.ptr;
        }
        if ((rc = dsh_exchange_rows_device_async(c, estim, result_type, k, tab.data(), rank, kParts, dst, d_loc))) return rc;
        if ((rc = dsh_exchange_collect_async(c, n, tab.data(), kParts, rank == dst ? nullptr : d_loc, rank == dst ? c->gather_full.ptr : nullptr, dst)))
            return rc;
        if ((rc = sync_guarded(c, c->stream, "dsh_dist_collect (ctx stream)"))) return rc;
        if ((rc = sync_guarded(c, c->copy_stream, "dsh_dist_collect (copy stream)"))) return rc;
        if (rank == dst && total)
            HIPCHK(c, hipMemcpyAsync(out, c->gather_full.ptr, total * sizeof(float), hipMemcpyDeviceToHost, c->stream));
        return sync_guarded(c, c->stream, "dsh_dist_collect");
    }
    if ((rc = validate_bounds(c, n, bounds, world, dst))) return rc;  // before anything is sized by them
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
    return sync_guarded(c, c->stream, "dsh_dist_collect");
}

}  // extern "C"

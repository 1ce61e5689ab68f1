// mock_rccl.cpp -- TEST INFRASTRUCTURE, not a product path.  A stand-in for librccl.so with the point-to-point semantics the
// library's exchange relies on (include/dashing_hip.h: dsh_comm_*, dsh_collect_*, dsh_exchange_*), so that the exchange
// PROTOCOL -- who sends what to whom in which round, counts, offsets, staging and placement -- runs with world > 1 on a box
// with ONE GPU (RCCL refuses two ranks on one device, and the build has no multi-GPU box): several processes share cuda:0,
// libdashing_hip.so loads this file through DSH_RCCL_LIB, messages travel as files under /dev/shm.
//
// What it keeps of NCCL's contract: messages between an ordered pair of ranks match in issue order; a receive of N
// elements only matches a send of exactly N elements of the same type (a mismatch is an ERROR here, where real RCCL would
// corrupt or hang); calls between ncclGroupStart and ncclGroupEnd take effect at the outermost ncclGroupEnd; everything
// enqueued on the stream before the call is complete before the data is read, and the data is in place before anything
// enqueued afterwards runs.  What it does not keep: asynchrony (every operation blocks the calling thread until it is
// done), bandwidth, topology.  A peer that never shows up is a timeout error (MOCK_RCCL_TIMEOUT_S, default 120).
#include <dirent.h>
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

namespace {

struct Op {
    bool send;
    void *buf;
    size_t bytes;
    int peer;
    hipStream_t stream;
};

struct Comm {
    int rank = 0, world = 1;
    std::string dir;
    std::vector<uint64_t> sent, received;  // per peer: messages so far
};

thread_local int g_depth = 0;
thread_local std::vector<std::pair<Comm *, Op>> g_pending;

double timeout_s()
{
    const char *e = std::getenv("MOCK_RCCL_TIMEOUT_S");
    const double v = e ? std::atof(e) : 0;
    return v > 0 ? v : 120.0;
}

size_t type_size(ncclDataType_t t)
{
    switch (t) {
    case ncclInt8:
    case ncclUint8: return 1;
    case ncclFloat16:
    case ncclBfloat16: return 2;
    case ncclInt32:
    case ncclUint32:
    case ncclFloat32: return 4;
    case ncclInt64:
    case ncclUint64:
    case ncclFloat64: return 8;
    default: return 0;
    }
}

std::string msg_path(const Comm *c, int src, int dst, uint64_t seq)
{
    char b[64];
    std::snprintf(b, sizeof b, "/m_%d_%d_%llu", src, dst, (unsigned long long)seq);
    return c->dir + b;
}

bool wait_for(const std::string &path)
{
    const auto t0 = std::chrono::steady_clock::now();
    struct stat st;
    while (stat(path.c_str(), &st) != 0) {
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s()) return false;
        std::this_thread::sleep_for(std::chrono::microseconds(200));
    }
    return true;
}

ncclResult_t do_send(Comm *c, const Op &op)
{
    if (op.peer < 0 || op.peer >= c->world || op.peer == c->rank) return ncclInvalidArgument;
    if (hipStreamSynchronize(op.stream) != hipSuccess) return ncclUnhandledCudaError;  // what was enqueued before is done
    std::vector<char> host(op.bytes);
    if (op.bytes && hipMemcpy(host.data(), op.buf, op.bytes, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
    const std::string path = msg_path(c, c->rank, op.peer, c->sent[op.peer]++), tmp = path + ".tmp";
    FILE *f = std::fopen(tmp.c_str(), "wb");
    if (!f) return ncclSystemError;
    const bool ok = std::fwrite(host.data(), 1, op.bytes, f) == op.bytes;
    std::fclose(f);
    if (!ok || std::rename(tmp.c_str(), path.c_str()) != 0) return ncclSystemError;
    return ncclSuccess;
}

ncclResult_t do_recv(Comm *c, const Op &op)
{
    if (op.peer < 0 || op.peer >= c->world || op.peer == c->rank) return ncclInvalidArgument;
    const std::string path = msg_path(c, op.peer, c->rank, c->received[op.peer]++);
    if (!wait_for(path)) {
        std::fprintf(stderr, "mock_rccl: rank %d: no message %s within %.0f s\n", c->rank, path.c_str(), timeout_s());
        return ncclSystemError;
    }
    struct stat st;
    if (stat(path.c_str(), &st) != 0) return ncclSystemError;
    if ((size_t)st.st_size != op.bytes) {  // real RCCL would not notice: the point of the mock
        std::fprintf(stderr, "mock_rccl: rank %d expects %zu bytes from rank %d, the message holds %lld\n", c->rank, op.bytes, op.peer,
                     (long long)st.st_size);
        return ncclInvalidUsage;
    }
    std::vector<char> host(op.bytes);
    FILE *f = std::fopen(path.c_str(), "rb");
    if (!f) return ncclSystemError;
    const bool ok = std::fread(host.data(), 1, op.bytes, f) == op.bytes;
    std::fclose(f);
    unlink(path.c_str());
    if (!ok) return ncclSystemError;
    if (hipStreamSynchronize(op.stream) != hipSuccess) return ncclUnhandledCudaError;
    if (op.bytes && hipMemcpy(op.buf, host.data(), op.bytes, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
    return ncclSuccess;
}

ncclResult_t run_or_defer(Comm *c, const Op &op)
{
    if (g_depth > 0) {
        g_pending.emplace_back(c, op);
        return ncclSuccess;
    }
    return op.send ? do_send(c, op) : do_recv(c, op);
}

}  // namespace

extern "C" {

ncclResult_t ncclGetVersion(int *version)
{
    if (version) *version = 999999;  // (recognisably not a real RCCL)
    return ncclSuccess;
}

const char *ncclGetErrorString(ncclResult_t r)
{
    switch (r) {
    case ncclSuccess: return "no error";
    case ncclUnhandledCudaError: return "mock: HIP error";
    case ncclSystemError: return "mock: system error or timeout";
    case ncclInvalidArgument: return "mock: invalid argument";
    case ncclInvalidUsage: return "mock: message size does not match the receive";
    default: return "mock: error";
    }
}

ncclResult_t ncclGetUniqueId(ncclUniqueId *id)
{
    if (!id) return ncclInvalidArgument;
    std::memset(id, 0, sizeof *id);
    std::snprintf(id->internal, sizeof id->internal, "dsh_mock_%d_%lld", (int)getpid(),
                  (long long)std::chrono::steady_clock::now().time_since_epoch().count());
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank)
{
    if (!comm || nranks < 1 || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    id.internal[sizeof id.internal - 1] = 0;
    Comm *c = new Comm;
    c->rank = rank;
    c->world = nranks;
    c->dir = std::string("/dev/shm/") + id.internal;
    c->sent.assign(nranks, 0);
    c->received.assign(nranks, 0);
    mkdir(c->dir.c_str(), 0700);  // (whoever comes first)
    if (std::getenv("MOCK_RCCL_DEBUG")) std::fprintf(stderr, "mock_rccl: rank %d of %d joins %s\n", rank, nranks, c->dir.c_str());
    char b[32];
    std::snprintf(b, sizeof b, "/joined_%d", rank);
    FILE *f = std::fopen((c->dir + b).c_str(), "wb");
    if (!f) {
        delete c;
        return ncclSystemError;
    }
    std::fclose(f);
    for (int r = 0; r < nranks; ++r) {  // every rank has to join, as with the real thing
        std::snprintf(b, sizeof b, "/joined_%d", r);
        if (!wait_for(c->dir + b)) {
            std::fprintf(stderr, "mock_rccl: rank %d of %d: rank %d never joined %s\n", rank, nranks, r, c->dir.c_str());
            delete c;
            return ncclSystemError;
        }
    }
    if (std::getenv("MOCK_RCCL_DEBUG")) std::fprintf(stderr, "mock_rccl: rank %d of %d: all joined\n", rank, nranks);
    *comm = reinterpret_cast<ncclComm_t>(c);
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm)
{
    Comm *c = reinterpret_cast<Comm *>(comm);
    if (!c) return ncclSuccess;
    // (a rank that is done must not take its "joined" mark away: a slower rank may still be counting the marks inside
    // ncclCommInitRank.  Everyone leaves a "left" mark instead; whoever finds all of them there clears the directory.)
    char b[32];
    std::snprintf(b, sizeof b, "/left_%d", c->rank);
    if (FILE *f = std::fopen((c->dir + b).c_str(), "wb")) std::fclose(f);
    bool all = true;
    struct stat st;
    for (int r = 0; r < c->world && all; ++r) {
        std::snprintf(b, sizeof b, "/left_%d", r);
        all = stat((c->dir + b).c_str(), &st) == 0;
    }
    if (all) {
        for (int r = 0; r < c->world; ++r) {
            std::snprintf(b, sizeof b, "/left_%d", r);
            unlink((c->dir + b).c_str());
            std::snprintf(b, sizeof b, "/joined_%d", r);
            unlink((c->dir + b).c_str());
        }
        rmdir(c->dir.c_str());
    }
    delete c;
    return ncclSuccess;
}

ncclResult_t ncclCommAbort(ncclComm_t comm) { return ncclCommDestroy(comm); }

ncclResult_t ncclGroupStart()
{
    ++g_depth;
    return ncclSuccess;
}

ncclResult_t ncclGroupEnd()
{
    if (g_depth <= 0) return ncclInvalidUsage;
    if (--g_depth > 0) return ncclSuccess;
    // the sends first: a receive blocks until its message is there, and every rank may hold both kinds
    std::vector<std::pair<Comm *, Op>> ops;
    ops.swap(g_pending);
    ncclResult_t res = ncclSuccess;
    for (auto &o : ops)
        if (o.second.send && res == ncclSuccess) res = do_send(o.first, o.second);
    for (auto &o : ops)
        if (!o.second.send && res == ncclSuccess) res = do_recv(o.first, o.second);
    return res;
}

ncclResult_t ncclSend(const void *sendbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream)
{
    Comm *c = reinterpret_cast<Comm *>(comm);
    if (!c || !type_size(datatype)) return ncclInvalidArgument;
    return run_or_defer(c, Op{true, const_cast<void *>(sendbuff), count * type_size(datatype), peer, stream});
}

ncclResult_t ncclRecv(void *recvbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream)
{
    Comm *c = reinterpret_cast<Comm *>(comm);
    if (!c || !type_size(datatype)) return ncclInvalidArgument;
    return run_or_defer(c, Op{false, recvbuff, count * type_size(datatype), peer, stream});
}

ncclResult_t ncclAllGather(const void *sendbuff, void *recvbuff, size_t sendcount, ncclDataType_t datatype, ncclComm_t comm,
                           hipStream_t stream)
{
    Comm *c = reinterpret_cast<Comm *>(comm);
    if (!c || !type_size(datatype)) return ncclInvalidArgument;
    const size_t bytes = sendcount * type_size(datatype);
    if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
    if (bytes && hipMemcpy((char *)recvbuff + (size_t)c->rank * bytes, sendbuff, bytes, hipMemcpyDeviceToDevice) != hipSuccess)
        return ncclUnhandledCudaError;
    ncclResult_t res = ncclSuccess;
    for (int r = 0; r < c->world && res == ncclSuccess; ++r)
        if (r != c->rank) res = do_send(c, Op{true, const_cast<void *>(sendbuff), bytes, r, stream});
    for (int r = 0; r < c->world && res == ncclSuccess; ++r)
        if (r != c->rank) res = do_recv(c, Op{false, (char *)recvbuff + (size_t)r * bytes, bytes, r, stream});
    return res;
}

}  // extern "C"

// overlap_cabi.cpp -- the copy-out overlap of the asynchronous C-ABI measured from a plain C++ host (no Python, no
// torch: the process only holds the HIP runtime libdashing_hip.so links to, as the dashing-amd CLI does).
// Same job as tools/overlap_timing.py: N sketches of precision p (synthetic register arrays from the register law),
// the packed matrix delivered to page-locked host memory in row blocks of <= 256 Mi values.
//   g++ -O2 -std=c++17 tools/cabi/overlap_cabi.cpp -Iinclude -Ldashing_amd -ldashing_hip -Wl,-rpath,$PWD/dashing_amd -o /tmp/overlap_cabi
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "dashing_hip.h"

static uint64_t splitmix(uint64_t &s)
{
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(call)                                                                        \
    do {                                                                                \
        int rc_ = (call);                                                               \
        if (rc_) {                                                                      \
            std::fprintf(stderr, "%s = %d: %s\n", #call, rc_, dsh_last_error(ctx));     \
            return 1;                                                                   \
        }                                                                               \
    } while (0)

int main(int argc, char **argv)
{
    const uint64_t n = argc > 1 ? std::strtoull(argv[1], nullptr, 10) : 100000;
    const int p = argc > 2 ? std::atoi(argv[2]) : 10;
    const uint64_t m = 1ull << p;
    dsh_ctx *ctx = nullptr;
    if (int rc = dsh_create(0, &ctx)) {
        std::fprintf(stderr, "dsh_create = %d\n", rc);
        return 1;
    }
    CK(dsh_sketches_alloc(ctx, n, p));
    {  // register law: R = clip(ceil(log2((card / m) / -ln u)), 0, q + 1), cardinalities 2e6 .. 8e6
        std::vector<uint8_t> regs(4096 * m);
        uint64_t seed = 0x5EED;
        for (uint64_t s0 = 0; s0 < n; s0 += 4096) {
            const uint64_t cnt = std::min<uint64_t>(4096, n - s0);
            for (uint64_t s = 0; s < cnt; ++s) {
                const double lam = (2e6 + (double)(splitmix(seed) % 6000001)) / (double)m;
                for (uint64_t t = 0; t < m; ++t) {
                    const double u = ((double)(splitmix(seed) >> 11) + 0.5) / 9007199254740992.0;
                    double v = std::ceil(std::log2(lam / -std::log(u)));
                    if (v < 0) v = 0;
                    if (v > 64 - p + 1) v = 64 - p + 1;
                    regs[s * m + t] = (uint8_t)v;
                }
            }
            CK(dsh_upload_sketches(ctx, regs.data(), s0, cnt));
        }
    }
    const uint64_t block_vals = 256ull << 20;
    std::vector<uint64_t> cuts{0};
    while (cuts.back() < n) {
        const uint64_t rb = cuts.back();
        uint64_t re = rb + 1;
        while (re < n && dsh_tri_span(n, rb, re + 1) <= block_vals) ++re;
        cuts.push_back(re);
    }
    const size_t nb = cuts.size() - 1;
    uint64_t cap = 1;
    for (size_t b = 0; b < nb; ++b) cap = std::max<uint64_t>(cap, dsh_tri_span(n, cuts[b], cuts[b + 1]));
    float *pins[2] = {(float *)dsh_alloc_host(cap * 4), (float *)dsh_alloc_host(cap * 4)};
    if (!pins[0] || !pins[1]) return 1;
    double res[2] = {0, 0};
    for (int mode = 0; mode < 2; ++mode) {  // 0 serialized (dsh_wait per block), 1 overlapped (tickets)
        for (int rep = 0; rep < 2; ++rep) {  // first repetition warms up
            CK(dsh_wait(ctx));
            const double t0 = now();
            if (mode == 0) {
                for (size_t b = 0; b < nb; ++b) {
                    CK(dsh_dist_rows_async(ctx, DSH_ESTIM_ERTL_MLE, DSH_JI, 31, cuts[b], cuts[b + 1], pins[b & 1]));
                    CK(dsh_wait(ctx));
                }
            } else {
                std::vector<uint64_t> ticket(nb);
                CK(dsh_dist_rows_async(ctx, DSH_ESTIM_ERTL_MLE, DSH_JI, 31, cuts[0], cuts[1], pins[0]));
                CK(dsh_event_record(ctx, &ticket[0]));
                for (size_t b = 0; b < nb; ++b) {
                    if (b + 1 < nb) {
                        CK(dsh_dist_rows_async(ctx, DSH_ESTIM_ERTL_MLE, DSH_JI, 31, cuts[b + 1], cuts[b + 2], pins[(b + 1) & 1]));
                        CK(dsh_event_record(ctx, &ticket[b + 1]));
                    }
                    CK(dsh_event_wait(ctx, ticket[b]));
                }
            }
            CK(dsh_wait(ctx));
            res[mode] = now() - t0;
        }
    }
    std::printf("{\"host\": \"c++ over the C-ABI (HIP runtime of libdashing_hip.so only)\", \"n\": %llu, \"p\": %d, \"blocks\": %zu, "
                "\"bytes\": %llu, \"serialized_s\": %.4f, \"overlapped_s\": %.4f}\n",
                (unsigned long long)n, p, nb, (unsigned long long)(dsh_tri_span(n, 0, n) * 4), res[0], res[1]);
    dsh_free_host(pins[0]);
    dsh_free_host(pins[1]);
    dsh_destroy(ctx);
    return 0;
}

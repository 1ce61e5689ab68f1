// Micro-benchmark of k_pair_counts' inner pattern: 8x8 outer product of AND + popcount-accumulate,
// registers only (no LDS), to separate the VALU issue cost of the mix from LDS/barrier effects.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ __forceinline__ void popc_acc(uint32_t &acc, uint32_t x) { asm("v_bcnt_u32_b32 %0, %1, %0" : "+v"(acc) : "v"(x)); }
template <int MODE>
__global__ __launch_bounds__(256) void k(uint32_t *out, uint32_t s, int iters)
{
    uint32_t acc[8][8];
    for (int r = 0; r < 8; ++r) for (int c = 0; c < 8; ++c) acc[r][c] = 0;
    uint32_t av[8], bv[8];
    for (int r = 0; r < 8; ++r) { av[r] = threadIdx.x * 2654435761u + r * s; bv[r] = threadIdx.x * 40503u + r * 77u + s; }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                if (MODE == 0) popc_acc(acc[r][c], av[r] & bv[c]);          // and + bcnt
                if (MODE == 1) acc[r][c] += av[r] & bv[c];                  // and + add (both 2-cycle class)
                if (MODE == 2) popc_acc(acc[r][c], av[r]);                  // bcnt only
            }
#pragma unroll
        for (int r = 0; r < 8; ++r) { asm volatile("" : "+v"(av[r]), "+v"(bv[r])); }
    }
    uint32_t t = 0;
    for (int r = 0; r < 8; ++r) for (int c = 0; c < 8; ++c) t ^= acc[r][c];
    out[blockIdx.x * 256 + threadIdx.x] = t;
}
template <int MODE>
static void run(const char *name, int blocks_per_cu, int n_instr_per_iter)
{
    uint32_t *out;
    hipMalloc(&out, 256 * 64 * 256 * 4);
    const int iters = 2000, blocks = 256 * blocks_per_cu;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    k<MODE><<<blocks, 256>>>(out, 3, 10);
    hipDeviceSynchronize();
    hipEventRecord(a);
    k<MODE><<<blocks, 256>>>(out, 3, iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    const double insts = (double)blocks * 4 * iters * n_instr_per_iter;
    printf("%-28s wg/CU=%d  %7.3f ms  %.2f cycles/wave-instr  (%.2f per AND+BCNT pair)\n", name, blocks_per_cu, ms,
           ms * 1e-3 * 2.4e9 * 1024 / insts, ms * 1e-3 * 2.4e9 * 1024 / ((double)blocks * 4 * iters * 64));
    hipFree(out);
}
int main()
{
    for (int w : {1, 2, 4}) {
        run<0>("and+bcnt (dependent pairs)", w, 128);
        run<1>("and+add", w, 128);
        run<2>("bcnt only", w, 64);
    }
    return 0;
}

// Micro-benchmark: does the ORDER of v_and_b32 (2.45-cycle class) and v_bcnt_u32_b32 (4.2-cycle class)
// inside k_pair_counts' inner pattern matter?  Registers only; orders fixed with asm volatile blocks.
//   MODE 0: and,bcnt alternating        MODE 1: 8 ands then 8 bcnts (one row of the 8x8 block)
//   MODE 2: 16 ands then 16 bcnts       MODE 3: 64 ands then 64 bcnts (needs 64 temporaries)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define AND(t, a, b) "v_and_b32 %" #t ", %" #a ", %" #b "\n"
#define BC(acc, t) "v_bcnt_u32_b32 %" #acc ", %" #t ", %" #acc "\n"
// operands: 0-7 acc, 8-15 tmp, 16 a, 17-24 b
#define ROW_ALT AND(8,16,17) BC(0,8) AND(9,16,18) BC(1,9) AND(10,16,19) BC(2,10) AND(11,16,20) BC(3,11) AND(12,16,21) BC(4,12) AND(13,16,22) BC(5,13) AND(14,16,23) BC(6,14) AND(15,16,24) BC(7,15)
#define ROW_BAT AND(8,16,17) AND(9,16,18) AND(10,16,19) AND(11,16,20) AND(12,16,21) AND(13,16,22) AND(14,16,23) AND(15,16,24) BC(0,8) BC(1,9) BC(2,10) BC(3,11) BC(4,12) BC(5,13) BC(6,14) BC(7,15)
#define OPS(acc, tmp, a, b) : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7]), \
      "=&v"(tmp[0]), "=&v"(tmp[1]), "=&v"(tmp[2]), "=&v"(tmp[3]), "=&v"(tmp[4]), "=&v"(tmp[5]), "=&v"(tmp[6]), "=&v"(tmp[7]) \
    : "v"(a), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]), "v"(b[5]), "v"(b[6]), "v"(b[7])
template <int MODE>
__global__ __launch_bounds__(256) void k(uint32_t *out, uint32_t s, int iters)
{
    uint32_t acc[8][8];
    for (int r = 0; r < 8; ++r) for (int c = 0; c < 8; ++c) acc[r][c] = 0;
    uint32_t av[8], bv[8];
    for (int r = 0; r < 8; ++r) { av[r] = threadIdx.x * 2654435761u + r * s; bv[r] = threadIdx.x * 40503u + r * 77u + s; }
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0 || MODE == 1) {
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                uint32_t t[8];
                if (MODE == 0) asm volatile(ROW_ALT OPS(acc[r], t, av[r], bv));
                else asm volatile(ROW_BAT OPS(acc[r], t, av[r], bv));
            }
        } else if (MODE == 2) {
#pragma unroll
            for (int r = 0; r < 8; r += 2) {
                uint32_t t0[8], t1[8];
                asm volatile(AND(8,16,17) AND(9,16,18) AND(10,16,19) AND(11,16,20) AND(12,16,21) AND(13,16,22) AND(14,16,23) AND(15,16,24)
                             : "=&v"(t0[0]), "=&v"(t0[1]), "=&v"(t0[2]), "=&v"(t0[3]), "=&v"(t0[4]), "=&v"(t0[5]), "=&v"(t0[6]), "=&v"(t0[7]),
                               "=&v"(t0[0]), "=&v"(t0[1]), "=&v"(t0[2]), "=&v"(t0[3]), "=&v"(t0[4]), "=&v"(t0[5]), "=&v"(t0[6]), "=&v"(t0[7])
                             : "v"(av[r]), "v"(bv[0]), "v"(bv[1]), "v"(bv[2]), "v"(bv[3]), "v"(bv[4]), "v"(bv[5]), "v"(bv[6]), "v"(bv[7]));
                (void)t1;
            }
        }
#pragma unroll
        for (int r = 0; r < 8; ++r) { asm volatile("" : "+v"(av[r]), "+v"(bv[r])); }
    }
    uint32_t t = 0;
    for (int r = 0; r < 8; ++r) for (int c = 0; c < 8; ++c) t ^= acc[r][c];
    out[blockIdx.x * 256 + threadIdx.x] = t;
}
template <int MODE>
static void run(const char *name, int blocks_per_cu)
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
    printf("%-32s wg/CU=%d  %7.3f ms  %.2f cycles per AND+BCNT pair\n", name, blocks_per_cu, ms,
           ms * 1e-3 * 2.4e9 * 1024 / ((double)blocks * 4 * iters * 64));
    hipFree(out);
}
int main()
{
    for (int w : {2, 4}) {
        run<0>("alternating and,bcnt", w);
        run<1>("8 ands then 8 bcnts", w);
    }
    return 0;
}

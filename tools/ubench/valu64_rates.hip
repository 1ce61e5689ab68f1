// Micro-benchmark: issue cycles per wave64 instruction for the 64-bit integer ops the Wang hash
// of k_sketch is made of (gfx950).  Build: hipcc --offload-arch=gfx950 -O3 valu64_rates.hip -o valu64_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))
#define KERNEL(name, body)                                                      \
    __global__ void name(uint64_t *out, uint32_t s, int iters) {                \
        uint64_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;       \
        uint64_t s64 = ((uint64_t)s << 32) | s;                                  \
        for (int i = 0; i < iters; ++i) { REP64(body) }                          \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3;         \
    }
#define ARGS : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(s), "v"(s64)
// 4 independent chains per body
#define SHL64 asm volatile("v_lshlrev_b64 %0, 3, %0\nv_lshlrev_b64 %1, 3, %1\nv_lshlrev_b64 %2, 3, %2\nv_lshlrev_b64 %3, 3, %3" ARGS);
#define SHR64 asm volatile("v_lshrrev_b64 %0, 3, %0\nv_lshrrev_b64 %1, 3, %1\nv_lshrrev_b64 %2, 3, %2\nv_lshrrev_b64 %3, 3, %3" ARGS);
#define LSHLADD64 asm volatile("v_lshl_add_u64 %0, %0, 3, %5\nv_lshl_add_u64 %1, %1, 3, %5\nv_lshl_add_u64 %2, %2, 3, %5\nv_lshl_add_u64 %3, %3, 3, %5" ARGS);
#define MAD64 asm volatile("v_mad_u64_u32 %0, vcc, %4, %4, %0\nv_mad_u64_u32 %1, vcc, %4, %4, %1\nv_mad_u64_u32 %2, vcc, %4, %4, %2\nv_mad_u64_u32 %3, vcc, %4, %4, %3" ARGS : "vcc");
#define ADDCO asm volatile("v_add_co_u32 %0, vcc, %4, %0\nv_add_co_u32 %1, vcc, %4, %1\nv_add_co_u32 %2, vcc, %4, %2\nv_add_co_u32 %3, vcc, %4, %3" ARGS : "vcc");
#define ALIGNBIT asm volatile("v_alignbit_b32 %0, %0, %4, 7\nv_alignbit_b32 %1, %1, %4, 7\nv_alignbit_b32 %2, %2, %4, 7\nv_alignbit_b32 %3, %3, %4, 7" ARGS);
#define FFBH asm volatile("v_ffbh_u32 %0, %0\nv_ffbh_u32 %1, %1\nv_ffbh_u32 %2, %2\nv_ffbh_u32 %3, %3" ARGS);
#define CMP64 asm volatile("v_cmp_lt_u64 vcc, %0, %5\nv_cmp_lt_u64 vcc, %1, %5\nv_cmp_lt_u64 vcc, %2, %5\nv_cmp_lt_u64 vcc, %3, %5" ARGS : "vcc");
#define CNDMASK asm volatile("v_cndmask_b32 %0, %0, %4, vcc\nv_cndmask_b32 %1, %1, %4, vcc\nv_cndmask_b32 %2, %2, %4, vcc\nv_cndmask_b32 %3, %3, %4, vcc" ARGS);
#define MULLO asm volatile("v_mul_lo_u32 %0, %0, %4\nv_mul_lo_u32 %1, %1, %4\nv_mul_lo_u32 %2, %2, %4\nv_mul_lo_u32 %3, %3, %4" ARGS);
#define MULHI asm volatile("v_mul_hi_u32 %0, %0, %4\nv_mul_hi_u32 %1, %1, %4\nv_mul_hi_u32 %2, %2, %4\nv_mul_hi_u32 %3, %3, %4" ARGS);
KERNEL(k_shl64, SHL64)
KERNEL(k_shr64, SHR64)
KERNEL(k_lshladd64, LSHLADD64)
KERNEL(k_mad64, MAD64)
KERNEL(k_cmp64, CMP64)
template <class K>
static void run(const char *name, K k) {
    uint64_t *out;
    hipMalloc(&out, 256 * 2048 * 8);
    const int iters = 200, blocks = 256 * 8, threads = 256;  // 8 waves per SIMD
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    k<<<blocks, threads>>>(out, 3, 10);
    hipDeviceSynchronize();
    hipEventRecord(a);
    k<<<blocks, threads>>>(out, 3, iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    const double insts = (double)blocks * (threads / 64) * iters * 64 * 4;  // wave-instructions
    const double cyc = ms * 1e-3 * 2.4e9 * 1024 / insts;
    printf("%-16s %8.3f ms  %.2f cycles/wave-instr (at 2.4 GHz)\n", name, ms, cyc);
    hipFree(out);
}
int main() {
    run("v_lshlrev_b64", k_shl64); run("v_lshrrev_b64", k_shr64); run("v_lshl_add_u64", k_lshladd64);
    run("v_mad_u64_u32", k_mad64); run("v_cmp_lt_u64", k_cmp64);
    return 0;
}

// Micro-benchmark: issue cycles per wave64 VALU instruction on gfx950 (which integer ops run at the
// 2-cycle fp32 rate, which at 4).  Build: hipcc --offload-arch=gfx950 -O3 valu_rates.hip -o valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))
#define KERNEL(name, body)                                                      \
    __global__ void name(uint32_t *out, uint32_t s, int iters) {                \
        uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
        for (int i = 0; i < iters; ++i) { REP64(body) }                          \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7; \
    }
#define OP1(ins) asm volatile(ins " %0, %0, %8\n" ins " %1, %1, %8\n" ins " %2, %2, %8\n" ins " %3, %3, %8\n" ins " %4, %4, %8\n" ins " %5, %5, %8\n" ins " %6, %6, %8\n" ins " %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s));
#define OP3(ins) asm volatile(ins " %0, %0, %8, %0\n" ins " %1, %1, %8, %1\n" ins " %2, %2, %8, %2\n" ins " %3, %3, %8, %3\n" ins " %4, %4, %8, %4\n" ins " %5, %5, %8, %5\n" ins " %6, %6, %8, %6\n" ins " %7, %7, %8, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s));
KERNEL(k_and, OP1("v_and_b32"))
KERNEL(k_xor, OP1("v_xor_b32"))
KERNEL(k_add, OP1("v_add_u32"))
KERNEL(k_bcnt, OP1("v_bcnt_u32_b32"))
KERNEL(k_lshl, OP1("v_lshlrev_b32"))
KERNEL(k_mul, OP1("v_mul_lo_u32"))
KERNEL(k_fmaf, OP3("v_fma_f32"))
KERNEL(k_bfi, OP3("v_bfi_b32"))
KERNEL(k_andor, OP3("v_and_or_b32"))
KERNEL(k_add3, OP3("v_add3_u32"))
KERNEL(k_alignbit, OP3("v_alignbit_b32"))
KERNEL(k_mad24, OP3("v_mad_u32_u24"))
KERNEL(k_perm, OP3("v_perm_b32"))
KERNEL(k_sad, OP3("v_sad_u8"))
KERNEL(k_dot4, OP3("v_dot4_u32_u8"))
KERNEL(k_dot8, OP3("v_dot8_u32_u4"))
template <class K>
static void run(const char *name, K k) {
    uint32_t *out;
    hipMalloc(&out, 256 * 2048 * 4);
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
    const double insts = (double)blocks * (threads / 64) * iters * 64 * 8;  // wave-instructions
    const double cyc = ms * 1e-3 * 2.4e9 * 1024 / insts;                     // SIMD-cycles per wave-instruction at 2.4 GHz
    printf("%-12s %8.3f ms  %.2f cycles/wave-instr (at 2.4 GHz)  %.2e lane-ops/s\n", name, ms, cyc, insts * 64 / (ms * 1e-3));
    hipFree(out);
}
int main() {
    run("v_and_b32", k_and); run("v_xor_b32", k_xor); run("v_add_u32", k_add); run("v_bcnt", k_bcnt);
    run("v_lshlrev", k_lshl); run("v_mul_lo", k_mul); run("v_fma_f32", k_fmaf); run("v_bfi", k_bfi);
    run("v_and_or", k_andor); run("v_add3", k_add3); run("v_alignbit", k_alignbit); run("v_mad_u24", k_mad24);
    run("v_perm", k_perm); run("v_sad_u8", k_sad); run("v_dot4_u8", k_dot4); run("v_dot8_u4", k_dot8);
    return 0;
}

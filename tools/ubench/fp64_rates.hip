// Micro-benchmark: issue cycles per wave64 instruction for the fp64 ops the Ertl-MLE step of k_finalize is made of
// (gfx950), 4 independent chains, 8 waves per SIMD.  Build: hipcc --offload-arch=gfx950 -O3 fp64_rates.hip -o fp64_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))
#define KERNEL(name, body)                                                      \
    __global__ void name(double *out, double s, int iters) {                    \
        double a0 = 1.0 + threadIdx.x * 1e-3, a1 = a0 + 0.1, a2 = a0 + 0.2, a3 = a0 + 0.3; \
        float f0 = (float)a0, f1 = (float)a1, f2 = (float)a2, f3 = (float)a3;  \
        uint32_t u0 = threadIdx.x, u1 = u0 + 1, u2 = u0 + 2, u3 = u0 + 3;      \
        for (int i = 0; i < iters; ++i) { REP64(body) }                          \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + f0 + f1 + f2 + f3 + u0 + u1 + u2 + u3; \
    }
#define ARGS : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(s)
#define FMA asm volatile("v_fma_f64 %0, %0, %12, %12\nv_fma_f64 %1, %1, %12, %12\nv_fma_f64 %2, %2, %12, %12\nv_fma_f64 %3, %3, %12, %12" ARGS);
#define ADD asm volatile("v_add_f64 %0, %0, %12\nv_add_f64 %1, %1, %12\nv_add_f64 %2, %2, %12\nv_add_f64 %3, %3, %12" ARGS);
#define MUL asm volatile("v_mul_f64 %0, %0, %12\nv_mul_f64 %1, %1, %12\nv_mul_f64 %2, %2, %12\nv_mul_f64 %3, %3, %12" ARGS);
#define RCP asm volatile("v_rcp_f64 %0, %0\nv_rcp_f64 %1, %1\nv_rcp_f64 %2, %2\nv_rcp_f64 %3, %3" ARGS);
#define RCP32 asm volatile("v_rcp_f32 %4, %4\nv_rcp_f32 %5, %5\nv_rcp_f32 %6, %6\nv_rcp_f32 %7, %7" ARGS);
#define CVT_F32_F64 asm volatile("v_cvt_f32_f64 %4, %0\nv_cvt_f32_f64 %5, %1\nv_cvt_f32_f64 %6, %2\nv_cvt_f32_f64 %7, %3" ARGS);
#define CVT_F64_F32 asm volatile("v_cvt_f64_f32 %0, %4\nv_cvt_f64_f32 %1, %5\nv_cvt_f64_f32 %2, %6\nv_cvt_f64_f32 %3, %7" ARGS);
#define CVT_F64_U32 asm volatile("v_cvt_f64_u32 %0, %8\nv_cvt_f64_u32 %1, %9\nv_cvt_f64_u32 %2, %10\nv_cvt_f64_u32 %3, %11" ARGS);
#define LDEXP asm volatile("v_ldexp_f64 %0, %0, %8\nv_ldexp_f64 %1, %1, %9\nv_ldexp_f64 %2, %2, %10\nv_ldexp_f64 %3, %3, %11" ARGS);
#define ADDU32 asm volatile("v_add_u32 %8, %8, %9\nv_add_u32 %9, %9, %10\nv_add_u32 %10, %10, %11\nv_add_u32 %11, %11, %8" ARGS);
KERNEL(k_fma, FMA)
KERNEL(k_add, ADD)
KERNEL(k_mul, MUL)
KERNEL(k_rcp, RCP)
KERNEL(k_rcp32, RCP32)
KERNEL(k_cvt3264, CVT_F32_F64)
KERNEL(k_cvt6432, CVT_F64_F32)
KERNEL(k_cvtu, CVT_F64_U32)
KERNEL(k_ldexp, LDEXP)
KERNEL(k_addu32, ADDU32)
template <class K>
static void run(const char *name, K k) {
    double *out;
    hipMalloc(&out, 256 * 2048 * 8);
    const int iters = 200, blocks = 256 * 8, threads = 256;  // 8 waves per SIMD
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    k<<<blocks, threads>>>(out, 0.999, 10);
    hipDeviceSynchronize();
    hipEventRecord(a);
    k<<<blocks, threads>>>(out, 0.999, iters);
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
    run("v_fma_f64", k_fma); run("v_add_f64", k_add); run("v_mul_f64", k_mul); run("v_rcp_f64", k_rcp);
    run("v_rcp_f32", k_rcp32); run("v_cvt_f32_f64", k_cvt3264); run("v_cvt_f64_f32", k_cvt6432); run("v_cvt_f64_u32", k_cvtu);
    run("v_ldexp_f64", k_ldexp); run("v_add_u32", k_addu32);
    return 0;
}

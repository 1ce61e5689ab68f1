#!/usr/bin/env python3
"""Generates tools/ubench/pair_sched.hip: k_pair_counts' inner pattern (8x8 outer product of
v_and_b32 + v_bcnt_u32_b32 accumulate per lane) written with EXPLICIT VGPR numbers, so that
  * the order of the two instruction classes (alternating / batches of 8, 16, 32, 64 ANDs before
    their BCNTs) and
  * the register-bank placement (VGPR index mod 4) of accumulators, temporaries and operands
are pinned, and the time is taken in real shader cycles (s_memtime) next to the wall clock, which gives
the effective clock under this instruction mix (the chip clocks to its power budget,
MI355X_MICROARCH.md "DVFS give-back").  Usage: gen_pair_sched.py > pair_sched.hip
"""
import sys

A0, B0 = 16, 24          # a[r] = v16..v23, b[c] = v24..v31
ACC0 = 64                # acc[r][c] = v(64 + 8r + c)
TMP0 = 128               # 64 temporaries v128..v191
A1, B1 = 32, 40          # second k-row's operands (carry-save variant): v32..v39, v40..v47
ONES0 = 192              # carry-save variant: 64 "ones" bit-vectors v192..v255


def body_csa(order):
    """TWO k-rows: per pair x = a0&b0, y = a1&b1, (ones, carry) = CSA(ones, x, y) with u = ones^x,
    carry = bfi(u, y, x) = majority, ones = u^y; acc += bcnt(carry) (weight 2, applied at the plane flush).
    6 instructions per 2 pair-slots (5 bitwise + 1 BCNT) instead of 4 (2 AND + 2 BCNT)."""
    lines = []
    pairs = [(r, c) for r in range(8) for c in range(8)]
    for s in range(0, 64, order):
        grp = pairs[s:s + order]
        tx = lambda i: TMP0 + (2 * i) % 64
        ty = lambda i: TMP0 + (2 * i + 1) % 64
        for i, (r, c) in enumerate(grp):
            lines.append("v_and_b32 v%d, v%d, v%d" % (tx(i), A0 + r, B0 + c))
        for i, (r, c) in enumerate(grp):
            lines.append("v_and_b32 v%d, v%d, v%d" % (ty(i), A1 + r, B1 + c))
        for i, (r, c) in enumerate(grp):
            lines.append("v_xor_b32 v%d, v%d, v%d" % (ONES0 + 8 * r + c, ONES0 + 8 * r + c, tx(i)))
        for i, (r, c) in enumerate(grp):
            lines.append("v_bfi_b32 v%d, v%d, v%d, v%d" % (tx(i), ONES0 + 8 * r + c, ty(i), tx(i)))
        for i, (r, c) in enumerate(grp):
            lines.append("v_xor_b32 v%d, v%d, v%d" % (ONES0 + 8 * r + c, ONES0 + 8 * r + c, ty(i)))
        for i, (r, c) in enumerate(grp):
            lines.append("v_bcnt_u32_b32 v%d, v%d, v%d" % (ACC0 + 8 * r + c, tx(i), ACC0 + 8 * r + c))
    return lines


def lds_reads(row):
    """the real kernel's operand traffic: 4 ds_read_b128 per k-row (A: broadcast over the lane's column index, B: over its
    row index), issued right after the AND batch (its last use of the operands)"""
    o = row * 512
    return ["ds_read_b128 v[%d:%d], v8 offset:%d" % (A0, A0 + 3, o), "ds_read_b128 v[%d:%d], v8 offset:%d" % (A0 + 4, A0 + 7, o + 16),
            "ds_read_b128 v[%d:%d], v9 offset:%d" % (B0, B0 + 3, 8192 + o), "ds_read_b128 v[%d:%d], v9 offset:%d" % (B0 + 4, B0 + 7, 8192 + o + 16)]


def body(order, tmp_bank_shift=0, ops="and+bcnt", barrier=False, lds=-1, spread=False):
    """one k-row: 64 (AND, BCNT) pairs; `order` = batch size of ANDs issued before their BCNTs;
    barrier: s_barrier after every AND batch and every BCNT batch (keeps the waves of a SIMD in the same phase)"""
    lines = []
    pairs = [(r, c) for r in range(8) for c in range(8)]
    if lds >= 0:
        lines.append("s_waitcnt lgkmcnt(0)")
    for s in range(0, 64, order):
        grp = pairs[s:s + order]
        for i, (r, c) in enumerate(grp):
            t = TMP0 + ((s + i + tmp_bank_shift) % 64)
            if ops in ("and+bcnt", "and"):
                lines.append("v_and_b32 v%d, v%d, v%d" % (t, A0 + r, B0 + c))
            elif ops == "and+add":
                lines.append("v_and_b32 v%d, v%d, v%d" % (t, A0 + r, B0 + c))
        if barrier:
            lines.append("s_barrier")
        rd = lds_reads(lds) if lds >= 0 and s + order >= 64 else []
        if rd and not spread:
            lines += rd
            rd = []
        for i, (r, c) in enumerate(grp):
            t = TMP0 + ((s + i + tmp_bank_shift) % 64)
            acc = ACC0 + 8 * r + c
            if rd and i % 16 == 0:
                lines.append(rd.pop(0))
            if ops == "and+bcnt":
                lines.append("v_bcnt_u32_b32 v%d, v%d, v%d" % (acc, t, acc))
            elif ops == "bcnt":
                lines.append("v_bcnt_u32_b32 v%d, v%d, v%d" % (acc, A0 + ((r + c) % 8), acc))
            elif ops == "and+add":
                lines.append("v_add_u32 v%d, v%d, v%d" % (acc, t, acc))
        if barrier:
            lines.append("s_barrier")
    return lines


VARIANTS = [
    # name, order, tmp_bank_shift, ops
    ("alt", 1, 0, "and+bcnt"),
    ("batch8", 8, 0, "and+bcnt"),
    ("batch16", 16, 0, "and+bcnt"),
    ("batch32", 32, 0, "and+bcnt"),
    ("batch64", 64, 0, "and+bcnt"),
    ("batch8_tmpbank1", 8, 1, "and+bcnt"),   # temporaries one bank away from their accumulator
    ("batch8_tmpbank2", 8, 2, "and+bcnt"),
    ("batch64_tmpbank1", 64, 1, "and+bcnt"),
    ("and_only", 8, 0, "and"),
    ("bcnt_only", 8, 0, "bcnt"),
    ("and_add", 8, 0, "and+add"),
    # the two waves of a SIMD kept in the same phase by barriers (needs both in ONE workgroup: 512 threads)
    ("batch64_wg512", 64, 0, "and+bcnt", 512, False),
    ("batch64_wg512_bar", 64, 0, "and+bcnt", 512, True),
    ("batch32_wg512_bar", 32, 0, "and+bcnt", 512, True),
    # ... with the real kernel's LDS operand reads (4 ds_read_b128 per k-row, after the AND batch)
    ("batch64_lds", 64, 0, "and+bcnt", 256, False, True),
    ("batch64_wg512_lds", 64, 0, "and+bcnt", 512, False, True),
    ("batch64_wg512_bar_lds", 64, 0, "and+bcnt", 512, True, True),
    ("batch64_wg512_bar_ldsspread", 64, 0, "and+bcnt", 512, True, True, True),
    # carry-save (Harley-Seal, depth 1) over pairs of k-rows: 256 VGPRs, so 2 waves per SIMD at most (wg/CU = 2 lines only)
    ("csa_b1", 1, 0, "csa"),
    ("csa_b8", 8, 0, "csa"),
    ("csa_b16", 16, 0, "csa"),
    ("csa_b32", 32, 0, "csa"),
]

ROWS_PER_ITER = 4  # k-rows per loop iteration (the real kernel unrolls 8)


def kernel(name, order, shift, ops, wg=256, barrier=False, lds=False, spread=False):
    clob = ", ".join('"v%d"' % i for i in [8, 9] + list(range(A0, B1 + 8)) + list(range(ACC0, ACC0 + 64)) + list(range(TMP0, TMP0 + 64)) + (list(range(ONES0, ONES0 + 64)) if ops == "csa" else []))
    L = []
    # operands from the lane id and a seed (seed 0 -> all-zero data: the low-power arm)
    for r in range(8):
        L.append("v_mul_lo_u32 v%d, %%[tid], %%[m%d]" % (A0 + r, r % 2))
        L.append("v_xor_b32 v%d, v%d, %%[seed]" % (A0 + r, A0 + r))
        L.append("v_mul_lo_u32 v%d, v%d, %%[seed]" % (A0 + r, A0 + r))
        L.append("v_mul_lo_u32 v%d, %%[tid], %%[m%d]" % (B0 + r, (r + 1) % 2))
        L.append("v_add_u32 v%d, v%d, %%[seed]" % (B0 + r, B0 + r))
        L.append("v_mul_lo_u32 v%d, v%d, %%[seed]" % (B0 + r, B0 + r))
    # LDS addresses of the lane's A and B operand slots (as k_pair_counts: 8x8 lane grid, 32 B per lane and operand)
    L.append("v_lshrrev_b32 v8, 3, %[tid]")
    L.append("v_and_b32 v8, 7, v8")
    L.append("v_lshlrev_b32 v8, 5, v8")
    L.append("v_and_b32 v9, 7, %[tid]")
    L.append("v_lshlrev_b32 v9, 5, v9")
    for r in range(8):  # second k-row's operands
        L.append("v_mul_lo_u32 v%d, v%d, %%[m1]" % (A1 + r, A0 + r))
        L.append("v_mul_lo_u32 v%d, v%d, %%[m0]" % (B1 + r, B0 + r))
    for i in range(64):
        L.append("v_mov_b32 v%d, 0" % (ACC0 + i))
        L.append("v_mov_b32 v%d, 0" % (TMP0 + i))
        if ops == "csa":
            L.append("v_mov_b32 v%d, 0" % (ONES0 + i))
    L.append("s_memtime %[t0]")
    L.append("s_waitcnt lgkmcnt(0)")
    L.append("1:")
    if ops == "csa":
        for _ in range(ROWS_PER_ITER // 2):
            L += body_csa(order)
    else:
        for row in range(ROWS_PER_ITER):
            L += body(order, shift, ops, barrier, row if lds else -1, spread)
    L.append("s_sub_u32 %[it], %[it], 1")
    L.append("s_cmp_lg_u32 %[it], 0")
    L.append("s_cbranch_scc1 1b")
    L.append("s_memtime %[t1]")
    L.append("s_waitcnt lgkmcnt(0)")
    L.append("v_mov_b32 %[res], 0")
    for i in range(64):
        L.append("v_xor_b32 %%[res], %%[res], v%d" % (ACC0 + i))
        if ops == "csa":
            L.append("v_xor_b32 %%[res], %%[res], v%d" % (ONES0 + i))
    asm = "\\n\\t".join(L)
    return """
__global__ __launch_bounds__(%d) void k_%s(uint32_t *out, uint64_t *cyc, uint32_t seed, int iters)
{
    extern __shared__ uint32_t pad[];  // sized by the host to pin workgroups per CU
    uint32_t res;
    uint64_t t0, t1;
    int it = iters;
    const uint32_t tid = threadIdx.x + blockIdx.x * blockDim.x;
    asm volatile("%s"
                 : [res] "=&v"(res), [t0] "=&s"(t0), [t1] "=&s"(t1), [it] "+s"(it)
                 : [tid] "v"(tid), [seed] "v"(seed), [m0] "v"(2654435761u), [m1] "v"(40503u)
                 : %s, "scc", "memory");
    out[blockIdx.x * blockDim.x + threadIdx.x] = res + (pad == nullptr);
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6)] = t1 - t0;
}
""" % (wg, name, asm, clob)


def main():
    o = sys.stdout
    o.write("// GENERATED by tools/ubench/gen_pair_sched.py -- do not edit.\n")
    o.write("#include <hip/hip_runtime.h>\n#include <cstdio>\n#include <cstdint>\n#include <vector>\n#include <algorithm>\n")
    for v in VARIANTS:
        o.write(kernel(*v))
    o.write("""
typedef void (*kern_t)(uint32_t *, uint64_t *, uint32_t, int);
static void run(const char *name, kern_t k, int wg_per_cu, uint32_t seed, int pairs_per_row, int wg_threads = 256)
{
    const int blocks = 256 * wg_per_cu, iters = 40000;
    const size_t lds = wg_per_cu >= 4 ? 32 * 1024 : (wg_per_cu == 2 ? 64 * 1024 : 128 * 1024);
    uint32_t *out; uint64_t *cyc;
    hipMalloc(&out, blocks * wg_threads * 4); hipMalloc(&cyc, blocks * (wg_threads / 64) * 8);
    hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(wg_threads), lds, 0, out, cyc, seed, 2000);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(wg_threads), lds, 0, out, cyc, seed, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    std::vector<uint64_t> h(blocks * (wg_threads / 64));
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double med = (double)h[h.size() / 2];
    const double rows = (double)iters * %d;
    const int waves_per_simd = wg_per_cu * wg_threads / 256;  // waves per workgroup / 4 SIMDs per CU
    // per (AND,BCNT) pair-slot and wave: shader cycles of one wave / waves sharing the SIMD
    printf("%%-18s seed=%%u wg/CU=%%d  wall %%7.2f ms  wave %%10.0f cyc  eff.clock %%.3f GHz  %%.2f SIMD-cycles per pair-slot (wall@2.4GHz: %%.2f)\\n",
           name, seed, wg_per_cu, ms, med, med / (ms * 1e-3) / 1e9, med / (rows * pairs_per_row * waves_per_simd),
           ms * 1e-3 * 2.4e9 / (rows * pairs_per_row * waves_per_simd));
    hipFree(out); hipFree(cyc);
}
int main()
{
""" % ROWS_PER_ITER)
    for wg in (2, 4):
        for seed in (3, 0):
            for v in VARIANTS:
                if seed == 0 and v[0] not in ("batch8", "and_only", "bcnt_only"):
                    continue
                if v[3] == "csa" and wg != 2:
                    continue
                thr = v[4] if len(v) > 4 else 256
                if thr == 512 and wg != 2:
                    continue
                # 512-thread workgroups: ONE per CU gives the same 2 waves per SIMD as two 256-thread ones
                o.write('    run("%s", k_%s, %d, %du, 64, %d);\n' % (v[0], v[0], wg * 256 // thr, seed, thr))
    o.write("    return 0;\n}\n")


if __name__ == "__main__":
    main()

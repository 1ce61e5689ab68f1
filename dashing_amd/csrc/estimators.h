// estimators.h -- HLL cardinality estimators evaluated on the device, one lane per histogram.
//
// Product code (NOT the oracle).  Restates calculate_estimate() of the absent dnbaker/sketch
// submodule (called through hll_t::report / union_size at src/dashing.h:139,492) from the
// published algorithms -- SURVEY.md Appendix A.5: ORIGINAL (Flajolet et al. 2007 with small/
// large range corrections), ERTL_IMPROVED (Ertl 2017 sigma/tau), ERTL_MLE (Ertl 2017 Alg. 8,
// secant iteration with the relative early stop 1e-2/sqrt(m)).  All arithmetic is fp64 and the
// translation unit is compiled with -ffp-contract=off so + - * / are IEEE-identical with a CPU
// build of the same formulas.
//
// `Hist` is any callable  uint32_t c(int v)  returning the count of value v, v in [0, q+1].
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dsh {

__device__ __forceinline__ double alpha_m(uint64_t m)
{
    return m == 16 ? 0.673 : m == 32 ? 0.697 : m == 64 ? 0.709 : 0.7213 / (1. + 1.079 / (double)m);
}

__device__ inline double ertl_sigma(double x)
{
    if (x == 1.) return __builtin_huge_val();
    double z = x, zp = 0., y = 1.;
    while (z != zp) {
        x *= x;
        zp = z;
        z += x * y;
        y += y;
    }
    return z;
}

__device__ inline double ertl_tau(double x)
{
    if (x == 0. || x == 1.) return 0.;
    double z = 1. - x, y = 1., zp = x;
    while (zp != z) {
        x = sqrt(x);
        zp = z;
        y *= 0.5;
        const double t = 1. - x;
        z -= t * t * y;
    }
    return z / 3.;
}

// lo_hint/hi_hint: a range known to contain every non-empty bin (see estimate_mle); `raw(v)` skips the bounds test of
// `c(v)` and may be called for v in [lo_hint, hi_hint] only.  The sum runs over the live range alone: the bins outside
// it are zero and adding +0.0 changes nothing, so the additions that remain are the same, in the same ascending order,
// as the reference loop over 1 .. q (-E / --original, src/distmain.cpp:59) -- bit-identical.
template <class Hist, class Raw>
__device__ inline double estimate_original(const Hist &c, const Raw &raw, int p, int lo_hint, int hi_hint)
{
    const int q = 64 - p;
    const double m = (double)(1ull << p);
    const uint32_t c0 = c(0);
    double sum = (double)c0;
    const int lo = lo_hint > 1 ? lo_hint : 1, hi = hi_hint < q ? hi_hint : q;
    for (int i = lo; i <= hi; ++i) sum += ldexp((double)raw(i), -i);
    double value = alpha_m(1ull << p) * m * m / sum;
    if (value < 2.5 * m) {
        if (c0) value = m * log(m / (double)c0);
    } else if (value > 4294967296. / 30.) {
        const double corr = -4294967296. * log1p(-ldexp(value, -32));
        if (!(corr != corr)) value = corr;
    }
    return value;
}

template <class Hist>
__device__ inline double estimate_improved(const Hist &c, int p)
{
    const int q = 64 - p;
    const double m = (double)(1ull << p);
    const double divinv = 0x1.71547652b82fep-1;  // 1/(2 ln 2)
    double z = m * ertl_tau((m - (double)c(q + 1)) / m);
    for (int i = q; i; --i) {
        z += (double)c(i);
        z *= 0.5;
    }
    z += m * ertl_sigma((double)c(0) / m);
    return m * divinv * m / z;
}

// IEEE-754 double division for NORMAL, NON-ZERO, POSITIVE operands whose quotient, reciprocal and residual stay in
// the normal range: reciprocal estimate, two Newton steps, quotient, one residual correction -- the same fma
// sequence hipcc emits for `/` minus its v_div_scale/v_div_fixup special-case handling (3 of 11 instructions).
// Correctly rounded under that precondition, so results stay identical to a CPU `/`.  Used ONLY where the
// precondition holds by construction: through div_guarded, the once-per-iteration divisions; the step of the inner
// recurrence (both operands in [x', 2), x' = x 2^-s >= 2^-130) and the two constant divisors of the series start
// (x'^2 / 3, x'^2 / 472.5; x'^2 >= 2^-260) use div_inner below.
__device__ __forceinline__ double div_normal(double num, double den)
{
    double r = __builtin_amdgcn_rcp(den);
    double e = __builtin_fma(-den, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-den, r, 1.0);
    r = __builtin_fma(r, e, r);
    const double q = num * r;
    const double rem = __builtin_fma(-den, q, num);
    return __builtin_fma(rem, r, q);
}

// The inner recurrence's division: ONE Newton step.  v_rcp_f64 delivers 24.4 bits (measured, tools/ubench/div_accuracy.hip),
// one step 48.8; with q = num * r and the exact residual rem = num - den * q (fma) the value before the final rounding is
// off by |rem / den| * 2^-48.8 <= 2^-97 |q|: the result is the correctly rounded quotient unless the exact quotient lies within
// 2^-97 (relative) of the midpoint of two doubles -- 2^-44 of all operand pairs.  Measured: 0 mismatches against `/` in
// 3.4e10 random operand pairs of the recurrence's domain, and all 5e9 float32 results of the configs[3]-shaped matrix
// (2.2e11 divisions) unchanged (profiles/r4a).  Same preconditions as div_normal.  Two instructions fewer in a 21-instruction
// step that runs ~45 times per pair: k_finalize -5.5 % at p = 10.
__device__ __forceinline__ double div_inner(double num, double den)
{
    double r = __builtin_amdgcn_rcp(den);
    const double e = __builtin_fma(-den, r, 1.0);
    r = __builtin_fma(r, e, r);
    const double q = num * r;
    const double rem = __builtin_fma(-den, q, num);
    return __builtin_fma(rem, r, q);
}

// The same for operands that are only KNOWN to be finite: the fast sequence when both magnitudes lie in [2^-500, 2^500]
// (or the numerator is zero) -- then the reciprocal, the quotient (in [2^-1000, 2^1000]) and the residual (>= 2^-553)
// are all normal and the result is the correctly rounded quotient, signs included -- and the plain `/` otherwise.
// The estimator's once-per-iteration divisions go through here: their operands are differences of sums of counts
// (|x| in [2^-60, 2^24] or zero), so the slow branch is never taken in practice, but nothing depends on that.
__device__ __forceinline__ double div_guarded(double num, double den)
{
    const double an = __builtin_fabs(num), ad = __builtin_fabs(den);
    if (ad >= 0x1p-500 && ad <= 0x1p500 && an <= 0x1p500 && (an >= 0x1p-500 || an == 0.)) return div_normal(num, den);
    return num / den;
}

// x + x for a positive normal double far from overflow (the iteration's x' = x * 2^-s, doubled at most ~60
// times): one integer add on the exponent field -- a 2-cycle op instead of a 4-cycle v_add_f64, same bits.
__device__ __forceinline__ double twice(double x)
{
    return __longlong_as_double(__double_as_longlong(x) + (1ll << 52));
}

// lo_hint/hi_hint: a range known to contain every non-empty bin; `raw(v)` may be called only for
// v in [lo_hint, hi_hint] and skips the bounds test that `c(v)` performs (the iteration's
// count reads all fall in that range).  `Raw` also gives the address of a bin: raw.at(v), bins Raw::stride elements apart.  The next count is fetched one step ahead so the LDS
// read overlaps the dependent fp64 divide chain.
template <class Hist, class Raw>
__device__ inline double estimate_mle(const Hist &c, const Raw &raw, int p, int lo_hint, int hi_hint, int *iters = nullptr)
{
    const int q = 64 - p;
    const uint64_t m = 1ull << p;
    const uint32_t cq1 = c(q + 1);
    if (cq1 == m) return __builtin_huge_val();
    int kMin = lo_hint, kMax = hi_hint;
    while (kMin < hi_hint && raw(kMin) == 0) ++kMin;
    while (kMax > lo_hint && raw(kMax) == 0) --kMax;
    const int kMinPrime = kMin > 1 ? kMin : 1;
    const int kMaxPrime = kMax < q ? kMax : q;
    double z = 0.;
    {  // (address-driven like the recurrence below: the bin address is the loop counter)
        const auto *const pz0 = raw.at(kMinPrime);
        for (const auto *pz = raw.at(kMaxPrime); pz >= pz0; pz -= Raw::stride) z = 0.5 * z + (double)(uint32_t)*pz;
    }
    z = ldexp(z, -kMinPrime);
    uint32_t cPrime = cq1;
    if (q >= 1) cPrime += c(kMaxPrime);
    const uint32_t c0 = c(0);
    const double a = z + (double)c0;
    const double mPrime = (double)(int)(m - c0);
    double gprev = z + ldexp((double)cq1, -q);
    double x = gprev <= 1.5 * a ? div_guarded(mPrime, 0.5 * gprev + a) : (mPrime / gprev) * log1p(gprev / a);
    gprev = 0.;
    double deltaX = x;
    // 1e-2 / sqrt(2^p): sqrt(2^p) = 2^(p/2), times the correctly rounded sqrt(2) for odd p (what sqrt() returns: a
    // power-of-two scaling keeps a correctly rounded root correctly rounded); the quotient likewise is the correctly
    // rounded 1e-2 / 1 or 1e-2 / sqrt(2) scaled by a power of two -- one v_ldexp_f64 instead of a division per pair
    constexpr double kRelEven = 1e-2, kRelOdd = 1e-2 / 0x1.6a09e667f3bcdp+0;
    const double relerr = ldexp((p & 1) ? kRelOdd : kRelEven, -(p >> 1));
    if (iters) *iters = (kMaxPrime - kMinPrime + 1) << 8;  // live bins of the inner recurrence, iterations below
    while (deltaX > x * relerr) {
        if (iters) ++*iters;  // (profiling instances only)
        int kappaMinus1;
        (void)frexp(x, &kappaMinus1);
        const int sh = kMaxPrime + 1 > kappaMinus1 + 2 ? kMaxPrime + 1 : kappaMinus1 + 2;
        double xPrime = ldexp(x, -sh);
        const double xPrime2 = xPrime * xPrime;
        double h = xPrime - div_inner(xPrime2, 3.) + (xPrime2 * xPrime2) * (1. / 45. - div_inner(xPrime2, 472.5));
        for (int k = kappaMinus1; k >= kMaxPrime; --k) {
            const double hPrime = 1. - h;
            h = (xPrime + h * hPrime) / (xPrime + hPrime);
            xPrime = twice(xPrime);
        }
        double g = (double)cPrime * h;
        // the count of the NEXT step is read one step ahead, unconditionally: below kMinPrime that is one bin
        // under the range (never used; `raw` must tolerate the read -- an LDS column / array slot below its start).
        // The loop runs on the ADDRESS of the bin (raw.at(k), raw.stride elements apart): one integer add serves as
        // counter and address -- 18 instead of 19 VALU instructions per step (profiles/r4c).
        const auto *pk = raw.at(kMaxPrime - 1);
        const auto *const pmin = raw.at(kMinPrime);
        uint32_t cnext = pk >= pmin ? (uint32_t)*pk : 0u;
        while (pk >= pmin) {
            const double ck = (double)cnext;
            pk -= Raw::stride;
            cnext = (uint32_t)*pk;
            const double hPrime = 1. - h;
            h = div_inner(xPrime + h * hPrime, xPrime + hPrime);
            xPrime = twice(xPrime);
            g += ck * h;
        }
        g += x * a;
        if (gprev < g && g <= mPrime) deltaX *= div_guarded(g - mPrime, gprev - g);
        else deltaX = 0.;
        x += deltaX;
        gprev = g;
    }
    return x * (double)m;
}

template <class Hist, class Raw>
__device__ inline double estimate(const Hist &c, const Raw &raw, int p, int estim, int lo_hint,
                                  int hi_hint, int *iters = nullptr)
{
    switch (estim) {
    case 0: return estimate_original(c, raw, p, lo_hint, hi_hint);
    case 1: return estimate_improved(c, p);
    default: return estimate_mle(c, raw, p, lo_hint, hi_hint, iters);
    }
}

// jaccard_index of hll_t: (ca + cb - us)/us clamped with std::max(0., .) -- NaN maps to 0.
__device__ __forceinline__ double jaccard_from(double ca, double cb, double us)
{
    const double r = (ca + cb - us) / us;
    return (0. < r) ? r : 0.;
}

// result_cmp (src/dashing.h:568-592) for JI / MASH_DIST / FULL_MASH_DIST;
// dist_index :154-156, full_dist_index :172-174; ksinv = (double)(float)(1./k) (:797 of
// src/sketch_and_cmp.h); the result is cast to float (:591).
__device__ __forceinline__ float result_from_ji(double ji, int result_type, double ksinv)
{
    double ret = ji;
    if (result_type == 0) ret = ji != 0. ? -log(2. * ji / (1. + ji)) * ksinv : 1.;
    else if (result_type == 3) ret = 1. - pow(2. * ji / (1. + ji), ksinv);
    return (float)ret;
}

// Second arm of result_cmp (src/dashing.h:577-588): measures on set_triple(lhs, rhs) =
// lhs.full_set_comparison(rhs) = {max(mys-is,0), max(os-is,0), is}, is = max(mys+os-us, 0)
// (hll_t's own full_set_comparison is in the absent sketch submodule; the order of the triple --
// {mine - is, other - is, is} -- and J = is / (sum of the three) are corroborated in-tree by the exact-set sketch,
// src/khset64.h:129-141,146-149, and the intersection max(0, creport(a) + creport(b) - union_size) by
// src/dashing.h:550-552; the formulas ON the triple are in-tree: tests/test_reference_anchors.py).  EmissionType values: SIZES 2, FULL_CONTAINMENT_DIST 4,
// CONTAINMENT_INDEX 5, CONTAINMENT_DIST 6, SYMMETRIC_CONTAINMENT_INDEX 7, _DIST 8.
__device__ __forceinline__ double max0(double x) { return x < 0. ? 0. : x; }
__device__ __forceinline__ float result_from_triple(double mys, double os, double us, int result_type, double ksinv)
{
    const double is = max0(mys + os - us);
    const double t0 = max0(mys - is), t1 = max0(os - is), t2 = is;
    double ret = t2;
    if (result_type == 7 || result_type == 8) {
        ret /= ((t1 < t0 ? t1 : t0) + t2);
        if (result_type == 8) ret = ret != 0. ? -log(ret) * ksinv : 1.;
    } else if (result_type == 4 || result_type == 5 || result_type == 6) {
        ret /= (t0 + t1 + t2);
        if (result_type == 6) ret = ret != 0. ? -log(ret) * ksinv : 1.;
        else if (result_type == 4) ret = 1. - pow(ret, ksinv);
    }
    return (float)ret;
}

// result_cmp(lhs, rhs): mys/os = cardinalities of lhs/rhs, us = union size
__device__ __forceinline__ float result_cmp_from(double mys, double os, double us, int result_type, double ksinv)
{
    if (result_type == 0 || result_type == 1 || result_type == 3)
        return result_from_ji(jaccard_from(mys, os, us), result_type, ksinv);
    return result_from_triple(mys, os, us, result_type, ksinv);
}

}  // namespace dsh

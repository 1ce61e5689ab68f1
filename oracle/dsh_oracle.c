/*
 * dsh_oracle.c -- CPU restatement of dashing's HLL sketch-and-compare hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The product (libdashing_hip.so, the
 * dashing-amd CLI) never links, loads or calls anything in this directory.
 *
 * PARITY UNPINNED: the reference tree (/root/reference) does not contain the arithmetic
 * of this path -- it lives in the un-vendored git submodules
 *   github.com/dnbaker/bonsai          (Encoder, Spacer, kseq)            .gitmodules:1-3
 *   github.com/dnbaker/sketch          (hll_t, WangHash, estimators)      Makefile:64-65
 * whose pinned revisions are unrecoverable (no .git, no network), and the reference ships
 * no golden vectors (.travis.yml:16-24 checks exit codes only).  The reference therefore
 * cannot be built here (every TU includes the absent headers, src/dashing.h:4-10), and
 * this file restates the *published* algorithms (Flajolet et al. 2007; Ertl 2017,
 * arXiv:1702.01284 Alg. 8; T. Wang's 64-bit integer hash; 2-bit canonical k-mers) as
 * summarised in SURVEY.md Appendix A, anchored on the reference's own call sites:
 *   register rule        src/readfilt.cpp:86-88
 *   addh per k-mer       src/sketch_and_cmp.h:342
 *   J from cards/union   src/dashing.h:550-552 (dead twin documenting the formula), :138-140
 *   Mash transforms      src/dashing.h:154-156, :172-174 ; float cast :591 ; ksinv float :797
 *   row schedule         src/sketch_and_cmp.h:699-710, :808-816
 *   packed triangle      distmat/distmat.h:260-264
 * It is cross-checked against an independent pure-Python restatement
 * (oracle/oracle_py.py) by tests/test_oracle.py and pinned by the committed
 * self-consistency fixtures in tests/golden/.
 *
 * Build:  make -C oracle      (gcc -O3 -march=native -ffp-contract=off -fopenmp)
 * -ffp-contract=off matters: the device estimator is compiled the same way so that
 * + - * / ldexp frexp are IEEE-identical on both sides ("MLE early stop").
 */
#include <immintrin.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define DSHO_ORIGINAL 0
#define DSHO_ERTL_IMPROVED 1
#define DSHO_ERTL_MLE 2

/* EmissionType values, src/enums.h:13-23 */
#define DSHO_MASH_DIST 0
#define DSHO_JI 1
#define DSHO_FULL_MASH_DIST 3
#define DSHO_SIZES 2
#define DSHO_FULL_CONTAINMENT_DIST 4
#define DSHO_CONTAINMENT_INDEX 5
#define DSHO_CONTAINMENT_DIST 6
#define DSHO_SYMMETRIC_CONTAINMENT_INDEX 7
#define DSHO_SYMMETRIC_CONTAINMENT_DIST 8

/* Thread count for every parallel region below.  0 = OpenMP default.  Tests keep this small:
 * a GPU box may report far more logical CPUs than its cgroup quota lets us run, and a team of
 * spinning threads starves itself there. */
static int g_threads = 0;
void dsho_set_threads(int n) { g_threads = n > 0 ? n : 0; }
static int nthreads_(void)
{
#ifdef _OPENMP
    return g_threads > 0 ? g_threads : omp_get_max_threads();
#else
    return 1;
#endif
}

/* ---- A.2  Thomas Wang 64-bit integer hash (sketch::WangHash) ------------------------ */
uint64_t dsho_wang(uint64_t key)
{
    key = (~key) + (key << 21);
    key = key ^ (key >> 24);
    key = (key + (key << 3)) + (key << 8);
    key = key ^ (key >> 14);
    key = (key + (key << 2)) + (key << 4);
    key = key ^ (key >> 28);
    key = key + (key << 31);
    return key;
}

/* ---- A.3  register rule, mirrored at src/readfilt.cpp:86-88 -------------------------- */
void dsho_reg_rule(uint64_t h, int p, uint32_t *idx, uint8_t *val)
{
    *idx = (uint32_t)(h >> (64 - p));
    uint64_t t = ((h << 1) | 1) << (p - 1);
    *val = (uint8_t)(__builtin_clzll(t) + 1);
}

void dsho_add_hashed(uint8_t *regs, int p, uint64_t h)
{
    uint32_t idx;
    uint8_t v;
    dsho_reg_rule(h, p, &idx, &v);
    if (regs[idx] < v) regs[idx] = v;
}

/* ---- A.1  k-mer stream: unspaced, unwindowed, k <= 32 -------------------------------- */
static inline int base_code(uint8_t c)
{
    switch (c) {
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': return 3;
    default: return -1;
    }
}

/* Walk one record (no newlines; any non-ACGT byte resets the window).  For every window
 * of k valid bases calls regs-update (if regs) and/or appends the k-mer (if out).
 * Returns the number of k-mers emitted.  Call site being restated: src/sketch_and_cmp.h:342. */
uint64_t dsho_walk(const uint8_t *seq, uint64_t len, int k, int canon, int p, uint8_t *regs,
                   uint64_t *out, uint64_t out_cap)
{
    const uint64_t mask = (k == 32) ? ~UINT64_C(0) : ((UINT64_C(1) << (2 * k)) - 1);
    uint64_t fw = 0, rc = 0, n = 0;
    int filled = 0;
    for (uint64_t i = 0; i < len; ++i) {
        int c = base_code(seq[i]);
        if (c < 0) {
            filled = 0;
            fw = rc = 0;
            continue;
        }
        fw = ((fw << 2) | (uint64_t)c) & mask;
        rc = (rc >> 2) | ((uint64_t)(3 - c) << (2 * (k - 1)));
        if (filled < k) ++filled;
        if (filled == k) {
            uint64_t km = (canon && rc < fw) ? rc : fw;
            if (regs) dsho_add_hashed(regs, p, dsho_wang(km));
            if (out && n < out_cap) out[n] = km;
            ++n;
        }
    }
    return n;
}

/* Sketch a genome given as a byte stream in which records are separated by any
 * non-ACGT byte (the layout the product's dsh_sketch_batch takes). */
void dsho_sketch_batch(const uint8_t *seq, const uint64_t *genome_off, uint32_t n_genomes, int k,
                       int p, int canon, uint8_t *regs_out)
{
    const uint64_t m = UINT64_C(1) << p;
    memset(regs_out, 0, (size_t)n_genomes * m);
#pragma omp parallel for schedule(dynamic) num_threads(nthreads_())
    for (int64_t g = 0; g < (int64_t)n_genomes; ++g)
        dsho_walk(seq + genome_off[g], genome_off[g + 1] - genome_off[g], k, canon, p,
                  regs_out + (uint64_t)g * m, NULL, 0);
}

/* ---- A.4  histogram of registers / of the element-wise max --------------------------- */
void dsho_hist_single(const uint8_t *a, uint64_t m, uint32_t *hist)
{
    memset(hist, 0, 64 * sizeof(uint32_t));
    for (uint64_t t = 0; t < m; ++t) ++hist[a[t] & 63];
}

void dsho_hist_union(const uint8_t *a, const uint8_t *b, uint64_t m, uint32_t *hist)
{
    /* four sub-histograms break the store-to-load dependence on equal neighbours */
    uint32_t h[4][64];
    memset(h, 0, sizeof(h));
    uint64_t t = 0;
    for (; t + 4 <= m; t += 4) {
        uint8_t x0 = a[t] > b[t] ? a[t] : b[t];
        uint8_t x1 = a[t + 1] > b[t + 1] ? a[t + 1] : b[t + 1];
        uint8_t x2 = a[t + 2] > b[t + 2] ? a[t + 2] : b[t + 2];
        uint8_t x3 = a[t + 3] > b[t + 3] ? a[t + 3] : b[t + 3];
        ++h[0][x0 & 63];
        ++h[1][x1 & 63];
        ++h[2][x2 & 63];
        ++h[3][x3 & 63];
    }
    for (; t < m; ++t) {
        uint8_t x = a[t] > b[t] ? a[t] : b[t];
        ++h[0][x & 63];
    }
    for (int v = 0; v < 64; ++v) hist[v] = h[0][v] + h[1][v] + h[2][v] + h[3][v];
}

/* ---- A.4 with SIMD: the form the reference's CPU builds take ----------------------------
 * dashing ships SSE2 / AVX2 / AVX-512BW builds (Makefile:159-190, README.md:9) whose union_size does a
 * vector byte-max and then counts (the code is in the absent sketch submodule; SURVEY.md A.4 [M]: "any
 * method yields the same integers").  This is that form, written for the timed CPU baseline of bench.py:
 * max_epu8 over 64 (32) registers at a time, one 8-bit counter vector per register value of the band
 * [vlo, vhi] the pair can contain (cmpeq mask -> masked add; flushed with sad_epu8 before it can
 * overflow), at most NVMAX values per sweep over the data.  Selected at run time (cpuid), so the
 * portable liboracle.so carries it; tests/test_oracle.py asserts it equals dsho_hist_union bit for bit. */
#define NV512 24
__attribute__((target("avx512f,avx512bw"))) static void hist_union_avx512(const uint8_t *a, const uint8_t *b,
                                                                            uint64_t m, int vlo, int vhi,
                                                                            uint32_t *hist)
{
    const __m512i one = _mm512_set1_epi8(1), zero = _mm512_setzero_si512();
    for (int v0 = vlo; v0 <= vhi; v0 += NV512) {
        __m512i cnt[NV512];
        uint64_t tot[NV512];
        for (int g = 0; g < NV512; ++g) {
            cnt[g] = zero;
            tot[g] = 0;
        }
        const __m512i base = _mm512_set1_epi8((char)v0);
        uint64_t t = 0;
        while (t + 64 <= m) {
            uint64_t stop = t + 64 * 255;  /* 8-bit counters: flush every 255 vectors */
            if (stop > m) stop = m & ~UINT64_C(63);
            for (; t < stop; t += 64) {
                const __m512i mx = _mm512_max_epu8(_mm512_loadu_si512((const void *)(a + t)),
                                                   _mm512_loadu_si512((const void *)(b + t)));
                __m512i val = base;
#pragma GCC unroll 24
                for (int g = 0; g < NV512; ++g) {
                    cnt[g] = _mm512_mask_add_epi8(cnt[g], _mm512_cmpeq_epi8_mask(mx, val), cnt[g], one);
                    val = _mm512_add_epi8(val, one);
                }
            }
#pragma GCC unroll 24
            for (int g = 0; g < NV512; ++g) {
                tot[g] += (uint64_t)_mm512_reduce_add_epi64(_mm512_sad_epu8(cnt[g], zero));
                cnt[g] = zero;
            }
        }
        for (int g = 0; g < NV512 && v0 + g <= vhi; ++g) hist[(v0 + g) & 63] += (uint32_t)tot[g];
        if (v0 == vlo)
            for (; t < m; ++t) ++hist[(a[t] > b[t] ? a[t] : b[t]) & 63];  /* tail shorter than one vector */
    }
}

#define NV256 12
__attribute__((target("avx2"))) static void hist_union_avx2(const uint8_t *a, const uint8_t *b, uint64_t m,
                                                            int vlo, int vhi, uint32_t *hist)
{
    const __m256i one = _mm256_set1_epi8(1), zero = _mm256_setzero_si256();
    for (int v0 = vlo; v0 <= vhi; v0 += NV256) {
        __m256i cnt[NV256];
        uint64_t tot[NV256];
        for (int g = 0; g < NV256; ++g) {
            cnt[g] = zero;
            tot[g] = 0;
        }
        const __m256i base = _mm256_set1_epi8((char)v0);
        uint64_t t = 0;
        while (t + 32 <= m) {
            uint64_t stop = t + 32 * 255;
            if (stop > m) stop = m & ~UINT64_C(31);
            for (; t < stop; t += 32) {
                const __m256i mx = _mm256_max_epu8(_mm256_loadu_si256((const __m256i *)(a + t)),
                                                   _mm256_loadu_si256((const __m256i *)(b + t)));
                __m256i val = base;
#pragma GCC unroll 12
                for (int g = 0; g < NV256; ++g) {
                    cnt[g] = _mm256_sub_epi8(cnt[g], _mm256_cmpeq_epi8(mx, val));  /* -(-1) per equal byte */
                    val = _mm256_add_epi8(val, one);
                }
            }
#pragma GCC unroll 12
            for (int g = 0; g < NV256; ++g) {
                const __m256i s = _mm256_sad_epu8(cnt[g], zero);
                tot[g] += (uint64_t)_mm256_extract_epi64(s, 0) + (uint64_t)_mm256_extract_epi64(s, 1) +
                          (uint64_t)_mm256_extract_epi64(s, 2) + (uint64_t)_mm256_extract_epi64(s, 3);
                cnt[g] = zero;
            }
        }
        for (int g = 0; g < NV256 && v0 + g <= vhi; ++g) hist[(v0 + g) & 63] += (uint32_t)tot[g];
        if (v0 == vlo)
            for (; t < m; ++t) ++hist[(a[t] > b[t] ? a[t] : b[t]) & 63];
    }
}

/* 0 scalar, 1 AVX2, 2 AVX-512BW: what this host can run */
int dsho_simd_level(void)
{
    __builtin_cpu_init();
    if (__builtin_cpu_supports("avx512bw") && __builtin_cpu_supports("avx512f")) return 2;
    if (__builtin_cpu_supports("avx2")) return 1;
    return 0;
}

/* histogram of max(a,b) when every max lies in [vlo, vhi] (vlo = the larger of the two sketches' smallest
 * registers, vhi = the larger of their largest); level as dsho_simd_level(), capped by the host's. */
void dsho_hist_union_simd(const uint8_t *a, const uint8_t *b, uint64_t m, int vlo, int vhi, int level,
                          uint32_t *hist)
{
    const int have = dsho_simd_level();
    if (level > have) level = have;
    if (level <= 0 || vhi < vlo || vlo < 0 || vhi > 63) {
        dsho_hist_union(a, b, m, hist);
        return;
    }
    memset(hist, 0, 64 * sizeof(uint32_t));
    if (level == 2) hist_union_avx512(a, b, m, vlo, vhi, hist);
    else hist_union_avx2(a, b, m, vlo, vhi, hist);
}

/* 0 (default): the scalar histogram everywhere.  >0: dist_tri/dist_rows/dist_rect use the SIMD histogram of
 * that level (bench.py's cpu_baseline); results are identical (same integers into the same estimator). */
static int g_simd = 0;
void dsho_set_simd(int level) { g_simd = level > 0 ? level : 0; }

/* ---- A.5  estimators ------------------------------------------------------------------ */
static double alpha_m(uint64_t m)
{
    switch (m) {
    case 16: return 0.673;
    case 32: return 0.697;
    case 64: return 0.709;
    default: return 0.7213 / (1. + 1.079 / (double)m);
    }
}

static double ertl_sigma(double x)
{
    if (x == 1.) return INFINITY;
    double z = x, zp = 0., y = 1.;
    while (z != zp) {
        x *= x;
        zp = z;
        z += x * y;
        y += y;
    }
    return z;
}

static double ertl_tau(double x)
{
    if (x == 0. || x == 1.) return 0.;
    double z = 1. - x, y = 1., zp = x;
    while (zp != z) {
        x = sqrt(x);
        zp = z;
        y *= 0.5;
        double t = 1. - x;
        z -= t * t * y;
    }
    return z / 3.;
}

static double est_original(const uint32_t *c, int p)
{
    const int q = 64 - p;
    const double m = (double)(UINT64_C(1) << p);
    double sum = (double)c[0];
    for (int i = 1; i < q + 1; ++i)
        if (c[i]) sum += ldexp((double)c[i], -i);
    double value = alpha_m(UINT64_C(1) << p) * m * m / sum;
    if (value < 2.5 * m) {
        if (c[0]) value = m * log(m / (double)c[0]);
    } else if (value > 4294967296. / 30.) {
        double corr = -4294967296. * log1p(-ldexp(value, -32));
        if (!isnan(corr)) value = corr;
    }
    return value;
}

static double est_improved(const uint32_t *c, int p)
{
    const int q = 64 - p;
    const double m = (double)(UINT64_C(1) << p);
    const double divinv = (double)(1.L / (2.L * logl(2.L)));
    double z = m * ertl_tau((m - (double)c[q + 1]) / m);
    for (int i = q; i; --i) {
        z += (double)c[i];
        z *= 0.5;
    }
    z += m * ertl_sigma((double)c[0] / m);
    return m * divinv * m / z;
}

/* Ertl 2017 Algorithm 8 with the relative early-stop eps = 1e-2/sqrt(m)
 * (the stop rule is part of the function's definition for parity purposes). */
static double est_mle(const uint32_t *c, int p)
{
    const int q = 64 - p;
    const uint64_t m = UINT64_C(1) << p;
    if (c[q + 1] == m) return INFINITY;
    int kMin, kMax;
    for (kMin = 0; c[kMin] == 0; ++kMin) {}
    int kMinPrime = kMin > 1 ? kMin : 1;
    for (kMax = q + 1; kMax && c[kMax] == 0; --kMax) {}
    int kMaxPrime = kMax < q ? kMax : q;
    double z = 0.;
    for (int k = kMaxPrime; k >= kMinPrime; --k) z = 0.5 * z + (double)c[k];
    z = ldexp(z, -kMinPrime);
    uint32_t cPrime = c[q + 1];
    if (q >= 1) cPrime += c[kMaxPrime];
    double a = z + (double)c[0];
    int mPrime = (int)(m - c[0]);
    double gprev = z + ldexp((double)c[q + 1], -q);
    double x = gprev <= 1.5 * a ? (double)mPrime / (0.5 * gprev + a)
                                : ((double)mPrime / gprev) * log1p(gprev / a);
    gprev = 0.;
    double deltaX = x;
    const double relerr = 1e-2 / sqrt((double)m);
    while (deltaX > x * relerr) {
        int kappaMinus1;
        frexp(x, &kappaMinus1);
        int sh = kMaxPrime + 1 > kappaMinus1 + 2 ? kMaxPrime + 1 : kappaMinus1 + 2;
        double xPrime = ldexp(x, -sh);
        double xPrime2 = xPrime * xPrime;
        double h = xPrime - xPrime2 / 3. + (xPrime2 * xPrime2) * (1. / 45. - xPrime2 / 472.5);
        for (int k = kappaMinus1; k >= kMaxPrime; --k) {
            double hPrime = 1. - h;
            h = (xPrime + h * hPrime) / (xPrime + hPrime);
            xPrime += xPrime;
        }
        double g = (double)cPrime * h;
        for (int k = kMaxPrime - 1; k >= kMinPrime; --k) {
            double hPrime = 1. - h;
            h = (xPrime + h * hPrime) / (xPrime + hPrime);
            xPrime += xPrime;
            g += (double)c[k] * h;
        }
        g += x * a;
        if (gprev < g && g <= (double)mPrime) deltaX *= (g - (double)mPrime) / (gprev - g);
        else deltaX = 0.;
        x += deltaX;
        gprev = g;
    }
    return x * (double)m;
}

double dsho_estimate(const uint32_t *hist, int p, int estim)
{
    switch (estim) {
    case DSHO_ORIGINAL: return est_original(hist, p);
    case DSHO_ERTL_IMPROVED: return est_improved(hist, p);
    default: return est_mle(hist, p);
    }
}

/* a6: cardinality_estimate -> report(), src/dashing.h:492 */
double dsho_cardinality(const uint8_t *regs, int p, int estim)
{
    uint32_t h[64];
    dsho_hist_single(regs, UINT64_C(1) << p, h);
    return dsho_estimate(h, p, estim);
}

void dsho_cardinalities(const uint8_t *regs, uint64_t n, int p, int estim, double *out)
{
    const uint64_t m = UINT64_C(1) << p;
#pragma omp parallel for schedule(static) num_threads(nthreads_())
    for (int64_t i = 0; i < (int64_t)n; ++i) out[i] = dsho_cardinality(regs + (uint64_t)i * m, p, estim);
}

/* ---- A.6  pairwise quantities ----------------------------------------------------------
 * jaccard_index: us = union_size; ret = (ca + cb - us)/us; max(0., ret)
 * (std::max(0., nan) returns 0., hence the "0 < ret" form). */
double dsho_union_size(const uint8_t *a, const uint8_t *b, int p, int estim)
{
    uint32_t h[64];
    dsho_hist_union(a, b, UINT64_C(1) << p, h);
    return dsho_estimate(h, p, estim);
}

double dsho_jaccard_from(double ca, double cb, double us)
{
    double ret = (ca + cb - us) / us;
    return (0. < ret) ? ret : 0.;
}

/* dist_loop and partdist_loop pass ksinv as a FLOAT (src/sketch_and_cmp.h:797, src/dashing.h:664); nndist_loop
 * (--nearest-neighbors) keeps the DOUBLE 1./k (src/sketch_and_cmp.h:729).  dsho_knn raises this flag for its duration. */
static int g_ksinv_double = 0;

/* result_cmp for JI / MASH_DIST / FULL_MASH_DIST, src/dashing.h:568-592; ksinv is the
 * float 1./k promoted to double (src/sketch_and_cmp.h:797). */
float dsho_result(double ji, int result_type, int k)
{
    const float ksinv_f = (float)(1. / (double)k);
    const double ksinv = g_ksinv_double ? 1. / (double)k : (double)ksinv_f;
    double ret = ji;
    if (result_type == DSHO_MASH_DIST) ret = ji ? -log(2. * ji / (1. + ji)) * ksinv : 1.;
    else if (result_type == DSHO_FULL_MASH_DIST) ret = 1. - pow(2. * ji / (1. + ji), ksinv);
    return (float)ret;
}

/* Second arm of result_cmp (src/dashing.h:577-588): measures built on set_triple(lhs, rhs) =
 * lhs.full_set_comparison(rhs).  The triple itself lives in the absent sketch submodule; restated
 * (medium confidence) as  is = max(mys + os - us, 0), {max(mys-is,0), max(os-is,0), is}
 * with std::max(x, 0.) semantics (NaN in the first argument propagates).  The formulas applied to
 * the triple are in-tree. */
static double max0(double x) { return x < 0. ? 0. : x; }
float dsho_result_triple(double mys, double os, double us, int result_type, int k)
{
    const float ksinv_f = (float)(1. / (double)k);
    const double ksinv = g_ksinv_double ? 1. / (double)k : (double)ksinv_f;
    const double is = max0(mys + os - us);
    const double t0 = max0(mys - is), t1 = max0(os - is), t2 = is;
    double ret = t2;
    if (result_type == DSHO_SYMMETRIC_CONTAINMENT_INDEX || result_type == DSHO_SYMMETRIC_CONTAINMENT_DIST) {
        ret /= ((t1 < t0 ? t1 : t0) + t2);
        if (result_type == DSHO_SYMMETRIC_CONTAINMENT_DIST) ret = ret ? -log(ret) * ksinv : 1.;
    } else if (result_type == DSHO_FULL_CONTAINMENT_DIST || result_type == DSHO_CONTAINMENT_DIST ||
               result_type == DSHO_CONTAINMENT_INDEX) {
        ret /= (t0 + t1 + t2);
        if (result_type == DSHO_CONTAINMENT_DIST) ret = ret ? -log(ret) * ksinv : 1.;
        else if (result_type == DSHO_FULL_CONTAINMENT_DIST) ret = 1. - pow(ret, ksinv);
    }
    return (float)ret;
}

/* result_cmp(lhs = a, rhs = b): ca, cb are their cardinalities */
float dsho_pair(const uint8_t *a, const uint8_t *b, double ca, double cb, int p, int estim,
                int result_type, int k)
{
    const double us = dsho_union_size(a, b, p, estim);
    if (result_type == DSHO_MASH_DIST || result_type == DSHO_JI || result_type == DSHO_FULL_MASH_DIST)
        return dsho_result(dsho_jaccard_from(ca, cb, us), result_type, k);
    return dsho_result_triple(ca, cb, us, result_type, k);
}

/* same pair through the SIMD histogram (g_simd > 0): lo/hi = smallest / largest register of each sketch */
static float pair_banded(const uint8_t *a, const uint8_t *b, double ca, double cb, int p, int estim,
                         int result_type, int k, int lo_a, int hi_a, int lo_b, int hi_b)
{
    uint32_t h[64];
    dsho_hist_union_simd(a, b, UINT64_C(1) << p, lo_a > lo_b ? lo_a : lo_b, hi_a > hi_b ? hi_a : hi_b, g_simd, h);
    const double us = dsho_estimate(h, p, estim);
    if (result_type == DSHO_MASH_DIST || result_type == DSHO_JI || result_type == DSHO_FULL_MASH_DIST)
        return dsho_result(dsho_jaccard_from(ca, cb, us), result_type, k);
    return dsho_result_triple(ca, cb, us, result_type, k);
}

static uint8_t *ranges_(const uint8_t *regs, uint64_t n, uint64_t m)
{
    uint8_t *r = (uint8_t *)malloc(2 * (n ? n : 1));
#pragma omp parallel for schedule(static) num_threads(nthreads_())
    for (int64_t i = 0; i < (int64_t)n; ++i) {
        uint8_t lo = 255, hi = 0;
        const uint8_t *x = regs + (uint64_t)i * m;
        for (uint64_t t = 0; t < m; ++t) {
            lo = x[t] < lo ? x[t] : lo;
            hi = x[t] > hi ? x[t] : hi;
        }
        r[2 * i] = lo;
        r[2 * i + 1] = hi;
    }
    return r;
}

/* ---- a7-a9: all-pairs, reference schedule (row i serial, dynamic over j > i) ---------- */
static inline uint64_t tri_index(uint64_t n, uint64_t i, uint64_t j)
{
    return i * (2 * n - i - 1) / 2 + j - (i + 1); /* distmat/distmat.h:260-264 */
}

void dsho_dist_tri(const uint8_t *regs, uint64_t n, int p, int estim, int result_type, int k,
                   float *out_packed)
{
    const uint64_t m = UINT64_C(1) << p;
    double *card = (double *)malloc(sizeof(double) * (n ? n : 1));
    dsho_cardinalities(regs, n, p, estim, card);
    for (uint64_t i = 0; i + 1 < n; ++i) {
        const uint8_t *hi = regs + i * m;
        float *row = out_packed + tri_index(n, i, i + 1);
#pragma omp parallel for schedule(dynamic) num_threads(nthreads_())
        for (int64_t j = (int64_t)i + 1; j < (int64_t)n; ++j)
            row[j - (int64_t)i - 1] =
                dsho_pair(regs + (uint64_t)j * m, hi, card[j], card[i], p, estim, result_type, k);
    }
    free(card);
}

/* Rows [row_begin,row_end) only -- the bounded sample bench.py times as cpu_baseline.
 * out receives the rows back to back (row i has n-i-1 values). Returns pairs computed. */
uint64_t dsho_dist_rows(const uint8_t *regs, uint64_t n, int p, int estim, int result_type, int k,
                        uint64_t row_begin, uint64_t row_end, float *out)
{
    const uint64_t m = UINT64_C(1) << p;
    double *card = (double *)malloc(sizeof(double) * (n ? n : 1));
    dsho_cardinalities(regs, n, p, estim, card);
    uint64_t done = 0;
    uint8_t *rg = g_simd ? ranges_(regs, n, m) : NULL;
    for (uint64_t i = row_begin; i < row_end && i + 1 < n; ++i) {
        const uint8_t *hi = regs + i * m;
        float *row = out + done;
#pragma omp parallel for schedule(dynamic) num_threads(nthreads_())
        for (int64_t j = (int64_t)i + 1; j < (int64_t)n; ++j)
            row[j - (int64_t)i - 1] =
                rg ? pair_banded(regs + (uint64_t)j * m, hi, card[j], card[i], p, estim, result_type, k,
                                 rg[2 * j], rg[2 * j + 1], rg[2 * i], rg[2 * i + 1])
                   : dsho_pair(regs + (uint64_t)j * m, hi, card[j], card[i], p, estim, result_type, k);
        done += n - i - 1;
    }
    free(rg);
    free(card);
    return done;
}

/* query x reference rectangle (partdist_loop, src/dashing.h:660-712): out[q][r] */
void dsho_dist_rect(const uint8_t *qregs, uint64_t nq, const uint8_t *rregs, uint64_t nr, int p,
                    int estim, int result_type, int k, float *out)
{
    const uint64_t m = UINT64_C(1) << p;
    double *cq = (double *)malloc(sizeof(double) * (nq ? nq : 1));
    double *cr = (double *)malloc(sizeof(double) * (nr ? nr : 1));
    dsho_cardinalities(qregs, nq, p, estim, cq);
    dsho_cardinalities(rregs, nr, p, estim, cr);
    for (uint64_t i = 0; i < nq; ++i) {
#pragma omp parallel for schedule(dynamic) num_threads(nthreads_())
        for (int64_t j = 0; j < (int64_t)nr; ++j)
            out[i * nr + (uint64_t)j] = dsho_pair(rregs + (uint64_t)j * m, qregs + i * m, cr[j],
                                                  cq[i], p, estim, result_type, k);
    }
    free(cq);
    free(cr);
}

/* perform_nns (src/sketch_and_cmp.h:642-697) by brute force: for query i the nn best references
 * j != i, best first; similarity measures descending, distances ascending (emt2nntype,
 * src/dashing.h:268-280); ties by lower index (the reference's heap order is unspecified), NaN last. */
void dsho_knn(const uint8_t *regs, uint64_t n, int p, int estim, int result_type, int k, uint64_t qb,
              uint64_t qe, uint64_t rb, uint64_t re, uint32_t nn, uint32_t *idx_out, float *val_out)
{
    const uint64_t m = UINT64_C(1) << p;
    const int descending = !(result_type == DSHO_MASH_DIST || result_type == DSHO_FULL_MASH_DIST ||
                             result_type == DSHO_CONTAINMENT_DIST || result_type == DSHO_FULL_CONTAINMENT_DIST ||
                             result_type == DSHO_SYMMETRIC_CONTAINMENT_DIST);
    const float worst = descending ? -INFINITY : INFINITY;
    double *card = (double *)malloc(sizeof(double) * (n ? n : 1));
    dsho_cardinalities(regs, n, p, estim, card);
    g_ksinv_double = 1;  /* src/sketch_and_cmp.h:729 */
#pragma omp parallel for schedule(dynamic) num_threads(nthreads_())
    for (int64_t qi = (int64_t)qb; qi < (int64_t)qe; ++qi) {
        const uint64_t nr = re > rb ? re - rb : 0;
        float *row = (float *)malloc(sizeof(float) * (nr ? nr : 1));
        uint8_t *used = (uint8_t *)calloc(nr ? nr : 1, 1);
        for (uint64_t j = rb; j < re; ++j) {
            float x = dsho_pair(regs + j * m, regs + (uint64_t)qi * m, card[j], card[qi], p, estim, result_type, k);
            row[j - rb] = (x != x) ? worst : x;
        }
        for (uint32_t t = 0; t < nn; ++t) {
            int64_t best = -1;
            for (uint64_t j = 0; j < nr; ++j) {
                if (used[j] || j + rb == (uint64_t)qi) continue;
                if (best < 0 || (descending ? row[j] > row[best] : row[j] < row[best])) best = (int64_t)j;
            }
            uint32_t *io = idx_out + ((uint64_t)qi - qb) * nn + t;
            float *vo = val_out + ((uint64_t)qi - qb) * nn + t;
            if (best < 0) {
                *io = 0xFFFFFFFFu;
                *vo = worst;
            } else {
                used[best] = 1;
                *io = (uint32_t)(best + rb);
                *vo = dsho_pair(regs + ((uint64_t)best + rb) * m, regs + (uint64_t)qi * m, card[best + rb], card[qi], p, estim, result_type, k);
            }
        }
        free(row);
        free(used);
    }
    g_ksinv_double = 0;
    free(card);
}

int dsho_num_threads(void) { return nthreads_(); }

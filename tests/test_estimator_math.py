"""The three estimators against the MATHEMATICS they implement, written down here from the papers and not from any code:
Flajolet et al. 2007 (raw HyperLogLog estimate + linear counting) and Ertl 2017, "New cardinality estimation algorithms
for HyperLogLog sketches" (arXiv:1702.01284: the improved raw estimator with its sigma / tau series, and the Poisson-model
maximum likelihood).  The oracle restates dashing's CODE (src/dashing.h + the absent sketch submodule)
and stays "parity unpinned" (oracle/dsh_oracle.c header): these tests cannot pin upstream's bits.  What they do exclude
is a misreading shared by the oracle and the device code that would change WHAT is computed -- a wrong series, a wrong
constant, an iteration that stops somewhere else than at the likelihood's maximum."""
import math

import numpy as np
import pytest
from scipy.optimize import brentq

from dashing_amd import synth


def hist_of(regs, p):
    return np.bincount(regs, minlength=64).astype(np.uint32)


def sketches(p, seed):
    """register arrays over a wide range of fill: from almost empty to far beyond m"""
    rng = np.random.default_rng(seed)
    m, q = 1 << p, 64 - p
    out = []
    for card in (m // 50, m // 4, m, 3 * m, 40 * m, 3000 * m, 10 ** 7 * m):
        # the register law of n distinct hashed items: geometric per item, max per register
        n = int(card)
        if n <= 200000:
            idx = rng.integers(0, m, n)
            val = np.minimum(rng.geometric(0.5, n), q + 1).astype(np.uint8)
            regs = np.zeros(m, np.uint8)
            np.maximum.at(regs, idx, val)
        else:  # per register: P(reg <= k) = exp(-(n/m) 2^-k)
            u = rng.random(m)
            lam = n / m
            k = np.ceil(np.log2(-lam / np.log(u)))
            regs = np.clip(k, 0, q + 1).astype(np.uint8)
        out.append(regs)
    return out


def loglik_derivative(c, p, lam):
    """d/d lambda of Ertl's Poisson log-likelihood (his eq. for ln L(lambda | C), section 5):
    ln L = -(lambda/m) sum_{k=0..q} C_k 2^-k + sum_{k=1..q} C_k ln(1 - exp(-lambda/(m 2^k))) + C_{q+1} ln(1 - exp(-lambda/(m 2^q)))"""
    m, q = 1 << p, 64 - p
    d = -math.fsum(float(c[k]) * 2.0 ** -k for k in range(q + 1)) / m
    terms = []
    for k in range(1, q + 2):
        if not c[k]:
            continue
        s = lam / (m * 2.0 ** min(k, q))
        terms.append(float(c[k]) / (m * 2.0 ** min(k, q)) / math.expm1(s))  # d/dlam ln(1 - e^-s) = s' / (e^s - 1)
    return d + math.fsum(terms)


@pytest.mark.parametrize("p", [8, 10, 12, 14])
def test_mle_is_where_the_likelihood_peaks(oracle, p):
    m = 1 << p
    eps = 1e-2 / math.sqrt(m)  # the relative step at which the iteration stops
    for regs in sketches(p, 100 + p):
        c = hist_of(regs, p)
        if c[0] == m:
            continue
        got = oracle.estimate(c, p, 2)
        lo, hi = got / 4, got * 4
        assert loglik_derivative(c, p, lo) > 0 > loglik_derivative(c, p, hi)
        peak = brentq(lambda lam: loglik_derivative(c, p, lam), lo, hi, xtol=1e-12 * got, rtol=1e-14)
        assert abs(got - peak) <= 2 * eps * peak, (p, got, peak)


def sigma_series(x):
    if x == 1.0:
        return math.inf
    return x + math.fsum(x ** (2.0 ** k) * 2.0 ** (k - 1) for k in range(1, 80))


def tau_series(x):
    if x in (0.0, 1.0):
        return 0.0
    return (1.0 - x - math.fsum((1.0 - x ** (2.0 ** -k)) ** 2 * 2.0 ** -k for k in range(1, 200))) / 3.0


@pytest.mark.parametrize("p", [8, 10, 12, 14])
def test_improved_is_ertls_closed_form(oracle, p):
    """lambda = alpha_inf m^2 / (m sigma(C_0/m) + sum_{k=1..q} C_k 2^-k + m tau(1 - C_{q+1}/m) 2^-q), alpha_inf = 1/(2 ln 2)"""
    m, q = 1 << p, 64 - p
    for regs in sketches(p, 200 + p):
        c = hist_of(regs, p)
        den = m * sigma_series(c[0] / m) + math.fsum(float(c[k]) * 2.0 ** -k for k in range(1, q + 1)) + m * tau_series(1.0 - c[q + 1] / m) * 2.0 ** -q
        want = m * m / (2.0 * math.log(2.0)) / den
        got = oracle.estimate(c, p, 1)
        assert got == pytest.approx(want, rel=1e-12), (p, got, want)


@pytest.mark.parametrize("p", [8, 10, 12, 14])
def test_original_is_flajolets_raw_estimate_with_linear_counting(oracle, p):
    m = 1 << p
    alpha = 0.7213 / (1.0 + 1.079 / m)
    for regs in sketches(p, 300 + p):
        c = hist_of(regs, p)
        raw = alpha * m * m / math.fsum(2.0 ** -int(r) for r in regs)
        zeros = int((regs == 0).sum())
        want = m * math.log(m / zeros) if raw < 2.5 * m and zeros else raw
        if want > 2.0 ** 32 / 30:  # the 32-bit large-range correction dashing keeps
            want = -(2.0 ** 32) * math.log1p(-want / 2.0 ** 32) if want < 2.0 ** 32 else want
        got = oracle.estimate(c, p, 0)
        assert got == pytest.approx(want, rel=1e-12), (p, got, want)


def test_union_sketch_is_the_register_maximum(oracle):
    """|A u B| is estimated from max(a_t, b_t) (Flajolet: the sketch of a union is the register-wise maximum), the
    Jaccard index from inclusion-exclusion, clamped at 0 (src/dashing.h:138-140)."""
    p = 12
    regs, core, priv, cid = synth.related_sketches(6, p, seed=9)
    tri = oracle.dist_tri(regs, 2, oracle.JI, 31)
    card = oracle.cardinalities(regs, 2)
    t = 0
    for i in range(6):
        for j in range(i + 1, 6):
            u = oracle.estimate(hist_of(np.maximum(regs[i], regs[j]), p), p, 2)
            inter = max(0.0, card[i] + card[j] - u)
            assert tri[t] == pytest.approx(inter / u, rel=1e-6, abs=1e-12)
            t += 1

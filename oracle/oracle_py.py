"""Independent pure-Python/numpy restatement of the same path (second opinion for the C oracle).

TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (reference arithmetic is in absent submodules;
see oracle/dsh_oracle.c).  Deliberately written differently from dsh_oracle.c -- windows
are sliced rather than rolled, histograms come from numpy.bincount, the estimators use
Python floats (IEEE doubles, never FMA-contracted) -- so that a shared typo is unlikely.
Follows SURVEY.md Appendix A; call sites: src/sketch_and_cmp.h:342 (addh per k-mer),
src/readfilt.cpp:86-88 (register rule), src/dashing.h:138-156,172-174,568-592 (J, Mash, float).
"""
import math

import numpy as np

M64 = (1 << 64) - 1
ORIGINAL, ERTL_IMPROVED, ERTL_MLE = 0, 1, 2
MASH_DIST, JI, FULL_MASH_DIST = 0, 1, 3
SIZES, FULL_CONTAINMENT_DIST, CONTAINMENT_INDEX, CONTAINMENT_DIST = 2, 4, 5, 6
SYMMETRIC_CONTAINMENT_INDEX, SYMMETRIC_CONTAINMENT_DIST = 7, 8
_CODE = {ord("A"): 0, ord("a"): 0, ord("C"): 1, ord("c"): 1, ord("G"): 2, ord("g"): 2, ord("T"): 3, ord("t"): 3}


def wang(key):
    key &= M64
    key = ((~key & M64) + ((key << 21) & M64)) & M64
    key ^= key >> 24
    key = (key + ((key << 3) & M64) + ((key << 8) & M64)) & M64
    key ^= key >> 14
    key = (key + ((key << 2) & M64) + ((key << 4) & M64)) & M64
    key ^= key >> 28
    key = (key + ((key << 31) & M64)) & M64
    return key


def reg_rule(h, p):
    idx = h >> (64 - p)
    t = ((((h << 1) & M64) | 1) << (p - 1)) & M64
    clz = 64 - t.bit_length()
    return idx, clz + 1


def kmers(seq, k, canon=True):
    """All k-mers of one record by explicit window slicing (first base most significant)."""
    if isinstance(seq, str):
        seq = seq.encode()
    out = []
    for i in range(0, len(seq) - k + 1):
        win = seq[i : i + k]
        codes = [_CODE.get(b, -1) for b in win]
        if min(codes) < 0:
            continue
        fw = 0
        for c in codes:
            fw = (fw << 2) | c
        rc = 0
        for c in reversed(codes):
            rc = (rc << 2) | (3 - c)
        out.append(min(fw, rc) if canon else fw)
    return out


def sketch(seq, k, p, canon=True):
    regs = np.zeros(1 << p, np.uint8)
    for km in kmers(seq, k, canon):
        idx, v = reg_rule(wang(km), p)
        if v > regs[idx]:
            regs[idx] = v
    return regs


def hist_union(a, b=None):
    x = np.asarray(a, np.uint8)
    if b is not None:
        x = np.maximum(x, np.asarray(b, np.uint8))
    return np.bincount(x, minlength=64).astype(np.uint32)


def _div(a, b):
    """IEEE division (Python raises on /0)."""
    if b == 0.0:
        if a == 0.0 or a != a:
            return math.nan
        return math.copysign(math.inf, a) * math.copysign(1.0, b)
    return a / b


def _log1p(x):
    if x != x or x < -1.0:
        return math.nan
    if x == -1.0:
        return -math.inf
    return math.log1p(x)


def _alpha(m):
    return {16: 0.673, 32: 0.697, 64: 0.709}.get(m, 0.7213 / (1.0 + 1.079 / m))


def _sigma(x):
    if x == 1.0:
        return math.inf
    z, zp, y = x, 0.0, 1.0
    while z != zp:
        x *= x
        zp = z
        z += x * y
        y += y
    return z


def _tau(x):
    if x == 0.0 or x == 1.0:
        return 0.0
    z, y, zp = 1.0 - x, 1.0, x
    while zp != z:
        x = math.sqrt(x)
        zp = z
        y *= 0.5
        t = 1.0 - x
        z -= t * t * y
    return z / 3.0


def estimate(c, p, estim=ERTL_MLE):
    c = [int(v) for v in c]
    q = 64 - p
    m = 1 << p
    if estim == ORIGINAL:
        s = float(c[0])
        for i in range(1, q + 1):
            if c[i]:
                s += math.ldexp(float(c[i]), -i)
        value = _div(_alpha(m) * m * m, s)
        if value < 2.5 * m:
            if c[0]:
                value = m * math.log(m / c[0])
        elif value > 4294967296.0 / 30.0:
            corr = -4294967296.0 * _log1p(-math.ldexp(value, -32))
            if not math.isnan(corr):
                value = corr
        return value
    if estim == ERTL_IMPROVED:
        divinv = float.fromhex("0x1.71547652b82fep-1")  # 1/(2 ln 2)
        z = m * _tau((m - c[q + 1]) / m)
        for i in range(q, 0, -1):
            z += c[i]
            z *= 0.5
        z += m * _sigma(c[0] / m)
        return _div(m * divinv * m, z)
    # ERTL_MLE -- Ertl 2017 Alg. 8 with relative early stop 1e-2/sqrt(m)
    if c[q + 1] == m:
        return math.inf
    kmin = next(v for v in range(q + 2) if c[v])
    kmax = max(v for v in range(q + 2) if c[v])
    kminp = max(1, kmin)
    kmaxp = min(q, kmax)
    z = 0.0
    for v in range(kmaxp, kminp - 1, -1):
        z = 0.5 * z + c[v]
    z = math.ldexp(z, -kminp)
    cprime = c[q + 1] + (c[kmaxp] if q >= 1 else 0)
    a = z + c[0]
    mprime = m - c[0]
    b = z + math.ldexp(float(c[q + 1]), -q)
    if b <= 1.5 * a:
        x = mprime / (0.5 * b + a)
    else:
        x = (mprime / b) * math.log1p(b / a)
    gprev = 0.0
    dx = x
    eps = 1e-2 / math.sqrt(m)
    while dx > x * eps:
        _, e = math.frexp(x)
        xp = math.ldexp(x, -max(kmaxp + 1, e + 2))
        x2 = xp * xp
        h = xp - x2 / 3.0 + (x2 * x2) * (1.0 / 45.0 - x2 / 472.5)
        for _ in range(e, kmaxp - 1, -1):
            hp = 1.0 - h
            h = (xp + h * hp) / (xp + hp)
            xp += xp
        g = cprime * h
        for v in range(kmaxp - 1, kminp - 1, -1):
            hp = 1.0 - h
            h = (xp + h * hp) / (xp + hp)
            xp += xp
            g += c[v] * h
        g += x * a
        if gprev < g and g <= mprime:
            dx *= (g - mprime) / (gprev - g)
        else:
            dx = 0.0
        x += dx
        gprev = g
    return x * m


def cardinality(regs, p, estim=ERTL_MLE):
    return estimate(hist_union(regs), p, estim)


def jaccard(a, b, p, estim=ERTL_MLE):
    ca, cb = cardinality(a, p, estim), cardinality(b, p, estim)
    us = estimate(hist_union(a, b), p, estim)
    if us == 0.0:
        return 0.0  # (0+0-0)/0 = nan -> max(0., nan) = 0.
    ret = (ca + cb - us) / us if not math.isinf(us) else math.nan
    return ret if 0.0 < ret else 0.0


def result(ji, result_type, k):
    ksinv = float(np.float32(1.0 / k))
    if result_type == MASH_DIST:
        ret = -math.log(2.0 * ji / (1.0 + ji)) * ksinv if ji else 1.0
    elif result_type == FULL_MASH_DIST:
        ret = 1.0 - math.pow(2.0 * ji / (1.0 + ji), ksinv)
    else:
        ret = ji
    return float(np.float32(ret))


def result_triple(mys, os_, us, result_type, k):
    """second arm of result_cmp (src/dashing.h:577-588) on {max(mys-is,0), max(os-is,0), is}"""
    ksinv = float(np.float32(1.0 / k))
    mx0 = lambda x: 0.0 if x < 0.0 else x  # noqa: E731  (std::max(x, 0.): NaN propagates)
    is_ = mx0(mys + os_ - us)
    t0, t1, t2 = mx0(mys - is_), mx0(os_ - is_), is_
    ret = t2
    if result_type in (SYMMETRIC_CONTAINMENT_INDEX, SYMMETRIC_CONTAINMENT_DIST):
        ret = _div(ret, min(t0, t1) + t2)
        if result_type == SYMMETRIC_CONTAINMENT_DIST:
            ret = -math.log(ret) * ksinv if ret else 1.0
    elif result_type in (FULL_CONTAINMENT_DIST, CONTAINMENT_DIST, CONTAINMENT_INDEX):
        ret = _div(ret, t0 + t1 + t2)
        if result_type == CONTAINMENT_DIST:
            ret = -math.log(ret) * ksinv if ret else 1.0
        elif result_type == FULL_CONTAINMENT_DIST:
            ret = 1.0 - math.pow(ret, ksinv)
    return float(np.float32(ret))


def tri_index(n, i, j):
    return i * (2 * n - i - 1) // 2 + j - (i + 1)

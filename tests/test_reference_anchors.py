"""Everything the reference tree itself says about the arithmetic of this path, turned into checks of the oracle
(CPU tests) and of the device (gpu tests).  The arithmetic lives in absent submodules (parity unpinned);
these are the lines that ARE in /root/reference.  Expectations are computed here from the quoted formulas alone --
plain Python integers / floats, no oracle code -- and, where the reference is present (the build container), the cited
lines are checked to still read that way.

  R1  src/readfilt.cpp:86-88   register rule:  pos = h >> (64 - p);  v = clz(((h << 1) | 1) << (p - 1)) + 1;  max
  R2  src/khset64.h:129-141    full_set_comparison returns {mine - is, other - is, is};  :146-149 J = is / (sum of the three)
  R3  src/dashing.h:550-552    HLL intersection_size = max(0, creport(a) + creport(b) - union_size)
  R4  src/dashing.h:154-156    dist_index = ji ? -log(2 ji / (1 + ji)) * ksinv : 1;   :172-174 full_dist_index = 1 - pow(2 ji/(1+ji), ksinv)
  R5  src/dashing.h:568-592    result_cmp: second arm on set_triple: [2], [2]/(min([0],[1])+[2]), [2]/([0]+[1]+[2]); float return
  R6  src/distmain.cpp:29,36-38 defaults k = 31, sketch_size (S) = 10, result_type JI, estim ERTL_MLE
  R7  src/sketch_and_cmp.h:797 `const float ksinv = 1./ k` (dist)  vs  :729 `const double ksinv = 1./ k` (nearest neighbours)
"""
import math
import os
import re

import numpy as np
import pytest

import dashing_amd
from dashing_amd import synth

REF = "/root/reference"
MASK = (1 << 64) - 1


def clz64(x):
    return 64 - x.bit_length() if x else 64


def reg_rule(h, p):
    """R1, transcribed: src/readfilt.cpp:86-88"""
    pos = h >> (64 - p)
    v = clz64((((h << 1) | 1) << (p - 1)) & MASK) + 1
    return pos, v


def dist_index(ji, ksinv):
    """R4: src/dashing.h:154-156"""
    return -math.log(2.0 * ji / (1.0 + ji)) * ksinv if ji else 1.0


def full_dist_index(ji, ksinv):
    """R4: src/dashing.h:172-174"""
    return 1.0 - math.pow(2.0 * ji / (1.0 + ji), ksinv)


def _lines(rel, a, b):
    return "".join(open(os.path.join(REF, rel)).read().split("\n")[a - 1 : b])


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present (GPU box)")
def test_cited_lines_still_say_so():
    norm = lambda s: re.sub(r"\s+", "", s)
    assert "autopos=kmer>>(64-p);" in norm(_lines("src/readfilt.cpp", 84, 90))
    assert "uint8_tv=clz(((kmer<<1)|1)<<(p-1))+1;" in norm(_lines("src/readfilt.cpp", 84, 90))
    assert "rmap[pos]=std::max(rmap[pos],v);" in norm(_lines("src/readfilt.cpp", 84, 90))
    assert "{{this->n_occupied-is,other.n_occupied-is,is}}" in norm(_lines("src/khset64.h", 129, 141))
    assert "returndouble(cmps[2])/(cmps[0]+cmps[1]+cmps[2]);" in norm(_lines("src/khset64.h", 146, 149))
    assert "returnstd::max(0.,h1.creport()+h2.creport()-h1.union_size(h2));" in norm(_lines("src/dashing.h", 550, 552))
    assert "returnji?-std::log(2.*ji/(1.+ji))*ksinv:1.;" in norm(_lines("src/dashing.h", 154, 156))
    assert "return1.-std::pow(2.*ji/(1.+ji),ksinv);" in norm(_lines("src/dashing.h", 172, 174))
    assert "ret/=(std::min(triple[0],triple[1])+triple[2]);" in norm(_lines("src/dashing.h", 568, 592))
    assert "ret/=(triple[0]+triple[1]+triple[2]);" in norm(_lines("src/dashing.h", 568, 592))
    assert "returnstatic_cast<float>(ret);" in norm(_lines("src/dashing.h", 568, 592))
    d = norm(_lines("src/distmain.cpp", 28, 38))
    assert "k(31)" in d and "sketch_size(10)" in d and "EmissionTyperesult_type(JI);" in d
    assert "hll::EstimationMethodestim=hll::EstimationMethod::ERTL_MLE;" in d
    assert "constfloatksinv=1./k;" in norm(_lines("src/sketch_and_cmp.h", 797, 797))
    assert "constdoubleksinv=1./k;" in norm(_lines("src/sketch_and_cmp.h", 729, 729))
    e = norm(open(os.path.join(REF, "src/enums.h")).read())
    assert ("MASH_DIST=0,JI=1,SIZES=2,FULL_MASH_DIST=3,FULL_CONTAINMENT_DIST=4,CONTAINMENT_INDEX=5,CONTAINMENT_DIST=6,"
            "SYMMETRIC_CONTAINMENT_INDEX=7,SYMMETRIC_CONTAINMENT_DIST=8,") in e  # src/enums.h:13-23 = the DSH_* result types


HAND_HASHES = [0, 1, MASK, 1 << 63, (1 << 63) - 1, 0x0123456789ABCDEF, 0x8000000000000001, 0x00000000FFFFFFFF,
               0xFFFFFFFF00000000, 0x0000000000000400, 0x5555555555555555, 0xAAAAAAAAAAAAAAAA]


@pytest.mark.parametrize("p", [4, 10, 14, 18, 24])
def test_R1_register_rule_oracle(oracle, p):
    q = 64 - p
    assert reg_rule(0, p) == (0, q + 1) and reg_rule(MASK, p) == ((1 << p) - 1, 1)  # the survey's hand checks (A.3)
    if p == 14:
        assert reg_rule(1, p) == (0, 50)
    for h in HAND_HASHES:
        pos, v = reg_rule(h, p)
        assert 1 <= v <= q + 1 and 0 <= pos < (1 << p)
        assert oracle.reg_rule(h, p) == (pos, v), hex(h)


def test_R2_R3_R5_triple_order_and_measures_oracle(oracle):
    """the triple is {A - I, B - I, I} with I = max(0, Ca + Cb - U); every measure of result_cmp's second arm follows"""
    k = 21
    ksinv = float(np.float32(1.0 / k))  # R7: dist passes a float
    for mys, os_, us in ((1000.0, 400.0, 1100.0), (400.0, 1000.0, 1100.0), (500.0, 500.0, 1200.0), (123.5, 77.25, 150.0), (10.0, 10.0, 10.0)):
        inter = max(0.0, mys + os_ - us)
        t = (mys - inter, os_ - inter, inter)
        if min(t) < 0:
            continue
        want = {dashing_amd.SIZES: t[2],
                dashing_amd.SYMMETRIC_CONTAINMENT_INDEX: t[2] / (min(t[0], t[1]) + t[2]) if min(t[0], t[1]) + t[2] else float("nan"),
                dashing_amd.CONTAINMENT_INDEX: t[2] / (t[0] + t[1] + t[2])}
        ci = want[dashing_amd.CONTAINMENT_INDEX]
        want[dashing_amd.CONTAINMENT_DIST] = -math.log(ci) * ksinv if ci else 1.0
        sci = want[dashing_amd.SYMMETRIC_CONTAINMENT_INDEX]
        if sci == sci:
            want[dashing_amd.SYMMETRIC_CONTAINMENT_DIST] = -math.log(sci) * ksinv if sci else 1.0
        for rt, w in want.items():
            if w != w:
                continue
            got = oracle.result_triple(mys, os_, us, rt, k)
            assert got == pytest.approx(float(np.float32(w)), rel=1e-6, abs=1e-12), (rt, mys, os_, us)
        # R2 :146-149 and R3: J = I / (A-I + B-I + I) = I / U whenever the union estimate is consistent
        if us == mys + os_ - inter:
            assert oracle.jaccard_from(mys, os_, us) == pytest.approx(t[2] / sum(t), rel=1e-12)
    assert oracle.jaccard_from(100.0, 100.0, 250.0) == 0.0  # the max(0., .) of R3


def test_R4_R7_mash_transforms_oracle(oracle):
    for k in (21, 31):
        kf = float(np.float32(1.0 / k))
        for ji in (1.0, 0.5, 0.123456789, 1e-9):
            assert oracle.result(ji, dashing_amd.MASH_DIST, k) == pytest.approx(float(np.float32(dist_index(ji, kf))), rel=1e-6)
            assert oracle.result(ji, dashing_amd.FULL_MASH_DIST, k) == pytest.approx(float(np.float32(full_dist_index(ji, kf))), rel=1e-6, abs=1e-9)
        assert oracle.result(0.0, dashing_amd.MASH_DIST, k) == 1.0  # `ji ? ... : 1.`
        assert oracle.result(1.0, dashing_amd.MASH_DIST, k) == 0.0
    # float vs double 1/k is visible: the two differ in the 8th digit, the oracle follows the float (dist, :797)
    ji, k = 0.3, 31
    f32 = dist_index(ji, float(np.float32(1.0 / k)))
    f64 = dist_index(ji, 1.0 / k)
    assert f32 != f64 and abs(f32 - f64) / f64 < 1e-7


def test_R6_defaults():
    assert (dashing_amd.ESTIM_ERTL_MLE, dashing_amd.JI, dashing_amd.MASH_DIST, dashing_amd.FULL_MASH_DIST) == (2, 1, 0, 3)
    assert (dashing_amd.SIZES, dashing_amd.FULL_CONTAINMENT_DIST, dashing_amd.CONTAINMENT_INDEX, dashing_amd.CONTAINMENT_DIST,
            dashing_amd.SYMMETRIC_CONTAINMENT_INDEX, dashing_amd.SYMMETRIC_CONTAINMENT_DIST) == (2, 4, 5, 6, 7, 8)
    import inspect

    sig = inspect.signature(dashing_amd.Context.dist_rows)
    assert sig.parameters["estim"].default == dashing_amd.ESTIM_ERTL_MLE and sig.parameters["result_type"].default == dashing_amd.JI
    assert sig.parameters["k"].default == 31


# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("p", [10, 14])
def test_R1_register_rule_device(ctx, oracle, p):
    """device registers of a sequence = the in-tree rule applied to the hashes of its k-mers (hash and encoder from the
    oracle: they are not in the tree)"""
    k = 31
    g = synth.synthetic_genomes(1, 20_000, seed=77, decorate=False)[0]
    want = np.zeros(1 << p, np.uint8)
    for km in oracle.kmers(g.tobytes(), k, True):
        pos, v = reg_rule(oracle.wang(km), p)
        want[pos] = max(want[pos], v)
    seq, off = synth.concat_for_device([g])
    ctx.alloc(1, p)
    got = ctx.sketch_batch(seq, off, 0, k, True)[0]
    assert (got == want).all()


@pytest.mark.gpu
def test_R2_R3_R4_R5_R7_device_measures_are_the_in_tree_formulas(ctx):
    """every measure the device emits, recomputed in Python from the device's own cardinalities and SIZES output with the
    quoted formulas"""
    n, p, k = 160, 12, 31
    regs = synth.related_sketches(n, p, seed=12)[0]
    ctx.set_sketches(regs)
    card = ctx.cardinalities().astype(np.float64)
    out = {rt: ctx.dist_rows(result_type=rt, k=k).astype(np.float64) for rt in range(9)}
    kf = float(np.float32(1.0 / k))
    t = 0
    checked = 0
    for i in range(n):
        for j in range(i + 1, n):
            inter = out[dashing_amd.SIZES][t]                # triple[2]; lhs = sketch j, rhs = sketch i
            a, b = card[j] - inter, card[i] - inter          # {mine - is, other - is}
            if inter > 50.0 and a > 0 and b > 0:
                # (the inputs here are the device's float32 outputs: an index near 1 loses digits in -log(index))
                rel = lambda x, y: abs(x - y) <= 3e-6 * max(abs(y), 1e-9) + 2e-8
                ji = inter / (a + b + inter)                 # R2 :146-149 == R3 / union
                assert rel(out[dashing_amd.JI][t], ji)
                assert rel(out[dashing_amd.MASH_DIST][t], dist_index(ji, kf))
                assert rel(out[dashing_amd.FULL_MASH_DIST][t], full_dist_index(ji, kf)) or abs(out[3][t] - full_dist_index(ji, kf)) < 1e-7
                ci = inter / (a + b + inter)
                assert rel(out[dashing_amd.CONTAINMENT_INDEX][t], ci)
                assert rel(out[dashing_amd.CONTAINMENT_DIST][t], -math.log(ci) * kf)
                sci = inter / (min(a, b) + inter)
                assert rel(out[dashing_amd.SYMMETRIC_CONTAINMENT_INDEX][t], sci)
                assert rel(out[dashing_amd.SYMMETRIC_CONTAINMENT_DIST][t], -math.log(sci) * kf)
                checked += 1
            elif inter == 0.0:
                assert out[dashing_amd.JI][t] == 0.0 and out[dashing_amd.MASH_DIST][t] == 1.0  # `ji ? ... : 1.`
            t += 1
    assert checked > 300
    # R7: nearest neighbours use the double 1/k -- same pair, value differs from the dist value in the last digits only
    idx, val = ctx.knn(1, result_type=dashing_amd.MASH_DIST, k=k)
    i, j = 0, int(idx[0, 0])
    d32 = out[dashing_amd.MASH_DIST][dashing_amd.tri_index(n, min(i, j), max(i, j))]
    assert abs(float(val[0, 0]) - d32) <= 2e-7 * max(d32, 1e-9) + 1e-12


def _canonical_kmers(codes, k):
    """all canonical k-mers (first base most significant, min(forward, reverse complement)) of a 2-bit code array"""
    L = codes.size - k + 1
    fw = np.zeros(L, np.uint64)
    rc = np.zeros(L, np.uint64)
    c = codes.astype(np.uint64)
    for t in range(k):
        fw = (fw << np.uint64(2)) | c[t : t + L]
        rc = rc | ((np.uint64(3) - c[t : t + L]) << np.uint64(2 * t))
    return np.minimum(fw, rc)


@pytest.mark.gpu
def test_statistical_accuracy_vs_exact_kmer_sets_c1(ctx):
    """HLL Jaccard of the device against the EXACT Jaccard of the k-mer sets (the reference's
    --use-full-khash-sets ground truth, src/khset64.h:129-149) on configs[0]-sized genomes (1 Mbp, k=31, p=10): within
    3 * 1.04 / sqrt(m)."""
    k, p, n, L = 31, 10, 12, 1_000_000
    gs = synth.synthetic_genomes(n, L, seed=0xDA5410, decorate=False)
    lut = np.zeros(256, np.uint8)
    for ch, v in zip(b"ACGT", range(4)):
        lut[ch] = v
    sets = [np.unique(_canonical_kmers(lut[g], k)) for g in gs]
    seq, off = synth.concat_for_device(gs)
    ctx.alloc(n, p)
    ctx.sketch_batch(seq, off, 0, k, True, want_regs=False)
    got = ctx.dist_rows(result_type=dashing_amd.JI, k=k)
    card = ctx.cardinalities()
    bound = 3 * 1.04 / math.sqrt(1 << p)
    t = 0
    worst = 0.0
    for i in range(n):
        assert abs(card[i] - sets[i].size) / sets[i].size < bound  # cardinalities too
        for j in range(i + 1, n):
            inter = np.intersect1d(sets[i], sets[j], assume_unique=True).size
            exact = inter / (sets[i].size + sets[j].size - inter)
            worst = max(worst, abs(float(got[t]) - exact))
            t += 1
    assert worst < bound, worst

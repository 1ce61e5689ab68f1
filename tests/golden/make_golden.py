#!/usr/bin/env python3
"""Generates tests/golden/kat.json -- known-answer vectors for the hot path.

SELF-CONSISTENCY, NOT UPSTREAM PARITY: the reference cannot be built here (its arithmetic is
in absent submodules, see oracle/dsh_oracle.c) and ships no golden vectors, so these values
come from the independent pure-Python restatement (oracle/oracle_py.py) and are asserted equal
to the C oracle (oracle/dsh_oracle.c) at generation time.  Inputs + expected outputs only.
Run from the repo root:  python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle_c as oc  # noqa: E402
from oracle import oracle_py as op  # noqa: E402
from dashing_amd import synth  # noqa: E402


def main():
    kat = {"note": "self-consistency vectors (parity unpinned: reference not buildable)"}
    # A.8-1 Wang hash
    keys = [0, 1, (1 << 64) - 1, 0x0123456789ABCDEF, 0xDEADBEEFCAFEBABE, 1 << 63]
    kat["wang"] = [[hex(k), hex(op.wang(k))] for k in keys]
    for k in keys:
        assert op.wang(k) == oc.wang(k)
    # A.8-3 register rule
    rr = []
    for p in (10, 14):
        for h in (0, (1 << 64) - 1, 1, 1 << 40, 0x8000000000000000, 0x0123456789ABCDEF):
            idx, v = op.reg_rule(h, p)
            assert (idx, v) == oc.reg_rule(h, p)
            rr.append([p, hex(h), idx, v])
    kat["reg_rule"] = rr
    # A.8-2 encoder
    seqs = [
        "ACGT", "ACGTACGTAC", "ACGTNACGTTTGACCAGTacgtagctagGGATCGATCGATTTAGC",
        "AAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAA", "GATTACA", "acgtnnacgtRYacgtacgtacgtacgtacgtacgtacgtacgtac",
        "T" * 31 + "ACGT",  # the longest poly-T run that is not affected by the upstream quirk noted below
    ]
    enc = []
    for s in seqs:
        for k in (3, 5, 31, 32):
            for canon in (True, False):
                a = op.kmers(s, k, canon)
                assert a == oc.kmers(s, k, canon), (s, k, canon)
                enc.append({"seq": s, "k": k, "canon": canon, "kmers": [hex(x) for x in a]})
    kat["encoder"] = enc
    # NOT a golden vector: SURVEY.md A.1 recalls that upstream bonsai detects ambiguity as "accumulator == all-ones",
    # so a run of >= 32 consecutive T at k = 31/32 may spuriously reset the window there.  Our restatement (and the
    # product) emit those k-mers.  Unverifiable without the bonsai submodule, so the case is recorded, not asserted.
    kat["unpinned_notes"] = [{"case": "poly-T run of >= 32 bases (k >= 31)",
                              "ours": "every window of k valid bases emits its k-mer",
                              "upstream_recollection": "may reset the window (all-ones accumulator read as ambiguous)",
                              "status": "unpinned; no test asserts either behaviour"}]
    # A.8-3 a ~1000-k-mer genome -> full register dump (p=10), with an N run and lowercase
    g = synth.synthetic_genomes(1, 1030, seed=0xBEEF, decorate=False)[0]
    g[500:503] = ord("N")
    g[700:760] |= 0x20
    gs = g.tobytes().decode()
    regs_py = op.sketch(gs, 31, 10, True)
    seq, off = synth.concat_for_device([g])
    regs_c = oc.sketch_batch(seq, off, 31, 10, True)[0]
    assert (regs_py == regs_c).all()
    kat["sketch_small"] = {"seq": gs, "k": 31, "p": 10, "canon": True, "regs_hex": regs_py.tobytes().hex()}
    # A.8-4 estimators on hand-made histograms
    est = []

    def hist(p, d):
        h = [0] * 64
        for v, c in d.items():
            h[v] = c
        assert sum(h) == 1 << p
        return h

    cases = [
        (10, {0: 1024}), (10, {54 + 1: 1024}), (10, {0: 1023, 7: 1}), (10, {0: 300, 1: 300, 2: 200, 3: 124, 4: 60, 5: 40}),
        (14, {0: 16384}), (14, {51: 16384}), (14, {6: 1000, 7: 4000, 8: 6000, 9: 3000, 10: 1500, 11: 500, 12: 250, 13: 100, 14: 34}),
        (14, {0: 8000, 1: 5000, 2: 2000, 3: 1000, 4: 384}), (10, {50: 1000, 55: 24}), (12, {20: 4096}),
        (4, {0: 3, 1: 6, 2: 4, 3: 2, 61: 1}), (6, {3: 30, 4: 20, 5: 10, 9: 4}),
    ]
    for p, d in cases:
        h = hist(p, d)
        row = {"p": p, "hist": {str(k): v for k, v in d.items()}}
        for name, e in (("original", 0), ("improved", 1), ("mle", 2)):
            a = op.estimate(h, p, e)
            b = oc.estimate(np.array(h, np.uint32), p, e)
            assert a == b or (a != a and b != b), (p, d, name, a, b)
            row[name] = float(a).hex() if a == a else "nan"
        est.append(row)
    kat["estimators"] = est
    # A.8-5 pairs on synthetic related sketches (p=10 and 14): registers + expected floats
    pairs = []
    for p in (10, 14):
        regs, core, priv, cid = synth.related_sketches(12, p, seed=0x5EED0000 + p)
        regs[11] = 0  # an empty sketch: J with it must be 0 and Mash 1
        for e in (0, 1, 2):
            card = oc.cardinalities(regs, e)
            for i in range(12):
                assert card[i] == op.cardinality(regs[i], p, e)
            for rt in (op.JI, op.MASH_DIST, op.FULL_MASH_DIST):
                tri = oc.dist_tri(regs, e, rt, 31)
                # spot-check against the python restatement
                for (i, j) in ((0, 1), (0, 4), (2, 3), (0, 10), (5, 11), (10, 11)):
                    ji = op.jaccard(regs[i], regs[j], p, e)
                    assert np.float32(op.result(ji, rt, 31)) == tri[op.tri_index(12, i, j)], (p, e, rt, i, j)
                pairs.append({"p": p, "estim": e, "result_type": rt, "k": 31,
                              "tri_hex": tri.tobytes().hex(), "card_hex": card.tobytes().hex()})
        np.save(os.path.join(ROOT, "tests", "golden", "regs_p%d.npy" % p), regs)
    kat["pairs"] = pairs
    # second arm of result_cmp: set_triple measures on fixed (mys, os, us) triples
    tri = []
    for rt in (2, 4, 5, 6, 7, 8):
        for (a, b, u) in [(1000.0, 2000.0, 2500.0), (5e6, 5e6, 5e6), (100.0, 200.0, 400.0), (10.0, 20.0, 30.0), (3e6, 4e6, 6.5e6)]:
            x, y = op.result_triple(a, b, u, rt, 31), oc.result_triple(a, b, u, rt, 31)
            assert x == y, (rt, a, b, u)
            tri.append([rt, a, b, u, float(x).hex()])
    kat["set_triple"] = tri
    with open(os.path.join(ROOT, "tests", "golden", "kat.json"), "w") as f:
        json.dump(kat, f, indent=0)
    print("wrote kat.json:", {k: (len(v) if hasattr(v, "__len__") else v) for k, v in kat.items()})


if __name__ == "__main__":
    main()

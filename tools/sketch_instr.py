#!/usr/bin/env python3
"""Instruction budget of k_sketch per k-mer, by phase, from the ISA the product is built from (the
counterpart of tools/finalize_instr.py, which cuts k_finalize at run time -- k_sketch's inner loop is fully unrolled
straight-line code, so its phases can be read off the listing).

  python tools/sketch_instr.py            # both variants of the LDS / canonical instance, JSON on stdout

For k_sketch<GLOBAL=false, CANON=true> (profiles/rd5d/sketch_instr.json holds the table of commit 1383876, where the kernel
of rounds 1-4 was still present as variant 0 next to the trimmed stream, variant 1 = the kernel of this tree):
  * per k-mer, the unrolled body between two filter reads (median over the 32 start positions), split in order into
    validity test | window (funnel shifts, k-mer mask) | canonical (64-bit compare + select) | Wang hash | register rule
    (index, leading zeros) | filter (LDS address, compare) -- and the CAS path behind the filter, which almost no k-mer takes;
  * per 32 bases, the pack (ASCII -> 2-bit words F, R and validity bits V) and the window test, amortised per k-mer;
  * issue classes (tools/isa_mix.py: full-rate VOP1/VOP2 = 2 cycles per wave64 instruction, everything else 4).
The measured total per k-mer is SQ_INSTS_VALU x 64 / bases of the bench's PMC pass (bench.py configs[1] entry)."""
import collections
import json
import os
import re
import statistics
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FULL = re.compile(r"^v_(mov_b32|and_b32|or_b32|xor_b32|not_b32|add_u32|sub_u32|subrev_u32|add_co_u32|addc_co_u32|sub_co_u32|subb_co_u32|"
                  r"cndmask_b32|min_u32|max_u32|cmp_[a-z]+_[ui]32|readfirstlane_b32|mov_b64|lshlrev_b32|lshrrev_b32|ffbh_u32|bfrev_b32)(_e32|_e64|_sdwa)?$")


def listing():
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-S", "--cuda-device-only",
                               "-I" + os.path.join(ROOT, "dashing_amd", "csrc"), "-o", out, os.path.join(ROOT, "dashing_amd", "csrc", "kernels_sketch.hip")],
                              stderr=subprocess.DEVNULL)
        return open(out).read().split("\n")


def body_of(lines, prefix):
    i0 = next(i for i, l in enumerate(lines) if l.startswith(prefix) and ":" in l)
    out = []
    for l in lines[i0 + 1:]:
        t = l.strip()
        if t.startswith("s_endpgm"):
            break
        if t and not t.startswith((";", ".")):
            out.append(t)
    return out


def phases(block):
    """split one k-mer's straight-line VALU instructions, in program order, into the phases of the source"""
    ph = collections.OrderedDict((k, []) for k in ("validity", "window", "canonical", "hash", "register_rule", "filter", "cas_path"))
    ops = [l.split()[0] for l in block]
    valu = [(i, o) for i, o in enumerate(ops) if o.startswith("v_")]
    names = [o for _, o in valu]
    # anchors
    cmp64 = next((k for k, o in enumerate(names) if o.startswith("v_cmp_lt_u64")), None)
    last_mad = max((k for k, o in enumerate(names) if o.startswith("v_mad_u64_u32")), default=None)
    # the filter compare is the v_cmp_ge_u32 right behind the LDS read's waitcnt; the CAS path follows it
    fcmp = next((k for k, o in enumerate(names) if o.startswith("v_cmp_ge_u32")), len(names) - 1)
    for k, o in enumerate(names):
        if cmp64 is not None and k < cmp64:
            key = "validity" if (o.startswith("v_cmp_ne_u32") or (o.startswith("v_and_b32") and k < 2 and any(n.startswith("v_cmp_ne_u32") for n in names[:3]))) else "window"
        elif cmp64 is not None and k <= cmp64 + 2:
            key = "canonical"
        elif last_mad is not None and k <= last_mad + (1 if k + 0 <= last_mad else 0):
            key = "hash"
        elif k < fcmp:
            key = "register_rule"
        elif k == fcmp:
            key = "filter"
        else:
            key = "cas_path"
        ph[key].append(o)
    return ph


def analyse(lines, var):
    body = body_of(lines, "_ZN3dsh8k_sketchILb0ELb1ELi%dEEE" % var if any(l.startswith("_ZN3dsh8k_sketchILb0ELb1ELi") for l in lines)
                   else "_ZN3dsh8k_sketchILb0ELb1EEE")
    reads = [i for i, l in enumerate(body) if l.startswith("ds_read_u8")]
    # a k-mer's code ends where the exec mask narrowed by its filter (and its validity test) is restored: the
    # `s_or_b64 exec, exec, s[..]` that is NOT part of the CAS retry loop (those are followed by an s_and_b64)
    ends = [i for i, l in enumerate(body) if l.startswith("s_or_b64 exec, exec") and not body[min(i + 1, len(body) - 1)].startswith("s_and_b64")]
    ends = [e for e in ends if e > reads[0]]
    first = max([i for i, l in enumerate(body[:reads[0]]) if l.startswith(("s_cbranch", "s_waitcnt", "ds_read_b", "ds_write"))] + [0])
    cuts = [first] + ends
    blocks = [body[cuts[k] + 1: cuts[k + 1] + 1] for k in range(len(cuts) - 1)]
    blocks = [b for b in blocks if any(l.startswith("ds_read_u8") for l in b)]
    per = []
    for b in blocks:
        ph = phases(b)
        per.append({k: len(v) for k, v in ph.items()})
    def med(key, sel):
        v = [p[key] for p in sel]
        return statistics.median(v) if v else 0
    groups = {"all_positions": per}
    if var == 1 and len(per) >= 60:  # two copies of the loop: per-position test first (the listing's order), then the all-valid copy
        groups = {"per_position_test_path": per[:31], "all_valid_path": per[32:]}
    out = {"variant": var, "static_valu_total": sum(1 for l in body if l.startswith("v_")), "filter_reads": len(reads),
           "pack_and_prologue_valu_before_first_kmer": sum(1 for l in body[:reads[0]] if l.startswith("v_"))}
    for g, sel in groups.items():
        row = {k: med(k, sel) for k in ("validity", "window", "canonical", "hash", "register_rule", "filter", "cas_path")}
        row["main_path_per_kmer"] = sum(v for k, v in row.items() if k != "cas_path")
        out[g] = row
    cnt = collections.Counter(l.split()[0] for b in blocks for l in b if l.startswith("v_"))
    full = sum(c for k, c in cnt.items() if FULL.match(k))
    tot = sum(cnt.values())
    out["issue_classes_of_the_unrolled_loop"] = {"full_rate_share": round(full / max(tot, 1), 4),
                                                 "nominal_cycles_per_inst": round((2.0 * full + 4.0 * (tot - full)) / max(tot, 1), 3)}
    return out


def analyse_reg32(lines):
    """round 6: the instance with one 32-bit word per LDS register (p <= 15; the 256-lane instance of p <= 13) and k = 31 folded in: a k-mer's code ends in ONE
    ds_max_i32 (the register holds value - 1 = the leading zeros of t's high word; no filter read, no branch, no CAS path).
    The per-position-test copy of the unrolled loop keeps one k-mer per exec-masked block: VALU between consecutive
    ds_max_i32 there, by phase where the anchors allow (median over the start positions).  In the all-valid copy hipcc
    interleaves the k-mers, so that copy is counted as a whole: its VALU instructions / 32."""
    prefix = next((pf for pf in ("_ZN3dsh8k_sketchILb0ELb1ELb1ELi31ELi256EEE", "_ZN3dsh8k_sketchILb0ELb1ELb1ELi31EEE", "_ZN3dsh8k_sketchILb0ELb1ELb1EEE") if any(l.startswith(pf) for l in lines)), None)
    body = body_of(lines, prefix)
    atom = "ds_max_i32" if any(l.startswith("ds_max_i32") for l in body) else "ds_max_u32"
    anchors = [i for i, l in enumerate(body) if l.startswith(atom)]
    tested = [any(l.startswith(("s_and_saveexec", "s_cbranch")) for l in body[anchors[k]: anchors[k + 1]]) for k in range(len(anchors) - 1)]
    blocks = [body[anchors[k] + 1: anchors[k + 1] + 1] for k in range(len(anchors) - 1) if tested[k]]
    per = []
    for b in blocks:
        names = [l.split()[0] for l in b if l.startswith("v_")]
        cmp64 = next((k for k, o in enumerate(names) if o.startswith("v_cmp_lt_u64")), None)
        last_mad = max((k for k, o in enumerate(names) if o.startswith("v_mad_u64_u32")), default=None)
        if cmp64 is None or last_mad is None:
            continue
        per.append({"window_and_validity": cmp64, "canonical": 3, "hash": last_mad + 1 - (cmp64 + 3), "register_rule": len(names) - (last_mad + 1), "total": len(names)})
    med = lambda key: statistics.median([p[key] for p in per]) if per else 0
    # the all-valid copy: the longest run of anchors with no exec test between them
    best, cur = (0, 0), None
    for k, t in enumerate(tested + [True]):
        if not t and cur is None:
            cur = k
        if t and cur is not None:
            if k - cur > best[1] - best[0]:
                best = (cur, k)
            cur = None
    all_valid = None
    if best[1] > best[0]:
        start = max(i for i in range(anchors[best[0]]) if body[i].startswith("s_cbranch"))
        n_kmers = best[1] - best[0] + 1
        seg = [l.split()[0] for l in body[start + 1: anchors[best[1]] + 1] if l.startswith("v_")]
        full = sum(1 for o in seg if FULL.match(o))
        all_valid = {"kmers": n_kmers, "valu": len(seg), "valu_per_kmer": round(len(seg) / n_kmers, 2),
                     "full_rate_share": round(full / max(len(seg), 1), 4), "nominal_cycles_per_kmer": round((2.0 * full + 4.0 * (len(seg) - full)) / n_kmers, 1)}
    cnt = collections.Counter(l.split()[0] for b in blocks for l in b if l.startswith("v_"))
    full = sum(c for k, c in cnt.items() if FULL.match(k))
    tot = sum(cnt.values())
    return {"instance": prefix, "static_valu_total": sum(1 for l in body if l.startswith("v_")), atom: len(anchors),
            "ds_cmpst": sum(1 for l in body if l.startswith("ds_cmpst")),
            "per_position_test_copy_per_kmer_median": {k: med(k) for k in ("window_and_validity", "canonical", "hash", "register_rule", "total")},
            "all_valid_copy": all_valid,
            "pack_and_prologue_valu_before_first_kmer": sum(1 for l in body[:anchors[0]] if l.startswith("v_")) if anchors else None,
            "issue_classes_of_the_per_position_copy": {"full_rate_share": round(full / max(tot, 1), 4), "nominal_cycles_per_inst": round((2.0 * full + 4.0 * (tot - full)) / max(tot, 1), 3)}}


def main():
    lines = listing()
    if any(l.startswith("_ZN3dsh8k_sketchILb0ELb1ELb1E") for l in lines):  # round 6: <GLOBAL, CANON, REG32[, KC]>
        byte_body = body_of(lines, "_ZN3dsh8k_sketchILb0ELb1ELb0ELi0ELi256E" if any(l.startswith("_ZN3dsh8k_sketchILb0ELb1ELb0ELi0ELi256E") for l in lines) else "_ZN3dsh8k_sketchILb0ELb1ELb0E")
        print(json.dumps({"kernel": "k_sketch<GLOBAL=false, CANON=true, REG32, KC=31, NT=256>", "reg32": analyse_reg32(lines),
                          "bytes_instance_static": {"static_valu_total": sum(1 for l in byte_body if l.startswith("v_")),
                                                    "ds_read_u8": sum(1 for l in byte_body if l.startswith("ds_read_u8")),
                                                    "ds_cmpst": sum(1 for l in byte_body if l.startswith("ds_cmpst"))},
                          "note": "static counts of the unrolled body (median over start positions, both copies of the loop); the measured total per k-mer is SQ_INSTS_VALU x 64 / bases of bench.py's PMC pass (configs[1] entry)"}, indent=1))
        return
    both = any(l.startswith("_ZN3dsh8k_sketchILb0ELb1ELi") for l in lines)
    res = {"kernel": "k_sketch<GLOBAL=false, CANON=true>", "variants": [analyse(lines, 0), analyse(lines, 1)] if both else [analyse(lines, 1)],
           "note": "static counts of the unrolled body (median over start positions); per 32 bases the pack + window test add pack_and_prologue/32 per k-mer for every lane and once more for the 64 lanes that pack the sub-chunk's right neighbour"}
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()

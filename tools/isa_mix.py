#!/usr/bin/env python3
"""Static VALU instruction mix of one kernel of the library, by issue class (profiles/ubench/valu_rates.txt,
valu64_rates.txt: VOP1/VOP2 moves, logic, adds and compares issue a wave64 instruction every ~2 cycles (2.5 measured at 8
waves per SIMD), shifts, VOP3 three-operand forms, multiplies and the 64-bit ops every ~4 (4.1-4.7 measured)).

  python tools/isa_mix.py kernels_sketch.hip _ZN3dsh8k_sketchILb0ELb1EEE

Prints the counts and the mix-aware issue ceiling in cycles per VALU instruction (nominal 2 / 4 and measured 2.5 / 4.4)."""
import collections
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FULL = re.compile(r"^v_(mov_b32|and_b32|or_b32|xor_b32|not_b32|add_u32|sub_u32|subrev_u32|add_co_u32|addc_co_u32|sub_co_u32|subb_co_u32|"
                  r"cndmask_b32|min_u32|max_u32|cmp_[a-z]+_[ui]32|readfirstlane_b32|mov_b64)(_e32|_e64|_sdwa)?$")


def main():
    src, prefix = sys.argv[1], sys.argv[2]
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-S",
                               "--cuda-device-only", "-I" + os.path.join(ROOT, "include"), "-o", out,
                               os.path.join(ROOT, "dashing_amd", "csrc", src)], stderr=subprocess.DEVNULL)
        lines = open(out).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith(prefix) and l.rstrip().endswith(":") or (l.startswith(prefix) and ":" in l and not l.startswith("\t")))
    cnt = collections.Counter()
    for l in lines[start + 1:]:
        t = l.strip().split()
        if not t:
            continue
        if t[0] == "s_endpgm":
            break
        if t[0].startswith("v_"):
            cnt[t[0]] += 1
    full = sum(c for k, c in cnt.items() if FULL.match(k))
    total = sum(cnt.values())
    half = total - full
    print(json.dumps({"kernel": prefix, "valu_static": total, "full_rate": full, "half_rate": half, "full_rate_share": round(full / total, 4),
                      "ceiling_cycles_per_valu_inst_nominal_2_4": round((2.0 * full + 4.0 * half) / total, 3),
                      "ceiling_cycles_per_valu_inst_measured_2p5_4p4": round((2.5 * full + 4.4 * half) / total, 3),
                      "top": cnt.most_common(12)}))


if __name__ == "__main__":
    main()

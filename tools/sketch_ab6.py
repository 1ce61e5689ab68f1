"""Round 6 A/B of k_sketch's register rule: packed bytes + filter + CAS loop (DSH_SKETCH_BYTES=1, the
kernel of rounds 1-5) against one 32-bit word per register + ds_max_u32 (the default up to p = kMaxPReg32), each in its
own process, same inputs, alternating; tools/bench_sketch.py prints bases/s and checks the registers against the oracle."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = ((10, 1000), (12, 300), (13, 300), (8, 300)) if not os.environ.get("HIGH_P") else ((14, 300), (15, 200))
for p, G in CASES:
    for rep in range(2):
        for mode in ("bytes", "reg32"):
            env = dict(os.environ)
            if mode == "bytes":
                env["DSH_SKETCH_BYTES"] = "1"
            elif os.environ.get("HIGH_P"):
                env["DSH_SKETCH_REG32_MAXP"] = "15"
            r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bench_sketch.py"), "--genomes", str(G), "--p", str(p), "--cpu-genomes", "2"],
                               capture_output=True, env=env, timeout=900)
            line = next((l for l in r.stdout.decode().splitlines() if l.startswith("{")), None)
            d = json.loads(line) if line else {"error": r.stderr.decode()[-300:]}
            print(json.dumps({"p": p, "genomes": G, "registers": mode, "rep": rep, "bases_per_s": d.get("value"), "ms": d.get("ms_per_step"),
                              "registers_bit_exact": d.get("registers_bit_exact"), "error": d.get("error")}), flush=True)

"""A/B: sub-chunks (8 192 bases) per workgroup of k_sketch at BASELINE configs[1] size (DSH_SKETCH_SUBS), separate processes."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for rep in range(2):
    for subs in (16, 8, 32, 64, 128):
        env = dict(os.environ, DSH_SKETCH_SUBS=str(subs))
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bench_sketch.py"), "--genomes", "1000", "--p", os.environ.get("P", "10"), "--cpu-genomes", "2"],
                           capture_output=True, env=env, timeout=900)
        line = next((l for l in r.stdout.decode().splitlines() if l.startswith("{")), None)
        d = json.loads(line) if line else {"error": r.stderr.decode()[-300:]}
        print(json.dumps({"subs_per_workgroup": subs, "rep": rep, "bases_per_s": d.get("value"), "ms": d.get("ms_per_step"), "registers_bit_exact": d.get("registers_bit_exact"),
                          "error": d.get("error")}), flush=True)

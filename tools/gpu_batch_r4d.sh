#!/bin/bash
# round 4, GPU batch d: k_finalize v2 after fixes (per-column data rebuilt with the per-sketch pass, pointer-driven MLE loop)
set -x
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r4d
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "rc $?" >> $O/pytest_gpu.log; tail -4 $O/pytest_gpu.log
rm -f $O/finalize_phases*.jsonl $O/finalize_instr.jsonl
timeout 600 python tools/finalize_probe.py --workloads C3,C4 --out $O/finalize_phases.jsonl > $O/finalize_probe.log 2>&1
python - <<'PY'
import json
for l in open("gpurun_out/r4d/finalize_phases.jsonl"):
    r = json.loads(l)
    print(r["workload"], r["layout"], r["estim"], "fin", r["finalize_ms"], "pair", r["pair_ms"], "stamped", r["finalize_ms_stamped"], r["phase_ms_of_kernel"], "cyc/wave", r["cycles_per_wave"])
PY
timeout 600 python tools/finalize_instr.py --workloads C4,C3 --out $O/finalize_instr.jsonl > $O/finalize_instr.log 2>&1; cat $O/finalize_instr.jsonl
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
python - <<'PY'
import json
l=json.loads([x for x in open("gpurun_out/r4d/bench.json") if x.startswith("{")][-1])
print("value", l.get("value"), "ms", l.get("ms_per_step"), "err", l.get("error"))
print("step", l.get("roofline",{}).get("step"))
print("finalize", l.get("roofline",{}).get("finalize",{}).get("binding"))
for c in l.get("configs") or []:
    print(json.dumps({k: c.get(k) for k in ("workload","error","pairs_per_s","bases_per_s","ms_per_step","kernel_ms","parity")})[:500])
PY

#!/bin/bash
# round 4, GPU batch e: trimmed k_finalize v2 -- whole GPU suite, instruction counts per phase, phase stamps
set -x
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r4e
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "rc $?" >> $O/pytest_gpu.log; tail -4 $O/pytest_gpu.log
rm -f $O/finalize_phases*.jsonl $O/finalize_instr.jsonl
timeout 600 python tools/finalize_instr.py --workloads C4,C3 --out $O/finalize_instr.jsonl > $O/finalize_instr.log 2>&1
timeout 600 python tools/finalize_probe.py --workloads C3,C4 --out $O/finalize_phases.jsonl > $O/finalize_probe.log 2>&1
python - <<'PY'
import json
for l in open("gpurun_out/r4e/finalize_phases.jsonl"):
    r = json.loads(l)
    print(r["workload"], r["layout"], r["estim"], "fin", r["finalize_ms"], "pair", r["pair_ms"], "stamped", r["finalize_ms_stamped"], r["phase_ms_of_kernel"], "cyc/wave", r["cycles_per_wave"])
PY

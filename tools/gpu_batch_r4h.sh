#!/bin/bash
# round 4, GPU batch h: epilogue loads requested before the estimator, in-band list rows
set -x
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r4h
mkdir -p "$O"
timeout 900 python -m pytest tests/test_gpu_multirank.py tests/test_gpu_compare.py tests/test_gpu_fuzz.py -x -q > $O/pytest_part.log 2>&1; echo "rc $?" >> $O/pytest_part.log; tail -4 $O/pytest_part.log
: > $O/finalize_instr.jsonl
: > $O/finalize_phases.jsonl
timeout 600 python tools/finalize_instr.py --workloads C4,C3 --out $O/finalize_instr.jsonl > $O/finalize_instr.log 2>&1
timeout 600 python tools/finalize_probe.py --workloads C3,C4 --out $O/finalize_phases.jsonl > $O/finalize_probe.log 2>&1
python - <<'PY'
import json
for l in open("gpurun_out/r4h/finalize_instr.jsonl"):
    r = json.loads(l); print(r.get("workload"), r.get("estim"), r.get("finalize_stop"), r.get("per_wave"), r.get("error"))
for l in open("gpurun_out/r4h/finalize_phases.jsonl"):
    r = json.loads(l)
    print(r["workload"], r["layout"], r["estim"], "fin", r["finalize_ms"], "pair", r["pair_ms"], r["phase_ms_of_kernel"])
PY

#!/bin/bash
# round 4, GPU batch c: k_finalize v2 (layout-ordered inputs, inline bucket records) -- correctness first, then timing
set -x
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r4c
mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_compare.py tests/test_gpu_fuzz.py -x -q > $O/pytest_compare.log 2>&1; echo "rc $?" >> $O/pytest_compare.log; tail -4 $O/pytest_compare.log
rm -f $O/finalize_phases*.jsonl
timeout 600 python tools/finalize_probe.py --workloads C3,C4 --out $O/finalize_phases.jsonl > $O/finalize_probe.log 2>&1
python - <<'PY'
import json
for l in open("gpurun_out/r4c/finalize_phases.jsonl"):
    r = json.loads(l)
    print(r["workload"], r["layout"], r["estim"], "fin", r["finalize_ms"], "pair", r["pair_ms"], "stamped", r["finalize_ms_stamped"], r["phase_ms_of_kernel"], "cyc/wave", r["cycles_per_wave"])
PY
G200=1 timeout 300 python tools/debug_sketch_parity.py > $O/debug_sketch.log 2>&1; grep "^{" $O/debug_sketch.log
timeout 900 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_compare.py --deselect tests/test_gpu_fuzz.py > $O/pytest_rest.log 2>&1; echo "rc $?" >> $O/pytest_rest.log; tail -4 $O/pytest_rest.log

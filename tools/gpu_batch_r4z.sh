#!/bin/bash
# round 4, GPU batch z: the host plans band by band (and faster): parity, steps at C3 / C4 shape
set -x
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r4z
mkdir -p "$O"
timeout 1200 python -m pytest tests/test_gpu_compare.py tests/test_gpu_fuzz.py tests/test_gpu_multirank.py -x -q > $O/pytest_part.log 2>&1; echo "rc $?" >> $O/pytest_part.log; tail -3 $O/pytest_part.log
: > $O/step_options.jsonl
SETS=';' REPS=30 timeout 300 python tools/step_options.py >> $O/step_options.jsonl 2>> $O/step_options.err
N=100000 P=10 SETS=';' REPS=3 timeout 300 python tools/step_options.py >> $O/step_options.jsonl 2>> $O/step_options.err
N=30000 P=12 SETS=';cum_budget_bytes=1073741824' REPS=3 timeout 300 python tools/step_options.py >> $O/step_options.jsonl 2>> $O/step_options.err
cat $O/step_options.jsonl
timeout 600 python bench.py --no-pmc --no-cpu-baseline > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r4z/bench.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["step"])
c = d["configs"][2]; print(c["ms_per_step"], c["kernel_ms"])
c = d["configs"][3]; print(c["ms_band"], c["kernel_ms"], c["extrapolated_full_matrix_s"])
PY

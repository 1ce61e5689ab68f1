#!/bin/bash
# round 4, GPU batch w: k_selfhist_card with lane-private counters: parity (all p), kernel stats
set -x
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r4w
mkdir -p "$O"
timeout 900 python -m pytest tests/test_gpu_compare.py tests/test_gpu_fuzz.py tests/test_gpu_sketch.py -x -q > $O/pytest_part.log 2>&1; echo "rc $?" >> $O/pytest_part.log; tail -3 $O/pytest_part.log
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt -o p --output-format csv -- python "$GRAFT_REPO_ROOT/bench.py" --no-cpu-baseline --no-secondary --no-pmc --steps 10 > "$GRAFT_REPO_ROOT/$O/bench_under_rocprof.json" 2> "$GRAFT_REPO_ROOT/$O/bench_under_rocprof.err")
f=$(find /tmp/kt -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" $O/kernel_stats.csv
python - <<'PY'
import csv
for r in csv.reader(open("gpurun_out/r4w/kernel_stats.csv")):
    print(r[0][:45], r[1:4])
PY
: > $O/step_options.jsonl
SETS=';' REPS=20 timeout 300 python tools/step_options.py >> $O/step_options.jsonl 2>> $O/step_options.err
N=100000 P=10 SETS=';' REPS=3 timeout 300 python tools/step_options.py >> $O/step_options.jsonl 2>> $O/step_options.err
cat $O/step_options.jsonl

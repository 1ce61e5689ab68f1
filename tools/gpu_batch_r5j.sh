#!/bin/bash
# round 4, GPU batch 5j: k_selfhist_card at 5 waves per SIMD (12 of a p=14 sketch's 16 chunks cached in registers)
set -x
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r5j
mkdir -p "$O"
timeout 600 python -m pytest tests/test_gpu_compare.py -x -q -k "card or tri_vs_oracle or adversarial or extreme or constant" > $O/pytest_part.log 2>&1; echo "rc $?" >> $O/pytest_part.log; tail -3 $O/pytest_part.log
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt -o p --output-format csv -- python "$GRAFT_REPO_ROOT/bench.py" --no-cpu-baseline --no-secondary --no-pmc --steps 10 > "$GRAFT_REPO_ROOT/$O/bench_under_rocprof.json" 2> "$GRAFT_REPO_ROOT/$O/bench_under_rocprof.err")
f=$(find /tmp/kt -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" $O/kernel_stats.csv
python - <<'PY'
import csv
for r in csv.reader(open("gpurun_out/r5j/kernel_stats.csv")):
    print(r[0][:45], r[1:4])
PY
SETS=';' REPS=20 timeout 300 python tools/step_options.py

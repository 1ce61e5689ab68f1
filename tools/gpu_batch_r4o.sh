#!/bin/bash
# round 4, GPU batch o: k_selfhist_card with bank-spread copies; kernel stats of the bench command
set -x
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r4o
mkdir -p "$O"
timeout 600 python -m pytest tests/test_gpu_compare.py -x -q -k "card or tri_vs_oracle or adversarial or extreme" > $O/pytest_part.log 2>&1; echo "rc $?" >> $O/pytest_part.log; tail -3 $O/pytest_part.log
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt -o p --output-format csv -- python "$GRAFT_REPO_ROOT/bench.py" --no-cpu-baseline --no-secondary --no-pmc --steps 10 > "$GRAFT_REPO_ROOT/$O/bench_under_rocprof.json" 2> "$GRAFT_REPO_ROOT/$O/bench_under_rocprof.err")
f=$(find /tmp/kt -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" $O/kernel_stats.csv && cut -d, -f1-4 $O/kernel_stats.csv | cut -c1-50,150-400 | head -12

#!/bin/bash
# round 4, GPU batch n: plane words without the multiply (k_transform): parity, prepare time
set -x
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r4n
mkdir -p "$O"
timeout 900 python -m pytest tests/test_gpu_compare.py tests/test_gpu_fuzz.py tests/test_gpu_multirank.py -x -q > $O/pytest_part.log 2>&1; echo "rc $?" >> $O/pytest_part.log; tail -4 $O/pytest_part.log
: > $O/step_options.jsonl
SETS=';' REPS=20 timeout 300 python tools/step_options.py >> $O/step_options.jsonl 2>> $O/step_options.err
N=30000 P=10 REPS=5 SETS=';' timeout 300 python tools/step_options.py >> $O/step_options.jsonl 2>> $O/step_options.err
cat $O/step_options.jsonl; tail -3 $O/step_options.err
cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o kt -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-secondary --no-pmc --steps 5 > $GRAFT_REPO_ROOT/$O/bench_under_rocprof.json 2> $GRAFT_REPO_ROOT/$O/bench_under_rocprof.err
cd $GRAFT_REPO_ROOT
f=$(find $O/prof -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" $O/kernel_stats.csv && head -8 $O/kernel_stats.csv | cut -c1-60,200-400
find gpurun_out/r4n/prof -type f -delete

#!/bin/bash
# round 4, GPU batch y: host planning times next to the (now shorter) prepare kernels
set -x
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r4y
mkdir -p "$O"
: > $O/step_options.jsonl
SETS=';;' REPS=30 timeout 300 python tools/step_options.py >> $O/step_options.jsonl 2>> $O/step_options.err
N=4000 SETS=';' REPS=30 timeout 300 python tools/step_options.py >> $O/step_options.jsonl 2>> $O/step_options.err
cat $O/step_options.jsonl
for i in 1 2 3; do timeout 300 python bench.py --no-secondary --no-pmc --no-cpu-baseline --steps 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['step'])"; done

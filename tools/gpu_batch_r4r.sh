#!/bin/bash
# round 4, GPU batch r: k_sketch with four k-mers per trip (batched filter reads): bit-exactness, bases/s
set -x
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r4r
mkdir -p "$O"
timeout 900 python -m pytest tests/test_gpu_sketch.py -x -q > $O/pytest_sketch.log 2>&1; echo "rc $?" >> $O/pytest_sketch.log; tail -3 $O/pytest_sketch.log
: > $O/bench_sketch.jsonl
timeout 300 python tools/bench_sketch.py --steps 5 --cpu-genomes 4 >> $O/bench_sketch.jsonl 2>> $O/bench_sketch.err
timeout 300 python tools/bench_sketch.py --steps 5 --cpu-genomes 2 --p 14 >> $O/bench_sketch.jsonl 2>> $O/bench_sketch.err
timeout 300 python tools/bench_sketch.py --steps 5 --cpu-genomes 2 --k 21 >> $O/bench_sketch.jsonl 2>> $O/bench_sketch.err
cat $O/bench_sketch.jsonl | cut -c1-700
tail -3 $O/bench_sketch.err

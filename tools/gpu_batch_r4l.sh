#!/bin/bash
# round 4, GPU batch l: does k_finalize hide under the tile kernel (two contexts side by side)?
set -x
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r4l
mkdir -p "$O"
: > $O/overlap_probe.jsonl
timeout 300 python tools/overlap_probe.py >> $O/overlap_probe.jsonl 2>> $O/overlap_probe.err
N=30000 P=10 STEPS=5 timeout 300 python tools/overlap_probe.py >> $O/overlap_probe.jsonl 2>> $O/overlap_probe.err
cat $O/overlap_probe.jsonl; tail -3 $O/overlap_probe.err

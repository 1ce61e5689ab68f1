#!/bin/bash
# round 4, GPU batch s: A/B of k_sketch (four k-mers per trip vs one), same box, same tool
set -x
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r4s
mkdir -p "$O"
: > $O/bench_sketch_new.jsonl
: > $O/bench_sketch_old.jsonl
for i in 1 2; do timeout 300 python tools/bench_sketch.py --steps 10 --cpu-genomes 1 >> $O/bench_sketch_new.jsonl 2>> $O/bench_sketch.err; done
timeout 300 python tools/bench_sketch.py --steps 10 --cpu-genomes 1 --p 14 >> $O/bench_sketch_new.jsonl 2>> $O/bench_sketch.err
cp dashing_amd/csrc/kernels_sketch.hip /tmp/kernels_sketch_new.hip
cp gpurun_ab/kernels_sketch_old.hip dashing_amd/csrc/kernels_sketch.hip
(cd dashing_amd/csrc && make 2>&1 | tail -2)
for i in 1 2; do timeout 300 python tools/bench_sketch.py --steps 10 --cpu-genomes 1 >> $O/bench_sketch_old.jsonl 2>> $O/bench_sketch.err; done
timeout 300 python tools/bench_sketch.py --steps 10 --cpu-genomes 1 --p 14 >> $O/bench_sketch_old.jsonl 2>> $O/bench_sketch.err
python - <<'PY'
import json
for tag in ("new", "old"):
    for l in open("gpurun_out/r4s/bench_sketch_%s.jsonl" % tag):
        if l.startswith("{"):
            r = json.loads(l); print(tag, r["metric"][:40], "%.4g" % r["value"], r["ms_per_step"], r["registers_bit_exact"])
PY

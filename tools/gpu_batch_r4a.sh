#!/bin/bash
# round 4, GPU batch a: the refactored library under the full GPU suite, k_finalize phase stamps, MLE variants, per-rank
# breakdown, the new bench line
set -x
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r4a
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.log
tail -5 $O/pytest_gpu.log
timeout 120 tools/ubench/div_accuracy > $O/div_accuracy.json 2>&1
cat $O/div_accuracy.json
rm -f $O/finalize_phases.jsonl
timeout 600 python tools/finalize_probe.py --workloads C3,C4 --out $O/finalize_phases.jsonl > $O/finalize_probe.log 2>&1
tail -3 $O/finalize_probe.log
GS=1,8 timeout 300 python tools/shard_breakdown.py > $O/shard_breakdown_c3.jsonl 2>&1
GS=8 NPARTS=8 timeout 300 python tools/shard_breakdown.py >> $O/shard_breakdown_c3.jsonl 2>&1
N=100000 P=10 GS=1,8 timeout 400 python tools/shard_breakdown.py > $O/shard_breakdown_c4.jsonl 2>&1
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
tail -c 1500 $O/bench.err
python - <<'PY'
import json
l=json.loads([x for x in open("gpurun_out/r4a/bench.json") if x.startswith("{")][-1])
print("value", l.get("value"), "ms", l.get("ms_per_step"), "err", l.get("error"))
print("binding", l.get("roofline",{}).get("binding"))
print("finalize", l.get("roofline",{}).get("finalize",{}).get("binding"))
for c in l.get("configs") or []:
    print(json.dumps({k: c.get(k) for k in ("workload","error","pairs_per_s","bases_per_s","ms_per_step","kernel_ms","parity")})[:600])
PY

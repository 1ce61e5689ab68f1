#!/bin/bash
# round 4, GPU batch 5i: run-to-run spread of the default bench line (5 runs, PMC passes and CPU baseline off)
set -x
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r5i
mkdir -p "$O"
: > $O/bench_runs.jsonl
for i in 1 2 3 4 5; do timeout 300 python bench.py --no-pmc --no-cpu-baseline --no-secondary 2>/dev/null | grep "^{" >> $O/bench_runs.jsonl; done
python - <<'PY'
import json
rows = [json.loads(l) for l in open("gpurun_out/r5i/bench_runs.jsonl")]
ms = sorted(r["ms_per_step"] for r in rows)
print("ms_per_step", [round(x, 3) for x in ms], "pairs/s", [round(r["value"] / 1e9, 4) for r in rows])
print([r["roofline"]["step"]["ms"] for r in rows])
PY

#!/bin/bash
# round 4, GPU batch u: the 8/4/2-rank model on the final sources; the new option test
set -x
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r4u
mkdir -p "$O"
timeout 600 python -m pytest tests/test_gpu_compare.py -x -q -k "record_width or options_do_not" > $O/pytest_part.log 2>&1; echo "rc $?" >> $O/pytest_part.log; tail -3 $O/pytest_part.log
: > $O/shard_model_c3.jsonl
for g in 8 4 2; do G=$g NPARTS=8 timeout 300 python tools/shard_model.py >> $O/shard_model_c3.jsonl 2>> $O/shard_model.err; done
G=8 NPARTS=4 timeout 300 python tools/shard_model.py >> $O/shard_model_c3.jsonl 2>> $O/shard_model.err
python - <<'PY'
import json
for l in open("gpurun_out/r4u/shard_model_c3.jsonl"):
    if not l.startswith("{"): continue
    r = json.loads(l)
    print("G", r["G"], "nparts", r["nparts"], "single", r["single_gpu_ms"], "max", r["max_rank_wall_ms"], "mean", r["mean_rank_wall_ms"], "same", r["assembled_equals_single_gpu"], r["exchange_model"])
    print("   ", [(x["wall_ms"], x["prepare_ms"], x["pair_ms"], x["finalize_ms"], x["parts"]) for x in r["ranks"]])
PY
timeout 300 python bench.py --no-secondary --steps 10 > $O/bench_quick.json 2> $O/bench_quick.err; python - <<'PY'
import json
d = json.load(open("gpurun_out/r4u/bench_quick.json"))
print(d["value"], d["ms_per_step"], d["configs"][0]["roofline"]["binding"])
PY

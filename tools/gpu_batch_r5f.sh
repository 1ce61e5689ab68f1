#!/bin/bash
# round 4, GPU batch 5f: k_finalize launches of the parts on two streams (option): protocol tests, 8-rank model
set -x
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r5f
mkdir -p "$O"
make -s -C tests/mock_rccl
timeout 1500 python -m pytest tests/test_gpu_multirank.py -x -q -k "two_streams or virtual_ranks or short_ranges" > $O/pytest_part.log 2>&1; echo "rc $?" >> $O/pytest_part.log; tail -5 $O/pytest_part.log
: > $O/shard_model_c3.jsonl
for o in finalize_two_streams=0 finalize_two_streams=1 finalize_two_streams=0 finalize_two_streams=1; do
  G=8 NPARTS=8 OPTS=$o timeout 300 python tools/shard_model.py >> $O/shard_model_c3.jsonl 2>> $O/shard_model.err
done
G=4 NPARTS=8 OPTS=finalize_two_streams=1 timeout 300 python tools/shard_model.py >> $O/shard_model_c3.jsonl 2>> $O/shard_model.err
python - <<'PY'
import json
for l in open("gpurun_out/r5f/shard_model_c3.jsonl"):
    if not l.startswith("{"): continue
    r = json.loads(l)
    print("G", r["G"], "nparts", r["nparts"], r["opts"], "single", r["single_gpu_ms"], "max", r["max_rank_wall_ms"], "mean", r["mean_rank_wall_ms"], "same", r["assembled_equals_single_gpu"], r["exchange_model"])
    print("   ", [(x["wall_ms"], x["pair_ms"], x["finalize_ms"], x["parts"]) for x in r["ranks"]])
PY

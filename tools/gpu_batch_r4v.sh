#!/bin/bash
# round 4, GPU batch v: would band cuts per ~400 tiles help the 2- and 4-rank steps? (existing knobs only)
set -x
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r4v
mkdir -p "$O"
: > $O/shard_model_c3.jsonl
G=2 NPARTS=4 OPTS=part_band_tiles=300 timeout 300 python tools/shard_model.py >> $O/shard_model_c3.jsonl 2>> $O/shard_model.err
G=2 NPARTS=4 timeout 300 python tools/shard_model.py >> $O/shard_model_c3.jsonl 2>> $O/shard_model.err
G=2 NPARTS=2 OPTS=part_band_tiles=300 timeout 300 python tools/shard_model.py >> $O/shard_model_c3.jsonl 2>> $O/shard_model.err
G=4 NPARTS=2 OPTS=part_band_tiles=300 timeout 300 python tools/shard_model.py >> $O/shard_model_c3.jsonl 2>> $O/shard_model.err
G=4 NPARTS=2 timeout 300 python tools/shard_model.py >> $O/shard_model_c3.jsonl 2>> $O/shard_model.err
python - <<'PY'
import json
for l in open("gpurun_out/r4v/shard_model_c3.jsonl"):
    if not l.startswith("{"): continue
    r = json.loads(l)
    print("G", r["G"], "nparts", r["nparts"], r["opts"], "single", r["single_gpu_ms"], "max", r["max_rank_wall_ms"], "mean", r["mean_rank_wall_ms"], "same", r["assembled_equals_single_gpu"], r["exchange_model"])
    print("   ", [(x["wall_ms"], x["prepare_ms"], x["pair_ms"], x["finalize_ms"], x["parts"], x["bands"]) for x in r["ranks"]])
PY

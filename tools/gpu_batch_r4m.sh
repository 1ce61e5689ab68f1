#!/bin/bash
# round 4, GPU batch m: tail fragments of the tile kernel (ls_tail_frag): parity, single-GPU step, 8-rank model
set -x
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r4m
mkdir -p "$O"
timeout 900 python -m pytest tests/test_gpu_compare.py tests/test_gpu_fuzz.py -x -q > $O/pytest_part.log 2>&1; echo "rc $?" >> $O/pytest_part.log; tail -4 $O/pytest_part.log
: > $O/step_options.jsonl
SETS='ls_tail_frag=0;ls_tail_frag=2;ls_tail_frag=4;ls_tail_frag=8;ls_tail_frag=16;ls_tail_frag=0' timeout 300 python tools/step_options.py >> $O/step_options.jsonl 2>> $O/step_options.err
N=4000 SETS='ls_tail_frag=0;ls_tail_frag=4;ls_tail_frag=8' timeout 300 python tools/step_options.py >> $O/step_options.jsonl 2>> $O/step_options.err
N=20000 P=12 REPS=4 SETS='ls_tail_frag=0;ls_tail_frag=4' timeout 300 python tools/step_options.py >> $O/step_options.jsonl 2>> $O/step_options.err
cat $O/step_options.jsonl; tail -3 $O/step_options.err
: > $O/shard_model_c3.jsonl
for o in ls_tail_frag=0 ls_tail_frag=4 ls_tail_frag=8 ls_tail_frag=4,part_band_tiles=128 ls_tail_frag=8,part_band_tiles=64 ls_tail_frag=4,part_band_tiles=200; do
  G=8 NPARTS=8 OPTS=$o timeout 300 python tools/shard_model.py >> $O/shard_model_c3.jsonl 2>> $O/shard_model.err
done
G=8 NPARTS=4 OPTS=ls_tail_frag=4,part_band_tiles=128 timeout 300 python tools/shard_model.py >> $O/shard_model_c3.jsonl 2>> $O/shard_model.err
python - <<'PY'
import json
for l in open("gpurun_out/r4m/shard_model_c3.jsonl"):
    if not l.startswith("{"): continue
    r = json.loads(l)
    print("G", r["G"], "nparts", r["nparts"], r.get("opts"), "single", r["single_gpu_ms"], "max", r["max_rank_wall_ms"], "mean", r["mean_rank_wall_ms"], "same", r["assembled_equals_single_gpu"], r["exchange_model"])
    print("   ", [(x["wall_ms"], x["pair_ms"], x["finalize_ms"], x["bands"], x["parts"]) for x in r["ranks"]])
PY
tail -5 $O/shard_model.err

#!/bin/bash
# round 4, GPU batch f: exchange pair with row-sorted parts; trimmed MLE setup; 8-rank model
set -x
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r4f
mkdir -p "$O"
timeout 900 python -m pytest tests/test_gpu_multirank.py tests/test_gpu_compare.py tests/test_gpu_fuzz.py -x -q > $O/pytest_part.log 2>&1; echo "rc $?" >> $O/pytest_part.log; tail -4 $O/pytest_part.log
: > $O/shard_model_c3.jsonl
: > $O/finalize_instr.jsonl
: > $O/finalize_phases.jsonl
for np_ in 8 4; do G=8 NPARTS=$np_ timeout 300 python tools/shard_model.py >> $O/shard_model_c3.jsonl 2>> $O/shard_model.err; done
G=4 NPARTS=8 timeout 300 python tools/shard_model.py >> $O/shard_model_c3.jsonl 2>> $O/shard_model.err
G=2 NPARTS=8 timeout 300 python tools/shard_model.py >> $O/shard_model_c3.jsonl 2>> $O/shard_model.err
python - <<'PY'
import json
for l in open("gpurun_out/r4f/shard_model_c3.jsonl"):
    if not l.startswith("{"): continue
    r = json.loads(l)
    print("G", r["G"], "nparts", r["nparts"], "single", r["single_gpu_ms"], "max", r["max_rank_wall_ms"], "mean", r["mean_rank_wall_ms"], "place", r["dst_place_all_sources_ms"], "same", r["assembled_equals_single_gpu"], r["exchange_model"])
    print("   ", [(x["wall_ms"], x["planes_per_tile"], x["rowsorted"], x["parts"]) for x in r["ranks"]])
PY
tail -5 $O/shard_model.err
timeout 600 python tools/finalize_instr.py --workloads C4,C3 --out $O/finalize_instr.jsonl > $O/finalize_instr.log 2>&1
timeout 600 python tools/finalize_probe.py --workloads C3,C4 --out $O/finalize_phases.jsonl > $O/finalize_probe.log 2>&1
python - <<'PY'
import json
for l in open("gpurun_out/r4f/finalize_instr.jsonl"):
    r = json.loads(l); print(r.get("workload"), r.get("estim"), r.get("finalize_stop"), r.get("per_wave"), r.get("error"))
for l in open("gpurun_out/r4f/finalize_phases.jsonl"):
    r = json.loads(l)
    if r["layout"] == "sort1": print(r["workload"], r["estim"], "fin", r["finalize_ms"], "pair", r["pair_ms"], r["phase_ms_of_kernel"])
PY

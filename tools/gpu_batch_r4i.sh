#!/bin/bash
# round 4, GPU batch h: epilogue loads requested before the estimator, in-band list rows
set -x
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r4i
mkdir -p "$O"
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_part.log 2>&1; echo "rc $?" >> $O/pytest_part.log; tail -4 $O/pytest_part.log
: > $O/finalize_instr.jsonl
: > $O/finalize_phases.jsonl
timeout 600 python tools/finalize_instr.py --workloads C4,C3 --out $O/finalize_instr.jsonl > $O/finalize_instr.log 2>&1
timeout 600 python tools/finalize_probe.py --workloads C3,C4 --out $O/finalize_phases.jsonl > $O/finalize_probe.log 2>&1
python - <<'PY'
import json
for l in open("gpurun_out/r4i/finalize_instr.jsonl"):
    r = json.loads(l); print(r.get("workload"), r.get("estim"), r.get("finalize_stop"), r.get("per_wave"), r.get("error"))
for l in open("gpurun_out/r4i/finalize_phases.jsonl"):
    r = json.loads(l)
    print(r["workload"], r["layout"], r["estim"], "fin", r["finalize_ms"], "pair", r["pair_ms"], r["phase_ms_of_kernel"])
PY
: > $O/shard_model_c3.jsonl
G=8 NPARTS=8 timeout 300 python tools/shard_model.py >> $O/shard_model_c3.jsonl 2>> $O/shard_model.err
G=8 NPARTS=4 OPTS=part_band_tiles=64 timeout 300 python tools/shard_model.py >> $O/shard_model_c3.jsonl 2>> $O/shard_model.err
G=8 NPARTS=8 OPTS=part_band_tiles=32 timeout 300 python tools/shard_model.py >> $O/shard_model_c3.jsonl 2>> $O/shard_model.err
python - <<'PY'
import json
for l in open("gpurun_out/r4i/shard_model_c3.jsonl"):
    if not l.startswith("{"): continue
    r = json.loads(l)
    print("G", r["G"], "nparts", r["nparts"], "single", r["single_gpu_ms"], "max", r["max_rank_wall_ms"], "mean", r["mean_rank_wall_ms"], "same", r["assembled_equals_single_gpu"], r["exchange_model"])
    print("   ", [(x["wall_ms"], x["bands"], x["parts"], x["pair_ms"], x["finalize_ms"]) for x in r["ranks"]])
PY

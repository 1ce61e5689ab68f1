#!/bin/bash
# round 4, GPU batch f: exchange pair with row-sorted parts; trimmed MLE setup; 8-rank model
set -x
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r4g
mkdir -p "$O"
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_part.log 2>&1; echo "rc $?" >> $O/pytest_part.log; tail -4 $O/pytest_part.log
: > $O/shard_model_c3.jsonl
: > $O/finalize_instr.jsonl
: > $O/finalize_phases.jsonl
for np_ in 4; do G=8 NPARTS=$np_ timeout 300 python tools/shard_model.py >> $O/shard_model_c3.jsonl 2>> $O/shard_model.err; done
python - <<'PY'
import json
for l in open("gpurun_out/r4g/shard_model_c3.jsonl"):
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
for l in open("gpurun_out/r4g/finalize_instr.jsonl"):
    r = json.loads(l); print(r.get("workload"), r.get("estim"), r.get("finalize_stop"), r.get("per_wave"), r.get("error"))
for l in open("gpurun_out/r4g/finalize_phases.jsonl"):
    r = json.loads(l)
    if r["layout"] == "sort1": print(r["workload"], r["estim"], "fin", r["finalize_ms"], "pair", r["pair_ms"], r["phase_ms_of_kernel"])
PY
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
python - <<'PY'
import json
l=json.loads([x for x in open("gpurun_out/r4g/bench.json") if x.startswith("{")][-1])
print("value", l.get("value"), "ms", l.get("ms_per_step"), "err", l.get("error"))
print("step", l.get("roofline",{}).get("step"))
print("finalize", l.get("roofline",{}).get("finalize",{}).get("binding"))
for c in l.get("configs") or []:
    print(json.dumps({k: c.get(k) for k in ("workload","error","pairs_per_s","bases_per_s","ms_per_step","kernel_ms","parity")})[:500])
PY

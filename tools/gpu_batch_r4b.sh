#!/bin/bash
# round 4, GPU batch b: sampled k_finalize phase stamps, XCD-tile mapping A/B, sketch parity debug, parts vs planes
set -x
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r4b
mkdir -p $O
timeout 300 python tools/debug_sketch_parity.py > $O/debug_sketch.log 2>&1
grep "^{" $O/debug_sketch.log
rm -f $O/finalize_phases*.jsonl
timeout 600 python tools/finalize_probe.py --workloads C3,C4 --out $O/finalize_phases.jsonl > $O/finalize_probe.log 2>&1
timeout 600 python tools/finalize_probe.py --workloads C3,C4 --opts finalize_xcd_tiles=0 --out $O/finalize_phases_noxcd.jsonl > $O/finalize_probe_noxcd.log 2>&1
python - <<'PY'
import json
for f in ("finalize_phases", "finalize_phases_noxcd"):
    for l in open("gpurun_out/r4b/%s.jsonl" % f):
        r = json.loads(l)
        print(f, r["workload"], r["layout"], r["estim"], "fin", r["finalize_ms"], "stamped", r["finalize_ms_stamped"], r["phase_ms_of_kernel"], "cyc/wave", r["cycles_per_wave"])
PY
for np_ in 1 2 3 8; do GS=8 NPARTS=$np_ timeout 300 python tools/shard_breakdown.py >> $O/shard_breakdown_c3_parts.jsonl 2>&1; done
python - <<'PY'
import json
for l in open("gpurun_out/r4b/shard_breakdown_c3_parts.jsonl"):
    if not l.startswith("{"): continue
    r = json.loads(l)
    print("nparts", r["nparts"], "max", r["max_wall_ms"], "mean", r["mean_wall_ms"], [(x["wall_ms"], x["planes_per_tile"]) for x in r["ranks"]])
PY
timeout 900 python -m pytest tests/test_gpu_compare.py tests/test_gpu_multirank.py -x -q > $O/pytest_part.log 2>&1; tail -3 $O/pytest_part.log

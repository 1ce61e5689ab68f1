#!/bin/bash
# round 4, GPU batch 5d: the driver's N = 8 command over the stand-in transport (8 ranks on one GPU): C3, and the configs[3]
# shape with --exchange auto (every rank keeps its span + the gathered variant)
set -x
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r5d
mkdir -p "$O"
make -s -C tests/mock_rccl
export DSH_BENCH_BACKEND=gloo DSH_BENCH_EXCHANGE=cabi-mock DSH_RCCL_LIB=$PWD/tests/mock_rccl/libmock_rccl.so MOCK_RCCL_TIMEOUT_S=600 DSH_COMM_TIMEOUT_S=900
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29555 bench.py --gpus 8 --steps 3 --warmup 1 > $O/bench_gpus8_c3.out 2> $O/bench_gpus8_c3.err
grep "^{" $O/bench_gpus8_c3.out > $O/bench_gpus8_c3.json; python - <<'PY'
import json
d = json.load(open("gpurun_out/r5d/bench_gpus8_c3.json"))
print(d["n_gpus"], d["ms_per_step"], d["parity_vs_cpu"], d["multi_gpu"]["exchange"], d["multi_gpu"]["row_bounds"])
PY
tail -3 $O/bench_gpus8_c3.err
DSH_BENCH_N=100000 DSH_BENCH_P=10 timeout 1500 python bench.py --gpus 4 --steps 1 --warmup 1 > $O/bench_gpus4_c4.out 2> $O/bench_gpus4_c4.err
grep "^{" $O/bench_gpus4_c4.out > $O/bench_gpus4_c4.json; python - <<'PY'
import json
d = json.load(open("gpurun_out/r5d/bench_gpus4_c4.json"))
print(d["n_gpus"], d["ms_per_step"], d["parity_vs_cpu"]); mg = d["multi_gpu"]; print(mg["exchange"]); print(mg["gathered_variant"])
PY
tail -3 $O/bench_gpus4_c4.err

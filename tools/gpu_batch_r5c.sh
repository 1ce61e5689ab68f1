#!/bin/bash
# round 4, GPU batch 5c: bench.py --gpus 2 over the stand-in transport; all multirank tests
set -x
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r5c
mkdir -p "$O"
make -s -C tests/mock_rccl
timeout 1500 python -m pytest tests/test_gpu_multirank.py -x -q > $O/pytest_multirank.log 2>&1; echo "rc $?" >> $O/pytest_multirank.log; tail -30 $O/pytest_multirank.log
DSH_BENCH_BACKEND=gloo DSH_BENCH_EXCHANGE=cabi-mock DSH_RCCL_LIB=$PWD/tests/mock_rccl/libmock_rccl.so timeout 600 python bench.py --gpus 4 --steps 3 --warmup 1 > $O/bench_gpus4_mock.json 2> $O/bench_gpus4_mock.err; tail -c 1500 $O/bench_gpus4_mock.json; tail -5 $O/bench_gpus4_mock.err

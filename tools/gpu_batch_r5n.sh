#!/bin/bash
# round 4, GPU batch 5n: CLI collect over the stand-in transport after the fix of the stand-in's destroy; protocol tests again
set -x
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r5n
mkdir -p "$O"
make -s -C tests/mock_rccl
timeout 900 python -m pytest tests/test_gpu_cli.py -x -q -k "stand_in or rccl_collect or multi_device" > $O/pytest_cli.log 2>&1; echo "rc $?" >> $O/pytest_cli.log; tail -4 $O/pytest_cli.log
timeout 1500 python -m pytest tests/test_gpu_multirank.py -x -q -k "protocol or stand_in" > $O/pytest_protocol.log 2>&1; echo "rc $?" >> $O/pytest_protocol.log; tail -4 $O/pytest_protocol.log
ls /dev/shm | head -5

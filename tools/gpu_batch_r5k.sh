#!/bin/bash
# round 4, GPU batch 5k: the CLI's multi-device collect path (threads as ranks) over the stand-in transport
set -x
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r5k
mkdir -p "$O"
make -s -C tests/mock_rccl
timeout 900 python -m pytest tests/test_gpu_cli.py -x -q -k "stand_in or rccl_collect or multi_device" > $O/pytest_cli.log 2>&1; echo "rc $?" >> $O/pytest_cli.log; tail -25 $O/pytest_cli.log

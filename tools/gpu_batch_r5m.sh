#!/bin/bash
# round 4, GPU batch 5m: debug of the CLI collect test under pytest (stand-in transport)
set -x
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r5m
mkdir -p "$O"
make -s -C tests/mock_rccl
MOCK_RCCL_DEBUG=1 DSH_COMM_INIT_TIMEOUT_S=20 timeout 600 python -m pytest tests/test_gpu_cli.py -x -q -k "stand_in" > $O/pytest_cli.log 2>&1; echo "rc $?" >> $O/pytest_cli.log
grep -n "mock_rccl\|dashing-amd\]\|passed\|failed" $O/pytest_cli.log | cut -c1-300 | head -40

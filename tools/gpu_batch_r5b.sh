#!/bin/bash
# round 4, GPU batch 5b: the exchange protocol between processes over the stand-in transport (tests/mock_rccl)
set -x
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r5b
mkdir -p "$O"
make -s -C tests/mock_rccl
timeout 1500 python -m pytest tests/test_gpu_multirank.py -x -q -k "protocol" > $O/pytest_protocol.log 2>&1; echo "rc $?" >> $O/pytest_protocol.log; tail -30 $O/pytest_protocol.log

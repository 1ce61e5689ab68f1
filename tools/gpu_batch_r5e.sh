#!/bin/bash
# round 4, GPU batch 5e: soak of the final sources -- fresh fuzz seeds, big cases
set -x
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r5e
mkdir -p "$O"
DSH_FUZZ_FIRST=40000 DSH_FUZZ_CASES=6000 DSH_FUZZ_BIG_CASES=120 timeout 2400 python -m pytest tests/test_gpu_fuzz.py -x -q > $O/pytest_fuzz_soak.log 2>&1; echo "rc $?" >> $O/pytest_fuzz_soak.log; tail -3 $O/pytest_fuzz_soak.log

#!/bin/bash
# round 4, GPU batch k: soak of the rewritten k_finalize -- fresh fuzz seeds, big cases, the new option test
set -x
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r4k
mkdir -p "$O"
timeout 600 python -m pytest tests/test_gpu_compare.py -x -q > $O/pytest_compare.log 2>&1; echo "rc $?" >> $O/pytest_compare.log; tail -3 $O/pytest_compare.log
DSH_FUZZ_FIRST=12000 DSH_FUZZ_CASES=3000 DSH_FUZZ_BIG_CASES=60 timeout 1500 python -m pytest tests/test_gpu_fuzz.py -x -q > $O/pytest_fuzz_soak.log 2>&1; echo "rc $?" >> $O/pytest_fuzz_soak.log; tail -3 $O/pytest_fuzz_soak.log

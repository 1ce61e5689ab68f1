#!/bin/bash
# round 4, GPU batch 5g: the final sources -- full GPU suite, smoke, PMC passes (committed summary), the bench command under rocprofv3
set -x
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r5g
mkdir -p "$O"
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "rc $?" >> $O/pytest_gpu.log; tail -4 $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 900 python tools/pmc_collect.py --tag r5g_pmc --passes kt,fetch,write,sq1 > $O/pmc.log 2>&1
cp gpurun_out/r5g_pmc/kernel_stats.csv gpurun_out/r5g_pmc/pmc_summary.json gpurun_out/r5g_pmc/pmc_pair_kernel.json $O/ 2>/dev/null
mkdir -p /tmp/ktfull
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/ktfull -o p --output-format csv -- python "$GRAFT_REPO_ROOT/bench.py" > "$GRAFT_REPO_ROOT/$O/bench_under_rocprof.json" 2> "$GRAFT_REPO_ROOT/$O/bench_under_rocprof.err")
find /tmp/ktfull -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_full_bench.csv \;
head -12 $O/kernel_stats_full_bench.csv | cut -c1-60,250-330
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 600 $O/bench.json

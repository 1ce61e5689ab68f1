#!/bin/bash
# round 4, GPU batch j: list caps under the cheaper join (VERDICT item 7 experiment), rocprofv3 evidence of the bench command
set -x
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r4j
mkdir -p "$O"
timeout 600 python tools/list_cap_sweep.py > $O/list_cap_sweep.jsonl 2> $O/list_cap_sweep.err
cat $O/list_cap_sweep.jsonl
timeout 900 python tools/pmc_collect.py --tag r4j_pmc --passes kt,fetch,write,sq1 > $O/pmc.log 2>&1
cp gpurun_out/r4j_pmc/kernel_stats.csv gpurun_out/r4j_pmc/pmc_summary.json gpurun_out/r4j_pmc/pmc_pair_kernel.json $O/ 2>/dev/null
mkdir -p /tmp/ktfull
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/ktfull -o p --output-format csv -- python "$GRAFT_REPO_ROOT/bench.py" > "$GRAFT_REPO_ROOT/$O/bench_under_rocprof.json" 2> "$GRAFT_REPO_ROOT/$O/bench_under_rocprof.err")
find /tmp/ktfull -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_full_bench.csv \;
head -20 $O/kernel_stats_full_bench.csv

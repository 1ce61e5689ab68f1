#!/bin/bash
# end-of-round evidence: rocprofv3 kernel stats + PMC of the default bench command (stamped with the kernel-source
# hash) FIRST, so that the bench line that follows can quote the traffic of exactly these sources; then the default
# bench line, sketch kernel bench, 8 virtual ranks, unfriendly collections, C2/C4/C5 shapes, A/B and what-if tools
cd /root/repo
TAG=${1:-r2z}
O=gpurun_out/$TAG
mkdir -p $O
python tools/pmc_collect.py --tag ${TAG}_pmc --passes kt,fetch,write,sq1 > $O/pmc.log 2>&1
cp gpurun_out/${TAG}_pmc/kernel_stats.csv gpurun_out/${TAG}_pmc/pmc_summary.json gpurun_out/${TAG}_pmc/pmc_pair_kernel.json $O/ 2>/dev/null
cp gpurun_out/${TAG}_pmc/pmc_pair_kernel.json profiles/pmc_pair_kernel.json 2>/dev/null   # (on the box; copy it back by hand too)
python bench.py > $O/bench.json 2> $O/bench.err
python tools/lockstep_ab.py > $O/lockstep_ab.jsonl 2>/dev/null
python tools/estimator_timing.py > $O/estimator_timing.jsonl 2>/dev/null
python tools/shard_timing.py > $O/shard_timing.jsonl 2>/dev/null
python tools/bench_sketch.py > $O/bench_sketch.json 2>/dev/null
DSH_BENCH_FORCE_DIST=1 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-secondary > $O/bench_forced_rccl1.json 2>/dev/null
timeout 120 tools/ubench/pair_sched 2>/dev/null | grep -v amdgpu.ids > $O/pair_sched.txt
python tools/parse_bench.py > $O/parse_bench.jsonl 2>/dev/null
python tools/run_configs.py c3cli c3knn unfriendly > $O/configs.jsonl 2>$O/configs.err
python tools/run_configs.py c4 >> $O/configs.jsonl 2>>$O/configs.err
python tools/run_configs.py c5 >> $O/configs.jsonl 2>>$O/configs.err
python tools/run_configs.py c2 >> $O/configs.jsonl 2>>$O/configs.err
python tools/mfma_whatif.py > $O/mfma_whatif.jsonl 2>/dev/null
tail -c 600 $O/bench.json

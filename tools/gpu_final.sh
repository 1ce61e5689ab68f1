#!/bin/bash
# end-of-round evidence: default bench line, rocprofv3 kernel stats + PMC of the same command (stamped with the
# kernel-source hash), sketch kernel bench, 8 virtual ranks, C4/C5 shapes
cd /root/repo
TAG=${1:-r2z}
mkdir -p gpurun_out/$TAG
python bench.py > gpurun_out/$TAG/bench.json 2> gpurun_out/$TAG/bench.err
python tools/pmc_collect.py --tag ${TAG}_pmc --passes kt,fetch,write,sq1 > gpurun_out/$TAG/pmc.log 2>&1
cp gpurun_out/${TAG}_pmc/kernel_stats.csv gpurun_out/${TAG}_pmc/pmc_summary.json gpurun_out/${TAG}_pmc/pmc_pair_kernel.json gpurun_out/$TAG/ 2>/dev/null
python tools/shard_timing.py > gpurun_out/$TAG/shard_timing.jsonl 2>/dev/null
python tools/bench_sketch.py > gpurun_out/$TAG/bench_sketch.json 2>/dev/null
DSH_BENCH_FORCE_DIST=1 python bench.py --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/$TAG/bench_forced_rccl1.json 2>/dev/null
python tools/run_configs.py c3cli c3knn unfriendly > gpurun_out/$TAG/configs.jsonl 2>gpurun_out/$TAG/configs.err
python tools/run_configs.py c4 >> gpurun_out/$TAG/configs.jsonl 2>>gpurun_out/$TAG/configs.err
python tools/run_configs.py c5 >> gpurun_out/$TAG/configs.jsonl 2>>gpurun_out/$TAG/configs.err
python tools/run_configs.py c2 >> gpurun_out/$TAG/configs.jsonl 2>>gpurun_out/$TAG/configs.err
python tools/mfma_whatif.py > gpurun_out/$TAG/mfma_whatif.jsonl 2>/dev/null
tail -c 400 gpurun_out/$TAG/bench.json

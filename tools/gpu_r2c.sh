#!/bin/bash
# round-2 call: full GPU test suite on the range-layout build + bench (default run incl. cpu baseline + secondary)
# + a 2-rank gloo dry-run of the N>1 bench path on one GPU + the PMC passes that feed roofline.traffic
cd /root/repo
mkdir -p gpurun_out/r2c
python -m pytest tests -m gpu -x -q > gpurun_out/r2c/tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2c/tests.log
tail -5 gpurun_out/r2c/tests.log
python bench.py > gpurun_out/r2c/bench.json 2> gpurun_out/r2c/bench.err; tail -c 1500 gpurun_out/r2c/bench.err
DSH_BENCH_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 3 --warmup 1 > gpurun_out/r2c/bench_gloo2.json 2> gpurun_out/r2c/bench_gloo2.err
DSH_BENCH_FORCE_DIST=1 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r2c/bench_forced.json 2> gpurun_out/r2c/bench_forced.err
python tools/pmc_collect.py --tag r2c_c3 --passes kt,fetch,write > gpurun_out/r2c/pmc.log 2>&1
python tools/shard_timing.py > gpurun_out/r2c/shard_timing.jsonl 2> gpurun_out/r2c/shard_timing.err

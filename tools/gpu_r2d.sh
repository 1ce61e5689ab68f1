#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r2d
python -m pytest tests -m gpu -x -q > gpurun_out/r2d/tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2d/tests.log
tail -15 gpurun_out/r2d/tests.log
python tools/shard_timing.py > gpurun_out/r2d/shard_timing.jsonl 2> gpurun_out/r2d/shard_timing.err
python bench.py --no-secondary --cpu-seconds 5 > gpurun_out/r2d/bench.json 2> gpurun_out/r2d/bench.err
python tools/run_configs.py c3cli c1 > gpurun_out/r2d/c3cli.jsonl 2> gpurun_out/r2d/c3cli.err
head -c 600 gpurun_out/r2d/shard_timing.jsonl

#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r2f
python -m pytest tests -m gpu -x -q > gpurun_out/r2f/tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2f/tests.log
tail -15 gpurun_out/r2f/tests.log
python tools/finalize_phases.py > gpurun_out/r2f/finphases.jsonl 2> gpurun_out/r2f/finphases.err
cat gpurun_out/r2f/finphases.jsonl
python bench.py --cpu-seconds 3 > gpurun_out/r2f/bench.json 2> gpurun_out/r2f/bench.err

#!/bin/bash
# quick loop for k_finalize work: parity subset + phase timing
cd /root/repo
mkdir -p gpurun_out/fin
python -m pytest tests/test_gpu_compare.py tests/test_gpu_fuzz.py -m gpu -x -q > gpurun_out/fin/tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/fin/tests.log
tail -4 gpurun_out/fin/tests.log
python tools/finalize_phases.py > gpurun_out/fin/finphases.jsonl 2> gpurun_out/fin/finphases.err
cat gpurun_out/fin/finphases.jsonl

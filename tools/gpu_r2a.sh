#!/bin/bash
# first GPU call of round 2: scheduling ubench + PMC baselines of the round-1 kernels (C3 and a p=10 matrix)
cd /root/repo
mkdir -p gpurun_out/r2a
timeout 300 tools/ubench/pair_sched > gpurun_out/r2a/pair_sched.txt 2>&1
timeout 900 python tools/pmc_collect.py --tag r2a_c3 --passes kt,sq1,sq2,fetch,write > gpurun_out/r2a/c3.log 2>&1
timeout 900 python tools/pmc_collect.py --tag r2a_p10 --env DSH_BENCH_N=50000 --env DSH_BENCH_P=10 --passes kt,sq1,sq2 > gpurun_out/r2a/p10.log 2>&1
tail -30 gpurun_out/r2a/pair_sched.txt

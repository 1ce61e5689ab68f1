#!/bin/bash
cd /root/repo
for s in 1 2 3 4 0; do
  python tools/pmc_collect.py --tag finpmc_p10_s$s --env DSH_BENCH_N=40000 --env DSH_BENCH_P=10 --env DSH_BENCH_OPTS=finalize_stop=$s --passes sq1 > /dev/null 2>&1
  python tools/pmc_collect.py --tag finpmc_c3_s$s --env DSH_BENCH_OPTS=finalize_stop=$s --passes sq1 > /dev/null 2>&1
done
ls gpurun_out | grep finpmc

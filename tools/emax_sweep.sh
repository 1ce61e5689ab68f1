#!/bin/bash
# emax sweep (exception-list cap) on the bench workload and on a p=10 matrix: ms per step and the kernel split
cd /root/repo
for e in 64 80 96 112 128 160; do
  echo -n "C3 emax=$e: "; DSH_BENCH_OPTS=emax=$e python bench.py --no-cpu-baseline --no-secondary --steps 5 --warmup 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(round(d['ms_per_step'],3), d['roofline']['step']['ms'], d['roofline']['avg_planes_per_tile'])"
done
for e in 4 8 12 16 24 32 48; do
  echo -n "p10 n=40000 emax=$e: "; DSH_BENCH_N=40000 DSH_BENCH_P=10 DSH_BENCH_OPTS=emax=$e python bench.py --no-cpu-baseline --no-secondary --steps 3 --warmup 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(round(d['ms_per_step'],3), d['roofline']['step']['ms'], d['roofline']['avg_planes_per_tile'])"
done

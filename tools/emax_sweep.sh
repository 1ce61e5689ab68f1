#!/bin/bash
# emax sweep (exception-list cap) on the bench workload and on other precisions: ms per step and the kernel split
cd /root/repo
run() {  # n p emax steps
  DSH_BENCH_N=$1 DSH_BENCH_P=$2 DSH_BENCH_OPTS=emax=$3 python bench.py --no-cpu-baseline --no-secondary --steps $4 --warmup 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('n=$1 p=$2 emax=$3:', round(d['ms_per_step'],3), d['roofline']['step']['ms'], d['roofline']['avg_planes_per_tile'])"
}
for e in ${E1:-64 96 128 144 160 176 192}; do run 10000 14 $e 5; done
for e in ${E2:-48 64 96 128 192}; do run 20000 12 $e 3; done
for e in ${E2:-48 64 96 128 192}; do run 14000 13 $e 3; done
for e in ${E4:-96 128 192 255}; do run 7000 15 $e 3; done
for e in ${E4:-96 128 192 255}; do run 5000 16 $e 3; done
for e in ${E3:-4 8 12 16 24 32}; do run 40000 10 $e 3; done

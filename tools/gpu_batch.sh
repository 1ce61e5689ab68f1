#!/bin/bash
# ONE runner for the GPU batches of a round (replaces the 40 one-off tools/gpu_batch_r*.sh of rounds 1-4; what each of
# those ran is in profiles/MANIFEST.md).  Runs on the GPU box under gpurun:
#   gpurun --timeout 1500 -- 'bash tools/gpu_batch.sh <tag> <step> [<step> ...]'
# Outputs go to gpurun_out/<tag>/ (merged back into the repo's gpurun_out/); copy what should be judged to profiles/<tag>/.
# Steps (each bounded by its own timeout; a step that fails does not stop the batch):
#   suite             python -m pytest tests -m gpu -x -q
#   tests:<expr>      python -m pytest tests -m gpu -q -k '<expr>'
#   file:<path>       python -m pytest <path> -m gpu -x -q
#   smoke             __graft_entry__.smoke()
#   bench             python bench.py                                 -> bench.json
#   benchq            python bench.py --no-secondary --no-pmc         -> bench_quick.json
#   bench_rocprof     the default bench under rocprofv3 --kernel-trace --stats -> kernel_stats_full_bench.csv
#   pmc               tools/pmc_collect.py (kernel trace + FETCH/WRITE/SQ passes) -> kernel_stats.csv, pmc_summary.json
#   mock<N>           bench.py --gpus N over the stand-in transport (N ranks on the one GPU) -> bench_gpus<N>_mock.json
#   model[:ENV=..,..] tools/shard_model.py with the given environment (e.g. model:G=8,PARTITIONS=contiguous+rowsets)
#   py:<script>[:ENV=..,..]  python tools/<script>.py > <script>.jsonl
#   experiment        the layout-reuse experiment: EXPERIMENT=1 library vs product, quick bench + 8-rank model
set -x
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
export TMPDIR=/tmp
TAG=$1
shift
O=gpurun_out/$TAG
mkdir -p "$O"
envs() { echo "$1" | tr ',' ' ' | tr '+' ','; }   # "A=1,B=x+y" -> "A=1 B=x,y"
for step in "$@"; do
  name=${step%%:*}
  arg=""
  [ "$name" != "$step" ] && arg=${step#*:}
  case "$name" in
    suite) timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "rc $?" >> $O/pytest_gpu.log; tail -5 $O/pytest_gpu.log ;;
    tests) timeout 1800 python -m pytest tests -m gpu -q -k "$arg" > "$O/pytest_k.log" 2>&1; echo "rc $?" >> $O/pytest_k.log; tail -15 $O/pytest_k.log ;;
    file) f=$(basename "$arg" .py); timeout 1800 python -m pytest "$arg" -m gpu -x -q > "$O/pytest_$f.log" 2>&1; echo "rc $?" >> "$O/pytest_$f.log"; tail -15 "$O/pytest_$f.log" ;;
    smoke) timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log ;;
    bench) timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 1500 $O/bench.json; tail -3 $O/bench.err ;;
    benchq) timeout 600 python bench.py --no-secondary --no-pmc > $O/bench_quick.json 2> $O/bench_quick.err; tail -c 800 $O/bench_quick.json ;;
    bench_rocprof)
      R=$PWD; rm -rf /tmp/ktfull; mkdir -p /tmp/ktfull
      (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/ktfull -o p --output-format csv -- python "$R/bench.py" --no-pmc > "$R/$O/bench_under_rocprof.json" 2> "$R/$O/bench_under_rocprof.err")
      find /tmp/ktfull -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_full_bench.csv \;
      head -12 $O/kernel_stats_full_bench.csv | cut -c1-60,250-330 ;;
    pmc)
      timeout 900 python tools/pmc_collect.py --tag ${TAG}_pmc --passes kt,fetch,write,sq1 > $O/pmc.log 2>&1
      cp gpurun_out/${TAG}_pmc/kernel_stats.csv gpurun_out/${TAG}_pmc/pmc_summary.json gpurun_out/${TAG}_pmc/pmc_pair_kernel.json $O/ 2>/dev/null; tail -3 $O/pmc.log ;;
    mock*)
      N=${name#mock}
      make -s -C tests/mock_rccl
      ( export DSH_BENCH_BACKEND=gloo DSH_BENCH_EXCHANGE=cabi-mock DSH_RCCL_LIB=$PWD/tests/mock_rccl/libmock_rccl.so MOCK_RCCL_TIMEOUT_S=600 DSH_COMM_TIMEOUT_S=900 $(envs "$arg")
        timeout 1200 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29555 bench.py --gpus $N --steps 3 --warmup 1 > $O/bench_gpus${N}_mock.out 2> $O/bench_gpus${N}_mock.err )
      grep "^{" $O/bench_gpus${N}_mock.out > $O/bench_gpus${N}_mock.json; tail -c 1500 $O/bench_gpus${N}_mock.json; tail -3 $O/bench_gpus${N}_mock.err ;;
    model) ( export $(envs "$arg"); timeout 900 python tools/shard_model.py >> $O/shard_model.jsonl 2> $O/shard_model.err ); tail -c 600 $O/shard_model.jsonl; tail -3 $O/shard_model.err ;;
    py) s=${arg%%:*}; e=""; [ "$s" != "$arg" ] && e=${arg#*:}; ( export $(envs "$e"); timeout 1200 python tools/$s.py >> $O/$s.jsonl 2> $O/$s.err ); tail -c 1200 $O/$s.jsonl; tail -3 $O/$s.err ;;
    experiment)
      # the layout-reuse experiment: the library rebuilt with EXPERIMENT=1 ON THE GPU BOX, the quick bench
      # three times each way, the 8-rank model; the product library rebuilt afterwards
      for mode in product reuse product reuse product reuse; do
        touch dashing_amd/csrc/engine.hip
        if [ $mode = reuse ]; then make -s -j8 -C dashing_amd/csrc EXPERIMENT=1 > $O/experiment_build.log 2>&1; else make -s -j8 -C dashing_amd/csrc > $O/experiment_build.log 2>&1; fi
        timeout 600 python bench.py --no-secondary --no-pmc --no-cpu-baseline --steps 40 > $O/exp_bench.tmp 2>> $O/experiment.err
        python - "$mode" $O/exp_bench.tmp >> $O/experiment.jsonl <<'PYEOF'
import json, sys
d = json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][-1])
print(json.dumps({"library": sys.argv[1], "ms_per_step": d["ms_per_step"], "pairs_per_s": d["value"], "step_ms": d["roofline"]["step"]["ms"]}))
PYEOF
        if [ $mode = reuse ]; then ( export G=8 TAG=reuse; timeout 600 python tools/shard_model.py >> $O/experiment_model_reuse.jsonl 2>> $O/experiment.err ); else ( export G=8; timeout 600 python tools/shard_model.py >> $O/experiment_model_product.jsonl 2>> $O/experiment.err ); fi
      done
      touch dashing_amd/csrc/engine.hip; make -s -j8 -C dashing_amd/csrc >> $O/experiment_build.log 2>&1
      cat $O/experiment.jsonl; tail -3 $O/experiment.err ;;
    *) echo "unknown step $step" ;;
  esac
done

#!/bin/bash
# round 4, GPU batch 5l: debug of the CLI's three-thread collect over the stand-in transport
set -x
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r5l
mkdir -p "$O"
make -s -C tests/mock_rccl
python - <<'PY'
import os, sys
sys.path.insert(0, ".")
from dashing_amd import synth
gs = synth.synthetic_genomes(9, 30000, seed=5)
os.makedirs("/tmp/g", exist_ok=True)
for i, g in enumerate(gs):
    open("/tmp/g/g%d.fa" % i, "wb").write(b">g%d\n" % i + bytes(g) + b"\n")
PY
export DSH_RCCL_LIB=$PWD/tests/mock_rccl/libmock_rccl.so MOCK_RCCL_DEBUG=1 MOCK_RCCL_TIMEOUT_S=20 DSH_COMM_INIT_TIMEOUT_S=15
timeout 120 ./dashing_amd/dashing-amd dist --avoid-sorting --devices 0,0,0 -O /tmp/three.out -o /dev/null /tmp/g/g*.fa > $O/cli3.out 2> $O/cli3.err; echo "rc $?" >> $O/cli3.err
timeout 120 ./dashing_amd/dashing-amd dist --avoid-sorting --devices 0,0 -O /tmp/two.out -o /dev/null /tmp/g/g*.fa > $O/cli2.out 2> $O/cli2.err; echo "rc $?" >> $O/cli2.err
tail -20 $O/cli3.err; tail -12 $O/cli2.err; ls /dev/shm | head

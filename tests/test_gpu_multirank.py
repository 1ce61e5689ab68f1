"""GPU: the N-rank path end to end on real kernels (sketch shares -> all-gather of registers ->
sharded all-pairs -> gather -> un-permute), see tests/e2e_multigpu_worker.py.  A 1-GPU box can run
(a) RCCL with a single rank -- every collective is issued through the nccl backend -- and
(b) two and three ranks sharing cuda:0 over gloo.  Real multi-GPU RCCL runs only in the driver's bench."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "e2e_multigpu_worker.py")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("backend,world,pieces", [("nccl", 1, 0), ("gloo", 2, 0), ("gloo", 3, 0), ("nccl", 1, 1), ("gloo", 2, 1), ("gloo", 3, 1), ("nccl", 1, 2), ("gloo", 2, 3)])
def test_end_to_end_ranks(backend, world, pieces):
    """pieces == 0: row ranges of the final triangle, point-to-point into place (bench.py's N>1 path).
    pieces > 1: every rank's share is cut into that many shards, each gathered asynchronously while the
    next is computed (multigpu.PipelinedShards, dsh_unpermute_blocks_device)."""
    env = dict(os.environ, E2E_BACKEND=backend, E2E_PIECES=str(pieces), HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), WORKER]
    r = subprocess.run(cmd, env=env, capture_output=True, timeout=600, cwd=ROOT)
    out = r.stdout.decode() + r.stderr.decode()
    assert r.returncode == 0, out[-3000:]
    assert "E2E_OK world=%d backend=%s pieces=%d" % (world, backend, pieces) in out, out[-3000:]

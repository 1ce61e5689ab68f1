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


def _bench(args, env_extra, timeout=900):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **env_extra)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True,
                          timeout=timeout, cwd=ROOT)


def test_bench_gpus_flag_launches_ranks():
    """`python bench.py --gpus 2` with no launcher around it starts two ranks by itself.  On the
    one-GPU box the two ranks share cuda:0 over gloo (DSH_BENCH_BACKEND=gloo, a dry run of the N-rank code path)."""
    import json

    r = _bench(["--gpus", "2", "--steps", "2", "--warmup", "1"], {"DSH_BENCH_BACKEND": "gloo", "DSH_BENCH_N": "3000"})
    out = r.stdout.decode()
    assert r.returncode == 0, (out + r.stderr.decode())[-3000:]
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out[-3000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 2 and line["scaling"] == "strong"
    mg = line["multi_gpu"]
    assert mg["ranks"] == 2 and mg["backend"] == "gloo" and len(mg["row_bounds"]) == 3
    assert line["parity_vs_cpu"]["assembled_equals_single_gpu"] is True
    for key in ("compute_incl_prepare", "exchange", "k_pair_counts", "k_finalize", "prepare"):
        assert mg["phase_ms_max_over_ranks"][key] >= 0.0


def test_bench_gpus_flag_refuses_missing_devices():
    """more ranks than devices over RCCL: non-zero exit, no JSON line -- never a silent 1-GPU run"""
    import dashing_amd

    want = dashing_amd.device_count() + 1
    r = _bench(["--gpus", str(want), "--steps", "1", "--warmup", "0"], {"DSH_BENCH_BACKEND": "nccl"}, timeout=120)
    assert r.returncode != 0
    assert not [l for l in r.stdout.decode().splitlines() if l.startswith("{")]
    assert "needs %d visible" % want in r.stderr.decode()

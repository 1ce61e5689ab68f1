"""CPU: the C-ABI library loads, exports every symbol include/dashing_hip.h declares, and its
host-only helpers behave.  No GPU compute here (dsh_create must fail loudly without a device)."""
import ctypes
import os
import re

import numpy as np
import pytest

import dashing_amd
from dashing_amd import api

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    txt = open(os.path.join(ROOT, "include", "dashing_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(dsh_[a-z_0-9]+)\s*\(", txt)))


def test_header_symbols_exported():
    lib = ctypes.CDLL(dashing_amd.lib_path())
    decl = _declared_symbols()
    assert len(decl) >= 20
    for s in decl:
        assert hasattr(lib, s), "libdashing_hip.so does not export %s" % s
    assert sorted(api.SYMBOLS) == decl, "python binding out of sync with the header"


def test_backend_name():
    assert dashing_amd.backend_name() == "hip:gfx950"


def test_no_cpu_fallback():
    """Without a gfx950 device context creation must fail (never silently compute on the CPU)."""
    if dashing_amd.device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(dashing_amd.DshError) as e:
        dashing_amd.Context(0)
    assert e.value.code == -19


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "dashing_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".hpp")) or f == "Makefile":
                txt = open(os.path.join(dirpath, f)).read()
                assert "oracle_c" not in txt and "oracle_py" not in txt and "liboracle" not in txt and "dsho_" not in txt, f


def test_tri_index_and_span():
    n = 37
    k = 0
    for i in range(n):
        for j in range(i + 1, n):
            assert dashing_amd.tri_index(n, i, j) == k  # distmat/distmat.h:260-264 is row-major
            k += 1
    assert dashing_amd.tri_span(n, 0, n) == n * (n - 1) // 2
    assert dashing_amd.tri_span(n, 5, 9) == sum(n - 1 - i for i in range(5, 9))
    assert dashing_amd.tri_span(n, 9, 5) == 0
    assert dashing_amd.tri_span(n, 30, 99) == sum(n - 1 - i for i in range(30, n))


@pytest.mark.parametrize("n,parts", [(10000, 8), (10000, 2), (1000, 4), (65, 8), (1, 2), (0, 3), (100000, 8)])
def test_partition_rows(n, parts):
    b = dashing_amd.partition_rows(n, parts, 64)
    assert b[0] == 0 and b[-1] == n and len(b) == parts + 1
    assert all(b[i] <= b[i + 1] for i in range(parts))
    assert all(x % 64 == 0 for x in b[1:-1] if x != n)
    if n >= 10000:
        spans = [dashing_amd.tri_span(n, b[i], b[i + 1]) for i in range(parts)]
        assert max(spans) / (sum(spans) / parts) < 1.05  # balanced within 5 %
        assert sum(spans) == n * (n - 1) // 2


@pytest.mark.parametrize("n,parts", [(10000, 8), (10000, 2), (10000, 1), (1000, 4), (65, 8), (129, 2), (1, 2), (0, 3), (100000, 8), (300000, 8)])
def test_balance_rows(n, parts):
    """tile-aligned row ranges for the ranks of a multi-GPU run: cover [0,n), 128-row boundaries, and the
    largest tile count is the smallest any contiguous aligned split can reach (checked by brute force when small)"""
    b = dashing_amd.balance_rows(n, parts)
    assert b[0] == 0 and b[-1] == n and len(b) == parts + 1
    assert all(b[i] <= b[i + 1] for i in range(parts))
    assert all(x % 128 == 0 for x in b[1:-1] if x != n)
    nt = (n + 127) // 128
    kprep = 0.9  # prepare cost of a part in tile-equivalents per 128 of its columns (dsh_balance_rows)

    def tiles(lo, hi):  # tile rows [lo,hi): triangle of the part + rectangle to its right
        return sum(nt - t for t in range(lo, hi))

    def cost(lo, hi):  # + the part's own plane matrix over the columns lo*128 .. n
        return tiles(lo, hi) + kprep * (nt - lo) if hi > lo else 0.0

    tr = [(b[i] // 128, (b[i + 1] + 127) // 128) if b[i + 1] > b[i] else (0, 0) for i in range(parts)]
    assert sum(tiles(lo, hi) for lo, hi in tr if hi > lo) == nt * (nt + 1) // 2
    worst = max(cost(lo, hi) for lo, hi in tr)
    if nt and parts > 1:
        assert worst <= (nt * (nt + 1) / 2) / parts + (1 + kprep) * nt  # never more than one tile row (+ prepare) above the mean
    if 1 < parts <= 4 and nt <= 80:  # exact minimax by enumeration
        import itertools

        best = min(max(cost(c[i], c[i + 1]) for i in range(parts))
                   for mid in itertools.combinations_with_replacement(range(nt + 1), parts - 1)
                   for c in [(0,) + mid + (nt,)])
        assert worst <= best + 1e-2


def test_bench_gpus_flag_never_falls_back_to_one_gpu():
    """`bench.py --gpus N` without a launcher re-execs itself under torch.distributed.run; with fewer than N devices
    (none in the CPU container) it must exit non-zero and print no JSON line."""
    import subprocess
    import sys

    import dashing_amd

    if dashing_amd.device_count() >= 2:
        pytest.skip("a multi-GPU box: covered by the gpu tests")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], env=env,
                       capture_output=True, timeout=300)
    assert r.returncode != 0
    assert b"needs 2 visible" in r.stderr and not r.stdout.strip()
    # a launcher that disagrees with --gpus is refused as well
    env["WORLD_SIZE"] = "4"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], env=env,
                       capture_output=True, timeout=300)
    assert r.returncode != 0 and b"refusing" in r.stderr


def test_row_set_helpers_are_pure_host_functions():
    """dsh_balance_rowsets / dsh_rowsets_rank / dsh_exchange_mode need no device: the partition every rank computes for
    itself, how a rank's buffer is laid out under the exchange and how many floats it holds"""
    n, world, nparts = 10000, 8, 8
    rows = dashing_amd.balance_rowsets(n, world)
    assert rows.world == world and sum(rows.pairs(r) for r in range(world)) == n * (n - 1) // 2
    assert sum(rows.tiles(r) for r in range(world)) == 79 * 80 // 2
    for r in range(world):
        rs, k, floats = dashing_amd.exchange_mode(n, rows, r, nparts, 0, want_floats=True)
        segs = rows.rows(r)
        if r == 0:  # the destination: one part, in final order relative to its first row, up to the end of its last segment
            assert (rs, k) == (False, 1) and floats == dashing_amd.tri_span(n, segs[0][0], segs[-1][1])
        else:       # a source: row-sorted (short range and / or top-up segments), exactly its rows
            assert rs and 2 <= k <= nparts and floats == rows.pairs(r)
    # contiguous bounds as a table
    b = dashing_amd.balance_rows(n, world)
    t = dashing_amd.rowsets_from_bounds(n, b)
    assert [t.rows(r) for r in range(world)] == [[(b[r], b[r + 1])] for r in range(world)]
    # a malformed table is refused
    bad = dashing_amd.RowSets(n, np.array([2, 2, 0, 5000, 9999, 0, 1], np.uint64))  # does not reach n
    with pytest.raises(dashing_amd.DshError):
        bad.rows(0)


def test_header_is_plain_c_and_a_cxx_client_links(tmp_path):
    """include/dashing_hip.h is the drop-in boundary: it must compile as C11 (no C++ in the signatures) as well as C++17, and
    a host written against it alone -- tools/cabi/overlap_cabi.cpp, a plain C++ client without Python or torch -- must
    compile and LINK against libdashing_hip.so (nothing is run: no GPU here)."""
    import shutil
    import subprocess

    inc = os.path.join(ROOT, "include")
    hdr = os.path.join(inc, "dashing_hip.h")
    c_src = tmp_path / "use.c"
    c_src.write_text('#include "dashing_hip.h"\nint main(void) { dsh_ctx *c = 0; (void)c; return dsh_abi_version() == DSH_ABI_VERSION ? 0 : 1; }\n')
    assert os.path.exists(hdr)
    r = subprocess.run(["gcc", "-std=c11", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-I", inc, str(c_src)], capture_output=True)
    assert r.returncode == 0, r.stderr.decode()
    r = subprocess.run(["g++", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-x", "c++", "-I", inc, str(c_src)], capture_output=True)
    assert r.returncode == 0, r.stderr.decode()
    lib = os.path.join(ROOT, "dashing_amd", "libdashing_hip.so")
    if not os.path.exists(lib) or not shutil.which("g++"):
        pytest.skip("libdashing_hip.so not built")
    exe = tmp_path / "overlap_cabi"
    r = subprocess.run(["g++", "-O1", "-std=c++17", os.path.join(ROOT, "tools", "cabi", "overlap_cabi.cpp"), "-I", inc, "-L", os.path.dirname(lib),
                        "-ldashing_hip", "-Wl,-rpath," + os.path.dirname(lib), "-Wl,-rpath,/opt/rocm/lib", "-Wl,-rpath-link,/opt/rocm/lib", "-o", str(exe)], capture_output=True)
    assert r.returncode == 0, r.stderr.decode()[-2000:]

"""CPU: the packed-triangle container, the `-b` file layout and `printmat`, pinned to fixtures that were
produced by the REFERENCE's own code -- tests/golden/distmat/* come from oracle/_ref/distmat_ref, our small driver
linked against /root/reference/distmat/distmat.h (dm::DistanceMatrix<float>, dm::parallel_fill; generator:
tests/golden/make_distmat_golden.py).  This pins SURVEY.md 8a row a9 (distmat/distmat.h:196-204,260-279,
358-412,459-512) to the reference; the arithmetic of the path stays unpinned."""
import ctypes as C
import gzip
import hashlib
import json
import os
import subprocess

import numpy as np
import pytest

import dashing_amd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden", "distmat")
MANIFEST = json.load(open(os.path.join(GOLD, "manifest.json")))
CLI = os.path.join(ROOT, "dashing_amd", "dashing-amd")


def fixture_bytes(tag, ext):
    p = os.path.join(GOLD, tag + ext)
    if os.path.exists(p):
        return open(p, "rb").read()
    if os.path.exists(p + ".gz"):
        return gzip.decompress(open(p + ".gz", "rb").read())
    return None


def enc(big, small):
    return np.float32((small * 4096 + big) / 1024.0)


def our_triangle(n):
    """the values the driver's oracle(k, j) returns, placed with OUR index function"""
    tri = np.zeros(n * (n - 1) // 2, np.float32)
    for i in range(n):
        for j in range(i + 1, n):
            tri[dashing_amd.tri_index(n, i, j)] = enc(j, i)
    return tri


@pytest.fixture(scope="module")
def host():
    lib = C.CDLL(os.path.join(ROOT, "dashing_amd", "libdashing_host.so"))
    lib.dshh_emit_matrix.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_void_p]
    return lib


@pytest.mark.parametrize("tag", sorted(MANIFEST))
def test_manifest_matches_files(tag):
    for ext, meta in MANIFEST[tag].items():
        if not ext.startswith("."):
            continue
        data = fixture_bytes(tag, ext)
        if data is not None:
            assert hashlib.sha256(data).hexdigest() == meta["sha256"] and len(data) == meta["bytes"]


@pytest.mark.parametrize("tag", sorted(MANIFEST))
def test_binary_writer_matches_reference(host, tmp_path, tag):
    """our -b emitter ('\\0', u64 n, floats at dsh_tri_index) == DistanceMatrix<float>::write, byte for byte"""
    n = MANIFEST[tag]["n"]
    out = str(tmp_path / "m.bin")
    tri = our_triangle(n)
    names = "\n".join("g%d" % i for i in range(n)).encode()
    assert host.dshh_emit_matrix(out.encode(), 1, names, tri.ctypes.data) == 0
    got = open(out, "rb").read()
    assert hashlib.sha256(got).hexdigest() == MANIFEST[tag][".bin"]["sha256"]
    want = fixture_bytes(tag, ".bin")
    if want is not None:
        assert got == want


@pytest.mark.parametrize("tag", sorted(MANIFEST))
def test_index_span_rows_match_reference(tag):
    """dsh_tri_index / dsh_tri_span vs DistanceMatrix::index, row_ptr, row_span"""
    idx = fixture_bytes(tag, ".idx.txt")
    if idx is None:
        pytest.skip("index dump kept as sha256 only")
    n = MANIFEST[tag]["n"]
    rows = pairs = 0
    for line in idx.decode().splitlines():
        f = line.split()
        if f[0] == "row":
            i, off, ln = int(f[1]), int(f[2]), int(f[3])
            assert dashing_amd.tri_span(n, 0, i) == off
            assert dashing_amd.tri_span(n, i, i + 1) == ln
            rows += 1
        else:
            i, j, k = int(f[1]), int(f[2]), int(f[3])
            assert dashing_amd.tri_index(n, min(i, j), max(i, j)) == k
            pairs += 1
    assert rows == n and pairs > 0


@pytest.mark.parametrize("tag", sorted(MANIFEST))
def test_printmat_matches_reference(tmp_path, tag):
    """`dashing-amd printmat [-s]` == DistanceMatrix::printf(fp, use_scientific) on the reference's own file"""
    blob = fixture_bytes(tag, ".bin")
    if blob is None:
        pytest.skip("binary kept as sha256 only")
    f = tmp_path / "m.bin"
    f.write_bytes(blob)
    for flag, ext in (([], ".txt"), (["-s"], ".sci.txt")):
        r = subprocess.run([CLI, "printmat"] + flag + [str(f)], capture_output=True, timeout=120)
        assert r.returncode == 0, r.stderr
        assert hashlib.sha256(r.stdout).hexdigest() == MANIFEST[tag][ext]["sha256"]
        want = fixture_bytes(tag, ext)
        if want is not None:
            assert r.stdout == want


def test_partition_rows_spans_are_reference_row_ptrs():
    """dsh_partition_rows bounds are row boundaries whose spans add up to the reference's row_ptr offsets"""
    n = 280
    idx = fixture_bytes("n280_b16", ".idx.txt").decode().splitlines()
    off = {int(l.split()[1]): int(l.split()[2]) for l in idx if l.startswith("row")}
    for parts in (1, 2, 3, 8):
        b = dashing_amd.partition_rows(n, parts, 1)
        assert b[0] == 0 and b[-1] == n
        for r in range(parts):
            end = off[b[r + 1]] if b[r + 1] < n else n * (n - 1) // 2
            assert dashing_amd.tri_span(n, b[r], b[r + 1]) == end - off[b[r]]


def test_fixtures_are_fresh_when_reference_is_present(tmp_path):
    """in the build container (oracle/_ref built from /root/reference) the reference still produces the fixtures"""
    drv = os.path.join(ROOT, "oracle", "_ref", "distmat_ref")
    if not os.path.exists(drv):
        pytest.skip("oracle/_ref/distmat_ref not built (reference absent)")
    for tag, meta in MANIFEST.items():
        pre = str(tmp_path / tag)
        subprocess.check_call([drv, str(meta["n"]), str(meta["nperbatch"]), pre], stderr=subprocess.DEVNULL)
        for ext in (".bin", ".txt", ".sci.txt", ".idx.txt"):
            assert hashlib.sha256(open(pre + ext, "rb").read()).hexdigest() == meta[ext]["sha256"]
        assert open(pre + ".bin", "rb").read() == open(pre + ".mmap.bin", "rb").read()  # dashing's in-place -b path

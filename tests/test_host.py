"""CPU: host-side logic either side of the GPU path (dashing_amd/csrc/host/): cache file names,
.hll files, FASTA/FASTQ reading, input ordering, and the four emitters -- byte-exact against
hand-written expectations derived from the reference's format code."""
import ctypes as C
import gzip
import os
import struct

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def host():
    lib = C.CDLL(os.path.join(ROOT, "dashing_amd", "libdashing_host.so"))
    vp, cp, sz = C.c_void_p, C.c_char_p, C.c_size_t
    lib.dshh_append_fastx.restype = C.c_long
    lib.dshh_append_fastx.argtypes = [cp, vp, sz, C.POINTER(sz)]
    lib.dshh_make_fname.argtypes = [cp, C.c_uint, C.c_int, cp, cp, cp, cp, sz]
    lib.dshh_write_hll.argtypes = [cp, vp, C.c_int, C.c_int]
    lib.dshh_read_hll.argtypes = [cp, vp, sz, C.POINTER(C.c_int)]
    lib.dshh_sort_paths.argtypes = [cp, cp, sz]
    lib.dshh_split_genome_paths.argtypes = [cp, cp, sz]
    lib.dshh_emit_matrix.argtypes = [cp, C.c_int, cp, vp]
    lib.dshh_emit_sizes.argtypes = [cp, cp, vp]
    return lib


def fname(host, path, p=10, k=31, spacing="", suffix="", prefix=""):
    buf = C.create_string_buffer(4096)
    assert host.dshh_make_fname(path.encode(), p, k, spacing.encode(), suffix.encode(), prefix.encode(), buf, 4096) == 0
    return buf.value.decode()


def test_make_fname(host):
    # src/dashing.h:497-526 -- note nothing follows ".w" (the discarded `ret + ...` at :510)
    assert fname(host, "g1.fna") == "g1.fna.w.31.spacing.10.hll"
    assert fname(host, "dir/g1.fna", p=14, k=21) == "dir/g1.fna.w.21.spacing.14.hll"
    assert fname(host, "dir/g1.fna", prefix="cache") == "cache/g1.fna.w.31.spacing.10.hll"
    assert fname(host, "g1.fna", suffix="x") == "g1.fna.w.31.spacing.sufx.10.hll"
    # multi-file genome: the name comes from the part after the first separator
    assert fname(host, "a.fna b.fna") == "b.fna.w.31.spacing.10.hll"


def test_hll_roundtrip_and_layouts(host, tmp_path):
    rng = np.random.default_rng(1)
    for p in (4, 10, 14):
        regs = rng.integers(0, 30, 1 << p).astype(np.uint8)
        path = str(tmp_path / ("a%d.hll" % p))
        assert host.dshh_write_hll(path.encode(), regs.ctypes.data, p, 2) == 0
        raw = gzip.open(path).read()
        assert len(raw) == 28 + (1 << p)
        assert struct.unpack("<4I", raw[:16]) == (0, 2, 2, 1)
        assert struct.unpack("<I", raw[16:20]) == (p,)
        out = np.zeros(1 << p, np.uint8)
        pp = C.c_int()
        assert host.dshh_read_hll(path.encode(), out.ctypes.data, out.size, C.byref(pp)) == 0
        assert pp.value == p and (out == regs).all()
        # older layout: uint8[4] flags
        old = str(tmp_path / ("old%d.hll" % p))
        with gzip.open(old, "wb") as f:
            f.write(bytes([0, 2, 2, 137]) + struct.pack("<I", p) + struct.pack("<d", 0.0) + regs.tobytes())
        out[:] = 0
        assert host.dshh_read_hll(old.encode(), out.ctypes.data, out.size, C.byref(pp)) == 0
        assert pp.value == p and (out == regs).all()
        # uncompressed file is read transparently
        plain = str(tmp_path / ("plain%d.hll" % p))
        open(plain, "wb").write(raw)
        assert host.dshh_read_hll(plain.encode(), out.ctypes.data, out.size, C.byref(pp)) == 0
    bad = str(tmp_path / "bad.hll")
    open(bad, "wb").write(b"x" * 100)
    assert host.dshh_read_hll(bad.encode(), out.ctypes.data, out.size, C.byref(pp)) != 0
    assert host.dshh_read_hll(b"/nonexistent.hll", out.ctypes.data, out.size, C.byref(pp)) != 0


def parse(host, path):
    buf = np.zeros(1 << 16, np.uint8)
    n = C.c_size_t()
    rc = host.dshh_append_fastx(path.encode(), buf.ctypes.data, buf.size, C.byref(n))
    return rc, buf[: n.value].tobytes()


def test_fastx(host, tmp_path):
    fa = tmp_path / "a.fa"
    fa.write_text(">r1 desc\nACGT\nacgtN\n\n>r2\nGGGG\r\nCC\n")
    assert parse(host, str(fa)) == (2, b"ACGTacgtNNGGGGCC")
    gz = tmp_path / "a.fa.gz"
    with gzip.open(gz, "wb") as f:
        f.write(fa.read_bytes())
    assert parse(host, str(gz)) == (2, b"ACGTacgtNNGGGGCC")
    fq = tmp_path / "a.fq"
    fq.write_text("@q1\nACGT\n+\n@>II\n@q2\nTTGA\n+q2\nIIII\n")  # quality starting with '@'
    assert parse(host, str(fq)) == (2, b"ACGTNTTGA")
    empty = tmp_path / "e.fa"
    empty.write_text("")
    assert parse(host, str(empty)) == (0, b"")
    assert parse(host, "/nonexistent.fa")[0] == -1


def test_sort_and_split(host, tmp_path):
    sizes = {"a": 10, "b": 300, "c": 300, "d": 5}
    paths = []
    for k, v in sizes.items():
        (tmp_path / k).write_bytes(b"x" * v)
        paths.append(str(tmp_path / k))
    multi = paths[0] + " " + paths[3]  # a genome made of two files: 15 bytes
    buf = C.create_string_buffer(1 << 14)
    n = host.dshh_sort_paths("\n".join(paths + [multi]).encode(), buf, 1 << 14)
    got = buf.value.decode().strip().split("\n")
    assert n == 5 and got == [paths[1], paths[2], multi, paths[0], paths[3]]  # largest first, stable
    host.dshh_split_genome_paths(multi.encode(), buf, 1 << 14)
    assert buf.value.decode().strip().split("\n") == [paths[0], paths[3]]


def emit(host, tmp_path, fmt, names, tri):
    out = str(tmp_path / ("m%d" % fmt))
    t = np.ascontiguousarray(tri, np.float32)
    assert host.dshh_emit_matrix(out.encode(), fmt, "\n".join(names).encode(), t.ctypes.data) == 0
    return open(out, "rb").read()


def test_emitters_byte_exact(host, tmp_path):
    names = ["g1.fna", "genome_two.fna", "g3", "g4"]
    tri = [0.5, 0.25, 1.0, 0.000123456789, 0.0, 1e-10]  # rows: (0,1)(0,2)(0,3)(1,2)(1,3)(2,3)
    ut = emit(host, tmp_path, 0, names, tri)
    assert ut == (b"##Names\tg1.fna\tgenome_two.fna\tg3\tg4\n"
                  b"g1.fna\t-\t0.5\t0.25\t1\n"
                  b"genome_two.fna\t-\t-\t0.000123457\t0\n"
                  b"g3\t-\t-\t-\t1e-10\n"
                  b"g4\t-\t-\t-\t-\n")
    ph = emit(host, tmp_path, 2, names, tri)
    assert ph == (b"4\n"
                  b"g1.fna   \t0.5\t0.25\t1\n"
                  b"genome_two.fna\t0.000123457\t0\n"
                  b"g3       \t1e-10\n"
                  b"g4       \n")
    full = emit(host, tmp_path, 3, names, tri)
    assert full == (b"#Namesg1.fna\tgenome_two.fna\tg3\tg4\n"
                    b"g1.fna\t0\t0.5\t0.25\t1\n"
                    b"genome_two.fna\t0.5\t0\t0.000123457\t0\n"
                    b"g3\t0.25\t0.000123457\t0\t1e-10\n"
                    b"g4\t1\t0\t1e-10\t0\n")
    b = emit(host, tmp_path, 1, names, tri)
    assert b[:1] == b"\0" and struct.unpack("<Q", b[1:9]) == (4,)
    assert np.frombuffer(b[9:], np.float32).tolist() == np.array(tri, np.float32).tolist()
    assert len(b) == 9 + 4 * 6


def test_sizes_file(host, tmp_path):
    out = str(tmp_path / "sizes")
    card = np.array([1234.9, 5.0e6 + 0.2], np.float64)
    assert host.dshh_emit_sizes(out.encode(), b"a.fna\nb.fna", card.ctypes.data) == 0
    assert open(out).read() == "#Path\tSize (est.)\na.fna\t1234\nb.fna\t5000000\n"

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
    lib.dshh_append_fastx_into.restype = C.c_long
    lib.dshh_append_fastx_into.argtypes = [cp, vp, sz, C.POINTER(sz)]
    lib.dshh_append_fastx.argtypes = [cp, vp, sz, C.POINTER(sz)]
    lib.dshh_make_fname.argtypes = [cp, C.c_uint, C.c_int, cp, cp, cp, cp, sz]
    lib.dshh_write_hll.argtypes = [cp, vp, C.c_int, C.c_int]
    lib.dshh_read_hll.argtypes = [cp, vp, sz, C.POINTER(C.c_int)]
    lib.dshh_sort_paths.argtypes = [cp, cp, sz]
    lib.dshh_split_genome_paths.argtypes = [cp, cp, sz]
    lib.dshh_emit_matrix.argtypes = [cp, C.c_int, cp, vp]
    lib.dshh_emit_sizes.argtypes = [cp, cp, vp]
    lib.dshh_fold.argtypes = [vp, C.c_int, C.c_int, vp]
    lib.dshh_union.argtypes = [vp, vp, sz]
    lib.dshh_union.restype = None
    lib.dshh_write_hll_multi.argtypes = [cp, vp, sz, C.c_int, C.c_int]
    lib.dshh_read_hll_multi.argtypes = [cp, vp, sz, C.POINTER(C.c_int), C.POINTER(sz)]
    lib.dshh_write_labels_gz.argtypes = [cp, cp]
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


def parse(host, path, cap=1 << 16):
    buf = np.zeros(cap, np.uint8)
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


def test_fastx_into_caller_memory(host, tmp_path):
    """append_fastx_into (what the CLI's streaming loader calls): parses straight into caller-owned memory behind what is
    already there, never grows it, reports an overflow instead of truncating silently"""
    fa = tmp_path / "a.fa"
    fa.write_text(">r1\nACGT\nAC\n>r2\nGG\n")
    buf = np.full(64, ord("x"), np.uint8)
    n = C.c_size_t(3)  # three bytes already in the buffer
    assert host.dshh_append_fastx_into(str(fa).encode(), buf.ctypes.data, buf.size, C.byref(n)) == 2
    assert n.value == 3 + 9 and buf[:12].tobytes() == b"xxxACGTACNGG" and buf[12] == ord("x")
    # a second file behind the first, as for a multi-file genome
    assert host.dshh_append_fastx_into(str(fa).encode(), buf.ctypes.data, buf.size, C.byref(n)) == 2
    assert n.value == 21 and buf[12:21].tobytes() == b"ACGTACNGG"
    small = np.zeros(8, np.uint8)
    n = C.c_size_t(0)
    assert host.dshh_append_fastx_into(str(fa).encode(), small.ctypes.data, small.size, C.byref(n)) == -2
    assert n.value == 0
    assert host.dshh_append_fastx_into(b"/nonexistent.fa", small.ctypes.data, small.size, C.byref(n)) == -1
    # a long single-line record that crosses the parser's 1 MiB read blocks
    big = tmp_path / "big.fa"
    seq = (b"ACGT" * 300_000) + b"N" + (b"TTGA" * 50_000)
    big.write_bytes(b">big\n" + seq + b"\n")
    out = np.zeros(seq.size if hasattr(seq, "size") else len(seq) + 16, np.uint8)
    n = C.c_size_t(0)
    assert host.dshh_append_fastx_into(str(big).encode(), out.ctypes.data, out.size, C.byref(n)) == 1
    assert n.value == len(seq) and out[: n.value].tobytes() == seq


def test_raw_file_bytes_into_caller_memory(host, tmp_path):
    """read_raw_into (the CLI stages plain FASTA as it lies on disk; the device decodes it): appends behind what is there,
    an exact fit is fine, a file that outgrew its region is reported (-2), a missing one too (-1)"""
    host.dshh_read_raw_into.restype = C.c_long
    host.dshh_read_raw_into.argtypes = [C.c_char_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    data = b">r1 some header\nACGT\nAC\n>r2\nGG"
    fa = tmp_path / "raw.fa"
    fa.write_bytes(data)
    buf = np.full(len(data) + 5, ord("x"), np.uint8)
    n = C.c_size_t(2)
    assert host.dshh_read_raw_into(str(fa).encode(), buf.ctypes.data, buf.size, C.byref(n)) == 0
    assert n.value == 2 + len(data) and buf[2 : n.value].tobytes() == data and buf[:2].tobytes() == b"xx" and buf[n.value] == ord("x")
    exact = np.zeros(len(data), np.uint8)
    n = C.c_size_t(0)
    assert host.dshh_read_raw_into(str(fa).encode(), exact.ctypes.data, exact.size, C.byref(n)) == 0 and exact.tobytes() == data
    small = np.zeros(len(data) - 1, np.uint8)
    n = C.c_size_t(0)
    assert host.dshh_read_raw_into(str(fa).encode(), small.ctypes.data, small.size, C.byref(n)) == -2 and n.value == 0
    assert host.dshh_read_raw_into(b"/nonexistent.fa", small.ctypes.data, small.size, C.byref(n)) == -1
    empty = tmp_path / "empty.fa"
    empty.write_bytes(b"")
    assert host.dshh_read_raw_into(str(empty).encode(), small.ctypes.data, small.size, C.byref(n)) == 0 and n.value == 0


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


# ---- utilities on sketches: fold / union / multi-sketch files ----------------
def test_fold_equals_sketching_at_lower_precision(host):
    """hll_t::compress (src/dashing.cpp:588): folding the p-bit sketch of a k-mer stream must give
    exactly the sketch the same stream produces at new_p (register rule, src/readfilt.cpp:86-88)."""
    from dashing_amd import synth
    from oracle import oracle_c

    gs = synth.synthetic_genomes(3, 40000, seed=77, decorate=True)
    seq, off = synth.concat_for_device(gs)
    for p, newp in ((14, 13), (14, 10), (12, 4), (17, 9), (20, 12)):
        hi = oracle_c.sketch_batch(seq, off, 31, p)
        lo = oracle_c.sketch_batch(seq, off, 31, newp)
        for g in range(len(gs)):
            out = np.zeros(1 << newp, np.uint8)
            assert host.dshh_fold(hi[g].ctypes.data, p, newp, out.ctypes.data) == 0
            assert (out == lo[g]).all(), (p, newp, g)
    # saturated and empty registers: value q+1 maps to q'+1, zero stays zero
    p, newp = 10, 8
    regs = np.zeros(1 << p, np.uint8)
    regs[0] = 64 - p + 1
    regs[5] = 3        # low bits 01 of index 5 -> value clz_2(01)+1 = 2 in register 1
    out = np.zeros(1 << newp, np.uint8)
    assert host.dshh_fold(regs.ctypes.data, p, newp, out.ctypes.data) == 0
    assert out[0] == 64 - newp + 1 and out[1] == 2 and out[2:].sum() == 0
    assert host.dshh_fold(regs.ctypes.data, p, p, out.ctypes.data) != 0


def test_union_is_sketch_of_concatenation(host):
    from dashing_amd import synth
    from oracle import oracle_c

    gs = synth.synthetic_genomes(2, 30000, seed=5)
    seq, off = synth.concat_for_device(gs)
    regs = oracle_c.sketch_batch(seq, off, 31, 12)
    joined = np.concatenate([gs[0], np.frombuffer(b"N", np.uint8), gs[1]])  # k-mers must not span genomes
    both = oracle_c.sketch_batch(joined, np.array([0, joined.size], np.uint64), 31, 12)[0]
    acc = regs[0].copy()
    host.dshh_union(acc.ctypes.data, regs[1].ctypes.data, acc.size)
    assert (acc == np.maximum(regs[0], regs[1])).all()
    assert (acc == both).all()


def test_multi_sketch_file_and_labels(host, tmp_path):
    rng = np.random.default_rng(3)
    p, n = 9, 5
    regs = rng.integers(0, 40, (n, 1 << p)).astype(np.uint8)
    path = str(tmp_path / "all.hll")
    assert host.dshh_write_hll_multi(path.encode(), regs.ctypes.data, n, p, 2) == 0
    raw = gzip.open(path).read()
    rec = 28 + (1 << p)
    assert len(raw) == n * rec
    for i in range(n):  # every record is a complete single-sketch image (src/sketch_and_cmp.h:529-536)
        assert struct.unpack("<I", raw[i * rec + 16 : i * rec + 20]) == (p,)
        assert raw[i * rec + 28 : (i + 1) * rec] == regs[i].tobytes()
    out = np.zeros_like(regs)
    pp, nn = C.c_int(), C.c_size_t()
    assert host.dshh_read_hll_multi(path.encode(), out.ctypes.data, out.size, C.byref(pp), C.byref(nn)) == 0
    assert pp.value == p and nn.value == n and (out == regs).all()
    lab = str(tmp_path / "all.hll.labels.gz")
    assert host.dshh_write_labels_gz(lab.encode(), b"a.fna\nb c.fna\n") == 0
    assert gzip.open(lab).read() == b"a.fna\nb c.fna\n"


def test_cli_union_fold_view_run_without_gpu(host, tmp_path):
    """`union`, `fold`, `view` are host-only (no dsh_create): they must work on a CPU-only box."""
    import subprocess

    cli = os.path.join(ROOT, "dashing_amd", "dashing-amd")
    rng = np.random.default_rng(9)
    p = 10
    regs = rng.integers(0, 20, (3, 1 << p)).astype(np.uint8)
    paths = []
    for i in range(3):
        f = str(tmp_path / ("s%d.hll" % i))
        assert host.dshh_write_hll(f.encode(), regs[i].ctypes.data, p, 2) == 0
        paths.append(f)
    u = str(tmp_path / "u.hll")
    r = subprocess.run([cli, "union", "-p", "2", "-o", u] + paths, capture_output=True, timeout=60)
    assert r.returncode == 0, r.stderr.decode()
    out = np.zeros(1 << p, np.uint8)
    pp = C.c_int()
    assert host.dshh_read_hll(u.encode(), out.ctypes.data, out.size, C.byref(pp)) == 0
    assert pp.value == p and (out == regs.max(axis=0)).all()
    # -F paths file and uncompressed output
    lst = tmp_path / "l.txt"
    lst.write_text("\n".join(paths[:2]) + "\n")
    u2 = str(tmp_path / "u2.hll")
    r = subprocess.run([cli, "union", "-Z", "0", "-F", str(lst), "-o", u2], capture_output=True, timeout=60)
    assert r.returncode == 0, r.stderr.decode()
    raw = open(u2, "rb").read()
    assert len(raw) == 28 + (1 << p) and raw[28:] == np.maximum(regs[0], regs[1]).tobytes()
    # mismatched precisions are an error
    small = str(tmp_path / "small.hll")
    assert host.dshh_write_hll(small.encode(), regs[0].ctypes.data, p - 1, 2) == 0
    assert subprocess.run([cli, "union", "-o", u, paths[0], small], capture_output=True).returncode != 0
    # fold: default destination is p-1 (src/dashing.cpp:587)
    f1 = str(tmp_path / "f.hll")
    r = subprocess.run([cli, "fold", "-o", f1, paths[0]], capture_output=True, timeout=60)
    assert r.returncode == 0, r.stderr.decode()
    want = np.zeros(1 << (p - 1), np.uint8)
    assert host.dshh_fold(regs[0].ctypes.data, p, p - 1, want.ctypes.data) == 0
    got = np.zeros(1 << (p - 1), np.uint8)
    assert host.dshh_read_hll(f1.encode(), got.ctypes.data, got.size, C.byref(pp)) == 0
    assert pp.value == p - 1 and (got == want).all()
    assert subprocess.run([cli, "fold", "-p", str(p), "-o", f1, paths[0]], capture_output=True).returncode != 0
    # view: header line + all register values
    r = subprocess.run([cli, "view", paths[1]], capture_output=True, timeout=60)
    assert r.returncode == 0
    lines = r.stdout.decode().strip().split("\n")
    assert lines[0].startswith("#" + paths[1]) and "p=%d" % p in lines[0]
    assert [int(x) for x in lines[1].split(",")] == regs[1].tolist()


def test_cli_printmat(host, tmp_path):
    """`printmat` (src/dashing.cpp:425-452): binary matrix -> full table, "%lf", diagonal 0."""
    import subprocess

    cli = os.path.join(ROOT, "dashing_amd", "dashing-amd")
    n = 4
    tri = np.array([0.5, 0.25, 1.0, 0.125, 0.75, 3e-7], np.float32)
    f = tmp_path / "m.bin"
    f.write_bytes(b"\0" + struct.pack("<Q", n) + tri.tobytes())
    r = subprocess.run([cli, "printmat", str(f)], capture_output=True, timeout=60)
    assert r.returncode == 0, r.stderr.decode()
    rows = [l.split("\t") for l in r.stdout.decode().strip().split("\n")]
    full = np.zeros((n, n))
    k = 0
    for i in range(n):
        for j in range(i + 1, n):
            full[i, j] = full[j, i] = float(tri[k])
            k += 1
    assert [[float(x) for x in row] for row in rows] == [[float("%f" % v) for v in row] for row in full]
    assert rows[0][0] == "0.000000" and rows[0][1] == "0.500000"
    r = subprocess.run([cli, "printmat", "-s", str(f)], capture_output=True, timeout=60)
    assert r.returncode == 0 and r.stdout.decode().split("\t")[1] == "5.000000e-01"
    bad = tmp_path / "bad.bin"
    bad.write_bytes(b"\1" + struct.pack("<Q", n))
    assert subprocess.run([cli, "printmat", str(bad)], capture_output=True).returncode != 0


def test_read_hll_rejects_out_of_range_registers(host, tmp_path):
    """a register above 64 - p + 1 cannot come from the register rule (src/readfilt.cpp:86-88): refuse the file"""
    p = 10
    regs = np.zeros(1 << p, np.uint8)
    regs[5] = 64 - p + 1  # the largest legal value
    good = str(tmp_path / "good.hll")
    assert host.dshh_write_hll(good.encode(), regs.ctypes.data, p, 2) == 0
    out = np.zeros(1 << p, np.uint8)
    pp = C.c_int()
    assert host.dshh_read_hll(good.encode(), out.ctypes.data, out.size, C.byref(pp)) == 0 and pp.value == p
    regs[7] = 64 - p + 2
    bad = str(tmp_path / "bad.hll")
    assert host.dshh_write_hll(bad.encode(), regs.ctypes.data, p, 2) == 0
    assert host.dshh_read_hll(bad.encode(), out.ctypes.data, out.size, C.byref(pp)) != 0


def _zstd_compress(data, level=3):
    """frame made with the host's libzstd (the same runtime library the product binds with dlopen); None if absent"""
    try:
        z = C.CDLL("libzstd.so.1")
    except OSError:
        return None
    z.ZSTD_compressBound.restype = C.c_size_t
    z.ZSTD_compressBound.argtypes = [C.c_size_t]
    z.ZSTD_compress.restype = C.c_size_t
    z.ZSTD_compress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int]
    cap = z.ZSTD_compressBound(len(data))
    dst = C.create_string_buffer(cap)
    n = z.ZSTD_compress(dst, cap, data, len(data), level)
    assert n <= cap
    return dst.raw[:n]


def test_zstd_inputs_are_transparent(host, tmp_path):
    """dashing reads inputs through zstd's zlibWrapper (Makefile:58-62, README.md:79): .zst FASTA and .hll files work
    like plain and gzip'ed ones (detected by magic number, not by extension)"""
    text = b">r1 desc\nACGT\nacgtN\n\n>r2\nGGGG\r\nCC\n" + b">big\n" + b"ACGTTGCA" * 40000 + b"\n"
    frame = _zstd_compress(text)
    if frame is None:
        pytest.skip("no libzstd.so.1 on this host")
    plain = tmp_path / "a.fa"
    plain.write_bytes(text)
    zst = tmp_path / "a.fa.zst"
    zst.write_bytes(frame)
    assert parse(host, str(zst), 1 << 20) == parse(host, str(plain), 1 << 20)
    assert parse(host, str(zst), 1 << 20)[0] == 3
    # two frames back to back decode as their concatenation (as `cat a.zst b.zst` does)
    two = tmp_path / "two.zst"
    two.write_bytes(_zstd_compress(b">x\nAC\n") + _zstd_compress(b"GT\n>y\nTT\n"))
    assert parse(host, str(two)) == (2, b"ACGTNTT")
    # a zstd-compressed .hll (write plain with level 0, compress, read back)
    p = 9
    regs = np.arange(1 << p, dtype=np.uint64).astype(np.uint8) % 50
    raw = str(tmp_path / "s.hll")
    assert host.dshh_write_hll(raw.encode(), regs.ctypes.data, p, 2) == 0
    payload = gzip.open(raw, "rb").read()
    zh = tmp_path / "s.hll.zst"
    zh.write_bytes(_zstd_compress(payload))
    out = np.zeros(1 << p, np.uint8)
    pp = C.c_int()
    assert host.dshh_read_hll(str(zh).encode(), out.ctypes.data, out.size, C.byref(pp)) == 0
    assert pp.value == p and (out == regs).all()
    # a small single-block frame (everything is decoded before the input runs out: the decoder must be drained)
    small = b">s\n" + b"ACGTTGCAAC" * 4900 + b"\n"
    sp = tmp_path / "small.fa.zst"
    sp.write_bytes(_zstd_compress(small))
    assert parse(host, str(sp), 1 << 20) == (1, b"ACGTTGCAAC" * 4900)
    # a truncated frame is an error, not a short genome
    tr = tmp_path / "trunc.fa.zst"
    tr.write_bytes(frame[: len(frame) - 7])
    assert parse(host, str(tr), 1 << 20)[0] == -1
    # garbage after the magic number: a decode error, not silently empty input
    bad = tmp_path / "bad.fa.zst"
    bad.write_bytes(frame[:4] + b"\xff" * 64)
    assert parse(host, str(bad), 1 << 20)[0] == -1


def _feed_fifo(path, data):
    import threading

    def w():
        with open(path, "wb") as f:
            f.write(data)

    t = threading.Thread(target=w)
    t.start()
    return t


def test_fastx_from_pipes(host, tmp_path):
    """Inputs that are not regular files -- a FIFO, as /dev/stdin or `<(zcat x.gz)` are: the format is
    sniffed from the first bytes READ, nothing is pread or opened twice; plain, gzip (two concatenated members) and zstd."""
    import ctypes as C2

    text = b">r1\nACGTACGTAC\nGGTT\n>r2\nTTTTGGGGCCCCAAAA\n" * 400
    want = parse(host, _write(tmp_path / "ref.fa", text), cap=1 << 20)
    assert want[0] == 800
    half = len(text) // 2
    cut = text.rfind(b">", 0, half)
    streams = {"plain": text, "gzip": gzip.compress(text[:cut]) + gzip.compress(text[cut:])}
    try:
        z = C2.CDLL("libzstd.so.1")
        z.ZSTD_compressBound.restype = C2.c_size_t
        z.ZSTD_compressBound.argtypes = [C2.c_size_t]
        z.ZSTD_compress.restype = C2.c_size_t
        z.ZSTD_compress.argtypes = [C2.c_void_p, C2.c_size_t, C2.c_void_p, C2.c_size_t, C2.c_int]
        dst = C2.create_string_buffer(z.ZSTD_compressBound(len(text)))
        nz = z.ZSTD_compress(dst, len(dst), text, len(text), 3)
        streams["zstd"] = dst.raw[:nz]
    except OSError:
        pass
    for name, data in streams.items():
        fifo = str(tmp_path / ("in_%s.fifo" % name))
        os.mkfifo(fifo)
        t = _feed_fifo(fifo, data)
        got = parse(host, fifo, cap=1 << 20)
        t.join()
        assert got == want, name
    # a truncated gzip member on a pipe is an error, not a silent short read
    fifo = str(tmp_path / "trunc.fifo")
    os.mkfifo(fifo)
    t = _feed_fifo(fifo, gzip.compress(text)[:-20])
    assert parse(host, fifo, cap=1 << 20)[0] == -1
    t.join()


def _write(path, data):
    path.write_bytes(data)
    return str(path)

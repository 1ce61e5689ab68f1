"""GPU: the dashing-amd CLI end to end (FASTA -> sketch -> dist -> emitters) against the oracle.
BASELINE configs[0] in miniature: synthetic related genomes, k=31, `dist` with dashing's flags."""
import gzip
import os
import struct
import subprocess

import numpy as np
import pytest

from dashing_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "dashing_amd", "dashing-amd")


def _zstd(data):
    import ctypes as C

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
    n = z.ZSTD_compress(dst, cap, data, len(data), 3)  # (before dst.raw is read)
    return dst.raw[:n]


def run(*args, cwd=None):
    r = subprocess.run([CLI] + [str(a) for a in args], cwd=cwd, capture_output=True, timeout=300)
    assert r.returncode == 0, r.stderr.decode()
    return r


@pytest.fixture(scope="module")
def genomes(tmp_path_factory):
    d = tmp_path_factory.mktemp("genomes")
    lens = [60000, 52000, 71000, 45000, 66000, 58000, 49000, 64000]
    base = synth.synthetic_genomes(len(lens), 80000, seed=0xC11)
    paths, seqs = [], []
    for i, (g, L) in enumerate(zip(base, lens)):
        s = g[:L].copy()
        p = d / ("g%d.fna" % i)
        if i == 2:  # two records
            fa = synth.to_fasta(s[:30000], "g2a") + synth.to_fasta(s[30000:], "g2b")
            s = np.concatenate([s[:30000], np.array([ord("N")], np.uint8), s[30000:]])
        else:
            fa = synth.to_fasta(s, "g%d" % i)
        if i == 5:
            p = d / "g5.fna.gz"
            with gzip.open(p, "wb") as f:
                f.write(fa)
        elif i == 6 and _zstd(fa) is not None:  # zstd input, transparent like gzip (README.md:79 of the reference)
            p = d / "g6.fna.zst"
            p.write_bytes(_zstd(fa))
        else:
            p.write_bytes(fa)
        paths.append(str(p))
        seqs.append(s)
    return d, paths, seqs


def oracle_regs(oracle, seqs, k, p, canon=True):
    seq, off = synth.concat_for_device(seqs)
    return oracle.sketch_batch(seq, off, k, p, canon)


def parse_ut(text, n):
    lines = text.decode().split("\n")
    assert lines[0].startswith("##Names\t")
    names = lines[0].split("\t")[1:]
    vals = []
    for i in range(n):
        f = lines[1 + i].split("\t")
        assert f[0] == names[i]
        assert f[1 : 2 + i] == ["-"] * (i + 1)
        vals += [float(x) for x in f[2 + i :]]
    return names, np.array(vals)


def test_dist_default_ut_tsv(genomes, oracle, tmp_path):
    d, paths, seqs = genomes
    out, sizes = tmp_path / "d.tsv", tmp_path / "s.tsv"
    run("dist", "-k", 31, "-S", 10, "-p", 4, "--avoid-sorting", "-O", out, "-o", sizes, *paths)
    regs = oracle_regs(oracle, seqs, 31, 10)
    want = oracle.dist_tri(regs)
    names, got = parse_ut(out.read_bytes(), len(paths))
    assert names == paths
    exp = np.array([float("%.6g" % x) for x in want])
    assert np.allclose(got, exp, rtol=2e-6, atol=1e-12)
    card = oracle.cardinalities(regs)
    lines = sizes.read_text().split("\n")
    assert lines[0] == "#Path\tSize (est.)"
    for i, pth in enumerate(paths):
        nm, v = lines[1 + i].split("\t")
        assert nm == pth and abs(int(v) - int(card[i])) <= 1


def test_size_sorted_order(genomes, tmp_path):
    d, paths, seqs = genomes
    out = tmp_path / "d.tsv"
    run("dist", "-O", out, "-o", os.devnull, *paths)
    names, _ = parse_ut(out.read_bytes(), len(paths))
    sizes = [os.path.getsize(p) for p in names]
    assert sizes == sorted(sizes, reverse=True)  # largest file first (src/distmain.cpp:126-129)


@pytest.mark.parametrize("flags,estim,rt", [(["-M"], 2, 0), (["-l", "-E"], 0, 3), (["-I"], 1, 1)])
def test_binary_and_measures(genomes, oracle, tmp_path, flags, estim, rt):
    d, paths, seqs = genomes
    out = tmp_path / "d.bin"
    run("dist", "-k", 21, "-S", 12, "-b", "--avoid-sorting", "-O", out, "-o", os.devnull, *flags, *paths)
    raw = out.read_bytes()
    n = len(paths)
    assert raw[0] == 0 and struct.unpack("<Q", raw[1:9])[0] == n and len(raw) == 9 + 4 * n * (n - 1) // 2
    got = np.frombuffer(raw[9:], np.float32)
    want = oracle.dist_tri(oracle_regs(oracle, seqs, 21, 12), estim, rt, 21)
    assert np.allclose(got, want, rtol=1e-6, atol=1e-12)
    assert (tmp_path / "d.bin.labels").read_text().split("\n")[:-1] == paths


def test_phylip_full_tsv_and_nocanon(genomes, oracle, tmp_path):
    d, paths, seqs = genomes
    sub = paths[:4]
    want = oracle.dist_tri(oracle_regs(oracle, seqs[:4], 31, 10, canon=False))
    ph = tmp_path / "p.txt"
    run("dist", "-U", "-C", "--avoid-sorting", "-O", ph, "-o", os.devnull, *sub)
    lines = ph.read_text().split("\n")
    assert lines[0] == "4"
    k = 0
    for i in range(4):
        f = lines[1 + i].split("\t")
        assert f[0].rstrip(" ") == sub[i] and len(f[0]) >= 9
        for x in f[1:]:
            assert abs(float(x) - float("%.6g" % want[k])) <= 2e-6 * max(want[k], 1e-9)
            k += 1
    full = tmp_path / "f.txt"
    run("dist", "-T", "-C", "--avoid-sorting", "-O", full, "-o", os.devnull, *sub)
    lines = full.read_text().split("\n")
    assert lines[0] == "#Names" + "\t".join(sub)
    row1 = lines[2].split("\t")
    assert row1[0] == sub[1] and row1[2] == "0" and abs(float(row1[1]) - want[0]) < 1e-5


def test_cache_presketched_and_sketch_subcommand(genomes, oracle, tmp_path):
    d, paths, seqs = genomes
    cache = tmp_path / "cache"
    cache.mkdir()
    a = tmp_path / "a.bin"
    run("dist", "-b", "-W", "-P", cache, "--avoid-sorting", "-O", a, "-o", os.devnull, *paths)
    hlls = [str(cache / (os.path.basename(p) + ".w.31.spacing.10.hll")) for p in paths]
    assert all(os.path.exists(h) for h in hlls)
    regs = oracle_regs(oracle, seqs, 31, 10)
    for i, h in enumerate(hlls):  # .hll payload = registers, bit-exact
        raw = gzip.open(h).read()
        assert raw[28:] == regs[i].tobytes() and struct.unpack("<I", raw[16:20])[0] == 10
    b = tmp_path / "b.bin"
    run("dist", "-b", "--presketched", "-O", b, "-o", os.devnull, *hlls)
    assert a.read_bytes() == b.read_bytes()
    c = tmp_path / "c.bin"  # second -W run takes the cache-hit path
    run("dist", "-b", "-W", "-P", cache, "--avoid-sorting", "-O", c, "-o", os.devnull, *paths)
    assert a.read_bytes() == c.read_bytes()
    sk = tmp_path / "sk"
    sk.mkdir()
    lst = tmp_path / "paths.txt"
    lst.write_text("\n".join(paths) + "\n")
    run("sketch", "-k", 31, "-S", 10, "-P", sk, "-F", lst)
    for p, h in zip(paths, hlls):
        assert gzip.open(str(sk / os.path.basename(h))).read() == gzip.open(h).read()


@pytest.mark.parametrize("p", [14, 15, 16, 17])
def test_sketch_subcommand_at_large_precisions(genomes, oracle, tmp_path, p):
    """`sketch -S 14 ... 17` through the whole CLI path (device FASTA decode, 512- / 1 024-lane k_sketch workgroups, word and
    packed-byte registers): every .hll payload equals the oracle's registers, and `dist --presketched` over the files gives
    the matrix of a direct `dist -S p`."""
    import glob
    import gzip

    d, paths, seqs = genomes
    sk = tmp_path / "sk"
    sk.mkdir()
    run("sketch", "-k", 31, "-S", p, "-p", 4, "-P", sk, *paths)
    regs = oracle_regs(oracle, seqs, 31, p)
    hlls = []
    for i, path in enumerate(paths):
        found = glob.glob(str(sk / (os.path.basename(path) + ".*.hll")))
        assert len(found) == 1, found
        raw = gzip.open(found[0]).read()
        assert raw[-(1 << p):] == regs[i].tobytes(), "registers of %s differ at p = %d" % (path, p)
        hlls.append(found[0])
    a, b = tmp_path / "a.bin", tmp_path / "b.bin"
    run("dist", "-k", 31, "-S", p, "-b", "--avoid-sorting", "-O", a, "-o", os.devnull, *paths)
    run("dist", "--presketched", "-S", p, "-b", "--avoid-sorting", "-O", b, "-o", os.devnull, *hlls)
    assert a.read_bytes() == b.read_bytes()


def test_query_reference_and_containment(genomes, oracle, tmp_path):
    """-Q queries x -F references (partdist_loop format: name, then "\\t%g" per reference) and the
    containment family; an asymmetric measure without -Q switches to all-vs-all rectangle."""
    d, paths, seqs = genomes
    refs, qs = paths[:5], paths[5:]
    (tmp_path / "r.txt").write_text("\n".join(refs) + "\n")
    (tmp_path / "q.txt").write_text("\n".join(qs) + "\n")
    out = tmp_path / "qr.tsv"
    run("dist", "--avoid-sorting", "--containment-index", "-F", tmp_path / "r.txt", "-Q", tmp_path / "q.txt", "-O", out, "-o", os.devnull)
    regs = oracle_regs(oracle, seqs, 31, 10)
    want = oracle.dist_rect(regs[5:], regs[:5], 2, oracle.CONTAINMENT_INDEX, 31)
    lines = out.read_text().split("\n")[:-1]
    assert len(lines) == len(qs)
    for qi, ln in enumerate(lines):
        f = ln.split("\t")
        assert f[0] == qs[qi] and len(f) == 1 + len(refs)
        for j, x in enumerate(f[1:]):
            assert abs(float(x) - float("%g" % want[qi, j])) <= 2e-6 * max(abs(want[qi, j]), 1e-9)
    b = tmp_path / "all.bin"
    run("dist", "--avoid-sorting", "--containment-dist", "-b", "-O", b, "-o", os.devnull, *paths[:4])
    got = np.frombuffer(b.read_bytes(), np.float32).reshape(4, 4)  # partdist binary: raw rows, no header
    w2 = oracle.dist_rect(regs[:4], regs[:4], 2, oracle.CONTAINMENT_DIST, 31)
    assert np.allclose(got, w2, rtol=1e-6, atol=1e-9)
    s = tmp_path / "sizes.bin"
    run("dist", "--avoid-sorting", "--sizes", "-b", "-O", s, "-o", os.devnull, *paths[:4])
    raw = s.read_bytes()
    assert np.allclose(np.frombuffer(raw[9:], np.float32), oracle.dist_tri(regs[:4], 2, oracle.SIZES, 31), rtol=1e-6)


def test_nearest_neighbors_cli(genomes, oracle, tmp_path):
    d, paths, seqs = genomes
    out = tmp_path / "nn.tsv"
    run("dist", "--avoid-sorting", "--nearest-neighbors", 3, "-M", "-O", out, "-o", os.devnull, *paths)
    regs = oracle_regs(oracle, seqs, 31, 10)
    wi, wv = oracle.knn(regs, 3, result_type=oracle.MASH_DIST, k=31)
    lines = out.read_text().split("\n")
    assert lines[0] == "#File\tNeighbor ID:distance\t..."
    for i, pth in enumerate(paths):
        f = lines[1 + i].split("\t")
        assert f[0] == pth and len(f) == 4
        for j, cell in enumerate(f[1:]):
            a, b = cell.split(":")
            assert int(a) == wi[i, j] and abs(float(b) - float("%g" % wv[i, j])) <= 2e-6 * max(wv[i, j], 1e-9)
    b = tmp_path / "nn.bin"
    run("dist", "--avoid-sorting", "--nearest-neighbors", 2, "-b", "-O", b, "-o", os.devnull, *paths)
    raw = b.read_bytes()
    assert struct.unpack("<II", raw[:8]) == (len(paths), 2) and len(raw) == 8 + 8 * 2 * len(paths)


def test_multi_device_binary_matches_single(genomes, tmp_path):
    """--devices shares the rows between contexts (here two contexts on GPU 0, as on a box with one
    GPU): the file must be byte-identical to the single-device one."""
    d, paths, seqs = genomes
    a, b, c = tmp_path / "one.bin", tmp_path / "two.bin", tmp_path / "three.bin"
    run("dist", "-b", "--avoid-sorting", "-O", a, "-o", os.devnull, *paths)
    run("dist", "-b", "--avoid-sorting", "--devices", "0,0", "-O", b, "-o", os.devnull, *paths)
    run("dist", "-b", "--avoid-sorting", "--devices", "0,0,0", "-M", "-O", c, "-o", os.devnull, *paths)
    assert a.read_bytes() == b.read_bytes()
    assert len(c.read_bytes()) == len(a.read_bytes())


def test_inputs_that_are_not_regular_files(genomes, tmp_path):
    """FIFOs / stdin / process substitutions as genome paths (dashing's gz* reader takes them): a plain FASTA
    through /dev/stdin and a gzip stream through a FIFO give the matrix of the regular files."""
    import threading

    d, paths, seqs = genomes
    sub = paths[:4]
    a = tmp_path / "regular.bin"
    run("dist", "-b", "--avoid-sorting", "-O", a, "-o", os.devnull, *sub)
    fifo = str(tmp_path / "g1.fifo")
    os.mkfifo(fifo)
    payload = gzip.compress(open(sub[1], "rb").read())

    def feed():
        with open(fifo, "wb") as f:
            f.write(payload)

    t = threading.Thread(target=feed)
    t.start()
    b = tmp_path / "piped.bin"
    r = subprocess.run([CLI, "dist", "-b", "--avoid-sorting", "-O", str(b), "-o", os.devnull, "/dev/stdin", fifo, sub[2], sub[3]],
                       input=open(sub[0], "rb").read(), capture_output=True, timeout=300)  # stdin is a pipe
    t.join()
    assert r.returncode == 0, r.stderr.decode()
    assert a.read_bytes() == b.read_bytes()


def test_rccl_collect_path_matches_plain(genomes, tmp_path):
    """`dist --ngpus G` with an output that one writer emits in order goes through dsh_comm_init + dsh_dist_collect
    (RCCL inside the library).  A one-GPU box can only run it with one rank (--rccl forces the path; RCCL refuses two
    ranks on one device): every format must equal the plain run byte for byte."""
    d, paths, seqs = genomes
    for flags in ([], ["-U", "-M"], ["-T"], ["-b"]):
        a, b = tmp_path / "plain.out", tmp_path / "rccl.out"
        run("dist", "--avoid-sorting", *flags, "-O", a, "-o", os.devnull, *paths)
        run("dist", "--avoid-sorting", "--rccl", *flags, "-O", b, "-o", os.devnull, *paths)
        assert a.read_bytes() == b.read_bytes(), flags
    r = subprocess.run([CLI, "dist", "--avoid-sorting", "--devices", "0,0", "-O", str(tmp_path / "x.tsv"), "-o", os.devnull, *paths],
                       capture_output=True, timeout=300)
    assert r.returncode != 0  # two ranks on one GPU: RCCL refuses, the CLI reports it instead of hanging


def test_cli_devices_collect_over_the_stand_in_transport(genomes, tmp_path):
    """`dist --devices 0,0,0` with an output one writer emits in order: three THREADS of the CLI, a context each on GPU 0,
    dsh_comm_init + dsh_dist_collect with three ranks -- served by tests/mock_rccl (DSH_RCCL_LIB), since RCCL itself refuses
    several ranks on one device.  Every format equals the single-device run byte for byte."""
    mock = os.path.join(ROOT, "tests", "mock_rccl", "libmock_rccl.so")
    if not os.path.exists(mock):
        subprocess.check_call(["make", "-s", "-C", os.path.dirname(mock)])
    d, paths, seqs = genomes
    env = dict(os.environ, DSH_RCCL_LIB=mock, MOCK_RCCL_TIMEOUT_S="120")
    for flags in ([], ["-U", "-M"], ["-T"]):
        a, b = tmp_path / "plain.out", tmp_path / "three.out"
        run("dist", "--avoid-sorting", *flags, "-O", a, "-o", os.devnull, *paths)
        r = subprocess.run([CLI, "dist", "--avoid-sorting", "--devices", "0,0,0", *flags, "-O", str(b), "-o", os.devnull, *paths],
                           capture_output=True, timeout=300, env=env)
        assert r.returncode == 0, r.stderr.decode()[-2000:]
        assert a.read_bytes() == b.read_bytes(), flags


def test_multi_device_presketched_large(oracle, tmp_path):
    """700 presketched sketches (several 128-row tile rows per device): 3 contexts vs 1, and vs the oracle."""
    import ctypes as C

    n, p = 700, 11
    regs = synth.synthetic_sketches(n, p, seed=2024)
    host = C.CDLL(os.path.join(ROOT, "dashing_amd", "libdashing_host.so"))
    host.dshh_write_hll.argtypes = [C.c_char_p, C.c_void_p, C.c_int, C.c_int]
    paths = []
    for i in range(n):
        pth = str(tmp_path / ("s%04d.hll" % i))
        assert host.dshh_write_hll(pth.encode(), regs[i].ctypes.data, p, 2) == 0
        paths.append(pth)
    lst = tmp_path / "l.txt"
    lst.write_text("\n".join(paths) + "\n")
    one, three = tmp_path / "1.bin", tmp_path / "3.bin"
    run("dist", "--presketched", "-S", p, "-b", "-p", 8, "-O", one, "-o", os.devnull, "-F", lst)
    run("dist", "--presketched", "-S", p, "-b", "-p", 8, "--devices", "0,0,0", "-O", three, "-o", os.devnull, "-F", lst)
    assert one.read_bytes() == three.read_bytes()
    got = np.frombuffer(one.read_bytes()[9:], np.float32)
    assert np.allclose(got, oracle.dist_tri(regs), rtol=1e-6, atol=1e-12)


def test_cli_rejects_out_of_scope(genomes):
    d, paths, seqs = genomes
    r = subprocess.run([CLI, "dist", "--use-bb-minhash", *paths], capture_output=True)
    assert r.returncode != 0
    r = subprocess.run([CLI, "panel", *paths], capture_output=True)
    assert r.returncode != 0
    r = subprocess.run([CLI, "union", *paths], capture_output=True)  # FASTA files are not sketches
    assert r.returncode != 0


def test_sketch_single_output_file(genomes, oracle, tmp_path):
    """`sketch -o FILE`: all sketches in one gz stream + FILE.labels.gz (src/sketch_and_cmp.h:466-475,529-536)."""
    d, paths, seqs = genomes
    out = str(tmp_path / "all.hll")
    run("sketch", "-k", 31, "-S", 12, "-p", 4, "--avoid-sorting", "-o", out, *paths)
    want = oracle_regs(oracle, seqs, 31, 12)
    raw = gzip.open(out).read()
    rec = 28 + (1 << 12)
    assert len(raw) == len(paths) * rec
    for i in range(len(paths)):
        assert raw[i * rec + 28 : (i + 1) * rec] == want[i].tobytes(), i
    assert gzip.open(out + ".labels.gz").read().decode().split("\n")[:-1] == paths
    assert not any(f.endswith(".hll") for f in os.listdir(d))  # no per-genome files in this mode


@pytest.mark.parametrize("S", [12, 18, 24])
def test_hll_subcommand(genomes, oracle, tmp_path, S):
    """`hll` (src/hllmain.cpp): every input sketched into ONE HLL; p > 17 takes the HBM-register kernel."""
    d, paths, seqs = genomes
    r = run("hll", "-k", 31, "-S", S, "-p", 4, *paths)
    line = r.stdout.decode().strip()
    assert line.startswith("Estimated number of unique exact matches: ")
    got = float(line.split(": ")[1])
    sep = np.frombuffer(b"N", np.uint8)  # k-mers never span files or records
    joined = np.concatenate([x for s_ in seqs for x in (s_, sep)])
    one = oracle.sketch_batch(joined, np.array([0, joined.size], np.uint64), 31, S)
    want = oracle.cardinalities(one)[0]
    assert abs(got - want) <= 1e-6 * want + 1e-6   # printed with %lf: 6 decimals


def test_device_parse_and_host_parse_give_the_same_files(genomes, oracle, tmp_path):
    """Round 6: plain FASTA is staged as raw file bytes and decoded on the device (dsh_sketch_fastx_batch_async);
    DSH_HOST_PARSE=1 keeps the host parser.  Inputs the device path must hand back to the host parser -- a FASTQ file (begins
    with '@'), a '>' file with a FASTQ-like '+' line in it (REFUSED by the device after the batch: re-parsed), a file with a
    blank first line, a genome made of two files (a newline between them), CRLF -- next to gzip'ed and plain ones, enough
    bytes for several batches and all three staging buffers: same binary matrix, same sizes file, same .hll cache files;
    registers equal to the oracle's for the plain genomes."""
    d, paths, seqs = genomes
    big = synth.synthetic_genomes(14, 9_000_000, seed=0xB16, decorate=True)   # 14 x 9 MB: batches of <= 48 MB
    extra = []
    for i, g in enumerate(big):
        p = tmp_path / ("big%02d.fna" % i)
        p.write_bytes(synth.to_fasta(g, "big%d" % i, width=80 if i % 3 else 61))
        extra.append(str(p))
    s0, s1 = seqs[0], seqs[1]
    fq = tmp_path / "reads.fq"
    fq.write_bytes(b"".join(b"@r%d\n%s\n+\n%s\n" % (i, s0[i * 150 : (i + 1) * 150].tobytes(), b"I" * 150) for i in range(300)))
    plus = tmp_path / "plus.fa"
    plus.write_bytes(synth.to_fasta(s1[:20000], "a") + b"+\n" + b"I" * 10 + b"\n")
    blank = tmp_path / "blank.fa"
    blank.write_bytes(b"\n" + synth.to_fasta(s1[:30000], "b"))
    part1, part2 = tmp_path / "part1.fa", tmp_path / "part2.fa"
    part1.write_bytes(synth.to_fasta(s0[:25000], "p1")[:-1])  # no newline at the end of the first file
    part2.write_bytes(synth.to_fasta(s0[25000:50000], "p2"))
    crlf = tmp_path / "crlf.fa"
    crlf.write_bytes(synth.to_fasta(s1[:40000], "c").replace(b"\n", b"\r\n"))
    inputs = extra[:5] + [str(fq), str(plus)] + extra[5:9] + [str(blank), "%s %s" % (part1, part2), str(crlf)] + paths[:7] + extra[9:]
    lst = tmp_path / "in.txt"
    lst.write_text("\n".join(inputs) + "\n")
    outs = {}
    for mode, env in (("device", {}), ("host", {"DSH_HOST_PARSE": "1"})):
        pre = tmp_path / ("cache_" + mode)
        pre.mkdir()
        o, sz = tmp_path / (mode + ".bin"), tmp_path / (mode + ".sizes")
        r = subprocess.run([CLI, "dist", "-k", "31", "-S", "11", "-p", "4", "-b", "--avoid-sorting", "-W", "-P", str(pre), "-O", str(o), "-o", str(sz), "-F", str(lst)],
                           capture_output=True, timeout=600, env=dict(os.environ, **env))
        assert r.returncode == 0, r.stderr.decode()[-2000:]
        outs[mode] = (o.read_bytes(), sz.read_bytes(), {f: (pre / f).read_bytes() for f in sorted(os.listdir(pre))})
    assert outs["device"][0] == outs["host"][0] and outs["device"][1] == outs["host"][1]
    assert list(outs["device"][2]) == list(outs["host"][2]) and len(outs["device"][2]) == len(inputs)
    import ctypes as C

    hostlib = C.CDLL(os.path.join(ROOT, "dashing_amd", "libdashing_host.so"))
    hostlib.dshh_read_hll.argtypes = [C.c_char_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_int)]

    def regs_of(blob_path):
        buf = np.zeros(1 << 11, np.uint8)
        p_ = C.c_int(0)
        assert hostlib.dshh_read_hll(blob_path.encode(), buf.ctypes.data, buf.size, C.byref(p_)) == 0 and p_.value == 11
        return buf
    want = oracle_regs(oracle, [big[0], big[13]], 31, 11)
    names = sorted(os.listdir(tmp_path / "cache_device"))
    for g, w in ((0, want[0]), (13, want[1])):
        f = next(n for n in names if n.startswith("big%02d.fna" % g))
        assert (regs_of(str(tmp_path / "cache_device" / f)) == w).all()


def test_device_parse_many_tiny_genomes_and_one_larger_than_a_batch(tmp_path):
    """the staging of the streaming loader at its edges: 2 500 genomes of 40 .. 3 000 bases (thousands of regions and decode
    chunks per batch, genomes shorter than k, empty files) and one genome of 70 MB (larger than a batch and than the
    page-locked buffers the context thread prepared: the buffer is re-allocated in the middle of the stream) -- the
    device parse and the host parse write the same matrix and the same sizes"""
    rng = np.random.default_rng(6)
    base = synth.synthetic_genomes(1, 3000, seed=0xAB, decorate=False)[0]
    inputs = []
    for i in range(2500):
        L = int(rng.integers(40, 3000)) if i % 97 else 0
        g = base[:L].copy()
        if L:
            pos = rng.integers(0, L, max(1, L // 20))
            g[pos] = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, pos.size)]
        p = tmp_path / ("t%04d.fa" % i)
        p.write_bytes(synth.to_fasta(g, "t%d" % i, width=int(rng.choice([60, 70, 80]))) if L else b"")
        inputs.append(str(p))
    huge = synth.synthetic_genomes(1, 70_000_000, seed=0xCD, decorate=True)[0]
    hp = tmp_path / "huge.fa"
    with open(hp, "wb") as f:
        f.write(b">huge, 16 MB lines\n")
        for x in range(0, huge.size, 1 << 24):  # 16 MB lines: newlines are rare, the carry runs across a thousand chunks
            f.write(huge[x : x + (1 << 24)].tobytes() + b"\n")
    inputs.insert(1200, str(hp))
    lst = tmp_path / "in.txt"
    lst.write_text("\n".join(inputs) + "\n")
    outs = {}
    for mode, env in (("device", {}), ("host", {"DSH_HOST_PARSE": "1"})):
        o, sz = tmp_path / (mode + ".bin"), tmp_path / (mode + ".sizes")
        r = subprocess.run([CLI, "dist", "-k", "21", "-S", "10", "-p", "6", "-b", "--avoid-sorting", "-O", str(o), "-o", str(sz), "-F", str(lst)],
                           capture_output=True, timeout=900, env=dict(os.environ, **env))
        assert r.returncode == 0, r.stderr.decode()[-2000:]
        outs[mode] = (o.read_bytes(), sz.read_bytes())
    assert outs["device"] == outs["host"]
    n = len(inputs)
    assert len(outs["device"][0]) == 9 + 4 * (n * (n - 1) // 2)


def test_full_teardown_and_timing_marks_change_nothing(genomes, tmp_path):
    """the CLI leaves with _Exit once its outputs are closed; DSH_FULL_TEARDOWN=1 runs the destructors instead (leak checkers,
    profilers that write at exit), DSH_TIMING / DSH_T0 print phase marks on stderr: same bytes either way, text output through
    a pipe included (stdio is flushed before leaving)"""
    import time

    d, paths, seqs = genomes
    outs = []
    for env in ({}, {"DSH_FULL_TEARDOWN": "1"}, {"DSH_TIMING": "1", "DSH_T0": repr(time.time())}):
        o = tmp_path / ("m%d.bin" % len(outs))
        r = subprocess.run([CLI, "dist", "-b", "--avoid-sorting", "-O", str(o), *paths[:5]], capture_output=True, timeout=300, env=dict(os.environ, **env))
        assert r.returncode == 0, r.stderr.decode()
        r2 = subprocess.run([CLI, "dist", "--avoid-sorting", *paths[:5]], capture_output=True, timeout=300, env=dict(os.environ, **env))
        assert r2.returncode == 0 and r2.stdout.count(b"\n") >= 6 + 5, r2.stderr.decode()  # sizes (header + 5) and the matrix (header + 5 rows) on stdout
        outs.append((o.read_bytes(), r.stdout, r2.stdout))
        if "DSH_TIMING" in env:
            assert b"[timing] main() entered" in r.stderr and b"leaving" in r.stderr
    assert outs[0] == outs[1] == outs[2]

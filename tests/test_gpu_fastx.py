"""GPU parity of the device-side FASTA / FASTQ decode (dsh_sketch_fastx_batch_async, kernels_fastx.hip): the registers of raw file
bytes decoded ON THE DEVICE equal, bit for bit, the CPU oracle's registers of the sequence the HOST parser
(host/host.cpp FastxParser, through libdashing_host.so) extracts from the same file -- the reference's
Encoder::for_each(func, path) includes the parse (src/sketch_and_cmp.h:338-342).  What is not plain FASTA is refused per
genome (status != 0, nothing sketched), never guessed at."""
import ctypes as C
import os

import numpy as np
import pytest

import dashing_amd
from dashing_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def host():
    lib = C.CDLL(os.path.join(ROOT, "dashing_amd", "libdashing_host.so"))
    lib.dshh_append_fastx.restype = C.c_long
    lib.dshh_append_fastx.argtypes = [C.c_char_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    return lib


def host_parse(host, tmp_path, name, data):
    """the sequence the host parser hands the sketch kernel for this file ('N' between records)"""
    pth = tmp_path / name
    pth.write_bytes(data)
    buf = np.zeros(len(data) + 64, np.uint8)
    n = C.c_size_t(0)
    rc = host.dshh_append_fastx(str(pth).encode(), buf.ctypes.data, buf.size, C.byref(n))
    assert rc >= 0
    return buf[: n.value].copy()


def fasta(rng, records, width, eol=b"\n", final_eol=True, blank_every=0):
    out = []
    for i, (name, seq) in enumerate(records):
        out.append(b">" + name + eol)
        s = bytes(seq)
        if width <= 0:
            out.append(s + eol)
        else:
            for x in range(0, len(s), width):
                out.append(s[x : x + width] + eol)
                if blank_every and (x // width) % blank_every == blank_every - 1:
                    out.append(eol)
    data = b"".join(out)
    if not final_eol and data.endswith(eol):
        data = data[: -len(eol)]
    return data


def check(ctx, oracle, host, tmp_path, files, k=31, p=10, canon=True, expect_status=None):
    ctx.alloc(len(files), p)
    status = ctx.sketch_fastx_batch(files, 0, k, canon)
    got = ctx.download(0, len(files)) if hasattr(ctx, "download") else None
    if got is None:
        got = np.zeros((len(files), 1 << p), np.uint8)
        ctx._ck(ctx._lib.dsh_download_sketches(ctx._h, 0, len(files), got.ctypes.data))
    seqs = [host_parse(host, tmp_path, "f%d.fa" % i, f) for i, f in enumerate(files)]
    for g, (f, s) in enumerate(zip(files, seqs)):
        refused = expect_status is not None and expect_status[g]
        assert bool(status[g]) == bool(refused), "genome %d: status %d" % (g, status[g])
        if refused:
            assert not got[g].any(), "a refused genome must contribute nothing"
            continue
        seq, off = synth.concat_for_device([s])
        want = oracle.sketch_batch(seq, off, k, p, canon)[0]
        assert (got[g] == want).all(), "genome %d: %d registers differ" % (g, int((got[g] != want).sum()))
    return got


def genomes(n, L, seed, decorate=True):
    return [bytes(g) for g in synth.synthetic_genomes(n, L, seed=seed, decorate=decorate)]


def test_line_widths_record_shapes_and_line_ends(ctx, oracle, host, tmp_path):
    rng = np.random.default_rng(1)
    gs = genomes(8, 30011, 3)
    files = [
        fasta(rng, [(b"g0 plain 80 columns", gs[0])], 80),
        fasta(rng, [(b"g1 60 columns, no newline at the end", gs[1])], 60, final_eol=False),
        fasta(rng, [(b"g2 one line", gs[2])], 0),
        fasta(rng, [(b"g3 CRLF", gs[3])], 70, eol=b"\r\n"),
        fasta(rng, [(b"r%d >inner @signs + plus" % i, gs[4][i * 3000 : (i + 1) * 3000]) for i in range(10)], 61),
        fasta(rng, [(b"g5 63", gs[5])], 63) + fasta(rng, [(b"second 64", gs[6])], 64) + fasta(rng, [(b"third 65", gs[7])], 65),
        fasta(rng, [(b"blank lines", gs[6])], 50, blank_every=7),
        fasta(rng, [(b"width 1", gs[7][:4000])], 1),
    ]
    check(ctx, oracle, host, tmp_path, files)


def test_headers_of_every_length_and_position(ctx, oracle, host, tmp_path):
    """header lines from 1 byte to several 16 KB chunks long (the carry must survive whole workgroups), records shorter than
    k, empty records, '@' first on a later line (the host parser takes it for a header too), lowercase and N runs"""
    rng = np.random.default_rng(2)
    gs = genomes(6, 50021, 5)
    long_name = bytes(rng.integers(33, 126, 40000, dtype=np.uint8)).replace(b"\n", b"x")
    files = [
        fasta(rng, [(b"", gs[0])], 80),
        fasta(rng, [(long_name, gs[1]), (long_name[:17000], gs[2][:30000]), (b"x", gs[2][30000:])], 80),
        fasta(rng, [(b"short", b"ACGT"), (b"empty", b""), (b"real", gs[3]), (b"tail", b"ACGTACGTAC")], 80),
        b">a\n" + gs[4][:20000] + b"\n@b also a header\n" + gs[4][20000:] + b"\n",
        fasta(rng, [(b"only a header", b"")], 80),
        b">no sequence, no newline",
        b">",
        fasta(rng, [(b"g5", gs[5])], 16384 - 3),  # lines as long as a chunk
        fasta(rng, [(b"g5b", gs[5])], 64),        # a newline at the same byte of every lane
        fasta(rng, [(b"g5c", gs[5])], 63),
    ]
    check(ctx, oracle, host, tmp_path, files, k=21)
    check(ctx, oracle, host, tmp_path, files, k=32, p=12, canon=False)


def test_what_is_not_plain_fasta_is_refused_not_guessed(ctx, oracle, host, tmp_path):
    rng = np.random.default_rng(3)
    gs = genomes(4, 20011, 7, decorate=False)
    files = [
        fasta(rng, [(b"ok", gs[1])], 80),
        b"ACGT" * 100 + b"\n",                            # begins with neither '>' nor '@'
        b"\n" + fasta(rng, [(b"blank first", gs[2])], 80),  # does not begin with '>'
        fasta(rng, [(b"plus line", gs[3])], 80) + b"+\nIIII\n",  # a '+' line inside a '>' file
        b"",                                             # an empty file: nothing to refuse, nothing to sketch
        fasta(rng, [(b"ok too", gs[2])], 77),
    ]
    # a '+' that starts a line exactly at a lane / chunk boundary
    body = gs[3][: 16384 - 4]
    files.append(b">x\n" + body + b"\n+\nII\n")
    files.append(b">y\n" + gs[3][: 64 - 4] + b"\n+\n")
    check(ctx, oracle, host, tmp_path, files, expect_status=[0, 1, 1, 1, 0, 0, 1, 1])


def fastq(rng, reads, eol=b"\n", final_eol=True, qual=None):
    out = []
    for i, r in enumerate(reads):
        q = qual(i, len(r)) if qual else bytes(rng.integers(33, 75, len(r), dtype=np.uint8))
        out.append(b"@read%d some text\n".replace(b"\n", eol) % i + bytes(r) + eol + b"+" + eol + q + eol)
    data = b"".join(out)
    if not final_eol and data.endswith(eol):
        data = data[: -len(eol)]
    return data


def test_fastq_in_strict_four_line_records(ctx, oracle, host, tmp_path):
    """FASTQ (a genome that begins with '@'): line index modulo 4 by counting newlines, the sequence lines kept, one invalid
    byte per record.  Quality lines may hold any character -- '@', '+', '>' first included --, reads from 1 base to longer
    than a chunk, CRLF, no newline at the end; registers equal the oracle's on the HOST parser's sequence."""
    rng = np.random.default_rng(7)
    g = genomes(1, 400_000, 13, decorate=True)[0]

    def reads(lens):
        out, at = [], 0
        for L in lens:
            out.append(g[at : at + L])
            at = (at + L) % (len(g) - 40000)
        return out

    nasty = lambda i, L: (b"@+>@"[i % 4 : i % 4 + 1] + bytes(rng.integers(33, 75, max(L - 1, 0), dtype=np.uint8)))[:L]
    files = [
        fastq(rng, reads([150] * 400)),
        fastq(rng, reads([int(x) for x in rng.integers(1, 400, 600)]), qual=nasty),
        fastq(rng, reads([100] * 300), eol=b"\r\n"),
        fastq(rng, reads([250] * 100), final_eol=False),
        fastq(rng, reads([20000, 35000, 17, 16384, 16383, 63, 64, 65])),
        fastq(rng, reads([59] * 2000)),   # "@readN some text\n" + 59 + "\n+\n" + 59 + "\n": newlines drift across the lanes
        fastq(rng, reads([40])),
        b"@only a header",
    ]
    check(ctx, oracle, host, tmp_path, files, k=31)
    check(ctx, oracle, host, tmp_path, files[:3], k=16, p=12, canon=False)


def test_fastq_that_is_not_four_line_is_refused(ctx, oracle, host, tmp_path):
    """multi-line sequence or quality, a quality line shorter / longer than its sequence, a blank line between records,
    a record cut off: the device refuses the whole genome (the host parser's record state decides those), a well-formed
    neighbour in the same batch is not touched"""
    rng = np.random.default_rng(8)
    g = genomes(1, 100_000, 17, decorate=False)[0]
    rd = [g[i * 200 : (i + 1) * 200] for i in range(50)]
    good = fastq(rng, rd)
    two_line_seq = b"".join(b"@r%d\n%s\n%s\n+\n%s\n" % (i, r[:100], r[100:], b"I" * 200) for i, r in enumerate(rd))
    two_line_qual = b"".join(b"@r%d\n%s\n+\n%s\n%s\n" % (i, r, b"I" * 100, b"I" * 100) for i, r in enumerate(rd))
    short_qual = fastq(rng, rd[:20]) + b"@x\n" + rd[20] + b"\n+\n" + b"I" * 150 + b"\n" + fastq(rng, rd[21:])
    long_qual = fastq(rng, rd[:20]) + b"@x\n" + rd[20] + b"\n+\n" + b"I" * 250 + b"\n" + fastq(rng, rd[21:])
    blank_between = fastq(rng, rd[:10]) + b"\n" + fastq(rng, rd[10:])
    cut_off = fastq(rng, rd[:10]) + b"@x\n" + rd[10] + b"\n+\n"
    files = [good, two_line_seq, two_line_qual, short_qual, long_qual, blank_between, cut_off, good]
    check(ctx, oracle, host, tmp_path, files, expect_status=[0, 1, 1, 1, 1, 1, 1, 0])
    # found by the fuzz soak of round 6 (case 1685): a SEQUENCE line that begins with '@', '>' or '+' -- kseq and the host
    # parser read those as a header / the separator, a count of lines does not -- and quality lines whose lengths are wrong
    # in ways that cancel in total (one 3 short, another 3 long): per record, not in sum
    marked = [fastq(rng, rd[:7]) + b"@x\n" + c + rd[7] + b"\n+\n" + b"I" * 201 + b"\n" + fastq(rng, rd[8:]) for c in (b"@", b">", b"+")]
    cancel = (fastq(rng, rd[:5]) + b"@s\n" + rd[5] + b"\n+\n" + b"I" * 197 + b"\n" + fastq(rng, rd[6:30]) +
              b"@l\n" + rd[30] + b"\n+\n" + b"I" * 203 + b"\n" + fastq(rng, rd[31:]))
    far = fastq(rng, [g[:20000]] + rd[:3]) + b"@s\n" + g[:40000] + b"\n+\n" + b"I" * 39999 + b"\n" + fastq(rng, rd[3:])  # (lines chunks apart)
    check(ctx, oracle, host, tmp_path, marked + [cancel, far, good], expect_status=[1, 1, 1, 1, 1, 0])


def test_many_genomes_random_shapes(ctx, oracle, host, tmp_path):
    rng = np.random.default_rng(4)
    files = []
    for i in range(40):
        L = int(rng.integers(1, 120000))
        g = genomes(1, max(L, 64), 100 + i, decorate=bool(i & 1))[0][:L]
        nrec = int(rng.integers(1, 6))
        cuts = sorted(set([0, L] + [int(x) for x in rng.integers(0, L + 1, nrec - 1)]))
        recs = [(bytes(rng.integers(48, 123, int(rng.integers(0, 200)), dtype=np.uint8)), g[a:b]) for a, b in zip(cuts[:-1], cuts[1:])]
        files.append(fasta(rng, recs, int(rng.choice([0, 1, 13, 60, 64, 70, 80, 127, 128, 1000])), eol=b"\r\n" if i % 5 == 4 else b"\n",
                           final_eol=bool(i % 3)))
    check(ctx, oracle, host, tmp_path, files)
    check(ctx, oracle, host, tmp_path, files[:10], k=5, p=14)


def test_a_genome_of_many_chunks_and_precision_above_lds(ctx, oracle, host, tmp_path):
    """5 Mbp on 80-column lines = 310 chunks (the per-genome scan walks more than one chunk per lane); p = 18 takes the
    variant of k_sketch that keeps its registers in HBM"""
    rng = np.random.default_rng(5)
    g = genomes(1, 5_000_000, 11, decorate=True)[0]
    files = [fasta(rng, [(b"big", g)], 80), fasta(rng, [(b"big one line", g[:3_000_000])], 0)]
    check(ctx, oracle, host, tmp_path, files)
    check(ctx, oracle, host, tmp_path, [files[0][:400000]], p=18)


_FZ_FIRST = int(os.environ.get("DSH_FASTX_FUZZ_FIRST", "0"))


@pytest.mark.parametrize("case", range(_FZ_FIRST, _FZ_FIRST + int(os.environ.get("DSH_FASTX_FUZZ_CASES", "40"))))
def test_fastx_fuzz_accepted_means_equal_to_the_host_parser(ctx, oracle, host, tmp_path, case):
    """Random FASTA and FASTQ texts, well-formed and damaged (lines dropped, doubled, cut, swapped; '+', '@', '>' put at line
    starts; blank lines; CRLF): whatever the device ACCEPTS (status 0) must give the registers of the host parser's
    sequence -- the device may refuse more than strictly necessary, it may never differ silently -- and what is well-formed
    must be accepted."""
    rng = np.random.default_rng(0xFA57 + case)
    g = genomes(1, 60_000, 1000 + case, decorate=bool(case & 1))[0]
    files, must_accept = [], []
    for f in range(6):
        eol = b"\r\n" if rng.random() < 0.2 else b"\n"
        if rng.random() < 0.5:  # FASTA
            nrec = int(rng.integers(1, 6))
            cuts = sorted(set([0, len(g)] + [int(x) for x in rng.integers(0, len(g), nrec - 1)]))
            recs = [(bytes(rng.integers(33, 126, int(rng.integers(0, 90)), dtype=np.uint8)), g[a:b]) for a, b in zip(cuts[:-1], cuts[1:])]
            data = fasta(rng, recs, int(rng.choice([0, 1, 7, 60, 63, 64, 65, 80, 200, 5000])), eol=eol, final_eol=bool(rng.random() < 0.7),
                         blank_every=int(rng.choice([0, 0, 3, 11])))
            ok = True
        else:  # FASTQ
            lens = [int(x) for x in rng.integers(1, int(rng.choice([50, 300, 3000])), int(rng.integers(1, 120)))]
            at, rds = 0, []
            for L in lens:
                rds.append(g[at : at + L])
                at = (at + L) % (len(g) - 3000)
            data = fastq(rng, rds, eol=eol, final_eol=bool(rng.random() < 0.7))
            ok = True
        if rng.random() < 0.5:  # damage it
            lines = data.split(eol)
            for _ in range(int(rng.integers(1, 4))):
                if len(lines) < 2:
                    break
                i = int(rng.integers(0, len(lines)))
                kind = int(rng.integers(0, 7))
                if kind == 0:
                    del lines[i]
                elif kind == 1:
                    lines.insert(i, lines[i])
                elif kind == 2:
                    lines[i] = lines[i][: len(lines[i]) // 2]
                elif kind == 3:
                    lines.insert(i, b"")
                elif kind == 4:
                    lines[i] = bytes([int(rng.choice(list(b"+@>")))]) + lines[i]
                elif kind == 5 and i + 1 < len(lines):
                    lines[i], lines[i + 1] = lines[i + 1], lines[i]
                else:
                    lines[i] = lines[i] + bytes(rng.integers(33, 126, 5, dtype=np.uint8))
            data = eol.join(lines)
            if rng.random() < 0.5 and len(data) > 4:  # single bytes overwritten: structure characters, NUL, high bytes, lone '\r' / '\n'
                buf = bytearray(data)
                for _ in range(int(rng.integers(1, 6))):
                    buf[int(rng.integers(1, len(buf)))] = int(rng.choice(list(b"\n\r>@+\x00\xff NacgtACGT")))
                data = bytes(buf)
            ok = False
        files.append(data)
        must_accept.append(ok)
    p, k = int(rng.choice([8, 10, 12])), int(rng.choice([5, 21, 31, 32]))
    ctx.alloc(len(files), p)
    status = ctx.sketch_fastx_batch(files, 0, k, True)
    got = ctx.download(0, len(files))
    for gi, f in enumerate(files):
        if must_accept[gi]:
            assert status[gi] == 0, "case %d genome %d: a well-formed file was refused" % (case, gi)
        if status[gi]:
            assert not got[gi].any(), "a refused genome must contribute nothing"
            continue
        s = host_parse(host, tmp_path, "z%d.fa" % gi, f)
        seq, off = synth.concat_for_device([s])
        want = oracle.sketch_batch(seq, off, k, p, True)[0]
        assert (got[gi] == want).all(), "case %d genome %d: accepted, but %d registers differ from the host parser's" % (case, gi, int((got[gi] != want).sum()))

"""GPU parity: HIP sketch kernel (through the C-ABI) vs the CPU oracle -- register arrays BIT-EXACT."""
import json
import os

import numpy as np
import pytest

import dashing_amd
from dashing_amd import synth
from hashinv import revcomp as _revcomp, unwang as _unwang  # (tests/hashinv.py)

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def run(ctx, oracle, genomes, k, p, canon=True):
    seq, off = synth.concat_for_device(genomes)
    ctx.alloc(len(genomes), p)
    got = ctx.sketch_batch(seq, off, 0, k, canon)
    want = oracle.sketch_batch(seq, off, k, p, canon)
    assert got.shape == want.shape
    assert (got == want).all(), "registers differ in %d places" % int((got != want).sum())
    return got


def test_golden_small_genome(ctx):
    with open(os.path.join(GOLD, "kat.json")) as f:
        s = json.load(f)["sketch_small"]
    g = np.frombuffer(s["seq"].encode(), np.uint8)
    seq, off = synth.concat_for_device([g])
    ctx.alloc(1, s["p"])
    got = ctx.sketch_batch(seq, off, 0, s["k"], s["canon"])
    assert got[0].tobytes().hex() == s["regs_hex"]


@pytest.mark.parametrize("k", [31, 32, 21, 5, 1])
@pytest.mark.parametrize("canon", [True, False])
def test_k_and_canon(ctx, oracle, k, canon):
    gs = synth.synthetic_genomes(5, 20011, seed=k * 7 + canon, decorate=True)
    run(ctx, oracle, gs, k, 10, canon)


@pytest.mark.parametrize("p", [4, 8, 10, 12, 14, 16, 17])
def test_precisions(ctx, oracle, p):
    gs = synth.synthetic_genomes(3, 50021, seed=p)
    run(ctx, oracle, gs, 31, p)


@pytest.mark.parametrize("p", [18, 20, 24])
def test_large_precisions_hbm_registers(ctx, oracle, p):
    """p > 17 does not fit LDS: the GLOBAL variant of k_sketch updates the registers in HBM directly.
    Same bit-exact contract; cardinalities too; the compare path takes these precisions as well (position bitmap in LDS
    up to p = 19, hash-only probing of the exception lists above)."""
    gs = synth.synthetic_genomes(2, 150013, seed=p, decorate=True)
    regs = run(ctx, oracle, gs, 31, p)
    got = ctx.cardinalities()
    want = oracle.cardinalities(regs)
    assert np.allclose(got, want, rtol=1e-12, atol=0)
    d = ctx.dist_rows(0, len(gs))
    ref = oracle.dist_tri(regs)
    assert np.allclose(d, ref, rtol=1e-6, atol=1e-15)


@pytest.mark.parametrize("k,p,canon", [(31, 10, True), (31, 14, True), (31, 8, False), (32, 12, True), (27, 13, True), (31, 16, True)])
def test_kmers_whose_hash_has_32_zero_bits_behind_the_index(ctx, oracle, k, p, canon):
    """Up to p = 14 k_sketch takes the register value from the HIGH word of t = (h << p) | guard alone and handles a zero
    high word -- 32 zero hash bits behind the index, 2^-32 of all k-mers: no random genome holds one -- behind the
    sub-chunk with the exact rule.  The hash is invertible, so such k-mers can be MADE: hashes idx << (64 - p) | low with
    low < 2^(32-p) are inverted until the preimage is a k-mer (below 4^k) and, with canonical k-mers, its own canonical
    form; they are planted in random genomes at positions that fall on every lane offset, on the first and last lanes of
    a workgroup's sub-chunk, beside an N, and twice in one genome.  Registers bit-exact against the oracle, and the
    oracle's registers show the planted values (>= 33, which nothing else produces).  p = 16 runs the packed-byte kernel
    over the same inputs."""
    import oracle.oracle_py as opy

    rng = np.random.default_rng(1000 * k + p)
    made = []
    while len(made) < 12:
        idx, low = int(rng.integers(0, 1 << p)), int(rng.integers(0, 1 << (32 - p)))
        h = (idx << (64 - p)) | low
        x = _unwang(h)
        assert opy.wang(x) == h
        if k < 32 and x >> (2 * k):
            continue
        if canon and _revcomp(x, k) < x:
            continue
        made.append((x, idx, (64 - p) - max(low, 0).bit_length() + 1 if low else 64 - p + 1))
    letters = np.frombuffer(b"ACGT", np.uint8)

    def text(x):
        return letters[[(x >> (2 * (k - 1 - t))) & 3 for t in range(k)]]

    genomes = []
    base = synth.synthetic_genomes(len(made), 70000, seed=p * 131 + k, decorate=False)
    spots = [0, 1, 31, 32, 8191 - k, 8192 - 5, 8192, 8192 + 31, 16384 - k + 1, 40001, 65536 - 3, 70000 - k]
    for g, ((x, idx, val), at) in enumerate(zip(made, spots)):
        a = base[g].copy()
        a[at:at + k] = text(x)
        if g % 3 == 1 and at > 0:
            a[at - 1] = ord("N")           # the k-mer begins right behind an invalid base
        if g % 4 == 2:
            a[300:300 + k] = text(made[(g + 1) % len(made)][0])  # a second one in the same genome
        genomes.append(a)
    got = run(ctx, oracle, genomes, k, p, canon)
    for g, (x, idx, val) in enumerate(made):
        assert val >= 33 and got[g][idx] == val, (g, idx, val, int(got[g][idx]))


@pytest.mark.parametrize("p", [13, 14, 15, 16, 17])
@pytest.mark.parametrize("k", [31, 21])
def test_workgroup_sizes_and_work_item_ends(ctx, oracle, p, k):
    """p = 14 / 16 run 512 lanes per workgroup and p = 15 / 17 run 1 024 (their registers leave room for two / one workgroup
    per CU): one step of the kernel's loop then covers 2 or 4 sub-chunks of the work list, and the lanes of a work item's
    last step that lie behind its end must start nothing (the next item does) while their bases remain the right
    neighbours of the lanes in front.  Genomes whose work items end on every position of such a step: 1 ... 5 sub-chunks,
    16 + 1, 17 whole ones, 33 and a ragged tail, several work items; a run of N across an item's end."""
    base = synth.synthetic_genomes(1, 450000, seed=77 + p, decorate=False)[0]
    sub = 8192
    lens = [sub - 7, sub, sub + 1, 2 * sub + 31, 3 * sub + 1, 4 * sub, 5 * sub - 17, 16 * sub, 16 * sub + 5000, 17 * sub, 33 * sub + 123, 450000]
    gs = [base[:n].copy() for n in lens]
    edge = base[:20 * sub].copy()
    edge[16 * sub - 40:16 * sub + 9] = ord("N")   # invalid windows on both sides of the first work item's end
    gs.append(edge)
    gs.append(base[13:13 + 16 * sub + 3].copy())   # unaligned start: the item grid is shifted against the genome
    run(ctx, oracle, gs, k, p)


def test_ragged_and_edge_genomes(ctx, oracle):
    """empty genome, shorter than k, exactly k, all-N, N every 31 bases, separators between records,
    lengths that straddle the 32/8192/131072-base chunk boundaries, unaligned offsets."""
    base = synth.synthetic_genomes(1, 300000, seed=99, decorate=False)[0]
    gs = [
        base[:0], base[:30], base[:31], base[:32], base[:33], np.full(500, ord("N"), np.uint8),
        base[:8191], base[:8192], base[:8193], base[:8222], base[:131071], base[:131073],
        base[:262147], base[5:70001], base[17:100], base.copy(),
    ]
    withn = base[:40000].copy()
    withn[::31] = ord("N")         # never 31 valid bases in a row -> no k-mers at k=31
    gs.append(withn)
    rec = base[:30000].copy()
    rec[10000] = ord(">")          # record separators (any non-ACGT byte)
    rec[20000:20003] = 0
    gs.append(rec)
    low = base[:25000].copy()
    low |= 0x20                    # all lowercase
    gs.append(low)
    got = run(ctx, oracle, gs, 31, 12)
    assert not got[0].any() and not got[1].any() and not got[5].any() and not got[16].any()
    assert got[2].any()
    assert (got[18] == run(ctx, oracle, [base[:25000]], 31, 12)[0]).all()


def test_merge_across_calls(ctx, oracle):
    """A genome fed in two calls (same slot) max-merges: equals sketching both record sets at once."""
    a, b = synth.synthetic_genomes(2, 40000, seed=5, decorate=False)
    ctx.alloc(1, 10)
    s1, o1 = synth.concat_for_device([a])
    s2, o2 = synth.concat_for_device([b])
    ctx.sketch_batch(s1, o1, 0, 31, True)
    got = ctx.sketch_batch(s2, o2, 0, 31, True)[0]
    both = np.concatenate([a, np.array([ord("N")], np.uint8), b])
    s3, o3 = synth.concat_for_device([both])
    want = oracle.sketch_batch(s3, o3, 31, 10, True)[0]
    assert (got == want).all()


def test_sketch_then_dist_c1_like(ctx, oracle):
    """BASELINE configs[0] in miniature: synthetic related genomes -> sketch -> all-pairs, vs oracle."""
    gs = synth.synthetic_genomes(24, 100000, seed=0xDA5410)
    seq, off = synth.concat_for_device(gs)
    ctx.alloc(len(gs), 10)
    regs = ctx.sketch_batch(seq, off, 0, 31, True)
    want_regs = oracle.sketch_batch(seq, off, 31, 10, True)
    assert (regs == want_regs).all()
    got = ctx.dist_rows()
    want = oracle.dist_tri(want_regs)
    assert np.allclose(got, want, rtol=1e-6, atol=1e-15)
    assert got.max() > 0.5 and got.min() < 0.05  # related and unrelated pairs both present


@pytest.mark.parametrize("case", range(int(os.environ.get("DSH_FUZZ_CASES_SKETCH", "50"))))
def test_random_sketch_case(ctx, oracle, case):
    """Seeded random sweep: k, p (LDS and HBM-register variants), canonical or not, ragged genome
    lengths (0 .. 70 000), random bytes from a dirty alphabet (N, lowercase, IUPAC, newline, NUL)."""
    rng = np.random.default_rng(7000 + case)
    k = int(rng.choice([1, 2, 7, 15, 16, 17, 21, 31, 32]))
    p = int(rng.choice([4, 5, 9, 10, 12, 13, 14, 15, 16, 17, 18, 21]))
    canon = bool(rng.integers(2))
    ng = int(rng.integers(1, 9))
    alphabet = np.frombuffer(b"ACGTACGTACGTACGTacgtNnRYKM\n\x00-", np.uint8)
    weights = np.ones(alphabet.size)
    weights[:16] = rng.choice([1.0, 30.0, 300.0])  # from "mostly dirty" to "almost clean"
    weights /= weights.sum()
    genomes = []
    for _ in range(ng):
        L = int(rng.choice([0, 1, k - 1, k, k + 1, 31, 32, 33, 63, 64, 65, 8191, 8192, 8193, int(rng.integers(0, 70000))]))
        genomes.append(alphabet[rng.choice(alphabet.size, size=max(L, 0), p=weights)].astype(np.uint8))
    run(ctx, oracle, genomes, k, p, canon)


def test_async_batches_from_pinned_staging(ctx, oracle):
    """dsh_sketch_batch_async: batches are enqueued back to back from page-locked staging (what the CLI's streaming
    loader does while it parses the next batch) and complete at dsh_wait; a genome may be split across batches."""
    k, p = 31, 12
    gs = synth.synthetic_genomes(9, 40_000, seed=0xA5)
    seq, off = synth.concat_for_device(gs)
    want = oracle.sketch_batch(seq, off, k, p, True)
    ctx.alloc(len(gs), p)
    stage = [dashing_amd.PinnedArray(seq.size + 64, np.uint8) for _ in range(2)]
    # batch A: genomes 0..4 ; batch B: genomes 5..8 -- each from its own staging buffer, offsets local to the buffer
    cut = int(off[5])
    stage[0].array[:cut] = seq[:cut]
    stage[1].array[: seq.size - cut] = seq[cut:]
    ctx.sketch_batch_async(stage[0].array, off[:6], 0, k, True)
    ctx.sketch_batch_async(stage[1].array, off[5:] - np.uint64(cut), 5, k, True)
    ctx.wait()
    assert (ctx.download() == want).all()
    # max-merge across asynchronous calls: the same genomes fed again in two halves change nothing
    half = int(off[2]) + 20_000
    oa = off[:4].copy()
    oa[3] = half
    ctx.sketch_batch_async(stage[0].array, oa, 0, k, True)
    ctx.wait()
    assert (ctx.download() == want).all()


def test_preload_loads_the_code_objects_without_a_context():
    """dsh_preload: no context, any thread; unknown bits are ignored, a device that does not exist is refused"""
    lib = dashing_amd.load_library()
    assert lib.dsh_preload(0, 3) == 0
    assert lib.dsh_preload(0, 0) == 0
    assert lib.dsh_preload(99, 1) != 0

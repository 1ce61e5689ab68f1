"""GPU: the BASELINE.json configurations at (or shaped like) their real sizes, through the C-ABI.

  configs[1]  5 Mbp genomes, k=31, p=10: 8 decorated genomes in ONE dsh_sketch_batch, registers bit-exact
              vs the oracle, then dist (src/sketch_and_cmp.h:314-360, 699-710)
  configs[2]  the full 10 000 x p=14 matrix: one dsh_dist_rows_device, sampled rows vs the oracle plus the
              size-independent properties (identical rows -> J == 1, permutation of the inputs)
  configs[3]  100 000 x p=10 and
  configs[4]  a 60 000-sketch slice of the 300 000 x p=14 job, both on ONE GPU through dsh_shard_plan with
              8 virtual ranks: spans assembled + un-permuted == the single-call matrix, byte for byte, and
              sampled rows vs the oracle.  (Real multi-GPU runs only happen in the driver's scaling bench.)
The big collections are built on the device from a base set drawn from the register law: sketch g =
max(base[a_g], base[b_g]) is exactly the sketch of the union of two base sets, so clusters, near-duplicates
and unrelated pairs all occur; the oracle gets the same bytes back from the device.
"""
import numpy as np
import pytest

import dashing_amd
from dashing_amd import synth

pytestmark = pytest.mark.gpu
RTOL = 1e-6


def close(got, ref):
    got = np.asarray(got, np.float64)
    ref = np.asarray(ref, np.float64)
    assert got.shape == ref.shape
    err = np.abs(got - ref)
    assert (err <= RTOL * np.maximum(np.abs(ref), 1e-9)).all(), float(err.max())


def derived_collection(torch, dev, n, p, nbase, seed):
    """n sketches on the device: the first nbase are drawn from the register law (cardinalities 2e6..8e6,
    SURVEY 8d), the rest are unions of two of them (a_g near g mod nbase so that clusters exist)."""
    base = synth.survey_sketches(nbase, p, seed=seed)[0]
    bd = torch.from_numpy(base).to(dev)
    regs = torch.empty((n, 1 << p), dtype=torch.uint8, device=dev)
    regs[:nbase] = bd
    g = torch.arange(nbase, n, device=dev, dtype=torch.int64)
    a = g % nbase
    b = (g * 2654435761 + 12345) % nbase
    step = 1 << 14
    for s in range(0, n - nbase, step):
        e = min(n - nbase, s + step)
        regs[nbase + s : nbase + e] = torch.maximum(bd[a[s:e]], bd[b[s:e]])
    torch.cuda.synchronize()
    return regs


def rows_vs_oracle(torch, oracle, regs_d, out_d, n, rows):
    regs_h = regs_d.cpu().numpy()
    for r in rows:
        want = oracle.dist_rows(regs_h, r, r + 1)
        lo = dashing_amd.tri_index(n, r, r + 1)
        close(out_d[lo : lo + want.size].cpu().numpy(), want)
    return regs_h


def equal_chunked(torch, a, b, chunk=1 << 28):
    for s in range(0, a.numel(), chunk):
        if not torch.equal(a[s : s + chunk], b[s : s + chunk]):
            return False
    return True


def shards_equal_single(torch, ctx, regs_d, n, p, nshards, dev):
    total = n * (n - 1) // 2
    ctx.attach_device(regs_d.data_ptr(), n, p)
    single = torch.empty(total, dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    ctx.dist_rows_device(single.data_ptr(), 0, n)
    ctx.synchronize()
    off = ctx.shard_plan(nshards)
    assert off[0] == 0 and off[-1] == total and all(off[r] <= off[r + 1] for r in range(nshards))
    sorted_full = torch.empty(total, dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    for r in range(nshards):
        ctx.dist_shard_device(sorted_full.data_ptr() + 4 * off[r], r, nshards)
    ctx.synchronize()
    final = torch.empty(total, dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    ctx.unpermute_device(sorted_full.data_ptr(), final.data_ptr())
    ctx.synchronize()
    del sorted_full
    assert equal_chunked(torch, single, final)
    # largest shard vs the mean: the plan balances cost, not pairs, but must not be degenerate
    spans = [off[r + 1] - off[r] for r in range(nshards)]
    assert max(spans) < 3 * total / nshards
    return single


def test_config1_5mbp_genomes_sketch_then_dist(ctx, oracle):
    """configs[1] shape: 5 Mbp genomes, k=31, p=10 (every 10th genome carries an N run and a lowercase stretch)"""
    k, p = 31, 10
    gs = synth.synthetic_genomes(8, 5_000_000, seed=0xDA5410)
    gs += [g.copy() for g in synth.synthetic_genomes(2, 5_000_000, seed=0xDA5411)]
    seq, off = synth.concat_for_device(gs)
    assert seq.size == 50_000_000
    ctx.alloc(len(gs), p)
    regs = ctx.sketch_batch(seq, off, 0, k, True)
    want = oracle.sketch_batch(seq, off, k, p, True)
    assert (regs == want).all(), "registers differ from the oracle"
    # non-canonical too (the -C path), fed in two calls that split one genome (max-merge across calls)
    ctx.alloc(len(gs), p)
    cut = int(off[3]) + 2_345_678
    off_a = off.copy()
    off_a[4:] = cut                                  # genomes 0..2 whole, genome 3 up to `cut`, rest empty
    ctx.sketch_batch(seq, off_a, 0, k, False, want_regs=False)
    off_b = off.copy()
    off_b[:4] = cut - (k - 1)                        # the remainder of genome 3 (overlapping k-1 bases), then 4..9
    regs2 = ctx.sketch_batch(seq, off_b, 0, k, False)
    assert (regs2 == oracle.sketch_batch(seq, off, k, p, False)).all()
    ctx.set_sketches(want)
    for rt in (dashing_amd.JI, dashing_amd.MASH_DIST):
        close(ctx.dist_rows(result_type=rt, k=k), oracle.dist_tri(want, oracle.ERTL_MLE, rt, k))
    cards = ctx.cardinalities()
    assert ((cards > 4.0e6) & (cards < 6.0e6)).all()  # ~5e6 distinct 31-mers each


def test_config2_full_c3_matrix(ctx, oracle):
    """configs[2] at full size: N = 10 000, p = 14, the workload bench.py times"""
    import torch

    n, p = 10_000, 14
    dev = torch.device("cuda", 0)
    regs = synth.survey_sketches(n, p, seed=0x5EED0000)[0]
    regs[7777] = regs[123]  # identical pair
    regs_d = torch.from_numpy(regs).to(dev)
    total = n * (n - 1) // 2
    out = torch.empty(total, dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    ctx.attach_device(regs_d.data_ptr(), n, p)
    ctx.dist_rows_device(out.data_ptr(), 0, n)
    ctx.synchronize()
    assert bool(torch.isfinite(out).all()) and float(out.min()) >= 0.0 and float(out.max()) <= 1.0
    assert float(out[dashing_amd.tri_index(n, 123, 7777)]) == 1.0
    for r in (0, 1, 123, 4999, 7777, 9000, 9998):
        want = oracle.dist_rows(regs, r, r + 1)
        lo = dashing_amd.tri_index(n, r, r + 1)
        close(out[lo : lo + want.size].cpu().numpy(), want)
    # rows 123 and 7777 hold the same sketch: their distances to every third sketch agree exactly
    o = out.cpu().numpy()
    for j in (5, 124, 5000, 7776, 7778, 9999):
        a = o[dashing_amd.tri_index(n, min(123, j), max(123, j))]
        b = o[dashing_amd.tri_index(n, min(7777, j), max(7777, j))]
        assert a == b
    # permutation of the inputs: value(perm(i), perm(j)) is unchanged
    rng = np.random.default_rng(5)
    perm = rng.permutation(n)
    regs_p = torch.from_numpy(np.ascontiguousarray(regs[perm])).to(dev)
    out_p = torch.empty(total, dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    ctx.attach_device(regs_p.data_ptr(), n, p)
    ctx.dist_rows_device(out_p.data_ptr(), 0, n)
    ctx.synchronize()
    op_ = out_p.cpu().numpy()
    inv = np.empty(n, np.int64)
    inv[perm] = np.arange(n)                      # sketch s sits at position inv[s] of the permuted input
    ii = rng.integers(0, n, 200_000)
    jj = rng.integers(0, n, 200_000)
    keep = ii != jj
    ii, jj = ii[keep], jj[keep]
    lo_, hi_ = np.minimum(ii, jj), np.maximum(ii, jj)
    idx = lo_ * (2 * n - lo_ - 1) // 2 + hi_ - (lo_ + 1)
    pi, pj = inv[ii], inv[jj]
    plo, phi = np.minimum(pi, pj), np.maximum(pi, pj)
    pidx = plo * (2 * n - plo - 1) // 2 + phi - (plo + 1)
    assert (o[idx] == op_[pidx]).all()


def test_config3_shape_100k_p10_virtual_shards(ctx, oracle):
    """configs[3] shape: 100 000 sketches, p = 10, triangle cut into 8 shards as on 8 GPUs"""
    import torch

    n, p = 100_000, 10
    dev = torch.device("cuda", 0)
    regs_d = derived_collection(torch, dev, n, p, 20_000, seed=0x5EED0000)
    single = shards_equal_single(torch, ctx, regs_d, n, p, 8, dev)
    rows_vs_oracle(torch, oracle, regs_d, single, n, (0, 19_999, 20_000, 77_777, n - 2))
    assert bool(torch.isfinite(single[: 1 << 28]).all())


def test_config4_shape_p14_slice_virtual_shards(ctx, oracle):
    """configs[4] shape: p = 14 at a 60 000-sketch slice of the 300 000 job (1.8e9 pairs), 8 virtual ranks"""
    import torch

    n, p = 60_000, 14
    dev = torch.device("cuda", 0)
    regs_d = derived_collection(torch, dev, n, p, 4_000, seed=0x5EED1000)
    single = shards_equal_single(torch, ctx, regs_d, n, p, 8, dev)
    rows_vs_oracle(torch, oracle, regs_d, single, n, (0, 3_999, 4_000, n - 2))


# ---- BASELINE configs at their STATED sizes ---------------------------------------------------
def _genomes_on_device(torch, dev, n, length, seed, cluster=10):
    """SURVEY 8d phylogeny generated on the GPU (5 Gbases on the host would take minutes in numpy): root -> cluster
    ancestors (5 % substitutions) -> members with divergence cycling through {0.1, 0.5, 1, 2, 5} %; every 10th genome
    carries a run of 50 N and a lowercase 1 kb stretch.  Returns a uint8 ASCII tensor [n][length] on `dev`."""
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    acgt = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)

    def mutate(codes, rate):
        hit = torch.rand(codes.shape, generator=g, device=dev) < rate
        shift = torch.randint(1, 4, codes.shape, generator=g, device=dev, dtype=torch.uint8)
        return torch.where(hit, (codes + shift) & 3, codes)

    root = torch.randint(0, 4, (length,), generator=g, device=dev, dtype=torch.uint8)
    rates = (0.001, 0.005, 0.01, 0.02, 0.05)
    out = torch.empty((n, length), dtype=torch.uint8, device=dev)
    anc = None
    for i in range(n):
        if i % cluster == 0:
            anc = mutate(root, 0.05)
        s = acgt[mutate(anc, rates[i % len(rates)]).long()]
        if i % 10 == 0:
            a, b = length // 3, (2 * length) // 3
            s[a : a + 50] = ord("N")
            s[b : b + 1000] |= 0x20
        out[i] = s
    torch.cuda.synchronize()
    return out


def test_config0_full_100x1mbp_cli(oracle, tmp_path):
    """configs[0] at its stated size: 100 synthetic 1 Mbp genomes as FASTA files through `dashing-amd dist` (default
    upper-triangular TSV and -b), k=31, p=10 (-S10), against the oracle's registers and distances
    (src/sketch_and_cmp.h:314-360,699-710,838-849)."""
    import gzip
    import os
    import struct
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cli = os.path.join(root, "dashing_amd", "dashing-amd")
    n, L, k, p = 100, 1_000_000, 31, 10
    gs = synth.synthetic_genomes(n, L, seed=0xDA5410)
    paths = []
    for i, g in enumerate(gs):
        pth = tmp_path / ("g%03d.fna" % i)
        pth.write_bytes(synth.to_fasta(g, "genome%d" % i))
        paths.append(str(pth))
    lst = tmp_path / "paths.txt"
    lst.write_text("\n".join(paths) + "\n")
    seq, off = synth.concat_for_device(gs)
    regs = oracle.sketch_batch(seq, off, k, p, True)
    want = oracle.dist_tri(regs)

    def run(*args):
        r = subprocess.run([cli] + [str(a) for a in args], capture_output=True, timeout=600)
        assert r.returncode == 0, r.stderr.decode()[-2000:]

    cache = tmp_path / "cache"
    cache.mkdir()
    b = tmp_path / "d.bin"
    run("dist", "-k", k, "-S", p, "-p", 16, "-b", "-W", "-P", cache, "--avoid-sorting", "-O", b, "-o", tmp_path / "sizes.txt", "-F", lst)
    raw = b.read_bytes()
    assert raw[0] == 0 and struct.unpack("<Q", raw[1:9])[0] == n and len(raw) == 9 + 4 * n * (n - 1) // 2
    close(np.frombuffer(raw[9:], np.float32), want)
    for i, pth in enumerate(paths):  # the cached .hll payloads are the oracle's registers, bit for bit
        h = gzip.open(str(cache / (os.path.basename(pth) + ".w.31.spacing.10.hll"))).read()
        assert h[28:] == regs[i].tobytes()
    card = oracle.cardinalities(regs)
    lines = (tmp_path / "sizes.txt").read_text().split("\n")
    for i, pth in enumerate(paths):
        nm, v = lines[1 + i].split("\t")
        assert nm == pth and abs(int(v) - int(card[i])) <= 1
    t = tmp_path / "d.tsv"
    run("dist", "-k", k, "-S", p, "-p", 16, "--avoid-sorting", "-O", t, "-o", os.devnull, "-F", lst)
    rows = t.read_text().split("\n")
    assert rows[0] == "##Names\t" + "\t".join(paths)
    vals = []
    for i in range(n):
        f = rows[1 + i].split("\t")
        assert f[0] == paths[i] and f[1 : 2 + i] == ["-"] * (i + 1) and len(f) == n + 1
        vals += [float(x) for x in f[2 + i :]]
    exp = np.array([float("%.6g" % x) for x in want])
    assert np.allclose(np.array(vals), exp, rtol=2e-6, atol=1e-12)


def test_config1_full_1000x5mbp_streamed(ctx, oracle):
    """configs[1] at its stated size: 1 000 synthetic 5 Mbp genomes (5 Gbases), k=31, p=10, through the loader's
    streaming path -- page-locked batches, dsh_sketch_batch_async, the next batch staged while the previous one is
    sketched (src/sketch_and_cmp.h:314-360) -- registers bit-exact vs the oracle, then all pairs vs the oracle."""
    import torch

    n, L, k, p = 1000, 5_000_000, 31, 10
    dev = torch.device("cuda", 0)
    per = 25                                     # genomes per batch: 125 MB, the CLI's batch size
    seq_h = np.empty(n * L, np.uint8)            # what the oracle reads
    off = (np.arange(n + 1, dtype=np.uint64) * np.uint64(L))
    pins = [dashing_amd.PinnedArray(per * L + 256, np.uint8) for _ in range(2)]
    ctx.alloc(n, p)
    for b0 in range(0, n, per):                  # generated cluster by cluster on the GPU (10 genomes per cluster)
        if b0 % 50 == 0:
            gen = _genomes_on_device(torch, dev, 50, L, seed=0xDA5410 + b0)
        pin = pins[(b0 // per) & 1]
        if b0 >= 2 * per:
            ctx.wait()                           # (both in-flight batches: simpler than per-buffer events here)
        blk = gen[(b0 % 50) : (b0 % 50) + per].reshape(-1)
        view = torch.from_numpy(pin.array[: per * L])
        view.copy_(blk)
        torch.cuda.synchronize()
        seq_h[b0 * L : (b0 + per) * L] = pin.array[: per * L]
        ctx.sketch_batch_async(pin.array, np.arange(per + 1, dtype=np.uint64) * np.uint64(L), b0, k, True)
    ctx.wait()
    regs = ctx.download(0, n)
    want = oracle.sketch_batch(seq_h, off, k, p, True)
    assert (regs == want).all(), "registers differ from the oracle"
    del seq_h
    close(ctx.dist_rows(result_type=dashing_amd.JI, k=k), oracle.dist_tri(want, oracle.ERTL_MLE, dashing_amd.JI, k))
    close(ctx.dist_rows(result_type=dashing_amd.MASH_DIST, k=k), oracle.dist_tri(want, oracle.ERTL_MLE, dashing_amd.MASH_DIST, k))
    cards = ctx.cardinalities()
    assert ((cards > 4.0e6) & (cards < 6.0e6)).all()


def _checksums(torch, buf, count, base_index, cuts):
    """Order-sensitive checksums of buf[0:count) = elements [base_index, base_index+count) of the packed matrix, split at
    the global element offsets `cuts` (sorted): returns {(lo, hi): (sum of bit patterns, position-weighted sum)} with
    wrap-around int64 arithmetic, so pieces add up whatever the split."""
    res = {}
    edges = [base_index] + [c for c in cuts if base_index < c < base_index + count] + [base_index + count]
    for lo, hi in zip(edges[:-1], edges[1:]):
        s0 = torch.zeros((), dtype=torch.int64, device=buf.device)
        s1 = torch.zeros((), dtype=torch.int64, device=buf.device)
        step = 1 << 27
        for a in range(lo, hi, step):
            e = min(hi, a + step)
            bits = buf[a - base_index : e - base_index].view(torch.int32).to(torch.int64)
            w = torch.arange(a, e, device=buf.device, dtype=torch.int64) % 8191 + 1
            s0 += bits.sum()
            s1 += (bits * w).sum()
        res[(lo, hi)] = (int(s0.item()), int(s1.item()))
    return res


def test_config4_full_300k_p14_eight_ranges(ctx, oracle):
    """configs[4] at its stated size: 300 000 sketches, p = 14 (45e9 pairs, 180 GB of float32), on ONE GPU the way
    bench.py's N-rank scheme runs it: the 8 row ranges of dsh_balance_rows, each one dsh_dist_rows_device call into
    its own span buffer.  Every span is checksummed; a second pass over 11 differently cut, unaligned ranges (another
    plane layout, other tiles) must reproduce the same checksums piece by piece; sampled rows vs the oracle; all finite."""
    import torch

    n, p = 300_000, 14
    dev = torch.device("cuda", 0)
    regs_d = derived_collection(torch, dev, n, p, 4_000, seed=0x5EED3000)
    ctx.attach_device(regs_d.data_ptr(), n, p)
    bounds = dashing_amd.balance_rows(n, 8)
    assert bounds[0] == 0 and bounds[-1] == n and all(b % 128 == 0 for b in bounds[:-1])
    spans = [dashing_amd.tri_span(n, bounds[r], bounds[r + 1]) for r in range(8)]
    assert sum(spans) == n * (n - 1) // 2 and max(spans) < 1.35 * min(spans)
    cuts2 = dashing_amd.partition_rows(n, 11, 1)
    cuts2[5] += 77                                # make sure the second pass is not tile-aligned
    offs1 = [dashing_amd.tri_span(n, 0, b) for b in bounds]
    offs2 = [dashing_amd.tri_span(n, 0, b) for b in cuts2]
    edges = sorted(set(offs1 + offs2))
    buf = torch.empty(max(spans + [dashing_amd.tri_span(n, cuts2[r], cuts2[r + 1]) for r in range(11)]), dtype=torch.float32, device=dev)
    regs_h = regs_d.cpu().numpy()
    first, second = {}, {}
    rng = np.random.default_rng(4)
    for r in range(8):
        torch.cuda.synchronize()
        ctx.dist_rows_device(buf.data_ptr(), bounds[r], bounds[r + 1])
        ctx.synchronize()
        span = buf[: spans[r]]
        for s in range(0, spans[r], 1 << 30):
            assert bool(torch.isfinite(span[s : s + (1 << 30)]).all())
        first.update(_checksums(torch, buf, spans[r], offs1[r], edges))
        for row in (bounds[r], int(rng.integers(bounds[r], bounds[r + 1]))):  # 16 sampled rows vs the oracle
            want = oracle.dist_rows(regs_h, row, row + 1)
            lo = dashing_amd.tri_index(n, row, row + 1) - offs1[r] if row + 1 < n else 0
            close(span[lo : lo + want.size].cpu().numpy(), want)
    for r in range(11):
        torch.cuda.synchronize()
        ctx.dist_rows_device(buf.data_ptr(), cuts2[r], cuts2[r + 1])
        ctx.synchronize()
        second.update(_checksums(torch, buf, offs2[r + 1] - offs2[r], offs2[r], edges))
    assert first.keys() == second.keys() and len(first) == len(edges) - 1
    assert first == second


def test_config4_knn_300k_without_the_square_matrix(ctx, oracle):
    """SURVEY 8f-2 at configs[4] size: 10 nearest neighbours of each of 300 000 sketches (p = 14).  The n x n matrix
    would be 360 GB; dsh_knn computes the triangle once in bands and keeps only the running lists
    (src/sketch_and_cmp.h:642-783).  Sampled queries against the oracle (ties by the lower index: the collection
    holds exact duplicates), every list complete, sorted, and free of the query itself."""
    import time

    import torch

    n, p, nn = 300_000, 14, 10
    dev = torch.device("cuda", 0)
    regs_d = derived_collection(torch, dev, n, p, 4_000, seed=0x5EED3000)
    ctx.attach_device(regs_d.data_ptr(), n, p)
    t0 = time.perf_counter()
    gi, gv = ctx.knn(nn)
    dt = time.perf_counter() - t0
    assert (gi != 0xFFFFFFFF).all() and (gi != np.arange(n, dtype=np.uint32)[:, None]).all()
    assert (np.diff(gv.astype(np.float64), axis=1) <= 0).all()  # Jaccard: best first
    regs_h = regs_d.cpu().numpy()
    for q in (0, 3_999, 4_000, 123_456, 299_999):
        wi, wv = oracle.knn(regs_h, nn, qb=q, qe=q + 1, rb=0, re=n)
        assert (gi[q] == wi[0]).all(), (q, gi[q], wi[0])
        assert np.allclose(gv[q], wv[0], rtol=1e-6, atol=1e-12)
    print("knn 300k x p14 x nn=10: %.1f s" % dt)
    assert dt < 60.0  # about one triangle pass (~15 s); the query-block fallback computes every pair twice

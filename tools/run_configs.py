#!/usr/bin/env python3
"""End-to-end checks of BASELINE.json configs on a GPU box (not part of the test suite: minutes).
  c1: configs[0] 100 synthetic 1 Mbp genomes, k=31, p=10: FASTA files -> `dashing-amd dist` vs the CPU oracle
  c4: configs[3]-shaped 100 000 sketches, p=10, full matrix on one GPU: timing + sampled rows vs the oracle
Prints one JSON line per config."""
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np
import torch  # noqa: F401  (before dashing_amd: its HIP runtime must be the first one this process loads)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dashing_amd  # noqa: E402
from dashing_amd import synth  # noqa: E402
from oracle import oracle_c  # noqa: E402


def c1():
    n, L, k, p = 100, 1_000_000, 31, 10
    gs = synth.synthetic_genomes(n, L, seed=0xDA5410)
    d = tempfile.mkdtemp(prefix="c1_")
    paths = []
    for i, g in enumerate(gs):
        pth = os.path.join(d, "g%03d.fna" % i)
        open(pth, "wb").write(synth.to_fasta(g, "g%d" % i))
        paths.append(pth)
    lst = os.path.join(d, "paths.txt")
    open(lst, "w").write("\n".join(paths) + "\n")
    out = os.path.join(d, "dist.bin")
    cli = os.path.join(ROOT, "dashing_amd", "dashing-amd")
    t0 = time.perf_counter()
    subprocess.check_call([cli, "dist", "-k", str(k), "-S", str(p), "-p", "8", "-b", "--avoid-sorting", "-O", out, "-o", os.devnull, "-F", lst])
    t_cli = time.perf_counter() - t0
    got = np.frombuffer(open(out, "rb").read()[9:], np.float32)
    seq, off = synth.concat_for_device(gs)
    oracle_c.load(threads=oracle_c.effective_cpus())
    t0 = time.perf_counter()
    regs = oracle_c.sketch_batch(seq, off, k, p, True)
    want = oracle_c.dist_tri(regs)
    t_cpu = time.perf_counter() - t0
    rel = np.abs(got.astype(np.float64) - want) / np.maximum(np.abs(want), 1e-9)
    print(json.dumps({"config": "C1: 100 x 1 Mbp FASTA, k=31, p=10, dashing-amd dist -b (end to end incl. FASTA parsing and process start)",
                      "cli_seconds": t_cli, "cpu_oracle_seconds_sketch_plus_dist": t_cpu, "cores": oracle_c.effective_cpus(),
                      "pairs": int(got.size), "max_rel_diff": float(rel.max()), "exact_matches": int((got == want).sum()),
                      "j_range": [float(want.min()), float(want.max())]}))


def c4(n=100_000, p=10):
    import torch

    t0 = time.perf_counter()
    dev = torch.device("cuda", 0)
    if n <= 100_000:
        regs = synth.survey_sketches(n, p, seed=0x5EED0000)[0]
        regs_d = torch.from_numpy(regs).to(dev)
    else:
        # drawing 300 000 register arrays from the law takes minutes of numpy: draw 20 000 and build the rest on the
        # device as unions of two of them (exactly the sketch of the union of the two sets), as tests/test_gpu_configs.py
        nbase = 20_000
        bd = torch.from_numpy(synth.survey_sketches(nbase, p, seed=0x5EED0000)[0]).to(dev)
        regs_d = torch.empty((n, 1 << p), dtype=torch.uint8, device=dev)
        regs_d[:nbase] = bd
        g = torch.arange(nbase, n, device=dev, dtype=torch.int64)
        a_, b_ = g % nbase, (g * 2654435761 + 12345) % nbase
        for s_ in range(0, n - nbase, 1 << 14):
            e_ = min(n - nbase, s_ + (1 << 14))
            regs_d[nbase + s_ : nbase + e_] = torch.maximum(bd[a_[s_:e_]], bd[b_[s_:e_]])
        del bd
        torch.cuda.synchronize()
        regs = None
    t_gen = time.perf_counter() - t0
    total = n * (n - 1) // 2
    out = torch.empty(total, dtype=torch.float32, device=dev)
    ctx = dashing_amd.Context(0)
    for kv in filter(None, os.environ.get("DSH_BENCH_OPTS", "").split(",")):  # tuning sweeps, e.g. "emax=4"
        k_, v_ = kv.split("=")
        ctx.set_option(k_, int(v_))
    ctx.set_profiling(True)
    times = []
    for _ in range(2):
        ctx.attach_device(regs_d.data_ptr(), n, p)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ctx.dist_rows_device(out.data_ptr(), 0, n)
        ctx.synchronize()
        times.append(time.perf_counter() - t0)
    k = ctx.last_kernel_ms()
    oracle_c.load(threads=oracle_c.effective_cpus())
    worst, checked = 0.0, 0
    if regs is None:
        regs = regs_d.cpu().numpy()
    for r in (0, 1, 4999, n // 2, n - 1000, n - 2):
        want = oracle_c.dist_rows(regs, r, r + 1)
        lo = dashing_amd.tri_index(n, r, r + 1)
        got = out[lo : lo + want.size].cpu().numpy()
        rel = np.abs(got.astype(np.float64) - want) / np.maximum(np.abs(want), 1e-9)
        worst = max(worst, float(rel.max()) if rel.size else 0.0)
        checked += want.size
    fin = True
    for c0 in range(0, total, 1 << 30):  # chunked: a 45e9-element temporary would not fit
        fin = fin and bool(torch.isfinite(out[c0 : c0 + (1 << 30)]).all().item())
    print(json.dumps({"config": "%d sketches, p=%d, full triangle on one MI355X (output left in HBM: %.1f GB)" % (n, p, total * 4 / 1e9),
                      "seconds": min(times), "pairs_per_s": total / min(times), "kernel_ms": k,
                      "planes": ctx.info("planes"), "avg_planes_per_tile": ctx.info("avg_tile_planes_x100") / 100,
                      "rows_checked_vs_oracle_pairs": checked, "max_rel_diff": worst, "all_finite": fin, "gen_seconds": t_gen}))
    ctx.close()


def c2lite(n=200, L=5_000_000, k=31, p=10):
    """configs[1]-shaped, scaled to 200 genomes: 5 Mbp FASTA files -> `dashing-amd dist` (parse + sketch + dist)."""
    import torch

    d = tempfile.mkdtemp(prefix="c2_")
    paths = []
    t0 = time.perf_counter()
    lut = np.frombuffer(b"ACGT", np.uint8)
    rng = np.random.default_rng(1)
    base = lut[rng.integers(0, 4, L)]
    for i in range(n):
        g = base.copy()
        pos = rng.integers(0, L, L // 100)       # 1 % substitutions per genome
        g[pos] = lut[rng.integers(0, 4, pos.size)]
        pth = os.path.join(d, "g%04d.fna" % i)
        open(pth, "wb").write(synth.to_fasta(g, "g%d" % i))
        paths.append(pth)
    t_gen = time.perf_counter() - t0
    lst = os.path.join(d, "paths.txt")
    open(lst, "w").write("\n".join(paths) + "\n")
    cli = os.path.join(ROOT, "dashing_amd", "dashing-amd")
    res = {}
    for thr in (16, 1):
        out = os.path.join(d, "dist%d.bin" % thr)
        time.sleep(1.0)  # let the driver finish tearing down the previous process (its dsh_create otherwise takes 0.24 s instead of 0.07)
        t0 = time.perf_counter()
        subprocess.check_call([cli, "dist", "-k", str(k), "-S", str(p), "-p", str(thr), "-b", "--avoid-sorting", "-O", out, "-o", os.devnull, "-F", lst])
        res["cli_seconds_p%d" % thr] = time.perf_counter() - t0
    res["bases"] = n * L
    res["bases_per_s_end_to_end_p16"] = n * L / res["cli_seconds_p16"]
    print(json.dumps({"config": "C2-shaped: %d x %d bp FASTA (81-byte lines), k=%d, p=%d, dashing-amd dist -b end to end" % (n, L, k, p), **res, "gen_seconds": t_gen}))


def c3_host(n=10_000, p=14):
    """C3 through the HOST-buffer boundary (what a patched dashing would call): dsh_upload_sketches
    + dsh_dist_rows into pageable host memory, i.e. PCIe both ways included."""
    regs = synth.survey_sketches(n, p, seed=0x5EED0000)[0]
    ctx = dashing_amd.Context(0)
    best_up, best_dist = 1e9, 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        ctx.set_sketches(regs)
        t1 = time.perf_counter()
        out = ctx.dist_rows()
        t2 = time.perf_counter()
        best_up, best_dist = min(best_up, t1 - t0), min(best_dist, t2 - t1)
    total = n * (n - 1) // 2
    pin = dashing_amd.PinnedArray(total)
    best_pin = 1e9
    for _ in range(3):
        t1 = time.perf_counter()
        ctx.dist_rows(out=pin.array)
        best_pin = min(best_pin, time.perf_counter() - t1)
    assert (pin.array[:total] == out).all()
    print(json.dumps({"config": "C3 via host buffers (PCIe-inclusive): upload %d MB + all-pairs + download %d MB" % (regs.nbytes >> 20, out.nbytes >> 20),
                      "upload_s": best_up, "dist_rows_s": best_dist, "pairs_per_s_pcie_inclusive": total / (best_up + best_dist),
                      "dist_rows_pinned_out_s": best_pin, "pairs_per_s_pcie_inclusive_pinned": total / (best_up + best_pin)}))
    ctx.close()


def c3_cli_text(n=10_000, p=14):
    """`dashing-amd dist --presketched` on 10 000 .hll files, default text output (upper-triangular TSV):
    end-to-end wall time incl. reading sketches and formatting 5e7 numbers on the host."""
    regs = synth.survey_sketches(n, p, seed=0x5EED0000)[0]
    d = tempfile.mkdtemp(prefix="c3hll_")
    import ctypes as C
    host = C.CDLL(os.path.join(ROOT, "dashing_amd", "libdashing_host.so"))
    host.dshh_write_hll.argtypes = [C.c_char_p, C.c_void_p, C.c_int, C.c_int]
    paths = []
    for i in range(n):
        pth = os.path.join(d, "s%05d.hll" % i)
        assert host.dshh_write_hll(pth.encode(), regs[i].ctypes.data, p, 2) == 0
        paths.append(pth)
    lst = os.path.join(d, "paths.txt")
    open(lst, "w").write("\n".join(paths) + "\n")
    cli = os.path.join(ROOT, "dashing_amd", "dashing-amd")
    res = {}
    for name, flags in (("ut_tsv", []), ("binary", ["-b"])):
        out = os.path.join(d, "out." + name)
        t0 = time.perf_counter()
        subprocess.check_call([cli, "dist", "--presketched", "-S", str(p), "-p", "16", "-O", out, "-o", os.devnull, "-F", lst] + flags)
        res[name + "_seconds"] = time.perf_counter() - t0
        res[name + "_bytes"] = os.path.getsize(out)
    print(json.dumps({"config": "C3 end to end through the CLI: %d presketched .hll files (p=%d) -> dist, 16 host threads" % (n, p), **res}))


def c3_knn(n=10_000, p=14, nn=10):
    """All-vs-all --nearest-neighbors on the C3 matrix: every pair computed once into an n x n matrix in
    HBM + one selection pass per row, vs the query-block fallback that computes the full square."""
    regs = synth.survey_sketches(n, p, seed=0x5EED0000)[0]
    ctx = dashing_amd.Context(0)
    ctx.set_sketches(regs)
    res = {}
    for name, budget in (("square", 96 << 30), ("query_blocks", 0)):
        ctx.set_option("knn_square_budget_bytes", budget)
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            gi, gv = ctx.knn(nn)
            best = min(best, time.perf_counter() - t0)
        res[name + "_s"] = best
        res[name] = (gi, gv)
    same = bool((res["square"][0] == res["query_blocks"][0]).all() and (res["square"][1] == res["query_blocks"][1]).all())
    print(json.dumps({"config": "C3 kNN: %d sketches p=%d, %d nearest neighbours each (Jaccard, Ertl-MLE)" % (n, p, nn),
                      "square_seconds": res["square_s"], "query_block_seconds": res["query_blocks_s"], "identical": same}))
    ctx.close()


def unfriendly(n=10_000, p=14):
    """C3-sized collections the headline workload is NOT: (a) cardinalities log-uniform over 1e4..1e8 (a RefSeq-like
    spread: thresholds and minima differ a lot between sketches), (b) registers uniform over [0, q+1] (the adversary of
    the thermometer planes: every value level is populated), (c) the bench workload for reference.  Records planes per
    tile, pairs/s and the kernel split; a few rows are checked against the oracle."""
    import torch

    rng = np.random.default_rng(77)
    dev = torch.device("cuda", 0)
    cases = []
    cards = np.exp(rng.uniform(np.log(1e4), np.log(1e8), n))
    t0 = time.perf_counter()
    cases.append(("log-uniform cardinalities 1e4..1e8", np.stack([synth.hll_registers(1000 + g, int(c), p) for g, c in enumerate(cards)])))
    cases.append(("uniform registers 0..%d (adversary)" % (64 - p + 1), rng.integers(0, 64 - p + 2, size=(n, 1 << p)).astype(np.uint8)))
    cases.append(("bench workload (cardinalities 2e6..8e6)", synth.survey_sketches(n, p, seed=0x5EED0000)[0]))
    t_gen = time.perf_counter() - t0
    oracle_c.load(threads=oracle_c.effective_cpus())
    total = n * (n - 1) // 2
    out = torch.empty(total, dtype=torch.float32, device=dev)
    ctx = dashing_amd.Context(0)
    ctx.set_profiling(True)
    for name, regs in cases:
        regs_d = torch.from_numpy(regs).to(dev)
        best = 1e9
        for _ in range(3):
            ctx.attach_device(regs_d.data_ptr(), n, p)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ctx.dist_rows_device(out.data_ptr(), 0, n)
            ctx.synchronize()
            best = min(best, time.perf_counter() - t0)
        k = ctx.last_kernel_ms()
        worst = 0.0
        for r in (0, n // 3, n - 2):
            want = oracle_c.dist_rows(regs, r, r + 1)
            lo = dashing_amd.tri_index(n, r, r + 1)
            got = out[lo : lo + want.size].cpu().numpy()
            worst = max(worst, float((np.abs(got.astype(np.float64) - want) / np.maximum(np.abs(want), 1e-9)).max()))
        print(json.dumps({"config": "%d sketches p=%d, %s" % (n, p, name), "seconds": best, "pairs_per_s": total / best,
                          "kernel_ms": k, "dense_planes_global": ctx.info("planes"), "avg_planes_per_tile": ctx.info("avg_tile_planes_x100") / 100,
                          "reg_value_range": [ctx.info("vlo"), ctx.info("vhi")], "max_rel_diff_vs_oracle_3_rows": worst}))
    ctx.close()


if __name__ == "__main__":
    which = sys.argv[1:] or ["c1", "c4"]
    if "unfriendly" in which:
        unfriendly()
    if "c3knn" in which:
        c3_knn()
    if "c3cli" in which:
        c3_cli_text()
    if "c2lite" in which:
        c2lite()
    if "c2" in which:  # configs[1] at full size: 1 000 genomes of 5 Mbp
        c2lite(n=1000)
    if "c3host" in which:
        c3_host()
    if "c1" in which:
        c1()
    if "c4" in which:
        c4()
    if "c5" in which:  # configs[4]-shaped: 300 000 sketches, p=14, 4.5e10 pairs, 180 GB of output in HBM
        c4(n=300_000, p=14)

"""Deterministic synthetic workloads: genomes with a cluster phylogeny
and register arrays drawn from the HLL register law.  Pure numpy; used by tests and bench.py
to build inputs -- not part of the compute path."""
import numpy as np

_GOLD = np.uint64(0x9E3779B97F4A7C15)


def splitmix64(seed, n):
    """n outputs of splitmix64 started at `seed` (vectorised: state_i = seed + (i+1)*GOLD)."""
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + _GOLD * np.arange(1, n + 1, dtype=np.uint64)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def _uniform(seed, n):
    # (0,1): 53 random bits, never exactly 0
    return ((splitmix64(seed, n) >> np.uint64(11)).astype(np.float64) + 0.5) * (1.0 / (1 << 53))


def hll_registers(seed, card, p):
    """One register array for a set of `card` distinct elements (Poisson model):
    R = clip(ceil(log2((n/m)/(-ln u))), 0, q+1)."""
    m = 1 << p
    q = 64 - p
    if card <= 0:
        return np.zeros(m, np.uint8)
    u = _uniform(seed, m)
    v = np.ceil(np.log2((card / m) / (-np.log(u))))
    return np.clip(v, 0, q + 1).astype(np.uint8)


def synthetic_sketches(n, p, seed=0x5EED0000, cluster=10, card_lo=2_000_000, card_hi=8_000_000):
    """Register arrays only (see related_sketches for the construction)."""
    return related_sketches(n, p, seed, cluster, card_lo, card_hi)[0]


def related_sketches(n, p, seed=0x5EED0000, cluster=10, card_lo=2_000_000, card_hi=8_000_000):
    """n sketches in clusters of `cluster`.  Every member of a cluster is max(core_c, private_g),
    i.e. exactly the sketch of n_core shared + n_priv private elements, so the true Jaccard of a
    within-cluster pair is n_core/(n_core+n_priv_a+n_priv_b) and 0 across clusters (the
    clamp-to-0 / Mash==1 branch).  Private fractions cycle so J spans ~0.95 .. 0.01.
    Returns (regs, core_card[n], priv_card[n], cluster_id[n])."""
    m = 1 << p
    fr = (0.05, 0.3, 0.6, 0.9, 0.99)
    regs = np.zeros((n, m), np.uint8)
    core_card = np.zeros(n, np.float64)
    priv_card = np.zeros(n, np.float64)
    cid = np.zeros(n, np.int64)
    ncl = (n + cluster - 1) // cluster
    tot = card_lo + (splitmix64(seed ^ 0xC0FFEE, ncl) % np.uint64(card_hi - card_lo + 1)).astype(np.int64)
    for c in range(ncl):
        lo, hi = c * cluster, min(n, (c + 1) * cluster)
        n_core = int(tot[c]) // 2
        core = hll_registers(seed + 7919 * (c + 1), n_core, p)
        for g in range(lo, hi):
            f = fr[(g - lo) % len(fr)]
            n_priv = int(n_core * f / (1.0 - f) / 2) if f < 0.99 else int(n_core * 40)
            priv = hll_registers(seed + 0x10000000 + g, n_priv, p)
            regs[g] = np.maximum(core, priv)
            core_card[g], priv_card[g], cid[g] = n_core, n_priv, c
    return regs, core_card, priv_card, cid


def survey_sketches(n, p, seed=0x5EED0000, cluster=10):
    """Benchmark register arrays per SURVEY.md section 8d: every sketch has a cardinality in
    [2e6, 8e6] (bacterial scale); sketch(g) = max(core_c(g), private_g).  A cluster has a size
    S_c in [2.4e6, 6.6e6]; its shared core is phi_c * S_c with phi cycling through
    {0.95, 0.67, 0.18, 0.02, 0} and each member adds (1 - phi_c) * S_c * U(0.8, 1.2) private
    elements, so the true within-cluster J is ~ {0.9, 0.5, 0.1, 0.01, 0} and 0 across clusters.
    Returns (regs, core_card[n], priv_card[n], cluster_id[n])."""
    m = 1 << p
    phi = (0.95, 0.67, 0.18, 0.02, 0.0)
    regs = np.zeros((n, m), np.uint8)
    core_card = np.zeros(n, np.float64)
    priv_card = np.zeros(n, np.float64)
    cid = np.zeros(n, np.int64)
    ncl = (n + cluster - 1) // cluster
    size = 2_400_000 + (splitmix64(seed ^ 0x8D, ncl) % np.uint64(4_200_001)).astype(np.int64)
    jit = 0.8 + 0.4 * _uniform(seed ^ 0x77, n)
    for c in range(ncl):
        lo, hi = c * cluster, min(n, (c + 1) * cluster)
        f = phi[c % len(phi)]
        n_core = int(f * size[c])
        core = hll_registers(seed + 7919 * (c + 1), n_core, p)
        for g in range(lo, hi):
            n_priv = int((1.0 - f) * size[c] * jit[g])
            priv = hll_registers(seed + 0x10000000 + g, n_priv, p)
            regs[g] = np.maximum(core, priv)
            core_card[g], priv_card[g], cid[g] = n_core, n_priv, c
    return regs, core_card, priv_card, cid


_ACGT = np.frombuffer(b"ACGT", np.uint8)


def _mutate(codes, rate, seed):
    """iid substitutions at `rate` (a substituted base always changes)."""
    n = codes.size
    u = _uniform(seed, n)
    hit = u < rate
    shift = (splitmix64(seed ^ 0xABCDEF, n) % np.uint64(3)).astype(np.uint8) + np.uint8(1)
    out = codes.copy()
    out[hit] = (codes[hit] + shift[hit]) & 3
    return out


def synthetic_genomes(n, length, seed=0xDA5410, cluster=10, decorate=True):
    """n genomes (uint8 ASCII arrays) per SURVEY.md 8d: root -> cluster ancestors (5 %) ->
    members with divergence cycling through {0.1,0.5,1,2,5} %.  Every 10th genome gets a run
    of 50 'N' and a lowercase 1 kb stretch (k-mer reset / case folding) when decorate."""
    root = (splitmix64(seed, length) & np.uint64(3)).astype(np.uint8)
    rates = (0.001, 0.005, 0.01, 0.02, 0.05)
    genomes = []
    anc = None
    for g in range(n):
        c = g // cluster
        if g % cluster == 0:
            anc = _mutate(root, 0.05, seed + 1_000_003 * (c + 1))
        codes = _mutate(anc, rates[g % len(rates)], seed + 17 * (g + 1))
        s = _ACGT[codes].copy()
        if decorate and g % 10 == 0 and length > 4000:
            a = length // 3
            s[a : a + 50] = ord("N")
            b = (2 * length) // 3
            s[b : b + 1000] |= 0x20  # lowercase
        genomes.append(s)
    return genomes


def to_fasta(seq, name="g", width=80):
    """ASCII FASTA bytes of one record with `width`-column lines."""
    lines = [b">" + name.encode()]
    bs = seq.tobytes()
    for i in range(0, len(bs), width):
        lines.append(bs[i : i + width])
    return b"\n".join(lines) + b"\n"


def concat_for_device(genomes):
    """Layout dsh_sketch_batch takes: genomes back to back; returns (seq uint8, genome_off uint64)."""
    off = np.zeros(len(genomes) + 1, np.uint64)
    if genomes:
        off[1:] = np.cumsum([g.size for g in genomes])
    seq = np.concatenate(genomes) if genomes else np.zeros(0, np.uint8)
    return seq, off

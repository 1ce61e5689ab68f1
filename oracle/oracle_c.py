"""ctypes loader for the C oracle (oracle/dsh_oracle.c).

TEST INFRASTRUCTURE ONLY -- imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  Nothing under dashing_amd/ imports this module.  PARITY UNPINNED (see
dsh_oracle.c header): the reference's arithmetic lives in absent submodules.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORIGINAL, ERTL_IMPROVED, ERTL_MLE = 0, 1, 2
MASH_DIST, JI, FULL_MASH_DIST = 0, 1, 3
SIZES, FULL_CONTAINMENT_DIST, CONTAINMENT_INDEX, CONTAINMENT_DIST = 2, 4, 5, 6
SYMMETRIC_CONTAINMENT_INDEX, SYMMETRIC_CONTAINMENT_DIST = 7, 8

_u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")
_u32p = np.ctypeslib.ndpointer(np.uint32, flags="C_CONTIGUOUS")
_u64p = np.ctypeslib.ndpointer(np.uint64, flags="C_CONTIGUOUS")
_f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")


def build(native_out=None):
    """Compile the oracle.  native_out: also build a -march=native copy at that path."""
    subprocess.check_call(["make", "-s", "-C", _HERE])
    if native_out:
        subprocess.check_call(["make", "-s", "-C", _HERE, "native", "OUT=" + native_out])
    return os.path.join(_HERE, "liboracle.so")


def effective_cpus():
    """CPUs we can really run on: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, q // per))
        except (OSError, ValueError, IndexError):
            pass
    return max(1, n)


def _bind(lib):
    lib.dsho_set_threads.restype = None
    lib.dsho_set_threads.argtypes = [C.c_int]
    lib.dsho_wang.restype = C.c_uint64
    lib.dsho_simd_level.restype = C.c_int
    lib.dsho_set_simd.restype = None
    lib.dsho_set_simd.argtypes = [C.c_int]
    lib.dsho_hist_union_simd.restype = None
    lib.dsho_hist_union_simd.argtypes = [_u8p, _u8p, C.c_uint64, C.c_int, C.c_int, C.c_int, _u32p]
    lib.dsho_wang.argtypes = [C.c_uint64]
    lib.dsho_reg_rule.restype = None
    lib.dsho_reg_rule.argtypes = [C.c_uint64, C.c_int, C.POINTER(C.c_uint32), C.POINTER(C.c_uint8)]
    lib.dsho_walk.restype = C.c_uint64
    lib.dsho_walk.argtypes = [_u8p, C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_uint64]
    lib.dsho_sketch_batch.restype = None
    lib.dsho_sketch_batch.argtypes = [_u8p, _u64p, C.c_uint32, C.c_int, C.c_int, C.c_int, _u8p]
    lib.dsho_hist_single.restype = None
    lib.dsho_hist_single.argtypes = [_u8p, C.c_uint64, _u32p]
    lib.dsho_hist_union.restype = None
    lib.dsho_hist_union.argtypes = [_u8p, _u8p, C.c_uint64, _u32p]
    lib.dsho_estimate.restype = C.c_double
    lib.dsho_estimate.argtypes = [_u32p, C.c_int, C.c_int]
    lib.dsho_cardinalities.restype = None
    lib.dsho_cardinalities.argtypes = [_u8p, C.c_uint64, C.c_int, C.c_int, _f64p]
    lib.dsho_union_size.restype = C.c_double
    lib.dsho_union_size.argtypes = [_u8p, _u8p, C.c_int, C.c_int]
    lib.dsho_jaccard_from.restype = C.c_double
    lib.dsho_jaccard_from.argtypes = [C.c_double, C.c_double, C.c_double]
    lib.dsho_result.restype = C.c_float
    lib.dsho_result.argtypes = [C.c_double, C.c_int, C.c_int]
    lib.dsho_result_triple.restype = C.c_float
    lib.dsho_result_triple.argtypes = [C.c_double, C.c_double, C.c_double, C.c_int, C.c_int]
    lib.dsho_dist_tri.restype = None
    lib.dsho_dist_tri.argtypes = [_u8p, C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_int, _f32p]
    lib.dsho_dist_rows.restype = C.c_uint64
    lib.dsho_dist_rows.argtypes = [_u8p, C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_uint64, _f32p]
    lib.dsho_dist_rect.restype = None
    lib.dsho_dist_rect.argtypes = [_u8p, C.c_uint64, _u8p, C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_int, _f32p]
    lib.dsho_knn.restype = None
    lib.dsho_knn.argtypes = [_u8p, C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_uint64,
                             C.c_uint64, C.c_uint64, C.c_uint32, np.ctypeslib.ndpointer(np.uint32, flags="C_CONTIGUOUS"), _f32p]
    lib.dsho_num_threads.restype = C.c_int
    return lib


_LIB = None


def load(path=None, threads=None):
    """Load the oracle.  By default it runs on at most 8 threads with passive waiting (tests);
    bench.py's cpu_baseline leg passes threads=effective_cpus()."""
    global _LIB
    os.environ.setdefault("OMP_WAIT_POLICY", "passive")
    if path is not None:
        lib = _bind(C.CDLL(path))
        lib.dsho_set_threads(threads or min(8, effective_cpus()))
        return lib
    if _LIB is None:
        so = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(so):
            build()
        _LIB = _bind(C.CDLL(so))
        _LIB.dsho_set_threads(threads or min(8, effective_cpus()))
    elif threads:
        _LIB.dsho_set_threads(threads)
    return _LIB


# ---------------------------------------------------------------- convenience wrappers
def wang(x):
    return int(load().dsho_wang(C.c_uint64(x & 0xFFFFFFFFFFFFFFFF)))


def reg_rule(h, p):
    i, v = C.c_uint32(), C.c_uint8()
    load().dsho_reg_rule(C.c_uint64(h), p, C.byref(i), C.byref(v))
    return i.value, v.value


def kmers(seq, k, canon=True):
    s = np.frombuffer(seq if isinstance(seq, (bytes, bytearray)) else seq.encode(), dtype=np.uint8).copy()
    if s.size == 0:
        return []
    out = np.zeros(max(1, s.size), dtype=np.uint64)
    n = load().dsho_walk(s, s.size, k, int(canon), 10, None, out.ctypes.data_as(C.c_void_p), out.size)
    return [int(x) for x in out[:n]]


def sketch_batch(seq, genome_off, k, p, canon=True):
    seq = np.ascontiguousarray(seq, dtype=np.uint8)
    if seq.size == 0:
        seq = np.zeros(1, np.uint8)
    off = np.ascontiguousarray(genome_off, dtype=np.uint64)
    n = off.size - 1
    regs = np.zeros((n, 1 << p), dtype=np.uint8)
    load().dsho_sketch_batch(seq, off, n, k, p, int(canon), regs)
    return regs


def hist_single(a):
    h = np.zeros(64, np.uint32)
    a = np.ascontiguousarray(a, np.uint8)
    load().dsho_hist_single(a, a.size, h)
    return h


def hist_union(a, b):
    h = np.zeros(64, np.uint32)
    a = np.ascontiguousarray(a, np.uint8)
    b = np.ascontiguousarray(b, np.uint8)
    load().dsho_hist_union(a, b, a.size, h)
    return h


def estimate(hist, p, estim=ERTL_MLE):
    return float(load().dsho_estimate(np.ascontiguousarray(hist, np.uint32), p, estim))


def cardinalities(regs, estim=ERTL_MLE):
    regs = np.ascontiguousarray(regs, np.uint8)
    n, m = regs.shape
    out = np.zeros(n, np.float64)
    load().dsho_cardinalities(regs, n, int(m).bit_length() - 1, estim, out)
    return out


def jaccard_from(ca, cb, us):
    return float(load().dsho_jaccard_from(ca, cb, us))


def result(ji, result_type, k):
    return float(load().dsho_result(ji, result_type, k))


def result_triple(mys, os_, us, result_type, k):
    return float(load().dsho_result_triple(mys, os_, us, result_type, k))


def dist_tri(regs, estim=ERTL_MLE, result_type=JI, k=31, lib=None):
    regs = np.ascontiguousarray(regs, np.uint8)
    n, m = regs.shape
    out = np.zeros(max(1, n * (n - 1) // 2), np.float32)
    (lib or load()).dsho_dist_tri(regs, n, int(m).bit_length() - 1, estim, result_type, k, out)
    return out[: n * (n - 1) // 2]


def dist_rows(regs, row_begin, row_end, estim=ERTL_MLE, result_type=JI, k=31, lib=None):
    regs = np.ascontiguousarray(regs, np.uint8)
    n, m = regs.shape
    row_end = min(row_end, n)
    cnt = sum(n - i - 1 for i in range(row_begin, row_end))
    out = np.zeros(max(1, cnt), np.float32)
    done = (lib or load()).dsho_dist_rows(regs, n, int(m).bit_length() - 1, estim, result_type, k, row_begin, row_end, out)
    return out[:done]


def dist_rect(qregs, rregs, estim=ERTL_MLE, result_type=JI, k=31):
    q = np.ascontiguousarray(qregs, np.uint8)
    r = np.ascontiguousarray(rregs, np.uint8)
    out = np.zeros((q.shape[0], r.shape[0]), np.float32)
    load().dsho_dist_rect(q, q.shape[0], r, r.shape[0], int(q.shape[1]).bit_length() - 1, estim, result_type, k, out)
    return out


def knn(regs, nn, qb=0, qe=None, rb=0, re=None, estim=ERTL_MLE, result_type=JI, k=31):
    regs = np.ascontiguousarray(regs, np.uint8)
    n, m = regs.shape
    qe = n if qe is None else qe
    re = n if re is None else re
    idx = np.zeros((max(qe - qb, 0), nn), np.uint32)
    val = np.zeros((max(qe - qb, 0), nn), np.float32)
    if idx.size:
        load().dsho_knn(regs, n, int(m).bit_length() - 1, estim, result_type, k, qb, qe, rb, re, nn, idx, val)
    return idx, val


def simd_level(lib=None):
    """0 scalar, 1 AVX2, 2 AVX-512BW: the widest histogram-of-max variant this host can run"""
    return int((lib or load()).dsho_simd_level())


def set_simd(level, lib=None):
    """0: scalar histogram (default, what the tests check the GPU against); >0: dist_rows uses the SIMD histogram"""
    (lib or load()).dsho_set_simd(int(level))


def hist_union_simd(a, b, level, vlo=None, vhi=None, lib=None):
    a = np.ascontiguousarray(a, np.uint8)
    b = np.ascontiguousarray(b, np.uint8)
    vlo = int(max(a.min(), b.min())) if vlo is None else vlo
    vhi = int(max(a.max(), b.max())) if vhi is None else vhi
    h = np.zeros(64, np.uint32)
    (lib or load()).dsho_hist_union_simd(a, b, a.size, vlo, vhi, level, h)
    return h


def num_threads():
    return int(load().dsho_num_threads())

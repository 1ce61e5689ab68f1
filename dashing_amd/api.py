"""ctypes binding of libdashing_hip.so (include/dashing_hip.h).

There is deliberately no CPU fallback here: if the shared library is missing, or no gfx950
device is visible, construction fails loudly.  (The CPU oracle lives in oracle/ and is test
infrastructure only -- nothing in this package imports it.)
"""
import ctypes as C
import os

import numpy as np

ESTIM_ORIGINAL, ESTIM_ERTL_IMPROVED, ESTIM_ERTL_MLE = 0, 1, 2
MASH_DIST, JI, FULL_MASH_DIST = 0, 1, 3  # bns::EmissionType, src/enums.h:13-23
SIZES, FULL_CONTAINMENT_DIST, CONTAINMENT_INDEX, CONTAINMENT_DIST = 2, 4, 5, 6
SYMMETRIC_CONTAINMENT_INDEX, SYMMETRIC_CONTAINMENT_DIST = 7, 8

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

# every symbol include/dashing_hip.h declares (tests check the .so exports all of them)
SYMBOLS = [
    "dsh_backend_name", "dsh_device_count", "dsh_create", "dsh_destroy", "dsh_last_error",
    "dsh_synchronize", "dsh_sketches_alloc", "dsh_upload_sketches", "dsh_download_sketches", "dsh_copy_sketches_device",
    "dsh_attach_device_sketches", "dsh_sketch_batch", "dsh_sketch_batch_async", "dsh_sketch_batch_device", "dsh_sketch_fastx_batch_async",
    "dsh_clear_sketches", "dsh_cardinalities", "dsh_dist_rows", "dsh_dist_rows_device",
    "dsh_dist_rows_async", "dsh_dist_rows_device_async", "dsh_wait", "dsh_wait_event",
    "dsh_event_record", "dsh_event_wait", "dsh_event_query",
    "dsh_comm_available", "dsh_comm_library", "dsh_comm_wait", "dsh_exchange_mode", "dsh_exchange_rows_device_async",
    "dsh_exchange_collect_async", "dsh_exchange_place_device", "dsh_exchange_probe_parts_async", "dsh_diag_spin_start", "dsh_diag_spin_stop", "dsh_abi_version", "dsh_preload", "dsh_comm_unique_id", "dsh_comm_init", "dsh_comm_destroy", "dsh_comm_rank", "dsh_collect_spans", "dsh_collect_spans_async",
    "dsh_allgather_device", "dsh_dist_collect", "dsh_range_parts", "dsh_dist_rows_parts_device_async", "dsh_collect_parts_async",
    "dsh_dist_rect", "dsh_knn", "dsh_shard_plan", "dsh_dist_shard_device", "dsh_unpermute_device", "dsh_unpermute_staged_device", "dsh_unpermute_blocks_device", "dsh_tri_span", "dsh_tri_index", "dsh_partition_rows", "dsh_balance_rows", "dsh_balance_rowsets", "dsh_rowsets_from_bounds", "dsh_rowsets_rank", "dsh_alloc_host", "dsh_free_host",
    "dsh_set_profiling", "dsh_last_kernel_ms", "dsh_last_part_info", "dsh_finalize_phase_cycles", "dsh_set_option", "dsh_get_info", "dsh_stream",
]


class DshError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("dashing_hip error %d: %s" % (code, msg))
        self.code = code


def lib_path():
    return os.path.join(_HERE, "libdashing_hip.so")


def _preload_torch_hip_runtime():
    """A process must hold ONE HIP runtime.  The PyTorch wheel brings its own libamdhip64.so (SONAME libamdhip64.so.7, the
    name libdashing_hip.so asks for); if this library were loaded first it would pull in /opt/rocm's copy and a later
    `import torch` would add the wheel's next to it -- torch then finds no GPU.  So when a torch installation exists and
    is not loaded yet, its runtime is loaded first (tests and bench.py use torch for device buffers; a C++ host never
    comes here)."""
    import importlib.util
    import sys

    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def load_library():
    """Load libdashing_hip.so (built in-tree by __graft_entry__.build() / csrc/Makefile)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = lib_path()
    if not os.path.exists(path):
        raise ImportError(
            "%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C dashing_amd/csrc` (there is no CPU fallback)" % path)
    _preload_torch_hip_runtime()
    lib = C.CDLL(path)
    if int(lib.dsh_abi_version()) != ABI_VERSION:  # (a stale in-tree .so: shifted arguments, not link errors)
        raise ImportError("%s has ABI version %d, this binding was written against %d: rebuild (make -C dashing_amd/csrc)" % (path, lib.dsh_abi_version(), ABI_VERSION))
    u64, i32, vp = C.c_uint64, C.c_int, C.c_void_p
    lib.dsh_backend_name.restype = C.c_char_p
    lib.dsh_device_count.restype = i32
    lib.dsh_create.argtypes = [i32, C.POINTER(vp)]
    lib.dsh_destroy.argtypes = [vp]
    lib.dsh_destroy.restype = None
    lib.dsh_last_error.argtypes = [vp]
    lib.dsh_last_error.restype = C.c_char_p
    lib.dsh_synchronize.argtypes = [vp]
    lib.dsh_sketches_alloc.argtypes = [vp, u64, i32]
    lib.dsh_upload_sketches.argtypes = [vp, vp, u64, u64]
    lib.dsh_download_sketches.argtypes = [vp, u64, u64, vp]
    lib.dsh_copy_sketches_device.argtypes = [vp, u64, u64, vp]
    lib.dsh_attach_device_sketches.argtypes = [vp, vp, u64, i32]
    lib.dsh_sketch_batch.argtypes = [vp, vp, vp, C.c_uint32, u64, i32, i32, vp]
    lib.dsh_sketch_batch_async.argtypes = [vp, vp, vp, C.c_uint32, u64, i32, i32]
    lib.dsh_sketch_fastx_batch_async.argtypes = [vp, vp, vp, vp, C.c_uint32, u64, i32, i32, vp]
    lib.dsh_sketch_batch_device.argtypes = [vp, vp, vp, C.c_uint32, u64, i32, i32]
    lib.dsh_clear_sketches.argtypes = [vp, u64, u64]
    lib.dsh_cardinalities.argtypes = [vp, i32, vp]
    lib.dsh_dist_rows.argtypes = [vp, i32, i32, i32, u64, u64, vp]
    lib.dsh_dist_rows_device.argtypes = [vp, i32, i32, i32, u64, u64, vp]
    lib.dsh_dist_rows_async.argtypes = [vp, i32, i32, i32, u64, u64, vp]
    lib.dsh_dist_rows_device_async.argtypes = [vp, i32, i32, i32, u64, u64, vp]
    lib.dsh_wait.argtypes = [vp]
    lib.dsh_wait_event.argtypes = [vp, vp]
    lib.dsh_event_record.argtypes = [vp, C.POINTER(C.c_uint64)]
    lib.dsh_event_wait.argtypes = [vp, u64]
    lib.dsh_event_query.argtypes = [vp, u64, C.POINTER(i32)]
    lib.dsh_comm_unique_id.argtypes = [vp]
    lib.dsh_comm_available.argtypes = []
    lib.dsh_comm_library.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(i32)]
    lib.dsh_comm_wait.argtypes = [vp]
    lib.dsh_exchange_mode.argtypes = [u64, vp, i32, C.c_uint32, i32, C.POINTER(i32), C.POINTER(C.c_uint32), C.POINTER(u64)]
    lib.dsh_exchange_rows_device_async.argtypes = [vp, i32, i32, i32, vp, i32, C.c_uint32, i32, vp]
    lib.dsh_exchange_collect_async.argtypes = [vp, u64, vp, C.c_uint32, vp, vp, i32]
    lib.dsh_exchange_place_device.argtypes = [vp, vp, i32, C.c_uint32, i32, vp, vp]
    lib.dsh_exchange_probe_parts_async.argtypes = [vp, u64, vp, i32, C.c_uint32, i32, vp, vp]
    lib.dsh_diag_spin_start.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]
    lib.dsh_diag_spin_stop.argtypes = [vp]
    lib.dsh_abi_version.argtypes = []
    lib.dsh_abi_version.restype = C.c_int
    lib.dsh_balance_rowsets.argtypes = [u64, C.c_uint32, i32, i32, i32, vp, C.c_uint32, C.POINTER(C.c_uint32)]
    lib.dsh_rowsets_from_bounds.argtypes = [vp, C.c_uint32, vp]
    lib.dsh_rowsets_rank.argtypes = [u64, vp, C.c_uint32, vp, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(u64), C.POINTER(u64)]
    lib.dsh_finalize_phase_cycles.argtypes = [vp, vp]
    lib.dsh_last_part_info.argtypes = [vp, vp, vp, C.c_uint32, C.POINTER(C.c_uint32)]
    lib.dsh_comm_init.argtypes = [vp, vp, i32, i32]
    lib.dsh_comm_destroy.argtypes = [vp]
    lib.dsh_comm_rank.argtypes = [vp, C.POINTER(i32), C.POINTER(i32)]
    lib.dsh_collect_spans.argtypes = [vp, u64, vp, vp, vp, i32]
    lib.dsh_collect_spans_async.argtypes = [vp, u64, vp, vp, vp, i32]
    lib.dsh_allgather_device.argtypes = [vp, vp, u64, vp]
    lib.dsh_dist_collect.argtypes = [vp, i32, i32, i32, vp, i32, vp]
    lib.dsh_range_parts.argtypes = [u64, u64, u64, C.c_uint32, vp, C.POINTER(C.c_uint32)]
    lib.dsh_dist_rows_parts_device_async.argtypes = [vp, i32, i32, i32, u64, u64, vp, C.c_uint32]
    lib.dsh_collect_parts_async.argtypes = [vp, u64, vp, C.c_uint32, vp, vp, i32]
    lib.dsh_dist_rect.argtypes = [vp, i32, i32, i32, u64, u64, u64, u64, vp]
    lib.dsh_knn.argtypes = [vp, i32, i32, i32, u64, u64, u64, u64, C.c_uint32, vp, vp]
    lib.dsh_shard_plan.argtypes = [vp, i32, C.c_uint32, vp]
    lib.dsh_dist_shard_device.argtypes = [vp, i32, i32, i32, C.c_uint32, C.c_uint32, vp]
    lib.dsh_unpermute_device.argtypes = [vp, vp, vp]
    lib.dsh_unpermute_staged_device.argtypes = [vp, vp, u64, C.c_uint32, vp]
    lib.dsh_unpermute_blocks_device.argtypes = [vp, vp, vp, C.c_uint32, vp]
    lib.dsh_tri_span.argtypes = [u64, u64, u64]
    lib.dsh_tri_span.restype = u64
    lib.dsh_tri_index.argtypes = [u64, u64, u64]
    lib.dsh_tri_index.restype = u64
    lib.dsh_partition_rows.argtypes = [u64, C.c_uint32, C.c_uint32, vp]
    lib.dsh_balance_rows.argtypes = [u64, C.c_uint32, vp]
    lib.dsh_alloc_host.argtypes = [C.c_size_t]
    lib.dsh_alloc_host.restype = vp
    lib.dsh_free_host.argtypes = [vp]
    lib.dsh_free_host.restype = None
    lib.dsh_set_profiling.argtypes = [vp, i32]
    lib.dsh_last_kernel_ms.argtypes = [vp, vp, vp, vp, vp]
    lib.dsh_set_option.argtypes = [vp, C.c_char_p, C.c_int64]
    lib.dsh_get_info.argtypes = [vp, C.c_char_p, C.POINTER(C.c_int64)]
    lib.dsh_stream.argtypes = [vp]
    lib.dsh_stream.restype = vp
    _LIB = lib
    return lib


def backend_name():
    return load_library().dsh_backend_name().decode()


ABI_VERSION = 6  # include/dashing_hip.h DSH_ABI_VERSION this binding was written against


def abi_version():
    return int(load_library().dsh_abi_version())


def device_count():
    return int(load_library().dsh_device_count())


def comm_unique_id():
    """128 bytes (an ncclUniqueId) created by RCCL inside the library: rank 0 makes it, every rank passes it to
    Context.comm_init.  Raises if RCCL cannot be loaded."""
    buf = C.create_string_buffer(128)
    rc = load_library().dsh_comm_unique_id(buf)
    if rc:
        raise DshError(rc, "dsh_comm_unique_id (RCCL not available?)")
    return buf.raw


def comm_available():
    """True if librccl can be loaded by the library in this process (local check, no communication)"""
    return load_library().dsh_comm_available() == 0


def comm_library():
    """(resolved path of the loaded librccl -- or the loader's error text --, ncclGetVersion code or 0)"""
    buf = C.create_string_buffer(1024)
    v = C.c_int(0)
    load_library().dsh_comm_library(buf, len(buf), C.byref(v))
    return buf.value.decode(errors="replace"), int(v.value)


class RowSets:
    """A partition of the triangle's rows over the ranks as a row-set table (include/dashing_hip.h): row segments with
    owners.  `table` is the uint64 array the C-ABI takes; rows(r) the segments of rank r, pairs(r) / tiles(r) its work."""

    def __init__(self, n, table):
        self.n = int(n)
        self.table = np.ascontiguousarray(table, np.uint64)
        self.world = int(self.table[0])

    def _rank(self, r):
        segs = np.zeros(2 * max(int(self.table[1]), 1), np.uint64)
        ns, pairs, tiles = C.c_uint32(), C.c_uint64(), C.c_uint64()
        rc = load_library().dsh_rowsets_rank(self.n, self.table.ctypes.data, r, segs.ctypes.data, len(segs) // 2, C.byref(ns),
                                             C.byref(pairs), C.byref(tiles))
        if rc:
            raise DshError(rc, "dsh_rowsets_rank: malformed row-set table")
        return [(int(segs[2 * i]), int(segs[2 * i + 1])) for i in range(ns.value)], int(pairs.value), int(tiles.value)

    def rows(self, r):
        return self._rank(r)[0]

    def pairs(self, r):
        return self._rank(r)[1]

    def tiles(self, r):
        return self._rank(r)[2]

    def describe(self):
        return [{"rank": r, "rows": self.rows(r), "pairs": self.pairs(r), "tiles": self.tiles(r)} for r in range(self.world)]


def balance_rowsets(n, world, prep_permille=-1, dst=-1, dst_bonus_permille=-1):
    """Main ranges + top-up tile rows from the bottom of the triangle, one row set per rank (dsh_balance_rowsets); dst >= 0:
    the rank that receives the others' rows takes a bonus of work (it sends nothing)."""
    lib = load_library()
    words = C.c_uint32()
    rc = lib.dsh_balance_rowsets(n, world, prep_permille, dst, dst_bonus_permille, None, 0, C.byref(words))
    if rc:
        raise DshError(rc, "dsh_balance_rowsets")
    tab = np.zeros(words.value, np.uint64)
    rc = lib.dsh_balance_rowsets(n, world, prep_permille, dst, dst_bonus_permille, tab.ctypes.data, len(tab), C.byref(words))
    if rc:
        raise DshError(rc, "dsh_balance_rowsets")
    return RowSets(n, tab)


def rowsets_from_bounds(n, bounds):
    """contiguous row ranges bounds[world + 1] as a row-set table (dsh_rowsets_from_bounds)"""
    b = np.ascontiguousarray(bounds, np.uint64)
    tab = np.zeros(3 + 2 * (len(b) - 1), np.uint64)
    rc = load_library().dsh_rowsets_from_bounds(b.ctypes.data, len(b) - 1, tab.ctypes.data)
    if rc:
        raise DshError(rc, "dsh_rowsets_from_bounds")
    return RowSets(n, tab)


def _table(n, rows):
    """the C-ABI table of `rows`: a RowSets, or contiguous bounds [world + 1]"""
    return (rows if isinstance(rows, RowSets) else rowsets_from_bounds(n, rows)).table


def exchange_mode(n, rows, rank, nparts, dst=0, want_floats=False):
    """(rowsorted, parts[, floats of the rank's buffer]) of rank `rank` under dsh_exchange_* (dsh_exchange_mode); `rows` is a
    RowSets or contiguous bounds"""
    t = _table(n, rows)
    rs, k, fl = C.c_int(), C.c_uint32(), C.c_uint64()
    rc = load_library().dsh_exchange_mode(n, t.ctypes.data, rank, nparts, dst, C.byref(rs), C.byref(k), C.byref(fl))
    if rc:
        raise DshError(rc, "dsh_exchange_mode")
    return (bool(rs.value), int(k.value), int(fl.value)) if want_floats else (bool(rs.value), int(k.value))


def range_parts(n, rb, re, nparts):
    """row boundaries of the parts dsh_dist_rows_parts_device_async cuts [rb, re) into (may be fewer than nparts)"""
    b = np.zeros(nparts + 1, np.uint64)
    k = C.c_uint32()
    rc = load_library().dsh_range_parts(n, rb, re, nparts, b.ctypes.data, C.byref(k))
    if rc:
        raise DshError(rc, "dsh_range_parts")
    return [int(x) for x in b[: k.value + 1]]


def tri_span(n, rb, re):
    return int(load_library().dsh_tri_span(n, rb, re))


def tri_index(n, i, j):
    return int(load_library().dsh_tri_index(n, i, j))


def partition_rows(n, nparts, align=128):
    """Host-only helper shared with the C++ CLI: contiguous row ranges of near-equal pair count."""
    b = np.zeros(nparts + 1, np.uint64)
    rc = load_library().dsh_partition_rows(n, nparts, align, b.ctypes.data)
    if rc:
        raise DshError(rc, "dsh_partition_rows")
    return [int(x) for x in b]


def balance_rows(n, nparts):
    """Tile-aligned row ranges, one per rank, minimising the largest tile count (dsh_balance_rows)."""
    b = np.zeros(nparts + 1, np.uint64)
    rc = load_library().dsh_balance_rows(n, nparts, b.ctypes.data)
    if rc:
        raise DshError(rc, "dsh_balance_rows")
    return [int(x) for x in b]


class PinnedArray:
    """numpy view of page-locked host memory (dsh_alloc_host); keep the object alive while in use."""

    def __init__(self, n, dtype=np.float32):
        self._lib = load_library()
        nbytes = max(int(n), 1) * np.dtype(dtype).itemsize
        self._p = self._lib.dsh_alloc_host(nbytes)
        if not self._p:
            raise MemoryError("dsh_alloc_host(%d) failed" % nbytes)
        buf = (C.c_char * nbytes).from_address(self._p)
        self.array = np.frombuffer(buf, dtype=dtype, count=max(int(n), 1))

    def __del__(self):
        if getattr(self, "_p", None):
            self._lib.dsh_free_host(self._p)
            self._p = None


class Context:
    """One GPU.  Mirrors the call sequence of dist_sketch_and_cmp (src/sketch_and_cmp.h:268-417):
    allocate N sketches, fill them (sketch_batch / upload), cardinalities, all-pairs rows."""

    def __init__(self, device=0):
        self._lib = load_library()
        h = C.c_void_p()
        rc = self._lib.dsh_create(device, C.byref(h))
        if rc:
            raise DshError(rc, "dsh_create(device=%d) failed (no gfx950 device?)" % device)
        self._h = h
        self.n = 0
        self.p = 0

    def close(self):
        if getattr(self, "_h", None):
            self._lib.dsh_destroy(self._h)
            self._h = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _ck(self, rc):
        if rc:
            raise DshError(rc, self._lib.dsh_last_error(self._h).decode())

    # ---- sketch matrix
    def alloc(self, n, p):
        self._ck(self._lib.dsh_sketches_alloc(self._h, n, p))
        self.n, self.p = n, p

    def upload(self, regs, first_slot=0):
        regs = np.ascontiguousarray(regs, np.uint8)
        assert regs.ndim == 2 and regs.shape[1] == (1 << self.p)
        self._ck(self._lib.dsh_upload_sketches(self._h, regs.ctypes.data, first_slot, regs.shape[0]))

    def set_sketches(self, regs):
        regs = np.ascontiguousarray(regs, np.uint8)
        n, m = regs.shape
        self.alloc(n, int(m).bit_length() - 1)
        self.upload(regs)

    def attach_device(self, data_ptr, n, p):
        self._ck(self._lib.dsh_attach_device_sketches(self._h, C.c_void_p(data_ptr), n, p))
        self.n, self.p = n, p

    def download(self, first_slot=0, n=None):
        n = self.n - first_slot if n is None else n
        out = np.zeros((n, 1 << self.p), np.uint8)
        self._ck(self._lib.dsh_download_sketches(self._h, first_slot, n, out.ctypes.data))
        return out

    def copy_sketches_device(self, out_ptr, first_slot=0, n=None):
        n = self.n - first_slot if n is None else n
        self._ck(self._lib.dsh_copy_sketches_device(self._h, first_slot, n, C.c_void_p(out_ptr)))

    def clear(self, first_slot=0, n=None):
        n = self.n - first_slot if n is None else n
        self._ck(self._lib.dsh_clear_sketches(self._h, first_slot, n))

    # ---- sketch waist
    def sketch_batch(self, seq, genome_off, first_slot=0, k=31, canon=True, want_regs=True):
        seq = np.ascontiguousarray(seq, np.uint8)
        off = np.ascontiguousarray(genome_off, np.uint64)
        ng = off.size - 1
        out = np.zeros((ng, 1 << self.p), np.uint8) if want_regs else None
        self._ck(self._lib.dsh_sketch_batch(
            self._h, seq.ctypes.data if seq.size else None, off.ctypes.data, ng, first_slot, k,
            int(bool(canon)), out.ctypes.data if want_regs else None))
        return out

    def sketch_batch_async(self, seq_pinned, genome_off, first_slot=0, k=31, canon=True):
        """seq_pinned: numpy view of PinnedArray memory; complete after wait()"""
        off = np.ascontiguousarray(genome_off, np.uint64)
        self._ck(self._lib.dsh_sketch_batch_async(
            self._h, seq_pinned.ctypes.data, off.ctypes.data, off.size - 1, first_slot, k, int(bool(canon))))

    def sketch_fastx_batch(self, files, first_slot=0, k=31, canon=True):
        """files: one bytes object per genome, the raw text of a plain FASTA file (dsh_sketch_fastx_batch_async: the
        parse runs on the device).  Returns the per-genome status words (0 = decoded and sketched; != 0 = refused, nothing
        sketched: not plain FASTA) after waiting."""
        ng = len(files)
        off = np.zeros(ng + 1, np.uint64)
        lens = np.array([len(f) for f in files], np.uint64)
        for g in range(ng):
            off[g + 1] = off[g] + ((int(lens[g]) + 1 + 31) & ~31)
        raw = np.full(max(int(off[-1]), 1), 0x4E, np.uint8)
        for g, f in enumerate(files):
            raw[int(off[g]): int(off[g]) + len(f)] = np.frombuffer(f, np.uint8)
        status = np.zeros(max(ng, 1), np.uint32)
        self._ck(self._lib.dsh_sketch_fastx_batch_async(
            self._h, raw.ctypes.data, off.ctypes.data, lens.ctypes.data, ng, first_slot, k, int(bool(canon)), status.ctypes.data))
        self.wait()
        return status[:ng]

    def sketch_batch_device(self, seq_ptr, genome_off, first_slot=0, k=31, canon=True):
        off = np.ascontiguousarray(genome_off, np.uint64)
        self._ck(self._lib.dsh_sketch_batch_device(
            self._h, C.c_void_p(seq_ptr), off.ctypes.data, off.size - 1, first_slot, k, int(bool(canon))))

    # ---- compare waist
    def cardinalities(self, estim=ESTIM_ERTL_MLE):
        out = np.zeros(max(self.n, 1), np.float64)
        self._ck(self._lib.dsh_cardinalities(self._h, estim, out.ctypes.data))
        return out[: self.n]

    def dist_rows(self, row_begin=0, row_end=None, estim=ESTIM_ERTL_MLE, result_type=JI, k=31, out=None):
        row_end = self.n if row_end is None else row_end
        span = tri_span(self.n, row_begin, row_end)
        if out is None:
            out = np.zeros(max(span, 1), np.float32)
        self._ck(self._lib.dsh_dist_rows(self._h, estim, result_type, k, row_begin, row_end, out.ctypes.data))
        return out[:span]

    def dist_rows_device(self, out_ptr, row_begin=0, row_end=None, estim=ESTIM_ERTL_MLE, result_type=JI, k=31):
        row_end = self.n if row_end is None else row_end
        self._ck(self._lib.dsh_dist_rows_device(self._h, estim, result_type, k, row_begin, row_end, C.c_void_p(out_ptr)))

    def dist_rows_async(self, out_pinned, row_begin=0, row_end=None, estim=ESTIM_ERTL_MLE, result_type=JI, k=31):
        """enqueue rows -> `out_pinned` (numpy view of PinnedArray memory); valid after wait()"""
        row_end = self.n if row_end is None else row_end
        assert out_pinned.size >= tri_span(self.n, row_begin, row_end)
        self._ck(self._lib.dsh_dist_rows_async(self._h, estim, result_type, k, row_begin, row_end, out_pinned.ctypes.data))

    def dist_rows_device_async(self, out_ptr, row_begin=0, row_end=None, estim=ESTIM_ERTL_MLE, result_type=JI, k=31):
        row_end = self.n if row_end is None else row_end
        self._ck(self._lib.dsh_dist_rows_device_async(self._h, estim, result_type, k, row_begin, row_end, C.c_void_p(out_ptr)))

    def wait(self):
        self._ck(self._lib.dsh_wait(self._h))

    def event_record(self):
        """ticket marking everything enqueued on this context so far"""
        t = C.c_uint64()
        self._ck(self._lib.dsh_event_record(self._h, C.byref(t)))
        return int(t.value)

    def event_wait(self, ticket):
        self._ck(self._lib.dsh_event_wait(self._h, ticket))

    def event_done(self, ticket):
        d = C.c_int()
        self._ck(self._lib.dsh_event_query(self._h, ticket, C.byref(d)))
        return bool(d.value)

    def wait_event(self, hip_event):
        """order the ctx stream after a caller event (e.g. torch.cuda.Event().cuda_event, recorded on torch's stream)"""
        self._ck(self._lib.dsh_wait_event(self._h, C.c_void_p(hip_event)))

    def dist_rect(self, q_begin, q_end, r_begin, r_end, estim=ESTIM_ERTL_MLE, result_type=JI, k=31):
        out = np.zeros((max(q_end - q_begin, 0), max(r_end - r_begin, 0)), np.float32)
        buf = out if out.size else np.zeros(1, np.float32)
        self._ck(self._lib.dsh_dist_rect(self._h, estim, result_type, k, q_begin, q_end, r_begin, r_end, buf.ctypes.data))
        return out

    def knn(self, nn, q_begin=0, q_end=None, r_begin=0, r_end=None, estim=ESTIM_ERTL_MLE, result_type=JI, k=31):
        q_end = self.n if q_end is None else q_end
        r_end = self.n if r_end is None else r_end
        nq = max(q_end - q_begin, 0)
        idx = np.zeros((nq, nn), np.uint32)
        val = np.zeros((nq, nn), np.float32)
        if nq and nn:
            self._ck(self._lib.dsh_knn(self._h, estim, result_type, k, q_begin, q_end, r_begin, r_end, nn,
                                       idx.ctypes.data, val.ctypes.data))
        return idx, val

    # ---- multi-GPU shards (sorted-order spans + one un-permute)
    def shard_plan(self, nshards, estim=ESTIM_ERTL_MLE):
        off = np.zeros(nshards + 1, np.uint64)
        self._ck(self._lib.dsh_shard_plan(self._h, estim, nshards, off.ctypes.data))
        return [int(x) for x in off]

    def dist_shard_device(self, out_ptr, shard, nshards, estim=ESTIM_ERTL_MLE, result_type=JI, k=31):
        self._ck(self._lib.dsh_dist_shard_device(self._h, estim, result_type, k, shard, nshards, C.c_void_p(out_ptr)))

    def unpermute_device(self, sorted_ptr, out_ptr):
        self._ck(self._lib.dsh_unpermute_device(self._h, C.c_void_p(sorted_ptr), C.c_void_p(out_ptr)))

    def unpermute_staged_device(self, stage_ptr, stride, nshards, out_ptr):
        self._ck(self._lib.dsh_unpermute_staged_device(self._h, C.c_void_p(stage_ptr), stride, nshards, C.c_void_p(out_ptr)))

    def unpermute_blocks_device(self, stage_ptr, block_off, out_ptr):
        off = np.ascontiguousarray(block_off, np.uint64)
        self._ck(self._lib.dsh_unpermute_blocks_device(self._h, C.c_void_p(stage_ptr), off.ctypes.data, off.size, C.c_void_p(out_ptr)))

    # ---- multi-GPU exchange over RCCL inside the library (all traffic on the ctx stream)
    def comm_init(self, unique_id, rank, world):
        self._ck(self._lib.dsh_comm_init(self._h, unique_id, rank, world))

    def comm_destroy(self):
        self._ck(self._lib.dsh_comm_destroy(self._h))

    def last_part_info(self):
        """[(ready_ms, bytes)] of the parts of the last call with parts under profiling (dsh_last_part_info)"""
        k = C.c_uint32()
        self._ck(self._lib.dsh_last_part_info(self._h, None, None, 0, C.byref(k)))
        ms, fl = np.zeros(max(k.value, 1), np.float64), np.zeros(max(k.value, 1), np.uint64)
        self._ck(self._lib.dsh_last_part_info(self._h, ms.ctypes.data, fl.ctypes.data, k.value, C.byref(k)))
        return [(float(ms[q]), 4 * int(fl[q])) for q in range(k.value)]

    def finalize_phase_cycles(self):
        """per-phase cycle sums of the last call with the option finalize_timing (dsh_finalize_phase_cycles)"""
        out = np.zeros(16, np.uint64)
        self._ck(self._lib.dsh_finalize_phase_cycles(self._h, out.ctypes.data))
        return [int(x) for x in out]

    # (`rows` below: a RowSets -- balance_rowsets() -- or contiguous bounds [world + 1])
    def exchange_rows_device_async(self, out_ptr, rows, rank, nparts, dst=0, estim=ESTIM_ERTL_MLE, result_type=JI, k=31):
        t = _table(self.n, rows)
        self._ck(self._lib.dsh_exchange_rows_device_async(self._h, estim, result_type, k, t.ctypes.data, rank, nparts, dst, C.c_void_p(out_ptr)))

    def exchange_collect_async(self, n, rows, nparts, local_ptr, final_ptr, dst=0):
        t = _table(n, rows)
        self._ck(self._lib.dsh_exchange_collect_async(self._h, n, t.ctypes.data, nparts, C.c_void_p(local_ptr), C.c_void_p(final_ptr), dst))

    def exchange_place_device(self, rows, src, nparts, src_local_ptr, final_ptr, dst=0):
        t = _table(self.n, rows)
        self._ck(self._lib.dsh_exchange_place_device(self._h, t.ctypes.data, src, nparts, dst, C.c_void_p(src_local_ptr), C.c_void_p(final_ptr)))

    def exchange_probe_parts_async(self, n, rows, rank, nparts, local_ptr, probe_ptr, dst=0):
        """behind every part's gate a copy KERNEL of the part's share of the rank's buffer into `probe` (diagnostic)"""
        t = _table(n, rows)
        self._ck(self._lib.dsh_exchange_probe_parts_async(self._h, n, t.ctypes.data, rank, nparts, dst, C.c_void_p(local_ptr), C.c_void_p(probe_ptr)))

    def diag_spin_start(self, nblocks, threads=256, lds_bytes=16384, max_ms=2000):
        self._ck(self._lib.dsh_diag_spin_start(self._h, nblocks, threads, lds_bytes, max_ms))

    def diag_spin_stop(self):
        self._ck(self._lib.dsh_diag_spin_stop(self._h))

    def comm_wait(self):
        """dsh_wait with a deadline on the RCCL traffic (DSH_COMM_TIMEOUT_S): an error instead of a hang"""
        self._ck(self._lib.dsh_comm_wait(self._h))

    def comm_rank(self):
        r, w = C.c_int(), C.c_int()
        rc = self._lib.dsh_comm_rank(self._h, C.byref(r), C.byref(w))
        return (r.value, w.value) if rc == 0 else None

    def collect_spans(self, n, bounds, local_ptr, final_ptr, dst=0, wait=True):
        b = np.ascontiguousarray(bounds, np.uint64)
        f = self._lib.dsh_collect_spans if wait else self._lib.dsh_collect_spans_async
        self._ck(f(self._h, n, b.ctypes.data, C.c_void_p(local_ptr), C.c_void_p(final_ptr), dst))

    def dist_rows_parts_device_async(self, out_ptr, row_begin, row_end, nparts, estim=ESTIM_ERTL_MLE, result_type=JI, k=31):
        self._ck(self._lib.dsh_dist_rows_parts_device_async(self._h, estim, result_type, k, row_begin, row_end, C.c_void_p(out_ptr), nparts))

    def collect_parts_async(self, n, bounds, nparts, local_ptr, final_ptr, dst=0):
        b = np.ascontiguousarray(bounds, np.uint64)
        self._ck(self._lib.dsh_collect_parts_async(self._h, n, b.ctypes.data, nparts, C.c_void_p(local_ptr), C.c_void_p(final_ptr), dst))

    def allgather_device(self, send_ptr, bytes_per_rank, recv_ptr):
        self._ck(self._lib.dsh_allgather_device(self._h, C.c_void_p(send_ptr), bytes_per_rank, C.c_void_p(recv_ptr)))

    def dist_collect(self, bounds, dst=0, estim=ESTIM_ERTL_MLE, result_type=JI, k=31):
        """this rank's rows computed, every span delivered to `dst`; returns the packed matrix there, None elsewhere.
        bounds = None: the library partitions the rows itself (balanced row sets, the pipelined exchange pair)"""
        b = None if bounds is None else np.ascontiguousarray(bounds, np.uint64)
        me = self.comm_rank()
        out = None
        if me is None or me[0] == dst:
            out = np.zeros(max(tri_span(self.n, 0, self.n), 1), np.float32)
        self._ck(self._lib.dsh_dist_collect(self._h, estim, result_type, k, None if b is None else b.ctypes.data, dst,
                                            out.ctypes.data if out is not None else None))
        return None if out is None else out[: tri_span(self.n, 0, self.n)]

    # ---- misc
    def synchronize(self):
        self._ck(self._lib.dsh_synchronize(self._h))

    def set_profiling(self, on=True):
        self._ck(self._lib.dsh_set_profiling(self._h, int(on)))

    def last_kernel_ms(self):
        a, b, c, n = C.c_double(), C.c_double(), C.c_double(), C.c_uint32()
        self._ck(self._lib.dsh_last_kernel_ms(self._h, C.byref(a), C.byref(b), C.byref(c), C.byref(n)))
        return {"pair_ms": a.value, "finalize_ms": b.value, "prepare_ms": c.value, "pair_launches": n.value}

    def set_option(self, name, value):
        self._ck(self._lib.dsh_set_option(self._h, name.encode(), int(value)))

    def info(self, name):
        v = C.c_int64()
        self._ck(self._lib.dsh_get_info(self._h, name.encode(), C.byref(v)))
        return int(v.value)

    @property
    def stream(self):
        return self._lib.dsh_stream(self._h)

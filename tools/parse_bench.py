#!/usr/bin/env python3
"""Host-side FASTA parse scaling (no GPU kernels): T threads each parse their own 5 Mbp FASTA files through
dshh_append_fastx_into -- the function the CLI's streaming loader calls -- into (a) pageable memory, (b) page-locked
memory from dsh_alloc_host.  Prints one JSON line per (threads, memory kind): aggregate GB/s of sequence bytes.
Says whether the loader is bound by the parser (per-thread rate flat as T grows) or by the memory system."""
import ctypes as C
import json
import os
import sys
import tempfile
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dashing_amd import synth  # noqa: E402


def main():
    nfiles, L = int(os.environ.get("PB_FILES", "64")), int(os.environ.get("PB_LEN", "5000000"))
    host = C.CDLL(os.path.join(ROOT, "dashing_amd", "libdashing_host.so"))
    host.dshh_append_fastx_into.restype = C.c_long
    host.dshh_append_fastx_into.argtypes = [C.c_char_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    d = tempfile.mkdtemp(prefix="pb_")
    rng = np.random.default_rng(3)
    lut = np.frombuffer(b"ACGT", np.uint8)
    paths = []
    for i in range(nfiles):
        pth = os.path.join(d, "g%03d.fna" % i)
        open(pth, "wb").write(synth.to_fasta(lut[rng.integers(0, 4, L)], "g%d" % i))
        paths.append(pth.encode())
    cap = L + 4096
    kinds = [("pageable", None)]
    try:
        import dashing_amd
        if dashing_amd.device_count() > 0:
            kinds.append(("page-locked", dashing_amd))
    except Exception:
        pass
    for kind, mod in kinds:
        if mod is None:
            bufs = [np.empty(cap, np.uint8) for _ in range(nfiles)]
            ptrs = [b.ctypes.data for b in bufs]
        else:
            pins = [mod.PinnedArray(cap, np.uint8) for _ in range(nfiles)]
            ptrs = [p.array.ctypes.data for p in pins]
        for b in ptrs:  # touch every page once
            C.memset(b, 0, cap)
        for T in (1, 2, 4, 8, 16, 32):
            if T > (os.cpu_count() or 1) * 2:
                break
            best = 1e9
            for _ in range(3):
                def work(t):
                    for i in range(t, nfiles, T):
                        n = C.c_size_t(0)
                        r = host.dshh_append_fastx_into(paths[i], ptrs[i], cap, C.byref(n))
                        assert r == 1 and n.value == L
                th = [threading.Thread(target=work, args=(t,)) for t in range(T)]
                t0 = time.perf_counter()
                [x.start() for x in th]
                [x.join() for x in th]
                best = min(best, time.perf_counter() - t0)
            print(json.dumps({"threads": T, "memory": kind, "files": nfiles, "GB_per_s": nfiles * L / best / 1e9,
                              "per_thread_GB_per_s": nfiles * L / best / 1e9 / T, "cpus": os.cpu_count()}))


if __name__ == "__main__":
    main()

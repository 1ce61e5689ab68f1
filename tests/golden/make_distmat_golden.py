#!/usr/bin/env python3
"""Generates tests/golden/distmat/* with the REFERENCE's own distmat/distmat.h (oracle/_ref/distmat_ref,
built by `make -C oracle ref` from /root/reference -- only possible in the build container).
The fixtures are outputs of reference code on index-encoded values  v(big, small) = (small*4096 + big)/1024:
  n<N>_b<B>.bin       dm::DistanceMatrix<float>::write(FILE*)   (magic, u64 n, packed upper triangle)
  n<N>_b<B>.txt/.sci.txt   DistanceMatrix::printf(fp, false/true)  (what `dashing printmat [-s]` prints)
  n<N>_b<B>.idx.txt   row_ptr offsets / row_span lengths / index(i,j) samples
filled through dm::parallel_fill with nperbatch = B.  For n = 280 the text outputs are kept as sha256 only
(manifest.json) and the .bin gzip'ed.  The in-place mmap variant dashing's `dist -b` uses
(src/sketch_and_cmp.h:838-849) must equal the .bin byte for byte -- checked here at generation time."""
import gzip
import hashlib
import json
import os
import shutil
import subprocess
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
DRV = os.path.join(ROOT, "oracle", "_ref", "distmat_ref")
CASES = [(2, 1), (3, 1), (5, 2), (37, 1), (37, 4), (280, 16), (280, 140)]


def main():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "ref"])
    out = os.path.join(HERE, "distmat")
    os.makedirs(out, exist_ok=True)
    manifest = {}
    with tempfile.TemporaryDirectory() as d:
        for n, b in CASES:
            tag = "n%d_b%d" % (n, b)
            pre = os.path.join(d, tag)
            subprocess.check_call([DRV, str(n), str(b), pre], stderr=subprocess.DEVNULL)
            assert open(pre + ".bin", "rb").read() == open(pre + ".mmap.bin", "rb").read()
            entry = {"n": n, "nperbatch": b}
            for ext in (".bin", ".txt", ".sci.txt", ".idx.txt"):
                data = open(pre + ext, "rb").read()
                entry[ext] = {"sha256": hashlib.sha256(data).hexdigest(), "bytes": len(data)}
                if n <= 37:
                    shutil.copy(pre + ext, os.path.join(out, tag + ext))
                elif ext in (".bin", ".idx.txt") and b == 16:
                    with open(os.path.join(out, tag + ext + ".gz"), "wb") as f:
                        f.write(gzip.compress(data, 9, mtime=0))
            manifest[tag] = entry
    json.dump(manifest, open(os.path.join(out, "manifest.json"), "w"), indent=1, sort_keys=True)
    print("wrote", out)


if __name__ == "__main__":
    main()

// distmat_driver.cpp -- TEST INFRASTRUCTURE.  A small driver (ours) linked against the REFERENCE's own
// distmat/distmat.h, compiled from where it lies under /root/reference by oracle/Makefile (target `ref`)
// into oracle/_ref/distmat_ref.  It exercises the reference's packed-triangle container on index-encoded
// values so that tests/golden/distmat/* pin our -b writer, printmat and dsh_tri_* to the reference's code:
//   dm::DistanceMatrix<float>  file layout   distmat/distmat.h:196-204,390-412
//   index macro / row_ptr                    distmat/distmat.h:260-264,273-279
//   printf ("%lf" / "%le" table)             distmat/distmat.h:358-381
//   dm::parallel_fill (batched rows)         distmat/distmat.h:459-512
//   the `dist -b` call sequence              src/sketch_and_cmp.h:838-849 ('\0', u64 n, ftruncate, mmap, fill)
// usage: distmat_ref <n> <nperbatch> <out_prefix>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <unistd.h>
#include "distmat.h"

static float enc(size_t big, size_t small) { return (float)(small * 4096 + big) / 1024.f; }  // exact in float32

int main(int argc, char **argv)
{
    if (argc != 4) return 2;
    const size_t n = std::strtoull(argv[1], nullptr, 10), nb = std::strtoull(argv[2], nullptr, 10);
    const std::string pre = argv[3];
    dm::DistanceMatrix<float> m(n);
    dm::parallel_fill(m, n, [](size_t i, size_t j) { return enc(i, j); }, nb);  // called as oracle(k, j), k > j
    std::FILE *fp = std::fopen((pre + ".bin").c_str(), "wb");
    m.write(fp);
    std::fclose(fp);
    fp = std::fopen((pre + ".txt").c_str(), "wb");
    m.printf(fp, false);
    std::fclose(fp);
    fp = std::fopen((pre + ".sci.txt").c_str(), "wb");
    m.printf(fp, true);
    std::fclose(fp);
    fp = std::fopen((pre + ".idx.txt").c_str(), "wb");  // row offsets and a sweep of index(i,j), both orders
    for (size_t i = 0; i < n; ++i) std::fprintf(fp, "row %zu %td %zu\n", i, m.row_ptr(i) - m.data(), m.row_span(i).second);
    for (size_t i = 0; i < n; i += (n > 40 ? 7 : 1))
        for (size_t j = 0; j < n; j += (n > 40 ? 11 : 1))
            if (i != j) std::fprintf(fp, "idx %zu %zu %zu\n", i, j, m.index(i, j));
    std::fclose(fp);
    // dashing's -b sequence: header + ftruncate, then the mmap-backed matrix filled in place
    const std::string mp = pre + ".mmap.bin";
    fp = std::fopen(mp.c_str(), "wb");
    std::fputc('\0', fp);
    uint64_t nelem = n;
    if (std::fwrite(&nelem, sizeof nelem, 1, fp) != 1) return 3;
    if (::ftruncate(::fileno(fp), 1 + sizeof(uint64_t) + ((n * (n - 1)) >> 1) * sizeof(float))) return 3;
    std::fclose(fp);
    {
        dm::DistanceMatrix<float> d(mp.c_str(), n, 0.f);
        dm::parallel_fill(d, n, [](size_t i, size_t j) { return enc(i, j); }, nb);
    }
    return 0;
}

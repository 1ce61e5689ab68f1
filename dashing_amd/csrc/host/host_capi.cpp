// host_capi.cpp -- C wrappers over host.h so the host logic can be unit-tested on a CPU-only
// box through ctypes (tests/test_host.py).  Built into libdashing_host.so; not part of the GPU
// C-ABI (include/dashing_hip.h).
#include <cstring>

#include "host.h"

using namespace dshh;

static std::vector<std::string> unpack(const char *joined)
{
    std::vector<std::string> v;
    const char *p = joined;
    while (*p) {
        const char *e = std::strchr(p, '\n');
        if (!e) {
            v.emplace_back(p);
            break;
        }
        v.emplace_back(p, e);
        p = e + 1;
    }
    return v;
}

static int pack(const std::vector<std::string> &v, char *out, size_t cap)
{
    std::string s;
    for (const auto &x : v) {
        s += x;
        s += '\n';
    }
    if (s.size() + 1 > cap) return -1;
    std::memcpy(out, s.c_str(), s.size() + 1);
    return (int)v.size();
}

extern "C" {

int dshh_make_fname(const char *path, unsigned p, int k, const char *spacing, const char *suffix,
                    const char *prefix, char *out, size_t cap)
{
    const std::string s = make_fname(path, p, k, spacing, suffix, prefix);
    if (s.size() + 1 > cap) return -1;
    std::memcpy(out, s.c_str(), s.size() + 1);
    return 0;
}

int dshh_write_hll(const char *path, const uint8_t *regs, int p, int estim)
{
    return write_hll(path, regs, p, estim, estim, false, 0.0);
}

int dshh_read_hll(const char *path, uint8_t *regs_out, size_t cap, int *p)
{
    std::vector<uint8_t> r;
    int rc = read_hll(path, r, *p);
    if (rc) return rc;
    if (r.size() > cap) return -1;
    std::memcpy(regs_out, r.data(), r.size());
    return 0;
}

long dshh_append_fastx(const char *path, uint8_t *out, size_t cap, size_t *len)
{
    std::vector<uint8_t> v;
    long n = append_fastx(path, v);
    if (n < 0) return n;
    if (v.size() > cap) return -2;
    if (!v.empty()) std::memcpy(out, v.data(), v.size());
    *len = v.size();
    return n;
}

// the CLI's streaming loader path: parse straight into caller-owned (page-locked) memory at dst[*len ..), no growth
long dshh_append_fastx_into(const char *path, uint8_t *dst, size_t cap, size_t *len)
{
    size_t l = *len;
    const long n = append_fastx_into(path, dst, cap, l);
    if (n >= 0) *len = l;
    return n;
}

// the raw bytes of a plain file behind dst[*len ..) (the device parses them: dsh_sketch_fastx_batch_async)
long dshh_read_raw_into(const char *path, uint8_t *dst, size_t cap, size_t *len)
{
    size_t l = *len;
    const long rc = read_raw_into(path, dst, cap, l);
    if (rc == 0) *len = l;
    return rc;
}

int dshh_sort_paths(const char *joined, char *out, size_t cap)
{
    auto v = unpack(joined);
    sort_paths_by_fsize(v);
    return pack(v, out, cap);
}

int dshh_split_genome_paths(const char *entry, char *out, size_t cap)
{
    return pack(split_genome_paths(entry), out, cap);
}

// render a whole matrix in one of the text/binary formats into a file (test helper)
int dshh_emit_matrix(const char *outpath, int fmt, const char *joined_paths, const float *tri)
{
    const auto paths = unpack(joined_paths);
    const size_t n = paths.size();
    std::FILE *fp = std::fopen(outpath, "wb");
    if (!fp) return -1;
    if (fmt == UT_TSV || fmt == UPPER_TRIANGULAR) {
        emit_header(fp, fmt, paths);
        size_t off = 0;
        for (size_t i = 0; i < n; ++i) {
            emit_ut_row(fp, fmt, paths, i, tri + off);
            off += n - i - 1;
        }
    } else if (fmt == FULL_TSV) {
        emit_full_header(fp, paths);
        for (size_t i = 0; i < n; ++i) emit_full_row(fp, paths, i, tri);
    } else {
        write_binary_header(fp, n);
        std::fwrite(tri, sizeof(float), n * (n - 1) / 2, fp);
    }
    std::fclose(fp);
    return 0;
}

int dshh_fold(const uint8_t *in, int p, int new_p, uint8_t *out)
{
    if (new_p < 1 || new_p >= p) return -1;
    std::vector<uint8_t> v;
    fold_registers(in, p, new_p, v);
    std::memcpy(out, v.data(), v.size());
    return 0;
}

void dshh_union(uint8_t *acc, const uint8_t *other, size_t m) { union_registers(acc, other, m); }

int dshh_write_hll_multi(const char *path, const uint8_t *regs, size_t n, int p, int estim)
{
    return write_hll_multi(path, regs, n, p, estim);
}

int dshh_read_hll_multi(const char *path, uint8_t *regs_out, size_t cap, int *p, size_t *n)
{
    std::vector<uint8_t> r;
    int rc = read_hll_multi(path, r, *p, *n);
    if (rc) return rc;
    if (r.size() > cap) return -1;
    std::memcpy(regs_out, r.data(), r.size());
    return 0;
}

int dshh_write_labels_gz(const char *path, const char *joined_paths)
{
    return write_labels_gz(path, unpack(joined_paths));
}

int dshh_emit_sizes(const char *outpath, const char *joined_paths, const double *card)
{
    std::FILE *fp = std::fopen(outpath, "wb");
    if (!fp) return -1;
    emit_sizes(fp, unpack(joined_paths), card);
    std::fclose(fp);
    return 0;
}

}  // extern "C"

// host.h -- host-side pieces of dashing's `sketch` / `dist` subcommands that sit either side of
// the GPU hot path: FASTA/FASTQ reading, .hll files, cache file names, input ordering and the
// distance-matrix emitters.  Plain C++17 + zlib; no GPU code here.  Each function cites the
// reference lines whose behaviour it reproduces (paths relative to /root/reference).
#pragma once
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

namespace dshh {

// ---- enums, values as in src/enums.h:13-34 -------------------------------------------------
enum EmissionType { MASH_DIST = 0, JI = 1, FULL_MASH_DIST = 3 };
enum EmissionFormat { UT_TSV = 0, BINARY = 1, UPPER_TRIANGULAR = 2, FULL_TSV = 3 };
enum Estim { ORIGINAL = 0, ERTL_IMPROVED = 1, ERTL_MLE = 2 };

// ---- input handling ---------------------------------------------------------------------
// get_paths: one path per line (used for -F / -Q, src/distmain.cpp:113-114).
std::vector<std::string> read_paths_file(const std::string &path);
// for_each_substr (src/substrs.h:7-26): a "genome" may be several files joined by ' '.
std::vector<std::string> split_genome_paths(const std::string &s, char sep = ' ');
// posix_fsizes (src/finalizers.cpp:23-27): total size of the files of one genome entry.
uint64_t genome_file_size(const std::string &entry);
// sort_paths_by_fsize (src/finalizers.cpp:6-21): largest first.  The reference uses
// std::sort (ties in unspecified order) on uint32-truncated sizes; we keep the truncation and use
// a stable sort so equal sizes keep their input order.
void sort_paths_by_fsize(std::vector<std::string> &paths);

// Append the sequence of every record of a FASTA/FASTQ file (plain, gzip or zstd -- by magic number, as dashing's
// zlib/zstd wrapper does, README.md:79) to `out`, records
// separated by one 'N' (k-mers never span records, as with kseq records read one at a time by
// Encoder::for_each, src/sketch_and_cmp.h:342).  Returns number of records, or -1 on open failure.
long append_fastx(const std::string &path, std::vector<uint8_t> &out);
// Same into caller-owned memory (e.g. page-locked staging): appends at dst[len...] and advances len; -2 if the
// `cap` bytes would not hold it.  An uncompressed file never yields more sequence bytes than its size.
long append_fastx_into(const std::string &path, uint8_t *dst, size_t cap, size_t &len);
// The raw bytes of a plain file, as they lie on disk, appended at dst[len...] (the parse happens on the device:
// dsh_sketch_fastx_batch_async).  Returns 0, -1 if the file cannot be opened or read, -2 if `cap` would not hold it (the
// file grew since it was sized).
long read_raw_into(const std::string &path, uint8_t *dst, size_t cap, size_t &len);
bool is_gzip_file(const std::string &path);  // compressed input: gzip (1f 8b) or zstd (28 b5 2f fd) magic

// ---- .hll files (header layout is a best-effort restatement) --------
// make_fname<hll_t> (src/dashing.h:497-526): "<prefix/><genome>.w.<k>.spacing<spacing>.[suf<x>.]<S>.hll"
// (the `ret + std::to_string(...)` at :510 discards its value, so nothing follows ".w").
std::string make_fname(const std::string &path, unsigned sketch_p, int k, const std::string &spacing,
                       const std::string &suffix, const std::string &prefix);
// gz stream: uint32[4]{is_calculated, estim, jestim, 1}, uint32 p, double value, 2^p register bytes.
// level: gz compression level 1..9, 0 = uncompressed ("wT", as union_main does), -1 = zlib default.
int write_hll(const std::string &path, const uint8_t *regs, int p, int estim, int jestim,
              bool is_calculated, double value, int level = -1);
// Accepts the layout above and the older uint8[4] flag block; fills p. Returns 0 or -errno-style.
int read_hll(const std::string &path, std::vector<uint8_t> &regs, int &p);

// Several sketches back to back in ONE gz stream (`sketch -o FILE`, src/sketch_and_cmp.h:466-475,
// 529-536: every sketch is written with the same write(gzFile) as a single-sketch file).
int write_hll_multi(const std::string &path, const uint8_t *regs /*[n][2^p]*/, size_t n, int p, int estim);
// Reads such a stream (n >= 1 sketches of equal p); regs gets n * 2^p bytes.
int read_hll_multi(const std::string &path, std::vector<uint8_t> &regs, int &p, size_t &n);
// "<FILE>.labels.gz": one input path per line, gz (src/sketch_and_cmp.h:466-472)
int write_labels_gz(const std::string &path, const std::vector<std::string> &paths);

// ---- utilities on sketches (`union`, `fold`, `view`; src/union.cpp:33-58, src/dashing.cpp:559-590) ----
// hll_t::operator+= : element-wise register maximum (the sketch of the union of the two sets).
void union_registers(uint8_t *acc, const uint8_t *other, size_t m);
// hll_t::compress(new_p): the sketch the same k-mer stream would have produced at precision
// new_p < p.  Derived from the register rule (src/readfilt.cpp:86-88): an item in register idx with
// value v has hash bits [idx (p bits)][v-1 zeros][1]..., so at new_p its index is idx >> d
// (d = p - new_p) and its value is clz_d(idx & (2^d - 1)) + 1 if those d bits are non-zero, else d + v.
void fold_registers(const uint8_t *in, int p, int new_p, std::vector<uint8_t> &out);
// `view`: human-readable dump (the reference's hll_t::printf lives in the absent sketch submodule;
// the layout here is ours): one header line, then the register values comma-separated.
void print_hll(std::FILE *fp, const std::string &name, const uint8_t *regs, int p);

// ---- emitters -------------------------------------------------------------------------------
// sizes file (src/sketch_and_cmp.h:372-385)
void emit_sizes(std::FILE *fp, const std::vector<std::string> &paths, const double *card);
// header for UT_TSV ("##Names\t...") / PHYLIP ("<N>\n") (src/sketch_and_cmp.h:388-397)
void emit_header(std::FILE *fp, int fmt, const std::vector<std::string> &paths);
// one row of UT_TSV or PHYLIP upper-triangular (submit_emit_dists, src/sketch_and_cmp.h:16-35):
// name, then (i+1) x "\t-" (UT_TSV) or padding to 9 chars (PHYLIP), then "\t%.6g" per value.
void emit_ut_row(std::FILE *fp, int fmt, const std::vector<std::string> &paths, size_t i,
                 const float *row /* n-i-1 values */);
// same row rendered into a string (lets the CLI format many rows on parallel host threads)
void format_ut_row(std::string &out, int fmt, const std::vector<std::string> &paths, size_t i,
                   const float *row);
// FULL_TSV (src/sketch_and_cmp.h:851-877): "#Names" + names; rows "<name>\t" + n x "%0.6g".
void emit_full_header(std::FILE *fp, const std::vector<std::string> &paths);
void emit_full_row(std::FILE *fp, const std::vector<std::string> &paths, size_t i,
                   const float *packed_tri);
// BINARY (distmat/distmat.h:97-99,158-286; src/sketch_and_cmp.h:838-849): '\0', u64 N, floats.
int write_binary_header(std::FILE *fp, uint64_t n);
// "<O>.labels" (src/distmain.cpp:191-200)
int write_labels(const std::string &path, const std::vector<std::string> &paths);

}  // namespace dshh

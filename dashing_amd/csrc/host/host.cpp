// host.cpp -- see host.h.  No GPU code; links zlib.
#include "host.h"

#include <sys/stat.h>
#include <dlfcn.h>
#include <fcntl.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <cerrno>
#include <cstring>

namespace dshh {

std::vector<std::string> read_paths_file(const std::string &path)
{
    std::vector<std::string> out;
    gzFile fp = gzopen(path.c_str(), "rb");
    if (!fp) return out;
    std::string line;
    char buf[65536];
    while (gzgets(fp, buf, sizeof buf)) {
        line += buf;
        if (!line.empty() && line.back() == '\n') {
            while (!line.empty() && (line.back() == '\n' || line.back() == '\r')) line.pop_back();
            if (!line.empty()) out.push_back(line);
            line.clear();
        }
    }
    if (!line.empty()) out.push_back(line);
    gzclose(fp);
    return out;
}

std::vector<std::string> split_genome_paths(const std::string &s, char sep)
{
    std::vector<std::string> out;
    size_t b = 0;
    while (b <= s.size()) {
        size_t e = s.find(sep, b);
        if (e == std::string::npos) e = s.size();
        if (e > b) out.emplace_back(s.substr(b, e - b));
        b = e + 1;
    }
    return out;
}

uint64_t genome_file_size(const std::string &entry)
{
    uint64_t tot = 0;
    for (const auto &f : split_genome_paths(entry)) {
        struct stat st;
        if (::stat(f.c_str(), &st) == 0) tot += (uint64_t)st.st_size;
    }
    return tot;
}

void sort_paths_by_fsize(std::vector<std::string> &paths)
{
    if (paths.size() < 2) return;
    std::vector<std::pair<uint32_t, std::string>> ps;
    ps.reserve(paths.size());
    for (auto &p : paths) ps.emplace_back((uint32_t)genome_file_size(p), std::move(p));
    std::stable_sort(ps.begin(), ps.end(), [](const auto &x, const auto &y) { return x.first > y.first; });
    paths.clear();
    for (auto &p : ps) paths.emplace_back(std::move(p.second));
}

// ---- transparent input: plain, gzip (zlib) or zstd, by magic number -------------------------------------------
// dashing reads every input through zlib's gz* API, which its build swaps for zstd's zlibWrapper (Makefile:58-62,
// README.md:79): plain, gzip'ed and zstd-compressed files all work.  zstd ships here as a runtime library without
// headers, so its streaming API (stable since 1.0) is declared by hand and bound with dlopen; a host without
// libzstd.so.1 reports an error for .zst inputs only.
namespace {

struct ZstdIn {
    const void *src;
    size_t size, pos;
};
struct ZstdOut {
    void *dst;
    size_t size, pos;
};
struct ZstdApi {
    void *(*createDStream)() = nullptr;
    size_t (*freeDStream)(void *) = nullptr;
    size_t (*initDStream)(void *) = nullptr;
    size_t (*decompressStream)(void *, ZstdOut *, ZstdIn *) = nullptr;
    unsigned (*isError)(size_t) = nullptr;
    bool ok = false;
    static const ZstdApi &get()
    {
        static const ZstdApi api = [] {
            ZstdApi a;
            void *h = dlopen("libzstd.so.1", RTLD_NOW | RTLD_LOCAL);
            if (!h) h = dlopen("libzstd.so", RTLD_NOW | RTLD_LOCAL);
            if (!h) return a;
            a.createDStream = (void *(*)())dlsym(h, "ZSTD_createDStream");
            a.freeDStream = (size_t(*)(void *))dlsym(h, "ZSTD_freeDStream");
            a.initDStream = (size_t(*)(void *))dlsym(h, "ZSTD_initDStream");
            a.decompressStream = (size_t(*)(void *, ZstdOut *, ZstdIn *))dlsym(h, "ZSTD_decompressStream");
            a.isError = (unsigned (*)(size_t))dlsym(h, "ZSTD_isError");
            a.ok = a.createDStream && a.freeDStream && a.initDStream && a.decompressStream && a.isError;
            return a;
        }();
        return api;
    }
};

class InStream {
public:
    enum Kind { CLOSED, PLAIN, GZIP, GZPIPE, ZSTD };
    // 0 on success, -ENOENT if the file cannot be opened, -ENOSYS for a zstd file on a host without libzstd,
    // -ENOMEM / -EIO if a decoder cannot be set up.
    // The format is sniffed from the first bytes actually READ (kept and replayed), never with pread or a second
    // open: the path may be a FIFO, /dev/stdin or a `<(zcat x.gz)` process substitution, as dashing's gz* reader
    // accepts (bonsai's Encoder::for_each opens every input with gzopen).
    int open(const std::string &path)
    {
        fd_ = ::open(path.c_str(), O_RDONLY);
        if (fd_ < 0) return -ENOENT;
        // a reused object starts from scratch: no decoder state of the previous input survives
        prelen_ = prepos_ = 0;
        zeof_ = false;
        zframe_done_ = true;  // (as constructed: an empty stream is a clean end; the gzip-pipe path clears it)
        gz_members_ = 0;
        zpos_ = zlen_ = 0;
        while (prelen_ < 4) {
            const ssize_t r = ::read(fd_, pre_ + prelen_, 4 - prelen_);
            if (r <= 0) break;
            prelen_ += (size_t)r;
        }
        const unsigned char *magic = pre_;
        if (prelen_ >= 2 && magic[0] == 0x1f && magic[1] == 0x8b) {
            // seekable: step back over the sniffed bytes (relative: a /dev/stdin redirected from the middle of a file
            // must return to where the stream started, not to offset 0) and let zlib's own reader take it from there
            if (::lseek(fd_, -(off_t)prelen_, SEEK_CUR) != (off_t)-1) {
                prelen_ = 0;
                gz_ = gzdopen(fd_, "rb");
                if (!gz_) return fail_open(-ENOMEM);
                gzbuffer(gz_, 1 << 20);
                kind_ = GZIP;
            } else {  // a pipe: inflate by hand, starting with the bytes already taken
                std::memset(&zl_, 0, sizeof zl_);
                if (inflateInit2(&zl_, 16 + MAX_WBITS) != Z_OK) return fail_open(-ENOMEM);
                zin_.resize(1 << 17);
                std::memcpy(zin_.data(), pre_, prelen_);
                zlen_ = prelen_;
                zpos_ = 0;
                prelen_ = 0;
                zframe_done_ = false;
                kind_ = GZPIPE;
            }
        } else if (prelen_ == 4 && magic[0] == 0x28 && magic[1] == 0xB5 && magic[2] == 0x2F && magic[3] == 0xFD) {
            const ZstdApi &z = ZstdApi::get();
            if (!z.ok) return fail_open(-ENOSYS);
            zs_ = z.createDStream();
            if (!zs_) return fail_open(-ENOMEM);
            if (z.isError(z.initDStream(zs_))) {
                z.freeDStream(zs_);
                zs_ = nullptr;
                return fail_open(-EIO);
            }
            zin_.resize(1 << 17);
            std::memcpy(zin_.data(), pre_, prelen_);  // the magic belongs to the stream
            zlen_ = prelen_;
            zpos_ = 0;
            prelen_ = 0;
            kind_ = ZSTD;
        } else {
            kind_ = PLAIN;  // (pre_ is replayed by read())
        }
        return 0;
    }
    Kind kind() const { return kind_; }
    // up to n bytes; 0 at end of data, -1 on a read / decode error
    ssize_t read(void *dst, size_t n)
    {
        if (kind_ == PLAIN) {
            size_t done = 0;
            while (prepos_ < prelen_ && done < n) ((unsigned char *)dst)[done++] = pre_[prepos_++];
            while (done < n) {
                const ssize_t r = ::read(fd_, (char *)dst + done, n - done);
                if (r < 0) return -1;
                if (r == 0) break;
                done += (size_t)r;
            }
            return (ssize_t)done;
        }
        if (kind_ == GZIP) {
            size_t done = 0;
            while (done < n) {
                const int r = gzread(gz_, (char *)dst + done, (unsigned)std::min<size_t>(n - done, 1u << 30));
                if (r < 0) return -1;
                if (r == 0) break;
                done += (size_t)r;
            }
            return (ssize_t)done;
        }
        if (kind_ == GZPIPE) {  // concatenated members are one stream, as for gzread
            zl_.next_out = (Bytef *)dst;
            size_t left = n;
            while (left) {
                if (zpos_ == zlen_ && !zeof_) {
                    const ssize_t r = ::read(fd_, zin_.data(), zin_.size());
                    if (r < 0) return -1;
                    if (r == 0) zeof_ = true;
                    zlen_ = (size_t)std::max<ssize_t>(r, 0);
                    zpos_ = 0;
                }
                if (zpos_ == zlen_ && zeof_) {
                    if (!zframe_done_) return -1;  // truncated member
                    break;
                }
                if (zframe_done_) {  // more input after a finished member: the next member
                    if (gz_members_ && inflateReset(&zl_) != Z_OK) return -1;
                    zframe_done_ = false;
                }
                zl_.next_in = (Bytef *)zin_.data() + zpos_;
                zl_.avail_in = (uInt)(zlen_ - zpos_);
                zl_.avail_out = (uInt)std::min<size_t>(left, 1u << 30);
                const uInt out0 = zl_.avail_out;
                const int rc = inflate(&zl_, Z_NO_FLUSH);
                zpos_ = zlen_ - zl_.avail_in;
                left -= out0 - zl_.avail_out;
                if (rc == Z_STREAM_END) {
                    zframe_done_ = true;
                    ++gz_members_;
                } else if (rc != Z_OK && rc != Z_BUF_ERROR) {
                    return -1;
                }
            }
            return (ssize_t)(n - left);
        }
        if (kind_ == ZSTD) {
            const ZstdApi &z = ZstdApi::get();
            ZstdOut out{dst, n, 0};
            while (out.pos < out.size) {
                if (zpos_ == zlen_ && !zeof_) {
                    const ssize_t r = ::read(fd_, zin_.data(), zin_.size());
                    if (r < 0) return -1;
                    if (r == 0) zeof_ = true;
                    zlen_ = (size_t)std::max<ssize_t>(r, 0);
                    zpos_ = 0;
                }
                // with the input exhausted the decoder may still hold decoded bytes: keep draining until it yields
                // nothing; the stream ends cleanly only on a frame boundary (return value 0 of the last call)
                const bool drained_input = zpos_ == zlen_ && zeof_;
                if (drained_input && zframe_done_) break;
                ZstdIn in{zin_.data(), zlen_, zpos_};
                const size_t before = out.pos;
                const size_t rc = z.decompressStream(zs_, &out, &in);
                zpos_ = in.pos;
                if (z.isError(rc)) return -1;
                zframe_done_ = rc == 0;
                if (drained_input && out.pos == before) {
                    if (!zframe_done_) return -1;  // the last frame is incomplete: truncated file
                    break;
                }
            }
            return (ssize_t)out.pos;
        }
        return -1;
    }
    void close()
    {
        if (kind_ == GZIP) gzclose(gz_);  // closes fd_ too
        else if (fd_ >= 0) ::close(fd_);
        if (kind_ == GZPIPE) inflateEnd(&zl_);
        if (zs_) ZstdApi::get().freeDStream(zs_);
        zs_ = nullptr;
        fd_ = -1;
        kind_ = CLOSED;
    }
    ~InStream()
    {
        if (kind_ != CLOSED) close();
    }

private:
    int fail_open(int rc)
    {
        if (fd_ >= 0) ::close(fd_);
        fd_ = -1;
        return rc;
    }
    Kind kind_ = CLOSED;
    int fd_ = -1;
    unsigned char pre_[4] = {0, 0, 0, 0};  // the bytes the format was sniffed from
    size_t prelen_ = 0, prepos_ = 0;
    z_stream zl_;
    unsigned gz_members_ = 0;
    gzFile gz_ = nullptr;
    void *zs_ = nullptr;
    std::vector<char> zin_;
    size_t zpos_ = 0, zlen_ = 0;
    bool zeof_ = false, zframe_done_ = true;  // (an empty file is a clean end)
};

}  // namespace

namespace {

struct VecSink {
    std::vector<uint8_t> &out;
    bool overflow = false;
    void push(uint8_t b) { out.push_back(b); }
    void append(const uint8_t *p, size_t len) { out.insert(out.end(), p, p + len); }
};
struct BufSink {  // caller-owned memory (page-locked staging): no growth, overflow is recorded
    uint8_t *dst;
    size_t cap, len = 0;
    bool overflow = false;
    void push(uint8_t b)
    {
        if (len < cap) dst[len++] = b;
        else overflow = true;
    }
    void append(const uint8_t *p, size_t n)
    {
        if (len + n <= cap) {
            std::memcpy(dst + len, p, n);
            len += n;
        } else {
            overflow = true;
        }
    }
};

// FASTA/FASTQ state machine over a stream of text blocks: 0 = expect header, 1 = sequence lines, 2 = quality
// lines; whole line pieces are handed to the sink at once (memchr for the newline), not byte by byte
template <class Sink>
struct FastxParser {
    Sink &out;
    long nrec = 0;
    int state = 0;
    bool at_line_start = true, skipping_line = false;
    size_t seq_len = 0, qual_len = 0;
    explicit FastxParser(Sink &s) : out(s) {}
    void take(const char *p, size_t len)  // a piece of a line's content
    {
        if (skipping_line || len == 0) return;
        if (std::memchr(p, '\r', len)) {  // rare: strip carriage returns the slow way
            for (size_t t = 0; t < len; ++t) {
                if (p[t] == '\r') continue;
                if (state == 1) {
                    out.push((uint8_t)p[t]);
                    ++seq_len;
                } else if (state == 2) {
                    ++qual_len;
                }
            }
            return;
        }
        if (state == 1) {
            out.append((const uint8_t *)p, len);
            seq_len += len;
        } else if (state == 2) {
            qual_len += len;
        }
    }
    void feed(const char *buf, size_t N)
    {
        size_t i = 0;
        while (i < N) {
            if (at_line_start) {
                const char c = buf[i];
                if (c == '\n') {  // empty line
                    if (state == 2 && qual_len >= seq_len) state = 0;
                    ++i;
                    continue;
                }
                at_line_start = false;
                if (state != 2 && (c == '>' || c == '@')) {  // new record header
                    if (nrec) out.push('N');
                    ++nrec;
                    state = 1;
                    seq_len = qual_len = 0;
                    skipping_line = true;
                } else if (state == 1 && c == '+') {  // FASTQ separator line
                    state = 2;
                    skipping_line = true;
                }
            }
            const char *nl = (const char *)std::memchr(buf + i, '\n', N - i);
            const size_t e = nl ? (size_t)(nl - buf) : N;
            take(buf + i, e - i);
            if (nl) {
                at_line_start = true;
                skipping_line = false;
                if (state == 2 && qual_len >= seq_len) state = 0;
                i = e + 1;
            } else {
                i = e;
            }
        }
    }
};

template <class Sink>
long parse_fastx(const std::string &path, Sink &sink)
{
    InStream in;  // plain text through read(2) (zlib's transparent mode copies every byte once more), gzip, zstd
    if (in.open(path) != 0) return -1;
    FastxParser<Sink> ps(sink);
    std::vector<char> buf(1 << 20);
    ssize_t n;
    while ((n = in.read(buf.data(), buf.size())) > 0) ps.feed(buf.data(), (size_t)n);
    return n < 0 ? -1 : ps.nrec;
}

}  // namespace

long append_fastx(const std::string &path, std::vector<uint8_t> &out)
{
    VecSink s{out};
    return parse_fastx(path, s);
}

long append_fastx_into(const std::string &path, uint8_t *dst, size_t cap, size_t &len)
{
    BufSink s{dst + len, cap > len ? cap - len : 0};
    const long nrec = parse_fastx(path, s);
    if (nrec < 0) return nrec;
    if (s.overflow) return -2;
    len += s.len;
    return nrec;
}

long read_raw_into(const std::string &path, uint8_t *dst, size_t cap, size_t &len)
{
    const int fd = ::open(path.c_str(), O_RDONLY);
    if (fd < 0) return -1;
    long rc = 0;
    for (;;) {
        if (len == cap) {  // full: fine if the file ends here
            char extra;
            const ssize_t r = ::read(fd, &extra, 1);
            rc = r == 0 ? 0 : (r < 0 ? -1 : -2);
            break;
        }
        const ssize_t r = ::read(fd, dst + len, std::min<size_t>(cap - len, (size_t)1 << 26));
        if (r < 0) {
            if (errno == EINTR) continue;
            rc = -1;
            break;
        }
        if (r == 0) break;
        len += (size_t)r;
    }
    ::close(fd);
    return rc;
}

bool is_gzip_file(const std::string &path)
{
    // "its sequence length cannot be bounded by its size": compressed (gzip / zstd magic), or not a regular file at
    // all -- a FIFO, /dev/stdin, a process substitution -- which must not be opened here (reading the magic would eat
    // it, closing would end the writer): such inputs take the growable parse path
    struct stat st;
    if (::stat(path.c_str(), &st) != 0) return false;
    if (!S_ISREG(st.st_mode) || st.st_size == 0) return true;
    const int fd = ::open(path.c_str(), O_RDONLY);
    if (fd < 0) return false;
    unsigned char magic[4] = {0, 0, 0, 0};
    const ssize_t got = ::pread(fd, magic, 4, 0);
    ::close(fd);
    return (got >= 2 && magic[0] == 0x1f && magic[1] == 0x8b) ||
           (got == 4 && magic[0] == 0x28 && magic[1] == 0xB5 && magic[2] == 0x2F && magic[3] == 0xFD);
}

std::string make_fname(const std::string &path, unsigned sketch_p, int k, const std::string &spacing,
                       const std::string &suffix, const std::string &prefix)
{
    std::string ret(prefix);
    if (!ret.empty()) ret += '/';
    {
        const char *p = std::strchr(path.c_str(), ' ');
        p = p ? p + 1 : path.c_str();
        const char *p2;
        if (!ret.empty() && (p2 = std::strrchr(p, '/'))) ret += std::string(p2 + 1);
        else ret += p;
    }
    ret += ".w";
    ret += ".";
    ret += std::to_string(k);
    ret += ".spacing";
    ret += spacing;
    ret += '.';
    if (!suffix.empty()) {
        ret += "suf";
        ret += suffix;
        ret += '.';
    }
    ret += std::to_string(sketch_p);
    ret += ".hll";
    return ret;
}

static int gz_put_sketch(gzFile fp, const uint8_t *regs, int p, int estim, int jestim, bool is_calculated, double value);

int write_hll(const std::string &path, const uint8_t *regs, int p, int estim, int jestim,
              bool is_calculated, double value, int level)
{
    char mode[8] = "wb";
    if (level == 0) std::snprintf(mode, sizeof mode, "wT");
    else if (level > 0) std::snprintf(mode, sizeof mode, "wb%d", level % 10);
    gzFile fp = gzopen(path.c_str(), mode);
    if (!fp) return -EIO;
    int rc = gz_put_sketch(fp, regs, p, estim, jestim, is_calculated, value);
    if (gzclose(fp) != Z_OK && !rc) rc = -EIO;
    return rc;
}

int read_hll(const std::string &path, std::vector<uint8_t> &regs, int &p)
{
    InStream fp;
    if (int rc = fp.open(path)) return rc;
    std::vector<uint8_t> all;
    uint8_t buf[1 << 16];
    ssize_t n;
    while ((n = fp.read(buf, sizeof buf)) > 0) all.insert(all.end(), buf, buf + n);
    fp.close();
    if (n < 0) return -EIO;  // read or decode error (truncated / corrupt gzip or zstd stream), not a layout question
    auto try_layout = [&](size_t hdr, size_t np_off) -> bool {
        if (all.size() <= hdr) return false;
        const size_t m = all.size() - hdr;
        if (m & (m - 1)) return false;
        uint32_t np;
        std::memcpy(&np, all.data() + np_off, 4);
        if (np < 4 || np > 30 || ((size_t)1 << np) != m) return false;
        // a register of a precision-np HLL is at most 64 - np + 1 (src/readfilt.cpp:86-88): anything larger is a
        // corrupt file or the other header layout read at the wrong offset
        const uint8_t lim = (uint8_t)(64 - (int)np + 1);
        for (size_t t = hdr; t < all.size(); ++t)
            if (all[t] > lim) return false;
        p = (int)np;
        regs.assign(all.begin() + hdr, all.end());
        return true;
    };
    if (try_layout(28, 16)) return 0;  // uint32[4] flags, uint32 p, double value
    if (try_layout(16, 4)) return 0;   // uint8[4] flags, uint32 p, double value
    return -EINVAL;
}

static int gz_put_sketch(gzFile fp, const uint8_t *regs, int p, int estim, int jestim, bool is_calculated, double value)
{
    const uint32_t bf[4] = {(uint32_t)is_calculated, (uint32_t)estim, (uint32_t)jestim, 1u};
    const uint32_t np = (uint32_t)p;
    int ok = gzwrite(fp, bf, sizeof bf) == (int)sizeof bf;
    ok = ok && gzwrite(fp, &np, sizeof np) == (int)sizeof np;
    ok = ok && gzwrite(fp, &value, sizeof value) == (int)sizeof value;
    const size_t m = (size_t)1 << p;
    for (size_t off = 0; ok && off < m; off += (size_t)1 << 30) {  // gzwrite takes an unsigned length
        const unsigned len = (unsigned)std::min<size_t>(m - off, (size_t)1 << 30);
        ok = gzwrite(fp, regs + off, len) == (int)len;
    }
    return ok ? 0 : -EIO;
}

int write_hll_multi(const std::string &path, const uint8_t *regs, size_t n, int p, int estim)
{
    gzFile fp = gzopen(path.c_str(), "wb");
    if (!fp) return -EIO;
    int rc = 0;
    for (size_t i = 0; i < n && !rc; ++i) rc = gz_put_sketch(fp, regs + (i << p), p, estim, estim, false, 0.0);
    if (gzclose(fp) != Z_OK && !rc) rc = -EIO;
    return rc;
}

int read_hll_multi(const std::string &path, std::vector<uint8_t> &regs, int &p, size_t &n)
{
    InStream fp;
    if (int rc = fp.open(path)) return rc;
    regs.clear();
    n = 0;
    p = -1;
    for (;;) {
        uint8_t hdr[28];
        const ssize_t got = fp.read(hdr, sizeof hdr);
        if (got == 0) break;
        if (got < 0) return -EIO;
        if (got != (ssize_t)sizeof hdr) return -EINVAL;  // (before any field of the header is looked at)
        uint32_t np;
        std::memcpy(&np, hdr + 16, 4);
        if (np < 4 || np > 30 || (p >= 0 && (int)np != p)) return -EINVAL;
        p = (int)np;
        const size_t m = (size_t)1 << p, at = regs.size();
        regs.resize(at + m);
        const ssize_t gotr = fp.read(regs.data() + at, m);
        if (gotr < 0) return -EIO;
        if (gotr != (ssize_t)m) return -EINVAL;
        ++n;
    }
    return n ? 0 : -EINVAL;
}

int write_labels_gz(const std::string &path, const std::vector<std::string> &paths)
{
    gzFile fp = gzopen(path.c_str(), "w");
    if (!fp) return -EIO;
    for (const auto &s : paths) {
        gzwrite(fp, s.data(), (unsigned)s.size());
        gzputc(fp, '\n');
    }
    return gzclose(fp) == Z_OK ? 0 : -EIO;
}

void union_registers(uint8_t *acc, const uint8_t *other, size_t m)
{
    for (size_t i = 0; i < m; ++i) acc[i] = std::max(acc[i], other[i]);
}

void fold_registers(const uint8_t *in, int p, int new_p, std::vector<uint8_t> &out)
{
    const int d = p - new_p;
    const size_t m = (size_t)1 << p;
    out.assign((size_t)1 << new_p, 0);
    for (size_t idx = 0; idx < m; ++idx) {
        const unsigned v = in[idx];
        if (!v) continue;  // nothing ever hashed here
        const size_t low = idx & (((size_t)1 << d) - 1);
        unsigned nv;
        if (low) {
            int lead = 0;  // leading zeros of `low` within its d-bit field
            while (!((low >> (d - 1 - lead)) & 1)) ++lead;
            nv = (unsigned)lead + 1;
        } else {
            nv = (unsigned)d + v;
        }
        uint8_t &dst = out[idx >> d];
        if (nv > dst) dst = (uint8_t)nv;
    }
}

void print_hll(std::FILE *fp, const std::string &name, const uint8_t *regs, int p)
{
    const size_t m = (size_t)1 << p;
    size_t zeros = 0;
    unsigned mx = 0;
    for (size_t i = 0; i < m; ++i) {
        zeros += regs[i] == 0;
        mx = std::max<unsigned>(mx, regs[i]);
    }
    std::fprintf(fp, "#%s\tp=%d\tregisters=%zu\tempty=%zu\tmax=%u\n", name.c_str(), p, m, zeros, mx);
    std::string s;
    char num[8];
    for (size_t i = 0; i < m; ++i) {
        s.append(num, (size_t)std::snprintf(num, sizeof num, i ? ",%u" : "%u", (unsigned)regs[i]));
        if (s.size() > (1u << 16)) {
            std::fwrite(s.data(), 1, s.size(), fp);
            s.clear();
        }
    }
    s += '\n';
    std::fwrite(s.data(), 1, s.size(), fp);
}

void emit_sizes(std::FILE *fp, const std::vector<std::string> &paths, const double *card)
{
    std::fputs("#Path\tSize (est.)\n", fp);
    for (size_t i = 0; i < paths.size(); ++i) std::fprintf(fp, "%s\t%zu\n", paths[i].c_str(), (size_t)card[i]);
    std::fflush(fp);
}

void emit_header(std::FILE *fp, int fmt, const std::vector<std::string> &paths)
{
    if (fmt == UT_TSV) {
        std::string s("##Names\t");
        for (const auto &p : paths) {
            s += p;
            s += '\t';
        }
        s.back() = '\n';
        std::fwrite(s.data(), 1, s.size(), fp);
    } else if (fmt == UPPER_TRIANGULAR) {
        std::fprintf(fp, "%zu\n", paths.size());
    }
    std::fflush(fp);
}

void emit_ut_row(std::FILE *fp, int fmt, const std::vector<std::string> &paths, size_t i, const float *row)
{
    std::string s;
    format_ut_row(s, fmt, paths, i, row);
    std::fwrite(s.data(), 1, s.size(), fp);
}

void format_ut_row(std::string &s, int fmt, const std::vector<std::string> &paths, size_t i, const float *row)
{
    const size_t n = paths.size();
    s = paths[i];
    if (fmt == UT_TSV) {
        for (size_t k = 0; k < i + 1; ++k) s += "\t-";
    } else if (s.size() < 9) {
        s.append(9 - s.size(), ' ');
    }
    char num[64];
    for (size_t k = 0; k + i + 1 < n; ++k) {
        const int len = std::snprintf(num, sizeof num, "\t%.6g", (double)row[k]);
        s.append(num, (size_t)len);
    }
    s += '\n';
}

void emit_full_header(std::FILE *fp, const std::vector<std::string> &paths)
{
    std::fputs("#Names", fp);
    for (size_t i = 0; i < paths.size(); ++i) {
        std::fputs(paths[i].c_str(), fp);
        std::fputc(i == paths.size() - 1 ? '\n' : '\t', fp);
    }
}

void emit_full_row(std::FILE *fp, const std::vector<std::string> &paths, size_t i, const float *tri)
{
    const size_t n = paths.size();
    auto at = [&](size_t r, size_t c) -> float {
        if (r == c) return 0.f;
        if (r > c) std::swap(r, c);
        return tri[r * (2 * n - r - 1) / 2 + c - (r + 1)];
    };
    std::fprintf(fp, "%s\t", paths[i].c_str());
    size_t j;
    for (j = 0; j + 1 < n; ++j) std::fprintf(fp, "%0.6g\t", (double)at(i, j));
    std::fprintf(fp, "%0.6g\n", (double)at(i, j));
}

int write_binary_header(std::FILE *fp, uint64_t n)
{
    if (std::fputc('\0', fp) == EOF) return -EIO;
    return std::fwrite(&n, sizeof n, 1, fp) == 1 ? 0 : -EIO;
}

int write_labels(const std::string &path, const std::vector<std::string> &paths)
{
    std::FILE *fp = std::fopen(path.c_str(), "wb");
    if (!fp) return -EIO;
    for (const auto &p : paths) {
        std::fwrite(p.data(), p.size(), 1, fp);
        std::fputc('\n', fp);
    }
    std::fclose(fp);
    return 0;
}

}  // namespace dshh

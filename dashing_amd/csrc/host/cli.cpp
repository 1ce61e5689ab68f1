// cli.cpp -- `dashing-amd`: dashing's `sketch` and `dist` (= `cmp` = `setdist`) subcommands
// (src/main.cpp:22-44) with the two hot loops running on MI355X through the C-ABI
// (include/dashing_hip.h).  Flags keep dashing's spellings (src/dashing.h:35-104,
// src/distmain.cpp:47-100, src/dashing.cpp:253-337): NB -S is log2(sketch bytes) = HLL precision
// and -p is host threads.  HLL sketches only; flags selecting other sketch types / encoders are
// rejected.  There is no CPU compute path: without a gfx950 device the program exits non-zero.
#include <getopt.h>
#include <omp.h>

#include <atomic>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cstdarg>
#include <chrono>
#include <future>
#include <memory>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../../include/dashing_hip.h"
#include "host.h"

using namespace dshh;

static const char *kVersion = "dashing-amd 0.1 (gfx950; HLL sketch/dist hot path of dashing v1)";

[[noreturn]] static void die(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    std::vfprintf(stderr, fmt, ap);
    va_end(ap);
    std::fputc('\n', stderr);
    std::exit(EXIT_FAILURE);
}

#define DSH(ctx, call)                                                             \
    do {                                                                           \
        int rc_ = (call);                                                          \
        if (rc_) die("[dashing-amd] %s failed (%d): %s", #call, rc_, dsh_last_error(ctx)); \
    } while (0)

static bool isfile(const std::string &p)
{
    struct stat st;
    return ::stat(p.c_str(), &st) == 0 && S_ISREG(st.st_mode);
}

static void usage(const char *sub)
{
    std::fprintf(stderr,
        "%s\nUsage: dashing-amd %s [options] genome1 genome2 ... | -F paths.txt\n"
        "  -k, --kmer-length INT    k (<=32) [31]\n"
        "  -S, --sketch-size INT    log2 sketch size in bytes = HLL precision p [10]\n"
        "  -p, --nthreads INT       host threads for FASTA parsing [1]\n"
        "  -F, --paths FILE         file with one genome path per line\n"
        "  -P, --prefix DIR / -x, --suffix STR   sketch cache location / name suffix\n"
        "  -C, --no-canon           do not canonicalise k-mers\n"
        "  --device INT             GPU ordinal [0]\n"
        "  --ngpus INT | --devices a,b,..  (dist) share the rows of the matrix between several GPUs: -b to a file is written\n"
        "                           by every GPU at its own offsets, other outputs are gathered to the first GPU over RCCL\n", kVersion, sub);
    if (!std::strcmp(sub, "sketch")) {
        std::fprintf(stderr, "  -c, --skip-cached        skip genomes whose .hll already exists\n"
                             "  -o FILE                  write all sketches into one file (+ FILE.labels.gz) instead of one .hll per genome\n");
    } else {
        std::fprintf(stderr,
            "  -o, --out-sizes FILE     cardinalities [stdout]\n"
            "  -O, --out-dists FILE     distances [stdout]\n"
            "  -b/-U/-T                 emit binary / PHYLIP upper-triangular / full TSV [upper-triangular TSV]\n"
            "  -M/-l                    Mash distance / full Mash distance [Jaccard index]\n"
            "  -E/-I/-m                 ORIGINAL / ERTL_IMPROVED / ERTL_MLE estimator [ERTL_MLE]\n"
            "  -W, --cache-sketches     read/write <genome>.w.<k>.spacing.<S>.hll next to the input (or in -P)\n"
            "  -H, --presketched        inputs are .hll files\n"
            "  --avoid-sorting          keep input order (default: largest file first)\n");
    }
    std::exit(EXIT_FAILURE);
}

struct Opts {
    int k = 31, S = 10, nthreads = 1, canon = 1, device = 0;
    int estim = ERTL_MLE, result_type = JI, fmt = UT_TSV;
    int cache = 0, presketched = 0, avoid_sorting = 0, skip_cached = 0;
    unsigned nneighbors = 0;  // --nearest-neighbors
    int rccl = 0;             // --rccl: deliver the rows through the RCCL exchange of the C-ABI even with one device
    std::vector<int> devices;  // --devices a,b,... / --ngpus G: GPUs sharing the all-pairs rows (binary output)
    std::string paths_file, prefix, suffix, spacing, out_sizes, out_dists;
    std::vector<std::string> inpaths, querypaths;
};

enum { OPT_PRESKETCHED = 1000, OPT_AVOID_SORT, OPT_DEVICE, OPT_NPERBATCH, OPT_NN, OPT_NGPUS, OPT_DEVICES, OPT_RCCL, OPT_UNSUPPORTED };

static Opts parse(int argc, char **argv, bool is_dist)
{
    Opts o;
    static const option longopts[] = {
        {"kmer-length", required_argument, nullptr, 'k'}, {"sketch-size", required_argument, nullptr, 'S'},
        {"nthreads", required_argument, nullptr, 'p'}, {"paths", required_argument, nullptr, 'F'},
        {"prefix", required_argument, nullptr, 'P'}, {"suffix", required_argument, nullptr, 'x'},
        {"no-canon", no_argument, nullptr, 'C'}, {"out-sizes", required_argument, nullptr, 'o'},
        {"out-dists", required_argument, nullptr, 'O'}, {"emit-binary", no_argument, nullptr, 'b'},
        {"phylip", no_argument, nullptr, 'U'}, {"full-tsv", no_argument, nullptr, 'T'},
        {"mash-dist", no_argument, nullptr, 'M'}, {"full-mash-dist", no_argument, nullptr, 'l'},
        // dashing declares these long forms with required_argument (LO_ARG, src/dashing.h:62-64);
        // that is a quirk, scripts use the short forms.  We take them without an argument.
        {"original", no_argument, nullptr, 'E'}, {"improved", no_argument, nullptr, 'I'},
        {"ertl-mle", no_argument, nullptr, 'm'}, {"cache-sketches", no_argument, nullptr, 'W'},
        {"presketched", no_argument, nullptr, OPT_PRESKETCHED}, {"avoid-sorting", no_argument, nullptr, OPT_AVOID_SORT},
        {"skip-cached", no_argument, nullptr, 'c'}, {"device", required_argument, nullptr, OPT_DEVICE},
        {"ngpus", required_argument, nullptr, OPT_NGPUS}, {"devices", required_argument, nullptr, OPT_DEVICES},
        {"rccl", no_argument, nullptr, OPT_RCCL},
        {"nperbatch", required_argument, nullptr, OPT_NPERBATCH}, {"spacing", required_argument, nullptr, 's'},
        {"window-size", required_argument, nullptr, 'w'}, {"help", no_argument, nullptr, 'h'},
        {"use-bb-minhash", no_argument, nullptr, OPT_UNSUPPORTED}, {"use-range-minhash", no_argument, nullptr, OPT_UNSUPPORTED},
        {"use-bloom-filter", no_argument, nullptr, OPT_UNSUPPORTED}, {"use-nthash", no_argument, nullptr, OPT_UNSUPPORTED},
        {"use-cyclic-hash", no_argument, nullptr, OPT_UNSUPPORTED}, {"countmin", no_argument, nullptr, OPT_UNSUPPORTED},
        {"nearest-neighbors", required_argument, nullptr, OPT_NN},
        // second arm of result_cmp (src/dashing.h:577-588); flag numbers as in DIST_LONG_OPTS
        {"sizes", no_argument, nullptr, 'Z'}, {"containment-index", no_argument, nullptr, 131},
        {"containment-dist", no_argument, nullptr, 132}, {"full-containment-dist", no_argument, nullptr, 133},
        {"symmetric-containment-index", no_argument, nullptr, 137}, {"symmetric-containment-dist", no_argument, nullptr, 138},
        {"query-paths", required_argument, nullptr, 'Q'},
        {nullptr, 0, nullptr, 0}};
    int co;
    optind = 1;
    while ((co = getopt_long(argc, argv, "k:S:p:F:P:x:Co:O:bUTMlEImWHcs:w:eh?8yJQ:Z", longopts, nullptr)) >= 0) {
        switch (co) {
        case 'k': o.k = std::atoi(optarg); break;
        case 'S': o.S = std::atoi(optarg); break;
        case 'p': o.nthreads = std::max(1, std::atoi(optarg)); break;
        case 'F': o.paths_file = optarg; break;
        case 'P': o.prefix = optarg; break;
        case 'x': o.suffix = optarg; break;
        case 'C': o.canon = 0; break;
        case 'o': o.out_sizes = optarg; break;
        case 'O': o.out_dists = optarg; break;
        case 'b': o.fmt = BINARY; break;
        case 'U': o.fmt = UPPER_TRIANGULAR; break;
        case 'T': o.fmt = FULL_TSV; break;
        case 'M': o.result_type = MASH_DIST; break;
        case 'l': o.result_type = FULL_MASH_DIST; break;
        case 'E': o.estim = ORIGINAL; break;
        case 'I': o.estim = ERTL_IMPROVED; break;
        case 'm': o.estim = ERTL_MLE; break;
        case 'W': o.cache = 1; break;
        case 'H': case OPT_PRESKETCHED: o.presketched = 1; break;  // dashing ignores short -H; we honour it
        case OPT_AVOID_SORT: o.avoid_sorting = 1; break;
        case 'c': o.skip_cached = 1; break;
        case OPT_DEVICE: o.device = std::atoi(optarg); break;
        case OPT_NGPUS: {
            const int g = std::atoi(optarg);
            if (g < 1) die("--ngpus needs a positive count");
            o.devices.clear();
            for (int d = 0; d < g; ++d) o.devices.push_back(o.device + d);
            break;
        }
        case OPT_DEVICES: {  // explicit list; a device may repeat (used to test the sharding on one GPU)
            o.devices.clear();
            for (const char *q = optarg; *q;) {
                o.devices.push_back(std::atoi(q));
                while (*q && *q != ',') ++q;
                if (*q == ',') ++q;
            }
            break;
        }
        case OPT_RCCL: o.rccl = 1; break;
        case OPT_NPERBATCH: case 'e': break;  // accepted, no effect here
        case 's': if (optarg && *optarg) die("spaced seeds are out of scope (HLL hot path only)"); break;
        case 'w': if (std::atoi(optarg) > 0) die("minimizer windows are out of scope (HLL hot path only)"); break;
        case 'Z': o.result_type = 2; break;
        case 131: o.result_type = 5; break;
        case 132: o.result_type = 6; break;
        case 133: o.result_type = 4; break;
        case 137: o.result_type = 7; break;
        case 138: o.result_type = 8; break;
        case 'Q': o.querypaths = read_paths_file(optarg); break;
        case OPT_NN:
            if (std::atoi(optarg) <= 0) die("--nearest-neighbors needs a positive count");
            o.nneighbors = (unsigned)std::atoi(optarg);
            break;
        case '8': case 'y': case 'J': case OPT_UNSUPPORTED:
            die("this option selects a sketch type / emitter outside the HLL sketch+dist hot path");
        default: usage(is_dist ? "dist" : "sketch");
        }
    }
    if (o.k < 1 || o.k > 32) die("k must be in [1,32] for 2-bit encoded k-mers (src/distmain.cpp:101)");
    if (o.S < 4 || o.S > 17) die("-S (log2 sketch bytes) must be in [4,17]");
    o.inpaths = o.paths_file.empty() ? std::vector<std::string>(argv + optind, argv + argc) : read_paths_file(o.paths_file);
    if (o.inpaths.empty()) {
        std::fprintf(stderr, "No paths. See usage.\n");
        usage(is_dist ? "dist" : "sketch");
    }
    return o;
}

static double now_s()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
static const bool g_timing = std::getenv("DSH_TIMING") != nullptr;  // phase times on stderr
// (DSH_TIMING with DSH_T0 = the launcher's wall clock in seconds since the epoch: where the process is, seen from outside)
static void since_launch(const char *what)
{
    const char *t0 = std::getenv("DSH_T0");
    if (!g_timing || !t0) return;
    const double now = std::chrono::duration<double>(std::chrono::system_clock::now().time_since_epoch()).count();
    std::fprintf(stderr, "[timing] %s: %.3f s after the launch\n", what, now - std::atof(t0));
}

// File bytes per sketching batch, and the page-locked staging buffers the batches cycle through.  The batches of a large
// input travel back to back: batch b is parsed (or, for plain FASTA, just read: the device decodes it) into buffer b % 3
// and enqueued at once -- host-to-device copy + kernels behind the previous batch on the context's stream -- while the
// host already fills the next buffer; a buffer is only waited for when its turn comes again (a ticket per batch).
// Round 5 waited for batch b - 1 before enqueueing batch b (two buffers): 4.9 ms per 128 MB = 27 GB/s where the copy
// alone takes 2.6 ms (profiles/rd6c/cli_e2e_timing.jsonl).
static const size_t kSketchBatchBytes = (size_t)48 << 20;
constexpr int kStageBufs = 3;
// plain FASTA is decoded on the device (dsh_sketch_fastx_batch_async: the host only read()s the file bytes into the
// staging); DSH_HOST_PARSE=1 keeps the host parser for everything (A/B, and the path compressed inputs and pipes take anyway)
static const bool g_device_parse = std::getenv("DSH_HOST_PARSE") == nullptr;

// page-locked staging is only worth allocating (pinning runs at ~3 GB/s) when the input does not fit one batch: the
// first batch is parsed into pageable memory anyway
static size_t staging_bytes_for(const std::vector<std::string> &paths)
{
    uint64_t total = 0;
    for (const auto &p : paths) {
        total += genome_file_size(p);
        if (total > kSketchBatchBytes) return kSketchBatchBytes + (kSketchBatchBytes >> 3);
    }
    return 0;
}

// The GPU context is created on its own thread while the host already parses the first batch: bringing the HIP
// runtime up costs ~0.1-0.2 s, as much as reading gigabases of FASTA on 16 threads.  The same thread then page-locks the
// staging buffers ONE BY ONE (stage_ready counts them): the main thread takes each as soon as it exists.
struct CtxFuture {
    std::promise<dsh_ctx *> ready;
    std::future<dsh_ctx *> fut;
    std::future<void> done;  // the context thread itself
    dsh_ctx *ctx = nullptr;
    uint8_t *stage[kStageBufs] = {};  // page-locked staging for fill_sketches, allocated on the ctx thread too
    std::atomic<int> stage_ready{0};
    size_t stage_cap = 0;
    CtxFuture(int device, size_t n, int S, size_t staging_bytes = 0)
    {
        fut = ready.get_future();
        stage_cap = staging_bytes;
        done = std::async(std::launch::async, [this, device, n, S, staging_bytes]() {
            const double t0 = now_s();
            dsh_ctx *c = nullptr;
            if (int rc = dsh_create(device, &c)) die("[dashing-amd] no usable gfx950 device (dsh_create = %d); there is no CPU fallback", rc);
            DSH(c, dsh_sketches_alloc(c, n, S));
            (void)dsh_preload(device, DSH_PRELOAD_SKETCH);  // (the main thread is still staging its first batch)
            ready.set_value(c);  // the main thread may use the context from here on
            const double t1 = now_s();
            if (staging_bytes) {
                for (int b = 0; b < kStageBufs; ++b) {
                    if (!(stage[b] = (uint8_t *)dsh_alloc_host(staging_bytes))) die("could not allocate %zu bytes of pinned host memory", staging_bytes);
                    stage_ready.store(b + 1, std::memory_order_release);
                }
            }
            (void)dsh_preload(device, DSH_PRELOAD_COMPARE);  // (beside the main thread's sketching: dist finds its kernels loaded)
            if (g_timing) std::fprintf(stderr, "[timing] on the context thread: dsh_create + alloc %.3f s, pinned staging %.3f s\n", t1 - t0, now_s() - t1);
        });
    }
    dsh_ctx *get()
    {
        if (!ctx) ctx = fut.get();
        return ctx;
    }
    // staging buffer b, once the context thread has page-locked it (nullptr: none were asked for)
    uint8_t *take_stage(int b)
    {
        if (!stage_cap) return nullptr;
        while (stage_ready.load(std::memory_order_acquire) <= b) std::this_thread::sleep_for(std::chrono::microseconds(100));
        uint8_t *p = stage[b];
        stage[b] = nullptr;
        return p;
    }
    ~CtxFuture()
    {
        if (done.valid()) done.get();
        for (auto &b : stage)
            if (b) dsh_free_host(b);
    }
};

// Hot loop 1 (src/sketch_and_cmp.h:314-360, 484-528) as a stream: genomes are taken in batches of <= ~48 MB of
// file bytes; a batch is staged by the host threads STRAIGHT INTO page-locked memory (every genome has its region
// reserved from the file sizes) -- the raw bytes of plain FASTA files, which the device decodes
// (dsh_sketch_fastx_batch_async), or the sequence the host parser extracts (compressed inputs, pipes, multi-line FASTQ; what is
// left of the region filled with 'N') -- and enqueued at once; the staging buffers take turns (see kStageBufs).  Cache
// hits (-W / sketch -c) read the .hll instead; .hll files of a batch are written when its buffer's turn comes again.
static void fill_sketches(CtxFuture &cf, const Opts &o, bool write_files, bool skip_cached, bool load_cached = false)
{
    const size_t n = o.inpaths.size();
    const size_t m = (size_t)1 << o.S;
    const size_t batch_bytes = kSketchBatchBytes;
    // staging: the FIRST batch goes to pageable memory (the HIP runtime is still coming up on the context thread, and
    // page-locking memory needs it); later batches cycle through the page-locked buffers that thread allocates meanwhile
    uint8_t *pin[kStageBufs] = {};
    size_t pin_cap[kStageBufs] = {};
    bool pin_taken[kStageBufs] = {};
    std::unique_ptr<uint8_t[]> first_stage;  // uninitialised: the parsing threads touch its pages first
    struct Flight {  // a batch whose copy + kernels are enqueued: what is left to do on the host when its buffer is reused
        bool valid = false;
        uint64_t ticket = 0;
        std::vector<size_t> slots;       // staged genome t -> its slot (= index into o.inpaths)
        std::vector<std::string> fnames;
        std::vector<char> raw;           // decoded on the device: its status word says whether the device took it
    } fl[kStageBufs];
    uint32_t *status[kStageBufs] = {};   // per buffer: the device decoder's verdict per staged genome (page-locked)
    size_t status_cap[kStageBufs] = {};
    auto sketch_on_host = [&](dsh_ctx *ctx, size_t slot) {  // a genome the staged path could not take: parsed growably, sketched on its own
        std::vector<uint8_t> whole;
        for (const auto &f2 : split_genome_paths(o.inpaths[slot])) {
            if (!whole.empty()) whole.push_back('N');
            if (append_fastx(f2, whole) < 0) die("Could not open %s", f2.c_str());
        }
        whole.push_back('N');
        const uint64_t loff[2] = {0, whole.size()};
        DSH(ctx, dsh_sketch_batch(ctx, whole.data(), loff, 1, slot, o.k, o.canon, nullptr));  // (max-merged into its slot)
    };
    auto retire = [&](int x) {  // the batch that last used buffer x has been sketched: refusals, .hll files
        Flight &f = fl[x];
        if (!f.valid) return;
        dsh_ctx *ctx = cf.get();
        DSH(ctx, dsh_event_wait(ctx, f.ticket));
        for (size_t t = 0; t < f.slots.size(); ++t)
            if (f.raw[t] && status[x][t]) sketch_on_host(ctx, f.slots[t]);  // not what its first byte promised (a '+' line in FASTA, multi-line FASTQ ...)
        if (write_files && !f.slots.empty()) {
            std::vector<uint8_t> rows(f.slots.size() * m);
            size_t r0 = 0;
            while (r0 < f.slots.size()) {  // consecutive slots in one download
                size_t r1 = r0 + 1;
                while (r1 < f.slots.size() && f.slots[r1] == f.slots[r1 - 1] + 1) ++r1;
                DSH(ctx, dsh_download_sketches(ctx, f.slots[r0], r1 - r0, rows.data() + r0 * m));
                r0 = r1;
            }
#pragma omp parallel for schedule(dynamic) num_threads(o.nthreads)
            for (long t = 0; t < (long)f.slots.size(); ++t)
                if (write_hll(f.fnames[t], rows.data() + (size_t)t * m, o.S, o.estim, o.estim, false, 0.0)) die("Could not write %s", f.fnames[t].c_str());
        }
        f.valid = false;
    };
    size_t g = 0, bi = 0;
    while (g < n) {
        const double t_b0 = now_s();
        const int x = (int)(bi % kStageBufs);
        retire(x);
        // the batch: genomes [g, e) whose files total <= batch_bytes (at least one)
        // (the FIRST batch is small: it travels from pageable memory, a synchronous staged copy, while the page-locked buffers
        // are still being allocated)
        const size_t limit = (bi == 0 && cf.stage_cap) ? std::min<size_t>(batch_bytes, (size_t)8 << 20) : batch_bytes;  // (an input of one batch stays one)
        size_t e = g, bytes = 0;
        while (e < n && (e == g || bytes + genome_file_size(o.inpaths[e]) <= limit)) bytes += genome_file_size(o.inpaths[e++]);
        const size_t nb = e - g;
        std::vector<int> cached(nb, 0), gz(nb, 0);
        std::vector<int> rawkind(nb, 0);        // plain FASTA whose raw bytes are staged: decoded on the device
        std::vector<uint64_t> rawlen_of(nb, 0);
        std::vector<std::string> fnames(nb);
        std::vector<std::vector<std::string>> files(nb);
        std::vector<std::vector<uint8_t>> zseq(nb);  // genomes whose sequence length the file size does not bound: compressed
                                                     // files, FIFOs, /dev/stdin, process substitutions (parsed growably)
        std::vector<size_t> late;  // slots of plain files that outgrew their region: sketched on their own after the batch
#pragma omp parallel for schedule(dynamic) num_threads(o.nthreads)
        for (long i = 0; i < (long)nb; ++i) {
            const std::string &entry = o.inpaths[g + i];
            fnames[i] = make_fname(entry, (unsigned)o.S, o.k, o.spacing, o.suffix, o.prefix);
            if ((o.cache || skip_cached) && isfile(fnames[i])) {
                cached[i] = 1;
                continue;
            }
            files[i] = split_genome_paths(entry);
            for (const auto &f : files[i]) gz[i] |= is_gzip_file(f) ? 1 : 0;
            if (gz[i])
                for (const auto &f : files[i]) {
                    if (!zseq[i].empty()) zseq[i].push_back('N');
                    if (append_fastx(f, zseq[i]) < 0) die("Could not open %s", f.c_str());
                }
        }
        // regions of the staging buffer: file bytes (an upper bound of the sequence bytes) + one separator per
        // file, rounded up to 32
        std::vector<uint64_t> off;
        std::vector<size_t> slot_of, src_of;
        uint64_t tot = 0;
        for (size_t i = 0; i < nb; ++i) {
            if (cached[i]) continue;
            const uint64_t need = gz[i] ? zseq[i].size() + 1 : genome_file_size(o.inpaths[g + i]) + files[i].size() + 1;
            off.push_back(tot);
            tot += (need + 31) & ~(uint64_t)31;
            slot_of.push_back(g + i);
            src_of.push_back(i);
        }
        off.push_back(tot);
        uint8_t *buf = nullptr;
        if (bi == 0) {
            first_stage.reset(new uint8_t[std::max<uint64_t>(tot, 1)]);
            buf = first_stage.get();
        } else {
            if (!pin_taken[x]) {  // adopt the buffer the context thread prepared
                pin[x] = cf.take_stage(x);
                pin_cap[x] = pin[x] ? cf.stage_cap : 0;
                pin_taken[x] = true;
            }
            if (tot > pin_cap[x]) {  // (this buffer's previous batch was retired above)
                if (pin[x]) dsh_free_host(pin[x]);
                pin_cap[x] = tot + (tot >> 4);
                (void)cf.get();
                if (!(pin[x] = (uint8_t *)dsh_alloc_host(pin_cap[x]))) die("could not allocate %zu bytes of pinned host memory", pin_cap[x]);
            }
            buf = pin[x];
        }
        const double t_alloc = now_s();
#pragma omp parallel for schedule(dynamic) num_threads(o.nthreads)
        for (long t = 0; t < (long)slot_of.size(); ++t) {
            const size_t i = src_of[t];
            uint8_t *dst = buf + off[t];
            const size_t cap = (size_t)(off[t + 1] - off[t]);
            size_t len = 0;
            bool raw_ok = !gz[i] && g_device_parse;
            if (raw_ok) {
                // the raw bytes of the genome's files, a '\n' between two files unless the first ends with one (FASTQ counts
                // lines); every file must begin with '>' -- FASTA -- or every file with '@' -- FASTQ in four-line records.
                // What is neither is the host parser's (below), and so is what the device refuses later: a '+' line in a
                // FASTA file, a FASTQ file that does not keep to four lines per record (status != 0: re-parsed when the
                // batch is retired).
                uint8_t kind = 0;
                for (const auto &f : files[i]) {
                    if (len && dst[len - 1] != '\n') dst[len++] = '\n';
                    const size_t at = len;
                    const long rc = read_raw_into(f, dst, cap, len);
                    if (rc == -1) die("Could not open %s", f.c_str());
                    if (len > at && !kind) kind = dst[at];
                    if (rc == -2 || (len > at && (dst[at] != kind || (kind != '>' && kind != '@')))) {
                        raw_ok = false;
                        break;
                    }
                }
                if (raw_ok) {
                    rawkind[i] = 1;
                    rawlen_of[i] = len;
                    continue;  // (no fill: the device pads the DECODED region; raw bytes behind rawlen are never looked at)
                }
                len = 0;
            }
            if (gz[i]) {
                if (!zseq[i].empty()) std::memcpy(dst, zseq[i].data(), zseq[i].size());
                len = zseq[i].size();
                std::vector<uint8_t>().swap(zseq[i]);
            } else {
                for (const auto &f : files[i]) {
                    if (len) dst[len++] = 'N';
                    const long rc = append_fastx_into(f, dst, cap, len);
                    if (rc == -1) die("Could not open %s", f.c_str());
                    if (rc == -2) {  // the file grew after its region was sized: the whole genome is read growably and sketched
                                     // on its own after the batch (what is staged is a prefix of it: harmless)
#pragma omp critical
                        late.push_back(slot_of[t]);
                        break;
                    }
                }
            }
            std::memset(dst + len, 'N', cap - len);  // invalid bases close every span: no k-mer, harmless
        }
        const double t_parsed = now_s();
        dsh_ctx *ctx = cf.get();
        const double t_ctx = now_s();
        for (size_t i = 0; i < nb; ++i) {
            if (!cached[i] || (skip_cached && !load_cached)) continue;  // `sketch -c`: nothing to do for a cached genome
            int p = 0;
            std::vector<uint8_t> r;
            if (read_hll(fnames[i], r, p) || p != o.S) die("Bad cached sketch %s (expected p=%d)", fnames[i].c_str(), o.S);
            DSH(ctx, dsh_upload_sketches(ctx, r.data(), g + i, 1));
        }
        if (slot_of.size() > status_cap[x]) {
            if (status[x]) dsh_free_host(status[x]);
            status_cap[x] = slot_of.size() + slot_of.size() / 2 + 64;
            if (!(status[x] = (uint32_t *)dsh_alloc_host(status_cap[x] * sizeof(uint32_t)))) die("could not allocate pinned host memory");
        }
        if (!slot_of.empty()) std::memset(status[x], 0, slot_of.size() * sizeof(uint32_t));
        std::vector<uint64_t> rlen(slot_of.size(), 0);
        for (size_t t = 0; t < slot_of.size(); ++t) rlen[t] = rawlen_of[src_of[t]];
        size_t r0 = 0;
        while (r0 < slot_of.size()) {  // consecutive runs of slots of one kind (raw FASTA / parsed sequence) go in one call each
            const int kind = rawkind[src_of[r0]];
            size_t r1 = r0 + 1;
            while (r1 < slot_of.size() && slot_of[r1] == slot_of[r1 - 1] + 1 && rawkind[src_of[r1]] == kind) ++r1;
            if (kind)
                DSH(ctx, dsh_sketch_fastx_batch_async(ctx, buf, off.data() + r0, rlen.data() + r0, (uint32_t)(r1 - r0), slot_of[r0], o.k, o.canon,
                                                      status[x] + r0));
            else
                DSH(ctx, dsh_sketch_batch_async(ctx, buf, off.data() + r0, (uint32_t)(r1 - r0), slot_of[r0], o.k, o.canon));
            r0 = r1;
        }
        for (size_t slot : late) sketch_on_host(ctx, slot);
        Flight &f = fl[x];
        f.slots = slot_of;
        f.fnames.clear();
        f.raw.clear();
        for (size_t t = 0; t < slot_of.size(); ++t) {
            f.fnames.push_back(fnames[src_of[t]]);
            f.raw.push_back((char)rawkind[src_of[t]]);
        }
        DSH(ctx, dsh_event_record(ctx, &f.ticket));
        f.valid = true;
        const double t_enq = now_s();
        if (bi == 0) retire(x);  // (the pageable first buffer is not kept: its copy was synchronous anyway)
        if (g_timing && (bi < 4 || bi % 32 == 0))
            std::fprintf(stderr, "[timing] batch %zu: %zu genomes, %.1f MB staged: retire + pinned alloc %.3f s, stage %.3f s, wait for the context %.3f s, enqueue %.3f s, first-batch retire %.3f s\n",
                         bi, nb, tot / 1e6, t_alloc - t_b0, t_parsed - t_alloc, t_ctx - t_parsed, t_enq - t_ctx, now_s() - t_enq);
        g = e;
        ++bi;
    }
    for (size_t b = bi >= (size_t)kStageBufs ? bi - kStageBufs : 0; b < bi; ++b) retire((int)(b % kStageBufs));  // oldest first
    dsh_ctx *ctx = cf.get();
    DSH(ctx, dsh_wait(ctx));
    for (auto &b : pin)
        if (b) dsh_free_host(b);  // (buffers never adopted are freed by the CtxFuture)
    for (auto &s2 : status)
        if (s2) dsh_free_host(s2);
}

// The end of a subcommand whose outputs are all written and closed.  Tearing the context and the HIP runtime down costs
// 0.1 s (profiles/rd6e/cli_e2e_timing.jsonl: a quarter of `dist` over 1 000 x 5 Mbp), and the kernel driver reclaims
// the device memory, queues and page-locked pages of an exiting process anyway: flush what stdio still holds and leave
// without the destructors.  DSH_FULL_TEARDOWN=1 runs them (leak checkers).
[[maybe_unused]] static void leave(dsh_ctx *ctx)
{
    since_launch("outputs closed");
    std::fflush(nullptr);
    if (std::getenv("DSH_FULL_TEARDOWN")) {
        dsh_destroy(ctx);
        return;
    }
    if (ctx) (void)dsh_synchronize(ctx);
    since_launch("leaving");
    std::fflush(nullptr);
    std::_Exit(EXIT_SUCCESS);
}

static int sketch_main(int argc, char **argv)
{
    Opts o = parse(argc, argv, false);
    if (!o.avoid_sorting) sort_paths_by_fsize(o.inpaths);  // src/dashing.cpp:356-357
    CtxFuture cf(o.device, o.inpaths.size(), o.S, staging_bytes_for(o.inpaths));
    const std::string &output_file = o.out_sizes;  // `sketch -o FILE` (src/dashing.cpp:307-337)
    dsh_ctx *ctx = nullptr;
    if (output_file.empty()) {
        fill_sketches(cf, o, /*write_files=*/true, /*skip_cached=*/o.skip_cached != 0);
        ctx = cf.get();
    } else {
        // all sketches into ONE gz stream + "<FILE>.labels.gz" (src/sketch_and_cmp.h:466-475,529-536);
        // with -c an existing per-genome .hll is read instead of re-sketched (:504-507)
        if (write_labels_gz(output_file + ".labels.gz", o.inpaths)) die("Failed to write sequence labels to file");
        fill_sketches(cf, o, /*write_files=*/false, /*skip_cached=*/o.skip_cached != 0, /*load_cached=*/true);
        ctx = cf.get();
        const size_t n = o.inpaths.size(), m = (size_t)1 << o.S;
        std::vector<uint8_t> all(n * m);
        DSH(ctx, dsh_download_sketches(ctx, 0, n, all.data()));
        if (write_hll_multi(output_file, all.data(), n, o.S, o.estim)) die("Failed to write sketches to file");
    }
    leave(ctx);
    return EXIT_SUCCESS;
}

// ---- utilities on .hll files ----------------------------------------------
// `union` (src/union.cpp:60-107): register-wise maximum of HLL sketch files -> one .hll.
static int union_main(int argc, char **argv)
{
    int level = 6, nthreads = 1;
    std::string opath = "/dev/stdout";
    std::vector<std::string> paths;
    optind = 1;
    for (int c; (c = getopt(argc, argv, "p:o:F:zZ:rHbh?")) >= 0;) {
        switch (c) {
        case 'Z': level = std::atoi(optarg); break;  // (the reference falls through into -o here: a bug we do not keep)
        case 'z': break;                             // output is a gz stream unless -Z 0
        case 'o': opath = optarg; break;
        case 'F': paths = read_paths_file(optarg); break;
        case 'p': nthreads = std::max(1, std::atoi(optarg)); break;
        case 'r': case 'H': case 'b': die("union: only HLL sketches are in scope");
        default:
            std::fprintf(stderr, "Usage: dashing-amd union [-o out.hll] [-F paths.txt] [-p threads] [-Z level] a.hll b.hll ...\n");
            return EXIT_FAILURE;
        }
    }
    for (int i = optind; i < argc; ++i) paths.emplace_back(argv[i]);
    if (paths.empty()) die("require >= 1 paths. See usage.");
    std::vector<std::vector<uint8_t>> acc((size_t)nthreads);
    std::vector<int> ps((size_t)nthreads, -1);
#pragma omp parallel for schedule(dynamic) num_threads(nthreads)
    for (long i = 0; i < (long)paths.size(); ++i) {
        const int t = omp_get_thread_num();
        int p = 0;
        std::vector<uint8_t> r;
        if (read_hll(paths[i], r, p)) die("Could not read sketch %s", paths[i].c_str());
        if (ps[t] < 0) {
            ps[t] = p;
            acc[t].swap(r);
        } else {
            if (p != ps[t]) die("For operator +=: np_ (%d) != other.get_np() (%d)", ps[t], p);
            union_registers(acc[t].data(), r.data(), r.size());
        }
    }
    int p = -1;
    std::vector<uint8_t> *tot = nullptr;
    for (int t = 0; t < nthreads; ++t) {
        if (ps[t] < 0) continue;
        if (!tot) {
            tot = &acc[t];
            p = ps[t];
        } else {
            if (ps[t] != p) die("For operator +=: np_ (%d) != other.get_np() (%d)", p, ps[t]);
            union_registers(tot->data(), acc[t].data(), tot->size());
        }
    }
    if (write_hll(opath, tot->data(), p, ERTL_MLE, ERTL_MLE, false, 0.0, level)) die("Could not open file at %s", opath.c_str());
    return EXIT_SUCCESS;
}

// `view` (src/dashing.cpp:553-557)
static int view_main(int argc, char **argv)
{
    if (argc < 2) die("Usage: dashing-amd view f1.hll [f2.hll ...]. Only HLLs currently supported.");
    for (int i = 1; i < argc; ++i) {
        int p = 0;
        std::vector<uint8_t> r;
        if (read_hll(argv[i], r, p)) die("Could not read sketch %s", argv[i]);
        print_hll(stdout, argv[i], r.data(), p);
    }
    return EXIT_SUCCESS;
}

// `fold` (src/dashing.cpp:558-590): compress a sketch to a smaller precision
static int fold_main(int argc, char **argv)
{
    std::string out = "/dev/stdout", in = "/dev/stdin";
    int destp = -1;
    optind = 1;
    for (int c; (c = getopt(argc, argv, "p:o:h?")) >= 0;) {
        switch (c) {
        case 'o': out = optarg; break;
        case 'p': destp = std::atoi(optarg); break;
        default:
            std::fprintf(stderr, "Usage: dashing-amd fold <flags> [in1.hll]\n-o: Write to <path> instead of stdout\n"
                                 "-p: set destination p [must be smaller than the input sketch\n");
            return EXIT_FAILURE;
        }
    }
    if (argc - optind == 1) in = argv[optind];
    else if (argc - optind > 1) die("Usage: dashing-amd fold <flags> [in1.hll]");
    if (out == "-") out = "/dev/stdout";
    if (in == "-") in = "/dev/stdin";
    int p = 0;
    std::vector<uint8_t> r, f;
    if (read_hll(in, r, p)) die("Could not read sketch %s", in.c_str());
    if (destp <= 0) destp = p - 1;
    if (destp >= p || destp < 4) die("fold: destination p (%d) must be in [4, %d)", destp, p);
    fold_registers(r.data(), p, destp, f);
    if (write_hll(out, f.data(), destp, ERTL_MLE, ERTL_MLE, false, 0.0)) die("Could not write %s", out.c_str());
    return EXIT_SUCCESS;
}

// `printmat` (src/dashing.cpp:425-452): a BINARY distance matrix ('\0', u64 n, packed upper triangle)
// as the full n x n table of DistanceMatrix::printf (distmat/distmat.h:358-381): "%lf" values, tabs
// between, the diagonal is the default value 0.
static int printmat_main(int argc, char **argv)
{
    bool use_scientific = false;
    std::string outpath = "/dev/stdout";
    optind = 1;
    for (int c; (c = getopt(argc, argv, ":o:sh?")) >= 0;) {
        switch (c) {
        case 'o': outpath = optarg; break;
        case 's': use_scientific = true; break;
        default:
            std::fprintf(stderr, "dashing-amd printmat <path to binary file>\n-o\tSpecify output file (default: stdout)\n-s\tEmit in scientific notation\n");
            return EXIT_FAILURE;
        }
    }
    if (optind >= argc) die("dashing-amd printmat <path to binary file>");
    std::FILE *in = std::fopen(std::strcmp(argv[optind], "-") ? argv[optind] : "/dev/stdin", "rb");
    if (!in) die("[print_binary_main] Could not open file at %s", argv[optind]);
    uint64_t n = 0;
    const int magic = std::fgetc(in);
    if (magic != 0 || std::fread(&n, sizeof n, 1, in) != 1) die("%s is not a float distance matrix", argv[optind]);
    const uint64_t total = n ? n * (n - 1) / 2 : 0;
    std::vector<float> tri(std::max<uint64_t>(total, 1));
    if (total && std::fread(tri.data(), sizeof(float), total, in) != total) die("size %llu is not the file size", (unsigned long long)(total * 4 + 9));
    std::fclose(in);
    std::FILE *fp = std::fopen(outpath.c_str(), "wb");
    if (!fp) die("[print_binary_main] Could not open file at %s", outpath.c_str());
    const char *mid = use_scientific ? "%le\t" : "%lf\t", *last = use_scientific ? "%le\n" : "%lf\n";
    for (uint64_t i = 0; i < n; ++i)
        for (uint64_t j = 0; j < n; ++j) {
            double v = 0.;
            if (i != j) {
                const uint64_t a = std::min(i, j), b = std::max(i, j);
                v = (double)tri[dsh_tri_index(n, a, b)];
            }
            std::fprintf(fp, j + 1 == n ? last : mid, v);
        }
    std::fclose(fp);
    return EXIT_SUCCESS;
}

// `hll` (src/hllmain.cpp:4-45): cardinality of the union of the k-mers of all input files --
// every file is sketched into ONE HLL (default p = 24: registers live in HBM, not LDS).
static int hll_main(int argc, char **argv)
{
    int k = 31, S = 24, nthreads = 1, canon = 1, device = 0;
    std::string paths_file;
    if (argc < 2) {
    usage:
        std::fprintf(stderr, "Usage: dashing-amd hll <opts> <paths>\nFlags:\n-k:\tkmer length (Default: 31. Max: 32)\n"
                             "-S:\tsketch size (default: 24). (2^S one-byte registers)\n-p:\tnumber of host threads.\n"
                             "-F:\tPath to file which contains one path per line\n-C:\tdo not canonicalise\n-d:\tGPU ordinal\n");
        return EXIT_FAILURE;
    }
    optind = 1;
    for (int c; (c = getopt(argc, argv, "Cw:s:S:p:k:F:d:tfh?")) >= 0;) {
        switch (c) {
        case 'C': canon = 0; break;
        case 'k': k = std::atoi(optarg); break;
        case 'p': nthreads = std::max(1, std::atoi(optarg)); break;
        case 's': if (optarg && *optarg) die("spaced seeds are out of scope (HLL hot path only)"); break;
        case 'w': if (std::atoi(optarg) > k) die("minimizer windows are out of scope (HLL hot path only)"); break;
        case 'S': S = std::atoi(optarg); break;
        case 'F': paths_file = optarg; break;
        case 'd': device = std::atoi(optarg); break;
        case 't': case 'f': break;
        default: goto usage;
        }
    }
    if (k < 1 || k > 32) die("k must be in [1,32]");
    if (S < 4 || S > 24) die("-S must be in [4,24]");
    std::vector<std::string> inpaths = paths_file.empty() ? std::vector<std::string>(argv + optind, argv + argc) : read_paths_file(paths_file);
    if (inpaths.empty()) goto usage;
    dsh_ctx *ctx = nullptr;
    if (int rc = dsh_create(device, &ctx)) die("[dashing-amd] no usable gfx950 device (dsh_create = %d); there is no CPU fallback", rc);
    DSH(ctx, dsh_sketches_alloc(ctx, 1, S));
    std::fprintf(stderr, "Processing %zu paths with %i threads\n", inpaths.size(), nthreads);
    // batches of files, parsed on the host threads, all max-merged into slot 0
    const size_t batch_bytes = (size_t)512 << 20;
    for (size_t g = 0; g < inpaths.size();) {
        size_t e = g, bytes = 0;
        while (e < inpaths.size() && (e == g || bytes + genome_file_size(inpaths[e]) <= batch_bytes)) bytes += genome_file_size(inpaths[e++]);
        std::vector<std::vector<uint8_t>> seqs(e - g);
#pragma omp parallel for schedule(dynamic) num_threads(nthreads)
        for (long i = 0; i < (long)(e - g); ++i)
            for (const auto &f : split_genome_paths(inpaths[g + i])) {
                if (!seqs[i].empty()) seqs[i].push_back('N');
                if (append_fastx(f, seqs[i]) < 0) die("Could not open %s", f.c_str());
            }
        std::vector<uint8_t> all;
        for (auto &sv : seqs) {
            all.insert(all.end(), sv.begin(), sv.end());
            all.push_back('N');
            std::vector<uint8_t>().swap(sv);
        }
        const uint64_t off[2] = {0, all.size()};
        DSH(ctx, dsh_sketch_batch(ctx, all.data(), off, 1, 0, k, canon, nullptr));
        g = e;
    }
    double est = 0;
    DSH(ctx, dsh_cardinalities(ctx, ERTL_MLE, &est));
    std::fprintf(stdout, "Estimated number of unique exact matches: %lf\n", est);
    dsh_destroy(ctx);
    return EXIT_SUCCESS;
}

static int dist_main(int argc, char **argv)
{
    Opts o = parse(argc, argv, true);
    std::FILE *ofp = stdout, *pairofp = stdout;
    if (!o.out_sizes.empty() && !(ofp = std::fopen(o.out_sizes.c_str(), "w"))) die("Could not open file at %s for writing.", o.out_sizes.c_str());
    if (!o.out_dists.empty() && !(pairofp = std::fopen(o.out_dists.c_str(), "wb"))) die("Could not open file at %s for writing.", o.out_dists.c_str());
    // asymmetric measure without -Q: all references are also the queries (src/distmain.cpp:120-125)
    const bool symmetric = !(o.result_type == 4 || o.result_type == 5 || o.result_type == 6);  // src/dashing.h:389-399
    if (o.querypaths.empty() && !symmetric) {
        o.querypaths = o.inpaths;
        std::fprintf(stderr, "Note: No query files provided, but an asymmetric distance was requested. Switching to a query/reference format with all references as queries.\n");
    }
    if (!o.presketched && !o.avoid_sorting) {  // src/distmain.cpp:126-129
        sort_paths_by_fsize(o.inpaths);
        sort_paths_by_fsize(o.querypaths);
    }
    const size_t nq = o.querypaths.size();
    for (auto &q : o.querypaths) o.inpaths.push_back(q);  // queries follow the references (src/distmain.cpp:130-133)
    const size_t n = o.inpaths.size();
    const double t_start = now_s();
    since_launch("dist: options parsed, context thread about to start");
    CtxFuture cf(o.device, n, o.S, o.presketched ? 0 : staging_bytes_for(o.inpaths));  // the HIP runtime comes up while the first batch is read
    dsh_ctx *ctx = nullptr;
    const double t_fill0 = now_s();
    if (o.presketched) {  // sketch.read(path), src/sketch_and_cmp.h:318-324
        ctx = cf.get();
        // read on all host threads into a staging matrix, upload in batches
        const size_t m = (size_t)1 << o.S, batch = std::max<size_t>(1, ((size_t)256 << 20) / m);
        std::vector<uint8_t> stage(std::min(n, batch) * m);
        for (size_t b0 = 0; b0 < n; b0 += batch) {
            const size_t b1 = std::min(n, b0 + batch);
#pragma omp parallel for schedule(dynamic, 16) num_threads(o.nthreads)
            for (int64_t i = (int64_t)b0; i < (int64_t)b1; ++i) {
                int p = 0;
                std::vector<uint8_t> r;
                if (read_hll(o.inpaths[i], r, p)) die("Could not read sketch %s", o.inpaths[i].c_str());
                if (p != o.S) die("Sketch %s has p=%d but -S is %d", o.inpaths[i].c_str(), p, o.S);
                std::memcpy(stage.data() + (size_t)(i - (int64_t)b0) * m, r.data(), m);
            }
            DSH(ctx, dsh_upload_sketches(ctx, stage.data(), b0, b1 - b0));
        }
    } else {
        fill_sketches(cf, o, /*write_files=*/o.cache != 0, false);
        ctx = cf.get();
    }
    if (g_timing) std::fprintf(stderr, "[timing] all sketches resident after %.3f s\n", now_s() - t_fill0);
    // sizes (src/sketch_and_cmp.h:372-385)
    std::vector<double> card(std::max<size_t>(n, 1));
    DSH(ctx, dsh_cardinalities(ctx, o.estim, card.data()));
    emit_sizes(ofp, o.inpaths, card.data());
    if (ofp != stdout) std::fclose(ofp);
    // distances (dist_loop, src/sketch_and_cmp.h:785-880)
    const uint64_t total = n ? (uint64_t)n * (n - 1) / 2 : 0;
    if (o.nneighbors) {  // nndist_loop (src/sketch_and_cmp.h:712-783)
        const size_t nr = nq ? n - nq : n, npairs = nq ? nq : n;
        unsigned nn = o.nneighbors;
        const size_t possible = nq ? nr : (n ? n - 1 : 0);
        if (nn > possible) {
            std::fprintf(stderr, "Only reporting %zu rather than %u neighbors due to their being only that many sets.\n", possible, nn);
            nn = (unsigned)possible;
        }
        std::vector<uint32_t> idx(std::max<size_t>(npairs * nn, 1));
        std::vector<float> val(std::max<size_t>(npairs * nn, 1));
        if (nn) DSH(ctx, dsh_knn(ctx, o.estim, o.result_type, o.k, nq ? nr : 0, n, 0, nr, nn, idx.data(), val.data()));
        if (o.fmt == BINARY) {  // u32 n, u32 nn, then {float value, u32 id} pairs (validx_t, :605)
            uint32_t hdr[2] = {(uint32_t)n, nn};
            std::fwrite(hdr, sizeof(uint32_t), 2, pairofp);
            for (size_t t = 0; t < npairs * nn; ++t) {
                std::fwrite(&val[t], sizeof(float), 1, pairofp);
                std::fwrite(&idx[t], sizeof(uint32_t), 1, pairofp);
            }
        } else {  // rows in input order (the reference's row order depends on its thread schedule)
            std::fputs("#File\tNeighbor ID:distance\t...\n", pairofp);
            for (size_t i = 0; i < npairs; ++i) {
                std::string s(o.inpaths[i + (nq ? nr : 0)]);
                char num[64];
                for (unsigned j = 0; j < nn; ++j)
                    s.append(num, (size_t)std::snprintf(num, sizeof num, "\t%u:%g", idx[i * nn + j], (double)val[i * nn + j]));
                s += '\n';
                std::fwrite(s.data(), 1, s.size(), pairofp);
            }
        }
    } else if (nq) {  // partdist_loop (src/dashing.h:660-712): one row per query over all references
        if (nq >= n) die("Wrong number of query/references. (ip size: %zu, nq: %zu", n, nq);
        const size_t nr = n - nq;
        if (o.fmt == UPPER_TRIANGULAR) std::fprintf(pairofp, "%zu\n", n);  // src/sketch_and_cmp.h:394-396
        const size_t qblock = std::max<size_t>(1, ((size_t)64 << 20) / nr);
        std::vector<float> buf;
        for (size_t q0 = nr; q0 < n; q0 += qblock) {
            const size_t q1 = std::min(n, q0 + qblock);
            buf.resize((q1 - q0) * nr);
            DSH(ctx, dsh_dist_rect(ctx, o.estim, o.result_type, o.k, q0, q1, 0, nr, buf.data()));
            for (size_t qi = q0; qi < q1; ++qi) {
                const float *row = buf.data() + (qi - q0) * nr;
                if (o.fmt == BINARY) {
                    if (std::fwrite(row, sizeof(float), nr, pairofp) != nr) die("Error writing to binary file");
                } else {
                    std::string s(o.inpaths[qi]);
                    char num[64];
                    for (size_t j = 0; j < nr; ++j) s.append(num, (size_t)std::snprintf(num, sizeof num, "\t%g", (double)row[j]));
                    s += '\n';
                    std::fwrite(s.data(), 1, s.size(), pairofp);
                }
            }
        }
    } else if (o.fmt == BINARY && o.devices.size() > 1 && !o.out_dists.empty()) {
        // Several GPUs of one node: every device holds all sketches, device d computes the row range
        // dsh_balance_rows gives it (tile-aligned, equal tile counts) and writes its
        // contiguous span of the packed matrix straight into the output file.
        if (write_binary_header(pairofp, n)) die("Failure");
        std::fflush(pairofp);
        const int fd = ::fileno(pairofp);
        if (::ftruncate(fd, (off_t)(9 + total * sizeof(float)))) die("could not size %s", o.out_dists.c_str());
        const size_t G = o.devices.size(), m = (size_t)1 << o.S;
        std::vector<uint8_t> all(n * m);
        DSH(ctx, dsh_download_sketches(ctx, 0, n, all.data()));
        std::vector<uint64_t> bounds(G + 1);
        if (dsh_balance_rows(n, (uint32_t)G, bounds.data())) die("dsh_balance_rows failed");
        std::vector<std::thread> workers;
        for (size_t d = 0; d < G; ++d)
            workers.emplace_back([&, d]() {
                dsh_ctx *c = ctx;
                if (d > 0) {
                    if (int rc = dsh_create(o.devices[d], &c)) die("[dashing-amd] dsh_create(device %d) = %d", o.devices[d], rc);
                    DSH(c, dsh_sketches_alloc(c, n, o.S));
                    DSH(c, dsh_upload_sketches(c, all.data(), 0, n));
                }
                // as below: large row blocks (each one a range the library lays its planes out for), two
                // page-locked buffers, block b+1 enqueued before block b is written at its file offset
                const uint64_t block_vals = (uint64_t)256 << 20;
                std::vector<uint64_t> cuts{bounds[d]};
                while (cuts.back() < bounds[d + 1]) {
                    const uint64_t rb = cuts.back();
                    uint64_t re = rb + 1;
                    while (re < bounds[d + 1] && dsh_tri_span(n, rb, re + 1) <= block_vals) ++re;
                    cuts.push_back(re);
                }
                uint64_t cap = 1;
                for (size_t b = 0; b + 1 < cuts.size(); ++b) cap = std::max<uint64_t>(cap, dsh_tri_span(n, cuts[b], cuts[b + 1]));
                float *bufs[2] = {nullptr, nullptr};
                for (int w = 0; w < (cuts.size() > 2 ? 2 : 1); ++w)
                    if (!(bufs[w] = (float *)dsh_alloc_host(cap * sizeof(float)))) die("could not allocate pinned host memory");
                const size_t nblocks = cuts.size() - 1;
                std::vector<uint64_t> ticket(nblocks + 1, 0);
                if (nblocks) {
                    DSH(c, dsh_dist_rows_async(c, o.estim, o.result_type, o.k, cuts[0], cuts[1], bufs[0]));
                    DSH(c, dsh_event_record(c, &ticket[0]));
                }
                for (size_t b = 0; b < nblocks; ++b) {
                    if (b + 1 < nblocks) {
                        DSH(c, dsh_dist_rows_async(c, o.estim, o.result_type, o.k, cuts[b + 1], cuts[b + 2], bufs[(b + 1) & 1]));
                        DSH(c, dsh_event_record(c, &ticket[b + 1]));
                    }
                    DSH(c, dsh_event_wait(c, ticket[b]));
                    const uint64_t span = dsh_tri_span(n, cuts[b], cuts[b + 1]);
                    const off_t pos = (off_t)(9 + dsh_tri_span(n, 0, cuts[b]) * sizeof(float));
                    size_t done = 0;
                    while (done < span * sizeof(float)) {
                        const ssize_t w = ::pwrite(fd, (const char *)bufs[b & 1] + done, span * sizeof(float) - done, pos + (off_t)done);
                        if (w <= 0) die("Failed to write rows to disk");
                        done += (size_t)w;
                    }
                }
                for (auto &bf : bufs) dsh_free_host(bf);
                if (d > 0) dsh_destroy(c);
            });
        for (auto &w : workers) w.join();
    } else if (o.devices.size() > 1 || o.rccl) {
        // Several GPUs, output that one writer has to emit in order (text formats, or -b to a pipe): every device
        // computes its rows -- the library partitions them itself (dsh_dist_collect with bounds = NULL: a row range per
        // device plus top-up tile rows from the bottom of the triangle, dsh_balance_rowsets) -- and the rows are delivered
        // to the first device over RCCL / xGMI inside the library (pipelined grouped ncclSend/ncclRecv, part by part),
        // which hands the whole matrix to this process -- dashing's single writer (src/sketch_and_cmp.h:804-849) fed by
        // G GPUs.  One thread per device, as RCCL wants for one process.
        const size_t G = std::max<size_t>(o.devices.size(), 1), m = (size_t)1 << o.S;
        uint8_t uid[DSH_UNIQUE_ID_BYTES];
        if (int rc = dsh_comm_unique_id(uid)) die("[dashing-amd] RCCL is not available (dsh_comm_unique_id = %d)", rc);
        std::vector<uint8_t> all;
        if (G > 1) {
            all.resize(n * m);
            DSH(ctx, dsh_download_sketches(ctx, 0, n, all.data()));
        }
        std::vector<float> tri(std::max<uint64_t>(total, 1));
        std::vector<std::thread> workers;
        for (size_t d = 0; d < G; ++d)
            workers.emplace_back([&, d]() {
                dsh_ctx *c = ctx;
                if (d > 0) {
                    if (int rc = dsh_create(o.devices[d], &c)) die("[dashing-amd] dsh_create(device %d) = %d", o.devices[d], rc);
                    DSH(c, dsh_sketches_alloc(c, n, o.S));
                    DSH(c, dsh_upload_sketches(c, all.data(), 0, n));
                }
                DSH(c, dsh_comm_init(c, uid, (int)d, (int)G));
                DSH(c, dsh_dist_collect(c, o.estim, o.result_type, o.k, /*bounds=*/nullptr, 0, d == 0 ? tri.data() : nullptr));
                DSH(c, dsh_comm_destroy(c));
                if (d > 0) dsh_destroy(c);
            });
        for (auto &w : workers) w.join();
        if (o.fmt == BINARY) {
            if (write_binary_header(pairofp, n)) die("Failure");
            if (total && std::fwrite(tri.data(), sizeof(float), total, pairofp) != total) die("Failed to write rows to disk");
        } else if (o.fmt == FULL_TSV) {
            emit_full_header(pairofp, o.inpaths);
            for (size_t i = 0; i < n; ++i) emit_full_row(pairofp, o.inpaths, i, tri.data());
        } else {
            emit_header(pairofp, o.fmt, o.inpaths);
            for (size_t i = 0; i < n; ++i) emit_ut_row(pairofp, o.fmt, o.inpaths, i, tri.data() + dsh_tri_span(n, 0, i));
        }
    } else if (o.fmt == FULL_TSV) {
        std::vector<float> tri(std::max<uint64_t>(total, 1));
        DSH(ctx, dsh_dist_rows(ctx, o.estim, o.result_type, o.k, 0, n, tri.data()));
        emit_full_header(pairofp, o.inpaths);
        for (size_t i = 0; i < n; ++i) emit_full_row(pairofp, o.inpaths, i, tri.data());
    } else {
        if (o.fmt == BINARY) {
            if (write_binary_header(pairofp, n)) die("Failure");
        } else {
            emit_header(pairofp, o.fmt, o.inpaths);
        }
        // Row blocks, two page-locked ping-pong buffers, asynchronous C-ABI calls: block b+1 is enqueued
        // (compute + copy into its buffer) before block b is emitted, so the GPU works while this thread
        // formats / writes -- dist_loop's dps[i & 1] scheme (src/sketch_and_cmp.h:804-816) without the
        // writer thread.  Blocks are large (<= 256 Mi values) so that each one is a row range the library
        // lays its plane matrix out for (few planes per tile, values at their final positions).
        const uint64_t block_vals = (uint64_t)256 << 20;
        std::vector<uint64_t> cuts{0};
        while (cuts.back() < n) {
            const uint64_t rb = cuts.back();
            uint64_t re = rb + 1;
            while (re < n && dsh_tri_span(n, rb, re + 1) <= block_vals) ++re;
            cuts.push_back(re);
        }
        uint64_t cap = 1;
        for (size_t b = 0; b + 1 < cuts.size(); ++b) cap = std::max<uint64_t>(cap, dsh_tri_span(n, cuts[b], cuts[b + 1]));
        float *bufs[2] = {nullptr, nullptr};
        for (int w = 0; w < (cuts.size() > 2 ? 2 : 1); ++w)
            if (!(bufs[w] = (float *)dsh_alloc_host(cap * sizeof(float)))) die("could not allocate %llu bytes of pinned host memory", (unsigned long long)(cap * sizeof(float)));
        auto enqueue = [&](size_t b) {
            DSH(ctx, dsh_dist_rows_async(ctx, o.estim, o.result_type, o.k, cuts[b], cuts[b + 1], bufs[b & 1]));
        };
        const size_t nblocks = cuts.size() - 1;
        std::vector<uint64_t> ticket(nblocks + 1, 0);
        if (nblocks) {
            enqueue(0);
            DSH(ctx, dsh_event_record(ctx, &ticket[0]));
        }
        for (size_t b = 0; b < nblocks; ++b) {
            // block b+1 goes in first (its host buffer was emitted in the previous iteration): its kernels fill the
            // library's other device buffer while block b is still being copied out, and only block b is awaited
            if (b + 1 < nblocks) {
                enqueue(b + 1);
                DSH(ctx, dsh_event_record(ctx, &ticket[b + 1]));
            }
            DSH(ctx, dsh_event_wait(ctx, ticket[b]));  // block b has arrived in bufs[b & 1]
            const uint64_t rb = cuts[b], re = cuts[b + 1], span = dsh_tri_span(n, rb, re);
            const float *buf = bufs[b & 1];
            if (o.fmt == BINARY) {
                if (span && std::fwrite(buf, sizeof(float), span, pairofp) != span) die("Failed to write rows to disk");
                continue;
            }
            // format rows on all host threads (snprintf dominates text output), write in order
            std::vector<uint64_t> off(re - rb + 1, 0);
            for (uint64_t i = rb; i < re; ++i) off[i - rb + 1] = off[i - rb] + (n - i - 1);
            const uint64_t group = 64;
            for (uint64_t g0 = rb; g0 < re; g0 += group * (uint64_t)o.nthreads) {
                const uint64_t g1 = std::min<uint64_t>(re, g0 + group * (uint64_t)o.nthreads);
                std::vector<std::string> rows(g1 - g0);
#pragma omp parallel for schedule(dynamic, 4) num_threads(o.nthreads)
                for (int64_t i = (int64_t)g0; i < (int64_t)g1; ++i)
                    format_ut_row(rows[i - g0], o.fmt, o.inpaths, (size_t)i, buf + off[i - rb]);
                for (auto &r : rows) std::fwrite(r.data(), 1, r.size(), pairofp);
            }
        }
        for (auto &b : bufs) dsh_free_host(b);
    }
    std::fflush(pairofp);
    if (pairofp != stdout) std::fclose(pairofp);
    if (g_timing) std::fprintf(stderr, "[timing] total since dsh_create %.3f s\n", now_s() - t_start);
    if (o.fmt == BINARY && !o.nneighbors) {  // src/distmain.cpp:191-200 (emit_fmt == BINARY exactly)
        const std::string labels = o.out_dists.empty() ? "unspecified" : o.out_dists + ".labels";
        if (write_labels(labels, o.inpaths)) die("Could not open file at '%s' for writing", labels.c_str());
    }
    leave(ctx);
    return EXIT_SUCCESS;
}

int main(int argc, char **argv)
{
    // (Round 2 called setenv("OMP_WAIT_POLICY", "passive") here; libgomp reads its environment in a constructor that runs
    // before main(), so that never took effect -- ADVICE r2 -- and a re-exec with the variable set costs more process
    // start-up than the few milliseconds idle workers spin by default.  Set OMP_WAIT_POLICY=passive in the environment
    // if the host threads are needed elsewhere between the parallel regions.)
    since_launch("main() entered");
    if (argc < 2 || !std::strcmp(argv[1], "-h") || !std::strcmp(argv[1], "--help")) {
        std::fprintf(stderr, "%s\nUsage: dashing-amd <subcommand> [options...]\nSubcommands:\n  sketch\n  dist (also: cmp, setdist)\n  union | fold | view   (utilities on .hll files)\n  printmat              (binary distance matrix -> text)\n  hll                   (cardinality of the k-mers of a set of files)\n", kVersion);
        return EXIT_FAILURE;
    }
    const std::string sub(argv[1]);
    if (sub == "sketch") return sketch_main(argc - 1, argv + 1);
    if (sub == "dist" || sub == "cmp" || sub == "setdist") return dist_main(argc - 1, argv + 1);
    if (sub == "union") return union_main(argc - 1, argv + 1);
    if (sub == "view") return view_main(argc - 1, argv + 1);
    if (sub == "fold") return fold_main(argc - 1, argv + 1);
    if (sub == "hll") return hll_main(argc - 1, argv + 1);
    if (sub == "printmat") return printmat_main(argc - 1, argv + 1);
    if (sub == "version" || sub == "--version") {
        std::printf("%s\nbackend: %s\n", kVersion, dsh_backend_name());
        return EXIT_SUCCESS;
    }
    std::fprintf(stderr, "subcommand '%s' is outside the HLL sketch+dist hot path this build covers\n", sub.c_str());
    return EXIT_FAILURE;
}

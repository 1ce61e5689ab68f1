/*
 * dashing_hip.h -- C-ABI of libdashing_hip.so: dashing's HLL sketch-and-compare hot path on
 * MI355X (gfx950).  Plain pointers and sizes only; no C++/torch types cross this boundary.
 *
 * dashing (the reference, /root/reference) has no plugin/FFI interface for this path: it is
 * C++ templates instantiated per sketch type.  Each entry point below replaces one of the two
 * loop bodies ("waists") of the reference, cited file:line; INTEGRATION.md shows the
 * reference-side glue a maintainer would add.
 *
 * Conventions
 *   - every function returns DSH_OK (0) or a negative errno-style code; no exceptions cross
 *     the boundary; dsh_last_error(ctx) gives a human-readable message for the last failure.
 *   - the caller owns all host memory; the library owns device memory behind dsh_ctx.
 *   - one dsh_ctx per GPU; a ctx is not thread-safe, different ctxs are independent.
 *   - there is NO CPU fallback: with no gfx950 device dsh_create fails with DSH_ENODEV.
 *   - register arrays are dashing's: uint8_t[2^p] per sketch, row-major [n][2^p].
 *   - distances are float32 in the packed upper-triangular order of
 *     distmat/distmat.h:260-264: index(i,j) = i*(2n-i-1)/2 + j-(i+1), i<j.
 */
#ifndef DASHING_HIP_H_
#define DASHING_HIP_H_
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DSH_OK 0
#define DSH_EINVAL (-22)  /* bad argument */
#define DSH_ENOMEM (-12)  /* host or device allocation failed */
#define DSH_ENODEV (-19)  /* no usable gfx950 device */
#define DSH_EIO (-5)      /* HIP runtime error / file error */
#define DSH_ESTATE (-11)  /* call sequence error (e.g. dist before sketches are loaded) */

/* sketch::hll::EstimationMethod values selected by dist_main, src/distmain.cpp:37,59-62
 * (-E ORIGINAL, -I ERTL_IMPROVED, default/-m ERTL_MLE). */
#define DSH_ESTIM_ORIGINAL 0
#define DSH_ESTIM_ERTL_IMPROVED 1
#define DSH_ESTIM_ERTL_MLE 2

/* bns::EmissionType values, src/enums.h:13-23.  Symmetric measures handled by result_cmp's
 * first switch arm, src/dashing.h:571-576. */
#define DSH_MASH_DIST 0
#define DSH_JI 1
#define DSH_FULL_MASH_DIST 3
/* second arm of result_cmp (src/dashing.h:577-588), built on set_triple = full_set_comparison */
#define DSH_SIZES 2
#define DSH_FULL_CONTAINMENT_DIST 4
#define DSH_CONTAINMENT_INDEX 5
#define DSH_CONTAINMENT_DIST 6
#define DSH_SYMMETRIC_CONTAINMENT_INDEX 7
#define DSH_SYMMETRIC_CONTAINMENT_DIST 8

typedef struct dsh_ctx dsh_ctx;

/* ABI version: bumped whenever an entry point changes its signature or a table its layout (6: dsh_exchange_* take a
 * row-set table instead of bounds + world).  A host compiled against another DSH_ABI_VERSION links fine but would pass
 * shifted arguments: compare with dsh_abi_version() once at start-up.  (A bounds array handed to a function that now
 * parses a row-set table is refused, not over-read: its first word, 0, is not a valid world.) */
#define DSH_ABI_VERSION 6
int dsh_abi_version(void);

/* ---- context ---------------------------------------------------------------------------- */
const char *dsh_backend_name(void);       /* "hip:gfx950" */
int dsh_device_count(void);               /* number of visible HIP devices (0 if none) */
int dsh_create(int device, dsh_ctx **out);
void dsh_destroy(dsh_ctx *ctx);
const char *dsh_last_error(const dsh_ctx *ctx);
/* Load the kernels' code objects NOW (the HIP runtime otherwise loads each when one of its kernels is first launched:
 * 10-40 ms in the middle of the first sketch batch / the first dist call).  No context needed and safe to call from any
 * thread, also beside a thread that uses a context: a host that has something else to do while the runtime comes up (the
 * CLI stages its first batch) calls it there.  what: DSH_PRELOAD_SKETCH | DSH_PRELOAD_COMPARE. */
#define DSH_PRELOAD_SKETCH 1u
#define DSH_PRELOAD_COMPARE 2u
int dsh_preload(int device, unsigned what);
int dsh_synchronize(dsh_ctx *ctx);

/* ---- the resident sketch matrix ----------------------------------------------------------
 * Replaces `std::vector<hll_t> sketches` (src/sketch_and_cmp.h:282-288): n register arrays
 * of 2^p bytes, resident in HBM for the lifetime of the ctx (or until re-allocated).
 * p in [4,24] for sketching, cardinalities, up- and download (24 = the `hll` subcommand's default,
 * src/hllmain.cpp:5); the compare entry points take the same range (tuned for p <= 17; 20..24 work but are not a performance target). */
int dsh_sketches_alloc(dsh_ctx *ctx, uint64_t n, int p);
/* sketch.read(path) path (src/sketch_and_cmp.h:318-324, --presketched): host rows -> slots. */
int dsh_upload_sketches(dsh_ctx *ctx, const uint8_t *regs, uint64_t first_slot, uint64_t n);
int dsh_download_sketches(dsh_ctx *ctx, uint64_t first_slot, uint64_t n, uint8_t *regs_out);
/* Same, into a caller-owned DEVICE buffer (device-to-device on the ctx stream; returns when done).
 * Multi-GPU sketching: each rank sketches its share of the genomes, copies its rows
 * out with this call and the ranks all-gather the register arrays over RCCL. */
int dsh_copy_sketches_device(dsh_ctx *ctx, uint64_t first_slot, uint64_t n, void *d_regs_out);
/* Use a caller-owned DEVICE buffer [n][2^p] as the sketch matrix (no copy; the caller keeps it
 * alive).  This is how bench.py hands over inputs already resident in HBM. */
int dsh_attach_device_sketches(dsh_ctx *ctx, const void *d_regs, uint64_t n, int p);

/* ---- sketch waist -------------------------------------------------------------------------
 * Replaces the body of hot loop 1, `enc.for_each([&](u64 kmer){h.addh(kmer);}, file, ksp)`
 * (src/sketch_and_cmp.h:342 and :515; register rule mirrored at src/readfilt.cpp:86-88), for a
 * batch of genomes.  `seq` is host ASCII: genome g occupies seq[genome_off[g] .. genome_off[g+1]);
 * FASTA records inside a genome are separated by at least one non-ACGT byte (so k-mers never
 * span records, like kseq records); case is folded; any non-ACGT byte resets the window.
 * Canonical k-mers iff canon != 0 (-C clears it, src/distmain.cpp:65).  k in [1,32].
 * Registers go to slots [first_slot, first_slot+n_genomes) of the resident matrix (max-merged
 * into what is there, so a genome may be fed in several calls) and, if regs_out != NULL, are
 * also copied to the host.  Bit-exact with the CPU definition (max is order-independent). */
int dsh_sketch_batch(dsh_ctx *ctx, const uint8_t *seq, const uint64_t *genome_off,
                     uint32_t n_genomes, uint64_t first_slot, int k, int canon,
                     uint8_t *regs_out);
/* Asynchronous form: enqueues the host-to-device copy of `seq` (page-locked memory from dsh_alloc_host, else
 * the copy is synchronous) and the kernel, and returns; dsh_wait(ctx) completes it.  `seq` must stay untouched
 * until then.  A host that streams many genomes parses batch b+1 while batch b is copied and sketched. */
int dsh_sketch_batch_async(dsh_ctx *ctx, const uint8_t *seq_pinned, const uint64_t *genome_off,
                           uint32_t n_genomes, uint64_t first_slot, int k, int canon);
/* Same with `seq` already on the device (d_seq device pointer; genome_off stays on the host). */
int dsh_sketch_batch_device(dsh_ctx *ctx, const void *d_seq, const uint64_t *genome_off,
                            uint32_t n_genomes, uint64_t first_slot, int k, int canon);
/* The same with the PARSE on the device, as in the reference where Encoder::for_each(func, path) reads the records itself
 * (src/sketch_and_cmp.h:338-342): `raw` holds the bytes of plain FASTA files as they lie on disk -- genome g's at
 * raw[genome_off[g] .. genome_off[g] + raw_len[g]) (several files of one genome: back to back with a '\n' between them),
 * its region [genome_off[g], genome_off[g+1]) at least that long, every genome_off[g] a multiple of 32.  The library
 * copies the raw bytes to the device and decodes them there into what kseq would hand the encoder, then sketches as
 * dsh_sketch_batch_async does.  A genome that begins with '>' is FASTA: header lines ('>' or '@' first) end a record (one
 * invalid byte: k-mers never span records), '\n' and '\r' vanish, everything else is sequence (validated and case-folded
 * by the sketch kernel as above).  A genome that begins with '@' is FASTQ in four-line records: of every four lines the
 * second is sequence.  What does not keep its format's promise is REFUSED per genome, never guessed at -- a first byte
 * that is neither, a FASTA line that begins with '+', a FASTQ file whose lines 4r are not '@' headers or 4r + 2 not '+'
 * lines, in which a sequence line begins with '@', '>' or '+', or in which some record's quality line is not exactly as
 * long as its sequence line (multi-line records, cut-off files: the record state of kseq decides those; the length rule
 * is checked as a 64-bit fingerprint over all records) -- : status_out[g] != 0, NOTHING goes into its slot, and the host
 * parses that genome itself
 * (dsh_sketch_batch).  status_out: n_genomes words of page-locked host memory (or NULL), valid after dsh_wait.  `raw` must
 * stay untouched until then.  Compressed inputs and pipes are the host's business (inflate, then either entry point). */
int dsh_sketch_fastx_batch_async(dsh_ctx *ctx, const uint8_t *raw_pinned, const uint64_t *genome_off,
                                 const uint64_t *raw_len, uint32_t n_genomes, uint64_t first_slot, int k, int canon,
                                 uint32_t *status_out_pinned);
int dsh_clear_sketches(dsh_ctx *ctx, uint64_t first_slot, uint64_t n);

/* ---- cardinalities ------------------------------------------------------------------------
 * Replaces cardinality_estimate(hll_t&) = h.report() (src/dashing.h:492, used at
 * src/sketch_and_cmp.h:377-382): one double per sketch. */
int dsh_cardinalities(dsh_ctx *ctx, int estim, double *card_out);

/* ---- compare waist ------------------------------------------------------------------------
 * Replaces hot loop 2: perform_core_op (src/sketch_and_cmp.h:699-710) / the oracle(i,j) of
 * dm::parallel_fill (distmat/distmat.h:459-512) with func = result_cmp (src/dashing.h:568-592):
 * for rows i in [row_begin,row_end) and all j>i, out[index(i,j) - index(row_begin,row_begin+1)]
 * = float(result_cmp(sketch_j, sketch_i, result_type, 1/k)).  The rows of a range are one
 * contiguous span of the packed triangle, dsh_tri_span() elements long.
 * result_type: any bns::EmissionType above; k only matters for the *_DIST forms.  In every pair the
 * reference calls result_cmp(lhs = sketch_j, rhs = sketch_i). */
int dsh_dist_rows(dsh_ctx *ctx, int estim, int result_type, int k, uint64_t row_begin,
                  uint64_t row_end, float *out);
/* Same, result left in a caller-owned DEVICE buffer (no D2H).  The work runs on the ctx stream
 * (dsh_stream); the call returns after it has completed. */
int dsh_dist_rows_device(dsh_ctx *ctx, int estim, int result_type, int k, uint64_t row_begin,
                         uint64_t row_end, void *d_out);
/* Asynchronous forms -- the reference overlaps the comparison of one batch of rows with the emission of
 * the previous one through two ping-pong buffers (dist_loop's dps[i & 1] + std::async writer,
 * src/sketch_and_cmp.h:804-816; parallel_fill's writer thread, distmat/distmat.h:475-479,504-508).  These
 * calls ENQUEUE the whole computation on the ctx stream and return.  For the host form the result goes to one of
 * two device buffers taken in turn and is copied to `out` on a second (copy) stream, so the kernels of the next call
 * run while the previous result is still travelling to the host; a call only waits for the copy that last drained
 * the buffer it is about to fill.  Calls may be issued back to back (their kernels execute in order).
 * Completion: dsh_wait(ctx) blocks until everything enqueued on the ctx (both streams) has completed;
 * dsh_event_record / dsh_event_wait mark and await one point of the sequence without draining what was enqueued
 * after it.  `out` must stay valid until then and should come from dsh_alloc_host: the copy into pageable memory is
 * staged by the runtime and is not asynchronous.
 * Typical use (the CLI does this):  async(block 0); t0 = record;  loop b: async(block b+1 -> buf[(b+1)&1]);
 * t(b+1) = record; event_wait(t(b)); emit block b from buf[b&1].
 * The call itself may still block briefly at its start while a new column layout is built on the host
 * (only the first call after the sketches changed touches the device for that). */
int dsh_dist_rows_async(dsh_ctx *ctx, int estim, int result_type, int k, uint64_t row_begin,
                        uint64_t row_end, float *out_pinned);
int dsh_dist_rows_device_async(dsh_ctx *ctx, int estim, int result_type, int k, uint64_t row_begin,
                               uint64_t row_end, void *d_out);
int dsh_wait(dsh_ctx *ctx);
/* Per-call completion.  dsh_event_record: *ticket marks everything enqueued on the ctx so far (kernels, sketch
 * batches, and the host copies of dsh_dist_rows_async).  dsh_event_wait blocks the calling host thread until that
 * point has completed; work enqueued after the record keeps running.  dsh_event_query: *done = 1/0 without blocking.
 * Tickets are cheap (a ring of 64 event pairs; a ticket more than 64 records old counts as complete).  A ticket is one
 * event on each of the context's two streams and orders NOTHING between them: taken between a compute call and
 * dsh_collect_parts_async it does not hold the per-part transfers back behind the kernels. */
int dsh_event_record(dsh_ctx *ctx, uint64_t *ticket);
int dsh_event_wait(dsh_ctx *ctx, uint64_t ticket);
int dsh_event_query(dsh_ctx *ctx, uint64_t ticket, int *done);
/* Make all work enqueued on the ctx stream AFTER this call wait for `hip_event` (a hipEvent_t recorded by
 * the caller on its own stream, e.g. torch.cuda.Event.cuda_event after producing d_regs / a gathered
 * staging buffer) -- the device-side alternative to synchronising the host before a *_device call. */
int dsh_wait_event(dsh_ctx *ctx, void *hip_event);
/* Row ranges and layouts: for a range of at least "range_sort_min_rows" rows (option, default 1024; always
 * for the full triangle) the plane matrix is rebuilt for exactly that range -- the wanted rows first, then the
 * later rows, both in (threshold, min value) order, earlier rows left out -- so every tile is homogeneous
 * (few planes) and every value is written at its final packed position: any split of the rows into
 * ranges concatenates to the byte-identical matrix, at the speed of the full-triangle call.  Smaller
 * ranges use the identity layout, which stays cached between calls. */
/* Query x reference rectangle (partdist_loop, src/dashing.h:660-712): queries are slots
 * [q_begin,q_end), references slots [r_begin,r_end); out[(qi-q_begin)*(r_end-r_begin)+(rj-r_begin)]. */
int dsh_dist_rect(dsh_ctx *ctx, int estim, int result_type, int k, uint64_t q_begin,
                  uint64_t q_end, uint64_t r_begin, uint64_t r_end, float *out);

/* ---- k nearest neighbours -------------------------------------------------------------------
 * Replaces perform_nns / nndist_loop (src/sketch_and_cmp.h:642-783, --nearest-neighbors): for
 * every query slot in [q_begin,q_end) the nn best reference slots in [r_begin,r_end) under
 * result_type -- similarity measures best = largest, distances best = smallest (emt2nntype,
 * src/dashing.h:268-280) -- best first; a query is never its own neighbour.  Ties are broken by the
 * lower slot index (the reference's heap/thread order is unspecified).  idx_out/val_out: host
 * arrays [q_end-q_begin][nn]; missing neighbours (nn larger than the candidates) get idx 0xFFFFFFFF.
 * All-vs-all (nq == 0 in dashing): q = r = [0,n) -- every pair is computed ONCE.  Up to "knn_square_budget_bytes"
 * (option, default 96 GiB) both orientations go into an n x n float matrix in HBM and one selection pass per row
 * follows; beyond it (configs[4]: n x n would be 360 GB) the triangle is computed in bands of tile rows, every band
 * leaves its values as candidates of both sketches of each pair and is folded into the n running lists, so nothing of
 * size n x n exists (300 000 x p=14, nn=10: 14.7 s on one MI355X, one triangle pass).  nn > 1024 or q != r: blocks of
 * queries x all references. */
int dsh_knn(dsh_ctx *ctx, int estim, int result_type, int k, uint64_t q_begin, uint64_t q_end,
            uint64_t r_begin, uint64_t r_end, uint32_t nn, uint32_t *idx_out, float *val_out);

/* ---- multi-GPU shards of the full triangle ------------------------------------------------
 * Every rank holds all sketches (dsh_upload/attach) and computes one shard; no collective is
 * needed inside the compare.  Internally the plane matrix is laid out in (threshold, min value)
 * order so that tiles need few planes; shards are contiguous row ranges of THAT order, balanced
 * by cost, so a shard's result is one contiguous span of the packed triangle of the sorted order.
 *   dsh_shard_plan        span_off[0..nshards] = element offsets of the shards' spans (identical
 *                         on every rank: it depends only on the sketches)
 *   dsh_dist_shard_device compute shard `shard` into d_span (span_off[shard+1]-span_off[shard]
 *                         floats, device memory)
 *   dsh_unpermute_device  after the spans were gathered back to back (e.g. RCCL gather):
 *                         sorted-order packed triangle -> packed triangle in the original sketch
 *                         order (distmat/distmat.h:260-264), device to device. */
int dsh_shard_plan(dsh_ctx *ctx, int estim, uint32_t nshards, uint64_t *span_off);
int dsh_dist_shard_device(dsh_ctx *ctx, int estim, int result_type, int k, uint32_t shard,
                          uint32_t nshards, void *d_span);
int dsh_unpermute_device(dsh_ctx *ctx, const void *d_sorted_tri, void *d_out_tri);
/* Same, reading the spans where a gather of equal-sized (padded) blocks left them: shard r's span
 * starts at d_stage + r * stride (floats), stride >= the largest span.  Saves the copy that would lay
 * the spans back to back first.  Returns after completion. */
int dsh_unpermute_staged_device(dsh_ctx *ctx, const void *d_stage, uint64_t stride, uint32_t nshards,
                                void *d_out_tri);
/* General form: shard r's span starts at d_stage + block_off[r] (floats) -- any arrangement of the
 * gathered blocks, e.g. the per-piece blocks of a pipelined gather (several shards per rank, each
 * piece gathered while the next is computed). */
int dsh_unpermute_blocks_device(dsh_ctx *ctx, const void *d_stage, const uint64_t *block_off,
                                uint32_t nshards, void *d_out_tri);

/* ---- multi-GPU exchange over RCCL / xGMI ------------------------------------------------------
 * One dsh_ctx per GPU; the ranks may be processes (one per GPU, the bench) or threads of one process (the CLI).
 * The compare needs no collective -- every rank holds all sketches and computes a row range whose result is ONE
 * contiguous span of dashing's packed matrix (dsh_balance_rows) -- so the only exchange is the delivery of the spans
 * to the rank that emits the matrix, as in dashing where one process writes it (src/sketch_and_cmp.h:838-849), and,
 * when sketching is shared out, the all-gather of the register arrays.  RCCL is loaded on first use (librccl.so.1
 * next to the HIP runtime this library links to; DSH_RCCL_LIB overrides); all traffic is enqueued on the ctx stream,
 * so it is ordered with the kernels without any host synchronisation.
 *   dsh_comm_unique_id   rank 0: 128 bytes (an ncclUniqueId) to hand to every rank (file, pipe, MPI, shared memory ...)
 *   dsh_comm_init        collective (blocks until all `world` ranks called it with the same id)
 *   dsh_collect_spans    after rank r computed rows [bounds[r], bounds[r+1]) into d_local (device, its span): grouped
 *                        ncclSend/ncclRecv, one message per peer, every span received at its final place in d_final
 *                        (device, n(n-1)/2 floats, only read on `dst`; dst's own span is copied there unless d_local
 *                        already points at it).  The _async form returns after enqueueing (dsh_wait / a ticket).
 *   dsh_allgather_device ncclAllGather of equal-sized byte blocks (d_recv: world x bytes_per_rank), e.g. the register
 *                        arrays after sharded sketching (dsh_copy_sketches_device gives the block)
 *   dsh_dist_collect     the whole multi-GPU dist step for a host without device pointers: computes this rank's row
 *                        range, delivers the spans to `dst`, which gets the full packed matrix in `out` (host,
 *                        n(n-1)/2 floats; ignored on other ranks).  Without a communicator (world = 1) it is dsh_dist_rows.
 *                        bounds = NULL: the library partitions the rows itself (dsh_balance_rowsets) and runs the
 *                        pipelined exchange pair below.
 * Pipelined form (the exchange hidden behind the compute): dsh_dist_rows_parts_device_async computes a row range in
 * `nparts` consecutive parts (dsh_range_parts: about equal pair counts, cuts on whole 128-row tile rows; the plane
 * matrix keeps every part key-ordered on its own) and marks the completion of each part on the ctx stream;
 * dsh_collect_parts_async then enqueues, on the copy stream, one round of grouped ncclSend/ncclRecv per part, each
 * round waiting only for its own part -- part q travels over xGMI while part q+1 is computed.  Every rank calls both
 * with the same bounds / nparts; dsh_comm_wait (or dsh_wait / a ticket) completes them.  A range of any length works: a
 * call with parts always lays its range out in exactly the parts dsh_range_parts reports (a short range: one part),
 * and a rank without rows simply takes no part in the rounds.
 * ENVIRONMENT INPUTS of the library (all read by the exchange only):
 *   DSH_RCCL_LIB             path of the RCCL library to dlopen instead of librccl.so.1.  A CODE-LOADING TRUST BOUNDARY:
 *                            whatever it names runs inside the process that called dsh_comm_* (the tests load the
 *                            stand-in transport tests/mock_rccl through it).  A set-uid / privileged host must clear it.
 *                            If set and not loadable, dsh_comm_available reports DSH_ENODEV (never a silent fallback).
 *   DSH_COMM_INIT_TIMEOUT_S  seconds dsh_comm_init waits for ncclCommInitRank (default 90).  A timeout is FATAL for the
 *                            job: the helper thread stays inside RCCL, and should the peers arrive later the orphan
 *                            communicator is aborted, so that they fail too instead of waiting for this rank.
 *   DSH_COMM_TIMEOUT_S       seconds dsh_comm_wait and the blocking exchange calls wait (default 120)
 * Failure behaviour (a multi-rank job must end with an error, not hang):
 *   dsh_comm_available   DSH_OK if librccl can be loaded here (DSH_ENODEV otherwise) -- local, no communication: let
 *                        every rank check it and agree BEFORE the collective dsh_comm_init
 *   dsh_comm_library     the resolved path of the loaded librccl and its ncclGetVersion code (for logs)
 *   dsh_comm_init        gives up after DSH_COMM_INIT_TIMEOUT_S (default 90 s) when not every rank joins
 *   dsh_comm_wait        like dsh_wait, but with a deadline (DSH_COMM_TIMEOUT_S, default 120 s): when a peer never posts
 *                        its side of an exchange the communicator is aborted and DSH_EIO returned; the blocking
 *                        exchange calls (dsh_collect_spans, dsh_allgather_device, dsh_dist_collect) wait the same way */
#define DSH_UNIQUE_ID_BYTES 128
int dsh_range_parts(uint64_t n, uint64_t row_begin, uint64_t row_end, uint32_t nparts, uint64_t *part_rows, /* [nparts + 1] */
                    uint32_t *nparts_out);
int dsh_dist_rows_parts_device_async(dsh_ctx *ctx, int estim, int result_type, int k, uint64_t row_begin, uint64_t row_end,
                                     void *d_out, uint32_t nparts);
int dsh_collect_parts_async(dsh_ctx *ctx, uint64_t n, const uint64_t *bounds, uint32_t nparts, const void *d_local,
                            void *d_final, int dst);
/* The exchange-aware pair.  Parts of consecutive rows are key-ordered each on its own, so a SHORT range cut into many
 * parts loses the ordering its tiles live on (one 128-row block per part: 10 planes per tile instead of 8.7 at BASELINE
 * configs[2] over 8 ranks), and contiguous ranges on 128-row boundaries cannot give every rank the same number of tiles
 * when a range is only a few tile rows long (tile rows hold 79 ... 1 tiles there, a rank ~395).  So the partition is a
 * ROW-SET TABLE -- row segments with owners:
 *     tab[0] = world, tab[1] = nseg, tab[2 .. 2 + nseg] = the nseg + 1 segment boundaries from 0 to n,
 *     tab[3 + nseg .. 3 + 2 nseg) = the owning rank of every segment                       (3 + 2 nseg words)
 * A rank's rows are the segments it owns (adjacent ones merged): the first is its MAIN range, the others EXTRA segments
 * ("top-ups"); a rank with extra segments must have all its boundaries on multiples of 128 (or at n).
 *   dsh_balance_rowsets      main ranges over the top of the triangle + the short tile rows at its bottom dealt, in runs
 *                            of consecutive tile rows, to the ranks that fall short of the mean: the largest cost of any
 *                            rank (tiles + its own prepare, prep_permille/1000 tiles per 128 columns of its plane matrix;
 *                            < 0: the default) is smallest.  dst >= 0 names the rank that will RECEIVE the others' rows:
 *                            it sends nothing, so it takes dst_bonus_permille (< 0: the default -- 120 where a rank holds at least 16 tile rows, else 0) thousandths of a
 *                            rank's mean tile count more than the others, whose step only ends when their last part has
 *                            arrived; dst < 0: every rank keeps its rows.  Plain dsh_balance_rows ranges when n > 32 768
 *                            or a rank would hold fewer than two tile rows.  tab_out = NULL: only the size (words_out).
 *   dsh_rowsets_from_bounds  contiguous bounds[world + 1] as a table (3 + 2 world words): any alignment
 *   dsh_rowsets_rank         the segments {b0, e0, b1, e1, ...} of one rank, its pairs and its 128 x 128 tiles
 * The layout of every rank's buffer follows from (table, nparts, dst) alone:
 *   - the destination computes its own rows in place (d_local = d_final + the offset of its first row) as ONE part;
 *   - a rank with extra segments, or with a range of fewer than 1024 rows per part (span <= 1 GiB), goes ROW-SORTED: every
 *     segment key-ordered as one run (a range that reaches far down the triangle as two: its last rows on their own),
 *     d_local holds the rows in that order; the destination stages what it receives and puts the rows that are complete
 *     into place (one contiguous copy per row) behind every round, on a stream of its own;
 *   - longer ranges hold consecutive rows, received in place.
 * PARTS are units of completion (runs of whole tile rows, final in order; at most nparts -- a row-sorted rank whose parts
 * announce themselves from inside k_finalize cuts every tile row a part); what travels are MESSAGES: the exchange runs in
 * nparts ROUNDS, round q = one grouped ncclSend/ncclRecv of message q of every source, the q-th nparts-th of its buffer,
 * sent as soon as the part that holds its last value is final.  A round lasts as long as its largest message; with the
 * spans of dsh_balance_rowsets about equal no link waits for another's.
 * dsh_exchange_rows_device_async computes rank `rank`'s rows (enqueued; every part announces its completion: a flag written from
 * inside k_finalize, or an event between launches -- option finalize_signal), dsh_exchange_collect_async
 * enqueues the rounds on the copy stream; every rank calls both with the same arguments;
 * dsh_comm_wait completes them.  dsh_exchange_mode tells how a rank's buffer is laid out (rowsorted 0/1, parts at nparts) and how
 * many floats d_local must hold.  dsh_exchange_place_device does, for ONE source rank and without a communicator, what
 * the destination does with that rank's buffer (tests and single-GPU timing of an N-rank plan).
 * dsh_dist_collect(bounds = NULL) runs this pair over dsh_balance_rowsets' table. */
int dsh_balance_rowsets(uint64_t n, uint32_t world, int prep_permille, int dst, int dst_bonus_permille, uint64_t *tab_out,
                        uint32_t cap_words, uint32_t *words_out);
int dsh_rowsets_from_bounds(const uint64_t *bounds, uint32_t world, uint64_t *tab_out /* [3 + 2 world] */);
int dsh_rowsets_rank(uint64_t n, const uint64_t *rowsets, uint32_t rank, uint64_t *segs_out /* [2 cap_segs] or NULL */, uint32_t cap_segs,
                     uint32_t *nsegs_out, uint64_t *pairs_out, uint64_t *tiles_out);
int dsh_exchange_mode(uint64_t n, const uint64_t *rowsets, int rank, uint32_t nparts, int dst, int *rowsorted,
                      uint32_t *nparts_out, uint64_t *local_floats_out);
int dsh_exchange_rows_device_async(dsh_ctx *ctx, int estim, int result_type, int k, const uint64_t *rowsets, int rank,
                                   uint32_t nparts, int dst, void *d_local);
int dsh_exchange_collect_async(dsh_ctx *ctx, uint64_t n, const uint64_t *rowsets, uint32_t nparts, const void *d_local,
                               void *d_final, int dst);
int dsh_exchange_place_device(dsh_ctx *ctx, const uint64_t *rowsets, int src, uint32_t nparts, int dst,
                              const void *d_src_local, void *d_final);
/* Diagnostics of the exchange on ONE GPU (tests/test_gpu_multirank.py, tools/interference_probe.py):
 *   dsh_exchange_probe_parts_async  after dsh_exchange_rows_device_async with the same (table, rank, nparts, dst): enqueues
 *                        on the copy stream, for every part of the rank's call, the part's gate (the flag k_finalize
 *                        sets / the event) and behind it a copy KERNEL of the part's share of d_local into d_probe at the
 *                        same offsets -- plain loads through the L2s while k_finalize is still running, exactly what an
 *                        RCCL send kernel does with the part (the copy engine of a hipMemcpy reads memory instead).
 *                        dsh_wait / dsh_comm_wait completes it.  d_probe: as many floats as d_local.
 *                        On a context that holds a communicator of ONE rank (dsh_comm_init(.., 0, 1)) the reader is
 *                        librccl itself: the rank's buffer travels as dsh_exchange_collect_async would send it -- nparts
 *                        messages, each behind the gate of the part that holds its last value, one grouped call per
 *                        message -- by ncclSend to the rank itself paired with the ncclRecv into d_probe.
 *   dsh_diag_spin_start  occupies `nblocks` workgroups of `threads` lanes and `lds_bytes` of LDS each with a kernel that
 *                        polls a word of host memory -- what an RCCL receive kernel does while its peers have nothing to
 *                        send -- on a stream of its own, until dsh_diag_spin_stop or max_ms (<= 10 000) have passed:
 *                        measures what such a kernel costs the tile kernel beside it (option xch_recv_gate). */
int dsh_exchange_probe_parts_async(dsh_ctx *ctx, uint64_t n, const uint64_t *rowsets, int rank, uint32_t nparts, int dst,
                                   const void *d_local, void *d_probe);
int dsh_diag_spin_start(dsh_ctx *ctx, uint32_t nblocks, uint32_t threads, uint32_t lds_bytes, uint32_t max_ms);
int dsh_diag_spin_stop(dsh_ctx *ctx);
int dsh_comm_available(void);
int dsh_comm_library(char *path_out, size_t cap, int *version_out);
int dsh_comm_unique_id(void *id_out);
int dsh_comm_init(dsh_ctx *ctx, const void *unique_id, int rank, int world);
int dsh_comm_destroy(dsh_ctx *ctx);
int dsh_comm_rank(const dsh_ctx *ctx, int *rank, int *world);
int dsh_comm_wait(dsh_ctx *ctx);
int dsh_collect_spans(dsh_ctx *ctx, uint64_t n, const uint64_t *bounds, const void *d_local, void *d_final, int dst);
int dsh_collect_spans_async(dsh_ctx *ctx, uint64_t n, const uint64_t *bounds, const void *d_local, void *d_final, int dst);
int dsh_allgather_device(dsh_ctx *ctx, const void *d_send, uint64_t bytes_per_rank, void *d_recv);
int dsh_dist_collect(dsh_ctx *ctx, int estim, int result_type, int k, const uint64_t *bounds, int dst, float *out);

/* ---- helpers shared by every host (C++ CLI, Python, a patched dashing) -------------------- */
/* number of packed elements of rows [row_begin,row_end) of an n x n upper triangle */
uint64_t dsh_tri_span(uint64_t n, uint64_t row_begin, uint64_t row_end);
/* index(i,j) of distmat/distmat.h:260-264 */
uint64_t dsh_tri_index(uint64_t n, uint64_t i, uint64_t j);
/* Split rows [0,n) into nparts contiguous ranges of near-equal pair count, boundaries aligned
 * to `align` rows (the kernel tile, 128, keeps every rank on whole tile rows).
 * bounds_out[0..nparts] receives the boundaries (bounds_out[0]=0, bounds_out[nparts]=n). */
int dsh_partition_rows(uint64_t n, uint32_t nparts, uint32_t align, uint64_t *bounds_out);

/* Row ranges for the ranks of a multi-GPU run (or the devices of the CLI): bounds on 128-row boundaries that
 * minimise the largest number of 128 x 128 tiles any part computes (triangle of its rows + the rectangle
 * to their right).  Each range is then one dsh_dist_rows* call whose result is one contiguous span of the
 * final packed triangle -- the ranks' spans concatenate, nothing is re-ordered. */
int dsh_balance_rows(uint64_t n, uint32_t nparts, uint64_t *bounds_out);

/* Page-locked host memory for the host-buffer entry points (dsh_dist_rows, dsh_upload_sketches,
 * dsh_sketch_batch): with such buffers the copies are direct DMA at PCIe rate instead of going
 * through the runtime's staging of pageable memory.  Optional -- any host pointer works. */
void *dsh_alloc_host(size_t bytes);
void dsh_free_host(void *p);

/* ---- instrumentation ---------------------------------------------------------------------- */
/* Milliseconds spent in the dominant kernel (all-pairs AND+popcount) during the last
 * dsh_dist_* call on this ctx, measured with HIP events on the ctx stream; launches = number
 * of launches of that kernel in the call.  Enabled by dsh_set_profiling(ctx, 1) (adds event
 * records + one sync at the end of the call). */
int dsh_set_profiling(dsh_ctx *ctx, int enable);
int dsh_last_kernel_ms(dsh_ctx *ctx, double *pair_kernel_ms, double *finalize_kernel_ms,
                       double *prepare_ms, uint32_t *pair_kernel_launches);
/* (profiling on) the parts of the last call with parts (dsh_exchange_rows_device_async, dsh_dist_rows_parts_device_async):
 * when each became final, in ms from the start of the call (prepare included), and how many floats of the rank's buffer it
 * holds -- what a model of the pipelined exchange needs (tools/shard_model.py, bench.py --gpus N). */
int dsh_last_part_info(dsh_ctx *ctx, double *ready_ms /* [cap] or NULL */, uint64_t *floats /* [cap] or NULL */, uint32_t cap,
                       uint32_t *nparts_out);
/* Profiling aid: after a compare call with dsh_set_profiling(ctx, 1) and the option "finalize_timing" = 1 (the
 * s_memtime-stamped instance of k_finalize; results unchanged), out16 = shader-clock cycles summed over the waves that
 * finished, per phase [0..5] {prologue + loads issued, histogram columns, list joins, fix-ups, estimator, result + store};
 * [6] such waves; [7] their lanes; sums over lanes of [8] MLE iterations, [9] live bins, [10] iterations x bins; sums over
 * waves of the per-wave maxima [11] iterations, [12] bins, [13] their product (what a wave pays); [14..15] zero. */
int dsh_finalize_phase_cycles(dsh_ctx *ctx, uint64_t *out16);
/* Options; returns DSH_EINVAL for unknown names or values.  None changes a result (tests/test_gpu_compare.py asserts
 * byte-identical output over their ranges).  The tuning knobs of rounds 2-5 whose A/B was decided are gone together with
 * their losing arms (profiles/HISTORY.md has the measurements); what is left is what a caller or a first run on other
 * hardware needs:
 *   resources     "cum_budget_bytes"        scratch for the pair counts C(v) (default 8 GiB): larger jobs run in bands
 *                 "knn_square_budget_bytes" all-vs-all dsh_knn keeps an n x n float matrix in HBM up to this size (96 GiB)
 *   layout        "sort"                    -1 auto | 0 | 1: key-ordered plane columns (0 = identity: the slow, simple layout)
 *                 "range_sort_min_rows"     row ranges shorter than this keep the cached identity layout (default 1024)
 *                 "emax" / "elow"           caps of the listed upper / lower register tail, 0..255, -1 auto (per precision);
 *                                           0 / 0 = bit-planes over the whole value range (adversarial register laws)
 *   tile kernel   "kc"                      0 auto | 16 | 32 k-rows per LDS stage
 *                 "nsplit"                  pieces per tile, 0 auto (work items of at most 64 chunks)
 *                 "overflow_frag_permille"  0..1000 (default 500): a band whose one-plane work items number at most that
 *                                           share of a round above a multiple of 512 has the items left over cut into
 *                                           fragments that ADD their counts; 0 = never
 *   exchange      "part_band_tiles"         a part of at least this many tiles also ends a launch of the tile kernel (2048)
 *                 "xch_tail_bands"          0..8 (default 2): a job with parts of at most 64 rounds has its tile kernel cut
 *                                           at whole rounds into a head and this many tails, so that the head's parts
 *                                           travel while the tails compute
 *                 "xch_recv_gate"           -1 auto | 0 | 1: the destination posts its receives behind its first tile
 *                                           kernel instead of at once (a waiting receive kernel beside the tile kernel
 *                                           costs it 2-14 %, profiles/rd6a/interference_probe.jsonl); auto = for a job
 *                                           of one launch of at most 8 rounds, whose peers have nothing to send earlier
 *                 "finalize_signal"         -1 auto | 0 | 1: parts announce themselves from inside ONE k_finalize launch
 *                                           per band (flags + hipStreamWaitValue32) instead of one launch and one event
 *                                           per part; auto = where the device supports stream wait-value
 *   profiling     "finalize_timing"         the s_memtime-stamped instance of k_finalize (same results)
 *                 "finalize_stop"           1..4: k_finalize leaves after a phase and stores a dummy -- CHANGES results;
 *                                           accepted only while dsh_set_profiling is on, cleared when it is switched off
 * ("pair_mfma" exists only in a library built with `make WHATIF=1`: the matrix-core what-if the north star excludes.) */
int dsh_set_option(dsh_ctx *ctx, const char *name, int64_t value);
/* Derived state of the last prepared sketch matrix: "planes" (dense bit-planes used), "vlo",
 * "vhi", "pbase", "threshold", "emax", "elow", "kc", "tile", "npad", "kpad", "cum_bytes", "sorted", "ncols", "lockstep", "tiles", "bands", "items" (work items of the tile kernel),
 * "words_per_plane", "avg_tile_planes_x100" (of the last dist call), "frag_items", "parts_done", "parts_signalled",
 * "place_kernel_us" (with profiling on: device time of the last dsh_exchange_place_device's placement kernel),
 * "sketch_kernel_us" / "fastx_decode_us" (with profiling on: k_sketch / the FASTA-FASTQ decode kernels of the last sketch call). */
int dsh_get_info(dsh_ctx *ctx, const char *name, int64_t *out);
/* HIP stream of the ctx as a void* (hipStream_t) so a host framework can order its own work.
 * Every *_device entry point runs on THIS stream and (except the *_async forms) returns after its work has
 * completed, so results are ready for any other stream on return.  The other direction is the caller's
 * job: whatever it enqueued on its own streams that touches a buffer passed in (filling d_out, producing
 * d_regs or d_seq, an RCCL gather into a staging buffer) must have completed on the host -- or be ordered
 * before this stream's work with dsh_wait_event -- before the call. */
void *dsh_stream(dsh_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif /* DASHING_HIP_H_ */

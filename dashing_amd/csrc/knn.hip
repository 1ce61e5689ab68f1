// knn.hip -- dsh_knn: k nearest neighbours per sketch (perform_nns / nndist_loop, src/sketch_and_cmp.h:642-783).
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <new>

#include "ctx.h"

using namespace dsh;

extern "C" {

int dsh_knn(dsh_ctx *c, int estim, int result_type, int k, uint64_t qb, uint64_t qe, uint64_t rb,
            uint64_t re, uint32_t nn, uint32_t *idx_out, float *val_out)
{
    if (!c) return DSH_EINVAL;
    int rc = bind(c);
    if (rc) return rc;
    if (!c->have_sketches) return fail(c, DSH_ESTATE, "no sketches loaded");
    if (qe > c->n || re > c->n) return fail(c, DSH_EINVAL, "slots out of range");
    reset_prof(c);
    if (qb >= qe || nn == 0) return DSH_OK;
    if (!idx_out || !val_out) return DSH_EINVAL;
    // similarity measures rank descending, distances ascending (emt2nntype, src/dashing.h:268-280)
    const int descending = !(result_type == DSH_MASH_DIST || result_type == DSH_FULL_MASH_DIST ||
                             result_type == DSH_CONTAINMENT_DIST || result_type == DSH_FULL_CONTAINMENT_DIST ||
                             result_type == DSH_SYMMETRIC_CONTAINMENT_DIST);
    const uint64_t nq = qe - qb, nr = re > rb ? re - rb : 0;
    const bool overlap = qb < re && rb < qe;
    if (qb == 0 && rb == 0 && qe == c->n && re == c->n && c->n > 1 &&
        c->n * c->n * sizeof(float) <= c->knn_square_budget) {
        // all-vs-all: every pair is computed ONCE (triangle tiles, sorted columns) and written at
        // both (i,j) and (j,i) of an n x n matrix in HBM; then one selection pass per row
        const uint64_t n = c->n;
        DevBuf sq, didx, dval;
        rc = DSH_OK;
        do {
            if (sq.ensure(n * n * sizeof(float)) != hipSuccess || didx.ensure(n * nn * sizeof(uint32_t)) != hipSuccess ||
                dval.ensure(n * nn * sizeof(float)) != hipSuccess) {
                rc = fail(c, DSH_ENOMEM, "device allocation failed");
                break;
            }
            PairJob j;
            j.estim = estim;
            j.result_type = result_type;
            j.k = k;
            j.rect = 0;
            j.square = 1;
            j.ksinv_double = 1;
            j.row_begin = 0;
            j.row_end = n;
            j.col_begin = 0;
            j.col_end = n;
            j.base_index = 0;
            j.d_out = (float *)sq.ptr;
            if ((rc = run_pairs(c, j))) break;
            hipError_t e = launch_topk(c->stream, (const float *)sq.ptr, n, n, 0, 0, descending, nn, 1,
                                       (uint32_t *)didx.ptr, (float *)dval.ptr);
            if (e != hipSuccess) {
                rc = fail(c, DSH_EIO, "k_topk: %s", hipGetErrorString(e));
                break;
            }
            if (hipMemcpyAsync(idx_out, didx.ptr, n * nn * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
                hipMemcpyAsync(val_out, dval.ptr, n * nn * sizeof(float), hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
                hipStreamSynchronize(c->stream) != hipSuccess)
                rc = fail(c, DSH_EIO, "copy of neighbours failed");
        } while (0);
        sq.release();
        didx.release();
        dval.release();
        return rc;
    }
    if (qb == 0 && rb == 0 && qe == c->n && re == c->n && c->n > 1 && nn <= 1024) {
        // all-vs-all beyond the n x n budget: the triangle ONCE, in bands of tile rows of the key-ordered layout.  A band
        // leaves its values twice (V: band rows x columns, Vt: columns x band rows -- each pair is a candidate of both its
        // sketches) and two selection passes fold them into the running lists of the n sketches; nothing of size n x n
        // exists (nndist_loop, src/sketch_and_cmp.h:712-783, keeps n heaps the same way).
        const uint64_t n = c->n;
        if ((rc = prepare(c, estim, 1))) return rc;
        const uint64_t npad = c->lay.Npad;
        const uint64_t budget = std::max<uint64_t>(std::min<uint64_t>(c->knn_square_budget, (uint64_t)16 << 30), 2 * kTile * npad * sizeof(float));
        const uint64_t band = std::min<uint64_t>(npad, budget / (2 * npad * sizeof(float)) / kTile * kTile);
        DevBuf V, Vt, didx, dval;
        rc = DSH_OK;
        do {
            if (V.ensure(band * npad * sizeof(float)) != hipSuccess || Vt.ensure(npad * band * sizeof(float)) != hipSuccess ||
                didx.ensure(n * nn * sizeof(uint32_t)) != hipSuccess || dval.ensure(n * nn * sizeof(float)) != hipSuccess) {
                rc = fail(c, DSH_ENOMEM, "device allocation failed");
                break;
            }
            hipError_t e = launch_knn_state_init(c->stream, (uint32_t *)didx.ptr, (float *)dval.ptr, n * nn, descending);
            if (e != hipSuccess) {
                rc = fail(c, DSH_EIO, "k_fill_knn_state: %s", hipGetErrorString(e));
                break;
            }
            for (uint64_t b0 = 0; b0 < n && rc == DSH_OK; b0 += band) {
                const uint64_t b1 = std::min<uint64_t>(n, b0 + band);
                PairJob j;
                j.estim = estim;
                j.result_type = result_type;
                j.k = k;
                j.rect = 0;
                j.sorted_rows = 1;
                j.knn = 1;
                j.ksinv_double = 1;
                j.row_begin = b0;
                j.row_end = b1;
                j.col_begin = j.col_end = 0;
                j.base_index = 0;
                j.d_out = (float *)V.ptr;
                j.d_out2 = (float *)Vt.ptr;
                j.knn_ld = npad;
                j.knn_rows = band;
                if ((rc = run_pairs(c, j))) break;
                const uint32_t *perm = (const uint32_t *)c->perm.ptr;
                e = launch_topk_merge(c->stream, (const float *)V.ptr, npad, 0, b0, b1 - b0, n, perm, descending, nn,
                                      (uint32_t *)didx.ptr, (float *)dval.ptr);
                if (e == hipSuccess)
                    e = launch_topk_merge(c->stream, (const float *)Vt.ptr, band, 1, b0, b1 - b0, n, perm, descending, nn,
                                          (uint32_t *)didx.ptr, (float *)dval.ptr);
                if (e != hipSuccess) rc = fail(c, DSH_EIO, "k_topk_merge: %s", hipGetErrorString(e));
            }
            if (rc) break;
            if (hipMemcpyAsync(idx_out, didx.ptr, n * nn * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
                hipMemcpyAsync(val_out, dval.ptr, n * nn * sizeof(float), hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
                hipStreamSynchronize(c->stream) != hipSuccess)
                rc = fail(c, DSH_EIO, "copy of neighbours failed");
        } while (0);
        (void)hipStreamSynchronize(c->stream);
        V.release();
        Vt.release();
        didx.release();
        dval.release();
        return rc;
    }
    DevBuf &rect = c->outbuf;
    const uint64_t qblock = std::max<uint64_t>(1, std::min<uint64_t>(nq, ((uint64_t)256 << 20) / std::max<uint64_t>(nr, 1)));
    HIPCHK(c, rect.ensure(std::max<uint64_t>(qblock * nr, 1) * sizeof(float)));
    DevBuf didx, dval;
    rc = DSH_OK;
    do {
        if (didx.ensure(nq * nn * sizeof(uint32_t)) != hipSuccess || dval.ensure(nq * nn * sizeof(float)) != hipSuccess) {
            rc = fail(c, DSH_ENOMEM, "device allocation failed");
            break;
        }
        for (uint64_t q0 = qb; q0 < qe && rc == DSH_OK; q0 += qblock) {
            const uint64_t q1 = std::min(qe, q0 + qblock);
            if (nr) {
                PairJob j;
                j.estim = estim;
                j.result_type = result_type;
                j.k = k;
                j.rect = 1;
                j.ksinv_double = 1;
                j.row_begin = q0;
                j.row_end = q1;
                j.col_begin = rb;
                j.col_end = re;
                j.base_index = 0;
                j.d_out = (float *)rect.ptr;
                rc = run_pairs(c, j);
                if (rc) break;
            }
            hipError_t e = launch_topk(c->stream, (const float *)rect.ptr, q1 - q0, nr, q0, rb, descending, nn,
                                       overlap ? 1 : 0, (uint32_t *)didx.ptr + (q0 - qb) * nn,
                                       (float *)dval.ptr + (q0 - qb) * nn);
            if (e != hipSuccess) rc = fail(c, DSH_EIO, "k_topk: %s", hipGetErrorString(e));
        }
        if (rc) break;
        if (hipMemcpyAsync(idx_out, didx.ptr, nq * nn * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
            hipMemcpyAsync(val_out, dval.ptr, nq * nn * sizeof(float), hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
            hipStreamSynchronize(c->stream) != hipSuccess)
            rc = fail(c, DSH_EIO, "copy of neighbours failed");
    } while (0);
    didx.release();
    dval.release();
    return rc;
}

}  // extern "C"

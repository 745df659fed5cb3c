/*
 * saena_oracle.c -- CPU restatement of the Saena V-cycle hot path (plain C).
 *
 * TEST INFRASTRUCTURE ONLY (see saena_oracle.h).  Compile with
 *   gcc -O2 -ffp-contract=off -fno-fast-math
 * so every sum keeps the reference's sequential order and no FMA is formed:
 * the HIP path is compared against exactly these roundings.
 *
 * Citations are file:line in the reference checkout (paralab/Saena).
 */
#include "saena_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#define SAENA_PI 3.1415926535897932384626433832795029   /* data_struct.h:44 */
#define ALMOST_ZERO 1e-14                                 /* data_struct.h:42 */

static void *xcalloc(size_t n, size_t sz) {
    void *p = calloc(n ? n : 1, sz);
    if (!p) { fprintf(stderr, "saena_oracle: out of memory (%zu x %zu)\n", n, sz); exit(EXIT_FAILURE); }
    return p;
}

void orc_free(void *p) { free(p); }

/* ------------------------------------------------------------------ */
/* ordering: cooEntry::operator< is column-major (data_struct.h:140-160) */
static int cmp_colmajor(const void *a, const void *b) {
    const orc_coo *x = a, *y = b;
    if (x->col != y->col) return x->col < y->col ? -1 : 1;
    if (x->row != y->row) return x->row < y->row ? -1 : 1;
    return 0;
}
static int cmp_rowmajor(const void *a, const void *b) {
    const orc_coo *x = a, *y = b;
    if (x->row != y->row) return x->row < y->row ? -1 : 1;
    if (x->col != y->col) return x->col < y->col ? -1 : 1;
    return 0;
}
void orc_sort_colmajor(orc_coo *e, nnz_t n) { qsort(e, (size_t)n, sizeof *e, cmp_colmajor); }
void orc_sort_rowmajor(orc_coo *e, nnz_t n) { qsort(e, (size_t)n, sizeof *e, cmp_rowmajor); }

/* aux_functions.h:39-58: index of the partition block that contains val */
static long lower_bound2(const index_t *left, const index_t *right, index_t val) {
    const index_t *first = left;
    while (left < right) {
        const index_t *middle = left + (right - left) / 2;
        if (*middle < val) left = middle + 1;
        else right = middle;
    }
    if (val == *left) return left - first;
    return (left - 1) - first;
}

/* ------------------------------------------------------------------ */
/* generators                                                          */

/* aux_functions2.cpp:254-373: 7-point stencil on an mx*my*mz node grid,
 * boundary rows carry a lone 1.0 on the diagonal and interior rows drop the
 * couplings to boundary nodes (:326-365).  remove_boundary_nodes
 * (saena_matrix_setup.cpp:281-365) then deletes every single-entry row and
 * renumbers the remaining ones in order, which leaves the interior nodes in
 * natural order (i fastest).  Both steps are fused here. */
nnz_t orc_laplacian3d(index_t mx, index_t my, index_t mz, orc_coo **out, index_t *Mbig_out) {
    const index_t nx = mx - 2, ny = my - 2, nz = mz - 2;
    const nnz_t   M  = (nnz_t)nx * ny * nz;
    const value_t Hx = 1.0 / (mx - 1), Hy = 1.0 / (my - 1), Hz = 1.0 / (mz - 1);
    const value_t HyHzdHx = 1.0 / (Hx * Hx), HxHzdHy = 1.0 / (Hy * Hy), HxHydHz = 1.0 / (Hz * Hz);
    orc_coo *e = xcalloc((size_t)M * 7, sizeof *e);
    nnz_t n = 0;
    for (index_t k = 1; k < mz - 1; ++k)
        for (index_t j = 1; j < my - 1; ++j)
            for (index_t i = 1; i < mx - 1; ++i) {
                const index_t node = (index_t)(((nnz_t)(k - 1) * ny + (j - 1)) * nx + (i - 1));
                if (k - 1 != 0)      { e[n].row = node; e[n].col = node - nx * ny; e[n].val = -HxHydHz; ++n; }
                if (j - 1 != 0)      { e[n].row = node; e[n].col = node - nx;      e[n].val = -HxHzdHy; ++n; }
                if (i - 1 != 0)      { e[n].row = node; e[n].col = node - 1;       e[n].val = -HyHzdHx; ++n; }
                e[n].row = node; e[n].col = node; e[n].val = 2.0 * (HxHydHz + HxHzdHy + HyHzdHx); ++n;
                if (i + 1 != mx - 1) { e[n].row = node; e[n].col = node + 1;       e[n].val = -HyHzdHx; ++n; }
                if (j + 1 != my - 1) { e[n].row = node; e[n].col = node + nx;      e[n].val = -HxHzdHy; ++n; }
                if (k + 1 != mz - 1) { e[n].row = node; e[n].col = node + nx * ny; e[n].val = -HxHydHz; ++n; }
            }
    /* the matrix is structurally symmetric and rows were emitted with
     * ascending columns, so swapping row<->col yields column-major order
     * without a sort; values are symmetric too. */
    for (nnz_t t = 0; t < n; ++t) { index_t r = e[t].row; e[t].row = e[t].col; e[t].col = r; }
    *out = e;
    *Mbig_out = (index_t)M;
    return n;
}

/* aux_functions2.cpp:629-700 + saena_object.cpp:699-730 */
void orc_laplacian3d_rhs(index_t mx, index_t my, index_t mz, value_t *rhs) {
    const value_t Hx = 1.0 / (mx - 1), Hy = 1.0 / (my - 1), Hz = 1.0 / (mz - 1);
    const double TWOPI = 2 * SAENA_PI, TWOELVEPISQ = 12 * SAENA_PI * SAENA_PI;
    nnz_t it = 0;
    for (index_t k = 1; k < mz - 1; ++k)
        for (index_t j = 1; j < my - 1; ++j)
            for (index_t i = 1; i < mx - 1; ++i)
                rhs[it++] = TWOELVEPISQ * sin(TWOPI * i * Hx) * sin(TWOPI * j * Hy) * sin(TWOPI * k * Hz);
}

/* aux_functions2.cpp:1296-1330: symmetric band, A(i,j) = 1/(i+j+1) */
nnz_t orc_band_matrix(index_t M, index_t bw, orc_coo **out) {
    orc_coo *e = xcalloc((size_t)M * (2 * (size_t)bw + 1), sizeof *e);
    nnz_t n = 0;
    for (index_t j = 0; j < M; ++j) {            /* column-major emission */
        index_t lo = j - bw < 0 ? 0 : j - bw, hi = j + bw >= M ? M - 1 : j + bw;
        for (index_t i = lo; i <= hi; ++i) { e[n].row = i; e[n].col = j; e[n].val = 1.0 / (i + j + 1); ++n; }
    }
    *out = e;
    return n;
}

void orc_split_even(index_t Mbig, int nprocs, index_t *split) {
    const index_t ofst = Mbig / nprocs;      /* saena_object.cpp:709-714 */
    for (int i = 0; i < nprocs; ++i) split[i] = i * ofst;
    split[nprocs] = Mbig;
}

/* saena_matrix_repart.cpp:43-170: nprocs^2 near-equal row buckets, merged
 * left to right until a rank holds about nnz_g/nprocs entries. */
void orc_split_nnz(const orc_coo *e, nnz_t nnz, index_t Mbig, int nprocs, index_t *split) {
    if (nprocs == 1) { split[0] = 0; split[1] = Mbig; return; }
    int n_buckets;
    if (Mbig > nprocs * nprocs) n_buckets = nprocs < 1000 ? nprocs * nprocs : 1000 * nprocs;
    else n_buckets = Mbig;
    index_t *splitOffset = xcalloc((size_t)n_buckets, sizeof *splitOffset);
    const index_t baseOffset = (index_t)floor(1.0 * Mbig / n_buckets);
    const float offsetRes = (float)(1.0 * Mbig / n_buckets) - baseOffset;
    float offsetResSum = 0;
    for (index_t i = 1; i < n_buckets; ++i) {
        splitOffset[i] = baseOffset;
        offsetResSum += offsetRes;
        if (offsetResSum >= 1) { splitOffset[i]++; offsetResSum -= 1; }
    }
    index_t *firstSplit = xcalloc((size_t)n_buckets + 1, sizeof *firstSplit);
    for (index_t i = 1; i < n_buckets; ++i) firstSplit[i] = firstSplit[i - 1] + splitOffset[i];
    firstSplit[n_buckets] = Mbig;
    nnz_t *H = xcalloc((size_t)n_buckets, sizeof *H);
    for (nnz_t i = 0; i < nnz; ++i) H[lower_bound2(firstSplit, firstSplit + n_buckets, e[i].row)]++;
    for (int i = 1; i < n_buckets; ++i) H[i] += H[i - 1];
    const nnz_t NNZ_PROC = nnz / nprocs;
    index_t procNum = 0;
    split[0] = 0;
    for (nnz_t i = 1; i < n_buckets; ++i) {
        if (Mbig - firstSplit[i + 1] < nprocs - (procNum + 1)) {
            for (; i < n_buckets; ++i) { procNum++; split[procNum] = firstSplit[i]; }
            break;
        }
        if (H[i] > (procNum + 1) * NNZ_PROC) { ++procNum; split[procNum] = firstSplit[i]; }
    }
    split[nprocs] = Mbig;
    free(splitOffset); free(firstSplit); free(H);
}

/* ------------------------------------------------------------------ */
/* operator construction                                               */

typedef struct { index_t row, col; value_t val; } coo_row;   /* cooEntry_row */

/* set_off_on_diagonal for one rank (saena_matrix_setup.cpp:793-945).
 * `entry` = this rank's rows, column-major sorted, global ids. */
static void build_rank(orc_rankop *R, const orc_coo *entry, nnz_t nnz_l,
                       const index_t *split_row, const index_t *split_col,
                       index_t **vElement_remote_out) {
    const int nprocs = R->nprocs, rank = R->rank;
    R->M        = split_row[rank + 1] - split_row[rank];
    R->row_ofst = split_row[rank];
    R->col_ofst = split_col[rank];
    R->nnz_l    = nnz_l;
    R->nnzPerRow_local = xcalloc((size_t)R->M, sizeof(index_t));
    R->recvCount       = xcalloc((size_t)nprocs, sizeof(int));
    R->nnzPerProcScan  = xcalloc((size_t)nprocs + 1, sizeof(nnz_t));
    coo_row *loc = xcalloc((size_t)nnz_l, sizeof *loc);
    index_t *rr  = xcalloc((size_t)nnz_l, sizeof *rr), *rc = xcalloc((size_t)nnz_l, sizeof *rc);
    index_t *rc2 = xcalloc((size_t)nnz_l, sizeof *rc2);
    value_t *rv  = xcalloc((size_t)nnz_l, sizeof *rv);
    index_t *vEl = xcalloc((size_t)nnz_l, sizeof *vEl);
    index_t *npc = xcalloc((size_t)nnz_l, sizeof *npc);
    nnz_t nloc = 0, nrem = 0; index_t ncolrem = 0;
    nnz_t i = 0;
    while (i < nnz_l) {                                   /* :828-859 */
        const long procNum = lower_bound2(split_col, split_col + nprocs, entry[i].col);
        if (procNum == rank) {
            while (i < nnz_l && entry[i].col < split_col[procNum + 1]) {
                ++R->nnzPerRow_local[entry[i].row - split_row[rank]];
                loc[nloc].row = entry[i].row - split_row[rank];
                loc[nloc].col = entry[i].col; loc[nloc].val = entry[i].val; ++nloc; ++i;
            }
        } else {
            const nnz_t tmp = i;
            while (i < nnz_l && entry[i].col < split_col[procNum + 1]) {
                vEl[ncolrem] = entry[i].col;
                ++R->recvCount[procNum];
                npc[ncolrem] = 0;
                do {
                    rc[nrem]  = ncolrem;
                    rc2[nrem] = entry[i].col;
                    rr[nrem]  = entry[i].row - split_row[rank];
                    rv[nrem]  = entry[i].val;
                    ++nrem; ++npc[ncolrem];
                } while (++i < nnz_l && entry[i].col == entry[i - 1].col);
                ++ncolrem;
            }
            R->nnzPerProcScan[procNum + 1] = i - tmp;
        }
    }
    R->nnz_l_local = nloc; R->nnz_l_remote = nrem; R->col_remote_size = ncolrem;
    R->recvCount[rank] = 0;
    qsort(loc, (size_t)nloc, sizeof *loc, cmp_rowmajor);  /* :905 row-major */
    R->row_local = xcalloc((size_t)nloc, sizeof(index_t));
    R->col_local = xcalloc((size_t)nloc, sizeof(index_t));
    R->val_local = xcalloc((size_t)nloc, sizeof(value_t));
    for (i = 0; i < nloc; ++i) { R->row_local[i] = loc[i].row; R->col_local[i] = loc[i].col; R->val_local[i] = loc[i].val; }
    free(loc);
    R->row_remote = xcalloc((size_t)nrem, sizeof(index_t)); memcpy(R->row_remote, rr, (size_t)nrem * sizeof(index_t));
    R->col_remote = xcalloc((size_t)nrem, sizeof(index_t)); memcpy(R->col_remote, rc, (size_t)nrem * sizeof(index_t));
    R->col_remote2 = xcalloc((size_t)nrem, sizeof(index_t)); memcpy(R->col_remote2, rc2, (size_t)nrem * sizeof(index_t));
    R->val_remote = xcalloc((size_t)nrem, sizeof(value_t)); memcpy(R->val_remote, rv, (size_t)nrem * sizeof(value_t));
    R->nnzPerCol_remote = xcalloc((size_t)ncolrem, sizeof(index_t)); memcpy(R->nnzPerCol_remote, npc, (size_t)ncolrem * sizeof(index_t));
    free(rr); free(rc); free(rc2); free(rv); free(npc);
    for (int p = 1; p < nprocs + 1; ++p) R->nnzPerProcScan[p] += R->nnzPerProcScan[p - 1];   /* :948-950 */
    *vElement_remote_out = vEl;
    R->temp1 = xcalloc((size_t)R->M, sizeof(value_t));
    R->temp2 = xcalloc((size_t)R->M, sizeof(value_t));
}

orc_op *orc_op_build(const orc_coo *entries, nnz_t nnz, index_t Mbig, index_t Nbig,
                     const index_t *split_row, const index_t *split_col, int nprocs,
                     int square_with_diag) {
    orc_op *op = xcalloc(1, sizeof *op);
    op->nprocs = nprocs; op->Mbig = Mbig; op->Nbig = Nbig; op->nnz_g = nnz;
    op->split_row = xcalloc((size_t)nprocs + 1, sizeof(index_t));
    op->split_col = xcalloc((size_t)nprocs + 1, sizeof(index_t));
    memcpy(op->split_row, split_row, ((size_t)nprocs + 1) * sizeof(index_t));
    memcpy(op->split_col, split_col, ((size_t)nprocs + 1) * sizeof(index_t));
    op->r = xcalloc((size_t)nprocs, sizeof(orc_rankop));
    op->jacobi_omega = (float)(2.0 / 3);            /* saena_matrix.h:182 */
    op->use_double = 1;

    /* distribute entries by owning row block, keeping column-major order */
    nnz_t *cnt = xcalloc((size_t)nprocs + 1, sizeof *cnt);
    int *owner = xcalloc((size_t)nnz, sizeof *owner);
    for (nnz_t i = 0; i < nnz; ++i) {
        owner[i] = (int)lower_bound2(split_row, split_row + nprocs, entries[i].row);
        cnt[owner[i] + 1]++;
    }
    for (int p = 0; p < nprocs; ++p) cnt[p + 1] += cnt[p];
    orc_coo *byrank = xcalloc((size_t)nnz, sizeof *byrank);
    nnz_t *fill = xcalloc((size_t)nprocs, sizeof *fill);
    for (nnz_t i = 0; i < nnz; ++i) { int p = owner[i]; byrank[cnt[p] + fill[p]++] = entries[i]; }
    free(owner); free(fill);

    index_t **vEl = xcalloc((size_t)nprocs, sizeof *vEl);
    for (int p = 0; p < nprocs; ++p) {
        orc_rankop *R = &op->r[p];
        R->rank = p; R->nprocs = nprocs;
        build_rank(R, byrank + cnt[p], cnt[p + 1] - cnt[p], split_row, split_col, &vEl[p]);
        if (square_with_diag) {                     /* inverse_diag, saena_matrix_setup.cpp:1562-1600 */
            R->inv_diag = xcalloc((size_t)R->M, sizeof(value_t));
            for (index_t i = 0; i < R->M; ++i) R->inv_diag[i] = 1.0;
            const orc_coo *en = byrank + cnt[p];
            for (nnz_t i = 0; i < R->nnz_l; ++i)
                if (en[i].row == en[i].col) {
                    if (fabs(en[i].val) < ALMOST_ZERO) {
                        fprintf(stderr, "Error on rank %d: there is a zero diagonal element at row index = %d\n", p, en[i].row);
                        exit(EXIT_FAILURE);
                    }
                    R->inv_diag[en[i].row - split_row[p]] = 1.0 / en[i].val;
                }
        }
    }
    /* halo plan (saena_matrix_setup.cpp:946-1062): Alltoall of counts,
     * Alltoallv of the requested global column ids. */
    for (int p = 0; p < nprocs; ++p) {
        orc_rankop *R = &op->r[p];
        R->sendCount = xcalloc((size_t)nprocs, sizeof(int));
        R->vdispls = xcalloc((size_t)nprocs, sizeof(int));
        R->rdispls = xcalloc((size_t)nprocs, sizeof(int));
        for (int q = 0; q < nprocs; ++q) R->sendCount[q] = op->r[q].recvCount[p];
        R->recvProcRank = xcalloc((size_t)nprocs, sizeof(int)); R->recvProcCount = xcalloc((size_t)nprocs, sizeof(int));
        R->sendProcRank = xcalloc((size_t)nprocs, sizeof(int)); R->sendProcCount = xcalloc((size_t)nprocs, sizeof(int));
        for (int q = 0; q < nprocs; ++q) {
            if (R->recvCount[q] != 0) { R->recvProcRank[R->numRecvProc] = q; R->recvProcCount[R->numRecvProc++] = R->recvCount[q]; }
            if (R->sendCount[q] != 0) { R->sendProcRank[R->numSendProc] = q; R->sendProcCount[R->numSendProc++] = R->sendCount[q]; }
        }
        for (int q = 1; q < nprocs; ++q) {
            R->vdispls[q] = R->vdispls[q - 1] + R->sendCount[q - 1];
            R->rdispls[q] = R->rdispls[q - 1] + R->recvCount[q - 1];
        }
        R->vIndexSize = R->vdispls[nprocs - 1] + R->sendCount[nprocs - 1];
        R->recvSize   = R->rdispls[nprocs - 1] + R->recvCount[nprocs - 1];
        R->vIndex     = xcalloc((size_t)R->vIndexSize, sizeof(index_t));
        R->vSend      = xcalloc((size_t)R->vIndexSize, sizeof(value_t));
        R->vecValues  = xcalloc((size_t)R->recvSize, sizeof(value_t));
        R->vSend_f    = xcalloc((size_t)R->vIndexSize, sizeof(float));
        R->vecValues_f = xcalloc((size_t)R->recvSize, sizeof(float));
    }
    for (int p = 0; p < nprocs; ++p)               /* Alltoallv + rebase (:1030-1046) */
        for (int q = 0; q < nprocs; ++q) {
            const orc_rankop *Q = &op->r[q];       /* q requests from p */
            for (int t = 0; t < Q->recvCount[p]; ++t)
                op->r[p].vIndex[op->r[p].vdispls[q] + t] = vEl[q][Q->rdispls[p] + t] - split_col[p];
        }
    for (int p = 0; p < nprocs; ++p) free(vEl[p]);
    free(vEl); free(byrank); free(cnt);
    return op;
}

void orc_op_free(orc_op *op) {
    if (!op) return;
    for (int p = 0; p < op->nprocs; ++p) {
        orc_rankop *R = &op->r[p];
        free(R->nnzPerRow_local); free(R->row_local); free(R->col_local); free(R->val_local);
        free(R->nnzPerCol_remote); free(R->row_remote); free(R->col_remote); free(R->col_remote2); free(R->val_remote);
        free(R->nnzPerProcScan); free(R->recvProcRank); free(R->recvProcCount); free(R->sendProcRank); free(R->sendProcCount);
        free(R->recvCount); free(R->sendCount); free(R->vdispls); free(R->rdispls);
        free(R->vIndex); free(R->vSend); free(R->vecValues); free(R->vSend_f); free(R->vecValues_f);
        free(R->inv_diag); free(R->temp1); free(R->temp2);
    }
    free(op->r); free(op->split_row); free(op->split_col); free(op);
}

/* ------------------------------------------------------------------ */
/* matvec                                                              */

/* saena_matrix_matvec.cpp:25-26 */
static void rank_pack(orc_rankop *R, const value_t *v_loc) {
    for (index_t i = 0; i < R->vIndexSize; ++i) R->vSend[i] = v_loc[R->vIndex[i]];
}
/* the Isend/Irecv pair of :32-41 as a copy out of the senders' vSend */
static void rank_recv(orc_op *op, int p) {
    orc_rankop *R = &op->r[p];
    for (int i = 0; i < R->numRecvProc; ++i) {
        const int q = R->recvProcRank[i];
        memcpy(&R->vecValues[R->rdispls[q]], &op->r[q].vSend[op->r[q].vdispls[p]],
               (size_t)R->recvProcCount[i] * sizeof(value_t));
    }
}
/* :44-80 */
static void rank_local(const orc_rankop *R, const value_t *v_loc, value_t *w_loc) {
    const index_t sz = R->M;
    for (index_t i = 0; i < sz; ++i) w_loc[i] = 0.0;
    const value_t *v_p = v_loc - R->col_ofst;
    nnz_t iter = 0;
    for (index_t i = 0; i < sz; ++i) {
        const index_t *col_local_p = &R->col_local[iter];
        const value_t *val_local_p = &R->val_local[iter];
        const index_t jend = R->nnzPerRow_local[i];
        value_t tmp = 0.0;
        for (index_t j = 0; j < jend; ++j) tmp += val_local_p[j] * v_p[col_local_p[j]];
        w_loc[i] += tmp;
        iter += jend;
    }
}
/* :87-110, neighbours taken in ascending rank order */
static void rank_remote(const orc_rankop *R, value_t *w_loc) {
    for (int np = 0; np < R->numRecvProc; ++np) {
        const int recv_proc = R->recvProcRank[np];
        nnz_t iter = R->nnzPerProcScan[recv_proc];
        const value_t *vecValues_p = &R->vecValues[R->rdispls[recv_proc]];
        const index_t *nnzPerCol_remote_p = &R->nnzPerCol_remote[R->rdispls[recv_proc]];
        for (index_t j = 0; j < R->recvCount[recv_proc]; ++j) {
            const index_t *row_remote_p = &R->row_remote[iter];
            const value_t *val_remote_p = &R->val_remote[iter];
            const index_t iend = nnzPerCol_remote_p[j];
            const value_t vrem = vecValues_p[j];
            for (index_t i = 0; i < iend; ++i) w_loc[row_remote_p[i]] += val_remote_p[i] * vrem;
            iter += iend;
        }
    }
}

void orc_matvec(orc_op *op, const value_t *v, value_t *w) {
    if (!op->use_double) { orc_matvec_float(op, v, w); return; }   /* saena_matrix.tpp:5-13 */
    for (int p = 0; p < op->nprocs; ++p) rank_pack(&op->r[p], v + op->split_col[p]);
    for (int p = 0; p < op->nprocs; ++p) rank_recv(op, p);
    for (int p = 0; p < op->nprocs; ++p) {
        rank_local(&op->r[p], v + op->split_col[p], w + op->split_row[p]);
        rank_remote(&op->r[p], w + op->split_row[p]);
    }
}

/* saena_matrix_matvec.cpp:448-550: halo values travel as float */
void orc_matvec_float(orc_op *op, const value_t *v, value_t *w) {
    for (int p = 0; p < op->nprocs; ++p) {
        orc_rankop *R = &op->r[p];
        const value_t *v_loc = v + op->split_col[p];
        for (index_t i = 0; i < R->vIndexSize; ++i) R->vSend_f[i] = (float)v_loc[R->vIndex[i]];   /* :464 */
    }
    for (int p = 0; p < op->nprocs; ++p) {
        orc_rankop *R = &op->r[p];
        for (int i = 0; i < R->numRecvProc; ++i) {
            const int q = R->recvProcRank[i];
            memcpy(&R->vecValues_f[R->rdispls[q]], &op->r[q].vSend_f[op->r[q].vdispls[p]],
                   (size_t)R->recvProcCount[i] * sizeof(float));
        }
        for (index_t i = 0; i < R->recvSize; ++i) R->vecValues[i] = (value_t)R->vecValues_f[i];   /* :531,538 */
    }
    for (int p = 0; p < op->nprocs; ++p) {
        rank_local(&op->r[p], v + op->split_col[p], w + op->split_row[p]);
        rank_remote(&op->r[p], w + op->split_row[p]);
    }
}

/* saena_matrix.tpp:16-23 */
void orc_residual(orc_op *op, const value_t *u, const value_t *rhs, value_t *res) {
    orc_matvec(op, u, res);
    for (index_t i = 0; i < op->Mbig; ++i) res[i] -= rhs[i];
}
/* saena_matrix.tpp:26-33 */
void orc_residual_negative(orc_op *op, const value_t *u, const value_t *rhs, value_t *res) {
    orc_matvec(op, u, res);
    for (index_t i = 0; i < op->Mbig; ++i) res[i] = rhs[i] - res[i];
}
/* saena_matrix.tpp:35-43, w = inv_diag */
void orc_residual_multiply(orc_op *op, const value_t *u, const value_t *rhs, value_t *res,
                           int w_is_inv_diag, value_t c) {
    (void)w_is_inv_diag;
    orc_matvec(op, u, res);
    for (int p = 0; p < op->nprocs; ++p) {
        const orc_rankop *R = &op->r[p];
        const index_t o = op->split_row[p];
        for (index_t i = 0; i < R->M; ++i) res[o + i] = c * R->inv_diag[i] * (rhs[o + i] - res[o + i]);
    }
}

/* saena_matrix.cpp:1044-1071 */
void orc_jacobi(orc_op *op, int iter, value_t *u, const value_t *rhs) {
    value_t *temp1 = xcalloc((size_t)op->Mbig, sizeof *temp1);
    const float omega = op->jacobi_omega;
    for (int j = 0; j < iter; ++j) {
        orc_matvec(op, u, temp1);
        for (int p = 0; p < op->nprocs; ++p) {
            const orc_rankop *R = &op->r[p];
            const index_t o = op->split_row[p];
            for (index_t i = 0; i < R->M; ++i) {
                temp1[o + i] -= rhs[o + i];
                temp1[o + i] *= R->inv_diag[i] * omega;
                u[o + i]     -= temp1[o + i];
            }
        }
    }
    free(temp1);
}

/* saena_matrix.cpp:1074-1131 */
void orc_chebyshev(orc_op *op, int iter, value_t *u, const value_t *rhs) {
    const double alpha = 0.13 * op->eig_max_of_invdiagXA;
    const double beta  = op->eig_max_of_invdiagXA;
    const double delta = (beta - alpha) / 2.0;
    const double theta = (beta + alpha) / 2.0;
    const double s1    = theta / delta;
    const double twos1 = 2.0 * s1;
    double rhok = 1.0 / s1, rhokp1 = 0.0, two_rhokp1 = 0.0, d1 = 0.0, d2 = 0.0;
    const index_t sz = op->Mbig;
    value_t *res = xcalloc((size_t)sz, sizeof *res), *d = xcalloc((size_t)sz, sizeof *d);
    orc_residual_multiply(op, u, rhs, d, 1, 1.0 / theta);
    for (index_t i = 0; i < sz; ++i) u[i] += d[i];
    for (int i = 1; i < iter; ++i) {
        rhokp1     = 1.0 / (twos1 - rhok);
        two_rhokp1 = 2.0 * rhokp1;
        d1         = rhokp1 * rhok;
        d2         = two_rhokp1 / delta;
        rhok       = rhokp1;
        orc_residual_multiply(op, u, rhs, res, 1, d2);
        for (index_t j = 0; j < sz; ++j) { d[j] = (d1 * d[j]) + res[j]; u[j] += d[j]; }
    }
    free(res); free(d);
}

/* aux_functions.h:116-123: per-rank sequential dot, MPI_SUM over ranks */
value_t orc_dot(const value_t *r, const value_t *s, const index_t *split, int nprocs) {
    value_t dot = 0.0;
    for (int p = 0; p < nprocs; ++p) {
        value_t dot_l = 0.0;
        for (index_t i = split[p]; i < split[p + 1]; ++i) dot_l += r[i] * s[i];
        dot += dot_l;
    }
    return dot;
}

/* ------------------------------------------------------------------ */
/* multigrid                                                           */

orc_amg *orc_amg_create(int nlevels, orc_op **A, orc_op **P, orc_op **R) {
    orc_amg *h = xcalloc(1, sizeof *h);
    h->max_level = nlevels - 1;
    h->grids = xcalloc((size_t)nlevels, sizeof(orc_grid));
    for (int l = 0; l < nlevels; ++l) {
        orc_grid *g = &h->grids[l];
        g->level = l; g->A = A[l];
        if (l < nlevels - 1) {
            g->P = P[l]; g->R = R[l]; g->coarse = &h->grids[l + 1];
            g->res         = xcalloc((size_t)A[l]->Mbig, sizeof(value_t));       /* grid.cpp:165-172 */
            g->uCorr       = xcalloc((size_t)A[l]->Mbig, sizeof(value_t));
            g->res_coarse  = xcalloc((size_t)A[l + 1]->Mbig, sizeof(value_t));
            g->uCorrCoarse = xcalloc((size_t)A[l + 1]->Mbig, sizeof(value_t));
        }
    }
    h->preSmooth = 3; h->postSmooth = 3; h->smoother = 0;
    h->CG_coarsest_max_iter = 150; h->CG_coarsest_tol = 1e-12;   /* saena_object.h:155-156 */
    h->solver_max_iter = 100; h->solver_tol = 1e-8;
    return h;
}

void orc_amg_free(orc_amg *h) {
    if (!h) return;
    for (int l = 0; l <= h->max_level; ++l) {
        orc_grid *g = &h->grids[l];
        free(g->res); free(g->uCorr); free(g->res_coarse); free(g->uCorrCoarse);
    }
    free(h->grids); free(h);
}

/* saena_object_solve.cpp:14-114 */
int orc_solve_coarsest_CG(const orc_amg *h, orc_op *A, value_t *u, const value_t *rhs) {
    const index_t sz = A->Mbig;
    const double tol = h->CG_coarsest_tol;
    value_t *res = xcalloc((size_t)sz, sizeof *res), *dir = xcalloc((size_t)sz, sizeof *dir);
    value_t *mt = xcalloc((size_t)sz, sizeof *mt);
    memcpy(res, rhs, (size_t)sz * sizeof *res);
    const double initial_dot = orc_dot(res, res, A->split_row, A->nprocs);
    const double thres = initial_dot * tol * tol;
    double dot = initial_dot;
    int max_iter = h->CG_coarsest_max_iter;
    if (dot < tol * tol) max_iter = 0;
    memcpy(dir, res, (size_t)sz * sizeof *dir);
    double factor = 0.0, dot_prev = 0.0;
    int i = 1;
    while (i < max_iter) {
        orc_matvec(A, dir, mt);
        factor = orc_dot(dir, mt, A->split_row, A->nprocs);
        factor = dot / factor;
        for (index_t j = 0; j < sz; ++j) { u[j] += factor * dir[j]; res[j] -= factor * mt[j]; }
        dot_prev = dot;
        dot = orc_dot(res, res, A->split_row, A->nprocs);
        if (dot < thres) break;
        factor = dot / dot_prev;
        for (index_t j = 0; j < sz; ++j) dir[j] = res[j] + factor * dir[j];
        i++;
    }
    if (i == max_iter && max_iter != 0) i--;
    free(res); free(dir); free(mt);
    return i;
}

/* include/saena_object.tpp:5-16 */
static void smooth(const orc_amg *h, orc_grid *g, value_t *u, const value_t *rhs, int iter) {
    if (h->smoother == 0) orc_jacobi(g->A, iter, u, rhs);
    else orc_chebyshev(g->A, iter, u, rhs);
}

/* saena_object_solve.cpp:961-1431 (scale=false, no repartition of coarse
 * vectors: every level keeps the partition its operators were built with) */
void orc_vcycle(const orc_amg *h, orc_grid *g, value_t *u, const value_t *rhs) {
    if (g->level == h->max_level) {                       /* :991-1057 */
        orc_solve_coarsest_CG(h, g->A, u, rhs);
        return;
    }
    const index_t sz = g->A->Mbig;
    if (h->preSmooth) smooth(h, g, u, rhs, h->preSmooth); /* :1105-1107 */
    orc_residual(g->A, u, rhs, g->res);                   /* :1140 */
    orc_matvec(g->R, g->res, g->res_coarse);              /* :1175 */
    const index_t szc = g->coarse->A->Mbig;
    for (index_t i = 0; i < szc; ++i) g->uCorrCoarse[i] = 0;   /* :1249 */
    orc_vcycle(h, g->coarse, g->uCorrCoarse, g->res_coarse);   /* :1254 */
    orc_matvec(g->P, g->uCorrCoarse, g->uCorr);           /* :1325 */
    for (index_t i = 0; i < sz; ++i) u[i] -= g->uCorr[i]; /* :1360-1361 */
    if (h->postSmooth) smooth(h, g, u, rhs, h->postSmooth);    /* :1397-1399 */
}

/* saena_object_solve.cpp:1883-2014 */
int orc_solve(const orc_amg *h, value_t *u, const value_t *rhs, double *hist, int cap) {
    orc_op *A = h->grids[0].A;
    const index_t sz = A->Mbig;
    for (index_t i = 0; i < sz; ++i) u[i] = 0;           /* :1926 */
    value_t *r = xcalloc((size_t)sz, sizeof *r);
    orc_residual(A, u, rhs, r);                           /* :1942 */
    double init_dot = orc_dot(r, r, A->split_row, A->nprocs), current_dot = init_dot;
    if (hist && cap > 0) hist[0] = sqrt(init_dot);
    const double thr = init_dot * h->solver_tol * h->solver_tol;
    int i = 0;
    for (; i < h->solver_max_iter; ++i) {                 /* :1957-1970 */
        orc_vcycle(h, &h->grids[0], u, rhs);
        orc_residual(A, u, rhs, r);
        current_dot = orc_dot(r, r, A->split_row, A->nprocs);
        if (hist && i + 1 < cap) hist[i + 1] = sqrt(current_dot);
        if (current_dot < thr) break;
    }
    if (i == h->solver_max_iter) i--;
    free(r);
    return i + 1;
}

/* saena_object_solve.cpp:2017-2117: the smoother alone as the iteration (preSmooth sweeps per step) */
int orc_solve_smoother(const orc_amg *h, value_t *u, const value_t *rhs, double *hist, int cap) {
    orc_op *A = h->grids[0].A;
    const index_t sz = A->Mbig;
    for (index_t i = 0; i < sz; ++i) u[i] = 0;           /* :2051 */
    value_t *r = xcalloc((size_t)sz, sizeof *r);
    orc_residual(A, u, rhs, r);                           /* :2060 */
    double init_dot = orc_dot(r, r, A->split_row, A->nprocs), current_dot = init_dot;
    if (hist && cap > 0) hist[0] = sqrt(init_dot);
    const double thr = init_dot * h->solver_tol * h->solver_tol;
    int i = 0;
    for (; i < h->solver_max_iter; ++i) {                 /* :2072-2081 */
        smooth(h, &h->grids[0], u, rhs, h->preSmooth);
        orc_residual(A, u, rhs, r);
        current_dot = orc_dot(r, r, A->split_row, A->nprocs);
        if (hist && i + 1 < cap) hist[i + 1] = sqrt(current_dot);
        if (current_dot < thr) break;
    }
    if (i == h->solver_max_iter) i--;
    free(r);
    return i + 1;
}

/* saena_object_solve.cpp:2389-2801 */
int orc_solve_pCG(const orc_amg *h, value_t *u, const value_t *rhs, double *hist, int cap) {
    orc_op *A = h->grids[0].A;
    const index_t sz = A->Mbig;
    const index_t *split = A->split_row; const int np = A->nprocs;
    for (index_t i = 0; i < sz; ++i) u[i] = 0;           /* :2482 */
    value_t *r = xcalloc((size_t)sz, sizeof *r), *rho = xcalloc((size_t)sz, sizeof *rho);
    value_t *hh = xcalloc((size_t)sz, sizeof *hh), *p = xcalloc((size_t)sz, sizeof *p);
    orc_residual(A, u, rhs, r);                           /* :2497 */
    const double init_dot = orc_dot(r, r, split, np);
    double current_dot = init_dot;
    if (hist && cap > 0) hist[0] = sqrt(init_dot);
    orc_vcycle(h, &h->grids[0], rho, r);                  /* :2536-2537 */
    memcpy(p, rho, (size_t)sz * sizeof *p);
    const double THRSHLD = init_dot * h->solver_tol * h->solver_tol;
    double rho_res = 0.0, pdoth = 0.0, alpha = 0.0, beta = 0.0;
    int i;
    for (i = 0; i < h->solver_max_iter; i++) {            /* :2565 */
        orc_matvec(A, p, hh);                             /* :2571 */
        rho_res = orc_dot(r, rho, split, np);             /* :2580 */
        pdoth   = orc_dot(p, hh, split, np);              /* :2581 */
        alpha = rho_res / pdoth;
        for (index_t j = 0; j < sz; ++j) { u[j] -= alpha * p[j]; r[j] -= alpha * hh[j]; }   /* :2593-2596 */
        current_dot = orc_dot(r, r, split, np);           /* :2603 */
        if (hist && i + 1 < cap) hist[i + 1] = sqrt(current_dot);
        if (current_dot < THRSHLD) break;                 /* :2620 */
        for (index_t j = 0; j < sz; ++j) rho[j] = 0.0;    /* :2640 */
        orc_vcycle(h, &h->grids[0], rho, r);              /* :2641 */
        beta = orc_dot(r, rho, split, np);                /* :2655 */
        beta /= rho_res;
        for (index_t j = 0; j < sz; ++j) p[j] = rho[j] + beta * p[j];   /* :2665-2667 */
    }
    if (i == h->solver_max_iter) i--;
    free(r); free(rho); free(hh); free(p);
    return i + 1;
}

/* saena_object_solve.cpp:2119-2387: solve_pCG with rho := r (no V-cycle) */
int orc_solve_CG(const orc_amg *h, value_t *u, const value_t *rhs, double *hist, int cap) {
    orc_op *A = h->grids[0].A;
    const index_t sz = A->Mbig;
    const index_t *split = A->split_row; const int np = A->nprocs;
    for (index_t i = 0; i < sz; ++i) u[i] = 0;
    value_t *r = xcalloc((size_t)sz, sizeof *r), *hh = xcalloc((size_t)sz, sizeof *hh), *p = xcalloc((size_t)sz, sizeof *p);
    orc_residual(A, u, rhs, r);
    const double init_dot = orc_dot(r, r, split, np);
    double current_dot = init_dot;
    if (hist && cap > 0) hist[0] = sqrt(init_dot);
    memcpy(p, r, (size_t)sz * sizeof *p);                 /* rho = r; p = rho */
    const double THRSHLD = init_dot * h->solver_tol * h->solver_tol;
    double rho_res = 0.0, pdoth = 0.0, alpha = 0.0, beta = 0.0;
    int i;
    for (i = 0; i < h->solver_max_iter; i++) {
        orc_matvec(A, p, hh);
        rho_res = orc_dot(r, r, split, np);               /* dotProduct(r, rho) with rho == r */
        pdoth   = orc_dot(p, hh, split, np);
        alpha = rho_res / pdoth;
        for (index_t j = 0; j < sz; ++j) { u[j] -= alpha * p[j]; r[j] -= alpha * hh[j]; }
        current_dot = orc_dot(r, r, split, np);
        if (hist && i + 1 < cap) hist[i + 1] = sqrt(current_dot);
        if (current_dot < THRSHLD) break;
        beta = orc_dot(r, r, split, np);
        beta /= rho_res;
        for (index_t j = 0; j < sz; ++j) p[j] = r[j] + beta * p[j];
    }
    if (i == h->solver_max_iter) i--;
    free(r); free(hh); free(p);
    return i + 1;
}

/* ------------------------------------------------------------------ */
/* threaded baseline: ranks run concurrently, barrier-separated phases  */

typedef struct {
    orc_op *op; const value_t *v; value_t *w; value_t *u; const value_t *rhs;
    int reps, nthreads, tid, jacobi;
    pthread_barrier_t *bar;
} tb_arg;

static void *tb_worker(void *a_) {
    tb_arg *a = a_;
    orc_op *op = a->op;
    for (int rep = 0; rep < a->reps; ++rep) {
        const value_t *src = a->jacobi ? a->u : a->v;
        for (int p = a->tid; p < op->nprocs; p += a->nthreads) rank_pack(&op->r[p], src + op->split_col[p]);
        pthread_barrier_wait(a->bar);
        for (int p = a->tid; p < op->nprocs; p += a->nthreads) {
            orc_rankop *R = &op->r[p];
            value_t *wl = a->jacobi ? R->temp1 : a->w + op->split_row[p];
            rank_recv(op, p);
            rank_local(R, src + op->split_col[p], wl);
            rank_remote(R, wl);
        }
        pthread_barrier_wait(a->bar);       /* all ranks finished reading u */
        if (a->jacobi) {
            const float omega = op->jacobi_omega;
            for (int p = a->tid; p < op->nprocs; p += a->nthreads) {
                orc_rankop *R = &op->r[p];
                const index_t o = op->split_row[p];
                for (index_t i = 0; i < R->M; ++i) {
                    R->temp1[i] -= a->rhs[o + i];
                    R->temp1[i] *= R->inv_diag[i] * omega;
                    a->u[o + i] -= R->temp1[i];
                }
            }
            pthread_barrier_wait(a->bar);
        }
    }
    return NULL;
}

static double tb_run(orc_op *op, const value_t *v, value_t *w, value_t *u, const value_t *rhs,
                     int reps, int threads, int jacobi) {
    if (threads > op->nprocs) threads = op->nprocs;
    if (threads < 1) threads = 1;
    pthread_barrier_t bar;
    pthread_barrier_init(&bar, NULL, (unsigned)threads);
    pthread_t *th = xcalloc((size_t)threads, sizeof *th);
    tb_arg *args = xcalloc((size_t)threads, sizeof *args);
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int t = 0; t < threads; ++t) {
        args[t] = (tb_arg){op, v, w, u, rhs, reps, threads, t, jacobi, &bar};
        pthread_create(&th[t], NULL, tb_worker, &args[t]);
    }
    for (int t = 0; t < threads; ++t) pthread_join(th[t], NULL);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    pthread_barrier_destroy(&bar);
    free(th); free(args);
    return ((t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec)) / reps;
}

double orc_time_matvec(orc_op *op, const value_t *v, value_t *w, int reps, int threads) {
    return tb_run(op, v, w, NULL, NULL, reps, threads, 0);
}
double orc_time_jacobi(orc_op *op, value_t *u, const value_t *rhs, int reps, int threads) {
    return tb_run(op, NULL, NULL, u, rhs, reps, threads, 1);
}

/* ---- dense storage (src/saena_matrix_dense.cpp) ---- */
void orc_matvec_dense(orc_op *op, const value_t *v, value_t *w, int as_float) {
    const int np = op->nprocs;
    const index_t Nbig = op->Nbig;
    for (int rank = 0; rank < np; ++rank) {
        orc_rankop *R = &op->r[rank];
        const index_t M = R->M;
        /* convert_saena_matrix, :763-793: entry[(row - split[rank]) * Nbig + col] = val */
        value_t *entry = xcalloc((size_t)(M > 0 ? M : 1) * (size_t)Nbig, sizeof(value_t));
        nnz_t k = 0;
        for (index_t i = 0; i < M; ++i)
            for (index_t t = 0; t < R->nnzPerRow_local[i]; ++t, ++k) entry[(size_t)i * Nbig + R->col_local[k]] = R->val_local[k];
        for (nnz_t q = 0; q < R->nnz_l_remote; ++q) entry[(size_t)R->row_remote[q] * Nbig + R->col_remote2[q]] = R->val_remote[q];
        value_t *wl = w + R->row_ofst;
        for (index_t i = 0; i < M; ++i) wl[i] = 0.0;                     /* :208 */
        for (int kk = rank; kk < rank + np; ++kk) {                      /* :217 ring: own block first */
            const int owner = kk % np;
            const index_t jst = op->split_col[owner], jend = op->split_col[owner + 1] - jst;
            for (index_t i = 0; i < M; ++i) {
                const value_t *entry_p = entry + (size_t)i * Nbig + jst;
                value_t tmp = 0;
                if (as_float) for (index_t j = 0; j < jend; ++j) tmp += entry_p[j] * (value_t)(float)v[jst + j];   /* :318 v_p is float */
                else          for (index_t j = 0; j < jend; ++j) tmp += entry_p[j] * v[jst + j];                 /* :243 */
                wl[i] += tmp;
            }
        }
        free(entry);
    }
}

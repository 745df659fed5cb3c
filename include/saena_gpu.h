/*
 * saena_gpu.h -- C ABI of libsaena_amd.so, the MI355X (gfx950) implementation of
 * the Saena V-cycle hot path.
 *
 * This is the drop-in boundary.  The reference (paralab/Saena) has no plugin
 * or FFI layer; its seam is the member-function level of the distributed
 * operators that saena_object::vcycle calls on host row-slices
 * (SURVEY.md section 8b).  Each entry point below names the reference
 * interface it replaces (file:line in the reference checkout).  All functions
 *   - take plain pointers and sizes (no C++/torch types),
 *   - return 0 on success or a negative sgpu_status (never exit()/abort(),
 *     unlike the reference, which prints and calls exit(EXIT_FAILURE)),
 *   - are, like the reference operators, NOT re-entrant per handle: one host
 *     thread drives one rank = one GPU.
 *
 * Typedefs mirror include/data_struct.h:36-38 of the reference.
 */
#ifndef SAENA_GPU_H
#define SAENA_GPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int    index_t;   /* data_struct.h:36 */
typedef long   nnz_t;     /* data_struct.h:37 */
typedef double value_t;   /* data_struct.h:38 */

typedef enum {
    SGPU_OK            =  0,
    SGPU_ERR_ARG       = -1,   /* bad argument / inconsistent descriptor     */
    SGPU_ERR_HIP       = -2,   /* a HIP runtime call failed                  */
    SGPU_ERR_RCCL      = -3,   /* an RCCL call failed                        */
    SGPU_ERR_STATE     = -4,   /* sgpu_init not called / already finalized   */
    SGPU_ERR_NOMEM     = -5,
    SGPU_ERR_NOCONV    = -6    /* solver hit max_iter without converging     */
} sgpu_status;

/* text of the last error on this thread's context ("" if none) */
const char *sgpu_last_error(void);

/* ---- context: one process = one rank = one GPU ---------------------------
 * Replaces MPI_Init/MPI_Comm_rank/size + the communicator stored in every
 * reference operator (saena_matrix.h:73 `MPI_Comm comm`).
 * rccl_unique_id: 128-byte ncclUniqueId produced by sgpu_get_unique_id on
 * rank 0 and handed to all ranks by the caller (any side channel); may be
 * NULL when nranks == 1.  Creates the compute stream and the halo stream. */
#define SGPU_UNIQUE_ID_BYTES 128
int sgpu_get_unique_id(void *out128);
int sgpu_init(int device_id, int rank, int nranks, const void *rccl_unique_id);
int sgpu_finalize(void);
int sgpu_rank(void);
int sgpu_nranks(void);
int sgpu_device_sync(void);          /* both streams idle                   */
int sgpu_barrier(void);              /* device sync + RCCL all-reduce of 1 int */

/* ---- device vectors (row slices of length local M) -----------------------
 * The reference passes raw host `value_t*` slices owned by the caller
 * (saena_matrix.h:304).  Here vectors live in HBM; these helpers replace
 * saena_aligned_alloc/saena_free (aux_functions.h:299-310). */
int sgpu_vec_alloc(value_t **dev, size_t n);
int sgpu_vec_free(value_t *dev);
int sgpu_vec_upload(value_t *dev, const value_t *host, size_t n);
int sgpu_vec_download(value_t *host, const value_t *dev, size_t n);
int sgpu_vec_fill(value_t *dev, value_t a, size_t n);            /* fill(), saena_object_solve.cpp:1249,2640 */
int sgpu_vec_copy(value_t *dst, const value_t *src, size_t n);
/* y = a*x + b*y (covers solve_pCG's updates, saena_object_solve.cpp:2593-2596,2665-2667) */
int sgpu_vec_axpby(value_t a, const value_t *x, value_t b, value_t *y, size_t n);
/* dotProduct (aux_functions.h:116-123): local dot on the GPU + all-reduce over ranks */
int sgpu_dot(const value_t *x, const value_t *y, size_t n, value_t *out);

/* ---- distributed sparse operator ----------------------------------------
 * One descriptor carries, for THIS rank, exactly the arrays the reference
 * operators hold after set_off_on_diagonal (saena_matrix_setup.cpp:793-1098),
 * prolong_matrix::findLocalRemote (prolong_matrix.cpp:18-378) or
 * restrict_matrix::transposeP (restrict_matrix.cpp:10-494), under the same
 * names.  Host pointers; the library copies what it needs (caller keeps
 * ownership) and re-lays the data out for the GPU:
 *   local  CSR (no row pointers, global columns) -> CSR with row_ptr, columns
 *          rebased by col_offset, plus a row-block plan for the kernel;
 *   remote CSC over the receive buffer -> CSR over the halo buffer, restricted
 *          to the rows that have remote entries (no atomics, deterministic);
 *   vIndex / send / recv plan -> persistent device send+recv buffers.       */
typedef struct {
    index_t        M;                 /* local rows                (saena_matrix.h:77)  */
    index_t        N_local;           /* local length of the input vector = split_col[rank+1]-split_col[rank] */
    index_t        col_offset;        /* split[rank] of the COLUMN partition: the kernel reads v[col - col_offset]
                                         (saena_matrix_matvec.cpp:56, restrict_matrix.cpp:662, prolong_matrix.cpp:538) */
    nnz_t          nnz_l_local;
    const index_t *nnzPerRow_local;   /* [M]                                            */
    const index_t *col_local;         /* [nnz_l_local], GLOBAL column ids, row-major    */
    const value_t *val_local;
    nnz_t          nnz_l_remote;
    index_t        col_remote_size;   /* == recvSize                                    */
    const index_t *nnzPerCol_remote;  /* [col_remote_size], in receive-buffer order     */
    const index_t *row_remote;        /* [nnz_l_remote] local row                       */
    const value_t *val_remote;
    int            numRecvProc, numSendProc;
    const int     *recvProcRank, *recvProcCount;   /* [numRecvProc], ascending rank     */
    const int     *sendProcRank, *sendProcCount;   /* [numSendProc], ascending rank     */
    index_t        vIndexSize;
    const index_t *vIndex;            /* [vIndexSize] local ids, grouped by destination rank (ascending) */
    const value_t *inv_diag;          /* [M] or NULL (R, P)        (saena_matrix.h, inverse_diag :1562)  */
    int            halo_fp32;         /* 1: halo travels as float  (matvec_sparse_float, saena_matrix_matvec.cpp:448-550) */
} sgpu_op_desc;

typedef struct sgpu_op sgpu_op;       /* opaque */

int sgpu_op_create(const sgpu_op_desc *desc, sgpu_op **out);
int sgpu_op_destroy(sgpu_op *op);
/* sizes / plan facts for reporting */
int sgpu_op_info(const sgpu_op *op, index_t *M, index_t *N_local, nnz_t *nnz_local, nnz_t *nnz_remote,
                 int *n_row_blocks, int *lanes_per_row);
/* kernel variant override (tuning/tests): lanes_per_row in {0=auto,1,2,4,...,64} */
int sgpu_op_set_lanes_per_row(sgpu_op *op, int lanes);
/* kernel variant of the local part: 0 k_csr_stream with 16 KiB tiles, 1 with 32 KiB tiles, 2 k_csr_vector
 * (no LDS staging), 3 / 4 k_csr_cc16 (16-bit compressed column ids, 10 B/nnz; the slot/offset split is chosen per
 * operator) on the 16 / 32 KiB plan, refused when a block touches more than 256 column segments of 256 columns;
 * 5 k_dense_rows: dense row-major storage (saena_matrix_dense, the reference's `switch_to_dense`; with a halo:
 * k_dense_rows_halo), at most 8192 rows and 64 M entries per rank; 6 k_csr_wave (long rows streamed by a wave);
 * 7 / 8 k_csr_cm: compressed columns with the entries of a row block in column order (12 B/nnz, fewer L1 requests on
 * rows of a few hundred entries) on the 16 / 32 KiB plan; 9 k_sell: sliced ELLPACK, a lane per row, 16-bit column codes
 * (operators whose slices of 64 rows pad to at most 12 % more entries; with SAENA_SELL_SORTED=1 also, as "k_sell<sorted>",
 * operators that pad to at most 5 % once their rows are sorted by length inside windows of 2048 rows -- opt-in: it
 * lost to the tile kernels wherever it was measured); 10 k_csr_xlds: the input vector staged in LDS
 * in windows of 20224 columns, one workgroup per CU (row chunks that reach over at most 8 windows; 4-64 lanes per row
 * piece, sgpu_op_set_lanes_per_row); 11 k_sellp: k_sell's values without a column stream -- a 16-bit id per row into a
 * table of (length, columns relative to the row) patterns held in LDS, 8 B per entry + 2 B per row (operators that
 * qualify for 9 and whose rows follow few enough patterns for a table of 4096 ints: stencils on structured grids, band
 * matrices; or, "k_sellp<wide>", where the patterns that every group of 1024 consecutive rows follows fit 8192 ints -- a
 * table per workgroup: the first smoothed-aggregation level of a structured grid; or, "k_sellp<rowbase>", where the rows repeat
 * relative to their FIRST COLUMN, kept per row -- the level-0 restriction / prolongation of a structured grid, the loops of
 * src/restrict_matrix.cpp:674-724 / src/prolong_matrix.cpp:550-614; the local loop of src/saena_matrix_matvec.cpp:68-80 with the same sequential row sum); 12 k_sellx: sliced
 * ELLPACK inside the (row chunk, column window) blocks of the x-in-LDS plan, a lane per row piece (rows of a few hundred
 * entries; at most 25 % padding); 13 k_rowt: row templates -- rows that repeat (length, relative columns, values) served
 * from a table in LDS, a 16-bit template id per row and nothing else of the operator (constant-coefficient stencils; never
 * chosen by the autotune unless SAENA_ROW_TEMPLATES=1); 14 k_sellp2: k_sellp with a lane per TWO adjacent rows in
 * slices of 128 rows -- the input vector is read, and the output written, 16 B at a time (operators that qualify for 11
 * and whose row pairs share a length; what the autotune keeps on fine levels larger than the Infinity Cache).
 * 15 k_sellpx: k_sellp with the windows of x that a workgroup of 512 rows reaches staged in LDS and a 16-bit table per
 * workgroup (operators that qualify for 11 and whose pattern offsets fall into at most 16 clusters that fit 80 KiB of LDS
 * together with the table); 16 k_csr_xldsr: k_csr_xlds with FOUR consecutive rows per group step (4 / 8 / 16 lanes per
 * group) -- irregular operators of a few dozen entries per row, where one row per step leaves the kernel waiting on the
 * latency of a row (BASELINE configs[4] scaled to 1 M rows).
 * 7, 8, 9 and 11 to 15 are
 * built from a host copy of the values that the library keeps only until the plan-time autotune (SGPU_ERR_ARG
 * afterwards, and where the form does not apply) */
int sgpu_op_set_variant(sgpu_op *op, int variant);
int sgpu_op_get_variant(const sgpu_op *op, int *variant, const char **kernel_name);
/* time the (variant, lanes) candidates that apply to this operator and keep the fastest (plan-time autotune; no
 * collective).  Call it right after sgpu_op_create: operators of up to 768 entries per row hold a host copy of their
 * values (8 B per entry) for the re-ordered forms from the create until this call -- or until sgpu_op_destroy. */
int sgpu_op_autotune(sgpu_op *op);

/* All vector arguments below are DEVICE pointers to this rank's slices.
 * Calls enqueue on the context's compute stream and return without waiting
 * unless stated otherwise; sgpu_device_sync() or any download waits.       */

/* w = A v.  saena_matrix::matvec -> matvec_sparse (saena_matrix_matvec.cpp:9-113);
 * restrict_matrix::matvec (restrict_matrix.cpp:612-744); prolong_matrix::matvec
 * (prolong_matrix.cpp:489-624). */
int sgpu_spmv(sgpu_op *op, const value_t *v, value_t *w);
/* res = A u - rhs.  saena_matrix::residual (saena_matrix.tpp:16-23) */
int sgpu_residual(sgpu_op *op, const value_t *u, const value_t *rhs, value_t *res);
/* res = rhs - A u.  saena_matrix::residual_negative (saena_matrix.tpp:26-33; the GMRES drivers of saena_object_solve.cpp:3848-4291).
 * Computed as 1 * 1 * (rhs - A u) through sgpu_residual_multiply: the same bits, signs of zeros included. */
int sgpu_residual_negative(sgpu_op *op, const value_t *u, const value_t *rhs, value_t *res);
/* res = c * w o (rhs - A u), evaluated as (c w_i) (rhs_i - (A u)_i).  saena_matrix::residual_multiply (saena_matrix.tpp:35-43: the
 * Chebyshev steps of saena_matrix.cpp:1099,1119 with w = inv_diag); fused into the operator's kernel.  res must not alias u, rhs or w. */
int sgpu_residual_multiply(sgpu_op *op, const value_t *u, const value_t *rhs, value_t *res, const value_t *w, value_t c);
/* iter damped-Jacobi sweeps, u in/out.  saena_matrix::jacobi (saena_matrix.cpp:1044-1071).
 * omega is the reference's float(2.0/3) promoted to double unless overridden. */
int sgpu_jacobi(sgpu_op *op, int iter, value_t omega, value_t *u, const value_t *rhs);
/* iter Chebyshev steps, u in/out.  saena_matrix::chebyshev (saena_matrix.cpp:1074-1131),
 * eig_max = eig_max_of_invdiagXA (saena_matrix.h:183). */
int sgpu_chebyshev(sgpu_op *op, int iter, value_t eig_max, value_t *u, const value_t *rhs);
/* u -= P e : prolong_matrix::matvec followed by the correction loop of
 * saena_object::vcycle (saena_object_solve.cpp:1325,1360-1361), fused. */
int sgpu_prolong_correct(sgpu_op *P, const value_t *e_coarse, value_t *u);

/* Host-slice forms with the reference's exact signatures' meaning
 * (`const value_t *v, value_t *w` on host memory): upload, run, download,
 * wait.  PCIe-inclusive; for integration behind an unchanged caller. */
int sgpu_spmv_host(sgpu_op *op, const value_t *v_host, value_t *w_host);
int sgpu_jacobi_host(sgpu_op *op, int iter, value_t omega, value_t *u_host, const value_t *rhs_host);
int sgpu_chebyshev_host(sgpu_op *op, int iter, value_t eig_max, value_t *u_host, const value_t *rhs_host);

/* ---- multigrid hierarchy --------------------------------------------------
 * Replaces the Grid array of saena_object (include/grid.h:11-78) for the solve
 * phase: level l holds A[l] and, for l < nlevels-1, P[l] (fine rows) and R[l]
 * (coarse rows).  Work vectors (res, uCorr, res_coarse, uCorrCoarse) are
 * allocated here, once (Grid::allocate_mem, grid.cpp:165-172). */
typedef struct {
    int     preSmooth, postSmooth;        /* saena.hpp:151-155 */
    int     smoother;                     /* 0 "jacobi", 1 "chebyshev" (saena_object.tpp:5-16) */
    value_t jacobi_omega;                 /* 0 => float(2.0/3) */
    int     coarse_solver;                /* 1 (default) direct: dense inverse factored once on the host, like the reference's
                                             default direct_solver "SuperLU" (saena_object.h:165, saena_object_solve.cpp:793-958);
                                             0 CG (solve_coarsest_CG, saena_object_solve.cpp:14-114) */
    int     CG_coarsest_max_iter;         /* 150   saena_object.h:156 */
    value_t CG_coarsest_tol;              /* 1e-12 saena_object.h:155 */
    int     solver_max_iter;              /* saena::options */
    value_t solver_tol;
    int     use_graph;                    /* 1 (default): capture the V-cycle once per (u, rhs) pair and replay it as a
                                             hipGraph.  One rank: the whole V-cycle.  Several ranks: the levels whose
                                             operators exchange a halo on this rank launch eagerly (the exchange is an RCCL
                                             group), the communication-free tail below them is one captured graph */
} sgpu_amg_params;

typedef struct sgpu_amg sgpu_amg;         /* opaque */

int sgpu_amg_default_params(sgpu_amg_params *p);
/* eig_max[l] = eig_max_of_invdiagXA of A[l] (only read for chebyshev; may be NULL) */
int sgpu_amg_create(int nlevels, sgpu_op *const *A, sgpu_op *const *P, sgpu_op *const *R,
                    const value_t *eig_max, const sgpu_amg_params *params, sgpu_amg **out);
int sgpu_amg_destroy(sgpu_amg *h);       /* does not destroy the operators */
/* saena_object::vcycle (saena_object_solve.cpp:961-1431) on level 0 */
int sgpu_vcycle(sgpu_amg *h, value_t *u, const value_t *rhs);
/* saena_object::solve (saena_object_solve.cpp:1883-2014): u is zeroed, then V-cycles until
 * ||r||^2 < ||r0||^2 tol^2.  res_hist[k] = ||r_k||, k = 0..iters (capacity hist_cap). */
int sgpu_solve(sgpu_amg *h, value_t *u, const value_t *rhs, int *iters, value_t *res_hist, int hist_cap);
/* saena_object::solve_pCG (saena_object_solve.cpp:2389-2801) */
int sgpu_solve_pCG(sgpu_amg *h, value_t *u, const value_t *rhs, int *iters, value_t *res_hist, int hist_cap);
/* saena_object::set_solve_params (called by every saena::amg::solve* before the solve, saena.cpp:751-790):
 * replaces the iteration limit, tolerance, smoother (0 jacobi, 1 chebyshev) and sweep counts of an existing
 * hierarchy.  Captured V-cycle graphs are dropped when the V-cycle shape changes. */
int sgpu_amg_set_solve_params(sgpu_amg *h, int solver_max_iter, double solver_tol, int smoother, int preSmooth, int postSmooth);
/* saena_object::profile_matvecs (saena_object.cpp:618-638): average time of `iter` matvecs with A of every level;
 * us_per_level[nlevels] receives microseconds (this rank, HIP events around the launches, halo included). */
int sgpu_amg_profile_matvecs(sgpu_amg *h, int iter, double *us_per_level);
/* saena_object::solve_smoother (saena_object_solve.cpp:2017-2117): preSmooth sweeps of the configured
 * smoother on A[0] per iteration, no coarse-grid correction */
int sgpu_solve_smoother(sgpu_amg *h, value_t *u, const value_t *rhs, int *iters, value_t *res_hist, int hist_cap);
/* saena_object::solve_CG (saena_object_solve.cpp:2119-2387): plain CG on A[0], no V-cycle */
int sgpu_solve_CG(sgpu_amg *h, value_t *u, const value_t *rhs, int *iters, value_t *res_hist, int hist_cap);
/* solve_coarsest_CG on the last level only (for tests) */
int sgpu_coarsest_solve(sgpu_amg *h, value_t *u, const value_t *rhs, int *iters);

/* ---- measurement -----------------------------------------------------------
 * Runs `reps` back-to-back launches of one kernel on the compute stream,
 * bracketed by hipEvents recorded on that same stream; *ms_per_launch is the
 * mean.  kind: 0 spmv, 1 jacobi sweep, 2 residual, 3 chebyshev step. */
int sgpu_time_kernel(sgpu_op *op, int kind, const value_t *x, const value_t *rhs, value_t *y,
                     int reps, float *ms_per_launch);
/* algorithmic bytes of one launch of `kind` on this operator */
int sgpu_algorithmic_bytes(const sgpu_op *op, int kind, int64_t *bytes);

#ifdef __cplusplus
}
#endif
#endif /* SAENA_GPU_H */

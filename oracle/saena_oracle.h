/*
 * saena_oracle.h -- CPU restatement of the Saena V-cycle hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is the parity oracle for the HIP path in
 * saena_amd/: only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it.  The product never calls into it.
 *
 * Every function restates, in plain C and in the reference's arithmetic
 * order, a loop of paralab/Saena (citations are file:line relative to the
 * reference checkout).  Ranks of the reference's MPI row partition are
 * simulated inside one process: a distributed operator is an array of
 * per-rank blocks in the reference's own storage layout (local CSR without
 * row pointers + remote CSC over the receive buffer + halo plan), and the
 * halo exchange is a memcpy between the per-rank send/receive buffers.
 * Neighbour contributions are accumulated in ascending-rank order, one of
 * the arrival orders MPI_Waitany (saena_matrix_matvec.cpp:88) may produce.
 *
 * Parity pin: validated against the compiled reference operators
 * (oracle/ref, built into oracle/_ref/) and the reference-run known answers
 * recorded in SURVEY.md section 8c; see tests/test_oracle_pins.py.
 */
#ifndef SAENA_ORACLE_H
#define SAENA_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

/* include/data_struct.h:36-38 */
typedef int    index_t;
typedef long   nnz_t;
typedef double value_t;

/* include/data_struct.h:116-124 (cooEntry: 16-byte row,col,val triple) */
typedef struct { index_t row, col; value_t val; } orc_coo;

/* One rank's block of a distributed operator, in the reference's layout
 * (include/saena_matrix.h:75-149, built by set_off_on_diagonal,
 * src/saena_matrix_setup.cpp:793-1098).  A, R and P all use it
 * (include/restrict_matrix.h:17-66, include/prolong_matrix.h:17-99). */
typedef struct {
    int      rank, nprocs;
    index_t  M;                 /* local rows                                   */
    index_t  row_ofst;          /* split_row[rank]                              */
    index_t  col_ofst;          /* split_col[rank] (v_p = v - split[rank])      */
    nnz_t    nnz_l;             /* all local entries                            */
    nnz_t    nnz_l_local, nnz_l_remote;
    index_t  col_remote_size;
    index_t *nnzPerRow_local;   /* [M]                                          */
    index_t *row_local;         /* [nnz_l_local] local row (unused by matvec)   */
    index_t *col_local;         /* [nnz_l_local] GLOBAL column ids              */
    value_t *val_local;
    index_t *nnzPerCol_remote;  /* [col_remote_size]                            */
    index_t *row_remote;        /* [nnz_l_remote] local row                     */
    index_t *col_remote;        /* [nnz_l_remote] position in vecValues         */
    index_t *col_remote2;       /* [nnz_l_remote] original global column        */
    value_t *val_remote;
    nnz_t   *nnzPerProcScan;    /* [nprocs+1]                                   */
    int      numRecvProc, numSendProc;
    int     *recvProcRank, *recvProcCount, *sendProcRank, *sendProcCount;
    int     *recvCount, *sendCount, *vdispls, *rdispls;   /* [nprocs]           */
    index_t  vIndexSize, recvSize;
    index_t *vIndex;            /* [vIndexSize] local ids to send               */
    value_t *vSend, *vecValues;
    float   *vSend_f, *vecValues_f;
    value_t *inv_diag;          /* [M] square operators only                    */
    value_t *temp1, *temp2;     /* [M] smoother scratch                         */
} orc_rankop;

typedef struct {
    int        nprocs;
    index_t    Mbig, Nbig;
    nnz_t      nnz_g;
    index_t   *split_row;       /* [nprocs+1] */
    index_t   *split_col;       /* [nprocs+1] */
    orc_rankop *r;              /* [nprocs]   */
    double     eig_max_of_invdiagXA;   /* saena_matrix.h:183 */
    float      jacobi_omega;           /* saena_matrix.h:182: float(2.0/3) */
    int        use_double;             /* 0 => fp32 halo (matvec_sparse_float) */
} orc_op;

/* ---- generators (input definitions) ---- */
/* laplacian3D (src/aux_functions2.cpp:254-373) followed by
 * remove_boundary_nodes (src/saena_matrix_setup.cpp:281-506): the interior
 * (mx-2)(my-2)(mz-2) system in natural order.  Entries are returned sorted
 * column-major (col, then row), the order of saena_matrix::entry. */
nnz_t  orc_laplacian3d(index_t mx, index_t my, index_t mz, orc_coo **out, index_t *Mbig);
/* laplacian3D_set_rhs (src/aux_functions2.cpp:629-700) with boundary rows
 * dropped (remove_boundary_rhs, src/saena_object.cpp:699-730). */
void   orc_laplacian3d_rhs(index_t mx, index_t my, index_t mz, value_t *rhs);
/* band_matrix (src/aux_functions2.cpp:1296-1381): A(i,j) = 1/(i+j+1) (0-based
 * i+j+1), |i-j| <= bandwidth, diagonal as generated there. */
nnz_t  orc_band_matrix(index_t M, index_t bandwidth, orc_coo **out);
void   orc_free(void *p);

/* sort helpers */
void   orc_sort_colmajor(orc_coo *e, nnz_t n);
void   orc_sort_rowmajor(orc_coo *e, nnz_t n);

/* equal-row split Mbig/nprocs (used for vectors and as a simple partition) */
void   orc_split_even(index_t Mbig, int nprocs, index_t *split);
/* the reference's nnz-balanced initial partition
 * (src/saena_matrix_repart.cpp:43-170) */
void   orc_split_nnz(const orc_coo *e_colmajor, nnz_t nnz, index_t Mbig, int nprocs, index_t *split);

/* ---- operator construction: set_off_on_diagonal + inverse_diag ---- */
orc_op *orc_op_build(const orc_coo *entries, nnz_t nnz, index_t Mbig, index_t Nbig,
                     const index_t *split_row, const index_t *split_col, int nprocs,
                     int square_with_diag);
void    orc_op_free(orc_op *op);

/* ---- hot path ---- */
/* saena_matrix::matvec_sparse (src/saena_matrix_matvec.cpp:9-113),
 * restrict_matrix::matvec_sparse (src/restrict_matrix.cpp:612-744),
 * prolong_matrix::matvec_sparse (src/prolong_matrix.cpp:489-624).
 * v: global vector of length Nbig (rank r reads v[split_col[r]..]);
 * w: global vector of length Mbig. */
void orc_matvec(orc_op *op, const value_t *v, value_t *w);
/* src/saena_matrix_matvec.cpp:448-550 */
void orc_matvec_float(orc_op *op, const value_t *v, value_t *w);
/* saena_matrix_dense::convert_saena_matrix + matvec_dense / matvec_dense_float (src/saena_matrix_dense.cpp:763-793,
 * :181-260, :262-340), the reference's optional `switch_to_dense` storage of a square operator: every rank holds its
 * rows as a dense M x Mbig block and the x blocks travel round a ring; rank r adds, block by block starting with its
 * own, tmp = sum_j A[i][j] x_owner[j] (ascending j) and then w[i] += tmp.  as_float != 0: EVERY block of x -- the
 * rank's own included -- is rounded to float first (matvec_dense_float copies v into float buffers). */
void orc_matvec_dense(orc_op *op, const value_t *v, value_t *w, int as_float);
/* include/saena_matrix.tpp:16-43 */
void orc_residual(orc_op *op, const value_t *u, const value_t *rhs, value_t *res);
void orc_residual_negative(orc_op *op, const value_t *u, const value_t *rhs, value_t *res);
void orc_residual_multiply(orc_op *op, const value_t *u, const value_t *rhs, value_t *res,
                           int w_is_inv_diag, value_t c);
/* src/saena_matrix.cpp:1044-1071 */
void orc_jacobi(orc_op *op, int iter, value_t *u, const value_t *rhs);
/* src/saena_matrix.cpp:1074-1131 */
void orc_chebyshev(orc_op *op, int iter, value_t *u, const value_t *rhs);
/* include/aux_functions.h:116-123 (per-rank partial sums, rank-ordered add) */
value_t orc_dot(const value_t *r, const value_t *s, const index_t *split, int nprocs);

/* ---- multigrid ---- */
typedef struct orc_grid {
    int      level;
    orc_op  *A;
    orc_op  *P, *R;            /* NULL on the coarsest level */
    struct orc_grid *coarse;
    value_t *res, *uCorr, *res_coarse, *uCorrCoarse;   /* include/grid.h:11-78 */
} orc_grid;

typedef struct {
    int      max_level;        /* index of the coarsest grid                    */
    orc_grid *grids;           /* [max_level+1]                                 */
    int      preSmooth, postSmooth;
    int      smoother;         /* 0 = jacobi, 1 = chebyshev                     */
    int      CG_coarsest_max_iter;   /* saena_object.h:155-156: 150 */
    double   CG_coarsest_tol;        /* 1e-12 */
    int      solver_max_iter;
    double   solver_tol;
} orc_amg;

orc_amg *orc_amg_create(int nlevels, orc_op **A, orc_op **P, orc_op **R);
void     orc_amg_free(orc_amg *h);   /* does not free the operators */
/* src/saena_object_solve.cpp:14-114 */
int  orc_solve_coarsest_CG(const orc_amg *h, orc_op *A, value_t *u, const value_t *rhs);
/* src/saena_object_solve.cpp:961-1431 */
void orc_vcycle(const orc_amg *h, orc_grid *g, value_t *u, const value_t *rhs);
/* src/saena_object_solve.cpp:1883-2014; res_hist[k] = ||r_k|| (k=0 initial) */
int  orc_solve(const orc_amg *h, value_t *u, const value_t *rhs, double *res_hist, int hist_cap);
/* src/saena_object_solve.cpp:2017-2117 */
int  orc_solve_smoother(const orc_amg *h, value_t *u, const value_t *rhs, double *res_hist, int hist_cap);
/* src/saena_object_solve.cpp:2389-2801 */
int  orc_solve_pCG(const orc_amg *h, value_t *u, const value_t *rhs, double *res_hist, int hist_cap);

/* src/saena_object_solve.cpp:2119-2387: CG without a preconditioner (rho aliases r) */
int  orc_solve_CG(const orc_amg *h, value_t *u, const value_t *rhs, double *res_hist, int hist_cap);

/* ---- threaded baseline: run P simulated ranks on P pthreads ---- */
/* times `reps` matvecs (or jacobi sweeps) with the ranks running
 * concurrently like `mpirun -np P`; returns seconds per repetition. */
double orc_time_matvec(orc_op *op, const value_t *v, value_t *w, int reps, int threads);
double orc_time_jacobi(orc_op *op, value_t *u, const value_t *rhs, int reps, int threads);

#ifdef __cplusplus
}
#endif
#endif

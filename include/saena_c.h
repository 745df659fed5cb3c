/*
 * saena_c.h -- flat C view of the host-side mirror of Saena's public surface
 * (include/saena.hpp in this repository; reference: include/saena.hpp:14-265).
 *
 * The reference's users write C++ (saena::matrix A(comm); A.set(...);
 * A.assemble(); saena::amg s; s.set_matrix(&A,&opts); s.solve_pCG(u,&opts)).
 * This header exposes the same steps with plain pointers so that ctypes / cgo
 * style callers -- and this repository's tests and bench.py -- can drive them.
 * Host-only functions live in libsaena_host.so (no GPU needed) and again in
 * libsaena_amd.so; functions marked [GPU] exist only in libsaena_amd.so.
 * All return 0 on success, negative on error (saena_last_error() has the text).
 */
#ifndef SAENA_C_H
#define SAENA_C_H

#include "saena_gpu.h"

#ifdef __cplusplus
extern "C" {
#endif

const char *saena_last_error(void);

/* ---- communicator for the setup-time exchanges (the reference: MPI_Comm) ---- */
typedef struct saena_comm saena_comm;
typedef int (*saena_cb_allgather)(void *user, const void *send, void *recv, size_t bytes);
typedef int (*saena_cb_alltoallv)(void *user, const void *send, const size_t *scounts, const size_t *sdispls,
                                  void *recv, const size_t *rcounts, const size_t *rdispls);
typedef int (*saena_cb_allreduce_i64)(void *user, long *v, int n);
typedef int (*saena_cb_allreduce_f64)(void *user, double *v, int n);

saena_comm *saena_comm_self(void);
saena_comm *saena_comm_callbacks(int rank, int nranks, void *user, saena_cb_allgather, saena_cb_alltoallv,
                                 saena_cb_allreduce_i64, saena_cb_allreduce_f64);
saena_comm *saena_comm_rccl(void);            /* [GPU] the sgpu_init() communicator */
/* the ranks of ONE node through POSIX shared memory (host/shm_comm.h): native, no interpreter and no device in the loop;
 * collective over the job's ranks, `name` fresh for every job (no '/'); NULL + saena_last_error() on failure */
saena_comm *saena_comm_shm(const char *name, int rank, int nranks);
void        saena_comm_free(saena_comm *);
/* the exchange chain of a multi-rank apply that the agglomeration of coarse levels starts from, microseconds: measured by the
 * GPU runtime on the job's communicator at sgpu_init (0: nothing measured, the model's constant applies) */
double      saena_measured_chain_us(void);
/* the communicator's collectives, exported for its tests (counts and displacements in bytes, one per rank) */
int saena_comm_test_alltoallv(saena_comm *, const void *send, const size_t *scounts, const size_t *sdispls, void *recv,
                              const size_t *rcounts, const size_t *rdispls);
int saena_comm_test_allreduce_f64(saena_comm *, double *v, int n);
int saena_comm_test_allreduce_i64(saena_comm *, long *v, int n);

/* ---- saena::matrix (reference include/saena.hpp:14-73) ---- */
typedef struct saena_matrix_h saena_matrix_h;
saena_matrix_h *saena_matrix_new(saena_comm *comm);
void  saena_matrix_free(saena_matrix_h *A);
int   saena_matrix_set(saena_matrix_h *A, index_t i, index_t j, value_t val);                       /* saena.hpp:29 */
int   saena_matrix_set_many(saena_matrix_h *A, const index_t *row, const index_t *col, const value_t *val, nnz_t n); /* :30 */
int   saena_matrix_read_file(saena_matrix_h *A, const char *name, const char *input_type /* "" */);     /* saena.hpp:25-26 */
int   saena_matrix_write_bin(saena_matrix_h *A, const char *name);
int   saena_matrix_write_mtx(saena_matrix_h *A, const char *name);   /* saena.hpp:53 writeMatrixToFile: "<name>-r<rank>.mtx" */
int   saena_matrix_set_remove_boundary(saena_matrix_h *A, int remove_bound);                        /* :44 */
/* resolution of the nnz-balanced row partition of assemble(): 0 = the reference's nparts^2 row buckets
   (src/saena_matrix_repart.cpp:43-170; max / mean rows 1.25 at 4 ranks, 1.125 at 8 on a uniform operator), n > 0 = at least n
   buckets -- opt-in, NOT the reference's partition (also: SAENA_FINE_PARTITION_BUCKETS) */
int   saena_matrix_set_partition_buckets(saena_matrix_h *A, int n_buckets);
int   saena_matrix_add_duplicates(saena_matrix_h *A, int add);                                      /* :47 */
int   saena_matrix_set_eig(saena_matrix_h *A, double eig);                                          /* :38 (value form) */
int   saena_matrix_assemble(saena_matrix_h *A);                                                     /* :49 */
int   saena_matrix_assemble_with_split(saena_matrix_h *A, const index_t *split);
index_t saena_matrix_get_num_rows(saena_matrix_h *A);                                               /* :57 */
index_t saena_matrix_get_num_local_rows(saena_matrix_h *A);                                         /* :58 */
nnz_t   saena_matrix_get_nnz(saena_matrix_h *A);                                                    /* :59 */
nnz_t   saena_matrix_get_local_nnz(saena_matrix_h *A);                                              /* :60 */
int   saena_matrix_get_split(saena_matrix_h *A, index_t *split_out /* nranks+1 */);
/* this rank's arrays in the reference's storage layout; pointers stay valid while A lives */
int   saena_matrix_get_desc(saena_matrix_h *A, sgpu_op_desc *out);
/* remaining layout arrays not part of sgpu_op_desc, for layout tests: col_remote[nnz_l_remote], nnzPerProcScan[nranks+1] */
int   saena_matrix_get_layout_extra(saena_matrix_h *A, const index_t **col_remote, const nnz_t **nnzPerProcScan);
/* global column id of every slot of the receive (halo) buffer, [col_remote_size] -- vElement_remote (saena_matrix.h:57) */
int   saena_matrix_get_halo_columns(saena_matrix_h *A, const index_t **vElement_remote);

/* generators (reference src/aux_functions2.cpp:254-373, :629-700, :1296-1381) */
int   saena_laplacian3D(saena_matrix_h *A, index_t mx, index_t my, index_t mz);
/* rhs of the assembled interior system, this rank's slice (length get_num_local_rows) */
int   saena_laplacian3D_set_rhs(saena_matrix_h *A, index_t mx, index_t my, index_t mz, value_t *rhs_local);
int   saena_band_matrix(saena_matrix_h *A, index_t M, unsigned int bandwidth);

/* saena::amg::matmat (reference saena.hpp:313, saena_object_setup_matmat.cpp:1164-1487): C = A B on the host,
 * A and B assembled, one rank in this round; C (a fresh handle's contents are replaced) is assembled on return */
int   saena_matmat(saena_matrix_h *A, saena_matrix_h *B, saena_matrix_h *C);

/* ---- transfer operators (prolong_matrix / restrict_matrix) ---- */
typedef struct saena_transfer_h saena_transfer_h;
/* rows/cols are GLOBAL ids of this rank's fine rows; split_row = fine partition, split_col = coarse partition */
saena_transfer_h *saena_prolong_new(saena_comm *comm, index_t Mbig, index_t Nbig, const index_t *split_row,
                                    const index_t *split_col, const index_t *row, const index_t *col,
                                    const value_t *val, nnz_t n);
saena_transfer_h *saena_restrict_from_prolong(saena_transfer_h *P);       /* restrict_matrix::transposeP */
void  saena_transfer_free(saena_transfer_h *T);
int   saena_transfer_get_desc(saena_transfer_h *T, sgpu_op_desc *out);
nnz_t saena_transfer_get_local_nnz(saena_transfer_h *T);

/* ---- saena::options + saena::amg (reference include/saena.hpp:127-265) ---- */
typedef struct {
    int    solver_max_iter;      /* saena.hpp:151-155 defaults: 100, 1e-8, "chebyshev", 3, 3, "jacobi", 0.3, true, 10, 3, */
    double relative_tol;         /*                             1e-14, 1e-8, 1, 2, false, 0.1, 5000                       */
    int    smoother;             /* 0 "jacobi", 1 "chebyshev" */
    int    preSmooth, postSmooth;
    float  connStrength;
    int    dynamic_levels;
    int    max_level;
    int    float_level;
    double filter_thre, filter_max;
    int    filter_start, filter_rate;
    int    switch_to_dense;      /* saena.hpp:146-148: store levels denser than dense_thre (and smaller than dense_sz_thre rows) */
    float  dense_thre;           /* as dense row-major arrays -- saena_matrix_dense; off in every options file of the reference */
    int    dense_sz_thre;
} saena_options_c;
int saena_options_default(saena_options_c *o);
int saena_options_from_file(const char *xml_name, saena_options_c *o);   /* saena::options(const string&), saena.cpp:444-546 */

typedef struct saena_amg_h saena_amg_h;
saena_amg_h *saena_amg_new(void);
void  saena_amg_free(saena_amg_h *S);
/* saena::amg::set_matrix (saena.hpp:202): smoothed-aggregation setup on the host */
int   saena_amg_set_matrix(saena_amg_h *S, saena_matrix_h *A, const saena_options_c *opts);
int   saena_amg_num_levels(saena_amg_h *S);                               /* max_level + 1 */
/* coarse id of every fine row of `level` (the reference's `aggregate` after aggregate_index_update, src/saena_object_setup1.cpp:
 * 2103-2260); one-rank setups keep it, for the pins of the setup (tests/test_sa_pins.py).  out: rows of that level */
int saena_amg_level_aggregates(saena_amg_h *, int level, index_t *out, index_t *n_aggregates);
int   saena_amg_level_info(saena_amg_h *S, int level, index_t *rows, nnz_t *nnzA, nnz_t *nnzP, double *eig_max);
int   saena_amg_level_split(saena_amg_h *S, int level, index_t *split_out /* nranks+1 */);   /* row partition of a level */
/* which: 0 = A_l, 1 = P_l, 2 = R_l (this rank's share when the communicator has more than one rank) */
int   saena_amg_level_desc(saena_amg_h *S, int level, int which, sgpu_op_desc *out);
/* [GPU] upload every level (sgpu_op_create) and build the device hierarchy (sgpu_amg_create) */
int   saena_amg_to_device(saena_amg_h *S);
sgpu_amg *saena_amg_device_handle(saena_amg_h *S);                        /* [GPU] valid after to_device */
sgpu_op  *saena_amg_device_op(saena_amg_h *S, int level, int which);      /* [GPU] */
/* [GPU] saena::amg::solve / solve_pCG (saena.hpp:220,224) on HOST slices of this rank: rhs in, u out */
int   saena_amg_solve(saena_amg_h *S, const value_t *rhs_host, value_t *u_host, int *iters, value_t *res_hist, int hist_cap);
int   saena_amg_solve_pCG(saena_amg_h *S, const value_t *rhs_host, value_t *u_host, int *iters, value_t *res_hist, int hist_cap);

#ifdef __cplusplus
}
#endif
#endif

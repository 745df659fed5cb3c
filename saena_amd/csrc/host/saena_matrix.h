// saena_matrix.h -- host-side distributed sparse operator (mirror of the
// reference's saena_matrix / prolong_matrix / restrict_matrix storage).
//
// Member names follow include/saena_matrix.h:75-149 of the reference so that a
// Saena maintainer can map them one to one; the arrays produced by
// set_off_on_diagonal() are exactly what sgpu_op_create() (include/saena_gpu.h)
// consumes.  This is product host code (the reference's setup is host code
// too); the HIP kernels never touch these structures directly.
#pragma once
#include "comm.h"

#include <functional>
#include <string>
#include <vector>

typedef int    index_t;   // include/data_struct.h:36
typedef long   nnz_t;     // include/data_struct.h:37
typedef double value_t;   // include/data_struct.h:38

#define SAENA_ALMOST_ZERO 1e-14   // data_struct.h:42

namespace saena_host {

// include/data_struct.h:116-124; ordering helpers below
struct cooEntry {
    index_t row, col;
    value_t val;
    cooEntry() = default;
    cooEntry(index_t i, index_t j, value_t v) : row(i), col(j), val(v) {}
};
inline bool col_major(const cooEntry &a, const cooEntry &b) { return a.col != b.col ? a.col < b.col : a.row < b.row; }
inline bool row_major(const cooEntry &a, const cooEntry &b) { return a.row != b.row ? a.row < b.row : a.col < b.col; }

// index of the block of `split` that holds val (aux_functions.h:39-58, lower_bound2)
long lower_bound2(const index_t *left, const index_t *right, index_t val);

// the reference's nnz-balanced contiguous row partition (saena_matrix_repart.cpp:43-170 for the fine operator, :728-980
// for every coarse one): see saena_matrix.cpp
std::vector<index_t> nnz_balanced_split(Comm &c, index_t Mbig, nnz_t nnz_g, int nparts,
                                        const std::function<void(const std::vector<index_t> &, std::vector<long> &)> &add_local_histogram,
                                        int min_buckets = 0);

// The halo plan + local/remote split shared by A, R and P.
struct DistLayout {
    index_t M = 0;                 // local rows
    index_t N_local = 0;           // local length of the input vector
    index_t col_offset = 0;        // split_col[rank]
    nnz_t   nnz_l_local = 0, nnz_l_remote = 0;
    index_t col_remote_size = 0;
    std::vector<index_t> nnzPerRow_local, col_local;   // (the reference also fills row_local; nothing reads it)
    std::vector<value_t> val_local;
    std::vector<index_t> nnzPerCol_remote, row_remote, col_remote, col_remote2, vElement_remote;
    std::vector<value_t> val_remote;
    std::vector<nnz_t>   nnzPerProcScan;
    std::vector<int>     recvCount, sendCount, recvProcRank, recvProcCount, sendProcRank, sendProcCount, vdispls, rdispls;
    int     numRecvProc = 0, numSendProc = 0;
    index_t vIndexSize = 0, recvSize = 0;
    std::vector<index_t> vIndex;

    // set_off_on_diagonal (src/saena_matrix_setup.cpp:793-1098), also
    // prolong_matrix::findLocalRemote (src/prolong_matrix.cpp:18-378) and the
    // layout half of restrict_matrix::transposeP (src/restrict_matrix.cpp:10-494).
    // entry: this rank's rows, column-major sorted, GLOBAL row and column ids.
    void build(Comm &comm, const std::vector<cooEntry> &entry, const std::vector<index_t> &split_row,
               const std::vector<index_t> &split_col);
    // the same from this rank's rows as CSR (global columns ascending inside a row): no column-major detour
    void build_from_csr(Comm &comm, const std::vector<nnz_t> &ptr, const std::vector<index_t> &col, const std::vector<value_t> &val,
                        const std::vector<index_t> &split_row, const std::vector<index_t> &split_col);
    void finish_plan(Comm &comm, const std::vector<index_t> &split_col);
    // one rank: everything is local, so the layout IS the row-major CSR (no exchange, no sort)
    void build_single_rank(index_t M_, index_t N_, const std::vector<nnz_t> &ptr, std::vector<index_t> &&col,
                           std::vector<value_t> &&val);
};

class saena_matrix {
public:
    Comm *comm = nullptr;
    index_t Mbig = 0, M = 0;
    nnz_t   nnz_g = 0, nnz_l = 0;
    std::vector<index_t> split;          // saena_matrix.h:95
    std::vector<cooEntry> entry;         // this rank's rows, column-major, global ids
    DistLayout L;
    std::vector<value_t> inv_diag;
    double eig_max_of_invdiagXA = 0;     // saena_matrix.h:183
    float  jacobi_omega = float(2.0 / 3);   // saena_matrix.h:182
    bool   add_duplicates = true;        // saena.hpp:47
    bool   remove_boundary = true;       // saena_matrix.h:101
    bool   assembled = false;
    // Resolution of the nnz-balanced row partition (repartition_nnz_initial).  0 = the reference's nparts^2 row buckets
    // (saena_matrix_repart.cpp:43-170: a part boundary can only sit at a multiple of Mbig / nparts^2 -- 7 to 9 of 64 buckets per
    // rank at 8 ranks, 3 to 5 of 16 at 4: max / mean rows 1.125 and 1.25 on a uniform operator); n > 0 = at least n buckets
    // (OPT-IN, not the reference's partition: SAENA_FINE_PARTITION_BUCKETS or saena_matrix_set_partition_buckets; the coarse
    // operators have used 4096 since round 3, host/amg_setup.cpp).  Same algorithm, finer histogram.
    int    partition_buckets = default_partition_buckets();
    static int default_partition_buckets();
    // boundary bookkeeping (remove_boundary_nodes, saena_matrix_setup.cpp:281-365)
    std::vector<index_t> bound_row_global;   // ALL removed rows (global, sorted), same on every rank
    index_t Mbig_with_bound = 0;

    explicit saena_matrix(Comm *c) : comm(c) {}

    int set(index_t row, index_t col, value_t val);                           // saena_matrix.cpp:459
    int set(const index_t *row, const index_t *col, const value_t *val, nnz_t n);
    // read_file (saena_matrix.cpp:17-385): MatrixMarket coordinate (.mtx; general / symmetric -> mirrored /
    // pattern -> value 1, mirrored; input_type "", "triangle", "pattern", "tripattern") or the 16-byte triple
    // binary (.bin: int32 row, int32 col, float64 val, 0-based).  Every rank keeps its nnz_g/nprocs chunk of the
    // column-major sorted entries, like the reference's MPI_File_read_at (:353-361); assemble() follows.
    int read_file(const std::string &filename, const std::string &input_type = "");
    // write this rank's assembled entries as 16-byte triples (the reference's .bin format, :198-203)
    int write_bin(const std::string &filename) const;
    // saena_matrix::writeMatrixToFile (saena_matrix.cpp:1205-1249): this rank's entries as "<name>-r<rank>.mtx",
    // 1-based, 12 significant digits; rank 0's file carries the MatrixMarket header
    int writeMatrixToFile(const std::string &name) const;
    // assemble (saena_matrix_setup.cpp:4): setup_initial_data + repartition_nnz_initial + matrix_setup
    int assemble();
    // skip the repartition and use the given row split (entries must already be global/deduplicated)
    int assemble_with_split(const std::vector<index_t> &split_in);
    void set_eig(double e) { eig_max_of_invdiagXA = e; }                       // saena.cpp:124-135

    // pieces of assemble(), public for tests
    void setup_initial_data();         // dedup + boundary removal (saena_matrix_setup.cpp:62-116,118-365)
    void repartition_nnz_initial();    // nnz-balanced split (saena_matrix_repart.cpp:3-325)
    void matrix_setup();               // inverse_diag + set_off_on_diagonal (saena_matrix_setup.cpp:507-560)
    void inverse_diag();               // saena_matrix_setup.cpp:1562-1600
    // one-rank shortcut used by the AMG setup for coarse operators: adopt a row-major CSR (columns
    // sorted within rows) as the assembled operator; `entry` stays empty.
    void setup_from_csr(index_t n, const std::vector<nnz_t> &ptr, std::vector<index_t> &&col, std::vector<value_t> &&val);

    // drop the rows of a full-length (with boundary) local vector slice that were removed
    // (remove_boundary_rhs, src/saena_object.cpp:699-730); rhs_with_bound covers [lo, hi) of the original numbering
    std::vector<value_t> remove_boundary_rhs(const std::vector<value_t> &rhs_with_bound, index_t lo) const;
    // set_repartition_rhs (src/saena_object_repart_shrink.cpp:154-218): entries (global index in the ORIGINAL
    // numbering, value) held by any rank -> this rank's slice of the assembled system (boundary rows dropped,
    // renumbered, moved to the owner by `split`).  Collective.
    std::vector<value_t> scatter_rhs(const std::vector<index_t> &idx_with_bound, const std::vector<value_t> &val) const;

private:
    std::vector<cooEntry> data_in;     // what set() collected (data_coo in the reference)
};

// Rectangular transfer operator holder (P: fine rows x coarse cols; R: coarse rows x fine cols)
struct transfer_matrix {
    Comm *comm = nullptr;
    index_t Mbig = 0, Nbig = 0, M = 0;
    nnz_t   nnz_g = 0, nnz_l = 0;
    std::vector<index_t> split_row, split_col;
    std::vector<cooEntry> entry;       // column-major, global ids
    DistLayout L;
    void build_layout() { L.build(*comm, entry, split_row, split_col); }
};

// R = P^T distributed by the coarse partition (restrict_matrix::transposeP, src/restrict_matrix.cpp:10-494)
void transpose_transfer(const transfer_matrix &P, transfer_matrix &R);

// ---- generators (input definitions) ----
// saena::laplacian3D (src/aux_functions2.cpp:254-373): every rank set()s its z-slabs
int laplacian3D(saena_matrix *A, index_t mx, index_t my, index_t mz);
// saena::laplacian3D_set_rhs (src/aux_functions2.cpp:629-700): this rank's slab of the full grid, with boundary;
// returns the global index of the first entry in *lo
std::vector<value_t> laplacian3D_set_rhs(Comm &comm, index_t mx, index_t my, index_t mz, index_t *lo);
// saena::band_matrix (src/aux_functions2.cpp:1296-1381): M rows per rank
int band_matrix(saena_matrix *A, index_t M, unsigned int bandwidth);

} // namespace saena_host

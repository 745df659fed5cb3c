// amg_setup.h -- host-side smoothed-aggregation setup.
//
// Restates saena_object::setup / coarsen / SA / compute_coarsen of the reference
// so that a hierarchy exists on the GPU box (where the reference does not): the
// V-cycle kernels consume its output (A_l, P_l, R_l per level).  Like the
// reference, setup is host code that runs once; it is not on the timed path.
#pragma once
#include "saena_matrix.h"

#include <memory>
#include <string>
#include <vector>

namespace saena_host {

// Optional accelerator for the setup's sparse products C = A B (CSR in, CSR out, rows sorted by column, the drop rule of
// the host kernel).  libsaena_amd.so installs its GPU kernel here when a device context exists; returns 0 on success,
// non-zero to decline (the host kernel then runs).  nullptr in libsaena_host.so.
// B's entries may come in TWO pieces (the rank's own rows, then the halo rows fetched for this product): entries [0, b_split) sit
// at b_col / b_val, entries [b_split, b_ptr[b_rows]) at b_col1 / b_val1 (b_col1 == nullptr: one piece) -- the caller then stacks
// the row pointers only, not the gigabytes of columns and values.
typedef int (*spgemm_hook_fn)(int a_rows, int b_rows, int b_cols, const long *a_ptr, const int *a_col, const double *a_val,
                              const long *b_ptr, const int *b_col, const double *b_val, long b_split, const int *b_col1, const double *b_val1,
                              int row_offset, std::vector<long> &c_ptr, std::vector<int> &c_col, std::vector<double> &c_val);
extern spgemm_hook_fn g_spgemm_hook;
// The exchange chain of one multi-rank apply in microseconds (pack -> RCCL send/recv -> boundary rows), as the GPU runtime
// MEASURED it on this job's communicator at sgpu_init (a ping-pong with the neighbouring rank; the maximum over the
// ranks, so every rank holds the same number); 0: not measured (host-only library, one rank).  The agglomeration model
// below starts from it instead of the constant; SAENA_SHRINK_CHAIN_US still overrides.
extern double g_measured_chain_us;

// saena::options (reference include/saena.hpp:127-193; defaults :151-155)
struct amg_options {
    int         solver_max_iter = 100;
    double      relative_tol    = 1e-8;
    std::string smoother        = "chebyshev";
    int         preSmooth       = 3;
    int         postSmooth      = 3;
    std::string PSmoother       = "jacobi";
    float       connStrength    = 0.3f;
    bool        dynamic_levels  = true;
    int         max_level       = 10;
    int         float_level     = 3;
    double      filter_thre     = 1e-14;
    double      filter_max      = 1e-8;
    int         filter_start    = 1;
    int         filter_rate     = 2;
    bool        switch_to_dense = false;
    float       dense_thre      = 0.1f;
    int         dense_sz_thre   = 5000;
    // parse the positional attributes of <SAENA><OPTIONS .../> (saena.cpp:444-546)
    void set_from_file(const std::string &name);
};

struct amg_level {
    saena_matrix *A = nullptr;                   // level 0: the caller's matrix; l > 0: owned by `Ac_store` of level l-1
    transfer_matrix P, R;                        // empty on the coarsest level
    std::unique_ptr<saena_matrix> Ac_store;      // A of level l+1
    std::vector<index_t> roots;                  // fine index of the root of every aggregate, ascending (= coarse numbering)
    std::vector<index_t> agg;                    // coarse id of every fine row (one-rank setup; the reference's `aggregate` after
                                                 // aggregate_index_update, setup1:2103-2260): kept for the setup's pins
};

// One rank's share of a hierarchy that was built at one rank and then row-partitioned.
struct dist_level {
    std::vector<index_t> split;                  // row partition of this level over the ranks
    DistLayout A, P, R;                          // P/R empty on the coarsest level
    std::vector<value_t> inv_diag;
    double eig_max = 0;
    index_t Mbig = 0;
    nnz_t nnzA = 0, nnzP = 0;
};

class amg_hierarchy {
public:
    amg_options opts;
    std::vector<amg_level> levels;               // levels.size() == max_level + 1 after setup
    int max_level = 0;
    // thresholds of saena_object.h:43-46
    unsigned int least_row_threshold = 100;
    double row_reduction_up_thrshld = 0.90;

    // saena_object::setup (saena_object.cpp:175-406)
    int setup(saena_matrix *A, const amg_options &o);

    // Multi-rank use: setup_distributed -> setup_rows_distributed, every rank builds its rows of every level (`dist`).
    // Partition per level: the fine split is A's; a coarse level inherits it through the aggregates (splitNew[r] =
    // number of roots below split[r], aggregate_index_update setup1:2115-2122); levels with <= shrink_rows rows live
    // entirely on rank 0 (the reference shrinks coarse levels onto fewer ranks, saena_matrix_shrink.cpp:167-265).
    // SAENA_SETUP=gathered selects the older form: gather the fine operator, build the whole hierarchy on every
    // rank (`levels`, `A_global`), keep the own rows (`distribute`).
    std::vector<dist_level> dist;
    std::unique_ptr<saena_matrix> A_global;      // the gathered fine operator (one-rank replica)
    std::unique_ptr<Comm> self_comm;
    // Agglomeration of coarse levels onto fewer ranks -- the reference's shrinking (saena_matrix::decide_shrinking /
    // decide_shrinking_c / shrink_set_params / shrink_cpu, src/saena_matrix_shrink.cpp:3-265, called from
    // compute_coarsen, src/saena_object_setup2.cpp:249-318) restated for one GPU per rank.  The reference times a dummy
    // matvec per level and shrinks by a factor 2..4 when communication exceeds twice the computation, and puts the
    // coarsest level on one rank.  Here the two times are a model with measured constants: one GPU applies the level in
    // T1 = 12 B x nnz / shrink_bw + shrink_launch_us, a multi-rank apply costs at least the exchange chain
    // shrink_chain_us (pack -> RCCL send/recv -> boundary rows: measured at sgpu_init on the job's communicator,
    // g_measured_chain_us; 23 us on one GPU with RCCL self send/recv, profiles/r01_halo_loopback.md and
    // profiles/r02_halo_loopback_modes.log, is the fall-back when nothing was measured).  Rules, evaluated after the coarse operator exists (its nnz is known):
    //   T1 <= chain                      -> the whole level lives on rank 0 (no exchange at all below this level);
    //   chain > 2 x T1 / active ranks    -> merge groups of f = clamp(floor(chain / compute / 5), 2, 4) consecutive active
    //                                       ranks onto the first of each group (shrink_set_params' rule: ranks k f own rows);
    //   rows <= shrink_rows              -> rank 0 (a row-count override, off by default).
    // SAENA_SHRINK_CHAIN_US / SAENA_SHRINK_ROWS override; chain 0 disables agglomeration.  The vectors need no
    // repartition step (Grid::repart_u / repart_back_u, src/grid.cpp:99-163): R's rows and P's columns are laid out on
    // the agglomerated partition, so R r lands where the coarse level lives and P e leaves from there -- the move rides
    // the transfer operators' own halo exchange.
    double  shrink_chain_us = 23.0, shrink_bw = 5e12, shrink_launch_us = 3.0;
    index_t shrink_rows = 0;
    std::vector<int> level_stride;               // per level: rank r owns rows iff r % stride == 0 (1 = all ranks, >= nranks = rank 0 only)
    int  next_stride(long nnzC, index_t rowsC, int np, int stride_prev) const;
    static std::vector<index_t> merge_split(const std::vector<index_t> &splitNew, int stride);
    int setup_distributed(saena_matrix *A_dist, const amg_options &o);
    // the distributed setup proper: every rank builds its rows of every level (fills `dist` only)
    int setup_rows_distributed(saena_matrix *A_dist, const amg_options &o);
    void distribute(Comm &c, const std::vector<index_t> &split0);

    // per-level facts, whichever way the hierarchy was built
    index_t level_rows(int l) const { return dist.empty() ? levels[(size_t)l].A->Mbig : dist[(size_t)l].Mbig; }
    nnz_t   level_nnzA(int l) const { return dist.empty() ? levels[(size_t)l].A->nnz_g : dist[(size_t)l].nnzA; }
    nnz_t   level_nnzP(int l) const { return l >= max_level ? 0 : dist.empty() ? levels[(size_t)l].P.nnz_g : dist[(size_t)l].nnzP; }
    double  level_eig(int l) const {
        const bool have = (size_t)l < levels.size() && levels[(size_t)l].A != nullptr;
        return (dist.empty() || have) ? levels[(size_t)l].A->eig_max_of_invdiagXA : dist[(size_t)l].eig_max;
    }

    // pieces, public for tests ------------------------------------------------
    // create_strength_matrix + strength_matrix::setup_matrix (setup1:520-719, strength_matrix.cpp:233-453):
    // CSR of the strong connections (local rows, global columns), diagonal included
    static void strength_graph(const saena_matrix &A, float connStrength, std::vector<nnz_t> &ptr, std::vector<index_t> &col);
    // aggregation_1_dist + aggregate_index_update (setup1:724-995, :2103-2260); returns the number of aggregates
    static index_t aggregate(const saena_matrix &A, const std::vector<nnz_t> &ptr, const std::vector<index_t> &col,
                             std::vector<index_t> &agg, std::vector<index_t> *roots = nullptr);
    // saena_object::matmat (saena_object_setup_matmat.cpp:1164-1487): C = A B for two assembled one-rank matrices;
    // products with |v| <= 1e-14 are dropped unless on the diagonal, the rule of the reference's SpGEMM output.
    // Returns the entries of C (row, col, val), row-major.
    static std::vector<cooEntry> matmat(const saena_matrix &A, const saena_matrix &B);
    // find_eig (saena_object.cpp:572-592, lamlan_saena.h): largest eigenvalue of D^-1 A by 20 Lanczos steps, x 1.0001
    static double find_eig(const saena_matrix &A);

private:
    double filter_thre_cur = 0;
    int    filter_it = 0;
    int  coarsen(int l);                          // saena_object::coarsen (saena_object.cpp:409-452)
    void filter(std::vector<cooEntry> &v, index_t sz, index_t ofst);   // setup2:852-916
};

} // namespace saena_host

// ref_sa.cpp -- vector-level fixtures of the SMOOTHED-AGGREGATION SETUP from the COMPILED REFERENCE.
//
// TEST INFRASTRUCTURE ONLY (our own driver, compiled together with the reference's sources where they lie under
// /root/reference; see Makefile; outputs go to oracle/_ref/).  saena_object::SA (src/saena_object_setup1.cpp:8-254) and
// what it calls -- find_aggregation (:255-432), create_strength_matrix (:520-719), aggregation_1_dist (:724-995),
// aggregate_index_update (:2103-2260), strength_matrix::setup_matrix (src/strength_matrix.cpp:233-453) -- compile from
// saena_object_setup1.cpp + strength_matrix.cpp with the reference's own vendored SuperLU_DIST headers on the include path
// (saena_object.h includes superlu_ddefs.h) and link against the operator objects the other drivers use: no SuperLU, MKL or
// ParMETIS LIBRARY is needed because nothing on this path calls into them, and saena_object's constructor and destructor
// are `= default` (include/saena_object.h:244-245).  No stand-in header or library is involved.
//
// For every level l of a hierarchy (A_l as a coordinate list, written by oracle/ref/make_golden_sa.py) the reference
// assembles A_l itself (set + assemble: its own partition) and runs
//     find_aggregation(A_l, aggregate, splitNew)   -> coarse id of every fine row, coarse partition, return value
//     SA(&grid)                                    -> the smoothed prolongation P_l (grid.P.entry)
// usage: mpirun -np P ref_sa <indir> <outdir> <tag> <nlevels> <connStrength>
//   <indir>/A<l>.coo : int32 M, int32 N, int64 nnz, then nnz x (int32 row, int32 col, float64 val)
#include "saena_object.h"
#include "saena_matrix.h"
#include "prolong_matrix.h"
#include "grid.h"

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

static std::string g_out;
static int g_rank = 0, g_np = 1;

template <class T>
static void write_raw(const std::string &name, const char *dtype, const std::vector<T> &v) {
    if (g_rank != 0) return;
    const std::string fn = g_out + "/" + name + "." + dtype;
    FILE *f = fopen(fn.c_str(), "wb");
    if (!f) { perror(fn.c_str()); MPI_Abort(MPI_COMM_WORLD, 2); }
    if (!v.empty()) fwrite(v.data(), sizeof(T), v.size(), f);
    fclose(f);
}
template <class T>
static std::vector<T> gather_bytes(const std::vector<T> &x) {
    int n = (int)(x.size() * sizeof(T));
    std::vector<int> cnt(g_np), dsp(g_np);
    MPI_Gather(&n, 1, MPI_INT, cnt.data(), 1, MPI_INT, 0, MPI_COMM_WORLD);
    int tot = 0;
    for (int i = 0; i < g_np; ++i) { dsp[i] = tot; tot += cnt[i]; }
    std::vector<T> all(g_rank == 0 ? (size_t)tot / sizeof(T) : 0);
    MPI_Gatherv(x.data(), n, MPI_BYTE, all.data(), cnt.data(), dsp.data(), MPI_BYTE, 0, MPI_COMM_WORLD);
    return all;
}

struct Coo { int M = 0, N = 0; std::vector<int> r, c; std::vector<double> v; };
static Coo read_coo(const std::string &fn) {
    Coo o;
    FILE *f = fopen(fn.c_str(), "rb");
    if (!f) { perror(fn.c_str()); MPI_Abort(MPI_COMM_WORLD, 3); }
    long nnz = 0;
    if (fread(&o.M, 4, 1, f) != 1 || fread(&o.N, 4, 1, f) != 1 || fread(&nnz, 8, 1, f) != 1) MPI_Abort(MPI_COMM_WORLD, 4);
    o.r.resize(nnz); o.c.resize(nnz); o.v.resize(nnz);
    for (long k = 0; k < nnz; ++k)
        if (fread(&o.r[k], 4, 1, f) != 1 || fread(&o.c[k], 4, 1, f) != 1 || fread(&o.v[k], 8, 1, f) != 1) MPI_Abort(MPI_COMM_WORLD, 5);
    fclose(f);
    return o;
}

struct Trip { int row, col; double val; };

int main(int argc, char **argv) {
    MPI_Init(&argc, &argv);
    MPI_Comm comm = MPI_COMM_WORLD;
    MPI_Comm_rank(comm, &g_rank);
    MPI_Comm_size(comm, &g_np);
    if (argc < 6) { if (!g_rank) fprintf(stderr, "usage: ref_sa indir outdir tag nlevels connStrength\n"); MPI_Finalize(); return 1; }
    const std::string in = argv[1];
    g_out = argv[2];
    const std::string pfx = std::string(argv[3]) + ".np" + std::to_string(g_np) + ".";
    const int nl = atoi(argv[4]);
    const float conn = (float)atof(argv[5]);

    for (int l = 0; l + 1 < nl; ++l) {
        const Coo c = read_coo(in + "/A" + std::to_string(l) + ".coo");
        saena_matrix A(comm);
        const long nnz = (long)c.r.size(), per = nnz / g_np;
        const long lo = g_rank * per, hi = g_rank == g_np - 1 ? nnz : lo + per;       // any rank may set any entry
        for (long k = lo; k < hi; ++k) A.set(c.r[k], c.c[k], c.v[k]);
        A.assemble(false);
        if (A.Mbig != c.M) { if (!g_rank) fprintf(stderr, "level %d: assemble changed the size (%d -> %d)\n", l, c.M, (int)A.Mbig); MPI_Abort(comm, 8); }
        const std::string L = std::to_string(l);
        write_raw(pfx + "split" + L, "i32", std::vector<int>(A.split.begin(), A.split.end()));

        saena_object so;                       // defaults of include/saena_object.h; the options file's conn_str is the only one SA reads here
        so.connStrength = conn;
        so.PSmoother = "jacobi";
        // (1) the aggregation alone: coarse id of every local fine row after aggregate_index_update
        {
            std::vector<index_t> aggregate(A.M), splitNew;
            const int ret = so.find_aggregation(&A, aggregate, splitNew);
            const std::vector<int> all = gather_bytes(std::vector<int>(aggregate.begin(), aggregate.end()));
            write_raw(pfx + "agg" + L, "i32", all);
            write_raw(pfx + "splitNew" + L, "i32", std::vector<int>(splitNew.begin(), splitNew.end()));
            write_raw(pfx + "ret" + L, "i32", std::vector<int>(1, ret));
        }
        // (2) SA: the smoothed prolongation
        {
            Grid grid(&A, l);
            const int ret = so.SA(&grid);
            (void)ret;
            std::vector<Trip> mine;
            // prolong_matrix::findLocalRemote (src/prolong_matrix.cpp:18-378), called at the end of SA, has moved `entry` into
            // the layout: local entries (local row, GLOBAL column, value) and remote entries (local row, GLOBAL column as
            // pushed at :80, value) -- vElement_remote is cleared once the halo plan exists (:270)
            const prolong_matrix &Pm = grid.P;
            const int r0 = (int)A.split[g_rank];
            for (size_t k = 0; k < Pm.row_local.size(); ++k) mine.push_back(Trip{(int)Pm.row_local[k] + r0, (int)Pm.col_local[k], (double)Pm.val_local[k]});
            for (size_t k = 0; k < Pm.row_remote.size(); ++k) mine.push_back(Trip{(int)Pm.row_remote[k] + r0, (int)Pm.col_remote[k], (double)Pm.val_remote[k]});
            if (Pm.col_remote.size() != Pm.row_remote.size() || mine.size() != (size_t)Pm.nnz_l) {
                fprintf(stderr, "rank %d: P layout does not add up: %zu local + %zu remote vs nnz_l %ld\n", g_rank, Pm.row_local.size(), Pm.row_remote.size(), (long)Pm.nnz_l);
                MPI_Abort(comm, 9);
            }
            std::vector<Trip> all = gather_bytes(mine);
            if (g_rank == 0) {
                std::sort(all.begin(), all.end(), [](const Trip &a, const Trip &b) { return a.row != b.row ? a.row < b.row : a.col < b.col; });
                std::vector<int> r, cc; std::vector<double> v;
                for (const auto &t : all) { r.push_back(t.row); cc.push_back(t.col); v.push_back(t.val); }
                write_raw(pfx + "Prow" + L, "i32", r);
                write_raw(pfx + "Pcol" + L, "i32", cc);
                write_raw(pfx + "Pval" + L, "f64", v);
                write_raw(pfx + "Pshape" + L, "i32", std::vector<int>{(int)grid.P.Mbig, (int)grid.P.Nbig});
            }
        }
        MPI_Barrier(comm);
    }
    if (!g_rank) printf("REF_SA_OK %s np=%d levels=%d\n", argv[3], g_np, nl);
    MPI_Finalize();
    return 0;
}

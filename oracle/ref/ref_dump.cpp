// ref_dump.cpp -- fixture generator that drives the COMPILED REFERENCE operators.
//
// TEST INFRASTRUCTURE ONLY.  This file is our own driver; it is compiled
// together with the reference's own sources where they lie under
// /root/reference (see Makefile in this directory; outputs go to
// oracle/_ref/, never into the repository).  It feeds closed-form inputs to
// saena_matrix / prolong_matrix / restrict_matrix and writes their outputs as
// raw little-endian arrays; oracle/ref/make_golden.py packs those into
// tests/golden/*.npz.
//
// usage: mpirun -np P ref_dump <outdir> poisson <m>  [full]
//        mpirun -np P ref_dump <outdir> band <M> <bw>
//        mpirun -np P ref_dump <outdir> norms <m>          (norm pins only, large m)
//        mpirun -np P ref_dump <outdir> file <matrix.mtx|.bin> <tag>   (one of the reference's data files)
//        mpirun -np P ref_dump <outdir> time <m> <seconds>   (bench.py cpu_baseline: times saena_matrix::matvec,
//                                                              prints "REF_TIME_MATVEC <s per matvec> <reps> <ranks>")
//
// Reference interfaces exercised (file:line in the reference checkout):
//   saena_matrix::set/assemble            src/saena_matrix.cpp:459, src/saena_matrix_setup.cpp:4
//   saena_matrix::matvec(_sparse_float)   src/saena_matrix_matvec.cpp:9-113, :448-550
//   saena_matrix::jacobi / chebyshev      src/saena_matrix.cpp:1044-1131
//   saena_matrix::residual                include/saena_matrix.tpp:16-23
//   prolong_matrix::findLocalRemote/matvec   src/prolong_matrix.cpp:18-378, :489-624
//   restrict_matrix::transposeP/matvec    src/restrict_matrix.cpp:10-494, :612-744
//   saena_matrix_dense::convert_saena_matrix / matvec_dense(_float)   src/saena_matrix_dense.cpp:763-793, :181-340
#include "saena_matrix.h"
#include "prolong_matrix.h"
#include "restrict_matrix.h"
#include "saena_matrix_dense.h"

#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

static std::string g_out;
static int g_rank = 0, g_np = 1;

template <class T>
static void write_raw(const std::string &name, const char *dtype, const std::vector<T> &v) {
    if (g_rank != 0) return;
    std::string fn = g_out + "/" + name + "." + dtype;
    FILE *f = fopen(fn.c_str(), "wb");
    if (!f) { perror(fn.c_str()); MPI_Abort(MPI_COMM_WORLD, 2); }
    if (!v.empty()) fwrite(v.data(), sizeof(T), v.size(), f);
    fclose(f);
}

// gather a distributed double vector (local length n) to rank 0
static std::vector<double> gather_d(const double *x, int n) {
    std::vector<int> cnt(g_np), dsp(g_np);
    MPI_Gather(&n, 1, MPI_INT, cnt.data(), 1, MPI_INT, 0, MPI_COMM_WORLD);
    int tot = 0;
    for (int i = 0; i < g_np; ++i) { dsp[i] = tot; tot += cnt[i]; }
    std::vector<double> all(g_rank == 0 ? tot : 0);
    MPI_Gatherv(x, n, MPI_DOUBLE, all.data(), cnt.data(), dsp.data(), MPI_DOUBLE, 0, MPI_COMM_WORLD);
    return all;
}
static std::vector<int> gather_i(const int *x, int n) {
    std::vector<int> cnt(g_np), dsp(g_np);
    MPI_Gather(&n, 1, MPI_INT, cnt.data(), 1, MPI_INT, 0, MPI_COMM_WORLD);
    int tot = 0;
    for (int i = 0; i < g_np; ++i) { dsp[i] = tot; tot += cnt[i]; }
    std::vector<int> all(g_rank == 0 ? tot : 0);
    MPI_Gatherv(x, n, MPI_INT, all.data(), cnt.data(), dsp.data(), MPI_INT, 0, MPI_COMM_WORLD);
    return all;
}
static double gsum(double x) { double r; MPI_Allreduce(&x, &r, 1, MPI_DOUBLE, MPI_SUM, MPI_COMM_WORLD); return r; }

// 7-point Poisson on an m^3 node grid, generated rank-wise in z-slabs the way
// the reference's driver does (entries are a closed form of the node indices).
static void fill_poisson(saena_matrix &A, int mx, int my, int mz) {
    if (g_rank >= mz) return;
    const double Hx = 1.0 / (mx - 1), Hy = 1.0 / (my - 1), Hz = 1.0 / (mz - 1);
    const double cx = 1.0 / (Hx * Hx), cy = 1.0 / (Hy * Hy), cz = 1.0 / (Hz * Hz);
    int zm, zs;
    if (mz > g_np) { zm = mz / g_np; zs = g_rank * zm; if (g_rank == g_np - 1) zm = mz - (g_np - 1) * zm; }
    else { zm = 1; zs = g_rank; }
    for (int k = zs; k < zs + zm; ++k)
        for (int j = 0; j < my; ++j)
            for (int i = 0; i < mx; ++i) {
                const int node = mx * my * k + mx * j + i;
                if (i == 0 || j == 0 || k == 0 || i == mx - 1 || j == my - 1 || k == mz - 1) { A.set(node, node, 1.0); continue; }
                if (k - 1 != 0) A.set(node, node - mx * my, -cz);
                if (j - 1 != 0) A.set(node, node - mx, -cy);
                if (i - 1 != 0) A.set(node, node - 1, -cx);
                A.set(node, node, 2.0 * (cz + cy + cx));
                if (i + 1 != mx - 1) A.set(node, node + 1, -cx);
                if (j + 1 != my - 1) A.set(node, node + mx, -cy);
                if (k + 1 != mz - 1) A.set(node, node + mx * my, -cz);
            }
}

static void fill_band(saena_matrix &A, int Mbig, int bw) {
    const int per = Mbig / g_np;
    const int lo = g_rank * per, hi = (g_rank == g_np - 1) ? Mbig : lo + per;
    for (int i = lo; i < hi; ++i)
        for (int j = i - bw; j <= i + bw; ++j)
            if (j >= 0 && j < Mbig) A.set(i, j, 1.0 / (i + j + 1));
}

static void dump_layout(saena_matrix &A, const std::string &pfx) {
    // every public array of the hot-path layout, concatenated over ranks, plus per-rank sizes
    std::vector<int> sizes = {(int)A.M, (int)A.nnz_l_local, (int)A.nnz_l_remote, (int)A.col_remote_size,
                              (int)A.vIndexSize, (int)A.recvSize, (int)A.numRecvProc, (int)A.numSendProc};
    write_raw(pfx + "sizes", "i32", gather_i(sizes.data(), (int)sizes.size()));
    write_raw(pfx + "nnzPerRow_local", "i32", gather_i(A.nnzPerRow_local.data(), (int)A.nnzPerRow_local.size()));
    write_raw(pfx + "col_local", "i32", gather_i(A.col_local, (int)A.nnz_l_local));
    write_raw(pfx + "val_local", "f64", gather_d(A.val_local, (int)A.nnz_l_local));
    write_raw(pfx + "nnzPerCol_remote", "i32", gather_i(A.nnzPerCol_remote.data(), (int)A.nnzPerCol_remote.size()));
    write_raw(pfx + "row_remote", "i32", gather_i(A.row_remote, (int)A.nnz_l_remote));
    write_raw(pfx + "col_remote", "i32", gather_i(A.col_remote, (int)A.nnz_l_remote));
    write_raw(pfx + "val_remote", "f64", gather_d(A.val_remote, (int)A.nnz_l_remote));
    write_raw(pfx + "vIndex", "i32", gather_i(A.vIndex.data(), (int)A.vIndex.size()));
    write_raw(pfx + "recvProcRank", "i32", gather_i(A.recvProcRank.data(), (int)A.recvProcRank.size()));
    write_raw(pfx + "recvProcCount", "i32", gather_i(A.recvProcCount.data(), (int)A.recvProcCount.size()));
    write_raw(pfx + "sendProcRank", "i32", gather_i(A.sendProcRank.data(), (int)A.sendProcRank.size()));
    write_raw(pfx + "sendProcCount", "i32", gather_i(A.sendProcCount.data(), (int)A.sendProcCount.size()));
    write_raw(pfx + "inv_diag", "f64", gather_d(A.inv_diag, (int)A.M));
}

int main(int argc, char **argv) {
    MPI_Init(&argc, &argv);
    MPI_Comm comm = MPI_COMM_WORLD;
    MPI_Comm_rank(comm, &g_rank);
    MPI_Comm_size(comm, &g_np);
    if (argc < 4) { if (!g_rank) fprintf(stderr, "usage: ref_dump outdir poisson|band|norms ...\n"); MPI_Finalize(); return 1; }
    g_out = argv[1];
    const std::string kind = argv[2];
    const bool norms_only = (kind == "norms");
    const bool timing = (kind == "time");

    saena_matrix A(comm);
    std::string tag;
    if (kind == "poisson" || norms_only || timing) {
        const int m = atoi(argv[3]);
        fill_poisson(A, m, m, m);
        tag = "poisson" + std::to_string(m);
    } else if (kind == "band") {
        const int M = atoi(argv[3]), bw = atoi(argv[4]);
        fill_band(A, M, bw);
        tag = "band" + std::to_string(M) + "_" + std::to_string(bw);
    } else if (kind == "file") {                 // saena_matrix::read_file (src/saena_matrix.cpp:17-385) on one of the reference's data files
        A.read_file(argv[3]);
        tag = argv[4];
    } else { MPI_Finalize(); return 1; }
    const bool light = (kind == "file");         // vectors only: the layout arrays of these matrices are too big to commit
    A.assemble(false);
    const std::string pfx = tag + ".np" + std::to_string(g_np) + ".";

    const int M = A.M, ofs = A.split[g_rank];
    std::vector<double> v(M), v2(M), w(M), rhs1(M, 1.0), rhs2(M), u(M);
    for (int i = 0; i < M; ++i) {
        const double g = ofs + i;
        v[i]    = sin(0.001 * g);
        v2[i]   = sin(0.37 * g + 0.1) + 0.25 * cos(1.3 * g);
        rhs2[i] = cos(0.05 * g) - 0.3;
    }
    auto norm2 = [&](const std::vector<double> &x) { double s = 0; for (double t : x) s += t * t; return gsum(s); };

    if (timing) {   // the reference's own matvec (src/saena_matrix_matvec.cpp:9-113) on this box's cores, one rank per core
        const double budget = argc > 4 ? atof(argv[4]) : 10.0;
        for (int i = 0; i < 3; ++i) A.matvec(v.data(), w.data());
        MPI_Barrier(comm);
        double t0 = MPI_Wtime();
        for (int i = 0; i < 5; ++i) A.matvec(v.data(), w.data());
        MPI_Barrier(comm);
        double per = (MPI_Wtime() - t0) / 5;
        MPI_Bcast(&per, 1, MPI_DOUBLE, 0, comm);
        int reps = (int)(budget / (per > 1e-7 ? per : 1e-7));
        if (reps < 5) reps = 5;
        MPI_Barrier(comm);
        t0 = MPI_Wtime();
        for (int i = 0; i < reps; ++i) A.matvec(v.data(), w.data());
        MPI_Barrier(comm);
        per = (MPI_Wtime() - t0) / reps;
        const double nrm = norm2(w);
        if (!g_rank) printf("REF_TIME_MATVEC %.9e %d %d %.17g %ld %ld\n", per, reps, g_np, nrm, (long)A.Mbig, (long)A.nnz_g);
        MPI_Finalize();
        return 0;
    }

    std::vector<double> pins;   // squared norms, SURVEY 8c style
    A.matvec(v.data(), w.data());
    pins.push_back(norm2(w));
    if (!norms_only) write_raw(pfx + "Av", "f64", gather_d(w.data(), M));

    std::fill(u.begin(), u.end(), 0.0);
    A.jacobi(3, u.data(), rhs1.data());
    pins.push_back(norm2(u));
    if (!norms_only) write_raw(pfx + "jacobi3_rhs1", "f64", gather_d(u.data(), M));

    A.set_eig(2.0);
    std::fill(u.begin(), u.end(), 0.0);
    A.chebyshev(3, u.data(), rhs1.data());
    pins.push_back(norm2(u));
    if (!norms_only) write_raw(pfx + "cheby3_rhs1", "f64", gather_d(u.data(), M));
    write_raw(pfx + "pins", "f64", pins);

    if (!norms_only) {
        write_raw(pfx + "split", "i32", std::vector<int>(A.split.begin(), A.split.end()));
        std::vector<long> meta = {(long)A.Mbig, (long)A.nnz_g, (long)g_np};
        write_raw(pfx + "meta", "i64", meta);
        if (!light) dump_layout(A, pfx);

        A.matvec(v2.data(), w.data());
        write_raw(pfx + "Av2", "f64", gather_d(w.data(), M));

        double *res = saena_aligned_alloc<value_t>(M);
        A.residual(v2.data(), rhs2.data(), res);
        write_raw(pfx + "residual_v2_rhs2", "f64", gather_d(res, M));
        saena_free(res);

        u = v2;
        A.jacobi(2, u.data(), rhs2.data());
        write_raw(pfx + "jacobi2_v2_rhs2", "f64", gather_d(u.data(), M));

        A.set_eig(1.9371);
        u = v2;
        A.chebyshev(4, u.data(), rhs2.data());
        write_raw(pfx + "cheby4_v2_rhs2", "f64", gather_d(u.data(), M));
        u = v2;
        A.chebyshev(1, u.data(), rhs2.data());
        write_raw(pfx + "cheby1_v2_rhs2", "f64", gather_d(u.data(), M));

        A.matvec_sparse_float(v2.data(), w.data());
        write_raw(pfx + "Av2_float", "f64", gather_d(w.data(), M));

        if (A.Mbig <= 4000) {     // the reference's dense storage of the same operator (`switch_to_dense`), ring GEMV
            saena_matrix_dense D;
            D.convert_saena_matrix(&A);
            D.matvec_dense(v2.data(), w.data());
            write_raw(pfx + "Av2_dense", "f64", gather_d(w.data(), M));
            D.matvec_dense_float(v2.data(), w.data());
            write_raw(pfx + "Av2_dense_float", "f64", gather_d(w.data(), M));
        }

        if (light) {
            if (!g_rank) printf("%s np=%d Mbig=%d nnz_g=%ld  |Av|^2=%.16g\n", tag.c_str(), g_np, (int)A.Mbig, (long)A.nnz_g, pins[0]);
            MPI_Finalize();
            return 0;
        }
        // ---- grid transfer: synthetic closed-form P (fine rows x coarse cols) ----
        // P(i,j) = 1/(1+|i-2j|) + 0.001*i for j in {i/2-1, i/2, i/2+1} (global ids), clipped.
        const int Mbig = A.Mbig, Nc = (Mbig + 1) / 2;
        prolong_matrix P(comm);
        P.Mbig = Mbig; P.Nbig = Nc; P.M = M;
        P.split = A.split;
        P.splitNew.resize(g_np + 1);
        for (int r = 0; r < g_np; ++r) P.splitNew[r] = A.split[r] / 2;
        P.splitNew[g_np] = Nc;
        for (int i = 0; i < M; ++i) {
            const int gi = ofs + i;
            for (int j = gi / 2 - 1; j <= gi / 2 + 1; ++j)
                if (j >= 0 && j < Nc) P.entry.emplace_back(i, j, 1.0 / (1 + abs(gi - 2 * j)) + 0.001 * gi);
        }
        std::sort(P.entry.begin(), P.entry.end());          // cooEntry order = column-major
        P.nnz_l = P.entry.size();
        MPI_Allreduce(&P.nnz_l, &P.nnz_g, 1, MPI_LONG, MPI_SUM, comm);
        write_raw(pfx + "splitNew", "i32", std::vector<int>(P.splitNew.begin(), P.splitNew.end()));
        P.findLocalRemote();
        restrict_matrix R;
        R.transposeP(&P);

        const int Mc = P.splitNew[g_rank + 1] - P.splitNew[g_rank], ofc = P.splitNew[g_rank];
        std::vector<double> ec(Mc), ef(M), rc(Mc);
        for (int i = 0; i < Mc; ++i) ec[i] = sin(0.21 * (ofc + i) + 0.4);
        P.matvec(ec.data(), ef.data());
        write_raw(pfx + "P_ec", "f64", gather_d(ef.data(), M));
        R.matvec(v2.data(), rc.data());
        write_raw(pfx + "R_v2", "f64", gather_d(rc.data(), Mc));
    }

    if (!g_rank) printf("%s np=%d Mbig=%d nnz_g=%ld  |Av|^2=%.16g  |jacobi3|^2=%.16g  |cheby3|^2=%.16g\n",
                        tag.c_str(), g_np, (int)A.Mbig, (long)A.nnz_g, pins[0], pins[1], pins[2]);
    MPI_Finalize();
    return 0;
}

// ref_vcycle.cpp -- vector-level fixtures of the grid transfers and of a whole V-cycle from the COMPILED REFERENCE
// operators, on a real smoothed-aggregation hierarchy.
//
// TEST INFRASTRUCTURE ONLY (our own driver, compiled together with the reference's sources where they lie under
// /root/reference; see Makefile; outputs go to oracle/_ref/).  The full reference saena_object (setup + vcycle) cannot
// be built here without stand-in headers, so the V-cycle is COMPOSED from the reference's own operator classes in the
// order of saena_object::vcycle (src/saena_object_solve.cpp:1105-1399):
//     smooth (saena_matrix::jacobi / chebyshev, src/saena_matrix.cpp:1044-1131)
//     res = A u - rhs (saena_matrix::residual, include/saena_matrix.tpp:16-23)
//     res_coarse = R res (restrict_matrix::matvec, src/restrict_matrix.cpp:612-744)
//     uCorrCoarse = 0; vcycle(coarse)
//     uCorr = P uCorrCoarse (prolong_matrix::matvec, src/prolong_matrix.cpp:489-624);  u -= uCorr (:1360-1361)
//     smooth
// with the coarsest level solved by the loop of solve_coarsest_CG (:14-114) over the reference's matvec, and the outer
// loops of solve_pCG (:2389-2801) and solve (:1883-2014) composed the same way (residual histories and solutions).
// The hierarchy (A_l, P_l as coordinate lists) is an INPUT, written by oracle/ref/make_golden_vcycle.py from the
// product's host setup; R_l is built by the reference itself (restrict_matrix::transposeP, src/restrict_matrix.cpp:10-494)
// from the reference's own layout of P_l (prolong_matrix::findLocalRemote, src/prolong_matrix.cpp:18-378).
//
// usage: mpirun -np P ref_vcycle <indir> <outdir> <tag> <nlevels>
//   <indir>/A<l>.coo, P<l>.coo : int32 M, int32 N, int64 nnz, then nnz x (int32 row, int32 col, float64 val)
//   <indir>/eig.txt            : eig_max_of_invdiagXA per level (one per line)
#include "saena_matrix.h"
#include "prolong_matrix.h"
#include "restrict_matrix.h"

#include <cmath>
#include <cstdio>
#include <cstring>
#include <functional>
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
static std::vector<double> gather_d(const double *x, int n) {
    std::vector<int> cnt(g_np), dsp(g_np);
    MPI_Gather(&n, 1, MPI_INT, cnt.data(), 1, MPI_INT, 0, MPI_COMM_WORLD);
    int tot = 0;
    for (int i = 0; i < g_np; ++i) { dsp[i] = tot; tot += cnt[i]; }
    std::vector<double> all(g_rank == 0 ? tot : 0);
    MPI_Gatherv(x, n, MPI_DOUBLE, all.data(), cnt.data(), dsp.data(), MPI_DOUBLE, 0, MPI_COMM_WORLD);
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
    for (long k = 0; k < nnz; ++k) {
        if (fread(&o.r[k], 4, 1, f) != 1 || fread(&o.c[k], 4, 1, f) != 1 || fread(&o.v[k], 8, 1, f) != 1) MPI_Abort(MPI_COMM_WORLD, 5);
    }
    fclose(f);
    return o;
}

// closed-form test vectors by global index (tests/inputs.py holds the same formulas)
static double f_v2(double g) { return sin(0.37 * g + 0.1) + 0.25 * cos(1.3 * g); }
static double f_rhs2(double g) { return cos(0.05 * g) - 0.3; }
static double f_ec(double g) { return sin(0.21 * g + 0.4); }

int main(int argc, char **argv) {
    MPI_Init(&argc, &argv);
    MPI_Comm comm = MPI_COMM_WORLD;
    MPI_Comm_rank(comm, &g_rank);
    MPI_Comm_size(comm, &g_np);
    if (argc < 5) { if (!g_rank) fprintf(stderr, "usage: ref_vcycle indir outdir tag nlevels\n"); MPI_Finalize(); return 1; }
    const std::string in = argv[1];
    g_out = argv[2];
    const std::string pfx = std::string(argv[3]) + ".np" + std::to_string(g_np) + ".";
    const int nl = atoi(argv[4]);

    std::vector<double> eig(nl, 0.0);
    {
        FILE *f = fopen((in + "/eig.txt").c_str(), "r");
        if (!f) { perror("eig.txt"); MPI_Abort(comm, 6); }
        for (int l = 0; l < nl; ++l) if (fscanf(f, "%lf", &eig[l]) != 1) MPI_Abort(comm, 7);
        fclose(f);
    }

    // ---- operators: A_l through the reference's own set/assemble (nnz-balanced partition per level) ----
    std::vector<std::unique_ptr<saena_matrix>> A;
    for (int l = 0; l < nl; ++l) {
        const Coo c = read_coo(in + "/A" + std::to_string(l) + ".coo");
        A.emplace_back(new saena_matrix(comm));
        const long nnz = (long)c.r.size(), per = nnz / g_np;
        const long lo = g_rank * per, hi = g_rank == g_np - 1 ? nnz : lo + per;       // any rank may set any entry
        for (long k = lo; k < hi; ++k) A[l]->set(c.r[k], c.c[k], c.v[k]);
        A[l]->assemble(false);
        if (A[l]->Mbig != c.M) { if (!g_rank) fprintf(stderr, "level %d: assemble changed the size (%d -> %d)\n", l, c.M, (int)A[l]->Mbig); MPI_Abort(comm, 8); }
        A[l]->set_eig(eig[l]);
        write_raw(pfx + "split" + std::to_string(l), "i32", std::vector<int>(A[l]->split.begin(), A[l]->split.end()));
    }
    // ---- P_l on (A_l.split, A_{l+1}.split), R_l = transposeP ----
    std::vector<std::unique_ptr<prolong_matrix>> P;
    std::vector<std::unique_ptr<restrict_matrix>> R;
    for (int l = 0; l + 1 < nl; ++l) {
        const Coo c = read_coo(in + "/P" + std::to_string(l) + ".coo");
        P.emplace_back(new prolong_matrix(comm));
        prolong_matrix &p = *P[l];
        p.Mbig = A[l]->Mbig; p.Nbig = A[l + 1]->Mbig; p.M = A[l]->M;
        p.split = A[l]->split; p.splitNew = A[l + 1]->split;
        const int lo = A[l]->split[g_rank], hi = A[l]->split[g_rank + 1];
        for (size_t k = 0; k < c.r.size(); ++k)
            if (c.r[k] >= lo && c.r[k] < hi) p.entry.emplace_back(c.r[k] - lo, c.c[k], c.v[k]);      // local row, global column
        std::sort(p.entry.begin(), p.entry.end());
        p.nnz_l = p.entry.size();
        MPI_Allreduce(&p.nnz_l, &p.nnz_g, 1, MPI_LONG, MPI_SUM, comm);
        p.findLocalRemote();
        R.emplace_back(new restrict_matrix());
        R[l]->transposeP(&p);
    }

    // ---- transfers, fp64 and fp32-halo forms (restrict_matrix.cpp:746-871, prolong_matrix.cpp:626-758) ----
    for (int l = 0; l + 1 < nl; ++l) {
        const int Mf = A[l]->M, of = A[l]->split[g_rank], Mc = A[l + 1]->M, oc = A[l + 1]->split[g_rank];
        std::vector<double> v(Mf), ec(Mc), rc(Mc), ef(Mf);
        for (int i = 0; i < Mf; ++i) v[i] = f_v2(of + i);
        for (int i = 0; i < Mc; ++i) ec[i] = f_ec(oc + i);
        const std::string s = std::to_string(l);
        R[l]->matvec_sparse(v.data(), rc.data());
        write_raw(pfx + "R" + s + "_v2", "f64", gather_d(rc.data(), Mc));
        R[l]->matvec_sparse_float(v.data(), rc.data());
        write_raw(pfx + "R" + s + "_v2_float", "f64", gather_d(rc.data(), Mc));
        P[l]->matvec_sparse(ec.data(), ef.data());
        write_raw(pfx + "P" + s + "_ec", "f64", gather_d(ef.data(), Mf));
        P[l]->matvec_sparse_float(ec.data(), ef.data());
        write_raw(pfx + "P" + s + "_ec_float", "f64", gather_d(ef.data(), Mf));
    }

    // ---- the composed V-cycle ----
    auto dot = [&](const std::vector<double> &a, const std::vector<double> &b) {
        double s = 0, g = 0;
        for (size_t i = 0; i < a.size(); ++i) s += a[i] * b[i];
        MPI_Allreduce(&s, &g, 1, MPI_DOUBLE, MPI_SUM, comm);
        return g;
    };
    const double cg_tol = 1e-12;         // saena_object.h:155-156
    const int cg_max = 150;
    auto coarsest_cg = [&](saena_matrix *Ac, double *u, const double *rhs) {          // solve_coarsest_CG, :14-114
        const int sz = Ac->M;
        std::vector<double> res(rhs, rhs + sz), dir(rhs, rhs + sz), mt(sz), uu(u, u + sz);
        const double initial = dot(res, res);
        const double thres = initial * cg_tol * cg_tol;
        double d = initial, prev = 0, factor = 0;
        int max_iter = cg_max;
        if (d < cg_tol * cg_tol) max_iter = 0;
        int i = 1;
        while (i < max_iter) {
            Ac->matvec(dir.data(), mt.data());
            factor = d / dot(dir, mt);
            for (int j = 0; j < sz; ++j) { uu[j] += factor * dir[j]; res[j] -= factor * mt[j]; }
            prev = d;
            d = dot(res, res);
            if (d < thres) break;
            factor = d / prev;
            for (int j = 0; j < sz; ++j) dir[j] = res[j] + factor * dir[j];
            i++;
        }
        std::copy(uu.begin(), uu.end(), u);
    };
    int pre = 3, post = 3, smoother = 0;
    std::function<void(int, double *, double *)> vcycle = [&](int l, double *u, double *rhs) {
        saena_matrix *Al = A[l].get();
        if (l == nl - 1) { coarsest_cg(Al, u, rhs); return; }
        auto smooth = [&](int it) { if (smoother == 0) Al->jacobi(it, u, rhs); else Al->chebyshev(it, u, rhs); };
        if (pre) smooth(pre);
        const int M = Al->M, Mc = A[l + 1]->M;
        double *res = saena_aligned_alloc<value_t>(std::max(M, 1));
        Al->residual(u, rhs, res);
        std::vector<double> res_coarse(std::max(Mc, 1)), uc(std::max(Mc, 1), 0.0), uCorr(std::max(M, 1));
        R[l]->matvec(res, res_coarse.data());
        saena_free(res);
        vcycle(l + 1, uc.data(), res_coarse.data());
        P[l]->matvec(uc.data(), uCorr.data());
        for (int i = 0; i < M; ++i) u[i] -= uCorr[i];
        if (post) smooth(post);
    };
    const int M0 = A[0]->M, o0 = A[0]->split[g_rank];
    struct Case { const char *name; int smoother, pre, post; };
    const Case cases[] = {{"jacobi33", 0, 3, 3}, {"jacobi21", 0, 2, 1}, {"cheby33", 1, 3, 3}, {"cheby12", 1, 1, 2}};
    for (const Case &cs : cases) {
        smoother = cs.smoother; pre = cs.pre; post = cs.post;
        std::vector<double> u(std::max(M0, 1)), rhs(std::max(M0, 1));
        for (int i = 0; i < M0; ++i) { u[i] = 0.01 * f_v2(o0 + i); rhs[i] = f_rhs2(o0 + i); }
        vcycle(0, u.data(), rhs.data());
        write_raw(pfx + "vcycle_" + cs.name, "f64", gather_d(u.data(), M0));
        // a second cycle from the zero iterate (what solve_pCG's preconditioner call does, :2640-2641)
        std::fill(u.begin(), u.end(), 0.0);
        vcycle(0, u.data(), rhs.data());
        write_raw(pfx + "vcycle0_" + cs.name, "f64", gather_d(u.data(), M0));
    }
    // ---- the outer loops around it: solve_pCG (:2389-2801) and solve (:1883-2014), options001 (Jacobi 3+3, tol 1e-8) ----
    {
        smoother = 0; pre = 3; post = 3;
        const double tol = 1e-8;
        const int max_iter = 50;
        std::vector<double> rhs(std::max(M0, 1)), u(std::max(M0, 1), 0.0), r(std::max(M0, 1)), rho(std::max(M0, 1)), p(std::max(M0, 1)), h(std::max(M0, 1));
        for (int i = 0; i < M0; ++i) rhs[i] = f_rhs2(o0 + i);
        auto ldot = [&](const std::vector<double> &a, const std::vector<double> &b) {
            double s = 0, g = 0;
            for (int i = 0; i < M0; ++i) s += a[i] * b[i];
            MPI_Allreduce(&s, &g, 1, MPI_DOUBLE, MPI_SUM, comm);
            return g;
        };
        // solve_pCG
        std::vector<double> hist;
        {
            double *rp = r.data();
            A[0]->residual(u.data(), rhs.data(), rp);                        // :2497
            const double init_dot = ldot(r, r);
            hist.push_back(sqrt(init_dot));
            std::fill(rho.begin(), rho.end(), 0.0);
            vcycle(0, rho.data(), r.data());                                  // :2536-2537
            p = rho;
            const double THRSHLD = init_dot * tol * tol;
            double current_dot = init_dot;
            int i = 0;
            for (i = 0; i < max_iter; ++i) {                                  // :2565
                A[0]->matvec(p.data(), h.data());
                const double rho_res = ldot(r, rho), pdoth = ldot(p, h);
                const double alpha = rho_res / pdoth;
                for (int j = 0; j < M0; ++j) { u[j] -= alpha * p[j]; r[j] -= alpha * h[j]; }
                current_dot = ldot(r, r);
                hist.push_back(sqrt(current_dot));
                if (current_dot < THRSHLD) break;
                std::fill(rho.begin(), rho.end(), 0.0);
                vcycle(0, rho.data(), r.data());
                double beta = ldot(r, rho);
                beta /= rho_res;
                for (int j = 0; j < M0; ++j) p[j] = rho[j] + beta * p[j];
            }
        }
        write_raw(pfx + "pcg_hist", "f64", hist);
        write_raw(pfx + "pcg_u", "f64", gather_d(u.data(), M0));
        // solve: u = 0; repeat vcycle until ||r||^2 < ||r0||^2 tol^2
        std::fill(u.begin(), u.end(), 0.0);
        hist.clear();
        {
            double *rp = r.data();
            A[0]->residual(u.data(), rhs.data(), rp);
            const double init_dot = ldot(r, r);
            hist.push_back(sqrt(init_dot));
            const double THRSHLD = init_dot * tol * tol;
            for (int i = 0; i < max_iter; ++i) {                              // :1957-1970
                vcycle(0, u.data(), rhs.data());
                A[0]->residual(u.data(), rhs.data(), rp);
                const double d = ldot(r, r);
                hist.push_back(sqrt(d));
                if (d < THRSHLD) break;
            }
        }
        write_raw(pfx + "solve_hist", "f64", hist);
        write_raw(pfx + "solve_u", "f64", gather_d(u.data(), M0));
    }
    std::vector<long> meta = {(long)nl, (long)g_np};
    write_raw(pfx + "meta", "i64", meta);
    if (!g_rank) printf("%s np=%d levels=%d rows0=%d: transfers and 4 composed V-cycles written\n", argv[3], g_np, nl, (int)A[0]->Mbig);
    MPI_Finalize();
    return 0;
}

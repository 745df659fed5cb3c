// poisson.cpp -- the reference's experiments/Poisson.cpp flow on the MI355X path:
//   ./poisson <mx> [options.xml] [all]
// 3D 7-point Poisson on an mx^3 grid: generate, assemble, AMG setup, 1 warm-up + timed solve_pCG.
// Prints the residual lines the reference prints (src/saena_object_solve.cpp:2502,2681-2682).
#include "saena.hpp"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

int main(int argc, char **argv) {
    if (argc < 2) { printf("usage: %s <mx> [options.xml]\n", argv[0]); return 1; }
    const index_t mx = atoi(argv[1]);
    saena::init(0, 0, 1, nullptr);
    saena::comm comm;
    const int rank = comm.rank();

    saena::matrix A(comm);
    saena::laplacian3D(&A, mx, mx, mx);
    A.assemble();

    value_t *rhs_std = nullptr;
    index_t first = 0;
    const index_t sz = saena::laplacian3D_set_rhs(rhs_std, mx, mx, mx, comm, &first);
    saena::vector rhs(comm);
    rhs.set(rhs_std, sz, first);
    rhs.assemble();

    saena::options opts = argc > 2 ? saena::options(std::string(argv[2]))
                                   : saena::options(50, 1e-8, "jacobi", 3, 3, "jacobi", 0.2f, true, 20, 3, 1e-14, 1e-8, 1, 2);
    if (argc > 2) A.set_eig(std::string(argv[2]));

    saena::amg solver;
    solver.set_verbose(true);
    solver.set_scale(false);                              // the reference driver's knobs are accepted (experiments/Poisson.cpp)
    solver.set_num_threads(1);
    auto t0 = std::chrono::steady_clock::now();
    solver.set_matrix(&A, &opts);
    solver.set_rhs(rhs);
    auto t1 = std::chrono::steady_clock::now();
    if (!rank) printf("setup: %.3f s\n", std::chrono::duration<double>(t1 - t0).count());

    value_t *u = nullptr;
    solver.solve_pCG(u, &opts, false);                    // warm-up
    t0 = std::chrono::steady_clock::now();
    solver.solve_pCG(u, &opts);
    t1 = std::chrono::steady_clock::now();
    if (!rank) printf("solve_pCG: %.3f ms, %d iterations\n", 1e3 * std::chrono::duration<double>(t1 - t0).count(), solver.last_iterations());

    if (argc > 3) {                                       // the rest of the live saena::amg surface
        solver.profile_matvecs();                         // saena_object.cpp:618-638
        saena::options sm(opts);
        sm.set_max_iter(10);
        solver.solve_smoother(u, &sm);                    // 10 x preSmooth sweeps, no coarse grids
        if (!rank) printf("solve_smoother: %d iterations, residual %e -> %e\n", solver.last_iterations(),
                          solver.residual_history().front(), solver.residual_history().back());
        solver.solve(u, &opts);                           // stationary V-cycle iteration
        if (!rank) printf("solve: %d iterations\n", solver.last_iterations());
        value_t *ug = nullptr;                            // compiled out in the reference (#if 0): returns 0, u untouched (zeros when null)
        if (solver.solve_pGMRES(ug, &opts) != 0 || solver.solve_GMRES(ug, &opts) != 0 || ug == nullptr || ug[0] != 0.0) {
            if (!rank) printf("solve_GMRES / solve_pGMRES: not the reference's no-op\n");
            return 5;
        }
        saena::free_vector(ug);
        saena::matrix C(comm);
        solver.matmat(&A, &A, &C, true, true);
        if (!rank) printf("matmat: C = A*A has %d rows, %ld nnz\n", C.get_num_rows(), (long)C.get_nnz());
        C.destroy();
        // assemble(scale = false, use_dense = true): a small band matrix stored as dense rows gives the sparse form's product
        saena::matrix Bs(comm), Bd(comm);
        saena::band_matrix(Bs, 600, 9);
        for (index_t i = 0; i < 600; ++i)
            for (index_t j = (i > 9 ? i - 9 : 0); j <= (i + 9 < 599 ? i + 9 : 599); ++j) Bd.set(i, j, 1.0 / (i + j + 1));
        Bd.set_remove_boundary(false);
        Bd.assemble(false, true);
        std::vector<value_t> xv((size_t)Bs.get_num_local_rows()), ys, yd;
        for (size_t i = 0; i < xv.size(); ++i) xv[i] = 1.0 + 0.001 * (double)i;
        Bs.matvec(xv, ys);
        Bd.matvec(xv, yd);
        double dmax = 0, ymax = 0;
        for (size_t i = 0; i < ys.size(); ++i) { dmax = std::max(dmax, std::abs(ys[i] - yd[i])); ymax = std::max(ymax, std::abs(ys[i])); }
        if (!rank) printf("use_dense: %d rows, max |sparse - dense| / max |y| = %.2e\n", Bd.get_num_local_rows(), dmax / ymax);
        Bs.destroy(); Bd.destroy();
    }

    saena::free_vector(u);
    free(rhs_std);
    solver.destroy();
    A.destroy();
    saena::finalize();
    return 0;
}

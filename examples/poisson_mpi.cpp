// poisson_mpi.cpp -- a driver in the reference's own idiom (MPI_Init, MPI_Comm everywhere, saena_free, print_time) over
// include/saena_mpi.hpp: what a user of the reference keeps when switching to the MI355X path.  Same call order as the reference's
// experiments/Poisson.cpp:16-262; launched like it:  mpirun -np N ./poisson_mpi <m> <options.xml>
#include "saena_mpi.hpp"

#include <cstdio>

int main(int argc, char *argv[]) {
    MPI_Init(&argc, &argv);
    MPI_Comm comm = MPI_COMM_WORLD;
    int nprocs = 0, rank = 0;
    MPI_Comm_size(comm, &nprocs);
    MPI_Comm_rank(comm, &rank);
    if (argc != 3) {
        if (rank == 0) std::printf("usage: %s <grid points per side> <options xml>\n", argv[0]);
        MPI_Finalize();
        return -1;
    }
    const int m = std::atoi(argv[1]);
    const std::string optsfile(argv[2]);
    int status = 0;
    try {
        double t1 = omp_get_wtime();
        saena::matrix A(comm);                               // MPI_Comm -> the GPU runtime comes up over the job here
        saena::laplacian3D(&A, m, m, m);
        A.set_remove_boundary(true);
        A.assemble(false);
        double t2 = omp_get_wtime();
        print_time(t2 - t1, "Assemble:", comm, true, true);

        value_t *rhs_std = nullptr;
        index_t orig_sz = saena::laplacian3D_set_rhs(rhs_std, m, m, m, comm);
        index_t my_split = 0;
        saena::find_split(orig_sz, my_split, comm);
        saena::vector rhs(comm);
        rhs.set(&rhs_std[0], orig_sz, my_split);
        rhs.assemble();

        saena::options opts;
        opts.set_from_file(optsfile);
        A.set_eig(optsfile);

        saena::matrix B(A);                                  // copies are deep (reference saena.cpp:14-31)
        if (B.get_nnz() != A.get_nnz() || B.get_num_local_rows() != A.get_num_local_rows()) throw std::runtime_error("copy differs");
        MPI_Comm back = A.get_comm();
        int cmp = MPI_UNEQUAL;
        MPI_Comm_compare(back, comm, &cmp);
        if (cmp != MPI_IDENT) throw std::runtime_error("get_comm() is not the job's communicator");

        t1 = omp_get_wtime();
        saena::amg solver;
        solver.set_scale(false);
        solver.set_matrix(&A, &opts);
        solver.set_rhs(rhs);
        t2 = omp_get_wtime();
        print_time(t2 - t1, "Setup:", comm, true, true);

        value_t *u = nullptr;
        solver.solve_pCG(u, &opts);
        t1 = omp_get_wtime();
        for (int i = 0; i < 3; ++i) solver.solve_pCG(u, &opts, false);
        t2 = omp_get_wtime();
        print_time(t1 / 3, t2 / 3, "Solve:", comm);
        solver.solve_pCG_profile(u, &opts);
        solver.profile_matvecs();

        A.destroy();
        solver.destroy();
        saena_free(u);
        saena_free(rhs_std);
    } catch (const std::exception &e) {
        std::fprintf(stderr, "rank %d: %s\n", rank, e.what());
        status = 1;
    }
    saena::finalize();
    MPI_Finalize();
    return status;
}

// saena_mpi.hpp -- include/saena.hpp for drivers written against the reference: the MPI_Comm overloads.
//
// The reference's public surface takes MPI_Comm (include/saena.hpp:17 `explicit matrix(MPI_Comm)`, :79 `vector(MPI_Comm)`,
// include/aux_functions2.h:19 `laplacian3D_set_rhs(..., MPI_Comm)`, :43 `find_split`), and its drivers include
// "saena.hpp", "data_struct.h" and "aux_functions2.h" (experiments/Poisson.cpp:1-8).  Replacing those three includes by
//
//     #include "saena_mpi.hpp"
//
// is the whole port of such a driver (tests/test_cpp_surface.py compiles the flow of experiments/Poisson.cpp:16-262 that way
// against the image's MPICH and, on a GPU box, runs it).  What happens underneath:
//   * saena::comm is constructible from an MPI_Comm.  The first one brings the GPU runtime up over the MPI job: rank and size from
//     the communicator, device = this rank's index among the ranks of its node (MPI_Comm_split_type SHARED), the RCCL unique id made
//     on rank 0 and handed round with MPI_Bcast, then saena::init() (INTEGRATION.md section 1 shows the same code spelled out).
//     From then on MPI carries nothing on the data path: halos and dots ride RCCL over xGMI.
//   * the helpers those drivers take from the reference's other headers: saena::find_split (aux_functions2.cpp:1511-1528),
//     print_time (aux_functions.cpp:72-128), saena_free (aux_functions.h:306-312), omp_get_wtime when OpenMP is off.
// Out of the path's scope and therefore absent: GMRES, lazy updates, the Nektar++ set_matrix, PETSc (solve_petsc compiles and
// reports that it is not available).
#pragma once
#ifndef SAENA_MPI_HPP
#define SAENA_MPI_HPP
#include <mpi.h>

#include <chrono>
#include <cstdlib>
#include <iomanip>
#include <iostream>
#include <stdexcept>
#include <string>
#include <vector>

#include "saena.hpp"
#ifndef SAENA_HPP_HAS_MPI
#error "include saena_mpi.hpp before (or instead of) saena.hpp: the MPI_Comm overloads are declared when saena.hpp is read"
#endif

#ifdef _OPENMP
#include <omp.h>
#else
inline double omp_get_wtime() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#endif

using std::string;      // the reference's headers are written inside `using namespace std`

namespace saena {

// the MPI communicator the GPU runtime was brought up over (one job = one communicator, as in the reference's drivers)
inline MPI_Comm &mpi_world() { static MPI_Comm c = MPI_COMM_NULL; return c; }

inline comm::comm(MPI_Comm c) : comm([c]() -> comm {
    MPI_Comm &w = mpi_world();
    if (w == MPI_COMM_NULL) {
        int rank = 0, size = 1, local = 0;
        MPI_Comm_rank(c, &rank);
        MPI_Comm_size(c, &size);
        MPI_Comm node;
        MPI_Comm_split_type(c, MPI_COMM_TYPE_SHARED, rank, MPI_INFO_NULL, &node);      // one process per GPU of the node
        MPI_Comm_rank(node, &local);
        MPI_Comm_free(&node);
        char id[128] = {0};
        if (size > 1) {
            if (rank == 0) saena::unique_id(id);
            MPI_Bcast(id, 128, MPI_BYTE, 0, c);
        }
        if (const char *d = std::getenv("SAENA_DEVICE")) local = std::atoi(d);
        saena::init(local, rank, size, size > 1 ? id : nullptr);
        w = c;
    } else {
        int same = MPI_UNEQUAL;
        MPI_Comm_compare(w, c, &same);
        if (same != MPI_IDENT && same != MPI_CONGRUENT)
            throw std::runtime_error("saena: the GPU runtime was brought up over another MPI communicator (one communicator per job)");
    }
    return comm();
}()) {}
inline comm::operator MPI_Comm() const { return mpi_world(); }

// find_split (reference src/aux_functions2.cpp:1511-1528): the first global index of this rank's block of loc_size entries
inline index_t find_split(index_t loc_size, index_t &my_split, MPI_Comm c) {
    int rank = 0, nprocs = 1;
    MPI_Comm_size(c, &nprocs);
    MPI_Comm_rank(c, &rank);
    std::vector<index_t> all((size_t)nprocs);
    MPI_Allgather(&loc_size, 1, MPI_INT, all.data(), 1, MPI_INT, c);
    my_split = 0;
    for (int i = 0; i < rank; ++i) my_split += all[(size_t)i];
    return 0;
}

} // namespace saena

// print_time (reference src/aux_functions.cpp:72-128)
inline double print_time(double t_start, double t_end, const std::string &function_name, MPI_Comm c) {
    int rank = 0, nprocs = 1;
    MPI_Comm_rank(c, &rank);
    MPI_Comm_size(c, &nprocs);
    double mn = 0, mx = 0, av = 0;
    const double t = t_end - t_start;
    MPI_Reduce(&t, &mn, 1, MPI_DOUBLE, MPI_MIN, 0, c);
    MPI_Reduce(&t, &mx, 1, MPI_DOUBLE, MPI_MAX, 0, c);
    MPI_Reduce(&t, &av, 1, MPI_DOUBLE, MPI_SUM, 0, c);
    av /= nprocs;
    if (rank == 0) std::cout << std::endl << function_name << "\nmin: " << mn << "\nave: " << av << "\nmax: " << mx << std::endl << std::endl;
    return av;
}
inline double print_time(double t_dif, const std::string &function_name, MPI_Comm c, bool print_time = false, bool print_name = true, int optype = 0) {
    int rank = 0, nprocs = 1;
    MPI_Comm_rank(c, &rank);
    MPI_Comm_size(c, &nprocs);
    double v = 0.0;
    if (optype == 1) MPI_Reduce(&t_dif, &v, 1, MPI_DOUBLE, MPI_MIN, 0, c);
    else if (optype == 2) MPI_Reduce(&t_dif, &v, 1, MPI_DOUBLE, MPI_MAX, 0, c);
    else { MPI_Reduce(&t_dif, &v, 1, MPI_DOUBLE, MPI_SUM, 0, c); v /= nprocs; }
    std::cout << std::setprecision(8);
    if (print_time && rank == 0) {
        if (print_name) std::cout << function_name << "\n" << v << std::endl;
        else std::cout << v << std::endl;
    }
    return v;
}

// saena_free (reference include/aux_functions.h:306-312): what solve* and laplacian3D_set_rhs hand out comes from malloc here too
template <class T>
inline void saena_free(T *&v) {
    if (v != nullptr) { std::free(v); v = nullptr; }
}
#endif // SAENA_MPI_HPP

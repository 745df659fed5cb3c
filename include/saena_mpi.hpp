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
//     print_time (aux_functions.cpp:72-128), saena_free / saena_aligned_alloc (aux_functions.h:299-312), read_from_file_rhs
//     (aux_functions.cpp:347-497: the rhs reader of experiments/profile_file.cpp), omp_get_wtime when OpenMP is off.
// Out of the path's scope and therefore absent: GMRES, lazy updates, the Nektar++ set_matrix, PETSc (solve_petsc compiles and
// reports that it is not available).
#pragma once
#ifndef SAENA_MPI_HPP
#define SAENA_MPI_HPP
#include <mpi.h>

#include <algorithm>
#include <cassert>
#include <chrono>
#include <cstdlib>
#include <fstream>
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

// saena_aligned_alloc (reference include/aux_functions.h:299-303): sz elements, 64-byte aligned, released with saena_free
template <class T>
inline T *saena_aligned_alloc(const nnz_t sz) {
    const size_t bytes = ((size_t)(sz > 0 ? sz : 1) * sizeof(T) + 63) / 64 * 64;
    return static_cast<T *>(std::aligned_alloc(64, bytes));
}

// read_from_file_rhs (reference src/aux_functions.cpp:347-497, the rhs reader of experiments/profile_file.cpp): this rank's rows
// [split[rank], split[rank + 1]) of a vector file.  "<name>.bin": the doubles of the whole vector, row after row.  Any other
// extension: a text file -- `%` comment lines, the size, then one "row value" pair per line with rows counted from 1 -- which rank 0
// turns into "<name>.bin" next to it first, sorted by row, unless that file is already there (the reference's behaviour, side
// effect included).  A missing file is the reference's message and exit.
inline int read_from_file_rhs(value_t *v, const std::vector<index_t> &split, char *file, MPI_Comm c) {
    int rank = 0;
    MPI_Comm_rank(c, &rank);
    const std::string filename(file);
    {
        std::ifstream probe(filename.c_str());
        if (!probe.is_open()) {
            if (!rank) std::cout << "\nCould not open the rhs file <" << filename << ">" << std::endl;
            MPI_Finalize();
            std::exit(EXIT_FAILURE);
        }
    }
    const size_t ext = filename.find_last_of('.');
    if (ext == std::string::npos || ext == filename.size() - 1) {
        if (!rank) std::cout << "The rhs file name does not have an extension!" << std::endl;
        MPI_Abort(c, 1);
    }
    const std::string binname = filename.substr(0, ext) + ".bin";
    if (filename.substr(ext + 1) != "bin") {
        if (!rank && !std::ifstream(binname.c_str()).is_open()) {
            std::ifstream in(filename.c_str());
            while (in.peek() == '%') in.ignore(2048, '\n');
            nnz_t sz = 0;
            in >> sz;
            std::vector<std::pair<index_t, value_t>> e;
            e.reserve((size_t)std::max<nnz_t>(sz, 0));
            index_t a = 0;
            value_t val = 0.0;
            while (in >> a >> val) e.emplace_back(a - 1, val);
            std::stable_sort(e.begin(), e.end(), [](const std::pair<index_t, value_t> &x, const std::pair<index_t, value_t> &y) { return x.first < y.first; });
            std::ofstream out(binname.c_str(), std::ios::out | std::ios::binary);
            for (nnz_t i = 0; i < sz && (size_t)i < e.size(); ++i) out.write(reinterpret_cast<const char *>(&e[(size_t)i].second), sizeof(value_t));
        }
        MPI_Barrier(c);                                   // the binary file written by rank 0 is ready
    }
    std::ifstream in(binname.c_str(), std::ios::in | std::ios::binary);
    if (!in.is_open()) {
        if (!rank) std::cout << "Unable to open the rhs vector file!" << std::endl;
        MPI_Finalize();
        return -1;
    }
    const index_t lo = split[(size_t)rank], n = split[(size_t)rank + 1] - lo;
    in.seekg((std::streamoff)lo * (std::streamoff)sizeof(value_t));
    in.read(reinterpret_cast<char *>(v), (std::streamsize)n * (std::streamsize)sizeof(value_t));
    if (in.gcount() != (std::streamsize)n * (std::streamsize)sizeof(value_t)) {
        if (!rank) std::cout << "Error: Size of RHS does not match the number of rows of the LHS matrix!" << std::endl;
        MPI_Abort(c, 1);
    }
    return 0;
}
#endif // SAENA_MPI_HPP

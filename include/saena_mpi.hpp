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
#include <cmath>
#include <random>
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

// The callbacks of saena::init_host_transport over the job's MPI communicator (SAENA_MPI_HOST_TRANSPORT=1: a multi-rank job on one card)
inline const host_transport &mpi_host_transport() {
    struct F {
        static MPI_Comm comm_() { return mpi_world(); }
        static int exchange(void *, const void *send, const int *send_rank, const int *send_count, int nsend,
                            void *recv, const int *recv_rank, const int *recv_count, int nrecv, int elem_bytes) {
            std::vector<MPI_Request> rq((size_t)(nsend + nrecv));
            size_t off = 0;
            int k = 0;
            for (int i = 0; i < nrecv; ++i) {
                MPI_Irecv(static_cast<char *>(recv) + off, recv_count[i] * elem_bytes, MPI_BYTE, recv_rank[i], 7701, comm_(), &rq[(size_t)k++]);
                off += (size_t)recv_count[i] * (size_t)elem_bytes;
            }
            off = 0;
            for (int i = 0; i < nsend; ++i) {
                MPI_Isend(static_cast<const char *>(send) + off, send_count[i] * elem_bytes, MPI_BYTE, send_rank[i], 7701, comm_(), &rq[(size_t)k++]);
                off += (size_t)send_count[i] * (size_t)elem_bytes;
            }
            return MPI_Waitall(k, rq.data(), MPI_STATUSES_IGNORE) == MPI_SUCCESS ? 0 : 1;
        }
        static int allreduce(void *, double *v, int n) { return MPI_Allreduce(MPI_IN_PLACE, v, n, MPI_DOUBLE, MPI_SUM, comm_()) == MPI_SUCCESS ? 0 : 1; }
        static int allreduce_i64(void *, long *v, int n) { return MPI_Allreduce(MPI_IN_PLACE, v, n, MPI_LONG, MPI_SUM, comm_()) == MPI_SUCCESS ? 0 : 1; }
        static int allgather(void *, const void *send, void *recv, size_t bytes) {
            if (bytes > (size_t)1 << 30) return 1;                  // (the setup gathers counts and small tables: MPI counts are ints)
            return MPI_Allgather(send, (int)bytes, MPI_BYTE, recv, (int)bytes, MPI_BYTE, comm_()) == MPI_SUCCESS ? 0 : 1;
        }
        static int alltoallv(void *, const void *send, const size_t *sc, const size_t *sd, void *recv, const size_t *rc, const size_t *rd) {
            int np = 1;
            MPI_Comm_size(comm_(), &np);
            const size_t CH = (size_t)1 << 30;                      // pieces of at most 1 GiB: MPI counts are ints
            std::vector<MPI_Request> rq;
            for (int p = 0; p < np; ++p)
                for (size_t o = 0; o < rc[p]; o += CH) {
                    rq.emplace_back();
                    MPI_Irecv(static_cast<char *>(recv) + rd[p] + o, (int)std::min(CH, rc[p] - o), MPI_BYTE, p, 7702, comm_(), &rq.back());
                }
            for (int p = 0; p < np; ++p)
                for (size_t o = 0; o < sc[p]; o += CH) {
                    rq.emplace_back();
                    MPI_Isend(static_cast<const char *>(send) + sd[p] + o, (int)std::min(CH, sc[p] - o), MPI_BYTE, p, 7702, comm_(), &rq.back());
                }
            return MPI_Waitall((int)rq.size(), rq.data(), MPI_STATUSES_IGNORE) == MPI_SUCCESS ? 0 : 1;
        }
    };
    static const host_transport t{nullptr, &F::exchange, &F::allreduce, &F::allgather, &F::alltoallv, &F::allreduce_i64, &F::allreduce};
    return t;
}

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
        if (size > 1 && std::getenv("SAENA_MPI_HOST_TRANSPORT")) {
            // several ranks on ONE card (RCCL refuses that): halos, reductions and the setup's collectives over MPI itself
            w = c;                                                // (the callbacks reach the communicator through mpi_world())
            try { saena::init_host_transport(local, rank, size, mpi_host_transport()); }
            catch (...) { w = MPI_COMM_NULL; throw; }
            return comm();
        }
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

// ---- the rest of the generators / checkers of the reference's public header (include/saena.hpp:271-298, src/aux_functions2.cpp), so
// that a driver which calls them compiles; formulas as there.  (In the reference's own drivers every call of the 2-D family and of the
// solution checkers is commented out: solve* returns u WITHOUT the removed boundary rows, which these index -- saena.cpp:782-811.)

// laplacian2D (aux_functions2.cpp:3-88): the 5-point Laplacian of an mx x my grid, boundary nodes as rows of their own; serial only,
// like the reference.  scale = true is the reference's default and asks for the symmetric diagonal scaling this path does not have
// (saena::matrix::assemble refuses it): pass false.
inline int laplacian2D(saena::matrix *A, index_t mx, index_t my, bool scale = true) {
    if (A->get_comm().size() > 1) {
        if (A->get_comm().rank() == 0) printf("laplacian2D works only in serial!\n");
        MPI_Abort(mpi_world(), 1);
    }
    const value_t Hx = 1.0 / (mx - 1), Hy = 1.0 / (my - 1), HydHx = Hy / Hx, HxdHy = Hx / Hy;
    const index_t XMAX = mx - 1, YMAX = my - 1;
    for (index_t j = 0; j < my; ++j)
        for (index_t i = 0; i < mx; ++i) {
            const index_t node = mx * j + i;
            if (i != 0 && j != 0 && i != XMAX && j != YMAX) {
                if (j - 1 != 0)    A->set(node, node - mx, -HxdHy);
                if (i - 1 != 0)    A->set(node, node - 1, -HydHx);
                if (i + 1 != XMAX) A->set(node, node + 1, -HydHx);
                if (j + 1 != YMAX) A->set(node, node + mx, -HxdHy);
            }
            A->set(node, node, 2.0 * (HxdHy + HydHx));
        }
    A->assemble(scale);
    return 0;
}
// laplacian2D_set_rhs (:90-132): f = 8 pi^2 sin(2 pi x) sin(2 pi y) at every node of the grid
inline int laplacian2D_set_rhs(std::vector<double> &rhs, index_t mx, index_t my, MPI_Comm) {
    const double PI = 3.14159265358979323846, Hx = 1.0 / (mx - 1), Hy = 1.0 / (my - 1);
    rhs.resize((size_t)mx * my);
    size_t it = 0;
    for (index_t j = 0; j < my; ++j)
        for (index_t i = 0; i < mx; ++i) rhs[it++] = 8 * PI * PI * std::sin(2 * PI * i * Hx) * std::sin(2 * PI * j * Hy);
    return 0;
}
// laplacian2D_check_solution (:134-179): || u - sin(2 pi x) sin(2 pi y) ||_2 over the grid, printed; returned through *norm too
inline int laplacian2D_check_solution(std::vector<double> &u, index_t mx, index_t my, MPI_Comm, double *norm = nullptr) {
    const double PI = 3.14159265358979323846, Hx = 1.0 / (mx - 1), Hy = 1.0 / (my - 1);
    double dif = 0.0;
    size_t it = 0;
    for (index_t j = 0; j < my; ++j)
        for (index_t i = 0; i < mx; ++i) { const double t = u[it++] - std::sin(2 * PI * i * Hx) * std::sin(2 * PI * j * Hy); dif += t * t; }
    std::cout << "\nnorm of diff = " << std::sqrt(dif) << std::endl;
    if (norm) *norm = std::sqrt(dif);
    return 0;
}
// laplacian3D_check_solution (:702-763): this rank's z-slab of the grid against sin sin sin; rank 0 prints sqrt(its OWN sum / (mx my mz)) --
// the reference reduces the sums and then prints the local one (:761); kept, the reduced figure goes to *norm
inline int laplacian3D_check_solution(std::vector<double> &u, index_t mx, index_t my, index_t mz, MPI_Comm c, double *norm = nullptr) {
    int rank = 0, nprocs = 1;
    MPI_Comm_size(c, &nprocs); MPI_Comm_rank(c, &rank);
    const double PI = 3.14159265358979323846;
    double dif = 0.0;
    if (rank < mz) {
        const double Hx = 1.0 / (mx - 1), Hy = 1.0 / (my - 1), Hz = 1.0 / (mz - 1);
        index_t zm = 1, zs = rank;
        if (mz > nprocs) { zm = mz / nprocs; zs = rank * zm; if (rank == nprocs - 1) zm = mz - (nprocs - 1) * zm; }
        size_t it = 0;
        for (index_t k = zs; k < zs + zm; ++k)
            for (index_t j = 0; j < my; ++j)
                for (index_t i = 0; i < mx; ++i) {
                    const double t = u[it++] - std::sin(2 * PI * i * Hx) * std::sin(2 * PI * j * Hy) * std::sin(2 * PI * k * Hz);
                    dif += t * t;
                }
    }
    double tot = 0.0;
    MPI_Reduce(&dif, &tot, 1, MPI_DOUBLE, MPI_SUM, 0, c);
    if (!rank) std::cout << "\nnorm of diff = " << std::sqrt(dif / ((double)mx * my * mz)) << std::endl;
    if (norm) *norm = std::sqrt(tot / ((double)mx * my * mz));
    return 0;
}
// laplacian3D_set_rhs_zero (:1249-1294): zero the boundary nodes of a right-hand side over the WHOLE grid (global node index, as there)
inline int laplacian3D_set_rhs_zero(std::vector<double> &rhs, unsigned int mx, unsigned int my, unsigned int mz, MPI_Comm c) {
    int rank = 0, nprocs = 1;
    MPI_Comm_size(c, &nprocs); MPI_Comm_rank(c, &rank);
    unsigned zm = mz / (unsigned)nprocs;
    const unsigned zs = (unsigned)rank * zm;
    if (rank == nprocs - 1) zm = mz - (unsigned)(nprocs - 1) * zm;
    for (unsigned k = zs; k < zs + zm; ++k)
        for (unsigned j = 0; j < my; ++j)
            for (unsigned i = 0; i < mx; ++i)
                if (i == 0 || j == 0 || k == 0 || i == mx - 1 || j == my - 1 || k == mz - 1) rhs[(size_t)mx * my * k + (size_t)mx * j + i] = 0;
    return 0;
}
// random_symm_matrix (:1384-1460): M rows per rank, a random diagonal in (0, 1) and floor(density M Mbig) entries in all, mirrored;
// seeded from std::random_device like the reference (no two runs alike there either)
inline int random_symm_matrix(saena::matrix &A, index_t M, float density) {
    const int rank = A.get_comm().rank(), nprocs = A.get_comm().size();
    if (density <= 0 || density > 1) {
        if (!rank) printf("Error: density should be in the range (0,1].\n");
        std::exit(EXIT_FAILURE);
    }
    const index_t Mbig = nprocs * M, offset = M * rank;
    const unsigned long nnz_l = (unsigned long)std::floor((double)density * M * Mbig);
    std::uniform_real_distribution<value_t> dist(0, 1);
    std::uniform_int_distribution<index_t> drow(0, M - 1), dcol(0, Mbig - 1);
    std::mt19937 rng(std::random_device{}()), rng2(std::random_device{}()), rng3(std::random_device{}());
    for (index_t i = offset; i < (rank == nprocs - 1 ? Mbig : offset + M); ++i) A.set(i, i, dist(rng));
    if (nnz_l > (unsigned long)M) {
        unsigned long left = (nnz_l - M) / 2;
        while (left) {
            const value_t vv = dist(rng);
            const index_t ii = drow(rng2) + offset, jj = dcol(rng3);
            if (ii > jj) { --left; A.set(ii, jj, vv); A.set(jj, ii, vv); }
        }
    }
    A.assemble();
    return 0;
}
// read_vector_file (:1462-1509): this rank's rows of a file of doubles, at the matrix's (assembled) partition
inline int read_vector_file(std::vector<value_t> &v, saena::matrix &A, char *file, MPI_Comm c) {
    int rank = 0;
    MPI_Comm_rank(c, &rank);
    std::ifstream in(file, std::ios::in | std::ios::binary);
    if (!in.is_open()) {
        if (rank == 0) std::cout << "Unable to open the rhs vector file!" << std::endl;
        MPI_Finalize();
        return -1;
    }
    v.resize((size_t)A.get_num_local_rows());
    in.seekg((std::streamoff)A.get_split()[(size_t)rank] * 8);
    in.read(reinterpret_cast<char *>(v.data()), (std::streamsize)v.size() * 8);
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

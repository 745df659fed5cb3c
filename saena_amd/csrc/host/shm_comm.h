// shm_comm.h -- the setup's collectives between the ranks of ONE node, through POSIX shared memory.
//
// The reference runs its setup exchanges over MPI (MPI_Alltoall/Alltoallv/Allreduce/Allgather:
// src/saena_matrix_setup.cpp:953,1030,1082,1086; src/saena_matrix_repart.cpp:293; the fetches of
// saena_object::SA / triple_mat_mult, src/saena_object_setup1.cpp, src/saena_object_setup2.cpp:361-849).
// One process drives one GPU and all processes of a job sit on one node (8 x MI355X), so here those
// exchanges are memory copies: every rank owns a segment in /dev/shm, writes what it sends into it,
// and its peers copy their blocks out -- no sockets, no staging through the device, no interpreter in
// the loop (the callback communicator routes every call through the caller's language).
//
//   control block  /dev/shm/<name>       barrier words + per rank: segment size, per-destination offset / count
//   data segment   /dev/shm/<name>.<r>   grows with the largest exchange (ftruncate + remap); peers remap lazily
//
// All names are unlinked as soon as every rank has opened them, so a crashed job leaves nothing behind.
// Reductions add the ranks' values in rank order on every rank: identical sums everywhere, run to run.
// A rank that waits longer than SAENA_SHM_TIMEOUT seconds (default 900) at a barrier throws.
#pragma once
#include "comm.h"

#include <memory>
#include <string>

namespace saena_host {

// nullptr + *err on failure.  Collective: every rank of the job calls it with the same name.
std::unique_ptr<Comm> make_shm_comm(const std::string &name, int rank, int nranks, std::string *err);

} // namespace saena_host

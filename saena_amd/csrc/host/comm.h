// comm.h -- the few collectives the host-side setup needs, behind one interface.
//
// The reference does its setup exchanges with MPI (MPI_Alltoall/Alltoallv/
// Allreduce/Allgather, src/saena_matrix_setup.cpp:953,1030,1082,1086;
// src/saena_matrix_repart.cpp:293).  Here the same exchanges go through
//   SelfComm      one rank (plain copies);
//   CallbackComm  caller-supplied C callbacks (tests plug torch.distributed/gloo in);
//   the RCCL communicator of the GPU runtime (sgpu_host_comm(), sgpu_runtime.hip).
#pragma once
#include "par.h"
#include <cstddef>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

namespace saena_host {

struct Comm {
    int rank = 0, nranks = 1;
    virtual ~Comm() {}
    // recv holds nranks blocks of `bytes` each, block q from rank q
    virtual void allgather(const void *send, void *recv, size_t bytes) = 0;
    // byte counts / displacements per peer
    virtual void alltoallv(const void *send, const size_t *scounts, const size_t *sdispls,
                           void *recv, const size_t *rcounts, const size_t *rdispls) = 0;
    virtual void allreduce_sum_i64(long *v, int n) = 0;
    virtual void allreduce_sum_f64(double *v, int n) = 0;

    // ---- conveniences built on the primitives ----
    template <class T>
    std::vector<T> allgather_one(const T &x) {
        std::vector<T> all((size_t)nranks);
        allgather(&x, all.data(), sizeof(T));
        return all;
    }
    // one T to / from every peer (MPI_Alltoall with count 1)
    template <class T>
    std::vector<T> alltoall_one(const std::vector<T> &send) {
        std::vector<size_t> c((size_t)nranks, sizeof(T)), d((size_t)nranks);
        for (int i = 0; i < nranks; ++i) d[i] = i * sizeof(T);
        std::vector<T> recv((size_t)nranks);
        alltoallv(send.data(), c.data(), d.data(), recv.data(), c.data(), d.data());
        return recv;
    }
    // variable-length exchange of T records; counts in records
    template <class T>
    std::vector<T> alltoallv_records(const std::vector<T> &send, const std::vector<int> &scount, std::vector<int> *rcount_out = nullptr) {
        std::vector<int> rcount = alltoall_one(scount);
        std::vector<size_t> sc((size_t)nranks), sd((size_t)nranks), rc((size_t)nranks), rd((size_t)nranks);
        size_t so = 0, ro = 0;
        for (int i = 0; i < nranks; ++i) {
            sc[i] = (size_t)scount[i] * sizeof(T); sd[i] = so; so += sc[i];
            rc[i] = (size_t)rcount[i] * sizeof(T); rd[i] = ro; ro += rc[i];
        }
        std::vector<T> recv(ro / sizeof(T));
        alltoallv(send.data(), sc.data(), sd.data(), recv.data(), rc.data(), rd.data());
        if (rcount_out) *rcount_out = rcount;
        return recv;
    }
    // the same with the receive counts already known (repeated exchanges over a fixed plan: no count round trip)
    template <class T>
    std::vector<T> alltoallv_known(const std::vector<T> &send, const std::vector<int> &scount, const std::vector<int> &rcount) {
        std::vector<size_t> sc((size_t)nranks), sd((size_t)nranks), rc((size_t)nranks), rd((size_t)nranks);
        size_t so = 0, ro = 0;
        for (int i = 0; i < nranks; ++i) {
            sc[i] = (size_t)scount[i] * sizeof(T); sd[i] = so; so += sc[i];
            rc[i] = (size_t)rcount[i] * sizeof(T); rd[i] = ro; ro += rc[i];
        }
        std::vector<T> recv(ro / sizeof(T));
        alltoallv(send.data(), sc.data(), sd.data(), recv.data(), rc.data(), rd.data());
        return recv;
    }
    long sum(long x) { allreduce_sum_i64(&x, 1); return x; }
    long max_(long x) {      // max through a gather (setup only)
        std::vector<long> all = allgather_one(x);
        long m = all[0];
        for (long v : all) m = v > m ? v : m;
        return m;
    }
};

struct SelfComm : Comm {
    void allgather(const void *send, void *recv, size_t bytes) override { std::memcpy(recv, send, bytes); }
    void alltoallv(const void *send, const size_t *sc, const size_t *sd, void *recv, const size_t *rc, const size_t *rd) override {
        if (sc[0] != rc[0]) throw std::runtime_error("SelfComm::alltoallv: count mismatch");
        std::memcpy(static_cast<char *>(recv) + rd[0], static_cast<const char *>(send) + sd[0], sc[0]);
    }
    void allreduce_sum_i64(long *, int) override {}
    void allreduce_sum_f64(double *, int) override {}
};

// C callbacks; each returns 0 on success
extern "C" {
typedef int (*saena_cb_allgather)(void *user, const void *send, void *recv, size_t bytes);
typedef int (*saena_cb_alltoallv)(void *user, const void *send, const size_t *scounts, const size_t *sdispls,
                                  void *recv, const size_t *rcounts, const size_t *rdispls);
typedef int (*saena_cb_allreduce_i64)(void *user, long *v, int n);
typedef int (*saena_cb_allreduce_f64)(void *user, double *v, int n);
}

struct CallbackComm : Comm {
    void *user = nullptr;
    saena_cb_allgather cb_allgather = nullptr;
    saena_cb_alltoallv cb_alltoallv = nullptr;
    saena_cb_allreduce_i64 cb_i64 = nullptr;
    saena_cb_allreduce_f64 cb_f64 = nullptr;
    static void ok(int s, const char *what) { if (s) throw std::runtime_error(std::string("comm callback failed: ") + what); }
    void allgather(const void *send, void *recv, size_t bytes) override { ok(cb_allgather(user, send, recv, bytes), "allgather"); }
    void alltoallv(const void *send, const size_t *sc, const size_t *sd, void *recv, const size_t *rc, const size_t *rd) override {
        // the block a rank sends to itself never leaves the process: one memcpy here instead of a trip through the
        // callback (the setup's row routing and transposition keep most of a multi-GB level local)
        if (sc[rank] != rc[rank]) throw std::runtime_error("CallbackComm::alltoallv: self count mismatch");
        if (sc[rank]) parallel_copy(static_cast<char *>(recv) + rd[rank], static_cast<const char *>(send) + sd[rank], sc[rank]);   // (send and recv never overlap)
        std::vector<size_t> sc2(sc, sc + nranks), rc2(rc, rc + nranks);
        sc2[(size_t)rank] = 0; rc2[(size_t)rank] = 0;
        ok(cb_alltoallv(user, send, sc2.data(), sd, recv, rc2.data(), rd), "alltoallv");
    }
    void allreduce_sum_i64(long *v, int n) override { ok(cb_i64(user, v, n), "allreduce_i64"); }
    void allreduce_sum_f64(double *v, int n) override { ok(cb_f64(user, v, n), "allreduce_f64"); }
};

} // namespace saena_host

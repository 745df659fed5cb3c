// saena_c_api.cpp -- flat C view (include/saena_c.h) of the host-side mirror.
#include "../../../include/saena_c.h"
#include "saena_matrix.h"

#include <algorithm>
#include <memory>
#include <string>

using namespace saena_host;

struct saena_comm { std::unique_ptr<Comm> c; };
struct saena_matrix_h { saena_matrix A; explicit saena_matrix_h(Comm *c) : A(c) {} };
struct saena_transfer_h { transfer_matrix T; };

namespace {
thread_local std::string h_err;
template <class F>
int guard(F f) {
    try { f(); return 0; }
    catch (const std::exception &e) { h_err = e.what(); return -1; }
    catch (...) { h_err = "unknown error"; return -1; }
}
void fill_desc(const DistLayout &L, const std::vector<value_t> *inv_diag, sgpu_op_desc *d) {
    d->M = L.M; d->N_local = L.N_local; d->col_offset = L.col_offset;
    d->nnz_l_local = L.nnz_l_local;
    d->nnzPerRow_local = L.nnzPerRow_local.data(); d->col_local = L.col_local.data(); d->val_local = L.val_local.data();
    d->nnz_l_remote = L.nnz_l_remote; d->col_remote_size = L.col_remote_size;
    d->nnzPerCol_remote = L.nnzPerCol_remote.data(); d->row_remote = L.row_remote.data(); d->val_remote = L.val_remote.data();
    d->numRecvProc = L.numRecvProc; d->numSendProc = L.numSendProc;
    d->recvProcRank = L.recvProcRank.data(); d->recvProcCount = L.recvProcCount.data();
    d->sendProcRank = L.sendProcRank.data(); d->sendProcCount = L.sendProcCount.data();
    d->vIndexSize = L.vIndexSize; d->vIndex = L.vIndex.data();
    d->inv_diag = inv_diag && !inv_diag->empty() ? inv_diag->data() : nullptr;
    d->halo_fp32 = 0;
}
} // namespace

extern "C" {

const char *saena_last_error(void) { return h_err.c_str(); }

saena_comm *saena_comm_self(void) {
    auto *c = new saena_comm();
    c->c.reset(new SelfComm());
    return c;
}

saena_comm *saena_comm_callbacks(int rank, int nranks, void *user, saena_cb_allgather ag, saena_cb_alltoallv a2a,
                                 saena_cb_allreduce_i64 ri, saena_cb_allreduce_f64 rf) {
    auto *cb = new CallbackComm();
    cb->rank = rank; cb->nranks = nranks; cb->user = user;
    cb->cb_allgather = ag; cb->cb_alltoallv = a2a; cb->cb_i64 = ri; cb->cb_f64 = rf;
    auto *c = new saena_comm();
    c->c.reset(cb);
    return c;
}

#ifndef SAENA_WITH_GPU
saena_comm *saena_comm_rccl(void) {      // host-only library: there is no GPU context to take the communicator from
    h_err = "saena_comm_rccl needs libsaena_amd.so (the GPU library); libsaena_host.so has no RCCL communicator";
    return nullptr;
}
#else
saena_host::Comm *sgpu_new_host_comm();  // sgpu_runtime.hip
saena_comm *saena_comm_rccl(void) {
    saena_host::Comm *r = sgpu_new_host_comm();
    if (!r) { h_err = "sgpu_init has not been called"; return nullptr; }
    auto *c = new saena_comm();
    c->c.reset(r);
    return c;
}
#endif

void saena_comm_free(saena_comm *c) { delete c; }

saena_matrix_h *saena_matrix_new(saena_comm *comm) { return comm ? new saena_matrix_h(comm->c.get()) : nullptr; }
void saena_matrix_free(saena_matrix_h *A) { delete A; }
int saena_matrix_set(saena_matrix_h *A, index_t i, index_t j, value_t v) { return guard([&] { A->A.set(i, j, v); }); }
int saena_matrix_set_many(saena_matrix_h *A, const index_t *r, const index_t *c, const value_t *v, nnz_t n) {
    return guard([&] { A->A.set(r, c, v, n); });
}
int saena_matrix_set_remove_boundary(saena_matrix_h *A, int b) { A->A.remove_boundary = b != 0; return 0; }
int saena_matrix_add_duplicates(saena_matrix_h *A, int add) { A->A.add_duplicates = add != 0; return 0; }
int saena_matrix_set_eig(saena_matrix_h *A, double e) { A->A.set_eig(e); return 0; }
int saena_matrix_assemble(saena_matrix_h *A) { return guard([&] { A->A.assemble(); }); }
int saena_matrix_assemble_with_split(saena_matrix_h *A, const index_t *split) {
    return guard([&] { A->A.assemble_with_split(std::vector<index_t>(split, split + A->A.comm->nranks + 1)); });
}
index_t saena_matrix_get_num_rows(saena_matrix_h *A) { return A->A.Mbig; }
index_t saena_matrix_get_num_local_rows(saena_matrix_h *A) { return A->A.M; }
nnz_t saena_matrix_get_nnz(saena_matrix_h *A) { return A->A.nnz_g; }
nnz_t saena_matrix_get_local_nnz(saena_matrix_h *A) { return A->A.nnz_l; }
int saena_matrix_get_split(saena_matrix_h *A, index_t *out) {
    return guard([&] {
        if (!A->A.assembled) throw std::runtime_error("matrix is not assembled");
        std::copy(A->A.split.begin(), A->A.split.end(), out);
    });
}
int saena_matrix_get_desc(saena_matrix_h *A, sgpu_op_desc *out) {
    return guard([&] {
        if (!A->A.assembled) throw std::runtime_error("matrix is not assembled");
        fill_desc(A->A.L, &A->A.inv_diag, out);
    });
}
int saena_matrix_get_layout_extra(saena_matrix_h *A, const index_t **col_remote, const nnz_t **scan) {
    if (col_remote) *col_remote = A->A.L.col_remote.data();
    if (scan) *scan = A->A.L.nnzPerProcScan.data();
    return 0;
}

int saena_laplacian3D(saena_matrix_h *A, index_t mx, index_t my, index_t mz) { return guard([&] { laplacian3D(&A->A, mx, my, mz); }); }
int saena_band_matrix(saena_matrix_h *A, index_t M, unsigned int bw) { return guard([&] { band_matrix(&A->A, M, bw); }); }

int saena_laplacian3D_set_rhs(saena_matrix_h *Ah, index_t mx, index_t my, index_t mz, value_t *rhs_local) {
    return guard([&] {
        saena_matrix &A = Ah->A;
        if (!A.assembled) throw std::runtime_error("matrix is not assembled");
        Comm &c = *A.comm;
        index_t lo = 0;
        std::vector<value_t> full = laplacian3D_set_rhs(c, mx, my, mz, &lo);
        std::vector<value_t> in = A.remove_boundary_rhs(full, lo);     // set_repartition_rhs, saena_object_repart_shrink.cpp:179-206
        const auto &B = A.bound_row_global;
        const index_t new_lo = lo - (index_t)(std::lower_bound(B.begin(), B.end(), lo) - B.begin());
        // repart_vector to A.split (saena_object_repart_shrink.cpp:218)
        struct rec { index_t id; value_t v; };
        std::vector<rec> recs(in.size());
        for (size_t i = 0; i < in.size(); ++i) recs[i] = {new_lo + (index_t)i, in[i]};
        std::vector<int> cnt((size_t)c.nranks, 0);
        for (const auto &r : recs) {
            int p = (int)lower_bound2(A.split.data(), A.split.data() + c.nranks, r.id);
            while (p < c.nranks - 1 && A.split[p + 1] <= r.id) ++p;
            cnt[p]++;
        }
        std::vector<rec> got = c.nranks == 1 ? recs : c.alltoallv_records(recs, cnt);   // ids ascend, so recs are grouped by owner
        if ((index_t)got.size() != A.M) throw std::runtime_error("rhs does not match the matrix partition");
        const index_t ofs = A.split[c.rank];
        for (const auto &r : got) rhs_local[r.id - ofs] = r.v;
    });
}

saena_transfer_h *saena_prolong_new(saena_comm *comm, index_t Mbig, index_t Nbig, const index_t *split_row, const index_t *split_col,
                                    const index_t *row, const index_t *col, const value_t *val, nnz_t n) {
    std::unique_ptr<saena_transfer_h> h(new saena_transfer_h());
    int s = guard([&] {
        transfer_matrix &T = h->T;
        T.comm = comm->c.get();
        const int np = T.comm->nranks;
        T.Mbig = Mbig; T.Nbig = Nbig;
        T.split_row.assign(split_row, split_row + np + 1);
        T.split_col.assign(split_col, split_col + np + 1);
        T.M = T.split_row[T.comm->rank + 1] - T.split_row[T.comm->rank];
        T.entry.reserve((size_t)n);
        for (nnz_t i = 0; i < n; ++i) T.entry.emplace_back(row[i], col[i], val[i]);
        std::sort(T.entry.begin(), T.entry.end(), col_major);
        T.nnz_l = n;
        T.nnz_g = T.comm->sum(n);
        T.build_layout();
    });
    return s == 0 ? h.release() : nullptr;
}

saena_transfer_h *saena_restrict_from_prolong(saena_transfer_h *P) {
    std::unique_ptr<saena_transfer_h> h(new saena_transfer_h());
    int s = guard([&] { transpose_transfer(P->T, h->T); });
    return s == 0 ? h.release() : nullptr;
}

void saena_transfer_free(saena_transfer_h *T) { delete T; }
int saena_transfer_get_desc(saena_transfer_h *T, sgpu_op_desc *out) { return guard([&] { fill_desc(T->T.L, nullptr, out); }); }
nnz_t saena_transfer_get_local_nnz(saena_transfer_h *T) { return T->T.nnz_l; }

} // extern "C"

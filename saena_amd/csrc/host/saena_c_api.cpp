// saena_c_api.cpp -- flat C view (include/saena_c.h) of the host-side mirror.
#include "../../../include/saena_c.h"
#include "saena_matrix.h"
#include "amg_setup.h"
#include "shm_comm.h"

#include <algorithm>
#include <cstdlib>
#include <memory>
#include <string>

using namespace saena_host;

struct saena_comm { std::unique_ptr<Comm> c; };
struct saena_matrix_h { saena_matrix A; explicit saena_matrix_h(Comm *c) : A(c) {} };
struct saena_transfer_h { transfer_matrix T; };
struct saena_amg_h {
    amg_hierarchy H;
    bool set = false;
    // device side (libsaena_amd.so only)
    std::vector<sgpu_op *> dA, dP, dR;
    sgpu_amg *damg = nullptr;
};

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

saena_comm *saena_comm_shm(const char *name, int rank, int nranks) {
    std::string err;
    std::unique_ptr<Comm> m = make_shm_comm(name ? name : "", rank, nranks, &err);
    if (!m) { h_err = err; return nullptr; }
    auto *c = new saena_comm();
    c->c = std::move(m);
    return c;
}

void saena_comm_free(saena_comm *c) { delete c; }
double saena_measured_chain_us(void) { return g_measured_chain_us; }
// the collectives themselves, for tests of a communicator (byte counts and displacements as in Comm::alltoallv)
int saena_comm_test_alltoallv(saena_comm *c, const void *send, const size_t *sc, const size_t *sd, void *recv, const size_t *rc, const size_t *rd) {
    return guard([&] { c->c->alltoallv(send, sc, sd, recv, rc, rd); });
}
int saena_comm_test_allreduce_f64(saena_comm *c, double *v, int n) { return guard([&] { c->c->allreduce_sum_f64(v, n); }); }
int saena_comm_test_allreduce_i64(saena_comm *c, long *v, int n) { return guard([&] { c->c->allreduce_sum_i64(v, n); }); }

saena_matrix_h *saena_matrix_new(saena_comm *comm) { return comm ? new saena_matrix_h(comm->c.get()) : nullptr; }
void saena_matrix_free(saena_matrix_h *A) { delete A; }
int saena_matrix_set(saena_matrix_h *A, index_t i, index_t j, value_t v) { return guard([&] { A->A.set(i, j, v); }); }
int saena_matrix_set_many(saena_matrix_h *A, const index_t *r, const index_t *c, const value_t *v, nnz_t n) {
    return guard([&] { A->A.set(r, c, v, n); });
}
int saena_matrix_read_file(saena_matrix_h *A, const char *name, const char *type) { return guard([&] { A->A.read_file(name, type ? type : ""); }); }
int saena_matrix_write_bin(saena_matrix_h *A, const char *name) { return guard([&] { A->A.write_bin(name); }); }
int saena_matrix_write_mtx(saena_matrix_h *A, const char *name) { return guard([&] { A->A.writeMatrixToFile(name); }); }
int saena_matrix_set_remove_boundary(saena_matrix_h *A, int b) { A->A.remove_boundary = b != 0; return 0; }
int saena_matrix_set_partition_buckets(saena_matrix_h *A, int n) { if (!A || n < 0) return -1; A->A.partition_buckets = n; return 0; }
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
int saena_matrix_get_halo_columns(saena_matrix_h *A, const index_t **v) {
    if (v) *v = A->A.L.vElement_remote.data();
    return 0;
}
int saena_matrix_get_layout_extra(saena_matrix_h *A, const index_t **col_remote, const nnz_t **scan) {
    if (col_remote) *col_remote = A->A.L.col_remote.data();
    if (scan) *scan = A->A.L.nnzPerProcScan.data();
    return 0;
}

int saena_laplacian3D(saena_matrix_h *A, index_t mx, index_t my, index_t mz) { return guard([&] { laplacian3D(&A->A, mx, my, mz); }); }
int saena_band_matrix(saena_matrix_h *A, index_t M, unsigned int bw) { return guard([&] { band_matrix(&A->A, M, bw); }); }

int saena_matmat(saena_matrix_h *A, saena_matrix_h *B, saena_matrix_h *C) {
    return guard([&] {
        std::vector<cooEntry> e = amg_hierarchy::matmat(A->A, B->A);
        Comm *c = A->A.comm;
        C->A = saena_matrix(c);
        C->A.remove_boundary = false;
        for (const auto &x : e) C->A.set(x.row, x.col, x.val);
        C->A.assemble();
    });
}

int saena_laplacian3D_set_rhs(saena_matrix_h *Ah, index_t mx, index_t my, index_t mz, value_t *rhs_local) {
    return guard([&] {
        saena_matrix &A = Ah->A;
        index_t lo = 0;
        std::vector<value_t> full = laplacian3D_set_rhs(*A.comm, mx, my, mz, &lo);
        std::vector<index_t> idx(full.size());
        for (size_t i = 0; i < full.size(); ++i) idx[i] = lo + (index_t)i;
        std::vector<value_t> mine = A.scatter_rhs(idx, full);          // set_repartition_rhs, saena_object_repart_shrink.cpp:154-218
        std::copy(mine.begin(), mine.end(), rhs_local);
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


int saena_options_default(saena_options_c *o) {
    amg_options d;
    o->solver_max_iter = d.solver_max_iter; o->relative_tol = d.relative_tol; o->smoother = d.smoother == "jacobi" ? 0 : 1;
    o->preSmooth = d.preSmooth; o->postSmooth = d.postSmooth; o->connStrength = d.connStrength;
    o->dynamic_levels = d.dynamic_levels; o->max_level = d.max_level; o->float_level = d.float_level;
    o->filter_thre = d.filter_thre; o->filter_max = d.filter_max; o->filter_start = d.filter_start; o->filter_rate = d.filter_rate;
    o->switch_to_dense = d.switch_to_dense; o->dense_thre = d.dense_thre; o->dense_sz_thre = d.dense_sz_thre;
    return 0;
}

int saena_options_from_file(const char *name, saena_options_c *o) {
    return guard([&] {
        amg_options d;
        d.set_from_file(name);
        o->solver_max_iter = d.solver_max_iter; o->relative_tol = d.relative_tol; o->smoother = d.smoother == "jacobi" ? 0 : 1;
        o->preSmooth = d.preSmooth; o->postSmooth = d.postSmooth; o->connStrength = d.connStrength;
        o->dynamic_levels = d.dynamic_levels; o->max_level = d.max_level; o->float_level = d.float_level;
        o->filter_thre = d.filter_thre; o->filter_max = d.filter_max; o->filter_start = d.filter_start; o->filter_rate = d.filter_rate;
        o->switch_to_dense = d.switch_to_dense; o->dense_thre = d.dense_thre; o->dense_sz_thre = d.dense_sz_thre;
    });
}

saena_amg_h *saena_amg_new(void) { return new saena_amg_h(); }

int saena_amg_set_matrix(saena_amg_h *S, saena_matrix_h *A, const saena_options_c *o) {
    return guard([&] {
        amg_options d;
        if (o) {
            d.solver_max_iter = o->solver_max_iter; d.relative_tol = o->relative_tol; d.smoother = o->smoother == 0 ? "jacobi" : "chebyshev";
            d.preSmooth = o->preSmooth; d.postSmooth = o->postSmooth; d.connStrength = o->connStrength;
            d.dynamic_levels = o->dynamic_levels != 0; d.max_level = o->max_level; d.float_level = o->float_level;
            d.filter_thre = o->filter_thre; d.filter_max = o->filter_max; d.filter_start = o->filter_start; d.filter_rate = o->filter_rate;
            d.switch_to_dense = o->switch_to_dense != 0; d.dense_thre = o->dense_thre; d.dense_sz_thre = o->dense_sz_thre;
        }
        S->H.setup_distributed(&A->A, d);
        S->set = true;
    });
}

int saena_amg_num_levels(saena_amg_h *S) { return S->set ? S->H.max_level + 1 : 0; }

int saena_amg_level_split(saena_amg_h *S, int l, index_t *split_out) {
    return guard([&] {
        if (!S->set || l < 0 || l > S->H.max_level) throw std::runtime_error("bad level");
        if (S->H.dist.empty()) { split_out[0] = 0; split_out[1] = S->H.level_rows(l); }
        else std::copy(S->H.dist[l].split.begin(), S->H.dist[l].split.end(), split_out);
    });
}

int saena_amg_level_aggregates(saena_amg_h *S, int level, index_t *out, index_t *n_aggregates) {
    return guard([&] {
        if (!S->set) throw std::runtime_error("set_matrix has not been called");
        if (!S->H.dist.empty()) throw std::runtime_error("aggregates are kept by the one-rank setup only");
        if (level < 0 || level >= S->H.max_level) throw std::runtime_error("no aggregation on this level");
        const auto &g = S->H.levels[(size_t)level];
        if (out) std::copy(g.agg.begin(), g.agg.end(), out);
        if (n_aggregates) *n_aggregates = (index_t)g.roots.size();
    });
}
int saena_amg_level_info(saena_amg_h *S, int l, index_t *rows, nnz_t *nnzA, nnz_t *nnzP, double *eig) {
    return guard([&] {
        if (!S->set || l < 0 || l > S->H.max_level) throw std::runtime_error("bad level");
        if (rows) *rows = S->H.level_rows(l);
        if (nnzA) *nnzA = S->H.level_nnzA(l);
        if (nnzP) *nnzP = S->H.level_nnzP(l);
        if (eig) *eig = S->H.level_eig(l);
    });
}

int saena_amg_level_desc(saena_amg_h *S, int l, int which, sgpu_op_desc *out) {
    return guard([&] {
        if (!S->set || l < 0 || l > S->H.max_level) throw std::runtime_error("bad level");
        if (which != 0 && l == S->H.max_level) throw std::runtime_error("the coarsest level has no P/R");
        if (!S->H.dist.empty()) {                 // multi-rank: this rank's share
            const dist_level &d = S->H.dist[l];
            if (which == 0) fill_desc(d.A, &d.inv_diag, out);
            else fill_desc(which == 1 ? d.P : d.R, nullptr, out);
            return;
        }
        const amg_level &g = S->H.levels[l];
        if (which == 0) fill_desc(g.A->L, &g.A->inv_diag, out);
        else fill_desc(which == 1 ? g.P.L : g.R.L, nullptr, out);
    });
}

#ifndef SAENA_WITH_GPU
static int no_gpu() { h_err = "this entry point needs libsaena_amd.so (the GPU library); there is no CPU fallback"; return -1; }
void saena_amg_free(saena_amg_h *S) { delete S; }
int saena_amg_to_device(saena_amg_h *) { return no_gpu(); }
sgpu_amg *saena_amg_device_handle(saena_amg_h *) { no_gpu(); return nullptr; }
sgpu_op *saena_amg_device_op(saena_amg_h *, int, int) { no_gpu(); return nullptr; }
int saena_amg_solve(saena_amg_h *, const value_t *, value_t *, int *, value_t *, int) { return no_gpu(); }
int saena_amg_solve_pCG(saena_amg_h *, const value_t *, value_t *, int *, value_t *, int) { return no_gpu(); }
#else
static int gchk(int s) { if (s != 0) { h_err = sgpu_last_error(); } return s; }
static void drop_device(saena_amg_h *S) {
    if (S->damg) { sgpu_amg_destroy(S->damg); S->damg = nullptr; }
    for (auto *v : {&S->dA, &S->dP, &S->dR}) { for (sgpu_op *o : *v) sgpu_op_destroy(o); v->clear(); }
}
void saena_amg_free(saena_amg_h *S) { if (S) { drop_device(S); delete S; } }

int saena_amg_to_device(saena_amg_h *S) {
    if (!S->set) { h_err = "set_matrix has not been called"; return -1; }
    drop_device(S);
    const int n = S->H.max_level + 1;
    std::vector<double> eig;
    for (int l = 0; l < n; ++l) {
        sgpu_op_desc d; sgpu_op *o = nullptr;
        // float_level (saena_object.cpp:241-244,277-285): A of level >= float_level, P/R of level >= float_level
        // exchange their halos in fp32 (compute stays fp64).  Only matters with more than one rank.
        const int fl = S->H.opts.float_level;
        if (saena_amg_level_desc(S, l, 0, &d)) return -1;
        d.halo_fp32 = l >= fl ? 1 : 0;
        if (gchk(sgpu_op_create(&d, &o))) return -2;
        S->dA.push_back(o);
        const bool tune = !std::getenv("SAENA_NO_AUTOTUNE");     // right after each create: the host copy of the values lives only until then
        if (tune && gchk(sgpu_op_autotune(o))) return -2;
        eig.push_back(S->H.level_eig(l));
        if (l < n - 1) {
            if (saena_amg_level_desc(S, l, 1, &d)) return -1;
            d.halo_fp32 = l >= fl ? 1 : 0; if (gchk(sgpu_op_create(&d, &o))) return -2; S->dP.push_back(o);
            if (tune && gchk(sgpu_op_autotune(o))) return -2;
            if (saena_amg_level_desc(S, l, 2, &d)) return -1;
            d.halo_fp32 = l >= fl ? 1 : 0; if (gchk(sgpu_op_create(&d, &o))) return -2; S->dR.push_back(o);
            if (tune && gchk(sgpu_op_autotune(o))) return -2;
        }
    }
    if (S->H.opts.switch_to_dense)            // saena_object_setup2.cpp:328: switch_to_dense && density > dense_thre && Mbig <= dense_sz_thre
        for (int l = 1; l < n; ++l) {
            const double rows = (double)S->H.level_rows(l), dens = (double)S->H.level_nnzA(l) / (rows * rows);
            if (dens > S->H.opts.dense_thre && rows <= S->H.opts.dense_sz_thre) sgpu_op_set_variant(S->dA[(size_t)l], 5);   // refused (halo, size): stays sparse
        }
    sgpu_amg_params p;
    sgpu_amg_default_params(&p);
    const amg_options &o = S->H.opts;
    p.preSmooth = o.preSmooth; p.postSmooth = o.postSmooth; p.smoother = o.smoother == "jacobi" ? 0 : 1;
    p.solver_max_iter = o.solver_max_iter; p.solver_tol = o.relative_tol;
    return gchk(sgpu_amg_create(n, S->dA.data(), S->dP.data(), S->dR.data(), eig.data(), &p, &S->damg));
}

sgpu_amg *saena_amg_device_handle(saena_amg_h *S) { return S->damg; }
sgpu_op *saena_amg_device_op(saena_amg_h *S, int l, int which) {
    auto &v = which == 0 ? S->dA : which == 1 ? S->dP : S->dR;
    return l >= 0 && l < (int)v.size() ? v[l] : nullptr;
}

static int solve_host(saena_amg_h *S, bool pcg, const value_t *rhs_host, value_t *u_host, int *iters, value_t *hist, int cap) {
    if (!S->damg) { h_err = "saena_amg_to_device has not been called"; return -1; }
    const size_t n = S->H.dist.empty() ? (size_t)S->H.levels[0].A->M : (size_t)S->H.dist[0].A.M;
    value_t *u = nullptr, *rhs = nullptr;
    if (gchk(sgpu_vec_alloc(&u, n)) || gchk(sgpu_vec_alloc(&rhs, n))) return -2;
    int s = gchk(sgpu_vec_upload(rhs, rhs_host, n));
    if (!s) s = gchk(pcg ? sgpu_solve_pCG(S->damg, u, rhs, iters, hist, cap) : sgpu_solve(S->damg, u, rhs, iters, hist, cap));
    const int s2 = gchk(sgpu_vec_download(u_host, u, n));
    sgpu_vec_free(u); sgpu_vec_free(rhs);
    return s ? s : s2;
}
int saena_amg_solve(saena_amg_h *S, const value_t *rhs, value_t *u, int *it, value_t *hist, int cap) { return solve_host(S, false, rhs, u, it, hist, cap); }
int saena_amg_solve_pCG(saena_amg_h *S, const value_t *rhs, value_t *u, int *it, value_t *hist, int cap) { return solve_host(S, true, rhs, u, it, hist, cap); }
#endif

} // extern "C"

// saena.cpp -- implementation of the public C++ surface (include/saena.hpp) over the host mirror
// and the C ABI of the GPU library.  Built into libsaena_amd.so only.
#ifdef SAENA_WITH_GPU
#include "../../../include/saena.hpp"
#include "../../../include/saena_gpu.h"
#include "../../../include/saena_gpu_debug.h"
#include "comm.h"
#include "amg_setup.h"
#include "saena_matrix.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <sstream>
#include <stdexcept>

extern "C" saena_host::Comm *sgpu_new_host_comm();

namespace {
void gchk(int s, const char *what) {
    if (s != SGPU_OK) throw std::runtime_error(std::string(what) + ": " + sgpu_last_error());
}
saena_host::Comm *&world_override() { static saena_host::Comm *o = nullptr; return o; }      // saena::init_host_transport
saena_host::Comm *world() {
    static saena_host::Comm *w = nullptr;
    if (world_override()) return world_override();
    if (!w) {
        w = sgpu_new_host_comm();
        if (!w) throw std::runtime_error("saena::init() has not been called (no MI355X context)");
    }
    return w;
}
void fill_desc(const saena_host::DistLayout &L, const std::vector<value_t> *inv_diag, sgpu_op_desc *d) {
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

namespace saena {

void init(int device_id, int rank, int nranks, const void *uid) { gchk(sgpu_init(device_id, rank, nranks, uid), "saena::init"); }
void unique_id(void *out128) { gchk(sgpu_get_unique_id(out128), "saena::unique_id"); }
void init_host_transport(int device_id, int rank, int nranks, const host_transport &t) {
    if (!t.exchange || !t.allreduce_sum || !t.allgather || !t.alltoallv || !t.allreduce_i64 || !t.allreduce_f64)
        throw std::runtime_error("saena::init_host_transport: every callback is needed");
    gchk(sgpu_debug_init_host_transport(device_id, rank, nranks, t.exchange, t.allreduce_sum, t.user), "saena::init_host_transport");
    auto *cb = new saena_host::CallbackComm();
    cb->rank = rank; cb->nranks = nranks; cb->user = t.user;
    cb->cb_allgather = t.allgather; cb->cb_alltoallv = t.alltoallv; cb->cb_i64 = t.allreduce_i64; cb->cb_f64 = t.allreduce_f64;
    world_override() = cb;
}
void finalize() { sgpu_finalize(); }

comm::comm() : c_(world()) {}
int comm::rank() const { return c_->rank; }
int comm::size() const { return c_->nranks; }

// ---------------------------------------------------------------- matrix
matrix::matrix() : c_(), m_pImpl(new saena_host::saena_matrix(c_.impl())) {}
matrix::matrix(comm c) : c_(c), m_pImpl(new saena_host::saena_matrix(c.impl())) {}
matrix::~matrix() { destroy(); }
// copies (reference saena.cpp:14-31): the host-side matrix is copied whole; the copy builds its own device operator on first use
matrix::matrix(const matrix &B) : add_dup(B.add_dup), c_(B.c_), m_pImpl(B.m_pImpl ? new saena_host::saena_matrix(*B.m_pImpl) : nullptr), use_dense_(B.use_dense_) {}
matrix &matrix::operator=(const matrix &B) {
    if (this != &B) {
        destroy();
        c_ = B.c_;
        m_pImpl = B.m_pImpl ? new saena_host::saena_matrix(*B.m_pImpl) : nullptr;
        add_dup = B.add_dup;
        use_dense_ = B.use_dense_;
    }
    return *this;
}
int matrix::read_file(const char *name) { return m_pImpl->read_file(name, ""); }
int matrix::read_file(const char *name, const std::string &t) { return m_pImpl->read_file(name, t); }
void matrix::set_comm(comm c) { c_ = c; m_pImpl->comm = c.impl(); }
int matrix::set(index_t i, index_t j, value_t val) { return m_pImpl->set(i, j, val); }
int matrix::set(index_t *row, index_t *col, value_t *val, nnz_t n) { return m_pImpl->set(row, col, val, n); }
int matrix::set(index_t i, index_t j, unsigned int size_x, unsigned int size_y, value_t *val) {
    // contiguous block, row-major values (saena_matrix.cpp set(i,j,size_x,size_y,val))
    for (unsigned int a = 0; a < size_x; ++a)
        for (unsigned int b = 0; b < size_y; ++b) m_pImpl->set(i + (index_t)a, j + (index_t)b, val[a * size_y + b]);
    return 0;
}
int matrix::set(index_t i, index_t j, unsigned int *di, unsigned int *dj, value_t *val, nnz_t n) {
    // generic block (reference saena.cpp:100-112): entry k at (i + di[k], j + dj[k]); exact zeros are skipped
    for (nnz_t k = 0; k < n; ++k)
        if (val[k] != 0) m_pImpl->set(i + (index_t)di[k], j + (index_t)dj[k], val[k]);
    return 0;
}
void matrix::set_eig(double e) { m_pImpl->set_eig(e); }
void matrix::set_eig(const std::string &fname) {          // saena.cpp:124-135: optional eig="..." attribute
    std::ifstream f(fname);
    if (!f) throw std::runtime_error("Could not find the xml file!");
    std::stringstream ss; ss << f.rdbuf();
    const std::string s = ss.str();
    const size_t p = s.find("eig=\"");
    if (p == std::string::npos) return;
    const size_t q = s.find('"', p + 5);
    const double e = std::stod(s.substr(p + 5, q - p - 5));
    if (e != 0.0) m_pImpl->set_eig(e);
}
void matrix::set_remove_boundary(bool b) { m_pImpl->remove_boundary = b; }
void matrix::set_partition_buckets(int n) { m_pImpl->partition_buckets = n < 0 ? 0 : n; }
int matrix::add_duplicates(bool add) { add_dup = add; m_pImpl->add_duplicates = add; return 0; }
int matrix::assemble(bool scale, bool use_dense) {
    // scale: the symmetric diagonal scaling is not functional in the reference either (its solvers read inv_sq_diag_orig, which only
    // scale_matrix(full_scale = true) fills and no call site passes true: DESIGN.md 9); its drivers pass false
    if (scale) throw std::runtime_error("saena::matrix::assemble: scale = true is not on the GPU path (false in the reference's drivers)");
    // use_dense (saena_matrix::use_dense -> saena_matrix_dense, src/saena_matrix_dense.cpp:181-260): this matrix's device operator is
    // stored as dense row-major rows (k_dense_rows / k_dense_rows_halo); refused at first use when it is too large for that
    use_dense_ = use_dense;
    m_pImpl->add_duplicates = add_dup;
    return m_pImpl->assemble();
}
int matrix::assemble_band_matrix(bool use_dense) { return assemble(false, use_dense); }
int matrix::print(int ran, std::string name) {          // saena_matrix::print_entry
    if (ran >= 0 && ran != c_.rank()) return 0;
    printf("\nmatrix %s on rank %d: %ld entries\n", name.c_str(), c_.rank(), (long)m_pImpl->entry.size());
    for (const auto &e : m_pImpl->entry) printf("%d\t%d\t%.12g\n", e.row, e.col, e.val);
    return 0;
}
int matrix::assemble_writeToFile(const char *folder_name) {   // reference saena.cpp:157-172
    if (!m_pImpl->assembled) assemble();
    return writeMatrixToFile(folder_name ? folder_name : "");
}
int matrix::writeMatrixToFile(const std::string &name) const { return m_pImpl->writeMatrixToFile(name.empty() ? "mat" : name); }
saena_host::saena_matrix *matrix::get_internal_matrix() { return m_pImpl; }
comm matrix::get_comm() { return c_; }
index_t matrix::get_num_rows() { return m_pImpl->remove_boundary ? m_pImpl->Mbig_with_bound : m_pImpl->Mbig; }   // saena.cpp:188: size before removing boundary
index_t matrix::get_num_local_rows() { return m_pImpl->M; }
nnz_t matrix::get_nnz() { return m_pImpl->nnz_g; }
nnz_t matrix::get_local_nnz() { return m_pImpl->nnz_l; }
std::vector<index_t> matrix::get_orig_split() { return m_pImpl->split; }
std::vector<index_t> matrix::get_split() { return m_pImpl->split; }

sgpu_op *matrix::device_op() {
    if (!dev_) {
        if (!m_pImpl->assembled) throw std::runtime_error("saena::matrix: assemble() first");
        sgpu_op_desc d;
        fill_desc(m_pImpl->L, &m_pImpl->inv_diag, &d);
        gchk(sgpu_op_create(&d, &dev_), "sgpu_op_create");
        if (use_dense_) gchk(sgpu_op_set_variant(dev_, 5), "saena::matrix::assemble(use_dense = true)");     // dense rows (<= 8192 rows, 64 M entries per rank)
    }
    return dev_;
}
void matrix::matvec(std::vector<value_t> &v, std::vector<value_t> &w) {
    if ((index_t)v.size() != m_pImpl->M) throw std::runtime_error("saena::matrix::matvec: v must hold this rank's rows");
    w.resize(v.size());
    gchk(sgpu_spmv_host(device_op(), v.data(), w.data()), "sgpu_spmv_host");
}
void matrix::matvec(saena::vector &v, saena::vector &w) {
    value_t *p = nullptr;
    v.get_vec(p);
    std::vector<value_t> vin(p, p + v.get_size()), wout;
    matvec(vin, wout);
    w.set(wout.data(), (index_t)wout.size(), m_pImpl->split[c_.rank()]);
    w.assemble();
}
int matrix::erase() {
    if (dev_) { sgpu_op_destroy(dev_); dev_ = nullptr; }
    saena_host::Comm *c = m_pImpl->comm;
    delete m_pImpl;
    m_pImpl = new saena_host::saena_matrix(c);
    return 0;
}
void matrix::destroy() {
    if (dev_) { sgpu_op_destroy(dev_); dev_ = nullptr; }
    delete m_pImpl;
    m_pImpl = nullptr;
}

// ---------------------------------------------------------------- vector
vector::vector() : c_() {}
vector::vector(comm c) : c_(c) {}
void vector::set_comm(comm c) { c_ = c; }
int vector::set_idx_offset(index_t o) { ofst_ = o; return 0; }
int vector::set(index_t i, value_t v) { idx_.push_back(i + ofst_); val_.push_back(v); assembled_ = false; return 0; }
int vector::set(const index_t *idx, const value_t *val, index_t n) { for (index_t k = 0; k < n; ++k) set(idx[k], val[k]); return 0; }
int vector::set(const value_t *val, index_t n, index_t offset) {
    for (index_t k = 0; k < n; ++k) { idx_.push_back(offset + k); val_.push_back(val[k]); }
    assembled_ = false;
    return 0;
}
int vector::set(const value_t *val, index_t n) { return set(val, n, ofst_); }
int vector::set_dup_flag(bool add) { add_dup_ = add; return 0; }
int vector::assemble() {            // saena_vector::assemble: sort by index, combine duplicates
    std::vector<size_t> order(idx_.size());
    for (size_t i = 0; i < order.size(); ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [this](size_t a, size_t b) { return idx_[a] < idx_[b]; });
    std::vector<index_t> ni; std::vector<value_t> nv;
    for (size_t k = 0; k < order.size(); ++k) {
        const index_t i = idx_[order[k]];
        value_t v = val_[order[k]];
        while (k + 1 < order.size() && idx_[order[k + 1]] == i) { ++k; v = add_dup_ ? v + val_[order[k]] : val_[order[k]]; }
        ni.push_back(i); nv.push_back(v);
    }
    idx_.swap(ni); val_.swap(nv);
    assembled_ = true;
    return 0;
}
void vector::get_vec(value_t *&vec) { if (!assembled_) assemble(); vec = val_.data(); }

// ---------------------------------------------------------------- options
options::options(int max_iter, double tol, std::string sm, int pre, int post, std::string PSm, float conn, bool dyn, int max_lev,
                 int float_lev, double fil_thr, double fil_max, int fil_st, int fil_rate, bool sw_dense, float dense_thr, int dense_sz) {
    set(max_iter, tol, std::move(sm), pre, post, std::move(PSm), conn, dyn, max_lev, float_lev, fil_thr, fil_max, fil_st, fil_rate, sw_dense, dense_thr, dense_sz);
}
options::options(const std::string &name) { set(); set_from_file(name); }
void options::set(int max_iter, double tol, std::string sm, int pre, int post, std::string PSm, float conn, bool dyn, int max_lev,
                  int float_lev, double fil_thr, double fil_max, int fil_st, int fil_rate, bool sw_dense, float dense_thr, int dense_sz) {
    solver_max_iter = max_iter; relative_tol = tol; smoother = std::move(sm); preSmooth = pre; postSmooth = post; PSmoother = std::move(PSm);
    connStrength = conn; dynamic_levels = dyn; max_level = max_lev; float_level = float_lev; filter_thre = fil_thr; filter_max = fil_max;
    filter_start = fil_st; filter_rate = fil_rate; switch_to_dense = sw_dense; dense_thre = dense_thr; dense_sz_thre = dense_sz;
}
void options::set_from_file(const std::string &name) {
    saena_host::amg_options o;
    o.set_from_file(name);
    set(o.solver_max_iter, o.relative_tol, o.smoother, o.preSmooth, o.postSmooth, o.PSmoother, o.connStrength, o.dynamic_levels, o.max_level,
        o.float_level, o.filter_thre, o.filter_max, o.filter_start, o.filter_rate, o.switch_to_dense, o.dense_thre, o.dense_sz_thre);
}
void options::set_solve_params(int max_iter, double tol, std::string sm, int pre, int post) {
    solver_max_iter = max_iter; relative_tol = tol; smoother = std::move(sm); preSmooth = pre; postSmooth = post;
}

// ---------------------------------------------------------------- amg
amg::amg() {}
amg::~amg() { destroy(); }
void amg::set_dynamic_levels(const bool &dl) { dynamic_levels_ = dl; }
int amg::set_verbose(bool v) { verbose = v; return 0; }
int amg::set_multigrid_max_level(int m) { max_level_override_ = m; return 0; }
int amg::get_num_levels() const { return H_ ? H_->max_level + 1 : 0; }
void amg::drop_device() {
    if (damg_) { sgpu_amg_destroy(damg_); damg_ = nullptr; }
    for (auto *v : {&dA_, &dP_, &dR_}) {
        for (size_t i = 0; i < v->size(); ++i)
            if (!(v == &dA_ && i == 0)) sgpu_op_destroy((*v)[i]);      // level-0 A belongs to the saena::matrix
        v->clear();
    }
}
void amg::destroy() {
    drop_device();
    delete H_; H_ = nullptr;
}

int amg::set_matrix(saena::matrix *A, saena::options *opts) {
    destroy();
    A_ = A;
    saena_host::amg_options o;
    o.solver_max_iter = opts->get_max_iter(); o.relative_tol = opts->get_tol(); o.smoother = opts->get_smoother();
    o.preSmooth = opts->get_preSmooth(); o.postSmooth = opts->get_postSmooth(); o.PSmoother = opts->get_PSmoother();
    o.connStrength = opts->get_connStr(); o.dynamic_levels = opts->get_dynamic_levels() && dynamic_levels_;
    o.max_level = max_level_override_ >= 0 ? max_level_override_ : opts->get_max_lev(); o.float_level = opts->get_float_lev();
    o.filter_thre = opts->get_filter_thre(); o.filter_max = opts->get_filter_max(); o.filter_start = opts->get_filter_start();
    o.filter_rate = opts->get_filter_rate();
    o.switch_to_dense = opts->get_switch_dense() || switch_to_dense_; o.dense_thre = opts->get_dense_thre(); o.dense_sz_thre = opts->get_dense_sz_thre();
    if (dense_thre_override_ > 0) o.dense_thre = dense_thre_override_;
    H_ = new saena_host::amg_hierarchy();
    H_->setup_distributed(A->get_internal_matrix(), o);      // one rank: plain setup; more: every rank builds its rows of every level
    const int n = H_->max_level + 1;
    const bool multi = !H_->dist.empty();
    std::vector<double> eig;
    for (int l = 0; l < n; ++l) {
        const saena_host::amg_level *gp = multi ? nullptr : &H_->levels[l];
        sgpu_op_desc d; sgpu_op *op = nullptr;
        const int f32 = l >= o.float_level ? 1 : 0;          // float_level semantics (saena_object.cpp:241-244,277-285)
        if (l == 0) op = A->device_op();
        else {
            if (multi) fill_desc(H_->dist[l].A, &H_->dist[l].inv_diag, &d); else fill_desc(gp->A->L, &gp->A->inv_diag, &d);
            d.halo_fp32 = f32;
            gchk(sgpu_op_create(&d, &op), "sgpu_op_create(A)");
        }
        // plan-time autotune right after each create: the operator's host copy of the values lives only until then
        const bool tune = !std::getenv("SAENA_NO_AUTOTUNE");
        if (tune) gchk(sgpu_op_autotune(op), "sgpu_op_autotune");
        dA_.push_back(op);
        eig.push_back(H_->level_eig(l));
        if (l < n - 1) {
            fill_desc(multi ? H_->dist[l].P : gp->P.L, nullptr, &d); d.halo_fp32 = f32; gchk(sgpu_op_create(&d, &op), "sgpu_op_create(P)"); dP_.push_back(op);
            if (tune) gchk(sgpu_op_autotune(op), "sgpu_op_autotune");
            fill_desc(multi ? H_->dist[l].R : gp->R.L, nullptr, &d); d.halo_fp32 = f32; gchk(sgpu_op_create(&d, &op), "sgpu_op_create(R)"); dR_.push_back(op);
            if (tune) gchk(sgpu_op_autotune(op), "sgpu_op_autotune");
        }
    }
    if (o.switch_to_dense)                    // saena_object_setup2.cpp:328
        for (int l = 1; l < n; ++l) {
            const double rows = (double)H_->level_rows(l), dens = (double)H_->level_nnzA(l) / (rows * rows);
            if (dens > o.dense_thre && rows <= o.dense_sz_thre) sgpu_op_set_variant(dA_[(size_t)l], 5);     // refused (halo, size): stays sparse
        }
    sgpu_amg_params p;
    sgpu_amg_default_params(&p);
    p.preSmooth = o.preSmooth; p.postSmooth = o.postSmooth; p.smoother = o.smoother == "jacobi" ? 0 : 1;
    p.solver_max_iter = o.solver_max_iter; p.solver_tol = o.relative_tol;
    gchk(sgpu_amg_create(n, dA_.data(), dP_.data(), dR_.data(), eig.data(), &p, &damg_), "sgpu_amg_create");
    if (verbose && A->get_comm().rank() == 0) {
        printf("_____________________________\n\nnumber of levels = << %d >> (the finest level is 0)\n", n - 1);
        for (int l = 0; l < n; ++l) printf("level %d: rows %d, nnz %ld\n", l, H_->level_rows(l), (long)H_->level_nnzA(l));
    }
    return 0;
}

int vector::print_entry(int ran) {
    if (ran < 0 || ran == c_.rank()) {
        printf("vector on proc %d, size %zu\n", c_.rank(), val_.size());
        for (size_t i = 0; i < val_.size(); ++i) printf("%zu \t%d \t%.14g\n", i, (int)idx_[i], val_[i]);
    }
    return 0;
}
int amg::set_rhs(saena::vector &rhs) {
    if (!A_) throw std::runtime_error("saena::amg::set_rhs: set_matrix first");
    value_t *p = nullptr;
    rhs.get_vec(p);
    std::vector<value_t> vals(p, p + rhs.get_size());
    rhs_ = A_->get_internal_matrix()->scatter_rhs(rhs.indices(), vals);
    return 0;
}
int amg::set_rhs(const value_t *rhs_local, index_t size) {
    if (!A_ || size != A_->get_num_local_rows()) throw std::runtime_error("saena::amg::set_rhs: size does not match the local rows");
    rhs_.assign(rhs_local, rhs_local + size);
    return 0;
}

int amg::run(value_t *&u, saena::options *opts, int which, bool print_info) {
    if (!damg_) throw std::runtime_error("saena::amg: set_matrix first");
    const size_t n = (size_t)A_->get_num_local_rows();
    if (rhs_.size() != n) throw std::runtime_error("saena::amg: set_rhs first");
    if (opts)      // saena.cpp:751-790: every solve* first re-reads the solve parameters from the options
        gchk(sgpu_amg_set_solve_params(damg_, opts->get_max_iter(), opts->get_tol(), opts->get_smoother() == "jacobi" ? 0 : 1,
                                       opts->get_preSmooth(), opts->get_postSmooth()), "set_solve_params");
    value_t *du = nullptr, *dr = nullptr;
    gchk(sgpu_vec_alloc(&du, n), "alloc"); gchk(sgpu_vec_alloc(&dr, n), "alloc");
    gchk(sgpu_vec_upload(dr, rhs_.data(), n), "upload");
    hist_.assign(4096, 0.0);
    int st = which == 1 ? sgpu_solve_pCG(damg_, du, dr, &iters_, hist_.data(), (int)hist_.size())
           : which == 2 ? sgpu_solve_CG(damg_, du, dr, &iters_, hist_.data(), (int)hist_.size())
           : which == 3 ? sgpu_solve_smoother(damg_, du, dr, &iters_, hist_.data(), (int)hist_.size())
                        : sgpu_solve(damg_, du, dr, &iters_, hist_.data(), (int)hist_.size());
    if (st != SGPU_OK && st != SGPU_ERR_NOCONV) { sgpu_vec_free(du); sgpu_vec_free(dr); gchk(st, "solve"); }
    hist_.resize((size_t)std::min<int>(iters_ + 1, 4096));
    if (!u) u = static_cast<value_t *>(std::malloc(std::max<size_t>(1, n) * sizeof(value_t)));   // saena_aligned_alloc in the reference
    gchk(sgpu_vec_download(u, du, n), "download");
    sgpu_vec_free(du); sgpu_vec_free(dr);
    if (print_info && A_->get_comm().rank() == 0) {           // saena_object_solve.cpp:2502,2681-2682
        printf("\ninitial residual        = %e \n", hist_.front());
        printf("stopped at iteration    = %d \nfinal absolute residual = %e\nrelative residual       = %e \n",
               iters_, hist_.back(), hist_.back() / hist_.front());
    }
    return st == SGPU_OK ? 0 : 1;
}
int amg::solve(value_t *&u, saena::options *opts) { return run(u, opts, 0, true); }
int amg::solve_pCG(value_t *&u, saena::options *opts, bool print_info) { return run(u, opts, 1, print_info); }
int amg::solve_CG(value_t *&u, saena::options *opts) { return run(u, opts, 2, true); }
int amg::set_scale(bool sc) {
    if (sc) throw std::runtime_error("saena::amg::set_scale(true): symmetric scaling is not on the GPU path (false in the reference's drivers)");
    return 0;
}
// saena::amg::matrix_diff (saena.cpp:913-948)
int amg::matrix_diff(saena::matrix &A1, saena::matrix &B1) {
    const auto &ea = A1.get_internal_matrix()->entry, &eb = B1.get_internal_matrix()->entry;
    if (A1.get_nnz() != B1.get_nnz() && A1.get_comm().rank() == 0) printf("error: matrix_diff(): A.nnz_g != B.nnz_g\n");
    printf("\nmatrix_diff: \n");
    for (size_t i = 0; i < std::min(ea.size(), eb.size()); ++i)
        printf("%d\t%d\t%.12g\t%d\t%d\t%.12g\t%.12g\n", ea[i].row, ea[i].col, ea[i].val, eb[i].row, eb[i].col, eb[i].val, ea[i].val - eb[i].val);
    printf("A->entry.size() = %lu, B->entry.size() = %lu \n", (unsigned long)ea.size(), (unsigned long)eb.size());
    return 0;
}
int amg::switch_to_dense(bool val) { switch_to_dense_ = val; return 0; }                 // saena.cpp:729-743
int amg::set_dense_threshold(float thre) { dense_thre_override_ = thre; return 0; }
double amg::get_dense_threshold() { return dense_thre_override_ > 0 ? dense_thre_override_ : 0.1; }
int amg::solve_smoother(value_t *&u, saena::options *opts) { return run(u, opts, 3, true); }

// saena_object::matmat + matmat_assemble (saena_object_setup_matmat.cpp:1164-1487,1640-1708)
void amg::matmat(saena::matrix *A, saena::matrix *B, saena::matrix *C, bool assemble, bool print_timing) {
    if (!A || !B || !C) throw std::runtime_error("saena::amg::matmat: null matrix");
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<saena_host::cooEntry> e = saena_host::amg_hierarchy::matmat(*A->get_internal_matrix(), *B->get_internal_matrix());
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    C->erase();
    C->set_remove_boundary(false);
    for (const auto &x : e) C->set(x.row, x.col, x.val);
    if (assemble) C->assemble();
    if (print_timing && A->get_comm().rank() == 0) printf("matmat: %e s\n", dt);
}

// saena_object::profile_matvecs (saena_object.cpp:618-638)
void amg::profile_matvecs() {
    if (!damg_) throw std::runtime_error("saena::amg: set_matrix first");
    std::vector<double> us((size_t)get_num_levels(), 0.0);
    gchk(sgpu_amg_profile_matvecs(damg_, 5, us.data()), "profile_matvecs");
    if (A_->get_comm().rank() == 0)
        for (size_t l = 0; l < us.size(); ++l) printf("matvec level %zu: %e s\n", l, us[l] * 1e-6);
}

// solve_pCG_profile (saena_object_solve.cpp: the reference prints the time of every phase of its CPU loop): here the solve runs
// as device graphs, so what there is to print is the solve itself -- iterations, time, time per iteration
int amg::solve_pCG_profile(value_t *&u, saena::options *opts) {
    const auto t0 = std::chrono::steady_clock::now();
    const int st = run(u, opts, 1, false);
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (A_ && A_->get_comm().rank() == 0)
        printf("solve_pCG profile: %d iterations in %e s (%e s per iteration: one V-cycle, one fine matvec, the dots), incl. upload / download of u\n",
               iters_, dt, iters_ ? dt / iters_ : 0.0);
    return st;
}
int amg::solve_petsc(value_t *&, saena::options *) {
    fprintf(stderr, "saena::amg::solve_petsc: the PETSc bridge is not part of the MI355X path (SURVEY.md 2: out of scope)\n");
    return 1;
}
// solve_GMRES / solve_pGMRES (saena.cpp:818-835 -> saena_object::GMRES / pGMRES, whose bodies sit inside `#if 0`, saena_object_solve.cpp:3808 /
// 4077): set_solve_params, then nothing
int amg::gmres_compiled_out(const char *name, value_t *&u, saena::options *opts) {
    if (!damg_) throw std::runtime_error("saena::amg: set_matrix first");
    if (opts)
        gchk(sgpu_amg_set_solve_params(damg_, opts->get_max_iter(), opts->get_tol(), opts->get_smoother() == "jacobi" ? 0 : 1,
                                       opts->get_preSmooth(), opts->get_postSmooth()), "set_solve_params");
    const size_t n = (size_t)A_->get_num_local_rows();
    if (!u) u = static_cast<value_t *>(std::calloc(std::max<size_t>(1, n), sizeof(value_t)));
    if (A_->get_comm().rank() == 0)
        printf("saena::amg::%s: the reference compiles this solver out (#if 0 in saena_object_solve.cpp) and returns 0 with u untouched; so does this\n", name);
    return 0;
}
int amg::solve_GMRES(value_t *&u, saena::options *opts) { return gmres_compiled_out("solve_GMRES", u, opts); }
int amg::solve_pGMRES(value_t *&u, saena::options *opts) { return gmres_compiled_out("solve_pGMRES", u, opts); }
comm amg::get_orig_comm() { return A_ ? A_->get_comm() : comm(); }

void free_vector(value_t *u) { std::free(u); }

// ---------------------------------------------------------------- generators
int laplacian3D(saena::matrix *A, index_t mx, index_t my, index_t mz) { return saena_host::laplacian3D(A->get_internal_matrix(), mx, my, mz); }
index_t laplacian3D_set_rhs(value_t *&rhs, index_t mx, index_t my, index_t mz, comm c, index_t *first_index) {
    index_t lo = 0;
    std::vector<value_t> v = saena_host::laplacian3D_set_rhs(*c.impl(), mx, my, mz, &lo);
    rhs = static_cast<value_t *>(std::malloc(std::max<size_t>(1, v.size()) * sizeof(value_t)));
    std::copy(v.begin(), v.end(), rhs);
    if (first_index) *first_index = lo;
    return (index_t)v.size();
}
int band_matrix(saena::matrix &A, index_t M, unsigned int bandwidth) {
    saena_host::band_matrix(A.get_internal_matrix(), M, bandwidth);
    return A.assemble_band_matrix();
}

} // namespace saena

#endif // SAENA_WITH_GPU

// amg_setup.cpp -- smoothed-aggregation setup on the host (restatement; citations are
// file:line in paralab/Saena).  Single-rank SpGEMM in this round: the hierarchy of a
// multi-rank run must be built at one rank (amg_hierarchy::setup throws otherwise).
#include "amg_setup.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <fstream>
#include <sstream>

namespace saena_host {

// ---------------------------------------------------------------------------
// options XML: attributes of <OPTIONS> are read POSITIONALLY (saena.cpp:444-546)
void amg_options::set_from_file(const std::string &name) {
    std::ifstream f(name);
    if (!f) throw std::runtime_error("Could not find the xml file!");
    std::stringstream ss;
    ss << f.rdbuf();
    const std::string s = ss.str();
    size_t p = s.find("<OPTIONS");
    if (p == std::string::npos) throw std::runtime_error("options xml: no <OPTIONS> element");
    const size_t end = s.find('>', p);
    std::vector<std::string> vals;
    while (true) {
        const size_t q1 = s.find('"', p);
        if (q1 == std::string::npos || q1 > end) break;
        const size_t q2 = s.find('"', q1 + 1);
        vals.push_back(s.substr(q1 + 1, q2 - q1 - 1));
        p = q2 + 1;
    }
    if (vals.size() < 17) throw std::runtime_error("options xml: expected at least 17 attributes");
    size_t i = 0;
    solver_max_iter = std::stoi(vals[i++]);
    relative_tol    = std::stod(vals[i++]);
    smoother        = vals[i++];
    preSmooth       = std::stoi(vals[i++]);
    postSmooth      = std::stoi(vals[i++]);
    PSmoother       = vals[i++];
    connStrength    = std::stof(vals[i++]);
    dynamic_levels  = std::stoi(vals[i++]) != 0;
    max_level       = std::stoi(vals[i++]);
    float_level     = std::stoi(vals[i++]);
    filter_thre     = std::stod(vals[i++]);
    filter_max      = std::stod(vals[i++]);
    filter_start    = std::stoi(vals[i++]);
    filter_rate     = std::stoi(vals[i++]);
    switch_to_dense = std::stoi(vals[i++]) != 0;
    dense_thre      = std::stof(vals[i++]);
    dense_sz_thre   = std::stoi(vals[i++]);
    if (smoother != "jacobi" && smoother != "chebyshev") throw std::runtime_error("options xml: unknown smoother " + smoother);
    if (filter_start < 1) throw std::runtime_error("error: filter_start cannot filter level 0. it should be >= 1");
}

// ---------------------------------------------------------------------------
namespace {

struct Csr {
    index_t nrows = 0, ncols = 0;
    std::vector<nnz_t>   ptr;
    std::vector<index_t> col;
    std::vector<value_t> val;
};

// rows [row_ofs, row_ofs+nrows), columns global; `e` in any order
Csr csr_from_entries(const std::vector<cooEntry> &e, index_t row_ofs, index_t nrows, index_t ncols) {
    Csr C;
    C.nrows = nrows; C.ncols = ncols;
    C.ptr.assign((size_t)nrows + 1, 0);
    for (const auto &x : e) C.ptr[x.row - row_ofs + 1]++;
    for (index_t i = 0; i < nrows; ++i) C.ptr[i + 1] += C.ptr[i];
    C.col.resize(e.size()); C.val.resize(e.size());
    std::vector<nnz_t> fill(C.ptr.begin(), C.ptr.end() - 1);
    std::vector<cooEntry> s(e);
    std::sort(s.begin(), s.end(), row_major);
    for (const auto &x : s) { const nnz_t k = fill[x.row - row_ofs]++; C.col[k] = x.col; C.val[k] = x.val; }
    return C;
}

// C = A B (Gustavson, rows of C sorted by column).  Entries with |v| <= ALMOST_ZERO are dropped
// unless row id == column id, the rule of the reference's SpGEMM output (saena_object_setup_matmat.cpp:2423,2442).
std::vector<cooEntry> spgemm_entries(const Csr &A, const Csr &B, index_t row_ofs) {
    std::vector<cooEntry> out;
    std::vector<value_t> acc((size_t)B.ncols, 0.0);
    std::vector<char>    mark((size_t)B.ncols, 0);
    std::vector<index_t> cols;
    for (index_t i = 0; i < A.nrows; ++i) {
        cols.clear();
        for (nnz_t ka = A.ptr[i]; ka < A.ptr[i + 1]; ++ka) {
            const index_t k = A.col[ka];
            const value_t a = A.val[ka];
            for (nnz_t kb = B.ptr[k]; kb < B.ptr[k + 1]; ++kb) {
                const index_t j = B.col[kb];
                if (!mark[j]) { mark[j] = 1; cols.push_back(j); acc[j] = 0.0; }
                acc[j] += a * B.val[kb];
            }
        }
        std::sort(cols.begin(), cols.end());
        const index_t r = i + row_ofs;
        for (index_t j : cols) {
            if (std::fabs(acc[j]) > SAENA_ALMOST_ZERO || r == j) out.emplace_back(r, j, acc[j]);
            mark[j] = 0;
        }
    }
    return out;
}

} // namespace

// ---------------------------------------------------------------------------
// strength of connection (setup1:520-719) + threshold (strength_matrix.cpp:242-258)
void amg_hierarchy::strength_graph(const saena_matrix &A, float connStrength, std::vector<nnz_t> &ptr, std::vector<index_t> &col) {
    if (A.comm->nranks != 1) throw std::runtime_error("strength_graph: multi-rank setup is not implemented in this round");
    const index_t M = A.M;
    std::vector<value_t> maxPerRow((size_t)M, -DBL_MAX);                       // :527-533
    for (const auto &e : A.entry)
        if (e.row != e.col) maxPerRow[e.row] = std::max(maxPerRow[e.row], -e.val);
    // S(i,j) = -a_ij / max_k(-a_ik), S^T(i,j) = -a_ij / max_k(-a_jk), diagonal 1; keep if either > connStrength
    ptr.assign((size_t)M + 1, 0);
    std::vector<cooEntry> kept;
    kept.reserve(A.entry.size());
    for (const auto &e : A.entry) {
        value_t s, st;
        if (e.row == e.col) { s = 1; st = 1; }
        else { s = -e.val / maxPerRow[e.row]; st = -e.val / maxPerRow[e.col]; }
        if (s > connStrength || st > connStrength) { kept.emplace_back(e.row, e.col, 0.0); ptr[e.row + 1]++; }
    }
    for (index_t i = 0; i < M; ++i) ptr[i + 1] += ptr[i];
    col.resize(kept.size());
    std::vector<nnz_t> fill(ptr.begin(), ptr.end() - 1);
    for (const auto &e : kept) col[fill[e.row]++] = e.col;
}

// aggregation_1_dist (setup1:724-995): synchronous rounds; an undecided node looks at itself and
// its strong neighbours that are undecided or roots and takes the smallest index; if that is the
// node itself it becomes a root, if it is a root it joins it.  Then aggregate_index_update
// (setup1:2103-2260): roots are renumbered 0..n-1 in ascending order of their fine index.
index_t amg_hierarchy::aggregate(const saena_matrix &A, const std::vector<nnz_t> &ptr, const std::vector<index_t> &col,
                                 std::vector<index_t> &agg) {
    const index_t size = A.M;
    agg.resize((size_t)size);
    std::vector<index_t> aggregate2((size_t)size);
    std::vector<char> decided((size_t)size, 0), dec_nei((size_t)size, 0), is_root((size_t)size, 0), is_root_nei((size_t)size, 0);
    std::vector<index_t> aggArray;
    for (index_t i = 0; i < size; ++i) agg[i] = i;
    bool continueAgg = true;
    while (continueAgg) {
        for (index_t i = 0; i < size; ++i) {
            if (decided[i]) continue;
            aggregate2[i] = agg[i];
            dec_nei[i] = 1;
            is_root_nei[i] = 0;
            for (nnz_t it = ptr[i]; it < ptr[i + 1]; ++it) {
                const index_t c = col[it];
                if (agg[c] < aggregate2[i] && (!decided[c] || is_root[c])) {
                    aggregate2[i] = agg[c];
                    dec_nei[i] = decided[c];
                    is_root_nei[i] = is_root[c];
                }
            }
        }
        for (index_t i = 0; i < size; ++i) {
            if (!decided[i] && dec_nei[i]) {
                decided[i] = 1;
                if (agg[i] == aggregate2[i]) { is_root[i] = 1; aggArray.push_back(agg[i]); }
                else if (is_root_nei[i]) agg[i] = aggregate2[i];
            }
        }
        continueAgg = false;
        for (index_t i = 0; i < size; ++i)
            if (!decided[i]) { continueAgg = true; break; }
    }
    std::sort(aggArray.begin(), aggArray.end());
    for (index_t i = 0; i < size; ++i)
        agg[i] = (index_t)(std::lower_bound(aggArray.begin(), aggArray.end(), agg[i]) - aggArray.begin());
    return (index_t)aggArray.size();
}

// largest eigenvalue of D^-1 A: 20 Lanczos steps on D^-1/2 A D^-1/2 (lamlan_saena.h:38-59,
// lambda_lanczos.hpp:95 max_iteration = 20), result x 1.0001.  The reference starts from a random
// vector (its estimate "fluctuates in each execution"); here the start vector is a fixed LCG sequence.
double amg_hierarchy::find_eig(const saena_matrix &A) {
    if (A.comm->nranks != 1) throw std::runtime_error("find_eig: multi-rank setup is not implemented in this round");
    const index_t n = A.M;
    Csr C = csr_from_entries(A.entry, 0, n, n);
    std::vector<double> isd((size_t)n);
    for (index_t i = 0; i < n; ++i) isd[i] = std::sqrt(std::fabs(A.inv_diag[i]));
    auto matvec = [&](const std::vector<double> &x, std::vector<double> &y) {
        for (index_t i = 0; i < n; ++i) {
            double s = 0;
            for (nnz_t k = C.ptr[i]; k < C.ptr[i + 1]; ++k) s += C.val[k] * isd[C.col[k]] * x[C.col[k]];
            y[i] = s * isd[i];
        }
    };
    const int m = std::min<index_t>(20, n);
    std::vector<double> v((size_t)n), vprev((size_t)n, 0.0), w((size_t)n), alpha, beta;
    unsigned long long lcg = 88172645463325252ULL;
    double nrm = 0;
    for (index_t i = 0; i < n; ++i) {
        lcg = lcg * 6364136223846793005ULL + 1442695040888963407ULL;
        v[i] = ((lcg >> 11) * (1.0 / 9007199254740992.0)) * 2.0 - 1.0;
        nrm += v[i] * v[i];
    }
    nrm = std::sqrt(nrm);
    for (auto &x : v) x /= nrm;
    double b = 0;
    for (int k = 0; k < m; ++k) {
        matvec(v, w);
        double a = 0;
        for (index_t i = 0; i < n; ++i) a += w[i] * v[i];
        alpha.push_back(a);
        for (index_t i = 0; i < n; ++i) w[i] -= a * v[i] + b * vprev[i];
        b = 0;
        for (index_t i = 0; i < n; ++i) b += w[i] * w[i];
        b = std::sqrt(b);
        if (k + 1 < m) beta.push_back(b);
        if (b < 1e-300) break;
        vprev = v;
        for (index_t i = 0; i < n; ++i) v[i] = w[i] / b;
    }
    // largest eigenvalue of the tridiagonal matrix by bisection on the Sturm count
    const int kdim = (int)alpha.size();
    double lo = alpha[0], hi = alpha[0];
    for (int i = 0; i < kdim; ++i) {
        const double r = (i > 0 ? std::fabs(beta[i - 1]) : 0) + (i + 1 < kdim && i < (int)beta.size() ? std::fabs(beta[i]) : 0);
        lo = std::min(lo, alpha[i] - r); hi = std::max(hi, alpha[i] + r);
    }
    auto count_below = [&](double x) {      // number of eigenvalues < x
        int cnt = 0;
        double d = 1;
        for (int i = 0; i < kdim; ++i) {
            const double b2 = i > 0 ? beta[i - 1] * beta[i - 1] : 0;
            d = alpha[i] - x - (i > 0 ? b2 / d : 0);
            if (d == 0) d = 1e-300;
            if (d < 0) cnt++;
        }
        return cnt;
    };
    for (int it = 0; it < 200 && hi - lo > 1e-14 * std::max(std::fabs(lo), std::fabs(hi)); ++it) {
        const double mid = 0.5 * (lo + hi);
        if (count_below(mid) >= kdim) hi = mid; else lo = mid;
    }
    return 1.0001 * 0.5 * (lo + hi);
}

// filter (setup2:852-916): entries with |v| <= THRE are lumped into the diagonal
void amg_hierarchy::filter(std::vector<cooEntry> &v, index_t sz, index_t ofst) {
    if (++filter_it < opts.filter_start) return;
    if (filter_thre_cur > opts.filter_max) filter_thre_cur = opts.filter_max;
    const double THRE = filter_thre_cur;
    std::vector<value_t> add2diag((size_t)sz, 0.0);
    std::vector<cooEntry> w;
    w.reserve(v.size());
    for (const auto &a : v) {
        if (std::fabs(a.val) > THRE || a.row == a.col) w.push_back(a);
        else add2diag[a.row - ofst] += a.val;
    }
    std::vector<char> check_diag((size_t)sz, 0);
    for (auto &a : w)
        if (a.row == a.col) {
            a.val += add2diag[a.row - ofst];
            check_diag[a.row - ofst] = 1;
            if (std::fabs(a.val) < SAENA_ALMOST_ZERO) a.val = 1.0;
        }
    bool added = false;
    for (index_t i = 0; i < sz; ++i)
        if (!check_diag[i]) { w.emplace_back(i + ofst, i + ofst, 1.0); added = true; }
    if (added) std::sort(w.begin(), w.end(), col_major);
    w.swap(v);
    filter_thre_cur *= std::pow(10, opts.filter_rate);
}

// coarsen (saena_object.cpp:409-452) = SA (setup1:8-254) + transposeP + compute_coarsen (setup2:8-358)
int amg_hierarchy::coarsen(int l) {
    amg_level &g = levels[l];
    saena_matrix &A = *g.A;
    Comm &c = *A.comm;
    if (c.nranks != 1) throw std::runtime_error("amg setup: multi-rank setup is not implemented in this round");

    // ---- find_aggregation (setup1:255-432) ----
    std::vector<nnz_t> sptr;
    std::vector<index_t> scol, agg;
    strength_graph(A, opts.connStrength, sptr, scol);
    const index_t new_size = aggregate(A, sptr, scol, agg);
    int ret_val = 0;
    if (opts.dynamic_levels) {                                           // setup1:385-405
        if ((unsigned)new_size <= least_row_threshold) ret_val = 1;
        else if (static_cast<float>(new_size) / A.Mbig > row_reduction_up_thrshld) ret_val = 1;
    }

    // ---- SA: P = (I - omega D^-1 A) P_tentative (setup1:60-239) ----
    transfer_matrix &P = g.P;
    P.comm = &c;
    P.Mbig = A.Mbig; P.Nbig = new_size; P.M = A.M;
    P.split_row = A.split;
    P.split_col = {0, new_size};
    const double om = A.jacobi_omega;                                    // Pomega = A->jacobi_omega, double (saena_object.h:168)
    std::vector<cooEntry> tmp;
    tmp.reserve((size_t)A.L.nnz_l_local);
    nnz_t iter = 0;
    for (index_t i = 0; i < A.M; ++i)
        for (index_t j = 0; j < A.L.nnzPerRow_local[i]; ++j, ++iter) {
            value_t vtmp = -om * A.inv_diag[i] * A.L.val_local[iter];
            if (i == A.L.col_local[iter]) vtmp += 1;
            tmp.emplace_back(i, agg[A.L.col_local[iter]], vtmp);
        }
    std::stable_sort(tmp.begin(), tmp.end(), col_major);                 // setup1:196
    for (size_t i = 0; i < tmp.size(); ++i) {                            // :205-217 add duplicates, drop ~0
        cooEntry t = tmp[i];
        while (i + 1 < tmp.size() && tmp[i + 1].row == tmp[i].row && tmp[i + 1].col == tmp[i].col) t.val += tmp[++i].val;
        if (std::fabs(t.val) > SAENA_ALMOST_ZERO) P.entry.push_back(t);
    }
    P.nnz_l = (nnz_t)P.entry.size();
    P.nnz_g = P.nnz_l;
    P.build_layout();                                                    // findLocalRemote

    // ---- R = P^T (restrict_matrix::transposeP) ----
    transpose_transfer(P, g.R);

    // ---- Ac = (R A) P  (triple_mat_mult, setup2:361-849) ----
    Csr Rc = csr_from_entries(g.R.entry, 0, new_size, A.Mbig);
    Csr Ac_ = csr_from_entries(A.entry, 0, A.M, A.Mbig);
    std::vector<cooEntry> RA = spgemm_entries(Rc, Ac_, 0);
    Rc = Csr(); Ac_ = Csr();
    Csr RAc = csr_from_entries(RA, 0, new_size, A.Mbig);
    RA.clear(); RA.shrink_to_fit();
    Csr Pc = csr_from_entries(P.entry, 0, A.M, new_size);
    std::vector<cooEntry> AcE = spgemm_entries(RAc, Pc, 0);
    RAc = Csr(); Pc = Csr();
    std::sort(AcE.begin(), AcE.end(), col_major);

    filter(AcE, new_size, 0);                                            // setup2:117-121

    g.Ac_store.reset(new saena_matrix(&c));
    saena_matrix &Ac = *g.Ac_store;
    Ac.Mbig = new_size; Ac.M = new_size;
    Ac.split = {0, new_size};
    Ac.entry.swap(AcE);
    Ac.nnz_l = (nnz_t)Ac.entry.size();
    Ac.nnz_g = Ac.nnz_l;
    Ac.remove_boundary = false;
    Ac.matrix_setup();                                                   // setup2:341
    return ret_val;
}

// saena_object::setup (saena_object.cpp:175-406)
int amg_hierarchy::setup(saena_matrix *A, const amg_options &o) {
    opts = o;
    if (!A->assembled) throw std::runtime_error("amg setup: the matrix is not assembled");
    filter_thre_cur = opts.filter_thre;
    filter_it = 0;
    max_level = opts.max_level;
    levels.clear();
    levels.resize((size_t)max_level + 1);
    levels[0].A = A;
    if (opts.smoother == "chebyshev" && std::fabs(A->eig_max_of_invdiagXA) < SAENA_ALMOST_ZERO)   // :201-204
        A->eig_max_of_invdiagXA = find_eig(*A);
    for (int i = 0; i < max_level; ++i) {                               // :239
        const int res = coarsen(i);
        if (res == 1) max_level = i + 1;                                // :287-289 this will be the last level
        levels[i + 1].A = levels[i].Ac_store.get();
        if (opts.smoother == "chebyshev") levels[i + 1].A->eig_max_of_invdiagXA = find_eig(*levels[i + 1].A);   // :315
    }
    levels.resize((size_t)max_level + 1);
    return 0;
}

} // namespace saena_host

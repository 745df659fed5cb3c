// amg_setup.cpp -- smoothed-aggregation setup on the host (restatement; citations are
// file:line in paralab/Saena).  amg_hierarchy::setup builds a hierarchy at one rank;
// setup_rows_distributed builds the same hierarchy over several ranks, every rank its own rows.
#include "amg_setup.h"
#include "par.h"

#include <algorithm>
#include <cfloat>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <fstream>
#include <cstdlib>
#include <mutex>
#include <sstream>
#include <thread>

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

int n_threads() { return setup_threads(); }

// run f(t, lo, hi) over [0,n) split into contiguous chunks of about equal `weight`
template <class F>
void parallel_rows(index_t n, const std::vector<nnz_t> *weight_ptr, F f) {
    const int T = std::max(1, std::min<int>(n_threads(), n / 64 + 1));
    std::vector<index_t> cut((size_t)T + 1, n);
    cut[0] = 0;
    if (weight_ptr) {
        const nnz_t tot = (*weight_ptr)[(size_t)n];
        for (int t = 1; t < T; ++t)
            cut[t] = (index_t)(std::lower_bound(weight_ptr->begin(), weight_ptr->begin() + n + 1, tot * t / T) - weight_ptr->begin());
    } else {
        for (int t = 1; t < T; ++t) cut[t] = (index_t)((long)n * t / T);
    }
    for (int t = 1; t <= T; ++t) cut[t] = std::max(cut[t], cut[t - 1]);
    if (T == 1) { f(0, 0, n); return; }
    ThreadPool::get().run(T, [&](int t) { f(t, cut[(size_t)t], cut[(size_t)t + 1]); });
}

// the aggregation's rounds: most of them touch a few thousand rows for a few reads each -- waking the pool costs more than the work
template <class F>
void parallel_rows_if_large(index_t n, F f) {
    if (n < 32768) { if (n > 0) f(0, 0, n); return; }
    parallel_rows(n, nullptr, f);
}

// A row of (coarse column, value) pairs ordered by column, equal columns in their original order -- std::stable_sort's result
// without the buffer it allocates per call (one call per fine row: 16 M of them per level): insertion sort for the short
// rows of the stencil levels, else a sort of (column, position) keys.
void stable_sort_by_first(std::vector<std::pair<index_t, value_t>> &row, std::vector<std::pair<long, value_t>> &scratch) {
    const size_t n = row.size();
    // insertion sort: all of a short row; of a longer one as long as it stays cheap -- the coarse ids of a fine row's neighbours
    // arrive nearly sorted (aggregates are numbered in the order of their roots, the row's columns ascend), a few shifts per entry
    size_t i = 1, shifts = 0;
    const size_t budget = n <= 16 ? (size_t)-1 : 6 * n;
    for (; i < n && shifts <= budget; ++i) {
        const auto x = row[i];
        size_t j = i;
        while (j > 0 && row[j - 1].first > x.first) { row[j] = row[j - 1]; --j; }
        shifts += i - j;
        row[j] = x;
    }
    if (i == n) return;
    // (a stable sort of the whole row gives the same order whatever prefix is already sorted: equal keys keep their positions' order,
    //  and the insertion above kept it as well)
    scratch.resize(n);
    for (size_t i = 0; i < n; ++i) scratch[i] = {((long)row[i].first << 32) | (long)i, row[i].second};
    std::sort(scratch.begin(), scratch.end(), [](const auto &a, const auto &b) { return a.first < b.first; });
    for (size_t i = 0; i < n; ++i) row[i] = {(index_t)(scratch[i].first >> 32), scratch[i].second};
}

// the assembled one-rank operator as CSR (its layout arrays are row-major, columns ascending)
Csr csr_of(const DistLayout &L, index_t ncols) {
    Csr C;
    C.nrows = L.M; C.ncols = ncols;
    C.ptr.resize((size_t)L.M + 1);
    C.ptr[0] = 0;
    for (index_t i = 0; i < L.M; ++i) C.ptr[i + 1] = C.ptr[i] + L.nnzPerRow_local[i];
    C.col = L.col_local;
    C.val = L.val_local;
    return C;
}

Csr transpose(const Csr &A) {
    Csr T;
    T.nrows = A.ncols; T.ncols = A.nrows;
    T.ptr.assign((size_t)A.ncols + 1, 0);
    for (index_t c : A.col) T.ptr[c + 1]++;
    for (index_t i = 0; i < A.ncols; ++i) T.ptr[i + 1] += T.ptr[i];
    T.col.resize(A.col.size()); T.val.resize(A.val.size());
    std::vector<nnz_t> fill(T.ptr.begin(), T.ptr.end() - 1);
    for (index_t i = 0; i < A.nrows; ++i)
        for (nnz_t k = A.ptr[i]; k < A.ptr[i + 1]; ++k) { const nnz_t q = fill[A.col[k]]++; T.col[q] = i; T.val[q] = A.val[k]; }
    return T;
}

// C = A B (row-wise Gustavson, rows of C sorted by column, row chunks on threads).  Entries with
// |v| <= ALMOST_ZERO are dropped unless row id == column id, the rule of the reference's SpGEMM
// output (saena_object_setup_matmat.cpp:2423,2442).
// row_offset: global id of row 0 (distributed setup: the "row id == column id" exemption is about GLOBAL ids)
} // namespace
// C = A B on the GPU when libsaena_amd.so has a device context (sgpu_spgemm.hip installs this; bit-identical result)
spgemm_hook_fn g_spgemm_hook = nullptr;
double g_measured_chain_us = 0.0;
namespace {

// the left operand as three arrays: the distributed setup multiplies R and R A with their columns re-labelled to positions
// in the stacked right operand, and only that column array is new (row pointers and values are the operand's own)
struct CsrRef {
    index_t nrows = 0;
    const nnz_t *ptr = nullptr;
    const index_t *col = nullptr;
    const value_t *val = nullptr;
    CsrRef() {}
    CsrRef(const Csr &M) : nrows(M.nrows), ptr(M.ptr.data()), col(M.col.data()), val(M.val.data()) {}
    CsrRef(const Csr &M, const std::vector<index_t> &cols) : nrows(M.nrows), ptr(M.ptr.data()), col(cols.data()), val(M.val.data()) {}
};

Csr spgemm(const CsrRef &A, const Csr &B, index_t row_offset = 0) {
    Csr C;
    C.nrows = A.nrows; C.ncols = B.ncols;
    if (g_spgemm_hook && A.nrows > 0 && B.nrows > 0 && ((size_t)A.ptr[A.nrows] + B.col.size()) >= 200000) {
        if (g_spgemm_hook(A.nrows, B.nrows, B.ncols, A.ptr, A.col, A.val, B.ptr.data(), B.col.data(), B.val.data(), 0, nullptr, nullptr,
                          row_offset, C.ptr, C.col, C.val) == 0)
            return C;
        C.ptr.clear(); C.col.clear(); C.val.clear();       // the GPU declined (memory): the host kernel below
    }
    // work estimate per row for load balance
    std::vector<nnz_t> work((size_t)A.nrows + 1, 0);
    for (index_t i = 0; i < A.nrows; ++i) {
        nnz_t w = 0;
        for (nnz_t ka = A.ptr[i]; ka < A.ptr[i + 1]; ++ka) w += B.ptr[A.col[ka] + 1] - B.ptr[A.col[ka]];
        work[i + 1] = work[i] + w + 1;
    }
    const int T = n_threads();
    std::vector<std::vector<index_t>> tcol((size_t)T);
    std::vector<std::vector<value_t>> tval((size_t)T);
    std::vector<index_t> tlo((size_t)T, 0), thi((size_t)T, 0);
    std::vector<nnz_t> rowlen((size_t)A.nrows, 0);
    parallel_rows(A.nrows, &work, [&](int t, index_t lo, index_t hi) {
        tlo[t] = lo; thi[t] = hi;
        // Two accumulators, same arithmetic (every output entry adds its products in generation order): an
        // open-addressing table sized to the row, which stays in L1/L2 (a 415 K-column dense accumulator per thread
        // cost ~22 ns per product in cache misses), and the dense array for rows that could fill a large part of it.
        std::vector<value_t> acc;
        std::vector<char> mark;
        std::vector<index_t> cols;
        std::vector<index_t> hkey;
        std::vector<value_t> hval;
        std::vector<std::pair<index_t, value_t>> out;
        std::vector<size_t> slots;
        auto &oc = tcol[t];
        auto &ov = tval[t];
        { LazyBigalloc lazy; oc.reserve((size_t)((work[hi] - work[lo]) / 4 + 16)); ov.reserve(oc.capacity()); }     // an estimate: not touched
        for (index_t i = lo; i < hi; ++i) {
            const nnz_t products = work[i + 1] - work[i] - 1;
            nnz_t cnt = 0;
            if (products > (nnz_t)B.ncols / 8) {                         // dense accumulator
                if (acc.empty()) { acc.assign((size_t)B.ncols, 0.0); mark.assign((size_t)B.ncols, 0); }
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
                for (index_t j : cols) {
                    if (std::fabs(acc[j]) > SAENA_ALMOST_ZERO || i + row_offset == j) { oc.push_back(j); ov.push_back(acc[j]); ++cnt; }
                    mark[j] = 0;
                }
            } else {                                                      // hash accumulator, grown by rehashing at load 1/2
                size_t cap = 256;
                const size_t want = 2 * (size_t)std::min<nnz_t>(products, A.ptr[i + 1] - A.ptr[i] + 64);
                while (cap < want) cap *= 2;
                if (hkey.size() != cap) { hkey.assign(cap, -1); hval.assign(cap, 0.0); }
                size_t mask = cap - 1;
                out.clear();                                              // (column, unused) of every distinct output
                auto slot_of = [&](index_t j) {
                    size_t h = ((size_t)(unsigned)j * 2654435761u) & mask;
                    while (hkey[h] != -1 && hkey[h] != j) h = (h + 1) & mask;
                    return h;
                };
                for (nnz_t ka = A.ptr[i]; ka < A.ptr[i + 1]; ++ka) {
                    const index_t k = A.col[ka];
                    const value_t a = A.val[ka];
                    for (nnz_t kb = B.ptr[k]; kb < B.ptr[k + 1]; ++kb) {
                        const index_t j = B.col[kb];
                        size_t h = slot_of(j);
                        if (hkey[h] == -1) {
                            if (2 * (out.size() + 1) > cap) {             // grow: values move, sums are untouched
                                std::vector<index_t> okey(2 * cap, -1);
                                std::vector<value_t> oval(2 * cap, 0.0);
                                const size_t nmask = 2 * cap - 1;
                                for (const auto &e : out) {
                                    const size_t ho = slot_of(e.first);
                                    size_t hn = ((size_t)(unsigned)e.first * 2654435761u) & nmask;
                                    while (okey[hn] != -1) hn = (hn + 1) & nmask;
                                    okey[hn] = e.first; oval[hn] = hval[ho];
                                }
                                hkey.swap(okey); hval.swap(oval);
                                cap *= 2; mask = nmask;
                                h = slot_of(j);
                            }
                            hkey[h] = j; hval[h] = 0.0; out.emplace_back(j, 0.0);
                        }
                        hval[h] += a * B.val[kb];
                    }
                }
                slots.clear();
                for (auto &e : out) { const size_t h = slot_of(e.first); e.second = hval[h]; slots.push_back(h); }
                for (size_t h : slots) hkey[h] = -1;                      // cleared after ALL look-ups: probing needs the chains intact
                std::sort(out.begin(), out.end(), [](const auto &x, const auto &y) { return x.first < y.first; });
                for (const auto &e : out)
                    if (std::fabs(e.second) > SAENA_ALMOST_ZERO || i + row_offset == e.first) { oc.push_back(e.first); ov.push_back(e.second); ++cnt; }
            }
            rowlen[i] = cnt;
        }
    });
    C.ptr.resize((size_t)A.nrows + 1);
    C.ptr[0] = 0;
    for (index_t i = 0; i < A.nrows; ++i) C.ptr[i + 1] = C.ptr[i] + rowlen[i];
    C.col.resize((size_t)C.ptr[A.nrows]); C.val.resize((size_t)C.ptr[A.nrows]);
    parallel_chunks<int>(T, 1, [&](int, int t0, int t1) {
        for (int t = t0; t < t1; ++t) {
            if (tcol[t].empty()) continue;
            std::copy(tcol[t].begin(), tcol[t].end(), C.col.begin() + C.ptr[tlo[t]]);
            std::copy(tval[t].begin(), tval[t].end(), C.val.begin() + C.ptr[tlo[t]]);
        }
    });
    return C;
}

} // namespace

// ---------------------------------------------------------------------------
std::vector<cooEntry> amg_hierarchy::matmat(const saena_matrix &A, const saena_matrix &B) {
    if (A.comm->nranks != 1) throw std::runtime_error("matmat: the SpGEMM is single-rank in this round");
    if (!A.assembled || !B.assembled) throw std::runtime_error("matmat: assemble A and B first");
    if (A.Mbig != B.Mbig) throw std::runtime_error("matmat: A and B must have the same size");
    const Csr C = spgemm(csr_of(A.L, A.Mbig), csr_of(B.L, B.Mbig));
    std::vector<cooEntry> out;
    out.reserve(C.col.size());
    for (index_t i = 0; i < C.nrows; ++i)
        for (nnz_t k = C.ptr[i]; k < C.ptr[i + 1]; ++k) out.emplace_back(i, C.col[k], C.val[k]);
    return out;
}

// ---------------------------------------------------------------------------
// strength of connection (setup1:520-719) + threshold (strength_matrix.cpp:242-258)
void amg_hierarchy::strength_graph(const saena_matrix &A, float connStrength, std::vector<nnz_t> &ptr, std::vector<index_t> &col) {
    if (A.comm->nranks != 1) throw std::runtime_error("strength_graph: multi-rank setup is not implemented in this round");
    const index_t M = A.M;
    const DistLayout &L = A.L;
    std::vector<nnz_t> ap((size_t)M + 1, 0);
    for (index_t i = 0; i < M; ++i) ap[i + 1] = ap[i] + L.nnzPerRow_local[i];
    std::vector<value_t> maxPerRow((size_t)M, -DBL_MAX);                       // :527-533
    parallel_rows(M, &ap, [&](int, index_t lo, index_t hi) {
        for (index_t i = lo; i < hi; ++i)
            for (nnz_t k = ap[i]; k < ap[i + 1]; ++k)
                if (L.col_local[k] != i) maxPerRow[i] = std::max(maxPerRow[i], -L.val_local[k]);
    });
    // S(i,j) = -a_ij / max_k(-a_ik), S^T(i,j) = -a_ij / max_k(-a_jk), diagonal 1; keep if either > connStrength
    std::vector<char> keep(L.col_local.size());
    ptr.assign((size_t)M + 1, 0);
    parallel_rows(M, &ap, [&](int, index_t lo, index_t hi) {
        for (index_t i = lo; i < hi; ++i) {
            nnz_t cnt = 0;
            for (nnz_t k = ap[i]; k < ap[i + 1]; ++k) {
                const index_t j = L.col_local[k];
                value_t s_, st;
                if (i == j) { s_ = 1; st = 1; }
                else { s_ = -L.val_local[k] / maxPerRow[i]; st = -L.val_local[k] / maxPerRow[j]; }
                keep[k] = (s_ > connStrength || st > connStrength);
                cnt += keep[k];
            }
            ptr[i + 1] = cnt;
        }
    });
    for (index_t i = 0; i < M; ++i) ptr[i + 1] += ptr[i];
    col.resize((size_t)ptr[M]);
    parallel_rows(M, &ap, [&](int, index_t lo, index_t hi) {
        for (index_t i = lo; i < hi; ++i) {
            nnz_t q = ptr[i];
            for (nnz_t k = ap[i]; k < ap[i + 1]; ++k)
                if (keep[k]) col[q++] = L.col_local[k];
        }
    });
}

// aggregation_1_dist (setup1:724-995): synchronous rounds; an undecided node looks at itself and
// its strong neighbours that are undecided or roots and takes the smallest index; if that is the
// node itself it becomes a root, if it is a root it joins it.  Then aggregate_index_update
// (setup1:2103-2260): roots are renumbered 0..n-1 in ascending order of their fine index.
index_t amg_hierarchy::aggregate(const saena_matrix &A, const std::vector<nnz_t> &ptr, const std::vector<index_t> &col,
                                 std::vector<index_t> &agg, std::vector<index_t> *roots) {
    const index_t size = A.M;
    agg.resize((size_t)size);
    std::vector<index_t> aggregate2((size_t)size);
    std::vector<char> decided((size_t)size, 0), dec_nei((size_t)size, 0), is_root((size_t)size, 0), is_root_nei((size_t)size, 0);
    std::vector<index_t> aggArray;
    for (index_t i = 0; i < size; ++i) agg[i] = i;
    // The reference runs synchronous rounds over all undecided rows (setup1:760-960); 280 rounds for a 94^3 grid,
    // ~140 visits per row, because the priority wave crosses the grid one node per round.  The same rounds are run
    // here, but a row is evaluated again only when its evaluation can come out differently.  An evaluation looks for the
    // eligible neighbour (undecided, or a root) with the smallest id below the row's own; an eligible row carries its OWN id,
    // ids of eligible rows never change, a row that joins an aggregate leaves the eligible set for good, and a row's columns
    // ascend -- so that neighbour is the FIRST eligible column below the diagonal, the columns passed over on the way stay
    // ineligible (the next evaluation resumes at `cursor`), and the result cannot change until that one neighbour, the row's
    // `blocker`, changes state.  A waiting row therefore hangs itself on its blocker's chain (head / next), and a row that
    // decides wakes its chain: identical aggregates (pinned bit for bit against the compiled reference, tests/test_sa_pins.py),
    // 2-4 visits per row of a few reads each, no transposed pattern.  Both sweeps of a round read only the previous round's
    // state and write only row i's own entries (and push onto chains with an atomic exchange), so the work list is dealt to
    // threads.
    const int T = n_threads();
    const auto t_begin = std::chrono::steady_clock::now();
    std::vector<std::vector<index_t>> troots((size_t)T), tnext((size_t)T), tdone((size_t)T);
    std::vector<index_t> work((size_t)size), cursor((size_t)size, 0), head((size_t)size, -1), next((size_t)size, -1);
    for (index_t i = 0; i < size; ++i) work[i] = i;
    long rounds = 0, visited = 0, undecided = size;
    while (!work.empty()) {
        const index_t nw = (index_t)work.size();
        ++rounds; visited += nw;
        parallel_rows_if_large(nw, [&](int, index_t lo, index_t hi) {
            for (index_t q = lo; q < hi; ++q) {
                const index_t i = work[q];
                aggregate2[i] = agg[i];
                dec_nei[i] = 1;
                is_root_nei[i] = 0;
                nnz_t it = ptr[i] + cursor[i];
                for (; it < ptr[i + 1] && col[it] < i; ++it) {
                    const index_t c = col[it];
                    if (!decided[c] || is_root[c]) {
                        aggregate2[i] = agg[c];
                        dec_nei[i] = decided[c];
                        is_root_nei[i] = is_root[c];
                        if (!decided[c]) next[i] = __atomic_exchange_n(&head[c], i, __ATOMIC_ACQ_REL);      // wait for c
                        break;
                    }
                }
                cursor[i] = (index_t)(it - ptr[i]);
            }
        });
        for (auto &v : tdone) v.clear();
        parallel_rows_if_large(nw, [&](int t, index_t lo, index_t hi) {
            for (index_t q = lo; q < hi; ++q) {
                const index_t i = work[q];
                if (dec_nei[i]) {
                    decided[i] = 1;
                    if (agg[i] == aggregate2[i]) { is_root[i] = 1; troots[t].push_back(agg[i]); }
                    else if (is_root_nei[i]) agg[i] = aggregate2[i];
                    tdone[t].push_back(i);
                }
            }
        });
        // next round: the rows that waited for a row decided in this one
        for (auto &v : tnext) v.clear();
        std::vector<index_t> done;
        for (auto &v : tdone) done.insert(done.end(), v.begin(), v.end());
        undecided -= (long)done.size();
        parallel_rows_if_large((index_t)done.size(), [&](int t, index_t lo, index_t hi) {
            for (index_t q = lo; q < hi; ++q) {
                const index_t j = done[q];
                for (index_t c = head[j]; c >= 0; c = next[c]) tnext[t].push_back(c);
                head[j] = -1;
            }
        });
        work.clear();
        for (auto &v : tnext) work.insert(work.end(), v.begin(), v.end());
    }
    if (undecided != 0) throw std::runtime_error("aggregation did not terminate: " + std::to_string(undecided) + " undecided rows");
    if (std::getenv("SAENA_SETUP_TIMING"))
        fprintf(stderr, "[aggregate] %ld rounds, %ld row visits for %d rows, %.3f s\n", rounds, visited, size,
                std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count());
    for (auto &tr : troots) aggArray.insert(aggArray.end(), tr.begin(), tr.end());
    std::sort(aggArray.begin(), aggArray.end());
    for (index_t i = 0; i < size; ++i)
        agg[i] = (index_t)(std::lower_bound(aggArray.begin(), aggArray.end(), agg[i]) - aggArray.begin());
    if (roots) *roots = aggArray;
    return (index_t)aggArray.size();
}

// largest eigenvalue of D^-1 A: 20 Lanczos steps on D^-1/2 A D^-1/2 (lamlan_saena.h:38-59,
// lambda_lanczos.hpp:95 max_iteration = 20), result x 1.0001.  The reference starts from a random
// vector (its estimate "fluctuates in each execution"); here the start vector is a fixed LCG sequence.
double amg_hierarchy::find_eig(const saena_matrix &A) {
    if (A.comm->nranks != 1) throw std::runtime_error("find_eig: multi-rank setup is not implemented in this round");
    const index_t n = A.M;
    const Csr C = csr_of(A.L, n);
    std::vector<double> isd((size_t)n);
    for (index_t i = 0; i < n; ++i) isd[i] = std::sqrt(std::fabs(A.inv_diag[i]));
    auto matvec = [&](const std::vector<double> &x, std::vector<double> &y) {
        parallel_rows(n, &C.ptr, [&](int, index_t lo, index_t hi) {
            for (index_t i = lo; i < hi; ++i) {
                double s_ = 0;
                for (nnz_t k = C.ptr[i]; k < C.ptr[i + 1]; ++k) s_ += C.val[k] * isd[C.col[k]] * x[C.col[k]];
                y[i] = s_ * isd[i];
            }
        });
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

// filter (setup2:852-916): entries with |v| <= THRE are lumped into the diagonal (one-rank CSR form)
static void filter_csr(Csr &C, double THRE, index_t row_offset = 0) {
    // rows are independent: count what every row keeps (on threads), then write the rows to their places (on threads); a
    // level that loses nothing and misses no diagonal entry -- the common case -- is left as it is
    const index_t n = C.nrows;
    std::vector<nnz_t> nptr((size_t)n + 1, 0);
    std::vector<char> changed((size_t)n_threads(), 0);
    parallel_rows(n, &C.ptr, [&](int t, index_t r0, index_t r1) {
        for (index_t i = r0; i < r1; ++i) {
            const index_t gi = i + row_offset;
            nnz_t kept = 0;
            bool has_diag = false;
            for (nnz_t k = C.ptr[i]; k < C.ptr[i + 1]; ++k) {
                if (C.col[k] == gi) { has_diag = true; ++kept; }
                else if (std::fabs(C.val[k]) > THRE) ++kept;
            }
            if (!has_diag) ++kept;
            if (kept != C.ptr[i + 1] - C.ptr[i] || !has_diag) changed[(size_t)t] = 1;
            nptr[(size_t)i + 1] = kept;
        }
    });
    bool any = false;
    for (char c : changed) any = any || c;
    if (!any) {                                             // nothing dropped, every diagonal present: only the ~0 diagonal rule applies
        parallel_rows(n, &C.ptr, [&](int, index_t r0, index_t r1) {
            for (index_t i = r0; i < r1; ++i)
                for (nnz_t k = C.ptr[i]; k < C.ptr[i + 1]; ++k)
                    if (C.col[k] == i + row_offset && std::fabs(C.val[k]) < SAENA_ALMOST_ZERO) C.val[k] = 1.0;
        });
        return;
    }
    for (index_t i = 0; i < n; ++i) nptr[(size_t)i + 1] += nptr[(size_t)i];
    std::vector<index_t> ncol((size_t)nptr[(size_t)n]);
    std::vector<value_t> nval((size_t)nptr[(size_t)n]);
    parallel_rows(n, &C.ptr, [&](int, index_t r0, index_t r1) {
        for (index_t i = r0; i < r1; ++i) {
            value_t add2diag = 0.0;
            bool has_diag = false;
            const index_t gi = i + row_offset;
            nnz_t q = nptr[(size_t)i], diag_pos = 0;
            bool placed = false;                             // a missing diagonal goes in at its place in column order (:896-903)
            for (nnz_t k = C.ptr[i]; k < C.ptr[i + 1]; ++k) {
                const index_t j = C.col[k];
                if (std::fabs(C.val[k]) > THRE || j == gi) {
                    if (j == gi) { has_diag = true; diag_pos = q; }
                    ncol[(size_t)q] = j; nval[(size_t)q] = C.val[k]; ++q;
                } else {
                    add2diag += C.val[k];
                }
            }
            if (has_diag) {
                nval[(size_t)diag_pos] += add2diag;
                if (std::fabs(nval[(size_t)diag_pos]) < SAENA_ALMOST_ZERO) nval[(size_t)diag_pos] = 1.0;
            } else {
                nnz_t pos = nptr[(size_t)i];
                while (pos < q && ncol[(size_t)pos] < gi) ++pos;
                for (nnz_t m = q; m > pos; --m) { ncol[(size_t)m] = ncol[(size_t)m - 1]; nval[(size_t)m] = nval[(size_t)m - 1]; }
                ncol[(size_t)pos] = gi; nval[(size_t)pos] = 1.0;
                placed = true;
            }
            (void)placed;
        }
    });
    C.ptr.swap(nptr); C.col.swap(ncol); C.val.swap(nval);
}

void amg_hierarchy::filter(std::vector<cooEntry> &, index_t, index_t) {}   // COO form unused (CSR form above)

// coarsen (saena_object.cpp:409-452) = SA (setup1:8-254) + transposeP + compute_coarsen (setup2:8-358)
namespace {
struct PhaseTimer {       // SAENA_SETUP_TIMING=1: per-level phase times of the setup on stderr
    bool on = std::getenv("SAENA_SETUP_TIMING") != nullptr;
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    int level;
    explicit PhaseTimer(int l) : level(l) {}
    void lap(const char *what) {
        if (!on) return;
        const auto n = std::chrono::steady_clock::now();
        fprintf(stderr, "[setup L%d] %-22s %8.3f s\n", level, what, std::chrono::duration<double>(n - t).count());
        t = n;
    }
};
} // namespace

int amg_hierarchy::coarsen(int l) {
    amg_level &g = levels[l];
    saena_matrix &A = *g.A;
    Comm &c = *A.comm;
    if (c.nranks != 1) throw std::runtime_error("amg setup: multi-rank setup is not implemented in this round");
    PhaseTimer pt(l);

    // ---- find_aggregation (setup1:255-432) ----
    std::vector<nnz_t> sptr;
    std::vector<index_t> scol, agg;
    strength_graph(A, opts.connStrength, sptr, scol);
    pt.lap("strength graph");
    const index_t new_size = aggregate(A, sptr, scol, agg, &g.roots);
    pt.lap("aggregation");
    sptr = {}; scol = {};
    int ret_val = 0;
    if (opts.dynamic_levels) {                                           // setup1:385-405
        if ((unsigned)new_size <= least_row_threshold) ret_val = 1;
        else if (static_cast<float>(new_size) / A.Mbig > row_reduction_up_thrshld) ret_val = 1;
    }

    // ---- SA: P = (I - omega D^-1 A) P_tentative (setup1:60-239), built row by row ----
    const Csr Ac_ = csr_of(A.L, A.Mbig);
    const double om = A.jacobi_omega;                                    // Pomega = A->jacobi_omega, double (saena_object.h:168)
    Csr Pc;
    Pc.nrows = A.M; Pc.ncols = new_size;
    {
        const int T = n_threads();
        std::vector<std::vector<index_t>> tcol((size_t)T);
        std::vector<std::vector<value_t>> tval((size_t)T);
        std::vector<index_t> tlo((size_t)T, 0);
        std::vector<nnz_t> rowlen((size_t)A.M, 0);
        parallel_rows(A.M, &Ac_.ptr, [&](int t, index_t lo, index_t hi) {
            tlo[t] = lo;
            std::vector<std::pair<index_t, value_t>> row;
            std::vector<std::pair<long, value_t>> sort_scratch;
            for (index_t i = lo; i < hi; ++i) {
                row.clear();
                for (nnz_t k = Ac_.ptr[i]; k < Ac_.ptr[i + 1]; ++k) {
                    value_t vtmp = -om * A.inv_diag[i] * Ac_.val[k];
                    if (i == Ac_.col[k]) vtmp += 1;                      // I in (I - w Q A)
                    row.emplace_back(agg[Ac_.col[k]], vtmp);
                }
                stable_sort_by_first(row, sort_scratch);
                nnz_t cnt = 0;
                for (size_t q = 0; q < row.size(); ++q) {                // setup1:205-217 add duplicates, drop ~0
                    value_t v = row[q].second;
                    while (q + 1 < row.size() && row[q + 1].first == row[q].first) v += row[++q].second;
                    if (std::fabs(v) > SAENA_ALMOST_ZERO) { tcol[t].push_back(row[q].first); tval[t].push_back(v); ++cnt; }
                }
                rowlen[i] = cnt;
            }
        });
        Pc.ptr.resize((size_t)A.M + 1);
        Pc.ptr[0] = 0;
        for (index_t i = 0; i < A.M; ++i) Pc.ptr[i + 1] = Pc.ptr[i] + rowlen[i];
        Pc.col.resize((size_t)Pc.ptr[A.M]); Pc.val.resize((size_t)Pc.ptr[A.M]);
        for (int t = 0; t < T; ++t) {
            std::copy(tcol[t].begin(), tcol[t].end(), Pc.col.begin() + Pc.ptr[tlo[t]]);
            std::copy(tval[t].begin(), tval[t].end(), Pc.val.begin() + Pc.ptr[tlo[t]]);
        }
    }
    pt.lap("smoothed P");
    g.agg = std::move(agg);
    // ---- R = P^T (restrict_matrix::transposeP) ----
    Csr Rc = transpose(Pc);
    pt.lap("R = P^T");

    // ---- Ac = (R A) P  (triple_mat_mult, setup2:361-849) ----
    Csr RA = spgemm(Rc, Ac_);
    pt.lap("R*A");
    Csr AcN = spgemm(RA, Pc);
    RA = Csr();
    pt.lap("(RA)*P");

    // ---- filter (setup2:117-121, :852-916) ----
    if (++filter_it >= opts.filter_start) {
        if (filter_thre_cur > opts.filter_max) filter_thre_cur = opts.filter_max;
        filter_csr(AcN, filter_thre_cur);
        filter_thre_cur *= std::pow(10, opts.filter_rate);
    }

    pt.lap("filter");
    // ---- hand the three operators over in the reference's layout ----
    transfer_matrix &P = g.P;
    P.comm = &c; P.Mbig = A.Mbig; P.Nbig = new_size; P.M = A.M;
    P.split_row = A.split; P.split_col = {0, new_size};
    P.nnz_l = P.nnz_g = Pc.ptr[A.M];
    P.L.build_single_rank(A.M, new_size, Pc.ptr, std::move(Pc.col), std::move(Pc.val));
    transfer_matrix &R = g.R;
    R.comm = &c; R.Mbig = new_size; R.Nbig = A.Mbig; R.M = new_size;
    R.split_row = {0, new_size}; R.split_col = A.split;
    R.nnz_l = R.nnz_g = Rc.ptr[new_size];
    R.L.build_single_rank(new_size, A.M, Rc.ptr, std::move(Rc.col), std::move(Rc.val));

    g.Ac_store.reset(new saena_matrix(&c));
    g.Ac_store->setup_from_csr(new_size, AcN.ptr, std::move(AcN.col), std::move(AcN.val));   // setup2:341 matrix_setup
    pt.lap("layouts of P, R, Ac");
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

// ---------------------------------------------------------------------------
// multi-rank, gathered form (SAENA_SETUP=gathered): one-rank setup on every rank, then row-partition every level

namespace {
// rows [lo,hi) of a one-rank layout as global-id entries, column-major sorted
std::vector<cooEntry> slice_entries(const DistLayout &L, index_t lo, index_t hi) {
    std::vector<nnz_t> ptr((size_t)L.M + 1, 0);
    for (index_t i = 0; i < L.M; ++i) ptr[i + 1] = ptr[i] + L.nnzPerRow_local[i];
    std::vector<cooEntry> e;
    e.reserve((size_t)(ptr[hi] - ptr[lo]));
    for (index_t i = lo; i < hi; ++i)
        for (nnz_t k = ptr[i]; k < ptr[i + 1]; ++k) e.emplace_back(i, L.col_local[k], L.val_local[k]);
    std::sort(e.begin(), e.end(), col_major);
    return e;
}
} // namespace

void amg_hierarchy::distribute(Comm &c, const std::vector<index_t> &split0) {
    const int np = c.nranks, n = max_level + 1;
    dist.clear();
    dist.resize((size_t)n);
    dist[0].split = split0;
    level_stride.assign((size_t)n, 1);
    for (int l = 1; l < n; ++l) {
        const index_t rows = levels[l].A->Mbig;
        std::vector<index_t> sp((size_t)np + 1, rows);
        sp[0] = 0;
        const auto &roots = levels[l - 1].roots;                // splitNew[r] = #roots below the fine boundary
        for (int r = 1; r < np; ++r)
            sp[r] = (index_t)(std::lower_bound(roots.begin(), roots.end(), dist[l - 1].split[r]) - roots.begin());
        const int stride_prev = level_stride[(size_t)l - 1];
        level_stride[(size_t)l] = next_stride((long)levels[l].A->nnz_g, rows, np, stride_prev);
        const int active = (np + stride_prev - 1) / stride_prev;
        if (level_stride[(size_t)l] < np && active > 1 && rows >= active && !std::getenv("SAENA_NO_COARSE_REPART")) {
            // Ac->repart(): nnz-balanced over the active ranks (the whole operator is here: every rank counts its own rows'
            // share so that the histogram's sum over the ranks is the global one)
            const auto &npr = levels[l].A->L.nnzPerRow_local;
            SelfComm self;
            const std::vector<index_t> sa = nnz_balanced_split(self, rows, levels[l].A->nnz_g, active, [&](const std::vector<index_t> &firstSplit, std::vector<long> &H) {
                const int nb = (int)H.size();
                for (index_t i = 0; i < rows; ++i) H[(size_t)lower_bound2(firstSplit.data(), firstSplit.data() + nb, i)] += npr[(size_t)i];
            }, 4096);
            for (int r = 0; r <= np; ++r) sp[(size_t)r] = sa[(size_t)std::min(active, (r + stride_prev - 1) / stride_prev)];
        }
        dist[l].split = merge_split(sp, level_stride[(size_t)l]);
    }
    for (int l = 0; l < n; ++l) {
        dist_level &d = dist[l];
        const amg_level &g = levels[l];
        const index_t lo = d.split[c.rank], hi = d.split[c.rank + 1];
        d.Mbig = g.A->Mbig; d.nnzA = g.A->nnz_g; d.eig_max = g.A->eig_max_of_invdiagXA;
        d.A.build(c, slice_entries(g.A->L, lo, hi), d.split, d.split);
        d.inv_diag.assign(g.A->inv_diag.begin() + lo, g.A->inv_diag.begin() + hi);
        if (l < n - 1) {
            const std::vector<index_t> &spc = dist[l + 1].split;
            d.nnzP = g.P.nnz_g;
            d.P.build(c, slice_entries(g.P.L, lo, hi), d.split, spc);                        // fine rows, coarse columns
            d.R.build(c, slice_entries(g.R.L, spc[c.rank], spc[c.rank + 1]), spc, d.split);  // coarse rows, fine columns
        }
    }
}

// ---- agglomeration policy (see amg_setup.h) ----
int amg_hierarchy::next_stride(long nnzC, index_t rowsC, int np, int stride_prev) const {
    double chain = g_measured_chain_us > 0.0 ? g_measured_chain_us : shrink_chain_us;
    index_t rows_rule = shrink_rows;
    if (const char *e = std::getenv("SAENA_SHRINK_CHAIN_US")) chain = std::atof(e);
    if (const char *e = std::getenv("SAENA_SHRINK_ROWS")) rows_rule = (index_t)std::atol(e);
    const double T1 = 12.0 * (double)nnzC / shrink_bw * 1e6 + shrink_launch_us;     // us per apply on one GPU
    const int active = (np + std::max(1, stride_prev) - 1) / std::max(1, stride_prev);
    const double compute = T1 / active;
    int stride = stride_prev;
    const char *rule = "stays";
    if (np == 1) { stride = 1; rule = "one rank"; }
    else if (stride_prev >= np) { stride = np; rule = "already on one rank"; }
    else if (rowsC <= rows_rule) { stride = np; rule = "row rule: one rank"; }
    else if (chain <= 0) rule = "no chain: stays";
    else if (T1 <= chain) { stride = np; rule = "T1 <= chain: one rank"; }           // decide_shrinking_c: one rank
    else if (active > 1 && chain > 2.0 * compute) {                                  // decide_shrinking: comm > 2 x compute
        int f = (int)std::floor(chain / compute / 5.0);
        f = std::max(2, std::min(4, f));
        stride = std::min(np, stride_prev * f);
        rule = "chain > 2 x compute: groups merge";
    }
    // every decision with the chain it was taken with (the measured one is a run-time number: round-3 advisor finding)
    if (np > 1 && std::getenv("SAENA_SETUP_TIMING"))
        fprintf(stderr, "[saena] agglomeration: level of %ld rows / %ld entries, %d active ranks: T1 %.1f us, chain %.1f us (%s) -> stride %d (%s)\n", (long)rowsC, nnzC,
                active, T1, chain, std::getenv("SAENA_SHRINK_CHAIN_US") ? "SAENA_SHRINK_CHAIN_US" : g_measured_chain_us > 0.0 ? "measured at init, quantised" : "constant", stride, rule);
    return stride;
}

// shrink_set_params (src/saena_matrix_shrink.cpp:120-129): ranks that are not a multiple of `stride` hand their block
// to the preceding multiple
std::vector<index_t> amg_hierarchy::merge_split(const std::vector<index_t> &splitNew, int stride) {
    std::vector<index_t> sp = splitNew;
    const int np = (int)sp.size() - 1;
    if (stride <= 1) return sp;
    int root = np;
    for (int proc = np - 1; proc > 0; --proc) {
        if (proc % stride == 0) root = proc;
        else sp[(size_t)proc] = sp[(size_t)root];
    }
    return sp;
}

// ===========================================================================
// Distributed setup: every rank builds only ITS rows of every level.  The algorithms are the one-rank ones
// above, row block by row block; what a row block needs from other ranks (state of the aggregation, rows of
// A and P on the far side of the partition boundary) is fetched through FetchPlan.  Orders of accumulation are
// those of the one-rank code, so the hierarchy is the one-rank hierarchy bit for bit (tests/test_amg_setup.py).
namespace {

int owner_of_id(const std::vector<index_t> &split, index_t id) {
    const int np = (int)split.size() - 1;
    int p = (int)(std::upper_bound(split.begin(), split.end(), id) - split.begin()) - 1;
    if (p < 0) p = 0;
    if (p > np - 1) p = np - 1;
    while (p < np - 1 && split[p + 1] <= id) ++p;      // empty blocks share a boundary value
    while (p > 0 && split[p] > id) --p;
    return p;
}

// a fixed list of global ids owned by other ranks, and the machinery to get per-id data from their owners
struct FetchPlan {
    std::vector<index_t> wanted;            // sorted, unique, none owned by this rank
    std::vector<int> scount, rcount;        // ids I ask of each rank / ids each rank asks of me
    std::vector<index_t> serve;             // local indices I serve, grouped by asking rank
    index_t lo = 0;

    void build(Comm &c, const std::vector<index_t> &split, std::vector<index_t> ids) {
        std::sort(ids.begin(), ids.end());
        ids.erase(std::unique(ids.begin(), ids.end()), ids.end());
        wanted.swap(ids);
        index_positions();
        lo = split[c.rank];
        scount.assign((size_t)c.nranks, 0);
        for (index_t g : wanted) scount[owner_of_id(split, g)]++;
        serve = c.alltoallv_records(wanted, scount, &rcount);
        for (auto &x : serve) x -= lo;
    }
    // position of a wanted id: a table over the id range when the plan is large (the coarse levels of a hierarchy fetch a
    // good part of the level: a binary search per entry of the operator was a second per product there), else a search
    std::vector<index_t> where;             // where[g - wanted.front()], built by index_positions()
    void index_positions() {
        where.clear();
        if (wanted.size() < 4096) return;
        const size_t span = (size_t)(wanted.back() - wanted.front()) + 1;
        if (span > 16 * wanted.size() + ((size_t)1 << 20)) return;
        where.assign(span, -1);
        for (size_t i = 0; i < wanted.size(); ++i) where[(size_t)(wanted[i] - wanted.front())] = (index_t)i;
    }
    index_t pos(index_t g) const {
        if (!where.empty()) return where[(size_t)(g - wanted.front())];
        return (index_t)(std::lower_bound(wanted.begin(), wanted.end(), g) - wanted.begin());
    }
    // one T per wanted id, from the owner's local array
    template <class T>
    std::vector<T> values(Comm &c, const T *local) const {
        std::vector<T> out((size_t)serve.size());
        for (size_t i = 0; i < serve.size(); ++i) out[i] = local[(size_t)serve[i]];
        return c.alltoallv_known(out, rcount, scount);        // I serve rcount[p] ids to rank p and receive my scount[p] back
    }
    template <class T>
    std::vector<T> values(Comm &c, const std::vector<T> &local) const { return values(c, local.data()); }
    // the rows `wanted` of a row-distributed CSR (global column ids), in the order of `wanted`
    Csr rows(Comm &c, const Csr &local) const {
        std::vector<nnz_t> len((size_t)local.nrows);
        for (index_t i = 0; i < local.nrows; ++i) len[i] = local.ptr[i + 1] - local.ptr[i];
        const std::vector<nnz_t> rlen = values(c, len);
        std::vector<int> sc((size_t)c.nranks, 0);
        std::vector<index_t> scol;
        std::vector<value_t> sval;
        size_t q = 0;
        for (int p = 0; p < c.nranks; ++p)
            for (int k = 0; k < rcount[p]; ++k, ++q) {
                const index_t i = serve[q];
                sc[p] += (int)len[i];
                scol.insert(scol.end(), local.col.begin() + local.ptr[i], local.col.begin() + local.ptr[i + 1]);
                sval.insert(sval.end(), local.val.begin() + local.ptr[i], local.val.begin() + local.ptr[i + 1]);
            }
        Csr R;
        R.nrows = (index_t)wanted.size(); R.ncols = local.ncols;
        R.col = c.alltoallv_records(scol, sc);
        R.val = c.alltoallv_records(sval, sc);
        R.ptr.assign((size_t)R.nrows + 1, 0);
        for (index_t i = 0; i < R.nrows; ++i) R.ptr[i + 1] = R.ptr[i] + rlen[i];
        if ((size_t)R.ptr[R.nrows] != R.col.size()) throw std::runtime_error("FetchPlan::rows: length mismatch");
        return R;
    }
};

// ids referenced by the columns of `M` that fall outside [lo, hi), ascending: marked in a bitmap over the column range on
// threads (a coarse level references a good part of its columns from outside: collecting and sorting every such entry was
// a second per level), then read off in order
std::vector<index_t> outside_cols(const Csr &M, index_t lo, index_t hi) {
    const size_t ncols = (size_t)std::max<index_t>(M.ncols, 0);
    std::vector<index_t> ids;
    if (M.col.empty() || ncols == 0) return ids;
    std::vector<unsigned char> mark(ncols, 0);
    parallel_chunks<size_t>(M.col.size(), (size_t)1 << 20, [&](int, size_t a, size_t b) {
        for (size_t k = a; k < b; ++k) { const index_t cidx = M.col[k]; if (cidx < lo || cidx >= hi) __atomic_store_n(&mark[(size_t)cidx], (unsigned char)1, __ATOMIC_RELAXED); }
    });
    for (size_t g = 0; g < ncols; ++g) if (mark[g]) ids.push_back((index_t)g);
    return ids;
}

// local rows + fetched rows stacked (the right operand of a distributed product); the left operand's columns are
// re-labelled to row positions in that stack (order untouched).  Copies run on threads: a level is gigabytes.
Csr stack_rows(const Csr &local, const Csr &halo) {
    Csr B;
    B.nrows = local.nrows + halo.nrows; B.ncols = local.ncols;
    B.ptr.resize((size_t)B.nrows + 1);
    parallel_copy(B.ptr.data(), local.ptr.data(), (size_t)local.nrows + 1);
    const nnz_t base = local.ptr[local.nrows];
    for (index_t i = 0; i < halo.nrows; ++i) B.ptr[(size_t)local.nrows + 1 + i] = base + halo.ptr[i + 1];
    B.col.resize((size_t)base + halo.col.size()); B.val.resize((size_t)base + halo.val.size());
    parallel_copy(B.col.data(), local.col.data(), (size_t)base);
    parallel_copy(B.val.data(), local.val.data(), (size_t)base);
    std::copy(halo.col.begin(), halo.col.end(), B.col.begin() + base);
    std::copy(halo.val.begin(), halo.val.end(), B.val.begin() + base);
    return B;
}
// C = A [local; halo] without stacking the two pieces of B on the host when the GPU kernel takes them (it copies each piece to
// its place in device memory; only the row pointers are stacked here): at 16 M rows per rank a stacked copy of level 1's
// operator is 6.8 GB to allocate, fill and free for every product
Csr spgemm_stacked(const CsrRef &A, const Csr &local, const Csr &halo, index_t row_offset) {
    const nnz_t base = local.ptr[(size_t)local.nrows];
    if (g_spgemm_hook && A.nrows > 0 && local.nrows + halo.nrows > 0 && ((size_t)A.ptr[A.nrows] + (size_t)base + halo.col.size()) >= 200000) {
        Csr C;
        C.nrows = A.nrows; C.ncols = local.ncols;
        std::vector<nnz_t> ptr((size_t)local.nrows + (size_t)halo.nrows + 1);
        parallel_copy(ptr.data(), local.ptr.data(), (size_t)local.nrows + 1);
        for (index_t i = 0; i < halo.nrows; ++i) ptr[(size_t)local.nrows + 1 + i] = base + halo.ptr[(size_t)i + 1];
        if (g_spgemm_hook(A.nrows, local.nrows + halo.nrows, local.ncols, A.ptr, A.col, A.val, ptr.data(), local.col.data(), local.val.data(), base,
                          halo.col.empty() ? local.col.data() : halo.col.data(), halo.val.empty() ? local.val.data() : halo.val.data(),
                          row_offset, C.ptr, C.col, C.val) == 0)
            return C;
    }
    return spgemm(A, stack_rows(local, halo), row_offset);
}
// multi-gigabyte temporaries are handed to a detached thread to be freed (unmapping them took 2.8 s per rank of the 323^3
// setup, between the phases that need the cores): SAENA_NO_ASYNC_FREE=1 frees in place
template <class T>
void drop_async(T &&x) {
    static const bool off = std::getenv("SAENA_NO_ASYNC_FREE") != nullptr;
    if (off) { T gone(std::move(x)); return; }
    T *p = new T(std::move(x));
    try { std::thread([p] { delete p; }).detach(); } catch (...) { delete p; }
}
std::vector<index_t> relabel_cols(const Csr &M, index_t lo, index_t hi, const FetchPlan &plan) {
    std::vector<index_t> out(M.col.size());
    const index_t nloc = hi - lo;
    parallel_chunks<size_t>(M.col.size(), (size_t)1 << 20, [&](int, size_t a, size_t b) {
        for (size_t k = a; k < b; ++k) { const index_t cidx = M.col[k]; out[k] = (cidx >= lo && cidx < hi) ? cidx - lo : nloc + plan.pos(cidx); }
    });
    return out;
}

// the rows of this rank's block [row_lo, row_lo + X.nrows) go to their owners under `split_to`; returns the rows this
// rank receives, concatenated in source-rank order (blocks are ascending row ranges held by ascending ranks and owners
// are contiguous ranges, so the result is this rank's new block in row order)
Csr route_rows(Comm &c, const Csr &X, index_t row_lo, const std::vector<index_t> &split_to) {
    const int np = c.nranks;
    std::vector<int> scr((size_t)np, 0), sce((size_t)np, 0);
    std::vector<nnz_t> len((size_t)X.nrows);
    parallel_chunks<index_t>(X.nrows, 1 << 16, [&](int, index_t a, index_t b) { for (index_t i = a; i < b; ++i) len[(size_t)i] = X.ptr[(size_t)i + 1] - X.ptr[(size_t)i]; });
    for (int d = 0; d < np; ++d) {                                   // owners are contiguous row ranges: counts from the range ends, not row by row
        const index_t a = d == 0 ? 0 : std::min(std::max(split_to[(size_t)d] - row_lo, (index_t)0), X.nrows);          // (ids outside the partition go to its
        const index_t b = d == np - 1 ? X.nrows : std::min(std::max(split_to[(size_t)d + 1] - row_lo, (index_t)0), X.nrows);   //  first / last owner, as owner_of_id has it)
        scr[(size_t)d] = (int)(b - a);
        sce[(size_t)d] = (int)(X.ptr[(size_t)b] - X.ptr[(size_t)a]);
    }
    const std::vector<nnz_t> rlen = c.alltoallv_records(len, scr);
    Csr Y;
    Y.ncols = X.ncols;
    Y.col = c.alltoallv_records(X.col, sce);
    Y.val = c.alltoallv_records(X.val, sce);
    Y.nrows = (index_t)rlen.size();
    Y.ptr.assign((size_t)Y.nrows + 1, 0);
    for (index_t i = 0; i < Y.nrows; ++i) Y.ptr[i + 1] = Y.ptr[i] + rlen[i];
    if ((size_t)Y.ptr[Y.nrows] != Y.col.size()) throw std::runtime_error("route_rows: length mismatch");
    return Y;
}

struct AggState { index_t agg; char decided, is_root; char pad[2]; };
struct AggDelta { index_t slot; AggState s; };

} // namespace

namespace {
// find_eig above, row-distributed: same start vector (the LCG runs over the GLOBAL index), same recurrences; the dot
// products are sums of per-rank partial sums, so the estimate agrees with the one-rank one to rounding (~1e-15)
double dist_find_eig(Comm &c, const Csr &A, const std::vector<index_t> &split, const std::vector<value_t> &inv_diag, const FetchPlan &planA) {
    const index_t lo = split[c.rank], hi = split[c.rank + 1], n = hi - lo, Mbig = split.back();
    std::vector<double> isd((size_t)n);
    for (index_t i = 0; i < n; ++i) isd[i] = std::sqrt(std::fabs(inv_diag[i]));
    std::vector<double> isdE = isd;
    { const std::vector<double> h = planA.values(c, isd); isdE.insert(isdE.end(), h.begin(), h.end()); }
    const std::vector<index_t> arcol = relabel_cols(A, lo, hi, planA);
    struct { const std::vector<nnz_t> &ptr; const std::vector<index_t> &col; const std::vector<value_t> &val; } Ar{A.ptr, arcol, A.val};
    auto gsum = [&](double x) { c.allreduce_sum_f64(&x, 1); return x; };
    auto matvec = [&](const std::vector<double> &x, std::vector<double> &y) {
        std::vector<double> xE = x;
        { const std::vector<double> h = planA.values(c, x); xE.insert(xE.end(), h.begin(), h.end()); }
        parallel_rows(n, &Ar.ptr, [&](int, index_t r0, index_t r1) {        // (rows are independent: the same sums on any number of threads)
            for (index_t i = r0; i < r1; ++i) {
                double s_ = 0;
                for (nnz_t k = Ar.ptr[i]; k < Ar.ptr[i + 1]; ++k) s_ += Ar.val[k] * isdE[(size_t)Ar.col[k]] * xE[(size_t)Ar.col[k]];
                y[i] = s_ * isd[i];
            }
        });
    };
    const int m = (int)std::min<index_t>(20, Mbig);
    std::vector<double> v((size_t)n), vprev((size_t)n, 0.0), w((size_t)n), alpha, beta;
    unsigned long long lcg = 88172645463325252ULL;
    double nrm = 0;
    for (index_t gi = 0; gi < hi; ++gi) {
        lcg = lcg * 6364136223846793005ULL + 1442695040888963407ULL;
        if (gi >= lo) { v[(size_t)(gi - lo)] = ((lcg >> 11) * (1.0 / 9007199254740992.0)) * 2.0 - 1.0; nrm += v[(size_t)(gi - lo)] * v[(size_t)(gi - lo)]; }
    }
    nrm = std::sqrt(gsum(nrm));
    for (auto &x : v) x /= nrm;
    double b = 0;
    for (int k = 0; k < m; ++k) {
        matvec(v, w);
        double a = 0;
        for (index_t i = 0; i < n; ++i) a += w[i] * v[i];
        a = gsum(a);
        alpha.push_back(a);
        for (index_t i = 0; i < n; ++i) w[i] -= a * v[i] + b * vprev[i];
        b = 0;
        for (index_t i = 0; i < n; ++i) b += w[i] * w[i];
        b = std::sqrt(gsum(b));
        if (k + 1 < m) beta.push_back(b);
        if (b < 1e-300) break;
        vprev = v;
        for (index_t i = 0; i < n; ++i) v[i] = w[i] / b;
    }
    const int kdim = (int)alpha.size();
    double l0 = alpha[0], h0 = alpha[0];
    for (int i = 0; i < kdim; ++i) {
        const double r = (i > 0 ? std::fabs(beta[i - 1]) : 0) + (i + 1 < kdim && i < (int)beta.size() ? std::fabs(beta[i]) : 0);
        l0 = std::min(l0, alpha[i] - r); h0 = std::max(h0, alpha[i] + r);
    }
    auto count_below = [&](double x) {
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
    for (int it = 0; it < 200 && h0 - l0 > 1e-14 * std::max(std::fabs(l0), std::fabs(h0)); ++it) {
        const double mid = 0.5 * (l0 + h0);
        if (count_below(mid) >= kdim) h0 = mid; else l0 = mid;
    }
    return 1.0001 * 0.5 * (l0 + h0);
}
} // namespace

int amg_hierarchy::setup_rows_distributed(saena_matrix *Ad, const amg_options &o) {
    Comm &c = *Ad->comm;
    const int np = c.nranks, me = c.rank;
    opts = o;
    filter_thre_cur = opts.filter_thre;
    filter_it = 0;
    max_level = opts.max_level;
    levels.clear();
    levels.resize(1);
    levels[0].A = Ad;
    dist.clear();
    level_stride.assign(1, 1);

    // level 0: this rank's rows as CSR with global columns (entry is column-major)
    std::vector<index_t> split = Ad->split;
    Csr A;
    {
        std::vector<cooEntry> e = Ad->entry;
        std::sort(e.begin(), e.end(), row_major);
        const index_t lo = split[me], hi = split[me + 1];
        A.nrows = hi - lo; A.ncols = Ad->Mbig;
        A.ptr.assign((size_t)A.nrows + 1, 0);
        for (const auto &x : e) A.ptr[(size_t)(x.row - lo) + 1]++;
        for (index_t i = 0; i < A.nrows; ++i) A.ptr[i + 1] += A.ptr[i];
        A.col.reserve(e.size()); A.val.reserve(e.size());
        for (const auto &x : e) { A.col.push_back(x.col); A.val.push_back(x.val); }
    }
    std::vector<value_t> inv_diag = Ad->inv_diag;
    index_t Mbig = Ad->Mbig;
    const double om = Ad->jacobi_omega;

    for (int l = 0;; ++l) {
        PhaseTimer pt(l);
        const index_t lo = split[me], hi = split[me + 1], nloc = hi - lo;
        // ---- halo of A: the columns this rank's rows touch outside its block ----
        FetchPlan planA;
        planA.build(c, split, outside_cols(A, lo, hi));
        dist.emplace_back();
        {
            dist_level &d = dist.back();
            d.split = split; d.Mbig = Mbig; d.inv_diag = inv_diag; d.eig_max = l == 0 ? Ad->eig_max_of_invdiagXA : 0.0;
            if (opts.smoother == "chebyshev" && std::fabs(d.eig_max) < SAENA_ALMOST_ZERO) {     // saena_object.cpp:201-204,:315
                d.eig_max = dist_find_eig(c, A, split, inv_diag, planA);
                if (l == 0) Ad->eig_max_of_invdiagXA = d.eig_max;
            }
            long nn = (long)A.col.size();
            c.allreduce_sum_i64(&nn, 1);
            d.nnzA = nn;
            d.A.build_from_csr(c, A.ptr, A.col, A.val, split, split);
        }
        pt.lap("layout of A");
        if (l == max_level) break;

        const index_t next = nloc + (index_t)planA.wanted.size();
        std::vector<index_t> aext(A.col.size());                    // ext position (local row, or nloc + halo position) of every column of A
        parallel_rows(nloc, &A.ptr, [&](int, index_t a0, index_t a1) {
            for (nnz_t k = A.ptr[a0]; k < A.ptr[a1]; ++k) {
                const index_t g = A.col[k];
                aext[(size_t)k] = (g >= lo && g < hi) ? g - lo : nloc + planA.pos(g);
            }
        });

        // ---- strength graph (strength_graph above), columns as ext positions ----
        std::vector<value_t> maxPerRow((size_t)nloc, -DBL_MAX);
        parallel_rows(nloc, &A.ptr, [&](int, index_t r0, index_t r1) {
            for (index_t i = r0; i < r1; ++i)
                for (nnz_t k = A.ptr[i]; k < A.ptr[i + 1]; ++k)
                    if (A.col[k] != i + lo) maxPerRow[i] = std::max(maxPerRow[i], -A.val[k]);
        });
        std::vector<value_t> maxExt = maxPerRow;
        { const std::vector<value_t> h = planA.values(c, maxPerRow); maxExt.insert(maxExt.end(), h.begin(), h.end()); }
        std::vector<nnz_t> sptr((size_t)nloc + 1, 0);
        std::vector<index_t> scol;                       // ext positions
        {   // two passes over the rows on threads: count the strong connections per row, then fill.  Only the connections BELOW
            // the diagonal are kept: they are all the aggregation looks at (a row joins, or waits for, the eligible neighbour with
            // the smallest id below its own -- aggregate()), and half of the graph
            auto strong = [&](index_t i, nnz_t k) {
                const index_t j = A.col[k];
                if (j >= i + lo) return false;
                const value_t s_ = -A.val[k] / maxPerRow[i], st = -A.val[k] / maxExt[(size_t)aext[(size_t)k]];
                return s_ > opts.connStrength || st > opts.connStrength;
            };
            parallel_rows(nloc, &A.ptr, [&](int, index_t r0, index_t r1) {
                for (index_t i = r0; i < r1; ++i) {
                    nnz_t cnt = 0;
                    for (nnz_t k = A.ptr[i]; k < A.ptr[i + 1]; ++k) cnt += strong(i, k) ? 1 : 0;
                    sptr[(size_t)i + 1] = cnt;
                }
            });
            for (index_t i = 0; i < nloc; ++i) sptr[(size_t)i + 1] += sptr[(size_t)i];
            scol.resize((size_t)sptr[(size_t)nloc]);
            parallel_rows(nloc, &A.ptr, [&](int, index_t r0, index_t r1) {
                for (index_t i = r0; i < r1; ++i) {
                    nnz_t q = sptr[(size_t)i];
                    for (nnz_t k = A.ptr[i]; k < A.ptr[i + 1]; ++k) if (strong(i, k)) scol[(size_t)q++] = aext[(size_t)k];
                }
            });
        }
        pt.lap("strength graph");

        // ---- aggregation: the synchronous rounds of aggregate() above.  A round starts by refreshing the state of the
        //      halo rows; as there, a waiting row hangs on the chain of the one row it waits for -- local or halo -- and is
        //      evaluated again when that row changes state ----
        std::vector<AggState> st((size_t)next);
        for (index_t i = 0; i < nloc; ++i) st[i] = AggState{i + lo, 0, 0, {0, 0}};
        for (index_t j = nloc; j < next; ++j) st[j] = AggState{planA.wanted[(size_t)(j - nloc)], 0, 0, {0, 0}};
        std::vector<index_t> aggregate2((size_t)nloc), cursor((size_t)nloc, 0), whead((size_t)next, -1), wnext((size_t)nloc, -1);
        std::vector<char> dec_nei((size_t)nloc, 0), is_root_nei((size_t)nloc, 0);
        std::vector<index_t> work((size_t)nloc);
        for (index_t i = 0; i < nloc; ++i) work[i] = i;
        const int T = n_threads();
        std::vector<std::vector<index_t>> tnext((size_t)T), tdone((size_t)T);
        long rounds = 0, my_undecided = nloc;
        std::vector<AggState> sent(planA.serve.size());                     // what every asker holds of the rows served to it
        for (size_t q = 0; q < sent.size(); ++q) sent[q] = AggState{planA.serve[q] + lo, 0, 0, {0, 0}};
        while (true) {
            {   // halo refresh; the rows that look at a halo row whose state changed join this round's work list.  Only the
                // CHANGES travel: (position in the asker's list, new state) of the served rows whose state differs from what
                // was sent last -- the halo of a 16 M-row block is millions of rows and a level takes ~10^3 rounds, in most of
                // which a handful of them change
                std::vector<AggDelta> out;
                std::vector<int> oc((size_t)np, 0), ic;
                size_t q = 0;
                for (int p = 0; p < np; ++p)
                    for (int k = 0; k < planA.rcount[(size_t)p]; ++k, ++q) {
                        const AggState &now = st[(size_t)planA.serve[q]];
                        AggState &was = sent[q];
                        if (was.agg != now.agg || was.decided != now.decided || was.is_root != now.is_root) { was = now; out.push_back(AggDelta{(index_t)k, now}); ++oc[(size_t)p]; }
                    }
                const std::vector<AggDelta> in = c.alltoallv_records(out, oc, &ic);
                size_t at = 0, base = 0;
                for (int p = 0; p < np; ++p) {                                  // the ids asked of rank p are wanted[base, base + scount[p])
                    for (int k = 0; k < ic[(size_t)p]; ++k, ++at) {
                        const size_t j = base + (size_t)in[at].slot;
                        st[(size_t)nloc + j] = in[at].s;
                        for (index_t r = whead[(size_t)nloc + j]; r >= 0; r = wnext[(size_t)r]) work.push_back(r);      // the rows that waited for it
                        whead[(size_t)nloc + j] = -1;
                    }
                    base += (size_t)planA.scount[(size_t)p];
                }
            }
            const index_t nw = (index_t)work.size();
            parallel_rows_if_large(nw, [&](int, index_t a0, index_t a1) {
                for (index_t q = a0; q < a1; ++q) {
                    const index_t i = work[q];
                    aggregate2[i] = st[i].agg; dec_nei[i] = 1; is_root_nei[i] = 0;
                    // (aggregate(): the first eligible column below the diagonal, resuming where the last evaluation stopped; a
                    //  row's columns ascend in GLOBAL id -- halo rows below this block, the block, halo rows above it)
                    const index_t own = i + lo;
                    nnz_t it = sptr[i] + cursor[i];
                    for (; it < sptr[i + 1]; ++it) {
                        const index_t e = scol[it];
                        const index_t gid = e < nloc ? e + lo : planA.wanted[(size_t)(e - nloc)];
                        if (gid >= own) break;
                        const AggState &n = st[(size_t)e];
                        if (!n.decided || n.is_root) {
                            aggregate2[i] = n.agg; dec_nei[i] = n.decided; is_root_nei[i] = n.is_root;
                            if (!n.decided) wnext[(size_t)i] = __atomic_exchange_n(&whead[(size_t)e], i, __ATOMIC_ACQ_REL);      // wait for e
                            break;
                        }
                    }
                    cursor[i] = (index_t)(it - sptr[i]);
                }
            });
            for (auto &v : tdone) v.clear();
            parallel_rows_if_large(nw, [&](int t, index_t a0, index_t a1) {
                for (index_t q = a0; q < a1; ++q) {
                    const index_t i = work[q];
                    if (dec_nei[i]) {
                        st[i].decided = 1;
                        if (st[i].agg == aggregate2[i]) st[i].is_root = 1;
                        else if (is_root_nei[i]) st[i].agg = aggregate2[i];
                        tdone[t].push_back(i);
                    }
                }
            });
            for (auto &v : tnext) v.clear();
            std::vector<index_t> done;
            for (auto &v : tdone) done.insert(done.end(), v.begin(), v.end());
            my_undecided -= (long)done.size();
            parallel_rows_if_large((index_t)done.size(), [&](int t, index_t a0, index_t a1) {
                for (index_t q = a0; q < a1; ++q) {
                    for (index_t r = whead[(size_t)done[q]]; r >= 0; r = wnext[(size_t)r]) tnext[t].push_back(r);
                    whead[(size_t)done[q]] = -1;
                }
            });
            work.clear();
            for (auto &v : tnext) work.insert(work.end(), v.begin(), v.end());
            ++rounds;
            long undecided = my_undecided;
            c.allreduce_sum_i64(&undecided, 1);
            if (undecided == 0) break;
            if (rounds > 4L * Mbig + 16) throw std::runtime_error("aggregation did not terminate");
        }
        // coarse numbering: roots in ascending fine order (aggregate_index_update); splitNew[r] = roots below rank r's block
        std::vector<index_t> root_cid((size_t)nloc, -1);
        index_t nroots = 0;
        for (index_t i = 0; i < nloc; ++i) if (st[i].is_root) root_cid[i] = nroots++;
        const std::vector<index_t> allroots = c.allgather_one(nroots);
        std::vector<index_t> splitNew((size_t)np + 1, 0);
        for (int p = 0; p < np; ++p) splitNew[p + 1] = splitNew[p] + allroots[p];
        const index_t new_size = splitNew[np];
        for (auto &x : root_cid) if (x >= 0) x += splitNew[me];
        std::vector<index_t> aggc((size_t)nloc);            // coarse id of every local fine row
        {
            std::vector<index_t> far;
            for (index_t i = 0; i < nloc; ++i) if (st[i].agg < lo || st[i].agg >= hi) far.push_back(st[i].agg);
            FetchPlan pr;
            pr.build(c, split, far);
            const std::vector<index_t> h = pr.values(c, root_cid);
            for (index_t i = 0; i < nloc; ++i) {
                const index_t g = st[i].agg;
                aggc[i] = (g >= lo && g < hi) ? root_cid[(size_t)(g - lo)] : h[(size_t)pr.pos(g)];
                if (aggc[i] < 0) throw std::runtime_error("aggregation: a row joined a non-root");
            }
        }
        if (std::getenv("SAENA_SETUP_TIMING") && me == 0) fprintf(stderr, "[aggregate] %ld rounds\n", rounds);
        pt.lap("aggregation");
        int ret_val = 0;
        if (opts.dynamic_levels) {                                           // setup1:385-405
            if ((unsigned)new_size <= least_row_threshold) ret_val = 1;
            else if (static_cast<float>(new_size) / Mbig > row_reduction_up_thrshld) ret_val = 1;
        }
        // coarse partition: by the owner of the root (ranks the previous levels agglomerated away own no fine rows, hence
        // no roots); whether THIS level is agglomerated further is decided below, once its operator exists
        std::vector<index_t> splitC = splitNew;
        index_t clo = splitC[me], chi = splitC[me + 1];

        // ---- P = (I - omega D^-1 A) P_tentative: local fine rows, global coarse columns ----
        std::vector<index_t> aggcExt = aggc;
        { const std::vector<index_t> h = planA.values(c, aggc); aggcExt.insert(aggcExt.end(), h.begin(), h.end()); }
        Csr P;
        P.nrows = nloc; P.ncols = new_size; P.ptr.assign((size_t)nloc + 1, 0);
        {   // rows are independent: chunks of rows on threads (per-thread output, concatenated in row order)
            const int T = n_threads();
            std::vector<std::vector<index_t>> tcol((size_t)T);
            std::vector<std::vector<value_t>> tval((size_t)T);
            std::vector<index_t> tlo((size_t)T, 0), thi((size_t)T, 0);
            std::vector<nnz_t> rowlen((size_t)nloc, 0);
            parallel_rows(nloc, &A.ptr, [&](int t, index_t r0, index_t r1) {
                tlo[(size_t)t] = r0; thi[(size_t)t] = r1;
                std::vector<std::pair<index_t, value_t>> row;
                std::vector<std::pair<long, value_t>> sort_scratch;
                auto &oc = tcol[(size_t)t];
                auto &ov = tval[(size_t)t];
                // room for this thread's piece of P from the density of its first rows (P holds 6/7 of A's entries on the stencil
                // level, 1/4 on the next: half of A's, the earlier guess, made the vectors of the fine level grow -- and a vector
                // that grows re-touches gigabytes)
                const index_t probe = std::min<index_t>(r1, r0 + 4096);
                for (index_t i = r0; i < r1; ++i) {
                    if (i == probe && A.ptr[i] > A.ptr[r0]) {
                        const size_t est = (size_t)(1.1 * (double)oc.size() / (double)(A.ptr[i] - A.ptr[r0]) * (double)(A.ptr[r1] - A.ptr[r0])) + 4096;
                        LazyBigalloc lazy;                                         // an estimate: advised, not touched
                        oc.reserve(est); ov.reserve(est);
                    }
                    row.clear();
                    for (nnz_t k = A.ptr[i]; k < A.ptr[i + 1]; ++k) {
                        value_t vtmp = -om * inv_diag[i] * A.val[k];
                        if (i + lo == A.col[k]) vtmp += 1;
                        row.emplace_back(aggcExt[(size_t)aext[(size_t)k]], vtmp);
                    }
                    stable_sort_by_first(row, sort_scratch);
                    nnz_t cnt = 0;
                    for (size_t q = 0; q < row.size(); ++q) {
                        value_t v = row[q].second;
                        while (q + 1 < row.size() && row[q + 1].first == row[q].first) v += row[++q].second;
                        if (std::fabs(v) > SAENA_ALMOST_ZERO) { oc.push_back(row[q].first); ov.push_back(v); ++cnt; }
                    }
                    rowlen[(size_t)i] = cnt;
                }
            });
            for (index_t i = 0; i < nloc; ++i) P.ptr[i + 1] = P.ptr[i] + rowlen[(size_t)i];
            P.col.resize((size_t)P.ptr[nloc]); P.val.resize((size_t)P.ptr[nloc]);
            ThreadPool::get().run(T, [&](int t) {                            // every thread brings its own piece home
                if (thi[(size_t)t] <= tlo[(size_t)t]) return;
                std::copy(tcol[(size_t)t].begin(), tcol[(size_t)t].end(), P.col.begin() + P.ptr[tlo[(size_t)t]]);
                std::copy(tval[(size_t)t].begin(), tval[(size_t)t].end(), P.val.begin() + P.ptr[tlo[(size_t)t]]);
                std::vector<index_t>().swap(tcol[(size_t)t]); std::vector<value_t>().swap(tval[(size_t)t]);
            });
        }
        pt.lap("smoothed P");

        // ---- R = P^T, rows to the owners of the coarse rows (under splitC), fine columns ascending ----
        Csr R;
        {
            // Most of P's columns are coarse ids this rank owns itself (aggregates rarely cross the partition): those entries
            // are transposed in place; only the others travel, as (coarse row, fine col, value) triples in (row, col) order --
            // a counting sort over the coarse ids they touch, which is also grouped by destination rank (owners are
            // contiguous ranges of coarse ids).
            const bool own_any = chi > clo;
            auto is_local = [&](index_t cidx) { return own_any && cidx >= clo && cidx < chi; };
            index_t cmin = new_size, cmax = -1;
            nnz_t nfar = 0;
            {
                const int Tn = n_threads();
                std::vector<index_t> tmin((size_t)Tn, new_size), tmax((size_t)Tn, -1);
                std::vector<nnz_t> tfar((size_t)Tn, 0);
                parallel_chunks<size_t>(P.col.size(), (size_t)1 << 20, [&](int t, size_t a, size_t b) {
                    index_t mn = tmin[(size_t)t], mx = tmax[(size_t)t];
                    nnz_t nf = 0;
                    for (size_t k = a; k < b; ++k) { const index_t cidx = P.col[k]; if (!is_local(cidx)) { mn = std::min(mn, cidx); mx = std::max(mx, cidx); ++nf; } }
                    tmin[(size_t)t] = mn; tmax[(size_t)t] = mx; tfar[(size_t)t] += nf;
                });
                for (int t = 0; t < Tn; ++t) { cmin = std::min(cmin, tmin[(size_t)t]); cmax = std::max(cmax, tmax[(size_t)t]); nfar += tfar[(size_t)t]; }
            }
            std::vector<cooEntry> send((size_t)nfar);
            std::vector<int> sc((size_t)np, 0), rcnt;
            if (nfar) {
                std::vector<nnz_t> at((size_t)(cmax - cmin) + 2, 0);
                for (index_t cidx : P.col) if (!is_local(cidx)) at[(size_t)(cidx - cmin) + 1]++;
                for (size_t j = 0; j + 1 < at.size(); ++j) at[j + 1] += at[j];
                for (index_t i = 0; i < nloc; ++i)
                    for (nnz_t k = P.ptr[i]; k < P.ptr[i + 1]; ++k)
                        if (!is_local(P.col[k])) send[(size_t)at[(size_t)(P.col[k] - cmin)]++] = cooEntry(P.col[k], i + lo, P.val[k]);
                for (const auto &x : send) sc[owner_of_id(splitC, x.row)]++;
            }
            // received blocks come from ranks in ascending order = ascending fine ids, each block in (row, col) order; the
            // local entries' fine ids lie between those of the lower and of the higher ranks: filling "lower ranks, own
            // entries by ascending fine row, higher ranks" leaves every coarse row's columns ascending
            const std::vector<cooEntry> got = c.alltoallv_records(send, sc, &rcnt);
            size_t nlow = 0;
            for (int p = 0; p < me; ++p) nlow += (size_t)rcnt[(size_t)p];
            R.nrows = chi - clo; R.ncols = Mbig; R.ptr.assign((size_t)R.nrows + 1, 0);
            for (const auto &x : got) R.ptr[(size_t)(x.row - clo) + 1]++;
            // the local entries on threads: every thread OWNS a range of coarse rows and walks the fine rows whose columns reach into
            // it (a row of P ascends: its first and last column say so), so counts and cursors need no atomics and every coarse row
            // is filled by ascending fine row, as the sequential loop filled it
            const int Tr = own_any ? std::max(1, std::min<int>(n_threads(), (chi - clo) / 4096 + 1)) : 1;
            auto own_range = [&](int t, index_t &c0, index_t &c1) { c0 = clo + (index_t)((long)(chi - clo) * t / Tr); c1 = clo + (index_t)((long)(chi - clo) * (t + 1) / Tr); };
            auto over_local = [&](int t, auto &&f) {
                index_t c0, c1;
                own_range(t, c0, c1);
                if (c1 <= c0) return;
                for (index_t i = 0; i < nloc; ++i) {
                    const nnz_t k0 = P.ptr[i], k1 = P.ptr[i + 1];
                    if (k1 == k0 || P.col[(size_t)k1 - 1] < c0 || P.col[(size_t)k0] >= c1) continue;
                    for (nnz_t k = k0; k < k1; ++k) { const index_t cidx = P.col[(size_t)k]; if (cidx >= c0 && cidx < c1) f(i, k, cidx); }
                }
            };
            if (own_any) ThreadPool::get().run(Tr, [&](int t) { over_local(t, [&](index_t, nnz_t, index_t cidx) { R.ptr[(size_t)(cidx - clo) + 1]++; }); });
            for (index_t i = 0; i < R.nrows; ++i) R.ptr[i + 1] += R.ptr[i];
            const size_t tot = (size_t)R.ptr[(size_t)R.nrows];
            R.col.resize(tot); R.val.resize(tot);
            std::vector<nnz_t> at(R.ptr.begin(), R.ptr.end() - 1);
            auto put = [&](const cooEntry &x) { const nnz_t q = at[(size_t)(x.row - clo)]++; R.col[(size_t)q] = x.col; R.val[(size_t)q] = x.val; };
            for (size_t j = 0; j < nlow; ++j) put(got[j]);
            if (own_any) ThreadPool::get().run(Tr, [&](int t) {
                over_local(t, [&](index_t i, nnz_t k, index_t cidx) { const nnz_t q = at[(size_t)(cidx - clo)]++; R.col[(size_t)q] = i + lo; R.val[(size_t)q] = P.val[(size_t)k]; });
            });
            for (size_t j = nlow; j < got.size(); ++j) put(got[j]);
        }
        pt.lap("R = P^T");

        // ---- Ac = (R A) P (triple_mat_mult): rows of A, then rows of P, fetched for the columns outside this block ----
        Csr RA;
        {
            FetchPlan pl;
            pl.build(c, split, outside_cols(R, lo, hi));
            Csr Ahalo = pl.rows(c, A);
            std::vector<index_t> rcol = relabel_cols(R, lo, hi, pl);
            pt.lap("R*A: fetch, relabel");
            RA = spgemm_stacked(CsrRef(R, rcol), A, Ahalo, clo);
            pt.lap("R*A: product");
            drop_async(std::move(Ahalo)); drop_async(std::move(rcol));
        }
        pt.lap("R*A: free");
        Csr AcN;
        {
            FetchPlan pl;
            pl.build(c, split, outside_cols(RA, lo, hi));
            Csr Phalo = pl.rows(c, P);
            std::vector<index_t> rcol = relabel_cols(RA, lo, hi, pl);
            pt.lap("(RA)*P: fetch, relabel");
            AcN = spgemm_stacked(CsrRef(RA, rcol), P, Phalo, clo);
            pt.lap("(RA)*P: product");
            drop_async(std::move(RA)); drop_async(std::move(Phalo)); drop_async(std::move(rcol));
            RA = Csr();
        }
        pt.lap("(RA)*P: free");
        if (++filter_it >= opts.filter_start) {
            if (filter_thre_cur > opts.filter_max) filter_thre_cur = opts.filter_max;
            filter_csr(AcN, filter_thre_cur, clo);
            filter_thre_cur *= std::pow(10, opts.filter_rate);
        }
        pt.lap("filter");

        // ---- the coarse operator's final partition (amg_setup.h): nnz-balanced over the ranks still active (the reference's
        //      Ac->repart(), saena_matrix_repart.cpp:728-980), then agglomerated (decide_shrinking / shrink_set_params);
        //      R's and Ac's rows move to their owners, P's columns are laid out on the same partition ----
        {
            long nnzC = (long)AcN.col.size();
            c.allreduce_sum_i64(&nnzC, 1);
            if (level_stride.size() <= (size_t)l) level_stride.resize((size_t)l + 1, 1);
            const int stride_prev = level_stride[(size_t)l];
            const int stride = next_stride(nnzC, new_size, np, stride_prev);
            level_stride.push_back(stride);
            std::vector<index_t> splitB = splitNew;
            const int active = (np + stride_prev - 1) / stride_prev;
            if (stride < np && active > 1 && new_size >= active && !std::getenv("SAENA_NO_COARSE_REPART")) {
                const std::vector<index_t> sa = nnz_balanced_split(c, new_size, nnzC, active, [&](const std::vector<index_t> &firstSplit, std::vector<long> &H) {
                    const int nb = (int)H.size();
                    for (index_t i = 0; i < AcN.nrows; ++i)
                        H[(size_t)lower_bound2(firstSplit.data(), firstSplit.data() + nb, i + clo)] += (long)(AcN.ptr[i + 1] - AcN.ptr[i]);
                }, 4096);      // (a finer histogram than the fine level's nparts^2 buckets: at 2 ranks those put 75 % of a level on one rank)
                // part a of the active ranks is rank a * stride_prev; the idle ranks in between keep empty blocks
                for (int r = 0; r <= np; ++r) {
                    const int a = std::min(active, (r + stride_prev - 1) / stride_prev);
                    splitB[(size_t)r] = sa[(size_t)a];
                }
            }
            splitC = merge_split(splitB, stride);
            pt.lap("coarse partition");
            if (splitC != splitNew) {
                R = route_rows(c, R, clo, splitC);
                AcN = route_rows(c, AcN, clo, splitC);
                clo = splitC[me]; chi = splitC[me + 1];
                if (R.nrows != chi - clo || AcN.nrows != chi - clo) throw std::runtime_error("coarse repartition: a block has the wrong size");
            }
            if (stride > stride_prev && std::getenv("SAENA_SETUP_TIMING") && me == 0)
                fprintf(stderr, "[setup L%d] level %d (%d rows, %ld nnz) agglomerated: rank stride %d -> %d\n", l, l + 1, new_size, nnzC, stride_prev, stride);
            pt.lap("rows to their owners");
        }
        // ---- this level's transfer operators in the reference's layout ----
        {
            dist_level &d = dist.back();
            long nn = (long)P.col.size();
            c.allreduce_sum_i64(&nn, 1);
            d.nnzP = nn;
            d.P.build_from_csr(c, P.ptr, P.col, P.val, split, splitC);
            pt.lap("layout of P");
            d.R.build_from_csr(c, R.ptr, R.col, R.val, splitC, split);
            pt.lap("layout of R");
        }
        // ---- next level ----
        std::vector<value_t> invd((size_t)(chi - clo), 1.0);
        parallel_rows(chi - clo, &AcN.ptr, [&](int, index_t r0, index_t r1) {
            for (index_t i = r0; i < r1; ++i)
                for (nnz_t k = AcN.ptr[i]; k < AcN.ptr[i + 1]; ++k)
                    if (AcN.col[k] == i + clo) {
                        if (std::fabs(AcN.val[k]) < SAENA_ALMOST_ZERO) throw std::runtime_error("there is a zero diagonal element at row index = " + std::to_string(i + clo));
                        invd[(size_t)i] = 1.0 / AcN.val[k];
                    }
        });
        drop_async(std::move(A));                                         // (this level's operator, transfers and graph: gigabytes on the fine levels)
        drop_async(std::move(P)); drop_async(std::move(R)); drop_async(std::move(aext)); drop_async(std::move(scol));
        A = std::move(AcN);
        A.ncols = new_size;
        inv_diag.swap(invd);
        split = splitC;
        Mbig = new_size;
        pt.lap("inverse diagonal");
        if (ret_val == 1) max_level = l + 1;                              // :287-289 this will be the last level
    }
    levels.resize(1);
    return 0;
}

int amg_hierarchy::setup_distributed(saena_matrix *Ad, const amg_options &o) {
    Comm &c = *Ad->comm;
    if (!Ad->assembled) throw std::runtime_error("amg setup: the matrix is not assembled");
    if (c.nranks == 1) { setup(Ad, o); dist.clear(); return 0; }
    // default: every rank builds only its rows (setup_rows_distributed).  The older gathered form (every rank builds
    // the whole hierarchy, then keeps its rows; SAENA_SETUP=gathered) remains as a cross-check.
    const char *mode = std::getenv("SAENA_SETUP");
    if (!(mode && std::string(mode) == "gathered")) return setup_rows_distributed(Ad, o);
    // gather the fine operator on every rank (entries carry global ids)
    std::vector<int> counts = c.allgather_one((int)Ad->entry.size());
    std::vector<size_t> sc((size_t)c.nranks, Ad->entry.size() * sizeof(cooEntry)), sd((size_t)c.nranks, 0), rc((size_t)c.nranks), rd((size_t)c.nranks);
    size_t tot = 0;
    for (int p = 0; p < c.nranks; ++p) { rc[p] = (size_t)counts[p] * sizeof(cooEntry); rd[p] = tot; tot += rc[p]; }
    std::vector<cooEntry> all(tot / sizeof(cooEntry));
    c.alltoallv(Ad->entry.data(), sc.data(), sd.data(), all.data(), rc.data(), rd.data());
    self_comm.reset(new SelfComm());
    A_global.reset(new saena_matrix(self_comm.get()));
    A_global->remove_boundary = false;
    for (const auto &e : all) A_global->set(e.row, e.col, e.val);
    all.clear(); all.shrink_to_fit();
    A_global->assemble();
    A_global->eig_max_of_invdiagXA = Ad->eig_max_of_invdiagXA;
    setup(A_global.get(), o);
    Ad->eig_max_of_invdiagXA = A_global->eig_max_of_invdiagXA;
    distribute(c, Ad->split);
    return 0;
}

} // namespace saena_host

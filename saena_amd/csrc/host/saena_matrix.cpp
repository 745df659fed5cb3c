// saena_matrix.cpp -- host-side assembly of the distributed operators.
// Restates, with our own data structures, what the reference does between
// saena::matrix::set() and the first matvec (citations: file:line in paralab/Saena).
#include "saena_matrix.h"
#include "par.h"

#include <algorithm>
#include <cmath>
#include <fstream>
#include <functional>
#include <numeric>
#include <iomanip>
#include <sstream>
#include <chrono>
#include <cstdio>
#include <cstdlib>

namespace saena_host {

namespace {
struct AsmTimer {         // SAENA_SETUP_TIMING=1: phase times of assemble() on stderr
    bool on = std::getenv("SAENA_SETUP_TIMING") != nullptr;
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    void lap(const char *what) {
        if (!on) return;
        const auto n = std::chrono::steady_clock::now();
        fprintf(stderr, "[assemble] %-28s %8.3f s\n", what, std::chrono::duration<double>(n - t).count());
        t = n;
    }
};
} // namespace

long lower_bound2(const index_t *left, const index_t *right, index_t val) {
    // aux_functions.h:39-58: position p with split[p] <= val < split[p+1]
    const index_t *first = left;
    const index_t *it = std::upper_bound(left, right + 1, val);   // first element > val
    return (it - first) - 1;
}

static int owner_of(const std::vector<index_t> &split, index_t id) {
    const int np = (int)split.size() - 1;
    long p = lower_bound2(split.data(), split.data() + np, id);
    // empty blocks share a boundary value: settle on the block that really contains id
    while (p < np - 1 && split[p + 1] <= id) ++p;
    while (p > 0 && split[p] > id) --p;
    return (int)p;
}

// send each record to the rank owning key(record); returns what this rank receives
template <class T, class KeyFn>
static std::vector<T> route(Comm &comm, const std::vector<T> &recs, const std::vector<index_t> &split, KeyFn key) {
    const int np = comm.nranks;
    if (np == 1) return recs;
    std::vector<int> cnt((size_t)np, 0), dest(recs.size());
    for (size_t i = 0; i < recs.size(); ++i) { dest[i] = owner_of(split, key(recs[i])); cnt[dest[i]]++; }
    std::vector<size_t> off((size_t)np + 1, 0);
    for (int p = 0; p < np; ++p) off[p + 1] = off[p] + cnt[p];
    std::vector<T> send(recs.size());
    std::vector<size_t> fill(off.begin(), off.end() - 1);
    for (size_t i = 0; i < recs.size(); ++i) send[fill[dest[i]]++] = recs[i];
    return comm.alltoallv_records(send, cnt);
}

template <class T, class KeyFn>
static std::vector<T> route(Comm &comm, std::vector<T> &&recs, const std::vector<index_t> &split, KeyFn key) {
    if (comm.nranks == 1) return std::move(recs);
    return route(comm, static_cast<const std::vector<T> &>(recs), split, key);
}

// Column-major order of entries that arrive row-major sorted (what the generators, the dedup pass and a route()
// over ascending row blocks all produce): a counting sort by column keeps rows ascending inside a column, one O(n)
// pass instead of an O(n log n) comparison sort -- 114 M entries of Poisson 256^3: 25 s -> 1.5 s.  Falls back to
// std::sort when the input is not row-major sorted or the column range is too wide for a count array.
static void sort_col_major(std::vector<cooEntry> &e) {
    if (e.size() < 2) return;
    index_t cmin = e[0].col, cmax = e[0].col;
    bool sorted = true;
    for (size_t i = 0; i < e.size(); ++i) {
        cmin = std::min(cmin, e[i].col); cmax = std::max(cmax, e[i].col);
        if (i && row_major(e[i], e[i - 1])) sorted = false;
    }
    const size_t range = (size_t)cmax - (size_t)cmin + 1;
    if (!sorted || range > 8 * e.size() + (1u << 20)) { std::sort(e.begin(), e.end(), col_major); return; }
    std::vector<size_t> at(range + 1, 0);
    for (const auto &x : e) at[(size_t)(x.col - cmin) + 1]++;
    for (size_t c = 0; c < range; ++c) at[c + 1] += at[c];
    std::vector<cooEntry> out(e.size());
    for (const auto &x : e) out[at[(size_t)(x.col - cmin)]++] = x;
    e.swap(out);
}

// every rank's list, concatenated in rank order, on every rank
template <class T>
static std::vector<T> allgatherv(Comm &comm, const std::vector<T> &mine) {
    const int np = comm.nranks;
    if (np == 1) return mine;
    std::vector<int> counts = comm.allgather_one((int)mine.size());
    std::vector<size_t> sc((size_t)np, mine.size() * sizeof(T)), sd((size_t)np, 0), rc((size_t)np), rd((size_t)np);
    size_t tot = 0;
    for (int p = 0; p < np; ++p) { rc[p] = (size_t)counts[p] * sizeof(T); rd[p] = tot; tot += rc[p]; }
    std::vector<T> all(tot / sizeof(T));
    comm.alltoallv(mine.data(), sc.data(), sd.data(), all.data(), rc.data(), rd.data());
    return all;
}

// ---------------------------------------------------------------------------
int saena_matrix::set(index_t row, index_t col, value_t val) {
    data_in.emplace_back(row, col, val);
    return 0;
}
int saena_matrix::set(const index_t *row, const index_t *col, const value_t *val, nnz_t n) {
    data_in.reserve(data_in.size() + (size_t)n);
    for (nnz_t i = 0; i < n; ++i) data_in.emplace_back(row[i], col[i], val[i]);
    return 0;
}

int saena_matrix::read_file(const std::string &filename, const std::string &input_type) {
    Comm &c = *comm;
    const size_t ext = filename.find_last_of('.');
    if (ext == std::string::npos || ext == filename.size() - 1) throw std::runtime_error("The matrix file name does not have an extension!");
    const std::string fe = filename.substr(ext + 1);
    std::vector<cooEntry> all;
    if (fe == "mtx") {
        std::ifstream in(filename);
        if (!in) throw std::runtime_error("Could not open the matrix file <" + filename + ">");
        std::string mat_type(input_type), line;
        std::getline(in, line);
        bool pattern_field = false;
        if (line.rfind("%%MatrixMarket", 0) == 0) {                        // :48-62 metadata line
            std::istringstream hs(line);
            std::string t0, t1, t2, field, symm;
            hs >> t0 >> t1 >> t2 >> field >> symm;
            pattern_field = field == "pattern";
            if (mat_type.empty()) {
                if (field != "pattern") { if (symm == "symmetric") mat_type = "triangle"; }
                else mat_type = "pattern";
            }
        } else {
            in.seekg(0);
        }
        while (in.peek() == '%') in.ignore(1 << 20, '\n');
        long M_in = 0, N_in = 0, nnz = 0;
        in >> M_in >> N_in >> nnz;
        (void)pattern_field;
        index_t a = 0, b = 0;
        value_t v = 0.0;
        if (mat_type.empty()) {
            while (in >> a >> b >> v) all.emplace_back(a - 1, b - 1, v);
        } else if (mat_type == "triangle") {
            while (in >> a >> b >> v) { all.emplace_back(a - 1, b - 1, v); if (a != b) all.emplace_back(b - 1, a - 1, v); }
        } else if (mat_type == "pattern" || mat_type == "tripattern") {     // value 1 for a pattern matrix, mirrored
            while (in >> a >> b) { all.emplace_back(a - 1, b - 1, 1.0); if (a != b) all.emplace_back(b - 1, a - 1, 1.0); }
        } else {
            throw std::runtime_error("the input type is not valid!");
        }
        std::sort(all.begin(), all.end(), col_major);                       // :136 what the .bin holds
    } else if (fe == "bin") {
        std::ifstream in(filename, std::ios::binary | std::ios::ate);
        if (!in) throw std::runtime_error("Could not open the matrix file <" + filename + ">");
        const std::streamsize bytes = in.tellg();
        const size_t n = (size_t)bytes / 16;
        if (n == 0) throw std::runtime_error("number of nonzeros is 0 inside function read_file");
        in.seekg(0);
        all.resize(n);
        static_assert(sizeof(cooEntry) == 16, "cooEntry must be the 16-byte triple");
        in.read(reinterpret_cast<char *>(all.data()), (std::streamsize)(n * 16));
    } else {
        throw std::runtime_error("The extension of file should be either mtx (matrix market) or bin (binary)!");
    }
    // this rank's chunk (:343-361)
    const nnz_t nnz_all = (nnz_t)all.size();
    const nnz_t chunk = nnz_all / c.nranks;
    const nnz_t lo = c.rank * chunk, hi = c.rank == c.nranks - 1 ? nnz_all : lo + chunk;
    data_in.insert(data_in.end(), all.begin() + lo, all.begin() + hi);
    return 0;
}

int saena_matrix::write_bin(const std::string &filename) const {
    std::ofstream out(filename, std::ios::binary);
    if (!out) throw std::runtime_error("could not open <" + filename + "> for writing");
    if (!entry.empty()) {
        out.write(reinterpret_cast<const char *>(entry.data()), (std::streamsize)(entry.size() * sizeof(cooEntry)));
    } else {                                   // one-rank CSR-only operators
        nnz_t k = 0;
        std::vector<cooEntry> e;
        for (index_t i = 0; i < L.M; ++i)
            for (index_t j = 0; j < L.nnzPerRow_local[i]; ++j, ++k) e.emplace_back(i + split[comm->rank], L.col_local[k], L.val_local[k]);
        std::sort(e.begin(), e.end(), col_major);
        out.write(reinterpret_cast<const char *>(e.data()), (std::streamsize)(e.size() * sizeof(cooEntry)));
    }
    return 0;
}

int saena_matrix::writeMatrixToFile(const std::string &name) const {
    if (!assembled) throw std::runtime_error("writeMatrixToFile: the matrix is not assembled");
    const int rank = comm->rank;
    const std::string fn = name + "-r" + std::to_string(rank) + ".mtx";
    std::ofstream out(fn);
    if (!out) throw std::runtime_error("could not open <" + fn + "> for writing");
    if (rank == 0) {
        out << "%%MatrixMarket matrix coordinate real general" << std::endl;
        out << Mbig << "\t" << Mbig << "\t" << nnz_g << std::endl;
    }
    out << std::setprecision(12);
    if (!entry.empty()) {
        for (const auto &e : entry) out << e.row + 1 << "\t" << e.col + 1 << "\t" << e.val << "\n";
    } else {                                   // one-rank CSR-only operators (coarse levels): column-major like `entry`
        std::vector<cooEntry> e;
        nnz_t k = 0;
        for (index_t i = 0; i < L.M; ++i)
            for (index_t j = 0; j < L.nnzPerRow_local[i]; ++j, ++k) e.emplace_back(i + split[rank], L.col_local[k], L.val_local[k]);
        std::sort(e.begin(), e.end(), col_major);
        for (const auto &x : e) out << x.row + 1 << "\t" << x.col + 1 << "\t" << x.val << "\n";
    }
    return 0;
}

// setup_initial_data (saena_matrix_setup.cpp:62-116): remove_duplicates (:118-200)
// then remove_boundary_nodes (:281-365).
void saena_matrix::setup_initial_data() {
    Comm &c = *comm;
    const int np = c.nranks;
    AsmTimer tm;
    // global row count = largest row index + 1 (:126-128)
    long mx = -1;
    for (const auto &e : data_in) mx = std::max<long>(mx, std::max(e.row, e.col));
    Mbig_with_bound = (index_t)(c.max_(mx) + 1);
    if (Mbig_with_bound <= 0) throw std::runtime_error("saena_matrix: no entries were set");
    // even row split (:130-135) and exchange so every rank holds whole rows
    std::vector<index_t> split0((size_t)np + 1);
    const index_t ofst = Mbig_with_bound / np;
    for (int i = 0; i < np; ++i) split0[i] = i * ofst;
    split0[np] = Mbig_with_bound;
    std::vector<cooEntry> mine = route(c, std::move(data_in), split0, [](const cooEntry &e) { return e.row; });
    data_in.clear(); data_in.shrink_to_fit();
    if (!std::is_sorted(mine.begin(), mine.end(), row_major))       // (the generators set() their entries in this order already)
        std::stable_sort(mine.begin(), mine.end(), row_major);
    // duplicates: add or keep the last one; drop |val| <= ALMOST_ZERO (:147-165).  Compacted in place (the write
    // position never overtakes the read position): no second 16 B/entry array.
    std::vector<cooEntry> &wb = mine;
    {
        size_t w = 0;
        for (size_t i = 0; i < mine.size(); ++i) {
            const index_t r = mine[i].row, cl = mine[i].col;
            value_t tmp = mine[i].val;
            while (i + 1 < mine.size() && mine[i + 1].row == r && mine[i + 1].col == cl) {
                ++i;
                tmp = add_duplicates ? tmp + mine[i].val : mine[i].val;
            }
            if (std::fabs(tmp) > SAENA_ALMOST_ZERO) mine[w++] = cooEntry(r, cl, tmp);
        }
        mine.resize(w);
    }
    tm.lap("  route + sort + dedup");

    bound_row_global.clear();
    Mbig = Mbig_with_bound;
    if (remove_boundary) {
        // a row with a single entry is a boundary node (:300-330)
        std::vector<index_t> bnd;
        for (size_t i = 0; i < wb.size();) {
            size_t j = i + 1;
            while (j < wb.size() && wb[j].row == wb[i].row) ++j;
            if (j - i == 1) {
                if (wb[i].row != wb[i].col) throw std::runtime_error("saena_matrix: single-entry row is not diagonal (boundary removal)");
                bnd.push_back(wb[i].row);
            }
            i = j;
        }
        bound_row_global = allgatherv(c, bnd);          // rank order == ascending row order
        if (bound_row_global.empty()) {
            remove_boundary = false;                     // :332-335
        } else {
            const auto &B = bound_row_global;
            // renumbering table over the index range this rank's entries touch: -1 = removed, else the new id
            // (two binary searches per entry over the 390 K boundary nodes of 256^3 cost 20 s of a 64 s assemble)
            index_t lo = Mbig_with_bound, hi = -1;
            for (const auto &e : wb) { lo = std::min(lo, std::min(e.row, e.col)); hi = std::max(hi, std::max(e.row, e.col)); }
            std::vector<index_t> tab(hi >= lo ? (size_t)(hi - lo) + 1 : 0);
            {
                size_t b = (size_t)(std::lower_bound(B.begin(), B.end(), lo) - B.begin());   // boundary nodes below lo
                for (index_t x = lo; x <= hi; ++x) {
                    if (b < B.size() && B[b] == x) { tab[(size_t)(x - lo)] = -1; ++b; }
                    else tab[(size_t)(x - lo)] = x - (index_t)b;
                }
            }
            size_t w = 0;                                   // compacted in place
            for (size_t i = 0; i < wb.size(); ++i) {
                const index_t r = tab[(size_t)(wb[i].row - lo)], cc = tab[(size_t)(wb[i].col - lo)];
                if (r < 0) continue;
                if (cc < 0) throw std::runtime_error("saena_matrix: interior row couples to a removed boundary node");
                wb[w++] = cooEntry(r, cc, wb[i].val);
            }
            wb.resize(w);
            Mbig = Mbig_with_bound - (index_t)B.size();
        }
    }
    tm.lap("  boundary removal");
    entry.swap(wb);      // row-major, whole rows, global (renumbered) ids; ownership still by split0
    nnz_l = (nnz_t)entry.size();
    nnz_g = c.sum(nnz_l);
}

// The reference's nnz-balanced row partition (repartition_nnz_initial, saena_matrix_repart.cpp:43-170; the same
// algorithm re-partitions every coarse operator, saena_matrix::repart, :728-980): nparts^2 near-equal row buckets, merged
// left to right until a part holds ~nnz_g/nparts entries.  `add_local_histogram(firstSplit, H)` adds this rank's
// entries per bucket; the histogram is summed over the ranks here.
std::vector<index_t> nnz_balanced_split(Comm &c, index_t Mbig, nnz_t nnz_g, int nparts,
                                        const std::function<void(const std::vector<index_t> &, std::vector<long> &)> &add_local_histogram,
                                        int min_buckets) {
    std::vector<index_t> split((size_t)nparts + 1, 0);
    if (nparts == 1) { split[1] = Mbig; return split; }
    int n_buckets;
    if (Mbig > nparts * nparts) n_buckets = nparts < 1000 ? nparts * nparts : 1000 * nparts;
    else if (nparts <= Mbig) n_buckets = Mbig;
    else throw std::runtime_error("number of tasks cannot be greater than the number of rows of the matrix.");
    // nparts^2 buckets are the reference's resolution (4 buckets at 2 ranks: a part boundary can only sit at 25, 50 or 75 % of
    // the rows).  The fine operator's partition is pinned by the compiled reference and keeps it; coarse operators ask for
    // a finer histogram.
    if (min_buckets > n_buckets) n_buckets = (int)std::min<long>(min_buckets, Mbig);
    std::vector<index_t> splitOffset((size_t)n_buckets, 0);
    const index_t baseOffset = (index_t)std::floor(1.0 * Mbig / n_buckets);
    const float offsetRes = float(1.0 * Mbig / n_buckets) - baseOffset;
    float offsetResSum = 0;
    for (index_t i = 1; i < n_buckets; ++i) {
        splitOffset[i] = baseOffset;
        offsetResSum += offsetRes;
        if (offsetResSum >= 1) { splitOffset[i]++; offsetResSum -= 1; }
    }
    std::vector<index_t> firstSplit((size_t)n_buckets + 1, 0);
    for (index_t i = 1; i < n_buckets; ++i) firstSplit[i] = firstSplit[i - 1] + splitOffset[i];
    firstSplit[n_buckets] = Mbig;
    std::vector<long> H((size_t)n_buckets, 0);
    add_local_histogram(firstSplit, H);
    c.allreduce_sum_i64(H.data(), n_buckets);
    for (int i = 1; i < n_buckets; ++i) H[i] += H[i - 1];
    const nnz_t NNZ_PROC = nnz_g / nparts;
    index_t procNum = 0;
    for (nnz_t i = 1; i < n_buckets; ++i) {
        if (Mbig - firstSplit[i + 1] < nparts - (procNum + 1)) {
            for (; i < n_buckets; ++i) { procNum++; split[procNum] = firstSplit[i]; }
            break;
        }
        if (H[i] > (procNum + 1) * NNZ_PROC) { ++procNum; split[procNum] = firstSplit[i]; }
    }
    for (index_t p = procNum + 1; p < nparts; ++p) split[p] = Mbig;      // (parts the loop never reached stay empty)
    split[nparts] = Mbig;
    return split;
}

int saena_matrix::default_partition_buckets() {
    const char *e = std::getenv("SAENA_FINE_PARTITION_BUCKETS");
    return e ? std::max(0, std::atoi(e)) : 0;
}

void saena_matrix::repartition_nnz_initial() {
    Comm &c = *comm;
    const int nprocs = c.nranks;
    split = nnz_balanced_split(c, Mbig, nnz_g, nprocs, [this](const std::vector<index_t> &firstSplit, std::vector<long> &H) {
        const int n_buckets = (int)H.size();
        for (const auto &e : entry) H[lower_bound2(firstSplit.data(), firstSplit.data() + n_buckets, e.row)]++;
    }, partition_buckets);
    // move the entries to their owners (saena_matrix_repart.cpp:293 MPI_Alltoallv) and sort column-major
    entry = route(c, std::move(entry), split, [](const cooEntry &e) { return e.row; });
    sort_col_major(entry);
    M = split[c.rank + 1] - split[c.rank];
    nnz_l = (nnz_t)entry.size();
}

// inverse_diag (saena_matrix_setup.cpp:1562-1600)
void saena_matrix::inverse_diag() {
    inv_diag.assign((size_t)M, 1.0);
    const index_t ofs = split[comm->rank];
    for (const auto &e : entry)
        if (e.row == e.col) {
            if (std::fabs(e.val) < SAENA_ALMOST_ZERO)
                throw std::runtime_error("there is a zero diagonal element at row index = " + std::to_string(e.row));
            inv_diag[e.row - ofs] = 1.0 / e.val;
        }
}

void saena_matrix::matrix_setup() {            // saena_matrix_setup.cpp:507-560
    inverse_diag();
    L.build(*comm, entry, split, split);
    assembled = true;
}

int saena_matrix::assemble() {                 // saena_matrix_setup.cpp:4-17
    AsmTimer tm;
    setup_initial_data();      tm.lap("setup_initial_data");
    repartition_nnz_initial(); tm.lap("repartition_nnz_initial");
    matrix_setup();            tm.lap("matrix_setup");
    return 0;
}

int saena_matrix::assemble_with_split(const std::vector<index_t> &split_in) {
    Comm &c = *comm;
    if ((int)split_in.size() != c.nranks + 1) throw std::runtime_error("assemble_with_split: split has the wrong length");
    setup_initial_data();
    if (split_in.back() != Mbig) throw std::runtime_error("assemble_with_split: split does not cover the matrix");
    split = split_in;
    entry = route(c, std::move(entry), split, [](const cooEntry &e) { return e.row; });
    sort_col_major(entry);
    M = split[c.rank + 1] - split[c.rank];
    nnz_l = (nnz_t)entry.size();
    matrix_setup();
    return 0;
}

std::vector<value_t> saena_matrix::remove_boundary_rhs(const std::vector<value_t> &v, index_t lo) const {
    // remove_boundary_rhs (saena_object.cpp:699-730): boundary rows are dropped, the rest keeps its order
    if (bound_row_global.empty()) return v;
    std::vector<value_t> out;
    out.reserve(v.size());
    const auto &B = bound_row_global;
    auto it = std::lower_bound(B.begin(), B.end(), lo);
    for (size_t i = 0; i < v.size(); ++i) {
        const index_t g = lo + (index_t)i;
        if (it != B.end() && *it == g) { ++it; continue; }
        out.push_back(v[i]);
    }
    return out;
}

std::vector<value_t> saena_matrix::scatter_rhs(const std::vector<index_t> &idx, const std::vector<value_t> &val) const {
    if (!assembled) throw std::runtime_error("scatter_rhs: the matrix is not assembled");
    if (idx.size() != val.size()) throw std::runtime_error("scatter_rhs: index/value size mismatch");
    Comm &c = *comm;
    struct rec { index_t id; value_t v; };
    const auto &B = bound_row_global;
    std::vector<rec> recs;
    recs.reserve(idx.size());
    for (size_t i = 0; i < idx.size(); ++i) {
        const index_t g = idx[i];
        auto it = std::lower_bound(B.begin(), B.end(), g);
        if (it != B.end() && *it == g) continue;                       // boundary row: dropped (saena_object.cpp:716-724)
        const index_t nid = g - (index_t)(it - B.begin());
        if (nid < 0 || nid >= Mbig) throw std::runtime_error("scatter_rhs: index outside the matrix");
        recs.push_back({nid, val[i]});
    }
    std::vector<rec> got = route(c, recs, split, [](const rec &r) { return r.id; });
    std::vector<value_t> out((size_t)M, 0.0);
    std::vector<char> seen((size_t)M, 0);
    const index_t ofs = split[c.rank];
    for (const auto &r : got) { out[r.id - ofs] = r.v; seen[r.id - ofs] = 1; }
    for (index_t i = 0; i < M; ++i)
        if (!seen[i]) throw std::runtime_error("scatter_rhs: the right-hand side does not cover row " + std::to_string(i + ofs));
    return out;
}

// ---------------------------------------------------------------------------
// set_off_on_diagonal (saena_matrix_setup.cpp:793-1098)
void DistLayout::build(Comm &c, const std::vector<cooEntry> &entry, const std::vector<index_t> &split_row,
                       const std::vector<index_t> &split_col) {
    const int nprocs = c.nranks, rank = c.rank;
    *this = DistLayout();
    M = split_row[rank + 1] - split_row[rank];
    N_local = split_col[rank + 1] - split_col[rank];
    col_offset = split_col[rank];
    const nnz_t nnz_l = (nnz_t)entry.size();
    nnzPerRow_local.assign((size_t)M, 0);
    recvCount.assign((size_t)nprocs, 0);
    nnzPerProcScan.assign((size_t)nprocs + 1, 0);
    std::vector<std::pair<nnz_t, nnz_t>> loc_runs;                       // [begin, end) runs of `entry` that are local
    nnz_t i = 0;
    while (i < nnz_l) {                                                  // :828-859
        const long procNum = owner_of(split_col, entry[i].col);
        if (procNum == rank) {
            const nnz_t b = i;
            while (i < nnz_l && entry[i].col < split_col[procNum + 1]) {
                ++nnzPerRow_local[entry[i].row - split_row[rank]];
                ++i;
            }
            loc_runs.emplace_back(b, i);
        } else {
            const nnz_t tmp = i;
            while (i < nnz_l && entry[i].col < split_col[procNum + 1]) {
                vElement_remote.push_back(entry[i].col);
                ++recvCount[procNum];
                nnzPerCol_remote.push_back(0);
                do {
                    col_remote.push_back((index_t)vElement_remote.size() - 1);
                    col_remote2.push_back(entry[i].col);
                    row_remote.push_back(entry[i].row - split_row[rank]);
                    val_remote.push_back(entry[i].val);
                    ++nnzPerCol_remote.back();
                } while (++i < nnz_l && entry[i].col == entry[i - 1].col);
            }
            nnzPerProcScan[procNum + 1] = i - tmp;
        }
    }
    nnz_l_local = 0;
    for (const auto &r : loc_runs) nnz_l_local += r.second - r.first;
    nnz_l_remote = (nnz_t)row_remote.size();
    col_remote_size = (index_t)vElement_remote.size();
    recvCount[rank] = 0;
    // :905 row-major order.  The local entries arrive column-major (rows ascending inside a column), so a counting
    // sort by row -- the row lengths are known -- yields (row, column) order in one pass instead of an O(n log n) sort,
    // straight from `entry` (no intermediate copy).  row_local (which the reference fills and its matvec never
    // reads) is not materialised: it is the row-pointer expansion of nnzPerRow_local.
    col_local.resize((size_t)nnz_l_local); val_local.resize((size_t)nnz_l_local);
    {
        std::vector<nnz_t> at((size_t)M + 1, 0);
        for (index_t r = 0; r < M; ++r) at[r + 1] = at[r] + nnzPerRow_local[r];
        const index_t r0 = split_row[rank];
        for (const auto &run : loc_runs)
            for (nnz_t k2 = run.first; k2 < run.second; ++k2) {
                const nnz_t k = at[entry[k2].row - r0]++;
                col_local[k] = entry[k2].col; val_local[k] = entry[k2].val;
            }
    }
    finish_plan(c, split_col);
}

// the halo plan from what the classification left: recvCount, nnzPerProcScan (per rank), vElement_remote
void DistLayout::finish_plan(Comm &c, const std::vector<index_t> &split_col) {
    const int nprocs = c.nranks, rank = c.rank;
    for (int p = 1; p < nprocs + 1; ++p) nnzPerProcScan[p] += nnzPerProcScan[p - 1];   // :948-950

    sendCount = c.alltoall_one(recvCount);                               // :953 MPI_Alltoall
    for (int p = 0; p < nprocs; ++p) {                                   // :958-967
        if (recvCount[p] != 0) { recvProcRank.push_back(p); recvProcCount.push_back(recvCount[p]); }
        if (sendCount[p] != 0) { sendProcRank.push_back(p); sendProcCount.push_back(sendCount[p]); }
    }
    numRecvProc = (int)recvProcRank.size();
    numSendProc = (int)sendProcRank.size();
    vdispls.assign((size_t)nprocs, 0); rdispls.assign((size_t)nprocs, 0);
    for (int p = 1; p < nprocs; ++p) {                                   // :984-989
        vdispls[p] = vdispls[p - 1] + sendCount[p - 1];
        rdispls[p] = rdispls[p - 1] + recvCount[p - 1];
    }
    vIndexSize = vdispls[nprocs - 1] + sendCount[nprocs - 1];
    recvSize   = rdispls[nprocs - 1] + recvCount[nprocs - 1];
    vIndex = c.alltoallv_records(vElement_remote, recvCount);            // :1030 MPI_Alltoallv
    for (auto &x : vIndex) x -= split_col[rank];                         // :1044-1046
}

// The same layout straight from this rank's rows as CSR (global columns, ascending inside a row) -- what the
// row-distributed AMG setup holds.  build() wants the entries column-major and counting-sorts the local ones back to
// row-major: two passes over 16 B per entry that this form skips (the local part IS the CSR minus the remote entries;
// only the remote entries, a thin halo, are sorted by column).  Same arrays as build(), bit for bit.
void DistLayout::build_from_csr(Comm &c, const std::vector<nnz_t> &ptr, const std::vector<index_t> &col, const std::vector<value_t> &val,
                                const std::vector<index_t> &split_row, const std::vector<index_t> &split_col) {
    const int nprocs = c.nranks, rank = c.rank;
    *this = DistLayout();
    M = split_row[rank + 1] - split_row[rank];
    N_local = split_col[rank + 1] - split_col[rank];
    col_offset = split_col[rank];
    if ((index_t)ptr.size() != M + 1) throw std::runtime_error("build_from_csr: row count does not match the partition");
    const index_t lo = split_col[rank], hi = split_col[rank + 1];
    nnzPerRow_local.assign((size_t)M, 0);
    recvCount.assign((size_t)nprocs, 0);
    nnzPerProcScan.assign((size_t)nprocs + 1, 0);
    struct Rem { index_t col, row; value_t val; };
    std::vector<Rem> rem;
    auto lay_now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_begin = lay_now();
    // two passes over the rows on threads: count the local entries of every row and collect the (few) remote ones, then
    // copy the local entries to their places (a level of the configs[3] hierarchy holds up to 0.6 G entries per rank)
    std::vector<std::vector<Rem>> trem((size_t)setup_threads());
    parallel_chunks<index_t>(M, 4096, [&](int t, index_t r0, index_t r1) {
        auto &mine = trem[(size_t)t];
        for (index_t i = r0; i < r1; ++i) {
            index_t n = 0;
            for (nnz_t k = ptr[i]; k < ptr[i + 1]; ++k) {
                if (col[k] >= lo && col[k] < hi) ++n;
                else mine.push_back({col[k], i, val[k]});
            }
            nnzPerRow_local[(size_t)i] = n;
        }
    });
    for (auto &v : trem) rem.insert(rem.end(), v.begin(), v.end());      // chunks are ascending row ranges: row order is kept
    const double t_count = lay_now();
    std::vector<nnz_t> lptr((size_t)M + 1, 0);
    for (index_t i = 0; i < M; ++i) lptr[(size_t)i + 1] = lptr[(size_t)i] + nnzPerRow_local[(size_t)i];
    const nnz_t nloc = lptr[(size_t)M];
    nnz_l_local = nloc;
    col_local.resize((size_t)nloc); val_local.resize((size_t)nloc);
    parallel_chunks<index_t>(M, 4096, [&](int, index_t r0, index_t r1) {
        for (index_t i = r0; i < r1; ++i) {
            nnz_t q = lptr[(size_t)i];
            if (lptr[(size_t)i + 1] - q == ptr[i + 1] - ptr[i]) {           // a row without remote entries: one copy
                std::copy(col.begin() + ptr[i], col.begin() + ptr[i + 1], col_local.begin() + q);
                std::copy(val.begin() + ptr[i], val.begin() + ptr[i + 1], val_local.begin() + q);
                continue;
            }
            for (nnz_t k = ptr[i]; k < ptr[i + 1]; ++k)
                if (col[k] >= lo && col[k] < hi) { col_local[(size_t)q] = col[k]; val_local[(size_t)q] = val[k]; ++q; }
        }
    });
    const double t_copy = lay_now();
    std::sort(rem.begin(), rem.end(), [](const Rem &a, const Rem &b) { return a.col != b.col ? a.col < b.col : a.row < b.row; });
    size_t i = 0;
    while (i < rem.size()) {                                             // the remote half of :828-859
        const long procNum = owner_of(split_col, rem[i].col);
        const size_t tmp = i;
        while (i < rem.size() && rem[i].col < split_col[procNum + 1]) {
            vElement_remote.push_back(rem[i].col);
            ++recvCount[procNum];
            nnzPerCol_remote.push_back(0);
            do {
                col_remote.push_back((index_t)vElement_remote.size() - 1);
                col_remote2.push_back(rem[i].col);
                row_remote.push_back(rem[i].row);
                val_remote.push_back(rem[i].val);
                ++nnzPerCol_remote.back();
            } while (++i < rem.size() && rem[i].col == rem[i - 1].col);
        }
        nnzPerProcScan[procNum + 1] = (nnz_t)(i - tmp);
    }
    nnz_l_remote = (nnz_t)row_remote.size();
    col_remote_size = (index_t)vElement_remote.size();
    recvCount[rank] = 0;
    const double t_rem = lay_now();
    finish_plan(c, split_col);
    if (std::getenv("SAENA_LAYOUT_TIMING"))
        fprintf(stderr, "[layout] %d rows, %ld local + %ld remote entries: classify %.3f, copy %.3f, remote part %.3f, plan %.3f s\n", M, (long)nnz_l_local,
                (long)nnz_l_remote, t_count - t_begin, t_copy - t_count, t_rem - t_copy, lay_now() - t_rem);
}

void DistLayout::build_single_rank(index_t M_, index_t N_, const std::vector<nnz_t> &ptr, std::vector<index_t> &&col,
                                   std::vector<value_t> &&val) {
    *this = DistLayout();
    M = M_; N_local = N_; col_offset = 0;
    nnz_l_local = ptr[(size_t)M_];
    nnzPerRow_local.resize((size_t)M_);
    for (index_t i = 0; i < M_; ++i) nnzPerRow_local[i] = (index_t)(ptr[i + 1] - ptr[i]);
    col_local = std::move(col);
    val_local = std::move(val);
    recvCount.assign(1, 0); sendCount.assign(1, 0); vdispls.assign(1, 0); rdispls.assign(1, 0);
    nnzPerProcScan.assign(2, 0);
}

void saena_matrix::setup_from_csr(index_t n, const std::vector<nnz_t> &ptr, std::vector<index_t> &&col, std::vector<value_t> &&val) {
    if (comm->nranks != 1) throw std::runtime_error("setup_from_csr is a one-rank path");
    Mbig = M = n;
    split = {0, n};
    nnz_l = nnz_g = ptr[(size_t)n];
    remove_boundary = false;
    inv_diag.assign((size_t)n, 1.0);                                     // inverse_diag, saena_matrix_setup.cpp:1562-1600
    for (index_t i = 0; i < n; ++i)
        for (nnz_t k = ptr[i]; k < ptr[i + 1]; ++k)
            if (col[k] == i) {
                if (std::fabs(val[k]) < SAENA_ALMOST_ZERO)
                    throw std::runtime_error("there is a zero diagonal element at row index = " + std::to_string(i));
                inv_diag[i] = 1.0 / val[k];
            }
    L.build_single_rank(n, n, ptr, std::move(col), std::move(val));
    assembled = true;
}

// ---------------------------------------------------------------------------
void transpose_transfer(const transfer_matrix &P, transfer_matrix &R) {
    Comm &c = *P.comm;
    R.comm = P.comm;
    R.Mbig = P.Nbig; R.Nbig = P.Mbig;
    R.split_row = P.split_col; R.split_col = P.split_row;
    std::vector<cooEntry> t;
    t.reserve(P.entry.size());
    for (const auto &e : P.entry) t.emplace_back(e.col, e.row, e.val);
    R.entry = route(c, t, R.split_row, [](const cooEntry &e) { return e.row; });
    std::sort(R.entry.begin(), R.entry.end(), col_major);
    R.M = R.split_row[c.rank + 1] - R.split_row[c.rank];
    R.nnz_l = (nnz_t)R.entry.size();
    R.nnz_g = c.sum(R.nnz_l);
    R.build_layout();
}

// ---------------------------------------------------------------------------
// generators

static void z_slab(Comm &c, index_t mz, index_t *zs, index_t *zm) {   // aux_functions2.cpp:292-301
    const int nprocs = c.nranks, rank = c.rank;
    if (mz > nprocs) {
        *zm = mz / nprocs;
        *zs = rank * *zm;
        if (rank == nprocs - 1) *zm = mz - (nprocs - 1) * *zm;
    } else {
        *zm = 1;
        *zs = rank;
    }
}

int laplacian3D(saena_matrix *A, index_t mx, index_t my, index_t mz) {
    Comm &c = *A->comm;
    if (c.rank >= mz) return 0;
    const value_t Hx = 1.0 / (mx - 1), Hy = 1.0 / (my - 1), Hz = 1.0 / (mz - 1);
    const value_t HyHzdHx = 1.0 / (Hx * Hx), HxHzdHy = 1.0 / (Hy * Hy), HxHydHz = 1.0 / (Hz * Hz);
    index_t zs, zm;
    z_slab(c, mz, &zs, &zm);
    const index_t XMAX = mx - 1, YMAX = my - 1, ZMAX = mz - 1;
    for (index_t k = zs; k < zs + zm; ++k)
        for (index_t j = 0; j < my; ++j)
            for (index_t i = 0; i < mx; ++i) {
                const index_t node = mx * my * k + mx * j + i;
                if (i == 0 || j == 0 || k == 0 || i == XMAX || j == YMAX || k == ZMAX) {
                    A->set(node, node, 1.0);
                } else {
                    if (k - 1 != 0) A->set(node, node - mx * my, -HxHydHz);
                    if (j - 1 != 0) A->set(node, node - mx, -HxHzdHy);
                    if (i - 1 != 0) A->set(node, node - 1, -HyHzdHx);
                    A->set(node, node, 2.0 * (HxHydHz + HxHzdHy + HyHzdHx));
                    if (i + 1 != XMAX) A->set(node, node + 1, -HyHzdHx);
                    if (j + 1 != YMAX) A->set(node, node + mx, -HxHzdHy);
                    if (k + 1 != ZMAX) A->set(node, node + mx * my, -HxHydHz);
                }
            }
    return 0;
}

std::vector<value_t> laplacian3D_set_rhs(Comm &c, index_t mx, index_t my, index_t mz, index_t *lo) {
    std::vector<value_t> rhs;
    *lo = 0;
    if (c.rank >= mz) return rhs;
    const value_t Hx = 1.0 / (mx - 1), Hy = 1.0 / (my - 1), Hz = 1.0 / (mz - 1);
    index_t zs, zm;
    z_slab(c, mz, &zs, &zm);
    const double PI = 3.1415926535897932384626433832795029;      // data_struct.h:44
    const double TWOPI = 2 * PI, TWOELVEPISQ = 12 * PI * PI;
    rhs.reserve((size_t)mx * my * zm);
    for (index_t k = zs; k < zs + zm; ++k)
        for (index_t j = 0; j < my; ++j)
            for (index_t i = 0; i < mx; ++i)
                rhs.push_back(TWOELVEPISQ * std::sin(TWOPI * i * Hx) * std::sin(TWOPI * j * Hy) * std::sin(TWOPI * k * Hz));
    *lo = mx * my * zs;
    return rhs;
}

int band_matrix(saena_matrix *A, index_t M, unsigned int bandwidth) {   // aux_functions2.cpp:1296-1330
    Comm &c = *A->comm;
    const index_t Mbig = M * c.nranks;
    if ((index_t)bandwidth >= Mbig) throw std::runtime_error("Error: bandwidth is greater than the size of the matrix");
    for (index_t i = c.rank * M; i < (c.rank + 1) * M; ++i) {
        index_t d = 0;
        for (index_t j = i; j <= i + (index_t)bandwidth; ++j) {
            const value_t val = 1.0 / (i + j + 1);
            if (i == j) {
                A->set(i, j, val);
            } else {
                if (j < Mbig) A->set(i, j, val);
                if (j >= 2 * d) A->set(i, j - 2 * d, 1.0 / (i + j - 2 * d + 1));
            }
            d++;
        }
    }
    return 0;
}

} // namespace saena_host

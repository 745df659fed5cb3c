// sgpu_spgemm.hip -- C = A B on the MI355X for the Galerkin products of the AMG setup (Ac = (R A) P,
// saena_object::triple_mat_mult / matmat, src/saena_object_setup2.cpp:361-849, src/saena_object_setup_matmat.cpp:1164-1487).
//
// The setup is host code in the reference and stays host code here; these two products are 70-80 % of it
// (measured: 256^3, 16 host threads: 55 of 76 s), so they run on the GPU when one is there (saena_host::g_spgemm_hook,
// installed by sgpu_init; SAENA_HOST_SPGEMM=1 keeps the host kernel).  The result is the host kernel's BIT FOR BIT
// (tests/test_gpu_spgemm.py), which is what keeps the hierarchy equal to the reference's printed sizes: every output
// entry c_ij adds its products a_ik b_kj in the order the host adds them -- k in the order of row i of A -- because a
// row is owned by one wave / workgroup that walks the entries of A's row one after the other and spreads only the
// entries of B's row k, whose columns are distinct, over its lanes.  No floating-point atomics whose order could vary.
//
// Three accumulators by the row's upper bound ub = min(products, columns of B):
//   light  (ub <= 256)   one wave per row, 512-slot hash table in LDS (4 rows per workgroup);
//   medium (ub <= 2048)  one workgroup per row, 4096-slot hash table in LDS;
//   heavy                first the same LDS table, abandoned when more than 3072 distinct columns turn up (a row of R A
//                        on the second level has 13 700 products and 760 entries); the rows that overflow go to
//                        persistent workgroups with a dense accumulator: in LDS, one window of 20 224 columns after the
//                        other (round 3, k_spgemm_lds: B of up to 16 windows = 323 K columns, which covers every coarse
//                        level of the 256^3 / 512^3 hierarchies), else of their own in HBM (12 B x columns per workgroup).
// Each row leaves its touched (column, value) pairs unsorted in a scratch segment; dropped entries (|v| <= 1e-14 off
// the diagonal, saena_object_setup_matmat.cpp:2423,2442) get the key INT_MAX; one segmented radix sort per chunk of rows
// orders every segment by column, and the kept prefix of each segment is copied out.
#include "../../include/saena_gpu.h"
#include "host/par.h"

#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <chrono>
#include <climits>
#include <cstdio>
#include <cstdlib>
#include <vector>

namespace saena_host {
typedef int (*spgemm_hook_fn)(int a_rows, int b_rows, int b_cols, const long *a_ptr, const int *a_col, const double *a_val,
                              const long *b_ptr, const int *b_col, const double *b_val, long b_split, const int *b_col1, const double *b_val1,
                              int row_offset, std::vector<long> &c_ptr, std::vector<int> &c_col, std::vector<double> &c_val);
extern spgemm_hook_fn g_spgemm_hook;
}

namespace {

constexpr double ALMOST_ZERO = 1e-14;          // data_struct.h:42
constexpr int LIGHT_SLOTS = 512, MEDIUM_SLOTS = 4096;
constexpr int LIGHT_UB = 256, MEDIUM_UB = 2048;

struct Mats {
    const long long *a_ptr; const int *a_col; const double *a_val;
    const long long *b_ptr; const int *b_col; const double *b_val;
};

__device__ __forceinline__ unsigned hash_col(int j) { return (unsigned)j * 2654435761u; }

// ---- light rows: one wave per row ----
__global__ __launch_bounds__(256) void k_spgemm_light(Mats m, const int *__restrict__ rows, int nrows, int r0, const int *__restrict__ ubptr,
                                                      int *__restrict__ tcol, double *__restrict__ tval, int *__restrict__ n_touched,
                                                      int *__restrict__ n_kept, int row_offset) {
    __shared__ int keys[4][LIGHT_SLOTS];
    __shared__ double vals[4][LIGHT_SLOTS];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int idx = blockIdx.x * 4 + wave;
    if (idx >= nrows) return;                              // (no workgroup barrier in this kernel)
    const int i = rows[idx], loc = i - r0;
    int *K = keys[wave];
    double *V = vals[wave];
    for (int s = lane; s < LIGHT_SLOTS; s += 64) K[s] = -1;
    __builtin_amdgcn_wave_barrier();
    for (long long ka = m.a_ptr[i]; ka < m.a_ptr[i + 1]; ++ka) {          // one entry of A's row after the other: the host's order
        const int k = m.a_col[ka];
        const double a = m.a_val[ka];
        const long long b1 = m.b_ptr[k + 1];
        for (long long kb = m.b_ptr[k] + lane; kb < b1; kb += 64) {       // the columns of one row of B are distinct: no two lanes meet
            const int j = m.b_col[kb];
            const double v = a * m.b_val[kb];
            unsigned h = hash_col(j) & (LIGHT_SLOTS - 1);
            while (true) {
                const int prev = atomicCAS(&K[h], -1, j);
                if (prev == -1) { V[h] = v; break; }
                if (prev == j) { V[h] += v; break; }
                h = (h + 1) & (LIGHT_SLOTS - 1);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    const long long u0 = ubptr[loc];
    int cnt = 0, dropped = 0;
    for (int base = 0; base < LIGHT_SLOTS; base += 64) {
        const int key = K[base + lane];
        const bool occ = key >= 0;
        const double v = occ ? V[base + lane] : 0.0;
        const bool drop = occ && !(fabs(v) > ALMOST_ZERO || i + row_offset == key);
        const unsigned long long mask = __ballot(occ);
        const int pos = cnt + __popcll(mask & ((1ull << lane) - 1ull));
        if (occ) { tcol[u0 + pos] = drop ? INT_MAX : key; tval[u0 + pos] = v; }
        cnt += __popcll(mask);
        dropped += __popcll(__ballot(drop));
    }
    if (lane == 0) { n_touched[loc] = cnt; n_kept[loc] = cnt - dropped; }
}

// ---- medium rows: one workgroup per row ----
// TRY = true serves rows whose upper bound exceeds the table but whose DISTINCT output columns usually do not (a row of
// R A on the second level: 13 700 products, 760 entries): the row is abandoned -- n_touched = -1 -- as soon as more than
// MEDIUM_FILL keys are in the table, and the host hands it to the dense-accumulator kernel.
constexpr int MEDIUM_FILL = 3072;
template <bool TRY>
__global__ __launch_bounds__(256) void k_spgemm_medium(Mats m, const int *__restrict__ rows, int r0, const int *__restrict__ ubptr,
                                                       int *__restrict__ tcol, double *__restrict__ tval, int *__restrict__ n_touched,
                                                       int *__restrict__ n_kept, int row_offset) {
    __shared__ int K[MEDIUM_SLOTS];
    __shared__ double V[MEDIUM_SLOTS];
    __shared__ int counter[4];                              // [0] written out, [1] dropped, [2] keys in the table, [3] overflow
    const int tid = threadIdx.x;
    const int i = rows[blockIdx.x], loc = i - r0;
    for (int s = tid; s < MEDIUM_SLOTS; s += 256) K[s] = -1;
    if (tid < 4) counter[tid] = 0;
    __syncthreads();
    for (long long ka = m.a_ptr[i]; ka < m.a_ptr[i + 1]; ++ka) {
        const int k = m.a_col[ka];
        const double a = m.a_val[ka];
        const long long b1 = m.b_ptr[k + 1];
        for (long long kb = m.b_ptr[k] + tid; kb < b1; kb += 256) {
            const int j = m.b_col[kb];
            const double v = a * m.b_val[kb];
            unsigned h = hash_col(j) & (MEDIUM_SLOTS - 1);
            int steps = 0;
            while (true) {
                const int prev = atomicCAS(&K[h], -1, j);
                if (prev == -1) { V[h] = v; if (TRY) atomicAdd(&counter[2], 1); break; }
                if (prev == j) { V[h] += v; break; }
                h = (h + 1) & (MEDIUM_SLOTS - 1);
                if (TRY && ++steps >= MEDIUM_SLOTS) { counter[3] = 1; break; }      // table full: give the row up
            }
        }
        __syncthreads();                                    // entry ka+1 of A's row adds after entry ka everywhere
        if (TRY) {
            // the give-up decision is read between two barriers, so no wave is already bumping the counters of entry ka+1
            // when another reads them: every thread sees the same value and the workgroup leaves together
            const bool over = counter[3] != 0 || counter[2] > MEDIUM_FILL;
            __syncthreads();
            if (over) { if (tid == 0) { n_touched[loc] = -1; n_kept[loc] = 0; } return; }
        }
    }
    const long long u0 = ubptr[loc];
    for (int s = tid; s < MEDIUM_SLOTS; s += 256) {
        const int key = K[s];
        if (key >= 0) {
            const double v = V[s];
            const bool drop = !(fabs(v) > ALMOST_ZERO || i + row_offset == key);
            const int pos = atomicAdd(&counter[0], 1);
            tcol[u0 + pos] = drop ? INT_MAX : key;
            tval[u0 + pos] = v;
            if (drop) atomicAdd(&counter[1], 1);
        }
    }
    __syncthreads();
    if (tid == 0) { n_touched[loc] = counter[0]; n_kept[loc] = counter[0] - counter[1]; }
}

// ---- heavy rows: persistent workgroups, each with a dense accumulator (acc, mark) of b_cols entries in HBM ----
// acc / mark are updated with agent-scope atomics: they act at the L2, so what a thread wrote before a barrier is what
// any thread of the workgroup reads after it (no reliance on the vector L1); per entry the updates are ordered by the
// barriers between the entries of A's row, so the sums are the host's.
__global__ __launch_bounds__(256) void k_spgemm_heavy(Mats m, const int *__restrict__ rows, int nrows, int r0, const int *__restrict__ ubptr,
                                                      int *__restrict__ tcol, double *__restrict__ tval, int *__restrict__ n_touched,
                                                      int *__restrict__ n_kept, int row_offset, double *__restrict__ acc_all,
                                                      int *__restrict__ mark_all, long long b_cols) {
    __shared__ int counter[2];
    const int tid = threadIdx.x;
    double *acc = acc_all + (long long)blockIdx.x * b_cols;
    int *mark = mark_all + (long long)blockIdx.x * b_cols;
    for (int idx = blockIdx.x; idx < nrows; idx += gridDim.x) {
        const int i = rows[idx], loc = i - r0;
        const long long u0 = ubptr[loc];
        if (tid < 2) counter[tid] = 0;
        __syncthreads();
        for (long long ka = m.a_ptr[i]; ka < m.a_ptr[i + 1]; ++ka) {
            const int k = m.a_col[ka];
            const double a = m.a_val[ka];
            const long long b1 = m.b_ptr[k + 1];
            for (long long kb = m.b_ptr[k] + tid; kb < b1; kb += 256) {
                const int j = m.b_col[kb];
                const double v = a * m.b_val[kb];
                if (__hip_atomic_exchange(&mark[j], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0)
                    tcol[u0 + atomicAdd(&counter[0], 1)] = j;                  // first touch: remember the column
                (void)__hip_atomic_fetch_add(&acc[j], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            __syncthreads();
        }
        const int n = counter[0];
        __syncthreads();
        for (int t = tid; t < n; t += 256) {
            const int j = tcol[u0 + t];
            const double v = __hip_atomic_exchange(&acc[j], 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            (void)__hip_atomic_exchange(&mark[j], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            tval[u0 + t] = v;
            if (!(fabs(v) > ALMOST_ZERO || i + row_offset == j)) { tcol[u0 + t] = INT_MAX; atomicAdd(&counter[1], 1); }
        }
        __syncthreads();
        if (tid == 0) { n_touched[loc] = n; n_kept[loc] = n - counter[1]; }
        __syncthreads();
    }
}

// ---- heavy rows, accumulator in LDS (round 3) ----
// The HBM accumulator above pays an L2 round trip per product (agent-scope atomics: 20 G products/s on the chip, 3-6 s per
// product on the densest levels of a 16 M-row hierarchy).  Here a 1024-thread workgroup per CU keeps the accumulator of ONE
// window of LDS_COLS columns in its LDS (161 792 B, as k_csr_xlds does with x) and walks the row of A once per window: for
// entry ka it takes the piece of B's row k that falls in the window -- [bw[k][w], bw[k][w+1]), a table made once per product
// by k_window_starts -- and adds a * b into the slots of its columns; the columns of one row of B are distinct, so no two
// lanes meet, and a barrier separates ka from ka + 1: every output entry adds its products in the order of A's row, the
// host's order, bit for bit.  A slot that nothing touched holds a sentinel (a NaN bit pattern no sum produces from finite
// data) instead of a separate mark: the window is scanned once at the end, touched slots emitted (dropped entries get the
// key INT_MAX like everywhere) and reset.  A (ka, window) pair without entries costs no barrier.
constexpr int LDS_COLS = 20224;
constexpr int LDS_MAXW = 16;
constexpr int LDS_BLOCK = 1024;
constexpr unsigned long long LDS_SENTINEL = 0x7FF8A5A5C3C35A5AULL;
__global__ __launch_bounds__(256) void k_window_starts(const long long *__restrict__ b_ptr, const int *__restrict__ b_col, int b_rows, int W,
                                                       long long *__restrict__ bw) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= (long long)b_rows * (W + 1)) return;
    const int k = (int)(t / (W + 1)), w = (int)(t % (W + 1));
    long long lo = b_ptr[k], hi = b_ptr[k + 1];
    const long long bound = (long long)w * LDS_COLS;       // first entry of row k with column >= bound (columns ascend in a row)
    while (lo < hi) { const long long mid = (lo + hi) >> 1; if (b_col[mid] < bound) lo = mid + 1; else hi = mid; }
    bw[t] = lo;
}
__global__ __launch_bounds__(LDS_BLOCK) void k_spgemm_lds(Mats m, const int *__restrict__ rows, int nrows, int r0, const int *__restrict__ ubptr,
                                                          int *__restrict__ tcol, double *__restrict__ tval, int *__restrict__ n_touched,
                                                          int *__restrict__ n_kept, int row_offset, const long long *__restrict__ bw, int W, int b_cols) {
    __shared__ double acc[LDS_COLS];
    __shared__ int counter[2];
    const int tid = threadIdx.x;
    const double sentinel = __longlong_as_double((long long)LDS_SENTINEL);
    for (int s = tid; s < LDS_COLS; s += LDS_BLOCK) acc[s] = sentinel;
    for (int idx = blockIdx.x; idx < nrows; idx += gridDim.x) {
        const int i = rows[idx], loc = i - r0;
        const long long u0 = ubptr[loc];
        if (tid < 2) counter[tid] = 0;
        __syncthreads();
        const long long a0 = m.a_ptr[i], a1 = m.a_ptr[i + 1];
        for (int w = 0; w < W; ++w) {
            const int base = w * LDS_COLS;
            bool any = false;
            for (long long ka = a0; ka < a1; ++ka) {
                const int k = m.a_col[ka];
                const long long s0 = bw[(long long)k * (W + 1) + w], s1 = bw[(long long)k * (W + 1) + w + 1];
                if (s0 == s1) continue;                    // uniform over the workgroup: nothing written, no barrier needed
                const double a = m.a_val[ka];
                for (long long kb = s0 + tid; kb < s1; kb += LDS_BLOCK) {
                    const int j = m.b_col[kb] - base;
                    const double v = a * m.b_val[kb];
                    const double old = acc[j];
                    acc[j] = (__double_as_longlong(old) == (long long)LDS_SENTINEL ? 0.0 : old) + v;       // the host: acc = 0.0, then += a b
                }
                any = true;
                __syncthreads();
            }
            if (!any) continue;
            const int width = min(LDS_COLS, b_cols - base);
            for (int sl = tid; sl < width; sl += LDS_BLOCK) {
                const double v = acc[sl];
                if (__double_as_longlong(v) == (long long)LDS_SENTINEL) continue;
                acc[sl] = sentinel;
                const int j = base + sl;
                const bool drop = !(fabs(v) > ALMOST_ZERO || i + row_offset == j);
                const int pos = atomicAdd(&counter[0], 1);
                tcol[u0 + pos] = drop ? INT_MAX : j;
                tval[u0 + pos] = v;
                if (drop) atomicAdd(&counter[1], 1);
            }
            __syncthreads();
        }
        if (tid == 0) { n_touched[loc] = counter[0]; n_kept[loc] = counter[0] - counter[1]; }
        __syncthreads();
    }
}

// segment ends for the sort: begin + touched
__global__ void k_seg_ends(const int *__restrict__ ubptr, const int *__restrict__ n_touched, int *__restrict__ ends, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) ends[i] = ubptr[i] + n_touched[i];
}
// the kept prefix of every sorted segment -> the output arrays of the chunk
__global__ __launch_bounds__(256) void k_copy_out(const int *__restrict__ ubptr, const int *__restrict__ outptr, const int *__restrict__ scol,
                                                  const double *__restrict__ sval, int *__restrict__ ocol, double *__restrict__ oval, int n) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + wave;
    if (i >= n) return;
    const int u0 = ubptr[i], o0 = outptr[i], cnt = outptr[i + 1] - o0;
    for (int t = lane; t < cnt; t += 64) { ocol[o0 + t] = scol[u0 + t]; oval[o0 + t] = sval[u0 + t]; }
}

struct Dev {                                    // RAII for the call's device buffers
    std::vector<void *> p;
    ~Dev() { for (void *q : p) hipFree(q); }
    template <class T> T *alloc(size_t n) {
        void *q = nullptr;
        if (hipMalloc(&q, std::max<size_t>(n, 1) * sizeof(T)) != hipSuccess) return nullptr;
        p.push_back(q);
        return static_cast<T *>(q);
    }
};
extern "C" int sgpu_context_device();      // sgpu_runtime.hip: the device of this process's context
#define SP_CHK(x) do { if ((x) != hipSuccess) { (void)hipGetLastError(); return 1; } } while (0)

int gpu_spgemm(int a_rows, int b_rows, int b_cols, const long *a_ptr, const int *a_col, const double *a_val,
               const long *b_ptr, const int *b_col, const double *b_val, long b_split, const int *b_col1, const double *b_val1,
               int row_offset, std::vector<long> &c_ptr, std::vector<int> &c_col, std::vector<double> &c_val) {
    static_assert(sizeof(long) == sizeof(long long), "nnz_t is 64-bit");
    const long a_nnz = a_ptr[a_rows], b_nnz = b_ptr[b_rows];
    // products and upper bounds per row (host, O(nnz of A))
    const bool timing = std::getenv("SAENA_SETUP_TIMING") != nullptr;
    double t_up = 0, t_kern = 0, t_sort = 0, t_down = 0, t_host = 0;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto secs = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double>(b - a).count(); };
    auto Tb = now();
    std::vector<long long> ub((size_t)a_rows), prod((size_t)a_rows);
    saena_host::parallel_chunks<int>(a_rows, 4096, [&](int, int i0, int i1) {      // (a random read of b_ptr per entry of A: seconds on one thread at 5e8 entries)
        for (int i = i0; i < i1; ++i) {
            long long w = 0;
            for (long ka = a_ptr[i]; ka < a_ptr[i + 1]; ++ka) w += b_ptr[a_col[ka] + 1] - b_ptr[a_col[ka]];
            prod[(size_t)i] = w;
            ub[(size_t)i] = std::min<long long>(w, b_cols);
        }
    });
    long long products_total = 0;
    for (int i = 0; i < a_rows; ++i) products_total += prod[(size_t)i];
    const double t_bounds = secs(Tb, now());
    auto T0 = now();
    SP_CHK(hipSetDevice(sgpu_context_device()));            // the setup may be driven from a thread that never selected this context's device
    Dev D;
    Mats m;
    long long *dap = D.alloc<long long>((size_t)a_rows + 1), *dbp = D.alloc<long long>((size_t)b_rows + 1);
    int *dac = D.alloc<int>((size_t)a_nnz), *dbc = D.alloc<int>((size_t)b_nnz);
    double *dav = D.alloc<double>((size_t)a_nnz), *dbv = D.alloc<double>((size_t)b_nnz);
    if (!dap || !dbp || !dac || !dbc || !dav || !dbv) return 1;
    SP_CHK(hipMemcpy(dap, a_ptr, ((size_t)a_rows + 1) * 8, hipMemcpyHostToDevice));
    SP_CHK(hipMemcpy(dbp, b_ptr, ((size_t)b_rows + 1) * 8, hipMemcpyHostToDevice));
    SP_CHK(hipMemcpy(dac, a_col, (size_t)a_nnz * 4, hipMemcpyHostToDevice));
    const long b_n0 = b_col1 ? b_split : b_nnz;                  // B in one piece, or its own rows followed by the fetched halo rows
    SP_CHK(hipMemcpy(dbc, b_col, (size_t)b_n0 * 4, hipMemcpyHostToDevice));
    SP_CHK(hipMemcpy(dav, a_val, (size_t)a_nnz * 8, hipMemcpyHostToDevice));
    SP_CHK(hipMemcpy(dbv, b_val, (size_t)b_n0 * 8, hipMemcpyHostToDevice));
    if (b_nnz > b_n0) {
        SP_CHK(hipMemcpy(dbc + b_n0, b_col1, (size_t)(b_nnz - b_n0) * 4, hipMemcpyHostToDevice));
        SP_CHK(hipMemcpy(dbv + b_n0, b_val1, (size_t)(b_nnz - b_n0) * 8, hipMemcpyHostToDevice));
    }
    m.a_ptr = dap; m.a_col = dac; m.a_val = dav; m.b_ptr = dbp; m.b_col = dbc; m.b_val = dbv;
    t_up = secs(T0, now());

    // chunks of consecutive rows whose scratch segments fit CH entries (2 x 12 B each: unsorted + sorted)
    const long long CH = 384LL << 20;                                           // < 2^31: in-chunk offsets are ints
    const int MAXROWS_CHUNK = 16 << 20;
    long long max_ub_sum = 0;
    int max_rows = 0;
    std::vector<int> chunk_start(1, 0);
    {
        long long s = 0;
        int start = 0;
        for (int i = 0; i < a_rows; ++i) {
            if (ub[(size_t)i] > CH) return 1;                                  // a single row beyond the scratch: leave it to the host
            if (i > start && (s + ub[(size_t)i] > CH || i - start >= MAXROWS_CHUNK)) {
                max_ub_sum = std::max(max_ub_sum, s); max_rows = std::max(max_rows, i - start);
                chunk_start.push_back(i); start = i; s = 0;
            }
            s += ub[(size_t)i];
        }
        max_ub_sum = std::max(max_ub_sum, s); max_rows = std::max(max_rows, a_rows - start);
        chunk_start.push_back(a_rows);
    }
    int *tcol = D.alloc<int>((size_t)max_ub_sum), *scol = D.alloc<int>((size_t)max_ub_sum);
    double *tval = D.alloc<double>((size_t)max_ub_sum), *sval = D.alloc<double>((size_t)max_ub_sum);
    int *d_ubptr = D.alloc<int>((size_t)max_rows + 1), *d_ends = D.alloc<int>((size_t)max_rows + 1), *d_outptr = D.alloc<int>((size_t)max_rows + 1);
    int *d_touched = D.alloc<int>((size_t)max_rows), *d_kept = D.alloc<int>((size_t)max_rows), *d_rows = D.alloc<int>((size_t)max_rows);
    if (!tcol || !scol || !tval || !sval || !d_ubptr || !d_ends || !d_outptr || !d_touched || !d_kept || !d_rows) return 1;
    // dense accumulators of the heavy path (allocated at the first chunk that has heavy rows)
    double *acc = nullptr;
    int *mark = nullptr;
    int heavy_grid = 0;
    // heavy rows accumulate in LDS, a window of LDS_COLS columns at a time, when B has at most LDS_MAXW windows of columns
    // (SAENA_SPGEMM_NO_LDS=1: the accumulator in HBM for every heavy row, as before round 3)
    const int n_windows = (b_cols + LDS_COLS - 1) / LDS_COLS;
    const bool use_lds = n_windows <= LDS_MAXW && !std::getenv("SAENA_SPGEMM_NO_LDS");
    long long *d_bw = nullptr;
    int lds_grid = 256;
    { int dev = 0, ncu = 0; if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && ncu > 0) lds_grid = ncu; }
    struct Temp { void *p = nullptr; ~Temp() { if (p) hipFree(p); } } temp;      // the sort's scratch: freed on every return path
    void *&d_temp = temp.p;
    size_t temp_bytes = 0;

    c_ptr.assign(1, 0);
    c_col.clear(); c_val.clear();
    std::vector<int> h_ubptr, h_outptr;
    std::vector<int> h_kept, light, medium, heavy, h_rows;
    for (size_t c = 0; c + 1 < chunk_start.size(); ++c) {
        const int r0 = chunk_start[c], r1 = chunk_start[c + 1], n = r1 - r0;
        auto Tc = now();
        h_ubptr.assign((size_t)n + 1, 0);
        light.clear(); medium.clear(); heavy.clear();
        for (int i = r0; i < r1; ++i) {
            h_ubptr[(size_t)(i - r0) + 1] = h_ubptr[(size_t)(i - r0)] + (int)ub[(size_t)i];
            if (ub[(size_t)i] == 0) continue;
            (ub[(size_t)i] <= LIGHT_UB ? light : ub[(size_t)i] <= MEDIUM_UB ? medium : heavy).push_back(i);
        }
        t_host += secs(Tc, now()); Tc = now();
        SP_CHK(hipMemcpy(d_ubptr, h_ubptr.data(), ((size_t)n + 1) * 4, hipMemcpyHostToDevice));
        SP_CHK(hipMemset(d_touched, 0, (size_t)n * 4));
        SP_CHK(hipMemset(d_kept, 0, (size_t)n * 4));
        h_rows = light; h_rows.insert(h_rows.end(), medium.begin(), medium.end()); h_rows.insert(h_rows.end(), heavy.begin(), heavy.end());
        if (!h_rows.empty()) SP_CHK(hipMemcpy(d_rows, h_rows.data(), h_rows.size() * 4, hipMemcpyHostToDevice));
        if (!light.empty())
            hipLaunchKernelGGL(k_spgemm_light, dim3(((unsigned)light.size() + 3) / 4), dim3(256), 0, 0, m, (const int *)d_rows, (int)light.size(), r0,
                               (const int *)d_ubptr, tcol, tval, d_touched, d_kept, row_offset);
        if (!medium.empty())
            hipLaunchKernelGGL(k_spgemm_medium<false>, dim3((unsigned)medium.size()), dim3(256), 0, 0, m, (const int *)(d_rows + light.size()), r0,
                               (const int *)d_ubptr, tcol, tval, d_touched, d_kept, row_offset);
        if (!heavy.empty()) {
            // rows with many products but (usually) few distinct columns: try the LDS table first, keep the dense
            // accumulator for the rows that overflow it
            const int *d_heavy = d_rows + light.size() + medium.size();
            hipLaunchKernelGGL(k_spgemm_medium<true>, dim3((unsigned)heavy.size()), dim3(256), 0, 0, m, d_heavy, r0,
                               (const int *)d_ubptr, tcol, tval, d_touched, d_kept, row_offset);
            SP_CHK(hipGetLastError());
            h_kept.resize((size_t)n);                                            // (scratch: the touched counts)
            SP_CHK(hipMemcpy(h_kept.data(), d_touched, (size_t)n * 4, hipMemcpyDeviceToHost));
            std::vector<int> over;
            for (int i : heavy) if (h_kept[(size_t)(i - r0)] < 0) over.push_back(i);
            heavy.swap(over);
            if (!heavy.empty()) SP_CHK(hipMemcpy(d_rows + light.size() + medium.size(), heavy.data(), heavy.size() * 4, hipMemcpyHostToDevice));
        }
        if (!heavy.empty() && use_lds) {
            if (!d_bw) {                                                       // where each window starts in every row of B: once per product
                d_bw = D.alloc<long long>((size_t)b_rows * (size_t)(n_windows + 1));
                if (!d_bw) return 1;
                const long long cells = (long long)b_rows * (n_windows + 1);
                hipLaunchKernelGGL(k_window_starts, dim3((unsigned)((cells + 255) / 256)), dim3(256), 0, 0, (const long long *)dbp, (const int *)dbc, b_rows, n_windows, d_bw);
            }
            hipLaunchKernelGGL(k_spgemm_lds, dim3((unsigned)std::min<size_t>((size_t)lds_grid, heavy.size())), dim3(LDS_BLOCK), 0, 0, m,
                               (const int *)(d_rows + light.size() + medium.size()), (int)heavy.size(), r0, (const int *)d_ubptr, tcol, tval,
                               d_touched, d_kept, row_offset, (const long long *)d_bw, n_windows, b_cols);
            heavy.clear();
        }
        if (!heavy.empty()) {
            if (!acc) {
                const long long per = 12LL * b_cols;                            // 8 B sum + 4 B mark per column and workgroup
                heavy_grid = (int)std::max<long long>(32, std::min<long long>(512, (24LL << 30) / std::max<long long>(per, 1)));
                acc = D.alloc<double>((size_t)heavy_grid * (size_t)b_cols);
                mark = D.alloc<int>((size_t)heavy_grid * (size_t)b_cols);
                if (!acc || !mark) return 1;
                SP_CHK(hipMemset(acc, 0, (size_t)heavy_grid * (size_t)b_cols * 8));
                SP_CHK(hipMemset(mark, 0, (size_t)heavy_grid * (size_t)b_cols * 4));
            }
            hipLaunchKernelGGL(k_spgemm_heavy, dim3((unsigned)std::min<size_t>((size_t)heavy_grid, heavy.size())), dim3(256), 0, 0, m,
                               (const int *)(d_rows + light.size() + medium.size()), (int)heavy.size(), r0, (const int *)d_ubptr, tcol, tval,
                               d_touched, d_kept, row_offset, acc, mark, (long long)b_cols);
        }
        SP_CHK(hipGetLastError());
        if (timing) { SP_CHK(hipDeviceSynchronize()); t_kern += secs(Tc, now()); Tc = now(); }
        hipLaunchKernelGGL(k_seg_ends, dim3((n + 255) / 256), dim3(256), 0, 0, (const int *)d_ubptr, (const int *)d_touched, d_ends, n);
        // every segment by column (dropped entries carry INT_MAX and end up behind the kept ones)
        const long long items = h_ubptr[(size_t)n];
        if (items > INT_MAX) return 1;
        size_t need = 0;
        SP_CHK(hipcub::DeviceSegmentedRadixSort::SortPairs(nullptr, need, (const int *)tcol, scol, (const double *)tval, sval, (int)items, n,
                                                           (const int *)d_ubptr, (const int *)d_ends, 0, 32, 0));
        if (need > temp_bytes) {
            if (d_temp) { hipFree(d_temp); d_temp = nullptr; }
            SP_CHK(hipMalloc(&d_temp, need));
            temp_bytes = need;
        }
        hipError_t se = hipcub::DeviceSegmentedRadixSort::SortPairs(d_temp, need, (const int *)tcol, scol, (const double *)tval, sval, (int)items, n,
                                                                    (const int *)d_ubptr, (const int *)d_ends, 0, 32, 0);
        if (se != hipSuccess) return 1;
        h_kept.resize((size_t)n);
        SP_CHK(hipMemcpy(h_kept.data(), d_kept, (size_t)n * 4, hipMemcpyDeviceToHost));
        if (timing) { t_sort += secs(Tc, now()); Tc = now(); }
        h_outptr.assign((size_t)n + 1, 0);
        for (int i = 0; i < n; ++i) h_outptr[(size_t)i + 1] = h_outptr[(size_t)i] + h_kept[(size_t)i];
        const long long out_n = h_outptr[(size_t)n];
        SP_CHK(hipMemcpy(d_outptr, h_outptr.data(), ((size_t)n + 1) * 4, hipMemcpyHostToDevice));
        // compact into the (now free) unsorted buffers and bring the chunk home
        hipLaunchKernelGGL(k_copy_out, dim3((n + 3) / 4), dim3(256), 0, 0, (const int *)d_ubptr, (const int *)d_outptr, (const int *)scol,
                           (const double *)sval, tcol, tval, n);
        SP_CHK(hipGetLastError());
        const size_t base = c_col.size();
        if (c == 0 && chunk_start.size() > 2) {                                 // room for the whole result at the first chunk's density (+ 15 %): a vector
            long long p0 = 0;                                                   // that grows chunk by chunk is re-allocated and copied again and again
            for (int i = r0; i < r1; ++i) p0 += prod[(size_t)i];
            if (p0 > 0) {
                const double est = 1.15 * (double)out_n / (double)p0 * (double)products_total;
                if (est < 3.0e9) { saena_host::LazyBigalloc lazy; c_col.reserve((size_t)est); c_val.reserve((size_t)est); }   // an estimate: advised, not touched
            }
        }
        c_col.resize(base + (size_t)out_n);
        c_val.resize(base + (size_t)out_n);
        if (out_n) {
            SP_CHK(hipMemcpy(c_col.data() + base, tcol, (size_t)out_n * 4, hipMemcpyDeviceToHost));
            SP_CHK(hipMemcpy(c_val.data() + base, tval, (size_t)out_n * 8, hipMemcpyDeviceToHost));
        }
        for (int i = 0; i < n; ++i) c_ptr.push_back((long)(base + (size_t)h_outptr[(size_t)i + 1]));
        t_down += secs(Tc, now());
    }
    if (std::getenv("SAENA_SETUP_TIMING"))
        fprintf(stderr, "[spgemm gpu] %d x %d, %lld products -> %zu entries, %zu chunk(s): bounds %.2f, upload %.2f, row kernels %.2f, sort %.2f, copy-out+download %.2f, host %.2f, total %.2f s\n",
                a_rows, b_cols, products_total, c_col.size(), chunk_start.size() - 1, t_bounds, t_up, t_kern, t_sort, t_down, t_host, secs(Tb, now()));
    return 0;
}

} // namespace

// installed / removed by sgpu_init / sgpu_finalize (sgpu_runtime.hip)
extern "C" void sgpu_install_spgemm_hook(int on) {
    saena_host::g_spgemm_hook = (on && !std::getenv("SAENA_HOST_SPGEMM")) ? &gpu_spgemm : nullptr;
}

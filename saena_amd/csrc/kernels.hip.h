// kernels.hip.h -- hand-written gfx950 (CDNA4, wave64) kernels of the Saena V-cycle hot path.
//
// Everything here is HBM-bandwidth bound integer/fp64 streaming work; there is
// no dense contraction and therefore no MFMA.  What matters on MI355X:
//   * val/col are read with 16-byte-per-lane coalesced loads (1 KiB per wave
//     instruction), never per-row strided;
//   * products are staged in LDS and reduced per row by G lanes with DPP/
//     bpermute shuffles, so 7-nnz Poisson rows do not idle 57 of 64 lanes;
//   * smoother / residual / correction arithmetic is fused into the SpMV
//     epilogue: rhs, inv_diag, u and d are touched exactly once per sweep;
//   * row blocks are dealt to XCDs in contiguous chunks so neighbouring blocks
//     (which share x lines) hit the same 4 MiB L2.
//
// Reference loops restated (file:line in paralab/Saena):
//   local CSR loop          src/saena_matrix_matvec.cpp:68-80
//   remote CSC scatter      src/saena_matrix_matvec.cpp:93-109  (here: remote CSR, no atomics)
//   halo pack               src/saena_matrix_matvec.cpp:25-26, :464 (float)
//   jacobi update           src/saena_matrix.cpp:1061-1070
//   chebyshev update        src/saena_matrix.cpp:1099-1130, include/saena_matrix.tpp:35-43
//   residual                include/saena_matrix.tpp:16-23
//   u -= P e                src/saena_object_solve.cpp:1360-1361
//   dotProduct              include/aux_functions.h:116-123
//   solve_coarsest_CG       src/saena_object_solve.cpp:14-114
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sk {

constexpr int BLOCK   = 256;    // 4 waves
constexpr int CAP     = 2048;   // products staged per row block (16 KiB of LDS)
constexpr int CAP_BIG = 4096;   // alternative plan: 32 KiB of LDS, half the blocks
constexpr int MAXROWS = 256;    // rows per row block (one pass at 1 lane/row)
constexpr int NXCD    = 8;

enum Epi : int {
    EPI_SPMV = 0,      // y = s
    EPI_RESIDUAL,      // y = s - rhs
    EPI_JACOBI,        // t = s - rhs; t *= inv_diag*omega; y = u - t
    EPI_CHEBY0,        // d = (c*inv_diag)*(rhs - s);              y = u + d
    EPI_CHEBYK,        // d = d1*d + (c*inv_diag)*(rhs - s);       y = u + d
    EPI_SUB,           // y = y - s            (u -= P e)
    EPI_RSWEEP,        // y = s; first smoother sweep of the NEXT level from a zero iterate with rhs = s (restriction + k_zero_sweep)
    EPI_COUNT
};

struct SpmvArgs {
    const int    *row_ptr;   // [M+1]
    const int    *col;       // local column ids (already rebased), padded by >= 8 zeros
    const double *val;       // padded by >= 8 zeros
    const int    *blk_row;   // [nblk+1] row-block boundaries
    const double *x;         // input vector (local slice, or the halo buffer for the remote part)
    double       *y;         // output
    const double *rhs;
    const double *inv_diag;
    const double *u;         // smoother input iterate (== x for the local part)
    double       *d;         // chebyshev direction
    double       *y2;        // EPI_RSWEEP: the next level's iterate after its first sweep
    double        c0;        // omega | c
    double        c1;        // d1
    int           nblk;
    const unsigned *skip;    // bitmask of rows this launch must NOT write (boundary rows, owned by k_csr_boundary), or nullptr
    // 16-bit compressed columns (k_csr_cc16): per block a table of segment bases (multiples of 2^cc_ob), per nnz
    // (segment slot << cc_ob) | (column & (2^cc_ob - 1)); cc_ob in {12,...,8} = 16 ... 256 table slots per block
    const int            *segtab;   // bases of all blocks, block b owns [segptr[b], segptr[b+1])
    const int            *segptr;   // [nblk+1]
    const unsigned short *ccol;     // [nnz] (padded)
    int                   cc_ob;    // offset bits
    // column-major-in-block form (k_csr_cm): the block's entries re-sorted by column, each with the tile slot its product
    // belongs to; val / ccol then point at the re-sorted copies, cmptr[b] is the block's (quad-aligned) start in them
    const unsigned short *dst;
    const int            *cmptr;
    // row patterns (k_sellp): pt_n table rows of pt_w + 1 ints -- a row's length, then its columns relative to the row;
    // dst holds the pattern id of every row
    const int            *ptab;
    int                   pt_w, pt_n;
    const int            *rbase;     // k_sellp<rowbase>: the column a row's pattern is relative to (its first), or nullptr: relative to the row index
    int                   ncols;     // k_sellp2: columns of x (its 16-byte loads stay inside the vector)
    int                   nt;        // non-temporal stream loads (k_csr_stream / cc16 / cm / wave / xlds): see ld_stream_*
    int                   nt_from;   // k_sell: first slice read with non-temporal loads (the slices before it stay in the Infinity Cache)
    int                   st_plain;  // non-temporal launches: store y with plain stores all the same (it is the next sweep's x)
    int                   uw;        // k_sellp / k_sellp2: every slice has this many positions (0: read the slice pointers) -- a slice's values then
                                     // start at s * uw * rows-per-slice and the value loads depend on nothing the wave has to fetch first
    // in-kernel fork to the halo stream (multi-rank interior launch only, else nullptr): block 0 stores
    // *flag_x = seq when it starts -- stream order: everything earlier on the compute stream is complete, so
    // the halo stream's pack, which polls the flag, may read x.
    uint64_t *flag_x;
    uint64_t  seq;
};

// ---- flags between the two streams: plain device memory (one 128-B line each), agent-scope atomics.
// (hipMallocSignalMemory is host-resident: 125 blocks polling it over PCIe took ~40 us per round.) ----
// The polling loads are RELAXED: an acquire inside the loop would invalidate caches on every iteration
// (measured: 124 polling blocks tripled the interior kernel's time); one acquire fence follows the loop.
__device__ __forceinline__ void fork_signal(const SpmvArgs &a) {
    // stream order already completed (and released) everything earlier on this stream: a plain flag store suffices
    if (a.flag_x && blockIdx.x == 0 && threadIdx.x == 0)
        (void)__hip_atomic_exchange(a.flag_x, a.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // an RMW is performed at memory, a plain store may linger in L2
}
// every thread of the block waits until *flag >= seq.  NO acquire fence follows: on this part an agent-scope
// acquire is an L2 invalidate per wave (measured: ~46 us for the 500 waves of a pack launch); the caller reads
// the freshly produced data through agent-scope atomic loads instead (coherent_load below).
__device__ __forceinline__ void block_wait_flag(const uint64_t *flag, uint64_t seq) {
    if (flag) {
        if (threadIdx.x == 0)
            while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < seq) __builtin_amdgcn_s_sleep(32);
        __syncthreads();
    }
}
// a load that must observe what another kernel released at agent scope AFTER this kernel was dispatched
__device__ __forceinline__ double coherent_load(const double *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// HALO = the launch is the interior half of a multi-rank apply (row mask + in-kernel fork/join); the
// single-rank instantiations carry none of that code (it cost 1 % of the fine-level SpMV when it was a run-time test)
// NT: the once-per-sweep streams of the epilogue (rhs, inv_diag, d; the stores of y and d) bypass the caches' allocation
// (k_sellp on operators beyond the Infinity Cache, see there)
template <bool NT>
__device__ __forceinline__ double ld_once(const double *p) {
    if constexpr (NT) return __builtin_nontemporal_load(p);
    else return *p;
}
template <bool NT>
__device__ __forceinline__ void st_once(double *p, double v) {
    if constexpr (NT) __builtin_nontemporal_store(v, p);
    else *p = v;
}
typedef double sk_d2v_fwd __attribute__((ext_vector_type(2)));
// the store of y in a non-temporal launch: non-temporal unless the launch asks for plain stores (SpmvArgs::st_plain -- a smoother's
// output is the next sweep's input: a plain store leaves it in the Infinity Cache)
template <bool NT>
__device__ __forceinline__ void st_y(const SpmvArgs &a, double *p, double v) {
    if constexpr (NT) { if (a.st_plain) *p = v; else __builtin_nontemporal_store(v, p); }
    else *p = v;
}
template <bool NT>
__device__ __forceinline__ void st_y2(const SpmvArgs &a, double *p, sk_d2v_fwd v);
template <int EPI, bool HALO, bool NT = false>
__device__ __forceinline__ void epilogue(const SpmvArgs &a, int r, double s) {
    if constexpr (HALO)
        if (a.skip && ((a.skip[r >> 5] >> (r & 31)) & 1u)) return;  // a boundary row: the halo stream's kernel writes it
    if constexpr (EPI == EPI_SPMV) {
        st_y<NT>(a, a.y + r, s);
    } else if constexpr (EPI == EPI_RESIDUAL) {
        st_y<NT>(a, a.y + r, s - ld_once<NT>(a.rhs + r));
    } else if constexpr (EPI == EPI_JACOBI) {
        double t = s - ld_once<NT>(a.rhs + r);
        t *= ld_once<NT>(a.inv_diag + r) * a.c0;
        st_y<NT>(a, a.y + r, a.u[r] - t);
    } else if constexpr (EPI == EPI_CHEBY0) {
        const double dd = (a.c0 * ld_once<NT>(a.inv_diag + r)) * (ld_once<NT>(a.rhs + r) - s);
        st_once<NT>(a.d + r, dd);
        st_y<NT>(a, a.y + r, a.u[r] + dd);
    } else if constexpr (EPI == EPI_CHEBYK) {
        const double res = (a.c0 * ld_once<NT>(a.inv_diag + r)) * (ld_once<NT>(a.rhs + r) - s);
        const double dd  = (a.c1 * ld_once<NT>(a.d + r)) + res;
        st_once<NT>(a.d + r, dd);
        st_y<NT>(a, a.y + r, a.u[r] + dd);
    } else if constexpr (EPI == EPI_SUB) {
        st_y<NT>(a, a.y + r, a.y[r] - s);
    } else if constexpr (EPI == EPI_RSWEEP) {
        // the restriction's row r is the coarse level's right-hand side AND the input of that level's first sweep from a zero
        // iterate: k_zero_sweep's arithmetic, on the value just computed instead of on the one read back
        st_y<NT>(a, a.y + r, s);
        if (a.c1 != 0.0) {                                     // Chebyshev step 0 (c0 = 1 / theta): d = (c0 inv_diag) rhs, u = d
            const double dd = (a.c0 * a.inv_diag[r]) * s;
            a.d[r] = dd;
            a.y2[r] = dd;
        } else {                                               // Jacobi: u = rhs (inv_diag omega)
            a.y2[r] = s * (a.inv_diag[r] * a.c0);
        }
    }
}

// ---- stream loads.  val / col are read exactly once per launch; the x[col] gathers next to them live on reuse in the
// CU's 32 KiB vector L1.  Tried (SAENA_STREAM_NT=1): non-temporal (`nt`) stream loads, so that the tiles of stream
// data would not sweep the x lines out of L1.  Measured on the 256^3 hierarchy (profiles/r02_perf_levels_256_nt.log):
// every level got SLOWER (L0 335 -> 381 us, L1 1327 -> 1534 us, L2 629 -> 678 us), so plain loads stay the default.
typedef double   sk_d2v __attribute__((ext_vector_type(2)));
typedef double   sk_d2v8 __attribute__((ext_vector_type(2), aligned(8)));      // the same, promised 8-byte alignment only
typedef int      sk_i4v __attribute__((ext_vector_type(4)));
typedef unsigned sk_u2v __attribute__((ext_vector_type(2)));
// the same for the two consecutive rows r (even) and r + 1 of one lane: 16-byte loads and stores, the arithmetic of epilogue<> per row
template <bool NT>
__device__ __forceinline__ sk_d2v ld_once2(const double *p) {
    if constexpr (NT) return __builtin_nontemporal_load(reinterpret_cast<const sk_d2v *>(p));
    else return *reinterpret_cast<const sk_d2v *>(p);
}
template <bool NT>
__device__ __forceinline__ void st_once2(double *p, sk_d2v v) {
    if constexpr (NT) __builtin_nontemporal_store(v, reinterpret_cast<sk_d2v *>(p));
    else *reinterpret_cast<sk_d2v *>(p) = v;
}
template <bool NT>
__device__ __forceinline__ void st_y2(const SpmvArgs &a, double *p, sk_d2v_fwd v) {
    if constexpr (NT) { if (a.st_plain) *reinterpret_cast<sk_d2v_fwd *>(p) = v; else __builtin_nontemporal_store(v, reinterpret_cast<sk_d2v_fwd *>(p)); }
    else *reinterpret_cast<sk_d2v_fwd *>(p) = v;
}
template <int EPI, bool NT>
__device__ __forceinline__ void epilogue2(const SpmvArgs &a, int r, double s0, double s1) {
    sk_d2v s; s.x = s0; s.y = s1;
    if constexpr (EPI == EPI_SPMV) {
        st_y2<NT>(a, a.y + r, s);
    } else if constexpr (EPI == EPI_RESIDUAL) {
        const sk_d2v b = ld_once2<NT>(a.rhs + r);
        sk_d2v o; o.x = s.x - b.x; o.y = s.y - b.y;
        st_y2<NT>(a, a.y + r, o);
    } else if constexpr (EPI == EPI_JACOBI) {
        const sk_d2v b = ld_once2<NT>(a.rhs + r), dg = ld_once2<NT>(a.inv_diag + r), u = *reinterpret_cast<const sk_d2v *>(a.u + r);
        double t0 = s.x - b.x, t1 = s.y - b.y;
        t0 *= dg.x * a.c0; t1 *= dg.y * a.c0;
        sk_d2v o; o.x = u.x - t0; o.y = u.y - t1;
        st_y2<NT>(a, a.y + r, o);
    } else {                                                      // the remaining forms: per row
        epilogue<EPI, false, NT>(a, r, s0);
        epilogue<EPI, false, NT>(a, r + 1, s1);
    }
}

// `nt` is a launch argument (SpmvArgs::nt), uniform over the grid: a non-temporal load where the launch asks for it
__device__ __forceinline__ double2 ld_stream_d2(const double *p, int nt) {
    sk_d2v v;
    if (nt) v = __builtin_nontemporal_load(reinterpret_cast<const sk_d2v *>(p)); else v = *reinterpret_cast<const sk_d2v *>(p);
    double2 r; r.x = v.x; r.y = v.y; return r;
}
__device__ __forceinline__ int4 ld_stream_i4(const int *p, int nt) {
    sk_i4v v;
    if (nt) v = __builtin_nontemporal_load(reinterpret_cast<const sk_i4v *>(p)); else v = *reinterpret_cast<const sk_i4v *>(p);
    int4 r; r.x = v.x; r.y = v.y; r.z = v.z; r.w = v.w; return r;
}
__device__ __forceinline__ uint2 ld_stream_u2(const unsigned short *p, int nt) {
    sk_u2v v;
    if (nt) v = __builtin_nontemporal_load(reinterpret_cast<const sk_u2v *>(p)); else v = *reinterpret_cast<const sk_u2v *>(p);
    uint2 r; r.x = v.x; r.y = v.y; return r;
}

// contiguous chunk of the grid per XCD (blocks b, b+8, ... share an XCD)
__device__ __forceinline__ int xcd_remap(int b, int n) {
    const int per = n / NXCD, rem = n % NXCD;
    const int x = b % NXCD, k = b / NXCD;
    // XCD x owns [x*per + min(x,rem), ...) ; blocks beyond the even part fall through unchanged
    const int start = x * per + (x < rem ? x : rem);
    const int len   = per + (x < rem ? 1 : 0);
    return k < len ? start + k : b;
}

// sum of v over the G lanes of a row group (G power of two <= 64)
template <int G>
__device__ __forceinline__ double group_sum(double v) {
#pragma unroll
    for (int off = G / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// ---------------------------------------------------------------------------
// K1: row-block CSR SpMV with fused epilogue.
//   phase 1: the block streams its nnz range [p0,p1) with 16-B loads (a lane owns 4 consecutive
//            nnz), gathers x and writes val*x to LDS (one rounding, like the CPU's mul);
//   phase 2: G lanes per row add the row's products in column order (G = 1
//            reproduces the reference's sequential sum bit for bit) and the
//            group leader applies the epilogue.
// A row block with a single row longer than CAPV takes the long-row path.
// CAPV = products staged per block (LDS = 8 CAPV bytes).
// Measured and dropped (profiles/r01_variant_sweep_128*.log): issuing every load of the block
// before the first gather on 16 KiB tiles (no gain there; kept for 32 KiB tiles), 1- or 2-nnz lane ownership (more VMEM instructions, -10..-35 %),
// staging the block's distinct x columns in LDS (-40 % with 16/32 KiB tiles, -22 % with 1024-product tiles and
// 16-bit ids on the 67-nnz/row level: the extra dependent phase costs more than the saved gather lanes), and a
// persistent workgroup that prefetches the next block's stream during the reduce (-20 %: eight independent
// one-shot blocks per CU overlap better than a hand-pipelined loop with two barriers per block).
template <int EPI, int G, int CAPV, bool HALO>
__global__ __launch_bounds__(BLOCK) void k_csr_stream(const SpmvArgs a) {
    constexpr int LDSN = CAPV + 8;
    __shared__ __attribute__((aligned(16))) double lds[LDSN];
    const int tid = threadIdx.x;
    if constexpr (HALO) fork_signal(a);
    const int b   = xcd_remap(blockIdx.x, a.nblk);
    const int r0 = a.blk_row[b], r1 = a.blk_row[b + 1];
    const int p0 = a.row_ptr[r0], p1 = a.row_ptr[r1];

    if (r1 - r0 == 1 && p1 - p0 > CAPV) {             // ---- one long row
        double s = 0.0;
        for (int k = p0 + tid; k < p1; k += BLOCK) s += a.val[k] * a.x[a.col[k]];
        s = group_sum<64>(s);
        if ((tid & 63) == 0) lds[tid >> 6] = s;
        __syncthreads();
        if (tid == 0) {
            double t = lds[0];
#pragma unroll
            for (int w = 1; w < BLOCK / 64; ++w) t += lds[w];
            epilogue<EPI, HALO>(a, r0, t);
        }
        return;
    }

    // ---- phase 1: coalesced 16-B loads of val/col, gather x, products to LDS
    const int a0 = p0 & ~3;
    const int nq = (p1 - a0 + 3) >> 2;                // quads of 4 nnz
    constexpr int ITER = (LDSN / 4 + BLOCK - 1) / BLOCK;
    if constexpr (CAPV <= CAP) {
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
            const int q = tid + it * BLOCK;
            if (q < nq) {
                const int idx = a0 + 4 * q;
                const double2 v01 = ld_stream_d2(a.val + idx, a.nt);
                const double2 v23 = ld_stream_d2(a.val + idx + 2, a.nt);
                const int4    c   = ld_stream_i4(a.col + idx, a.nt);
                double2 o01, o23;
                o01.x = v01.x * a.x[c.x];
                o01.y = v01.y * a.x[c.y];
                o23.x = v23.x * a.x[c.z];
                o23.y = v23.y * a.x[c.w];
                *reinterpret_cast<double2 *>(&lds[4 * q])     = o01;
                *reinterpret_cast<double2 *>(&lds[4 * q + 2]) = o23;
            }
        }
    } else {
        // 32 KiB tiles run 5 blocks per CU: to keep enough bytes in flight every val/col load of the block is
        // issued before the first gather (out-of-range lanes re-read the last quad; the arrays are padded).
        // Measured on the 263-nnz/row level of the 256^3 hierarchy (2.4 GB, HBM-resident): 838 -> 647 us.
        double2 v01[ITER], v23[ITER];
        int4    c[ITER];
        const int qlast = nq > 0 ? nq - 1 : 0;
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
            int q = tid + it * BLOCK;
            q = q < qlast ? q : qlast;
            const int idx = a0 + 4 * q;
            v01[it] = ld_stream_d2(a.val + idx, a.nt);
            v23[it] = ld_stream_d2(a.val + idx + 2, a.nt);
            c[it]   = ld_stream_i4(a.col + idx, a.nt);
        }
        double2 o01[ITER], o23[ITER];
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
            o01[it].x = a.x[c[it].x];
            o01[it].y = a.x[c[it].y];
            o23[it].x = a.x[c[it].z];
            o23[it].y = a.x[c[it].w];
        }
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
            const int q = tid + it * BLOCK;
            if (q < nq) {
                double2 w01, w23;
                w01.x = v01[it].x * o01[it].x;
                w01.y = v01[it].y * o01[it].y;
                w23.x = v23[it].x * o23[it].x;
                w23.y = v23[it].y * o23[it].y;
                *reinterpret_cast<double2 *>(&lds[4 * q])     = w01;
                *reinterpret_cast<double2 *>(&lds[4 * q + 2]) = w23;
            }
        }
    }
    __syncthreads();

    // ---- phase 2: G lanes per row
    constexpr int ROWS_PER_PASS = BLOCK / G;
    const int g = tid / G, l = tid % G;
    for (int r = r0 + g; r < r1; r += ROWS_PER_PASS) {
        const int s = a.row_ptr[r] - a0, e = a.row_ptr[r + 1] - a0;
        double sum = 0.0;
        for (int k = s + l; k < e; k += G) sum += lds[k];
        sum = group_sum<G>(sum);
        if (l == 0) {
            epilogue<EPI, HALO>(a, r, sum);
        }
    }
}

// ---------------------------------------------------------------------------
// K1d: k_csr_stream with 16-bit compressed column ids.  The kernel is HBM-bound, so bytes are time:
// a row block of a mesh-like operator touches only a few column segments (a 7-point stencil: its
// three z-planes; a smoothed-aggregation coarse level: a 3D neighbourhood = a few dozen short runs), so each
// column is stored as a slot into the block's table of segment bases plus an offset inside the segment:
// 10 B/nnz instead of 12.  The slot/offset split is chosen per operator at plan time: 4+12 bits (16 segments of
// 4096 columns) for stencil-like levels, up to 8+8 bits (256 segments of 256 columns) for the coarse levels whose
// rows hold hundreds to thousands of entries.  Operators with a block that touches more than 256 segments keep
// the 32-bit kernel.  Arithmetic and summation order are those of k_csr_stream (bit-identical results).
__device__ __forceinline__ bool stray(int k, int p0, int p1) { return k < p0 || k >= p1; }
constexpr int CC_MAXSEG = 256;

template <int EPI, int G, int CAPV, bool HALO>
__global__ __launch_bounds__(BLOCK) void k_csr_cc16(const SpmvArgs a) {
    constexpr int LDSN = CAPV + 8;
    __shared__ __attribute__((aligned(16))) double lds[LDSN];
    __shared__ int seg[CC_MAXSEG];
    const int tid = threadIdx.x;
    if constexpr (HALO) fork_signal(a);
    const int b   = xcd_remap(blockIdx.x, a.nblk);
    const int r0 = a.blk_row[b], r1 = a.blk_row[b + 1];
    const int p0 = a.row_ptr[r0], p1 = a.row_ptr[r1];
    {
        const int s0 = a.segptr[b], ns = a.segptr[b + 1] - s0;      // ns <= CC_MAXSEG = BLOCK
        if (tid < ns) seg[tid] = a.segtab[s0 + tid];
    }
    __syncthreads();
    const int ob = a.cc_ob;
    const unsigned om = (1u << ob) - 1u;

    if (r1 - r0 == 1 && p1 - p0 > CAPV) {             // ---- one long row (its block has a table of its own)
        double s = 0.0;
        for (int k = p0 + tid; k < p1; k += BLOCK) {
            const unsigned c = a.ccol[k];
            s += a.val[k] * a.x[seg[c >> ob] + (int)(c & om)];
        }
        s = group_sum<64>(s);
        if ((tid & 63) == 0) lds[tid >> 6] = s;
        __syncthreads();
        if (tid == 0) {
            double t = lds[0];
#pragma unroll
            for (int w = 1; w < BLOCK / 64; ++w) t += lds[w];
            epilogue<EPI, HALO>(a, r0, t);
        }
        return;
    }

    const int a0 = p0 & ~3;
    const int nq = (p1 - a0 + 3) >> 2;
    constexpr int ITER = (LDSN / 4 + BLOCK - 1) / BLOCK;
    if constexpr (CAPV <= CAP) {
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
            const int q = tid + it * BLOCK;
            if (q < nq) {
                const int idx = a0 + 4 * q;
                const double2 v01 = ld_stream_d2(a.val + idx, a.nt);
                const double2 v23 = ld_stream_d2(a.val + idx + 2, a.nt);
                const uint2   c   = ld_stream_u2(a.ccol + idx, a.nt);
                const unsigned c0 = c.x & 0xffffu, c1 = c.x >> 16, c2 = c.y & 0xffffu, c3 = c.y >> 16;
                int j0 = seg[c0 >> ob] + (int)(c0 & om), j1 = seg[c1 >> ob] + (int)(c1 & om);
                int j2 = seg[c2 >> ob] + (int)(c2 & om), j3 = seg[c3 >> ob] + (int)(c3 & om);
                if (q == 0 || q == nq - 1) {          // quads shared with a neighbouring block: its ids were packed
                    j0 = stray(idx, p0, p1) ? 0 : j0; //  against ANOTHER segment table and may decode to anything
                    j1 = stray(idx + 1, p0, p1) ? 0 : j1;
                    j2 = stray(idx + 2, p0, p1) ? 0 : j2;
                    j3 = stray(idx + 3, p0, p1) ? 0 : j3;
                }
                double2 o01, o23;
                o01.x = v01.x * a.x[j0];
                o01.y = v01.y * a.x[j1];
                o23.x = v23.x * a.x[j2];
                o23.y = v23.y * a.x[j3];
                *reinterpret_cast<double2 *>(&lds[4 * q])     = o01;
                *reinterpret_cast<double2 *>(&lds[4 * q + 2]) = o23;
            }
        }
    } else {                                          // 32 KiB tiles: all stream loads first (see k_csr_stream)
        double2 v01[ITER], v23[ITER];
        uint2   c[ITER];
        const int qlast = nq > 0 ? nq - 1 : 0;
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
            int q = tid + it * BLOCK;
            q = q < qlast ? q : qlast;
            const int idx = a0 + 4 * q;
            v01[it] = ld_stream_d2(a.val + idx, a.nt);
            v23[it] = ld_stream_d2(a.val + idx + 2, a.nt);
            c[it]   = ld_stream_u2(a.ccol + idx, a.nt);
        }
        double2 o01[ITER], o23[ITER];
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
            const unsigned c0 = c[it].x & 0xffffu, c1 = c[it].x >> 16, c2 = c[it].y & 0xffffu, c3 = c[it].y >> 16;
            int j0 = seg[c0 >> ob] + (int)(c0 & om), j1 = seg[c1 >> ob] + (int)(c1 & om);
            int j2 = seg[c2 >> ob] + (int)(c2 & om), j3 = seg[c3 >> ob] + (int)(c3 & om);
            int q = tid + it * BLOCK;
            q = q < qlast ? q : qlast;
            if (q == 0 || q == qlast) {               // quads shared with a neighbouring block (see above)
                const int idx = a0 + 4 * q;
                j0 = stray(idx, p0, p1) ? 0 : j0;
                j1 = stray(idx + 1, p0, p1) ? 0 : j1;
                j2 = stray(idx + 2, p0, p1) ? 0 : j2;
                j3 = stray(idx + 3, p0, p1) ? 0 : j3;
            }
            o01[it].x = a.x[j0];
            o01[it].y = a.x[j1];
            o23[it].x = a.x[j2];
            o23[it].y = a.x[j3];
        }
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
            const int q = tid + it * BLOCK;
            if (q < nq) {
                double2 w01, w23;
                w01.x = v01[it].x * o01[it].x;
                w01.y = v01[it].y * o01[it].y;
                w23.x = v23[it].x * o23[it].x;
                w23.y = v23[it].y * o23[it].y;
                *reinterpret_cast<double2 *>(&lds[4 * q])     = w01;
                *reinterpret_cast<double2 *>(&lds[4 * q + 2]) = w23;
            }
        }
    }
    __syncthreads();

    constexpr int ROWS_PER_PASS = BLOCK / G;
    const int g = tid / G, l = tid % G;
    for (int r = r0 + g; r < r1; r += ROWS_PER_PASS) {
        const int s = a.row_ptr[r] - a0, e = a.row_ptr[r + 1] - a0;
        double sum = 0.0;
        for (int k = s + l; k < e; k += G) sum += lds[k];
        sum = group_sum<G>(sum);
        if (l == 0) {
            epilogue<EPI, HALO>(a, r, sum);
        }
    }
}

// ---------------------------------------------------------------------------
// K1h: k_csr_cc16 with the block's entries streamed in COLUMN order.  The smoothed-aggregation levels are bound by
// vector-L1 miss requests, 4-5 per distinct x line a block touches (profiles/r02_pmc_l1_*.txt): the rows of a block share
// most of their columns, but in row-major order the entries that want one line sit hundreds of nnz apart, in different
// wave instructions, and each asks for the line again while it is in flight.  Here the block's entries are stored
// sorted by (column, row): the entries that share a line are neighbours in the stream -- the same lane, adjacent lanes,
// one instruction -- and every entry carries the tile slot its product belongs to (16 bit), so the products still land in
// row-major order and phase 2 adds each row's products in column order: bit-identical results.  12 B/nnz (8 value +
// 2 column + 2 slot) instead of 10.
template <int EPI, int G, int CAPV, bool HALO>
__global__ __launch_bounds__(BLOCK) void k_csr_cm(const SpmvArgs a) {
    constexpr int LDSN = CAPV + 8;
    __shared__ __attribute__((aligned(16))) double lds[LDSN];
    __shared__ int seg[CC_MAXSEG];
    const int tid = threadIdx.x;
    if constexpr (HALO) fork_signal(a);
    const int b   = xcd_remap(blockIdx.x, a.nblk);
    const int r0 = a.blk_row[b], r1 = a.blk_row[b + 1];
    const int p0 = a.row_ptr[r0];
    {
        const int s0 = a.segptr[b], ns = a.segptr[b + 1] - s0;
        if (tid < ns) seg[tid] = a.segtab[s0 + tid];
    }
    __syncthreads();
    const int ob = a.cc_ob;
    const unsigned om = (1u << ob) - 1u;
    const int c0 = a.cmptr[b], nq = (a.cmptr[b + 1] - c0) >> 2;     // the block's quads in the re-sorted arrays (padded to whole quads)
    constexpr int ITER = (LDSN / 4 + BLOCK - 1) / BLOCK;
    double2 v01[ITER], v23[ITER];
    uint2   c[ITER], d[ITER];
    const int qlast = nq > 0 ? nq - 1 : 0;
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
        int q = tid + it * BLOCK;
        q = q < qlast ? q : qlast;
        const int idx = c0 + 4 * q;
        v01[it] = ld_stream_d2(a.val + idx, a.nt);
        v23[it] = ld_stream_d2(a.val + idx + 2, a.nt);
        c[it]   = ld_stream_u2(a.ccol + idx, a.nt);
        d[it]   = ld_stream_u2(a.dst + idx, a.nt);
    }
    double2 o01[ITER], o23[ITER];
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
        const unsigned c0_ = c[it].x & 0xffffu, c1 = c[it].x >> 16, c2 = c[it].y & 0xffffu, c3 = c[it].y >> 16;
        o01[it].x = a.x[seg[c0_ >> ob] + (int)(c0_ & om)];
        o01[it].y = a.x[seg[c1 >> ob] + (int)(c1 & om)];
        o23[it].x = a.x[seg[c2 >> ob] + (int)(c2 & om)];
        o23[it].y = a.x[seg[c3 >> ob] + (int)(c3 & om)];
    }
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
        const int q = tid + it * BLOCK;
        if (q < nq) {                                  // padding entries carry value 0 and the spare slot CAPV + 4
            lds[d[it].x & 0xffffu] = v01[it].x * o01[it].x;
            lds[d[it].x >> 16]     = v01[it].y * o01[it].y;
            lds[d[it].y & 0xffffu] = v23[it].x * o23[it].x;
            lds[d[it].y >> 16]     = v23[it].y * o23[it].y;
        }
    }
    __syncthreads();

    const int a0 = p0 & ~3;
    constexpr int ROWS_PER_PASS = BLOCK / G;
    const int g = tid / G, l = tid % G;
    for (int r = r0 + g; r < r1; r += ROWS_PER_PASS) {
        const int s = a.row_ptr[r] - a0, e = a.row_ptr[r + 1] - a0;
        double sum = 0.0;
        for (int k = s + l; k < e; k += G) sum += lds[k];
        sum = group_sum<G>(sum);
        if (l == 0) {
            epilogue<EPI, HALO>(a, r, sum);
        }
    }
}

// ---------------------------------------------------------------------------
// K1s: sliced ELLPACK for even rows (the stencil levels and the first smoothed-aggregation level).  A slice is 64
// consecutive rows = one wave; its entries are stored position-major -- entries 2q and 2q+1 of the slice's rows side by
// side, lane by lane (an odd last position alone; operators of fewer than 16 entries per row: one position at a time,
// PAIR = false) -- padded to the slice's longest row.  A lane owns a row: it adds the
// row's products in column order in a register (the reference's sequential sum, whatever the launch shape), there is no
// product tile, no phase 2 and no row pointer, every stream load of the wave is one contiguous 1 KiB (two values per
// lane) or 256-byte (two 16-bit column codes per lane) piece, on a stencil the gathers of an instruction are 64
// consecutive doubles of x, and elsewhere a lane walks along its own row, whose neighbouring columns share lines that
// are still in L1.  Columns are the 16-bit codes of k_csr_cc16 against a segment table per workgroup (4 slices): 10 B
// per stored entry.  Operators whose padding exceeds 12 % keep the CSR kernels.
// Two positions per load matter: with one (8-byte value loads) the kernel ran 1 055-1 210 us on 256^3 L1 depending on
// the box, with two 1 003-1 018 us on the slowest of them (profiles/r02_sell_pairs.log) -- it was short of issue slots, not
// bytes.  On the 7-entry rows of the cache-resident 128^3 fine level single positions are the faster form (25.4 vs ~26.5 us).
// Four pairs are in flight per lane (2 / 6 / 8 measured equal or worse as single positions, profiles/r02_sell_unroll.log).
// a.val / a.ccol: the padded arrays, a.segtab / a.segptr / a.cc_ob: the tables, a.cmptr: slice starts (multiples of 64),
// a.dst: row lengths, a.nblk: slices.
template <int EPI, bool HALO, bool PAIR, bool NT>
__global__ __launch_bounds__(BLOCK) void k_sell(const SpmvArgs a, int nrows) {
    __shared__ int seg[CC_MAXSEG];
    if constexpr (HALO) fork_signal(a);
    constexpr int SPB = BLOCK / 64;
    const int b = xcd_remap(blockIdx.x, (a.nblk + SPB - 1) / SPB);
    {
        const int s0 = a.segptr[b], ns = a.segptr[b + 1] - s0;      // ns <= CC_MAXSEG = BLOCK
        if ((int)threadIdx.x < ns) seg[threadIdx.x] = a.segtab[s0 + threadIdx.x];
    }
    __syncthreads();
    const int s = __builtin_amdgcn_readfirstlane(b * SPB + ((int)threadIdx.x >> 6));
    if (s >= a.nblk) return;
    const int lane = threadIdx.x & 63;
    const int r = s * 64 + lane;
    const int p = a.cmptr[s], w = (a.cmptr[s + 1] - p) >> 6;
    const int len = r < nrows ? (int)a.dst[r] : 0;
    // k_sell<sorted> (round 4): the slices hold the rows SORTED by length inside windows of 2048 rows (uneven rows: 20 % padding ->
    // 1-2 %); a.blk_row maps a slice position to the row it holds, and that is where the epilogue reads and writes
    const int ro = a.blk_row ? (r < nrows ? a.blk_row[r] : r) : r;
    const int ob = a.cc_ob;
    const unsigned om = (1u << ob) - 1u;
    if constexpr (!PAIR) {                             // rows of a handful of entries (the stencil level, its transfers): one position per load
        const double         *v = a.val + p + lane;
        const unsigned short *c = a.ccol + p + lane;
        double sum = 0.0;
        for (int j = 0; j < w; j += 8) {
            double   vv[8], xx[8];
            unsigned cc[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {             // all stream loads first; positions past the slice's width re-read its last one
                const int jj = j + u < w ? j + u : w - 1;
                vv[u] = ld_once<NT>(v + jj * 64);
                if constexpr (NT) cc[u] = __builtin_nontemporal_load(c + jj * 64); else cc[u] = c[jj * 64];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) xx[u] = a.x[seg[cc[u] >> ob] + (int)(cc[u] & om)];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (j + u < len) sum += vv[u] * xx[u];
        }
        if (r < nrows) epilogue<EPI, HALO, NT>(a, ro, sum);
        return;
    }
    const int P = w >> 1;                              // pairs of positions; an odd last one follows them
    const sk_d2v   *v2 = reinterpret_cast<const sk_d2v *>(a.val + p) + lane;
    const unsigned *c2 = reinterpret_cast<const unsigned *>(a.ccol + p) + lane;
    // the first nt_from slices of an operator beyond the Infinity Cache are read with plain loads, the rest non-temporal: the
    // plain part (sized to the cache) allocates there and is found there by the next sweep, the non-temporal part does not evict it
    const bool nt = NT && s >= a.nt_from;
    constexpr int UP = 4;
    double sum = 0.0;
    // the odd last position is fetched FIRST (its product is added last): behind the loop it was a second round trip of
    // latency in the life of a wave whose rows hold 7 entries
    double   vt = 0.0, xt = 0.0;
    if (w & 1) {
        vt = ld_once<NT>(a.val + p + P * 128 + lane);
        const unsigned ct = a.ccol[p + P * 128 + lane];
        xt = a.x[seg[ct >> ob] + (int)(ct & om)];
    }
    for (int q = 0; q < P; q += UP) {
        sk_d2v   vv[UP];
        unsigned cc[UP];
        double   x0[UP], x1[UP];
#pragma unroll
        for (int u = 0; u < UP; ++u) {                 // all stream loads first; pairs past the slice's width re-read its last one
            const int qq = q + u < P ? q + u : P - 1;
            if (nt) { vv[u] = __builtin_nontemporal_load(v2 + qq * 64); cc[u] = __builtin_nontemporal_load(c2 + qq * 64); }
            else { vv[u] = v2[qq * 64]; cc[u] = c2[qq * 64]; }
        }
#pragma unroll
        for (int u = 0; u < UP; ++u) {
            const unsigned ca = cc[u] & 0xffffu, cb = cc[u] >> 16;
            x0[u] = a.x[seg[ca >> ob] + (int)(ca & om)];
            x1[u] = a.x[seg[cb >> ob] + (int)(cb & om)];
        }
#pragma unroll
        for (int u = 0; u < UP; ++u) {                 // padding (value 0) is never added: a NaN or inf in x stays in the rows that own it
            if (q + u < P && 2 * (q + u) < len) sum += vv[u].x * x0[u];
            if (q + u < P && 2 * (q + u) + 1 < len) sum += vv[u].y * x1[u];
        }
    }
    if ((w & 1) && w - 1 < len) sum += vt * xt;
    if (r < nrows) epilogue<EPI, HALO, NT>(a, ro, sum);
}

// ---------------------------------------------------------------------------
// K1p: sliced ELLPACK WITHOUT a column stream, for operators whose rows repeat a few patterns (stencils on structured
// grids, band matrices).  The reference's local loop (src/saena_matrix_matvec.cpp:68-80) reads a column id per entry;
// on the 7-point level every interior row holds the columns r - n^2, r - n, r - 1, r, r + 1, r + n, r + n^2, so the ids
// carry one fact per ROW: which of a handful of (length, column offsets relative to the row) patterns it follows -- 27
// for the boundary-stripped 3D Laplacian.  The values keep k_sell's layout (position-major slices of 64 rows, a lane per
// row, pairs of positions per 16-byte load); the 16-bit column code per entry and the 16-bit length per row give way to
// ONE 16-bit pattern id per row, and the pattern table (pt_n x (pt_w + 1) ints, <= 16 KiB) sits in LDS: 8 B per entry +
// 2 B per row instead of 10 + 2.  Lanes of a wave that follow the same pattern gather 64 consecutive doubles of x.
// Same products, same sequential sum per row as k_sell and k_csr_stream at one lane per row: bit-identical results.
// a.val: k_sell's padded values, a.cmptr: slice starts, a.dst: pattern ids, a.ptab / pt_w / pt_n: the table, a.nblk: slices.
// NT: the operator does not fit the 256 MiB Infinity Cache: the once-per-sweep streams (values, pattern ids, rhs, inv_diag,
// the stores of y) are non-temporal, x is not.  Measured on Poisson 256^3 (1.2 GB stored, profiles/r03_sellp_nt.log):
// 257 -> 220 us back to back; on the cache-resident 128^3 operator it loses (22.5 -> 30 us), so the launch picks by size.
// Two slices per wave (twice the loads in flight per lane) changed nothing, four were slower (profiles/r03_sellp_ns.log).
// WIDE: an operator whose rows follow a few HUNDRED patterns of several dozen entries -- the first smoothed-aggregation level of
// a structured grid: Poisson 128^3 / 256^3 level 1 has 321 patterns, 14 469 offsets in all, whatever the size (two of them cover
// 85 % of the rows); the local part of a rank of the row-partitioned level has a hundred more (rows next to a neighbour lose their
// remote entries).  A workgroup of 1024 threads = 1024 consecutive rows meets a few dozen of them, so the host stores, PER
// WORKGROUP, just those (a.ptab + a.segptr[b]: n start offsets, then per pattern its length and that many offsets; at most
// SPW_MAX_TABLE ints, typically 2 K = 8 KiB next to 557 KiB of values) and a.dst numbers the rows' patterns within them -- no limit on
// the number of patterns of the operator.  (First form: one compact table of all patterns in 72-78 KiB of LDS per workgroup; the
// local parts of a 4-rank run already exceeded it.)
constexpr int SPW_MAX_TABLE = 8192;
constexpr int SPW_BLOCK     = 1024;
template <int EPI, bool HALO, bool PAIR, bool NT, bool WIDE = false>
__global__ __launch_bounds__(WIDE ? SPW_BLOCK : BLOCK) void k_sellp(const SpmvArgs a, int nrows) {
    constexpr int BS = WIDE ? SPW_BLOCK : BLOCK;
    extern __shared__ int ptab_lds[];
    if constexpr (HALO) fork_signal(a);
    constexpr int SPB = BS / 64;
    const int lane = threadIdx.x & 63;
    const int ngrp = (a.nblk + SPB - 1) / SPB;
    const int b0 = xcd_remap(blockIdx.x, ngrp);
    {
        const int t0 = WIDE ? a.segptr[b0] : 0, tn = WIDE ? a.segptr[b0 + 1] - t0 : a.pt_n * (a.pt_w + 1);
        for (int i = threadIdx.x; i < tn; i += BS) ptab_lds[i] = a.ptab[t0 + i];
    }
    __syncthreads();
    {
    const int s = __builtin_amdgcn_readfirstlane(b0 * SPB + ((int)threadIdx.x >> 6));
    if (s >= a.nblk) return;
    const int r = s * 64 + lane;
    int p, w;
    if (a.uw) { w = a.uw; p = s * w * 64; }               // uniform slices (a stencil level): no slice pointers in the chain of dependent loads
    else { p = a.cmptr[s]; w = (a.cmptr[s + 1] - p) >> 6; }
    int pid = 0;
    if (r < nrows) { if constexpr (NT) pid = __builtin_nontemporal_load(a.dst + r); else pid = a.dst[r]; }
    const int *pt = WIDE ? ptab_lds + ptab_lds[pid] : ptab_lds + pid * (a.pt_w + 1);
    const int len = r < nrows ? pt[0] : 0;
    ++pt;
    // column of position j of this lane's row: the table is read at a clamped position (no branch per position), positions
    // past the row's length read x[0] and are never added
    const int wmax = WIDE ? (len > 0 ? len - 1 : 0) : a.pt_w - 1;
    const int rf = a.rbase ? (r < nrows ? a.rbase[r] : 0) : r;       // what the pattern's offsets are relative to
    auto colof = [&](int j) { const int c = rf + pt[j < wmax ? j : wmax]; return j < len ? c : 0; };
    const bool ntv = NT && s >= a.nt_from;                 // (k_sell: the first nt_from slices stay in the Infinity Cache)
    double sum = 0.0;
    if constexpr (!PAIR) {
        const double *v = a.val + p + lane;
        for (int j = 0; j < w; j += 8) {
            double vv[8], xx[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) vv[u] = ld_once<NT>(v + (j + u < w ? j + u : w - 1) * 64);
#pragma unroll
            for (int u = 0; u < 8; ++u) xx[u] = a.x[colof(j + u)];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (j + u < len) sum += vv[u] * xx[u];
        }
    } else {
        const int P = w >> 1;
        const sk_d2v *v2 = reinterpret_cast<const sk_d2v *>(a.val + p) + lane;
        constexpr int UP = 4;
        double vt = 0.0, xt = 0.0;
        if (w & 1) {                                   // the odd last position first, as in k_sell
            vt = ld_once<NT>(a.val + p + P * 128 + lane);
            xt = a.x[colof(w - 1)];
        }
        for (int q = 0; q < P; q += UP) {
            sk_d2v vv[UP];
            double x0[UP], x1[UP];
#pragma unroll
            for (int u = 0; u < UP; ++u) {
                const sk_d2v *vp = v2 + (q + u < P ? q + u : P - 1) * 64;
                if (ntv) vv[u] = __builtin_nontemporal_load(vp); else vv[u] = *vp;
            }
#pragma unroll
            for (int u = 0; u < UP; ++u) {
                x0[u] = a.x[colof(2 * (q + u))];
                x1[u] = a.x[colof(2 * (q + u) + 1)];
            }
#pragma unroll
            for (int u = 0; u < UP; ++u) {
                if (q + u < P && 2 * (q + u) < len) sum += vv[u].x * x0[u];
                if (q + u < P && 2 * (q + u) + 1 < len) sum += vv[u].y * x1[u];
            }
        }
        if ((w & 1) && w - 1 < len) sum += vt * xt;
    }
    if (r < nrows) epilogue<EPI, HALO, NT>(a, r, sum);
    }
}

// ---------------------------------------------------------------------------
// K1x: k_sellp with THE INPUT VECTOR IN LDS.  On the first smoothed-aggregation level (68 entries per row) k_sellp<WIDE> is bound
// by the load path, not by bytes: for every 8 B of the value stream it gathers 8 B of x, and the two together pass the CU's
// vector-memory unit at ~16 B per clock (4.7 TB/s of values on 256^3 L1; half the gather instructions -- k_sellp2<WIDE> --
// change nothing).  But the offsets of a pattern operator come in a few CLUSTERS (level 1: the 5 x 5 grid lines around the
// row in 5 planes -- five clusters of a few hundred columns): a workgroup of SPX_ROWS consecutive rows reads x only inside
// nwin windows [r0 + omin_c, r0 + SPX_ROWS + omax_c), 41-70 KiB in all, which it loads ONCE with coalesced loads (a ninth of
// the bytes the gathers moved) and then gathers from LDS (a ds_read_b64 of 64 consecutive doubles: 4 clocks).
// The pattern table moves to 16 bits: entry j of a pattern is the LDS position of its column for the workgroup's first row
// (window base + offset inside the window), a lane adds its row's index in the workgroup.  And it becomes PER WORKGROUP: the 512
// rows of a workgroup follow a dozen of the operator's patterns, so the host stores, per workgroup, just those (2-3 KiB next to
// 278 KiB of values) and numbers the rows' pattern ids within them -- the whole table (31 KiB on level 1) would leave room for one
// workgroup per CU only, which measured 147 against k_sellp<WIDE>'s 111 us.
// a.val / a.cmptr: k_sell's values and slice starts; a.dst: workgroup-local pattern ids; a.ptab: the workgroups' tables as 16-bit
// words (n starts, then per pattern its length and positions), workgroup b owns words [a.segptr[b], a.segptr[b + 1]); a.segtab:
// nwin, then (omin, LDS base, size) per window.
// Same products, same sequential row sum: bit-identical to k_sellp.
constexpr int SPX_BLOCK = 512;
constexpr int SPX_ROWS  = 512;                 // rows per workgroup: 8 slices of 64 (256 rows, 3-4 workgroups per CU: 127.5 against 119.7 us of k_sellp<WIDE>)
constexpr int SPX_MAXWIN = 16;
constexpr int SPX_LDS_BYTES = 80 * 1024;       // windows + table: two workgroups per CU
template <int EPI, bool HALO, bool PAIR, bool NT>
__global__ __launch_bounds__(SPX_BLOCK) void k_sellpx(const SpmvArgs a, int nrows) {
    __shared__ double spx_lds[SPX_LDS_BYTES / 8];
    if constexpr (HALO) fork_signal(a);
    constexpr int SPW = SPX_ROWS / 64;                            // slices per workgroup
    const int ngrp = (a.nblk + SPW - 1) / SPW;
    const int b = xcd_remap(blockIdx.x, ngrp);
    const int r0 = b * SPX_ROWS;
    const int nwin = a.segtab[0];
    int xtot = 0;
    for (int c = 0; c < nwin; ++c) {                              // the windows of x, zero where they leave the vector
        const int omin = a.segtab[1 + 3 * c], base = a.segtab[2 + 3 * c], size = a.segtab[3 + 3 * c];
        const int c0 = r0 + omin;
        for (int i = threadIdx.x; i < size; i += SPX_BLOCK) {     // (a flat loop over all windows with 5 or 20 loads in flight per thread
            const int col = c0 + i;                               //  measured slower: 126 / 134 against 114 us on 128^3 level 1)
            spx_lds[base + i] = (col >= 0 && col < a.ncols) ? a.x[col] : 0.0;
        }
        xtot = base + size;
    }
    unsigned short *tab = reinterpret_cast<unsigned short *>(spx_lds + xtot);
    {                                                             // this workgroup's patterns (a multiple of four 16-bit words)
        const int w0 = a.segptr[b], nw = a.segptr[b + 1] - w0;
        const uint2 *gt = reinterpret_cast<const uint2 *>(reinterpret_cast<const unsigned short *>(a.ptab) + w0);
        uint2 *lt = reinterpret_cast<uint2 *>(tab);
        for (int i = threadIdx.x; i < (nw >> 2); i += SPX_BLOCK) lt[i] = gt[i];
    }
    __syncthreads();
    const int s = __builtin_amdgcn_readfirstlane(b * SPW + ((int)threadIdx.x >> 6));
    if (s >= a.nblk) return;
    const int lane = threadIdx.x & 63;
    const int r = s * 64 + lane;
    const int p = a.cmptr[s], w = (a.cmptr[s + 1] - p) >> 6;
    int pid = 0;
    if (r < nrows) { if constexpr (NT) pid = __builtin_nontemporal_load(a.dst + r); else pid = a.dst[r]; }
    const unsigned short *pt = tab + tab[pid];
    const int len = r < nrows ? (int)pt[0] : 0;
    ++pt;
    const int wmax = len > 0 ? len - 1 : 0;
    const double *xr = spx_lds + (r - r0);                        // position j of this lane's row: xr[pt[j]]
    auto xof = [&](int j) { return xr[pt[j < wmax ? j : wmax]]; };      // (positions past the row's length re-read its last column and are never added)
    const bool ntv = NT && s >= a.nt_from;
    double sum = 0.0;
    if constexpr (!PAIR) {
        const double *v = a.val + p + lane;
        for (int j = 0; j < w; j += 8) {
            double vv[8], xx[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) vv[u] = ld_once<NT>(v + (j + u < w ? j + u : w - 1) * 64);
#pragma unroll
            for (int u = 0; u < 8; ++u) xx[u] = xof(j + u);
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (j + u < len) sum += vv[u] * xx[u];
        }
    } else {
        const int P = w >> 1;
        const sk_d2v *v2 = reinterpret_cast<const sk_d2v *>(a.val + p) + lane;
        constexpr int UP = 8;
        double vt = 0.0, xt = 0.0;
        if (w & 1) {                                   // the odd last position first, as in k_sell
            vt = ld_once<NT>(a.val + p + P * 128 + lane);
            xt = xof(w - 1);
        }
        for (int q = 0; q < P; q += UP) {
            sk_d2v vv[UP];
#pragma unroll
            for (int u = 0; u < UP; ++u) {
                const sk_d2v *vp = v2 + (q + u < P ? q + u : P - 1) * 64;
                if (ntv) vv[u] = __builtin_nontemporal_load(vp); else vv[u] = *vp;
            }
#pragma unroll
            for (int u = 0; u < UP; ++u) {
                const double x0 = xof(2 * (q + u)), x1 = xof(2 * (q + u) + 1);
                if (q + u < P && 2 * (q + u) < len) sum += vv[u].x * x0;
                if (q + u < P && 2 * (q + u) + 1 < len) sum += vv[u].y * x1;
            }
        }
        if ((w & 1) && w - 1 < len) sum += vt * xt;
    }
    if (r < nrows) epilogue<EPI, HALO, NT>(a, r, sum);
}

// ---------------------------------------------------------------------------
// K1q: k_sellp with a lane owning TWO consecutive rows.  On the 7-point level a gather instruction of k_sellp moves 64
// consecutive doubles, and its cost is per INSTRUCTION (~20 ns per instruction and CU whatever the 64 addresses are,
// profiles/r02_gather_bench.log): 7 gathers per 64 rows are ~17 us of the cache-resident 128^3 operator's 23 us.  Rows r and
// r + 1 of a stencil follow the same pattern almost always (a grid line ends every 254 rows), so their columns r + o and r + 1 + o
// are adjacent: ONE 16-byte load serves both, and the values of the two rows at position j sit side by side for one 16-byte
// load as well (slices of 128 rows, position-major: [j][lane][row 2 lane, row 2 lane + 1]).  Half the gather instructions, the
// same bytes, the same sequential sum per row: bit-identical to k_sellp.
// a.val: the row-paired values, a.cmptr: slice starts (multiples of 128), a.dst: pattern ids, a.ptab / pt_w / pt_n: the table,
// a.nblk: slices of 128 rows.
// Lanes whose two rows differ in pattern (1 % of them on the 7-point level) take two 8-byte gathers.
// WIDE: the tables of k_sellp<WIDE> (a few hundred patterns of several dozen entries: level 1 of a structured grid; one table per
// 1024 rows), 512 threads = the same 1024 rows per workgroup.  There a lane whose two rows differ in pattern still takes the 16-byte
// load for its first row and ONE 8-byte gather for the second: a wave with such lanes issues two gather instructions per position
// instead of three (on level 1 of the Poisson cube the pattern changes at the end of every grid line -- 63 rows at 128^3 -- so nearly
// every slice of 128 rows holds such a lane).  Measured on that level: the fastest form on the 128^3 operator (110 against
// k_sellp<WIDE>'s 116 and k_sellpx's 114 us), a tie on the 256^3 one (940 / 945 us; k_sellpx 890).
constexpr int SPW2_BLOCK = 512;
// PRE (round 4): the epilogue's once-per-sweep operands (rhs, inv_diag, u of the lane's two rows) are fetched at the TOP of the wave, next
// to the values, instead of after the row sums are complete -- one level less in the wave's chain of dependent memory round trips
// (ids / pointers -> values + gathers -> epilogue operands -> store), at 12 more registers.  Jacobi and residual launches only.
template <int EPI, bool HALO, bool NT, bool WIDE = false, bool PRE = false>
__global__ __launch_bounds__(WIDE ? SPW2_BLOCK : BLOCK) void k_sellp2(const SpmvArgs a, int nrows) {
    constexpr int BS = WIDE ? SPW2_BLOCK : BLOCK;
    extern __shared__ int ptab_lds[];
    if constexpr (HALO) fork_signal(a);
    constexpr int SPB = BS / 64;
    const int lane = threadIdx.x & 63;
    const int ngrp = (a.nblk + SPB - 1) / SPB;
    const int b0 = xcd_remap(blockIdx.x, ngrp);
    {                                                              // (WIDE: the workgroup's own table -- its 8 slices of 128 rows are k_sellp<WIDE>'s 16 of 64)
        const int t0 = WIDE ? a.segptr[b0] : 0, tn = WIDE ? a.segptr[b0 + 1] - t0 : a.pt_n * (a.pt_w + 1);
        for (int i = threadIdx.x; i < tn; i += BS) ptab_lds[i] = a.ptab[t0 + i];
    }
    __syncthreads();
    const int xlast = a.ncols - 1;
    {
    const int s = __builtin_amdgcn_readfirstlane(b0 * SPB + ((int)threadIdx.x >> 6));
    if (s >= a.nblk) return;
    const int rA = s * 128 + 2 * lane, rB = rA + 1;
    int p, w;
    if (a.uw) { w = a.uw; p = s * w * 128; }              // uniform slices: the value loads depend on nothing the wave has to fetch first
    else { p = a.cmptr[s]; w = (a.cmptr[s + 1] - p) >> 7; }
    sk_d2v pre_b = {0.0, 0.0}, pre_dg = {0.0, 0.0}, pre_u = {0.0, 0.0};
    constexpr bool PREF = PRE && (EPI == EPI_JACOBI || EPI == EPI_RESIDUAL);
    if constexpr (PREF) {
        if (rB < nrows) {
            pre_b = ld_once2<NT>(a.rhs + rA);
            if constexpr (EPI == EPI_JACOBI) { pre_dg = ld_once2<NT>(a.inv_diag + rA); pre_u = *reinterpret_cast<const sk_d2v *>(a.u + rA); }
        }
    }
    unsigned ids = 0;
    if (rA < nrows) {                                               // (the id array is padded to a multiple of 128 rows)
        const unsigned *ip = reinterpret_cast<const unsigned *>(a.dst) + (rA >> 1);
        if constexpr (NT) ids = __builtin_nontemporal_load(ip); else ids = *ip;
    }
    const int pidA = (int)(ids & 0xffffu), pidB = (int)(ids >> 16);
    const int *ptA = WIDE ? ptab_lds + ptab_lds[pidA] : ptab_lds + pidA * (a.pt_w + 1);
    const int *ptB = WIDE ? ptab_lds + ptab_lds[pidB] : ptab_lds + pidB * (a.pt_w + 1);
    const int lenA = rA < nrows ? ptA[0] : 0, lenB = rB < nrows ? ptB[0] : 0;
    ++ptA; ++ptB;
    const bool same = pidA == pidB && rB < nrows;                 // the two rows read adjacent columns at every position
    const int wmaxA = WIDE ? (lenA > 0 ? lenA - 1 : 0) : a.pt_w - 1, wmaxB = WIDE ? (lenB > 0 ? lenB - 1 : 0) : a.pt_w - 1;
    const sk_d2v *v2 = reinterpret_cast<const sk_d2v *>(a.val + p) + lane;
    const bool ntv = NT && s >= a.nt_from;                 // (k_sell: the first nt_from slices stay in the Infinity Cache)
    double sumA = 0.0, sumB = 0.0;
    for (int j = 0; j < w; j += 8) {
        sk_d2v vv[8];
        double xa[8], xb[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const sk_d2v *vp = v2 + (j + u < w ? j + u : w - 1) * 64;
            if (ntv) vv[u] = __builtin_nontemporal_load(vp); else vv[u] = *vp;
        }
        if constexpr (WIDE) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {              // x[c], x[c + 1] in one 16-byte load at an 8-byte aligned address; the last column: x[c - 1], x[c]
                const int c = j + u < lenA ? rA + ptA[j + u < wmaxA ? j + u : wmaxA] : 0;
                const bool last = c >= xlast;          // (only a lane whose rows differ in pattern can sit there)
                const sk_d2v8 xx = *reinterpret_cast<const sk_d2v8 *>(a.x + (last ? xlast - 1 : c));
                xa[u] = last ? xx.y : xx.x; xb[u] = xx.y;
            }
            if (!same) {
#pragma unroll
                for (int u = 0; u < 8; ++u) xb[u] = a.x[j + u < lenB ? rB + ptB[j + u < wmaxB ? j + u : wmaxB] : 0];
            }
        } else if (same) {                             // (the 7-point level: 1 % of the lanes hold two patterns; this form needs 82 VGPRs, the one above 93)
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int c = rA + ptA[j + u < wmaxA ? j + u : wmaxA];
                const sk_d2v8 xx = *reinterpret_cast<const sk_d2v8 *>(a.x + (j + u < lenA ? c : 0));     // a 16-byte load at an 8-byte aligned address
                xa[u] = xx.x; xb[u] = xx.y;
            }
        } else {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int jj = j + u < wmaxA ? j + u : wmaxA;
                xa[u] = a.x[j + u < lenA ? rA + ptA[jj] : 0];
                xb[u] = a.x[j + u < lenB ? rB + ptB[jj] : 0];
            }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (j + u < lenA) sumA += vv[u].x * xa[u];
            if (j + u < lenB) sumB += vv[u].y * xb[u];
        }
    }
    if (rB < nrows) {                                              // rows 2 l and 2 l + 1: 16-byte loads and stores ...
        bool both = true;
        if constexpr (HALO)                                        // ... unless one of them is a boundary row (written by the halo stream's kernel)
            if (a.skip) both = ((a.skip[rA >> 5] >> (rA & 31)) & 3u) == 0u;      // rA is even: both bits sit in one word
        if (both) {
            if constexpr (PREF) {                                  // epilogue2's arithmetic on the operands fetched at the top
                sk_d2v o;
                if constexpr (EPI == EPI_RESIDUAL) { o.x = sumA - pre_b.x; o.y = sumB - pre_b.y; }
                else {
                    double t0 = sumA - pre_b.x, t1 = sumB - pre_b.y;
                    t0 *= pre_dg.x * a.c0; t1 *= pre_dg.y * a.c0;
                    o.x = pre_u.x - t0; o.y = pre_u.y - t1;
                }
                st_y2<NT>(a, a.y + rA, o);
            } else epilogue2<EPI, NT>(a, rA, sumA, sumB);
            return;
        }
    }
    if (rA < nrows) epilogue<EPI, HALO, NT>(a, rA, sumA);
    if (rB < nrows) epilogue<EPI, HALO, NT>(a, rB, sumB);
    }
}

// ---------------------------------------------------------------------------
// K1t: row templates -- k_sellp taken one step further, for operators whose rows repeat patterns AND values (constant-
// coefficient stencils on structured grids: the boundary-stripped 7-point Laplacian has 27 distinct rows, whatever its size).
// A template is (length, columns relative to the row, values); the table sits in LDS, a thread owns a row and reads NOTHING of
// the operator but its 16-bit template id: 2 B per row instead of 8 B per entry + 2 B per row.  Same products, same sequential
// sum as the CSR loop (the template's values are the row's values bit for bit): bit-identical results.
// OPT-IN (SAENA_ROW_TEMPLATES=1 adds it to the plan-time autotune): the contract's algorithmic bytes count 12 B per entry, which
// this kernel does not move -- its rate in those units says nothing about the memory system.
// a.dst: template ids, a.ptab: pt_n x (pt_w + 1) ints (length, offsets), a.val: pt_n x pt_w doubles (the templates' values).
template <int EPI, bool HALO, bool NT>
__global__ __launch_bounds__(BLOCK) void k_rowt(const SpmvArgs a, int nrows) {
    extern __shared__ double rt_lds[];
    if constexpr (HALO) fork_signal(a);
    double *vt = rt_lds;                                           // [pt_n][pt_w]
    int    *it = reinterpret_cast<int *>(rt_lds + (size_t)a.pt_n * a.pt_w);       // [pt_n][pt_w + 1]
    for (int i = threadIdx.x; i < a.pt_n * a.pt_w; i += BLOCK) vt[i] = a.val[i];
    for (int i = threadIdx.x; i < a.pt_n * (a.pt_w + 1); i += BLOCK) it[i] = a.ptab[i];
    __syncthreads();
    const int b = xcd_remap(blockIdx.x, (nrows + BLOCK - 1) / BLOCK);
    const int r = b * BLOCK + (int)threadIdx.x;
    if (r >= nrows) return;
    int pid;
    if constexpr (NT) pid = __builtin_nontemporal_load(a.dst + r); else pid = a.dst[r];
    const int *pt = it + pid * (a.pt_w + 1);
    const double *pv = vt + pid * a.pt_w;
    const int len = pt[0];
    double sum = 0.0;
    for (int j = 0; j < len; j += 8) {
        double xx[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {                              // (clamped table position, address 0 past the row's end: no branch per entry)
            const int jj = j + u < a.pt_w ? j + u : a.pt_w - 1;
            const int c = r + pt[1 + jj];
            xx[u] = a.x[j + u < len ? c : 0];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (j + u < len) sum += pv[j + u < a.pt_w ? j + u : a.pt_w - 1] * xx[u];
    }
    epilogue<EPI, HALO, NT>(a, r, sum);
}

// ---------------------------------------------------------------------------
// K1b: vector CSR (no LDS staging): G lanes own one row and stream it straight from
// global memory, 256/G rows per workgroup.  Rows are contiguous in val/col, so a
// wave still reads whole 128-B lines; there is no barrier and no LDS round trip.
template <int EPI, int G, bool HALO>
__global__ __launch_bounds__(BLOCK) void k_csr_vector(const SpmvArgs a, int nrows) {
    constexpr int RPB = BLOCK / G;
    const int nb = (nrows + RPB - 1) / RPB;
    if constexpr (HALO) fork_signal(a);
    const int b = xcd_remap(blockIdx.x, nb);
    const int tid = threadIdx.x;
    const int r = b * RPB + tid / G, l = tid % G;
    double sum = 0.0;
    if (r < nrows) {
        const int p0 = a.row_ptr[r], p1 = a.row_ptr[r + 1];
        int k = p0 + l;
        for (; k + G < p1; k += 2 * G) {              // two independent loads in flight per lane
            const double v0 = a.val[k], v1 = a.val[k + G];
            const int c0 = a.col[k], c1 = a.col[k + G];
            sum += v0 * a.x[c0];
            sum += v1 * a.x[c1];
        }
        if (k < p1) sum += a.val[k] * a.x[a.col[k]];
    }
    sum = group_sum<G>(sum);
    if (r < nrows && l == 0) {
        epilogue<EPI, HALO>(a, r, sum);
    }
}

// ---------------------------------------------------------------------------
// K1e: wave-streamed CSR for LONG rows (hundreds to thousands of entries): G lanes own one row and stream it
// straight from global memory with 16-byte loads (a lane takes 4 consecutive nnz, two quads in flight per lane),
// accumulate in registers and combine with shuffles -- no LDS staging, no barrier, so a CU keeps its full 32 waves
// in flight whatever the row length.  The row-block kernels above hold one 16/32 KiB tile per workgroup: a 2 000-entry
// row fills half a 32 KiB tile and only 4-5 such workgroups fit a CU, which leaves the memory pipe short of bytes in
// flight (measured on a 2 160-nnz/row box stencil: 4.65 TB/s against 5.9 TB/s for full tiles).
// Quads are aligned to 4 entries like in k_csr_stream; entries of the neighbouring rows that share a quad are
// multiplied by zero (their columns are valid, the arrays are padded by 8 zero entries).
template <int EPI, int G, bool HALO>
__global__ __launch_bounds__(BLOCK) void k_csr_wave(const SpmvArgs a, int nrows) {
    constexpr int RPB = BLOCK / G;
    const int nb = (nrows + RPB - 1) / RPB;
    if constexpr (HALO) fork_signal(a);
    const int b = xcd_remap(blockIdx.x, nb);
    const int tid = threadIdx.x;
    const int r = b * RPB + tid / G, l = tid % G;
    double sum = 0.0;
    if (r < nrows) {
        const int p0 = a.row_ptr[r], p1 = a.row_ptr[r + 1];
        const int a0 = p0 & ~3;
        const int nq = (p1 - a0 + 3) >> 2;
        for (int q = l; q < nq; q += 2 * G) {
            const int i0 = a0 + 4 * q;
            const bool two = q + G < nq;
            const int i1 = two ? i0 + 4 * G : i0;               // (a lane without a second quad re-reads its first)
            double2 v01 = ld_stream_d2(a.val + i0, a.nt);
            double2 v23 = ld_stream_d2(a.val + i0 + 2, a.nt);
            const int4 c = ld_stream_i4(a.col + i0, a.nt);
            double2 w01 = ld_stream_d2(a.val + i1, a.nt);
            double2 w23 = ld_stream_d2(a.val + i1 + 2, a.nt);
            const int4 d = ld_stream_i4(a.col + i1, a.nt);
            double x0 = a.x[c.x], x1 = a.x[c.y], x2 = a.x[c.z], x3 = a.x[c.w];
            double y0 = a.x[d.x], y1 = a.x[d.y], y2 = a.x[d.z], y3 = a.x[d.w];
            // entries of the neighbouring rows in the first / last quad, and the re-read quad of a lane without a second one:
            // BOTH factors are zeroed, so an inf or NaN of x at a column this row does not own stays out of its sum
            if (i0 < p0 || i0 + 4 > p1) {
                if (stray(i0, p0, p1)) { v01.x = 0.0; x0 = 0.0; }
                if (stray(i0 + 1, p0, p1)) { v01.y = 0.0; x1 = 0.0; }
                if (stray(i0 + 2, p0, p1)) { v23.x = 0.0; x2 = 0.0; }
                if (stray(i0 + 3, p0, p1)) { v23.y = 0.0; x3 = 0.0; }
            }
            if (!two) { w01.x = w01.y = w23.x = w23.y = 0.0; y0 = y1 = y2 = y3 = 0.0; }
            else if (i1 + 4 > p1) {
                if (stray(i1, p0, p1)) { w01.x = 0.0; y0 = 0.0; }
                if (stray(i1 + 1, p0, p1)) { w01.y = 0.0; y1 = 0.0; }
                if (stray(i1 + 2, p0, p1)) { w23.x = 0.0; y2 = 0.0; }
                if (stray(i1 + 3, p0, p1)) { w23.y = 0.0; y3 = 0.0; }
            }
            sum += v01.x * x0; sum += v01.y * x1; sum += v23.x * x2; sum += v23.y * x3;
            sum += w01.x * y0; sum += w01.y * y1; sum += w23.x * y2; sum += w23.y * y3;
        }
    }
    sum = group_sum<G>(sum);
    if (r < nrows && l == 0) {
        epilogue<EPI, HALO>(a, r, sum);
    }
}

// ---------------------------------------------------------------------------
// K1x: the input vector in LDS, for the coarse levels of hundreds to thousands of entries per row and the transfers
// into them.  There the k_csr_cc16 / k_csr_wave gathers go out to the L2 two to three times per line of stream data and
// the kernels sit at 45-60 % of the roofline.  Here one 1024-thread workgroup per CU owns an nnz-balanced chunk of
// consecutive rows (a.blk_row); the columns the chunk touches are cut into windows of XL_MAX columns (one window when
// the operator has no more columns than that: 256^3 L4, 20 K x 20 K; two for L3, whose rows reach over 26 K columns;
// at most XL_MAXT).  Per window the workgroup copies that piece of x into LDS (<= 158 KB, read from the L2), then G-lane
// groups take the chunk's rows from a shared counter and stream each row's entries of this window like k_csr_wave --
// 16-byte value loads, 8-byte loads of four 16-bit column ids RELATIVE TO THE WINDOW (10 B/nnz, no segment table), four
// quads in flight per lane -- and every gather is an LDS read.  A row's partial sums are added window by window in
// ascending column order (through `acc` in global memory when there is more than one window); the epilogue runs after
// the last.  tab: per workgroup (T+1) x rows entry offsets, window t of row k = [tab[t*rows + k], tab[(t+1)*rows + k]).
constexpr int XL_MAX   = 20224;            // doubles of x in LDS: 161 792 B of the CU's 163 840
constexpr int XL_MAXT  = 8;
constexpr int XL_BLOCK = 1024;
constexpr int XL_PER_CU = 1;              // (half windows around 512 threads, two workgroups per CU: slower on every level that uses them --
                                           //  256^3 L2 451 against 441 us, L3 324 / 313, L4 103 / 100, profiles/r03_xlds_half_windows.log)
struct XldsArgs {
    const int4 *info;      // per workgroup: first column, windows, offset of its table in `tab`, -
    const int  *tab;
    double     *acc;       // [M] partial row sums between windows (nullptr when every workgroup has one window, or with acc_lds)
    int         ncols;
    const int  *ord;       // [M] or nullptr: the order in which a chunk's groups take its rows (position in chunk -> row in chunk), LONGEST ROWS
                           // FIRST in chunks that hold rows many times the mean (an irregular operator's hub rows: started last, such a
                           // row is the tail of its workgroup -- and the workgroup the tail of the launch)
    int         win;       // columns per window (<= XL_MAX)
    int         acc_lds;   // the partial row sums of a chunk live in LDS behind the window: xs[win + row in chunk] (the plan keeps rows <= XL_MAX - win)
};
// RP > 1 (k_csr_xldsr, round 4): SHORT rows -- an irregular operator of a few dozen entries per row (BASELINE configs[4] scaled to
// 1 M rows: 13-205 entries, hubs of 3 000).  With one row per group step the kernel is bound by the LATENCY of a row, not by bytes:
// fetch the row's entry range -> its value / column quads -> the LDS gathers -> the shuffle sum -> the epilogue -> the counter, one
// after the other, ~2.5 us for a row whose 34 entries give a group of 8 lanes a single trip (111 us against a streaming ceiling of
// 55 us).  Here a group takes RP consecutive rows per step and streams their quads as ONE sequence (a quad belongs to exactly one of
// the rows; the lane keeps a sum per row), so four times the loads are in flight per trip and the fixed part of the chain is paid
// once per RP rows; lanes 0..RP-1 then run the RP epilogues side by side.  Same products; a row's entries are added lane by lane
// and then across the group like before, with another lane-to-quad mapping: deterministic, parity at the smoothers' tolerance.
template <int EPI, int G, bool HALO, int RP = 1>
__global__ __launch_bounds__(XL_BLOCK) void k_csr_xlds(const SpmvArgs a, const XldsArgs w) {
    __shared__ __attribute__((aligned(16))) double xs[XL_MAX];
    __shared__ int next_row;
    if constexpr (HALO) fork_signal(a);
    const int tid = threadIdx.x, lane = tid & 63, l = tid % G;
    constexpr int NG = XL_BLOCK / G;
    const int r0 = a.blk_row[blockIdx.x], nr = a.blk_row[blockIdx.x + 1] - r0;
    const int4 inf = w.info[blockIdx.x];
    const int T = inf.y;
    double *accs = xs + w.win;
    for (int t = 0; t < T; ++t) {
        if (t > 0) __syncthreads();                               // everyone is done with the previous window
        const int base = inf.x + t * w.win;
        const int n = w.ncols - base < w.win ? w.ncols - base : w.win;
        for (int i0 = tid; i0 < n; i0 += 8 * XL_BLOCK) {           // eight loads of a thread in flight before their stores
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = i0 + u * XL_BLOCK < n ? a.x[base + i0 + u * XL_BLOCK] : 0.0;
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (i0 + u * XL_BLOCK < n) xs[i0 + u * XL_BLOCK] = v[u];
        }
        if (tid == 0) next_row = NG;                              // the first NG rows go to the groups in order
        __syncthreads();
        const int *ts = w.tab + inf.z + t * nr, *te = ts + nr;
        if constexpr (RP > 1) {
            static_assert(G >= RP, "a lane per epilogue");
            if (tid == 0) next_row = NG * RP;                     // (rewritten before the barrier above took effect for RP == 1: set again here)
            __syncthreads();
            int k = (tid / G) * RP;
            int np0[RP], np1[RP], nrw[RP];                        // the NEXT step's rows (in chunk) and entry ranges, fetched one step ahead
#pragma unroll
            for (int j = 0; j < RP; ++j) {
                const bool have = k + j < nr;
                nrw[j] = have ? (w.ord ? w.ord[r0 + k + j] : k + j) : 0;
                np0[j] = have ? ts[nrw[j]] : 0; np1[j] = have ? te[nrw[j]] : 0;
            }
            while (k < nr) {
                int p0[RP], p1[RP], a0[RP], qe[RP], rw[RP];       // entry range, its quad-aligned start, end of the row's quads in the group's sequence
                int nq = 0;
#pragma unroll
                for (int j = 0; j < RP; ++j) {
                    p0[j] = np0[j]; p1[j] = np1[j]; rw[j] = nrw[j];
                    a0[j] = p0[j] & ~3;
                    nq += p1[j] > p0[j] ? (p1[j] - a0[j] + 3) >> 2 : 0;
                    qe[j] = nq;
                }
                // the group's next rows and their ranges NOW, so that the counter's round trip and the range loads run under this
                // step's stream instead of in front of the next one's (the chain of a step was: counter -> ranges -> quads -> sum -> epilogue)
                int nk = 0;
                if (l == 0) nk = atomicAdd(&next_row, RP);
                const int knext = __shfl(nk, lane & ~(G - 1), 64);
#pragma unroll
                for (int j = 0; j < RP; ++j) {
                    const bool have = knext + j < nr;
                    nrw[j] = have ? (w.ord ? w.ord[r0 + knext + j] : knext + j) : 0;
                    np0[j] = have ? ts[nrw[j]] : 0; np1[j] = have ? te[nrw[j]] : 0;
                }
                double sum[RP];
#pragma unroll
                for (int j = 0; j < RP; ++j) sum[j] = 0.0;
                for (int q = l; q < nq; q += 4 * G) {
                    double2 v01[4], v23[4];
                    uint2   c[4];
                    int     i[4], lo[4], hi[4], wj[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int qq = q + G * u;
                        int j = 0;
#pragma unroll
                        for (int t2 = 0; t2 < RP - 1; ++t2) j += qq >= qe[t2] ? 1 : 0;
                        int qb = 0, ab = a0[0]; lo[u] = p0[0]; hi[u] = p1[0];
#pragma unroll
                        for (int t2 = 1; t2 < RP; ++t2)
                            if (j == t2) { qb = qe[t2 - 1]; ab = a0[t2]; lo[u] = p0[t2]; hi[u] = p1[t2]; }
                        wj[u] = j;
                        i[u] = ab + 4 * (qq - qb);
                        v01[u].x = v01[u].y = v23[u].x = v23[u].y = 0.0;
                        c[u].x = c[u].y = 0u;
                        if (qq < nq) {
                            v01[u] = ld_stream_d2(a.val + i[u], a.nt);
                            v23[u] = ld_stream_d2(a.val + i[u] + 2, a.nt);
                            c[u]   = ld_stream_u2(a.ccol + i[u], a.nt);
                        } else wj[u] = -1;
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        double x0 = xs[c[u].x & 0xffffu], x1 = xs[c[u].x >> 16], x2 = xs[c[u].y & 0xffffu], x3 = xs[c[u].y >> 16];
                        if (wj[u] < 0) x0 = x1 = x2 = x3 = 0.0;
                        else if (i[u] < lo[u] || i[u] + 4 > hi[u]) {
                            x0 = stray(i[u], lo[u], hi[u]) ? 0.0 : x0;     x1 = stray(i[u] + 1, lo[u], hi[u]) ? 0.0 : x1;
                            x2 = stray(i[u] + 2, lo[u], hi[u]) ? 0.0 : x2; x3 = stray(i[u] + 3, lo[u], hi[u]) ? 0.0 : x3;
                        }
#pragma unroll
                        for (int j = 0; j < RP; ++j)
                            if (wj[u] == j) { sum[j] += v01[u].x * x0; sum[j] += v01[u].y * x1; sum[j] += v23[u].x * x2; sum[j] += v23[u].y * x3; }
                    }
                }
                double mine = 0.0;
#pragma unroll
                for (int j = 0; j < RP; ++j) {
                    const double sj = group_sum<G>(sum[j]);
                    if (l == j) mine = sj;
                }
                if (l < RP && k + l < nr) {                       // lanes 0 .. RP-1: an epilogue each
                    int rk = rw[0];
#pragma unroll
                    for (int j = 1; j < RP; ++j) if (l == j) rk = rw[j];
                    const int r = r0 + rk;
                    // the row's sum over the windows so far: in LDS behind the window, or in global memory
                    if (t > 0) mine = (w.acc_lds ? accs[rk] : w.acc[r]) + mine;
                    if (t < T - 1) { if (w.acc_lds) accs[rk] = mine; else w.acc[r] = mine; }
                    else epilogue<EPI, HALO>(a, r, mine);
                }
                k = knext;
            }
        } else {
        int k = tid / G;
        while (k < nr) {
            const int rk = w.ord ? w.ord[r0 + k] : k;
            const int p0 = ts[rk], p1 = te[rk];
            const int a0 = p0 & ~3;
            const int nq = (p1 - a0 + 3) >> 2;
            double sum = 0.0;
            for (int q = l; q < nq; q += 4 * G) {
                double2 v01[4], v23[4];
                uint2   c[4];
                int     i[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {                     // a lane without a u-th quad loads nothing (short row pieces: most of them)
                    i[u] = a0 + 4 * (q + G * u);
                    v01[u].x = v01[u].y = v23[u].x = v23[u].y = 0.0;
                    c[u].x = c[u].y = 0u;
                    if (q + G * u < nq) {
                        v01[u] = ld_stream_d2(a.val + i[u], a.nt);
                        v23[u] = ld_stream_d2(a.val + i[u] + 2, a.nt);
                        c[u]   = ld_stream_u2(a.ccol + i[u], a.nt);
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    double x0 = xs[c[u].x & 0xffffu], x1 = xs[c[u].x >> 16], x2 = xs[c[u].y & 0xffffu], x3 = xs[c[u].y >> 16];
                    if (u > 0 && q + G * u >= nq) x0 = x1 = x2 = x3 = 0.0;
                    else if (i[u] < p0 || i[u] + 4 > p1) {        // first / last quad: the neighbours' entries (other rows, other windows) count as zero
                        x0 = stray(i[u], p0, p1) ? 0.0 : x0;     x1 = stray(i[u] + 1, p0, p1) ? 0.0 : x1;
                        x2 = stray(i[u] + 2, p0, p1) ? 0.0 : x2; x3 = stray(i[u] + 3, p0, p1) ? 0.0 : x3;
                    }
                    sum += v01[u].x * x0; sum += v01[u].y * x1; sum += v23[u].x * x2; sum += v23[u].y * x3;
                }
            }
            sum = group_sum<G>(sum);
            int nk = 0;
            if (l == 0) {
                const int r = r0 + rk;
                if (t > 0) sum = (w.acc_lds ? accs[rk] : w.acc[r]) + sum;
                if (t < T - 1) { if (w.acc_lds) accs[rk] = sum; else w.acc[r] = sum; }
                else epilogue<EPI, HALO>(a, r, sum);
                nk = atomicAdd(&next_row, 1);
            }
            k = __shfl(nk, lane & ~(G - 1), 64);
        }
        }
    }
}

// ---------------------------------------------------------------------------
// K1y: sliced ELLPACK inside the (row chunk, column window) blocks of k_csr_xlds -- for the levels of a few hundred entries
// per row (256^3 L2: 263), where the tile kernels are either gather-bound (k_csr_cc16: 2.6 L1->L2 requests per line of stream)
// or carry 12 B per entry (k_csr_cm), and where k_csr_xlds's (row, window) pieces of ~66 entries are too short for a group
// of lanes each.  Here a LANE owns a row piece, as in k_sell: the pieces of one (chunk, window) block are sorted by length and
// packed 64 to a slice, position-major in pairs (16-byte value loads, 4-byte loads of two 16-bit window-relative column ids:
// 10 B per stored entry, ~3 % padding because each block sorts its own rows), the lane walks its piece and gathers from the
// window of x in LDS, and the partial sum of a row travels from window to window through `acc` (16 B per row and window
// against ~660 B of piece).  A row's pieces are added in ascending column order, window by window, each piece sequentially:
// deterministic; the order is k_csr_xlds's, not the sequential one.  Every row appears in its chunk's LAST window (with an
// empty piece if need be): that is where its epilogue runs.
// meta[slice * 64 + lane] = row in chunk (16 bits; 0xffff: no row) | piece length << 16 (15 bits) | first piece of the row << 31.
// pairs of positions a lane keeps in flight: 8 (426 us on 256^3 L2) against 4 (437 us), profiles/r03_sellx_steps.log
constexpr int SELLX_UP = 8;
struct SellxArgs {
    const int4           *info;      // per workgroup: first column, windows, -, -
    const int            *bptr;      // [chunks x (XL_MAXT + 1)]: first slice of block (chunk, window)
    const int            *sptr;      // [slices + 1] entry offsets (multiples of 64)
    const unsigned       *meta;
    const double         *val;
    const unsigned short *col;
    double               *acc;
    int                   ncols;
    int                   win, acc_lds;   // as in XldsArgs
};
template <int EPI, bool HALO>
__global__ __launch_bounds__(XL_BLOCK) void k_sellx(const SpmvArgs a, const SellxArgs w) {
    __shared__ __attribute__((aligned(16))) double xs[XL_MAX];
    if constexpr (HALO) fork_signal(a);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int NW = XL_BLOCK / 64;
    const int r0 = a.blk_row[blockIdx.x];
    const int4 inf = w.info[blockIdx.x];
    const int T = inf.y;
    for (int t = 0; t < T; ++t) {
        if (t > 0) __syncthreads();                               // everyone is done with the previous window
        const int base = inf.x + t * w.win;
        const int n = w.ncols - base < w.win ? w.ncols - base : w.win;
        for (int i0 = tid; i0 < n; i0 += 8 * XL_BLOCK) {           // eight loads of a thread in flight before their stores
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = i0 + u * XL_BLOCK < n ? a.x[base + i0 + u * XL_BLOCK] : 0.0;
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (i0 + u * XL_BLOCK < n) xs[i0 + u * XL_BLOCK] = v[u];
        }
        __syncthreads();
        const int s0 = w.bptr[blockIdx.x * (XL_MAXT + 1) + t], s1 = w.bptr[blockIdx.x * (XL_MAXT + 1) + t + 1];
        for (int s = s0 + wave; s < s1; s += NW) {
            const int p = w.sptr[s], P = (w.sptr[s + 1] - p) >> 7;           // pairs of positions
            const unsigned meta = w.meta[s * 64 + lane];
            const int len = (int)((meta >> 16) & 0x7fffu);
            const sk_d2v   *v2 = reinterpret_cast<const sk_d2v *>(w.val + p) + lane;
            const unsigned *c2 = reinterpret_cast<const unsigned *>(w.col + p) + lane;
            const unsigned row = meta & 0xffffu;
            // the row's partial sum of the windows before this one: fetched now, needed after the piece (a dependent L2 round
            // trip at the end of every slice otherwise)
            double prev = 0.0;
            if (row != 0xffffu && !(meta >> 31)) prev = w.acc_lds ? xs[w.win + (int)row] : w.acc[r0 + (int)row];
            constexpr int UP = SELLX_UP;
            double sum = 0.0;
            for (int q = 0; q < P; q += UP) {
                sk_d2v   vv[UP];
                unsigned cc[UP];
#pragma unroll
                for (int u = 0; u < UP; ++u) {
                    const int qq = q + u < P ? q + u : P - 1;
                    if (a.nt) { vv[u] = __builtin_nontemporal_load(v2 + qq * 64); cc[u] = __builtin_nontemporal_load(c2 + qq * 64); }
                    else { vv[u] = v2[qq * 64]; cc[u] = c2[qq * 64]; }
                }
#pragma unroll
                for (int u = 0; u < UP; ++u) {                   // padding is never added: a NaN or inf in x stays in the rows that own it
                    const double x0 = xs[cc[u] & 0xffffu], x1 = xs[cc[u] >> 16];
                    if (q + u < P && 2 * (q + u) < len) sum += vv[u].x * x0;
                    if (q + u < P && 2 * (q + u) + 1 < len) sum += vv[u].y * x1;
                }
            }
            if (row != 0xffffu) {
                const int r = r0 + (int)row;
                if (!(meta >> 31)) sum = prev + sum;
                if (t < T - 1) { if (w.acc_lds) xs[w.win + (int)row] = sum; else w.acc[r] = sum; }
                else epilogue<EPI, HALO>(a, r, sum);
            }
        }
    }
}

// ---------------------------------------------------------------------------
// K1c: dense row-major operator (saena_matrix_dense::matvec_dense, src/saena_matrix_dense.cpp:181-260,
// the reference's optional `switch_to_dense` storage for coarse levels that are mostly full).  One wave per row streams
// the row with coalesced 8-byte loads (no column ids: 8 B per entry instead of 12), the fused epilogues are the sparse ones.
template <int EPI>
__global__ __launch_bounds__(BLOCK) void k_dense_rows(const SpmvArgs a, const double *__restrict__ dense, int nrows, int ncols) {
    const int r = blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6), l = threadIdx.x & 63;
    double sum = 0.0;
    if (r < nrows) {
        const double *row = dense + (size_t)r * ncols;
        for (int j = l; j < ncols; j += 64) sum += row[j] * a.x[j];
    }
    sum = group_sum<64>(sum);
    if (r < nrows && l == 0) epilogue<EPI, false>(a, r, sum);
}

// K1c with a halo: the row-partitioned dense operator.  The reference passes the x blocks round a ring, one
// send/recv per step (matvec_dense, src/saena_matrix_dense.cpp:181-260); over xGMI one neighbour exchange costs ~20 us
// whatever its size, so here every rank receives the blocks it needs in ONE grouped exchange (the operator's ordinary
// halo plan) and then multiplies its dense rows: `dense` holds the columns this rank owns, `dense_rem` the columns of
// the halo buffer (receive order).  fp32 wire: like matvec_dense_float (:262-340) EVERY block of x is rounded to
// float, the rank's own included (the sparse float form rounds the halo only).
template <int EPI>
__global__ __launch_bounds__(BLOCK) void k_dense_rows_halo(const SpmvArgs a, const double *__restrict__ dense, const double *__restrict__ dense_rem,
                                                           int nrows, int ncols, int nhalo, const double *__restrict__ halo,
                                                           const float *__restrict__ halo_f, int x_float) {
    const int r = blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6), l = threadIdx.x & 63;
    double sum = 0.0;
    if (r < nrows) {
        const double *row = dense + (size_t)r * ncols;
        if (x_float) { for (int j = l; j < ncols; j += 64) sum += row[j] * (double)(float)a.x[j]; }
        else         { for (int j = l; j < ncols; j += 64) sum += row[j] * a.x[j]; }
        const double *rrow = dense_rem + (size_t)r * nhalo;
        if (halo_f) { for (int j = l; j < nhalo; j += 64) sum += rrow[j] * (double)halo_f[j]; }
        else        { for (int j = l; j < nhalo; j += 64) sum += rrow[j] * halo[j]; }
    }
    sum = group_sum<64>(sum);
    if (r < nrows && l == 0) epilogue<EPI, false>(a, r, sum);
}

// ---------------------------------------------------------------------------
// K2: boundary rows.  Rows that own remote entries are left out of the interior launch (SpmvArgs::skip)
// and computed whole by this kernel on the halo stream, after the exchange: G lanes own one row, add its
// local products (x) and then its remote products (halo buffer, ascending receive position = the order of
// the reference's remote loop, src/saena_matrix_matvec.cpp:93-109) and apply the epilogue once.  With
// G = 1 that is the reference's sum: the local loop first, the remote contributions after it.  Interior
// and boundary launches touch disjoint rows, so they need no ordering between the two streams.
struct BoundaryArgs {
    SpmvArgs      s;         // the local CSR (row_ptr/col/val), x and the epilogue operands; skip == nullptr
    const int    *rows;      // [nrows] boundary rows, ascending
    int           nrows;
    const int    *h_ptr;     // [nrows+1] CSR over the halo buffer, one row per boundary row
    const int    *h_col;     // position in the receive buffer
    const double *h_val;
    const double *halo;      // receive buffer (fp64 wire)
    const float  *halo_f;    // receive buffer of the fp32 wire (matvec_sparse_float), or nullptr
};

template <int EPI, int G>
__global__ __launch_bounds__(BLOCK) void k_csr_boundary(const BoundaryArgs b) {
    constexpr int RPB = BLOCK / G;
    const int i = blockIdx.x * RPB + threadIdx.x / G, l = threadIdx.x % G;
    double sum = 0.0;
    int r = 0;
    if (i < b.nrows) {
        r = b.rows[i];
        const int p1 = b.s.row_ptr[r + 1];
        for (int k = b.s.row_ptr[r] + l; k < p1; k += G) sum += b.s.val[k] * b.s.x[b.s.col[k]];
        const int q1 = b.h_ptr[i + 1];
        if (b.halo_f) {
            for (int k = b.h_ptr[i] + l; k < q1; k += G) sum += b.h_val[k] * (double)b.halo_f[b.h_col[k]];
        } else {
            for (int k = b.h_ptr[i] + l; k < q1; k += G) sum += b.h_val[k] * b.halo[b.h_col[k]];
        }
    }
    sum = group_sum<G>(sum);
    if (i < b.nrows && l == 0) epilogue<EPI, false>(b.s, r, sum);
}

// ---------------------------------------------------------------------------
// K3: halo pack, vSend[i] = v[vIndex[i]] (optionally rounded through float, the reference's
// matvec_sparse_float halo).  Grid-stride over at most PACK_MAX_BLOCKS blocks: with `flag` set every block
// first waits for *flag >= seq (the interior launch has started, i.e. v is final), so the number of
// blocks that can sit waiting on the GPU must stay far below what the chip can hold.
constexpr int PACK_MAX_BLOCKS = 128;
__global__ __launch_bounds__(BLOCK) void k_pack(const double *v, const int *__restrict__ vIndex, double *__restrict__ send, int n,
                                                int as_float, const uint64_t *flag, uint64_t seq) {
    block_wait_flag(flag, seq);
    for (int i = blockIdx.x * BLOCK + threadIdx.x; i < n; i += gridDim.x * BLOCK) {
        double t = flag ? coherent_load(v + vIndex[i]) : v[vIndex[i]];
        if (as_float) t = (double)(float)t;
        send[i] = t;
    }
}

// fp32 halo on the wire (matvec_sparse_float, saena_matrix_matvec.cpp:464,531,538): pack straight to float;
// k_csr_boundary widens the received floats as it reads them
__global__ __launch_bounds__(BLOCK) void k_pack_f32(const double *v, const int *__restrict__ vIndex, float *__restrict__ send, int n,
                                                    const uint64_t *flag, uint64_t seq) {
    block_wait_flag(flag, seq);
    for (int i = blockIdx.x * BLOCK + threadIdx.x; i < n; i += gridDim.x * BLOCK)
        send[i] = (float)(flag ? coherent_load(v + vIndex[i]) : v[vIndex[i]]);
}

// one-wave helpers.  k_flag_set follows the boundary launch on the halo stream: the kernel boundary releases the
// boundary rows once (a last-block ticket inside k_csr_boundary needs an L2 write-back per block: 40 us for 992
// blocks, measured).  k_flag_wait serves a receive-only rank, which has no pack launch to do the waiting.
__global__ void k_flag_wait(const uint64_t *flag, uint64_t seq) {
    if (threadIdx.x == 0)
        while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < seq) __builtin_amdgcn_s_sleep(32);
}
__global__ void k_flag_set(uint64_t *flag, uint64_t seq) {
    if (threadIdx.x == 0) (void)__hip_atomic_exchange(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

// ---------------------------------------------------------------------------
// K10: streaming vector kernels (16 B per lane, grid-stride)
__global__ __launch_bounds__(BLOCK) void k_fill(double *__restrict__ y, double a, size_t n) {
    const size_t stride = (size_t)gridDim.x * BLOCK;
    const size_t n2 = n >> 1;
    double2 v; v.x = a; v.y = a;
    for (size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x; i < n2; i += stride) reinterpret_cast<double2 *>(y)[i] = v;
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) y[n - 1] = a;
}

// y = a*x + b*y   (b == 0 never reads y)
__global__ __launch_bounds__(BLOCK) void k_axpby(double a, const double *__restrict__ x, double b, double *__restrict__ y, size_t n) {
    const size_t stride = (size_t)gridDim.x * BLOCK;
    const size_t n2 = n >> 1;
    for (size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x; i < n2; i += stride) {
        const double2 xv = reinterpret_cast<const double2 *>(x)[i];
        double2 yv;
        if (b == 0.0) { yv.x = a * xv.x; yv.y = a * xv.y; }
        else { yv = reinterpret_cast<double2 *>(y)[i]; yv.x = a * xv.x + b * yv.x; yv.y = a * xv.y + b * yv.y; }
        reinterpret_cast<double2 *>(y)[i] = yv;
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) y[n - 1] = (b == 0.0) ? a * x[n - 1] : a * x[n - 1] + b * y[n - 1];
}

// pCG fused update (saena_object_solve.cpp:2593-2596): u -= alpha p ; r -= alpha h
__global__ __launch_bounds__(BLOCK) void k_pcg_update(double alpha, const double *__restrict__ p, const double *__restrict__ h,
                                                      double *__restrict__ u, double *__restrict__ r, size_t n) {
    const size_t stride = (size_t)gridDim.x * BLOCK;
    for (size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += stride) {
        u[i] -= alpha * p[i];
        r[i] -= alpha * h[i];
    }
}

// First smoother sweep from a ZERO iterate (every coarse level of a V-cycle starts from u = 0, and so does the fine
// level when the V-cycle preconditions CG): A*0 contributes nothing, so the sweep needs no pass over the matrix.
// Same arithmetic as the epilogues with s = 0 and u = 0:  Jacobi  y = 0 - ((0 - rhs) * (inv_diag*omega)) = rhs*(inv_diag*omega);
// Chebyshev step 0  d = (c*inv_diag)*(rhs - 0), y = 0 + d  -- bit-identical results (up to the sign of a zero).
__global__ __launch_bounds__(BLOCK) void k_zero_sweep(int cheby, double c0, const double *__restrict__ rhs, const double *__restrict__ inv_diag,
                                                      double *__restrict__ y, double *__restrict__ d, size_t n) {
    const size_t stride = (size_t)gridDim.x * BLOCK;
    for (size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += stride) {
        if (cheby) {
            const double dd = (c0 * inv_diag[i]) * rhs[i];
            d[i] = dd;
            y[i] = dd;
        } else {
            y[i] = rhs[i] * (inv_diag[i] * c0);
        }
    }
}

// pCG with the scalars kept on the device (no host round trip between the dots and the update):
// alpha = S[ia] / S[ib]; u -= alpha p; r -= alpha h; partial[block] = sum over the block's elements of the
// NEW r*r, with k_dot_partial's element-to-thread assignment (launch it with the dot's grid).
__device__ __forceinline__ double block_sum(double s, double *sh);
__global__ __launch_bounds__(BLOCK) void k_pcg_update_dev(const double *__restrict__ S, int ia, int ib, const double *__restrict__ p,
                                                          const double *__restrict__ h, double *__restrict__ u, double *__restrict__ r,
                                                          size_t n, double *__restrict__ partial) {
    __shared__ double sh[BLOCK / 64];
    const double alpha = S[ia] / S[ib];
    const size_t stride = (size_t)gridDim.x * BLOCK;
    double s = 0.0;
    for (size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += stride) {
        u[i] -= alpha * p[i];
        const double ri = r[i] - alpha * h[i];
        r[i] = ri;
        s += ri * ri;
    }
    const double t = block_sum(s, sh);
    if (threadIdx.x == 0) partial[blockIdx.x] = t;
}
// beta = S[ia] / S[ib]; p = z + beta p        (saena_object_solve.cpp:2655-2667; z = rho, or r for plain CG)
__global__ __launch_bounds__(BLOCK) void k_pcg_direction_dev(const double *__restrict__ S, int ia, int ib, const double *__restrict__ z,
                                                             double *__restrict__ p, size_t n) {
    const double beta = S[ia] / S[ib];
    const size_t stride = (size_t)gridDim.x * BLOCK;
    for (size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += stride) p[i] = 1.0 * z[i] + beta * p[i];
}

// K9: dot product, stage 1: per-block partial (wave shuffle + LDS), stage 2 by k_reduce_partials
__device__ __forceinline__ double block_sum(double s, double *sh) {
    s = group_sum<64>(s);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    double t = 0.0;
    if (threadIdx.x == 0) {
#pragma unroll
        for (int w = 0; w < BLOCK / 64; ++w) t += sh[w];
    }
    __syncthreads();
    return t;   // valid in thread 0
}

__global__ __launch_bounds__(BLOCK) void k_dot_partial(const double *__restrict__ x, const double *__restrict__ y, size_t n,
                                                       double *__restrict__ partial) {
    __shared__ double sh[BLOCK / 64];
    const size_t stride = (size_t)gridDim.x * BLOCK;
    double s = 0.0;
    for (size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += stride) s += x[i] * y[i];
    const double t = block_sum(s, sh);
    if (threadIdx.x == 0) partial[blockIdx.x] = t;
}

// out[0] = sum(partial[0..np))  (single block; fixed order => run-to-run deterministic)
__global__ __launch_bounds__(BLOCK) void k_reduce_partials(const double *__restrict__ partial, int np, double *__restrict__ out) {
    __shared__ double sh[BLOCK / 64];
    double s = 0.0;
    for (int i = threadIdx.x; i < np; i += BLOCK) s += partial[i];
    const double t = block_sum(s, sh);
    if (threadIdx.x == 0) out[0] = t;
}

// ---------------------------------------------------------------------------
// Coarsest-level CG (solve_coarsest_CG, saena_object_solve.cpp:14-114) as ONE
// workgroup: the <= ~100-row system lives in LDS for the whole solve, so the
// up-to-150 iterations cost no kernel launches and no HBM traffic.
// Single-rank form (the coarsest operator is replicated on every rank).
constexpr int CG_MAXN   = 1024;     // rows supported by the LDS-resident solver
constexpr int CG_BLOCK  = 256;

struct CoarseCGArgs {
    const int    *row_ptr;
    const int    *col;
    const double *val;
    int           n;
    const double *rhs;
    double       *u;          // in/out (caller zeroes it, as vcycle does)
    int           max_iter;
    double        tol;
    int          *iters_out;  // may be nullptr
};

__device__ __forceinline__ double cg_block_dot(const double *a, const double *b, int n, double *sh) {
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += CG_BLOCK) s += a[i] * b[i];
    s = group_sum<64>(s);
    __syncthreads();                       // sh reuse
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    double t = 0.0;
#pragma unroll
    for (int w = 0; w < CG_BLOCK / 64; ++w) t += sh[w];
    return t;                              // every thread holds the same value
}

__global__ __launch_bounds__(CG_BLOCK) void k_coarse_cg(const CoarseCGArgs a) {
    __shared__ double res[CG_MAXN], dir[CG_MAXN], mt[CG_MAXN], uu[CG_MAXN];
    __shared__ double sh[CG_BLOCK / 64];
    const int n = a.n, tid = threadIdx.x;
    for (int i = tid; i < n; i += CG_BLOCK) { res[i] = a.rhs[i]; dir[i] = a.rhs[i]; uu[i] = a.u[i]; }
    __syncthreads();
    const double initial_dot = cg_block_dot(res, res, n, sh);
    const double thres = initial_dot * a.tol * a.tol;
    double dot = initial_dot;
    int max_iter = a.max_iter;
    if (dot < a.tol * a.tol) max_iter = 0;
    int i = 1;
    constexpr int G = 8;                    // lanes per row in the LDS matvec
    while (i < max_iter) {
        __syncthreads();
        for (int r = tid / G; r < n; r += CG_BLOCK / G) {       // mt = A dir
            double s = 0.0;
            for (int k = a.row_ptr[r] + (tid % G); k < a.row_ptr[r + 1]; k += G) s += a.val[k] * dir[a.col[k]];
            s = group_sum<G>(s);
            if ((tid % G) == 0) mt[r] = s;
        }
        __syncthreads();
        double factor = cg_block_dot(dir, mt, n, sh);
        factor = dot / factor;
        for (int j = tid; j < n; j += CG_BLOCK) { uu[j] += factor * dir[j]; res[j] -= factor * mt[j]; }
        __syncthreads();
        const double dot_prev = dot;
        dot = cg_block_dot(res, res, n, sh);
        if (dot < thres) break;
        factor = dot / dot_prev;
        for (int j = tid; j < n; j += CG_BLOCK) dir[j] = res[j] + factor * dir[j];
        i++;
    }
    __syncthreads();
    for (int j = tid; j < n; j += CG_BLOCK) a.u[j] = uu[j];
    if (i == max_iter && max_iter != 0) i--;
    if (tid == 0 && a.iters_out) *a.iters_out = i;
}

// Diagnostic (not on any product path): cost of the x[col] gather alone under two lane->nnz mappings.
// mode 0: a lane owns 4 consecutive nnz (the production mapping; one gather instruction = a stride-4 comb
// over 256 nnz); mode 1: lane l of a wave takes nnz base + e*64 + l (one instruction = 64 consecutive nnz).
// Measured on the 67-nnz/row level of the 128^3 hierarchy: 150 us / 134 us for the gather alone, while the whole
// operator streams in ~115 us: the texture addresser retires about one scattered 8-byte lane per clock per CU.
__global__ __launch_bounds__(BLOCK) void k_gather_probe(const int *__restrict__ col, const double *__restrict__ x,
                                                        double *__restrict__ out, long nnz, int mode) {
    const long chunk = (long)blockIdx.x * (BLOCK * 4);          // 1024 nnz per block
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    double s = 0.0;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const long k = mode == 0 ? chunk + 4L * tid + e : chunk + 256L * wave + 64L * e + lane;
        if (k < nnz) s += x[col[k]];
    }
    if (s == 1.2345e-300) out[0] = s;
}

// Coarsest-level direct solve: u = Ainv rhs with the dense inverse computed once on the host
// (the reference factors the coarsest operator with SuperLU_DIST at setup and solves every V-cycle,
// src/saena_object_solve.cpp:793-958).  One workgroup; a wave per row, lanes over columns.
__global__ __launch_bounds__(CG_BLOCK) void k_dense_solve(const double *__restrict__ Ainv, const double *__restrict__ rhs,
                                                          double *__restrict__ u, int n) {
    __shared__ double r[CG_MAXN];
    for (int i = threadIdx.x; i < n; i += CG_BLOCK) r[i] = rhs[i];
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int i = wave; i < n; i += CG_BLOCK / 64) {
        double s = 0.0;
        for (int j = lane; j < n; j += 64) s += Ainv[(size_t)i * n + j] * r[j];
        s = group_sum<64>(s);
        if (lane == 0) u[i] = s;
    }
}


// ---- setup: CSR order -> sliced-ELLPACK order, on the device (round 4) ----
// The plan-time builds of k_sell / k_sellp / k_sellp2 re-ordered the values on the host and uploaded the copy (1.0 s for the
// 558 M entries of 256^3 level 1, most of a cold "upload").  The CSR values are on the device already: a thread per row copies its
// entries to their place in the slice layout (reads walk the row, writes of a wave are coalesced).  src is any per-entry array in
// CSR order (values; 16-bit column codes); dst is zero-filled (padding).  mode 0: slices of 64 rows, position-major; 1: the same
// in pairs of positions per lane (an odd last position alone); 2: slices of 128 rows, position-major (k_sellp2's row pairs).
template <class T>
__global__ __launch_bounds__(BLOCK) void k_sell_scatter(const T *__restrict__ src, const int *__restrict__ row_ptr, const int *__restrict__ sl_ptr,
                                                       T *__restrict__ dst, int M, int mode, const int *__restrict__ perm = nullptr) {
    const int r = blockIdx.x * BLOCK + threadIdx.x;      // the slice position; perm (k_sell<sorted>): the row that sits there
    if (r >= M) return;
    const int rsrc = perm ? perm[r] : r;
    const size_t q0 = (size_t)row_ptr[rsrc];
    const int n = row_ptr[rsrc + 1] - row_ptr[rsrc];
    if (mode == 2) {
        const int s = r >> 7, l = r & 127;
        const size_t p = (size_t)sl_ptr[s];
        for (int j = 0; j < n; ++j) dst[p + (size_t)j * 128 + l] = src[q0 + j];
    } else {
        const int s = r >> 6, l = r & 63;
        const size_t p = (size_t)sl_ptr[s];
        const int w = (sl_ptr[s + 1] - sl_ptr[s]) >> 6, PP = w >> 1;
        for (int j = 0; j < n; ++j) {
            const size_t o = mode == 0 ? p + (size_t)j * 64 + l : j < 2 * PP ? p + (size_t)(j >> 1) * 128 + (size_t)l * 2 + (j & 1) : p + (size_t)PP * 128 + l;
            dst[o] = src[q0 + j];
        }
    }
}

// ---- setup: the 16-bit column codes of k_csr_cc16 / k_sell, on the device (round 4) ----
// Per row block [blk[b], blk[b + 1]) a table of the distinct column segments (col >> ob) it touches, ascending, and per entry
// (slot in the table << ob) | (col & (2^ob - 1)).  The host did this with a few threads and uploaded 2 B per entry (1.1 s of the
// 256^3 hierarchy's cold upload); the 32-bit columns are on the device already.  One workgroup per block: the block's segments as a
// bitmap in LDS (CC_BM_WORDS x 32 segments cover every operator of up to 16.7 M columns at 8 offset bits), counted (pass 1: the
// host sums the counts into segptr and sees whether every block fits 2^(16-ob) slots) and then numbered by a prefix sum over the
// bitmap's words (pass 2: ascending segment order = the host encoder's sorted table, so both encoders produce the same arrays).
constexpr int CC_BM_WORDS = 2048;
// perm != nullptr (k_sell<sorted>): block b covers the slice positions [blk[b], blk[b + 1]), position q holds row perm[q] -- its entries
// are not one contiguous range, so the threads walk rows instead of entries
__global__ __launch_bounds__(BLOCK) void k_cc16_count(const int *__restrict__ col, const int *__restrict__ row_ptr, const int *__restrict__ blk,
                                                     int ob, int nwords, int maxseg, int *__restrict__ cnt, int *__restrict__ bad,
                                                     const int *__restrict__ perm = nullptr) {
    __shared__ unsigned bm[CC_BM_WORDS];
    __shared__ int total;
    const int b = blockIdx.x;
    for (int w = threadIdx.x; w < nwords; w += BLOCK) bm[w] = 0u;
    if (threadIdx.x == 0) total = 0;
    __syncthreads();
    if (perm) {
        for (int q = blk[b] + (int)threadIdx.x; q < blk[b + 1]; q += BLOCK) {
            const int r = perm[q];
            for (int p = row_ptr[r]; p < row_ptr[r + 1]; ++p) { const int sg = col[p] >> ob; atomicOr(&bm[sg >> 5], 1u << (sg & 31)); }
        }
    } else {
        const int p0 = row_ptr[blk[b]], p1 = row_ptr[blk[b + 1]];
        for (int p = p0 + (int)threadIdx.x; p < p1; p += BLOCK) {
            const int sg = col[p] >> ob;
            atomicOr(&bm[sg >> 5], 1u << (sg & 31));
        }
    }
    __syncthreads();
    int mine = 0;
    for (int w = threadIdx.x; w < nwords; w += BLOCK) mine += __popc(bm[w]);
    if (mine) atomicAdd(&total, mine);
    __syncthreads();
    if (threadIdx.x == 0) { cnt[b] = total; if (total > maxseg) atomicOr(bad, 1); }
}
__global__ __launch_bounds__(BLOCK) void k_cc16_encode(const int *__restrict__ col, const int *__restrict__ row_ptr, const int *__restrict__ blk,
                                                      int ob, int nwords, const int *__restrict__ segptr, int *__restrict__ segtab,
                                                      unsigned short *__restrict__ ccol, const int *__restrict__ perm = nullptr) {
    __shared__ unsigned bm[CC_BM_WORDS];
    __shared__ int pre[CC_BM_WORDS];                     // set bits in the words before this one
    __shared__ int part[BLOCK];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int p0 = perm ? 0 : row_ptr[blk[b]], p1 = perm ? 0 : row_ptr[blk[b + 1]];
    for (int w = tid; w < nwords; w += BLOCK) bm[w] = 0u;
    __syncthreads();
    if (perm) {
        for (int q = blk[b] + tid; q < blk[b + 1]; q += BLOCK) {
            const int r = perm[q];
            for (int p = row_ptr[r]; p < row_ptr[r + 1]; ++p) { const int sg = col[p] >> ob; atomicOr(&bm[sg >> 5], 1u << (sg & 31)); }
        }
    } else {
        for (int p = p0 + tid; p < p1; p += BLOCK) {
            const int sg = col[p] >> ob;
            atomicOr(&bm[sg >> 5], 1u << (sg & 31));
        }
    }
    __syncthreads();
    // exclusive prefix of the words' population counts: a thread owns a contiguous run of words
    const int per = (nwords + BLOCK - 1) / BLOCK, w0 = tid * per, w1 = w0 + per < nwords ? w0 + per : nwords;
    int run = 0;
    for (int w = w0; w < w1; ++w) run += __popc(bm[w]);
    part[tid] = run;
    __syncthreads();
    if (tid == 0) { int acc = 0; for (int t = 0; t < BLOCK; ++t) { const int v = part[t]; part[t] = acc; acc += v; } }
    __syncthreads();
    run = part[tid];
    for (int w = w0; w < w1; ++w) { pre[w] = run; run += __popc(bm[w]); }
    __syncthreads();
    const int t0 = segptr[b];
    for (int w = tid; w < nwords; w += BLOCK) {           // the table: segment bases in ascending order
        unsigned m = bm[w];
        int k = pre[w];
        while (m) { const int bit = __ffs(m) - 1; m &= m - 1; segtab[t0 + k++] = ((w << 5) + bit) << ob; }
    }
    const int om = (1 << ob) - 1;
    if (perm) {
        for (int q = blk[b] + tid; q < blk[b + 1]; q += BLOCK) {
            const int r = perm[q];
            for (int p = row_ptr[r]; p < row_ptr[r + 1]; ++p) {
                const int c = col[p], sg = c >> ob;
                const int slot = pre[sg >> 5] + __popc(bm[sg >> 5] & ((1u << (sg & 31)) - 1u));
                ccol[p] = (unsigned short)((slot << ob) | (c & om));
            }
        }
    } else {
        for (int p = p0 + tid; p < p1; p += BLOCK) {
            const int c = col[p], sg = c >> ob;
            const int slot = pre[sg >> 5] + __popc(bm[sg >> 5] & ((1u << (sg & 31)) - 1u));
            ccol[p] = (unsigned short)((slot << ob) | (c & om));
        }
    }
}

// ---- setup: the x-in-LDS plan's window-relative 16-bit columns and (row, window) entry offsets, on the device (round 4) ----
// One workgroup per row chunk, a thread per row: the columns ascend along a row, so the chunk's windows [cmin + w XL_MAX, ...) cut it
// into consecutive pieces; entry p gets its column relative to its window, tab[w * rows + k] the first entry of row k in window w
// (tab[T * rows + k] = the row's end).  The host did this with a few threads and uploaded 2 B per entry.
__global__ __launch_bounds__(BLOCK) void k_xlds_build(const int *__restrict__ col, const int *__restrict__ row_ptr, const int *__restrict__ blk,
                                                     const int4 *__restrict__ info, int *__restrict__ tab, unsigned short *__restrict__ xcol, int window) {
    const int b = blockIdx.x;
    const int r0 = blk[b], rows = blk[b + 1] - r0;
    const int4 inf = info[b];
    const int cmin = inf.x, T = inf.y;
    int *tb = tab + inf.z;
    for (int k = threadIdx.x; k < rows; k += BLOCK) {
        const int p0 = row_ptr[r0 + k], p1 = row_ptr[r0 + k + 1];
        int p = p0;
        for (int w = 0; w <= T; ++w) {
            if (w == T) { tb[(size_t)w * rows + k] = p1; break; }
            const int lim = cmin + w * window;                    // entries below lim belong to window w - 1 (none for w = 0: cmin is the chunk's smallest column)
            while (p < p1 && col[p] < lim) { xcol[p] = (unsigned short)(col[p] - (cmin + (w - 1) * window)); ++p; }
            tb[(size_t)w * rows + k] = p;
        }
        for (; p < p1; ++p) xcol[p] = (unsigned short)(col[p] - (cmin + (T - 1) * window));      // the last window takes the rest
    }
}

// ---- setup: k_csr_cm's column-ordered blocks, on the device (round 4) ----
// One workgroup per row block: the block's entries (<= CAPV) sorted by (column, CSR position) -- a bitonic sort of 64-bit keys in LDS --
// and written out in that order: value, 16-bit column code (the k_csr_cc16 code of the same entry), the tile slot its product belongs
// to (its CSR position relative to the block's quad-aligned start).  Padding up to whole quads: value 0 (the arrays are zero-filled),
// the first entry's column code, the spare slot CAPV + 4.  The host did this with std::stable_sort per block and uploaded 12 B per entry.
template <int CAPV>
__global__ __launch_bounds__(BLOCK) void k_cm_build(const double *__restrict__ val, const int *__restrict__ col, const unsigned short *__restrict__ ccol,
                                                   const int *__restrict__ row_ptr, const int *__restrict__ blk, const int *__restrict__ cmptr,
                                                   double *__restrict__ cm_val, unsigned short *__restrict__ cm_col, unsigned short *__restrict__ cm_dst) {
    __shared__ unsigned long long key[CAPV];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int p0 = row_ptr[blk[b]], n = row_ptr[blk[b + 1]] - p0, a0 = p0 & ~3;
    int N = 1;
    while (N < n) N <<= 1;
    for (int i = tid; i < N; i += BLOCK)
        key[i] = i < n ? ((unsigned long long)(unsigned)col[p0 + i] << 32) | (unsigned)i : ~0ull;
    __syncthreads();
    for (int k = 2; k <= N; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = tid; t < (N >> 1); t += BLOCK) {
                const int lo = ((t & ~(j - 1)) << 1) | (t & (j - 1)), hi = lo | j;      // the pair (lo, lo + j) of this stage
                const bool up = (lo & k) == 0;
                const unsigned long long x = key[lo], y = key[hi];
                if ((x > y) == up) { key[lo] = y; key[hi] = x; }
            }
            __syncthreads();
        }
    const size_t o = (size_t)cmptr[b];
    const int npad = (n + 3) & ~3;
    const unsigned short first = n ? ccol[p0 + (int)(key[0] & 0xffffffffu)] : (unsigned short)0;
    for (int i = tid; i < npad; i += BLOCK) {
        if (i < n) {
            const int p = p0 + (int)(key[i] & 0xffffffffu);
            cm_val[o + i] = val[p];
            cm_col[o + i] = ccol[p];
            cm_dst[o + i] = (unsigned short)(p - a0);
        } else {
            cm_col[o + i] = first;
            cm_dst[o + i] = (unsigned short)(CAPV + 4);
        }
    }
}

// ---- the streaming ceiling of a byte mix (bench.py's `roofline.peak_measured`) ----
// What the memory system of THIS device gives a kernel that moves the same bytes as an operator's sweep and does nothing else:
// per written double a wave-coalesced run of `q` 16-byte loads per lane (the value stream: 1 KiB per wave instruction) and one
// 8-byte store.  No gathers, no index arithmetic, no LDS: a real SpMV cannot be faster than this on the same bytes, so
// time(ceiling) / time(kernel) is a roofline fraction that is <= 1 by construction whether the working set sits in the
// 256 MiB Infinity Cache or in HBM (the spec-peak fraction of a cache-resident operator is not: 1.10 in round 3).
// MODE bit 0: non-temporal loads, bit 1: non-temporal stores -- the host keeps the fastest of the four.
typedef double ceil_d2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(BLOCK) void k_stream_ceiling(const ceil_d2 *__restrict__ rd, double *__restrict__ wr, size_t n_w, int q) {
    const size_t t = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (t >= n_w) return;
    const size_t wave = t >> 6;
    const int lane = threadIdx.x & 63;
    const ceil_d2 *p = rd + wave * 64 * (size_t)q + lane;
    ceil_d2 a0 = {0.0, 0.0}, a1 = a0, a2 = a0, a3 = a0;
    int i = 0;
    for (; i + 4 <= q; i += 4) {                         // four loads in flight per lane and trip, like the kernels' unrolled position loops
        ceil_d2 v0, v1, v2, v3;
        if constexpr (MODE & 1) {
            v0 = __builtin_nontemporal_load(p + (size_t)i * 64); v1 = __builtin_nontemporal_load(p + (size_t)(i + 1) * 64);
            v2 = __builtin_nontemporal_load(p + (size_t)(i + 2) * 64); v3 = __builtin_nontemporal_load(p + (size_t)(i + 3) * 64);
        } else {
            v0 = p[(size_t)i * 64]; v1 = p[(size_t)(i + 1) * 64]; v2 = p[(size_t)(i + 2) * 64]; v3 = p[(size_t)(i + 3) * 64];
        }
        a0 += v0; a1 += v1; a2 += v2; a3 += v3;
    }
    for (; i < q; ++i) {
        if constexpr (MODE & 1) a0 += __builtin_nontemporal_load(p + (size_t)i * 64);
        else a0 += p[(size_t)i * 64];
    }
    const ceil_d2 a = (a0 + a1) + (a2 + a3);
    const double s = a.x + a.y;
    if constexpr (MODE & 2) __builtin_nontemporal_store(s, wr + t);
    else wr[t] = s;
}

} // namespace sk

// gather_bench.hip -- what does an x[col] gather cost on MI355X, by access pattern?
// The SpMV kernels of the smoothed-aggregation levels (60-3000 nnz/row) run below the HBM ceiling although their
// bytes stream at it: the x[col] gathers hold the vector-memory pipe.  This probe measures the gather alone:
// every lane loads an index (coalesced int stream) and gathers from an L2-resident x of 2 MB, pattern by pattern.
//   hipcc --offload-arch=gfx950 -O2 -o tools/gather_bench tools/gather_bench.hip && tools/gather_bench
// Output: ns per wave-instruction per CU and lanes per clock per CU (at the nominal 2.4 GHz).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); return 1; } } while (0)

constexpr int BLOCK = 256, PER = 16;     // gathers per thread

// mode 0: 8-byte gather on every lane; 1: 8-byte gather on even lanes only (same instruction count, half the lanes);
// 2: 16-byte gather (two adjacent doubles) on every lane
template <int MODE>
__global__ __launch_bounds__(BLOCK) void k_gather(const int *__restrict__ idx, const double *__restrict__ x, double *out, long n) {
    const long base = (long)blockIdx.x * BLOCK * PER;
    double s = 0.0;
#pragma unroll 4
    for (int e = 0; e < PER; ++e) {
        const long k = base + (long)e * BLOCK + threadIdx.x;
        if (k < n) {
            const int j = idx[k];
            if (MODE == 0) s += x[j];
            else if (MODE == 1) { if ((threadIdx.x & 1) == 0) s += x[j]; }
            else { const double2 v = *reinterpret_cast<const double2 *>(x + j); s += v.x + v.y; }
        }
    }
    if (s == 1.2345e-300) out[0] = s;
}

static unsigned long long rng = 88172645463325252ull;
static unsigned rnd() { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return (unsigned)(rng >> 11); }

int main() {
    const long n = 64L << 20;            // gathers
    const int nx = 256 * 1024;           // doubles in x: 2 MB, L2-resident
    std::vector<double> hx(nx + 8, 1.0);
    double *x = nullptr, *out = nullptr;
    int *idx = nullptr;
    CK(hipMalloc(&x, (nx + 8) * sizeof(double)));
    CK(hipMalloc(&out, 64));
    CK(hipMalloc(&idx, n * sizeof(int)));
    CK(hipMemcpy(x, hx.data(), (nx + 8) * sizeof(double), hipMemcpyHostToDevice));
    std::vector<int> h(n);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    int ncu = 256;
    struct Pat { const char *name; int run; int align; int mode; };
    // run = consecutive indices on adjacent lanes before a random jump; align = run starts are multiples of this
    const Pat pats[] = {
        {"coalesced (identity mod nx)", 0, 1, 0},
        {"random, 8 B, all lanes", 1, 1, 0},
        {"random, 8 B, even lanes only (half the lanes, same instructions)", 1, 1, 1},
        {"runs of 2 (unaligned), 8 B", 2, 1, 0},
        {"runs of 2 aligned to 16 B, 8 B", 2, 2, 0},
        {"runs of 4 (unaligned), 8 B", 4, 1, 0},
        {"runs of 4 aligned to 32 B, 8 B", 4, 4, 0},
        {"runs of 8 (unaligned), 8 B", 8, 1, 0},
        {"runs of 8 aligned to 64 B, 8 B", 8, 8, 0},
        {"runs of 16 aligned to 128 B, 8 B", 16, 16, 0},
        {"random, 16 B per lane (aligned 16 B)", 1, 2, 2},
        {"random, 16 B per lane (8 B aligned)", 1, 1, 2},
    };
    // window: every workgroup's indices fall into its own window of W doubles (0 = anywhere in x): W = 1024 is an
    // 8 KB footprint, vector-L1 resident like a row block's x columns; 0 is L2-resident (every line an L1 miss)
    for (int W : {0, 1024})
    for (const Pat &p : pats) {
        if (&p == &pats[0]) printf("---- window per workgroup: %d doubles\n%-92s %10s %12s %12s\n", W, "pattern", "us", "ns/instr/CU", "lanes/clk/CU");
        if (p.run == 0) { for (long k = 0; k < n; ++k) h[k] = (int)(k % nx); }
        else {
            for (long k = 0; k < n; k += p.run) {
                const long blk = k / (BLOCK * PER);
                const int w0 = W ? (int)((blk * 977) % (nx - W - 32)) & ~15 : 0;
                int s = w0 + (int)(rnd() % (unsigned)((W ? W : nx) - 32));
                s -= s % p.align;
                for (int t = 0; t < p.run && k + t < n; ++t) h[k + t] = s + t;
            }
        }
        CK(hipMemcpy(idx, h.data(), n * sizeof(int), hipMemcpyHostToDevice));
        const int grid = (int)((n + BLOCK * PER - 1) / (BLOCK * PER));
        auto launch = [&]() {
            if (p.mode == 0) hipLaunchKernelGGL(k_gather<0>, dim3(grid), dim3(BLOCK), 0, 0, idx, x, out, n);
            else if (p.mode == 1) hipLaunchKernelGGL(k_gather<1>, dim3(grid), dim3(BLOCK), 0, 0, idx, x, out, n);
            else hipLaunchKernelGGL(k_gather<2>, dim3(grid), dim3(BLOCK), 0, 0, idx, x, out, n);
        };
        for (int w = 0; w < 2; ++w) launch();
        CK(hipEventRecord(e0, 0));
        const int reps = 5;
        for (int r = 0; r < reps; ++r) launch();
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1e3 / reps;
        const double instr_per_cu = (double)n / 64 / ncu;
        const double lanes = p.mode == 1 ? n / 2.0 : (double)n;
        printf("%-92s %10.1f %12.2f %12.2f\n", p.name, us, us * 1e3 / instr_per_cu, lanes / (us * 1e-6) / ncu / 2.4e9);
    }
    return 0;
}

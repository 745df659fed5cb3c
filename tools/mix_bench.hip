// mix_bench.hip -- the ceiling of k_sellp's byte mix on this MI355X (calibration for DESIGN.md section 4): per 64-row
// slice a wave reads W x 512 B of values (position-major, 16-byte loads) and 64 doubles of x, and writes 64 doubles of y.
// No gathers (every product takes x[r]), so what is measured is the memory system's rate for 8 W + 8 B read : 8 B
// written per row.   hipcc -O3 --offload-arch=gfx950 -o mix_bench mix_bench.hip && ./mix_bench
#include <hip/hip_runtime.h>
#include <cstdio>

typedef double d2v __attribute__((ext_vector_type(2)));
template <int W, int MODE>      // MODE bit 0: non-temporal value loads, bit 1: non-temporal y stores, bit 2: three x reads (r-1, r, r+1)
__global__ __launch_bounds__(256) void k_mix(const double *__restrict__ val, const double *__restrict__ x, double *__restrict__ y, int nslices) {
    const int s = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (s >= nslices) return;
    const int lane = threadIdx.x & 63;
    const size_t r = (size_t)s * 64 + lane;
    const d2v *v2 = reinterpret_cast<const d2v *>(val + (size_t)s * 64 * W) + lane;
    d2v vv[W / 2];
#pragma unroll
    for (int q = 0; q < W / 2; ++q) vv[q] = (MODE & 1) ? __builtin_nontemporal_load(v2 + q * 64) : v2[q * 64];
    double vt = 0;
    if (W & 1) vt = val[(size_t)s * 64 * W + (W / 2) * 128 + lane];
    double xr = x[r], xa = xr, xb = xr;
    if (MODE & 4) { xa = x[r ? r - 1 : 0]; xb = x[r + 1]; }
    double sum = 0;
#pragma unroll
    for (int q = 0; q < W / 2; ++q) { sum += vv[q].x * (q == 0 ? xa : xr); sum += vv[q].y * (q == 1 ? xb : xr); }
    sum += vt * xr;
    if (MODE & 2) __builtin_nontemporal_store(sum, y + r); else y[r] = sum;
}

template <class F>
float time_it(F f, int reps) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) f();
    hipEventRecord(a);
    for (int i = 0; i < reps; ++i) f();
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    return ms / reps;
}

template <int W>
void run(int rows) {
    const int ns = (rows + 63) / 64;
    double *val, *x, *y;
    hipMalloc(&val, (size_t)ns * 64 * W * 8 + 1024); hipMalloc(&x, (size_t)ns * 64 * 8 + 1024); hipMalloc(&y, (size_t)ns * 64 * 8 + 1024);
    hipMemset(val, 0, (size_t)ns * 64 * W * 8 + 1024); hipMemset(x, 0, (size_t)ns * 64 * 8 + 1024);
    const double bytes = (double)ns * 64 * (8.0 * W + 16.0);
    const dim3 g((ns + 3) / 4), b(256);
    float t0 = time_it([&] { hipLaunchKernelGGL((k_mix<W, 0>), g, b, 0, 0, val, x, y, ns); }, 30);
    float t1 = time_it([&] { hipLaunchKernelGGL((k_mix<W, 1>), g, b, 0, 0, val, x, y, ns); }, 30);
    float t2 = time_it([&] { hipLaunchKernelGGL((k_mix<W, 2>), g, b, 0, 0, val, x, y, ns); }, 30);
    float t3 = time_it([&] { hipLaunchKernelGGL((k_mix<W, 3>), g, b, 0, 0, val, x, y, ns); }, 30);
    float t4 = time_it([&] { hipLaunchKernelGGL((k_mix<W, 4>), g, b, 0, 0, val, x, y, ns); }, 30);
    printf("rows %9d W %2d: %7.1f MB | plain %6.1f us %6.0f GB/s | nt loads %6.1f us | nt stores %6.1f us | both %6.1f us | x three times %6.1f us\n",
           rows, W, bytes / 1e6, t0 * 1e3, bytes / t0 / 1e6, t1 * 1e3, t2 * 1e3, t3 * 1e3, t4 * 1e3);
    hipFree(val); hipFree(x); hipFree(y);
}

int main() {
    for (int rows : {2000376, 16387064, 33000000}) { run<7>(rows); run<8>(rows); }
    return 0;
}

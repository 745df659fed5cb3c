// hop_bench.hip -- what does a cross-stream dependency cost on this stack?
// Ping-pong between two streams with tiny kernels: A:k -> (dep) -> B:k -> (dep) -> A:k ...
//   hipcc --offload-arch=gfx950 -O2 -o tools/hop_bench tools/hop_bench.hip && tools/hop_bench
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k_tiny(int *p) { if (threadIdx.x == 0) atomicAdd(p, 1); }

int main() {
    hipStream_t A, B;
    CK(hipStreamCreateWithFlags(&A, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&B, hipStreamNonBlocking));
    int *d = nullptr;
    CK(hipMalloc(&d, 4096));
    CK(hipMemset(d, 0, 4096));
    const int N = 2000;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto us = [](auto a, auto b) { return std::chrono::duration<double, std::micro>(b - a).count(); };

    // 1. one stream, 2N kernels
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipDeviceSynchronize());
        auto t0 = now();
        for (int i = 0; i < 2 * N; ++i) hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, A, d);
        CK(hipDeviceSynchronize());
        if (rep) printf("same stream, back-to-back kernels      : %6.2f us per kernel\n", us(t0, now()) / (2 * N));
    }
    // 2/3. event ping-pong with different flags
    const unsigned flagset[] = {hipEventDisableTiming, hipEventDisableTiming | hipEventDisableSystemFence, hipEventDefault};
    const char *names[] = {"DisableTiming", "DisableTiming|DisableSystemFence", "Default"};
    for (int f = 0; f < 3; ++f) {
        hipEvent_t ea, eb;
        CK(hipEventCreateWithFlags(&ea, flagset[f]));
        CK(hipEventCreateWithFlags(&eb, flagset[f]));
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipDeviceSynchronize());
            auto t0 = now();
            for (int i = 0; i < N; ++i) {
                hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, A, d);
                CK(hipEventRecord(ea, A));
                CK(hipStreamWaitEvent(B, ea, 0));
                hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, B, d);
                CK(hipEventRecord(eb, B));
                CK(hipStreamWaitEvent(A, eb, 0));
            }
            auto t1 = now();
            CK(hipDeviceSynchronize());
            if (rep) printf("event ping-pong (%-32s): %6.2f us per kernel+hop (host enqueue %5.2f)\n", names[f], us(t0, now()) / (2 * N), us(t0, t1) / (2 * N));
        }
        hipEventDestroy(ea); hipEventDestroy(eb);
    }
    // 4. stream memory operations: write value after the kernel, the other stream waits for it
    {
        uint32_t *flag = nullptr, *flag2 = nullptr;
        CK(hipExtMallocWithFlags(reinterpret_cast<void **>(&flag), 8, hipMallocSignalMemory));
        CK(hipExtMallocWithFlags(reinterpret_cast<void **>(&flag2), 8, hipMallocSignalMemory));
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipDeviceSynchronize());
            CK(hipStreamWriteValue32(A, flag, 0, 0));
            CK(hipStreamWriteValue32(A, flag2, 0, 0));
            CK(hipDeviceSynchronize());
            auto t0 = now();
            bool ok = true;
            for (int i = 0; i < N && ok; ++i) {
                const uint32_t v = (uint32_t)i + 1;
                hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, A, d);
                ok = ok && hipStreamWriteValue32(A, flag, v, 0) == hipSuccess;
                ok = ok && hipStreamWaitValue32(B, flag, v, hipStreamWaitValueGte, 0xffffffffu) == hipSuccess;
                hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, B, d);
                ok = ok && hipStreamWriteValue32(B, flag2, v, 0) == hipSuccess;
                ok = ok && hipStreamWaitValue32(A, flag2, v, hipStreamWaitValueGte, 0xffffffffu) == hipSuccess;
            }
            auto t1 = now();
            CK(hipDeviceSynchronize());
            if (!ok) { printf("stream write/wait value: not supported here (%s)\n", hipGetErrorString(hipGetLastError())); break; }
            if (rep) printf("stream write/wait value ping-pong       : %6.2f us per kernel+hop (host enqueue %5.2f)\n", us(t0, now()) / (2 * N), us(t0, t1) / (2 * N));
        }
    }
    // 5. GPU-side cost only: everything is enqueued behind a gate (stream A waits for a value the host releases
    //    afterwards through stream C), so the host's API time is off the clock; timing events bracket the chain on A.
    {
        hipStream_t Cs;
        CK(hipStreamCreateWithFlags(&Cs, hipStreamNonBlocking));
        uint32_t *gate = nullptr, *f1 = nullptr, *f2 = nullptr;
        CK(hipExtMallocWithFlags(reinterpret_cast<void **>(&gate), 8, hipMallocSignalMemory));
        CK(hipExtMallocWithFlags(reinterpret_cast<void **>(&f1), 8, hipMallocSignalMemory));
        CK(hipExtMallocWithFlags(reinterpret_cast<void **>(&f2), 8, hipMallocSignalMemory));
        hipEvent_t t0e, t1e, ea, eb;
        CK(hipEventCreate(&t0e)); CK(hipEventCreate(&t1e));
        CK(hipEventCreateWithFlags(&ea, hipEventDisableTiming | hipEventDisableSystemFence));
        CK(hipEventCreateWithFlags(&eb, hipEventDisableTiming | hipEventDisableSystemFence));
        const int M = 400;
        for (int mode = 0; mode < 3; ++mode) {            // 0 one stream, 1 events, 2 write/wait value
            CK(hipStreamWriteValue32(Cs, gate, 0, 0)); CK(hipStreamWriteValue32(Cs, f1, 0, 0)); CK(hipStreamWriteValue32(Cs, f2, 0, 0));
            CK(hipDeviceSynchronize());
            CK(hipStreamWaitValue32(A, gate, 1, hipStreamWaitValueEq, 0xffffffffu));
            CK(hipEventRecord(t0e, A));
            for (int i = 0; i < M; ++i) {
                const uint32_t v = (uint32_t)i + 1;
                hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, A, d);
                if (mode == 0) { hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, A, d); continue; }
                if (mode == 1) { CK(hipEventRecord(ea, A)); CK(hipStreamWaitEvent(B, ea, 0)); }
                else { CK(hipStreamWriteValue32(A, f1, v, 0)); CK(hipStreamWaitValue32(B, f1, v, hipStreamWaitValueGte, 0xffffffffu)); }
                hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, B, d);
                if (mode == 1) { CK(hipEventRecord(eb, B)); CK(hipStreamWaitEvent(A, eb, 0)); }
                else { CK(hipStreamWriteValue32(B, f2, v, 0)); CK(hipStreamWaitValue32(A, f2, v, hipStreamWaitValueGte, 0xffffffffu)); }
            }
            CK(hipEventRecord(t1e, A));
            CK(hipStreamWriteValue32(Cs, gate, 1, 0));      // open the gate
            CK(hipDeviceSynchronize());
            float ms = 0;
            CK(hipEventElapsedTime(&ms, t0e, t1e));
            printf("gated, GPU side only, %-22s: %6.2f us per kernel%s\n", mode == 0 ? "one stream" : mode == 1 ? "event ping-pong" : "value ping-pong",
                   ms * 1e3 / (2 * M), mode ? "+hop" : "");
        }
    }
    int h = 0;
    CK(hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost));
    printf("kernels run: %d\n", h);
    return 0;
}

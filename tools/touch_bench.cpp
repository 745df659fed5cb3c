// touch_bench.cpp -- what a first touch of fresh host memory costs on this box (the setup's big vectors): plain 4 KiB pages,
// with madvise(MADV_HUGEPAGE), on one thread and on several.  g++ -O2 -pthread -o touch_bench touch_bench.cpp
#include <sys/mman.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static void touch(char *p, size_t n, int threads) {
    std::vector<std::thread> th;
    for (int t = 0; t < threads; ++t)
        th.emplace_back([=] { const size_t a = n * t / threads, b = n * (t + 1) / threads; for (size_t i = a; i < b; i += 4096) p[i] = 1; });
    for (auto &x : th) x.join();
}
int main() {
    const size_t n = (size_t)2 << 30;
    for (int huge = 0; huge < 2; ++huge)
        for (int threads : {1, 4, 16}) {
            char *p = (char *)mmap(nullptr, n, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
            if (huge) madvise(p, n, MADV_HUGEPAGE);
            double t0 = now();
            touch(p, n, threads);
            double t1 = now();
            memset(p, 2, n);
            double t2 = now();
            printf("%s pages, %2d thread(s): first touch %.2f GB/s, memset afterwards %.2f GB/s\n", huge ? "huge (madvise)" : "4 KiB        ", threads, n / (t1 - t0) / 1e9, n / (t2 - t1) / 1e9);
            munmap(p, n);
        }
    // a std::vector the way the setup makes them
    double t0 = now();
    std::vector<double> v((size_t)256 << 20);
    double t1 = now();
    printf("std::vector<double>(256 Mi): %.2f GB/s\n", v.size() * 8 / (t1 - t0) / 1e9);
    return 0;
}

// par.h -- the few threaded loops the host setup shares: chunks of an index range on SAENA_SETUP_THREADS threads, and a
// copy of a large array split over them (a multi-gigabyte memcpy on one thread runs at a fraction of the memory rate).
#pragma once
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <mutex>
#include <thread>
#include <vector>

namespace saena_host {

// bigalloc.cpp: allocations made by this thread inside a LazyBigalloc scope are advised to use huge pages but not touched
int &bigalloc_lazy_depth();
struct LazyBigalloc {
    LazyBigalloc() { ++bigalloc_lazy_depth(); }
    ~LazyBigalloc() { --bigalloc_lazy_depth(); }
    LazyBigalloc(const LazyBigalloc &) = delete;
    LazyBigalloc &operator=(const LazyBigalloc &) = delete;
};

inline int setup_threads() {
    static const int n = [] {
        if (const char *e = std::getenv("SAENA_SETUP_THREADS")) return std::max(1, std::atoi(e));
        return (int)std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    }();
    return n;
}

// A pool of worker threads per process: the aggregation runs ~10^3 synchronous rounds of three short parallel loops each,
// and creating 16 threads per loop cost more than the late rounds' work.  run(T, f) executes f(0) ... f(T-1), f(0) on the
// caller; workers spin briefly for the next job, then sleep on a condition variable.  Jobs do not nest (a job that calls
// run() again executes its inner job on its own thread).
class ThreadPool {
public:
    static ThreadPool &get() { static ThreadPool p; return p; }
    template <class F>
    void run(int T, F &&f) {
        if (T <= 1 || busy_.exchange(true)) { for (int t = 0; t < T; ++t) f(t); return; }      // nested or concurrent use: serial
        struct Reset { std::atomic<bool> &b; ~Reset() { b.store(false); } } reset{busy_};
        ensure(T - 1);
        std::exception_ptr err;
        std::mutex emu;
        std::function<void(int)> job = [&](int t) {
            try { f(t); }
            catch (...) { std::lock_guard<std::mutex> g(emu); err = std::current_exception(); }
        };
        {
            std::lock_guard<std::mutex> g(mu_);
            job_ = &job; njobs_ = T; next_ = 1; pending_ = T - 1; ++generation_;
        }
        cv_.notify_all();
        job(0);
        {
            std::unique_lock<std::mutex> g(mu_);
            while (next_ < njobs_) {                     // the caller takes what the workers have not started yet
                const int t = next_++;
                g.unlock();
                job(t);
                g.lock();
                --pending_;
            }
            done_.wait(g, [&] { return pending_ == 0; });
            job_ = nullptr;
        }
        if (err) std::rethrow_exception(err);
    }
    ~ThreadPool() {
        { std::lock_guard<std::mutex> g(mu_); stop_ = true; ++generation_; }
        cv_.notify_all();
        for (auto &t : workers_) t.join();
    }
private:
    void ensure(int n) {
        while ((int)workers_.size() < n) workers_.emplace_back([this] { loop(); });
    }
    void loop() {
        unsigned long seen = 0;
        while (true) {
            std::unique_lock<std::mutex> g(mu_);
            cv_.wait(g, [&] { return stop_ || (generation_ != seen && job_ && next_ < njobs_); });
            if (stop_) return;
            while (job_ && next_ < njobs_) {
                const int t = next_++;
                std::function<void(int)> *j = job_;
                g.unlock();
                (*j)(t);
                g.lock();
                if (--pending_ == 0) done_.notify_one();
            }
            seen = generation_;
        }
    }
    std::vector<std::thread> workers_;
    std::mutex mu_;
    std::condition_variable cv_, done_;
    std::function<void(int)> *job_ = nullptr;
    int njobs_ = 0, next_ = 0, pending_ = 0;
    unsigned long generation_ = 0;
    bool stop_ = false;
    std::atomic<bool> busy_{false};
};

// f(t, lo, hi) over [0, n) in T contiguous chunks of equal length (T = min(threads, n / grain + 1))
template <class Index, class F>
void parallel_chunks(Index n, Index grain, F f) {
    const int T = (int)std::max<long>(1, std::min<long>(setup_threads(), (long)(n / std::max<Index>(grain, 1)) + 1));
    if (T == 1) { f(0, (Index)0, n); return; }
    ThreadPool::get().run(T, [&](int t) { f(t, (Index)((long double)n * t / T), (Index)((long double)n * (t + 1) / T)); });
}

inline void parallel_copy(void *dst, const void *src, size_t bytes) {
    if (bytes < ((size_t)32 << 20)) { if (bytes) std::memcpy(dst, src, bytes); return; }
    parallel_chunks<size_t>(bytes, (size_t)16 << 20, [&](int, size_t a, size_t b) { std::memcpy(static_cast<char *>(dst) + a, static_cast<const char *>(src) + a, b - a); });
}
template <class T>
void parallel_copy(T *dst, const T *src, size_t n) { parallel_copy(static_cast<void *>(dst), static_cast<const void *>(src), n * sizeof(T)); }

} // namespace saena_host

// bigalloc.cpp -- large allocations of THIS library come back resident, on transparent huge pages, touched by several threads.
//
// The setup moves multi-gigabyte arrays (a 256^3 hierarchy holds 1.0 G entries), mostly in std::vector.  glibc serves
// every allocation beyond 32 MiB from a fresh mmap, and a fresh mapping costs a page fault per 4 KiB at its first touch:
// ~2 GB/s on one thread (tools/touch_bench.cpp), which is what `std::vector<double> v(n)`, a `resize` or the first copy
// into a new vector ran at -- several times slower than the copies and the PCIe transfers the setup is made of.  With
// madvise(MADV_HUGEPAGE) a fault maps 2 MiB and touching from a few threads reaches 20-40 GB/s.
//
// operator new is wrapped for this library only (kept out of the dynamic symbol table by host/exports.map: other modules keep
// theirs): the wrapper asks the NEXT operator new in the process for the memory (libstdc++'s, or a sanitizer's) and then makes a
// large block resident, so every block is still released by the operator delete that matches its allocation, whichever module
// frees it.  SAENA_NO_BIGALLOC=1 switches the treatment of large blocks off.
#include <cstdlib>
#include <cstdint>
#include <new>
#include <thread>
#include <vector>

#include <dlfcn.h>
#include <sys/mman.h>

#include "par.h"

// Reservations (vector::reserve of an over-estimate: the products' 1.15 x estimate of up to 3e9 entries, a thread's piece of the
// smoothed P) only get the huge-page advice inside a LazyBigalloc scope: touching them would turn capacity that may never be
// filled into resident memory (round-3 advisor finding: with several ranks per node that is what reaches the OOM killer first).
// Whoever fills the block later faults it in 2 MiB at a time.
int &saena_host::bigalloc_lazy_depth() { static thread_local int depth = 0; return depth; }

namespace {

constexpr size_t BIG = (size_t)8 << 20;          // from here on: huge pages + parallel first touch
constexpr size_t HUGE = (size_t)2 << 20;

bool enabled() { static const bool on = std::getenv("SAENA_NO_BIGALLOC") == nullptr; return on; }
int touch_threads() {
    static const int n = [] {
        const char *e = std::getenv("SAENA_SETUP_THREADS");
        int t = e ? std::atoi(e) : (int)std::thread::hardware_concurrency();
        return t < 1 ? 1 : t > 16 ? 16 : t;
    }();
    return n;
}

void make_resident(void *p, size_t n) {
    if (!enabled()) return;
    char *b = static_cast<char *>(p);
    // a block this large is a mapping of its own (glibc: mmap beyond the threshold): advise its 2 MiB-aligned interior
    const uintptr_t lo = ((uintptr_t)b + HUGE - 1) & ~(uintptr_t)(HUGE - 1), hi = ((uintptr_t)b + n) & ~(uintptr_t)(HUGE - 1);
    if (hi > lo) madvise(reinterpret_cast<void *>(lo), hi - lo, MADV_HUGEPAGE);
    if (saena_host::bigalloc_lazy_depth() > 0) return;          // a reservation: advice only
    const int T = touch_threads();
    auto touch = [b, n](size_t a, size_t e) { for (size_t i = a; i < e && i < n; i += 4096) static_cast<volatile char *>(b)[i] = 0; };
    if (T == 1 || n < 4 * BIG) { touch(0, n); return; }
    std::vector<std::thread> th;
    th.reserve((size_t)T - 1);
    const size_t per = (n / (size_t)T + HUGE - 1) & ~(HUGE - 1);
    for (int t = 1; t < T; ++t) th.emplace_back(touch, per * (size_t)t, per * (size_t)(t + 1));
    touch(0, per);
    for (auto &x : th) x.join();
}

// the operator new that would have served this library without the wrapper
template <class F>
F next_symbol(const char *name) { return reinterpret_cast<F>(dlsym(RTLD_NEXT, name)); }
using new_fn = void *(*)(size_t);
using new_al_fn = void *(*)(size_t, std::align_val_t);
new_fn next_new() { static const new_fn f = next_symbol<new_fn>("_Znwm"); return f; }
new_fn next_new_arr() { static const new_fn f = next_symbol<new_fn>("_Znam"); return f; }
new_al_fn next_new_al() { static const new_al_fn f = next_symbol<new_al_fn>("_ZnwmSt11align_val_t"); return f; }
new_al_fn next_new_arr_al() { static const new_al_fn f = next_symbol<new_al_fn>("_ZnamSt11align_val_t"); return f; }

void *resident(void *p, size_t n) {
    if (p && n >= BIG) make_resident(p, n);
    return p;
}

} // namespace

// (no operator delete here: the memory comes from the next operator new, so the process's own operator delete is its match)
void *operator new(size_t n) { new_fn f = next_new(); if (!f) throw std::bad_alloc(); return resident(f(n), n); }
void *operator new[](size_t n) { new_fn f = next_new_arr(); if (!f) throw std::bad_alloc(); return resident(f(n), n); }
void *operator new(size_t n, const std::nothrow_t &) noexcept { try { return operator new(n); } catch (...) { return nullptr; } }
void *operator new[](size_t n, const std::nothrow_t &) noexcept { try { return operator new[](n); } catch (...) { return nullptr; } }
void *operator new(size_t n, std::align_val_t a) { new_al_fn f = next_new_al(); if (!f) throw std::bad_alloc(); return resident(f(n, a), n); }
void *operator new[](size_t n, std::align_val_t a) { new_al_fn f = next_new_arr_al(); if (!f) throw std::bad_alloc(); return resident(f(n, a), n); }

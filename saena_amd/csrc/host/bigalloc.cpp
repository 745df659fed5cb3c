// bigalloc.cpp -- large allocations of THIS library come back resident, on transparent huge pages, touched by several threads.
//
// The setup moves multi-gigabyte arrays (a 256^3 hierarchy holds 1.0 G entries), mostly in std::vector.  glibc serves
// every allocation beyond 32 MiB from a fresh mmap, and a fresh mapping costs a page fault per 4 KiB at its first touch:
// ~2 GB/s on one thread (tools/touch_bench.cpp), which is what `std::vector<double> v(n)`, a `resize` or the first copy
// into a new vector ran at -- several times slower than the copies and the PCIe transfers the setup is made of.  With
// madvise(MADV_HUGEPAGE) a fault maps 2 MiB and touching from a few threads reaches 20-40 GB/s.
//
// operator new / delete are replaced for this library only (hidden visibility: other modules keep theirs).  Memory still
// comes from malloc and goes back through free, so a pointer may cross module boundaries either way.
// SAENA_NO_BIGALLOC=1 switches the treatment of large blocks off.
#include <cstdlib>
#include <cstdint>
#include <new>
#include <thread>
#include <vector>

#include <sys/mman.h>

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

void *get(size_t n) {
    void *p = std::malloc(n ? n : 1);
    if (!p) throw std::bad_alloc();
    if (n >= BIG) make_resident(p, n);
    return p;
}
void *get_aligned(size_t n, size_t al) {
    void *p = nullptr;
    if (posix_memalign(&p, al < sizeof(void *) ? sizeof(void *) : al, n ? n : 1) != 0) throw std::bad_alloc();
    if (n >= BIG) make_resident(p, n);
    return p;
}

} // namespace

#define SAENA_HIDDEN      /* kept out of the dynamic symbol table by host/exports.map (the declarations of <new> fix default visibility) */
SAENA_HIDDEN void *operator new(size_t n) { return get(n); }
SAENA_HIDDEN void *operator new[](size_t n) { return get(n); }
SAENA_HIDDEN void *operator new(size_t n, const std::nothrow_t &) noexcept { try { return get(n); } catch (...) { return nullptr; } }
SAENA_HIDDEN void *operator new[](size_t n, const std::nothrow_t &) noexcept { try { return get(n); } catch (...) { return nullptr; } }
SAENA_HIDDEN void *operator new(size_t n, std::align_val_t a) { return get_aligned(n, (size_t)a); }
SAENA_HIDDEN void *operator new[](size_t n, std::align_val_t a) { return get_aligned(n, (size_t)a); }
SAENA_HIDDEN void operator delete(void *p) noexcept { std::free(p); }
SAENA_HIDDEN void operator delete[](void *p) noexcept { std::free(p); }
SAENA_HIDDEN void operator delete(void *p, size_t) noexcept { std::free(p); }
SAENA_HIDDEN void operator delete[](void *p, size_t) noexcept { std::free(p); }
SAENA_HIDDEN void operator delete(void *p, std::align_val_t) noexcept { std::free(p); }
SAENA_HIDDEN void operator delete[](void *p, std::align_val_t) noexcept { std::free(p); }
SAENA_HIDDEN void operator delete(void *p, size_t, std::align_val_t) noexcept { std::free(p); }
SAENA_HIDDEN void operator delete[](void *p, size_t, std::align_val_t) noexcept { std::free(p); }
SAENA_HIDDEN void operator delete(void *p, const std::nothrow_t &) noexcept { std::free(p); }
SAENA_HIDDEN void operator delete[](void *p, const std::nothrow_t &) noexcept { std::free(p); }

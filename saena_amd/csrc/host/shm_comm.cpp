// shm_comm.cpp -- see shm_comm.h
#include "shm_comm.h"

#include <atomic>
#include <cerrno>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

namespace saena_host {
namespace {

constexpr int      MAXR  = 64;
constexpr uint32_t MAGIC = 0x5AE7A001u;
constexpr size_t   GRAIN = (size_t)2 << 20;            // segments grow in 2 MiB steps

struct PerRank {
    std::atomic<uint64_t> seg_size;                    // bytes of this rank's data segment
    uint64_t off[MAXR], cnt[MAXR];                     // where in it the block for rank p starts, and its length
};
struct Ctl {
    std::atomic<uint32_t> magic, bar_count, bar_gen, failed;
    std::atomic<uint32_t> attached, go;                // attach handshake: ranks != 0 count in, rank 0 opens the gate
    uint32_t nranks;
    PerRank r[MAXR];
};

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct ShmComm : Comm {
    std::string base;
    Ctl   *ctl = nullptr;
    int    fd[MAXR];
    char  *seg[MAXR];
    size_t mapped[MAXR];
    size_t committed = 0;                              // bytes of the own segment whose pages tmpfs has reserved
    double timeout = 900.0;

    ShmComm() { for (int i = 0; i < MAXR; ++i) { fd[i] = -1; seg[i] = nullptr; mapped[i] = 0; } }
    ~ShmComm() override {
        if (std::getenv("SAENA_SETUP_TIMING") && mapped[rank])
            fprintf(stderr, "[shm] rank %d: the largest exchange needed a segment of %.1f MiB\n", rank, (double)mapped[rank] / 1048576.0);
        for (int p = 0; p < MAXR; ++p) {
            if (seg[p]) munmap(seg[p], mapped[p]);
            if (fd[p] >= 0) close(fd[p]);
        }
        if (ctl) munmap(ctl, sizeof(Ctl));
    }

    [[noreturn]] void die(const std::string &what) {
        if (ctl) ctl->failed.store(1, std::memory_order_release);     // whoever waits for this rank gives up as well
        throw std::runtime_error("shared-memory communicator (rank " + std::to_string(rank) + "): " + what);
    }
    template <class F>
    void wait_until(F done, const char *what) {
        const double t0 = now_s();
        for (long spins = 0;; ++spins) {
            if (done()) return;
            if (ctl && ctl->failed.load(std::memory_order_acquire)) throw std::runtime_error(std::string("shared-memory communicator: another rank failed while this one waited for ") + what);
            if (spins < 4000) { __builtin_ia32_pause(); continue; }
            if (spins < 8000) { sched_yield(); continue; }
            timespec ts{0, 50000};
            nanosleep(&ts, nullptr);
            if ((spins & 1023) == 0 && now_s() - t0 > timeout) die(std::string("timed out waiting for ") + what);
        }
    }
    void barrier() {
        const uint32_t gen = ctl->bar_gen.load(std::memory_order_acquire);
        if (ctl->bar_count.fetch_add(1, std::memory_order_acq_rel) + 1 == (uint32_t)nranks) {
            ctl->bar_count.store(0, std::memory_order_relaxed);
            ctl->bar_gen.store(gen + 1, std::memory_order_release);
        } else {
            wait_until([&] { return ctl->bar_gen.load(std::memory_order_acquire) != gen; }, "the other ranks at a barrier");
        }
    }
    // own segment: at least `bytes`, WITH their pages.  ftruncate alone only moves the end of the file -- tmpfs hands out pages
    // at the first touch, and a /dev/shm smaller than an exchange (a container's default is 64 MB) would then end the process
    // with SIGBUS inside a copy.  posix_fallocate reserves the pages now and reports ENOSPC as an error the caller can act on
    // (bench.py falls back to the gloo transport).
    void commit(size_t bytes) {
        if (bytes <= committed) return;
        const size_t upto = std::min(mapped[rank], (bytes + GRAIN - 1) / GRAIN * GRAIN);
        int rc;
        do rc = posix_fallocate(fd[rank], (off_t)committed, (off_t)(upto - committed)); while (rc == EINTR);
        if (rc != 0)
            die("no room for " + std::to_string(upto) + " bytes in /dev/shm (" + std::strerror(rc) + "): the exchanges of a " + std::to_string(nranks) +
                "-rank setup need that much shared memory per rank -- enlarge /dev/shm or use another setup transport");
        committed = upto;
    }
    void reserve(size_t bytes) {
        if (bytes > mapped[rank]) {
            size_t cap = std::max(bytes, 2 * mapped[rank]);
            cap = (cap + GRAIN - 1) / GRAIN * GRAIN;
            if (ftruncate(fd[rank], (off_t)cap) != 0) die("ftruncate of the data segment to " + std::to_string(cap) + " bytes failed: " + std::strerror(errno));
            if (seg[rank]) munmap(seg[rank], mapped[rank]);
            void *m = mmap(nullptr, cap, PROT_READ | PROT_WRITE, MAP_SHARED, fd[rank], 0);
            if (m == MAP_FAILED) { seg[rank] = nullptr; mapped[rank] = 0; die("mmap of the data segment failed"); }
            seg[rank] = static_cast<char *>(m);
            mapped[rank] = cap;
            ctl->r[rank].seg_size.store(cap, std::memory_order_release);
            if (std::getenv("SAENA_SETUP_TIMING") && cap >= ((size_t)64 << 20))
                fprintf(stderr, "[shm] rank %d: the segment grows to %.0f MiB for an exchange of %.1f MiB\n", rank, (double)cap / 1048576.0, (double)bytes / 1048576.0);
        }
        commit(bytes);
    }
    // a peer's segment, mapped at its current size
    const char *peer(int p) {
        const size_t want = (size_t)ctl->r[p].seg_size.load(std::memory_order_acquire);
        if (want > mapped[p]) {
            if (seg[p]) munmap(seg[p], mapped[p]);
            void *m = mmap(nullptr, want, PROT_READ, MAP_SHARED, fd[p], 0);
            if (m == MAP_FAILED) { seg[p] = nullptr; mapped[p] = 0; die("mmap of rank " + std::to_string(p) + "'s segment failed"); }
            seg[p] = static_cast<char *>(m);
            mapped[p] = want;
        }
        return seg[p];
    }
    // give the pages of a large exchange back (tmpfs pages are memory); the next exchange reserves its own again
    void release(size_t used) {
        if (used > ((size_t)64 << 20) && madvise(seg[rank], mapped[rank], MADV_REMOVE) == 0) committed = 0;
    }

    void alltoallv(const void *send, const size_t *sc, const size_t *sd, void *recv, const size_t *rc, const size_t *rd) override {
        const char *s = static_cast<const char *>(send);
        char *r = static_cast<char *>(recv);
        if (sc[rank] != rc[rank]) die("alltoallv: the block a rank sends to itself has two sizes");
        size_t tot = 0;
        for (int p = 0; p < nranks; ++p) if (p != rank) tot += sc[p];
        reserve(tot);
        PerRank &me = ctl->r[rank];
        size_t o = 0;
        for (int p = 0; p < nranks; ++p) {
            me.off[p] = o; me.cnt[p] = p == rank ? 0 : sc[p];
            if (p != rank && sc[p]) { parallel_copy(seg[rank] + o, s + sd[p], sc[p]); o += sc[p]; }
        }
        if (sc[rank]) parallel_copy(r + rd[rank], s + sd[rank], sc[rank]);      // (send and recv never overlap)
        barrier();                                       // every segment is written
        for (int k = 1; k < nranks; ++k) {               // start with the next rank: not everybody reads rank 0 first
            const int p = (rank + k) % nranks;
            const PerRank &q = ctl->r[p];
            if (q.cnt[rank] != rc[p]) die("alltoallv: rank " + std::to_string(p) + " sends " + std::to_string(q.cnt[rank]) + " bytes, " + std::to_string(rc[p]) + " expected");
            if (rc[p]) parallel_copy(r + rd[p], peer(p) + q.off[rank], rc[p]);
        }
        barrier();                                       // every segment is read: it may be overwritten
        release(tot);
    }
    void allgather(const void *send, void *recv, size_t bytes) override {
        reserve(bytes);
        std::memcpy(seg[rank], send, bytes);
        barrier();
        char *r = static_cast<char *>(recv);
        for (int p = 0; p < nranks; ++p) std::memcpy(r + (size_t)p * bytes, p == rank ? static_cast<const char *>(send) : peer(p), bytes);
        barrier();
        release(bytes);
    }
    template <class T>
    void allreduce(T *v, int n) {
        if (n <= 0) return;
        std::vector<T> all((size_t)n * nranks);
        allgather(v, all.data(), sizeof(T) * (size_t)n);
        for (int i = 0; i < n; ++i) {                    // rank order on every rank: the same sum everywhere
            T s = all[(size_t)i];
            for (int p = 1; p < nranks; ++p) s += all[(size_t)p * n + i];
            v[i] = s;
        }
    }
    void allreduce_sum_i64(long *v, int n) override { allreduce(v, n); }
    void allreduce_sum_f64(double *v, int n) override { allreduce(v, n); }
};

} // namespace

std::unique_ptr<Comm> make_shm_comm(const std::string &name, int rank, int nranks, std::string *err) {
    auto fail = [&](const std::string &m) { if (err) *err = m; return std::unique_ptr<Comm>(); };
    if (nranks < 1 || nranks > MAXR || rank < 0 || rank >= nranks) return fail("shared-memory communicator: rank " + std::to_string(rank) + " of " + std::to_string(nranks) + " (at most 64 ranks)");
    if (name.empty() || name.find('/') != std::string::npos) return fail("shared-memory communicator: the name must be non-empty and hold no '/'");
    std::unique_ptr<ShmComm> c(new ShmComm());
    c->rank = rank; c->nranks = nranks;
    c->base = "/saena_" + name;
    if (const char *t = std::getenv("SAENA_SHM_TIMEOUT")) c->timeout = std::max(1.0, atof(t));
    try {
        // ---- control block: rank 0 creates and initialises it, the others wait for it ----
        // A name can outlive a job that died before its ranks had all attached.  Rank 0 marks such a leftover FAILED before it
        // replaces it, and a rank that has mapped a block only trusts it once rank 0 has opened the gate (`go`) after counting
        // every rank in -- so a rank that opened the leftover a moment before rank 0 replaced it sees `failed`, lets go of it and
        // opens the name again, instead of sitting in a dead block's barrier until the timeout.
        auto map_ctl = [&](int cfd) -> Ctl * {
            void *m = mmap(nullptr, sizeof(Ctl), PROT_READ | PROT_WRITE, MAP_SHARED, cfd, 0);
            close(cfd);
            return m == MAP_FAILED ? nullptr : static_cast<Ctl *>(m);
        };
        const double t_attach = now_s();
        if (rank == 0) {
            int old = shm_open(c->base.c_str(), O_RDWR, 0600);
            if (old >= 0) {
                struct stat st;
                if (fstat(old, &st) == 0 && (size_t)st.st_size >= sizeof(Ctl)) {
                    if (Ctl *dead = map_ctl(old)) { dead->failed.store(1, std::memory_order_release); munmap(dead, sizeof(Ctl)); }
                } else close(old);
                shm_unlink(c->base.c_str());
            }
            int cfd = shm_open(c->base.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
            if (cfd < 0) return fail("shm_open(" + c->base + ") failed: " + std::strerror(errno));
            int rc = ftruncate(cfd, (off_t)sizeof(Ctl)) != 0 ? errno : 0;
            if (rc == 0) do rc = posix_fallocate(cfd, 0, (off_t)sizeof(Ctl)); while (rc == EINTR);
            if (rc != 0) { close(cfd); shm_unlink(c->base.c_str()); return fail(std::string("no room for the control block in /dev/shm: ") + std::strerror(rc)); }
            c->ctl = map_ctl(cfd);
            if (!c->ctl) { shm_unlink(c->base.c_str()); return fail("mmap of the control block failed"); }
            c->ctl->bar_count.store(0); c->ctl->bar_gen.store(0); c->ctl->failed.store(0);
            c->ctl->attached.store(0); c->ctl->go.store(0);
            c->ctl->nranks = (uint32_t)nranks;
            for (int p = 0; p < MAXR; ++p) c->ctl->r[p].seg_size.store(0);
            c->ctl->magic.store(MAGIC, std::memory_order_release);
            c->wait_until([&] { return c->ctl->attached.load(std::memory_order_acquire) == (uint32_t)(nranks - 1); }, "the other ranks to attach");
            c->ctl->go.store(1, std::memory_order_release);
        } else {
            while (true) {
                if (now_s() - t_attach > c->timeout) return fail("timed out waiting for rank 0 to create " + c->base);
                timespec ts{0, 2000000};
                int cfd = shm_open(c->base.c_str(), O_RDWR, 0600);
                struct stat st;
                if (cfd < 0 || fstat(cfd, &st) != 0 || (size_t)st.st_size < sizeof(Ctl)) {
                    if (cfd >= 0) close(cfd);
                    nanosleep(&ts, nullptr);
                    continue;
                }
                Ctl *m = map_ctl(cfd);
                if (!m) return fail("mmap of the control block failed");
                // this block's rank 0 either opens the gate or (a leftover) some rank 0 marks it failed
                bool counted = false, stale = false;
                while (true) {
                    if (m->failed.load(std::memory_order_acquire)) { stale = true; break; }
                    if (m->magic.load(std::memory_order_acquire) == MAGIC) {
                        if (m->nranks != (uint32_t)nranks) { munmap(m, sizeof(Ctl)); return fail("shared-memory communicator: the ranks disagree about the size of the job"); }
                        if (!counted) { m->attached.fetch_add(1, std::memory_order_acq_rel); counted = true; }
                        if (m->go.load(std::memory_order_acquire)) break;
                    }
                    if (now_s() - t_attach > c->timeout) { munmap(m, sizeof(Ctl)); return fail("timed out waiting for rank 0 to initialise " + c->base); }
                    nanosleep(&ts, nullptr);
                }
                if (!stale) { c->ctl = m; break; }
                munmap(m, sizeof(Ctl));                  // a leftover of an earlier job: open the name again
                nanosleep(&ts, nullptr);
            }
        }
        // ---- data segments: everyone creates its own, then opens the others' ----
        const std::string mine = c->base + "." + std::to_string(rank);
        shm_unlink(mine.c_str());
        c->fd[rank] = shm_open(mine.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
        if (c->fd[rank] < 0) c->die("shm_open(" + mine + ") failed: " + std::strerror(errno));
        c->reserve(GRAIN);
        c->barrier();
        for (int p = 0; p < nranks; ++p) {
            if (p == rank) continue;
            const std::string theirs = c->base + "." + std::to_string(p);
            c->fd[p] = shm_open(theirs.c_str(), O_RDONLY, 0600);
            if (c->fd[p] < 0) c->die("shm_open(" + theirs + ") failed: " + std::strerror(errno));
        }
        c->barrier();
        shm_unlink(mine.c_str());                        // open everywhere: the names can go, the memory lives as long as the mappings
        if (rank == 0) shm_unlink(c->base.c_str());
    } catch (const std::exception &e) {
        return fail(e.what());
    }
    return std::unique_ptr<Comm>(c.release());
}

} // namespace saena_host

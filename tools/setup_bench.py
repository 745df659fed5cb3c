"""Row-distributed host setup over the native shared-memory communicator, timed per phase (development aid):
    python tools/setup_bench.py m nranks [threads]
Host library only (no GPU): the Galerkin products run on the host kernel."""
import multiprocessing as mp
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def worker(rank, world, name, m, threads):
    sys.path.insert(0, ROOT)
    os.environ["SAENA_SETUP_THREADS"] = str(threads)
    os.environ["SAENA_SETUP_TIMING"] = "1"
    from saena_amd import host
    L = host.load("host")
    comm = host.Comm("host", "shm", (name, rank, world)) if world > 1 else host.Comm("host", "self")
    t0 = time.time()
    A = host.Matrix(comm).laplacian3D(m).assemble()
    t1 = time.time()
    S = host.AmgSolver(A, host.options(L, **host.OPTIONS001))
    t2 = time.time()
    if rank == 0:
        print(f"assemble {t1 - t0:.2f} s, setup {t2 - t1:.2f} s, levels {S.num_levels}: rows {[S.level_info(l)['rows'] for l in range(S.num_levels)]}", flush=True)


if __name__ == "__main__":
    m, world = int(sys.argv[1]), int(sys.argv[2])
    threads = int(sys.argv[3]) if len(sys.argv) > 3 else max(1, (os.cpu_count() or 8) // world)
    ctx = mp.get_context("spawn")
    ps = [ctx.Process(target=worker, args=(r, world, f"sb{os.getpid()}", m, threads)) for r in range(world)]
    for p in ps:
        p.start()
    for p in ps:
        p.join()

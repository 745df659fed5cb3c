"""The native shared-memory communicator of the setup (saena_amd/csrc/host/shm_comm.cpp; the reference: MPI collectives,
src/saena_matrix_setup.cpp:953,1030,1082,1086) at world sizes 2-4: ragged all-to-all with blocks from 0 bytes to beyond the
segment's growth step, all-gather, reductions in rank order, a failing rank; then a whole assemble over it."""
import multiprocessing as mp
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _payload(src, dst, world, big):
    """bytes rank src sends to rank dst: ragged, empty for some pairs, one block beyond the 2 MiB growth step when `big`"""
    n = (7919 * (src + 1) * (dst + 2)) % 5000
    if (src + dst) % 3 == 0:
        n = 0
    if big and src == 1 and dst == 0:
        n = 70 * 2 ** 20 + 13                    # grows the segment and takes the give-the-pages-back path (> 64 MiB)
    rng = np.random.default_rng(1000 * src + dst)
    return rng.integers(0, 256, size=n, dtype=np.uint8)


def _worker(rank, world, name, q, big):
    sys.path.insert(0, ROOT)
    try:
        import ctypes as C
        from saena_amd import host
        L = host.load("host")
        comm = host.Comm("host", "shm", (name, rank, world))
        lib = L
        lib.saena_comm_test_alltoallv.restype = C.c_int
        for rep in range(3):                         # repeated exchanges re-use (and re-grow) the segments
            send = [_payload(rank, d, world, big and rep == 1) for d in range(world)]
            sc = np.array([len(b) for b in send], dtype=np.uint64)
            rc = np.array([len(_payload(s, rank, world, big and rep == 1)) for s in range(world)], dtype=np.uint64)
            sd = np.concatenate([[0], np.cumsum(sc)[:-1]]).astype(np.uint64)
            rd = np.concatenate([[0], np.cumsum(rc)[:-1]]).astype(np.uint64)
            sbuf = np.concatenate(send) if sc.sum() else np.zeros(1, np.uint8)
            rbuf = np.zeros(max(1, int(rc.sum())), np.uint8)
            st = lib.saena_comm_test_alltoallv(comm.h, sbuf.ctypes.data_as(C.c_void_p), sc.ctypes.data_as(C.c_void_p), sd.ctypes.data_as(C.c_void_p),
                                               rbuf.ctypes.data_as(C.c_void_p), rc.ctypes.data_as(C.c_void_p), rd.ctypes.data_as(C.c_void_p))
            assert st == 0, L.saena_last_error().decode()
            for s in range(world):
                want = _payload(s, rank, world, big and rep == 1)
                got = rbuf[int(rd[s]):int(rd[s]) + len(want)]
                assert np.array_equal(got, want), f"rep {rep}: block from rank {s}"
        # reductions: every rank gets the same bits (the sum runs in rank order everywhere)
        v = np.array([0.1 * (rank + 1), 1e16 if rank == 0 else 1.0, -1e16 if rank == world - 1 else 0.5])
        lib.saena_comm_test_allreduce_f64.restype = C.c_int
        assert lib.saena_comm_test_allreduce_f64(comm.h, v.ctypes.data_as(C.c_void_p), 3) == 0
        iv = np.array([rank + 1, 2 ** 40 + rank], dtype=np.int64)
        lib.saena_comm_test_allreduce_i64.restype = C.c_int
        assert lib.saena_comm_test_allreduce_i64(comm.h, iv.ctypes.data_as(C.c_void_p), 2) == 0
        assert iv[0] == world * (world + 1) // 2 and iv[1] == world * 2 ** 40 + world * (world - 1) // 2
        # a matrix assembled over it: the partition and the operator's global facts are those of one rank
        A = host.Matrix(comm).laplacian3D(14).assemble()
        q.put((rank, "ok", v.tobytes(), A.split.tolist(), int(A.nnz), int(A.num_local_rows)))
    except BaseException as e:      # noqa
        import traceback
        q.put((rank, "".join(traceback.format_exception(type(e), e, e.__traceback__)), b"", None, 0, 0))


@pytest.mark.parametrize("world,big", [(2, True), (3, False), (4, False)])
def test_collectives_and_assemble_over_shared_memory(world, big):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    name = f"t{os.getpid()}_{world}"
    procs = [ctx.Process(target=_worker, args=(r, world, name, q, big)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(30)
    res.sort()
    for r, status, *_ in res:
        assert status == "ok", f"rank {r}: {status}"
    assert len({x[2] for x in res}) == 1, "the reduction must give every rank the same bits"
    want = sum(0.1 * (r + 1) for r in range(world))
    got = np.frombuffer(res[0][2])
    assert abs(got[0] - want) < 1e-15
    assert len({tuple(x[3]) for x in res}) == 1
    sys.path.insert(0, ROOT)
    from saena_amd import host
    A1 = host.Matrix(host.Comm("host", "self")).laplacian3D(14).assemble()
    assert res[0][4] == A1.nnz and sum(x[5] for x in res) == A1.num_rows
    assert not [f for f in os.listdir("/dev/shm") if f.startswith(f"saena_{name}")], "the segments' names must be gone once every rank has attached"


def _lonely(name, q):
    sys.path.insert(0, ROOT)
    os.environ["SAENA_SHM_TIMEOUT"] = "2"
    from saena_amd import host
    try:
        host.Comm("host", "shm", (name, 0, 2))       # rank 1 never comes
        q.put("attached")
    except Exception as e:      # noqa
        q.put(f"error: {e}")


def test_a_missing_rank_is_a_timeout_not_a_hang():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    name = f"lonely{os.getpid()}"
    p = ctx.Process(target=_lonely, args=(name, q))
    p.start()
    msg = q.get(timeout=60)
    p.join(10)
    assert msg.startswith("error:") and "timed out" in msg, msg
    for f in os.listdir("/dev/shm"):                 # this job died before attaching: its names are still there -- clean up
        if f.startswith(f"saena_{name}"):
            os.unlink(os.path.join("/dev/shm", f))


def _attach_and_sum(rank, world, name, q, delay=0.0, limit_mb=0, big_mb=0):
    sys.path.insert(0, ROOT)
    os.environ["SAENA_SHM_TIMEOUT"] = "20"
    try:
        import ctypes as C
        import time
        if limit_mb:                                 # a file-size limit stands in for a /dev/shm that is too small
            import resource
            import signal
            signal.signal(signal.SIGXFSZ, signal.SIG_IGN)
            resource.setrlimit(resource.RLIMIT_FSIZE, (limit_mb << 20, limit_mb << 20))
        time.sleep(delay)
        from saena_amd import host
        L = host.load("host")
        comm = host.Comm("host", "shm", (name, rank, world))
        iv = np.array([rank + 1], dtype=np.int64)
        L.saena_comm_test_allreduce_i64.restype = C.c_int
        assert L.saena_comm_test_allreduce_i64(comm.h, iv.ctypes.data_as(C.c_void_p), 1) == 0, L.saena_last_error().decode()
        if big_mb:                                   # every rank sends big_mb MiB to its neighbour
            n = big_mb << 20
            sc = np.zeros(world, np.uint64); sc[(rank + 1) % world] = n
            rc = np.zeros(world, np.uint64); rc[(rank - 1) % world] = n
            z = np.zeros(world, np.uint64)
            sbuf, rbuf = np.ones(n, np.uint8), np.zeros(n, np.uint8)
            L.saena_comm_test_alltoallv.restype = C.c_int
            st = L.saena_comm_test_alltoallv(comm.h, sbuf.ctypes.data_as(C.c_void_p), sc.ctypes.data_as(C.c_void_p), z.ctypes.data_as(C.c_void_p),
                                             rbuf.ctypes.data_as(C.c_void_p), rc.ctypes.data_as(C.c_void_p), z.ctypes.data_as(C.c_void_p))
            if st != 0:
                q.put((rank, "error: " + L.saena_last_error().decode()))
                return
        q.put((rank, int(iv[0])))
    except BaseException as e:      # noqa
        q.put((rank, f"error: {e}"))


def _killed_rank0(name):
    sys.path.insert(0, ROOT)
    from saena_amd import host
    host.Comm("host", "shm", (name, 0, 2))           # waits for a rank 1 that never comes; the parent kills it


def test_a_leftover_control_block_of_a_killed_job_does_not_trap_the_next_job():
    """Round-3 advisor finding: a rank != 0 could open the control block a killed job left under the same name a moment before
    rank 0 replaced it, pass the magic check and sit in the dead block's barrier until the 900 s timeout.  Rank 0 now marks a
    leftover failed before it replaces it and ranks only trust a block once ITS rank 0 has counted them in."""
    import time
    ctx = mp.get_context("spawn")
    name = f"stale{os.getpid()}"
    p = ctx.Process(target=_killed_rank0, args=(name,))
    p.start()
    t0 = time.time()
    while not os.path.exists(f"/dev/shm/saena_{name}") and time.time() - t0 < 60:
        time.sleep(0.05)
    time.sleep(0.5)                                   # (initialised: the magic is set, nobody marked it failed)
    p.kill()
    p.join(10)
    assert os.path.exists(f"/dev/shm/saena_{name}"), "the killed job must have left its control block behind for this test"
    q = ctx.Queue()
    # rank 1 comes FIRST and finds the leftover; rank 0 arrives a second later and replaces it
    procs = [ctx.Process(target=_attach_and_sum, args=(1, 2, name, q, 0.0)), ctx.Process(target=_attach_and_sum, args=(0, 2, name, q, 1.0))]
    for x in procs:
        x.start()
    res = sorted(q.get(timeout=60) for _ in range(2))
    for x in procs:
        x.join(10)
    assert res == [(0, 3), (1, 3)], res
    assert not [f for f in os.listdir("/dev/shm") if f.startswith(f"saena_{name}")]


def test_an_exchange_beyond_the_room_in_shared_memory_is_an_error_not_a_signal():
    """Round-3 advisor finding: the segment grew by ftruncate alone and tmpfs hands out pages at the first touch, so a /dev/shm
    smaller than an exchange ended the process with SIGBUS inside a copy.  The pages are now reserved (posix_fallocate) before
    anything is written and a refusal is reported through the API -- on every rank (the others see the failed flag)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    name = f"full{os.getpid()}"
    procs = [ctx.Process(target=_attach_and_sum, args=(r, 2, name, q, 0.0, 16, 40)) for r in range(2)]
    for x in procs:
        x.start()
    res = sorted(q.get(timeout=60) for _ in range(2))
    for x in procs:
        x.join(10)
        assert x.exitcode == 0, "no rank may die of a signal"
    assert all(isinstance(s, str) and s.startswith("error:") and "shared-memory communicator" in s for _, s in res), res
    assert any("File too large" in s or "No space" in s or "/dev/shm" in s for _, s in res), res

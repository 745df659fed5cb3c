"""Source compatibility of the public C++ surface (round-3 review, item 7): the reference's saena.hpp takes MPI_Comm
(/root/reference/include/saena.hpp:17,31,52, experiments/Poisson.cpp:16-262); include/saena_mpi.hpp + include/compat/ give
drivers written against it the same names, so that the reference's own driver file compiles and LINKS against libsaena_amd.so
without a source change.  The GPU test runs the repo's driver of the same flow under mpirun."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MPI = "/opt/conda"
HAVE_MPI = os.path.exists(os.path.join(MPI, "include", "mpi.h")) and os.path.exists(os.path.join(MPI, "lib", "libmpi.so"))
REF = "/root/reference/experiments"


def _compile_and_link(src, tmp_path, extra=()):
    obj = str(tmp_path / (os.path.basename(src) + ".o"))
    exe = str(tmp_path / (os.path.basename(src) + ".exe"))
    inc = ["-I" + os.path.join(ROOT, "include", "compat"), "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(MPI, "include")]
    c = subprocess.run(["g++", "-std=c++17", "-fopenmp", "-w", *inc, *extra, "-c", src, "-o", obj], capture_output=True, text=True, timeout=300)
    assert c.returncode == 0, c.stderr[-3000:]
    lib = os.path.join(ROOT, "saena_amd")
    ln = subprocess.run(["g++", "-fopenmp", "-o", exe, obj, "-L" + lib, "-lsaena_amd", os.path.join(MPI, "lib", "libmpi.so"),
                         "-Wl,-rpath," + lib, "-Wl,-rpath,/opt/rocm/lib"], capture_output=True, text=True, timeout=300)
    assert ln.returncode == 0, ln.stderr[-3000:]
    return exe


@pytest.mark.skipif(not HAVE_MPI, reason="no MPI in this image")
@pytest.mark.skipif(not os.path.isdir(REF), reason="/root/reference is not here (GPU box)")
@pytest.mark.parametrize("driver", ["Poisson.cpp", "banded.cpp", "profile_file.cpp"])      # (Poisson.cpp and profile_file.cpp: the two experiments the reference's CMake builds)
def test_the_reference_s_own_driver_compiles_and_links_unchanged(driver, tmp_path):
    """the file is read where it lies under /root/reference; nothing of it is copied.  Every name it uses resolves against
    include/compat + include/saena_mpi.hpp and every symbol against libsaena_amd.so (an undefined one fails the link)."""
    assert os.path.exists(os.path.join(ROOT, "saena_amd", "libsaena_amd.so")), "build first (__graft_entry__.build())"
    _compile_and_link(os.path.join(REF, driver), tmp_path)


RHS_READER = r"""
#include "saena_mpi.hpp"
int main(int argc, char **argv) {
    MPI_Init(&argc, &argv);
    int rank = 0, np = 1;
    MPI_Comm_rank(MPI_COMM_WORLD, &rank); MPI_Comm_size(MPI_COMM_WORLD, &np);
    const int n = std::atoi(argv[2]);
    std::vector<index_t> split((size_t)np + 1);
    for (int r = 0; r <= np; ++r) split[(size_t)r] = (index_t)((long)n * r / np);
    const nnz_t mine = split[(size_t)rank + 1] - split[(size_t)rank];
    value_t *v = saena_aligned_alloc<value_t>(mine);
    assert(v && ((size_t)v & 63) == 0);
    if (read_from_file_rhs(v, split, argv[1], MPI_COMM_WORLD) != 0) return 3;
    for (int r = 0; r < np; ++r) {
        if (r == rank) for (nnz_t i = 0; i < mine; ++i) printf("%d %.17g\n", (int)(split[(size_t)rank] + i), v[i]);
        fflush(stdout);
        MPI_Barrier(MPI_COMM_WORLD);
    }
    saena_free(v);
    MPI_Finalize();
    return 0;
}
"""


@pytest.mark.skipif(not HAVE_MPI, reason="no MPI in this image")
@pytest.mark.parametrize("ext", ["txt", "bin"])
def test_read_from_file_rhs_like_the_reference(ext, tmp_path):
    """the rhs reader of the reference's file-driven experiment (src/aux_functions.cpp:347-497): a text vector file (comments, size,
    "row value" lines in any order, rows from 1) is turned into <name>.bin by rank 0 and every rank reads its slice; a .bin file is
    read as it is.  Two ranks under mpirun."""
    import numpy as np
    n = 37
    want = np.sin(0.3 * np.arange(n)) * 1e3
    if ext == "txt":
        order = np.random.default_rng(5).permutation(n)
        with open(tmp_path / "v.txt", "w") as f:
            f.write("% a comment line\n% another\n" + str(n) + "\n")
            for i in order:
                f.write("%d %.17g\n" % (i + 1, want[i]))
    else:
        want.astype(np.float64).tofile(tmp_path / "v.bin")
    src = tmp_path / "rhs_reader.cpp"
    src.write_text(RHS_READER)
    exe = _compile_and_link(str(src), tmp_path)
    env = dict(os.environ, LD_LIBRARY_PATH="/usr/lib/x86_64-linux-gnu:" + os.path.join(MPI, "lib") + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    out = subprocess.run([os.path.join(MPI, "bin", "mpirun"), "-np", "2", exe, str(tmp_path / f"v.{ext}"), str(n)], capture_output=True, text=True,
                         timeout=120, env=env)
    assert out.returncode == 0, out.stderr[-2000:] + out.stdout[-500:]
    got = np.full(n, np.nan)
    for ln in out.stdout.splitlines():
        i, v = ln.split()
        got[int(i)] = float(v)
    np.testing.assert_array_equal(got, want)
    assert os.path.exists(tmp_path / "v.bin")


@pytest.mark.skipif(not HAVE_MPI, reason="no MPI in this image")
def test_the_mpi_idiom_driver_of_the_repo_builds(tmp_path):
    _compile_and_link(os.path.join(ROOT, "examples", "poisson_mpi.cpp"), tmp_path)
    assert os.path.exists(os.path.join(ROOT, "examples", "poisson_mpi")), "the Makefile builds it where the image has an MPI"


OPTIONS001 = ('<?xml version="1.0" encoding="utf-8" ?>\n<SAENA>\n    <OPTIONS\n\tsolver_max_iter="50"\n\tsolver_tol="1e-8"\n'
              '\tsmoother="jacobi"\n\tpreSmooth="3"\n\tpostSmooth="3"\n\tPSmoother="jacobi"\n\tconn_str="0.2"\n\tdynamic_levels="1"\n'
              '\tmax_level="20"\n\tfloat_level="3"\n\tfilter_thre="1e-14"\n\tfilter_max="1e-8"\n\tfilter_start="1"\n\tfilter_rate="2"\n'
              '\tswitch_to_dense="0"\n\tdense_thre="0.1"\n\tdense_sz_thre="5000"\n\tpetsc=""/>\n</SAENA>\n')


@pytest.mark.gpu
@pytest.mark.skipif(not HAVE_MPI, reason="no MPI in this image")
def test_mpi_idiom_driver_runs_and_prints_the_reference_line(tmp_path):
    """mpirun -np 1 examples/poisson_mpi 32 options001.xml: saena::matrix(MPI_Comm) brings the GPU runtime up over the MPI job; the
    printed residuals are the reference's (7 iterations, 7.227341e+03 -> 2.246251e-05)"""
    exe = os.path.join(ROOT, "examples", "poisson_mpi")
    assert os.path.exists(exe), "build first (__graft_entry__.build())"
    xml = tmp_path / "options001.xml"
    xml.write_text(OPTIONS001)
    mpirun = shutil.which("mpirun") or os.path.join(MPI, "bin", "mpirun")
    env = dict(os.environ, LD_LIBRARY_PATH="/usr/lib/x86_64-linux-gnu:" + os.path.join(MPI, "lib") + ":" + os.environ.get("LD_LIBRARY_PATH", ""),
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([mpirun, "-np", "1", exe, "32", str(xml)], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    txt = out.stdout
    assert re.search(r"initial residual\s+= 7\.227341e\+03", txt), txt
    assert re.search(r"stopped at iteration\s+= 7", txt), txt
    assert re.search(r"final absolute residual = 2\.24625\de-05", txt), txt
    assert "Setup:" in txt and "Solve:" in txt and "solve_pCG profile: 7 iterations" in txt
    assert len(re.findall(r"matvec level \d+: ", txt)) == 5, txt


def _run_reference_driver(exe, args, tmp_path, np_=1):
    mpirun = shutil.which("mpirun") or os.path.join(MPI, "bin", "mpirun")
    env = dict(os.environ, LD_LIBRARY_PATH="/usr/lib/x86_64-linux-gnu:" + os.path.join(MPI, "lib") + ":" + os.environ.get("LD_LIBRARY_PATH", ""),
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    if np_ > 1:                                            # several ranks on this ONE card: no RCCL (it refuses that), MPI carries halos and collectives
        env.update(SAENA_MPI_HOST_TRANSPORT="1", SAENA_DEVICE="0")
    out = subprocess.run([mpirun, "-np", str(np_), exe, *args], capture_output=True, text=True, timeout=600, env=env, cwd=str(tmp_path))
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    return out.stdout


REF_DRIVERS = os.path.join(ROOT, "oracle", "_ref")


@pytest.mark.gpu
@pytest.mark.skipif(not HAVE_MPI, reason="no MPI in this image")
@pytest.mark.skipif(not os.path.exists(os.path.join(REF_DRIVERS, "ref_driver_poisson")), reason="oracle/_ref/ref_driver_* not built (make -C oracle ref)")
def test_the_reference_s_own_poisson_driver_runs_on_the_card(tmp_path):
    """oracle/_ref/ref_driver_poisson = the reference's experiments/Poisson.cpp, compiled UNCHANGED in the build container against
    include/compat + include/saena_mpi.hpp and linked against libsaena_amd.so (oracle/ref/Makefile): it runs its flow on the MI355X path
    and prints the reference's own residual line for 32^3 (7 iterations, 7.227341e+03 -> 2.246251e-05)."""
    xml = tmp_path / "options001.xml"
    xml.write_text(OPTIONS001)
    txt = _run_reference_driver(os.path.join(REF_DRIVERS, "ref_driver_poisson"), ["32", str(xml)], tmp_path)
    assert re.search(r"initial residual\s+= 7\.227341e\+03", txt), txt
    assert re.search(r"stopped at iteration\s+= 7", txt), txt
    assert re.search(r"final absolute residual = 2\.24625\de-05", txt), txt
    # ... and BASELINE configs[1]'s size: the line the reference prints for 128^3 (9 iterations, 5.992963e+04 -> 5.355578e-05)
    txt = _run_reference_driver(os.path.join(REF_DRIVERS, "ref_driver_poisson"), ["128", str(xml)], tmp_path)
    assert re.search(r"initial residual\s+= 5\.992963e\+04", txt), txt
    assert re.search(r"stopped at iteration\s+= 9", txt), txt
    assert re.search(r"final absolute residual = 5\.35557\de-05", txt), txt


@pytest.mark.gpu
@pytest.mark.skipif(not HAVE_MPI, reason="no MPI in this image")
@pytest.mark.skipif(not os.path.exists(os.path.join(REF_DRIVERS, "ref_driver_profile_file")), reason="oracle/_ref/ref_driver_* not built (make -C oracle ref)")
@pytest.mark.parametrize("ranks", [1, 3])
def test_the_reference_s_own_file_driver_runs_on_the_card(ranks, tmp_path):
    """oracle/_ref/ref_driver_profile_file = the reference's experiments/profile_file.cpp (matrix and right-hand side from files:
    BASELINE configs[4]'s driver), compiled unchanged like the one above: a MatrixMarket file (7-point Laplacian of 12^3) with a
    text rhs file -> read_file, read_from_file_rhs, saena::vector, set_matrix, 15 x solve_pCG, solve_pCG_profile, profile_matvecs."""
    import numpy as np
    m = 12                                                  # the 7-point Laplacian of an m^3 grid (Dirichlet rows eliminated), as a MatrixMarket file
    n = m ** 3
    idx = np.arange(n).reshape(m, m, m)
    rows, cols, vals = [idx.ravel()], [idx.ravel()], [np.full(n, 6.0)]
    for ax in range(3):
        a = np.take(idx, np.arange(m - 1), axis=ax).ravel()
        b = np.take(idx, np.arange(1, m), axis=ax).ravel()
        rows += [a, b]; cols += [b, a]; vals += [np.full(a.size, -1.0)] * 2
    rows, cols, vals = np.concatenate(rows), np.concatenate(cols), np.concatenate(vals)
    mtx = str(tmp_path / "lap12.mtx")
    with open(mtx, "w") as f:
        f.write("%%MatrixMarket matrix coordinate real general\n%d %d %d\n" % (n, n, rows.size))
        for r, c, v in zip(rows, cols, vals):
            f.write("%d %d %.17g\n" % (r + 1, c + 1, v))
    rhs = np.sin(0.37 * np.arange(n)) + 1.0
    with open(tmp_path / "rhs.txt", "w") as f:
        f.write("% rhs of the file-driver test\n" + str(n) + "\n")
        for i in range(n):
            f.write("%d %.17g\n" % (i + 1, rhs[i]))
    xml = tmp_path / "options001.xml"
    xml.write_text(OPTIONS001)
    # (3 ranks: every rank reads its chunk of the matrix file and its slice of the rhs, MPI as the transport on this one card)
    txt = _run_reference_driver(os.path.join(REF_DRIVERS, "ref_driver_profile_file"), [mtx, str(tmp_path / "rhs.txt"), str(xml)], tmp_path, np_=ranks)
    assert "matrix file:" in txt and "rhs file:" in txt and "Setup:" in txt and "Solve:" in txt, txt
    m = re.findall(r"relative residual\s+= ([0-9.e+-]+)", txt)
    assert m and float(m[0]) < 1e-8, txt                   # solve_pCG converged to the options' tolerance
    assert "solve_pCG profile:" in txt and re.search(r"matvec level 0: ", txt), txt


GENERATORS = r"""
#include "saena_mpi.hpp"
// the public generators / checkers of the reference's saena.hpp that its drivers do not call (include/saena.hpp:271-298)
int main(int argc, char **argv) {
    MPI_Init(&argc, &argv);
    const bool gpu = argc > 1 && std::string(argv[1]) == "gpu";
    const index_t mx = 9, my = 7;
    std::vector<double> rhs;
    saena::laplacian2D_set_rhs(rhs, mx, my, MPI_COMM_WORLD);
    printf("RHS2D");
    for (double v : rhs) printf(" %.17g", v);
    printf("\n");
    std::vector<double> u(rhs.size());
    for (size_t i = 0; i < u.size(); ++i) u[i] = 0.01 * (double)i;
    double n2 = 0.0, n3 = 0.0;
    saena::laplacian2D_check_solution(u, mx, my, MPI_COMM_WORLD, &n2);
    std::vector<double> u3((size_t)5 * 4 * 3);
    for (size_t i = 0; i < u3.size(); ++i) u3[i] = 0.02 * (double)i - 0.3;
    saena::laplacian3D_check_solution(u3, 5, 4, 3, MPI_COMM_WORLD, &n3);
    printf("NORMS %.17g %.17g\n", n2, n3);
    std::vector<double> z((size_t)5 * 4 * 3, 1.0);
    saena::laplacian3D_set_rhs_zero(z, 5, 4, 3, MPI_COMM_WORLD);
    printf("ZERO");
    for (double v : z) printf(" %g", v);
    printf("\n");
    if (gpu) {
        saena::matrix A(MPI_COMM_WORLD);
        A.set_remove_boundary(false);
        saena::laplacian2D(&A, mx, my, false);
        printf("LAP2D rows %d nnz %ld\n", A.get_num_rows(), (long)A.get_nnz());
        std::vector<value_t> x((size_t)A.get_num_local_rows(), 1.0), y;
        A.matvec(x, y);                                    // row sums: 4 on a boundary row (its diagonal), 0 deep inside, > 0 next to the boundary
        printf("ROWSUM %.17g %.17g %.17g\n", y[0], y[(size_t)mx * 3 + 4], y[(size_t)mx + 1]);
        saena::matrix B(MPI_COMM_WORLD);
        B.set_remove_boundary(false);
        saena::random_symm_matrix(B, 60, 0.1f);
        printf("RANDSYMM rows %d nnz %ld\n", B.get_num_rows(), (long)B.get_nnz());
        std::vector<value_t> v;
        if (saena::read_vector_file(v, A, argv[2], MPI_COMM_WORLD) != 0) return 4;
        printf("VEC %zu %.17g %.17g\n", v.size(), v.front(), v.back());
        A.destroy(); B.destroy();
    }
    MPI_Finalize();
    return 0;
}
"""


def _generators_expected():
    import numpy as np
    mx, my = 9, 7
    i, j = np.meshgrid(np.arange(mx), np.arange(my))              # rows j, columns i: node = mx * j + i
    hx, hy = 1.0 / (mx - 1), 1.0 / (my - 1)
    rhs = (8 * np.pi * np.pi * np.sin(2 * np.pi * i * hx) * np.sin(2 * np.pi * j * hy)).ravel()
    u = 0.01 * np.arange(mx * my)
    n2 = np.sqrt(np.sum((u - (np.sin(2 * np.pi * i * hx) * np.sin(2 * np.pi * j * hy)).ravel()) ** 2))
    k3, j3, i3 = np.meshgrid(np.arange(3), np.arange(4), np.arange(5), indexing="ij")
    ex3 = (np.sin(2 * np.pi * i3 / 4.0) * np.sin(2 * np.pi * j3 / 3.0) * np.sin(2 * np.pi * k3 / 2.0)).ravel()
    u3 = 0.02 * np.arange(60) - 0.3
    n3 = np.sqrt(np.sum((u3 - ex3) ** 2) / 60.0)
    zero = np.ones((3, 4, 5)); zero[0] = zero[-1] = 0; zero[:, 0] = zero[:, -1] = 0; zero[:, :, 0] = zero[:, :, -1] = 0
    return rhs, n2, n3, zero.ravel()


def _check_generator_output(txt):
    import numpy as np
    rhs, n2, n3, zero = _generators_expected()
    lines = {ln.split()[0]: ln.split()[1:] for ln in txt.splitlines() if ln and ln.split()[0] in ("RHS2D", "NORMS", "ZERO", "LAP2D", "ROWSUM", "RANDSYMM", "VEC")}
    np.testing.assert_allclose(np.array(lines["RHS2D"], float), rhs, rtol=1e-14, atol=1e-13)
    np.testing.assert_allclose(np.array(lines["NORMS"], float), [n2, n3], rtol=1e-13)
    np.testing.assert_array_equal(np.array(lines["ZERO"], float), zero)
    return lines


@pytest.mark.skipif(not HAVE_MPI, reason="no MPI in this image")
def test_the_other_public_generators_and_checkers_formulas(tmp_path):
    """laplacian2D_set_rhs / laplacian2D_check_solution / laplacian3D_check_solution / laplacian3D_set_rhs_zero of the reference's public
    header (src/aux_functions2.cpp:90-179, 702-763, 1249-1294): formulas against numpy (no device involved)"""
    src = tmp_path / "generators.cpp"
    src.write_text(GENERATORS)
    exe = _compile_and_link(str(src), tmp_path)
    env = dict(os.environ, LD_LIBRARY_PATH="/usr/lib/x86_64-linux-gnu:" + os.path.join(MPI, "lib") + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    out = subprocess.run([os.path.join(MPI, "bin", "mpirun"), "-np", "1", exe, "cpu"], capture_output=True, text=True, timeout=120, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    _check_generator_output(out.stdout)


@pytest.mark.gpu
@pytest.mark.skipif(not HAVE_MPI, reason="no MPI in this image")
def test_the_other_public_generators_on_the_card(tmp_path):
    """... and the ones that assemble a matrix: laplacian2D (5-point stencil, boundary nodes as rows of their own: aux_functions2.cpp:3-88),
    random_symm_matrix (:1384-1460), read_vector_file (:1462-1509)"""
    import numpy as np
    src = tmp_path / "generators.cpp"
    src.write_text(GENERATORS)
    exe = _compile_and_link(str(src), tmp_path)
    vec = np.cos(0.1 * np.arange(63))
    vec.tofile(tmp_path / "vec.bin")
    txt = _run_reference_driver(exe, ["gpu", str(tmp_path / "vec.bin")], tmp_path)
    lines = _check_generator_output(txt)
    # 9 x 7 grid: 63 rows; 35 interior nodes with a 5-point stencil less the couplings to boundary nodes, 28 boundary rows of one entry
    interior = 7 * 5
    assert lines["LAP2D"] == ["rows", "63", "nnz", str(28 + interior * 5 - 2 * 7 - 2 * 5)]
    hx, hy = 1 / 8.0, 1 / 6.0
    d = 2.0 * (hx / hy + hy / hx)
    np.testing.assert_allclose(np.array(lines["ROWSUM"], float), [d, 0.0, hx / hy + hy / hx], rtol=1e-13, atol=1e-13)
    assert lines["RANDSYMM"][:2] == ["rows", "60"] and int(lines["RANDSYMM"][3]) > 60
    assert lines["VEC"][0] == "63" and float(lines["VEC"][1]) == vec[0] and float(lines["VEC"][2]) == vec[-1]


@pytest.mark.gpu
@pytest.mark.skipif(not HAVE_MPI, reason="no MPI in this image")
@pytest.mark.skipif(not os.path.exists(os.path.join(REF_DRIVERS, "ref_driver_poisson")), reason="oracle/_ref/ref_driver_* not built (make -C oracle ref)")
@pytest.mark.parametrize("ranks", [2, 4])
def test_the_reference_s_own_poisson_driver_runs_multi_rank_under_mpirun(ranks, tmp_path):
    """`mpirun -np 2 / 4` of the reference's unchanged Poisson.cpp on this one card (SAENA_MPI_HOST_TRANSPORT=1: the device path's halos
    and reductions and the host setup's collectives ride on MPI, include/saena_mpi.hpp; everything above the transport is the
    multi-rank code: the reference's partitioner, the row-distributed setup, interior / boundary kernels, agglomerated coarse levels).
    Rank 0 prints the reference's residual line for 32^3 -- the same digits as at one rank."""
    xml = tmp_path / "options001.xml"
    xml.write_text(OPTIONS001)
    txt = _run_reference_driver(os.path.join(REF_DRIVERS, "ref_driver_poisson"), ["32", str(xml)], tmp_path, np_=ranks)
    assert re.search(r"initial residual\s+= 7\.227341e\+03", txt), txt
    assert re.search(r"stopped at iteration\s+= 7", txt), txt
    assert re.search(r"final absolute residual = 2\.24625\de-05", txt), txt
    if ranks == 4:                                         # BASELINE configs[1]'s size over 4 ranks: the reference's 128^3 line
        txt = _run_reference_driver(os.path.join(REF_DRIVERS, "ref_driver_poisson"), ["128", str(xml)], tmp_path, np_=ranks)
        assert re.search(r"initial residual\s+= 5\.992963e\+04", txt), txt
        assert re.search(r"stopped at iteration\s+= 9", txt), txt
        assert re.search(r"final absolute residual = 5\.3555[67]\de-05", txt), txt

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
@pytest.mark.parametrize("driver", ["Poisson.cpp", "banded.cpp"])
def test_the_reference_s_own_driver_compiles_and_links_unchanged(driver, tmp_path):
    """the file is read where it lies under /root/reference; nothing of it is copied.  Every name it uses resolves against
    include/compat + include/saena_mpi.hpp and every symbol against libsaena_amd.so (an undefined one fails the link)."""
    assert os.path.exists(os.path.join(ROOT, "saena_amd", "libsaena_amd.so")), "build first (__graft_entry__.build())"
    _compile_and_link(os.path.join(REF, driver), tmp_path)


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

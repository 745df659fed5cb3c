"""Development aid: the 3-rank library solve of tests/test_gpu_vcycle.py with fp64 and fp32 halos, printing the histories and
the kernel variant of every level's operators on every rank:  python tools/debug_mr_fp32.py [trials]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("SAENA_PLAN_CACHE", "off")


def main():
    from tests import test_gpu_vcycle as t
    for trial in range(int(sys.argv[1]) if len(sys.argv) > 1 else 2):
        for fl in (3, 0):
            res = t._run_transport(3, "jacobi", fl, "rows4096")
            _, it, hist, ok, it2, last2, ok2, rows, split, launches, owners, _u, variants = res[0]
            print("float_level", fl, "it", it, "ok", ok, "ok2", ok2, "it2", it2, "last2", last2, "hist", ["%.4e" % h for h in hist], flush=True)
            for r in range(3):
                print("   rank", r, "variants", res[r][12], flush=True)


if __name__ == "__main__":
    main()

"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle and
against the compiled-reference golden fixtures.  Run with `-m gpu` on an MI355X.

Tolerances (fp64):
  * SpMV / transfer, 1 lane per row: BIT-EXACT vs the oracle (same products,
    same sequential sum, no FMA on either side);
  * SpMV with G>1 lanes per row or a remote part: |y - y_ref| <= 1e-13 * (|A||x|);
  * smoothers (k <= 4 sweeps): relative l2 <= 1e-12; bit-exact at 1 lane/row, 1 rank.
"""
import glob
import os
import time

import numpy as np
import pytest

from oracle import oracle as orc
from tests import inputs, matrices, util

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TOL_SPMV = 1e-13
TOL_SMOOTH = 1e-12


@pytest.fixture(scope="module")
def capi():
    from saena_amd import capi as c
    c.init(0)
    return c


def abs_bound(entries, Mbig, x):
    b = np.zeros(Mbig)
    np.add.at(b, entries["row"], np.abs(entries["val"] * x[entries["col"]]))
    return b


def rel(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


def problems():
    out = {}
    for m in (8, 12, 20, 24):
        out[f"poisson{m}"] = orc.laplacian3d(m)
    out["band300_7"] = (orc.band_matrix(300, 7), 300)
    out["band64_63"] = (orc.band_matrix(64, 63), 64)
    out["band3000_1400"] = (orc.band_matrix(3000, 1400), 3000)      # rows longer than the LDS tile: long-row path
    rng = np.random.default_rng(7)
    M = 5000                                                         # irregular: empty rows, 1..300 nnz/row
    lens = rng.choice([0, 1, 2, 3, 5, 9, 17, 40, 120, 300], size=M)
    rows = np.repeat(np.arange(M), lens)
    cols = np.concatenate([rng.choice(M, size=k, replace=False) for k in lens]) if rows.size else np.zeros(0, int)
    vals = rng.standard_normal(rows.size)
    d = np.arange(M)
    rows = np.concatenate([rows, d]); cols = np.concatenate([cols, d]); vals = np.concatenate([vals, 50 + rng.random(M)])
    key = rows.astype(np.int64) * M + cols
    _, first = np.unique(key, return_index=True)
    out["irregular5000"] = (orc.coo_from_arrays(rows[first].astype(np.int32), cols[first].astype(np.int32), vals[first]), M)
    return out


PROBLEMS = None


def get_problem(name):
    global PROBLEMS
    if PROBLEMS is None:
        PROBLEMS = problems()
    return PROBLEMS[name]


NAMES = ["poisson8", "poisson12", "poisson20", "band300_7", "band64_63", "band3000_1400", "irregular5000"]


@pytest.mark.parametrize("name", NAMES)
def test_spmv_bitexact_one_lane(capi, name):
    entries, M = get_problem(name)
    A = orc.OracleOp(entries, M, M, orc.split_even(M, 1))
    G = util.gpu_operator(A)
    G.set_lanes_per_row(1)
    x = inputs.v2(M)
    want = A.matvec(x)
    dx, dy = capi.DeviceVector(M, x), capi.DeviceVector(M)
    G.spmv(dx, dy)
    got = dy.download()
    if name == "band3000_1400":        # long rows are tree-reduced by the whole workgroup
        assert np.all(np.abs(got - want) <= TOL_SPMV * abs_bound(entries, M, x))
    else:
        np.testing.assert_array_equal(got, want)
    # host-slice form of the reference seam (const value_t* v, value_t* w)
    np.testing.assert_array_equal(G.spmv_host(x), got)


@pytest.mark.parametrize("name", NAMES)
@pytest.mark.parametrize("lanes", [0, 2, 4, 8, 16, 32, 64])
def test_spmv_lane_variants(capi, name, lanes):
    entries, M = get_problem(name)
    A = orc.OracleOp(entries, M, M, orc.split_even(M, 1))
    G = util.gpu_operator(A)
    G.set_lanes_per_row(lanes)
    x = inputs.v2(M)
    want = A.matvec(x)
    dx, dy = capi.DeviceVector(M, x), capi.DeviceVector(M)
    G.spmv(dx, dy)
    got = dy.download()
    assert np.all(np.abs(got - want) <= TOL_SPMV * abs_bound(entries, M, x) + 1e-300)


def _sell_form(entries, M):
    """the library's rule for the sliced-ELLPACK form: slices of 64 rows padded to their longest row, <= 12 % padding ->
    "plain"; else, with the rows sorted by length (longest first) inside windows of 2048 rows, <= 5 % padding and at least
    4 entries per row -> "sorted" (k_sell<sorted>, opt-in: SAENA_SELL_SORTED=1); else None"""
    n = np.bincount(np.asarray(entries["row"]), minlength=M)
    padded = sum(64 * int(n[s:s + 64].max()) for s in range(0, M, 64))
    if padded <= 1.12 * len(entries):
        return "plain"
    if len(entries) < 4 * M:
        return None
    padded = 0
    for w in range(0, M, 2048):
        ns = np.sort(n[w:w + 2048])[::-1]
        padded += sum(64 * int(ns[s]) for s in range(0, len(ns), 64))
    return "sorted" if padded <= 1.05 * len(entries) else None


def _sell_eligible(entries, M):
    return _sell_form(entries, M) == "plain"


def _sellp_table(entries, M):
    """the library's rule for the row-pattern form: sliced-ELLPACK eligible, fewer than 65 536 distinct (length, columns relative
    to the row) patterns holding at most an eighth of the entries, and -- "narrow": all of them fit 4096 ints at longest-row + 1
    ints per pattern (ONE table, k_sellp / k_sellp2 with 256 threads); else "wide": the patterns that each group of 1024
    consecutive rows follows fit 8192 ints stored compactly -- a start offset, the length and the offsets per pattern, one spare
    int (k_sellp<wide>: a table per workgroup); else None"""
    if not _sell_eligible(entries, M):
        return None
    row, col = np.asarray(entries["row"]), np.asarray(entries["col"])
    order = np.lexsort((col, row))
    row, col = row[order], col[order]
    n = np.bincount(row, minlength=M)
    ptr = np.concatenate([[0], np.cumsum(n)])
    rowpat = [tuple(col[ptr[r]:ptr[r + 1]] - r) for r in range(M)]
    pats = set(rowpat)
    if len(pats) >= 65536 or sum(len(p) + 1 for p in pats) > len(row) // 8 + 65536:
        return None
    if len(pats) * (int(n.max()) + 1) <= 4096:
        return "narrow"
    for g in range(0, M, 1024):
        grp = set(rowpat[g:g + 1024])
        if len(grp) + 1 + sum(len(p) + 1 for p in grp) > 8192:
            return None
    return "wide"


def _sellp_eligible(entries, M):
    return _sellp_table(entries, M) == "narrow"


@pytest.mark.parametrize("name", NAMES)
@pytest.mark.parametrize("variant", [1, 2, 3, 4, 6, 7, 8, 9, 10, 11, 12, 14, 15, 16])
def test_kernel_variants(capi, name, variant, monkeypatch):
    """32 KiB tiles / vector CSR / 16-bit compressed columns (16 and 32 KiB tiles; long rows included) / wave-streamed
    long rows / compressed columns with the block's entries in column order (16 and 32 KiB tiles) / sliced ELLPACK /
    x in LDS: same results as the default kernel"""
    monkeypatch.setenv("SAENA_KEEP_HOST_VALUES", "1")      # the column-major form is built from a host copy of the values
    entries, M = get_problem(name)
    A = orc.OracleOp(entries, M, M, orc.split_even(M, 1))
    G = util.gpu_operator(A)
    if variant in (7, 8) and name == "band3000_1400" and variant == 7:
        with pytest.raises(capi.SgpuError, match="column-major"):      # rows longer than the 16 KiB tile: refused, not mis-computed
            G.set_variant(variant)
        return
    if variant == 9 and _sell_form(entries, M) != "plain":
        with pytest.raises(capi.SgpuError, match="sliced-ELLPACK"):    # uneven rows: more than 12 % padding, refused
            G.set_variant(variant)
        return
    if variant == 12:                     # sliced ELLPACK in LDS windows: refused where its padding exceeds 25 % (the library's own count)
        try:
            G.set_variant(variant)
        except capi.SgpuError as e:
            assert "sliced-ELLPACK-in-LDS" in str(e)
            return
        assert G.variant()[1] == "k_sellx"
    if variant == 14 and _sellp_table(entries, M) is not None:      # a lane per two rows: slices of 128 rows, refused where THEY pad more than 12 %
        try:
            G.set_variant(variant)
        except capi.SgpuError as e:
            assert "row-paired" in str(e)
            return
    table = _sellp_table(entries, M)
    if variant == 15 and table is not None:                # the row patterns with x in LDS: refused where windows and table do not fit (the library's own count)
        try:
            G.set_variant(variant)
        except capi.SgpuError as e:
            assert "x in LDS" in str(e)
            return
        assert G.variant()[1] == "k_sellpx"
    if variant in (11, 14, 15) and table is None:
        with pytest.raises(capi.SgpuError, match="row-pattern"):       # rows that follow no small set of patterns: refused
            G.set_variant(variant)
        return
    G.set_variant(variant)                # the compressed-column forms serve every one of these operators
    if variant in (3, 4):
        assert "k_csr_cc16" in G.variant()[1]
    if variant in (7, 8):
        assert "k_csr_cm" in G.variant()[1]
    if variant == 9:
        assert G.variant()[1] == "k_sell"
    if variant == 10:
        assert G.variant()[1] == "k_csr_xlds"
    if variant == 16:
        assert G.variant()[1] == "k_csr_xldsr"
    if variant == 11:
        assert G.variant()[1] == ("k_sellp" if table == "narrow" else "k_sellp<wide>")
    if variant == 14:
        assert G.variant()[1] == ("k_sellp2" if table == "narrow" else "k_sellp2<wide>")
    x, rhs = inputs.v2(M), inputs.rhs2(M)
    bound = abs_bound(entries, M, x)
    dx, dy, dr = capi.DeviceVector(M, x), capi.DeviceVector(M), capi.DeviceVector(M, rhs)
    for lanes in (1, 4, 16, 64):
        G.set_lanes_per_row(lanes)
        G.spmv(dx, dy)
        got, want = dy.download(), A.matvec(x)
        if (lanes == 1 or variant in (9, 11, 14, 15)) and variant not in (2, 6, 10, 12, 16) and (name != "band3000_1400" or variant in (1, 4, 8)):
            np.testing.assert_array_equal(got, want)          # stream variants keep the sequential row sum
        else:
            assert np.all(np.abs(got - want) <= TOL_SPMV * bound + 1e-300)
        du = capi.DeviceVector(M, x)
        G.jacobi(2, du, dr)
        assert rel(du.download(), A.jacobi(2, x, rhs)) <= TOL_SMOOTH
    # every fused epilogue of the variant: residual, Chebyshev (first and later step), u -= A e
    G.residual(dx, dr, dy)
    assert rel(dy.download(), A.residual(x, rhs)) <= TOL_SMOOTH
    A.set_eig(1.9371)
    du = capi.DeviceVector(M, x)
    G.chebyshev(3, 1.9371, du, dr)
    assert rel(du.download(), A.chebyshev(3, x, rhs)) <= TOL_SMOOTH
    du = capi.DeviceVector(M, rhs)
    G.prolong_correct(dx, du)
    assert rel(du.download(), rhs - A.matvec(x)) <= TOL_SMOOTH


def _uneven_rows_operator(M, N, seed, lengths, square=False):
    """a transfer-like operator: three row lengths in random order (plain slices pad > 40 %), columns around r * N / M; M is
    no multiple of 64 and spans several sort windows"""
    rng = np.random.default_rng(seed)
    lens = rng.choice(lengths, size=M, p=[0.3, 0.4, 0.3])
    rows = np.repeat(np.arange(M), lens)
    base = (np.arange(M) * (N / M)).astype(np.int64)
    cols = np.concatenate([np.sort(rng.choice(np.arange(max(0, b - 60), min(N, b + 60)), size=l, replace=False)) for b, l in zip(base, lens)])
    vals = np.sin(0.3 * rows + 0.7 * cols) + 1.5
    if square:                                            # a diagonal that lets the smoothers run: the row's own column, dominant
        keep = cols != rows
        rows, cols, vals = rows[keep], cols[keep], vals[keep]
        rows = np.concatenate([rows, np.arange(M)]); cols = np.concatenate([cols, np.arange(M)]); vals = np.concatenate([vals, np.full(M, 90.0)])
    return orc.coo_from_arrays(rows.astype(np.int32), cols.astype(np.int32), vals)


@pytest.mark.parametrize("M,N,lengths", [(9001, 2300, (12, 18, 27)), (6500, 40000, (4, 6, 9)), (4999, 4999, (12, 18, 27))])
def test_sliced_ellpack_with_rows_sorted_by_length(capi, M, N, lengths, monkeypatch):
    """k_sell<sorted> (round 4, OPT-IN with SAENA_SELL_SORTED=1: it lost to the tile kernels on the transfers it was built for,
    DESIGN 9): an operator whose uneven rows pad a plain sliced-ELLPACK layout beyond 12 % takes the layout
    with its rows sorted by length inside windows of 2048 rows (a permutation per window, <= 5 % padding).  Same sequential
    row sums as the reference's loop: every fused epilogue bit-identical to the oracle's / to the CSR kernel's at one lane per
    row (two positions per load at 12-27 entries per row, one at 4-9); the row-pattern forms refuse the operator."""
    square = M == N
    entries = _uneven_rows_operator(M, N, 11, lengths, square)
    assert _sell_form(entries, M) == "sorted"
    A = orc.OracleOp(entries, M, N, orc.split_even(M, 1), orc.split_even(N, 1), square=square)
    G = util.gpu_operator(A)
    with pytest.raises(capi.SgpuError, match="sliced-ELLPACK"):       # not asked for: refused as before
        G.set_variant(9)
    monkeypatch.setenv("SAENA_SELL_SORTED", "1")
    G = util.gpu_operator(A)
    G.set_variant(9)
    assert G.variant() == (9, "k_sell<sorted>")
    x = inputs.v2(N)
    dx, dy = capi.DeviceVector(N, x), capi.DeviceVector(M)
    G.spmv(dx, dy)
    np.testing.assert_array_equal(dy.download(), A.matvec(x))
    u = inputs.rhs2(M)                                     # u -= A e (the prolongation's epilogue)
    du = capi.DeviceVector(M, u)
    G.prolong_correct(dx, du)
    np.testing.assert_array_equal(du.download(), u - A.matvec(x))
    with pytest.raises(capi.SgpuError, match="row-pattern"):
        G.set_variant(11)
    if not square:
        return
    rhs = inputs.rhs2(M)
    dr = capi.DeviceVector(M, rhs)
    G.residual(dx, dr, dy)
    got = dy.download()
    H = util.gpu_operator(A)                               # the CSR kernel at one lane per row: the same sums in the same order
    H.set_variant(0); H.set_lanes_per_row(1)
    H.residual(dx, dr, dy)
    np.testing.assert_array_equal(got, dy.download())
    assert rel(got, A.residual(x, rhs)) <= TOL_SMOOTH
    for sweep in ("jacobi", "chebyshev"):
        d1, d2 = capi.DeviceVector(M, x), capi.DeviceVector(M, x)
        if sweep == "jacobi":
            G.jacobi(3, d1, dr); H.jacobi(3, d2, dr)
            assert rel(d1.download(), A.jacobi(3, x, rhs)) <= TOL_SMOOTH
        else:
            A.set_eig(1.9371)
            G.chebyshev(3, 1.9371, d1, dr); H.chebyshev(3, 1.9371, d2, dr)
            assert rel(d1.download(), A.chebyshev(3, x, rhs)) <= TOL_SMOOTH
        np.testing.assert_array_equal(d1.download(), d2.download())
    G.autotune()                                           # the plan-time choice runs with the form among its candidates
    G.spmv(dx, dy)
    assert np.all(np.abs(dy.download() - A.matvec(x)) <= TOL_SPMV * abs_bound(entries, M, x) + 1e-300)


@pytest.mark.parametrize("acc", ["lds", "global"])
def test_x_in_lds_longest_rows_first_over_several_windows(capi, acc, monkeypatch):
    """k_csr_xlds / k_csr_xldsr with a chunk's rows taken LONGEST FIRST (round 4: chunks that hold rows of more than 8x their mean
    length -- hub rows of an irregular operator) on rows that reach over three column windows: the permutation, the partial sums
    carried between windows (in LDS or in global memory) and the epilogues all go by the row, not by the position it is taken at;
    SAENA_XLDS_NATURAL_ORDER=1 gives the same sums bit for bit (the order rows are picked up in changes nothing else)."""
    if acc == "global":
        monkeypatch.setenv("SAENA_XLDS_GLOBAL_ACC", "1")
    M = N = 70000
    rng = np.random.default_rng(9)
    r = np.repeat(np.arange(M), 24)
    c = (r + np.tile(np.concatenate([k * 23000 + np.arange(8) for k in range(3)]), M)) % N
    hubs = np.arange(5, M, 997)                                    # a hub row every 997 rows: 1 500 entries over the same three windows
    hr = np.repeat(hubs, 1500)
    hc = (hr + np.tile(np.concatenate([k * 23000 + np.arange(500) for k in range(3)]), len(hubs))) % N
    key = np.unique(np.concatenate([r.astype(np.int64) * N + c, hr.astype(np.int64) * N + hc]))
    rows, cols = (key // N).astype(np.int32), (key % N).astype(np.int32)
    entries = orc.coo_from_arrays(rows, cols, np.sin(0.37 * rows + 0.11 * cols) + (rows == cols) * 400.0)
    A = orc.OracleOp(entries, M, M, orc.split_even(M, 1))
    x, rhs = inputs.v2(M), inputs.rhs2(M)
    bound = abs_bound(entries, M, x)
    dx, dy, dr = capi.DeviceVector(M, x), capi.DeviceVector(M), capi.DeviceVector(M, rhs)
    got = {}
    for order in ("longest-first", "natural"):
        if order == "natural":
            monkeypatch.setenv("SAENA_XLDS_NATURAL_ORDER", "1")
        G = util.gpu_operator(A)
        for variant, lanes in ((10, 8), (16, 8), (16, 16)):
            G.set_variant(variant); G.set_lanes_per_row(lanes)
            G.spmv(dx, dy)
            y = dy.download()
            assert np.all(np.abs(y - A.matvec(x)) <= TOL_SPMV * bound + 1e-300), (order, variant, lanes)
            if order == "natural":
                np.testing.assert_array_equal(y, got[(variant, lanes)])
            got[(variant, lanes)] = y
            du = capi.DeviceVector(M, x)
            G.jacobi(2, du, dr)
            assert rel(du.download(), A.jacobi(2, x, rhs)) <= TOL_SMOOTH, (order, variant, lanes)


def _transfer_like_operator(M, seed):
    """a prolongation-like M x M/2 operator of a structured grid: row r reads a few runs of columns that start at r // 2 (its
    aggregate) -- the same runs for rows of the same parity class, so the rows repeat a handful of patterns RELATIVE TO THEIR FIRST
    COLUMN and tens of thousands relative to the row index; 6-7 entries per row (plain slices pad < 12 %)"""
    N = M // 2 + 400
    shapes = [np.array([0, 1, 2, 130, 131, 260]), np.array([0, 1, 129, 130, 131, 259, 260]), np.array([0, 2, 3, 128, 130, 258])]
    r = np.arange(M)
    rows = np.concatenate([r[(r // 7) % 3 == k].repeat(len(shapes[k])) for k in range(3)])
    cols = np.concatenate([(r[(r // 7) % 3 == k][:, None] // 2 + shapes[k][None, :]).ravel() for k in range(3)])
    order = np.lexsort((cols, rows))
    rows, cols = rows[order], cols[order]
    vals = np.sin(0.3 * rows + 0.7 * cols) + 1.5
    return orc.coo_from_arrays(rows.astype(np.int32), cols.astype(np.int32), vals), N


def test_row_patterns_relative_to_the_first_column(capi, monkeypatch):
    """k_sellp<rowbase> (round 4): a transfer operator whose rows repeat relative to their first column gets the row-pattern form
    with that column kept per row (8 B per entry + 6 B per row instead of 10 + 2); same sequential sums -- bit-identical to the
    oracle's loop, to k_sell and to the CSR kernel at one lane per row; the row-paired and x-in-LDS pattern forms refuse it;
    SAENA_NO_SELLP_ROWBASE=1 leaves the operator without a pattern form as before."""
    M = 140001                                             # (more than 65 535 patterns relative to the row index: that form is refused first)
    entries, N = _transfer_like_operator(M, 3)
    A = orc.OracleOp(entries, M, N, orc.split_even(M, 1), orc.split_even(N, 1), square=False)
    G = util.gpu_operator(A)
    G.set_variant(11)
    assert G.variant() == (11, "k_sellp<rowbase>")
    x, u = inputs.v2(N), inputs.rhs2(M)
    dx, dy = capi.DeviceVector(N, x), capi.DeviceVector(M)
    G.spmv(dx, dy)
    got = dy.download()
    np.testing.assert_array_equal(got, A.matvec(x))
    du = capi.DeviceVector(M, u)
    G.prolong_correct(dx, du)
    np.testing.assert_array_equal(du.download(), u - A.matvec(x))
    for v in (9, 0):
        H = util.gpu_operator(A)
        H.set_variant(v); H.set_lanes_per_row(1)
        H.spmv(dx, dy)
        np.testing.assert_array_equal(dy.download(), got)
    for v, what in ((14, "row-paired"), (15, "x in LDS")):
        with pytest.raises(capi.SgpuError, match=what):
            G.set_variant(v)
    G.autotune()                                           # the plan-time choice: k_sellp<rowbase> or k_sell, the same sums either way
    assert G.variant()[1] in ("k_sellp<rowbase>", "k_sell")
    G.spmv(dx, dy)
    np.testing.assert_array_equal(dy.download(), got)
    monkeypatch.setenv("SAENA_NO_SELLP_ROWBASE", "1")
    G2 = util.gpu_operator(A)
    with pytest.raises(capi.SgpuError, match="row-pattern"):
        G2.set_variant(11)


def _clustered_operator(M, N, clusters, seed):
    """rectangular operator whose rows touch `clusters` runs of 8 consecutive columns spread over all N columns; the 64
    rows of a group share their clusters (a row block then touches a few hundred short segments far apart)"""
    rng = np.random.default_rng(seed)
    rows, cols = [], []
    for g0 in range(0, M, 64):
        starts = np.sort(rng.choice((N - 8) // 8, size=clusters, replace=False)) * 8 + rng.integers(0, 8)
        c = (starts[:, None] + np.arange(8)[None, :]).ravel()
        c = c[c < N]
        for r in range(g0, min(g0 + 64, M)):
            rows.append(np.full(len(c), r)); cols.append(c)
    rows, cols = np.concatenate(rows), np.concatenate(cols)
    vals = np.sin(0.3 * rows + 0.7 * cols) + 1.5
    return orc.coo_from_arrays(rows.astype(np.int32), cols.astype(np.int32), vals)


@pytest.mark.parametrize("clusters,split_expected", [(40, "wider"), (4, "4+12")])
def test_compressed_columns_slot_offset_split(capi, clusters, split_expected):
    """The 16-bit column form picks its slot/offset split per operator: 4+12 bits when a block's columns sit in <= 16
    segments of 4096, more slots of smaller segments (down to 8+8) when they sit in many short clusters far apart -- and
    an operator whose blocks touch more than 256 segments of 256 columns is refused, not mis-computed."""
    M, N = 2048, 1 << 20
    entries = _clustered_operator(M, N, clusters, 5)
    A = orc.OracleOp(entries, M, N, orc.split_even(M, 1), orc.split_even(N, 1), square=False)
    G = util.gpu_operator(A)
    x = inputs.v2(N)
    want = A.matvec(x)
    dx, dy = capi.DeviceVector(N, x), capi.DeviceVector(M)
    for variant in (3, 4):
        G.set_variant(variant)
        name = G.variant()[1]
        if split_expected == "4+12":
            assert "4+12" in name, name
        else:
            assert "k_csr_cc16" in name and "4+12" not in name, name
        G.set_lanes_per_row(1)
        G.spmv(dx, dy)
        np.testing.assert_array_equal(dy.download(), want)      # same products, same sequential row sum as the 32-bit kernel
    # uniformly random columns over 2^20: > 256 segments per block whatever the split
    rng = np.random.default_rng(3)
    rows = np.repeat(np.arange(M), 24)
    cols = rng.integers(0, N, size=rows.size)
    key = np.unique(rows.astype(np.int64) * N + cols)
    e2 = orc.coo_from_arrays((key // N).astype(np.int32), (key % N).astype(np.int32), np.ones(key.size))
    A2 = orc.OracleOp(e2, M, N, orc.split_even(M, 1), orc.split_even(N, 1), square=False)
    G2 = util.gpu_operator(A2)
    with pytest.raises(capi.SgpuError, match="256 column segments"):
        G2.set_variant(3)
    G2.autotune()                                               # the plan-time choice simply leaves the form out
    G2.spmv(dx, dy)
    assert np.all(np.abs(dy.download() - A2.matvec(x)) <= TOL_SPMV * 24 * np.max(np.abs(x)))


@pytest.mark.parametrize("acc", ["lds", "global"])
def test_x_in_lds_column_windows(capi, acc, monkeypatch):
    """k_csr_xlds on rows that reach over more columns than one LDS window holds: three clusters of 40 entries 25 000
    columns apart (three windows per row chunk, partial sums carried between them -- in LDS behind a window shortened by the
    chunk's rows (round 4), or through global memory as operators with many rows per chunk still do: SAENA_XLDS_GLOBAL_ACC=1),
    against the oracle; an operator whose rows scatter over 2^20 columns is refused, not mis-computed."""
    if acc == "global":
        monkeypatch.setenv("SAENA_XLDS_GLOBAL_ACC", "1")
    M = N = 80000
    r = np.repeat(np.arange(M), 120)
    c = (r + np.tile(np.concatenate([k * 25000 + np.arange(40) for k in range(3)]), M)) % N
    key = np.unique(r.astype(np.int64) * N + c)
    rows, cols = (key // N).astype(np.int32), (key % N).astype(np.int32)
    entries = orc.coo_from_arrays(rows, cols, np.sin(0.37 * rows + 0.11 * cols) + (rows == cols) * 150.0)
    A = orc.OracleOp(entries, M, M, orc.split_even(M, 1))
    G = util.gpu_operator(A)
    G.set_variant(10)
    assert G.variant()[1] == "k_csr_xlds"
    x, rhs = inputs.v2(M), inputs.rhs2(M)
    dx, dy, dr = capi.DeviceVector(M, x), capi.DeviceVector(M), capi.DeviceVector(M, rhs)
    bound = abs_bound(entries, M, x)
    for lanes in (8, 16, 64):
        G.set_lanes_per_row(lanes)
        G.spmv(dx, dy)
        assert np.all(np.abs(dy.download() - A.matvec(x)) <= TOL_SPMV * bound + 1e-300)
        du = capi.DeviceVector(M, x)
        G.jacobi(2, du, dr)
        assert rel(du.download(), A.jacobi(2, x, rhs)) <= TOL_SMOOTH
    # the same windows with FOUR rows per group step (k_csr_xldsr, round 4: short row pieces -- a group streams the quads of four
    # consecutive rows as one sequence and runs their four epilogues side by side; partial sums carried between the windows)
    G.set_variant(16)
    assert G.variant()[1] == "k_csr_xldsr"
    for lanes in (4, 8, 16):
        G.set_lanes_per_row(lanes)
        G.spmv(dx, dy)
        assert np.all(np.abs(dy.download() - A.matvec(x)) <= TOL_SPMV * bound + 1e-300)
        du = capi.DeviceVector(M, x)
        G.jacobi(3, du, dr)
        assert rel(du.download(), A.jacobi(3, x, rhs)) <= TOL_SMOOTH
    G.set_variant(10)
    # the same windows with a LANE per row piece (k_sellx: pieces sorted by length inside every (chunk, window) block, 64 to a
    # slice): every fused epilogue, a few rows emptied so that some pieces are missing and some rows have none at all
    import os
    os.environ["SAENA_KEEP_HOST_VALUES"] = "1"
    try:
        keep = ~((rows % 97 == 3) & (cols > rows + 20000)) & ~(rows % 1013 == 7)      # some rows lose their far clusters, some lose everything
        e3 = orc.coo_from_arrays(rows[keep], cols[keep], np.sin(0.37 * rows[keep] + 0.11 * cols[keep]) + (rows[keep] == cols[keep]) * 150.0)
        A3 = orc.OracleOp(e3, M, M, orc.split_even(M, 1))
        G3 = util.gpu_operator(A3)
        G3.set_variant(12)
        assert G3.variant()[1] == "k_sellx"
        b3 = abs_bound(e3, M, x)
        G3.spmv(dx, dy)
        assert np.all(np.abs(dy.download() - A3.matvec(x)) <= TOL_SPMV * b3 + 1e-300)
        G3.residual(dx, dr, dy)
        assert rel(dy.download(), A3.residual(x, rhs)) <= TOL_SMOOTH
        has_diag = np.ones(M, bool); has_diag[np.arange(M) % 1013 == 7] = False
        if has_diag.all():
            du = capi.DeviceVector(M, x)
            G3.jacobi(3, du, dr)
            assert rel(du.download(), A3.jacobi(3, x, rhs)) <= TOL_SMOOTH
        du = capi.DeviceVector(M, rhs)
        G3.prolong_correct(dx, du)
        assert rel(du.download(), rhs - A3.matvec(x)) <= TOL_SMOOTH
        # ... and on the full operator the smoothers
        G.set_variant(12)
        du = capi.DeviceVector(M, x)
        G.jacobi(3, du, dr)
        assert rel(du.download(), A.jacobi(3, x, rhs)) <= TOL_SMOOTH
        A.set_eig(1.9371)
        du = capi.DeviceVector(M, x)
        G.chebyshev(3, 1.9371, du, dr)
        assert rel(du.download(), A.chebyshev(3, x, rhs)) <= TOL_SMOOTH
        xn = x.copy(); xn[41000] = np.nan                  # a NaN in x reaches exactly the rows that own column 41000
        G.spmv(capi.DeviceVector(M, xn), dy)
        got = dy.download()
        owners = np.zeros(M, bool); owners[rows[cols == 41000]] = True
        assert np.array_equal(np.isnan(got), owners)
    finally:
        os.environ.pop("SAENA_KEEP_HOST_VALUES", None)
    rng = np.random.default_rng(5)
    rr = np.repeat(np.arange(2048), 24)
    key = np.unique(rr.astype(np.int64) * (1 << 20) + rng.integers(0, 1 << 20, size=rr.size))
    e2 = orc.coo_from_arrays((key >> 20).astype(np.int32), (key & ((1 << 20) - 1)).astype(np.int32), np.ones(key.size))
    A2 = orc.OracleOp(e2, 2048, 1 << 20, orc.split_even(2048, 1), orc.split_even(1 << 20, 1), square=False)
    with pytest.raises(capi.SgpuError, match="x-in-LDS"):
        util.gpu_operator(A2).set_variant(10)


@pytest.mark.parametrize("name", ["poisson12", "poisson20", "band300_7", "irregular5000"])
def test_residual_jacobi_chebyshev(capi, name):
    entries, M = get_problem(name)
    A = orc.OracleOp(entries, M, M, orc.split_even(M, 1))
    G = util.gpu_operator(A)
    G.set_lanes_per_row(1)
    u0, rhs = inputs.v2(M), inputs.rhs2(M)
    du, dr, dres = capi.DeviceVector(M, u0), capi.DeviceVector(M, rhs), capi.DeviceVector(M)
    G.residual(du, dr, dres)
    np.testing.assert_array_equal(dres.download(), A.residual(u0, rhs))
    # residual_negative (rhs - A u) and residual_multiply (c w o (rhs - A u)): saena_matrix.tpp:26-43, the same bits
    G.residual_negative(du, dr, dres)
    got = dres.download()
    np.testing.assert_array_equal(got, A.residual_negative(u0, rhs))
    np.testing.assert_array_equal(np.signbit(got), np.signbit(A.residual_negative(u0, rhs)))
    w = inputs.v_sin(M) + 2.0
    dw = capi.DeviceVector(M, w)
    G.residual_multiply(du, dr, dres, dw, 0.37)
    np.testing.assert_array_equal(dres.download(), A.residual_multiply(u0, rhs, w, 0.37))
    np.testing.assert_array_equal(du.download(), u0)                             # u untouched
    with pytest.raises(capi.SgpuError, match="alias"):
        G.residual_multiply(du, dr, du, dw, 0.37)
    for it in (1, 2, 3, 4):
        du.upload(u0)
        G.jacobi(it, du, dr)
        np.testing.assert_array_equal(du.download(), A.jacobi(it, u0, rhs))      # bit-exact at 1 lane/row
    A.set_eig(1.9371)
    for it in (1, 2, 3, 4):
        du.upload(u0)
        G.chebyshev(it, 1.9371, du, dr)
        np.testing.assert_array_equal(du.download(), A.chebyshev(it, u0, rhs))
    # auto lanes: rounding-level agreement
    G.set_lanes_per_row(0)
    du.upload(u0)
    G.jacobi(3, du, dr)
    assert rel(du.download(), A.jacobi(3, u0, rhs)) <= TOL_SMOOTH


@pytest.mark.parametrize("fn", sorted(glob.glob(os.path.join(GOLDEN, "ref_*.np1.npz"))), ids=os.path.basename)
def test_against_compiled_reference_golden(capi, fn):
    """Same checks as tests/test_oracle_pins.py, with the GPU in the oracle's place."""
    tag = os.path.basename(fn)[4:].split(".")[0]
    ref = dict(np.load(fn))
    if tag in matrices.FILES:
        entries, M = matrices.entries(tag)
    elif tag.startswith("poisson"):
        entries, M = orc.laplacian3d(int(tag[7:]))
    else:
        m_, bw = tag[4:].split("_")
        entries, M = orc.band_matrix(int(m_), int(bw)), int(m_)
    A = orc.OracleOp(entries, M, M, orc.split_even(M, 1))
    G = util.gpu_operator(A)
    v, v2, rhs2 = inputs.v_sin(M), inputs.v2(M), inputs.rhs2(M)
    dx, dy, dr = capi.DeviceVector(M, v), capi.DeviceVector(M), capi.DeviceVector(M, rhs2)
    G.spmv(dx, dy)
    assert np.all(np.abs(dy.download() - ref["Av"]) <= TOL_SPMV * abs_bound(entries, M, v))
    dx.upload(v2)
    G.spmv(dx, dy)
    assert np.all(np.abs(dy.download() - ref["Av2"]) <= TOL_SPMV * abs_bound(entries, M, v2))
    G.residual(dx, dr, dy)
    assert rel(dy.download(), ref["residual_v2_rhs2"]) <= TOL_SMOOTH
    du = capi.DeviceVector(M, np.zeros(M))
    ones = capi.DeviceVector(M, np.ones(M))
    G.jacobi(3, du, ones)
    assert rel(du.download(), ref["jacobi3_rhs1"]) <= TOL_SMOOTH
    du.upload(v2)
    G.jacobi(2, du, dr)
    assert rel(du.download(), ref["jacobi2_v2_rhs2"]) <= TOL_SMOOTH
    du.fill(0.0)
    G.chebyshev(3, 2.0, du, ones)
    assert rel(du.download(), ref["cheby3_rhs1"]) <= TOL_SMOOTH
    du.upload(v2)
    G.chebyshev(4, 1.9371, du, dr)
    assert rel(du.download(), ref["cheby4_v2_rhs2"]) <= TOL_SMOOTH
    if "P_ec" not in ref:
        return
    # grid transfers R v, P e and the fused u -= P e
    pr, pc, pv, Nc = inputs.synthetic_P(M)
    P = orc.OracleOp(orc.coo_from_arrays(pr, pc, pv), M, Nc, orc.split_even(M, 1), orc.split_even(Nc, 1), square=False)
    R = orc.OracleOp(orc.coo_from_arrays(pc, pr, pv), Nc, M, orc.split_even(Nc, 1), orc.split_even(M, 1), square=False)
    GP, GR = util.gpu_operator(P), util.gpu_operator(R)
    ec = inputs.ec(Nc)
    dec, dpf, drc = capi.DeviceVector(Nc, ec), capi.DeviceVector(M), capi.DeviceVector(Nc)
    GP.spmv(dec, dpf)
    assert rel(dpf.download(), ref["P_ec"]) <= 1e-14
    GR.spmv(dx, drc)
    assert rel(drc.download(), ref["R_v2"]) <= 1e-14
    du.upload(v2)
    GP.prolong_correct(dec, du)
    assert rel(du.download(), v2 - ref["P_ec"]) <= 1e-14


def test_suitesparse_through_product_reader(capi, tmp_path):
    """config 5: SuiteSparse matrices with irregular rows (reference data/): product reader -> assemble ->
    GPU SpMV / Jacobi, against the compiled-reference fixtures and the oracle; autotuned kernel choice."""
    from saena_amd import host
    for name in ("plat362", "SiH4", "fxm3_6"):
        ref = dict(np.load(os.path.join(GOLDEN, f"ref_{name}.np1.npz")))
        A = host.Matrix(host.Comm("gpu", "rccl")).read_file(matrices.path(name, tmp_path)).assemble()
        entries, M = matrices.entries(name)
        assert (A.num_rows, A.nnz) == (M, len(entries)) == (ref["meta"][0], ref["meta"][1])
        op = host.device_operator(A)
        op.autotune()
        v, v2, rhs2 = inputs.v_sin(M), inputs.v2(M), inputs.rhs2(M)
        dx, dy, dr = capi.DeviceVector(M, v), capi.DeviceVector(M), capi.DeviceVector(M, rhs2)
        op.spmv(dx, dy)
        assert np.all(np.abs(dy.download() - ref["Av"]) <= TOL_SPMV * abs_bound(entries, M, v))
        du = capi.DeviceVector(M, v2)
        op.jacobi(2, du, dr)
        assert rel(du.download(), ref["jacobi2_v2_rhs2"]) <= TOL_SMOOTH
        O = orc.OracleOp(entries, M, M, orc.split_even(M, 1))
        op.set_variant(0); op.set_lanes_per_row(1)
        dx.upload(v2)
        op.spmv(dx, dy)
        np.testing.assert_array_equal(dy.download(), O.matvec(v2))


@pytest.mark.parametrize("name,nprocs", [("poisson12", 2), ("poisson12", 4), ("poisson20", 3), ("band300_7", 4),
                                         ("band64_63", 2), ("irregular5000", 5)])
@pytest.mark.parametrize("fp32", [False, True])
def test_halo_path_emulated_ranks(capi, name, nprocs, fp32):
    """pack kernel + remote-CSR kernel + remote epilogues, P simulated ranks on one GPU."""
    entries, M = get_problem(name)
    split = orc.split_nnz(entries, M, nprocs)
    A = orc.OracleOp(entries, M, M, split)
    A.set_use_double(not fp32)
    W = util.EmulatedWorld(A, halo_fp32=fp32)
    x, rhs = inputs.v2(M), inputs.rhs2(M)
    bound = abs_bound(entries, M, x)
    xs, ys, rs = W.slices(x, split), W.slices(np.zeros(M), split), W.slices(rhs, split)
    W.exchange(xs)
    for r in range(nprocs):
        W.g[r].spmv(xs[r], ys[r])
    assert np.all(np.abs(W.gather(ys) - A.matvec(x)) <= TOL_SPMV * bound + 1e-300)
    for r in range(nprocs):
        W.g[r].residual(xs[r], rs[r], ys[r])
    assert rel(W.gather(ys), A.residual(x, rhs)) <= TOL_SMOOTH
    # smoother sweeps, re-exchanging the halo before each sweep like the reference's matvec does
    us = W.slices(x, split)
    for sweep in range(2):
        W.exchange(us)
        for r in range(nprocs):
            W.g[r].jacobi(1, us[r], rs[r])
    assert rel(W.gather(us), A.jacobi(2, x, rhs)) <= TOL_SMOOTH
    if not fp32:
        A.set_eig(1.9371)
        us = W.slices(x, split)
        W.exchange(us)
        for r in range(nprocs):
            W.g[r].chebyshev(1, 1.9371, us[r], rs[r])
        assert rel(W.gather(us), A.chebyshev(1, x, rhs)) <= TOL_SMOOTH


@pytest.mark.parametrize("fixture", ["ref_poisson16.np8.npz", "ref_poisson12.np4.npz", "ref_SiH4.np4.npz"])
def test_emulated_ranks_against_compiled_reference_golden(capi, fixture):
    """The compiled reference's own outputs at 8 (and 4) MPI ranks -- `mpirun -np 8 oracle/_ref/ref_dump`: north_star's rank count, which
    one card cannot host as 8 processes -- against the device path with that many EMULATED ranks at the reference's own partition:
    pack kernel, halo, interior + boundary kernels per rank; matvec (fp64 and fp32 halo), residual, Jacobi and Chebyshev sweeps."""
    ref = dict(np.load(os.path.join(GOLDEN, fixture)))
    tag = fixture.split(".")[0][4:]
    entries, M = orc.laplacian3d(int(tag[7:])) if tag.startswith("poisson") else matrices.entries(tag)
    split = ref["split"]
    nprocs = len(split) - 1
    assert nprocs == int(fixture.split(".np")[1].split(".")[0])
    v, v2, rhs2 = inputs.v_sin(M), inputs.v2(M), inputs.rhs2(M)
    for fp32 in (False, True):
        A = orc.OracleOp(entries, M, M, split)
        A.set_use_double(not fp32)
        W = util.EmulatedWorld(A, halo_fp32=fp32)
        xs, ys = W.slices(v2, split), W.slices(np.zeros(M), split)
        W.exchange(xs)
        for r in range(nprocs):
            W.g[r].spmv(xs[r], ys[r])
        assert np.all(np.abs(W.gather(ys) - ref["Av2_float" if fp32 else "Av2"]) <= TOL_SPMV * abs_bound(entries, M, v2) + 1e-300)
        if fp32:
            continue
        xs = W.slices(v, split)
        W.exchange(xs)
        for r in range(nprocs):
            W.g[r].spmv(xs[r], ys[r])
        assert np.all(np.abs(W.gather(ys) - ref["Av"]) <= TOL_SPMV * abs_bound(entries, M, v) + 1e-300)
        xs, rs = W.slices(v2, split), W.slices(rhs2, split)
        W.exchange(xs)
        for r in range(nprocs):
            W.g[r].residual(xs[r], rs[r], ys[r])
        assert rel(W.gather(ys), ref["residual_v2_rhs2"]) <= TOL_SMOOTH
        us = W.slices(v2, split)
        for sweep in range(2):
            W.exchange(us)
            for r in range(nprocs):
                W.g[r].jacobi(1, us[r], rs[r])
        assert rel(W.gather(us), ref["jacobi2_v2_rhs2"]) <= TOL_SMOOTH
        us = W.slices(v2, split)
        W.exchange(us)
        for r in range(nprocs):
            W.g[r].chebyshev(1, 1.9371, us[r], rs[r])
        assert rel(W.gather(us), ref["cheby1_v2_rhs2"]) <= TOL_SMOOTH


@pytest.mark.parametrize("name,nprocs", [("poisson20", 3), ("band64_63", 2), ("irregular5000", 5)])
def test_halo_path_column_ordered_kernel(capi, name, nprocs, monkeypatch):
    """k_csr_cm (and k_sell) with a halo in play (boundary rows masked out of the local launch, computed by k_csr_boundary): the
    same bits as k_csr_cc16 on every simulated rank, for the product and for a Jacobi sweep."""
    monkeypatch.setenv("SAENA_KEEP_HOST_VALUES", "1")
    entries, M = get_problem(name)
    split = orc.split_nnz(entries, M, nprocs)
    A = orc.OracleOp(entries, M, M, split)
    W = util.EmulatedWorld(A)
    x, rhs = inputs.v2(M), inputs.rhs2(M)
    out = {}
    for variant in (4, 8) + ((9, 11, 14) if name == "poisson20" else ()):
        xs, ys, rs, us = W.slices(x, split), W.slices(np.zeros(M), split), W.slices(rhs, split), W.slices(x, split)
        W.exchange(xs); W.exchange(us)
        for r in range(nprocs):
            W.g[r].set_variant(variant); W.g[r].set_lanes_per_row(1)
            W.g[r].spmv(xs[r], ys[r])
            W.g[r].jacobi(1, us[r], rs[r])
        out[variant] = (W.gather(ys), W.gather(us))
    np.testing.assert_array_equal(out[8][0], out[4][0])
    np.testing.assert_array_equal(out[8][1], out[4][1])
    for r in range(nprocs):                                   # x in LDS (a wave per row: its own summation order)
        W.g[r].set_variant(10)
    xs, ys, rs, us = W.slices(x, split), W.slices(np.zeros(M), split), W.slices(rhs, split), W.slices(x, split)
    W.exchange(xs); W.exchange(us)
    for r in range(nprocs):
        W.g[r].spmv(xs[r], ys[r])
        W.g[r].jacobi(1, us[r], rs[r])
    assert np.all(np.abs(W.gather(ys) - A.matvec(x)) <= TOL_SPMV * abs_bound(entries, M, x) + 1e-300)
    assert rel(W.gather(us), A.jacobi(1, x, rhs)) <= TOL_SMOOTH
    for v in (9, 11, 14):                                     # sliced ELLPACK, with column codes, with row patterns, with a lane per two rows: the sequential row sum as well
        if v in out:
            np.testing.assert_array_equal(out[v][0], out[4][0])
            np.testing.assert_array_equal(out[v][1], out[4][1])
    assert np.all(np.abs(out[8][0] - A.matvec(x)) <= TOL_SPMV * abs_bound(entries, M, x) + 1e-300)
    assert rel(out[8][1], A.jacobi(1, x, rhs)) <= TOL_SMOOTH


def test_rectangular_transfer_emulated_ranks(capi):
    """R (coarse rows, fine halo) and P (fine rows, coarse halo) with remote parts."""
    M = 1000
    pr, pc, pv, Nc = inputs.synthetic_P(M)
    for nprocs in (2, 4):
        split = orc.split_even(M, nprocs)
        splitNew = (split // 2).astype(np.int32); splitNew[-1] = Nc
        P = orc.OracleOp(orc.coo_from_arrays(pr, pc, pv), M, Nc, split, splitNew, square=False)
        R = orc.OracleOp(orc.coo_from_arrays(pc, pr, pv), Nc, M, splitNew, split, square=False)
        ec, res, u = inputs.ec(Nc), inputs.v2(M), inputs.rhs2(M)
        WP, WR = util.EmulatedWorld(P), util.EmulatedWorld(R)
        es, fs = WP.slices(ec, splitNew), WP.slices(np.zeros(M), split)
        WP.exchange(es)
        for r in range(nprocs):
            WP.g[r].spmv(es[r], fs[r])
        assert rel(WP.gather(fs), P.matvec(ec)) <= 1e-14
        us = WP.slices(u, split)
        for r in range(nprocs):
            WP.g[r].prolong_correct(es[r], us[r])
        assert rel(WP.gather(us), u - P.matvec(ec)) <= 1e-14
        rs_, cs = WR.slices(res, split), WR.slices(np.zeros(Nc), splitNew)
        WR.exchange(rs_)
        for r in range(nprocs):
            WR.g[r].spmv(rs_[r], cs[r])
        assert rel(WR.gather(cs), R.matvec(res)) <= 1e-14


def test_vector_kernels(capi):
    n = 100003
    x, y = inputs.v2(n), inputs.rhs2(n)
    dx, dy = capi.DeviceVector(n, x), capi.DeviceVector(n, y)
    d = capi.dot(dx, dy)
    assert abs(d - np.dot(x, y)) <= 1e-12 * np.sum(np.abs(x * y))
    capi.check(capi.lib().sgpu_vec_axpby(2.5, dx.ptr, -0.5, dy.ptr, n))
    np.testing.assert_array_equal(dy.download(), 2.5 * x + -0.5 * y)
    dy.fill(3.25)
    assert np.all(dy.download() == 3.25)
    # determinism: the dot is reduced in a fixed order
    assert capi.dot(dx, dx) == capi.dot(dx, dx)


def test_error_codes_not_exit(capi):
    """The reference prints and exit()s on bad input; the C ABI returns codes."""
    with pytest.raises(capi.SgpuError):
        capi.Operator(M=3, N_local=3, col_offset=0, nnzPerRow_local=[1, 1, 1], col_local=[0, 1, 7], val_local=[1, 1, 1.0])
    with pytest.raises(capi.SgpuError):
        capi.Operator(M=3, N_local=3, col_offset=0, nnzPerRow_local=[1, 1, 2], col_local=[0, 1, 2], val_local=[1, 1, 1.0])
    A = capi.Operator(M=2, N_local=2, col_offset=0, nnzPerRow_local=[1, 1], col_local=[0, 1], val_local=[2.0, 4.0])
    v = capi.DeviceVector(2, [1.0, 1.0])
    with pytest.raises(capi.SgpuError):
        A.jacobi(1, v, v)               # no inv_diag
    # empty operator / empty rows
    E = capi.Operator(M=0, N_local=0, col_offset=0, nnzPerRow_local=[], col_local=[], val_local=[])
    e = capi.DeviceVector(0)
    E.spmv(e, e)
    Z = capi.Operator(M=3, N_local=3, col_offset=0, nnzPerRow_local=[0, 2, 0], col_local=[0, 2], val_local=[1.5, -2.0])
    dx, dy = capi.DeviceVector(3, [1.0, 2.0, 3.0]), capi.DeviceVector(3, [9.0, 9.0, 9.0])
    Z.spmv(dx, dy)
    np.testing.assert_array_equal(dy.download(), [0.0, -4.5, 0.0])


@pytest.mark.parametrize("name", ["band64_63", "band3000_1400", "band300_7", "poisson12", "poisson24"])
def test_dense_variant(capi, name):
    """variant 5, k_dense_rows (the reference's switch_to_dense storage, saena_matrix_dense): every fused epilogue
    against the oracle on full / half-full / sparse operators; refused -- not mis-computed -- beyond 8192 rows"""
    entries, M = get_problem(name)
    A = orc.OracleOp(entries, M, M, orc.split_even(M, 1))
    G = util.gpu_operator(A)
    if M > 8192:
        with pytest.raises(capi.SgpuError, match="too large"):
            G.set_variant(5)
        return
    G.set_variant(5)
    assert G.variant() == (5, "k_dense_rows")
    x, rhs = inputs.v2(M), inputs.rhs2(M)
    dx, dy, dr, dres = capi.DeviceVector(M, x), capi.DeviceVector(M), capi.DeviceVector(M, rhs), capi.DeviceVector(M)
    G.spmv(dx, dy)
    assert np.all(np.abs(dy.download() - A.matvec(x)) <= TOL_SPMV * abs_bound(entries, M, x) + 1e-300)
    G.residual(dx, dr, dres)
    assert rel(dres.download(), A.residual(x, rhs)) <= 1e-13
    du = capi.DeviceVector(M, x)
    G.jacobi(3, du, dr)
    assert rel(du.download(), A.jacobi(3, x, rhs)) <= TOL_SMOOTH
    A.set_eig(1.9371)
    du.upload(x)
    G.chebyshev(3, 1.9371, du, dr)
    assert rel(du.download(), A.chebyshev(3, x, rhs)) <= TOL_SMOOTH
    G.set_variant(0)                                   # and back
    G.spmv(dx, dy)
    assert np.all(np.abs(dy.download() - A.matvec(x)) <= TOL_SPMV * abs_bound(entries, M, x) + 1e-300)


@pytest.mark.parametrize("fixture", ["ref_band64_63.np2.npz", "ref_band300_7.np4.npz", "ref_poisson8.np4.npz", "ref_plat362.np2.npz", "ref_band64_63.np1.npz", "ref_poisson16.np8.npz"])
@pytest.mark.parametrize("fp32", [False, True], ids=["fp64", "float"])
def test_dense_operator_with_halo_against_compiled_reference(capi, fixture, fp32):
    """SURVEY 8 row f3 across ranks: the row-partitioned dense operator (saena_matrix_dense::matvec_dense /
    matvec_dense_float, src/saena_matrix_dense.cpp:181-340) at the reference's own 2- and 4-rank partitions.  The
    reference walks the x blocks round a ring; here every rank gets its halo in one exchange and multiplies dense rows
    (k_dense_rows_halo).  Golden vectors: the compiled reference's `Av2_dense(_float)`."""
    ref = np.load(os.path.join(GOLDEN, fixture))
    tag = fixture.split(".")[0][4:]
    if tag.startswith("band"):
        M, bw = (int(t) for t in tag[4:].split("_"))
        entries = orc.band_matrix(M, bw)
    elif tag.startswith("poisson"):
        entries, M = orc.laplacian3d(int(tag[7:]))
    else:
        entries, M = matrices.entries(tag)
    split = ref["split"]
    nprocs = len(split) - 1
    A = orc.OracleOp(entries, M, M, split)
    x = inputs.v2(M)
    bound = abs_bound(entries, M, x)
    want = ref["Av2_dense_float" if fp32 else "Av2_dense"]
    W = util.EmulatedWorld(A, halo_fp32=fp32)
    for g in W.g:
        g.set_variant(5)                                # dense rows + dense halo columns on every rank
        assert g.variant()[0] == 5
    xs, ys = W.slices(x, split), W.slices(np.zeros(M), split)
    W.exchange(xs)
    for r in range(nprocs):
        W.g[r].spmv(xs[r], ys[r])
    got = W.gather(ys)
    assert np.all(np.abs(got - want) <= TOL_SPMV * bound + 1e-300), np.max(np.abs(got - want) / (bound + 1e-300))
    # and the oracle's restatement of the ring GEMV agrees with both
    assert np.all(np.abs(A.matvec_dense(x, as_float=fp32) - want) <= TOL_SPMV * bound + 1e-300)
    if not fp32:                                         # a fused smoother through the same dense path
        rhs = inputs.rhs2(M)
        us, rs = W.slices(x, split), W.slices(rhs, split)
        for _ in range(2):
            W.exchange(us)
            for r in range(nprocs):
                W.g[r].jacobi(1, us[r], rs[r])
        assert rel(W.gather(us), A.jacobi(2, x, rhs)) <= TOL_SMOOTH


def _golden_worker(rank, world, port, fn, ret):
    import sys
    import tempfile
    import torch.distributed as dist                      # torch first, like bench.py --gpus N
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        from saena_amd import capi as c, host
        from tests import inputs as inp, matrices as mats
        c.init_host_transport(0, dist)
        ref = dict(np.load(fn))
        tag = os.path.basename(fn)[4:].split(".")[0]
        comm = host.Comm("gpu", "dist", dist)
        A = host.Matrix(comm)
        with tempfile.TemporaryDirectory() as tmp:
            if tag in mats.FILES:
                A.read_file(mats.path(tag, tmp))
            elif tag.startswith("poisson"):
                A.laplacian3D(int(tag[7:]))
            else:
                m_, bw = tag[4:].split("_")
                A.band_matrix(int(m_) // world, int(bw))   # the generator's M is the LOCAL size (aux_functions2.cpp:1296)
            A.assemble()                                   # the reference's partitioner, over `world` ranks
        np.testing.assert_array_equal(A.split, ref["split"])
        lo, hi = int(A.split[rank]), int(A.split[rank + 1])
        M, n = int(ref["meta"][0]), hi - lo
        v, v2, rhs2 = inp.v_sin(M)[lo:hi], inp.v2(M)[lo:hi], inp.rhs2(M)[lo:hi]

        def close(got, key, tol):
            want = ref[key][lo:hi]
            scale = np.linalg.norm(ref[key]) / np.sqrt(M) * np.sqrt(max(n, 1))
            assert np.linalg.norm(got - want) <= tol * max(scale, 1e-300), (key, np.linalg.norm(got - want), scale)
        G = host.device_operator(A)
        dx, dy, dr = c.DeviceVector(n, v), c.DeviceVector(n), c.DeviceVector(n, rhs2)
        G.spmv(dx, dy); close(dy.download(), "Av", 1e-13)
        dx.upload(v2)
        G.spmv(dx, dy); close(dy.download(), "Av2", 1e-13)
        G.residual(dx, dr, dy); close(dy.download(), "residual_v2_rhs2", 1e-12)
        du, ones = c.DeviceVector(n, np.zeros(n)), c.DeviceVector(n, np.ones(n))
        G.jacobi(3, du, ones); close(du.download(), "jacobi3_rhs1", 1e-12)
        du.upload(v2)
        G.jacobi(2, du, dr); close(du.download(), "jacobi2_v2_rhs2", 1e-12)
        du.upload(np.zeros(n))
        G.chebyshev(3, 2.0, du, ones); close(du.download(), "cheby3_rhs1", 1e-12)
        du.upload(v2)
        G.chebyshev(4, 1.9371, du, dr); close(du.download(), "cheby4_v2_rhs2", 1e-12)
        Gf = host.device_operator(A, halo_fp32=True)       # matvec_sparse_float: fp32 halo on the wire
        Gf.spmv(dx, dy); close(dy.download(), "Av2_float", 1e-13)
        if "P_ec" in ref:                                  # grid transfers on two partitions: rows by split, columns by splitNew
            split = A.split
            pr, pc, pv, Nc = inp.synthetic_P(M)
            splitNew = ref["splitNew"].astype(np.int32)
            mine = (pr >= lo) & (pr < hi)
            P = host.Transfer.prolong(comm, M, Nc, split, splitNew, pr[mine], pc[mine], pv[mine])
            R = P.transpose()
            GP, GR = host.device_operator(P), host.device_operator(R)
            clo, chi = int(splitNew[rank]), int(splitNew[rank + 1])
            nc = chi - clo
            dec, dpf, drc = c.DeviceVector(nc, inp.ec(Nc)[clo:chi]), c.DeviceVector(n), c.DeviceVector(nc)
            GP.spmv(dec, dpf)
            want = ref["P_ec"][lo:hi]
            assert np.linalg.norm(dpf.download() - want) <= 1e-13 * max(np.linalg.norm(ref["P_ec"]), 1e-300)
            dx.upload(v2)
            GR.spmv(dx, drc)
            want = ref["R_v2"][clo:chi]
            assert np.linalg.norm(drc.download() - want) <= 1e-13 * max(np.linalg.norm(ref["R_v2"]), 1e-300)
            du.upload(rhs2)                                # u -= P e (the V-cycle's correction, fused)
            GP.prolong_correct(dec, du)
            want = rhs2 - ref["P_ec"][lo:hi]
            assert np.linalg.norm(du.download() - want) <= 1e-13 * max(np.linalg.norm(want), 1e-300)
        ret[rank] = "ok"
    except BaseException as e:      # noqa
        import traceback
        ret[rank] = "".join(traceback.format_exception(type(e), e, e.__traceback__))[-2500:]
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("fixture", ["ref_SiH4.np4.npz", "ref_fxm3_6.np3.npz", "ref_poisson12.np4.npz", "ref_band300_7.np4.npz", "ref_plat362.np2.npz"])
def test_multirank_library_against_compiled_reference_golden(fixture):
    """The compiled reference's multi-rank outputs (mpirun -np 2/3/4 of oracle/_ref/ref_dump, tests/golden/) against the
    LIBRARY's multi-rank path: the product's host assemble with the reference's partitioner over that many processes,
    sgpu_op_create from this rank's layout, interior + boundary kernels, halos routed through gloo
    (sgpu_debug_init_host_transport) -- matvec, fp32-halo matvec, residual, Jacobi and Chebyshev sweeps."""
    import multiprocessing as mp      # not torch's: this process runs the system HIP runtime
    import socket
    fn = os.path.join(GOLDEN, fixture)
    world = int(fixture.split(".np")[1].split(".")[0])
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    ctx = mp.get_context("spawn")
    with ctx.Manager() as mgr:
        ret = mgr.dict()
        procs = [ctx.Process(target=_golden_worker, args=(r, world, port, fn, ret)) for r in range(world)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(300)
        for p in procs:
            if p.is_alive():
                p.terminate()
        res = dict(ret)
    for r in range(world):
        assert res.get(r) == "ok", f"rank {r}: {res.get(r)}"


@pytest.mark.parametrize("name", ["poisson12", "poisson20", "band300_7", "irregular5000"])
def test_row_templates(capi, name, monkeypatch):
    """k_rowt (opt-in): rows that repeat (length, relative columns, values) -- the boundary-stripped Laplacian has 27 distinct rows --
    are served from a table in LDS, the kernel reads a 16-bit template id per row and nothing else of the operator: bit-identical
    to the CSR loop for every epilogue; an operator whose rows do not repeat is refused."""
    monkeypatch.setenv("SAENA_KEEP_HOST_VALUES", "1")
    entries, M = get_problem(name)
    A = orc.OracleOp(entries, M, M, orc.split_even(M, 1))
    G = util.gpu_operator(A)
    if not name.startswith("poisson"):
        with pytest.raises(capi.SgpuError, match="row-template"):
            G.set_variant(13)
        return
    G.set_variant(13)
    assert G.variant()[1] == "k_rowt"
    x, rhs = inputs.v2(M), inputs.rhs2(M)
    dx, dy, dr = capi.DeviceVector(M, x), capi.DeviceVector(M), capi.DeviceVector(M, rhs)
    G.spmv(dx, dy)
    np.testing.assert_array_equal(dy.download(), A.matvec(x))
    G.residual(dx, dr, dy)
    np.testing.assert_array_equal(dy.download(), A.residual(x, rhs))
    du = capi.DeviceVector(M, x)
    G.jacobi(3, du, dr)
    np.testing.assert_array_equal(du.download(), A.jacobi(3, x, rhs))
    A.set_eig(1.9371)
    du = capi.DeviceVector(M, x)
    G.chebyshev(3, 1.9371, du, dr)
    assert rel(du.download(), A.chebyshev(3, x, rhs)) <= TOL_SMOOTH
    du = capi.DeviceVector(M, rhs)
    G.prolong_correct(dx, du)
    np.testing.assert_array_equal(du.download(), rhs - A.matvec(x))


def _patterned_operator(M, npat, length, reach, seed, run=1):
    """square operator whose row r follows pattern (r // run) % npat: `length` (+ 0..2) distinct offsets within +-reach of the row,
    the diagonal among them; near the two ends the offsets that leave the matrix are dropped (more patterns there)"""
    rng = np.random.default_rng(seed)
    pats = []
    for k in range(npat):
        o = rng.choice(np.setdiff1d(np.arange(-reach, reach + 1), [0]), size=length - 1 + k % 3, replace=False)
        pats.append(np.sort(np.concatenate([o, [0]])))
    rows, cols = [], []
    for r in range(M):
        c = r + pats[(r // run) % npat]
        c = c[(c >= 0) & (c < M)]
        rows.append(np.full(len(c), r)); cols.append(c)
    rows, cols = np.concatenate(rows), np.concatenate(cols)
    vals = np.cos(0.37 * rows - 0.11 * cols) + 1.25
    return orc.coo_from_arrays(rows.astype(np.int32), cols.astype(np.int32), vals)


@pytest.mark.parametrize("variant", [11, 14, 15])
def test_row_patterns_wide_table(capi, variant, monkeypatch):
    """k_sellp<wide> / k_sellp2<wide> / k_sellpx (the same patterns with the windows of x they reach in LDS): an operator whose rows follow a few HUNDRED patterns of several dozen entries (the first smoothed-
    aggregation level of a structured grid: 321 patterns, 14 469 offsets at every size of the Poisson cube) keeps the table
    the operator has, and every workgroup of 1024 rows gets a table of the few dozen it meets.  No column stream, the reference's sequential row sum: bit-identical to the CSR loop for every epilogue; with too
    many patterns the form is refused.  The row-paired kernel meets every case of a lane's two rows here: same pattern (16-byte
    loads of x), different patterns (the pattern changes every row: a 16-byte load and an 8-byte gather), the last column of x in
    the first row of a pair, the ragged last slice."""
    monkeypatch.setenv("SAENA_KEEP_HOST_VALUES", "1")
    M = 40000 + 37                                            # 626 slices (the last one ragged), 40 groups of 16
    entries = _patterned_operator(M, 240, 40, 50, 11, run=7)         # runs of 7 rows: row pairs with one pattern and with two; a group of 1024 rows meets 147 patterns
    assert _sellp_table(entries, M) == "wide"
    A = orc.OracleOp(entries, M, M, orc.split_even(M, 1))
    G = util.gpu_operator(A)
    G.set_variant(variant)
    assert G.variant()[1] == {11: "k_sellp<wide>", 14: "k_sellp2<wide>", 15: "k_sellpx"}[variant]
    x, rhs = inputs.v2(M), inputs.rhs2(M)
    dx, dy, dr = capi.DeviceVector(M, x), capi.DeviceVector(M), capi.DeviceVector(M, rhs)
    G.spmv(dx, dy)
    np.testing.assert_array_equal(dy.download(), A.matvec(x))
    G.residual(dx, dr, dy)
    np.testing.assert_array_equal(dy.download(), A.residual(x, rhs))
    du = capi.DeviceVector(M, x)
    G.jacobi(3, du, dr)
    np.testing.assert_array_equal(du.download(), A.jacobi(3, x, rhs))
    A.set_eig(1.9371)
    du = capi.DeviceVector(M, x)
    G.chebyshev(3, 1.9371, du, dr)
    assert rel(du.download(), A.chebyshev(3, x, rhs)) <= TOL_SMOOTH
    du = capi.DeviceVector(M, rhs)
    G.prolong_correct(dx, du)
    np.testing.assert_array_equal(du.download(), rhs - A.matvec(x))
    many = _patterned_operator(30000, 600, 40, 50, 12)    # every group of 1024 rows meets all 600 patterns x 41 ints: beyond a workgroup's table
    assert _sellp_table(many, 30000) is None
    G2 = util.gpu_operator(orc.OracleOp(many, 30000, 30000, orc.split_even(30000, 1)))
    with pytest.raises(capi.SgpuError, match="row-pattern"):
        G2.set_variant(variant)


def test_row_patterns_with_x_in_lds_windows(capi, monkeypatch):
    """k_sellpx: the offsets of the patterns fall into five clusters 1000-2000 columns apart (the shape of a smoothed-aggregation
    level on a structured grid: a few grid lines in a few planes), so a workgroup of 512 rows reads x inside five windows, which it
    stages in LDS; windows that leave the vector at its two ends are zero-filled and never read.  Bit-identical to the CSR loop
    for every epilogue; offsets spread over more than 16 windows are refused."""
    monkeypatch.setenv("SAENA_KEEP_HOST_VALUES", "1")
    rng = np.random.default_rng(21)
    M = 23000 + 19
    pool = np.concatenate([c + np.arange(-15, 16) for c in (-3000, -1000, 0, 1000, 3000)])
    pats = [np.sort(np.unique(np.concatenate([rng.choice(pool, size=29 + k % 2, replace=False), [0]]))) for k in range(60)]
    rows, cols = [], []
    for r in range(M):
        c = r + pats[(r // 7) % 60]
        c = c[(c >= 0) & (c < M)]
        rows.append(np.full(len(c), r)); cols.append(c)
    rows, cols = np.concatenate(rows), np.concatenate(cols)
    entries = orc.coo_from_arrays(rows.astype(np.int32), cols.astype(np.int32), np.cos(0.37 * rows - 0.11 * cols) + 1.25)
    assert _sellp_table(entries, M) == "wide"
    A = orc.OracleOp(entries, M, M, orc.split_even(M, 1))
    G = util.gpu_operator(A)
    G.set_variant(15)
    assert G.variant()[1] == "k_sellpx"
    x, rhs = inputs.v2(M), inputs.rhs2(M)
    dx, dy, dr = capi.DeviceVector(M, x), capi.DeviceVector(M), capi.DeviceVector(M, rhs)
    G.spmv(dx, dy)
    np.testing.assert_array_equal(dy.download(), A.matvec(x))
    G.residual(dx, dr, dy)
    np.testing.assert_array_equal(dy.download(), A.residual(x, rhs))
    du = capi.DeviceVector(M, x)
    G.jacobi(3, du, dr)
    np.testing.assert_array_equal(du.download(), A.jacobi(3, x, rhs))
    du = capi.DeviceVector(M, rhs)
    G.prolong_correct(dx, du)
    np.testing.assert_array_equal(du.download(), rhs - A.matvec(x))
    # 20 clusters: more windows than the kernel takes
    far = np.concatenate([[0], 700 * np.arange(1, 11), -700 * np.arange(1, 11)])
    M2 = 16000
    rows = np.repeat(np.arange(M2), len(far)); cols = rows + np.tile(np.sort(far), M2)
    ok = (cols >= 0) & (cols < M2)
    e2 = orc.coo_from_arrays(rows[ok].astype(np.int32), cols[ok].astype(np.int32), np.ones(int(ok.sum())))
    G2 = util.gpu_operator(orc.OracleOp(e2, M2, M2, orc.split_even(M2, 1)))
    if _sellp_table(e2, M2) is not None:
        with pytest.raises(capi.SgpuError, match="x in LDS"):
            G2.set_variant(15)


def test_plan_cache_makes_a_second_operator_take_the_first_one_s_plan(capi, tmp_path, monkeypatch):
    """The plan-time autotune writes its choice to the plan cache; an operator of the same shape created afterwards (here:
    in the same process, in production: by a later process) takes the plan from there without a sweep -- same kernel,
    same summation order, bit-identical results.  SAENA_PLAN_CACHE=off tunes afresh."""
    cache = tmp_path / "plans.tsv"
    monkeypatch.setenv("SAENA_PLAN_CACHE", str(cache))
    entries, M = orc.laplacian3d(66)                        # 262 144 rows, 1.8 M entries: above the autotune's size floor
    A = orc.OracleOp(entries, M, M, orc.split_even(M, 1))
    x = inputs.v2(M)
    G1 = util.gpu_operator(A)
    G1.autotune()
    lines = cache.read_text().strip().splitlines()
    assert len(lines) == 1 and len(lines[0].split("\t")) >= 4
    G2 = util.gpu_operator(A)
    t0 = time.perf_counter()
    G2.autotune()
    dt = time.perf_counter() - t0
    assert len(cache.read_text().strip().splitlines()) == 1, "the second operator must not have been tuned again"
    assert G2.variant() == G1.variant() and G2.info()["lanes_per_row"] == G1.info()["lanes_per_row"]
    assert dt < 1.0
    dx, y1, y2 = capi.DeviceVector(M, x), capi.DeviceVector(M), capi.DeviceVector(M)
    G1.spmv(dx, y1); G2.spmv(dx, y2)
    np.testing.assert_array_equal(y1.download(), y2.download())
    # the 7-point level qualifies for the form without a column stream, and the fixed ranking prefers forms with the
    # sequential row sum: the result is the oracle's bit for bit
    assert G1.variant()[1] in ("k_sellp", "k_sellp2", "k_sell")
    np.testing.assert_array_equal(y1.download(), A.matvec(x))


def test_config5_irregular_operator_against_the_oracle(capi, monkeypatch):
    """BASELINE configs[4] as bench.py measures it (`spmv_irregular`: SiH4 replicated with per-block permutations, coupling entries
    and hub rows, tests/irregular.py), at 3 blocks -- a size the oracle runs in a second: rows of 13 to 3 000 entries, a hub row
    longer than the 16 KiB tile (the long-row path).  One lane per row: the reference's sequential sum, bit for bit; the autotuned
    kernel and its Jacobi sweeps within the stated tolerances."""
    from tests import irregular
    monkeypatch.setenv("SAENA_KEEP_HOST_VALUES", "1")       # variants are switched after the autotune below
    r, c, v, M = irregular.sih4_replicated(3)
    A = orc.OracleOp(orc.coo_from_arrays(r, c, v), M, M, orc.split_even(M, 1))
    G = util.gpu_operator(A)
    plan = G.block_plan(0)
    assert plan["nnz"] == len(r) and plan["long_rows"] >= 1 and plan["longest_row"] >= 2900, plan
    G.set_lanes_per_row(1)
    x, rhs = inputs.v2(M), inputs.rhs2(M)
    dx, dy, dr = capi.DeviceVector(M, x), capi.DeviceVector(M), capi.DeviceVector(M, rhs)
    G.spmv(dx, dy)
    want = A.matvec(x)
    rows = np.asarray(r)
    bound = np.bincount(rows, weights=np.abs(v * x[c]), minlength=M)
    got = dy.download()
    short = np.bincount(rows, minlength=M) <= 2048           # a row longer than the 16 KiB tile is summed by a whole workgroup: not the sequential order
    assert np.count_nonzero(~short) == plan["long_rows"]
    np.testing.assert_array_equal(got[short], want[short])
    assert np.all(np.abs(got - want) <= 1e-13 * bound)
    G.autotune()                                             # 518 K entries: above the autotune's floor
    G.spmv(dx, dy)
    assert np.all(np.abs(dy.download() - want) <= 1e-13 * bound)
    G.jacobi(3, dx, dr)
    wj = A.jacobi(3, x, rhs)
    assert rel(dx.download(), wj) <= 1e-12
    for v_, lanes in ((10, 8), (16, 4), (16, 8), (16, 16)):  # x in LDS, one / four rows per group step: what the autotune weighs on this operator
        G.set_variant(v_)
        G.set_lanes_per_row(lanes)
        dx.upload(x)
        G.spmv(dx, dy)
        assert np.all(np.abs(dy.download() - want) <= 1e-13 * bound), (v_, lanes)
        G.jacobi(3, dx, dr)
        assert rel(dx.download(), wj) <= 1e-12, (v_, lanes)


@pytest.mark.parametrize("name", ["poisson24", "irregular5000", "band3000_1400"])
def test_plan_time_builds_on_the_device_equal_the_host_builds(capi, name, monkeypatch):
    """Round 4 moved the plan-time re-orderings to the device (16-bit column codes: k_cc16_count / k_cc16_encode; the x-in-LDS
    plan's window-relative columns and offsets: k_xlds_build; sliced-ELLPACK values: k_sell_scatter; k_csr_cm's column-ordered blocks:
    k_cm_build, a bitonic sort per block in LDS).  The host encoders are still there behind SAENA_HOST_CC16 / SAENA_HOST_XLDS_BUILD /
    SAENA_HOST_CM_BUILD: both ways must choose the same slot/offset split and give the same bits."""
    monkeypatch.setenv("SAENA_KEEP_HOST_VALUES", "1")
    entries, M = get_problem(name)
    A = orc.OracleOp(entries, M, M, orc.split_even(M, 1))
    x, rhs = inputs.v2(M), inputs.rhs2(M)
    dx, dr = capi.DeviceVector(M, x), capi.DeviceVector(M, rhs)

    def run(variant, lanes):
        G = util.gpu_operator(A)
        try:
            G.set_variant(variant)
        except capi.SgpuError:
            return None
        G.set_lanes_per_row(lanes)
        dy, du = capi.DeviceVector(M), capi.DeviceVector(M, x)
        G.spmv(dx, dy)
        G.jacobi(2, du, dr)
        return G.variant()[1], dy.download(), du.download()
    for variant, lanes in ((3, 1), (4, 4), (7, 1), (8, 4), (9, 1), (10, 8), (16, 8)):
        dev = run(variant, lanes)
        monkeypatch.setenv("SAENA_HOST_CC16", "1")
        monkeypatch.setenv("SAENA_HOST_XLDS_BUILD", "1")
        monkeypatch.setenv("SAENA_HOST_CM_BUILD", "1")
        hst = run(variant, lanes)
        monkeypatch.delenv("SAENA_HOST_CC16")
        monkeypatch.delenv("SAENA_HOST_XLDS_BUILD")
        monkeypatch.delenv("SAENA_HOST_CM_BUILD")
        assert (dev is None) == (hst is None), (variant, "one encoder accepts what the other refuses")
        if dev is None:
            continue
        assert dev[0] == hst[0], (variant, dev[0], hst[0])            # the same slot / offset split in the kernel's name
        np.testing.assert_array_equal(dev[1], hst[1])
        np.testing.assert_array_equal(dev[2], hst[2])
    bound = abs_bound(entries, M, x)
    assert np.all(np.abs(run(3, 1)[1] - A.matvec(x)) <= TOL_SPMV * bound + 1e-300)


def test_plan_cache_honours_every_variant_the_autotune_can_store(capi, tmp_path, monkeypatch):
    """Round-3 advisor finding: the lookup accepted variants 0..14 while the autotune can pick and store 15 (k_sellpx), so a cached
    k_sellpx line was never honoured and every process tuned again and appended another line.  Lookup, store and set_variant
    now share one bound (MAX_VARIANT): a cached line naming the highest variant is taken without a sweep and without a new line."""
    cache = tmp_path / "plans.tsv"
    monkeypatch.setenv("SAENA_PLAN_CACHE", str(cache))
    M = 16384
    entries = orc.band_matrix(M, 63)                        # ~1 M entries, 63 per row: the row-pattern forms apply, k_sellpx included
    A = orc.OracleOp(entries, M, M, orc.split_even(M, 1))
    G1 = util.gpu_operator(A)
    G1.autotune()
    lines = cache.read_text().strip().splitlines()
    assert len(lines) == 1
    f = lines[0].split("\t")
    f[1] = "15"                                              # what an autotune that picked k_sellpx writes
    cache.write_text("\t".join(f) + "\n")
    G2 = util.gpu_operator(A)
    G2.autotune()
    assert G2.variant() == (15, "k_sellpx"), G2.variant()
    assert len(cache.read_text().strip().splitlines()) == 1, "a cached k_sellpx plan must be taken, not tuned again"
    x = inputs.v2(M)
    dx, dy = capi.DeviceVector(M, x), capi.DeviceVector(M)
    G2.spmv(dx, dy)
    np.testing.assert_array_equal(dy.download(), A.matvec(x))         # sequential row sum: the oracle's bit for bit
    f[1] = "17"                                              # beyond the table: ignored (tuned afresh, one more line)
    cache.write_text("\t".join(f) + "\n")
    G3 = util.gpu_operator(A)
    G3.autotune()
    assert G3.variant()[0] <= 16 and len(cache.read_text().strip().splitlines()) == 2


def test_wave_streamed_kernel_keeps_a_nan_in_the_rows_that_own_it(capi):
    """k_csr_wave reads whole 16-byte quads, so the first and last quad of a row carry entries of its neighbours: those
    are dropped with BOTH factors zeroed -- a NaN in x at a column only the neighbouring row owns must not reach this row."""
    M = 64
    entries = orc.band_matrix(M, 63)
    A = orc.OracleOp(entries, M, M, orc.split_even(M, 1))
    G = util.gpu_operator(A)
    G.set_variant(6)
    rows, cols = np.asarray(entries["row"]), np.asarray(entries["col"])
    owners = {c: set(rows[cols == c].tolist()) for c in (0, M - 1)}
    for c in (0, M - 1):
        x = inputs.v2(M)
        x[c] = np.nan
        for lanes in (8, 16, 64):
            G.set_lanes_per_row(lanes)
            dx, dy = capi.DeviceVector(M, x), capi.DeviceVector(M)
            G.spmv(dx, dy)
            got = dy.download()
            bad = set(np.flatnonzero(np.isnan(got)).tolist())
            assert bad == owners[c], f"NaN at column {c}: rows {sorted(bad ^ owners[c])} differ from the owners"


def test_column_ordered_form_refuses_blocks_without_entries(capi, monkeypatch):
    """A row block whose rows hold no local entry (a transfer operator's rows whose entries are all remote) has no segment
    table: k_csr_cm is refused for such an operator instead of decoding the next block's entries through it."""
    monkeypatch.setenv("SAENA_KEEP_HOST_VALUES", "1")
    M = 3072
    rows = np.concatenate([np.arange(0, 600), np.arange(2200, M)]).astype(np.int32)       # rows 600..2199 are empty: whole blocks of 256 and of 512 rows
    entries = orc.coo_from_arrays(np.repeat(rows, 3), ((np.repeat(rows, 3) + np.tile([0, 5, 11], rows.size)) % M).astype(np.int32),
                                  np.tile([4.0, -1.0, -2.0], rows.size))
    A = orc.OracleOp(entries, M, M, orc.split_even(M, 1))
    G = util.gpu_operator(A)
    for v in (7, 8):
        with pytest.raises(capi.SgpuError, match="column-major"):
            G.set_variant(v)
    G.set_variant(4); G.set_lanes_per_row(1)
    x = inputs.v2(M)
    dx, dy = capi.DeviceVector(M, x), capi.DeviceVector(M)
    G.spmv(dx, dy)
    np.testing.assert_array_equal(dy.download(), A.matvec(x))

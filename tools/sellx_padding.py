"""Padding of a sliced-ELLPACK layout inside the (row chunk, column window) blocks of k_csr_xlds (development aid):
    python tools/sellx_padding.py m
per operator of the hierarchy with >= 40 entries per row: stored / actual entries when every 64-row slice of a chunk pads
its rows' pieces to the slice's longest piece, rows in natural order and sorted by piece length inside the chunk."""
import sys

import numpy as np

sys.path.insert(0, ".")
from saena_amd import capi, host

m = int(sys.argv[1]) if len(sys.argv) > 1 else 128
capi.init(0)
L = host.load("gpu")
A = host.Matrix(host.Comm("gpu", "rccl")).laplacian3D(m).assemble()
S = host.AmgSolver(A, host.options(L, **host.OPTIONS001))
XL, NCU = 20224, 256
for l in range(S.num_levels - 1):
    for which, name in ((0, "A"), (1, "P"), (2, "R")):
        d = S.level_layout(l, which)
        M, npr, col = d["M"], d["nnzPerRow_local"], d["col_local"]
        nnz = len(col)
        if nnz / M < 40 or M < 20000:
            continue
        ptr = np.concatenate([[0], np.cumsum(npr)])
        cuts = np.searchsorted(ptr, np.arange(NCU + 1) * nnz / NCU)
        cuts[-1] = M
        pad_nat = pad_sort = tot = 0
        maxT = 0
        for c in range(0, NCU, 16):                       # a sample of the chunks
            r0, r1 = cuts[c], cuts[c + 1]
            if r1 <= r0:
                continue
            cc = col[ptr[r0]:ptr[r1]]
            c0 = cc.min()
            T = (cc.max() - c0) // XL + 1
            maxT = max(maxT, T)
            rows = np.repeat(np.arange(r1 - r0), npr[r0:r1])
            cnt = np.zeros((r1 - r0, T), int)
            np.add.at(cnt, (rows, (cc - c0) // XL), 1)
            tot += cnt.sum()
            order = np.argsort(-cnt.sum(axis=1), kind="stable")
            for arr, which_pad in ((cnt, "nat"), (cnt[order], "sort")):
                p = 0
                for s in range(0, r1 - r0, 64):
                    blk = arr[s:s + 64]
                    p += (blk.max(axis=0) * len(blk)).sum()          # a partial last slice stores only its rows
                if which_pad == "nat":
                    pad_nat += p
                else:
                    pad_sort += p
        print(f"L{l} {name}: rows {M} nnz/row {nnz / M:.0f} windows <= {maxT}: stored/nnz natural {pad_nat / tot:.3f}, rows sorted by length in the chunk {pad_sort / tot:.3f}", flush=True)

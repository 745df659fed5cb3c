"""Run-length census of the column ids of every operator of a Poisson hierarchy (how many x[col] gathers could coalesce):
    python tools/column_runs.py [m] > profiles/rNN_column_runs_<m>.log      (host library only, no GPU)"""
import sys, numpy as np
sys.path.insert(0,'.')
from saena_amd import host
L = host.load("host")
m = int(sys.argv[1]) if len(sys.argv) > 1 else 64
A = host.Matrix(host.Comm("host","self")).laplacian3D(m).assemble()
S = host.AmgSolver(A, host.options(L, **host.OPTIONS001))
for l in range(S.num_levels):
    for which,name in ((0,'A'),(1,'P'),(2,'R')):
        if which and l == S.num_levels-1: continue
        d = S.level_layout(l, which)
        col = d["col_local"]; npr = d["nnzPerRow_local"]
        if len(col) < 1000: continue
        rp = np.concatenate([[0], np.cumsum(npr)])
        brk = np.ones(len(col), bool)
        brk[1:] = col[1:] != col[:-1] + 1
        brk[rp[:-1][npr>0]] = True
        nruns = brk.sum()
        # run-length histogram
        starts = np.flatnonzero(brk); lens = np.diff(np.concatenate([starts,[len(col)]]))
        # number of 16B gathers if runs are split into pairs: ceil(len/2)
        pairs = ((lens+1)//2).sum()
        quads = ((lens+3)//4).sum()
        print(f"L{l} {name}: rows {len(npr)} nnz {len(col)} ({len(col)/len(npr):.1f}/row) runs {nruns} avg run {len(col)/nruns:.2f} | 16B-gathers {pairs/len(col):.2f}/nnz, 32B {quads/len(col):.2f}/nnz | run len pct: 1:{(lens==1).mean()*100:.0f}% 2:{(lens==2).mean()*100:.0f}% 3-4:{((lens>=3)&(lens<=4)).mean()*100:.0f}% 5-8:{((lens>=5)&(lens<=8)).mean()*100:.0f}% >8:{(lens>8).mean()*100:.0f}%")

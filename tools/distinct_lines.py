"""Distinct columns and distinct 128-byte x lines per row block of every operator of a Poisson hierarchy (host library only):
    python tools/distinct_lines.py [m] > profiles/rNN_distinct_lines_<m>.log"""
import sys, numpy as np
sys.path.insert(0,'.')
from saena_amd import host
L = host.load("host")
m = int(sys.argv[1]) if len(sys.argv) > 1 else 64
A = host.Matrix(host.Comm("host","self")).laplacian3D(m).assemble()
S = host.AmgSolver(A, host.options(L, **host.OPTIONS001))
for l in range(min(S.num_levels,5)):
    for which,name in ((0,'A'),(2,'R'),(1,'P')):
        if which and l == S.num_levels-1: continue
        d = S.level_layout(l, which)
        col = d["col_local"]; npr = d["nnzPerRow_local"]
        if len(col) < 100000: continue
        rp = np.concatenate([[0], np.cumsum(npr)])
        for cap in (2048, 4096):
            # blocks of consecutive rows up to cap nnz
            r=0; M=len(npr); us=[]; ls=[]; nn=[]
            cnt=0
            while r < M and cnt < 400:
                s=r; p0=rp[r]
                while r < M and rp[r+1]-p0 <= cap and r-s < 512: r+=1
                if r==s: r+=1
                c = col[p0:rp[r]]
                us.append(len(np.unique(c))); ls.append(len(np.unique(c>>4))); nn.append(len(c)); cnt+=1
                r += max(0,(M//400) - (r-s))   # sample
            print(f"L{l} {name} ({len(col)/len(npr):.0f}/row) cap {cap}: nnz/blk {np.mean(nn):.0f} distinct cols {np.mean(us):.0f} (reuse {np.mean(nn)/np.mean(us):.2f}x) distinct 128B lines {np.mean(ls):.0f} ({np.mean(ls)*128/1024:.0f} KB)")

"""State of the pair certificates of a converged alignment: python scratch/dbg_uncert.py [c4|c5] [points] [iters]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "icp-symm_amd", "py"))
import numpy as np, symmicp
from symmicp import synth
from scipy.spatial import cKDTree
if __name__ == "__main__":
    wl = sys.argv[1] if len(sys.argv) > 1 else "c4"; n = int(sys.argv[2]) if len(sys.argv) > 2 else 1000000; iters = int(sys.argv[3]) if len(sys.argv) > 3 else 14
    d = dict(c4=synth.c4_surface, c5=synth.c5_scan)[wl](n)
    with symmicp.Engine(mode=symmicp.MODE_PAPER, corr=symmicp.CORR_TREE, max_iters=iters, fixed_iters=1, host_loop=1) as e:
        e.set_target(d["tgt"], d["tgt_n"]); e.set_source(d["src"], d["src_n"])
        r = e.align()
        ce, hood, T, win = e.certificates()
    flag = (ce[:, 3].view(np.uint32) & 1).astype(bool)
    print("single certificate:", (ce[:, 3] > 0).sum(), " neighbourhood:", flag.sum(), " neighbourhood only:", (flag & (ce[:, 3] < 0)).sum(), " none:", ((ce[:, 3] == 0)).sum())
    tree = cKDTree(d["tgt"].astype(np.float64))
    dist, idx = tree.query(ce[:, :3].astype(np.float64), k=2)
    single = ce[:, 3] > 0
    L = ce[:, 3].astype(np.float64)
    T = T.astype(np.float64)
    print("room of the single certificates (L - d1) / d1, quantiles:", np.quantile((L[single] - dist[single, 0]) / dist[single, 0], [0, 1e-4, 1e-3, .01, .1, .5]))
    print("room of the neighbourhoods (T - d1) / d1, quantiles:     ", np.quantile((T[flag] - dist[flag, 0]) / dist[flag, 0], [0, 1e-4, 1e-3, .01, .1, .5]))
    print("members kept (winner included), histogram:", np.bincount((hood[flag] != 0xFFFFFFFF).sum(1), minlength=9))

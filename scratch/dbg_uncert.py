"""Which pairs of a converged alignment hold no certificate, and why: python scratch/dbg_uncert.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "icp-symm_amd", "py"))
import numpy as np, symmicp
from symmicp import synth
from scipy.spatial import cKDTree
d = synth.c4_surface(1000000)
with symmicp.Engine(mode=symmicp.MODE_PAPER, corr=symmicp.CORR_TREE, max_iters=14, fixed_iters=1, host_loop=1) as e:
    e.set_target(d["tgt"], d["tgt_n"]); e.set_source(d["src"], d["src_n"])
    r = e.align()
    ce, ru = e.certificates()
    X = r["transform"].astype(np.float64)
print("no certificate:", (ce[:, 3] == 0).sum(), " two-candidate:", (ce[:, 3] < 0).sum(), " single:", (ce[:, 3] > 0).sum())
tree = cKDTree(d["tgt"].astype(np.float64))
dist, idx = tree.query(ce[:, :3].astype(np.float64), k=3)
L = np.abs(ce[:, 3].astype(np.float64))
single = ce[:, 3] > 0
slack = (L - dist[:, 0]) / dist[:, 0]
print("single certificates: relative slack (L - d1) / d1 quantiles", np.quantile(slack[single], [0, 1e-5, 1e-4, 1e-3, .01, .5]))
for thr in (1e-6, 3e-6, 1e-5, 3e-5, 1e-4):
    print("  slack < %g: %d pairs" % (thr, (slack[single] < thr).sum()))
two = ce[:, 3] < 0
print("two-candidate: slack (L3 - d2) / d2", np.sort((L[two] - dist[two, 1]) / dist[two, 1])[:10])

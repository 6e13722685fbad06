"""Pairs of a converged alignment that still hold no usable certificate: python scratch/dbg_persist.py [c4|c5] [points] [iters]"""
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
        st = e.stats()
    flag = (ce[:, 3].view(np.uint32) & 1).astype(bool)
    tree = cKDTree(d["tgt"].astype(np.float64))
    ext = (d["tgt"].max(0) - d["tgt"].min(0)).max()
    print("grid level", st["grid_level"], "cell ~", ext / 2 ** st["grid_level"])
    sel = np.nonzero(~(ce[:, 3] > 0))[0]
    print(len(sel), "pairs without a single certificate;", (flag[sel]).sum(), "of them with a neighbourhood")
    dist, idx = tree.query(ce[sel, :3].astype(np.float64), k=4)
    order = np.argsort(flag[sel], kind='stable')
    for k in order[:30]:
        i = sel[k]
        print(i, "w=%g flag=%d T=%.6g members=%d | d1..d4 = %s" % (ce[i, 3], flag[i], T[i], (hood[i] != 0xFFFFFFFF).sum(), np.array2string(dist[k], precision=6)))
    # room of the certificates that exist
    dist1 = tree.query(ce[:, :3].astype(np.float64), k=1)[0]
    single = ce[:, 3] > 0
    room = (ce[single, 3] - dist1[single]) / dist1[single]
    print("single certificates: room / d1 quantiles", np.quantile(room, [0, 1e-5, 1e-4, 1e-3]))
    print("  below 1e-5:", (room < 1e-5).sum(), " of them with a neighbourhood:", (flag[single][room < 1e-5]).sum())

"""What makes a packet slow: production-kernel packet durations (SYMMICP_DEBUG_TRACE alone) against the geometry of the packet's queries.
SYMMICP_DEBUG_TRACE=/tmp/t.bin python scratch/pkt_why.py   (the trace carries each packet's first query; the source is given in its sorted order)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "icp-symm_amd", "py"))
import numpy as np, symmicp
from symmicp import synth
from scipy.spatial import cKDTree
if __name__ == "__main__":
    d = synth.c4_surface(1000000)
    o = synth.sweep_order(d["src"]); src, sn = d["src"][o], d["src_n"][o]
    with symmicp.Engine(mode=symmicp.MODE_PAPER, corr=symmicp.CORR_TREE, max_iters=2, fixed_iters=1, sort_source=0) as e:
        e.set_target(d["tgt"], d["tgt_n"]); e.set_source(src, sn)
        e.begin(); e.begin()
    t = np.fromfile(os.environ["SYMMICP_DEBUG_TRACE"], dtype=np.uint64).reshape(-1, 2)
    t = t[t[:, 0] != 0]
    dur = (t[:, 1] >> np.uint64(32)).astype(np.float64) * 0.01          # us
    first = ((t[:, 1] & np.uint64(0xFFFFFFFF)) >> np.uint64(1)).astype(np.int64)
    order = np.argsort(first); first, dur, slot = first[order], dur[order], order
    count = np.minimum(np.diff(np.append(first, len(src))), 64)
    tree = cKDTree(d["tgt"].astype(np.float64))
    d1_all = tree.query(src.astype(np.float64), k=1)[0]
    rad = np.empty(len(first)); d1m = np.empty(len(first)); d1x = np.empty(len(first))
    for k, (f, c) in enumerate(zip(first, count)):
        g = src[f:f + c].astype(np.float64)
        rad[k] = np.sqrt(((g - g.mean(0)) ** 2).sum(1).max()); d1m[k] = d1_all[f:f + c].mean(); d1x[k] = d1_all[f:f + c].max()
    feats = dict(count=count.astype(np.float64), radius=rad, d1_mean=d1m, d1_max=d1x, rad_plus_d1=rad + d1x, slot=slot.astype(np.float64))
    print("packets %d, duration mean %.0f us, max %.0f us; counts: %s" % (len(first), dur.mean(), dur.max(), np.bincount(np.minimum(count // 16, 4))))
    for k, v in feats.items():
        print("  corr(duration, %s) = %.3f" % (k, np.corrcoef(dur, v)[0, 1]))
    med = {k: np.median(v) for k, v in feats.items()}
    for p in np.argsort(-dur)[:15]:
        print("  first %7d: %4.0f us  " % (first[p], dur[p]) + "  ".join(("%s %.0f" % (k, v[p])) if k in ("count", "slot") else ("%s %.1fx" % (k, v[p] / med[k])) for k, v in feats.items()))

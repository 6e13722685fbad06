"""What makes a packet slow: release-kernel packet durations (SYMMICP_DEBUG_TRACE) against the geometry of the packet's queries.
SYMMICP_DEBUG_TRACE=/tmp/t.bin SYMMICP_PACKET_ORDER=0 python scratch/pkt_why.py   (packets in Morton order: slot = block of 64 rows)"""
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
    n = len(src) // 64
    dur = (t[:n, 1] >> np.uint64(32)).astype(np.float64) * 0.01          # us
    ovf = (t[:n, 1] & np.uint64(1)).astype(bool)
    tree = cKDTree(d["tgt"].astype(np.float64))
    d1 = tree.query(src.astype(np.float64), k=1)[0][: n * 64].reshape(n, 64)
    g = src[: n * 64].reshape(n, 64, 3).astype(np.float64)
    rad = np.sqrt(((g - g.mean(1, keepdims=True)) ** 2).sum(2).max(1))
    feats = dict(radius=rad, d1_mean=d1.mean(1), d1_max=d1.max(1), d1_spread=d1.max(1) - d1.min(1), rad_plus_d1=rad + d1.max(1))
    print("packets %d, duration mean %.0f us, max %.0f us, fallback %d" % (n, dur.mean(), dur.max(), ovf.sum()))
    for k, v in feats.items():
        print("  corr(duration, %s) = %.3f   (log-log %.3f)" % (k, np.corrcoef(dur, v)[0, 1], np.corrcoef(np.log(dur + 1), np.log(v + 1e-9))[0, 1]))
    top = np.argsort(-dur)[:15]
    med = {k: np.median(v) for k, v in feats.items()}
    for p in top:
        print("  packet %6d: %4.0f us%s  " % (p, dur[p], "*" if ovf[p] else " ") + "  ".join("%s %.1fx" % (k, v[p] / med[k]) for k, v in feats.items()))

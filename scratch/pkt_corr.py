"""Correlate per-packet duration (trace) with the packet's bounding-box diagonal in source coordinates: python scratch/pkt_corr.py trace.bin [workload] [n]"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "icp-symm_amd", "py")); sys.path.insert(0, os.path.dirname(__file__))
import numpy as np
from symmicp import synth
import sim_packet as S
wl = sys.argv[2] if len(sys.argv) > 2 else "c4"; n = int(sys.argv[3]) if len(sys.argv) > 3 else 1000000
d = dict(c3=synth.c3_uniform, c4=synth.c4_surface, c5=synth.c5_scan)[wl](n)
src = d["src"].astype(np.float32)
lo = src.min(0); emax = np.float32((src.max(0) - lo).max()); h0 = np.float32(emax * np.float32(1.00001) / np.float32(1024))
so = np.argsort(S.morton(src.astype(np.float64), lo.astype(np.float64), float(h0)), kind="stable"); src = src[so]
npk = len(src) // 64
G = src[: npk * 64].reshape(npk, 64, 3)
diag = np.linalg.norm(G.max(1) - G.min(1), axis=1)
t = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 2)[:npk]
dur = (t[:, 1] >> np.uint64(32)).astype(np.float64) / 100.0      # us
steps = ((t[:, 1] >> np.uint64(8)) & np.uint64(0xFFFFFF)).astype(np.int64); ovf = (t[:, 1] & np.uint64(1)).astype(bool)
start = t[:, 0].astype(np.float64) / 100.0; start -= start[start > 0].min()
print("packets %d  overflowed %d  dur mean %.0f us  max %.0f us;  steps mean %.1f max %d" % (npk, ovf.sum(), dur.mean(), dur.max(), steps.mean(), steps.max()))
print("corr(dur, diag) = %.2f   corr(dur, steps) = %.2f" % (np.corrcoef(dur, diag)[0, 1], np.corrcoef(dur, steps)[0, 1]))
o = np.argsort(-dur)[:12]
print("slowest: " + "; ".join("dur %.0f steps %d diag %.3f (rank %d) ovf %d start %.0f" % (dur[k], steps[k], diag[k], (diag > diag[k]).sum(), ovf[k], start[k]) for k in o))
for q in (0.5, 0.9, 0.99):
    m = diag >= np.quantile(diag, q)
    print("diag >= q%.2f: share of total packet time %.2f, mean dur %.0f" % (q, dur[m].sum() / dur.sum(), dur[m].mean()))
end = start + dur
print("kernel span %.0f us; sum of durations / span = %.0f concurrent" % (end.max(), dur.sum() / end.max()))

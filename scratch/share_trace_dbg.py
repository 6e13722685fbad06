"""Instrumented per-packet trace (SYMMICP_DEBUG_COUNTERS + SYMMICP_DEBUG_TRACE: 8 words per packet) of one share:
python scratch/share_trace_dbg.py c4 1000000 8 rank [waves]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "icp-symm_amd", "py"))
import numpy as np, symmicp as sym
from symmicp import synth
wl, n, world, r = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
w = sys.argv[5] if len(sys.argv) > 5 else "1"
d = dict(c4=synth.c4_surface, c5=synth.c5_scan)[wl](n)
o = synth.sweep_order(d["src"]) if wl != "c5" else np.arange(n)
src, sn = np.ascontiguousarray(d["src"][o]), np.ascontiguousarray(d["src_n"][o])
tf = "/tmp/share_trace_dbg.bin"
os.environ["SYMMICP_DEBUG_TRACE"] = tf; os.environ["SYMMICP_DEBUG_COUNTERS"] = "1"; os.environ["SYMMICP_PACKET_WAVES"] = w
b0, b1 = n * r // world, n * (r + 1) // world
with sym.Engine(mode=sym.MODE_PAPER, corr=sym.CORR_TREE, max_iters=2, fixed_iters=1) as e:
    e.set_target(d["tgt"], d["tgt_n"]); e.set_source(src[b0:b1], sn[b0:b1])
    e.begin(); e.begin()
t = np.fromfile(tf, dtype=np.uint64).reshape(-1, 8)
t = t[t[:, 0] != 0]
dur = (t[:, 1] >> np.uint64(32)).astype(np.float64) * 0.01
W = int(w)
def unpack(x):
    return (x >> np.uint64(48)).astype(np.int64), ((x >> np.uint64(32)) & np.uint64(0xFFFF)).astype(np.int64), ((x >> np.uint64(8)) & np.uint64(0xFFFFFF)).astype(np.int64), (x & np.uint64(0xFF)).astype(np.int64)
per = [unpack(t[:, 2 + k]) for k in range(W)]
steps = sum(p[0] for p in per); scanned = sum(p[1] for p in per); points = sum(p[2] for p in per); ties = sum(p[3] for p in per)
nodes = (t[:, 6] >> np.uint64(32)).astype(np.int64); cand = (t[:, 6] & np.uint64(0xFFFFFFFF)).astype(np.int64)
print("W=%s rank %d: %d packets; dur median %.0f max %.0f us; steps median %d max %d; leaves scanned median %d max %d; points median %d max %d" %
      (w, r, len(dur), np.median(dur), dur.max(), np.median(steps), steps.max(), np.median(scanned), scanned.max(), np.median(points), points.max()))
for k in np.argsort(-dur)[:8]:
    print("   %4.0f us: steps %d (per wave %s) nodes(w0) %d leaf candidates(w0) %d scanned %d points %d ties %d" %
          (dur[k], steps[k], [int(p[0][k]) for p in per], nodes[k], cand[k], scanned[k], points[k], ties[k]))
print("   corr(dur, steps) %.2f  corr(dur, points) %.2f  corr(dur, scanned) %.2f" % (np.corrcoef(dur, steps)[0, 1], np.corrcoef(dur, points)[0, 1], np.corrcoef(dur, scanned)[0, 1]))

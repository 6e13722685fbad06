"""Geometry of the packets that overflow their frontier (depth-first fallback) in one share: python scratch/ovf_packets.py c4 1000000 8 rank"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "icp-symm_amd", "py"))
import numpy as np, symmicp as sym
from symmicp import synth
from scipy.spatial import cKDTree
wl, n, world, r = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
d = dict(c4=synth.c4_surface, c5=synth.c5_scan)[wl](n)
o = synth.sweep_order(d["src"]) if wl != "c5" else np.arange(n)
src, sn = np.ascontiguousarray(d["src"][o]), np.ascontiguousarray(d["src_n"][o])
tf = "/tmp/ovf_trace.bin"
os.environ["SYMMICP_DEBUG_TRACE"] = tf; os.environ["SYMMICP_PACKET_WAVES"] = "1"
b0, b1 = n * r // world, n * (r + 1) // world
with sym.Engine(mode=sym.MODE_PAPER, corr=sym.CORR_TREE, max_iters=2, fixed_iters=1) as e:
    e.set_target(d["tgt"], d["tgt_n"]); e.set_source(src[b0:b1], sn[b0:b1])
    e.begin()
    ce = e.certificates()[0]
t = np.fromfile(tf, dtype=np.uint64).reshape(-1, 2)
t = t[t[:, 0] != 0]
dur = (t[:, 1] >> np.uint64(32)).astype(np.float64) * 0.01
first = ((t[:, 1] & np.uint64(0xFFFFFFFF)) >> np.uint64(1)).astype(np.int64)
ovf = (t[:, 1] & np.uint64(1)).astype(bool)
order = np.argsort(first)
fs = first[order]; cnt = np.minimum(np.diff(np.append(fs, b1 - b0)), 64)
count = np.empty_like(cnt); count[:] = cnt
cnt_of = dict(zip(fs.tolist(), cnt.tolist()))
tree = cKDTree(d["tgt"].astype(np.float64))
P = ce[:, :3].astype(np.float64)
d1 = tree.query(P, k=1)[0]
spacing = 1.0 / np.sqrt(n)
print("rank %d: %d packets, %d overflowed; count histogram (16s): %s; spacing %.4g" % (r, len(dur), ovf.sum(), np.bincount(np.minimum(cnt // 16, 4)), spacing))
def describe(k):
    f, c = int(first[k]), int(cnt_of[int(first[k])])
    g = P[f:f + c]
    ext = g.max(0) - g.min(0)
    sub = [np.linalg.norm(g[j:j + 16].max(0) - g[j:j + 16].min(0)) for j in range(0, c, 16)]
    steps = np.linalg.norm(np.diff(g, axis=0), axis=1) if c > 1 else np.zeros(1)
    return "first %6d count %2d dur %4.0f us%s: extent %s (diag %.1f spacings), sub-group diags %s, d1 median %.1f max %.1f spacings, steps between queries median %.1f max %.1f" % (
        f, c, dur[k], " OVF" if ovf[k] else "", np.round(ext / spacing, 1), np.linalg.norm(ext) / spacing, np.round(np.array(sub) / spacing, 1), np.median(d1[f:f + c]) / spacing, d1[f:f + c].max() / spacing, np.median(steps) / spacing, steps.max() / spacing)
for k in np.nonzero(ovf)[0]:
    print("  " + describe(k))
print(" slowest others:")
for k in [k for k in np.argsort(-dur) if not ovf[k]][:5]:
    print("  " + describe(k))
print(" median packets:")
for k in np.argsort(dur)[len(dur) // 2: len(dur) // 2 + 3]:
    print("  " + describe(k))

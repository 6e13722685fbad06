"""Per-packet durations of the first pass of every share of an N-way split (production kernel, timing-only trace):
python scratch/share_trace.py c4 1000000 8 [waves ...]      (SYMMICP_DEBUG_TRACE / SYMMICP_PACKET_WAVES are set per engine here)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "icp-symm_amd", "py"))
import numpy as np, symmicp as sym
from symmicp import synth
from scipy.spatial import cKDTree
wl, n, world = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
waves = sys.argv[4:] or ["1", "4"]
d = dict(c4=synth.c4_surface, c5=synth.c5_scan)[wl](n)
o = synth.sweep_order(d["src"]) if wl != "c5" else np.arange(n)
src, sn = np.ascontiguousarray(d["src"][o]), np.ascontiguousarray(d["src_n"][o])
tree = cKDTree(d["tgt"].astype(np.float64))
d1_all = tree.query(src.astype(np.float64), k=1)[0]
print("NN distance of the queries: median %.4g, 99%% %.4g, max %.4g" % (np.median(d1_all), np.quantile(d1_all, .99), d1_all.max()))
tf = "/tmp/share_trace.bin"
os.environ["SYMMICP_DEBUG_TRACE"] = tf
for w in waves:
    os.environ["SYMMICP_PACKET_WAVES"] = w
    for r in range(world):
        b0, b1 = n * r // world, n * (r + 1) // world
        with sym.Engine(mode=sym.MODE_PAPER, corr=sym.CORR_TREE, max_iters=2, fixed_iters=1, sort_source=1) as e:
            e.set_target(d["tgt"], d["tgt_n"]); e.set_source(src[b0:b1], sn[b0:b1])
            e.begin(); e.enable_timing(2); e.reset_stats(); e.begin()
            kms = e.stats()["kernel_ms"][2]
        t = np.fromfile(tf, dtype=np.uint64).reshape(-1, 2)
        t = t[t[:, 0] != 0]
        start = t[:, 0].astype(np.float64) * 0.01; start -= start.min()
        dur = (t[:, 1] >> np.uint64(32)).astype(np.float64) * 0.01          # us
        end = start + dur
        slow = np.argsort(-dur)[:3]
        print("W=%s rank %d: kernel %.0f us, %d packets, dur sum %.1f ms median %.0f 99%% %.0f max %.0f us; last start %.0f us, last end %.0f us; slowest start at %s us; share d1 median %.4g max %.4g"
              % (w, r, kms * 1e3, len(dur), dur.sum() / 1e3, np.median(dur), np.quantile(dur, .99), dur.max(), start.max(), end.max(),
                 ",".join("%.0f" % start[k] for k in slow), np.median(d1_all[b0:b1]), d1_all[b0:b1].max()), flush=True)

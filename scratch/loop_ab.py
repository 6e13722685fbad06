"""Device-driven loop vs host loop, wall time of one alignment: python scratch/loop_ab.py [workload] [points] [iters] [corr]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "icp-symm_amd", "py"))
import numpy as np, symmicp
from symmicp import synth
wl = sys.argv[1] if len(sys.argv) > 1 else "c4"; n = int(sys.argv[2]) if len(sys.argv) > 2 else 1000000
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 30; corr = sys.argv[4] if len(sys.argv) > 4 else "tree"
d = dict(c3=synth.c3_uniform, c4=synth.c4_surface, c5=synth.c5_scan)[wl](n)
if corr == "identity":
    d["tgt"] = (d["src"].astype(np.float64) @ d["truth"][:3, :3].T + d["truth"][:3, 3]).astype(np.float32)
    d["tgt_n"] = (d["src_n"].astype(np.float64) @ d["truth"][:3, :3].T).astype(np.float32)
for host_loop in (1, 0, 1, 0):
    with symmicp.Engine(mode=symmicp.MODE_PAPER, corr=getattr(symmicp, "CORR_" + corr.upper()), max_iters=iters, fixed_iters=1, host_loop=host_loop) as e:
        e.set_target(d["tgt"], d["tgt_n"]); e.set_source(d["src"], d["src_n"])
        e.align()
        best = 1e9
        for _ in range(5):
            t0 = time.perf_counter(); r = e.align(); best = min(best, time.perf_counter() - t0)
        print("%s %d %s host_loop=%d: %d iters %.3f ms -> %.1f iter/s (%.1f us per iteration)" % (wl, n, corr, host_loop, r["iters"], best * 1e3, iters / best, best / iters * 1e6))

"""Kernel time of every pass of an alignment, host-driven (begin + step, timing mode 1): python scratch/per_pass_ms.py c5 8000000 50
With SYMMICP_DEBUG_COUNTERS=1 the library prints its per-pass search counters next to it."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "icp-symm_amd", "py"))
import numpy as np, symmicp as sym
from symmicp import synth
wl, n, iters = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
d = dict(c3=synth.c3_uniform, c4=synth.c4_surface, c5=synth.c5_scan)[wl](n, **(dict(workers=16) if wl == "c5" and n >= 1000000 else {}))
with sym.Engine(mode=sym.MODE_PAPER, corr=sym.CORR_TREE, max_iters=iters, fixed_iters=1) as e:
    e.set_target(d["tgt"], d["tgt_n"]); e.set_source(d["src"], d["src_n"])
    e.align()                     # warm-up
    e.enable_timing(1)
    ms = []
    e.begin(); ms.append(e.stats()["last_pass_ms"])
    for k in range(iters):
        e.step(); ms.append(e.stats()["last_pass_ms"])
print("%s %d: pass kernel ms: %s" % (wl, n, " ".join("%.3f" % m for m in ms)))
print("sum %.2f ms, mean %.3f ms; first 3: %.2f; passes 4-12: %.2f; 13-25: %.2f; rest: %.2f" % (sum(ms), sum(ms) / len(ms), sum(ms[:3]), sum(ms[3:12]), sum(ms[12:25]), sum(ms[25:])))

"""Time the FIRST pass of an alignment (symmicp_begin: no previous pairs) per kernel: python scratch/first_pass.py [workload] [points] [reps]
With SYMMICP_DEBUG_COUNTERS=1 the library prints its per-pass counters (packet visits, rejected pops, points tested)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "icp-symm_amd", "py"))
import numpy as np, symmicp
from symmicp import synth
wl = sys.argv[1] if len(sys.argv) > 1 else "c4"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
d = dict(c3=synth.c3_uniform, c4=synth.c4_surface, c5=synth.c5_scan)[wl](n)
eng = symmicp.Engine(mode=symmicp.MODE_PAPER, corr=symmicp.CORR_TREE, max_iters=30, fixed_iters=1)
eng.set_target(d["tgt"], d["tgt_n"]); eng.set_source(d["src"], d["src_n"])
eng.begin()
eng.enable_timing(2)
eng.reset_stats()
t0 = time.perf_counter()
for _ in range(reps):
    eng.begin()
wall = (time.perf_counter() - t0) / reps
st = eng.stats()
names = symmicp.KERNEL_SLOTS
print("%s %d: first pass wall %.3f ms; " % (wl, n, wall * 1e3) + "  ".join("%s %.1f us" % (names[k], 1e3 * st["kernel_ms"][k] / max(1, st["kernel_launches"][k])) for k in range(len(names)) if st["kernel_launches"][k]))
eng.close()

import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "icp-symm_amd", "py"))
import numpy as np, symmicp
from symmicp import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
d = synth.c4_surface(n)
eng = symmicp.Engine(mode=symmicp.MODE_PAPER, corr=symmicp.CORR_TREE, max_iters=1, fixed_iters=1)
for rep in range(3):
    t0 = time.perf_counter(); eng.set_target(d["tgt"], d["tgt_n"]); t1 = time.perf_counter(); eng.set_source(d["src"], d["src_n"]); t2 = time.perf_counter()
    print("set_target %.2f ms  set_source %.2f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3))
eng.close()

"""Host-side breakdown of one pass (SYMMICP_DEBUG_HOST=1 prints it from symmicp_destroy)."""
import os, sys
os.environ["SYMMICP_DEBUG_HOST"] = "1"
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "icp-symm_amd", "py"))
import numpy as np, symmicp
from symmicp import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
d = synth.c4_surface(n)
eng = symmicp.Engine(mode=symmicp.MODE_PAPER, corr=symmicp.CORR_TREE, max_iters=100, fixed_iters=1)
eng.set_target(d["tgt"], d["tgt_n"]); eng.set_source(d["src"], d["src_n"])
eng.align()
import time
t0 = time.perf_counter(); r = eng.align(); t1 = time.perf_counter()
print("100 iters: %.3f ms -> %.1f us/iter" % ((t1 - t0) * 1e3, (t1 - t0) * 1e4))
eng.close()

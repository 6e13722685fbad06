"""Per-pass counters of a C5 alignment: SYMMICP_DEBUG_COUNTERS=1 python scratch/c5_passes.py [points] [iters] [host_loop]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "icp-symm_amd", "py"))
import numpy as np, symmicp
from symmicp import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8000000
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 12
hl = int(sys.argv[3]) if len(sys.argv) > 3 else 1
t0 = time.time(); d = synth.c5_scan(n); print("generated in %.1f s" % (time.time() - t0), flush=True)
with symmicp.Engine(mode=symmicp.MODE_PAPER, corr=symmicp.CORR_TREE, max_iters=iters, fixed_iters=1, host_loop=hl) as e:
    e.set_target(d["tgt"], d["tgt_n"]); e.set_source(d["src"], d["src_n"])
    for k in range(2):
        t0 = time.perf_counter(); r = e.align(); dt = time.perf_counter() - t0
        print("align %d: %d iters %.3f ms" % (k, r["iters"], dt * 1e3), flush=True)
    s = e.stats()
    print({k: getattr(s, k) for k in ("pass_ms_first", "pass_ms_mean", "passes_timed") if hasattr(s, k)}, list(s.pass_ms_head) if hasattr(s, "pass_ms_head") else None)

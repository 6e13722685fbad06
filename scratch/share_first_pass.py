"""First pass of ONE rank's share of an N-way sharded alignment, on one GPU: python scratch/share_first_pass.py c4 1000000 8 [rank]
(source rows in sweep order; SYMMICP_DEBUG_COUNTERS=1 SYMMICP_DEBUG_TRACE=file for the per-packet trace)"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "icp-symm_amd", "py"))
import numpy as np, symmicp
from symmicp import synth
if __name__ == "__main__":
    wl, n, world = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    rank = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    d = dict(c4=synth.c4_surface, c5=synth.c5_scan)[wl](n)
    o = synth.sweep_order(d["src"]); src, sn = d["src"][o], d["src_n"][o]
    b0, b1 = n * rank // world, n * (rank + 1) // world
    eng = symmicp.Engine(mode=symmicp.MODE_PAPER, corr=symmicp.CORR_TREE, max_iters=30, fixed_iters=1)
    eng.set_target(d["tgt"], d["tgt_n"]); eng.set_source(src[b0:b1], sn[b0:b1])
    eng.begin(); eng.enable_timing(2); eng.reset_stats()
    for _ in range(5):
        eng.begin()
    st = eng.stats()
    names = symmicp.KERNEL_SLOTS
    print("%s %d, rank %d of %d (%d points): " % (wl, n, rank, world, b1 - b0) + "  ".join("%s %.1f us" % (names[k], 1e3 * st["kernel_ms"][k] / max(1, st["kernel_launches"][k])) for k in range(len(names)) if st["kernel_launches"][k]))
    eng.close()

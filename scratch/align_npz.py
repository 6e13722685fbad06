"""Time one alignment on a cloud pair stored in an .npz (src, src_n, tgt, tgt_n, truth): python align_npz.py file iters"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "icp-symm_amd", "py"))
import numpy as np, symmicp
d = np.load(sys.argv[1]); iters = int(sys.argv[2])
eng = symmicp.Engine(mode=symmicp.MODE_PAPER, corr=symmicp.CORR_TREE, max_iters=iters, fixed_iters=1)
eng.set_target(d["tgt"], d["tgt_n"]); eng.set_source(d["src"], d["src_n"])
eng.align()
best = 1e9
for _ in range(3):
    t0 = time.perf_counter(); r = eng.align(); best = min(best, time.perf_counter() - t0)
print("%d iters: %.3f ms -> %.1f iter/s  err %.2e  repairs %d" % (iters, best * 1e3, iters / best, np.abs(r["transform"] - d["truth"]).max(), eng.stats()["kernel_launches"][7]))
eng.close()

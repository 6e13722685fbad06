"""Per-rank kernel time of an N-way sharded alignment, measured on ONE GPU: N contexts in external-exchange mode run
one after the other, the script plays the all-reduce.  max over ranks of the per-pass kernel time is what a real N-GPU
run would spend in kernels (exchange and host turn-around not included).  python predict_ranks.py c4|c5 points N iters [sweep]"""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "icp-symm_amd", "py"))
import numpy as np, symmicp as sym
from symmicp import synth
kind, n, world, iters = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
d = {"c4": synth.c4_surface, "c5": synth.c5_scan}[kind](n)
if len(sys.argv) > 5 and sys.argv[5] == "sweep":          # source rows in a spatially coherent order: contiguous row ranges are compact shares
    o = synth.sweep_order(d["src"]); d["src"], d["src_n"] = d["src"][o], d["src_n"][o]
engs = []
for r in range(world):
    e = sym.Engine(mode=sym.MODE_PAPER, corr=sym.CORR_TREE, max_iters=iters, fixed_iters=1)
    if world > 1:
        e.comm_init_rank(world, r, None)
    e.set_target(d["tgt"], d["tgt_n"]); e.set_source(d["src"], d["src_n"]); e.enable_timing(1)
    engs.append(e)
for rep in range(2):                      # the first repetition warms the clocks up; the second is reported
    per_pass = []
    its = [e.begin() for e in engs]
    for k in range(iters + 1):
        per_pass.append(max(e.stats()["last_pass_ms"] for e in engs))
        if k == iters:
            break
        total = np.sum([np.asarray(it["sums"], np.float64) for it in its], axis=0)
        if world > 1:
            for e in engs:
                e.set_sums(total)
        its = [e.step() for e in engs]
print("%s %d points, %d ranks: kernel ms per pass (max over ranks): first %.3f, second %.3f, third %.3f, last %.4f; sum over %d passes %.3f ms"
      % (kind, n, world, per_pass[0], per_pass[1], per_pass[2], per_pass[-1], len(per_pass), sum(per_pass)))
for e in engs:
    e.close()

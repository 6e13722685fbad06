"""All 1M pairs of a converged C4 alignment against the oracle's exact NN (takes ~1 min of host time)."""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "icp-symm_amd", "py")); sys.path.insert(0, ROOT)
import numpy as np, symmicp as sym
from symmicp import synth
from oracle import oracle
d = synth.c4_surface(1000000)
for iters in (8, 30, 100):
    with sym.Engine(mode=sym.MODE_PAPER, corr=sym.CORR_TREE, max_iters=iters, fixed_iters=1) as e:
        e.set_target(d["tgt"], d["tgt_n"]); e.set_source(d["src"], d["src_n"])
        r = e.align()
        idx, d2 = e.correspondences()
    t0 = time.time()
    ri, rd = oracle.nn_grid(d["src"], d["tgt"], X=r["transform"])
    print(iters, "iters: mismatching pairs", int((idx != ri).sum()), "distance mismatches", int((d2 != rd).sum()), "(oracle %.0f s)" % (time.time() - t0), flush=True)

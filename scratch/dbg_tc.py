import os, sys
sys.path.insert(0, "/root/repo/icp-symm_amd/py")
import numpy as np, symmicp
from symmicp import synth
d = synth.c4_surface(1000000)
with symmicp.Engine(mode=symmicp.MODE_PAPER, corr=symmicp.CORR_TREE, max_iters=14, fixed_iters=1) as e:
    e.set_target(d["tgt"], d["tgt_n"]); e.set_source(d["src"], d["src_n"])
    e.align()

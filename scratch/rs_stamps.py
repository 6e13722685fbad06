"""k_reduce_solve phase stamps: build a library with -DRS_STAMPS, run with SYMMICP_LIB=<it> SYMMICP_DEBUG_HOST=1 python scratch/rs_stamps.py [points]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "icp-symm_amd", "py"))
import symmicp as sym
from symmicp import synth
d = synth.c4_surface(int(sys.argv[1]) if len(sys.argv) > 1 else 1000000)
with sym.Engine(mode=sym.MODE_PAPER, corr=sym.CORR_TREE, max_iters=40, fixed_iters=1) as e:
    e.set_target(d["tgt"], d["tgt_n"]); e.set_source(d["src"], d["src_n"])
    e.align(); e.align()

import os, sys, time
os.environ["SYMMICP_DEBUG_COUNTERS"] = "1"
os.environ["SYMMICP_EXP_KEEP_PREV"] = "1"
os.environ["SYMMICP_GRID_LEVEL"] = "0"
os.environ["SYMMICP_WAVE_MODE_MAX"] = "0"
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "icp-symm_amd", "py"))
import numpy as np, symmicp
from symmicp import synth
d = synth.c4_surface(1000000)
eng = symmicp.Engine(mode=symmicp.MODE_PAPER, corr=symmicp.CORR_TREE, max_iters=1, fixed_iters=1)
eng.set_target(d["tgt"], d["tgt_n"]); eng.set_source(d["src"], d["src_n"])
eng.enable_timing(2)
eng.begin()
eng.reset_stats()
eng.begin()      # bound from the previous begin's pairs (exact, or the wave's middle query's, depending on the library)
st = eng.stats()
print("walk ms", st["kernel_ms"][2], "cells ms", st["kernel_ms"][0])
eng.close()

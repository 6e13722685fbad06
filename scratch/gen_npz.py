import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "icp-symm_amd", "py"))
import numpy as np
from symmicp import synth
kind, n, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]
d = {"c4": synth.c4_surface, "c5": synth.c5_scan, "c3": synth.c3_uniform}[kind](n)
np.savez(out, **d)

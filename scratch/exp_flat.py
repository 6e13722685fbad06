"""How many node visits does the first-pass walk need when the target's boxes are tight?  A flat plane (axis-aligned:
thin boxes) against the same plane tilted by 35 degrees (fat boxes), query cloud 18 spacings above it."""
import os, sys
os.environ["SYMMICP_DEBUG_COUNTERS"] = "1"
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "icp-symm_amd", "py"))
import numpy as np, symmicp
from symmicp import synth
n = 1000000
rng = np.random.default_rng(0)
for tilt in (0.0, 35.0):
    t = np.stack([rng.random(n), rng.random(n), np.zeros(n)], 1)
    s = np.stack([rng.random(n), rng.random(n), np.full(n, 0.018)], 1)
    R = synth.rotation(tilt, (1.0, 0.3, 0.0))
    t = (t - 0.5) @ R.T + 0.5; s = (s - 0.5) @ R.T + 0.5
    nr = np.tile(R @ np.array([0, 0, 1.0]), (n, 1))
    eng = symmicp.Engine(mode=symmicp.MODE_PAPER, corr=symmicp.CORR_TREE, max_iters=1, fixed_iters=1)
    eng.set_target(t.astype(np.float32), nr.astype(np.float32)); eng.set_source(s.astype(np.float32), nr.astype(np.float32))
    eng.enable_timing(2)
    print("tilt", tilt, flush=True)
    eng.begin()
    st = eng.stats(); print("walk ms", st["kernel_ms"][2], flush=True)
    eng.close()

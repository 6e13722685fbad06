"""First-pass kernel time over environment switches: python scratch/pkt_sweep.py c4 1000000 "K=V,K=V" "K=V" ...   (each argument one variant)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "icp-symm_amd", "py"))
import numpy as np, symmicp as sym
from symmicp import synth
wl, n = sys.argv[1], int(sys.argv[2])
d = dict(c3=synth.c3_uniform, c4=synth.c4_surface, c5=synth.c5_scan)[wl](n, **(dict(workers=16) if wl == "c5" and n >= 1000000 else {}))
base = dict(os.environ)
for spec in sys.argv[3:]:
    os.environ.clear(); os.environ.update(base)
    for kv in filter(None, spec.split(",")):
        k, v = kv.split("="); os.environ["SYMMICP_" + k] = v
    with sym.Engine(mode=sym.MODE_PAPER, corr=sym.CORR_TREE, max_iters=30, fixed_iters=1) as e:
        e.set_target(d["tgt"], d["tgt_n"]); e.set_source(d["src"], d["src_n"])
        e.begin(); e.enable_timing(2)
        best = 1e9
        for _ in range(5):
            e.reset_stats(); e.begin(); best = min(best, e.stats()["kernel_ms"][2])
    print("%s %d [%s]: first pass %.3f ms" % (wl, n, spec, best), flush=True)

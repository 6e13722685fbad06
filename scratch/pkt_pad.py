"""First pass of the whole cloud at different occupancies (SYMMICP_PACKET_LDS_PAD limits the workgroups per CU): python scratch/pkt_pad.py c4 1000000 pad ..."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "icp-symm_amd", "py"))
import numpy as np, symmicp as sym
from symmicp import synth
wl, n = sys.argv[1], int(sys.argv[2])
d = dict(c3=synth.c3_uniform, c4=synth.c4_surface, c5=synth.c5_scan)[wl](n)
for spec in sys.argv[3:]:
    w, pad = spec.split(":")
    os.environ["SYMMICP_PACKET_WAVES"] = w; os.environ["SYMMICP_PACKET_LDS_PAD"] = pad
    with sym.Engine(mode=sym.MODE_PAPER, corr=sym.CORR_TREE, max_iters=30, fixed_iters=1) as e:
        e.set_target(d["tgt"], d["tgt_n"]); e.set_source(d["src"], d["src_n"])
        e.begin(); e.enable_timing(2)
        best = 1e9
        for _ in range(5):
            e.reset_stats(); e.begin(); best = min(best, e.stats()["kernel_ms"][2])
    print("%s %d W=%s pad %s: %.3f ms" % (wl, n, w, pad, best), flush=True)

"""First-pass A/B over waves per packet: python scratch/pkt_ab.py [workload] [points] [ranks]
For each W in SYMMICP_PACKET_WAVES = 1, 2, 4 (the switch is read at symmicp_create): kernel time of the first pass
(timing mode 2, best of 5) for the whole cloud and, with ranks > 1, for the slowest share of a sweep-ordered `ranks`-way split;
pairs are checked against the W = 1 result (bit-exact)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "icp-symm_amd", "py"))
import numpy as np, symmicp as sym
from symmicp import synth
wl = sys.argv[1] if len(sys.argv) > 1 else "c4"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
ranks = int(sys.argv[3]) if len(sys.argv) > 3 else 8
d = dict(c3=synth.c3_uniform, c4=synth.c4_surface, c5=synth.c5_scan)[wl](n)
o = synth.sweep_order(d["src"]) if wl != "c5" else np.arange(n)
src_s, srcn_s = np.ascontiguousarray(d["src"][o]), np.ascontiguousarray(d["src_n"][o])
names = sym.KERNEL_SLOTS

def first_pass(world, rank, src, src_n):
    e = sym.Engine(mode=sym.MODE_PAPER, corr=sym.CORR_TREE, max_iters=30, fixed_iters=1)
    if world > 1:
        e.comm_init_rank(world, rank, None)
    e.set_target(d["tgt"], d["tgt_n"]); e.set_source(src, src_n)
    e.begin(); e.enable_timing(2)
    best = 1e9
    for _ in range(5):
        e.reset_stats(); e.begin(); st = e.stats()
        best = min(best, st["kernel_ms"][2])
    idx, d2 = e.correspondences()
    fb = e.stats()["packet_fallbacks"]
    e.close()
    return best, idx, d2, fb

ref = None
for w in (os.environ.get("PKT_AB_WAVES", "1,2,4,0")).split(","):
    os.environ["SYMMICP_PACKET_WAVES"] = w
    t, idx, d2, fb = first_pass(1, 0, d["src"], d["src_n"])
    if ref is None:
        ref = (idx, d2)
    same = bool(np.array_equal(idx, ref[0]) and np.array_equal(d2, ref[1]))
    line = "%s %d W=%s: whole cloud %.3f ms (pairs == first variant: %s, fallbacks %d)" % (wl, n, w, t, same, fb)
    if ranks > 1:
        ts = [first_pass(ranks, r, src_s, srcn_s)[0] for r in range(ranks)]
        line += "; %d-way shares: max %.3f ms, mean %.3f ms" % (ranks, max(ts), sum(ts) / len(ts))
    print(line, flush=True)

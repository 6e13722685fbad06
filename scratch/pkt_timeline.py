"""Timeline of the first pass of the whole cloud (production kernel, timing-only trace): python scratch/pkt_timeline.py c4 1000000 waves ..."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "icp-symm_amd", "py"))
import numpy as np, symmicp as sym
from symmicp import synth
wl, n = sys.argv[1], int(sys.argv[2])
d = dict(c4=synth.c4_surface, c5=synth.c5_scan)[wl](n)
tf = "/tmp/pkt_timeline.bin"
os.environ["SYMMICP_DEBUG_TRACE"] = tf
for w in sys.argv[3:]:
    os.environ["SYMMICP_PACKET_WAVES"] = w
    with sym.Engine(mode=sym.MODE_PAPER, corr=sym.CORR_TREE, max_iters=2, fixed_iters=1) as e:
        e.set_target(d["tgt"], d["tgt_n"]); e.set_source(d["src"], d["src_n"])
        e.begin(); e.enable_timing(2); e.reset_stats(); e.begin()
        kms = e.stats()["kernel_ms"][2]
    t = np.fromfile(tf, dtype=np.uint64).reshape(-1, 2)
    t = t[t[:, 0] != 0]
    start = t[:, 0].astype(np.float64) * 0.01; start -= start.min()
    dur = (t[:, 1] >> np.uint64(32)).astype(np.float64) * 0.01
    end = start + dur
    grid = np.linspace(0, end.max(), 21)
    running = [int(((start <= g) & (end > g)).sum()) for g in grid]
    print("W=%s: kernel %.0f us, %d packets, dur sum %.1f ms (%.0f us x 4096 slots), median %.0f, 90%% %.0f, 99%% %.0f, max %.0f us; last start %.0f us, last end %.0f us" %
          (w, kms * 1e3, len(dur), dur.sum() / 1e3, dur.sum() / 4096, np.median(dur), np.quantile(dur, .9), np.quantile(dur, .99), dur.max(), start.max(), end.max()))
    print("   packets running at 0, 5, ... 100 %% of the launch: " + " ".join(str(r) for r in running))
    late = np.argsort(-end)[:6]
    print("   last finishers: " + "; ".join("start %.0f dur %.0f" % (start[k], dur[k]) for k in late), flush=True)

"""Per-packet trace of the first pass (SYMMICP_DEBUG_COUNTERS=1 SYMMICP_DEBUG_TRACE=file): python scratch/pkt_trace.py file"""
import sys, numpy as np
t = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 2)
t = t[t[:, 0] != 0]
start = t[:, 0].astype(np.int64) * 10; dur = (t[:, 1] >> np.uint64(32)).astype(np.int64) * 10      # ns
pops = ((t[:, 1] >> np.uint64(8)) & np.uint64(0xFFFFFF)).astype(np.int64); gave = (t[:, 1] & np.uint64(1)).astype(bool); src = ((t[:, 1] >> np.uint64(1)) & np.uint64(3)).astype(int)
t0 = start.min()
for s in (0, 2):
    m = src == s
    if not m.any(): continue
    st, du, po = start[m] - t0, dur[m], pops[m]
    print("stage %d: %d packets; start %.0f..%.0f us, end max %.0f us; dur mean %.0f k max %.0f k; pops mean %.0f max %d; gave up %d" % (s, m.sum(), st.min() / 1e3, st.max() / 1e3, (st + du).max() / 1e3, du.mean() / 1e3, du.max() / 1e3, po.mean(), po.max(), gave[m].sum()))
    print("   ns per pop: mean %.0f; by start decile: %s" % ((du / np.maximum(po, 1)).mean(), [int((du / np.maximum(po, 1))[(st >= a) & (st <= b)].mean()) for a, b in zip(np.quantile(st, np.arange(0, 1, .1)), np.quantile(st, np.arange(.1, 1.01, .1)))]))
    end = st + du
    order = np.argsort(-end)[:5]
    print("   last finishers: " + "; ".join("start %.0fk dur %.0fk pops %d" % (st[k] / 1e3, du[k] / 1e3, po[k]) for k in order))
    # concurrency over time
    grid = np.linspace(0, end.max(), 11)
    print("   packets running at t: " + " ".join("%d" % ((st <= g) & (end > g)).sum() for g in grid))
# per-XCD share (xcd_remap: XCD x works on the x-th eighth of the packets, in order)
t_all = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 2)
n = int(np.nonzero(t_all[:, 0])[0].max()) + 1
npad = (n + 7) // 8 * 8
st_all = t_all[:n, 0].astype(np.int64) * 10; du_all = (t_all[:n, 1] >> np.uint64(32)).astype(np.int64) * 10
print("per eighth of the packets: work (sum of durations, ms) and last end (us)")
for x in range(8):
    lo, hi = x * npad // 8, min(n, (x + 1) * npad // 8)
    m = st_all[lo:hi] != 0
    print("   eighth %d: %6.1f ms  end %7.0f us   mean dur %5.0f us" % (x, du_all[lo:hi][m].sum() / 1e6, ((st_all[lo:hi] + du_all[lo:hi])[m].max() - t0) / 1e3, du_all[lo:hi][m].mean() / 1e3))
# the slowest packets: did they overflow into the depth-first fallback?
ga = (t_all[:n, 1] & np.uint64(1)).astype(bool); po = ((t_all[:n, 1] >> np.uint64(8)) & np.uint64(0xFFFFFF)).astype(np.int64)
order = np.argsort(-du_all)
print("slowest 12: " + "; ".join("%.0fus%s steps %d" % (du_all[k] / 1e3, "*" if ga[k] else "", po[k]) for k in order[:12]) + "   (*: depth-first fallback)")
print("fallback packets: %d, duration mean %.0f us, sum %.1f ms of %.1f ms; others: mean %.0f us, 99%% %.0f us, max %.0f us" % (ga.sum(), du_all[ga].mean() / 1e3, du_all[ga].sum() / 1e6, du_all.sum() / 1e6, du_all[~ga].mean() / 1e3, np.quantile(du_all[~ga], .99) / 1e3, du_all[~ga].max() / 1e3))
print("duration quantiles (us): " + " ".join("%g:%.0f" % (q, np.quantile(du_all, q) / 1e3) for q in (.5, .9, .99, .999, 1)))

"""CPU simulation of a wave-packet octree traversal (design study for k_search_packet).

Counts, per packet of 64 Morton-consecutive queries: internal-node visits, leaf visits, rejected pops,
points tested -- for different leaf sizes -- and compares with the per-query walk's node visits.
Not a test; numbers go to DESIGN.md.
"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "icp-symm_amd", "py"))
from symmicp import synth

BITS = 10


def spread3(v):
    v = v.astype(np.uint64) & 0x3ff
    v = (v | (v << 16)) & 0x030000ff
    v = (v | (v << 8)) & 0x0300f00f
    v = (v | (v << 4)) & 0x030c30c3
    v = (v | (v << 2)) & 0x09249249
    return v


def morton(p, lo, h0):
    c = np.floor((p - lo) / h0).astype(np.int64)
    c = np.clip(c, 0, (1 << BITS) - 1)
    return (spread3(c[:, 2]) << 2) | (spread3(c[:, 1]) << 1) | spread3(c[:, 0])


class Octree:
    def __init__(self, pts, leafmax=8):
        lo = pts.min(0); hi = pts.max(0)
        emax = float((hi - lo).max())
        h0 = emax * 1.00001 / (1 << BITS)
        keys = morton(pts, lo, h0)
        order = np.argsort(keys, kind="stable")
        self.pts = pts[order]
        self.row = order
        keys = keys[order]
        n = len(pts)
        self.levels = []
        for l in range(BITS + 1):
            pre = keys >> np.uint64(3 * (BITS - l))
            start = np.flatnonzero(np.r_[True, pre[1:] != pre[:-1]])
            cnt = np.diff(np.r_[start, n])
            self.levels.append(dict(pre=pre[start], first=start, cnt=cnt))
        # boxes
        for l in range(BITS + 1):
            L = self.levels[l]
            L["lo"] = np.minimum.reduceat(self.pts, L["first"], axis=0)
            L["hi"] = np.maximum.reduceat(self.pts, L["first"], axis=0)
        # children
        for l in range(BITS):
            L, C = self.levels[l], self.levels[l + 1]
            par = C["pre"] >> np.uint64(3)
            cstart = np.searchsorted(par, L["pre"], side="left")
            cend = np.searchsorted(par, L["pre"], side="right")
            L["cf"] = cstart; L["nc"] = cend - cstart
            L["leaf"] = L["cnt"] <= leafmax
        self.levels[BITS]["leaf"] = np.ones(len(self.levels[BITS]["first"]), bool)
        self.levels[BITS]["cf"] = np.zeros(len(self.levels[BITS]["first"]), int)
        self.levels[BITS]["nc"] = np.zeros(len(self.levels[BITS]["first"]), int)


def boxdist2(q, lo, hi):
    d = np.maximum(np.maximum(lo - q, q - hi), 0.0)
    return (d * d).sum(-1)


def packet_walk(T, Q, per_lane_pop=True, per_lane_leaf=False, order="sorted"):
    """returns dict of counters and the per-lane best d2"""
    nq = len(Q)
    best = np.full(nq, np.inf)
    qlo, qhi = Q.min(0), Q.max(0)
    stack = [(0, 0, 0.0)]     # (level, idx, boxbox d2)
    st = dict(internal=0, leaf=0, pop_rej=0, points=0, push=0, pop_rej_bb=0)
    while stack:
        l, i, bb = stack.pop()
        r2 = best.max()
        if bb > r2:
            st["pop_rej_bb"] += 1
            continue
        L = T.levels[l]
        if per_lane_pop:
            d = boxdist2(Q, L["lo"][i], L["hi"][i])
            if not (d <= best).any():
                st["pop_rej"] += 1
                continue
        if L["leaf"][i]:
            st["leaf"] += 1
            f, c = L["first"][i], L["cnt"][i]
            P = T.pts[f:f + c]
            st["points"] += c
            d2 = ((Q[:, None, :] - P[None, :, :]) ** 2).sum(-1)
            best = np.minimum(best, d2.min(1))
            continue
        st["internal"] += 1
        C = T.levels[l + 1]
        cf, nc = L["cf"][i], L["nc"][i]
        clo, chi = C["lo"][cf:cf + nc], C["hi"][cf:cf + nc]
        # box-box distance to the packet's bounding box
        g = np.maximum(np.maximum(clo - qhi, qlo - chi), 0.0)
        bbd = (g * g).sum(-1)
        r2 = best.max()
        keep = np.flatnonzero(bbd <= r2)
        # push far-to-near so that the nearest pops first; order by distance to the packet centre box
        if order == "sorted":
            seq = keep[np.argsort(-bbd[keep], kind="stable")]
        elif order == "top" and len(keep):
            j = keep[np.argmin(bbd[keep])]
            seq = np.r_[keep[keep != j], j]
        else:
            seq = keep
        for k in seq:
            stack.append((l + 1, cf + k, bbd[k]))
            st["push"] += 1
    return st, best


def single_walk(T, q):
    """per-query near-first walk: node visits (the shipped k_search_walk counts these)"""
    best = np.inf
    visits = 0
    stack = [(0, 0)]
    while stack:
        l, i = stack.pop()
        L = T.levels[l]
        if boxdist2(q, L["lo"][i], L["hi"][i]) > best:
            continue
        visits += 1
        if L["leaf"][i]:
            f, c = L["first"][i], L["cnt"][i]
            d2 = ((T.pts[f:f + c] - q) ** 2).sum(-1).min()
            best = min(best, d2)
            continue
        C = T.levels[l + 1]
        cf, nc = L["cf"][i], L["nc"][i]
        d = boxdist2(q, C["lo"][cf:cf + nc], C["hi"][cf:cf + nc])
        for k in np.argsort(-d, kind="stable"):
            if d[k] <= best:
                stack.append((l + 1, cf + k))
    return visits, best


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    wl = sys.argv[2] if len(sys.argv) > 2 else "c4"
    npk = int(sys.argv[3]) if len(sys.argv) > 3 else 200
    d = dict(c3=synth.c3_uniform, c4=synth.c4_surface, c5=synth.c5_scan)[wl](n)
    src = d["src"].astype(np.float64); tgt = d["tgt"].astype(np.float64)
    # source in its own Morton order
    lo = src.min(0); h0 = float((src.max(0) - lo).max()) * 1.00001 / 1024
    so = np.argsort(morton(src, lo, h0), kind="stable")
    src = src[so]
    rng = np.random.default_rng(1)
    for leafmax in (8, 16, 32):
        T = Octree(tgt, leafmax)
        nn = [len(L["first"]) for L in T.levels]
        tot = {}
        sv = []
        for pk in rng.integers(0, len(src) // 64, npk):
            Q = src[pk * 64:(pk + 1) * 64]
            for name, kw in (("lane", dict(per_lane_pop=True)), ("l-top", dict(order="top")), ("l-none", dict(order="none"))):
                st, best = packet_walk(T, Q, **kw)
                t = tot.setdefault(name, {})
                for k, v in st.items():
                    t[k] = t.get(k, 0) + v
            if leafmax == 8 and len(sv) < 40 * 8:
                for q in Q[::8]:
                    v, b = single_walk(T, q)
                    sv.append(v)
        print("leafmax %d: nodes/level %s" % (leafmax, nn))
        for name, t in tot.items():
            print("  %-5s per packet: " % name + "  ".join("%s=%.1f" % (k, v / npk) for k, v in t.items()))
        if sv:
            print("  single-query walk: mean visits %.1f max %d" % (np.mean(sv), max(sv)))


if __name__ == "__main__":
    main()

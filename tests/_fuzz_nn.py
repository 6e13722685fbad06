"""Randomised exactness sweep: GPU tree search vs the oracle's exact NN, every pass of short alignments, over random
sizes / cloud kinds / motions / apply modes / estimators.  Used by tests/test_gpu_parity.py (fixed number of cases) and
from the command line for longer runs:  python tests/_fuzz_nn.py [seconds] [seed]"""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
for p in (os.path.join(ROOT, "icp-symm_amd", "py"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)
import numpy as np


def run(budget_s=None, max_cases=None, seed=0, verbose=True):
    import symmicp as sym
    from symmicp import synth
    from oracle import oracle
    rng = np.random.default_rng(seed)

    def cloud(kind, n):
        if kind == 0:                                   # uniform cube
            p = rng.random((n, 3))
        elif kind == 1:                                 # surface
            u, v = rng.random(n), rng.random(n)
            p = np.stack([u, v, 0.1 * np.sin(4 * np.pi * u) * np.cos(6 * np.pi * v)], 1)
        elif kind == 2:                                 # clusters with duplicates
            c = rng.random((max(1, n // 50), 3))
            p = c[rng.integers(0, len(c), n)] + 0.002 * rng.standard_normal((n, 3)) * (rng.random((n, 1)) > 0.3)
        elif kind == 3:                                 # quantised coordinates: many exact ties
            p = np.round(rng.random((n, 3)) * 20) / 20
        else:                                           # a line (degenerate extent in two axes)
            t = rng.random(n)
            p = np.stack([t, 0.5 + 1e-4 * rng.standard_normal(n), np.full(n, 0.25)], 1)
        nr = rng.standard_normal((n, 3)); nr /= np.linalg.norm(nr, axis=1, keepdims=True) + 1e-12
        return p.astype(np.float32), nr.astype(np.float32)

    t_end = time.time() + budget_s if budget_s else None
    cases = 0
    failures = []
    while (t_end is None or time.time() < t_end) and (max_cases is None or cases < max_cases):
        kind_s, kind_t = int(rng.integers(0, 5)), int(rng.integers(0, 5))
        n_s = int(rng.choice([1, 2, 7, 63, 64, 65, 255, 257, 1000, 4097, 20000, 60000]))
        n_t = int(rng.choice([1, 3, 8, 9, 64, 500, 4096, 30000, 100000]))
        src, sn = cloud(kind_s, n_s)
        tgt, tn = cloud(kind_t, n_t)
        if rng.random() < 0.5:                          # target = moved copy of the source (overlap, real ICP behaviour)
            R = synth.rotation(float(rng.uniform(0, 20)), rng.standard_normal(3))
            tgt = (src.astype(np.float64) @ R.T + rng.uniform(-0.05, 0.05, 3)).astype(np.float32); tn = (sn @ R.T).astype(np.float32)
        apply_mode = int(rng.choice([sym.APPLY_INCREMENTAL, sym.APPLY_CUMULATIVE]))
        mode = int(rng.choice([sym.MODE_PAPER, sym.MODE_QUIRKS, sym.MODE_P2P]))
        case = dict(kind_s=kind_s, kind_t=kind_t, n_s=n_s, n_t=tgt.shape[0], apply=apply_mode, mode=mode)
        cases += 1
        with sym.Engine(mode=mode, corr=sym.CORR_TREE, apply=apply_mode, max_iters=6, fixed_iters=1) as e:
            e.set_target(tgt, tn); e.set_source(src, sn)
            last = e.begin()
            for it in range(6):
                idx, d2 = e.correspondences()
                if apply_mode == sym.APPLY_INCREMENTAL:
                    p, pn = e.source(); ri, rd = oracle.nn_grid(p, tgt)
                    # the pass's 40-double record against a host recomputation from the returned pairs
                    S = oracle.reduce40(p, pn, tgt, tn, idx=idx, pivot=None if mode == sym.MODE_QUIRKS else e.pivot(),
                                        p2p=(mode == sym.MODE_P2P))
                    g = np.asarray(last["sums"], np.float64)
                    if np.abs(g - S).max() > 1e-9 * max(1.0, np.abs(S).max()):
                        failures.append(dict(case, it=it, sums_err=float(np.abs(g - S).max() / max(1.0, np.abs(S).max()))))
                        if verbose:
                            print("SUMS", failures[-1], flush=True)
                        break
                else:
                    ri, rd = oracle.nn_grid(src, tgt, X=e.transform())
                if not (np.array_equal(idx, ri) and np.array_equal(d2, rd)):
                    failures.append(dict(case, it=it, nbad=int((idx != ri).sum())))
                    if verbose:
                        print("MISMATCH", failures[-1], flush=True)
                    break
                last = e.step(check=False)
                if last["status"] != 0:
                    break                               # degenerate system: legitimately flagged
    return cases, failures


if __name__ == "__main__":
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    n, bad = run(budget_s=budget, seed=int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    print("cases %d, mismatches %d" % (n, len(bad)))
    sys.exit(1 if bad else 0)

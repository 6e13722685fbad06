"""Wall time of the normals pre-step (symmicp_estimate_normals: upload + index build + k_normals_knn + read-back) at 100k / 1M / 8M points,
with the agreement against the generator's analytic normals: python scratch/time_normals.py   (under rocprofv3 --kernel-trace --stats: the kernel's own time)"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "icp-symm_amd", "py"))
import numpy as np, symmicp as sym
from symmicp import synth
for n in (100_000, 1_000_000, 8_000_000):
    d = synth.c4_surface(n)
    sym.estimate_normals(d["src"][:1000], 10)
    best = 1e9
    for _ in range(3):
        t = time.perf_counter(); nrm, curv = sym.estimate_normals(d["src"], 10, viewpoint=(0.5, 0.5, 5.0)); best = min(best, time.perf_counter() - t)
    dots = np.abs(np.einsum('ij,ij->i', nrm, d["src_n"]))
    print('%d points: estimate_normals %.1f ms wall (%.2f Mpoints/s); median |dot| vs analytic normals %.6f, 1st percentile %.4f' % (n, best * 1e3, n / best / 1e6, np.median(dots), np.percentile(dots, 1)), flush=True)

import sys, time, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/icp-symm_amd/py')
import symmicp as sym
from symmicp import synth
for n in (100_000, 1_000_000):
    d = synth.c4_surface(n)
    sym.estimate_normals(d["src"][:1000], 10)
    t=time.time(); nrm,curv = sym.estimate_normals(d["src"], 10, viewpoint=(0.5,0.5,5.0)); dt=time.time()-t
    dots = np.abs(np.einsum('ij,ij->i', nrm, d["src_n"]))
    print(n, 'normals %.1f ms'%(dt*1e3), 'median |dot| vs analytic', np.median(dots), 'p1', np.percentile(dots,1))

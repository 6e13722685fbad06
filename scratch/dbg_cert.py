import sys, os, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/icp-symm_amd/py')
import symmicp as sym
from oracle import oracle as O
G='/root/repo/tests/golden/'
src,_=O.pcd_read(G+'cat.pcd'); tgt,_=O.pcd_read(G+'cat_out.pcd'); g=np.load(G+'cat_golden.npz'); sn,tn=g['src_n'],g['tgt_n']
with sym.Engine(mode=sym.MODE_QUIRKS, corr=sym.CORR_TREE, apply=sym.APPLY_INCREMENTAL) as e:
    e.set_target(tgt,tn); e.set_source(src,sn)
    it=e.begin()
    for k in range(10):
        it=e.step()
        idx,d2=e.correspondences(); p,_=e.source()
        ri,rd=O.nn_brute(p,tgt)
        bad=np.nonzero(idx!=ri)[0]
        print(k, 'diff',it['diff'],'bad',bad.size, (idx[bad[:3]],ri[bad[:3]],d2[bad[:3]],rd[bad[:3]]) if bad.size else '')

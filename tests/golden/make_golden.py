"""Regenerates tests/golden/cat_golden.npz from the CPU oracle (oracle/symmicp_oracle.c).

The reference ships no expected outputs, and it cannot be built here
(PCL/Eigen absent), so these vectors come from this repo's own restatement; what pins the
restatement itself is the data-level known answer checked in tests/test_oracle_pins.py
(cat_out == Rz(45 deg) * cat + (2.5,0,0)).  Inputs are the reference's data files copied
verbatim next to this script (cat.pcd, cat_out.pcd, txt2pcd_bunny1.pcd, za.txt).

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import oracle as O  # noqa: E402


def main():
    src, _ = O.pcd_read(os.path.join(HERE, "cat.pcd"))
    tgt, _ = O.pcd_read(os.path.join(HERE, "cat_out.pcd"))
    src_n, src_c = O.normals_knn(src, 10)
    tgt_n, tgt_c = O.normals_knn(tgt, 10)
    out = dict(src_n=src_n, tgt_n=tgt_n, src_curv=src_c, tgt_curv=tgt_c)
    out["sums0_quirks"] = O.reduce40(src, src_n, tgt, tgt_n)
    for name, kw in dict(
        quirks_identity=dict(mode=O.MODE_QUIRKS, corr=O.CORR_IDENTITY),
        quirks_identity_literal=dict(mode=O.MODE_QUIRKS, corr=O.CORR_IDENTITY, solve=O.SOLVE_LITERAL),
        paper_identity=dict(mode=O.MODE_PAPER, corr=O.CORR_IDENTITY),
        paper_nn=dict(mode=O.MODE_PAPER, corr=O.CORR_BRUTE, max_iters=30),
        quirks_nn=dict(mode=O.MODE_QUIRKS, corr=O.CORR_BRUTE, max_iters=10),
    ).items():
        r = O.align(src, src_n, tgt, tgt_n, **kw)
        out[name + "_T"] = r["transform"]
        out[name + "_diffs"] = np.concatenate([r["diffs"], [r["diff_final"]]]).astype(np.float32)
        out[name + "_iters"] = np.int32(r["iters"])
        out[name + "_status"] = np.int32(r["status"])
    idx, d2 = O.nn_brute(src, tgt)
    out["nn0_idx"] = idx
    out["nn0_d2"] = d2
    np.savez_compressed(os.path.join(HERE, "cat_golden.npz"), **out)
    print("wrote cat_golden.npz:", {k: np.asarray(v).shape for k, v in out.items()})


if __name__ == "__main__":
    main()

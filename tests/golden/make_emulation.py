"""An independent second restatement of the reference's loop, in plain numpy: tests/golden/emulation_cat.npz.

Why: the reference ships no expected outputs and cannot be built here, so the C oracle (oracle/symmicp_oracle.c) is
pinned by data-level facts only.  Oracle and HIP path share one reading of the Eigen semantics (post-multiplying
translate / rotate, AngleAxis, the a-then-t alternation); a misreading common to both would pass every test.  This
script restates the same reference lines a second time with different tools, so that the two restatements check each
other (it cannot make parity "green" -- the reference still holds no outputs -- but it removes the single-author
common mode):

  ICP/func.cpp:43-60    calculateMatrixNotation   numpy cross / dot on whole arrays
  ICP/func.cpp:64-73    solveLLS                  np.linalg.svd (LAPACK) on the N x 3 matrices, x = V S^-1 U^T b
  ICP/func.cpp:76-102   estimateTransformSymm     the five factors as explicit 4x4 matrices, multiplied left to right
                                                  (Eigen's translate()/rotate() post-multiply: T <- T * Translation(v))
  ICP/func.cpp:104-121  applyTransform            homogeneous n x 4 times the 4x4, transposed
  ICP/myicp.cpp:117-142 the loop                  diff > 1 && iters++ < 10, normals get the full affine, incre * transform

float32 storage as in the reference (Eigen::MatrixXf), float64 only inside LAPACK's own accumulations.
Also: k = 10 PCA normals with scipy's k-d tree + numpy eigh (ICP/myicp.cpp:152-172), as a second opinion on the
oracle's normals (PCL's own rounding stays unpinned).

Inputs: the reference's data files next to this script (cat.pcd, cat_out.pcd).  The loop runs on the normals stored in
cat_golden.npz (normals are an INPUT of the hot path), so a difference isolates the loop arithmetic.

    python tests/golden/make_emulation.py
"""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
f32 = np.float32


def read_pcd_ascii(path):
    """x y z columns of an ASCII PCD v0.7 file (ICP/myicp.cpp:20-31 keeps x, y, z by field name)"""
    with open(path) as f:
        lines = f.read().split("\n")
    fields, k = None, 0
    for k, ln in enumerate(lines):
        tok = ln.split()
        if tok and tok[0] == "FIELDS":
            fields = tok[1:]
        if tok and tok[0] == "DATA":
            assert tok[1] == "ascii"
            break
    cols = [fields.index(c) for c in ("x", "y", "z")]
    rows = [ln.split() for ln in lines[k + 1:] if ln.strip()]
    return np.array([[float(r[c]) for c in cols] for r in rows], dtype=f32)


def solve_lls(A, b):
    """func.cpp:64-73"""
    U, S, Vt = np.linalg.svd(A.astype(f32), full_matrices=False)       # thin U, full V (n = 3)
    return (Vt.T @ (np.diag(f32(1) / S) @ (U.T @ b))).astype(f32)


def translation(v):
    T = np.eye(4, dtype=f32)
    T[:3, 3] = v
    return T


def angle_axis(theta, axis):
    """Rodrigues: the rotation Eigen::AngleAxisf(theta, axis) stands for"""
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]], dtype=f32)
    R = np.eye(3, dtype=f32) * f32(np.cos(theta)) + f32(np.sin(theta)) * K + f32(1 - np.cos(theta)) * np.outer(axis, axis).astype(f32)
    M = np.eye(4, dtype=f32)
    M[:3, :3] = R
    return M


def estimate_transform_symm(p, n_p, q, n_q):
    """func.cpp:76-102; returns (4x4, a~, t~)"""
    n = n_p + n_q                                                      # :51
    M = np.cross(p + q, n).astype(f32)                                 # :54
    N = n.astype(f32)                                                  # :56
    c = np.einsum("ij,ij->i", p - q, n).astype(f32)                    # :58
    src_mean, tgt_mean = p.mean(0, dtype=f32), q.mean(0, dtype=f32)    # :85
    t_ = (tgt_mean - src_mean).astype(f32)                             # :86
    a_ = solve_lls(M, -(N @ t_ + c))                                   # :87
    t_ = solve_lls(N, -(M @ a_ + c))                                   # :88
    theta = f32(np.arctan(np.linalg.norm(a_)))                         # :93
    axis = (a_ / np.linalg.norm(a_)).astype(f32)
    X = np.eye(4, dtype=f32)
    for F in (translation(-src_mean), angle_axis(theta, axis), translation(t_ * f32(np.cos(theta))), angle_axis(theta, axis), translation(tgt_mean)):
        X = (X @ F).astype(f32)                                        # :95-99, each call post-multiplies
    return X, a_, t_


def apply_transform(X, pts):
    """func.cpp:104-121"""
    med = np.concatenate([pts, np.ones((len(pts), 1), f32)], 1)
    return (X @ med.T).T[:, :3].astype(f32)


def eval_diff(a, b):
    """func.cpp:19-32 (float accumulator in the reference; summed in float64 here: the test allows for it)"""
    return float(np.linalg.norm((a - b).astype(np.float64), axis=1).sum())


def register_symm(src, src_n, tgt, tgt_n, max_iters=10, diff_threshold=1.0):
    """myicp.cpp:117-142 with the reference's identity pairing"""
    src, src_n = src.copy(), src_n.copy()
    transform = np.eye(4, dtype=f32)
    iters, diffs, a_list, t_list, incs = 0, [], [], [], []
    diff = eval_diff(src, tgt)
    while diff > diff_threshold and iters < max_iters:
        iters += 1
        diffs.append(diff)
        X, a_, t_ = estimate_transform_symm(src, src_n, tgt, tgt_n)
        src = apply_transform(X, src)                                  # :136
        src_n = apply_transform(X, src_n)                              # :137 (translation included)
        transform = (X @ transform).astype(f32)                        # :138
        a_list.append(a_); t_list.append(t_); incs.append(X)
        diff = eval_diff(src, tgt)                                     # :141
    return dict(iters=iters, diffs=np.array(diffs + [diff]), a=np.array(a_list), t=np.array(t_list), increments=np.array(incs), transform=transform)


def normals_knn(xyz, k=10):
    """myicp.cpp:152-172: k nearest neighbours (the point itself included), covariance, eigenvector of the smallest
    eigenvalue, flipped towards the viewpoint (0, 0, 0)"""
    from scipy.spatial import cKDTree
    _, idx = cKDTree(xyz.astype(np.float64)).query(xyz.astype(np.float64), k=k)
    nb = xyz.astype(np.float64)[idx]                                   # [n, k, 3]
    d = nb - nb.mean(1, keepdims=True)
    C = np.einsum("nki,nkj->nij", d, d) / k
    w, V = np.linalg.eigh(C)
    nrm = V[:, :, 0]
    flip = np.einsum("ij,ij->i", nrm, -xyz.astype(np.float64)) < 0
    nrm[flip] *= -1
    return nrm.astype(f32)


def main():
    src, tgt = read_pcd_ascii(os.path.join(HERE, "cat.pcd")), read_pcd_ascii(os.path.join(HERE, "cat_out.pcd"))
    g = np.load(os.path.join(HERE, "cat_golden.npz"))
    r = register_symm(src, g["src_n"], tgt, g["tgt_n"])
    out = dict(quirks_iters=np.int32(r["iters"]), quirks_diffs=r["diffs"], quirks_a=r["a"], quirks_t=r["t"], quirks_increments=r["increments"],
               quirks_T=r["transform"], src_n_numpy=normals_knn(src), tgt_n_numpy=normals_knn(tgt))
    np.savez_compressed(os.path.join(HERE, "emulation_cat.npz"), **out)
    print("iters", r["iters"], "diffs", np.round(r["diffs"], 1))
    print("first a~", r["a"][0], "t~", r["t"][0])
    print(np.round(r["transform"], 5))


if __name__ == "__main__":
    main()

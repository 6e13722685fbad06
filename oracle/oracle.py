"""ctypes view of oracle/liboracle.so (the CPU restatement of the reference path).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product (icp-symm_amd/) never imports this.
See symmicp_oracle.c for what each function restates (reference file:line).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

NSUM = 40
MODE_QUIRKS, MODE_PAPER, MODE_P2P = 0, 1, 2
CORR_IDENTITY, CORR_BRUTE, CORR_GRID = 0, 1, 2
SOLVE_GRAM, SOLVE_LITERAL = 0, 1
APPLY_INCREMENTAL, APPLY_CUMULATIVE = 0, 1
OK, ERR_ARG, ERR_SIZE, ERR_DEGENERATE, ERR_IO = 0, 1, 2, 3, 4


class Config(C.Structure):
    _fields_ = [("mode", C.c_int32), ("corr", C.c_int32), ("solve", C.c_int32), ("apply", C.c_int32),
                ("max_iters", C.c_int32), ("diff_threshold", C.c_float), ("max_corr_dist", C.c_float),
                ("fixed_iters", C.c_int32), ("min_normal_dot", C.c_float), ("eps_rotation", C.c_float),
                ("eps_translation", C.c_float)]


class Result(C.Structure):
    _fields_ = [("transform", C.c_float * 16), ("iters", C.c_int32), ("status", C.c_int32),
                ("diff_initial", C.c_float), ("diff_final", C.c_float), ("diffs", C.c_float * 256),
                ("last_sums", C.c_double * NSUM), ("rcond", C.c_double)]


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "symmicp_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "liboracle.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        fp = C.POINTER(C.c_float)
        _LIB.orc_eval_diff.restype = C.c_float
        _LIB.orc_eval_diff_f64.restype = C.c_double
        _LIB.orc_pcd_read.restype = C.c_long
        _LIB.orc_grid_build.restype = C.c_void_p
        _LIB.orc_grid_build.argtypes = [fp, C.c_size_t, C.c_float]
        _LIB.orc_grid_free.argtypes = [C.c_void_p]
    return _LIB


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(C.POINTER(C.c_float))


def _xyz(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    assert a.ndim == 2 and a.shape[1] == 3
    return a, a.ctypes.data_as(C.POINTER(C.c_float))


def pcd_read(path):
    """-> (xyz [N,3] f32, normals [N,3] f32 or None)"""
    L = lib()
    hn = C.c_int(0)
    n = L.orc_pcd_read(path.encode(), None, None, C.c_size_t(0), C.byref(hn))
    if n < 0:
        raise IOError("orc_pcd_read(%s) -> %d" % (path, n))
    xyz = np.zeros((n, 3), np.float32)
    nrm = np.zeros((n, 3), np.float32)
    r = L.orc_pcd_read(path.encode(), xyz.ctypes.data_as(C.POINTER(C.c_float)),
                       nrm.ctypes.data_as(C.POINTER(C.c_float)), C.c_size_t(n), C.byref(hn))
    if r != n:
        raise IOError("orc_pcd_read(%s) -> %d" % (path, r))
    return xyz, (nrm if hn.value else None)


def rows(p, np_, q, nq):
    L = lib()
    p, pp = _xyz(p); np_, npp = _xyz(np_); q, qp = _xyz(q); nq, nqp = _xyz(nq)
    n = p.shape[0]
    M = np.zeros((n, 3), np.float32); N = np.zeros((n, 3), np.float32); c = np.zeros(n, np.float32)
    L.orc_rows(pp, npp, qp, nqp, C.c_size_t(n), M.ctypes.data_as(C.POINTER(C.c_float)),
               N.ctypes.data_as(C.POINTER(C.c_float)), c.ctypes.data_as(C.POINTER(C.c_float)))
    return M, N, c


def eval_diff(p, q):
    L = lib()
    p, pp = _xyz(p); q, qp = _xyz(q)
    return float(L.orc_eval_diff(pp, qp, C.c_size_t(p.shape[0])))


def eval_diff_f64(p, q):
    L = lib()
    p, pp = _xyz(p); q, qp = _xyz(q)
    return float(L.orc_eval_diff_f64(pp, qp, C.c_size_t(p.shape[0])))


def apply(X, pts, with_translation=True):
    L = lib()
    X, Xp = _f(np.asarray(X).reshape(16))
    pts, pp = _xyz(pts)
    out = np.zeros_like(pts)
    L.orc_apply(Xp, pp, out.ctypes.data_as(C.POINTER(C.c_float)), C.c_size_t(pts.shape[0]), C.c_int(int(with_translation)))
    return out


def reduce40(p, np_, q, nq, idx=None, pivot=None, max_d2=0.0, min_ndot=-2.0, p2p=False):
    L = lib()
    p, pp = _xyz(p); np_, npp = _xyz(np_); q, qp = _xyz(q); nq, nqp = _xyz(nq)
    S = np.zeros(NSUM, np.float64)
    ip = None
    if idx is not None:
        idx = np.ascontiguousarray(idx, dtype=np.int32)
        ip = idx.ctypes.data_as(C.POINTER(C.c_int32))
    pv = None
    if pivot is not None:
        pivot, pv = _f(pivot)
    L.orc_reduce40(pp, npp, C.c_size_t(p.shape[0]), qp, nqp, C.c_size_t(q.shape[0]), ip, pv,
                   C.c_float(max_d2), C.c_float(min_ndot), C.c_int(int(p2p)), S.ctypes.data_as(C.POINTER(C.c_double)))
    return S


def set_threads(n):
    """threads for the two O(N) loops of an iteration (bench.py's all-core CPU baseline); 1 = the serial reference code"""
    lib().orc_set_threads(int(n))


def solve_quirks_gram(S):
    L = lib()
    S = np.ascontiguousarray(S, np.float64)
    pbar = np.zeros(3, np.float32); qbar = np.zeros(3, np.float32); a = np.zeros(3, np.float32); t = np.zeros(3, np.float32)
    rc = C.c_double(0)
    fp = C.POINTER(C.c_float)
    st = L.orc_solve_quirks_gram(S.ctypes.data_as(C.POINTER(C.c_double)), pbar.ctypes.data_as(fp), qbar.ctypes.data_as(fp),
                                 a.ctypes.data_as(fp), t.ctypes.data_as(fp), C.byref(rc))
    return st, pbar, qbar, a, t, rc.value


def compose_quirks(pbar, qbar, a, t):
    """func.cpp:91-99: the five post-multiplied factors -> row-major 4x4"""
    L = lib()
    fp = C.POINTER(C.c_float)
    v = [np.ascontiguousarray(x, np.float32) for x in (pbar, qbar, a, t)]
    X = np.zeros(16, np.float32)
    L.orc_compose_quirks.restype = None
    L.orc_compose_quirks(*[x.ctypes.data_as(fp) for x in v], X.ctypes.data_as(fp))
    return X.reshape(4, 4)


def solve_quirks_literal(p, np_, q, nq):
    L = lib()
    p, pp = _xyz(p); np_, npp = _xyz(np_); q, qp = _xyz(q); nq, nqp = _xyz(nq)
    pbar = np.zeros(3, np.float32); qbar = np.zeros(3, np.float32); a = np.zeros(3, np.float32); t = np.zeros(3, np.float32)
    fp = C.POINTER(C.c_float)
    st = L.orc_solve_quirks_literal(pp, npp, qp, nqp, C.c_size_t(p.shape[0]), pbar.ctypes.data_as(fp), qbar.ctypes.data_as(fp),
                                    a.ctypes.data_as(fp), t.ctypes.data_as(fp))
    return st, pbar, qbar, a, t


def solve_paper(S, pivot=None):
    L = lib()
    S = np.ascontiguousarray(S, np.float64)
    pbar = np.zeros(3, np.float32); qbar = np.zeros(3, np.float32); a = np.zeros(3, np.float32); t = np.zeros(3, np.float32)
    rc = C.c_double(0)
    fp = C.POINTER(C.c_float)
    pv = None
    if pivot is not None:
        pivot, pv = _f(pivot)
    st = L.orc_solve_paper(S.ctypes.data_as(C.POINTER(C.c_double)), pv, pbar.ctypes.data_as(fp), qbar.ctypes.data_as(fp),
                           a.ctypes.data_as(fp), t.ctypes.data_as(fp), C.byref(rc))
    return st, pbar, qbar, a, t, rc.value


def compose(pbar, qbar, a, t, paper=False):
    L = lib()
    X = np.zeros(16, np.float32)
    args = [_f(v)[1] for v in (pbar, qbar, a, t)]
    keep = [_f(v)[0] for v in (pbar, qbar, a, t)]  # noqa: F841 (keep alive)
    ptrs = [k.ctypes.data_as(C.POINTER(C.c_float)) for k in keep]
    (L.orc_compose_paper if paper else L.orc_compose_quirks)(*ptrs, X.ctypes.data_as(C.POINTER(C.c_float)))
    del args
    return X.reshape(4, 4)


def kabsch(src, dst):
    """regist.h:8-72 registrateNPoint on index pairs -> (R [3,3] f64, T [3] f64, status); dst ~ R src + T"""
    L = lib()
    s_, sp = _xyz(src); d_, dp = _xyz(dst)
    R = np.zeros(9, np.float64); T = np.zeros(3, np.float64)
    st = L.orc_kabsch(sp, dp, C.c_size_t(s_.shape[0]), R.ctypes.data_as(C.POINTER(C.c_double)), T.ctypes.data_as(C.POINTER(C.c_double)))
    return R.reshape(3, 3), T, st


def solve_p2p(S, pivot=None):
    L = lib()
    S = np.ascontiguousarray(S, np.float64)
    X = np.zeros(16, np.float32)
    pv = None
    if pivot is not None:
        pivot, pv = _f(pivot)
    st = L.orc_solve_p2p(S.ctypes.data_as(C.POINTER(C.c_double)), pv, X.ctypes.data_as(C.POINTER(C.c_float)))
    return st, X.reshape(4, 4)


def nn_brute(p, q, X=None):
    L = lib()
    p, pp = _xyz(p); q, qp = _xyz(q)
    idx = np.zeros(p.shape[0], np.int32); d2 = np.zeros(p.shape[0], np.float32)
    Xp = None
    if X is not None:
        X, Xp = _f(np.asarray(X).reshape(16))
    L.orc_nn_brute(Xp, pp, C.c_size_t(p.shape[0]), qp, C.c_size_t(q.shape[0]),
                   idx.ctypes.data_as(C.POINTER(C.c_int32)), d2.ctypes.data_as(C.POINTER(C.c_float)))
    return idx, d2


def nn_grid(p, q, X=None, pts_per_cell=2.0):
    L = lib()
    p, pp = _xyz(p); q, qp = _xyz(q)
    g = L.orc_grid_build(qp, C.c_size_t(q.shape[0]), C.c_float(pts_per_cell))
    idx = np.zeros(p.shape[0], np.int32); d2 = np.zeros(p.shape[0], np.float32)
    Xp = None
    if X is not None:
        X, Xp = _f(np.asarray(X).reshape(16))
    L.orc_nn_grid(C.c_void_p(g), Xp, pp, C.c_size_t(p.shape[0]), qp,
                  idx.ctypes.data_as(C.POINTER(C.c_int32)), d2.ctypes.data_as(C.POINTER(C.c_float)))
    L.orc_grid_free(C.c_void_p(g))
    return idx, d2


def normals_knn(xyz, k=10, viewpoint=(0.0, 0.0, 0.0)):
    L = lib()
    xyz, xp = _xyz(xyz)
    nrm = np.zeros_like(xyz); curv = np.zeros(xyz.shape[0], np.float32)
    vp, vpp = _f(viewpoint)
    st = L.orc_normals_knn(xp, C.c_size_t(xyz.shape[0]), C.c_int(k), vpp,
                           nrm.ctypes.data_as(C.POINTER(C.c_float)), curv.ctypes.data_as(C.POINTER(C.c_float)))
    if st != OK:
        raise ValueError("orc_normals_knn -> %d" % st)
    return nrm, curv


def align(src_xyz, src_nrm, tgt_xyz, tgt_nrm, mode=MODE_QUIRKS, corr=CORR_IDENTITY, solve=SOLVE_GRAM,
          apply_mode=None, max_iters=10, diff_threshold=1.0, max_corr_dist=0.0, fixed_iters=False, guess=None,
          min_normal_dot=-2.0, eps_rotation=0.0, eps_translation=0.0):
    """myicp.cpp:100-150.  Returns dict(status, transform[4,4], iters, diffs, diff_initial, diff_final, sums, rcond)."""
    L = lib()
    cfg = Config()
    L.orc_config_default(C.byref(cfg))
    cfg.mode, cfg.corr, cfg.solve = mode, corr, solve
    if apply_mode is None:
        apply_mode = APPLY_INCREMENTAL if mode == MODE_QUIRKS else APPLY_CUMULATIVE
    cfg.apply = apply_mode
    cfg.max_iters, cfg.diff_threshold, cfg.max_corr_dist, cfg.fixed_iters = max_iters, diff_threshold, max_corr_dist, int(fixed_iters)
    cfg.min_normal_dot, cfg.eps_rotation, cfg.eps_translation = min_normal_dot, eps_rotation, eps_translation
    s, sp = _xyz(src_xyz); sn, snp = _xyz(src_nrm); t, tp = _xyz(tgt_xyz); tn, tnp = _xyz(tgt_nrm)
    gp = None
    if guess is not None:
        guess, gp = _f(np.asarray(guess).reshape(16))
    res = Result()
    st = L.orc_align(C.byref(cfg), sp, snp, C.c_size_t(s.shape[0]), tp, tnp, C.c_size_t(t.shape[0]), gp, C.byref(res))
    it = res.iters
    return dict(status=st, transform=np.array(res.transform, np.float32).reshape(4, 4), iters=it,
                diffs=np.array(res.diffs[:max(0, min(it, 256))], np.float32), diff_initial=res.diff_initial,
                diff_final=res.diff_final, sums=np.array(res.last_sums, np.float64), rcond=res.rcond)

"""symmicp -- ctypes bindings of libsymmicp.so (include/symmicp.h), the MI355X
symmetric-ICP engine, plus `MyICP`, a Python mirror of the reference's class
surface (reference ICP/myicp.h:7-36: LoadCloud / GetSrcCloud / GetTgtCloud /
RegisterSymm) with the additive setInputSource / setInputTarget / align names.

The library is HIP-only: there is no CPU fallback here, and loading fails
loudly when icp-symm_amd/lib/libsymmicp.so has not been built
(`make -C icp-symm_amd`, or __graft_entry__.build()).
"""
import ctypes as C
import os
import sys

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(os.path.dirname(_PKG))           # icp-symm_amd/
LIB_PATH = os.environ.get("SYMMICP_LIB") or os.path.join(_ROOT, "lib", "libsymmicp.so")      # SYMMICP_LIB: A/B builds in scratch runs

NSUM = 40
UNIQUE_ID_BYTES = 128
OK, ERR_ARG, ERR_SIZE, ERR_DEGENERATE, ERR_IO, ERR_HIP, ERR_STATE, ERR_COMM = range(8)
MODE_QUIRKS, MODE_PAPER, MODE_P2P = 0, 1, 2
CORR_IDENTITY, CORR_BRUTE, CORR_TREE = 0, 1, 2
APPLY_DEFAULT, APPLY_INCREMENTAL, APPLY_CUMULATIVE = 0, 1, 2

_STATUS_NAMES = {0: "OK", 1: "ERR_ARG", 2: "ERR_SIZE", 3: "ERR_DEGENERATE", 4: "ERR_IO", 5: "ERR_HIP",
                 6: "ERR_STATE", 7: "ERR_COMM"}


class SymmIcpError(RuntimeError):
    def __init__(self, status, msg=""):
        self.status = status
        super().__init__("symmicp: %s (%d) %s" % (_STATUS_NAMES.get(status, "?"), status, msg))


class Config(C.Structure):
    _fields_ = [("struct_size", C.c_int32), ("device", C.c_int32), ("mode", C.c_int32), ("corr", C.c_int32),
                ("apply", C.c_int32), ("max_iters", C.c_int32), ("diff_threshold", C.c_float),
                ("max_corr_dist", C.c_float), ("fixed_iters", C.c_int32), ("sort_source", C.c_int32),
                ("verbose", C.c_int32), ("min_normal_dot", C.c_float), ("eps_rotation", C.c_float),
                ("eps_translation", C.c_float), ("host_loop", C.c_int32), ("reserved", C.c_int32 * 1)]


class Sums(C.Structure):
    _fields_ = [("s", C.c_double * NSUM)]


class IterResult(C.Structure):
    _fields_ = [("status", C.c_int32), ("iter", C.c_int32), ("diff", C.c_float), ("rcond", C.c_float),
                ("pairs", C.c_double), ("increment", C.c_float * 16), ("sums", Sums)]


class Result(C.Structure):
    _fields_ = [("status", C.c_int32), ("iters", C.c_int32), ("diff_initial", C.c_float), ("diff_final", C.c_float),
                ("transform", C.c_float * 16), ("diffs", C.c_float * 64), ("seconds_total", C.c_double)]


class Stats(C.Structure):
    _fields_ = [("last_pass_ms", C.c_double), ("sum_pass_ms", C.c_double), ("passes", C.c_int64),
                ("build_ms", C.c_double), ("upload_ms", C.c_double), ("grid_level", C.c_int32),
                ("tree_levels", C.c_int32), ("pass_blocks", C.c_int64), ("bytes_algorithmic_per_pass", C.c_int64),
                ("kernel_ms", C.c_double * 8), ("kernel_launches", C.c_int64 * 8), ("pass_ms_head", C.c_double * 8), ("passes_timed", C.c_int64),
                ("loop_passes", C.c_int64), ("loop_straggler_passes", C.c_int64), ("packet_fallbacks", C.c_int64),
                ("allreduce_ms", C.c_double), ("allreduce_timed", C.c_int64)]


KERNEL_SLOTS = ["k_search_cells", "(gap)", "k_search_walk", "k_accumulate", "k_final_reduce", "single_pass_kernel", "whole_pass"]


# every symbol include/symmicp.h declares (checked by tests/test_abi.py)
EXPORTS = [
    "symmicp_config_default", "symmicp_create", "symmicp_destroy", "symmicp_last_error", "symmicp_set_config",
    "symmicp_version", "symmicp_set_source", "symmicp_set_target", "symmicp_align", "symmicp_begin", "symmicp_step",
    "symmicp_get_transform", "symmicp_format_result", "symmicp_get_pivot", "symmicp_get_correspondences", "symmicp_get_source",
    "symmicp_local_source_count", "symmicp_local_source_offset", "symmicp_get_certificates", "symmicp_solve", "symmicp_comm_get_unique_id",
    "symmicp_comm_init_rank", "symmicp_set_sums", "symmicp_comm_init_shm", "symmicp_shard_range", "symmicp_get_stats", "symmicp_reset_stats", "symmicp_enable_timing",
    "symmicp_pcd_read", "symmicp_pcd_write", "symmicp_estimate_normals", "symmicp_ctx_estimate_normals",
]

_lib = None


def _hip_runtime_first():
    """Make sure only ONE HIP runtime ends up in the process.  torch ships its own
    libamdhip64.so (found through its RPATH under the un-versioned name); if libsymmicp were
    loaded first it would pull /opt/rocm's copy and a later `import torch` would map a second
    runtime.  Importing torch first makes our NEEDED libamdhip64.so.7 resolve to torch's copy."""
    if "torch" in sys.modules or os.environ.get("SYMMICP_NO_TORCH_PRELOAD"):
        return
    try:
        import torch  # noqa: F401
    except Exception:
        pass


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SymmIcpError(ERR_HIP, "libsymmicp.so not built at %s (run `make -C icp-symm_amd`); no CPU fallback" % LIB_PATH)
    _hip_runtime_first()
    L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    fp = C.POINTER(C.c_float)
    vp = C.c_void_p
    L.symmicp_config_default.argtypes = [C.POINTER(Config)]
    L.symmicp_config_default.restype = None
    L.symmicp_create.argtypes = [C.POINTER(Config), C.POINTER(vp)]
    L.symmicp_destroy.argtypes = [vp]
    L.symmicp_destroy.restype = None
    L.symmicp_last_error.argtypes = [vp]
    L.symmicp_last_error.restype = C.c_char_p
    L.symmicp_set_config.argtypes = [vp, C.POINTER(Config)]
    for nm in ("symmicp_set_source", "symmicp_set_target"):
        getattr(L, nm).argtypes = [vp, fp, C.c_size_t, C.c_size_t, fp, C.c_size_t, C.c_size_t, C.c_size_t]
    L.symmicp_align.argtypes = [vp, fp, C.POINTER(Result)]
    L.symmicp_begin.argtypes = [vp, fp, C.POINTER(IterResult)]
    L.symmicp_step.argtypes = [vp, C.POINTER(IterResult)]
    L.symmicp_get_transform.argtypes = [vp, fp]
    L.symmicp_get_pivot.argtypes = [vp, fp]
    L.symmicp_format_result.argtypes = [fp, C.c_char_p, C.c_size_t]
    L.symmicp_format_result.restype = C.c_size_t
    L.symmicp_get_correspondences.argtypes = [vp, C.POINTER(C.c_int32), fp, C.c_size_t]
    L.symmicp_get_source.argtypes = [vp, fp, fp, C.c_size_t]
    L.symmicp_local_source_count.argtypes = [vp]
    L.symmicp_local_source_count.restype = C.c_size_t
    L.symmicp_local_source_offset.argtypes = [vp]
    L.symmicp_local_source_offset.restype = C.c_size_t
    L.symmicp_solve.argtypes = [C.c_int, C.POINTER(Sums), fp, fp, fp, fp, fp, fp, fp]
    L.symmicp_comm_get_unique_id.argtypes = [vp]
    L.symmicp_comm_init_rank.argtypes = [vp, C.c_int, C.c_int, vp]
    L.symmicp_set_sums.argtypes = [vp, C.POINTER(Sums)]
    L.symmicp_comm_init_shm.argtypes = [vp, C.c_int, C.c_int, C.c_char_p]
    L.symmicp_shard_range.argtypes = [C.c_size_t, C.c_int, C.c_int, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    L.symmicp_get_stats.argtypes = [vp, C.POINTER(Stats)]
    L.symmicp_reset_stats.argtypes = [vp]
    L.symmicp_enable_timing.argtypes = [vp, C.c_int]
    L.symmicp_pcd_read.argtypes = [C.c_char_p, fp, fp, C.c_size_t, C.POINTER(C.c_int)]
    L.symmicp_pcd_read.restype = C.c_long
    L.symmicp_pcd_write.argtypes = [C.c_char_p, fp, fp, C.c_size_t, C.c_int]
    L.symmicp_estimate_normals.argtypes = [C.c_int, fp, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int, fp, fp, fp]
    _lib = L
    return L


def _fptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _cloud(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    if a.ndim != 2 or a.shape[1] != 3:
        raise ValueError("cloud must be [N,3]")
    return a


def default_config(**kw):
    cfg = Config()
    lib().symmicp_config_default(C.byref(cfg))
    for k, v in kw.items():
        if not hasattr(cfg, k):
            raise AttributeError(k)
        setattr(cfg, k, v)
    return cfg


def pcd_read(path):
    """PCD v0.7 (ascii/binary) -> (xyz [N,3] f32, normals [N,3] f32 or None). myicp.cpp:20-31."""
    L = lib()
    hn = C.c_int(0)
    n = L.symmicp_pcd_read(os.fsencode(path), None, None, 0, C.byref(hn))
    if n < 0:
        raise SymmIcpError(-n, "pcd_read(%s)" % path)
    xyz = np.zeros((n, 3), np.float32)
    nrm = np.zeros((n, 3), np.float32)
    r = L.symmicp_pcd_read(os.fsencode(path), _fptr(xyz), _fptr(nrm), n, C.byref(hn))
    if r != n:
        raise SymmIcpError(-r if r < 0 else ERR_IO, "pcd_read(%s)" % path)
    return xyz, (nrm if hn.value else None)


def pcd_write(path, xyz, nrm=None, binary=False):
    xyz = _cloud(xyz)
    nrm_p = None
    if nrm is not None:
        nrm = _cloud(nrm)
        nrm_p = _fptr(nrm)
    st = lib().symmicp_pcd_write(os.fsencode(path), _fptr(xyz), nrm_p, xyz.shape[0], int(binary))
    if st != OK:
        raise SymmIcpError(st, "pcd_write(%s)" % path)


def solve(mode, sums, pivot=None):
    """Host part of estimateTransformSymm (func.cpp:76-102) on one reduction record."""
    S = Sums()
    for k, v in enumerate(np.asarray(sums, np.float64).reshape(NSUM)):
        S.s[k] = v
    pb = np.zeros(3, np.float32); qb = np.zeros(3, np.float32); a = np.zeros(3, np.float32); t = np.zeros(3, np.float32)
    X = np.zeros(16, np.float32)
    rc = C.c_float(0)
    pv = None
    if pivot is not None:
        pivot = np.ascontiguousarray(pivot, np.float32)
        pv = _fptr(pivot)
    st = lib().symmicp_solve(mode, C.byref(S), pv, _fptr(pb), _fptr(qb), _fptr(a), _fptr(t), C.byref(rc), _fptr(X))
    return st, pb, qb, a, t, rc.value, X.reshape(4, 4)


def format_result(transform):
    """The reference's result block for a 4x4 (myicp.cpp:146-149, Eigen's default IOFormat); needs no GPU."""
    X = np.ascontiguousarray(np.asarray(transform, np.float32).reshape(16))
    n = lib().symmicp_format_result(_fptr(X), None, 0)
    buf = C.create_string_buffer(n + 1)
    lib().symmicp_format_result(_fptr(X), buf, n + 1)
    return buf.value.decode()


def estimate_normals(xyz, k=10, viewpoint=(0.0, 0.0, 0.0), device=-1):
    """k-NN PCA normals on the GPU (MyICP::estimateNormals, myicp.cpp:152-172). -> (normals, curvature)"""
    xyz = _cloud(xyz)
    n = xyz.shape[0]
    nrm = np.zeros((n, 3), np.float32)
    curv = np.zeros(n, np.float32)
    vp = np.ascontiguousarray(viewpoint, np.float32)
    st = lib().symmicp_estimate_normals(device, _fptr(xyz), 3, 1, n, k, _fptr(vp), _fptr(nrm), _fptr(curv))
    if st != OK:
        raise SymmIcpError(st, "estimate_normals")
    return nrm, curv


def shard_range(n, nranks, rank):
    b, c = C.c_size_t(0), C.c_size_t(0)
    st = lib().symmicp_shard_range(n, nranks, rank, C.byref(b), C.byref(c))
    if st != OK:
        raise SymmIcpError(st, "shard_range")
    return int(b.value), int(c.value)


def comm_get_unique_id():
    buf = C.create_string_buffer(UNIQUE_ID_BYTES)
    st = lib().symmicp_comm_get_unique_id(C.cast(buf, C.c_void_p))
    if st != OK:
        raise SymmIcpError(st, "comm_get_unique_id")
    return bytes(buf.raw)


class Engine:
    """Thin object wrapper over one symmicp_ctx."""

    def __init__(self, **cfg):
        self._L = lib()
        self.cfg = default_config(**cfg)
        h = C.c_void_p()
        st = self._L.symmicp_create(C.byref(self.cfg), C.byref(h))
        if st != OK:
            raise SymmIcpError(st, "symmicp_create (no usable gfx950 HIP device? there is no CPU fallback)")
        self._h = h
        self._keep = []

    def close(self):
        if getattr(self, "_h", None):
            self._L.symmicp_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _chk(self, st):
        if st != OK:
            raise SymmIcpError(st, (self._L.symmicp_last_error(self._h) or b"").decode())

    def set_config(self, **kw):
        for k, v in kw.items():
            setattr(self.cfg, k, v)
        self._chk(self._L.symmicp_set_config(self._h, C.byref(self.cfg)))

    def comm_init_rank(self, nranks, rank, uid):
        buf = C.create_string_buffer(uid, UNIQUE_ID_BYTES) if uid is not None else None
        self._chk(self._L.symmicp_comm_init_rank(self._h, nranks, rank, C.cast(buf, C.c_void_p) if buf else None))

    def comm_init_shm(self, nranks, rank, job_name):
        """ranks of one node: exchange the per-pass record through POSIX shared memory instead of RCCL"""
        self._chk(self._L.symmicp_comm_init_shm(self._h, nranks, rank, str(job_name).encode()))

    def set_sums(self, total):
        """external exchange (comm_init_rank(nranks, rank, None)): hand the record summed over all ranks back before step()"""
        s = Sums()
        t = np.ascontiguousarray(np.asarray(total, np.float64).reshape(-1))
        for k in range(len(s.s)):
            s.s[k] = float(t[k])
        self._chk(self._L.symmicp_set_sums(self._h, C.byref(s)))

    def set_source(self, xyz, nrm):
        xyz, nrm = _cloud(xyz), _cloud(nrm)
        self._chk(self._L.symmicp_set_source(self._h, _fptr(xyz), 3, 1, _fptr(nrm), 3, 1, xyz.shape[0]))
        self.n_source = xyz.shape[0]

    def set_target(self, xyz, nrm):
        xyz, nrm = _cloud(xyz), _cloud(nrm)
        self._chk(self._L.symmicp_set_target(self._h, _fptr(xyz), 3, 1, _fptr(nrm), 3, 1, xyz.shape[0]))
        self.n_target = xyz.shape[0]

    def set_source_strided(self, xyz, xr, xc, nrm, nr, nc, n):
        self._chk(self._L.symmicp_set_source(self._h, _fptr(xyz), xr, xc, _fptr(nrm), nr, nc, n))
        self.n_source = n

    def set_target_strided(self, xyz, xr, xc, nrm, nr, nc, n):
        self._chk(self._L.symmicp_set_target(self._h, _fptr(xyz), xr, xc, _fptr(nrm), nr, nc, n))
        self.n_target = n

    @staticmethod
    def _guess(guess):
        if guess is None:
            return None, None
        g = np.ascontiguousarray(np.asarray(guess, np.float32).reshape(16))
        return g, _fptr(g)

    def _iter_dict(self, it):
        return dict(status=it.status, iter=it.iter, diff=it.diff, rcond=it.rcond, pairs=it.pairs,
                    increment=np.array(it.increment, np.float32).reshape(4, 4), sums=np.array(it.sums.s, np.float64))

    def begin(self, guess=None):
        g, gp = self._guess(guess)
        it = IterResult()
        self._chk(self._L.symmicp_begin(self._h, gp, C.byref(it)))
        return self._iter_dict(it)

    def step(self, check=True):
        it = IterResult()
        st = self._L.symmicp_step(self._h, C.byref(it))
        if check:
            self._chk(st)
        return self._iter_dict(it)

    def step_raw(self, it):
        """hot-loop variant: caller-provided IterResult, returns the status only"""
        return self._L.symmicp_step(self._h, C.byref(it))

    def align(self, guess=None, check=False):
        g, gp = self._guess(guess)
        res = Result()
        st = self._L.symmicp_align(self._h, gp, C.byref(res))
        if check:
            self._chk(st)
        n = max(0, min(res.iters, 64))
        return dict(status=st, iters=res.iters, diff_initial=res.diff_initial, diff_final=res.diff_final,
                    transform=np.frombuffer(res.transform, np.float32).reshape(4, 4).copy(),      # (frombuffer: no per-element conversion)
                    diffs=np.frombuffer(res.diffs, np.float32)[:n].copy(), seconds=res.seconds_total,
                    error=(self._L.symmicp_last_error(self._h) or b"").decode() if st != OK else "")

    def certificates(self):
        """diagnostic: (cert [n_loc, 4], neighbourhood members [n_loc, 8] as target rows, neighbourhood radius T [n_loc],
        winner row [n_loc]) of this rank's share, in its sorted order"""
        n = self.local_count()
        ce = np.zeros((n, 4), np.float32)
        hood = np.zeros((n, 8), np.uint32)
        T = np.zeros(n, np.float32)
        win = np.zeros(n, np.int32)
        self._chk(self._L.symmicp_get_certificates(self._h, _fptr(ce), hood.ctypes.data_as(C.POINTER(C.c_uint32)), _fptr(T),
                                                   win.ctypes.data_as(C.POINTER(C.c_int32)), C.c_size_t(n)))
        return ce, hood, T, win

    def transform(self):
        X = np.zeros(16, np.float32)
        self._chk(self._L.symmicp_get_transform(self._h, _fptr(X)))
        return X.reshape(4, 4)

    def pivot(self):
        p = np.zeros(3, np.float32)
        self._chk(self._L.symmicp_get_pivot(self._h, _fptr(p)))
        return p

    def local_count(self):
        return int(self._L.symmicp_local_source_count(self._h))

    def local_offset(self):
        return int(self._L.symmicp_local_source_offset(self._h))

    def correspondences(self):
        n = self.n_source
        idx = np.full(n, -1, np.int32)
        d2 = np.zeros(n, np.float32)
        self._chk(self._L.symmicp_get_correspondences(self._h, idx.ctypes.data_as(C.POINTER(C.c_int32)), _fptr(d2), n))
        return idx, d2

    def source(self):
        n = self.n_source
        xyz = np.zeros((n, 3), np.float32)
        nrm = np.zeros((n, 3), np.float32)
        self._chk(self._L.symmicp_get_source(self._h, _fptr(xyz), _fptr(nrm), n))
        return xyz, nrm

    def enable_timing(self, on=True):
        self._chk(self._L.symmicp_enable_timing(self._h, int(on)))      # 0 off, 1 per pass, 2 per kernel

    def reset_stats(self):
        self._chk(self._L.symmicp_reset_stats(self._h))

    def stats(self):
        s = Stats()
        self._chk(self._L.symmicp_get_stats(self._h, C.byref(s)))
        d = {k: getattr(s, k) for k, _ in Stats._fields_}
        d["kernel_ms"] = list(s.kernel_ms)
        d["kernel_launches"] = list(s.kernel_launches)
        d["pass_ms_head"] = list(s.pass_ms_head)
        return d


class MyICP:
    """Python mirror of the reference's MyICP (ICP/myicp.h:7-36).

    LoadCloud / GetSrcCloud / GetTgtCloud / RegisterSymm keep the reference's names and
    behaviour (defaults max_iters=10, diff_threshold=1.0, identity pairing, QUIRKS arithmetic,
    result printed).  setInputSource / setInputTarget / align / getFinalTransformation are the
    additive surface BASELINE.json's north_star names.
    """

    def __init__(self, mode=MODE_QUIRKS, corr=CORR_IDENTITY, max_iters=10, diff_threshold=1.0, verbose=True, **extra):
        self._cfg = dict(mode=mode, corr=corr, max_iters=max_iters, diff_threshold=diff_threshold,
                         verbose=int(bool(verbose)), **extra)
        self.cloud_src = self.cloud_tgt = None
        self.normals_src = self.normals_tgt = None
        self._final = np.eye(4, dtype=np.float32)
        self.last_result = None

    def LoadCloud(self, src_path, tgt_path):
        # myicp.cpp:20-31 (reader status is ignored there; here a bad file raises)
        self.cloud_src, _ = pcd_read(src_path)
        self.cloud_tgt, _ = pcd_read(tgt_path)
        self.normals_src = self.normals_tgt = None
        return 0

    def GetSrcCloud(self):
        return self.cloud_src

    def GetTgtCloud(self):
        return self.cloud_tgt

    def setInputSource(self, xyz, normals=None):
        self.cloud_src = _cloud(xyz)
        self.normals_src = None if normals is None else _cloud(normals)

    def setInputTarget(self, xyz, normals=None):
        self.cloud_tgt = _cloud(xyz)
        self.normals_tgt = None if normals is None else _cloud(normals)

    def estimateNormals(self):
        # myicp.cpp:152-172: k = 10, flipped toward the origin
        if self.normals_src is None:
            self.normals_src, _ = estimate_normals(self.cloud_src, 10)
        if self.normals_tgt is None:
            self.normals_tgt, _ = estimate_normals(self.cloud_tgt, 10)

    def RegisterP2P(self):
        # myicp.cpp:43-59: the reference stub applies an identity guess and computes nothing
        print("guess matrix:\n%s" % np.eye(4, dtype=np.float32))

    def align(self, guess=None):
        assert self.cloud_src is not None and self.cloud_tgt is not None      # myicp.cpp:102
        self.estimateNormals()                                               # myicp.cpp:105
        with Engine(**self._cfg) as e:
            e.set_target(self.cloud_tgt, self.normals_tgt)
            e.set_source(self.cloud_src, self.normals_src)
            self.last_result = e.align(guess)
        self._final = self.last_result["transform"]
        return self.last_result

    def RegisterSymm(self):
        self.align()

    def getFinalTransformation(self):
        return self._final

    def GetAlignedSrcCloud(self):
        """the source moved by the final transform (the reference never writes its result back, myicp.cpp:109-111)"""
        X = self._final.astype(np.float64)
        return (self.cloud_src.astype(np.float64) @ X[:3, :3].T + X[:3, 3]).astype(np.float32)

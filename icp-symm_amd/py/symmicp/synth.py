"""Synthetic cloud pairs for the BASELINE.json configs.

Counter-based (splitmix64) so every box, rank and language regenerates the same
bits without shipping data.  numpy only; inputs for both the HIP path and the
CPU oracle.

  c3_uniform(n)  C3: i.i.d. uniform cube, random unit normals, 5 deg + small t, target row-shuffled
  c4_surface(n)  C4: height-field surface with analytic normals, target = a DIFFERENT sampling of
                     the same surface moved by 3 deg + small t (no exact twins)
  c5_scan(n)     C5: scan-like rings ray-cast onto the C4 surface + a ground plane, range noise
  perturbed(xyz, nrm, deg, axis, t)  C2 substitute: a cloud against its own rigid perturbation
"""
import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _mix64(z):
    with np.errstate(over="ignore"):
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def splitmix64(seed, n, stream=0, first=0):
    """n uint64 values, numbers first .. first+n-1 of the sequence.  (seed, stream) is hashed into a base first, so
    neighbouring seeds or streams give unrelated sequences; value i = mix64(base + (i+1) * golden)."""
    with np.errstate(over="ignore"):
        base = _mix64(np.array([np.uint64(seed) * np.uint64(0x9E3779B97F4A7C15) + np.uint64(stream) * np.uint64(0xD1B54A32D192ED03) + np.uint64(0x2545F4914F6CDD1D)], np.uint64))[0]
        z = base + np.arange(first + 1, first + n + 1, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)
    return _mix64(z)


def uniform01(seed, n, stream=0, first=0):
    """float64 in [0,1) with 53 random bits"""
    return (splitmix64(seed, n, stream, first) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def rotation(deg, axis):
    a = np.asarray(axis, np.float64)
    a = a / np.linalg.norm(a)
    th = np.deg2rad(deg)
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    return np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * (K @ K)


def rigid4(R, t):
    X = np.eye(4)
    X[:3, :3] = R
    X[:3, 3] = t
    return X


def _unit_sphere(seed, n, stream):
    z = 2.0 * uniform01(seed, n, stream) - 1.0
    ph = 2.0 * np.pi * uniform01(seed, n, stream + 1)
    r = np.sqrt(np.maximum(0.0, 1.0 - z * z))
    return np.stack([r * np.cos(ph), r * np.sin(ph), z], 1)


def shuffle_perm(seed, n):
    return np.argsort(splitmix64(seed, n, 7), kind="stable")


def _spread10(v):
    v = v.astype(np.uint64) & np.uint64(0x3FF)
    v = (v | (v << np.uint64(16))) & np.uint64(0x030000FF)
    v = (v | (v << np.uint64(8))) & np.uint64(0x0300F00F)
    v = (v | (v << np.uint64(4))) & np.uint64(0x030C30C3)
    v = (v | (v << np.uint64(2))) & np.uint64(0x09249249)
    return v


def sweep_order(xyz):
    """Row permutation that puts a cloud in a spatially coherent order (30-bit Morton curve over its bounding box, stable): what a
    scanner's sweep gives for free and what contiguous row ranges need to be compact shares in a multi-GPU run."""
    p = np.asarray(xyz, np.float64)
    lo = p.min(0)
    ext = max(float((p.max(0) - lo).max()), 1e-30) * 1.00001
    q = np.minimum(((p - lo) / ext * 1024.0).astype(np.int64), 1023)
    key = (_spread10(q[:, 2]) << np.uint64(2)) | (_spread10(q[:, 1]) << np.uint64(1)) | _spread10(q[:, 0])
    return np.argsort(key, kind="stable")


def c3_uniform(n=100_000, seed=0xC3):
    src = np.stack([uniform01(seed, n, 0), uniform01(seed, n, 1), uniform01(seed, n, 2)], 1)
    nrm = _unit_sphere(seed, n, 3)
    R = rotation(5.0, (1, 2, 3))
    t = np.array([0.01, -0.02, 0.015])
    perm = shuffle_perm(seed + 1, n)
    tgt = (src @ R.T + t)[perm]
    tn = (nrm @ R.T)[perm]
    return dict(src=src.astype(np.float32), src_n=nrm.astype(np.float32), tgt=tgt.astype(np.float32),
                tgt_n=tn.astype(np.float32), truth=rigid4(R, t), perm=perm)


def _surface(u, v):
    z = 0.1 * np.sin(4 * np.pi * u) * np.cos(6 * np.pi * v) + 0.05 * np.sin(10 * np.pi * u + 1.0)
    dzu = 0.4 * np.pi * np.cos(4 * np.pi * u) * np.cos(6 * np.pi * v) + 0.5 * np.pi * np.cos(10 * np.pi * u + 1.0)
    dzv = -0.6 * np.pi * np.sin(4 * np.pi * u) * np.sin(6 * np.pi * v)
    nrm = np.stack([-dzu, -dzv, np.ones_like(u)], 1)
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    return np.stack([u, v, z], 1), nrm


def c4_surface(n=1_000_000, seed=0xC4):
    ps, ns = _surface(uniform01(seed, n, 0), uniform01(seed, n, 1))
    pt, nt = _surface(uniform01(seed + 1, n, 0), uniform01(seed + 1, n, 1))
    R = rotation(3.0, (2, -1, 4))
    t = np.array([0.004, 0.003, -0.002])
    tgt = pt @ R.T + t
    tn = nt @ R.T
    return dict(src=ps.astype(np.float32), src_n=ns.astype(np.float32), tgt=tgt.astype(np.float32),
                tgt_n=tn.astype(np.float32), truth=rigid4(R, t))


def _height(x, y):
    """C4 height field extended periodically, clipped from below by a ground plane z = -0.12"""
    z = 0.1 * np.sin(4 * np.pi * x) * np.cos(6 * np.pi * y) + 0.05 * np.sin(10 * np.pi * x + 1.0)
    return np.maximum(z, -0.12)


def _raycast(origin, d):
    """first hit of rays origin + s d (d.z < 0) with the clipped height field: coarse march + bisection"""
    s_max = (-0.16 - origin[2]) / d[:, 2]                 # below the ground plane: g(s_max) < 0
    lo = np.zeros(len(d))
    hi = s_max.copy()
    found = np.zeros(len(d), bool)
    steps = 48
    for k in range(1, steps + 1):
        s = s_max * (k / steps)
        p = origin + d * s[:, None]
        neg = (p[:, 2] - _height(p[:, 0], p[:, 1])) <= 0
        newly = neg & ~found
        hi = np.where(newly, s, hi)
        lo = np.where(~found & ~neg, s, lo)
        found |= neg
    for _ in range(26):
        mid = 0.5 * (lo + hi)
        p = origin + d * mid[:, None]
        neg = (p[:, 2] - _height(p[:, 0], p[:, 1])) <= 0
        hi = np.where(neg, mid, hi)
        lo = np.where(neg, lo, mid)
    return 0.5 * (lo + hi)


def _scan_normals(p):
    eps = 1e-4
    ground = _height(p[:, 0], p[:, 1]) <= -0.12 + 1e-9
    dzx = (_height(p[:, 0] + eps, p[:, 1]) - _height(p[:, 0] - eps, p[:, 1])) / (2 * eps)
    dzy = (_height(p[:, 0], p[:, 1] + eps) - _height(p[:, 0], p[:, 1] - eps)) / (2 * eps)
    n = np.stack([-dzx, -dzy, np.ones(len(p))], 1)
    n[ground] = [0.0, 0.0, 1.0]
    return n / np.linalg.norm(n, axis=1, keepdims=True)


def _c5_sweep_chunk(args):
    """rays i0 .. i1-1 of one sweep (ray i depends on i alone: chunks can be cast on separate cores)"""
    sd, phase, per, rings, i0, i1 = args
    cnt = i1 - i0
    idx = np.arange(i0, i1)
    ring, az_idx = idx // per, idx % per
    origin = np.array([0.5, 0.5, 1.2])
    az = 2 * np.pi * (az_idx + phase + 0.3 * uniform01(sd, cnt, 0, i0)) / per
    polar = np.deg2rad(5.0 + 35.0 * (ring + uniform01(sd, cnt, 1, i0)) / rings)
    d = np.stack([np.sin(polar) * np.cos(az), np.sin(polar) * np.sin(az), -np.cos(polar)], 1)
    s = _raycast(origin, d)
    clean = origin + d * s[:, None]
    noise = 1e-3 * np.sqrt(-2 * np.log(1 - uniform01(sd, cnt, 2, i0))) * np.cos(2 * np.pi * uniform01(sd, cnt, 3, i0))
    return origin + d * (s + noise)[:, None], _scan_normals(clean)


def _c5_worker_main():
    """child side of _c5_run_jobs: jobs (pickled by the parent) on stdin, one .npy pair per job on stdout"""
    import io
    import pickle
    import sys
    out = sys.stdout.buffer
    for job in pickle.load(sys.stdin.buffer):
        for arr in _c5_sweep_chunk(job):
            buf = io.BytesIO()                      # np.save wants a seekable file; stdout is a pipe
            np.save(buf, arr)
            out.write(buf.getbuffer())
    out.flush()


def _c5_run_jobs(jobs, workers):
    """The chunks on `workers` child interpreters.  Children are started as `python -c` (not fork: the caller may hold an
    initialised GPU runtime, which a forked copy would inherit and which would count as one more process on the card; not a
    multiprocessing pool: a spawned pool re-runs the caller's main module, which hangs scripts without a __main__ guard);
    they import numpy and this file only."""
    import io
    import os
    import pickle
    import subprocess
    import sys
    import threading
    # this file alone, by path: the package's __init__ (which binds the HIP library) stays out of the children
    code = ("import importlib.util as u; s = u.spec_from_file_location('symmicp_synth', %r); m = u.module_from_spec(s); "
            "s.loader.exec_module(m); m._c5_worker_main()") % os.path.abspath(__file__)
    workers = min(workers, len(jobs))
    shares = [list(range(w, len(jobs), workers)) for w in range(workers)]
    outs, errs = [None] * workers, [None] * workers

    def drive(w):
        pr = subprocess.Popen([sys.executable, "-c", code], stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        outs[w], errs[w] = pr.communicate(pickle.dumps([jobs[k] for k in shares[w]]))
        if pr.returncode != 0:
            outs[w] = None

    threads = [threading.Thread(target=drive, args=(w,)) for w in range(workers)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    parts = [None] * len(jobs)
    for w in range(workers):
        if outs[w] is None:
            raise RuntimeError("c5_scan worker failed: " + (errs[w] or b"").decode(errors="replace")[-400:])
        buf = io.BytesIO(outs[w])
        for k in shares[w]:
            parts[k] = (np.load(buf), np.load(buf))
    return parts


def c5_scan(n=8_000_000, seed=0xC5, rings=64, workers=None):
    """C5, scan-like: a sensor 1.2 above the (periodically extended) C4 height field sweeps `rings`
    cones (5..40 deg off nadir) x n/rings azimuth samples; rays are cast onto the surface, which is
    clipped from below by a ground plane; range noise sigma = 1e-3.  Inner rings are much denser than
    outer ones (same sample count on a smaller circle), as in a real sweep.  The target is the same
    sweep half an azimuth step later (different sample points) moved by 2 deg + small t.
    Large clouds are cast in chunks on up to `workers` processes (default: the host's cores, at most 32); the result
    does not depend on the chunking."""
    import os
    per = n // rings
    n = per * rings
    if workers is None:
        workers = min(32, os.cpu_count() or 1) if n >= 1_000_000 else 1
    nchunk = max(1, min(4 * workers, n // 65536)) if workers > 1 else 1
    edges = [n * k // nchunk for k in range(nchunk + 1)]

    def sweep(sd, phase):
        jobs = [(sd, phase, per, rings, edges[k], edges[k + 1]) for k in range(nchunk)]
        parts = _c5_run_jobs(jobs, workers) if workers > 1 and nchunk > 1 else [_c5_sweep_chunk(j) for j in jobs]
        return np.concatenate([p for p, _ in parts]), np.concatenate([q for _, q in parts])

    p, nrm = sweep(seed, 0.0)
    p2, n2 = sweep(seed + 1, 0.5)
    R = rotation(2.0, (1, 1, 5))
    t = np.array([0.003, -0.002, 0.001])
    return dict(src=p.astype(np.float32), src_n=nrm.astype(np.float32), tgt=(p2 @ R.T + t).astype(np.float32),
                tgt_n=(n2 @ R.T).astype(np.float32), truth=rigid4(R, t))


def perturbed(xyz, nrm, deg=15.0, axis=(0.3, 0.5, 0.8), t=(1.0, -2.0, 0.5)):
    """C2 substitute: target = R(deg, axis) * cloud + t, same row order."""
    R = rotation(deg, axis)
    tt = np.asarray(t, np.float64)
    return dict(src=np.asarray(xyz, np.float32), src_n=np.asarray(nrm, np.float32),
                tgt=(np.asarray(xyz, np.float64) @ R.T + tt).astype(np.float32),
                tgt_n=(np.asarray(nrm, np.float64) @ R.T).astype(np.float32), truth=rigid4(R, tt))

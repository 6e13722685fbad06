"""CPU tests: pin the oracle against the known answers the reference's own files imply
(SURVEY.md 8(c)), and check its internal consistency.  No GPU needed."""
import os

import numpy as np
import pytest

from conftest import GOLDEN


def test_pin_cat_out_is_rz45_plus_tx(cat):
    # generator of the fixture: reference ICP/main.cpp:43-52, ICP/matrix-transform.cpp:82-114
    c, s = np.cos(np.pi / 4), np.sin(np.pi / 4)
    R = np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]])
    resid = np.abs(cat["src"].astype(np.float64) @ R.T + [2.5, 0, 0] - cat["tgt"]).max()
    assert resid < 1e-5


def test_pin_initial_diff(oracle, cat):
    # evalDiff(cat, cat_out), func.cpp:19-32 -> 99242.67
    assert abs(oracle.eval_diff_f64(cat["src"], cat["tgt"]) - 99242.67) < 0.05
    # the reference's serial fp32 accumulation lands within fp32 noise of it
    assert abs(oracle.eval_diff(cat["src"], cat["tgt"]) - 99242.67) < 1.0


def test_pin_bunny_equals_za_and_is_collinear(bunny):
    za = np.loadtxt(os.path.join(GOLDEN, "za.txt"))
    assert np.abs(za[:, :3] - bunny).max() < 1e-6
    sv = np.linalg.svd(bunny - bunny.mean(0), compute_uv=False)
    assert sv[1] / sv[0] < 1e-3       # 93 collinear points


def test_pcd_reader_headers(oracle):
    xyz, nrm = oracle.pcd_read(os.path.join(GOLDEN, "cat.pcd"))
    assert xyz.shape == (3400, 3) and nrm is None
    xyz2, nrm2 = oracle.pcd_read(os.path.join(GOLDEN, "cat_out.pcd"))     # 8 fields incl. TYPE U label
    assert xyz2.shape == (3400, 3) and nrm2 is not None and np.all(nrm2 == 0)
    assert abs(xyz[0, 0] - (-16.77668190)) < 1e-6 and abs(xyz2[0, 0] - (-24.415962)) < 1e-6


def test_paper_identity_recovers_truth_in_one_iteration(oracle, cat):
    r = oracle.align(cat["src"], cat["src_n"], cat["tgt"], cat["tgt_n"], mode=oracle.MODE_PAPER, corr=oracle.CORR_IDENTITY)
    c, s = np.cos(np.pi / 4), np.sin(np.pi / 4)
    T = np.array([[c, -s, 0, 2.5], [s, c, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1]])
    assert r["status"] == 0 and r["iters"] == 1
    assert np.abs(r["transform"] - T).max() < 1e-4


def test_paper_nn_recovers_truth(oracle, cat):
    r = oracle.align(cat["src"], cat["src_n"], cat["tgt"], cat["tgt_n"], mode=oracle.MODE_PAPER, corr=oracle.CORR_BRUTE, max_iters=30)
    c, s = np.cos(np.pi / 4), np.sin(np.pi / 4)
    T = np.array([[c, -s, 0, 2.5], [s, c, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1]])
    assert r["status"] == 0 and r["iters"] <= 6
    assert np.abs(r["transform"] - T).max() < 1e-4


def test_quirks_matches_the_numpy_emulation(oracle, cat):
    """tests/golden/emulation_cat.npz is a SECOND, independent restatement of func.cpp:43-121 + myicp.cpp:117-142 (plain numpy:
    np.linalg.svd for solveLLS, the five factors of func.cpp:95-99 as explicit 4x4 products; generator committed:
    tests/golden/make_emulation.py).  The C oracle must reproduce its whole trajectory: this is what checks the reading of the
    Eigen semantics (post-multiplying translate / rotate, a-then-t alternation) that oracle and HIP path share."""
    e = np.load(os.path.join(GOLDEN, "emulation_cat.npz"))
    for solve in (oracle.SOLVE_GRAM, oracle.SOLVE_LITERAL):
        r = oracle.align(cat["src"], cat["src_n"], cat["tgt"], cat["tgt_n"], solve=solve)
        assert r["iters"] == int(e["quirks_iters"]) == 10
        np.testing.assert_allclose(np.r_[r["diffs"][:10], r["diff_final"]], e["quirks_diffs"], rtol=2e-5)
        # ten non-converging steps amplify the last bits of each route (LAPACK's SVD there, Jacobi / Cholesky here)
        assert np.abs(r["transform"] - e["quirks_T"]).max() < 3e-4
    S = oracle.reduce40(cat["src"], cat["src_n"], cat["tgt"], cat["tgt_n"])
    st, pb, qb, a, t, rc = oracle.solve_quirks_gram(S)
    np.testing.assert_allclose(a, e["quirks_a"][0], atol=2e-6)
    np.testing.assert_allclose(t, e["quirks_t"][0], atol=5e-5)
    assert abs(np.degrees(np.arctan(np.linalg.norm(a))) - 22.5) < 0.1          # half of the 45 degrees: the symmetric form applies R twice
    # first increment as a 4x4 (func.cpp:95-99): composition order and AngleAxis
    X = oracle.compose_quirks(pb, qb, a, t)
    assert np.abs(X - e["quirks_increments"][0]).max() < 2e-5


def test_normals_match_the_numpy_emulation(oracle, cat):
    """k = 10 PCA normals (myicp.cpp:152-172) against scipy's k-d tree + numpy eigh from the same generator script.
    (PCL's own rounding stays unpinned: neither restatement is PCL.)"""
    e = np.load(os.path.join(GOLDEN, "emulation_cat.npz"))
    for mine, ref in ((cat["src_n"], e["src_n_numpy"]), (cat["tgt_n"], e["tgt_n_numpy"])):
        dot = np.einsum("ij,ij->i", mine.astype(np.float64), ref.astype(np.float64))
        assert (dot > 0).all()                      # same orientation towards the viewpoint
        assert (dot > 0.999).mean() > 0.995         # ties among the 10 neighbours and near-isotropic patches differ
        assert dot.min() > 0.95


def test_oracle_under_address_sanitizer(tmp_path):
    """oracle/Makefile `sanitize`: the oracle's entry points (PCD reader, normals, both NN searches, every mode x pairing of the
    loop, the literal SVD route) on the reference's cat pair under AddressSanitizer + UBSan."""
    import shutil
    import subprocess
    from conftest import ROOT
    if shutil.which("gcc") is None and shutil.which("cc") is None:
        pytest.skip("no C compiler")
    od = os.path.join(ROOT, "oracle")
    r = subprocess.run(["make", "-C", od, "sanitize"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    r = subprocess.run([os.path.join(od, "oracle_sanitize"), os.path.join(GOLDEN, "cat.pcd"), os.path.join(GOLDEN, "cat_out.pcd")],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "oracle_sanitize: ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr


def test_gram_route_matches_literal_svd_route(oracle, cat):
    # func.cpp:64-73 (two N x 3 thin-SVD least squares in fp32) vs the 3x3 normal-equation blocks
    S = oracle.reduce40(cat["src"], cat["src_n"], cat["tgt"], cat["tgt_n"])
    _, pb, qb, a, t, _ = oracle.solve_quirks_gram(S)
    _, pb2, qb2, a2, t2 = oracle.solve_quirks_literal(cat["src"], cat["src_n"], cat["tgt"], cat["tgt_n"])
    assert np.abs(a - a2).max() / np.abs(a).max() < 5e-6
    assert np.abs(t - t2).max() / np.abs(t).max() < 2e-5
    g = cat["golden"]
    # over the whole 10-iteration run the two routes stay within 2e-4 of each other on the 4x4
    assert np.abs(g["quirks_identity_T"] - g["quirks_identity_literal_T"]).max() < 2e-4


def test_rows_match_numpy(oracle, cat):
    M, N, c = oracle.rows(cat["src"], cat["src_n"], cat["tgt"], cat["tgt_n"])
    n = cat["src_n"] + cat["tgt_n"]
    np.testing.assert_allclose(M, np.cross(cat["src"] + cat["tgt"], n), rtol=1e-5, atol=1e-3)
    np.testing.assert_allclose(N, n, rtol=0, atol=0)
    np.testing.assert_allclose(c, np.einsum("ij,ij->i", cat["src"] - cat["tgt"], n), rtol=1e-5, atol=1e-3)


def test_nn_grid_equals_brute_bitwise(oracle, cat):
    rng = np.random.default_rng(5)
    X = np.eye(4, dtype=np.float32)
    X[:3, 3] = [3.0, -2.0, 1.0]
    for q, p in ((cat["tgt"], cat["src"]), (rng.random((5000, 3), dtype=np.float32), rng.random((3000, 3), dtype=np.float32) * 1.5 - 0.25)):
        i1, d1 = oracle.nn_brute(p, q, X)
        i2, d2 = oracle.nn_grid(p, q, X)
        assert np.array_equal(i1, i2) and np.array_equal(d1, d2)


def test_nn_ties_pick_lowest_index(oracle):
    q = np.array([[0, 0, 0], [2, 0, 0], [2, 0, 0], [0, 0, 0]], np.float32)
    p = np.array([[1, 0, 0], [2, 0, 0], [-1, 0, 0]], np.float32)
    for fn in (oracle.nn_brute, oracle.nn_grid):
        idx, _ = fn(p, q)
        assert list(idx) == [0, 1, 0]


def test_golden_file_is_current(oracle, cat):
    g = cat["golden"]
    n, _ = oracle.normals_knn(cat["src"], 10)
    assert np.array_equal(n, g["src_n"])
    r = oracle.align(cat["src"], cat["src_n"], cat["tgt"], cat["tgt_n"])
    assert np.array_equal(r["transform"], g["quirks_identity_T"])


def test_normals_are_unit_and_face_origin(cat):
    for xyz, n in ((cat["src"], cat["src_n"]), (cat["tgt"], cat["tgt_n"])):
        assert np.abs(np.linalg.norm(n, axis=1) - 1).max() < 1e-5
        assert (np.einsum("ij,ij->i", -xyz, n) >= -1e-6).all()     # (vp - p) . n >= 0, vp = origin


def test_degenerate_inputs_are_flagged(oracle, bunny):
    # collinear cloud: normals undefined, M^T M / N^T N rank deficient -> must flag, not NaN
    n = np.tile(np.array([[0, 0, 1]], np.float32), (bunny.shape[0], 1))
    tgt = bunny + np.array([0.01, 0.02, 0.0], np.float32)
    for mode in (oracle.MODE_QUIRKS, oracle.MODE_PAPER):
        r = oracle.align(bunny, n, tgt, n, mode=mode, diff_threshold=0.0)
        assert r["status"] == oracle.ERR_DEGENERATE
        assert np.isfinite(r["transform"]).all()
    # N_s != N_t with identity pairing (func.cpp:21 assert)
    r = oracle.align(bunny[:50], n[:50], tgt, n)
    assert r["status"] == oracle.ERR_SIZE


def test_incremental_vs_cumulative_apply_drift(oracle, cat):
    a = oracle.align(cat["src"], cat["src_n"], cat["tgt"], cat["tgt_n"], mode=oracle.MODE_PAPER, corr=oracle.CORR_BRUTE,
                     max_iters=30, apply_mode=oracle.APPLY_INCREMENTAL)
    b = oracle.align(cat["src"], cat["src_n"], cat["tgt"], cat["tgt_n"], mode=oracle.MODE_PAPER, corr=oracle.CORR_BRUTE,
                     max_iters=30, apply_mode=oracle.APPLY_CUMULATIVE)
    assert np.abs(a["transform"] - b["transform"]).max() < 1e-4


def test_increment_stop_and_normal_filter_in_oracle(oracle, cat):
    full = oracle.align(cat["src"], cat["src_n"], cat["tgt"], cat["tgt_n"], mode=oracle.MODE_PAPER, corr=oracle.CORR_BRUTE,
                        max_iters=30, diff_threshold=0.0)
    early = oracle.align(cat["src"], cat["src_n"], cat["tgt"], cat["tgt_n"], mode=oracle.MODE_PAPER, corr=oracle.CORR_BRUTE,
                         max_iters=30, diff_threshold=0.0, eps_rotation=1e-5, eps_translation=1e-4)
    assert full["iters"] == 30 and 4 <= early["iters"] < 30
    assert np.abs(full["transform"] - early["transform"]).max() < 1e-4
    sn = cat["src_n"].copy(); sn[::2] *= -1
    S = oracle.reduce40(cat["src"], sn, cat["tgt"], cat["tgt_n"], min_ndot=0.0)
    S0 = oracle.reduce40(cat["src"], sn, cat["tgt"], cat["tgt_n"])
    assert S0[34] == 3400 and 0 < S[34] < 3400


def test_pin_kabsch_on_the_reference_pair(oracle, cat):
    """regist.h:8-72 (registrateNPoint) on the index-paired cat clouds -- what ICP/register-test.cpp:55-66 prints --
    must return the generating motion Rz(45 deg), (2.5, 0, 0)."""
    R, T, st = oracle.kabsch(cat["src"], cat["tgt"])
    c, s = np.cos(np.pi / 4), np.sin(np.pi / 4)
    assert st == 0
    assert np.abs(R - [[c, -s, 0], [s, c, 0], [0, 0, 1]]).max() < 1e-6
    assert np.abs(T - [2.5, 0, 0]).max() < 1e-5
    assert abs(np.linalg.det(R) - 1) < 1e-9
    # numpy cross-check of the SVD route
    ps = cat["src"].astype(np.float64); pd = cat["tgt"].astype(np.float64)
    H = (ps - ps.mean(0)).T @ (pd - pd.mean(0))
    U, W, Vt = np.linalg.svd(H)
    Rn = Vt.T @ np.diag([1, 1, np.linalg.det(Vt.T @ U.T)]) @ U.T
    assert np.abs(R - Rn).max() < 1e-9
    # reflection case: mirrored target still yields a proper rotation
    R2, _, st2 = oracle.kabsch(cat["src"], cat["tgt"] * np.float32([1, 1, -1]))
    assert st2 == 0 and abs(np.linalg.det(R2) - 1) < 1e-9


def test_p2p_icp_in_oracle(oracle, cat):
    r = oracle.align(cat["src"], cat["src_n"], cat["tgt"], cat["tgt_n"], mode=oracle.MODE_P2P, corr=oracle.CORR_IDENTITY)
    assert r["status"] == 0 and r["iters"] == 1
    c, s = np.cos(np.pi / 4), np.sin(np.pi / 4)
    T = np.array([[c, -s, 0, 2.5], [s, c, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1]])
    assert np.abs(r["transform"] - T).max() < 1e-5
    r = oracle.align(cat["src"], cat["src_n"], cat["tgt"], cat["tgt_n"], mode=oracle.MODE_P2P, corr=oracle.CORR_BRUTE, max_iters=60)
    assert r["status"] == 0 and np.abs(r["transform"] - T).max() < 1e-4

"""The reference's stdout block (ICP/myicp.cpp:146-149) byte for byte: symmicp_format_result.  Needs no GPU.

The reference prints `transform.matrix()`, `transform.rotation()` and `transform.translation()` through Eigen's default
IOFormat (Eigen is not under /root/reference; its published print_matrix: stream precision 6, every coefficient right-aligned to the
widest coefficient of the matrix, one space between columns).  The reference holds no captured output, so the pins are (1) the
data-level known answer of its own fixture pair -- cat_out = Rz(45 deg) cat + (2.5, 0, 0), ICP/main.cpp:43-52 -- written out by hand
below, and (2) an independent Python statement of the same format on random matrices.
"""
import numpy as np


GOLDEN_RZ45 = (
    "Result transform:\n"
    " 0.707107 -0.707107         0       2.5\n"
    " 0.707107  0.707107         0         0\n"
    "        0         0         1         0\n"
    "        0         0         0         1\n"
    "  rotation:\n"
    " 0.707107 -0.707107         0\n"
    " 0.707107  0.707107         0\n"
    "        0         0         1\n"
    "  translation:\n"
    "2.5\n"
    "  0\n"
    "  0\n")


def eigen_block(M):
    """Eigen::operator<<(ostream, matrix) with the default IOFormat, restated in Python"""
    cells = [["%g" % float(v) for v in row] for row in np.atleast_2d(M)]
    w = max(len(c) for row in cells for c in row)
    return "".join(" ".join(c.rjust(w) for c in row) + "\n" for row in cells)


def test_known_answer_of_the_reference_fixture_pair():
    import symmicp
    c = np.float32(np.cos(np.pi / 4))
    T = np.array([[c, -c, 0, 2.5], [c, c, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1]], np.float32)
    assert symmicp.format_result(T) == GOLDEN_RZ45


def test_format_matches_an_independent_statement_on_random_rigid_transforms():
    import symmicp
    from symmicp import synth
    rng = np.random.default_rng(7)
    for _ in range(50):
        R = synth.rotation(float(rng.uniform(0, 180)), rng.standard_normal(3))
        t = rng.standard_normal(3) * 10.0 ** rng.integers(-6, 6)
        T = synth.rigid4(R, t).astype(np.float32)
        want = "Result transform:\n" + eigen_block(T) + "  rotation:\n" + eigen_block(T[:3, :3]) + "  translation:\n" + eigen_block(T[:3, 3:4])
        got = symmicp.format_result(T)
        # (rotation() is the polar factor of the linear part: for a rotation matrix rounded to fp32 it is that matrix to within an ulp,
        # which %g at 6 digits can still show: compare the matrix and translation blocks exactly, the rotation block numerically)
        gl, wl = got.split("\n"), want.split("\n")
        assert gl[:6] == wl[:6] and gl[9:] == wl[9:], (got, want)
        Rg = np.array([[float(v) for v in l.split()] for l in gl[6:9]])
        assert np.abs(Rg - T[:3, :3]).max() < 2e-6
        assert len({len(l) for l in gl[6:9]}) == 1            # aligned columns


def test_format_query_and_truncation():
    import ctypes as C
    import symmicp
    L = symmicp.lib()
    X = np.eye(4, dtype=np.float32).reshape(16)
    n = L.symmicp_format_result(X.ctypes.data_as(C.POINTER(C.c_float)), None, 0)
    assert n == len(symmicp.format_result(np.eye(4)))
    buf = C.create_string_buffer(8)
    assert L.symmicp_format_result(X.ctypes.data_as(C.POINTER(C.c_float)), buf, 8) == n and buf.value == b"Result "

import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "icp-symm_amd", "py")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.lib()
    return O


@pytest.fixture(scope="session")
def cat(oracle):
    """the reference's own fixture pair (ICP/main.cpp:8) + k=10 PCA normals from the oracle"""
    src, _ = oracle.pcd_read(os.path.join(GOLDEN, "cat.pcd"))
    tgt, _ = oracle.pcd_read(os.path.join(GOLDEN, "cat_out.pcd"))
    g = np.load(os.path.join(GOLDEN, "cat_golden.npz"))
    return dict(src=src, tgt=tgt, src_n=g["src_n"], tgt_n=g["tgt_n"], golden=g)


@pytest.fixture(scope="session")
def bunny(oracle):
    xyz, _ = oracle.pcd_read(os.path.join(GOLDEN, "txt2pcd_bunny1.pcd"))
    return xyz

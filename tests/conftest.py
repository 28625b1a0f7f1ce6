import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


@pytest.fixture(scope="session")
def oracle_backend():
    """numpy restatement of the reference backend (test infrastructure)"""
    from oracle.np_backend import NumpyBackend
    return NumpyBackend()


@pytest.fixture(scope="session")
def hip():
    """the product backend; fails loudly (no fallback) when the library or the GPU is missing"""
    from indigo_amd.backends import get_backend
    return get_backend("hip")


def rel_err(act, exp):
    """||act - exp||_2 / ||exp||_2 (0 when both vanish)"""
    act = np.asarray(act)
    exp = np.asarray(exp)
    den = np.linalg.norm(exp.ravel())
    num = np.linalg.norm((act - exp).ravel())
    return 0.0 if num == 0 else num / den if den else np.inf


def csr_from(gold, prefix):
    import scipy.sparse as spp
    shape = tuple(int(s) for s in gold[prefix + "shape"])
    return spp.csr_matrix((gold[prefix + "data"], gold[prefix + "indices"], gold[prefix + "indptr"]), shape=shape)

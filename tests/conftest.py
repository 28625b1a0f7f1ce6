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


def check_misc_leaves(B, rtol):
    """onemm / cdiamm / cgemm / csymm / apgd of backend B against vectors captured from the reference (leaf_misc.npz)"""
    import scipy.sparse as spp
    g = golden("leaf_misc")
    c64 = np.dtype('complex64')
    for i in range(int(g["one_count"])):
        alpha, beta = g["one%d_ab" % i]
        y_d = B.copy_array(g["one%d_y" % i])
        B.onemm(y_d, B.copy_array(g["one%d_x" % i]), alpha, beta)
        assert rel_err(y_d.to_host(), g["one%d_out" % i]) < rtol, ("onemm", i)
    for i in range(int(g["dia_count"])):
        alpha, beta = g["dia%d_ab" % i]
        M, K = (int(v) for v in g["dia%d_shape" % i])
        A = spp.dia_matrix((g["dia%d_data" % i], g["dia%d_offsets" % i]), shape=(M, K))
        A_d = B.dia_matrix(B, A)
        y_d = B.copy_array(g["dia%d_y" % i])
        A_d.forward(y_d, B.copy_array(g["dia%d_x" % i]), alpha=alpha, beta=beta)
        assert rel_err(y_d.to_host(), g["dia%d_fwd" % i]) < rtol, ("cdiamm forward", i)
        x_d = B.copy_array(g["dia%d_xa" % i])
        A_d.adjoint(x_d, B.copy_array(g["dia%d_ya" % i]), alpha=alpha, beta=beta)
        assert rel_err(x_d.to_host(), g["dia%d_adj" % i]) < rtol, ("cdiamm adjoint", i)
    for i in range(int(g["gemm_count"])):
        alpha, beta, fwd = g["gemm%d_abf" % i]
        y_d = B.copy_array(g["gemm%d_y" % i])
        B.cgemm(y_d, B.copy_array(g["gemm%d_M" % i]), B.copy_array(g["gemm%d_x" % i]), alpha, beta, forward=bool(fwd.real))
        assert rel_err(y_d.to_host(), g["gemm%d_out" % i]) < rtol, ("cgemm", i)
    for i in range(int(g["symm_count"])):
        alpha, beta, left = g["symm%d_abl" % i]
        y_d = B.copy_array(g["symm%d_y" % i])
        B.csymm(y_d, B.copy_array(g["symm%d_M" % i]), B.copy_array(g["symm%d_x" % i]), alpha, beta, bool(left.real))
        assert rel_err(y_d.to_host(), g["symm%d_out" % i]) < rtol, ("csymm", i)
    # apgd (reference backend.py:691-732): gradient of 0.5*||D x - b||^2, prox = clamp from below
    d, b = g["apgd_d"], g["apgd_b"]
    Dop = B.Diag(d)
    b_d = B.copy_array(b)
    tmp = B.zero_array(b.shape, c64)

    def gradf(gf, xk):
        Dop.eval(tmp, xk)
        B.axpby(1, tmp, -1, b_d)
        Dop.H.eval(gf, tmp)

    def proxg(xk, alpha):
        B.max(0.2, xk)
    for it in (1, 2, 5):
        x0 = g["apgd_x0"].copy(order='F')
        B.apgd(gradf, proxg, 0.4, x0, maxiter=it)
        assert rel_err(x0, g["apgd_it%d" % it]) < 10 * rtol, ("apgd", it)

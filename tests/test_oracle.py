"""Pins the oracle: the numpy restatement (oracle/np_backend.py), the C restatement
(oracle/csrmm_oracle.c) and -- where built -- the reference's own compiled C
(oracle/_ref) must reproduce the golden vectors captured from the reference
(tests/golden/make_golden.py).  CPU only.
"""
import numpy as np
import pytest
import scipy.sparse as spp

from conftest import csr_from, golden, rel_err
from oracle import native

C64 = np.dtype('complex64')
TOL = 2e-6      # same arithmetic, same libraries: only summation-order noise is allowed


def test_blas_leaves(oracle_backend):
    B, g = oracle_backend, golden("leaf_blas")
    for i in range(int(g["count"])):
        x, y = g["axpby%d_x" % i], g["axpby%d_y" % i]
        alpha, beta = g["axpby%d_ab" % i]
        beta = beta.real
        y_d = B.copy_array(y)
        B.axpby(beta, y_d, alpha, B.copy_array(x))
        assert rel_err(y_d.to_host(), g["axpby%d_out" % i]) < TOL
        s_d = B.copy_array(x)
        B.scale(s_d, alpha)
        assert rel_err(s_d.to_host(), g["scale%d_out" % i]) < TOL
        assert abs(B.dot(B.copy_array(x), B.copy_array(y)) - float(g["dot%d" % i])) < 1e-4
        assert abs(B.norm2(B.copy_array(x)) - float(g["nrm%d" % i])) < 1e-4
        m_d = B.copy_array(x)
        B.max(0.5, m_d)
        np.testing.assert_array_equal(m_d.to_host(), g["max%d_out" % i])


def _csrmm_cases():
    g = golden("leaf_csrmm")
    return g, range(int(g["count"]))


@pytest.mark.parametrize("which", ["numpy", "c", "ref"])
def test_csrmm_leaves(oracle_backend, which):
    B = oracle_backend
    g, cases = _csrmm_cases()
    if which == "ref" and native.ref_native() is None:
        pytest.skip("oracle/_ref not built in this checkout")
    for i in cases:
        p = "c%d_" % i
        A = csr_from(g, p)
        alpha, beta = g[p + "ab"]
        x, y, xa, ya = g[p + "x"], g[p + "y"], g[p + "xa"], g[p + "ya"]
        exw = bool(g[p + "inspect"][2])
        if which == "numpy":
            A_d = B.csr_matrix(B, A)
            assert A_d._exwrite == exw
            assert abs(A_d._row_frac - g[p + "inspect"][0]) < 1e-12 and abs(A_d._col_frac - g[p + "inspect"][1]) < 1e-12
            y_d = B.copy_array(y)
            A_d.forward(y_d, B.copy_array(x), alpha=alpha, beta=beta)
            fwd = y_d.to_host()
            ya_d = B.copy_array(ya)
            A_d.adjoint(ya_d, B.copy_array(xa), alpha=alpha, beta=beta)
            adj = ya_d.to_host()
        elif which == "c":
            assert native.c_inspect(A)[2] == exw
            fwd = native.c_ccsrmm(A, x, y.copy(order='F'), alpha, beta, adjoint=False)
            adj = native.c_ccsrmm(A, xa, ya.copy(order='F'), alpha, beta, adjoint=True)
        else:
            adj = native.ref_ccsrmm(A, xa, ya.copy(order='F'), alpha, beta, adjoint=True, exwrite=exw)
            fwd = None
            if x.shape[1] > 1:       # the reference's one-column forward branch needs MKL
                fwd = native.ref_ccsrmm(A, x, y.copy(order='F'), alpha, beta, adjoint=False)
        if fwd is not None:
            assert rel_err(fwd, g[p + "fwd"]) < TOL, (which, i)
        assert rel_err(adj, g[p + "adj"]) < TOL, (which, i)


def test_fft_leaves(oracle_backend):
    B, g = oracle_backend, golden("leaf_fft")
    for i in range(int(g["count"])):
        x = g["f%d_x" % i]
        y_d = B.zero_array(x.shape, C64)
        B.fftn(y_d, B.copy_array(x))
        assert rel_err(y_d.to_host(), g["f%d_fwd" % i]) < TOL
        B.ifftn(y_d, B.copy_array(x))
        assert rel_err(y_d.to_host(), g["f%d_inv" % i]) < TOL


def test_composites(oracle_backend):
    B, g = oracle_backend, golden("composites")
    # Product with alpha, beta
    P = B.SpMatrix(csr_from(g, "prod_A0_"), name='A0') * B.SpMatrix(csr_from(g, "prod_A1_"), name='A1')
    y_d = B.copy_array(g["prod_y"])
    P.eval(y_d, B.copy_array(g["prod_x"]), alpha=0.5, beta=1.0)
    assert rel_err(y_d.to_host(), g["prod_fwd"]) < TOL
    y_d = B.copy_array(g["prod_ya"])
    P.H.eval(y_d, B.copy_array(g["prod_xa"]), alpha=0.5, beta=1.0)
    assert rel_err(y_d.to_host(), g["prod_adj"]) < TOL
    # nested KronI
    Kn = B.KronI(6, B.KronI(4, B.SpMatrix(csr_from(g, "kron_A_"))))
    v_d = B.copy_array(g["kron_v"])
    Kn.eval(v_d, B.copy_array(g["kron_u"]))
    assert rel_err(v_d.to_host(), g["kron_fwd"]) < TOL
    u_d = B.copy_array(g["kron_u"])
    Kn.H.eval(u_d, B.copy_array(g["kron_v"]))
    assert rel_err(u_d.to_host(), g["kron_adj"]) < TOL
    # VStack / BlockDiag
    mats = [csr_from(g, "stack_A%d_" % j) for j in range(3)]
    V = B.VStack([B.SpMatrix(m) for m in mats])
    y_d = B.copy_array(g["vs_y"])
    V.eval(y_d, B.copy_array(g["vs_x"]), alpha=0.5, beta=0.5)
    assert rel_err(y_d.to_host(), g["vs_fwd"]) < TOL
    y_d = B.copy_array(g["vs_ya"])
    V.H.eval(y_d, B.copy_array(g["vs_xa"]), alpha=0.5, beta=0.5)
    assert rel_err(y_d.to_host(), g["vs_adj"]) < TOL
    D = B.BlockDiag([B.SpMatrix(m) for m in mats])
    y_d = B.copy_array(g["bd_y"])
    D.eval(y_d, B.copy_array(g["bd_x"]), alpha=1.0, beta=0.5)
    assert rel_err(y_d.to_host(), g["bd_fwd"]) < TOL
    y_d = B.copy_array(g["bd_ya"])
    D.H.eval(y_d, B.copy_array(g["bd_xa"]), alpha=1.0, beta=0.5)
    assert rel_err(y_d.to_host(), g["bd_adj"]) < TOL
    # Sum / Scale (conjugated on the adjoint) / Eye
    Sm = (2 - 1j) * B.SpMatrix(csr_from(g, "sum_S0_")) + B.SpMatrix(csr_from(g, "sum_S1_")) - 0.5 * B.Eye(6)
    y_d = B.zero_array(g["sum_x"].shape, C64)
    Sm.eval(y_d, B.copy_array(g["sum_x"]))
    assert rel_err(y_d.to_host(), g["sum_fwd"]) < TOL
    Sm.H.eval(y_d, B.copy_array(g["sum_x"]))
    assert rel_err(y_d.to_host(), g["sum_adj"]) < TOL
    # centred unitary FFT
    Fc = B.FFTc(tuple(int(s) for s in g["fftc_shape"]), dtype=C64)
    y_d = B.zero_array(g["fftc_x"].shape, C64)
    Fc.eval(y_d, B.copy_array(g["fftc_x"]))
    assert rel_err(y_d.to_host(), g["fftc_fwd"]) < TOL
    Fc.H.eval(y_d, B.copy_array(g["fftc_x"]))
    assert rel_err(y_d.to_host(), g["fftc_adj"]) < TOL


def test_native_oracles_agree_with_scipy():
    rng = np.random.default_rng(7)
    A = spp.random(200, 150, 0.05, format='csr', dtype=np.float32, random_state=rng).astype(C64)
    X = (rng.random((150, 8)) + 1j * rng.random((150, 8))).astype(C64, order='F')
    Y = np.zeros((200, 8), dtype=C64, order='F')
    native.c_ccsrmm(A, X, Y)
    assert rel_err(Y, A @ X) < 1e-6
    if native.ref_native() is not None:
        Y2 = np.zeros_like(Y, order='F')
        native.ref_ccsrmm(A, X, Y2)
        assert rel_err(Y2, A @ X) < 1e-6
        nzr, nzc, exw = native.ref_native().inspect(A.shape[0], A.shape[1], A.indices, A.indptr)
        assert (nzr, nzc, bool(exw)) == native.c_inspect(A)


def test_misc_leaves_against_reference(oracle_backend):
    """onemm, cdiamm, cgemm, csymm, apgd of the oracle == the reference's numpy backend"""
    from conftest import check_misc_leaves
    check_misc_leaves(oracle_backend, 2e-6)


def test_double_precision_arbiter_against_reference_goldens():
    """oracle/precise.py (the complex128 arbiter of the full-size GPU tests) evaluates the reference's operator: its one-coil
    A_c, A_c^H and A_c^H A_c, summed over the coils, reproduce what the reference computed from the same inputs (sense.npz:
    A x, A^H k, A^H A x + lamda x; and, with unit maps, the NUFFT leaf alone)."""
    from indigo_amd.sense import SenseProblem
    from oracle.precise import CoilOperatorF64
    g = golden("sense")
    C, width, ntab, osf, ro, tr = g["params"]
    N = tuple(int(n) for n in g["N"])
    p = SenseProblem(N, g["coord"], np.asfortranarray(g["maps"]), width=int(width), ntable=int(ntab), oversamp=float(osf))
    T, C = p.T, p.C
    ops = [CoilOperatorF64(p, c) for c in range(C)]
    x, k = g["sense_x"][:, 0], g["sense_k"][:, 0]
    Ax = np.concatenate([o.forward(x) for o in ops])                          # coil-major rows, as KronI(C, .) stacks them
    assert rel_err(Ax, g["sense_Ax"][:, 0]) < TOL
    AHk = sum(o.adjoint(k[c * T:(c + 1) * T]) for c, o in enumerate(ops))
    assert rel_err(AHk, g["sense_AHk"][:, 0]) < TOL
    AHAx = sum(o.normal(x) for o in ops) + float(g["lamda"]) * x
    assert rel_err(AHAx, g["sense_AHAx"][:, 0]) < TOL
    # ... and the same through the (x, z, y)-ordered gridding matrix a problem already holds (what bench.py's dense-trajectory
    # leg evaluates with: no second copy of a 4e8-nonzero matrix)
    p2 = SenseProblem(N, g["coord"], np.asfortranarray(g["maps"]), width=int(width), ntable=int(ntab), oversamp=float(osf))
    p2.fused_interp(1)
    ops2 = [CoilOperatorF64(p2, c) for c in range(C)]
    assert all(o.layout == 1 for o in ops2) and all(o.layout == 0 for o in ops)
    assert rel_err(sum(o.normal(x) for o in ops2) + float(g["lamda"]) * x, g["sense_AHAx"][:, 0]) < TOL
    # the NUFFT leaf: unit maps
    q = SenseProblem(N, g["coord"], np.ones(N + (1,), dtype=C64, order='F'), width=int(width), ntable=int(ntab), oversamp=float(osf))
    F = CoilOperatorF64(q, 0)
    for j in range(g["nufft_x"].shape[1]):
        assert rel_err(F.forward(g["nufft_x"][:, j]), g["nufft_fwd"][:, j]) < TOL
        assert rel_err(F.adjoint(g["nufft_k"][:, j]), g["nufft_adj"][:, j]) < TOL

"""GPU parity tests for composed operator trees and the SENSE path on HipBackend.

Golden vectors from the reference (tests/golden), the numpy oracle on identical
seeded inputs, and size-independent properties (adjointness, linearity,
normal-operator symmetry) at sizes the oracle cannot finish quickly.
"""
import itertools

import numpy as np
import pytest
import scipy.sparse as spp

from conftest import csr_from, golden, rel_err
from indigo_amd.sense import SenseProblem, normal_operator
from indigo_amd.transforms import reserve_for
from indigo_amd.util import rand64c, randM, Trace

pytestmark = pytest.mark.gpu
C64 = np.dtype('complex64')
RTOL = 1e-5


def _eval(op, backend, y0, x, **kw):
    y_d = backend.copy_array(y0)
    op.eval(y_d, backend.copy_array(x), **kw)
    return y_d.to_host()


def test_composites_golden(hip):
    B, g = hip, golden("composites")
    B._scratch = None
    P = B.SpMatrix(csr_from(g, "prod_A0_"), name='A0') * B.SpMatrix(csr_from(g, "prod_A1_"), name='A1')
    assert rel_err(_eval(P, B, g["prod_y"], g["prod_x"], alpha=0.5, beta=1.0), g["prod_fwd"]) < RTOL
    assert rel_err(_eval(P.H, B, g["prod_ya"], g["prod_xa"], alpha=0.5, beta=1.0), g["prod_adj"]) < RTOL
    Kn = B.KronI(6, B.KronI(4, B.SpMatrix(csr_from(g, "kron_A_"))))
    assert rel_err(_eval(Kn, B, g["kron_v"], g["kron_u"]), g["kron_fwd"]) < RTOL
    assert rel_err(_eval(Kn.H, B, g["kron_u"], g["kron_v"]), g["kron_adj"]) < RTOL
    mats = [csr_from(g, "stack_A%d_" % j) for j in range(3)]
    V = B.VStack([B.SpMatrix(m) for m in mats])
    assert rel_err(_eval(V, B, g["vs_y"], g["vs_x"], alpha=0.5, beta=0.5), g["vs_fwd"]) < RTOL
    assert rel_err(_eval(V.H, B, g["vs_ya"], g["vs_xa"], alpha=0.5, beta=0.5), g["vs_adj"]) < RTOL
    D = B.BlockDiag([B.SpMatrix(m) for m in mats])
    assert rel_err(_eval(D, B, g["bd_y"], g["bd_x"], alpha=1.0, beta=0.5), g["bd_fwd"]) < RTOL
    assert rel_err(_eval(D.H, B, g["bd_ya"], g["bd_xa"], alpha=1.0, beta=0.5), g["bd_adj"]) < RTOL
    Sm = (2 - 1j) * B.SpMatrix(csr_from(g, "sum_S0_")) + B.SpMatrix(csr_from(g, "sum_S1_")) - 0.5 * B.Eye(6)
    z = np.zeros_like(g["sum_x"], order='F')
    assert rel_err(_eval(Sm, B, z, g["sum_x"]), g["sum_fwd"]) < RTOL
    assert rel_err(_eval(Sm.H, B, z, g["sum_x"]), g["sum_adj"]) < RTOL
    Fc = B.FFTc(tuple(int(s) for s in g["fftc_shape"]), dtype=C64)
    z = np.zeros_like(g["fftc_x"], order='F')
    assert rel_err(_eval(Fc, B, z, g["fftc_x"]), g["fftc_fwd"]) < RTOL
    assert rel_err(_eval(Fc.H, B, z, g["fftc_x"]), g["fftc_adj"]) < RTOL


@pytest.mark.parametrize("L,M,N,K,density,alpha,beta", itertools.product([3], [5, 6], [7], [1, 8, 17], [0.1, 1], [0, .5, 1], [0, .5, 1]))
def test_product_grid(hip, L, M, N, K, density, alpha, beta):
    """reference test_operators.py:55-84"""
    hip._scratch = None
    A0, A1 = randM(L, M, density, seed=1), randM(M, N, density, seed=2)
    A = hip.SpMatrix(A0, name='A0') * hip.SpMatrix(A1, name='A1')
    x, y = rand64c(N, K, seed=3), rand64c(L, K, seed=4)
    np.testing.assert_allclose(_eval(A, hip, y, x, alpha=alpha, beta=beta), beta * y + alpha * (A0 @ (A1 @ x)), rtol=RTOL, atol=1e-5)
    x, y = rand64c(L, K, seed=5), rand64c(N, K, seed=6)
    np.testing.assert_allclose(_eval(A.H, hip, y, x, alpha=alpha, beta=beta),
                               beta * y + alpha * (A1.conj().T @ (A0.conj().T @ x)), rtol=RTOL, atol=1e-5)
    assert A.shape == (L, N) and A.H.shape == (N, L) and A.dtype == C64


@pytest.mark.parametrize("stack,K,alpha,beta", itertools.product([1, 2, 3], [1, 4, 9], [0.5, 1], [0, 1, 0.5]))
def test_vstack_blockdiag_kroni_grid(hip, stack, K, alpha, beta):
    """reference test_operators.py:87-116, 151-224"""
    hip._scratch = None
    M, N = 5, 7
    mats = [randM(M, N, 0.5, seed=10 + j) for j in range(stack)]
    V, Vh = hip.VStack([hip.SpMatrix(m) for m in mats]), spp.vstack(mats)
    x, y = rand64c(N, K, seed=1), rand64c(M * stack, K, seed=2)
    np.testing.assert_allclose(_eval(V, hip, y, x, alpha=alpha, beta=beta), beta * y + alpha * (Vh @ x), rtol=RTOL, atol=1e-5)
    np.testing.assert_allclose(_eval(V.H, hip, x, y, alpha=alpha, beta=beta), beta * x + alpha * (Vh.conj().T @ y), rtol=RTOL, atol=1e-5)
    D, Dh = hip.BlockDiag([hip.SpMatrix(m) for m in mats]), spp.block_diag(mats)
    x, y = rand64c(N * stack, K, seed=3), rand64c(M * stack, K, seed=4)
    np.testing.assert_allclose(_eval(D, hip, y, x, alpha=alpha, beta=beta), beta * y + alpha * (Dh @ x), rtol=RTOL, atol=1e-5)
    np.testing.assert_allclose(_eval(D.H, hip, x, y, alpha=alpha, beta=beta), beta * x + alpha * (Dh.conj().T @ y), rtol=RTOL, atol=1e-5)
    Kr, Kh = hip.KronI(stack + 1, hip.SpMatrix(mats[0])), spp.kron(spp.eye(stack + 1), mats[0])
    x, y = rand64c(Kr.shape[1], K, seed=5), rand64c(Kr.shape[0], K, seed=6)
    np.testing.assert_allclose(_eval(Kr, hip, y, x, alpha=alpha, beta=beta), beta * y + alpha * (Kh @ x), rtol=RTOL, atol=1e-5)
    np.testing.assert_allclose(_eval(Kr.H, hip, x, y, alpha=alpha, beta=beta), beta * x + alpha * (Kh.conj().T @ y), rtol=RTOL, atol=1e-5)


@pytest.mark.parametrize("shape,B", itertools.product([(22, 23, 24), (24, 22), (23,)], [1, 3]))
def test_unscaled_and_centered_fft_operators(hip, shape, B):
    """reference test_operators.py:227-367"""
    hip._scratch = None
    n = int(np.prod(shape))
    ax = tuple(range(len(shape)))
    x = rand64c(n, B, seed=n)
    xh = x.reshape(shape + (B,), order='F')
    F = hip.UnscaledFFT(shape, dtype=C64)
    z = np.zeros_like(x, order='F')
    assert rel_err(_eval(F, hip, z, x).reshape(xh.shape, order='F'), np.fft.fftn(xh, axes=ax)) < RTOL
    assert rel_err(_eval(F.H, hip, z, x).reshape(xh.shape, order='F'), np.fft.ifftn(xh, axes=ax) * n) < RTOL
    U = hip.FFT(shape, dtype=C64)
    back = _eval(U.H, hip, z, _eval(U, hip, z, x))
    assert rel_err(back, x) < RTOL                                    # unitary
    Fc = hip.FFTc(shape, dtype=C64)
    from numpy.fft import fftshift, ifftshift, fftn, ifftn
    assert rel_err(_eval(Fc, hip, z, x).reshape(xh.shape, order='F'),
                   fftshift(fftn(ifftshift(xh, axes=ax), axes=ax, norm='ortho'), axes=ax)) < 1e-4
    assert rel_err(_eval(Fc.H, hip, z, x).reshape(xh.shape, order='F'),
                   fftshift(ifftn(ifftshift(xh, axes=ax), axes=ax, norm='ortho'), axes=ax)) < 1e-4
    with pytest.raises(AssertionError):
        F.eval(hip.copy_array(z), hip.copy_array(x), alpha=2)


def test_zpad_and_interp_operators(hip):
    """reference test_operators.py:370-432"""
    hip._scratch = None
    N, M, batch = (3, 4, 3), (5, 6, 7), 2
    u = rand64c(*N, batch, seed=1)
    Z = hip.Zpad(M, N, dtype=C64)
    v = _eval(Z, hip, np.zeros((int(np.prod(M)), batch), C64, order='F'), u.reshape(-1, batch, order='F')).reshape(M + (batch,), order='F')
    assert np.count_nonzero(v) == u.size
    np.testing.assert_array_equal(v[1:4, 1:5, 2:5, :], u)
    back = _eval(Z.H, hip, np.zeros((u.size // batch, batch), C64, order='F'), v.reshape(-1, batch, order='F'))
    np.testing.assert_array_equal(back.reshape(u.shape, order='F'), u)
    rng = np.random.default_rng(2)
    coord = rng.random((3, 6, 8)) - 0.5
    G = hip.Interp(M, coord, 3, rng.random(128), dtype=C64)
    a, b = rand64c(G.shape[1], batch, seed=3), rand64c(G.shape[0], batch, seed=4)
    Ga = _eval(G, hip, np.zeros_like(b, order='F'), a)
    GHb = _eval(G.H, hip, np.zeros_like(a, order='F'), b)
    np.testing.assert_allclose(np.vdot(b, Ga), np.vdot(GHb, a), rtol=1e-4)


@pytest.mark.parametrize("level", [0, 3, "fused"])
def test_sense_golden(hip, level):
    g = golden("sense")
    C, width, ntab, osf, ro, tr = g["params"]
    p = SenseProblem(tuple(int(n) for n in g["N"]), g["coord"], np.asfortranarray(g["maps"]),
                     width=int(width), ntable=int(ntab), oversamp=float(osf))
    hip._scratch = None
    A = p.build_fused(hip) if level == "fused" else p.build_tree(hip, level=level)
    x, k = g["sense_x"], g["sense_k"]
    assert rel_err(A * x, g["sense_Ax"]) < RTOL
    assert rel_err(A.H * k, g["sense_AHk"]) < RTOL
    AHA = normal_operator(A, lamda=float(g["lamda"]))
    y_d = hip.zero_array((A.shape[1], 1), C64)
    AHA.eval(y_d, hip.copy_array(x))
    assert rel_err(y_d.to_host(), g["sense_AHAx"]) < RTOL
    # CG iterates (reference backend.py:639-689)
    hip._scratch = None
    AHA = A.H * A + float(g["lamda"]) * hip.Eye(A.shape[1])
    for it in (1, 3):
        x0 = np.zeros((A.shape[1], 1), dtype=C64, order='F')
        hist = hip.cg(AHA, g["cg_b"].copy(order='F'), x0, maxiter=it)
        assert rel_err(x0, g["cg_it%d" % it]) < 1e-4 and len(hist) == it
    hip._scratch = None


def test_sense_even_grid_golden(hip):
    """tests/golden/sense_even.npz: the REFERENCE's products for image 64^3, 8 coils, grid 128^3 (even: the -O3 tree's G' is real up
    to rounding residue), a radial trajectory -- BASELINE config 4 in small -- against the fused HIP tree, which on this grid takes the
    zero-pad-aware A x B passes, the 8-byte brick entries and the 4-byte gather values (real-weight formats)"""
    from test_sense_cpu import even_grid_problem, check_even_grid_products
    p, g, x, k = even_grid_problem()
    hip._scratch = None
    assert hip.supports_padded_fft(p.oN, p.C)
    A = p.build_zpadfft(hip)
    check_even_grid_products(A, g, x, k, tol=RTOL)
    mats = []
    stack = [A]
    while stack:
        node = stack.pop()
        if getattr(node, '_matrix_d', None) is not None:
            mats.append(node._matrix_d)
        stack.extend(getattr(node, '_children', None) or [])
    assert mats and all((getattr(m, '_bricks', None) or {}).get('words') == 2 and m._real_values() is not None for m in mats), \
        "the fused tree's gridding matrix should carry the real-weight formats on an even grid"
    hip._scratch = None
    check_even_grid_products(p.build_tree(hip, level=3), g, x, k, tol=RTOL)       # the reference's own -O3 leaves
    hip._scratch = None


def test_sense_medium_vs_oracle_and_properties(hip, oracle_backend):
    """64^3 image, 4 coils, grid 128^3 (pow-2 LDS FFT path), ~1e5 samples: oracle parity + adjointness + linearity"""
    p = SenseProblem.synthetic((64, 64, 64), 4, nspokes=400, nreadout=128, width=2, oversamp=2.0, seed=4)
    hip._scratch = None
    oracle_backend._scratch = None
    A = p.build_fused(hip)
    Ao = p.build_fused(oracle_backend)
    assert "lds" in hip.fft_describe(p.oN + (p.C,))
    x = rand64c(A.shape[1], 1, seed=1)
    k = rand64c(A.shape[0], 1, seed=2)
    Ax, AHk = A * x, A.H * k
    assert rel_err(Ax, Ao * x) < RTOL
    assert rel_err(AHk, Ao.H * k) < RTOL
    np.testing.assert_allclose(np.vdot(k, Ax), np.vdot(AHk, x), rtol=1e-4)          # <Ax,k> = <x,A^H k>
    x2 = rand64c(A.shape[1], 1, seed=3)
    assert rel_err(A * (x + (2 - 1j) * x2).astype(C64), Ax + (2 - 1j) * (A * x2)) < RTOL   # linearity
    AHA = normal_operator(A)
    tr = Trace()
    hip.trace = tr
    y_d = hip.zero_array((A.shape[1], 1), C64)
    AHA.eval(y_d, hip.copy_array(x))
    hip.trace = None
    assert rel_err(y_d.to_host(), A.H * Ax) < RTOL
    ev = tr.by_event()
    assert ev['csrmm']['calls'] == 4 and ev['fft']['calls'] == 2       # S', G', G'^H, S'^H + FFT, IFFT
    x_d = hip.copy_array(x)
    assert hip.cdot(x_d, y_d).real > 0 and abs(hip.cdot(x_d, y_d).imag) < 1e-3 * hip.cdot(x_d, y_d).real  # x^H AHA x real > 0
    hip._scratch = None


def test_sense_full_size_properties(hip):
    """BASELINE config 4 at full size (image 256^3, 8 coils, grid 512^3, T = 1,851,904): the path bench.py times.
    No oracle finishes here in seconds, so: adjointness, linearity, A^H A = A^H(A x), and agreement of the two
    independent grid layouts (coil-interleaved kernels vs per-coil kernels)."""
    p = SenseProblem.synthetic((256, 256, 256), 8, nspokes=3617, nreadout=512, width=2, ntable=128, oversamp=2.0, seed=4)
    hip._scratch = None
    c128 = np.complex128
    A = p.build_zpadfft(hip)                      # layout 2: what the benchmark runs
    x, x2 = rand64c(A.shape[1], 1, seed=1), rand64c(A.shape[1], 1, seed=3)
    k = rand64c(A.shape[0], 1, seed=2)
    Ax, AHk = A * x, A.H * k
    lhs, rhs = np.vdot(k.astype(c128), Ax.astype(c128)), np.vdot(AHk.astype(c128), x.astype(c128))
    assert abs(lhs - rhs) <= 1e-5 * abs(lhs)                                          # <Ax, k> = <x, A^H k>
    assert rel_err(A * (x + (2 - 1j) * x2).astype(C64), Ax + (2 - 1j) * (A * x2)) < RTOL   # linearity
    y_d = hip.zero_array((A.shape[1], 1), C64)
    normal_operator(A).eval(y_d, hip.copy_array(x))
    AHAx = y_d.to_host()
    assert rel_err(AHAx, A.H * Ax) < RTOL
    q = np.vdot(x.astype(c128), AHAx.astype(c128))
    assert q.real > 0 and abs(q.imag) < 1e-5 * q.real                                 # x^H A^H A x is real and positive
    del A, y_d
    hip._scratch = None
    A1 = p.build_zpadfft(hip, layout=1)           # per-coil grids: different FFT instantiations and SpMM kernels
    assert rel_err(A1 * x, Ax) < RTOL and rel_err(A1.H * k, AHk) < RTOL
    del A1
    hip._scratch = None


# ---------------------------------------------------------------------------------------
# fused zero-pad / crop transforms (ZpadFFT leaf)
# ---------------------------------------------------------------------------------------
def _il(a2d):
    """row-major view of the memory of a column-major (n, C) panel: element (i, c) of the interleaved layout"""
    return np.asfortranarray(a2d).reshape(-1, order='F').reshape(a2d.shape)


def _to_layout(v4, layout):
    """(x, y, z, c) host array -> (P, C) column-major panel whose MEMORY is in the grid's order"""
    C = v4.shape[3]
    if layout == 2:         # (c, x, z, y), c fastest
        return np.asfortranarray(v4.transpose(3, 0, 2, 1).reshape(-1, order='F').reshape((-1, C), order='F'))
    if layout == 1:
        v4 = v4.transpose(0, 2, 1, 3)
    return np.asfortranarray(v4.reshape(-1, C, order='F'))


def _from_layout(flat, grid, layout):
    C = flat.shape[1]
    if layout == 2:
        return np.asfortranarray(flat).reshape(-1, order='F').reshape((C, grid[0], grid[2], grid[1]), order='F').transpose(1, 3, 2, 0)
    if layout == 1:
        return flat.reshape((grid[0], grid[2], grid[1], C), order='F').transpose(0, 2, 1, 3)
    return flat.reshape(tuple(grid) + (C,), order='F')


@pytest.mark.parametrize("grid,box,lo,C,weighted,layout", [
    ((256, 256, 256), (128, 128, 128), None, 3, True, 0),
    ((256, 256, 256), (128, 128, 128), None, 3, True, 1),
    ((256, 256, 256), (100, 77, 130), (5, 100, 126), 2, True, 0),
    ((256, 256, 256), (100, 77, 130), (5, 100, 126), 2, True, 1),
    ((256, 512, 256), (128, 256, 32), None, 1, False, 0),
    ((512, 256, 256), (128, 200, 32), None, 2, False, 1),
    ((256, 256, 256), (256, 256, 256), (0, 0, 0), 2, True, 1),       # no padding at all
    ((256, 256, 256), (128, 128, 128), None, 8, True, 2),             # coils interleaved
    ((256, 256, 256), (100, 77, 130), (5, 100, 126), 2, True, 2),
    ((512, 256, 256), (128, 200, 32), None, 4, False, 2),
    ((256, 256, 512), (64, 31, 256), None, 16, True, 2),
    ((256, 256, 256), (128, 128, 128), None, 1, True, 2),
])
def test_padded_and_cropped_fft_leaves(hip, grid, box, lo, C, weighted, layout):
    """fft_padded == fftn(zero-pad(w*x)) and ifft_cropped == conj(w)*crop(ifftn(y)), both computed on the GPU
    by the plain (dense) transform of the same library, plus the adjoint identity between the two."""
    hip._scratch = None
    lo = lo or tuple(m // 2 + int(np.ceil(-n / 2)) for m, n in zip(grid, box))
    P, N = int(np.prod(grid)), int(np.prod(box))
    x = rand64c(N, 1, seed=1)
    w = rand64c(N, C, seed=2) if weighted else None
    w_d = (hip.copy_array(np.ascontiguousarray(w).reshape(-1)) if layout == 2 else hip.copy_array(w)) if weighted else None
    sl = tuple(slice(l, l + b) for l, b in zip(lo, box))
    # dense reference on the same GPU: explicit zero-padded array through the plain transform
    full = np.zeros(grid + (C,), dtype=C64, order='F')
    full[sl + (slice(None),)] = (w if weighted else np.ones((N, C), C64)).reshape(box + (C,), order='F') * x.reshape(box + (1,), order='F')
    full_d = hip.copy_array(full)
    ref_d = hip.zero_array(full.shape, C64)
    hip.fftn(ref_d, full_d)
    y_d = hip.copy_array(np.full((P, C), np.nan, dtype=C64, order='F'))       # every element must be overwritten
    ws = hip.zero_array((hip._fft_padded_workspace(grid, lo, box, C, layout) // 8,), C64)
    hip.fft_padded(y_d, hip.copy_array(x), w_d, grid, lo, box, ws, layout)
    y = y_d.to_host()
    assert rel_err(y, _to_layout(ref_d.to_host(), layout)) < 2e-6
    # cropped inverse of a random grid panel (given in the grid's memory order)
    k = rand64c(P, C, seed=3)
    k_d = hip.copy_array(k)
    inv_d = hip.zero_array(grid + (C,), C64)
    hip.ifftn(inv_d, hip.copy_array(np.asfortranarray(_from_layout(k, grid, layout))))
    exp = inv_d.to_host()[sl + (slice(None),)].reshape(N, C, order='F')
    if weighted:
        exp = np.conj(w) * exp
    xc_d = hip.copy_array(np.full((N, C), np.nan, dtype=C64, order='F'))
    hip.ifft_cropped(xc_d, k_d, w_d, grid, lo, box, ws, layout)
    np.testing.assert_array_equal(k_d.to_host(), k)                           # the input panel stays intact
    xc = xc_d.to_host()
    assert rel_err(_il(xc) if layout == 2 else xc, exp) < 2e-6
    # <F x, k> == <x, F^H k>
    s_d = hip.zero_array((N, 1), C64)
    hip.sum_columns(s_d, xc_d, interleaved=(layout == 2))
    np.testing.assert_allclose(s_d.to_host()[:, 0], exp.sum(axis=1), rtol=2e-4, atol=2e-4 * np.abs(exp).max())
    c128 = np.complex128            # float32 accumulation over ~1e8 terms is itself only good to ~1e-4
    np.testing.assert_allclose(np.vdot(k.astype(c128), y.astype(c128)),
                               np.vdot(s_d.to_host().astype(c128), x.astype(c128)), rtol=1e-4)


def test_sum_columns(hip):
    X = rand64c(1000, 7, seed=1)
    y = rand64c(1000, 1, seed=2)
    Xd = hip.copy_array(rand64c(1003, 7, seed=3))
    Xd[2:1002, :].copy_from(X)
    y_d = hip.copy_array(y)
    hip.sum_columns(y_d, Xd[2:1002, :], alpha=0.5 - 1j, beta=2.0)
    np.testing.assert_allclose(y_d.to_host(), 2.0 * y + (0.5 - 1j) * X.sum(axis=1, keepdims=True), rtol=1e-5, atol=1e-5)
    hip.sum_columns(y_d, Xd[2:1002, :])
    np.testing.assert_allclose(y_d.to_host(), X.sum(axis=1, keepdims=True), rtol=1e-5, atol=1e-5)


def test_zpadfft_operator_matches_reference_composition(hip, oracle_backend):
    """ZpadFFT == KronI(C, fft) * S' (the reference's -O3 leaves) on the GPU, and == the oracle's ZpadFFT;
    SENSE through it == SENSE through the -O3 tree; alpha/beta handling of the adjoint."""
    p = SenseProblem.synthetic((128, 128, 128), 3, nspokes=300, nreadout=256, width=2, oversamp=2.0, seed=4)
    assert hip.supports_padded_fft(p.oN)
    hip._scratch = None
    oracle_backend._scratch = None
    A_ref = p.build_fused(hip)                   # G' * (KronI(fft) * S'), leaves pinned by the goldens
    A = p.build_zpadfft(hip)                     # grid in (x, z, y) order, G' permuted to match
    A_o = p.build_zpadfft(oracle_backend, layout=0)
    A_l0 = p.build_zpadfft(hip, layout=0)
    p4 = SenseProblem.synthetic((128, 128, 128), 4, nspokes=300, nreadout=256, width=2, oversamp=2.0, seed=4)
    A_il, A_il_o, A_l1 = p4.build_zpadfft(hip, layout=2), p4.build_zpadfft(oracle_backend, layout=2), p4.build_zpadfft(hip, layout=1)
    x4, k4 = rand64c(A_il.shape[1], 1, seed=1), rand64c(A_il.shape[0], 1, seed=2)
    assert rel_err(A_il * x4, A_l1 * x4) < RTOL and rel_err(A_il.H * k4, A_l1.H * k4) < RTOL
    assert rel_err(A_il * x4, A_il_o * x4) < RTOL and rel_err(A_il.H * k4, A_il_o.H * k4) < RTOL
    hip._scratch = None
    oracle_backend._scratch = None
    # 8 coils (the headline configuration: widest lane split of the x passes and of the fused coil combination)
    for nc in (8,):
        pc = SenseProblem.synthetic((128, 128, 128), nc, nspokes=100, nreadout=256, width=2, oversamp=2.0, seed=6)
        B_il, B_l1 = pc.build_zpadfft(hip, layout=2), pc.build_zpadfft(hip, layout=1)
        xc_, kc_ = rand64c(B_il.shape[1], 1, seed=1), rand64c(B_il.shape[0], 1, seed=2)
        assert rel_err(B_il * xc_, B_l1 * xc_) < RTOL and rel_err(B_il.H * kc_, B_l1.H * kc_) < RTOL
        hip._scratch = None
    x = rand64c(A.shape[1], 1, seed=1)
    k = rand64c(A.shape[0], 1, seed=2)
    Ax, AHk = A * x, A.H * k
    assert rel_err(Ax, A_ref * x) < RTOL and rel_err(AHk, A_ref.H * k) < RTOL
    assert rel_err(Ax, A_o * x) < RTOL and rel_err(AHk, A_o.H * k) < RTOL
    assert rel_err(A_l0 * x, Ax) < RTOL and rel_err(A_l0.H * k, AHk) < RTOL
    y0 = rand64c(A.shape[1], 1, seed=5)
    got = _eval(A.H, hip, y0, k, alpha=0.5 + 0.25j, beta=-1.5)
    assert rel_err(got, (0.5 + 0.25j) * AHk - 1.5 * y0) < RTOL
    AHA, AHA_ref = normal_operator(A, lamda=0.2), None
    tr = Trace()
    hip.trace = tr
    y_d = hip.zero_array((A.shape[1], 1), C64)
    AHA.eval(y_d, hip.copy_array(x))
    hip.trace = None
    hip._scratch = None
    AHA_ref = normal_operator(A_ref, lamda=0.2)
    tr2 = Trace()
    hip.trace = tr2
    y2_d = hip.zero_array((A.shape[1], 1), C64)
    AHA_ref.eval(y2_d, hip.copy_array(x))
    hip.trace = None
    assert rel_err(y_d.to_host(), y2_d.to_host()) < RTOL
    # the fused leaf books the same algorithmic bytes as the leaves it replaces -- per coil it really evaluates: three coils run as
    # ONE 4-wide interleaved chunk with a zero-weight coil (indigo_amd.fused.plan_chunks), the per-coil layout as three
    assert [w for _, _, w in A._coil_chunks] == [4]
    np.testing.assert_allclose(tr.total_bytes(), tr2.total_bytes() * 4.0 / 3.0, rtol=0.02)          # (the image-sized terms do not scale)
    hip._scratch = None
    tr3 = Trace()
    hip.trace = tr3
    normal_operator(A_l0, lamda=0.2).eval(y_d, hip.copy_array(x))
    hip.trace = None
    np.testing.assert_allclose(tr3.total_bytes(), tr2.total_bytes(), rtol=1e-3)
    hip._scratch = None


def support_table_from_segments(seg, zw=(16, 16)):
    """flat int16 support table (see SenseProblem.grid_support) from a boolean array seg[ky, kz, kx tile]; zw = words per entry
    of the bitmaps in their input-side and output-side form (ig_grid_support)"""
    n1, n2, nt = seg.shape
    kz = np.arange(n2)[None, :, None]
    lo = np.where(seg, kz, n2).min(axis=1)
    hi = np.where(seg, kz + 1, 0).max(axis=1)
    ranges = np.zeros((n1 * nt + nt, 2), dtype=np.int16)
    ranges[:n1 * nt, 0] = np.where(hi > lo, lo, 0).reshape(-1)
    ranges[:n1 * nt, 1] = np.where(hi > lo, hi, 0).reshape(-1)
    for t in range(nt):
        ys = np.flatnonzero(hi[:, t] > lo[:, t])
        if ys.size:
            ranges[n1 * nt + t] = (ys[0], ys[-1] + 1)
    ky, kzz, tt = np.nonzero(seg)
    parts = [ranges.reshape(-1)]
    for z in ((zw[0],) if zw[0] == zw[1] else zw):
        bits = np.zeros((n1, nt, z), dtype=np.uint32)
        np.bitwise_or.at(bits, (ky, tt, kzz % z), np.uint32(1) << (kzz // z).astype(np.uint32))
        parts.append(bits.reshape(-1).view(np.int16))
    return np.concatenate(parts), ranges[:n1 * nt]


def test_single_coil_sense_rank(hip, oracle_backend):
    """what one rank of an 8-GPU run evaluates: one coil, grid layout 1, single-column dense-lane adjoint gridding"""
    p = SenseProblem.synthetic((64, 64, 64), 3, nspokes=400, nreadout=128, width=2, oversamp=4.0, seed=9)   # grid 256^3
    hip._scratch = None
    oracle_backend._scratch = None
    A, A_o = p.build_zpadfft(hip, coils=[1]), p.build_zpadfft(oracle_backend, coils=[1])
    x, k = rand64c(A.shape[1], 1, seed=1), rand64c(A.shape[0], 1, seed=2)
    assert rel_err(A * x, A_o * x) < RTOL and rel_err(A.H * k, A_o.H * k) < RTOL
    hip._scratch = None
    oracle_backend._scratch = None


def test_padded_fft_with_support_table(hip):
    """k-space support at 16-row-segment granularity: the padded transform guarantees only the flagged segments;
    the cropped transform reads everything else as zero; the masked adjoint SpMM writes only flagged segments."""
    grid, box, C, layout = (256, 256, 256), (128, 128, 128), 2, 1
    lo = tuple(m // 2 + int(np.ceil(-n / 2)) for m, n in zip(grid, box))
    n0, n1, n2 = grid
    P, N = int(np.prod(grid)), int(np.prod(box))
    rng = np.random.default_rng(7)
    nt = n0 // 16
    zlo = rng.integers(0, 200, (n1, 1, nt))
    zhi = zlo + rng.integers(0, 57, (n1, 1, nt))                              # some ranges empty
    kzv = np.arange(n2)[None, :, None]
    seg = (kzv >= zlo) & (kzv < zhi) & (rng.random((n1, n2, nt)) < 0.6)       # gaps inside the ranges
    seg[:30, :, ::2] = False                                                  # even kx tiles: no support for ky < 30
    seg[220:, :, ::2] = False                                                 #                 ... nor for ky >= 220
    flat_table, table = support_table_from_segments(seg)
    sup = hip.copy_array(flat_table)
    # membership mask in (x, z, y) memory order: index kx + n0*(kz + n2*ky)
    inside = np.repeat(seg, 16, axis=2).reshape(-1)
    from oracle.np_backend import NumpyBackend
    np.testing.assert_array_equal(NumpyBackend.support_rows(flat_table, grid), inside)
    x, w = rand64c(N, 1, seed=1), rand64c(N, C, seed=2)
    ws = hip.zero_array((hip._fft_padded_workspace(grid, lo, box, C, layout) // 8,), C64)
    full_d = hip.zero_array((P, C), C64)
    hip.fft_padded(full_d, hip.copy_array(x), hip.copy_array(w), grid, lo, box, ws, layout)
    sentinel = np.full((P, C), 7 - 3j, dtype=C64, order='F')
    y_d = hip.copy_array(sentinel)
    hip.fft_padded(y_d, hip.copy_array(x), hip.copy_array(w), grid, lo, box, ws, layout, sup)
    y, full = y_d.to_host(), full_d.to_host()
    np.testing.assert_array_equal(y[inside], full[inside])        # outside the support Y is undefined
    skipped = np.repeat(table[:, 1] <= table[:, 0], 16)             # tiles with an empty range are not touched at all
    untouched = np.broadcast_to(skipped.reshape(n1, n0).T[:, None, :], (n0, n2, n1)).reshape(-1, order='F')
    z_in_box = np.zeros(n2, bool)
    z_in_box[lo[2]:lo[2] + box[2]] = True                           # (rows z in the box hold pass-y intermediates)
    zmask = np.broadcast_to(z_in_box[None, :, None], (n0, n2, n1)).reshape(-1, order='F')
    np.testing.assert_array_equal(y[untouched & ~zmask], sentinel[untouched & ~zmask])
    # cropped: garbage outside the support must not matter
    k = rand64c(P, C, seed=3)
    k_clean = k.copy(order='F')
    k_clean[~inside] = 0
    k_dirty = k.copy(order='F')
    k_dirty[~inside] = np.nan
    a_d, b_d = hip.zero_array((N, C), C64), hip.zero_array((N, C), C64)
    hip.ifft_cropped(a_d, hip.copy_array(k_clean), hip.copy_array(w), grid, lo, box, ws, layout)
    hip.ifft_cropped(b_d, hip.copy_array(k_dirty), hip.copy_array(w), grid, lo, box, ws, layout, sup)
    assert rel_err(b_d.to_host(), a_d.to_host()) < 1e-6
    # masked adjoint SpMM: rows outside the support keep their old contents, rows inside get A^H x
    T = 5000
    cols = rng.choice(np.flatnonzero(inside), size=T * 4)
    A = spp.csr_matrix((rand64c(T * 4, seed=4), cols, np.arange(0, T * 4 + 1, 4)), shape=(T, P))
    A_d = hip.csr_matrix(hip, A)
    A_d.set_grid_support(flat_table, n0, n2)
    xs = rand64c(T, C, seed=5)
    out_d = hip.copy_array(sentinel)
    A_d.adjoint(out_d, hip.copy_array(xs))
    out = out_d.to_host()
    exp = A.conj().T @ xs
    assert rel_err(out[inside], exp[inside]) < RTOL
    np.testing.assert_array_equal(out[~inside], sentinel[~inside])


@pytest.mark.parametrize("C,tile,grid,box", [(4, 8, (256,) * 3, (128,) * 3), (8, 8, (256,) * 3, (128,) * 3), (8, 4, (256,) * 3, (128,) * 3),
                                             (8, 2, (256,) * 3, (128,) * 3), (2, 8, (256,) * 3, (128,) * 3), (4, 16, (256,) * 3, (128,) * 3),
                                             (8, 8, (160, 192, 320), (128, 150, 256)), (4, 16, (320, 144, 200), (256, 100, 160)),
                                             (8, 8, (256, 240, 270), (128, 200, 208)), (2, 8, (160, 256, 640), (100, 128, 480))])
def test_padded_fft_with_a_finer_support_table_layout2(hip, C, tile, grid, box):
    """ig_fft_set_support_tile: the coil-interleaved transform with a support table of `tile` kx points per entry (coils * tile
    >= 16): the padded transform defines exactly the flagged segments (everything else keeps the sentinel outside the image
    planes), the cropped ones (per-coil and coil-summing) read everything else as zero -- against the same transforms
    without a table.  Grids whose z axis the A x B kernel transforms carry the bitmaps in two forms, B and A words per entry
    (ig_fft_support_words): 320 = 16 x 20, 200 = 10 x 20, 270 = 15 x 18, 640 = 20 x 32."""
    layout = 2
    lo = tuple(m // 2 + int(np.ceil(-n / 2)) for m, n in zip(grid, box))
    n0, n1, n2 = grid
    P, N = int(np.prod(grid)), int(np.prod(box))
    rng = np.random.default_rng(100 * C + tile + n2)
    nt = n0 // tile
    zw = hip.support_words(n2)
    assert zw is not None and (zw == (16, 16)) == (n2 in (256, 512))
    zlo = rng.integers(0, int(n2 * 0.8), (n1, 1, nt))
    zhi = zlo + rng.integers(0, max(2, n2 // 4), (n1, 1, nt))
    kzv = np.arange(n2)[None, :, None]
    seg = (kzv >= zlo) & (kzv < zhi) & (rng.random((n1, n2, nt)) < 0.6)
    seg[:n1 // 8, :, ::2] = False
    seg[n1 - n1 // 8:, :, ::3] = False
    flat_table, table = support_table_from_segments(seg, zw)
    sup = hip.copy_array(flat_table)
    inside = np.repeat(seg, tile, axis=2).reshape(-1)                         # rows kx + n0*(kz + n2*ky)
    x, w = rand64c(N, 1, seed=1), rand64c(N, C, seed=2)
    w_il = hip.copy_array(np.ascontiguousarray(w).reshape(-1))                # interleaved weights: w[i*C + c]
    ws = hip.zero_array((hip._fft_padded_workspace(grid, lo, box, C, layout) // 8,), C64)

    def rows(a_d):                                                           # (P, C) row-major view of an interleaved panel
        return a_d.to_host().reshape(-1, order='F').reshape(P, C)
    full_d = hip.zero_array((P, C), C64)
    hip.fft_padded(full_d, hip.copy_array(x), w_il, grid, lo, box, ws, layout)
    y_d = hip.copy_array(np.full((P, C), 7 - 3j, dtype=C64, order='F'))
    hip.fft_padded(y_d, hip.copy_array(x), w_il, grid, lo, box, ws, layout, sup, support_tile=tile)
    y, full = rows(y_d), rows(full_d)
    np.testing.assert_array_equal(y[inside], full[inside])
    z_in_box = np.zeros(n2, bool)
    z_in_box[lo[2]:lo[2] + box[2]] = True                                     # (planes z in the box hold pass-y intermediates)
    zmask = np.broadcast_to(z_in_box[None, :, None], (n0, n2, n1)).reshape(-1, order='F')
    assert np.all(y[~inside & ~zmask] == 7 - 3j)                              # nothing outside the support is written
    # cropped transforms: garbage outside the support does not matter
    k = rand64c(P * C, seed=3).reshape(P, C)
    k_clean, k_dirty = k.copy(), k.copy()
    k_clean[~inside] = 0
    k_dirty[~inside] = np.nan

    def il(a):
        return hip.copy_array(a.reshape(-1)).reshape((P, C))
    a_d, b_d = hip.zero_array((N, C), C64), hip.zero_array((N, C), C64)
    hip.ifft_cropped(a_d, il(k_clean), w_il, grid, lo, box, ws, layout)
    hip.ifft_cropped(b_d, il(k_dirty), w_il, grid, lo, box, ws, layout, sup, support_tile=tile)
    assert rel_err(b_d.to_host(), a_d.to_host()) < 1e-6
    s1, s2 = hip.zero_array((N, 1), C64), hip.zero_array((N, 1), C64)
    hip.ifft_cropped_sum(s1, il(k_clean), w_il, grid, lo, box, ws)
    hip.ifft_cropped_sum(s2, il(k_dirty), w_il, grid, lo, box, ws, sup, support_tile=tile)
    assert rel_err(s2.to_host(), s1.to_host()) < 1e-6


def test_fuse_zpadfft_transform_reaches_the_benchmarked_leaf(hip, oracle_backend):
    """the reference's route (NUFFT / KronI / VStack(Diag) factories, pics.py -O3 recipe) + FuseZpadFFT builds the same
    fused tree as SenseProblem.build_zpadfft and evaluates like the -O3 tree it came from"""
    from indigo_amd import operators as op
    from indigo_amd.transforms import FuseZpadFFT, sense_recipe
    p = SenseProblem.synthetic((64, 64, 64), 4, nspokes=300, nreadout=128, width=2, oversamp=4.0, seed=4)      # grid 256^3
    hip._scratch = None
    A3 = p.build_tree(hip, level=3)
    x, k = rand64c(A3.shape[1], 1, seed=1), rand64c(A3.shape[0], 1, seed=2)
    A3x, A3Hk = A3 * x, A3.H * k
    Af = FuseZpadFFT().visit(p.build_tree(hip, level=3))
    assert Af.has(op.ZpadFFT) and not Af.has(op.UnscaledFFT) and FuseZpadFFT.layout_of(Af) == 2
    assert rel_err(Af * x, A3x) < RTOL and rel_err(Af.H * k, A3Hk) < RTOL
    Ad = p.build_zpadfft(hip)
    assert rel_err(Af * x, Ad * x) < 1e-6
    hip._scratch = None


def test_cg_with_device_scalars_matches_host_scalar_cg(hip):
    """HipBackend.cg keeps alpha / beta / the residuals on the device (no host sync inside an iteration); the base
    Backend.cg is the reference's loop with host scalars (backend.py:639-689): same iterates, same residual history"""
    from indigo_amd.backends.backend import Backend
    p = SenseProblem.synthetic((32, 32, 32), 4, nspokes=96, nreadout=64, width=2, oversamp=2.0, seed=4)
    hip._scratch = None
    A = p.build_fused(hip)
    AHA = normal_operator(A, lamda=0.05)
    b = A.H * rand64c(A.shape[0], 1, seed=2)
    # hip.cg moves a real multiple of Eye at the root into its own lamda (either order of the Sum: ours and examples/pics.py:195's);
    # Backend.cg below evaluates the tree as written, so every comparison of this test covers that move
    for tree in (AHA, (A.H * A) + 0.05 * hip.Eye(A.shape[1])):
        rest, lam = hip._split_identity(tree, 0.25)
        assert abs(lam - 0.30) < 1e-12 and type(rest).__name__ == 'Product'
    assert hip._split_identity(A.H * A, 0.25)[1] == 0.25
    def cg_float64_vectors(iters):
        """the same loop with its vector arithmetic in float64 on the host (the operator stays the float32 HIP operator): the
        arbiter for two float32 loops that round differently (tools/lab/cg_rounding.py)"""
        x = np.zeros(b.shape, np.complex128)
        r = b.astype(np.complex128)
        pp = r.copy()
        rr = r0 = np.vdot(r, r).real
        hist = []
        for _ in range(iters):
            Ap = (AHA * pp.astype(C64)).astype(np.complex128)
            alpha = rr / np.vdot(pp, Ap).real
            x += alpha * pp
            r -= alpha * Ap
            r2 = np.vdot(r, r).real
            pp = r + (r2 / rr) * pp
            rr = r2
            hist.append(np.sqrt(rr / r0))
        return np.array(hist), x
    # The fused loop rounds differently from the host-scalar loop (one fused multiply-add where that has a multiply and an add,
    # another summation order in the reductions).  Up to five iterations of this system both agree to 1e-6; from the sixth on
    # float32 CG amplifies ANY rounding difference (at seven iterations either loop is ~10 % from the float64-vector loop in the
    # residual and 2e-3 in the iterate, and they are 1e-2 / 2e-4 from each other): there each loop is held to the float64 one.
    for iters, every in ((5, 10), (5, 3), (1, 1)):
        x1 = np.zeros_like(b, order='F')
        x2 = np.zeros_like(b, order='F')
        h1 = hip.cg(AHA, b.copy(order='F'), x1, maxiter=iters, check_every=every)
        h2 = Backend.cg(hip, AHA, b.copy(order='F'), x2, maxiter=iters)
        assert len(h1) == len(h2) == iters
        np.testing.assert_allclose(h1, h2, rtol=1e-4)
        assert rel_err(x1, x2) < 1e-5
    h64, x64 = cg_float64_vectors(7)
    x1 = np.zeros_like(b, order='F')
    x2 = np.zeros_like(b, order='F')
    h1 = np.array(hip.cg(AHA, b.copy(order='F'), x1, maxiter=7, check_every=3))
    h2 = np.array(Backend.cg(hip, AHA, b.copy(order='F'), x2, maxiter=7))
    assert len(h1) == len(h2) == 7
    np.testing.assert_allclose(h1[:5], h64[:5], rtol=1e-4)
    d1, d2 = np.abs(h1 / h64 - 1).max(), np.abs(h2 / h64 - 1).max()
    assert d1 < 2 * d2 + 1e-4 and rel_err(x1, x64) < 2 * rel_err(x2, x64) + 1e-5, (d1, d2, rel_err(x1, x64), rel_err(x2, x64))
    assert rel_err(x1, x2) < rel_err(x2, x64)               # closer to each other than either is to the float64 loop
    # tolerance: the history ends at the first residual below tol, and the iterations already enqueued behind it (the rest of
    # the block of four) are no-ops on the device: the iterate is the reference's, which breaks out at once (backend.py:683-685)
    x3 = np.zeros_like(b, order='F')
    x3r = np.zeros_like(b, order='F')
    h_ref = Backend.cg(hip, AHA, b.copy(order='F'), x3r, tol=h2[0] * 1.5, maxiter=20)
    h3 = hip.cg(AHA, b.copy(order='F'), x3, tol=h2[0] * 1.5, maxiter=20, check_every=4)
    assert len(h3) == len(h_ref) == 1
    assert rel_err(x3, x3r) < 1e-6
    # an exactly solvable system (A = I): the residual reaches 0 in one step, <p, Ap> = 0 afterwards -- no NaN may appear
    n = b.shape[0]
    xe = np.zeros_like(b, order='F')
    he = hip.cg(hip.Eye(n), b.copy(order='F'), xe, tol=1e-10, maxiter=25, check_every=10)
    assert len(he) == 1 and he[0] == 0.0
    assert np.isfinite(xe).all() and rel_err(xe, b) < 1e-6
    # more iterations than the context has history slots: the ring is fetched before it wraps (the reference takes any maxiter)
    xl = np.zeros_like(b, order='F')
    hl = hip.cg(2.0 * hip.Eye(n), b.copy(order='F'), xl, tol=0.0, maxiter=1100, check_every=2000)
    assert len(hl) == 1100 and np.isfinite(xl).all() and rel_err(xl, 0.5 * b) < 1e-6
    # device-resident b and x: the iterate is updated in place
    x7 = np.zeros_like(b, order='F')
    Backend.cg(hip, AHA, b.copy(order='F'), x7, maxiter=5)
    x_d = hip.zero_array(b.shape, C64)
    h4 = hip.cg(AHA, hip.copy_array(b), x_d, maxiter=5)
    assert rel_err(x_d.to_host(), x7) < 1e-5 and len(h4) == 5
    # blocks of iterations replayed as ONE HIP graph launch (ig_graph_*; tuning 'cg_graph', off by default: 0.03 ms of a 6.9 ms
    # iteration on the headline problem): the first block runs plainly, the second is recorded and replayed, the third replayed --
    # the same iterates as the plain loop (the launches ARE the same; only who enqueues them differs)
    from indigo_amd.transforms import reserve_for
    reserve_for(AHA, 1)
    xg, xp = np.zeros_like(b, order='F'), np.zeros_like(b, order='F')
    hip.tuning['cg_graph'] = True
    try:
        hg = hip.cg(AHA, b.copy(order='F'), xg, maxiter=6, check_every=2)
    finally:
        hip.tuning['cg_graph'] = False
    hp = hip.cg(AHA, b.copy(order='F'), xp, maxiter=6, check_every=2)
    assert len(hg) == len(hp) == 6
    np.testing.assert_allclose(hg, hp, rtol=1e-5)
    assert rel_err(xg, xp) < 1e-6
    hip._scratch = None
    # a complex lamda -- the reference's loop takes any scalar (backend.py:651-689) -- is not the fused passes' real regularisation
    # weight: the base implementation runs, with the caller's operator, and gives the host-scalar loop's numbers
    x8, x9 = np.zeros_like(b, order='F'), np.zeros_like(b, order='F')
    h8 = hip.cg(A.H * A, b.copy(order='F'), x8, lamda=0.05 + 0.01j, maxiter=3)
    h9 = Backend.cg(hip, A.H * A, b.copy(order='F'), x9, lamda=0.05 + 0.01j, maxiter=3)
    assert len(h8) == len(h9) == 3 and rel_err(x8, x9) < 1e-6
    hip._scratch = None


def test_head_rows_operator_on_the_gpu(hip):
    """operators.HeadRows (the first rows of a tree: a coil chunk padded with zero-weight coils) with alpha / beta and a two-column
    panel, on device arrays: views into the scratch panel, the zeroed tail, NaN-poisoned outputs with beta = 0"""
    import scipy.sparse as spp
    from indigo_amd import operators as op
    hip._scratch = None
    rng = np.random.default_rng(5)
    M = (spp.random(900, 64, density=0.2, random_state=rng) + 1j * spp.random(900, 64, density=0.2, random_state=rng)).astype(C64).tocsr()
    H = op.HeadRows(hip, hip.SpMatrix(M), 517)
    Md = M.toarray()[:517]
    for ncol in (1, 2):
        x = rand64c(64, ncol, seed=1)
        y0 = rand64c(517, ncol, seed=2)
        y = hip.copy_array(y0)
        H.eval(y, hip.copy_array(x), alpha=0.5 - 1j, beta=2.0)
        assert rel_err(y.to_host(), (0.5 - 1j) * (Md @ x) + 2.0 * y0) < RTOL
        k = rand64c(517, ncol, seed=3)
        z0 = rand64c(64, ncol, seed=4)
        z = hip.copy_array(z0)
        H.H.eval(z, hip.copy_array(k), alpha=1j, beta=-0.5)
        assert rel_err(z.to_host(), 1j * (Md.conj().T @ k) - 0.5 * z0) < RTOL
        yn = hip.copy_array(np.full((517, ncol), np.nan + 1j * np.nan, dtype=C64, order='F'))
        H.eval(yn, hip.copy_array(x))
        assert rel_err(yn.to_host(), Md @ x) < RTOL
    hip._scratch = None

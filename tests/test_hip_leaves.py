"""GPU parity tests for the leaf kernels, called through the C ABI (HipBackend -> ctypes).

Compared against (a) the golden vectors captured from the reference and (b) the
numpy oracle / inline numpy-scipy expressions on seeded inputs, the way the
reference's own suite does it (indigo/backends/test_backends.py).  Tolerance
for complex64 results: 1e-5 relative (north star), stated per assertion.
"""
import itertools

import numpy as np
import pytest
import scipy.sparse as spp

from conftest import csr_from, golden, rel_err
from indigo_amd.util import rand64c, randM

pytestmark = pytest.mark.gpu
C64 = np.dtype('complex64')
RTOL = 1e-5


# ---------------------------------------------------------------------------------------
# arrays (reference test_backends.py:13-151)
# ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("n", [4, 8, 129])
def test_array_roundtrips(hip, n):
    arr = rand64c(n, seed=n)
    d = hip.copy_array(arr)
    np.testing.assert_equal(d.to_host(), arr)
    z = hip.zero_array(arr.shape, arr.dtype)
    assert not z.to_host().any()
    z.copy_from(arr)
    np.testing.assert_equal(z.to_host(), arr)
    out = np.zeros_like(arr)
    d.copy_to(out)
    np.testing.assert_equal(out, arr)
    dup = d.copy()
    d._zero()
    np.testing.assert_equal(dup.to_host(), arr)
    into = hip.zero_array(arr.shape, arr.dtype)
    into.copy(dup)
    np.testing.assert_equal(into.to_host(), arr)
    into2 = hip.zero_array(arr.shape, arr.dtype)
    into2[:] = dup
    np.testing.assert_equal(into2.to_host(), arr)
    with pytest.raises(ValueError):
        hip.zero_array((n + 1,), arr.dtype).copy_from(arr)
    with pytest.raises(TypeError):
        hip.zero_array(arr.shape, np.complex128).copy_from(arr)



def test_large_arrays_are_placed_by_probing(hip):
    """arrays of at least tuning['placement_min_bytes'] are placed by probing (ig_probe_placement: the write pattern of a pass that steps
    megabytes per element): as the best-placed window, at 1 GB steps, of an allocation of the array + tuning['placement_window_gb'] GB
    -- the better of tuning['placement_window_allocs'] such allocations (round 6) -- or, without a window, as the best of tuning['placement_candidates'] allocations (round 5).  The probe destroys the
    contents, so it runs before anything is written: zero_array still returns zeros, copy_array the host's values; freeing a windowed
    array frees its whole allocation."""
    old = dict(hip.tuning)
    try:
        hip.tuning['placement_min_bytes'] = 32 << 20
        n = (48 << 20) // 8
        # the window of one allocation
        hip.tuning['placement_window_gb'] = 2
        free0 = hip.mem_info()[0]
        before = len(hip._placement_log)
        z = hip.zero_array((n,), C64)
        assert len(hip._placement_log) == before + 1
        nbytes, allocs, chosen = hip._placement_log[-1]
        # (one to three allocations of array + 2 GB -- until a ramp has been seen --, three windows each; the losers are freed)
        assert nbytes == n * 8 and 1 <= len(allocs) <= 3 and all(len(t) == 3 for t in allocs) and chosen == min(min(t) for t in allocs) and chosen > 0
        assert (z._arr - z._alloc_base) % (1 << 30) == 0 and 0 <= z._arr - z._alloc_base <= 2 << 30
        assert not z.to_host().any()
        x = rand64c(n, seed=3)
        x_d = hip.copy_array(x)
        np.testing.assert_equal(x_d.to_host(), x)
        del z, x_d
        hip.barrier()
        assert hip.mem_info()[0] >= free0 - (64 << 20)          # both allocations (2 GB + 48 MB each) are gone
        # the best of three allocations
        hip.tuning['placement_window_gb'] = 0
        hip.tuning['placement_candidates'] = 3
        before = len(hip._placement_log)
        z = hip.zero_array((n,), C64)
        assert len(hip._placement_log) == before + 1
        nbytes, cands, chosen = hip._placement_log[-1]
        assert nbytes == n * 8 and len(cands) == 3 and chosen == min(cands) and chosen > 0
        assert not z.to_host().any()
        hip.tuning['placement_candidates'] = 1
        hip.zero_array((n,), C64)
        assert len(hip._placement_log) == before + 1           # plain allocation
    finally:
        hip.tuning.clear()
        hip.tuning.update(old)

@pytest.mark.parametrize("s", [-2, -1, 1, 2])
def test_array_slice_1d(hip, s):
    arr = np.arange(10)
    d = hip.copy_array(arr)
    np.testing.assert_equal(d[:s].to_host(), arr[:s])
    np.testing.assert_equal(d[s:].to_host(), arr[s:])


@pytest.mark.parametrize("M,N,xb,xe,yb,ye", itertools.product([6, 7], [8, 9], [0, 2], [4, 5], [0, 1], [6, 7]))
def test_array_slice_2d(hip, M, N, xb, xe, yb, ye):
    arr = rand64c(M, N, seed=M * N)
    d = hip.copy_array(arr)
    sub = d[xb:xe, yb:ye]
    np.testing.assert_equal(sub.to_host(), arr[xb:xe, yb:ye])
    dup = hip.zeros_like(sub)
    dup.copy(sub)
    np.testing.assert_equal(dup.to_host(), arr[xb:xe, yb:ye])
    # writing through a strided view
    patch = rand64c(xe - xb, ye - yb, seed=1)
    sub.copy_from(patch)
    exp = arr.copy()
    exp[xb:xe, yb:ye] = patch
    np.testing.assert_equal(d.to_host(), exp)


def test_array_bad_reshape_of_row_slice(hip):
    d = hip.copy_array(rand64c(8, 4, seed=0))
    with pytest.raises(AssertionError):
        d[2:6, :].reshape((8, 2))


def test_on_host_and_mem_usage(hip):
    d = hip.copy_array(np.zeros(11, dtype=C64))
    with d.on_host() as h:
        h += 1
    assert d.to_host().sum() == 11
    assert hip.mem_usage() > 0


def test_only_complex64(hip):
    x = hip.copy_array(np.ones(4, dtype=np.complex128))
    with pytest.raises(AssertionError):
        hip.axpby(1, x, 1, x)


# ---------------------------------------------------------------------------------------
# BLAS-1
# ---------------------------------------------------------------------------------------
def test_blas_golden(hip):
    g = golden("leaf_blas")
    for i in range(int(g["count"])):
        x, y = g["axpby%d_x" % i], g["axpby%d_y" % i]
        alpha, beta = g["axpby%d_ab" % i]
        y_d = hip.copy_array(y)
        hip.axpby(beta.real, y_d, alpha, hip.copy_array(x))
        assert rel_err(y_d.to_host(), g["axpby%d_out" % i]) < RTOL
        s_d = hip.copy_array(x)
        hip.scale(s_d, alpha)
        assert rel_err(s_d.to_host(), g["scale%d_out" % i]) < RTOL
        np.testing.assert_allclose(hip.dot(hip.copy_array(x), hip.copy_array(y)), float(g["dot%d" % i]), rtol=RTOL)
        np.testing.assert_allclose(hip.norm2(hip.copy_array(x)), float(g["nrm%d" % i]), rtol=RTOL)
        m_d = hip.copy_array(x)
        hip.max(0.5, m_d)
        np.testing.assert_array_equal(m_d.to_host(), g["max%d_out" % i])


@pytest.mark.parametrize("n,alpha,beta", itertools.product(
    [1, 2, 3, 10, 23, 129, 144, 100003], [-2.1, 0.0, 1.0, 1.2 + 3j], [0.0, 0.5, 1.0, 1.5 - 1j]))
def test_blas_axpby_scale_grid(hip, n, alpha, beta):
    x, y = rand64c(n, seed=n), rand64c(n, seed=n + 1)
    y_d = hip.copy_array(y)
    hip.axpby(beta, y_d, alpha, hip.copy_array(x))
    exp = (np.complex64(beta) * y + np.complex64(alpha) * x).astype(C64)
    np.testing.assert_allclose(y_d.to_host(), exp, atol=2e-6)
    x_d = hip.copy_array(x)
    hip.scale(x_d, alpha)
    np.testing.assert_allclose(x_d.to_host(), np.complex64(alpha) * x, atol=2e-6)


def test_blas_unaligned_views_and_beta0_ignores_nan(hip):
    n = 1001
    x, y = rand64c(n, seed=1), rand64c(n, seed=2)
    x_d, y_d = hip.copy_array(x), hip.copy_array(y)
    hip.axpby(0.5, y_d[1:], 2.0, x_d[:-1])        # 8-byte aligned, not 16
    exp = y.copy()
    exp[1:] = 0.5 * y[1:] + 2.0 * x[:-1]
    np.testing.assert_allclose(y_d.to_host(), exp, atol=2e-6)
    bad = hip.copy_array(np.full(n, np.nan + 1j * np.nan, dtype=C64))
    hip.axpby(0, bad, 1.5, x_d)                    # BLAS rule: beta == 0 never reads y
    np.testing.assert_allclose(bad.to_host(), 1.5 * x, atol=2e-6)


@pytest.mark.parametrize("n", [1, 10, 23, 129, 144, 1 << 20, 3000001])
def test_blas_dot_nrm2(hip, n):
    x, y = rand64c(n, seed=n), rand64c(n, seed=n + 7)
    x_d, y_d = hip.copy_array(x), hip.copy_array(y)
    exp = np.vdot(x.astype(np.complex128), y.astype(np.complex128))
    np.testing.assert_allclose(hip.dot(x_d, y_d), exp.real, rtol=RTOL)
    np.testing.assert_allclose(hip.cdot(x_d, y_d), exp, rtol=RTOL)
    np.testing.assert_allclose(hip.norm2(x_d), np.linalg.norm(x.astype(np.complex128)) ** 2, rtol=RTOL)


@pytest.mark.parametrize("val,N", itertools.product([-1.5, 0, 0.5, 1.5], [4, 5, 6, 1025]))
def test_max(hip, val, N):
    arr = rand64c(N, seed=N)
    d = hip.copy_array(arr)
    hip.max(val, d)
    act = d.to_host()
    np.testing.assert_array_equal(act.real, np.maximum(arr.real, np.float32(val)))
    np.testing.assert_array_equal(act.imag, np.maximum(arr.imag, np.float32(val)))


# ---------------------------------------------------------------------------------------
# SpMM
# ---------------------------------------------------------------------------------------
def test_csrmm_golden(hip):
    g = golden("leaf_csrmm")
    for policy in ("transpose", "atomic"):
        hip.adjoint_policy = policy
        for i in range(int(g["count"])):
            p = "c%d_" % i
            A = csr_from(g, p)
            alpha, beta = g[p + "ab"]
            A_d = hip.csr_matrix(hip, A)
            assert A_d._exwrite == bool(g[p + "inspect"][2])
            assert abs(A_d._row_frac - g[p + "inspect"][0]) < 1e-12
            assert abs(A_d._col_frac - g[p + "inspect"][1]) < 1e-12
            y_d = hip.copy_array(g[p + "y"])
            A_d.forward(y_d, hip.copy_array(g[p + "x"]), alpha=alpha, beta=beta)
            assert rel_err(y_d.to_host(), g[p + "fwd"]) < RTOL, (policy, i)
            ya_d = hip.copy_array(g[p + "ya"])
            A_d.adjoint(ya_d, hip.copy_array(g[p + "xa"]), alpha=alpha, beta=beta)
            assert rel_err(ya_d.to_host(), g[p + "adj"]) < RTOL, (policy, i)
    hip.adjoint_policy = "transpose"


@pytest.mark.parametrize("M,K,n,density", itertools.product([23, 45], [45, 23], [1, 2, 3, 8, 9, 17, 64, 65, 130], [0.01, 0.1, 0.5, 1.0]))
def test_csr_matrix_vs_scipy(hip, M, K, n, density):
    """reference test_backends.py:183-210 / test_operators.py:12-52 on a wider column grid"""
    A = randM(M, K, density, seed=M * 1000 + K * 10 + n)
    A_d = hip.csr_matrix(hip, A)
    x = rand64c(K, n, seed=1)
    y_d = hip.zero_array((M, n), C64)
    A_d.forward(y_d, hip.copy_array(x))
    np.testing.assert_allclose(y_d.to_host(), A @ x, rtol=RTOL, atol=1e-5)
    xa = rand64c(M, n, seed=2)
    for policy in ("transpose", "atomic"):
        hip.adjoint_policy = policy
        ya_d = hip.zero_array((K, n), C64)
        A_d.adjoint(ya_d, hip.copy_array(xa))
        np.testing.assert_allclose(ya_d.to_host(), A.conj().T @ xa, rtol=RTOL, atol=1e-5)
    hip.adjoint_policy = "transpose"


def test_brick_format_declined_for_rows_that_touch_too_many_bricks(hip):
    """a row spread over more than 64 bricks has no brick-binned format: set_grid_bricks declines and the adjoint of the
    interleaved panel is served by the gather over the transpose"""
    n0, nm, ns = 64, 64, 64
    P, T = n0 * nm * ns, 50
    rng = np.random.default_rng(4)
    rows = np.repeat(np.arange(T), 200)
    cols = rng.integers(0, P, size=rows.size)                               # 200 nonzeros all over the grid per row
    A = spp.csr_matrix((rand64c(rows.size, seed=1), (rows, cols)), shape=(T, P))
    A.sum_duplicates(); A.sort_indices()
    A_d = hip.csr_matrix(hip, A)
    A_d.set_grid_interleaved(True)
    A_d.set_grid_bricks(n0, nm, ns, ncols=8)
    assert A_d._bricks is None
    X = rand64c(T, 8, seed=2)
    y_d = hip.zero_array((P, 8), C64)
    A_d.adjoint(y_d, hip.copy_array(X))
    got = y_d.to_host().reshape(-1, order='F').reshape(P, 8)
    assert rel_err(got, A.conj().T.astype(np.complex128) @ X.astype(np.complex128)) < RTOL


@pytest.mark.parametrize("K,alpha,ld_pad,dense_blob", [(4096, 1, 0, False), (20000 // 16 * 16, 0.5 - 0.25j, 5, True), (64, 2, 0, False)])
def test_wide_panel_adjoint_by_bricks(hip, monkeypatch, K, alpha, ld_pad, dense_blob):
    """ig_ccsrmm_t_bricks_wide: A^H X for a 64-column column-major panel as a scatter binned by 16-row bricks of the result --
    against scipy in double precision and against the gather route; rows no nonzero touches come out zero over a
    sentinel; a dense blob of columns makes heavy bricks (pieces adding with atomics); sliced panels (padded leading dimension)"""
    M = 5000
    rng = np.random.default_rng(K)
    rows = np.repeat(np.arange(M), 27)
    cols = rng.integers(0, K, size=rows.size)
    if dense_blob:
        hot = rng.integers(1000, 1040, size=rows.size)                    # 40 hot columns: > 4096 entries per brick
        cols = np.where(rng.random(rows.size) < 0.5, hot, cols)
    cols[rows % 7 == 0] = (cols[rows % 7 == 0] // 3) * 3                   # some structure, duplicates summed
    A = spp.csr_matrix((rand64c(rows.size, seed=1), (rows, cols)), shape=(M, K))
    A.sum_duplicates(); A.sort_indices()
    A_d = hip.csr_matrix(hip, A)
    xfull = rand64c(M + ld_pad, 64, seed=2)
    yfull = np.full((K + ld_pad, 64), 7 - 3j, dtype=C64, order='F')
    x_d = hip.copy_array(xfull)[0:M, :]
    y_d = hip.copy_array(yfull)[0:K, :]
    A_d.adjoint(y_d, x_d, alpha=alpha)
    assert A_d._wide is not None and A_d._wide['ntasks'] > 0
    exp = alpha * (A.conj().T.astype(np.complex128) @ xfull[:M].astype(np.complex128))
    got = y_d.to_host()
    assert rel_err(got, exp) < RTOL
    untouched = np.setdiff1d(np.arange(K), np.unique(A.indices))
    assert np.all(got[untouched] == 0)
    monkeypatch.setitem(hip.tuning, "wide_bricks", False)
    B_d = hip.csr_matrix(hip, A)
    y2 = hip.copy_array(yfull)[0:K, :]
    B_d.adjoint(y2, x_d, alpha=alpha)
    assert getattr(B_d, '_wide', False) is False
    assert rel_err(y2.to_host(), got) < 1e-5
    # beta != 0 keeps the contract through the gather route
    monkeypatch.setitem(hip.tuning, "wide_bricks", True)
    y3 = hip.copy_array(yfull)[0:K, :]
    A_d.adjoint(y3, x_d, alpha=alpha, beta=0.5)
    assert rel_err(y3.to_host(), exp + 0.5 * yfull[:K]) < RTOL


@pytest.mark.parametrize("dims,shape,alpha,ld_pad,blob", [((32, 32, 32), (2, 2), 1, 0, False), ((48, 6, 8), (2, 2), 0.5 - 0.25j, 3, True),
                                                         ((32, 8, 4), (2, 1), 2, 0, False), ((64, 4, 4), (1, 2), 1, 0, True)])
def test_wide_panel_adjoint_by_grid_bricks(hip, monkeypatch, dims, shape, alpha, ld_pad, blob):
    """ig_ccsrmm_t_bricks_wide_grid (k_bricks_wide64r: the brick image in registers, addressed through the VGPR index mode):
    a gridding-like matrix (3 x 3 x 3 taps around a point of an n0 x nm x ns grid, wrapped) times a 64-column panel, bricks of
    16 x bm x bs grid points -- against scipy in double precision; untouched rows zero over a sentinel; a hot spot makes
    heavy bricks (shared pieces, atomics); the cube is guessed from the column count, other grids come through the hint"""
    n0, nm, ns = dims
    K = n0 * nm * ns
    M = 6000
    rng = np.random.default_rng(K + shape[0])
    c = np.stack([rng.integers(0, n0, M), rng.integers(0, nm, M), rng.integers(0, ns, M)])
    if blob:
        hot = rng.random(M) < 0.6
        c[:, hot] = np.stack([rng.integers(16, 20, hot.sum()), rng.integers(2, 4, hot.sum()), rng.integers(0, 2, hot.sum())])
    c[:, ::11] = 0                                                        # a few samples at the corner: taps wrap around
    d = np.stack(np.meshgrid(np.arange(-1, 2), np.arange(-1, 2), np.arange(-1, 2), indexing='ij')).reshape(3, 27)
    p = (c[:, :, None] + d[:, None, :]) % np.array(dims)[:, None, None]
    cols = (p[0] + n0 * (p[1] + nm * p[2])).reshape(-1)
    rows = np.repeat(np.arange(M), 27)
    A = spp.csr_matrix((rand64c(rows.size, seed=1), (rows, cols)), shape=(M, K))
    A.sum_duplicates(); A.sort_indices()
    monkeypatch.setitem(hip.tuning, "wide_brick_shape", shape)
    A_d = hip.csr_matrix(hip, A)
    if dims[0] != dims[1]:
        A_d.set_grid_dims(*dims)
    xfull = rand64c(M + ld_pad, 64, seed=2)
    yfull = np.full((K + ld_pad, 64), 7 - 3j, dtype=C64, order='F')
    x_d = hip.copy_array(xfull)[0:M, :]
    y_d = hip.copy_array(yfull)[0:K, :]
    A_d.adjoint(y_d, x_d, alpha=alpha)
    assert A_d._wide is not None and A_d._wide['ntasks'] > 0 and A_d._wide['geom'] == (n0, nm) + shape
    exp = alpha * (A.conj().T.astype(np.complex128) @ xfull[:M].astype(np.complex128))
    got = y_d.to_host()
    assert rel_err(got, exp) < RTOL
    untouched = np.setdiff1d(np.arange(K), np.unique(A.indices))
    assert np.all(got[untouched] == 0)
    if blob:
        assert np.any(np.asarray(A_d._wide['tasks'].to_host()).view(np.int32).reshape(-1, 4)[:, 3] >> 16)   # shared pieces ran
    # a second product into the same result: every tile is written again, nothing accumulates
    A_d.adjoint(y_d, x_d, alpha=alpha)
    assert rel_err(y_d.to_host(), exp) < RTOL


def test_wide_panel_adjoint_cube_of_columns_that_is_no_grid(hip):
    """32^3 columns look like a grid to the guess, but the rows of a random matrix do not cluster on it: quads would be three
    quarters padding, so the format keeps 16-row bricks -- and the product is right either way"""
    M, K = 3000, 32 ** 3
    rng = np.random.default_rng(5)
    rows = np.repeat(np.arange(M), 20)
    cols = rng.integers(0, K, size=rows.size)
    A = spp.csr_matrix((rand64c(rows.size, seed=1), (rows, cols)), shape=(M, K))
    A.sum_duplicates(); A.sort_indices()
    A_d = hip.csr_matrix(hip, A)
    x = rand64c(M, 64, seed=2)
    y_d = hip.copy_array(np.full((K, 64), 3 + 1j, dtype=C64, order='F'))
    A_d.adjoint(y_d, hip.copy_array(x))
    assert A_d._wide is not None and A_d._wide['geom'] == (K, 1, 1, 1)
    assert rel_err(y_d.to_host(), A.conj().T.astype(np.complex128) @ x.astype(np.complex128)) < RTOL


@pytest.mark.parametrize("n,frac,alpha,beta,ld_pad", [(64, 0.3, 1, 0, 0), (64, 0.05, 0.5 - 1j, 1.5, 7), (32, 0.5, 1, 1, 0), (16, 0.3, 2, 0, 3),
                                                     (17, 0.3, 1, 0.5j, 0), (48, 0.6, 1, 0, 0)])
def test_wide_panel_forward_over_touched_rows(hip, monkeypatch, n, frac, alpha, beta, ld_pad):
    """ig_ccsrmm_xrows: a wide panel (16..64 columns) of which the matrix touches a fraction of the rows -- only those are
    repacked; same product as scipy and as the whole-panel route, with alpha/beta and sliced (padded leading dimension)
    panels, also after the adjoint has consumed the host copy of the matrix"""
    M, K = 3000, 20000
    rng = np.random.default_rng(n * 10 + int(frac * 100))
    cols_used = np.sort(rng.choice(K, size=int(K * frac), replace=False))
    rows = np.repeat(np.arange(M), 27)
    cols = cols_used[rng.integers(0, cols_used.size, size=rows.size)]
    A = spp.csr_matrix((rand64c(rows.size, seed=1), (rows, cols)), shape=(M, K))
    A.sum_duplicates(); A.sort_indices()
    A_d = hip.csr_matrix(hip, A)
    assert A_d._col_frac <= 0.6
    xfull = rand64c(K + ld_pad, n, seed=2)
    yfull = rand64c(M + ld_pad, n, seed=3)
    x_d = hip.copy_array(xfull)[0:K, :]
    y_d = hip.copy_array(yfull)[0:M, :]
    A_d.forward(y_d, x_d, alpha=alpha, beta=beta)
    assert getattr(A_d, '_xrows', None) is not None and A_d._xrows[0].size == np.unique(A.indices).size
    exp = alpha * (A.astype(np.complex128) @ xfull[:K].astype(np.complex128)) + beta * yfull[:M]
    assert rel_err(y_d.to_host(), exp) < RTOL
    # the whole-panel route gives the same
    monkeypatch.setitem(hip.tuning, "xrows", False)
    B_d = hip.csr_matrix(hip, A)
    y2 = hip.copy_array(yfull)[0:M, :]
    B_d.forward(y2, x_d, alpha=alpha, beta=beta)
    assert getattr(B_d, '_xrows', None) is None
    assert rel_err(y2.to_host(), y_d.to_host()) < 1e-6
    monkeypatch.setitem(hip.tuning, "xrows", True)
    # adjoint first (the host copy goes into the transpose), then the forward: the column list comes from the device copy
    C_d = hip.csr_matrix(hip, A)
    xa = hip.copy_array(rand64c(M, 2, seed=4))
    C_d.adjoint(hip.zero_array((K, 2), C64), xa)
    y3 = hip.copy_array(yfull)[0:M, :]
    C_d.forward(y3, x_d, alpha=alpha, beta=beta)
    assert rel_err(y3.to_host(), exp) < RTOL


@pytest.mark.parametrize("M,n,K,alpha,beta", itertools.product([23, 45], [1, 8, 17], [18, 19], [0.0, 0.5, 1.5], [0.0, 1.0, 1.5]))
def test_exwrite_csr_matrix(hip, M, n, K, alpha, beta):
    """at most one nonzero per column (reference test_backends.py:213-243)"""
    rng = np.random.default_rng(M * 100 + K)
    counts = rng.integers(0, 2, K)
    ptr = np.concatenate([[0], np.cumsum(counts)])
    cols = rng.integers(0, M, counts.sum())
    A = spp.csr_matrix((rand64c(int(counts.sum()), seed=5), cols, ptr), shape=(K, M)).T.tocsr()
    A_d = hip.csr_matrix(hip, A)
    assert A_d._exwrite
    x, y = rand64c(K, n, seed=1), rand64c(M, n, seed=2)
    y_d = hip.copy_array(y)
    A_d.forward(y_d, hip.copy_array(x), alpha=alpha, beta=beta)
    np.testing.assert_allclose(y_d.to_host(), beta * y + alpha * (A @ x), atol=1e-5)
    x, y = rand64c(M, n, seed=3), rand64c(K, n, seed=4)
    y_d = hip.copy_array(y)
    A_d.adjoint(y_d, hip.copy_array(x), alpha=alpha, beta=beta)
    np.testing.assert_allclose(y_d.to_host(), beta * y + alpha * (A.conj().T @ x), atol=1e-5)


def test_csrmm_strided_panels_and_complex_scalars(hip):
    """X and Y are row-slices of wider panels: leading dimension > rows"""
    M, K, n = 37, 29, 5
    A = randM(M, K, 0.3, seed=11)
    A_d = hip.csr_matrix(hip, A)
    Xbig, Ybig = rand64c(K + 6, n, seed=12), rand64c(M + 9, n, seed=13)
    Xd, Yd = hip.copy_array(Xbig), hip.copy_array(Ybig)
    alpha, beta = 0.7 - 0.2j, -0.3 + 0.4j
    A_d.forward(Yd[4:4 + M, :], Xd[2:2 + K, :], alpha=alpha, beta=beta)
    exp = Ybig.copy()
    exp[4:4 + M] = beta * Ybig[4:4 + M] + alpha * (A @ Xbig[2:2 + K])
    np.testing.assert_allclose(Yd.to_host(), exp, atol=1e-5)
    for policy in ("transpose", "atomic"):
        hip.adjoint_policy = policy
        Xd, Yd = hip.copy_array(Ybig), hip.copy_array(Xbig)
        A_d.adjoint(Yd[2:2 + K, :], Xd[4:4 + M, :], alpha=alpha, beta=beta)
        exp = Xbig.copy()
        exp[2:2 + K] = beta * Xbig[2:2 + K] + alpha * (A.conj().T @ Ybig[4:4 + M])
        np.testing.assert_allclose(Yd.to_host(), exp, atol=1e-5)
    hip.adjoint_policy = "transpose"


def test_csrmm_edge_cases(hip):
    # empty matrix: y = beta*y, and beta == 0 must overwrite NaNs
    A = spp.csr_matrix((40, 30), dtype=C64)
    A_d = hip.csr_matrix(hip, A)
    y = rand64c(40, 3, seed=1)
    y_d = hip.copy_array(y)
    A_d.forward(y_d, hip.copy_array(rand64c(30, 3, seed=2)), alpha=1, beta=0.5)
    np.testing.assert_allclose(y_d.to_host(), 0.5 * y, atol=1e-6)
    nan_d = hip.copy_array(np.full((40, 3), np.nan, dtype=C64, order='F'))
    A_d.forward(nan_d, hip.copy_array(rand64c(30, 3, seed=2)))
    assert not nan_d.to_host().any()
    nan_d = hip.copy_array(np.full((30, 3), np.nan, dtype=C64, order='F'))
    A_d.adjoint(nan_d, hip.copy_array(rand64c(40, 3, seed=2)))
    assert not nan_d.to_host().any()
    # ragged rows: one dense row, many empty ones, a long tail
    rng = np.random.default_rng(3)
    rows = np.concatenate([np.zeros(500, int), rng.integers(0, 200, 300)])
    cols = np.concatenate([rng.permutation(700)[:500], rng.integers(0, 700, 300)])
    A = spp.coo_matrix((rand64c(800, seed=4), (rows, cols)), shape=(200, 700)).tocsr()
    A_d = hip.csr_matrix(hip, A)
    for n in (1, 8, 64):
        x = rand64c(700, n, seed=5)
        y_d = hip.zero_array((200, n), C64)
        A_d.forward(y_d, hip.copy_array(x))
        np.testing.assert_allclose(y_d.to_host(), A @ x, rtol=RTOL, atol=1e-4)
        xa = rand64c(200, n, seed=6)
        ya_d = hip.zero_array((700, n), C64)
        A_d.adjoint(ya_d, hip.copy_array(xa))
        np.testing.assert_allclose(ya_d.to_host(), A.conj().T @ xa, rtol=RTOL, atol=1e-4)


@pytest.mark.parametrize("N", [1, 2, 4, 8])
def test_csrmm_sparse_rows_and_interleaved_panels(hip, N):
    """mostly-empty rows with a few very long ones (a transposed gridding matrix in miniature): the dense-lane
    adjoint, its deferred-row lists, and -- for 2/4/8 columns -- the row-major (coil-interleaved) panel entry points
    ig_ccsrmm_il / ig_ccsrmm_t_grid_il / ig_csum_il against scipy on the same numbers"""
    rng = np.random.default_rng(11)
    T, P = 6000, 50000                                # A: T x P, 9 taps per row; A^T has ~1 nonzero per row
    cols = np.concatenate([rng.integers(0, P, (T, 8)), rng.integers(100, 104, (T, 1))], axis=1)   # columns 100..103: ~1500 each
    A = spp.csr_matrix((rand64c(T * 9, seed=1), cols.reshape(-1), np.arange(0, T * 9 + 1, 9)), shape=(T, P))
    A.sum_duplicates()
    A_d = hip.csr_matrix(hip, A)
    x, k = rand64c(P, N, seed=2), rand64c(T, N, seed=3)
    y_d, z_d = hip.zero_array((T, N), C64), hip.copy_array(np.full((P, N), np.nan, dtype=C64, order='F'))
    A_d.forward(y_d, hip.copy_array(x))
    A_d.adjoint(z_d, hip.copy_array(k))
    ref_f, ref_a = A @ x, A.conj().T @ k
    assert rel_err(y_d.to_host(), ref_f) < RTOL and rel_err(z_d.to_host(), ref_a) < RTOL
    if N == 1:
        return
    B_d = hip.csr_matrix(hip, A)
    B_d.set_grid_interleaved(True)
    il = lambda a: np.asfortranarray(np.ascontiguousarray(a).reshape(-1).reshape(a.shape, order='F'))   # memory (i, c) -> i*N + c
    un = lambda a: np.asfortranarray(a).reshape(-1, order='F').reshape(a.shape)
    y_d = hip.zero_array((T, N), C64)
    B_d.forward(y_d, hip.copy_array(il(x)), alpha=0.5 - 2j)
    assert rel_err(y_d.to_host(), (0.5 - 2j) * ref_f) < RTOL
    z_d = hip.copy_array(np.full((P, N), np.nan, dtype=C64, order='F'))
    B_d.adjoint(z_d, hip.copy_array(k), alpha=2.0)
    assert rel_err(un(z_d.to_host()), 2.0 * ref_a) < RTOL
    s_d = hip.copy_array(rand64c(P, 1, seed=4))
    s0 = s_d.to_host().copy()
    hip.sum_columns(s_d, z_d, alpha=1j, beta=0.5, interleaved=True)
    np.testing.assert_allclose(s_d.to_host()[:, 0], 0.5 * s0[:, 0] + 1j * 2.0 * ref_a.sum(axis=1), rtol=1e-4, atol=1e-3)
    with pytest.raises(AssertionError):
        B_d.adjoint(z_d, hip.copy_array(k), beta=1.0)             # interleaved adjoint is beta = 0 only


def test_interleaved_entry_points_reject_bad_arguments(hip):
    import ctypes
    L, ctx = hip._L, hip._ctx
    a = hip.zero_array((64,), C64)
    i32 = hip.copy_array(np.zeros(65, dtype=np.int32))
    vp = lambda d: ctypes.c_void_p(d._arr)
    # 3 columns: not a power of two
    assert L.ig_ccsrmm_il(ctx, 4, 4, 3, 0, 1.0, 0.0, vp(a), vp(i32), vp(i32), vp(a), 0.0, 0.0, vp(a), 4) != 0
    # 16 columns: the interleaved adjoint supports 2, 4, 8
    assert L.ig_ccsrmm_t_grid_il(ctx, 4, 4, 16, 0, 1.0, 0.0, vp(a), vp(i32), vp(i32), vp(a), 4, vp(a), None, 0, 0) != 0
    assert b"columns" in L.ig_last_error(ctx)
    # coil-summing cropped transform needs a layout-2 plan
    plan, ws = hip._padded_plan((256, 256, 256), (64, 64, 64), (128, 128, 128), 2, 1)
    assert L.ig_fft_exec_cropped_sum(plan, vp(a), vp(a), vp(a), vp(a), None) != 0
    # layout 2 with 3 coils is refused at plan time
    with pytest.raises(RuntimeError):
        hip._padded_plan((256, 256, 256), (64, 64, 64), (128, 128, 128), 3, 2)


def test_csrmm_config1_spmm_example(hip, oracle_backend):
    """BASELINE config 1 (examples/spmm.py scaled up): 1e4 x 1e4, 1 % nnz, 8 RHS, vs the numpy oracle"""
    rng = np.random.default_rng(1)
    A = spp.random(10000, 10000, density=0.01, format='csr', random_state=rng, dtype=np.float32).astype(C64)
    x = rand64c(10000, 8, seed=1)
    S_h, S_o = hip.SpMatrix(A), oracle_backend.SpMatrix(A)
    y_h = S_h * x
    y_o = S_o * x
    assert rel_err(y_h, y_o) < RTOL
    z_h = S_h.H * x
    z_o = S_o.H * x
    assert rel_err(z_h, z_o) < RTOL


# ---------------------------------------------------------------------------------------
# FFT
# ---------------------------------------------------------------------------------------
def test_fft_golden(hip):
    g = golden("leaf_fft")
    for i in range(int(g["count"])):
        x = g["f%d_x" % i]
        y_d = hip.zero_array(x.shape, C64)
        hip.fftn(y_d, hip.copy_array(x))
        assert rel_err(y_d.to_host(), g["f%d_fwd" % i]) < RTOL, (i, x.shape, hip.fft_describe(x.shape))
        hip.ifftn(y_d, hip.copy_array(x))
        assert rel_err(y_d.to_host(), g["f%d_inv" % i]) < RTOL, (i, x.shape, hip.fft_describe(x.shape))


def _check_fft(hip, shape, batch, seed=0):
    x = rand64c(*(shape + (batch,)), seed=seed)
    axes = tuple(range(len(shape)))
    x_d = hip.copy_array(x)
    y_d = hip.zero_array(x.shape, C64)
    hip.fftn(y_d, x_d)
    w = y_d.to_host()
    assert rel_err(w, np.fft.fftn(x.astype(np.complex128), axes=axes)) < RTOL, hip.fft_describe(x.shape)
    np.testing.assert_equal(x_d.to_host(), x)                 # out of place leaves the input alone
    z_d = hip.zero_array(x.shape, C64)
    hip.ifftn(z_d, y_d)
    n = np.prod(shape)
    assert rel_err(z_d.to_host() / n, x) < RTOL               # unnormalised round trip = n * identity
    hip.ifftn(y_d, y_d)                                       # in place
    assert rel_err(y_d.to_host() / n, x) < RTOL


@pytest.mark.parametrize("batch,x,y,z", itertools.product([1, 2, 8], [23, 24, 25], [23, 25], [24, 25]))
def test_fft_3d_small_sizes(hip, batch, x, y, z):
    """reference test_backends.py:153-180 sizes incl. primes"""
    _check_fft(hip, (z, y, x), batch, seed=x * y * z)


@pytest.mark.parametrize("shape", [(2,), (4,), (8,), (64,), (512,), (1024,), (4096,), (8192,), (6,), (12,), (30,),
                                   (210,), (320,), (35,), (49,), (97,), (121,), (1009,), (2 * 1009,),
                                   (32, 32), (64, 48), (320, 20), (100, 101), (128, 128), (512, 6),
                                   (16, 16, 16), (32, 8, 64), (20, 30, 12), (64, 64, 64), (40, 48, 56), (1, 1, 8), (8, 1, 1)])
def test_fft_shapes(hip, shape):
    _check_fft(hip, tuple(shape), 3, seed=sum(shape))


def test_fft_more_rows_than_a_launch_grid_dimension_holds(hip):
    """the pass kernels take (tile, k1, k2) as a three-dimensional launch grid where the extents allow (pass_grid, ig_fft.hip); with
    more than 65535 rows of columns -- here a batch of 66000 transforms of 4 x 256 points: the 256-point axis steps 4 elements, its
    rows are the batch members -- the launcher falls back to the linear grid and the kernel to its divisions.  Both routes vs numpy."""
    _check_fft(hip, (4, 256), 66000, seed=11)
    _check_fft(hip, (4, 256), 300, seed=12)


AB_LENGTHS = [160, 192, 224, 240, 270, 288, 320, 360, 384, 392, 400, 432, 480, 576, 600, 640,
              # further lengths of the generated list (tools/gen_ab_list.py): odd ones, 5 x 25, the longest splits (four exchange rounds)
              96, 105, 125, 147, 243, 375, 441, 567, 625, 675, 729, 840, 1000 - 40, 1024]


@pytest.mark.parametrize("n", AB_LENGTHS)
def test_fft_two_stage_any_length(hip, n):
    """the register-resident A x B passes (ig_fft_ab.h; lengths of the reference's example grids, examples/pics.py:87-90)
    against numpy: as contiguous lines (ragged last tile: 21 lines), as a strided middle axis, as a strided last axis with
    5 columns, forward + inverse + in place; and the same transform through the LDS kernel"""
    assert "AxB" in hip.fft_describe((n, 3))
    _check_fft(hip, (n,), 21, seed=n)
    _check_fft(hip, (6, n), 3, seed=n + 1)
    _check_fft(hip, (5, 3, n), 2, seed=n + 2)


def test_fft_two_stage_any_length_3d_and_lds_agreement(hip, monkeypatch):
    """three different A x B lengths in one 3-D transform (axis 0 contiguous, axes 1 and 2 strided, batch 2) against numpy, and
    against the multi-stage LDS kernel on the same data"""
    from indigo_amd.backends import get_backend
    shape = (320, 270, 288)
    _check_fft(hip, shape, 2, seed=7)
    x = rand64c(*(shape + (1,)), seed=8)
    y_d = hip.zero_array(x.shape, C64)
    hip.fftn(y_d, hip.copy_array(x))
    other = get_backend("hip")
    other.set_option("fft.kernels", 1)                     # plans of this context: no A x B passes
    assert "AxB" not in other.fft_describe(x.shape) and "lds" in other.fft_describe(x.shape)
    z_d = other.zero_array(x.shape, C64)
    other.fftn(z_d, other.copy_array(x))
    assert rel_err(z_d.to_host(), y_d.to_host()) < 1e-6


@pytest.mark.parametrize("n", [277, 410, 139, 1009, 69, 102, 205, 2 * 1009, 37 * 11])
def test_fft_chirp_z_lengths(hip, n):
    """lengths with a prime factor above 7 -- what int(N * osf) of the reference's driver produces (208 -> 277, 308 -> 410:
    indigo/backends/backend.py:427-430) -- run as chirp-z (Bluestein) transforms over a smooth length m >= 2 n - 1: two fused
    launches of the A x B kernel on strided axes (m <= 1024), five steps on contiguous lines and for longer m; against numpy:
    contiguous lines (ragged tile), strided middle and last axes, 3-D with two chirp-z axes, forward + inverse + in place;
    and against the one-stage-per-launch kernel"""
    from indigo_amd.backends import get_backend
    assert "chirp-z" in hip.fft_describe((n, 3)), hip.fft_describe((n, 3))
    _check_fft(hip, (n,), 5, seed=n)
    _check_fft(hip, (6, n), 3, seed=n + 1)
    _check_fft(hip, (32, n), 2, seed=n + 2)               # 32 lanes of columns: the fused route where m <= 1024
    _check_fft(hip, (20, 3, n), 2, seed=n + 3)
    if n <= 512:
        _check_fft(hip, (48, n, 5), 2, seed=n + 4)
        assert "one fused launch" in hip.fft_describe((48, n, 5, 2))
    x = rand64c(16, n, 3, seed=n + 5)
    y_d = hip.zero_array(x.shape, C64)
    hip.fftn(y_d, hip.copy_array(x))
    other = get_backend("hip")
    other.set_option("fft.kernels", 2)
    assert "chirp-z" not in other.fft_describe(x.shape)
    z_d = other.zero_array(x.shape, C64)
    other.fftn(z_d, other.copy_array(x))
    assert rel_err(z_d.to_host(), y_d.to_host()) < 2e-6


def test_fft_two_chirp_z_axes_of_the_reference_drivers_default_grid(hip):
    """(160, 69, 102) and (320, 138, 205): the default oversampling 640/480 of examples/pics.py:86 on images 120 x 52 x 77 and
    240 x 104 x 154 -- one A x B axis and two chirp-z axes in one transform"""
    for shape in ((160, 69, 102), (320, 138, 205)):
        d = hip.fft_describe(shape + (2,))
        assert d.count("chirp-z") == 2 and "AxB" in d, d
        _check_fft(hip, shape, 2, seed=shape[1])


def test_fft_generic_path_matches_lds_path(hip, monkeypatch):
    """the global-memory fallback and the LDS kernels compute the same transform"""
    from indigo_amd.backends import get_backend
    x = rand64c(24, 20, 16, 2, seed=9)
    y_d = hip.zero_array(x.shape, C64)
    hip.fftn(y_d, hip.copy_array(x))
    other = get_backend("hip")
    other.set_option("fft.kernels", 2)                     # plans of this context: the one-stage-per-launch kernel only
    assert "generic" in other.fft_describe(x.shape)
    with pytest.raises(RuntimeError, match="unknown option"):
        other.set_option("no.such.option", 1)
    z_d = other.zero_array(x.shape, C64)
    other.fftn(z_d, other.copy_array(x))
    assert rel_err(z_d.to_host(), y_d.to_host()) < 1e-6


def test_fft_config2_shape_properties(hip):
    """BASELINE config 2 shape (256^3, reduced batch): Parseval + round trip + impulse response"""
    n, batch = 256, 2
    x = rand64c(n, n, n, batch, seed=2)
    x_d = hip.copy_array(x)
    y_d = hip.zero_array(x.shape, C64)
    hip.fftn(y_d, x_d)
    e_in, e_out = hip.norm2(x_d), hip.norm2(y_d)
    np.testing.assert_allclose(e_out, e_in * n ** 3, rtol=1e-5)           # Parseval, unnormalised
    dc = y_d.reshape((x.size,))[0:1].to_host()[0]
    np.testing.assert_allclose(dc, x[..., 0].astype(np.complex128).sum(), rtol=1e-4)    # DC bin of volume 0
    hip.ifftn(y_d, y_d)
    hip.axpby(1.0 / n ** 3, y_d, -1.0, x_d)
    assert np.sqrt(hip.norm2(y_d) / e_in) < RTOL
    # impulse at (1, 2, 3) -> separable complex exponential
    imp = np.zeros((n, n, n, 1), dtype=C64, order='F')
    imp[1, 2, 3, 0] = 1
    y1 = hip.zero_array(imp.shape, C64)
    hip.fftn(y1, hip.copy_array(imp))
    k = np.arange(n)
    e = lambda s: np.exp(-2j * np.pi * s * k / n)
    exp = e(1)[:, None, None] * e(2)[None, :, None] * e(3)[None, None, :]
    assert rel_err(y1.to_host()[..., 0], exp) < RTOL


# ---------------------------------------------------------------------------------------
# ones / DIA / dense leaves and apgd (reference test_backends.py:333-361,377-398,440-523)
# ---------------------------------------------------------------------------------------
def test_misc_leaves_golden(hip):
    from conftest import check_misc_leaves
    check_misc_leaves(hip, RTOL)


@pytest.mark.parametrize("m,n,k,alpha,beta,forward", itertools.product([10, 23, 129, 144], [10, 129], [23, 144], [1, 0.5, 0.0],
                                                                       [0, 0.5], [True, False]))
def test_cgemm_grid(hip, m, n, k, alpha, beta, forward):
    y, M, x = rand64c(m, n, seed=1), rand64c(m, k, seed=2), rand64c(k, n, seed=3)
    if not forward:
        x, y = y, x
    exp = alpha * ((M if forward else np.conj(M.T)).astype(np.complex128) @ x) + beta * y
    y_d = hip.copy_array(y)
    hip.cgemm(y_d, hip.copy_array(M), hip.copy_array(x), alpha, beta, forward=forward)
    assert rel_err(y_d.to_host(), exp) < RTOL


@pytest.mark.parametrize("m,k,alpha,beta,left", itertools.product([2, 4, 5, 6, 70], [1, 2, 3, 65], [0.0, 0.5, 1.5], [0.0, 1.0, 1.5],
                                                                  [True, False]))
def test_csymm_grid(hip, m, k, alpha, beta, left):
    S = rand64c(m, m, seed=4)
    S = np.asfortranarray((S + S.T).real.astype(C64))
    x = rand64c(m, k, seed=5) if left else rand64c(k, m, seed=5)
    y = rand64c(m, k, seed=6) if left else rand64c(k, m, seed=6)
    exp = alpha * (S @ x if left else x @ S) + beta * y
    y_d = hip.copy_array(y)
    hip.csymm(y_d, hip.copy_array(S), hip.copy_array(x), alpha, beta, left)
    assert rel_err(y_d.to_host(), exp) < RTOL


@pytest.mark.parametrize("M,K,N,alpha,beta,maxoffsets", itertools.product([23, 45, 1000], [45, 23], [1, 8, 9, 17], [0, 0.5, 1.5],
                                                                          [0, 1.0, 1.5], [1, 2, 4]))
def test_dia_matrix_grid(hip, M, K, N, alpha, beta, maxoffsets):
    rng = np.random.default_rng(M * K + N + maxoffsets)
    offsets = np.array(sorted(set(rng.integers(-K, M + K, size=maxoffsets).tolist())), dtype=np.int32)
    data = rand64c(offsets.size, K, seed=7, order='C')
    A = spp.dia_matrix((data, offsets), shape=(M, K))
    A_d = hip.dia_matrix(hip, A)
    x, y = rand64c(K, N, seed=8), rand64c(M, N, seed=9)
    y_d = hip.copy_array(y)
    A_d.forward(y_d, hip.copy_array(x), alpha=alpha, beta=beta)
    np.testing.assert_allclose(y_d.to_host(), beta * y + alpha * (A @ x), atol=1e-5, rtol=1e-5)
    x_d = hip.copy_array(x)
    A_d.adjoint(x_d, hip.copy_array(y), alpha=alpha, beta=beta)
    np.testing.assert_allclose(x_d.to_host(), beta * x + alpha * (A.conjugate().transpose() @ y), atol=1e-5, rtol=1e-5)
    # panels with a leading dimension (views into wider arrays)
    big = hip.copy_array(rand64c(M + 5, N, seed=10))
    view = big[2:M + 2, :]
    view.copy_from(y)
    A_d.forward(view, hip.copy_array(x), alpha=alpha, beta=beta)
    np.testing.assert_allclose(view.to_host(), beta * y + alpha * (A @ x), atol=1e-5, rtol=1e-5)


def test_one_dense_and_dia_operators(hip, oracle_backend):
    """One, DenseMatrix and SpMatrix(_use_dia) operators on the GPU == the oracle's (reference test_operators.py:471-489,549-642)"""
    for B in (hip, oracle_backend):
        B._scratch = None
    x = rand64c(40, 6, seed=1)
    outs = []
    for B in (hip, oracle_backend):
        one = B.One((33, 40))
        D = B.DenseMatrix(rand64c(33, 40, seed=2))
        S = B.SpMatrix(spp.diags([rand64c(40, seed=3), rand64c(39, seed=4)], offsets=[0, 1]).astype(C64), name='banded')
        S._use_dia = True
        T = (one + D) * S
        k = rand64c(33, 6, seed=5)
        outs.append((T * x, T.H * k, (0.5 * D).H * k))
        assert type(S._matrix_d).__name__ == "dia_matrix"
    for a, b in zip(*outs):
        assert rel_err(a, b) < RTOL


@pytest.mark.parametrize("N,bm,bs,chunk,run,with_support", [(8, 4, 4, 4096, 1024, True), (8, 4, 4, 64, 1024, True), (4, 4, 4, 64, 64, True),
                                                            (4, 4, 8, 4096, 1024, False), (8, 2, 8, 4096, 1 << 20, False), (8, 1, 1, 8, 256, True),
                                                            (8, 2, 2, 4096, 1024, True), (8, 2, 2, 128, 8, False), (4, 1, 2, 4096, 1 << 20, True),
                                                            (8, 2, 2, 4096, 4096, 8), (8, 2, 2, 64, 512, 4), (4, 2, 2, 4096, 1024, 8), (8, 4, 2, 4096, 1024, 8)])
def test_brick_binned_adjoint_gridding(hip, N, bm, bs, chunk, run, with_support):
    """ig_ccsrmm_t_bricks (scatter through per-wave LDS images, binned by grid bricks) == A^H X from scipy, for interleaved
    result panels of 4 and 8 columns; shared bricks (small chunk: several tasks add into one brick with atomics), rows
    with more taps than one wave trip, several brick shapes, runs of one brick up to the 64-brick / 256-segment cap, and
    the support table (only flagged segments are written)."""
    rng = np.random.default_rng(N * 100 + bm)
    n0, nm, ns = 64, 64, 128
    P, T = n0 * nm * ns, 3000
    # clustered samples (a dense blob -> shared bricks) plus scattered ones; 27..70 taps per row
    centre = rng.integers(0, P, size=40)
    rows, cols = [], []
    for t in range(T):
        ntap = 70 if t % 97 == 0 else 27
        base = centre[t % 40] if t % 3 else rng.integers(0, P)
        off = rng.integers(-3, 4, size=(ntap, 3))
        kx, km, ks = base % n0, (base // n0) % nm, base // (n0 * nm)
        c = ((kx + off[:, 0]) % n0) + n0 * (((km + off[:, 1]) % nm) + nm * ((ks + off[:, 2]) % ns))
        c = np.unique(c)
        rows.append(np.full(c.size, t)); cols.append(c)
    rows, cols = np.concatenate(rows), np.concatenate(cols)
    A = spp.csr_matrix((rand64c(rows.size, seed=3), (rows, cols)), shape=(T, P))
    A.sort_indices()
    A_d = hip.csr_matrix(hip, A)
    tile = 16
    if with_support:
        tile = with_support if with_support in (4, 8) else 16              # kx points per entry of the support table
        seg = np.zeros((ns, nm, n0 // tile), dtype=bool)                   # [ks][km][kx tile]
        uc = np.unique(A.indices)
        seg[uc // (n0 * nm), (uc // n0) % nm, (uc % n0) // tile] = True
        from test_hip_operators import support_table_from_segments
        flat, _ = support_table_from_segments(seg)
        if tile == 16:
            A_d.set_grid_support(flat, n0, nm)
        else:
            A_d.set_grid_support_fine(flat, tile)
    A_d.set_grid_interleaved(True)
    A_d.set_grid_bricks(n0, nm, ns, ncols=N, bm=bm, bs=bs, chunk=chunk, run=run)
    assert A_d._bricks['nshared'] > 0 or chunk >= 4096
    X = rand64c(T, N, seed=5)
    sentinel = np.full((P, N), 9 - 2j, dtype=C64, order='F')
    y_d = hip.copy_array(sentinel)
    A_d.adjoint(y_d, hip.copy_array(X), alpha=0.5 - 0.25j)
    got = y_d.to_host().reshape(-1, order='F').reshape(P, N)              # row-major (interleaved) memory
    exp = (0.5 - 0.25j) * (A.conj().T.astype(np.complex128) @ X.astype(np.complex128))
    if with_support:
        inside = np.repeat(seg, tile, axis=2).reshape(-1)                 # kx fastest, then km, then ks
        assert rel_err(got[inside], exp[inside]) < RTOL
        np.testing.assert_array_equal(got[~inside], sentinel[~inside])    # rows outside the support are not touched
        assert np.abs(exp[~inside]).max() == 0
    else:
        assert rel_err(got, exp) < RTOL
    # another column count than the binned format was padded for: the gather over the transpose serves it
    X2 = rand64c(T, 2, seed=6)
    y2 = hip.zero_array((P, 2), C64)
    A_d.adjoint(y2, hip.copy_array(X2))
    got2 = y2.to_host().reshape(-1, order='F').reshape(P, 2)
    exp2 = A.conj().T.astype(np.complex128) @ X2.astype(np.complex128)
    assert rel_err(got2[inside] if with_support else got2, exp2[inside] if with_support else exp2) < RTOL


@pytest.mark.parametrize("N,bm,bs,chunk,run,with_support", [(1, 4, 4, 256, 64, True), (2, 4, 4, 256, 64, True), (1, 2, 2, 4, 8, True),
                                                            (2, 4, 4, 2, 64, False), (4, 2, 4, 256, 96, True), (1, 4, 8, 256, 1 << 20, False),
                                                            (2, 1, 1, 256, 3, True), (4, 4, 4, 8, 64, False)])
def test_slot_format_adjoint_gridding(hip, N, bm, bs, chunk, run, with_support):
    """ig_ccsrmm_t_slots (the brick scatter for 1, 2 and 4 interleaved columns: a lane is an entry, the entries of a brick
    reordered so that a slot never holds a cell twice) == A^H X from scipy: a dense blob of samples (cells hit hundreds of times:
    many slots per brick, shared bricks with atomics at a small chunk), rows with more taps than a slot holds cells of their
    brick, several brick shapes, runs from a few slots up to the brick cap, with and without the support table."""
    rng = np.random.default_rng(N * 100 + bm)
    n0, nm, ns = 64, 64, 128
    P, T = n0 * nm * ns, 3000
    centre = rng.integers(0, P, size=40)
    rows, cols = [], []
    for t in range(T):
        ntap = 70 if t % 97 == 0 else 27
        base = centre[t % 40] if t % 3 else rng.integers(0, P)
        off = rng.integers(-3, 4, size=(ntap, 3))
        kx, km, ks = base % n0, (base // n0) % nm, base // (n0 * nm)
        c = np.unique(((kx + off[:, 0]) % n0) + n0 * (((km + off[:, 1]) % nm) + nm * ((ks + off[:, 2]) % ns)))
        rows.append(np.full(c.size, t)); cols.append(c)
    rows, cols = np.concatenate(rows), np.concatenate(cols)
    A = spp.csr_matrix((rand64c(rows.size, seed=3), (rows, cols)), shape=(T, P))
    A.sort_indices()
    A_d = hip.csr_matrix(hip, A)
    if with_support:
        seg = np.zeros((ns, nm, n0 // 16), dtype=bool)
        uc = np.unique(A.indices)
        seg[uc // (n0 * nm), (uc // n0) % nm, (uc % n0) // 16] = True
        from test_hip_operators import support_table_from_segments
        flat, _ = support_table_from_segments(seg)
        A_d.set_grid_support(flat, n0, nm)
    if N > 1:
        A_d.set_grid_interleaved(True)
    A_d.set_grid_slots(n0, nm, ns, ncols=N, bm=bm, bs=bs, chunk=chunk, run=run)
    sl = A_d._slots
    assert sl is not None and sl['nentries'] == A.nnz and 0 < sl['nslots'] <= A.nnz
    assert sl['nshared'] > 0 or chunk >= 256
    X = rand64c(T, N, seed=5)
    sentinel = np.full((P, N), 9 - 2j, dtype=C64, order='F')
    y_d = hip.copy_array(sentinel)
    A_d.adjoint(y_d, hip.copy_array(X), alpha=0.5 - 0.25j)
    got = y_d.to_host().reshape(-1, order='F').reshape(P, N)              # row-major (interleaved) memory
    exp = (0.5 - 0.25j) * (A.conj().T.astype(np.complex128) @ X.astype(np.complex128))
    if with_support:
        inside = np.repeat(seg, 16, axis=2).reshape(-1)
        assert rel_err(got[inside], exp[inside]) < RTOL
        np.testing.assert_array_equal(got[~inside], sentinel[~inside])    # rows outside the support are not touched
    else:
        assert rel_err(got, exp) < RTOL


@pytest.mark.parametrize("N,residue", [(8, 0.0), (8, 3e-17), (4, 3e-17), (2, 0.0), (1, 3e-17)])
def test_real_weight_formats_of_a_gridding_matrix(hip, monkeypatch, N, residue):
    """A gridding matrix times the +-1 modulation of a centred transform on an even grid is REAL up to the 1e-16 rounding residue of
    exp(i pi k) (weights_are_real).  Its formats then store 4-byte weights -- 8-byte brick entries (ig_ccsrmm_t_bricks, entry_words 2),
    12-byte slot entries (ig_ccsrmm_t_slots, entry_words 3), float values for the forward gather over an interleaved panel
    (ig_ccsrmm_il_rw): against scipy in complex128 and against the complex formats of the same matrix; a matrix with genuinely
    complex weights keeps the complex formats."""
    from indigo_amd.backends.hip import weights_are_real
    rng = np.random.default_rng(7 + N)
    n0, nm, ns = 64, 32, 32
    P, T = n0 * nm * ns, 2500
    centre = rng.integers(0, P, size=30)
    rows, cols = [], []
    for t in range(T):
        base = centre[t % 30] if t % 3 else rng.integers(0, P)
        off = rng.integers(-2, 3, size=(40 if t % 89 == 0 else 27, 3))
        kx, km, ks = base % n0, (base // n0) % nm, base // (n0 * nm)
        c = np.unique(((kx + off[:, 0]) % n0) + n0 * (((km + off[:, 1]) % nm) + nm * ((ks + off[:, 2]) % ns)))
        rows.append(np.full(c.size, t)); cols.append(c)
    rows, cols = np.concatenate(rows), np.concatenate(cols)
    w = rng.standard_normal(rows.size).astype(np.float32) * np.where(rng.random(rows.size) < 0.5, -1, 1)
    vals = (w + 1j * (residue * rng.standard_normal(rows.size))).astype(C64)
    assert weights_are_real(vals) and not weights_are_real(rand64c(16, seed=1)) and not weights_are_real(np.zeros(0, C64))
    A = spp.csr_matrix((vals, (rows, cols)), shape=(T, P))
    A.sort_indices()
    X = rand64c(T, N, seed=5)
    G = rand64c(P, N, seed=6)

    def products(real_entries):
        monkeypatch.setitem(hip.tuning, "real_entries", real_entries)
        A_d = hip.csr_matrix(hip, A)
        if N > 1:
            A_d.set_grid_interleaved(True)
        if N in (4, 8):
            A_d.set_grid_bricks(n0, nm, ns, ncols=N, bm=2, bs=2, chunk=64, run=512)
            assert A_d._bricks['words'] == (2 if real_entries else 3) and A_d._bricks['nshared'] > 0
        else:
            A_d.set_grid_slots(n0, nm, ns, ncols=N, bm=4, bs=4, chunk=8, run=64)
            assert A_d._slots['words'] == (3 if real_entries else 4)
        y_d = hip.zero_array((P, N), C64)
        A_d.adjoint(y_d, hip.copy_array(X), alpha=0.5 - 0.25j)
        adj = y_d.to_host().reshape(-1, order='F').reshape(P, N)
        fwd = None
        if N > 1:       # forward over the interleaved grid panel (row-major memory)
            g_d = hip.copy_array(np.asfortranarray(G.reshape(-1).reshape(P, N, order='F')))
            k_d = hip.zero_array((T, N), C64)
            A_d.forward(k_d, g_d, alpha=1.5)
            assert (A_d._real_values() is not None) == real_entries
            fwd = k_d.to_host()
        return adj, fwd
    adj_r, fwd_r = products(True)
    adj_c, fwd_c = products(False)
    exp = (0.5 - 0.25j) * (A.conj().T.astype(np.complex128) @ X.astype(np.complex128))
    assert rel_err(adj_r, exp) < RTOL and rel_err(adj_c, exp) < RTOL and rel_err(adj_r, adj_c) < 2e-6
    if N > 1:
        expf = 1.5 * (A.astype(np.complex128) @ G.astype(np.complex128))
        assert rel_err(fwd_r, expf) < RTOL and rel_err(fwd_c, expf) < RTOL and rel_err(fwd_r, fwd_c) < 2e-6
    # genuinely complex weights: no real formats, whatever the tuning says
    monkeypatch.setitem(hip.tuning, "real_entries", True)
    B_d = hip.csr_matrix(hip, spp.csr_matrix((rand64c(rows.size, seed=3), (rows, cols)), shape=(T, P)))
    assert B_d._real_values() is None


def test_real_weight_wide_adjoint(hip, monkeypatch):
    """the 64-column adjoint by grid bricks with the image in registers (k_bricks_wide64r) on 8-byte entries {cell, re}: a plain
    gridding matrix has real weights (ig_ccsrmm_t_bricks_wide_grid, entry_words 2); against scipy and against the 12-byte form"""
    rng = np.random.default_rng(21)
    n0, nm, ns = 32, 32, 32
    P, T = n0 * nm * ns, 4000
    centre = rng.integers(0, P, size=25)
    rows, cols = [], []
    for t in range(T):
        base = centre[t % 25] if t % 4 else rng.integers(0, P)
        off = rng.integers(-1, 2, size=(27, 3))
        kx, km, ks = base % n0, (base // n0) % nm, base // (n0 * nm)
        c = np.unique(((kx + off[:, 0]) % n0) + n0 * (((km + off[:, 1]) % nm) + nm * ((ks + off[:, 2]) % ns)))
        rows.append(np.full(c.size, t)); cols.append(c)
    rows, cols = np.concatenate(rows), np.concatenate(cols)
    A = spp.csr_matrix((rng.standard_normal(rows.size).astype(np.float32).astype(C64), (rows, cols)), shape=(T, P))
    A.sort_indices()
    X = rand64c(T, 64, seed=8)
    exp = (0.5 + 2j) * (A.conj().T.astype(np.complex128) @ X.astype(np.complex128))
    outs = []
    for real_entries in (True, False):
        monkeypatch.setitem(hip.tuning, "real_entries", real_entries)
        monkeypatch.setitem(hip.tuning, "wide_task_shape", (512, 256))      # heavy bricks in shared pieces as well
        A_d = hip.csr_matrix(hip, A)
        A_d.set_grid_dims(n0, nm, ns)
        y_d = hip.copy_array(np.full((P, 64), 7 - 3j, dtype=C64, order='F'))
        A_d.adjoint(y_d, hip.copy_array(X), alpha=0.5 + 2j)
        assert A_d._wide is not None and A_d._wide['geom'][2:] == (2, 2) and A_d._wide['words'] == (2 if real_entries else 3)
        outs.append(y_d.to_host())
        assert rel_err(outs[-1], exp) < RTOL
    assert rel_err(outs[0], outs[1]) < 2e-6


@pytest.mark.parametrize("alpha,beta,ld_pad", [(1, 0, 0), (0.5 - 1j, 1.5, 5)])
def test_wide_panel_forward_ragged_rows(hip, alpha, beta, ld_pad):
    """the 16-row-tile gather (k_csrmm_gather_tile64: 64 columns, software-pipelined over the rows of a tile): rows of every
    length -- empty ones, more than one pass (> 32 nonzeros), a few beyond the long-row threshold (> 256: handed to the
    workgroup-per-row kernel, their tile slots skipped) -- and a row count that is not a multiple of 16"""
    M, K = 1003, 6000
    rng = np.random.default_rng(11)
    lens = rng.integers(0, 80, size=M)
    lens[[5, 300, 1002]] = [700, 300, 257]
    lens[[0, 17, 18, 19, 640]] = 0
    rows = np.repeat(np.arange(M), lens)
    cols = rng.integers(0, K, size=rows.size)
    A = spp.csr_matrix((rand64c(rows.size, seed=1), (rows, cols)), shape=(M, K))
    A.sum_duplicates(); A.sort_indices()
    A_d = hip.csr_matrix(hip, A)
    xfull = rand64c(K + ld_pad, 64, seed=2)
    yfull = rand64c(M + ld_pad, 64, seed=3)
    x_d = hip.copy_array(xfull)[0:K, :]
    y_d = hip.copy_array(yfull)[0:M, :]
    A_d.forward(y_d, x_d, alpha=alpha, beta=beta)
    exp = alpha * (A.astype(np.complex128) @ xfull[:K].astype(np.complex128)) + beta * yfull[:M]
    assert rel_err(y_d.to_host(), exp) < RTOL
    if ld_pad:
        np.testing.assert_array_equal(hip.copy_array(yfull)[M:, :].to_host(), yfull[M:])      # (rows past M of the parent are not ours)


@pytest.mark.parametrize("real_weights,alpha,beta,ld_pad", [(False, 1, 0, 0), (True, 1, 0, 0), (False, 0.5 - 1j, 1.5, 5), (True, 2, 1j, 3)])
def test_wide_panel_forward_by_runs(hip, monkeypatch, real_weights, alpha, beta, ld_pad):
    """the run format (ig_csr_runs_build + k_csrmm_runs64r: 64 columns, the panel rows of a run of 16 matrix rows loaded once, the
    run's results in registers through the VGPR index mode): rows of every length -- empty rows, whole empty runs, runs with
    more than 64 distinct columns and more than 64 entries (several windows), clustered columns (many entries per panel row: the
    quad and single paths) -- a row count that is not a multiple of 16, real and complex values, alpha / beta and padded
    leading dimensions; against scipy in complex128 and against the per-nonzero gather"""
    M, K = 1003, 12000
    rng = np.random.default_rng(12)
    lens = rng.integers(0, 40, size=M)
    lens[[5, 300, 1002]] = [700, 300, 257]
    lens[[0, 17, 18, 19, 640]] = 0
    lens[160:192] = 0                                           # two empty runs
    rows = np.repeat(np.arange(M), lens)
    # columns cluster around a centre per run of 16 rows, inside a pool of 30 % of the columns (col_frac <= 0.6: the xrows route)
    pool = np.sort(rng.choice(K, size=int(K * 0.3), replace=False))
    centre = rng.integers(0, pool.size, size=(M + 15) // 16)[rows // 16]
    spread = np.where(lens[rows] > 200, 400, 20)
    cols = pool[(centre + rng.integers(-spread, spread + 1)) % pool.size]
    vals = rand64c(rows.size, seed=1)
    if real_weights:
        vals = vals.real.astype(C64)
    A = spp.csr_matrix((vals, (rows, cols)), shape=(M, K))
    A.sum_duplicates(); A.sort_indices()
    A_d = hip.csr_matrix(hip, A)
    assert A_d._col_frac <= 0.6 and A.nnz >= K, (A_d._col_frac, A.nnz)
    xfull = rand64c(K + ld_pad, 64, seed=2)
    yfull = rand64c(M + ld_pad, 64, seed=3)
    x_d = hip.copy_array(xfull)[0:K, :]
    y_d = hip.copy_array(yfull)[0:M, :]
    A_d.forward(y_d, x_d, alpha=alpha, beta=beta)
    fmt = getattr(A_d, '_runs_fmt', None)
    assert fmt is not None and fmt['all_real'] == int(real_weights) and fmt['ndistinct'] < A.nnz
    exp = alpha * (A.astype(np.complex128) @ xfull[:K].astype(np.complex128)) + beta * yfull[:M]
    assert rel_err(y_d.to_host(), exp) < RTOL
    if ld_pad:
        np.testing.assert_array_equal(hip.copy_array(yfull)[M:, :].to_host(), yfull[M:])
    monkeypatch.setitem(hip.tuning, "runs", False)
    B_d = hip.csr_matrix(hip, A)
    y2 = hip.copy_array(yfull)[0:M, :]
    B_d.forward(y2, x_d, alpha=alpha, beta=beta)
    assert getattr(B_d, '_runs_fmt', None) is None
    assert rel_err(y2.to_host(), y_d.to_host()) < 2e-6

"""GPU parity of the gridding kernels that COMPUTE their taps from the separable form of the gridding matrix (round 6:
ig_interp3_sep, ig_grid_gather_sep, ig_grid_scatter_sep) -- against scipy products with the stored matrix the reference's own
construction gives (indigo/interp.py:18-60 times the -O3 recipe's modulation and scale, examples/pics.py:104-177; our builder of
that CSR is pinned to the reference's goldens in tests/test_sense_cpu.py) and against the stored-tap kernels of the same backend.
Everything goes through the C ABI (HipBackend.csr_matrix -> ctypes)."""
import numpy as np
import pytest
import scipy.sparse as spp

from conftest import rel_err
from indigo_amd.util import rand64c

pytestmark = pytest.mark.gpu
C64 = np.dtype('complex64')
RTOL = 1e-5          # complex64 parity bar of BASELINE.json's north_star


def _problem(N, osf, width, nspokes, seed=4, edge=False):
    from indigo_amd.sense import SenseProblem
    p = SenseProblem.synthetic(N, 2, nspokes=nspokes, nreadout=int(N[0] * osf), width=width, oversamp=osf, seed=seed)
    if edge:
        # push samples onto the grid's faces and corners (wrap-around taps on every axis) and exactly onto grid points (2 width taps)
        c = p.coord.reshape(3, -1, order='F').copy()
        rng = np.random.default_rng(seed)
        k = c.shape[1] // 4
        c[:, :k] = rng.choice([-0.5, -0.5 + 1.0 / p.oN[0], 0.5 - 1.0 / p.oN[0], 0.0, 0.25], size=(3, k))
        c[:, k:2 * k] = np.clip(c[:, k:2 * k] * 2.2, -0.5, 0.4999)
        p.coord = c.reshape(p.coord.shape, order='F')
        p.drop_cache()
    return p


@pytest.mark.parametrize("NC", [8, 4, 2])
@pytest.mark.parametrize("N,osf,width,edge", [((32, 32, 32), 2.0, 2, False), ((32, 32, 32), 2.0, 2, True), ((24, 32, 40), 2.0, 3, True),
                                              ((32, 32, 32), 1.5, 2, True), ((32, 16, 24), 2.0, 2.5, True), ((16, 13, 16), 2.0, 2, True)])
def test_forward_gridding_from_separable_records(hip, monkeypatch, NC, N, osf, width, edge):
    """Y = alpha G' X + beta Y over a coil-interleaved grid panel: taps computed from the records (ig_grid_gather_sep) against scipy in
    complex128 on the stored matrix, and against the stored-tap gather of the same backend"""
    p = _problem(N, osf, width, nspokes=97, edge=edge)
    G = p.fused_interp(1)
    sep = p.fused_interp_sep(1)
    assert sep is not None and sep['records'].shape[0] == p.T
    P = G.shape[1]
    X = rand64c(P, NC, seed=6)
    Y0 = rand64c(p.T, NC, seed=7)
    x_d = hip.copy_array(np.asfortranarray(X.reshape(-1).reshape(P, NC, order='F')))          # row-major memory: a grid point's coils side by side
    for alpha, beta in ((1.0, 0.0), (0.5 - 0.25j, 0.0), (1.5, -0.5 + 2j)):
        exp = alpha * (G.astype(np.complex128) @ X.astype(np.complex128)) + beta * Y0
        got = {}
        for use in (True, False):
            monkeypatch.setitem(hip.tuning, "sep_gather", use)
            A_d = hip.csr_matrix(hip, G)
            A_d.set_grid_interleaved(True)
            A_d.set_grid_separable(sep)
            y_d = hip.copy_array(Y0)
            A_d.forward(y_d, x_d, alpha=alpha, beta=beta)
            got[use] = y_d.to_host()
        assert rel_err(got[True], exp) < RTOL, (alpha, beta)
        assert rel_err(got[True], got[False]) < 2e-6
        # the order in which the workgroups take their groups of samples (sorted by grid block by default) changes nothing, bit for bit
        monkeypatch.setitem(hip.tuning, "sep_gather", True)
        monkeypatch.setitem(hip.tuning, "gather_order", False)
        y_d = hip.copy_array(Y0)
        A_d = hip.csr_matrix(hip, G)
        A_d.set_grid_interleaved(True)
        A_d.set_grid_separable(sep)
        A_d.forward(y_d, x_d, alpha=alpha, beta=beta)
        monkeypatch.setitem(hip.tuning, "gather_order", True)
        assert np.array_equal(y_d.to_host(), got[True])


def _expected_adjoint(G, X, alpha):
    return alpha * (G.conj().T.astype(np.complex128) @ X.astype(np.complex128))


@pytest.mark.parametrize("NC", [8, 4])
@pytest.mark.parametrize("N,osf,width,edge,shape", [((32, 32, 32), 2.0, 2, False, (4, 4)), ((32, 32, 32), 2.0, 2, True, (4, 4)),
                                                    ((24, 32, 40), 2.0, 3, True, (4, 4)), ((32, 16, 24), 2.0, 2.5, True, (2, 4)),
                                                    ((32, 32, 32), 2.0, 2, True, (4, 2)), ((16, 16, 16), 2.0, 4, True, (4, 4)),
                                                    ((32, 32, 32), 2.0, 2, True, (1, 1)), ((24, 32, 40), 2.0, 3, False, (2, 2)),
                                                    ((16, 13, 16), 2.0, 2, True, (4, 2)),           # (a 26-point axis: the modulation's constant is -+i)
                                                    ((16, 13, 17), 2.0, 3, True, (4, 4)),           # bricks that do not divide the grid: 26 = 6 * 4 + 2 slabs,
                                                    ((16, 15, 13), 2.0, 2, True, (4, 4))])          # 34 = 8 * 4 + 2 lines; 26 lines x 30 slabs
def test_adjoint_gridding_from_shares(hip, monkeypatch, NC, N, osf, width, edge, shape):
    """Y_il = alpha G'^H X as the scatter of (sample, brick) shares with computed taps (ig_grid_scatter_sep: brick image in registers,
    outer products on the matrix cores), no support table: every grid row is defined.  Heavy bricks cut into shared pieces (atomics)
    and runs of light bricks both occur (small chunk / run).  Against scipy in complex128 on the stored matrix and against the
    stored-tap adjoint of the same backend."""
    p = _problem(N, osf, width, nspokes=97, edge=edge)
    G = p.fused_interp(1)
    sep = p.fused_interp_sep(1)
    P = G.shape[1]
    X = rand64c(p.T, NC, seed=5)
    x_d = hip.copy_array(X)
    for alpha, (chunk, run) in ((1.0, (64, 128)), (0.5 - 0.25j, (1024, 1024)), (2.0, (16, 16))):
        exp = _expected_adjoint(G, X, alpha)
        A_d = hip.csr_matrix(hip, G)
        A_d.set_grid_interleaved(True)
        A_d.set_grid_separable(sep)
        A_d.set_grid_shares(NC, shape[0], shape[1], chunk, run)
        sh = A_d._shares_by[NC]
        assert sh is not None and sh['ntasks'] > 0 and (chunk > 64 or sh['nshared'] > 0)
        y_d = hip.copy_array(np.full((P, NC), np.nan + 0j, dtype=C64))
        A_d.adjoint(y_d, x_d, alpha=alpha)
        got = y_d.to_host().reshape(-1, order='F').reshape(P, NC)          # row-major memory -> (grid point, coil)
        assert rel_err(got, exp) < RTOL, (alpha, chunk, run)
        monkeypatch.setitem(hip.tuning, "sep_scatter", False)
        y2 = hip.zero_array((P, NC), C64)
        try:
            A_d.adjoint(y2, x_d, alpha=alpha)
        except RuntimeError:          # (the stored-tap gather over G'^T declines small dense grids: nothing to compare with)
            y2 = None
        monkeypatch.setitem(hip.tuning, "sep_scatter", True)
        assert y2 is None or rel_err(got, y2.to_host().reshape(-1, order='F').reshape(P, NC)) < 2e-6


@pytest.mark.parametrize("NC,tile", [(8, 4), (8, 8), (4, 8), (8, 16), (4, 16)])
def test_adjoint_shares_write_only_flagged_segments(hip, NC, tile):
    """with a k-space support table the scatter stores exactly the flagged segments of the bricks that hold a share -- into a grid
    poisoned with NaN: flagged segments equal scipy's G'^H X, everything else is still NaN; a second evaluation is bit-identical on
    every brick no shared task touches"""
    from indigo_amd import fused
    p = _problem((64, 64, 64), 2.0, 2, nspokes=211, edge=False)
    oN = p.oN
    zw = fused.support_words(hip, oN)
    if zw is None:
        pytest.skip("no support table for this grid on this backend")
    G = p.fused_interp(1)
    sep = p.fused_interp_sep(1)
    P = G.shape[1]
    n0, nm, ns = sep['dims']
    table16 = fused.grid_support(G, oN, 16, zw)
    X = rand64c(p.T, NC, seed=5)
    A_d = hip.csr_matrix(hip, G)
    A_d.set_grid_interleaved(True)
    A_d.set_grid_support(table16, n0, nm, zw[0])
    table = table16
    if tile != 16:
        table = fused.grid_support(G, oN, tile, zw)
        A_d.set_grid_support_fine(table, tile, ncols=NC)
    A_d.set_grid_separable(sep)
    A_d.set_grid_shares(NC, 4, 4, 256, 512)
    sh = A_d._shares_by[NC]
    assert sh['nshared'] > 0 and sh['tile'] == tile
    outs = []
    for _ in range(2):
        y_d = hip.copy_array(np.full((P, NC), np.nan + 0j, dtype=C64))
        A_d.adjoint(y_d, hip.copy_array(X), alpha=1.0)
        outs.append(y_d.to_host().reshape(-1, order='F').reshape(P, NC))
    got = outs[0]
    exp = _expected_adjoint(G, X, 1.0)
    # flagged cells by the table (input-side bitmaps): segment (kx tile, km, ks)
    _, _, bits = fused.split_support(table, oN, tile, zw[0])
    nt = n0 // tile
    ks, km, kxt = np.meshgrid(np.arange(ns), np.arange(nm), np.arange(nt), indexing='ij')
    flag = ((bits[ks * nt + kxt, km % zw[0]] >> (km // zw[0]).astype(np.uint32)) & 1).astype(bool)          # (ns, nm, nt)
    cell_flag = np.repeat(flag, tile, axis=2).reshape(-1)                                                    # kx fastest, then km, then ks
    touched = np.zeros(P, dtype=bool)
    touched[G.indices] = True
    assert not (touched & ~cell_flag).any()
    assert np.isnan(got[~cell_flag]).all()
    assert not np.isnan(got[cell_flag]).any()
    assert rel_err(got[cell_flag], exp[cell_flag]) < RTOL
    # repeatability: plain stores everywhere except the shared bricks
    tab = sh['table'].to_host().reshape(-1, 4)
    shb = set(sh['shared'].to_host().reshape(-1, 4)[:sh['nshared'], 0].tolist())
    nbx, nbm = n0 // 16, nm // sh['bm']
    kx, kmm, kss = np.arange(P) % n0, (np.arange(P) // n0) % nm, np.arange(P) // (n0 * nm)
    brick_of = kx // 16 + nbx * (kmm // sh['bm'] + nbm * (kss // sh['bs']))
    own = cell_flag & ~np.isin(brick_of, np.fromiter(shb, dtype=np.int64, count=len(shb)))
    assert np.array_equal(outs[0][own], outs[1][own])
    assert rel_err(outs[1][cell_flag], outs[0][cell_flag]) < 2e-6


def test_default_share_pieces_are_short(hip):
    """a heavy brick of the k-space centre is cut into pieces of at most 128 shares by default: a wave sums a piece into ONE float32 brick
    image before it adds it to the grid, and pieces of 1024 shares put an ill-conditioned evaluation (oversampling 1.25, half-width 3) 2.1e-5
    from the float64 one where 128 give 5e-6 (profiles/r06_share_pieces.txt); the task list of a problem with a hot centre obeys it"""
    assert all(v[2] <= 128 for v in hip.tuning['share_shape'].values())
    p = _problem((32, 32, 32), 2.0, 3, nspokes=601, edge=False)
    sep = p.fused_interp_sep(1)
    A_d = hip.csr_matrix(hip, p.fused_interp(1))
    A_d.set_grid_interleaved(True)
    A_d.set_grid_separable(sep)
    bm, bs, chunk, run = hip.tuning['share_shape'][8]
    A_d.set_grid_shares(8, bm, bs, chunk, run)
    sh = A_d._shares_by[8]
    tasks = sh['tasks'].to_host().reshape(-1, 4)[:sh['ntasks']]
    shared = (tasks[:, 3] >> 16) & 1
    assert sh['nshared'] > 0 and shared.any()
    assert (tasks[shared == 1, 1] - tasks[shared == 1, 0]).max() <= chunk

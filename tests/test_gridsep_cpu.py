"""Host side of the separable gridding format (round 6), no GPU: the records of ig_interp3_sep against the stored matrix our builder
makes of the reference's construction (indigo/interp.py:18-60 times the -O3 recipe's modulation and scale, examples/pics.py:104-177;
that CSR is pinned to the reference's goldens in test_sense_cpu.py), and the (sample, brick) shares of ig_grid_shares_count / _fill:
expanded, they must reproduce every tap of every sample exactly once, inside the brick they are filed under."""
import numpy as np
import pytest
import scipy.sparse as spp

from indigo_amd import _lib
from indigo_amd.interp import interp_sep_records, sep_expand
from indigo_amd.sense import SenseProblem, _mod_axis_phases


def _problem(N, osf, width, nspokes=53, seed=4):
    p = SenseProblem.synthetic(N, 2, nspokes=nspokes, nreadout=int(N[0] * osf), width=width, oversamp=osf, seed=seed)
    c = p.coord.reshape(3, -1, order='F').copy()
    rng = np.random.default_rng(seed)
    k = c.shape[1] // 3          # a third of the samples onto faces, corners and exact grid points: wrap-around, 2 width taps per axis
    c[:, :k] = rng.choice([-0.5, -0.5 + 1.0 / p.oN[0], 0.5 - 1.0 / p.oN[0], 0.0, 0.25], size=(3, k))
    p.coord = c.reshape(p.coord.shape, order='F')
    p.drop_cache()
    return p


@pytest.mark.parametrize("N,osf,width", [((16, 16, 16), 2.0, 2), ((12, 16, 20), 2.0, 3), ((13, 16, 16), 2.0, 2), ((16, 16, 16), 1.5, 2.5), ((8, 8, 8), 2.0, 4)])
@pytest.mark.parametrize("layout", [0, 1])
def test_separable_records_describe_the_stored_matrix(N, osf, width, layout):
    p = _problem(N, osf, width)
    G = p.fused_interp(layout)
    G.sort_indices()
    sep = p.fused_interp_sep(layout)
    assert sep is not None and sep['records'].shape[0] == p.T and sep['tw'] == (4 if 2 * width <= 4 else 6 if 2 * width <= 6 else 8)
    r, c, v = sep_expand(sep)
    S = spp.csr_matrix((v, (r, c)), shape=G.shape)
    S.sort_indices()
    assert S.nnz == G.nnz and np.array_equal(S.indices, G.indices) and np.array_equal(S.indptr, G.indptr)
    assert np.abs(S.data - G.data).max() <= 4e-7 * np.abs(G.data).max()          # five float32 roundings against two
    if any(n % 4 == 2 for n in p.oN):
        assert abs(sep['gconst'].imag) > 0.5 or sep['gconst'].real < 0             # exp(-i pi n / 4) per such axis: no +1


def test_odd_axes_and_wide_kernels_decline():
    p = _problem((16, 16, 16), 2.0, 2)
    coord = p.coord.reshape(3, -1, order='F')
    assert interp_sep_records(p.T, (32, 31, 32), 2, p.table, coord, _mod_axis_phases((32, 31, 32)), 1.0, 1) is None          # a complex modulation
    assert interp_sep_records(p.T, (32, 31, 32), 2, p.table, coord, None, 1.0, 1) is not None                              # the plain matrix is fine
    assert interp_sep_records(p.T, (32, 32, 32), 4.5, p.table, coord, None, 1.0, 1) is None                                # 9 taps per axis


def _shares(sep, bm, bs):
    L = _lib.lib()
    rec, tw = sep['records'], sep['tw']
    n0, nm, ns = sep['dims']
    nb = (n0 // 16) * (-(-nm // bm)) * (-(-ns // bs))
    counts = np.zeros(nb, np.int32)
    assert L.ig_grid_shares_count(rec.shape[0], rec.ctypes.data, tw, n0, nm, ns, bm, bs, counts.ctypes.data) == 0
    ptr = np.zeros(nb + 1, np.int64)
    np.cumsum(counts, out=ptr[1:])
    sh = np.empty((int(ptr[-1]), 2), np.uint32)
    assert L.ig_grid_shares_fill(rec.shape[0], rec.ctypes.data, tw, n0, nm, ns, bm, bs, ptr.ctypes.data, sh.ctypes.data) == 0
    return counts, sh


@pytest.mark.parametrize("N,osf,width,bm,bs", [((16, 16, 16), 2.0, 2, 8, 2), ((16, 16, 16), 2.0, 2, 4, 4), ((8, 16, 20), 2.0, 3, 4, 4),
                                               ((8, 8, 8), 2.0, 2, 16, 1), ((8, 8, 8), 2.0, 2.5, 2, 8), ((8, 8, 8), 2.0, 4, 4, 4), ((8, 8, 8), 2.0, 2, 1, 1),
                                               ((8, 13, 9), 2.0, 2, 4, 4), ((8, 9, 15), 2.0, 3, 4, 8)])          # bricks that do not divide the middle / slow axis
def test_shares_hold_every_tap_exactly_once(N, osf, width, bm, bs):
    p = _problem(N, osf, width)
    sep = p.fused_interp_sep(1)
    rec, tw = sep['records'], sep['tw']
    n0, nm, ns = sep['dims']
    counts, sh = _shares(sep, bm, bs)
    brick = np.repeat(np.arange(counts.size), counts)
    nbx, nbm = n0 // 16, -(-nm // bm)
    bx, bmi, bsi = brick % nbx, (brick // nbx) % nbm, brick // (nbx * nbm)
    t, gmask, geo = (sh[:, 0] & 0x0fffffff).astype(np.int64), sh[:, 0] >> 28, sh[:, 1]
    assert (np.diff(t)[np.diff(brick) == 0] >= 0).all()                       # sample order inside a brick
    ox, om, os_ = (geo & 31).astype(np.int64) - 8, ((geo >> 5) & 31).astype(np.int64) - 8, ((geo >> 10) & 31).astype(np.int64) - 8
    blo, bhi, clo, chi = (geo >> 15) & 7, (geo >> 18) & 15, (geo >> 22) & 7, (geo >> 25) & 15
    if bs <= 4:          # the slow-axis cells of the brick that hold a tap: what the MFMA form skips by
        assert np.array_equal(gmask, ((((1 << (chi - clo).astype(np.int64)) - 1) << (os_ + clo)) & 15).astype(np.uint32))
    w = rec[:, :3 * tw].view(np.float32).reshape(-1, 3, tw)
    cnt0 = (rec[:, 3 * tw + 1] >> 16) & 15
    rows, cols, vals = [], [], []
    for a in range(tw):
        for b in range(tw):
            for c in range(tw):
                cx = ox + a
                ok = (cx >= 0) & (cx < 16) & (a < cnt0[t]) & (b >= blo) & (b < bhi) & (c >= clo) & (c < chi)
                km, ks = bmi * bm + om + b, bsi * bs + os_ + c
                assert ((om + b)[ok] >= 0).all() and ((om + b)[ok] < bm).all() and ((os_ + c)[ok] >= 0).all() and ((os_ + c)[ok] < bs).all()
                rows.append(t[ok]); cols.append((bx * 16 + cx + n0 * (km + nm * ks))[ok])
                vals.append(((w[t, 1, b] * w[t, 2, c]).astype(np.float32) * w[t, 0, a]).astype(np.float32)[ok])
    rows, cols, vals = np.concatenate(rows), np.concatenate(cols), np.concatenate(vals)
    r, c, v = sep_expand(sep)
    A = spp.csr_matrix((v.real / np.real(sep['gconst']) if sep['gconst'].imag == 0 else np.abs(v), (r, c)), shape=(p.T, n0 * nm * ns))
    assert rows.size == A.nnz and np.unique(rows * A.shape[1] + cols).size == rows.size          # every tap once
    B = spp.csr_matrix((vals if sep['gconst'].imag == 0 else np.abs(vals), (rows, cols)), shape=A.shape)
    assert abs(A - B).max() == 0


def test_odd_axis_ramp_and_constant_leave_a_real_matrix():
    """round 6: what the fused leaf does on a grid with an odd chirp-z axis (HipBackend.fold_axis_shifts / split_gridding_constant, called here
    without a device: both are host logic over the library's host entry points).  G' of the reference's construction (interp * centred-transform
    modulation * scale: pinned to the goldens in test_sense_cpu.py) must equal  g * (real matrix) * diag(ramp on the folded axis)  with the ramp
    exp(2 pi i k c / n), c = n // 2 -- the circular shift the transform pass takes over (ig_fft_set_axis_shift)"""
    import types
    from indigo_amd.backends.hip import HipBackend
    from indigo_amd.interp import interp_csr_modulated
    fake = types.SimpleNamespace(_L=_lib.lib(), tuning={})
    N = (64, 35, 51)            # image; grid 128 x 69 (odd, 3 * 23: chirp-z) x 102 (= 2 mod 4: constant -+i)
    p = SenseProblem.synthetic(N, 2, nspokes=41, nreadout=128, width=2, oversamp=2.0, seed=3)
    p.oN = (128, 69, 102)
    p.drop_cache()
    ph = _mod_axis_phases(p.oN)
    ks, folded = HipBackend.fold_axis_shifts(fake, p.oN, ph)
    assert ks == (0, 34, 0) and np.ptp(folded[1]) == 0 and folded[1][0] == ph[1][0] and folded[0] is not None
    assert HipBackend.fold_axis_shifts(fake, (128, 64, 102), _mod_axis_phases((128, 64, 102))) == (None, None)           # nothing odd
    assert HipBackend.fold_axis_shifts(fake, (128, 75, 102), _mod_axis_phases((128, 75, 102))) == (None, None)           # 75 = 3 * 5 * 5: an A x B axis
    g, split = HipBackend.split_gridding_constant(fake, folded)
    assert abs(abs(g) - 1.0) < 1e-12 and split is not None
    coord = p.coord.reshape(3, -1, order='F')
    scale = np.float32(1.0) / np.sqrt(np.float32(np.prod(p.oN)))
    P = int(np.prod(p.oN))
    full = spp.csr_matrix(interp_csr_modulated(p.T, p.oN, p.width, p.table, coord, ph, scale, grid_order=1)[::-1], shape=(p.T, P))
    real = spp.csr_matrix(interp_csr_modulated(p.T, p.oN, p.width, p.table, coord, split, scale, grid_order=1)[::-1], shape=(p.T, P))
    assert np.abs(real.data.imag).max() <= 2.0 ** -34 * np.abs(real.data.real).max()          # real up to the residue weights_are_real allows
    # columns in (x, z, y) order: column = kx + n0 * (kz + n2 * ky)
    ky = np.arange(P) // (p.oN[0] * p.oN[2])
    ramp = np.exp(2j * np.pi * ky * 34 / 69)
    rebuilt = (real.astype(np.complex128) @ spp.diags(ramp)) * g
    assert abs(rebuilt - full.astype(np.complex128)).max() <= 3e-7 * np.abs(full.data).max()
    sep = interp_sep_records(p.T, p.oN, p.width, p.table, coord, split, scale, 1)
    assert sep is not None and sep['gconst'] == 1

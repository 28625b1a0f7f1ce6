"""BASELINE.json configurations 2-5 at FULL size on the GPU, each against an oracle that shares no code with the
library: numpy's pocketfft, scipy's csr product and the numpy restatement of the reference backend
(oracle/np_backend.py).  These are the kernel instantiations bench.py times -- the two-stage FFT kernel on a STRIDED
512-point axis (R1 = 32, 16- and 32-column tiles, half-box variants), the coil-interleaved gridding kernels, the
64-column SpMM -- which the small golden-vector cases never reach (reference pattern: test_backends.py:153-210).

Where the full result is too large to compare on the host, linearity restricts it: a SENSE operator whose other
coils are zero, a k-space panel whose other columns are zero, or a subset of the panel columns.
Tolerance: 1e-5 relative (L2) in complex64, the north star's bar.
"""
import numpy as np
import pytest
import scipy.sparse as spp

from conftest import rel_err
from indigo_amd.sense import SenseProblem, normal_operator
from indigo_amd.util import rand64c

pytestmark = pytest.mark.gpu
C64 = np.dtype('complex64')
RTOL = 1e-5


# ---------------------------------------------------------------------------------------
# plain fftn/ifftn on 512-point axes at a stride (k_fft_2stage<32,16,16,16,false,...>) and config 2
# ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("shape", [(4, 512), (48, 512), (16, 512, 2), (16, 512, 8), (2, 3, 512), (32, 16, 512),
                                   (32, 512, 512), (512, 512, 4), (8, 256), (16, 256, 6), (4, 5, 256)])
def test_fft_strided_two_stage_axes_vs_numpy(hip, shape):
    batch = 3
    x = rand64c(*(shape + (batch,)), seed=sum(shape))
    x_d = hip.copy_array(x)
    y_d = hip.zero_array(x.shape, C64)
    hip.fftn(y_d, x_d)
    assert "2stage" in hip.fft_describe(x.shape)
    axes = tuple(range(len(shape)))
    assert rel_err(y_d.to_host(), np.fft.fftn(x.astype(np.complex128), axes=axes)) < RTOL, hip.fft_describe(x.shape)
    hip.ifftn(y_d, x_d)
    assert rel_err(y_d.to_host(), np.fft.ifftn(x.astype(np.complex128), axes=axes) * np.prod(shape)) < RTOL


def test_fft_plain_512_axis_at_a_huge_stride_vs_numpy(hip):
    """a plain (unboxed) 512-point axis 1 MB per element apart takes the 32-column run-time-box variant: (256, 512, 512) against
    numpy, forward and inverse, one volume"""
    shape = (256, 512, 512)
    x = rand64c(*(shape + (1,)), seed=9)
    x_d = hip.copy_array(x)
    y_d = hip.zero_array(x.shape, C64)
    hip.fftn(y_d, x_d)
    ref = np.fft.fftn(x[..., 0].astype(np.complex128))
    assert rel_err(y_d.to_host()[..., 0], ref) < RTOL
    hip.ifftn(y_d, y_d)                                         # in place, inverse: back to n * x
    assert rel_err(y_d.to_host() / np.prod(shape), x) < RTOL
    del x_d, y_d


def test_config2_fft_256_cubed_batch16(hip):
    """BASELINE config 2 as quoted: 256^3 complex64, 16 batches.  Two of the volumes against numpy, all of them by
    Parseval and the unnormalised round trip; in place and out of place."""
    n, batch = 256, 16
    x = rand64c(n, n, n, batch, seed=2)
    x_d = hip.copy_array(x)
    y_d = hip.zero_array(x.shape, C64)
    hip.fftn(y_d, x_d)
    for j in (0, 11):
        got = y_d[:, :, :, j:j + 1].to_host()[..., 0]
        assert rel_err(got, np.fft.fftn(x[..., j])) < RTOL, j
    e_in, e_out = hip.norm2(x_d), hip.norm2(y_d)
    np.testing.assert_allclose(e_out, e_in * n ** 3, rtol=1e-5)
    hip.ifftn(y_d, y_d)                                         # in place, inverse
    got = y_d[:, :, :, 5:6].to_host()[..., 0]
    assert rel_err(got / n ** 3, x[..., 5]) < RTOL
    hip.axpby(1.0 / n ** 3, y_d, -1.0, x_d)
    assert np.sqrt(hip.norm2(y_d) / e_in) < RTOL
    del x_d, y_d


@pytest.mark.parametrize("batch", [1, 3])
def test_fft_256_cubed_two_launch_transform(hip, batch):
    """256^3 volumes take the two-launch transform (x + y/64 | y/4 + z, indigo_amd/csrc/ig_fft.hip k_fft3d_a/_b): forward and
    inverse, out of place and in place (launch A cannot run in place: the plan stages through its own buffer), vs numpy"""
    n = 256
    x = (rand64c(n, n, n, batch, seed=7) - np.complex64(0.5 + 0.5j)).astype(C64)
    assert "two launches" in hip.fft_describe(x.shape)
    x_d = hip.copy_array(x)
    y_d = hip.zero_array(x.shape, C64)
    hip.fftn(y_d, x_d)
    np.testing.assert_array_equal(x_d[:, :, :, 0:1].to_host(), x[..., 0:1])        # the input is left alone
    for j in range(batch):
        assert rel_err(y_d[:, :, :, j:j + 1].to_host()[..., 0], np.fft.fftn(x[..., j].astype(np.complex128))) < RTOL
    hip.ifftn(y_d, y_d)                                                            # in place, inverse: n^3 * x
    assert rel_err(y_d.to_host() / n ** 3, x) < RTOL
    hip.ifftn(y_d, x_d)                                                            # out of place, inverse
    assert rel_err(y_d[:, :, :, 0:1].to_host()[..., 0], np.fft.ifftn(x[..., 0].astype(np.complex128)) * n ** 3) < RTOL
    hip.fftn(x_d, x_d)                                                             # in place, forward
    assert rel_err(x_d[:, :, :, batch - 1:batch].to_host()[..., 0], np.fft.fftn(x[..., batch - 1].astype(np.complex128))) < RTOL


# ---------------------------------------------------------------------------------------
# config 3: 3-D radial gridding CSR (T x 256^3, 27 taps per row, 5e7 nonzeros) x 64-column panel
# ---------------------------------------------------------------------------------------
def gridding_problem(grid, nspokes, nreadout, seed=3):
    """SenseProblem used only for its trajectory and gridding matrix (oversamp 1: the grid IS the image)"""
    from indigo_amd.sense import radial_trajectory
    coord = radial_trajectory(nspokes, nreadout, seed=seed)
    return SenseProblem(grid, coord, lambda c: None, width=2, ntable=128, oversamp=1.0, ncoils=1)


def test_config3_gridding_csr_times_64_columns(hip):
    """the workload of BASELINE config 3 (3617 spokes x 512 samples on 256^3, width 2 -> 27 taps), forward and
    adjoint through the reference's SpMatrix surface, eight of the 64 columns compared with scipy on the host"""
    from indigo_amd.interp import interp_csr_arrays
    from scipy.signal.windows import kaiser
    N, ncol = (256, 256, 256), 64
    gp = gridding_problem(N, 3617, 512)
    beta = np.pi * np.sqrt(((2 * 2.0 / 2.0) * (2.0 - 0.5)) ** 2 - 0.8)
    table = kaiser(2 * 128 + 1, beta)[128:]
    indptr, indices, w = interp_csr_arrays(gp.T, N, 2, table, gp.coord.reshape(3, -1, order='F'), dtype=np.float32)
    G = spp.csr_matrix((w.astype(C64), indices, indptr), shape=(gp.T, int(np.prod(N))))
    assert 4.9e7 < G.nnz < 5.1e7
    S = hip.SpMatrix(G, name='gridding')
    P, T = G.shape[1], G.shape[0]
    cols = [0, 9, 18, 27, 36, 45, 54, 63]
    # the 8.6 GB panel is generated and uploaded eight columns at a time; the compared columns stay on the host
    X_d = hip.empty_array((P, ncol), C64)
    keep = {}
    for j0 in range(0, ncol, 8):
        blk = rand64c(P, 8, seed=100 + j0)
        X_d[:, j0:j0 + 8].copy_from(blk)
        for j in cols:
            if j0 <= j < j0 + 8:
                keep[j] = blk[:, j - j0].copy()
        del blk
    Xs = np.stack([keep[j] for j in cols], axis=1)
    Y_d = hip.zero_array((T, ncol), C64)
    S.eval(Y_d, X_d)
    Y = Y_d.to_host()
    assert rel_err(Y[:, cols], G @ Xs) < RTOL
    # a second, independent check of ALL columns: column sums commute with the product
    np.testing.assert_allclose(Y.sum(axis=0)[cols], (G @ Xs).sum(axis=0), rtol=1e-4)
    del X_d
    # adjoint: Z = G^H Y (256^3 x 64); compare the same eight columns
    Z_d = hip.empty_array((P, ncol), C64)
    S.eval(Z_d, Y_d, forward=False)
    # (double-precision oracle: the rows at the k-space centre sum ~1e5 terms each and carry most of the norm; scipy's own
    # complex64 running sum is itself off by more than 1e-5 there)
    GH = G.conj().T.tocsr().astype(np.complex128)
    exp = GH @ Y[:, cols].astype(np.complex128)
    for i, j in enumerate(cols):
        got = Z_d[:, j:j + 1].to_host()[:, 0]
        assert rel_err(got, exp[:, i]) < RTOL, j
    # beta / alpha on the wide adjoint (a second evaluation accumulates)
    S.eval(Z_d, Y_d, alpha=0.5, beta=-1.0, forward=False)
    got = Z_d[:, 27:28].to_host()[:, 0]
    assert rel_err(got, -0.5 * exp[:, 3]) < RTOL
    del Z_d, Y_d


# ---------------------------------------------------------------------------------------
# config 4: the headline SENSE problem at full size against the numpy oracle, coil by coil
# ---------------------------------------------------------------------------------------
def masked_coils(p, keep):
    """the same problem with every coil outside `keep` switched off (zero map): the interleaved 8-coil kernels then
    compute sum_{c in keep} A_c^H A_c x, which the one-coil oracle can check"""
    zero = np.zeros(p.N, dtype=C64, order='F')
    q = SenseProblem(p.N, p.coord, lambda c: p.coil_map(c) if c in keep else zero, width=p.width, ntable=p.ntable,
                     oversamp=p.oversamp, ncoils=p.C)
    q._interp_cache = p._interp_cache
    return q


def oracle_coil_ops(p, oracle_backend, coils):
    return {c: p.build_zpadfft(oracle_backend, coils=[c], layout=0, support=False) for c in coils}


def test_config4_full_size_vs_numpy_oracle(hip, oracle_backend):
    """image 256^3, 8 coils, grid 512^3, T = 1,851,904 (what bench.py times, grid layout 2):
       forward  : k-space columns of coils 0 and 5 == oracle A_c x
       adjoint  : A^H of a panel that is non-zero in columns 2 and 7 == oracle A_2^H k_2 + A_7^H k_7
       normal   : A^H A with all coils but one switched off == oracle A_0^H A_0 x
       one coil : the per-coil kernels a rank of an 8-GPU run uses (layout 1) == the same oracle results"""
    p = SenseProblem.synthetic((256, 256, 256), 8, nspokes=3617, nreadout=512, width=2, ntable=128, oversamp=2.0, seed=4)
    hip._scratch = None
    oracle_backend._scratch = None
    T, C = p.T, p.C
    x = rand64c(int(np.prod(p.N)), 1, seed=1)
    k = np.zeros((T, C), dtype=C64, order='F')
    k[:, 2] = rand64c(T, seed=2)
    k[:, 7] = rand64c(T, seed=3)
    A = p.build_zpadfft(hip)                                   # layout 2, with the k-space support table
    Ax = (A * x).reshape(T, C, order='F')
    AHk = A.H * k.reshape(-1, 1, order='F')
    del A
    hip._scratch = None
    A1 = p.build_zpadfft(hip, coils=[0])                       # layout 1, single-column gridding kernels
    A1x = A1 * x
    A1Hk = A1.H * np.asfortranarray(k[:, 2:3])                 # (coil 0's weights applied to column 2's data)
    del A1
    hip._scratch = None
    A0 = masked_coils(p, {0}).build_zpadfft(hip)               # the benchmarked operator, coils 1..7 switched off
    y_d = hip.zero_array((x.shape[0], 1), C64)
    normal_operator(A0).eval(y_d, hip.copy_array(x))
    AHA0 = y_d.to_host()
    del A0, y_d
    hip._scratch = None
    # ---- oracle, one coil at a time (numpy pocketfft + scipy csr) ----
    O = oracle_coil_ops(p, oracle_backend, [0, 5, 2, 7])
    o0x = O[0] * x
    assert rel_err(Ax[:, 0:1], o0x) < RTOL
    assert rel_err(A1x, o0x) < RTOL
    # A^H A of this input (uniform[0,1): a large DC term, ~1e4 samples on the central grid point) is where the complex64
    # oracle is itself only good to ~2.6e-5 (its running complex64 sums, see oracle/precise.py): arbitrate in double
    from oracle.precise import CoilOperatorF64
    exact = CoilOperatorF64(p, 0).normal(x).reshape(-1, 1)
    o_aha = O[0].H * o0x
    oracle_own = rel_err(o_aha, exact)
    assert rel_err(AHA0, exact) < RTOL, "HIP A^H A vs the double-precision evaluation"
    assert rel_err(AHA0, o_aha) < oracle_own + RTOL, "HIP vs the complex64 oracle, beyond the oracle's own error (%.2e)" % oracle_own
    del exact, o_aha
    assert rel_err(A1Hk, O[0].H * np.asfortranarray(k[:, 2:3])) < RTOL
    del o0x
    assert rel_err(Ax[:, 5:6], O[5] * x) < RTOL
    exp = O[2].H * np.asfortranarray(k[:, 2:3]) + O[7].H * np.asfortranarray(k[:, 7:8])
    assert rel_err(AHk, exp) < RTOL
    oracle_backend._scratch = None
    p.drop_cache()


# ---------------------------------------------------------------------------------------
# config 5: image 320^3 in a 512^3 grid (oversampling 1.6), 32 coils in chunks / 4 coils per rank
# ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("C,layout", [(2, 2), (1, 1)])
def test_config5_box_320_in_512_leaves_vs_numpy(hip, C, layout):
    """the zero-pad-aware transforms on the box of config 5 (not the half box: run-time box predicates on a
    512-point axis), coil 0 against numpy"""
    grid, box = (512, 512, 512), (320, 320, 320)
    lo = tuple(m // 2 + int(np.ceil(-n / 2)) for m, n in zip(grid, box))
    assert lo == (96, 96, 96)
    P, N = int(np.prod(grid)), int(np.prod(box))
    hip._scratch = None
    x = rand64c(N, 1, seed=1)
    w = rand64c(N, C, seed=2)
    w_d = hip.copy_array(np.ascontiguousarray(w).reshape(-1)) if layout == 2 else hip.copy_array(w)
    sl = tuple(slice(l, l + b) for l, b in zip(lo, box))
    ws = hip.zero_array((hip._fft_padded_workspace(grid, lo, box, C, layout) // 8,), C64)
    y_d = hip.copy_array(np.full((P, C), np.nan, dtype=C64, order='F'))
    hip.fft_padded(y_d, hip.copy_array(x), w_d, grid, lo, box, ws, layout)
    y = y_d.to_host()
    full = np.zeros(grid, dtype=C64, order='F')
    full[sl] = (w[:, 0] * x[:, 0]).reshape(box, order='F')
    ref = np.fft.fftn(full)                                    # (x, y, z)
    if layout == 2:
        got = y.reshape(-1, order='F').reshape((C, grid[0], grid[2], grid[1]), order='F')[0].transpose(0, 2, 1)
    else:
        got = y[:, 0].reshape((grid[0], grid[2], grid[1]), order='F').transpose(0, 2, 1)
    assert rel_err(got, ref) < RTOL
    del full, ref, got
    # cropped inverse of y itself: crop(IFFT(FFT(pad(w x)))) * conj(w) = P * |w|^2 x  -- and against numpy for coil 0
    xc_d = hip.copy_array(np.full((N, C), np.nan, dtype=C64, order='F'))
    hip.ifft_cropped(xc_d, y_d, w_d, grid, lo, box, ws, layout)
    xc = xc_d.to_host()
    xc0 = xc.reshape(-1, order='F').reshape(N, C)[:, 0] if layout == 2 else xc[:, 0]
    assert rel_err(xc0, P * np.abs(w[:, 0]) ** 2 * x[:, 0]) < RTOL
    del y_d, xc_d, ws
    hip._scratch = None


def test_config5_shard_of_four_coils_vs_oracle(hip, oracle_backend):
    """what one rank of the 8-GPU run of config 5 evaluates: 4 of the 32 coils of a 320^3 image on the 512^3 grid
    (fewer spokes than the benchmark, the same grid, box and kernels), against the one-coil numpy oracle"""
    p = SenseProblem.synthetic((320, 320, 320), 32, nspokes=600, nreadout=512, width=2, ntable=128, oversamp=1.6,
                               seed=5, lazy_maps=True)
    assert p.oN == (512, 512, 512)
    hip._scratch = None
    oracle_backend._scratch = None
    coils = [12, 13, 14, 15]                                   # rank 3 of 8
    A = p.build_zpadfft(hip, coils=coils)
    T = p.T
    x = rand64c(int(np.prod(p.N)), 1, seed=1)
    k = np.zeros((T, 4), dtype=C64, order='F')
    k[:, 0] = rand64c(T, seed=2)
    k[:, 3] = rand64c(T, seed=3)
    Ax = (A * x).reshape(T, 4, order='F')
    AHk = A.H * k.reshape(-1, 1, order='F')
    del A
    hip._scratch = None
    O = oracle_coil_ops(p, oracle_backend, [12, 13, 15])
    assert rel_err(Ax[:, 1:2], O[13] * x) < RTOL
    exp = O[12].H * np.asfortranarray(k[:, 0:1]) + O[15].H * np.asfortranarray(k[:, 3:4])
    assert rel_err(AHk, exp) < RTOL
    oracle_backend._scratch = None
    p.drop_cache()


def test_config5_coil_chunks_vs_oracle(hip, oracle_backend):
    """more coils than one interleaved grid holds: a VStack of chunks sharing one device gridding matrix.
    Reduced size (image 160^3 in a 256^3 grid, oversampling 1.6, 6 coils in chunks of 2 + an odd 7th alone)."""
    p = SenseProblem.synthetic((160, 160, 160), 7, nspokes=300, nreadout=256, width=2, ntable=128, oversamp=1.6, seed=5,
                               lazy_maps=True)
    assert p.oN == (256, 256, 256)
    hip._scratch = None
    oracle_backend._scratch = None
    A = p.build_zpadfft(hip, chunk=2)
    from indigo_amd import operators as op
    assert isinstance(A, op.VStack) and len(A.children) == 4
    A_o = p.build_zpadfft(oracle_backend, layout=0, support=False)
    x = rand64c(A.shape[1], 1, seed=1)
    k = rand64c(A.shape[0], 1, seed=2)
    assert rel_err(A * x, A_o * x) < RTOL
    ref_h = A_o.H * k
    got_h = A.H * k
    if not rel_err(got_h, ref_h) < RTOL:                 # which chunk, and is a second evaluation the same?
        T = p.T
        again = A.H * k
        msg = ["adjoint err %.3e, second evaluation err %.3e, first vs second %.3e" % (rel_err(got_h, ref_h), rel_err(again, ref_h), rel_err(again, got_h))]
        lo = 0
        for ci, ch in enumerate(A.children):
            nc = ch.shape[0] // T
            kk = k[lo:lo + ch.shape[0]]
            Ao_c = p.build_zpadfft(oracle_backend, coils=list(range(lo // T, lo // T + nc)), layout=0, support=False)
            msg.append("chunk %d (%d coils): err %.3e" % (ci, nc, rel_err(ch.H * kk, Ao_c.H * kk)))
            lo += ch.shape[0]
        raise AssertionError("; ".join(msg))
    AHA = normal_operator(A, lamda=0.3)
    y_d = hip.zero_array((A.shape[1], 1), C64)
    AHA.eval(y_d, hip.copy_array(x))
    exp = A_o.H * (A_o * x) + np.float32(0.3) * x
    assert rel_err(y_d.to_host(), exp) < RTOL
    hip._scratch = None
    oracle_backend._scratch = None


def test_config5_as_benchmarked_on_one_gpu(hip, oracle_backend):
    """BASELINE config 5 the way bench.py runs it at N = 1: image 320^3 on the 512^3 grid at full T = 2,893,824, 32 coils as a
    VStack of four 8-coil chunks sharing one gridding matrix.  Reaches the 8-coil kernels on the run-time 320-of-512 box
    (the coil-summing x pass, the 8-coil brick scatter) that the per-rank shapes above do not.
       one 8-coil chunk : forward column of coil 3, adjoint of a panel non-zero in columns 1 and 6, A^H A with seven coils
                          switched off -- against the one-coil numpy oracle / its double-precision evaluation
       four chunks      : A^H A of the 32-coil VStack with only coils 3 and 20 (chunks 0 and 2) switched on"""
    from oracle.precise import CoilOperatorF64
    p = SenseProblem.synthetic((320, 320, 320), 32, nspokes=5652, nreadout=512, width=2, ntable=128, oversamp=1.6,
                               seed=5, lazy_maps=True)
    assert p.oN == (512, 512, 512) and p.T == 2893824
    hip._scratch = None
    oracle_backend._scratch = None
    T = p.T
    x = rand64c(int(np.prod(p.N)), 1, seed=1)
    k = np.zeros((T, 8), dtype=C64, order='F')
    k[:, 1] = rand64c(T, seed=2)
    k[:, 6] = rand64c(T, seed=3)
    A = p.build_zpadfft(hip, coils=range(8))
    from indigo_amd import operators as op
    assert not isinstance(A, op.VStack)
    Ax3 = (A * x).reshape(T, 8, order='F')[:, 3:4].copy()
    AHk = A.H * k.reshape(-1, 1, order='F')
    del A
    hip._scratch = None
    y_d = hip.zero_array((x.shape[0], 1), C64)
    A0 = masked_coils(p, {3}).build_zpadfft(hip, coils=range(8))
    normal_operator(A0).eval(y_d, hip.copy_array(x))
    AHA3 = y_d.to_host()
    del A0
    hip._scratch = None
    q = masked_coils(p, {3, 20})
    A32 = q.build_zpadfft(hip)
    assert isinstance(A32, op.VStack) and len(A32.children) == 4
    normal_operator(A32).eval(y_d, hip.copy_array(x))
    AHA_2of32 = y_d.to_host()
    del A32, y_d
    hip._scratch = None
    # ---- oracle ----
    O = oracle_coil_ops(p, oracle_backend, [3, 1, 6])
    o3x = O[3] * x
    assert rel_err(Ax3, o3x) < RTOL
    exp = O[1].H * np.asfortranarray(k[:, 1:2]) + O[6].H * np.asfortranarray(k[:, 6:7])
    assert rel_err(AHk, exp) < RTOL
    del exp
    e3 = CoilOperatorF64(p, 3).normal(x).reshape(-1, 1)
    o_aha = O[3].H * o3x
    oracle_own = rel_err(o_aha, e3)
    assert rel_err(AHA3, e3) < RTOL, "8-coil chunk, A^H A vs the double-precision evaluation"
    assert rel_err(AHA3, o_aha) < oracle_own + RTOL
    del O, o_aha, o3x
    oracle_backend._scratch = None
    e20 = CoilOperatorF64(p, 20).normal(x).reshape(-1, 1)
    assert rel_err(AHA_2of32, e3 + e20) < RTOL, "four-chunk VStack: the chunks' images accumulate"
    p.drop_cache()


# ---------------------------------------------------------------------------------------
# the reference driver's own oversampled grids (examples/pics.py:87-90: 320 ... 640, not powers of two): zero-pad-aware
# passes on the A x B two-stage kernel (k_fft_ab_desc), coil-interleaved layout
# ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("grid,box,C", [((320, 320, 320), (256, 256, 256), 8), ((480, 384, 640), (360, 300, 500), 2),
                                        ((400, 432, 512), (300, 310, 256), 4),
                                        # lengths of the generated list (tools/gen_ab_list.py): odd ones, splits with A != B, 3 / 5 / 7 in both factors
                                        ((288, 270, 392), (208, 208, 308), 4), ((600, 135, 175), (480, 100, 140), 2),
                                        ((144, 625, 128), (100, 500, 96), 16), ((360, 250, 567), (256, 200, 400), 2)])
def test_padded_transforms_on_non_power_of_two_grids_vs_numpy(hip, grid, box, C):
    """fft_padded / ifft_cropped / ifft_cropped_sum (layout 2) on grids whose axes are 320 ... 640 points long (and mixed
    with 512): one coil of the forward grid against numpy, the cropped inverse and the coil combination against their
    closed forms and numpy"""
    lo = tuple(m // 2 + int(np.ceil(-n / 2)) for m, n in zip(grid, box))
    P, N = int(np.prod(grid)), int(np.prod(box))
    assert hip.supports_padded_fft(grid, C)
    hip._scratch = None
    x = rand64c(N, 1, seed=1)
    w = rand64c(N, C, seed=2)
    w_d = hip.copy_array(np.ascontiguousarray(w).reshape(-1))
    sl = tuple(slice(l, l + b) for l, b in zip(lo, box))
    ws = hip.zero_array((hip._fft_padded_workspace(grid, lo, box, C, 2) // 8,), C64)
    y_d = hip.copy_array(np.full((P, C), np.nan, dtype=C64, order='F'))
    hip.fft_padded(y_d, hip.copy_array(x), w_d, grid, lo, box, ws, 2)
    y = y_d.to_host().reshape(-1, order='F').reshape((C, grid[0], grid[2], grid[1]), order='F')        # (c, x, z, y)
    for c in (0, C - 1):
        full = np.zeros(grid, dtype=C64, order='F')
        full[sl] = (w[:, c] * x[:, 0]).reshape(box, order='F')
        ref = np.fft.fftn(full)
        assert rel_err(y[c].transpose(0, 2, 1), ref) < RTOL, c
        del full, ref
    del y
    # cropped inverse of the grid just made: conj(w_c) * crop(IFFT(FFT(pad(w_c x)))) = P |w_c|^2 x
    xc_d = hip.copy_array(np.full((N, C), np.nan, dtype=C64, order='F'))
    hip.ifft_cropped(xc_d, y_d, w_d, grid, lo, box, ws, 2)
    xc = xc_d.to_host().reshape(-1, order='F').reshape(N, C)
    assert rel_err(xc, P * np.abs(w) ** 2 * x) < RTOL
    # ... and with the coil combination inside the last pass
    xs_d = hip.copy_array(np.full((N, 1), np.nan, dtype=C64, order='F'))
    hip.ifft_cropped_sum(xs_d, y_d, w_d, grid, lo, box, ws)
    assert rel_err(xs_d.to_host(), P * (np.abs(w) ** 2).sum(axis=1, keepdims=True) * x) < RTOL
    # an arbitrary grid (not a transform of anything zero-padded): coil 0 against numpy's inverse
    g = rand64c(P, C, seed=3)
    g_d = hip.copy_array(np.ascontiguousarray(g).reshape(-1)).reshape((P, C))
    hip.ifft_cropped(xc_d, g_d, w_d, grid, lo, box, ws, 2)
    vol = g[:, 0].reshape((grid[0], grid[2], grid[1]), order='F').transpose(0, 2, 1)                       # memory (x, z, y) -> (x, y, z)
    ref = (np.fft.ifftn(vol) * P)[sl].reshape(-1, order='F') * np.conj(w[:, 0])
    got = xc_d.to_host().reshape(-1, order='F').reshape(N, C)[:, 0]
    assert rel_err(got, ref) < RTOL
    del y_d, xc_d, xs_d, g_d, ws
    hip._scratch = None


def test_padded_transforms_with_chirp_z_axes_vs_numpy(hip):
    """the reference driver's default oversampling (640/480, examples/pics.py:86) on a 240 x 104 x 154 image: grid 320 x 138 x 205
    -- 138 = 2 * 3 * 23 and 205 = 5 * 41 run as chirp-z passes inside the fused leaf (two launches of the A x B kernel each, no
    support table).  fft_padded against numpy, the cropped transforms against their closed forms and numpy"""
    grid, box, C = (320, 138, 205), (240, 104, 154), 4
    assert hip.supports_padded_fft(grid, C) and hip.padded_axis_kind(138) == 5 and hip.padded_axis_kind(205) == 5
    assert not hip.supports_padded_fft((138, 320, 205), C)          # the x axis carries the weights: no chirp-z there
    lo = tuple(m // 2 + int(np.ceil(-n / 2)) for m, n in zip(grid, box))
    P, N = int(np.prod(grid)), int(np.prod(box))
    hip._scratch = None
    x = rand64c(N, 1, seed=1)
    w = rand64c(N, C, seed=2)
    w_d = hip.copy_array(np.ascontiguousarray(w).reshape(-1))
    sl = tuple(slice(l, l + b) for l, b in zip(lo, box))
    ws = hip.zero_array((hip._fft_padded_workspace(grid, lo, box, C, 2) // 8,), C64)
    y_d = hip.copy_array(np.full((P, C), np.nan, dtype=C64, order='F'))
    hip.fft_padded(y_d, hip.copy_array(x), w_d, grid, lo, box, ws, 2)
    y = y_d.to_host().reshape(-1, order='F').reshape((C, grid[0], grid[2], grid[1]), order='F')        # (c, x, z, y)
    for c in (0, C - 1):
        full = np.zeros(grid, dtype=C64, order='F')
        full[sl] = (w[:, c] * x[:, 0]).reshape(box, order='F')
        assert rel_err(y[c].transpose(0, 2, 1), np.fft.fftn(full)) < RTOL, c
    xs_d = hip.copy_array(np.full((N, 1), np.nan, dtype=C64, order='F'))
    hip.ifft_cropped_sum(xs_d, y_d, w_d, grid, lo, box, ws)
    assert rel_err(xs_d.to_host(), P * (np.abs(w) ** 2).sum(axis=1, keepdims=True) * x) < RTOL
    g = rand64c(P, C, seed=3)
    g_d = hip.copy_array(np.ascontiguousarray(g).reshape(-1)).reshape((P, C))
    xc_d = hip.copy_array(np.full((N, C), np.nan, dtype=C64, order='F'))
    hip.ifft_cropped(xc_d, g_d, w_d, grid, lo, box, ws, 2)
    vol = g[:, 1].reshape((grid[0], grid[2], grid[1]), order='F').transpose(0, 2, 1)
    ref = (np.fft.ifftn(vol) * P)[sl].reshape(-1, order='F') * np.conj(w[:, 1])
    assert rel_err(xc_d.to_host().reshape(-1, order='F').reshape(N, C)[:, 1], ref) < RTOL
    hip._scratch = None


def test_padded_transforms_carry_the_modulation_of_odd_axes(hip):
    """round 6 (ig_fft_set_axis_shift): a circular shift by c on the image side of a chirp-z axis = the modulation exp(2 pi i k c / n) on its
    k-space side -- what the reference's centred transform (Backend.fftc_mod, indigo/backends/backend.py:352-366) puts on an ODD axis -- folded
    into the pass's tables.  Grid 160 x 69 x 205 (both odd axes chirp-z), shifts n // 2 and an arbitrary one: fft_padded against numpy times the
    modulation, the cropped transforms against numpy of the conjugate-modulated grid; an axis that is no chirp-z axis refuses"""
    grid, box, C = (160, 69, 205), (120, 52, 154), 4
    assert hip.supports_padded_fft(grid, C) and hip.padded_axis_kind(69) == 5 and hip.padded_axis_kind(205) == 5
    lo = tuple(m // 2 + int(np.ceil(-n / 2)) for m, n in zip(grid, box))
    P, N = int(np.prod(grid)), int(np.prod(box))
    hip._scratch = None
    x = rand64c(N, 1, seed=1)
    w = rand64c(N, C, seed=2)
    w_d = hip.copy_array(np.ascontiguousarray(w).reshape(-1))
    sl = tuple(slice(l, l + b) for l, b in zip(lo, box))
    ws = hip.zero_array((hip._fft_padded_workspace(grid, lo, box, C, 2) // 8,), C64)
    g = rand64c(P, C, seed=3)
    g_d = hip.copy_array(np.ascontiguousarray(g).reshape(-1)).reshape((P, C))
    for ks in ((0, 34, 102), (0, 0, 7), (0, 68, 0)):
        mod = np.ones(grid, dtype=np.complex128)
        for a in (1, 2):
            ph = np.exp(2j * np.pi * np.arange(grid[a]) * ks[a] / grid[a])
            mod = mod * ph.reshape([-1 if d == a else 1 for d in range(3)])
        y_d = hip.copy_array(np.full((P, C), np.nan, dtype=C64, order='F'))
        hip.fft_padded(y_d, hip.copy_array(x), w_d, grid, lo, box, ws, 2, kshift=ks)
        y = y_d.to_host().reshape(-1, order='F').reshape((C, grid[0], grid[2], grid[1]), order='F')        # (c, x, z, y)
        for c in (0, C - 1):
            full = np.zeros(grid, dtype=np.complex128, order='F')
            full[sl] = (w[:, c] * x[:, 0]).reshape(box, order='F')
            assert rel_err(y[c].transpose(0, 2, 1), mod * np.fft.fftn(full)) < RTOL, (ks, c)
        xc_d = hip.copy_array(np.full((N, C), np.nan, dtype=C64, order='F'))
        hip.ifft_cropped(xc_d, g_d, w_d, grid, lo, box, ws, 2, kshift=ks)
        vol = g[:, 1].reshape((grid[0], grid[2], grid[1]), order='F').transpose(0, 2, 1)
        ref = (np.fft.ifftn(np.conj(mod) * vol) * P)[sl].reshape(-1, order='F') * np.conj(w[:, 1])
        assert rel_err(xc_d.to_host().reshape(-1, order='F').reshape(N, C)[:, 1], ref) < RTOL, ks
        xs_d = hip.copy_array(np.full((N, 1), np.nan, dtype=C64, order='F'))
        hip.ifft_cropped_sum(xs_d, y_d, w_d, grid, lo, box, ws, kshift=ks)          # the adjoint undoes the forward's modulation
        assert rel_err(xs_d.to_host(), P * (np.abs(w) ** 2).sum(axis=1, keepdims=True) * x) < RTOL, ks
    with pytest.raises(RuntimeError):
        hip._padded_plan((160, 160, 205), (20, 20, 25), (120, 120, 154), C, 2, 16, (0, 80, 0))          # 160 points: an A x B axis
    hip._scratch = None


def test_sense_on_a_grid_with_odd_axes_keeps_a_real_gridding_matrix(hip, oracle_backend):
    """round 6: image 120 x 52 x 77 on the grid the reference driver's default oversampling gives, 160 x 69 x 102 -- 69 is odd (the centred
    transform's modulation there is a genuine phase ramp), 102 = 2 mod 4 (its constant is -+i).  The fused leaf moves the ramp into the chirp-z
    pass of that axis and the constant into the transform's weights: the gridding matrix keeps REAL weights -- 8-byte brick entries, separable
    records with gconst = 1 -- and the operator still equals the oracle's, which knows none of this"""
    from indigo_amd import operators as op
    p = SenseProblem.synthetic((120, 52, 77), 8, nspokes=300, nreadout=160, width=2, ntable=128, oversamp=640 / 480, seed=6)
    assert p.oN == (160, 69, 102)
    hip._scratch = None
    oracle_backend._scratch = None
    ks, folded = hip.fold_axis_shifts(p.oN, __import__('indigo_amd.sense', fromlist=['x'])._mod_axis_phases(p.oN))
    assert ks == (0, 34, 0) and np.ptp(folded[1]) == 0
    g, split = hip.split_gridding_constant(folded)
    assert abs(abs(g) - 1) < 1e-12 and abs(g - 1) > 0.5 and all(np.allclose(np.exp(2j * np.pi * ph).imag, 0, atol=1e-12) for ph in split)
    A = p.build_zpadfft(hip)
    Z = A.right
    assert isinstance(Z, op.ZpadFFT) and Z._tile_kw.get('kshift') == (0, 34, 0)
    x = rand64c(A.shape[1], 1, seed=1)
    k = rand64c(A.shape[0], 1, seed=2)
    A_o = p.build_zpadfft(oracle_backend, layout=0, support=False)
    assert rel_err(A * x, A_o * x) < RTOL
    assert rel_err(A.H * k, A_o.H * k) < RTOL
    M = A.left.right._matrix_d
    assert M._sep is not None and M._sep['gconst'] == 1 and M._bricks_by[8]['words'] == 2
    y_d = hip.zero_array((A.shape[1], 1), C64)
    normal_operator(A, lamda=0.2).eval(y_d, hip.copy_array(x))
    exp = A_o.H * (A_o * x) + np.float32(0.2) * x
    assert rel_err(y_d.to_host(), exp) < RTOL
    # the same at the reference's default half-width 3: record gather + share scatter on bricks that do not divide 69
    p3 = SenseProblem.synthetic((120, 52, 77), 8, nspokes=300, nreadout=160, width=3, ntable=128, oversamp=640 / 480, seed=6)
    A3 = p3.build_zpadfft(hip)
    A3_o = p3.build_zpadfft(oracle_backend, layout=0, support=False)
    assert rel_err(A3 * x, A3_o * x) < RTOL
    k3 = rand64c(A3.shape[0], 1, seed=2)
    assert rel_err(A3.H * k3, A3_o.H * k3) < RTOL
    M3 = A3.left.right._matrix_d
    assert M3._shares_by[8] is not None and (M3._shares_by[8]['bm'], M3._shares_by[8]['bs']) == (4, 4)
    hip._scratch = None
    oracle_backend._scratch = None


@pytest.mark.parametrize("C", [4, 8])
def test_sense_on_the_reference_drivers_grid_vs_oracle(hip, oracle_backend, C):
    """image 256^3 on a 320^3 grid (oversampling 1.25, the reference driver's choice of grid, examples/pics.py:87-90): the fused
    leaf on the A x B passes WITH the k-space support table (bitmaps of 20 and 16 words per entry for the 16 x 20 split of the z
    axis; the 4-point table with 8 coils, the 8-point one with 4; the 8- and the 4-coil brick scatter), forward / adjoint / normal
    operator against the numpy oracle, which knows no table"""
    p = SenseProblem.synthetic((256, 256, 256), C, nspokes=400, nreadout=320, width=2, ntable=128, oversamp=1.25, seed=6)
    assert p.oN == (320, 320, 320)
    hip._scratch = None
    oracle_backend._scratch = None
    A = p.build_zpadfft(hip)
    from indigo_amd import operators as op
    assert A.has(op.ZpadFFT) and A.right._layout == 2 and A.right._support_h is not None and p.last_support_zw == (20, 16)
    assert p.last_support_fine is not None and p.last_support_fine[1] == (4 if C == 8 else 8)       # (coils * tile >= 32)
    x = rand64c(A.shape[1], 1, seed=1)
    k = rand64c(A.shape[0], 1, seed=2)
    A_o = p.build_zpadfft(oracle_backend, layout=0, support=False)
    assert rel_err(A * x, A_o * x) < RTOL
    assert rel_err(A.H * k, A_o.H * k) < RTOL
    y_d = hip.zero_array((A.shape[1], 1), C64)
    normal_operator(A, lamda=0.2).eval(y_d, hip.copy_array(x))
    exp = A_o.H * (A_o * x) + np.float32(0.2) * x
    assert rel_err(y_d.to_host(), exp) < RTOL
    hip._scratch = None
    oracle_backend._scratch = None
    p.drop_cache()


def _chunk_trees(A):
    from indigo_amd import operators as op
    out = []
    for ch in (A.children if isinstance(A, op.VStack) else [A]):
        out.append(ch.child if isinstance(ch, op.HeadRows) else ch)
    return out


@pytest.mark.parametrize("C", [3, 6, 9, 12])
@pytest.mark.parametrize("image,osf", [((128, 128, 128), 2.0), ((256, 256, 256), 1.25)])
def test_every_coil_count_takes_the_fast_routes(hip, oracle_backend, C, image, osf):
    """A KronI takes any coil count (reference backend.py:311-314, operators.py:374-375; examples/pics.py:93).  The coil-interleaved
    kernels take 2, 4 or 8: other counts are cut into chunks of those widths that share ONE device gridding matrix carrying a binned
    adjoint format and a fine support table per width present, the last chunk padded with zero-weight coils where needed
    (indigo_amd.fused.plan_chunks: 3 -> 4 wide; 6 -> 4 + 2; 9 -> 8 + 1 on a power-of-two grid, 8 + 2 wide elsewhere; 12 -> 8 + 4).
    Forward / adjoint (/ normal operator on the small grid) against the per-coil numpy oracle on a 256^3 grid (power-of-two
    passes) and on a 320^3 grid (A x B passes), and: every chunk runs the fused leaf, a scatter format of its own width and --
    8 and 4 wide -- its fine table."""
    from indigo_amd import operators as op
    p = SenseProblem.synthetic(image, C, nspokes=300, nreadout=2 * image[0] if osf == 2.0 else 320, width=2, ntable=128, oversamp=osf, seed=7)
    pow2 = p.oN == (256, 256, 256)
    assert pow2 or p.oN == (320, 320, 320)
    hip._scratch = None
    oracle_backend._scratch = None
    A = p.build_zpadfft(hip)
    widths = [w for _, _, w in A._coil_chunks]
    assert widths == {3: [4], 6: [4, 2], 9: [8, 1] if pow2 else [8, 2], 12: [8, 4]}[C]
    assert A.shape == (C * p.T, int(np.prod(image)))
    G_il = None
    for tree, w in zip(_chunk_trees(A), widths):
        Z, G = tree.right, tree.left.right
        assert isinstance(Z, op.ZpadFFT) and Z._C == w and Z._layout == (2 if w > 1 else 1)
        M = G._get_or_create_device_matrix()
        if w > 1:
            assert G_il is None or G is G_il, "the interleaved chunks share one gridding matrix"
            G_il = G
            assert M._format('_bricks' if w in (4, 8) else '_slots', w, exact=True) is not None
            if w in (4, 8):
                assert M._format('_support_fine', w)[1] == (4 if w == 8 else 8) and Z._tile_kw == {'support_tile': 4 if w == 8 else 8}
        else:
            assert M._format('_slots', 1, exact=True) is not None
    x = rand64c(A.shape[1], 1, seed=1)
    k = rand64c(A.shape[0], 1, seed=2)
    A_o = p.build_zpadfft(oracle_backend, layout=0, support=False)
    assert rel_err(A * x, A_o * x) < RTOL
    assert rel_err(A.H * k, A_o.H * k) < RTOL
    if pow2:
        y_d = hip.zero_array((A.shape[1], 1), C64)
        normal_operator(A, lamda=0.2).eval(y_d, hip.copy_array(x))
        exp = A_o.H * (A_o * x) + np.float32(0.2) * x
        assert rel_err(y_d.to_host(), exp) < RTOL
    hip._scratch = None
    oracle_backend._scratch = None
    p.drop_cache()


@pytest.mark.parametrize("C", [4, 3, 8])
def test_sense_on_a_chirp_z_grid_with_its_support_table(hip, oracle_backend, C):
    """The reference driver's default oversampling (640/480, examples/pics.py:86) with its sizing rule int(N * osf)
    (indigo/backends/backend.py:427-430) at a quarter of its scan: image 120 x 52 x 77 on 160 x 69 x 102 -- 69 = 3 * 23 and
    102 = 2 * 3 * 17 are chirp-z axes.  The fused leaf takes the k-space support table there too: a chirp-z thread holds the rows
    b + B a of its length-m transform on both sides, so the bitmaps of such a z axis are B words per entry
    (ig_fft_support_words); the brick scatter writes by them (no zero-fill of the grid), the chirp-z passes skip the tiles outside
    the hulls and mask their loads and stores with the bitmaps, 8- and 4-wide chunks take the fine table.
    Forward / adjoint / normal operator against the per-coil numpy oracle, which knows no table."""
    from indigo_amd import operators as op
    p = SenseProblem.synthetic((120, 52, 77), C, nspokes=200, nreadout=160, width=3, ntable=128, oversamp=640 / 480, seed=8)
    assert p.oN == (160, 69, 102)
    hip._scratch = None
    oracle_backend._scratch = None
    A = p.build_zpadfft(hip)
    assert p.last_support_table is not None
    zw = p.last_support_zw
    assert zw[0] == zw[1] and zw[0] * 32 >= 102 and zw == hip.support_words(102)
    for tree in _chunk_trees(A):
        Z, G = tree.right, tree.left.right
        assert isinstance(Z, op.ZpadFFT) and Z._layout == 2 and Z._support_h is not None and getattr(G, '_grid_support', None) is not None
    if C in (4, 8):
        assert p.last_support_fine is not None and p.last_support_fine[1] == (8 if C == 4 else 4)
    # the hulls really cut something: a radial trajectory leaves the corners of the (ky, kx tile) plane empty
    zr, yr, _ = p.split_support(p.last_support_table, 16)
    assert np.count_nonzero(zr[:, 1] > zr[:, 0]) < zr.shape[0]
    x = rand64c(A.shape[1], 1, seed=1)
    k = rand64c(A.shape[0], 1, seed=2)
    A_o = p.build_zpadfft(oracle_backend, layout=0, support=False)
    assert rel_err(A * x, A_o * x) < RTOL
    assert rel_err(A.H * k, A_o.H * k) < RTOL
    y_d = hip.zero_array((A.shape[1], 1), C64)
    normal_operator(A, lamda=0.2).eval(y_d, hip.copy_array(x))
    exp = A_o.H * (A_o * x) + np.float32(0.2) * x
    assert rel_err(y_d.to_host(), exp) < RTOL
    hip._scratch = None
    oracle_backend._scratch = None
    p.drop_cache()

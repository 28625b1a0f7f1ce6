"""The reconstruction driver (indigo_amd/pics.py; reference examples/pics.py) end to end on the MI355X backend: a synthetic
radial scan, the reference's recipe at -O3 plus FuseZpadFFT (the benchmarked leaf: grid 256^3, two interleaved coils), CG --
against the same driver on the numpy oracle backend, against the unfused -O3 leaves, and against the phantom.  A second scan
at osf 1.25 (grid 160 = 10 x 16: the fused leaf on the zero-pad-aware A x B passes)."""
import os

import numpy as np
import pytest

from indigo_amd import pics
from indigo_amd.sense import radial_trajectory

pytestmark = pytest.mark.gpu


def _scan(tmpdir, B, N, C, nro, nsp, osf, width=2):
    g = np.mgrid[tuple(slice(-1, 1, n * 1j) for n in N)]
    img = (np.exp(-4 * (g[0] ** 2 + 1.5 * g[1] ** 2 + 0.7 * g[2] ** 2)) * (1 + 0.3j)).astype(np.complex64)
    centres = [(-1, 0, 0.3), (1, 0.5, -0.4), (0, -1, 1.0), (0.3, 0.9, 2.0)][:C]
    mps = np.stack([np.exp(-((g[0] - cx) ** 2 + (g[1] - cy) ** 2)) * np.exp(1j * ph) for cx, cy, ph in centres],
                   axis=3).astype(np.complex64)
    coord = radial_trajectory(nsp, nro, seed=2)
    traj = coord * np.array(N, dtype=np.float64)[:, None, None]
    F1 = B.NUFFT((1, nro, nsp), N, coord, width=width, oversamp=(osf, osf, osf), dtype=np.dtype('complex64'))
    A = B.KronI(C, F1) * B.VStack([B.Diag(mps[:, :, :, c:c + 1]) for c in range(C)])
    ksp = (A * np.asfortranarray(img.reshape(-1, 1, order='F'))).reshape((1, nro, nsp, C), order='F')
    path = os.path.join(str(tmpdir), "scan.npz")
    np.savez(path, data=ksp.reshape(ksp.shape + (1,)).T, maps=mps.reshape(mps.shape + (1,)).T, traj=traj.T)
    return path, img


def _rel(a, b):
    return np.linalg.norm((a - b).ravel()) / np.linalg.norm(b.ravel())


def test_pics_on_the_gpu_matches_the_oracle_backend(tmp_path, hip, oracle_backend):
    N, C = (128, 128, 128), 2
    path, img = _scan(tmp_path, hip, N, C, nro=256, nsp=600, osf=2.0)
    args = ["-i", "4", "--osf", "2.0", "--width", "2", "--lamda", "1e-3", "--debug", "40", path]
    fused = pics.main(["-O", "3"] + args, backend=hip)
    assert fused.shape == N + (1, 1)
    plain = pics.main(["-O", "3", "--no-fuse"] + args, backend=hip)
    assert _rel(fused, plain) < 2e-4                       # same operator, different leaves: CG amplifies complex64 rounding
    oracle_backend._scratch = None
    ref = pics.main(["-O", "3", "--no-fuse"] + args, backend=oracle_backend)
    oracle_backend._scratch = None
    assert _rel(fused, ref) < 1e-3
    assert _rel(plain, ref) < 1e-3
    # parity proper: ONE iteration (x = alpha r: a forward, two adjoints, two reductions; nothing amplified yet) is held to north_star's
    # 1e-5 against the oracle backend running the same driver
    one = ["-i", "1"] + args[2:]
    fused1 = pics.main(["-O", "3"] + one, backend=hip)
    ref1 = pics.main(["-O", "3", "--no-fuse"] + one, backend=oracle_backend)
    oracle_backend._scratch = None
    assert _rel(fused1, ref1) < 1e-5, _rel(fused1, ref1)
    # more iterations on the GPU alone: CG recovers the phantom up to the driver's normalisation of the right-hand side
    full = pics.main(["-O", "3", "-i", "30", "--osf", "2.0", "--width", "2", "--lamda", "1e-4", "--debug", "40", path], backend=hip)
    x, t = full.reshape(-1, order='F'), img.reshape(-1, order='F')
    scale = np.vdot(x, t) / np.vdot(x, x)
    assert np.linalg.norm(scale * x - t) < 0.1 * np.linalg.norm(t)


def test_pics_on_a_non_power_of_two_grid(tmp_path, hip, oracle_backend, caplog):
    """the reference driver's own kind of grid (oversampling 1.25: 160 = 10 x 16 points): FuseZpadFFT takes the fused leaf on
    the A x B zero-pad-aware passes (4 coils interleaved); same image as the unfused -O3 leaves, the unoptimised tree and the
    oracle backend"""
    import logging
    N, C = (128, 128, 128), 4
    path, img = _scan(tmp_path, hip, N, C, nro=160, nsp=300, osf=1.25, width=3)
    assert "AxB" in hip.fft_describe((160, 160, 160, C)) and hip.supports_padded_fft((160, 160, 160), C)
    args = ["-i", "4", "--osf", "1.25", "--width", "3", "--lamda", "1e-3", "--debug", "40", path]
    with caplog.at_level(logging.INFO, logger="pics"):
        out = pics.main(["-O", "3"] + args, backend=hip)
    tree = [r.getMessage() for r in caplog.records if r.getMessage().startswith("tree:")][-1]
    assert "ZpadFFT" in tree and "UnscaledFFT" not in tree, tree
    plain = pics.main(["-O", "3", "--no-fuse"] + args, backend=hip)
    assert _rel(out, plain) < 2e-4
    oracle_backend._scratch = None
    ref = pics.main(["-O", "3", "--no-fuse"] + args, backend=oracle_backend)
    oracle_backend._scratch = None
    assert _rel(out, ref) < 1e-3
    base = pics.main(["-O", "0"] + args, backend=hip)
    assert _rel(out, base) < 2e-4
    # three coils (the reference takes any count, examples/pics.py:93): ONE 4-wide interleaved chunk whose fourth coil has zero
    # weights -- the fused leaf on this grid too, the same image as the unfused leaves
    assert hip.supports_padded_fft((160, 160, 160), 3)
    (tmp_path / "c3").mkdir()
    path3, _ = _scan(tmp_path / "c3", hip, N, 3, nro=160, nsp=300, osf=1.25, width=3)
    args3 = ["-i", "3", "--osf", "1.25", "--width", "3", "--lamda", "1e-3", "--debug", "40", path3]
    caplog.clear()
    with caplog.at_level(logging.INFO, logger="pics"):
        out3 = pics.main(["-O", "3"] + args3, backend=hip)
    tree3 = [r.getMessage() for r in caplog.records if r.getMessage().startswith("tree:")][-1]
    assert "ZpadFFT" in tree3 and "HeadRows" in tree3 and "UnscaledFFT" not in tree3, tree3
    assert _rel(out3, pics.main(["-O", "3", "--no-fuse"] + args3, backend=hip)) < 2e-4


def test_pics_at_the_reference_drivers_default_oversampling(tmp_path, hip, oracle_backend, caplog):
    """examples/pics.py:86 oversamples by 640/480; indigo/backends/backend.py:427-430 sizes the grid as int(N * osf): the
    reference's own 480 x 208 x 308 scan lands on 640 x 277 x 410 -- 277 is prime, 410 = 2 * 5 * 41.  The same at a quarter of
    the size: image 120 x 52 x 77, grid 160 x 69 x 102 (69 = 3 * 23, 102 = 2 * 3 * 17).  FuseZpadFFT takes the fused leaf -- the x
    axis on the A x B kernel, the y and z axes as chirp-z passes -- and the image equals the unfused -O3 leaves' and the
    oracle backend's"""
    import logging
    N, C = (120, 52, 77), 4
    path, img = _scan(tmp_path, hip, N, C, nro=160, nsp=300, osf=640 / 480, width=3)
    args = ["-i", "4", "--width", "3", "--lamda", "1e-3", "--debug", "40", path]          # (--osf: the driver's default)
    assert hip.supports_padded_fft((160, 69, 102), C)
    with caplog.at_level(logging.INFO, logger="pics"):
        out = pics.main(["-O", "3"] + args, backend=hip)
    tree = [r.getMessage() for r in caplog.records if r.getMessage().startswith("tree:")][-1]
    assert "ZpadFFT" in tree and "UnscaledFFT" not in tree and "(160, 69, 102)" in hip_grid_of(tree), tree
    plain = pics.main(["-O", "3", "--no-fuse"] + args, backend=hip)
    assert _rel(out, plain) < 2e-4
    oracle_backend._scratch = None
    ref = pics.main(["-O", "3", "--no-fuse"] + args, backend=oracle_backend)
    oracle_backend._scratch = None
    assert _rel(out, ref) < 1e-3


def hip_grid_of(tree):
    """the ZpadFFT line of a dumped tree names its shape (C * P, N): recover P's factors from the test's known grid"""
    return "(160, 69, 102)" if str(4 * 160 * 69 * 102) in tree else ""

"""The reconstruction driver (indigo_amd/pics.py; reference examples/pics.py) on a synthetic scan, CPU only: the
numpy oracle backend is injected where the product would construct the MI355X backend.  Checks the container
handling (BART-style reversed dimension order, --crop), that every -O level and the FuseZpadFFT route give the
same image, and that CG actually reconstructs a phantom from its own simulated k-space."""
import os

import numpy as np
import pytest

from indigo_amd import pics
from indigo_amd.sense import radial_trajectory
from indigo_amd.util import rand64c


@pytest.fixture(scope="module")
def scan(tmp_path_factory, oracle_backend):
    N, C, nro, nsp = (12, 10, 8), 3, 24, 80
    rng = np.random.default_rng(0)
    # smooth positive phantom and smooth maps
    g = np.mgrid[tuple(slice(-1, 1, n * 1j) for n in N)]
    img = (np.exp(-3 * (g[0] ** 2 + g[1] ** 2 + g[2] ** 2)) * (1 + 0.3j)).astype(np.complex64)
    mps = np.stack([np.exp(-((g[0] - cx) ** 2 + (g[1] - cy) ** 2)) * np.exp(1j * ph)
                    for cx, cy, ph in [(-1, 0, 0.3), (1, 0.5, -0.4), (0, -1, 1.0)]], axis=3).astype(np.complex64)
    coord = radial_trajectory(nsp, nro, seed=2)                       # (3, nro, nsp) in units of the FOV
    traj = coord * np.array(N, dtype=np.float64)[:, None, None]       # pixels, as the reference's files store it
    B = oracle_backend
    B._scratch = None
    F1 = B.NUFFT((1, nro, nsp), N, coord, width=3, oversamp=(1.5, 1.5, 1.5), dtype=np.dtype('complex64'))
    A = B.KronI(C, F1) * B.VStack([B.Diag(mps[:, :, :, c:c + 1]) for c in range(C)])
    ksp = (A * np.asfortranarray(img.reshape(-1, 1, order='F'))).reshape((1, nro, nsp, C), order='F')
    B._scratch = None
    # the file stores the arrays with reversed dimension order (examples/pics.py:52-54 reads them back with .T)
    path = os.path.join(str(tmp_path_factory.mktemp("scan")), "scan.npz")
    np.savez(path, data=ksp.reshape(ksp.shape + (1,)).T, maps=mps.reshape(mps.shape + (1,)).T, traj=traj.T)
    return path, img, N, C


def test_pics_reconstructs_the_phantom_and_all_levels_agree(scan, oracle_backend):
    path, img, N, C = scan
    recs = {}
    for level, extra in ((0, []), (3, ["--no-fuse"]), (3, [])):
        oracle_backend._scratch = None
        out = pics.main(["-i", "6", "-O", str(level), "--osf", "1.5", "--lamda", "1e-4", "--debug", "40", path] + extra,
                        backend=oracle_backend)
        assert out.shape == N + (1, 1)
        recs[(level, tuple(extra))] = out
        rec = np.load(os.path.splitext(path)[0] + ".rec.npy")
        np.testing.assert_array_equal(rec, out.T)                    # written back in the file's orientation
    base = recs[(0, ())]
    for k, v in recs.items():
        assert np.linalg.norm(v - base) < 2e-3 * np.linalg.norm(base), k      # complex64 CG: rounding differs per tree
    # CG on the normalised right-hand side recovers the phantom up to that normalisation
    oracle_backend._scratch = None
    full = pics.main(["-i", "25", "--osf", "1.5", "--lamda", "1e-4", "--debug", "40", path], backend=oracle_backend)
    x = full.reshape(-1, order='F')
    t = img.reshape(-1, order='F')
    scale = np.vdot(x, t) / np.vdot(x, x)
    assert np.linalg.norm(scale * x - t) < 0.05 * np.linalg.norm(t)
    oracle_backend._scratch = None


def test_pics_crop_and_errors(scan, oracle_backend):
    path, img, N, C = scan
    oracle_backend._scratch = None
    out2 = pics.main(["-i", "2", "--osf", "1.5", "--crop", "COIL:2", "--debug", "40", path], backend=oracle_backend)
    assert out2.shape == N + (1, 1)
    crops = pics.crop_limits("COIL:2,TIME:1")
    assert crops[-(pics.dim.COIL + 1)] == 2 and crops[-(pics.dim.TIME + 1)] == 1
    with pytest.raises(SystemExit):
        pics.load(os.path.join(os.path.dirname(path), "scan.h5"))   # h5py is absent in this image: a clear message
    oracle_backend._scratch = None

"""`python -m indigo_amd.pics` on a synthetic scan of the reference's own size -- image 480 x 208 x 308, 8 coils -- at the driver's
default oversampling 640/480 (examples/pics.py:86): the grid is int(N * osf) = 640 x 277 x 410 (indigo/backends/backend.py:427-430),
277 prime, 410 = 2 * 5 * 41.  Prints the tree the driver evaluates (the fused ZpadFFT leaf with two chirp-z axes) and CG's residuals."""
import logging, os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from indigo_amd import pics
from indigo_amd.backends import get_backend
from indigo_amd.sense import SenseProblem, radial_trajectory
from indigo_amd.util import rand64c

B = get_backend("hip")
N, C, nro, nsp = (480, 208, 308), 8, 640, 2000
g = [np.linspace(-1, 1, n) for n in N]
img = (np.exp(-4 * (g[0][:, None, None] ** 2 + 1.5 * g[1][None, :, None] ** 2 + 0.7 * g[2][None, None, :] ** 2)) * (1 + 0.3j)).astype(np.complex64)
mps = np.stack([np.exp(-((g[0][:, None, None] - np.cos(c)) ** 2 + (g[1][None, :, None] - np.sin(c)) ** 2)) * np.exp(0.3j * c) * np.ones((1, 1, N[2]))
                for c in range(C)], axis=3).astype(np.complex64)
coord = radial_trajectory(nsp, nro, seed=2)
traj = coord * np.array(N, dtype=np.float64)[:, None, None]
# the scan: k-space of the phantom through the fused operator itself (the driver's own factories would build the same samples)
p = SenseProblem(N, coord, np.asfortranarray(mps), width=3, oversamp=640 / 480)
print("grid", p.oN, flush=True)
A = p.build_zpadfft(B)
ksp = (A * np.asfortranarray(img.reshape(-1, 1, order='F'))).reshape((1, nro, nsp, C), order='F')
del A
B._scratch = None
path = os.path.join(tempfile.mkdtemp(), "scan.npz")
np.savez(path, data=ksp.reshape(ksp.shape + (1,)).T, maps=mps.reshape(mps.shape + (1,)).T, traj=traj.T)
logging.basicConfig(level=logging.INFO)
t0 = time.time()
out = pics.main(["-O", "3", "-i", "5", "--width", "3", "--lamda", "1e-3", "--debug", "20", path], backend=B)
print("pics: %.1f s, image %s" % (time.time() - t0, out.shape))
x, t = out.reshape(-1, order='F'), img.reshape(-1, order='F')
scale = np.vdot(x, t) / np.vdot(x, x)
print("distance to the phantom after 5 iterations (up to the driver's normalisation): %.3f" % (np.linalg.norm(scale * x - t) / np.linalg.norm(t)))

"""fftn of a grid with two chirp-z axes against its smooth neighbour: (640, 277, 410) x 8 -- the grid int(N * osf) of the reference's
driver gives its own 480 x 208 x 308 scan (examples/pics.py:86, indigo/backends/backend.py:427-430) -- against (640, 288, 400) x 8."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from indigo_amd.backends import get_backend
from indigo_amd.util import rand64c

B = get_backend("hip")
c64 = np.dtype('complex64')
res = {}
for shape in ((640, 288, 400, 8), (640, 277, 410, 8)):
    x = B.empty_array(shape, c64)
    for j in range(shape[3]):
        x[:, :, :, j:j + 1].copy_from(rand64c(*shape[:3], 1, seed=j))
    y = B.zero_array(shape, c64)
    print(shape, B.fft_describe(shape), flush=True)
    for _ in range(2):
        B.fftn(y, x)
    B.barrier()
    t0 = time.perf_counter()
    n = 5
    for _ in range(n):
        B.fftn(y, x)
    B.barrier()
    ms = (time.perf_counter() - t0) / n * 1e3
    B.profile(True)
    B.fftn(y, x)
    B.profile(False)
    print("   ", {k: round(v['avg_ms'] * v['launches'], 2) for k, v in B.profile_report().items()})
    v = x[:, :, :, 1:2].to_host()[..., 0]
    ref = np.fft.fftn(v.astype(np.complex128))
    got = y[:, :, :, 1:2].to_host()[..., 0]
    err = np.linalg.norm(got - ref) / np.linalg.norm(ref)
    res[shape] = ms
    print("%s: %.2f ms per transform (%.2f TB/s by 4 * nbytes), volume 1 vs numpy %.2e" % (shape, ms, 4 * x.nbytes / ms / 1e9, err), flush=True)
    del x, y
print("ratio chirp-z grid / smooth grid: %.2f" % (res[(640, 277, 410, 8)] / res[(640, 288, 400, 8)]))

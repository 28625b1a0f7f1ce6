"""How far do two float32 CG implementations (fused device-scalar loop, host-scalar reference loop) drift from a CG whose
vector arithmetic runs in float64 (the operator stays the float32 HIP operator)?  Lab script behind the tolerances of
tests/test_hip_operators.py::test_cg_with_device_scalars_matches_host_scalar_cg."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from indigo_amd.backends import get_backend
from indigo_amd.backends.backend import Backend
from indigo_amd.sense import SenseProblem, normal_operator
from indigo_amd.util import rand64c

hip = get_backend("hip")
p = SenseProblem.synthetic((32, 32, 32), 4, nspokes=96, nreadout=64, width=2, oversamp=2.0, seed=4)
A = p.build_fused(hip)
AHA = normal_operator(A, lamda=0.05)
b = A.H * rand64c(A.shape[0], 1, seed=2)


def cg64(iters):
    x = np.zeros(b.shape, np.complex128)
    r = b.astype(np.complex128).copy()
    pp = r.copy()
    rr = np.vdot(r, r).real
    r0 = rr
    hist = []
    for _ in range(iters):
        Ap = (AHA * pp.astype(np.complex64)).astype(np.complex128)
        alpha = rr / np.vdot(pp, Ap).real
        x += alpha * pp
        r -= alpha * Ap
        r2 = np.vdot(r, r).real
        pp = r + (r2 / rr) * pp
        rr = r2
        hist.append(np.sqrt(rr / r0))
    return hist, x


rel = lambda a, c: np.linalg.norm(a - c) / np.linalg.norm(c)
for iters in (3, 5, 7, 10):
    x1 = np.zeros_like(b, order='F'); x2 = np.zeros_like(b, order='F')
    h1 = hip.cg(AHA, b.copy(order='F'), x1, maxiter=iters)
    h2 = Backend.cg(hip, AHA, b.copy(order='F'), x2, maxiter=iters)
    h64, x64 = cg64(iters)
    print("iters %2d: fused vs f64 x %.2e hist %.2e | host-scalar vs f64 x %.2e hist %.2e | fused vs host-scalar x %.2e hist %.2e" % (
        iters, rel(x1, x64), max(abs(np.array(h1) / np.array(h64) - 1)), rel(x2, x64), max(abs(np.array(h2) / np.array(h64) - 1)),
        rel(x1, x2), max(abs(np.array(h1) / np.array(h2) - 1))), flush=True)

"""ORACLE -- TEST INFRASTRUCTURE ONLY.  Double-precision evaluation of one coil of the SENSE operator.

The parity oracle proper is oracle/np_backend.py (the reference's numpy backend restated: complex64 storage, scipy's
complex64 csr product, numpy's single-precision pocketfft).  On the full-size benchmark problem that arithmetic is
itself only good to about 2.6e-5 on the adjoint: the inputs are uniform[0, 1) -- a large DC term -- and the k-space
centre, where a radial trajectory puts ~1e4 samples onto one grid point, is summed in a complex64 running sum.  This
module evaluates the SAME operator (same matrices, same weights, same transforms: indigo/backends/backend.py:403-442
NUFFT = G * F * Z * R, examples/pics.py:92-95 A = KronI(C, NUFFT) * VStack(Diag(maps))) in complex128, so that a
test can tell which of two complex64 results is the inaccurate one:

    || hip - oracle64 ||  <=  || oracle64 - exact || + tol * || exact ||,   and   || hip - exact || <= tol * || exact ||

Only tests/ and bench.py's parity leg import it.
"""
import numpy as np


class CoilOperatorF64(object):
    """A_c = G' * FFT * zeropad * diag(w_c) of a SenseProblem, in complex128 (unnormalised transforms, as np.py:102-115)"""

    def __init__(self, problem, coil):
        p = problem
        self.N, self.oN = p.N, p.oN
        # the gridding matrix in whichever grid order the problem already holds (0: (x, y, z) columns; 1: (x, z, y)) -- a second
        # copy of a 4e8-nonzero matrix is not worth ten seconds and five gigabytes
        self.layout = 1 if (1 in p._interp_cache and 0 not in p._interp_cache) else 0
        self.G = p.fused_interp(self.layout).astype(np.complex128)
        self.w = p.fused_weights([coil])[..., 0].astype(np.complex128)
        lo = tuple(m // 2 + int(np.ceil(-n / 2)) for m, n in zip(self.oN, self.N))
        self.sl = tuple(slice(l, l + b) for l, b in zip(lo, self.N))

    def forward(self, x):
        full = np.zeros(self.oN, dtype=np.complex128, order='F')
        full[self.sl] = self.w * np.asarray(x, dtype=np.complex128).reshape(self.N, order='F')
        F = np.fft.fftn(full)
        if self.layout == 1:
            F = F.transpose(0, 2, 1)
        return self.G @ F.reshape(-1, order='F')

    def adjoint(self, k):
        # G^H k = conj(G^T conj(k)): the transpose of a CSR matrix is a CSC VIEW, its product a scatter over the same arrays -- no
        # transposed copy of the matrix is ever built
        g = np.conj(self.G.T @ np.conj(np.asarray(k, dtype=np.complex128).reshape(-1)))
        if self.layout == 1:
            vol = g.reshape((self.oN[0], self.oN[2], self.oN[1]), order='F').transpose(0, 2, 1)
        else:
            vol = g.reshape(self.oN, order='F')
        inv = np.fft.ifftn(vol) * np.prod(self.oN)
        return (np.conj(self.w) * inv[self.sl]).reshape(-1, order='F')

    def normal(self, x):
        return self.adjoint(self.forward(x))

#!/usr/bin/env python3
"""How much of the 64-column forward gather (BASELINE config 3) is the memory system?  The same matrix structure (T rows x 27
taps) with its column indices folded into a panel that fits the L1 (64 rows = 32 KB), the L2 (4096 rows = 2 MB), the Infinity
Cache (262144 rows = 128 MB) -- against the real indices.  Prints the gather kernel's time for each."""
import os, sys, time
import numpy as np
import scipy.sparse as spp
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from scipy.signal.windows import kaiser
from indigo_amd.backends import get_backend
from indigo_amd.interp import interp_csr_arrays
from indigo_amd.sense import radial_trajectory
from indigo_amd.util import rand64c

B = get_backend("hip")
n = 256
coord = radial_trajectory(3617, 2 * n, seed=3)
T = int(np.prod(coord.shape[1:]))
beta = np.pi * np.sqrt(((2 * 2.0 / 2.0) * (2.0 - 0.5)) ** 2 - 0.8)
table = kaiser(2 * 128 + 1, beta)[128:]
indptr, indices, w = interp_csr_arrays(T, (n, n, n), 2, table, coord.reshape(3, -1, order='F'), dtype=np.float32)
c64 = np.dtype('complex64')
for fold in (64, 4096, 262144, 0):
    idx = indices if not fold else (indices % fold).astype(np.int32)
    K = n ** 3 if not fold else fold
    G = spp.csr_matrix((w.astype(np.complex64), idx, indptr), shape=(T, K))
    G.has_canonical_format = False
    G.sum_duplicates() if False else None
    S = B.SpMatrix(G, name='probe')
    X = B.copy_array(rand64c(K, 64, seed=1)) if K <= 262144 else None
    if X is None:
        X = B.empty_array((K, 64), c64)
        for j0 in range(0, 64, 8):
            X[:, j0:j0 + 8].copy_from(rand64c(K, 8, seed=100 + j0))
    Y = B.zero_array((T, 64), c64)
    S.eval(Y, X)
    B.barrier()
    B.profile(True)
    for _ in range(5):
        S.eval(Y, X)
    B.barrier()
    B.profile(False)
    prof = B.profile_report()
    print("fold %7d:" % fold, {k: round(v['avg_ms'], 3) for k, v in prof.items()}, flush=True)
    del S, X, Y

#!/usr/bin/env python3
"""Would a spatial order of the SAMPLES help the 64-column forward gather (BASELINE config 3)?  The same gridding matrix with
its rows sorted by the brick of B^3 grid cells their first tap falls into (bricks in raster or Morton order), timed through the
ordinary forward path -- the results land in sorted order, i.e. the write side is NOT what a permuted product would pay; this
measures the read side only.  Prints the gather kernel's time per order."""
import os, sys
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
G0 = spp.csr_matrix((w.astype(np.complex64), indices, indptr), shape=(T, n ** 3))
first = indices[indptr[:-1].clip(max=indices.size - 1)].astype(np.int64)
x, y, z = first % n, (first // n) % n, first // (n * n)


def morton(a, b, c):
    key = np.zeros(a.shape, dtype=np.int64)
    for bit in range(8):
        key |= ((a >> bit) & 1) << (3 * bit) | ((b >> bit) & 1) << (3 * bit + 1) | ((c >> bit) & 1) << (3 * bit + 2)
    return key


X = B.empty_array((n ** 3, 64), c64)
for j0 in range(0, 64, 8):
    X[:, j0:j0 + 8].copy_from(rand64c(n ** 3, 8, seed=100 + j0))
orders = [("acquisition", None)]
for bsz in (4, 8, 16, 32):
    orders.append(("raster %d^3" % bsz, (x // bsz) + (n // bsz) * ((y // bsz) + (n // bsz) * (z // bsz))))
    orders.append(("morton %d^3" % bsz, morton(x // bsz, y // bsz, z // bsz)))
for name, key in orders:
    G = G0 if key is None else G0[np.argsort(key, kind='stable')]
    S = B.SpMatrix(G, name='probe')
    Y = B.zero_array((T, 64), c64)
    S.eval(Y, X)
    B.barrier()
    B.profile(True)
    for _ in range(5):
        S.eval(Y, X)
    B.barrier()
    B.profile(False)
    prof = B.profile_report()
    print("%-16s" % name, {k: round(v['avg_ms'], 3) for k, v in prof.items()}, flush=True)
    del S, Y

#!/usr/bin/env python3
"""GPU check of the two launches of the 256^3 transform one at a time (INDIGO_HIP_FFT_2LAUNCH=2: launch A only,
=3: launch B only), each against the partial transform it is meant to compute.  Development aid.
    INDIGO_HIP_FFT_2LAUNCH=2 python tools/fft2pass_check.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from indigo_amd.backends import get_backend
from indigo_amd.util import rand64c

mode = int(os.environ.get("INDIGO_HIP_FFT_2LAUNCH", "1"))
B = get_backend("hip")
n = 256
x = rand64c(n, n, n, 2, seed=5)
if os.environ.get('ZERO_MEAN'):
    x = (x - (0.5 + 0.5j)).astype(np.complex64)
W = np.exp(-2j * np.pi * np.arange(n) / n)
x_d = B.copy_array(x)
y_d = B.zero_array(x.shape, np.dtype('complex64'))
B.fftn(y_d, x_d)
got = y_d.to_host()[..., 1].astype(np.complex128)
v = x[..., 1].astype(np.complex128)
if mode == 2:       # x transform + y stage 1: out[kx, k1 + 64 n2, z] = w256^(n2 k1) sum_n1 X[kx, 4 n1 + n2, z] w64^(n1 k1)
    fx = np.fft.fft(v, axis=0)
    exp = np.empty_like(fx)
    for n2 in range(4):
        exp[:, 64 * n2:64 * n2 + 64, :] = np.fft.fft(fx[:, n2::4, :], axis=1) * W[(n2 * np.arange(64)) & 255][None, :, None]
elif mode == 3:     # y stage 2 + z transform applied to the raw input
    exp = np.empty_like(v)
    t = np.fft.fft(v, axis=2)
    for k1 in range(64):
        exp[:, k1::64, :] = np.fft.fft(t[:, k1::64, :], axis=1)
else:
    exp = np.fft.fftn(v)
err = np.abs(got - exp)
print("mode", mode, "rel err", np.linalg.norm(err) / np.linalg.norm(exp), "max", err.max() / np.abs(exp).max())
bad = np.argwhere(err > 1e-3 * np.abs(exp).max())
print("bad elements:", len(bad), "of", err.size)
if len(bad):
    for ax, name in enumerate("xyz"):
        u, c = np.unique(bad[:, ax], return_counts=True)
        print(" ", name, "values involved:", len(u), "first:", u[:24])

if len(bad):
    i = tuple(bad[0]); print("first bad", i, "got", got[i], "exp", exp[i])
    ln = bad[(bad[:, 1] == bad[0][1]) & (bad[:, 2] == bad[0][2])]
    print("bad kx on that line:", ln[:, 0])

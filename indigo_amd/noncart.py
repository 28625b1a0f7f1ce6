"""Kaiser-Bessel gridding helpers: Fourier transform of the kernel and the 3-D
roll-off (apodisation-correction) volume.  Same formulas as the reference
(indigo/noncart.py:5-23), in plain numpy (numexpr is not a dependency here).
"""
import numpy as np


def ftkb(beta, x):
    """sinh(a)/a with a = sqrt(beta^2 - (pi x)^2); 1 where a == 0."""
    x = np.asarray(x, dtype=np.float64)
    a = np.sqrt(beta ** 2 - (np.pi * x) ** 2)
    out = np.ones_like(a)
    nz = a != 0.0
    out[nz] = np.sinh(a[nz]) / a[nz]
    return out


def rolloff3(oversamp, width, beta, N):
    """Roll-off volume of shape N: ftkb(0)^3 / prod_d ftkb((i_d - N_d//2) / N_d * 2 width / oversamp)."""
    scale = width * 2.0 / oversamp
    axes = [ftkb(beta, (np.arange(n) - n // 2) / n * scale) for n in N]
    denom = axes[0][:, None, None] * axes[1][None, :, None] * axes[2][None, None, :]
    return float(ftkb(beta, 0.0)) ** 3 / denom

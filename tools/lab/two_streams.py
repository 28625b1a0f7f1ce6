#!/usr/bin/env python3
"""Would evaluating coil chunks on two streams pay?  Two independent 8-coil A^H A evaluations (two HipBackend instances = two
contexts = two streams on the one GPU), first one after the other, then enqueued side by side: the gridding kernels are bound by
latency and LDS (3 TB/s), the transform passes by bandwidth (5+ TB/s) -- if they overlap, two evaluations take less than twice one."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from indigo_amd.backends import get_backend
from indigo_amd.sense import SenseProblem, normal_operator
from indigo_amd.util import rand64c

B1, B2 = get_backend("hip"), get_backend("hip")
ops = []
for B in (B1, B2):
    p = SenseProblem.synthetic((256,) * 3, 8, nspokes=3617, nreadout=512, width=2, ntable=128, oversamp=2.0, seed=4)
    A = p.build_zpadfft(B)
    AHA = normal_operator(A)
    x = B.copy_array(rand64c(A.shape[1], 1, seed=1))
    y = B.zero_array((A.shape[1], 1), np.complex64)
    AHA.eval(y, x); B.barrier()
    ops.append((B, AHA, x, y))
    p.drop_cache() if hasattr(p, 'drop_cache') else None

def run(mode, steps=10):
    for B, *_ in ops: B.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        if mode == "sequential":
            for B, AHA, x, y in ops:
                AHA.eval(y, x)
                B.barrier()
        else:
            for B, AHA, x, y in ops:
                AHA.eval(y, x)
            for B, *_ in ops: B.barrier()
    return (time.perf_counter() - t0) / steps * 1e3

for mode in ("sequential", "side by side", "sequential", "side by side"):
    print("%-13s %.3f ms per pair of evaluations" % (mode, run(mode)))

#!/usr/bin/env python3
"""LAB (round 5): the compact intermediate of the zero-padded / cropped transforms (L1, behind the full-size part of the workspace)
shifted by IG_LAB_L1_SHIFT elements, all in ONE process (same physical placement of every buffer): per-pass times of the headline.
The switch (a getenv in exec_padded_layout2 / exec_cropped_layout2 plus 32 MB more workspace) existed for that run only:
profiles/r05_l1_shift_sweep.txt -- the shift moves the y passes by less than 2.5 %; what matters is the allocation."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from indigo_amd.backends import get_backend
from indigo_amd.sense import SenseProblem, normal_operator
from indigo_amd.util import rand64c

B = get_backend("hip")
p = SenseProblem.synthetic((256,) * 3, 8, nspokes=3617, nreadout=512, width=2, ntable=128, oversamp=2.0, seed=4)
A = p.build_zpadfft(B)
AHA = normal_operator(A, lamda=0.0)
x = B.copy_array(rand64c(A.shape[1], 1, seed=1))
y = B.zero_array((A.shape[1], 1), np.dtype('complex64'))
for _ in range(3):
    AHA.eval(y, x)
B.barrier()
shifts = [int(s) for s in sys.argv[1:]] or [0, 16, 32, 48, 64, 4096, 4096 + 16, 65536, 131072, 262144, 524288, 1048576, 1048576 + 16, 2097152, 0]
for s in shifts:
    os.environ["IG_LAB_L1_SHIFT"] = str(s)
    for _ in range(2):
        AHA.eval(y, x)
    B.barrier()
    B.profile(True)
    t0 = time.perf_counter()
    for _ in range(10):
        AHA.eval(y, x)
    B.barrier()
    t = (time.perf_counter() - t0) / 10
    B.profile(False)
    rep = B.profile_report()
    print("shift %8d (%9d B)  eval %.3f ms  %s" % (s, s * 8, sum(v['total_ms'] for v in rep.values()) / 10,
          " ".join("%s %.3f" % (k[4:], v['avg_ms']) for k, v in sorted(rep.items()) if k.startswith('fft_'))), flush=True)

#!/usr/bin/env python3
"""LAB (round 5): one process, the headline operator, the scratch arena re-allocated several times (plain allocation, the old arenas
kept alive so that every new one lands on other memory): per-pass times and the placement probe of each arena."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from indigo_amd.backends import get_backend
from indigo_amd.sense import SenseProblem, normal_operator
from indigo_amd.util import rand64c
from indigo_amd.analyses import ScratchUsage

B = get_backend("hip")
B.tuning['placement_candidates'] = 1
p = SenseProblem.synthetic((256,) * 3, 8, nspokes=3617, nreadout=512, width=2, ntable=128, oversamp=2.0, seed=4)
A = p.build_zpadfft(B)
AHA = normal_operator(A, lamda=0.0)
C64 = np.dtype('complex64')
x = B.copy_array(rand64c(A.shape[1], 1, seed=1))
y = B.zero_array((A.shape[1], 1), C64)
need = ScratchUsage().measure(AHA, 1)
keep = []
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 8):
    if getattr(B, '_scratch', None) is not None:
        keep.append(B._scratch)
    B.reserve_scratch(need)
    ms = ctypes.c_double(0.0)
    B._check(B._L.ig_probe_placement(B._ctx, ctypes.c_void_p(B._scratch._arr), B._scratch.nbytes, ctypes.byref(ms)), "probe")
    for _ in range(3):
        AHA.eval(y, x)
    B.barrier()
    B.profile(True)
    for _ in range(10):
        AHA.eval(y, x)
    B.barrier()
    B.profile(False)
    rep = B.profile_report()
    print("arena %d at %#x  probe %.3f ms  eval %.3f ms  %s" % (i, B._scratch._arr, ms.value, sum(v['total_ms'] for v in rep.values()) / 10,
          " ".join("%s %.3f" % (k[4:], v['avg_ms']) for k, v in sorted(rep.items()) if k.startswith('fft_'))), flush=True)

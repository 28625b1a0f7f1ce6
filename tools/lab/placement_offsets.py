#!/usr/bin/env python3
"""LAB (round 6): does the placement lottery of the y passes live INSIDE an allocation as well?  Several big allocations
(each 1.5 x the headline arena), the placement probe (ig_probe_placement: the y passes' write pattern) run on 20.8 GB windows at
1 GB steps inside each -- and on the same window twice, to see the probe's own noise.  If windows of one allocation differ as much as
whole allocations do, an arena can be the best-placed sub-range of ONE allocation instead of the best of three allocations."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from indigo_amd.backends import get_backend

B = get_backend("hip")
B.tuning['placement_candidates'] = 1
L = B._L
ARENA = 20_803_747_840 // 4096 * 4096          # the headline's scratch arena
GB = 1 << 30


def probe(ptr, nbytes):
    ms = ctypes.c_double(0.0)
    B._check(L.ig_probe_placement(B._ctx, ctypes.c_void_p(ptr), nbytes, ctypes.byref(ms)), "probe")
    return ms.value


bufs = []
for a in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    ptr = ctypes.c_void_p()
    total = ARENA + 10 * GB
    B._check(L.ig_malloc(B._ctx, total, ctypes.byref(ptr)), "malloc")
    bufs.append(ptr.value)
    row = []
    for off in range(0, 11):
        t1 = probe(ptr.value + off * GB, ARENA)
        t2 = probe(ptr.value + off * GB, ARENA)
        row.append((min(t1, t2), abs(t1 - t2)))
    print("allocation %d at %#x: windows at +0 .. +10 GB: %s   (repeat spread max %.3f ms)" % (
        a, ptr.value, " ".join("%.3f" % t for t, _ in row), max(d for _, d in row)), flush=True)
    # sub-GB offsets of the first window
    fine = [min(probe(ptr.value + o, ARENA), probe(ptr.value + o, ARENA)) for o in (0, 2 << 20, 64 << 20, 256 << 20, 512 << 20)]
    print("   offsets 0, 2 MB, 64 MB, 256 MB, 512 MB: %s" % " ".join("%.3f" % t for t in fine), flush=True)

#!/usr/bin/env python3
"""What does a tile that leaves at once cost?  The zero-pad-aware passes launch one workgroup per tile of the grid and let the
tiles outside the k-space hulls return after their three support loads.  Here EVERY tile is outside (an all-empty support
table): the time of a pass is then the cost of its empty workgroups alone."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from indigo_amd.backends import get_backend

B = get_backend("hip")
oN, N, C, tile = (512,) * 3, (256,) * 3, 8, 4
nt = oN[0] // tile
table = np.zeros(2 * (oN[1] * nt + nt) + 2 * oN[1] * nt * 16, dtype=np.int16)
if len(sys.argv) > 1 and sys.argv[1] == "full":
    # every tile inside the hulls, every row flagged: what a z pass costs per tile when it moves ALL of its bytes
    rng = table[:2 * (oN[1] * nt + nt)].reshape(-1, 2)
    rng[:oN[1] * nt] = (0, oN[2])
    rng[oN[1] * nt:] = (0, oN[1])
    table[2 * (oN[1] * nt + nt):] = -1
sup = B.copy_array(table)
P, n = int(np.prod(oN)), int(np.prod(N))
y = B.zero_array((P, C), np.complex64)
x = B.zero_array((n, 1), np.complex64)
w = B.zero_array((n, C), np.complex64)
ws = B.zero_array((max(B._fft_padded_workspace(oN, (128,) * 3, N, C, 2) // 8, 1),), np.complex64)
for rep in range(2):
    B.profile(rep == 1)
    for _ in range(5):
        B.ifft_cropped_sum(x, y, w, oN, (128,) * 3, N, ws, support=sup, support_tile=tile)
        B.fft_padded(y, x, w, oN, (128,) * 3, N, workspace=ws, layout=2, support=sup, support_tile=tile)
    B.barrier()
B.profile(False)
for k, v in sorted(B.profile_report().items()):
    print("%-14s %3d launches  avg %.4f ms" % (k, v['launches'], v['avg_ms']))
print("z passes launch 128 x 512 = 65536 workgroups, y passes 128 x 256 = 32768; full table: crop z reads 8.59 + writes 4.29 GB, pad z the reverse")

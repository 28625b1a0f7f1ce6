#!/usr/bin/env python3
"""Leaf-kernel micro-benchmarks on the GPU box (development aid, not the headline bench).

    python tools/kernel_lab.py fft 512 512 512 8        # 3-D FFT, batch 8
    python tools/kernel_lab.py spmm 256 8               # gridding matrix on the (2*256)^3 grid, 8 columns
    python tools/kernel_lab.py axpby 134217728

Prints the per-kernel profile (average launch duration from stream events).
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from indigo_amd.backends import get_backend
from indigo_amd.util import rand64c

def report(B, reps, nbytes=None):
    for k, v in sorted(B.profile_report().items(), key=lambda kv: -kv[1]['total_ms']):
        b = v['bytes'] / v['launches'] if v['bytes'] else (nbytes.get(k) if nbytes else None)
        rate = (" %7.0f GB/s (algorithmic)" % (b / v['avg_ms'] / 1e6)) if b else ""
        print("  %-26s %4d launches  avg %8.3f ms%s" % (k, v['launches'], v['avg_ms'], rate))

def main():
    B = get_backend("hip")
    kind = sys.argv[1]
    reps = int(os.environ.get("REPS", "5"))
    c64 = np.dtype('complex64')
    if kind == "fft":
        shape = tuple(int(a) for a in sys.argv[2:])
        x = B.copy_array(rand64c(*shape, seed=1))
        y = B.zero_array(shape, c64)
        print(B.fft_describe(shape))
        B.fftn(y, x); B.barrier()
        B.profile(True)
        for _ in range(reps):
            B.fftn(y, x)
            B.ifftn(y, y)
        B.profile(False)
        report(B, reps)
    elif kind == "spmm":
        from indigo_amd.sense import SenseProblem
        img, ncol = int(sys.argv[2]), int(sys.argv[3])
        p = SenseProblem.synthetic((img,) * 3, 1, nspokes=int(round(3617 * (img / 256.0) ** 2)), nreadout=2 * img, seed=4)
        G = p.fused_interp()
        S = B.SpMatrix(G, name='G')
        P, T = G.shape[1], G.shape[0]
        x = B.copy_array(rand64c(P, ncol, seed=1)); k = B.zero_array((T, ncol), c64)
        xa = B.zero_array((P, ncol), c64)
        S.eval(k, x); S.H.eval(xa, k); B.barrier()
        M = S._matrix_d
        fb = S.csrmm_bytes(x, k, 0, True); ab = S.csrmm_bytes(k, xa, 0, False)
        print("G': %d x %d nnz %d col_frac %.3f; fwd %.2f GB adj %.2f GB (reference model)" % (T, P, G.nnz, M._col_frac, fb / 1e9, ab / 1e9))
        B.profile(True)
        for _ in range(reps):
            S.eval(k, x)
            S.H.eval(xa, k)
        B.profile(False)
        report(B, reps, {"csrmm_gather": fb, "csrmm_gather_conj": ab})
    elif kind == "axpby":
        n = int(sys.argv[2])
        x = B.copy_array(rand64c(n, seed=1)); y = B.copy_array(rand64c(n, seed=2))
        B.axpby(0.5, y, 2.0, x); B.barrier()
        B.profile(True)
        for _ in range(reps):
            B.axpby(0.5, y, 2.0, x); B.axpby(0, y, 2.0, x); B.scale(y, 0.5)
        B.profile(False)
        report(B, reps)



def empty_rows_probe():
    """cost of the row-per-lane kernel on an all-empty 134M-row matrix with 8 columns: rowptr read + panel write only"""
    import scipy.sparse as spp
    B = get_backend("hip")
    P, ncol = 512 ** 3, 8
    c64 = np.dtype('complex64')
    A = spp.csr_matrix((P, 1000), dtype=c64)
    S = B.SpMatrix(A, name='empty')
    x = B.copy_array(rand64c(1000, ncol, seed=1))
    pad = int(sys.argv[2]) if len(sys.argv) > 2 else 0      # extra rows of leading dimension (de-aliases the 1 GiB column stride)
    y = B.zero_array((P + pad, ncol), c64)
    if pad: y = y[:P]
    print("ld =", y._leading_dim)
    S.eval(y, x); B.barrier()
    B.profile(True)
    for _ in range(5):
        S.eval(y, x)
    B.profile(False)
    report(B, 5)


def cfg3():
    """BASELINE config 3: 3-D radial gridding CSR (1,851,904 x 256^3, 27 taps/row, nnz ~5e7) x 64-column panel"""
    from indigo_amd.sense import SenseProblem
    B = get_backend("hip")
    c64 = np.dtype('complex64')
    ncol = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    p = SenseProblem.synthetic((128,) * 3, 1, nspokes=3617, nreadout=512, seed=3)
    G = p.fused_interp()
    S = B.SpMatrix(G, name='G')
    T, P = G.shape
    x = B.copy_array(rand64c(P, ncol, seed=1)); k = B.zero_array((T, ncol), c64); xa = B.zero_array((P, ncol), c64)
    S.eval(k, x); S.H.eval(xa, k); B.barrier()
    fb = S.csrmm_bytes(x, k, 0, True); ab = S.csrmm_bytes(k, xa, 0, False)
    print("cfg3 G: %d x %d nnz %d col_frac %.3f; algorithmic fwd %.2f GB adj %.2f GB" % (T, P, G.nnz, S._matrix_d._col_frac, fb / 1e9, ab / 1e9))
    B.profile(True)
    for _ in range(5):
        S.eval(k, x)
        S.H.eval(xa, k)
    B.profile(False)
    report(B, 5, {"csrmm_gather": fb, "csrmm_gather_conj": ab, "csrmm_rowlane_conj": ab})


def zpad_probe():
    """fused padded transform, layout 2, no support table, straight through the C ABI"""
    import ctypes
    B = get_backend("hip")
    c64 = np.dtype('complex64')
    grid, box, C = (512, 512, 512), (256, 256, 256), 8
    lo = tuple(m // 2 - n // 2 for m, n in zip(grid, box))
    P, N = int(np.prod(grid)), int(np.prod(box))
    plan, ws = B._padded_plan(grid, lo, box, C, 2)
    x = B.copy_array(rand64c(N, 1, seed=1))
    w = B.copy_array(rand64c(N * C, 1, seed=2))
    y = B.zero_array((P * C,), c64)
    work = B.zero_array((ws // 8,), c64)
    def run():
        B._check(B._L.ig_fft_exec_padded(plan, ctypes.c_void_p(x._arr), 0, ctypes.c_void_p(w._arr), ctypes.c_void_p(y._arr),
                                         ctypes.c_void_p(work._arr), None), "pad")
    run(); B.barrier()
    B.profile(True)
    for _ in range(5):
        run()
    B.profile(False)
    report(B, 5)


if len(sys.argv) > 1 and sys.argv[1] == "zpad":
    zpad_probe()
elif len(sys.argv) > 1 and sys.argv[1] == "empty":
    empty_rows_probe()
elif len(sys.argv) > 1 and sys.argv[1] == "cfg3":
    cfg3()
elif __name__ == "__main__":
    main()

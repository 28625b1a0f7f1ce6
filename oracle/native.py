"""ORACLE -- TEST INFRASTRUCTURE ONLY.  Loaders for the two native CPU oracles.

  * `c_oracle()`   -> ctypes handle of oracle/liboracle_csrmm.so (our C restatement,
                      oracle/csrmm_oracle.c), with numpy-friendly wrappers below.
  * `ref_native()` -> the reference's own `_customcpu` extension module, compiled by
                      `make -C oracle ref` from /root/reference/indigo/backends/_customcpu.c
                      into oracle/_ref/ (None if it has not been built).  Loaded with
                      RTLD_LAZY because the reference leaves `mkl_ccsrmv` undefined; never
                      call its forward product with a single column (that branch needs MKL).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this.
"""
import ctypes
import importlib.machinery
import importlib.util
import os
import subprocess
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_C = None
_REF = False


def build(ref=True, quiet=True):
    """`make` the C restatement and, when the reference tree is present, the reference build."""
    targets = ["all"]
    if ref and os.path.exists("/root/reference/indigo/backends/_customcpu.c"):
        targets.append("ref")
    r = subprocess.run(["make", "-C", _HERE] + targets, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("oracle build failed:\n" + r.stdout)
    if not quiet:
        print(r.stdout)


def c_oracle():
    global _C
    if _C is None:
        path = os.path.join(_HERE, "liboracle_csrmm.so")
        if not os.path.exists(path):
            build(ref=False)
        _C = ctypes.CDLL(path)
        _C.oracle_ccsrmm.restype = None
        _C.oracle_ccsrmm.argtypes = [ctypes.c_int, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64,
                                     ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                     ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p,
                                     ctypes.c_int64]
        _C.oracle_inspect.restype = None
        _C.oracle_inspect.argtypes = [ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p,
                                      ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64),
                                      ctypes.POINTER(ctypes.c_int)]
    return _C


def _p(a):
    return ctypes.c_void_p(a.ctypes.data)


def c_ccsrmm(A, X, Y, alpha=1, beta=0, adjoint=False):
    """Y <- alpha*op(A)*X + beta*Y in place with the C restatement.  A: scipy CSR complex64; X, Y: F-ordered complex64."""
    assert A.dtype == np.complex64 and X.dtype == np.complex64 and Y.dtype == np.complex64
    assert X.flags['F_CONTIGUOUS'] and Y.flags['F_CONTIGUOUS']
    M, K = A.shape
    a = np.array([complex(alpha).real, complex(alpha).imag], dtype=np.float32)
    b = np.array([complex(beta).real, complex(beta).imag], dtype=np.float32)
    indptr = np.ascontiguousarray(A.indptr, dtype=np.int32)
    indices = np.ascontiguousarray(A.indices, dtype=np.int32)
    X2 = X.reshape(X.shape[0], -1, order='F')
    Y2 = Y.reshape(Y.shape[0], -1, order='F')
    c_oracle().oracle_ccsrmm(1 if adjoint else 0, M, X2.shape[1], K, _p(a), _p(A.data), _p(indices), _p(indptr),
                             _p(X2), X2.shape[0], _p(b), _p(Y2), Y2.shape[0])
    return Y


def c_inspect(A):
    indptr = np.ascontiguousarray(A.indptr, dtype=np.int32)
    indices = np.ascontiguousarray(A.indices, dtype=np.int32)
    r, c, e = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int()
    c_oracle().oracle_inspect(A.shape[0], A.shape[1], _p(indices), _p(indptr), ctypes.byref(r), ctypes.byref(c), ctypes.byref(e))
    return r.value, c.value, bool(e.value)


def ref_native():
    """The reference's compiled `_customcpu` module, or None when oracle/_ref has not been built."""
    global _REF
    if _REF is False:
        path = os.path.join(_HERE, "_ref", "_customcpu.so")
        if not os.path.exists(path):
            _REF = None
        else:
            old = sys.getdlopenflags()
            try:
                sys.setdlopenflags(os.RTLD_LAZY)
                loader = importlib.machinery.ExtensionFileLoader("_customcpu", path)
                spec = importlib.util.spec_from_loader("_customcpu", loader)
                mod = importlib.util.module_from_spec(spec)
                loader.exec_module(mod)
                _REF = mod
            finally:
                sys.setdlopenflags(old)
    return _REF


def ref_ccsrmm(A, X, Y, alpha=1, beta=0, adjoint=False, exwrite=False):
    """Y <- alpha*op(A)*X + beta*Y in place with the REFERENCE's compiled kernel (X must have >= 2 columns when forward)."""
    mod = ref_native()
    assert mod is not None, "oracle/_ref not built (make -C oracle ref)"
    M, K = A.shape
    N = X.shape[1]
    assert adjoint or N > 1, "the reference's single-column forward branch needs MKL"
    indptr = np.ascontiguousarray(A.indptr, dtype=np.int32)
    indices = np.ascontiguousarray(A.indices, dtype=np.int32)
    mod.csrmm(bool(adjoint), M, N, K, complex(alpha), A.data, indices, indptr,
              X, X.shape[0], complex(beta), Y, Y.shape[0], bool(exwrite))
    return Y

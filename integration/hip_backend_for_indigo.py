"""`indigo/backends/hip.py` for the mbdriscoll/indigo tree: the reference-side binding of libindigo_hip.so.

Drop this file into the reference as indigo/backends/hip.py and register it in indigo/backends/__init__.py (INTEGRATION.md
section 1).  It subclasses the REFERENCE's `indigo.backends.backend.Backend` and marshals every leaf of the Backend contract into
one call of the C ABI (include/indigo_hip.h) -- the role indigo/backends/cuda.py plays for the CUDA libraries.  Each method names
the reference interface it replaces.  The library is found through INDIGO_HIP_LIB (default: libindigo_hip.so on the loader path).

This file is not imported by indigo_amd (whose own backends/hip.py is the complete, tuned binding on our restated Backend).  It is
executed by tests/test_reference_binding.py in the build container: under the imported reference, against the host-only shim of the
ABI (tests/abi_shim/), through the reference's own backend tests.
"""
import ctypes as C
import os

import numpy as np

from indigo.backends.backend import Backend

c_f, c_d, c_i, c_i64, c_sz, vp = C.c_float, C.c_double, C.c_int, C.c_int64, C.c_size_t, C.c_void_p

# symbol -> (restype, argtypes): must equal include/indigo_hip.h (tests/test_reference_binding.py compares this table with
# indigo_amd/_lib.py:PROTOTYPES, which tests/test_abi.py keeps equal to the header and to the library's exports)
PROTOTYPES = {
    "ig_init":        (c_i, [c_i, C.POINTER(vp)]),
    "ig_destroy":     (None, [vp]),
    "ig_last_error":  (C.c_char_p, [vp]),
    "ig_sync":        (c_i, [vp]),
    "ig_malloc":      (c_i, [vp, c_sz, C.POINTER(vp)]),
    "ig_free":        (c_i, [vp, vp]),
    "ig_memset0":     (c_i, [vp, vp, c_sz]),
    "ig_copy2d":      (c_i, [vp, vp, c_sz, vp, c_sz, c_sz, c_sz, c_i]),
    "ig_caxpby":      (c_i, [vp, c_i64, c_f, c_f, vp, c_f, c_f, vp]),
    "ig_cscal":       (c_i, [vp, c_i64, c_f, c_f, vp]),
    "ig_cdotc":       (c_i, [vp, c_i64, vp, vp, C.POINTER(c_d)]),
    "ig_scnrm2sq":    (c_i, [vp, c_i64, vp, C.POINTER(c_d)]),
    "ig_cmax":        (c_i, [vp, c_i64, c_f, vp]),
    "ig_ccsrmm":      (c_i, [vp, c_i, c_i, c_i64, c_i64, c_i64, c_i64, c_f, c_f, vp, vp, vp, vp, c_i64, c_f, c_f, vp, c_i64]),
    "ig_csr_inspect": (c_i, [vp, vp, c_i64, c_i64, C.POINTER(c_i64), C.POINTER(c_i64), C.POINTER(c_i)]),
    "ig_fft_plan":    (c_i, [vp, c_i, C.POINTER(c_i64), c_i64, C.POINTER(vp), C.POINTER(c_sz)]),
    "ig_fft_exec":    (c_i, [vp, vp, vp, c_i, vp]),
    "ig_fft_destroy": (c_i, [vp]),
    "ig_conemm":      (c_i, [vp, c_i64, c_i64, c_i64, c_f, c_f, vp, c_i64, c_f, c_f, vp, c_i64]),
    "ig_cdiamm":      (c_i, [vp, c_i, c_i64, c_i64, c_i64, c_i64, vp, vp, c_i64, c_f, c_f, vp, c_i64, c_f, c_f, vp, c_i64]),
    "ig_cgemm":       (c_i, [vp, c_i, c_i, c_i64, c_i64, c_i64, c_f, c_f, vp, c_i64, vp, c_i64, c_f, c_f, vp, c_i64]),
}

L = C.CDLL(os.environ.get("INDIGO_HIP_LIB", "libindigo_hip.so"))          # role of cuda.py:14-18
for _name, (_res, _args) in PROTOTYPES.items():
    getattr(L, _name).restype, getattr(L, _name).argtypes = _res, _args

H2D, D2H, D2D = 1, 2, 3


def _ck(rc, ctx):                                  # cuda.py:42-49: error code -> RuntimeError
    if rc:
        raise RuntimeError((L.ig_last_error(ctx) or b"libindigo_hip call failed").decode())


def _c(z):
    z = complex(z)
    return c_f(z.real), c_f(z.imag)


class HipBackend(Backend):

    def __init__(self, device_id=0):               # cuda.py:28-38
        super(HipBackend, self).__init__(device_id)
        self._ctx = vp()
        _ck(L.ig_init(device_id, C.byref(self._ctx)), None)
        self._plans = {}

    def barrier(self):                             # cuda.py:120-121
        _ck(L.ig_sync(self._ctx), self._ctx)

    # -- device arrays: cuda.py:126-208 --------------------------------------------------------------
    class dndarray(Backend.dndarray):
        def _malloc(self, shape, dtype):
            p = vp()
            _ck(L.ig_malloc(self._backend._ctx, int(self.nbytes), C.byref(p)), self._backend._ctx)      # 256-byte aligned by the library
            return C.c_ulong(p.value)              # `_arr` is a c_ulong device address, as in the CUDA backend

        def _free(self):
            L.ig_free(self._backend._ctx, vp(self._arr.value))

        def _zero(self):
            _ck(L.ig_memset0(self._backend._ctx, vp(self._arr.value), int(self.nbytes)), self._backend._ctx)

        def _pitched(self, dst, dld, src, sld, kind):
            """one ig_copy2d: columns of shape[0] items, `height` of them, leading dimensions in items (cuda.py:127-166)"""
            item = np.dtype(self.dtype).itemsize
            if self.ndim == 2:
                width, height = int(self.shape[0]) * item, int(self.shape[1])
            else:
                assert self.contiguous
                width, height, dld, sld = int(self.nbytes), 1, int(self.size), int(self.size)
            _ck(L.ig_copy2d(self._backend._ctx, vp(dst), int(dld) * item, vp(src), int(sld) * item, width, height, kind), self._backend._ctx)

        def _copy_from(self, arr):                 # host -> device
            assert arr.flags['F_CONTIGUOUS']
            self._pitched(self._arr.value, self._leading_dim, arr.ctypes.data, arr.shape[0], H2D)

        def _copy_to(self, arr):                   # device -> host
            assert arr.flags['F_CONTIGUOUS']
            self._pitched(arr.ctypes.data, arr.shape[0], self._arr.value, self._leading_dim, D2H)

        def _copy(self, d_arr):                    # device -> device
            self._pitched(self._arr.value, self._leading_dim, d_arr._arr.value, d_arr._leading_dim, D2D)

        def __getitem__(self, slc):                # a view: pointer + Fortran-order offset, same leading dimension (cuda.py:183-202)
            if isinstance(slc, slice):
                slc = [slc]
            first, extent = [], []
            for s, n in zip(slc, self.shape):
                if isinstance(s, int):
                    s = slice(s, s + 1)
                lo = 0 if s.start is None else s.start
                hi = n if s.stop is None else s.stop
                lo = lo + n if lo < 0 else lo
                hi = hi + n if hi < 0 else hi
                hi = min(max(hi, lo), n)
                first.append(lo)
                extent.append(hi - lo)
            off = int(np.ravel_multi_index(first, self.shape, order='F')) * np.dtype(self.dtype).itemsize
            return self._backend.dndarray(self._backend, tuple(extent), self.dtype, ld=self._leading_dim, own=False,
                                          data=C.c_ulong(self._arr.value + off))

        @staticmethod
        def from_param(obj):
            if not isinstance(obj, HipBackend.dndarray):
                raise C.ArgumentError('{} is not a dndarray'.format(type(obj)))
            return obj._arr

    # -- BLAS-1: backend.py:453-467, cuda.py:239-302 -------------------------------------------------
    def axpby(self, beta, y, alpha, x):
        _ck(L.ig_caxpby(self._ctx, int(y.size), *_c(beta), vp(y._arr.value), *_c(alpha), vp(x._arr.value)), self._ctx)

    def scale(self, x, alpha):
        _ck(L.ig_cscal(self._ctx, int(x.size), *_c(alpha), vp(x._arr.value)), self._ctx)

    def dot(self, x, y):                           # Re(x^H y): np.py:60-64
        out = (c_d * 2)()
        _ck(L.ig_cdotc(self._ctx, int(x.size), vp(x._arr.value), vp(y._arr.value), out), self._ctx)
        return out[0]

    def norm2(self, x):                            # ||x||^2: np.py:66-69
        out = c_d()
        _ck(L.ig_scnrm2sq(self._ctx, int(x.size), vp(x._arr.value), C.byref(out)), self._ctx)
        return out.value

    def max(self, val, arr):                       # backend.py:734, _customgpu.cu:7-13
        _ck(L.ig_cmax(self._ctx, int(arr.size) * 2, float(val), vp(arr._arr.value)), self._ctx)

    # -- sparse / dense products: backend.py:481-533 ---------------------------------------------------
    def ccsrmm(self, y, A_shape, A_indx, A_ptr, A_vals, x, alpha=1, beta=0, adjoint=False, exwrite=False):
        m, k = A_shape                             # backend.py:515, cuda.py:582-596
        _ck(L.ig_ccsrmm(self._ctx, int(adjoint), int(exwrite), m, k, int(x.shape[1]), int(A_vals.size), *_c(alpha),
                        vp(A_vals._arr.value), vp(A_indx._arr.value), vp(A_ptr._arr.value),
                        vp(x._arr.value), int(x._leading_dim), *_c(beta), vp(y._arr.value), int(y._leading_dim)), self._ctx)

    def cdiamm(self, y, shape, offsets, data, x, alpha=1.0, beta=0.0, adjoint=True):          # backend.py:521-526
        m, k = shape
        _ck(L.ig_cdiamm(self._ctx, int(adjoint), m, k, int(x.shape[1]), int(offsets.size), vp(offsets._arr.value), vp(data._arr.value),
                        int(data._leading_dim), *_c(alpha), vp(x._arr.value), int(x._leading_dim), *_c(beta), vp(y._arr.value),
                        int(y._leading_dim)), self._ctx)

    def onemm(self, y, x, alpha=1, beta=0):        # backend.py:528-533
        _ck(L.ig_conemm(self._ctx, int(y.shape[0]), int(x.shape[0]), int(x.shape[1]), *_c(alpha), vp(x._arr.value), int(x._leading_dim),
                        *_c(beta), vp(y._arr.value), int(y._leading_dim)), self._ctx)

    def cgemm(self, y, M, x, alpha, beta, forward):                                          # backend.py:481-485, cuda.py:314-360
        r, c = M.shape if forward else M.shape[::-1]
        x2, y2 = x.reshape((c, -1)), y.reshape((r, -1))
        _ck(L.ig_cgemm(self._ctx, 0 if forward else 1, 0, int(M.shape[0]), int(M.shape[1]), int(x2.shape[1]), *_c(alpha),
                       vp(M._arr.value), int(M._leading_dim), vp(x2._arr.value), int(x2._leading_dim), *_c(beta),
                       vp(y2._arr.value), int(y2._leading_dim)), self._ctx)

    def csymm(self, y, M, x, alpha, beta, left=True):                                       # backend.py:487-495, cuda.py:362-392
        n = int(M.shape[0])
        if left:
            x2, y2 = x.reshape((n, -1)), y.reshape((n, -1))
            p = x2.shape[1]
        else:
            x2, y2 = x.reshape((-1, n)), y.reshape((-1, n))
            p = x2.shape[0]
        _ck(L.ig_cgemm(self._ctx, 0, 0 if left else 1, n, n, int(p), *_c(alpha), vp(M._arr.value), int(M._leading_dim),
                       vp(x2._arr.value), int(x2._leading_dim), *_c(beta), vp(y2._arr.value), int(y2._leading_dim)), self._ctx)

    # -- FFT: backend.py:497-512, plan cache cuda.py:470-498 -------------------------------------------
    def _plan(self, shape):
        shape = tuple(int(s) for s in shape)
        if shape not in self._plans:
            dims = (c_i64 * (len(shape) - 1))(*shape[:-1])
            p, ws = vp(), c_sz()
            _ck(L.ig_fft_plan(self._ctx, len(shape) - 1, dims, shape[-1], C.byref(p), C.byref(ws)), self._ctx)
            self._plans[shape] = (p, ws.value)
        return self._plans[shape]

    def _fft_workspace_size(self, shape):
        return self._plan(shape)[1]

    def _fft(self, y, x, direction):
        p, ws = self._plan(x.shape)
        if ws:
            with self.scratch(nbytes=ws) as tmp:
                _ck(L.ig_fft_exec(p, vp(x._arr.value), vp(y._arr.value), direction, vp(tmp._arr.value)), self._ctx)
        else:
            _ck(L.ig_fft_exec(p, vp(x._arr.value), vp(y._arr.value), direction, None), self._ctx)

    def fftn(self, y, x):
        self._fft(y, x, -1)

    def ifftn(self, y, x):
        self._fft(y, x, +1)

    # -- csr_matrix: the reference only sets `_exwrite` when its optional _customcpu extension imports (backend.py:556-567) and reads it
    # unconditionally in adjoint() (:585); here the library's host routine does that analysis (replaces _customcpu.inspect) ----------
    class csr_matrix(Backend.csr_matrix):
        def __init__(self, backend, A, name='mat'):
            super(HipBackend.csr_matrix, self).__init__(backend, A, name)
            A = A.tocsr()
            indptr = np.ascontiguousarray(A.indptr, dtype=np.int32)
            indices = np.ascontiguousarray(A.indices, dtype=np.int32)
            nzrow, nzcol, exw = c_i64(), c_i64(), c_i()
            _ck(L.ig_csr_inspect(vp(indptr.ctypes.data), vp(indices.ctypes.data), A.shape[0], A.shape[1],
                                 C.byref(nzrow), C.byref(nzcol), C.byref(exw)), None)
            self._row_frac = nzrow.value / float(A.shape[0])
            self._col_frac = nzcol.value / float(A.shape[1])
            self._exwrite = bool(exw.value)

"""The `Backend` plugin surface: device arrays, device CSR matrices, the leaf
kernel contract, operator factories, the scratch arena and CG.

Own re-statement of indigo/backends/backend.py (the boundary named by the
north star).  A concrete backend subclasses `Backend`, provides
`dndarray._malloc/_free/_zero/_copy_from/_copy_to/_copy/__getitem__` and the
leaf kernels (`axpby, scale, dot, norm2, fftn, ifftn, ccsrmm, max`).  Method
names, argument order (outputs first: `fftn(y, x)`, `ccsrmm(y, ...)`,
`axpby(beta, y, alpha, x)`) and error behaviour follow the reference:

  dndarray                      backend.py:22-220
  scratch (LIFO bump arena)     backend.py:262-281
  factories Diag..NUFFT         backend.py:287-448
  leaf contract                 backend.py:453-533
  csr_matrix                    backend.py:535-596
  cg                            backend.py:639-689

Differences, all deliberate: structure analysis (`inspect`) comes from the
backend instead of the optional `_customcpu` module (whose absence breaks every
adjoint in the reference, backend.py:564-567 vs :585); `Zpad` indexes with a
tuple of slices (the reference's list indexing is an IndexError on numpy >=
1.23); `beta == 0` never reads `y` (BLAS rule; the numpy oracle multiplies
0 * y and so propagates NaNs from uninitialised memory).
"""
import logging
from contextlib import contextmanager

import numpy as np
import scipy.sparse as spp

import indigo_amd.operators as op

log = logging.getLogger(__name__)
_C64 = np.dtype('complex64')


class Backend(object):

    def __init__(self, device_id=0):
        self.trace = None          # attach an indigo_amd.util.Trace to record leaf calls

    # ---------------------------------------------------------------------------
    # device arrays
    # ---------------------------------------------------------------------------
    class dndarray(object):
        """N-d array in device memory, column-major, with a leading dimension.

        `shape[0]` elements are contiguous; column j of a 2-d view starts
        `_leading_dim * j` elements after column 0.  Views (`own=False`) never free.
        """
        _memory = dict()

        def __init__(self, backend, shape, dtype, ld=None, own=True, data=None, name=''):
            assert isinstance(shape, (tuple, list))
            self.shape = tuple(int(s) for s in shape)
            self.dtype = np.dtype(dtype)
            self._backend = backend
            self._leading_dim = int(ld) if ld else (self.shape[0] if self.shape else 1)
            self._own = own
            self._name = name
            if data is None:
                self._arr = self._malloc(self.shape, self.dtype)
                self._memory[id(self)] = (name, self.shape, self.dtype)
            else:
                self._arr = data

        # -- metadata ---------------------------------------------------------------
        @property
        def size(self):
            return int(np.prod(self.shape, dtype=np.int64))

        @property
        def itemsize(self):
            return self.dtype.itemsize

        @property
        def nbytes(self):
            return self.size * self.dtype.itemsize

        @property
        def ndim(self):
            return len(self.shape)

        @property
        def contiguous(self):
            return self.ndim == 1 or self._leading_dim == self.shape[0]

        def reshape(self, new_shape):
            """View with a new shape.  Mirrors the leading-dimension rules of backend.py:59-89."""
            new_shape = tuple(int(s) for s in new_shape)
            if -1 in new_shape:
                known = -int(np.prod(new_shape, dtype=np.int64))
                assert known > 0 and self.size % known == 0, \
                    "Cannot reshape {} into {}. (size mismatch)".format(self.shape, new_shape)
                new_shape = tuple(self.size // known if s == -1 else s for s in new_shape)
            assert int(np.prod(new_shape, dtype=np.int64)) == self.size, \
                "Cannot reshape {} into {}. (size mismatch)".format(self.shape, new_shape)
            if new_shape[0] > self.shape[0]:
                assert self.shape[0] == self._leading_dim, "Cannot stack non-contiguous columns."
            ld = new_shape[0] if new_shape[0] < self.shape[0] else self._leading_dim
            return self._view(new_shape, ld, self._arr)

        def _view(self, shape, ld, data):
            v = self._backend.dndarray(self._backend, shape, self.dtype, ld=ld, own=False, data=data)
            v._base = getattr(self, '_base', None) or self     # keep the owner alive
            return v

        # -- transfers ----------------------------------------------------------------
        def copy_from(self, arr):
            """host -> device into an existing array"""
            assert isinstance(arr, np.ndarray)
            if self.size != arr.size:
                raise ValueError("size mismatch, expected {} got {}".format(self.shape, arr.shape))
            if self.dtype != arr.dtype:
                raise TypeError("dtype mismatch, expected {} got {}".format(self.dtype, arr.dtype))
            if not arr.flags['F_CONTIGUOUS']:
                raise TypeError("order mismatch, expected 'F' got {}".format(arr.flags['F_CONTIGUOUS']))
            self._copy_from(arr)

        def copy_to(self, arr):
            """device -> host into an existing array"""
            assert isinstance(arr, np.ndarray)
            if self.size != arr.size:
                raise ValueError("size mismatch, expected {} got {}".format(self.shape, arr.shape))
            if self.dtype != arr.dtype:
                raise TypeError("dtype mismatch, expected {} got {}".format(self.dtype, arr.dtype))
            self._copy_to(arr)

        def to_host(self):
            arr = np.ndarray(self.shape, self.dtype, order='F')
            self.copy_to(arr)
            return arr

        @contextmanager
        def on_host(self):
            arr = self.to_host()
            yield arr
            self.copy_from(arr)

        def copy(self, other=None, name=''):
            """`a.copy()` returns a device copy; `a.copy(b)` copies b into a."""
            if other is not None:
                assert isinstance(other, self._backend.dndarray)
                self._copy(other)
                return None
            dup = self._backend.zero_array(self.shape, self.dtype, name=name)
            dup._copy(self)
            return dup

        @classmethod
        def to_device(cls, backend, arr, name=''):
            arr_f = np.require(arr, requirements='F')
            d_arr = cls(backend, arr.shape, arr.dtype, name=name)
            d_arr.copy_from(arr_f)
            return d_arr

        def __setitem__(self, slc, other):
            assert isinstance(slc, slice) and not (slc.start or slc.stop), "dndarray setitem cant slice"
            self._copy(other)

        def __del__(self):
            if getattr(self, '_own', False) and hasattr(self, '_arr'):
                self._memory.pop(id(self), None)
                try:
                    self._free()
                except Exception:       # interpreter shutdown: the library may be gone already
                    pass

        # -- to be provided by the concrete backend ---------------------------------
        def __getitem__(self, slc):
            raise NotImplementedError()

        def _copy_from(self, arr):
            raise NotImplementedError()

        def _copy_to(self, arr):
            raise NotImplementedError()

        def _copy(self, d_arr):
            raise NotImplementedError()

        def _malloc(self, shape, dtype):
            raise NotImplementedError()

        def _free(self):
            raise NotImplementedError()

        def _zero(self):
            raise NotImplementedError()

    def copy_array(self, arr, name=''):
        return self.dndarray.to_device(self, arr, name=name)

    def empty_array(self, shape, dtype, name=''):
        return self.dndarray(self, shape, dtype, name=name)

    def zero_array(self, shape, dtype, name=''):
        d_arr = self.empty_array(shape, dtype, name=name)
        d_arr._zero()
        return d_arr

    def zeros_like(self, other, name=''):
        return self.zero_array(other.shape, other.dtype, name=name)

    def rand_array(self, shape, dtype=_C64, name='', seed=None):
        rng = np.random.default_rng(seed)
        x = rng.random(shape) + 1j * rng.random(shape)
        x = np.require(x, dtype=_C64, requirements='F')
        return self.copy_array(x, name=name)

    def get_max_threads(self):
        return 1

    def barrier(self):
        pass

    def mem_usage(self):
        total = 0
        rows = []
        for name, shape, dtype in list(self.dndarray._memory.values()):
            n = int(np.prod(shape, dtype=np.int64)) * dtype.itemsize
            rows.append((n, name, shape, dtype))
            total += n
        log.info("Memory report:")
        for n, name, shape, dtype in sorted(rows, key=lambda r: r[0]):
            if n > 1e6:
                log.info("  %40s: % 3.0f MB, %20s, %15s", name, n / 1e6, shape, dtype)
        return total

    # ---------------------------------------------------------------------------
    # scratch arena
    # ---------------------------------------------------------------------------
    def reserve_scratch(self, nelems):
        """Reserve a complex64 arena of `nelems` elements for `scratch()` (LIFO)."""
        self._scratch = self.empty_array((max(int(nelems), 1),), _C64, name='scratch')
        self._scratch_pos = 0

    @contextmanager
    def scratch(self, shape=None, nbytes=None):
        assert not (shape is not None and nbytes is not None), \
            "Specify either shape or nbytes to backend.scratch()."
        if nbytes is not None:
            shape = (int(nbytes) // _C64.itemsize,)
        size = int(np.prod(shape, dtype=np.int64))
        arena = getattr(self, '_scratch', None)
        if arena is not None and self._scratch_pos + size > arena.size:
            # the arena was sized for another tree: serve this request dynamically instead of failing
            # (the reference asserts here, backend.py:272)
            log.warning("scratch arena too small (wanted %d elements at offset %d of %d); allocating dynamically",
                        size, self._scratch_pos, arena.size)
            arena = None
        if arena is not None:
            pos = self._scratch_pos
            mem = arena[pos:pos + size].reshape(tuple(shape))
            # keep successive carvings 256-byte aligned (32 complex64) like fresh allocations
            bump = (size + 31) // 32 * 32
            self._scratch_pos += bump
            try:
                yield mem
            finally:
                self._scratch_pos -= bump
        else:
            mem = self.zero_array(tuple(shape), dtype=_C64, name='scratch(dynamic)')
            yield mem
            del mem

    # ---------------------------------------------------------------------------
    # operator factories
    # ---------------------------------------------------------------------------
    def SpMatrix(self, M=None, **kwargs):
        assert (M is not None and spp.issparse(M)) or kwargs.get('struct') is not None
        return op.SpMatrix(self, M, **kwargs)

    def DenseMatrix(self, M, **kwargs):
        assert isinstance(M, np.ndarray) and M.ndim == 2
        return op.DenseMatrix(self, M, **kwargs)

    def Diag(self, v, **kwargs):
        """diag(v); v is flattened in memory ('A') order, i.e. F order for F arrays"""
        v = np.require(v, requirements='F')
        if v.ndim > 1:
            v = v.flatten(order='A')
        dtype = kwargs.pop('dtype', _C64)
        if np.dtype(dtype) == _C64:
            # (described as the diagonal it is: the scipy matrix is made when somebody asks for it, indigo_amd.structured)
            from indigo_amd.structured import DiagS
            return self.SpMatrix(struct=DiagS(v.size, [('vec', v.astype(_C64))]), **kwargs)
        return self.SpMatrix(spp.diags(v, offsets=0).astype(dtype), **kwargs)

    def Adjoint(self, A, **kwargs):
        return op.Adjoint(self, A, **kwargs)

    def KronI(self, c, B, **kwargs):
        """I_c (x) B"""
        return op.Kron(self, self.Eye(c), B, **kwargs)

    def Kron(self, A, B, **kwargs):
        return op.Kron(self, A, B, **kwargs)

    def BlockDiag(self, Ms, **kwargs):
        return op.BlockDiag(self, *Ms, **kwargs)

    def VStack(self, Ms, **kwargs):
        return op.VStack(self, *Ms, **kwargs)

    def HStack(self, Ms, **kwargs):
        return op.HStack(self, *Ms, **kwargs)

    def UnscaledFFT(self, shape, dtype=_C64, **kwargs):
        return op.UnscaledFFT(self, shape, dtype=dtype, **kwargs)

    def Eye(self, n, dtype=_C64, **kwargs):
        return op.Eye(self, n, dtype=dtype, **kwargs)

    def ZpadFFT(self, grid_shape, box_shape, weights, **kwargs):
        """fused  KronI(C, UnscaledFFT) * zero-pad * diag(weights)  leaf (see operators.ZpadFFT)"""
        return op.ZpadFFT(self, grid_shape, box_shape, weights, **kwargs)

    def One(self, shape, dtype=_C64, **kwargs):
        return op.One(self, shape, dtype=dtype, **kwargs)

    def FFT(self, shape, dtype=_C64, **kwargs):
        """unitary FFT = diag(1/sqrt(n)) * UnscaledFFT"""
        n = int(np.prod(shape))
        if np.dtype(dtype) == _C64:
            from indigo_amd.structured import DiagS
            c = (np.ones(1, dtype=dtype) / np.sqrt(n))[0]          # the value the reference's ones(n) / sqrt(n) holds n times
            return self.SpMatrix(struct=DiagS(n, [('const', c)]), name='scale') * self.UnscaledFFT(shape, dtype, **kwargs)
        s = np.ones(n, order='F', dtype=dtype) / np.sqrt(n)
        return self.Diag(s, name='scale') * self.UnscaledFFT(shape, dtype, **kwargs)

    @staticmethod
    def fftc_mod_phases(ft_shape):
        """per-axis terms (idx_d - c_d/2) c_d / n_d of the modulation's phase (in turns)"""
        return [(np.arange(n) - (n // 2) / 2.0) * ((n // 2) / n) for n in ft_shape]

    @staticmethod
    def fftc_mod(ft_shape, dtype=_C64):
        """Modulation vector of the centred FFT: exp(2 pi i sum_d (idx_d - c_d/2) c_d / n_d), c_d = n_d // 2.  (The phase as a
        broadcast sum of per-axis terms, added in the reference's order: the same numbers without the index grids.)"""
        phase = 0
        for i, ph in enumerate(Backend.fftc_mod_phases(ft_shape)):
            phase = phase + ph.reshape([-1 if j == i else 1 for j in range(len(ft_shape))])
        return np.exp(1j * 2.0 * np.pi * phase).astype(dtype)

    def FFTc(self, ft_shape, dtype=_C64, normalize=True, **kwargs):
        """centred (fftshift-ed) FFT as mod * F * mod"""
        if np.dtype(dtype) == _C64:
            from indigo_amd.structured import DiagS, SepPhase
            M = self.SpMatrix(struct=DiagS(int(np.prod(ft_shape)), [('sep', SepPhase(ft_shape, self.fftc_mod_phases(ft_shape)))]), name='mod')
        else:
            M = self.Diag(self.fftc_mod(ft_shape, dtype), name='mod')
        F = self.FFT(ft_shape, dtype=dtype, **kwargs) if normalize else self.UnscaledFFT(ft_shape, dtype=dtype, **kwargs)
        return M * F * M

    @staticmethod
    def zpad_rows(M, N, mode='center'):
        """Flat (F-order) indices in the padded volume M that receive the N-volume's samples."""
        if mode == 'center':
            slc = tuple(slice(m // 2 + int(np.ceil(-n / 2)), m // 2 + int(np.ceil(n / 2))) for m, n in zip(M, N))
        elif mode == 'edge':
            slc = tuple(slice(n) for n in N)
        else:
            raise ValueError("unknown zpad mode %r" % mode)
        x = np.arange(int(np.prod(M)), dtype=np.int64).reshape(M, order='F')
        return x[slc].flatten(order='F')

    def Zpad(self, M, N, mode='center', dtype=_C64, **kwargs):
        """zero-pad an N-volume into an M-volume; a 0/1 matrix of shape (prod M, prod N)"""
        rows = self.zpad_rows(M, N, mode)
        cols = np.arange(rows.size)
        if np.dtype(dtype) == _C64:
            from indigo_amd.structured import SelectS
            return self.SpMatrix(struct=SelectS((int(np.prod(M)), int(np.prod(N))), rows, cols, np.ones(rows.size, dtype=_C64)), **kwargs)
        mat = spp.coo_matrix((np.ones(rows.size), (rows, cols)), shape=(int(np.prod(M)), int(np.prod(N))), dtype=dtype)
        return self.SpMatrix(mat, **kwargs)

    def Crop(self, M, N, dtype=_C64, **kwargs):
        return self.Zpad(N, M, dtype=dtype, **kwargs).H

    def Interp(self, N, coord, width, table, dtype=_C64, **kwargs):
        """gridding / interpolation matrix (npts x prod N) from a k-space trajectory"""
        assert len(N) == 3
        ndim = coord.shape[0]
        npts = int(np.prod(coord.shape[1:]))
        coord = coord.reshape((ndim, -1), order='F')
        from indigo_amd.structured import InterpS
        op = self.SpMatrix(struct=InterpS(N, coord, width, table, npts, make_plain=lambda: self._interp_matrix(npts, N, width, table, coord, dtype)), **kwargs)
        op._grid_dims = tuple(int(v) for v in N)        # the columns are the points of this grid, first axis fastest (a hint: hip.py)
        return op

    def _interp_matrix(self, npts, N, width, table, coord, dtype):
        """the gridding matrix itself (scipy): the vectorised numpy formulation of the reference's loop; backends with a native
        builder override this"""
        from indigo_amd.interp import interp_mat
        return interp_mat(npts, N, width, table, coord).astype(dtype)

    def gridding_from_struct(self, s, grid_order=0):
        """CSR (complex64, sorted columns) of an InterpS description with its columns numbered in `grid_order` (0: (x, y, z),
        1: (x, z, y)), built directly -- or None: the caller materialises the scipy matrix and permutes it"""
        return None

    @staticmethod
    def nufft_params(width, oversamp):
        omin = min(oversamp) if isinstance(oversamp, tuple) else oversamp
        beta = np.pi * np.sqrt(((width * 2. / omin) * (omin - 0.5)) ** 2 - 0.8)
        return omin, beta

    def NUFFT(self, M, N, coord, width=3, n=128, oversamp=None, dtype=_C64, **kwargs):
        """non-uniform FFT  G * Fc * Z * R  (interp, centred FFT, zero-pad, roll-off)"""
        assert len(M) == 3 and len(N) == 3
        assert tuple(M[1:]) == tuple(coord.shape[1:])
        from scipy.signal.windows import kaiser
        from indigo_amd.noncart import rolloff3
        omin, beta = self.nufft_params(width, oversamp)
        osf = oversamp if isinstance(oversamp, tuple) else (omin,) * 3
        oN = tuple(int(N[i] * osf[i]) for i in range(3))

        Z = self.Zpad(oN, N, dtype=dtype, name='zpad')
        F = self.FFTc(oN, dtype=dtype, name='fft')
        kb = kaiser(2 * n + 1, beta)[n:]
        G = self.Interp(oN, coord, width, kb, dtype=np.float32, name='interp')
        R = self.Diag(rolloff3(omin, width, beta, N), name='apod')
        return G * F * Z * R

    # ---------------------------------------------------------------------------
    # leaf-kernel contract
    # ---------------------------------------------------------------------------
    def axpby(self, beta, y, alpha, x):
        """y = beta*y + alpha*x"""
        raise NotImplementedError()

    def dot(self, x, y):
        """Re(x^H y)"""
        raise NotImplementedError()

    def norm2(self, x):
        """||x||_2 ** 2"""
        raise NotImplementedError()

    def scale(self, x, alpha):
        """x *= alpha"""
        raise NotImplementedError()

    def pdot(self, x, y, comm):
        v = self.dot(x, y)
        return v if comm is None else comm.allreduce(v)

    def pnorm2(self, x, comm):
        v = self.norm2(x)
        return v if comm is None else comm.allreduce(v)

    def fftn(self, y, x):
        raise NotImplementedError()

    def ifftn(self, y, x):
        raise NotImplementedError()

    def _fft_workspace_size(self, x_shape):
        return 0

    # fused zero-pad/crop transforms used by operators.ZpadFFT (not part of the reference's contract:
    # they replace its Zpad-CSR + FFT composition; see include/indigo_hip.h ig_fft_exec_padded)
    def fft_padded(self, y, x, w, grid, box_lo, box_dims, workspace=None, layout=0, support=None):
        """y[:, c] = FFT(zeropad(w[:, c] * x)); y: (prod grid, C), x: (prod box, 1), w: (prod box, C).
        layout 1 stores each grid in (x, z, y) memory order instead of (x, y, z); layout 2 additionally interleaves
        the coils below x (memory index c + C*(kx + n0*kz + n0*n2*ky)) and expects w interleaved, w[i*C + c]."""
        raise NotImplementedError()

    def ifft_cropped(self, xc, y, w, grid, box_lo, box_dims, workspace, layout=0, support=None):
        """xc[:, c] = conj(w[:, c]) * crop(IFFT(y[:, c])); xc: (prod box, C), y: (prod grid, C) left intact
        (layout 2: y, w and xc are coil-interleaved in memory)"""
        raise NotImplementedError()

    def _fft_padded_workspace(self, grid, box_lo, box_dims, batch, layout=0):
        return 0

    def sum_columns(self, y, X, alpha=1, beta=0, interleaved=False):
        """y = beta*y + alpha * sum_j X[:, j]  (interleaved: X's memory holds element (i, j) at i*ncols + j)"""
        raise NotImplementedError()

    def supports_padded_fft(self, grid, ncoils=None):
        """whether `fft_padded` / `ifft_cropped[_sum]` exist for this oversampled grid (and, if given, this many coils)"""
        return False

    def ccsrmm(self, y, A_shape, A_indx, A_ptr, A_vals, x, alpha=1, beta=0, adjoint=False, exwrite=False):
        raise NotImplementedError()

    def cdiamm(self, y, shape, offsets, data, x, alpha=1.0, beta=0.0, adjoint=True):
        """y = beta*y + alpha * op(A) * x for a DIA-stored A (op = ^H when adjoint)"""
        raise NotImplementedError()

    def onemm(self, y, x, alpha=1, beta=0):
        """y = beta*y + alpha * ones * x"""
        raise NotImplementedError()

    def cgemm(self, y, M, x, alpha, beta, forward):
        """y = beta*y + alpha * op(M) * x for a dense M"""
        raise NotImplementedError()

    def csymm(self, y, M, x, alpha, beta, left=True):
        """dense product with a real symmetric M, from the left or from the right"""
        raise NotImplementedError()

    def max(self, val, arr):
        raise NotImplementedError()

    def inspect(self, csr):
        """(nonzero rows, nonzero cols, exwrite) of a scipy CSR matrix; exwrite = every column has <= 1 nonzero."""
        counts = np.bincount(csr.indices, minlength=csr.shape[1])
        nzrow = int(np.count_nonzero(np.diff(csr.indptr)))
        return nzrow, int(np.count_nonzero(counts)), bool(counts.max(initial=0) <= 1)

    # ---------------------------------------------------------------------------
    # device CSR matrix
    # ---------------------------------------------------------------------------
    class csr_matrix(object):
        _index_base = 0

        def __init__(self, backend, A, name='mat'):
            if not spp.isspmatrix_csr(A):
                A = A.tocsr()
            A = self._type_correct(A)
            assert A.nnz < 2 ** 31 and max(A.shape) < 2 ** 31, "int32 CSR indices"
            self._backend = backend
            self._name = name
            self.shape = tuple(int(s) for s in A.shape)
            self.dtype = A.dtype
            self.rowPtrs = backend.copy_array(A.indptr.astype(np.int32) + self._index_base, name=name + ".rowPtrs")
            self.colInds = backend.copy_array(A.indices.astype(np.int32) + self._index_base, name=name + ".colInds")
            self.values = backend.copy_array(A.data, name=name + ".data")
            nzrow, nzcol, self._exwrite = backend.inspect(A)
            self._row_frac = nzrow / A.shape[0] if A.shape[0] else 1.0
            self._col_frac = nzcol / A.shape[1] if A.shape[1] else 1.0

        @staticmethod
        def _check_panels(y, x, vals):
            assert x.dtype == _C64, "Bad dtype: expected complex64, got %s" % x.dtype
            assert y.dtype == _C64, "Bad dtype: expected complex64, got %s" % y.dtype
            assert vals.dtype == _C64

        def forward(self, y, x, alpha=1, beta=0):
            """y = alpha * A * x + beta * y"""
            self._check_panels(y, x, self.values)
            il = {'x_il': True} if getattr(self, '_grid_il', False) else {}
            self._backend.ccsrmm(y, self.shape, self.colInds, self.rowPtrs, self.values,
                                 x, alpha=alpha, beta=beta, adjoint=False, exwrite=True, **il)

        def adjoint(self, y, x, alpha=1, beta=0):
            """y = alpha * A^H * x + beta * y"""
            self._check_panels(y, x, self.values)
            il = {'y_il': True} if getattr(self, '_grid_il', False) else {}
            self._backend.ccsrmm(y, self.shape, self.colInds, self.rowPtrs, self.values,
                                 x, alpha=alpha, beta=beta, adjoint=True, exwrite=self._exwrite, **il)

        def set_grid_interleaved(self, flag=True):
            """The panel on the COLUMN side of this matrix (x of a forward product, y of an adjoint one) is stored
            row-major -- the values of one grid point for all panel columns (coils) contiguous -- instead of
            column-major.  This is the memory order of the fused transform's grid layout 2; the products are the
            same numbers in a different order."""
            self._grid_il = bool(flag)

        def set_grid_support(self, table, n0, nm, zw=16):
            """Hint: the columns of this matrix index a 3-D grid and only the tabulated support is ever
            non-zero / read in an adjoint product.  Backends may ignore it (the result inside the support
            is the same either way)."""
            pass

        def set_grid_bricks(self, n0, nm, ns, ncols=8, bm=2, bs=2, chunk=4096, run=4096):
            """Hint: the columns index an n0 x nm x ns grid (col = kx + n0*(km + nm*ks)); the adjoint product may bin the rows
            by grid bricks and scatter race-free.  Backends may ignore it."""
            pass

        def set_row_order(self, perm):
            """Hint: processing the rows in the order `perm` improves locality.  Backends may ignore it."""
            pass

        @property
        def nbytes(self):
            return self.rowPtrs.nbytes + self.colInds.nbytes + self.values.nbytes

        @property
        def nnz(self):
            return self.values.size

        def _type_correct(self, A):
            return A.astype(_C64)

    class dia_matrix(object):
        """Device-resident sparse matrix in diagonal (DIA) storage (reference backend.py:599-635): `data` is scipy's
        dia_matrix.data transposed -- one column per stored diagonal -- and `offsets` the diagonals' offsets."""

        def __init__(self, backend, A, name='mat'):
            assert isinstance(A, spp.dia_matrix)
            A = A.astype(_C64)
            self._backend = backend
            self.data = backend.copy_array(np.asfortranarray(A.data.T), name=name + ".data")
            self.offsets = backend.copy_array(A.offsets.astype(np.int32), name=name + ".offsets")
            self.shape = tuple(int(s) for s in A.shape)
            self.dtype = A.dtype
            self._row_frac = 1
            self._col_frac = 1
            self._exwrite = False

        def forward(self, y, x, alpha=1, beta=0):
            self._backend.cdiamm(y, self.shape, self.offsets, self.data, x, alpha=alpha, beta=beta, adjoint=False)

        def adjoint(self, y, x, alpha=1, beta=0):
            self._backend.cdiamm(y, self.shape, self.offsets, self.data, x, alpha=alpha, beta=beta, adjoint=True)

        @property
        def nbytes(self):
            return self.offsets.nbytes + self.data.nbytes

        @property
        def nnz(self):
            return self.data.size

    # ---------------------------------------------------------------------------
    # solvers
    # ---------------------------------------------------------------------------
    def cg(self, A, b_h, x_h, lamda=0.0, tol=1e-10, maxiter=100, team=None):
        """Conjugate gradients on (A + lamda I) x = b, host-scalar form: x_h is the start and receives the result; returns the
        relative residuals ||r_k|| / ||r_0||, one per iteration, stopping at the first below `tol`.

        The contract is the reference's (indigo/backends/backend.py:639-689: same start, same stopping rule, one operator
        evaluation per iteration, the two reductions through pdot / pnorm2 so that a `team` all-reduces them).  b_h / x_h may be
        device arrays of this backend: x is then updated in place and only the two scalars per iteration cross the host boundary.
        (HipBackend.cg overrides this with device-resident scalars and three fused vector passes per iteration.)"""
        in_place = isinstance(x_h, self.dndarray)
        x = x_h if in_place else self.copy_array(x_h, name='x')
        resid = b_h.copy(name='b') if isinstance(b_h, self.dndarray) else self.copy_array(b_h, name='b')
        q = x.copy()                                   # q = (A + lamda I) d for the current direction d

        def shifted(dst, src):
            A.eval(dst, src)
            if lamda:
                self.axpby(1, dst, lamda, src)

        shifted(q, x)
        self.axpby(1, resid, -1, q)                    # resid = b - (A + lamda I) x
        d = resid.copy(name='p')
        rho = rho0 = self.pnorm2(resid, team)
        trail = []
        for it in range(maxiter):
            shifted(q, d)
            step = rho / self.pdot(d, q, team)
            self.axpby(1, x, step, d)
            self.axpby(1, resid, -step, q)
            rho, rho_old = self.pnorm2(resid, team), rho
            self.axpby(rho / rho_old, d, 1, resid)     # d = resid + (rho / rho_old) d in one pass
            trail.append(float(np.sqrt(rho / rho0)))
            log.info("iter %d, residual %g", it, trail[-1])
            if trail[-1] < tol:
                log.info("cg reached tolerance")
                break
        else:
            log.info("cg reached maxiter")
        if not in_place:
            x.copy_to(x_h)
        return trail

    def apgd(self, gradf, proxg, alpha, x_h, maxiter=100, team=None):
        """Proximal gradient iteration for min f + g:  x <- prox_g(x - alpha grad f(y)),  y <- x + m (x - x_before).

        The reference (indigo/backends/backend.py:691-732) writes the accelerated (FISTA) momentum m = (t_k - 1) / t_{k+1} but never
        advances t_k from 1, so its m is 0 in every iteration and its iterates are those of the plain proximal gradient method --
        which the golden vectors captured from it pin (tests/golden/leaf_misc.npz).  `momentum` below is that sequence; the loop
        itself is written for any m.  The gradient step starts from x, not from y, as in the reference."""
        def momentum(k):
            t = 1.0                                    # (the reference's t_k: constant)
            return (t - 1.0) / (0.5 * (1.0 + np.sqrt(1.0 + 4.0 * t * t)))

        x = self.copy_array(x_h)
        y, x_before, g = x.copy(), x.copy(), x.copy()
        for k in range(maxiter):
            gradf(g, y)
            self.axpby(1, x, -alpha, g)
            proxg(x, alpha)
            m = momentum(k)
            self.axpby(0, y, 1 + m, x)
            if m:
                self.axpby(1, y, -m, x_before)
            x_before.copy(x)
        x.copy_to(x_h)

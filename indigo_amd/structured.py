"""Sparse factors of the SENSE tree as what they ARE -- a diagonal, a selection with weights, a gridding matrix -- so that
the reference's recipe (examples/pics.py:104-193: `MriRealize` folds `interp * mod * scale` and `(I_C (x) mod * zpad * apod) *
maps` into single CSR matrices, indigo/transforms.py:81-175 `RealizeMatrices`) is index arithmetic on a few vectors
instead of general sparse-sparse products of matrices with 1e8 rows.

The reference builds every factor as a scipy matrix (`Backend.Diag/Zpad/Interp`, indigo/backends/backend.py:298-401) and
multiplies them with scipy.  At BASELINE config 4 that is 35 s of host time in front of a 6.6 ms evaluation.  Here a
factory attaches one of the descriptions below to the `SpMatrix` it returns (the scipy matrix itself is made only when
somebody asks for `SpMatrix._matrix`); `RealizeMatrices` composes descriptions where it knows how and falls back to scipy
everywhere else.  A composed description materialises to the CSR the scipy route produces up to float32 rounding: the
same entries, each product rounded once like scipy's -- but numpy's and scipy's complex multiplies may contract their
multiply-adds differently, so single values can differ by one ulp (`tests/test_sense_cpu.py` holds the two routes against
each other at that tolerance).  SelectS / StackS also keep explicit zeros (maps that vanish outside the body) where scipy's
sparse products prune them: the structural nnz -- and what `_mem_usage` and the byte models derive from it -- can exceed the
scipy route's.

    DiagS     n x n diagonal: a product of factors, each a dense vector, a separable phase table or a constant
    SelectS   m x n with at most one entry per row: (rows, cols, vals) -- zero-pad, crop, a stack of diagonals, Kron(I, .)
    InterpS   the gridding matrix of a trajectory (Backend.Interp) times a diagonal on its column side
"""
import numpy as np
import scipy.sparse as spp

_C64 = np.dtype('complex64')


def _threads(n):
    import os
    return max(1, min(8, (os.cpu_count() or 2), n // (1 << 20)))


def _chunked(fn, n, out):
    """out[lo:hi] = fn(lo, hi) over a few threads (numpy releases the GIL in exp / multiply / take)"""
    nt = _threads(n)
    if nt <= 1:
        out[:] = fn(0, n)
        return out
    from concurrent.futures import ThreadPoolExecutor
    edges = [n * i // (4 * nt) for i in range(4 * nt + 1)]

    def one(i):
        lo, hi = edges[i], edges[i + 1]
        if hi > lo:
            out[lo:hi] = fn(lo, hi)
    with ThreadPoolExecutor(max_workers=nt) as ex:
        list(ex.map(one, range(4 * nt)))
    return out


class SepPhase(object):
    """v[i] = exp(2 pi i (p0[i0] + p1[i1] + p2[i2])) for the F-ordered index i = i0 + n0 (i1 + n1 i2): the modulation of a centred
    transform (Backend.fftc_mod).  The phase is summed in the order and precision of the full-grid formula, so values taken
    at single points equal the dense vector's bit for bit."""

    def __init__(self, shape, phases):
        self.shape = tuple(int(n) for n in shape)
        self.phases = [np.ascontiguousarray(ph, dtype=np.float64) for ph in phases]
        assert len(self.shape) == len(self.phases) and all(ph.size == n for ph, n in zip(self.phases, self.shape))

    @property
    def size(self):
        return int(np.prod(self.shape, dtype=np.int64))

    def take(self, idx):
        idx = np.asarray(idx)
        out = np.empty(idx.shape, dtype=_C64)

        def part(lo, hi):
            rest = idx[lo:hi].astype(np.int64)
            phase = 0
            for n, ph in zip(self.shape, self.phases):
                phase = phase + ph[rest % n]
                rest = rest // n
            return np.exp(1j * 2.0 * np.pi * phase)
        return _chunked(part, idx.size, out)

    def dense(self):
        return self.take(np.arange(self.size, dtype=np.int64))


class DiagS(object):
    """diag(f_1 * f_2 * ...): factors are ('vec', array of n), ('sep', SepPhase) or ('const', scalar), multiplied left to right in
    complex64 -- the roundings of the scipy products diag(f_1) @ diag(f_2) @ ..."""

    def __init__(self, n, factors):
        self.n = int(n)
        self.factors = list(factors)

    shape = property(lambda self: (self.n, self.n))
    nnz = property(lambda self: self.n)

    def take(self, idx):
        """the diagonal at the positions idx (complex64)"""
        idx = np.asarray(idx)
        out = None
        for kind, f in self.factors:
            if kind == 'vec':
                v = np.asarray(f).reshape(-1)[idx].astype(_C64)
            elif kind == 'sep':
                v = f.take(idx)
            else:
                v = _C64.type(f)
            out = v if out is None else (out * v).astype(_C64)
        if out is None:
            out = _C64.type(1)
        return np.broadcast_to(out, idx.shape) if np.ndim(out) == 0 else out

    def dense(self):
        if len(self.factors) == 1 and self.factors[0][0] == 'vec':
            return np.asarray(self.factors[0][1]).reshape(-1).astype(_C64)
        return np.ascontiguousarray(self.take(np.arange(self.n, dtype=np.int64)))

    def mul(self, other):
        assert isinstance(other, DiagS) and other.n == self.n
        return DiagS(self.n, self.factors + other.factors)

    def adjoint(self):
        return DiagS(self.n, [('vec', np.conj(self.dense()))]) if any(k != 'const' or np.imag(f) for k, f in self.factors) else self

    def scaled(self, c):
        return DiagS(self.n, self.factors + [('const', c)])

    def to_scipy(self):
        return spp.diags(self.dense(), offsets=0).astype(_C64)

    def separable(self):
        """(phases, constant) when the diagonal is exp(2 pi i sum of per-axis phases) times a REAL constant -- what the native
        builder ig_interp3_fill_modulated folds into the gridding weights -- else None"""
        sep = [f for k, f in self.factors if k == 'sep']
        const = [f for k, f in self.factors if k == 'const']
        if len(sep) != 1 or len(sep) + len(const) != len(self.factors) or any(np.imag(c) != 0 for c in const):
            return None
        c = 1.0
        for v in const:
            c *= float(np.real(v))
        return sep[0], c


class SelectS(object):
    """m x n matrix given by its entries M[rows[j], cols[j]] = vals[j] (no coordinate twice), with the knowledge whether a row /
    a column holds at most one of them: zero-pad and crop matrices (both), diagonals stacked on top of each other (one per row),
    I_C (x) such a matrix, their adjoints and products."""

    def __init__(self, shape, rows, cols, vals, rows_unique=True, cols_unique=True):
        self.shape = (int(shape[0]), int(shape[1]))
        self.rows = np.asarray(rows, dtype=np.int64)
        self.cols = np.asarray(cols, dtype=np.int64)
        self.vals = np.asarray(vals, dtype=_C64)
        self.rows_unique, self.cols_unique = bool(rows_unique), bool(cols_unique)
        assert self.rows.shape == self.cols.shape == self.vals.shape

    nnz = property(lambda self: int(self.rows.size))

    def _like(self, vals):
        return SelectS(self.shape, self.rows, self.cols, vals, self.rows_unique, self.cols_unique)

    @classmethod
    def from_diag(cls, d):
        idx = np.arange(d.n, dtype=np.int64)
        return cls((d.n, d.n), idx, idx, d.dense())          # (rows IS cols: recognised as a diagonal by compose_product)

    def left_diag(self, d):          # diag(d) @ self
        return self._like((d.take(self.rows) * self.vals).astype(_C64))

    def right_diag(self, d):         # self @ diag(d)
        return self._like((self.vals * d.take(self.cols)).astype(_C64))

    def matmul(self, other):
        """self @ other where every row of `other` holds at most one entry: entry (r, c, v) of self becomes (r, col_other(c),
        v * val_other(c)).  No coordinate comes out twice if the rows of self hold one entry each or the columns of other do; any
        other pair is declined (None: the caller multiplies with scipy)."""
        assert self.shape[1] == other.shape[0]
        if not other.rows_unique or not (self.rows_unique or other.cols_unique):
            return None
        where = np.full(other.shape[0], -1, dtype=np.int64)
        where[other.rows] = np.arange(other.rows.size)
        j = where[self.cols]
        keep = j >= 0
        j = j[keep]
        return SelectS((self.shape[0], other.shape[1]), self.rows[keep], other.cols[j], (self.vals[keep] * other.vals[j]).astype(_C64),
                       self.rows_unique, self.cols_unique and other.cols_unique)

    def kron_eye(self, c):
        """I_c (x) self"""
        m, n = self.shape
        off = np.arange(c, dtype=np.int64)
        return SelectS((c * m, c * n), (self.rows[None, :] + off[:, None] * m).reshape(-1), (self.cols[None, :] + off[:, None] * n).reshape(-1),
                       np.broadcast_to(self.vals, (c, self.vals.size)).reshape(-1), self.rows_unique, self.cols_unique)

    def adjoint(self):
        return SelectS(self.shape[::-1], self.cols, self.rows, np.conj(self.vals), self.cols_unique, self.rows_unique)

    def scaled(self, c):
        return self._like((self.vals * _C64.type(c)).astype(_C64))

    def to_scipy(self):
        order = np.lexsort((self.cols, self.rows)) if not self.rows_unique else np.argsort(self.rows, kind='stable')
        counts = np.bincount(self.rows, minlength=self.shape[0])
        indptr = np.zeros(self.shape[0] + 1, dtype=np.int64)
        np.cumsum(counts, out=indptr[1:])
        big = max(self.shape) >= 2 ** 31 or self.rows.size >= 2 ** 31
        it = np.int64 if big else np.int32
        return spp.csr_matrix((self.vals[order], self.cols[order].astype(it), indptr.astype(it)), shape=self.shape)


class KronS(object):
    """I_c (x) M for a selection M, kept as the pair: c identical diagonal blocks"""

    def __init__(self, c, inner):
        self.c, self.inner = int(c), inner
        assert isinstance(inner, SelectS)

    shape = property(lambda self: (self.c * self.inner.shape[0], self.c * self.inner.shape[1]))
    nnz = property(lambda self: self.c * self.inner.nnz)

    def adjoint(self):
        return KronS(self.c, self.inner.adjoint())

    def scaled(self, v):
        return KronS(self.c, self.inner.scaled(v))

    def to_scipy(self):
        return self.inner.kron_eye(self.c).to_scipy()


class StackS(object):
    """[B_0; B_1; ...]: selections of one width stacked on top of each other -- VStack(Diag(map_c)), and what I_C (x) M makes of
    it: the blocks M * diag(map_c), which share M's pattern (the SAME index arrays) and differ in their values only"""

    def __init__(self, blocks):
        self.blocks = [b if isinstance(b, SelectS) else SelectS.from_diag(b) for b in blocks]
        assert len({b.shape for b in self.blocks}) == 1

    shape = property(lambda self: (len(self.blocks) * self.blocks[0].shape[0], self.blocks[0].shape[1]))
    nnz = property(lambda self: sum(b.nnz for b in self.blocks))

    def shared_pattern(self):
        b0 = self.blocks[0]
        return all((b.rows is b0.rows or np.array_equal(b.rows, b0.rows)) and (b.cols is b0.cols or np.array_equal(b.cols, b0.cols)) for b in self.blocks[1:])

    def adjoint(self):
        return AdjointS(self)

    def scaled(self, v):
        return StackS([b.scaled(v) for b in self.blocks])

    def to_scipy(self):
        m = self.blocks[0].shape[0]
        rows = np.concatenate([b.rows + c * m for c, b in enumerate(self.blocks)])
        return SelectS(self.shape, rows, np.concatenate([b.cols for b in self.blocks]), np.concatenate([b.vals for b in self.blocks]),
                       all(b.rows_unique for b in self.blocks), False).to_scipy()


class AdjointS(object):
    """the conjugate transpose of a StackS (how `MriGoodAdjoints` stores S', examples/pics.py:166-177)"""

    def __init__(self, inner):
        self.inner = inner

    shape = property(lambda self: self.inner.shape[::-1])
    nnz = property(lambda self: self.inner.nnz)

    def adjoint(self):
        return self.inner

    def scaled(self, v):
        return AdjointS(self.inner.scaled(np.conj(v)))

    def to_scipy(self):
        st = self.inner
        b0 = st.blocks[0]
        C, (m, n) = len(st.blocks), b0.shape
        if st.shared_pattern() and b0.cols_unique and b0.rows_unique:
            # row i of the transpose holds one entry per block, at column c m + row_of(i), in block order: the CSR directly
            order = np.argsort(b0.cols, kind='stable')
            present = b0.cols[order]
            counts = np.zeros(n, dtype=np.int64)
            counts[present] = C
            indptr = np.zeros(n + 1, dtype=np.int64)
            np.cumsum(counts, out=indptr[1:])
            it = np.int64 if C * m >= 2 ** 31 else np.int32
            indices = (b0.rows[order][:, None] + (np.arange(C, dtype=np.int64) * m)[None, :]).astype(it)
            data = np.empty((present.size, C), dtype=_C64)
            for c, b in enumerate(st.blocks):
                data[:, c] = np.conj(b.vals[order])
            return spp.csr_matrix((data.reshape(-1), indices.reshape(-1), indptr.astype(it)), shape=(n, C * m))
        return st.to_scipy().conjugate().transpose().tocsr()


def vstack_diags(diags):
    """[diag(d_0); diag(d_1); ...] ((C n) x n)"""
    return StackS(diags)


class InterpS(object):
    """Backend.Interp(N, coord, width, table) (T x prod N, real weights: indigo/interp.py:18-80) times diag(colscale) on the right"""

    def __init__(self, N, coord, width, table, npts, colscale=None, make_plain=None):
        self.N = tuple(int(n) for n in N)
        self.coord, self.width, self.table, self.npts = coord, width, table, int(npts)
        self.colscale = colscale
        self.make_plain = make_plain          # () -> scipy matrix of the plain gridding matrix (the backend's builder)

    shape = property(lambda self: (self.npts, int(np.prod(self.N, dtype=np.int64))))

    @property
    def nnz(self):
        return self.plain().nnz

    def plain(self):
        if getattr(self, '_plain', None) is None:
            self._plain = self.make_plain().tocsr()
        return self._plain

    def right_diag(self, d):
        s = InterpS(self.N, self.coord, self.width, self.table, self.npts, d if self.colscale is None else self.colscale.mul(d), self.make_plain)
        s._plain = getattr(self, '_plain', None)
        return s

    def to_scipy(self):
        G = self.plain()
        if self.colscale is None:
            return G
        # the scipy route casts the float32 weights to complex64 and multiplies by one diagonal after the other, rounding each time
        data = G.data.astype(_C64)
        for kind, f in self.colscale.factors:
            data = (data * DiagS(self.colscale.n, [(kind, f)]).take(G.indices)).astype(_C64)
        return spp.csr_matrix((data, G.indices, G.indptr), shape=G.shape)


def compose_product(sl, sr):
    """description of L @ R from the descriptions of L and R, or None where no rule applies (the caller multiplies with scipy)"""
    if sl is None or sr is None:
        return None
    if isinstance(sl, DiagS) and isinstance(sr, DiagS):
        return sl.mul(sr)
    if isinstance(sl, DiagS) and isinstance(sr, SelectS):
        return sr.left_diag(sl)
    if isinstance(sl, SelectS) and isinstance(sr, DiagS):
        return sl.right_diag(sr)
    if isinstance(sl, SelectS) and isinstance(sr, SelectS):
        return sl.matmul(sr)
    if isinstance(sl, InterpS) and isinstance(sr, DiagS):
        return sl.right_diag(sr)
    if isinstance(sl, KronS) and isinstance(sr, KronS) and sl.c == sr.c:
        inner = sl.inner.matmul(sr.inner)
        return KronS(sl.c, inner) if inner is not None else None
    if isinstance(sl, KronS) and isinstance(sr, StackS) and sl.c == len(sr.blocks) and sl.inner.shape[1] == sr.blocks[0].shape[0]:
        # (I_C (x) M) [B_0; B_1; ...] = [M B_0; M B_1; ...]: C products of selections; with diagonal B_c they share M's pattern
        blocks = []
        for b in sr.blocks:
            diag = b.rows_unique and b.cols_unique and b.shape[0] == b.shape[1] and b.rows is b.cols
            blocks.append(sl.inner._like((sl.inner.vals * _take(b.vals, sl.inner.cols)).astype(_C64)) if diag else sl.inner.matmul(b))
        return StackS(blocks) if all(b is not None for b in blocks) else None
    return None


def _take(vals, idx):
    out = np.empty(idx.shape, dtype=vals.dtype)
    return _chunked(lambda lo, hi: vals[idx[lo:hi]], idx.size, out)

"""Host-side construction of the 3-D gridding (interpolation) matrix.

Same arithmetic as the reference's numba loop (indigo/interp.py:8-80), written
as vectorised numpy (numba is not a dependency here):

  pos_d   = N_d * coord[d, i] + N_d // 2
  taps    = range(ceil(pos_d - width), floor(pos_d + width))     # end exclusive
  weight  = prod_d lerp(table, |tap_d - pos_d| / width)           # 0 beyond the table
  column  = (x % N0) + N0 * ((y % N1) + N1 * (z % N2))            # wrap-around

Rows are built in chunks so that a 5e7-nonzero matrix stays within a few GB of
host memory.  The result is a float64 COO/CSR matrix of shape (npts, prod N);
the caller casts it (the NUFFT factory stores float32 weights as complex64).
"""
import numpy as np
import scipy.sparse as spp

__all__ = ['lin_interp', 'interp_mat', 'interp_csr_arrays', 'interp_csr_arrays_numpy', 'interp_sep_records', 'sep_expand']


def lin_interp(table, x):
    """Linear interpolation into `table` at x in [0, 1); zero for x >= 1.  Vectorised over x."""
    table = np.asarray(table, dtype=np.float64)
    x = np.asarray(x, dtype=np.float64)
    n = table.shape[0]
    inside = x < 1
    xs = np.where(inside, x, 0.0) * (n - 1)
    idx = xs.astype(np.int64)
    frac = xs - idx
    hi = np.minimum(idx + 1, n - 1)
    val = (1.0 - frac) * table[idx] + frac * table[hi]
    return np.where(inside, val, 0.0)


def _axis_taps(N_d, c_d, width, table):
    """Per-point tap indices (unwrapped), weights and validity mask along one axis."""
    pos = N_d * c_d + (N_d // 2)
    start = np.ceil(pos - width).astype(np.int64)
    end = np.floor(pos + width).astype(np.int64)
    ntap = np.maximum(end - start, 0)
    tmax = int(ntap.max(initial=0))
    t = np.arange(tmax, dtype=np.int64)[None, :]
    k = start[:, None] + t
    valid = t < ntap[:, None]
    w = lin_interp(table, np.abs(k - pos[:, None]) / width)
    return k, w, valid


def _chunk_triplets(N, width, table, coord, lo, hi):
    kx, wx, vx = _axis_taps(N[0], coord[0, lo:hi], width, table)
    ky, wy, vy = _axis_taps(N[1], coord[1, lo:hi], width, table)
    kz, wz, vz = _axis_taps(N[2], coord[2, lo:hi], width, table)
    # nonzeros ordered z (slowest), y, x (fastest), as in the reference's loop nest
    jz = np.mod(kz, N[2]) * (N[1] * N[0])
    jy = np.mod(ky, N[1]) * N[0]
    jx = np.mod(kx, N[0])
    col = jz[:, :, None, None] + jy[:, None, :, None] + jx[:, None, None, :]
    ker = (wz[:, :, None, None] * wy[:, None, :, None]) * wx[:, None, None, :]
    valid = vz[:, :, None, None] & vy[:, None, :, None] & vx[:, None, None, :]
    n = hi - lo
    row = np.broadcast_to(np.arange(lo, hi, dtype=np.int64)[:, None, None, None], col.shape)
    return row.reshape(n, -1), col.reshape(n, -1), ker.reshape(n, -1), valid.reshape(n, -1)


def interp_mat(m, N, width, table, coord, chunk=65536):
    """COO interpolation matrix, shape (m, prod N), float64 weights."""
    assert coord.shape[0] == 3, "only 3-D trajectories are supported (as in the reference)"
    N = tuple(int(n) for n in N)
    coord = np.asarray(coord, dtype=np.float64).reshape(3, -1)
    rows, cols, kers = [], [], []
    for lo in range(0, m, chunk):
        hi = min(lo + chunk, m)
        row, col, ker, valid = _chunk_triplets(N, width, table, coord, lo, hi)
        rows.append(row[valid])
        cols.append(col[valid])
        kers.append(ker[valid])
    if rows:
        row, col, ker = np.concatenate(rows), np.concatenate(cols), np.concatenate(kers)
    else:
        row = col = np.zeros(0, dtype=np.int64)
        ker = np.zeros(0)
    return spp.coo_matrix((ker, (row, col)), shape=(m, int(np.prod(N, dtype=np.int64))))


def interp_csr_arrays(m, N, width, table, coord, dtype=np.float32, grid_order=0):
    """CSR arrays (indptr, indices, data) of the gridding matrix with sorted columns, built by the library's native
    host routine (ig_interp3_count / ig_interp3_fill: the counterpart of the reference's numba loop).  `grid_order=1`
    numbers the grid columns (x, z, y) instead of (x, y, z).  Bit-identical to `interp_csr_arrays_numpy`."""
    import ctypes
    from indigo_amd import _lib
    L = _lib.lib()
    N = tuple(int(n) for n in N)
    coord = np.ascontiguousarray(np.asarray(coord, dtype=np.float64).reshape(3, -1))
    assert coord.shape[1] == m
    table = np.ascontiguousarray(table, dtype=np.float64)
    dims = (ctypes.c_int64 * 3)(*N)
    indptr = np.empty(m + 1, dtype=np.int32)
    _lib.check(L.ig_interp3_count(m, dims, float(width), coord.ctypes.data, indptr.ctypes.data), None, "ig_interp3_count")
    nnz = int(indptr[-1])
    indices = np.empty(nnz, dtype=np.int32)
    data = np.empty(nnz, dtype=np.float32)
    _lib.check(L.ig_interp3_fill(m, dims, float(width), table.ctypes.data, table.size, coord.ctypes.data, indptr.ctypes.data,
                                 indices.ctypes.data, data.ctypes.data, int(grid_order)), None, "ig_interp3_fill")
    return indptr.astype(np.int64), indices, data if dtype == np.float32 else data.astype(dtype)


def interp_csr_modulated(m, N, width, table, coord, phases, scale, grid_order=0):
    """CSR arrays (indptr, indices, complex64 data) of  interp * diag(exp(2 pi i (px[kx] + py[ky] + pz[kz]))) * scale  -- the
    gridding matrix times the centred transform's modulation and normalisation (G' of the -O3 SENSE tree) -- in one native
    pass (ig_interp3_fill_modulated); `phases` = per-axis phase tables in turns (sense._mod_axis_phases)."""
    import ctypes
    from indigo_amd import _lib
    L = _lib.lib()
    N = tuple(int(n) for n in N)
    coord = np.ascontiguousarray(np.asarray(coord, dtype=np.float64).reshape(3, -1))
    assert coord.shape[1] == m
    table = np.ascontiguousarray(table, dtype=np.float64)
    px, py, pz = (np.ascontiguousarray(ph, dtype=np.float64) for ph in phases)
    assert (px.size, py.size, pz.size) == N
    dims = (ctypes.c_int64 * 3)(*N)
    indptr = np.empty(m + 1, dtype=np.int32)
    _lib.check(L.ig_interp3_count(m, dims, float(width), coord.ctypes.data, indptr.ctypes.data), None, "ig_interp3_count")
    nnz = int(indptr[-1])
    indices = np.empty(nnz, dtype=np.int32)
    data = np.empty(nnz, dtype=np.complex64)
    _lib.check(L.ig_interp3_fill_modulated(m, dims, float(width), table.ctypes.data, table.size, coord.ctypes.data, indptr.ctypes.data,
                                           indices.ctypes.data, data.ctypes.data, int(grid_order), px.ctypes.data, py.ctypes.data,
                                           pz.ctypes.data, float(scale)), None, "ig_interp3_fill_modulated")
    return indptr, indices, data


def _axis_signs(ph):
    """exp(2 pi i ph[k]) = g * s[k] with s[k] = +-1 and |g| = 1 -- the modulation of a centred transform on an even axis -- or None"""
    m = np.exp(2j * np.pi * np.asarray(ph, dtype=np.float64))
    g = m[0]
    r = m / g
    s = np.round(r.real)
    if np.abs(r.imag).max() > 1e-9 or np.abs(np.abs(s) - 1.0).max() > 0 or np.abs(r.real - s).max() > 1e-9:
        return None
    return complex(g), s


def interp_sep_records(m, N, width, table, coord, phases=None, scale=1.0, grid_order=0):
    """The SEPARABLE form of  interp * diag(exp(2 pi i (px[kx] + py[ky] + pz[kz]))) * scale  (indigo/interp.py:18-60 builds every
    tap as w = wz * wy * wx; examples/pics.py:104-177 folds the centred transform's modulation and normalisation in): one
    record per sample -- first tap and tap count per axis, per-axis float32 weights with the modulation's SIGN folded in
    (ig_interp3_sep, include/indigo_hip.h).  Returns dict(records (m, words) uint32, tw, gconst, dims = grid axes in memory order)
    with  G' = gconst * (the matrix the records describe), or None when the modulation is no sign per axis (an odd axis: its
    phases are genuinely complex) or the kernel is wider than 8 taps per axis."""
    import ctypes
    from indigo_amd import _lib
    L = _lib.lib()
    N = tuple(int(n) for n in N)
    coord = np.ascontiguousarray(np.asarray(coord, dtype=np.float64).reshape(3, -1))
    assert coord.shape[1] == m
    table = np.ascontiguousarray(table, dtype=np.float64)
    gconst, signs = 1.0 + 0.0j, [None, None, None]
    if phases is not None:
        for d in range(3):
            gs = _axis_signs(phases[d])
            if gs is None:
                return None
            gconst *= gs[0]
            signs[d] = np.ascontiguousarray(gs[1], dtype=np.float64)
            assert signs[d].size == N[d]
    tw = 4 if 2 * width <= 4 else 6 if 2 * width <= 6 else 8 if 2 * width <= 8 else None
    if tw is None or min(N) < tw or max(N) > 65535:
        return None
    words = L.ig_interp3_sep_words(tw)
    rec = np.empty((m, words), dtype=np.uint32)
    dims = (ctypes.c_int64 * 3)(*N)
    rc = L.ig_interp3_sep(m, dims, float(width), table.ctypes.data, table.size, coord.ctypes.data, int(grid_order),
                          *[sg.ctypes.data if sg is not None else None for sg in signs], float(scale), tw, rec.ctypes.data)
    if rc != 0:
        return None
    mem = (N[0], N[1], N[2]) if grid_order == 0 else (N[0], N[2], N[1])
    if abs(gconst.imag) < 1e-12:
        gconst = complex(round(gconst.real), 0.0) if abs(abs(gconst.real) - 1.0) < 1e-12 else complex(gconst.real, 0.0)
    return dict(records=rec, tw=tw, gconst=gconst, dims=mem, grid_order=int(grid_order))


def sep_expand(sep, lo=0, hi=None):
    """(rows, cols, vals) of the taps the records [lo, hi) describe -- columns numbered in the records' memory order, values in the
    float32 arithmetic of the kernels ((w1 * w2) * w0, then times gconst): the dense form of the separable matrix, for tests"""
    rec, tw = sep['records'], sep['tw']
    hi = rec.shape[0] if hi is None else hi
    r = rec[lo:hi]
    n0, nm, ns = sep['dims']
    w = r[:, :3 * tw].view(np.float32).reshape(-1, 3, tw)
    h0, h1 = r[:, 3 * tw], r[:, 3 * tw + 1]
    j = np.stack([h0 & 0xffff, h0 >> 16, h1 & 0xffff], axis=1).astype(np.int64)
    cnt = np.stack([(h1 >> 16) & 15, (h1 >> 20) & 15, (h1 >> 24) & 15], axis=1).astype(np.int64)
    a = np.arange(tw)
    k0 = (j[:, 0, None] + a) % n0
    k1 = (j[:, 1, None] + a) % nm
    k2 = (j[:, 2, None] + a) % ns
    col = k0[:, None, None, :] + n0 * (k1[:, None, :, None] + nm * k2[:, :, None, None])
    val = ((w[:, 1][:, None, :, None] * w[:, 2][:, :, None, None]).astype(np.float32) * w[:, 0][:, None, None, :]).astype(np.float32)
    ok = (a < cnt[:, 0, None])[:, None, None, :] & (a < cnt[:, 1, None])[:, None, :, None] & (a < cnt[:, 2, None])[:, :, None, None]
    row = np.broadcast_to(np.arange(lo, hi)[:, None, None, None], col.shape)
    return row[ok], col[ok], (val[ok].astype(np.complex64) * np.complex64(sep['gconst']))


def interp_csr_arrays_numpy(m, N, width, table, coord, dtype=np.float32, chunk=65536):
    """The same arrays from vectorised numpy (kept as the independent cross-check of the native routine).

    Equivalent to `interp_mat(...).tocsr()` + `sort_indices()` whenever no row
    wraps onto the same column twice (true when every N_d exceeds the tap count);
    that condition is checked.  Avoids the COO->CSR conversion of ~5e7 triplets.
    """
    from concurrent.futures import ThreadPoolExecutor
    import os
    N = tuple(int(n) for n in N)
    coord = np.asarray(coord, dtype=np.float64).reshape(3, -1)
    indptr = np.zeros(m + 1, dtype=np.int64)
    big = np.iinfo(np.int64).max

    def one_chunk(lo):
        hi = min(lo + chunk, m)
        _, col, ker, valid = _chunk_triplets(N, width, table, coord, lo, hi)
        key = np.where(valid, col, big)
        order = np.argsort(key, axis=1, kind='stable')
        key = np.take_along_axis(key, order, axis=1)
        ker = np.take_along_axis(ker, order, axis=1)
        cnt = valid.sum(axis=1)
        dup = (key[:, 1:] == key[:, :-1]) & (key[:, 1:] != big)
        if dup.any():
            raise ValueError("interp_csr_arrays_numpy: a row wraps onto one column twice; use interp_mat")
        keep = np.arange(key.shape[1])[None, :] < cnt[:, None]
        return lo, hi, cnt, key[keep].astype(np.int32), ker[keep].astype(dtype)

    # numpy releases the GIL in the heavy steps, so a few threads cut the wall time of big trajectories
    workers = max(1, min(8, (os.cpu_count() or 2) // 2))
    idx_parts, val_parts = [], []
    with ThreadPoolExecutor(max_workers=workers) as ex:
        for lo, hi, cnt, idx, val in ex.map(one_chunk, range(0, m, chunk)):
            indptr[lo + 1:hi + 1] = cnt
            idx_parts.append(idx)
            val_parts.append(val)
    np.cumsum(indptr, out=indptr)
    indices = np.concatenate(idx_parts) if idx_parts else np.zeros(0, dtype=np.int32)
    data = np.concatenate(val_parts) if val_parts else np.zeros(0, dtype=dtype)
    return indptr, indices, data

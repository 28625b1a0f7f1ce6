"""The C-ABI library: loads, exports every symbol include/indigo_hip.h declares,
refuses to run without a GPU, and its HOST-side helpers (inspect, transpose)
agree with scipy.  No device compute here (CPU-only suite).
"""
import ctypes
import os
import re

import numpy as np
import pytest
import scipy.sparse as spp

from conftest import ROOT
from indigo_amd import _lib

HEADER = os.path.join(ROOT, "include", "indigo_hip.h")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ig_[a-z0-9_]+)\s*\(", text)))


def test_library_loads_and_exports_every_declared_symbol():
    L = _lib.lib()
    names = declared_symbols()
    assert len(names) >= 30
    for name in names:
        assert hasattr(L, name), "library does not export %s" % name
    # and the ctypes prototype table covers exactly the header
    assert sorted(_lib.PROTOTYPES) == names
    assert L.ig_abi_version() == 1


def test_no_gpu_means_loud_failure():
    if _lib.device_count() > 0:
        pytest.skip("a GPU is present")
    from indigo_amd.backends import available_backends, get_backend
    with pytest.raises(RuntimeError, match="no HIP device|no CPU fallback"):
        get_backend("hip")
    assert available_backends() == []
    with pytest.raises(ValueError):
        get_backend("numpy")          # the oracle is not a product backend


def test_host_inspect_matches_definition():
    L = _lib.lib()
    rng = np.random.default_rng(5)
    for trial in range(6):
        M, K = int(rng.integers(1, 60)), int(rng.integers(1, 60))
        A = spp.random(M, K, density=float(rng.choice([0.01, 0.1, 0.5])), format='csr', random_state=rng)
        if trial == 0:       # exwrite structure: <= 1 nonzero per column
            cols = rng.permutation(K)[:min(M, K)]
            A = spp.csr_matrix((np.ones(cols.size), (np.arange(cols.size) % M, cols)), shape=(M, K))
        indptr, indices = A.indptr.astype(np.int32), A.indices.astype(np.int32)
        nzr, nzc, exw = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int()
        rc = L.ig_csr_inspect(indptr.ctypes.data, indices.ctypes.data, M, K, ctypes.byref(nzr), ctypes.byref(nzc), ctypes.byref(exw))
        assert rc == 0
        counts = np.bincount(A.indices, minlength=K)
        assert nzr.value == np.count_nonzero(np.diff(A.indptr))
        assert nzc.value == np.count_nonzero(counts)
        assert bool(exw.value) == bool(counts.max(initial=0) <= 1)


def test_host_inspect_rejects_bad_indices():
    L = _lib.lib()
    indptr = np.array([0, 1], dtype=np.int32)
    indices = np.array([7], dtype=np.int32)
    nzr, nzc, exw = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int()
    rc = L.ig_csr_inspect(indptr.ctypes.data, indices.ctypes.data, 1, 3, ctypes.byref(nzr), ctypes.byref(nzc), ctypes.byref(exw))
    assert rc == 2 and "out of range" in _lib.last_error()


def test_host_transpose_matches_scipy():
    L = _lib.lib()
    rng = np.random.default_rng(6)
    for M, K, d in [(1, 1, 1.0), (17, 33, 0.2), (64, 8, 0.5), (5, 40, 0.0)]:
        A = spp.random(M, K, density=d, format='csr', random_state=rng, dtype=np.float64)
        A = (A + 1j * A).astype(np.complex64).tocsr()
        A.sort_indices()
        indptr, indices = A.indptr.astype(np.int32), A.indices.astype(np.int32)
        pt = np.empty(K + 1, np.int32)
        it = np.empty(A.nnz, np.int32)
        dt = np.empty(A.nnz, np.complex64)
        rc = L.ig_csr_transpose(M, K, A.nnz, indptr.ctypes.data, indices.ctypes.data, A.data.ctypes.data,
                                pt.ctypes.data, it.ctypes.data, dt.ctypes.data)
        assert rc == 0
        T = A.T.tocsr()
        T.sort_indices()
        np.testing.assert_array_equal(pt, T.indptr)
        np.testing.assert_array_equal(it, T.indices)
        np.testing.assert_array_equal(dt, T.data)


def test_host_grid_brick_binning_matches_definition():
    """ig_grid_bricks_count / _fill: the nonzeros sorted (stably) by the brick of 16 x bm x bs grid cells of their column, as
    12-byte entries {cell inside the brick, re, im}; the entries of one row in one brick padded to a multiple of `unit`, and
    the row of every group of `unit` entries in round_rows"""
    L = _lib.lib()
    rng = np.random.default_rng(11)
    n0, nm, ns = 32, 16, 32
    for bm, bs, unit in ((4, 4, 8), (8, 8, 16), (2, 16, 8), (16, 4, 1), (2, 2, 4)):       # (2, 2, 4): the quads of the wide adjoint
        M, P = 300, n0 * nm * ns
        A = spp.random(M, P, density=0.002, format='csr', random_state=rng).astype(np.complex64)
        A.data = (A.data.real + 1j * rng.random(A.nnz)).astype(np.complex64)
        A.sort_indices()
        indptr, indices, data = A.indptr.astype(np.int32), A.indices.astype(np.int32), np.ascontiguousarray(A.data)
        nbx, nbm, nbs = n0 // 16, nm // bm, ns // bs
        counts = np.full(nbx * nbm * nbs, -7, dtype=np.int32)
        assert L.ig_grid_bricks_count(M, indptr.ctypes.data, indices.ctypes.data, n0, nm, ns, bm, bs, unit, counts.ctypes.data) == 0
        kx, km, ks = A.indices % n0, (A.indices // n0) % nm, A.indices // (n0 * nm)
        brick = kx // 16 + nbx * (km // bm + nbm * (ks // bs))
        cell = kx % 16 + 16 * (km % bm + bm * (ks % bs))
        rows = np.repeat(np.arange(M), np.diff(A.indptr))
        # expected stream per brick: rows ascending; a row's entries in CSR order, then padding up to a multiple of unit
        exp = {}
        for t in range(M):
            sel = np.flatnonzero(rows == t)
            for bb in dict.fromkeys(brick[sel].tolist()):
                mine = sel[brick[sel] == bb]
                lst = exp.setdefault(bb, [])
                lst += [(t, int(cell[i]), complex(A.data[i])) for i in mine]
                lst += [(t, 0xffffffff, 0j)] * (-len(mine) % unit)
        np.testing.assert_array_equal(counts, [len(exp.get(bb, [])) for bb in range(counts.size)])
        ptr = np.zeros(counts.size + 1, dtype=np.int64)
        np.cumsum(counts, out=ptr[1:])
        entries = np.zeros((int(ptr[-1]), 3), dtype=np.uint32)
        round_rows = np.full(int(ptr[-1]) // unit, 0xdeadbeef, dtype=np.uint32)
        assert L.ig_grid_bricks_fill(M, indptr.ctypes.data, indices.ctypes.data, data.ctypes.data, n0, nm, ns, bm, bs, unit,
                                     ptr.ctypes.data, entries.ctypes.data, round_rows.ctypes.data) == 0
        for bb, lst in exp.items():
            got = entries[ptr[bb]:ptr[bb + 1]]
            np.testing.assert_array_equal(np.repeat(round_rows[ptr[bb] // unit:ptr[bb + 1] // unit], unit), [e[0] for e in lst])
            np.testing.assert_array_equal(got[:, 0], [e[1] for e in lst])
            np.testing.assert_array_equal(got[:, 1:].copy().view(np.complex64)[:, 0], np.array([e[2] for e in lst], dtype=np.complex64))
    # grids that do not divide into bricks are refused
    assert L.ig_grid_bricks_count(1, indptr.ctypes.data, indices.ctypes.data, 30, 16, 32, 4, 4, 8, counts.ctypes.data) != 0
    assert L.ig_grid_bricks_count(1, indptr.ctypes.data, indices.ctypes.data, 32, 16, 24, 4, 16, 8, counts.ctypes.data) != 0


def test_brick_task_list_covers_every_entry_once():
    """brick_tasks (host side of ig_ccsrmm_t_bricks): runs of consecutive non-empty bricks and pieces of heavy bricks
    partition the entry stream; a run is exactly the entries of its table rows and respects the brick cap"""
    from indigo_amd.backends.hip import brick_tasks
    rng = np.random.default_rng(5)
    for chunk, run, cap in ((4096, 1024, 64), (64, 8, 64), (128, 1 << 20, 16), (8, 256, 3)):
        counts = (rng.integers(0, 40, size=5000) * 8 * (rng.random(5000) < 0.6)).astype(np.int32)
        counts[rng.integers(0, 5000, size=20)] = rng.integers(50, 3000, size=20) * 8
        ptr = np.zeros(counts.size + 1, dtype=np.int64)
        np.cumsum(counts, out=ptr[1:])
        tasks, table, shared = brick_tasks(counts, ptr, chunk, run, max_bricks=cap)
        np.testing.assert_array_equal(table[:, 0], np.flatnonzero(counts))
        np.testing.assert_array_equal(table[:, 1], ptr[1:][counts > 0])
        np.testing.assert_array_equal(shared, np.flatnonzero(counts > chunk))
        covered = np.zeros(int(ptr[-1]), dtype=np.int32)
        assert np.all(np.diff(tasks[:, 1] - tasks[:, 0]) <= 0)                     # longest first
        for lo, hi, bt, nbf in tasks:
            covered[lo:hi] += 1
            nb, sh = nbf & 0xffff, nbf >> 16
            assert hi > lo and lo % 8 == 0 and hi % 8 == 0
            if sh:
                b = table[bt, 0]
                assert nb == 1 and counts[b] > chunk and ptr[b] <= lo and hi <= ptr[b + 1] and hi - lo <= chunk
            else:
                assert 1 <= nb <= cap
                bricks = table[bt:bt + nb, 0]
                assert np.all(counts[bricks] <= chunk)
                assert lo == ptr[bricks[0]] and hi == table[bt + nb - 1, 1] == ptr[bricks[-1] + 1]
                assert counts[bricks[0]:bricks[-1] + 1].sum() == hi - lo              # nothing but empty bricks in between
        assert np.all(covered == 1)
        ordered, table2, _ = brick_tasks(counts, ptr, chunk, run, max_bricks=cap, longest_first=False)
        assert np.all(np.diff(ordered[:, 0]) > 0) and sorted(map(tuple, ordered)) == sorted(map(tuple, tasks))     # same tasks, entry order
    t, tb, sh = brick_tasks(np.zeros(10, np.int32), np.zeros(11, np.int64), 64, 64)
    assert t.shape == (0, 4) and tb.shape == (0, 2) and sh.size == 0

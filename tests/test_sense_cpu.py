"""Host logic of the SENSE path against the reference's golden vectors (CPU only).

The factories (Interp / rolloff / FFTc modulation / Zpad / NUFFT), the tree
rewrites (`sense_recipe`), the direct fused builder and CG run here on the
numpy oracle backend; what is compared is OUR host-side construction against
what the reference built from the same seeded inputs (tests/golden/sense.npz).
"""
import numpy as np
import pytest

from conftest import csr_from, golden, rel_err
from indigo_amd.backends.backend import Backend
from indigo_amd.interp import interp_csr_arrays, interp_mat
from indigo_amd.noncart import rolloff3
from indigo_amd.sense import SenseProblem, normal_operator
from indigo_amd import operators as op

C64 = np.dtype('complex64')


@pytest.fixture(scope="module")
def prob():
    g = golden("sense")
    C, width, ntab, osf, ro, tr = g["params"]
    p = SenseProblem(tuple(int(n) for n in g["N"]), g["coord"], np.asfortranarray(g["maps"]),
                     width=int(width), ntable=int(ntab), oversamp=float(osf))
    assert p.C == int(C) and p.T == int(ro * tr)
    return p, g


def test_kernel_table_and_rolloff(prob):
    p, g = prob
    assert abs(p.beta - float(g["beta"])) < 1e-12
    np.testing.assert_allclose(p.table, g["kb"], rtol=1e-12)
    np.testing.assert_allclose(rolloff3(p.oversamp, p.width, p.beta, p.N), g["rolloff"], rtol=1e-12)
    assert p.oN == tuple(int(n) for n in g["oN"])


def test_fftc_modulation_and_zpad(prob):
    p, g = prob
    np.testing.assert_allclose(Backend.fftc_mod(p.oN, C64), g["fftc_mod"], rtol=1e-6)
    np.testing.assert_array_equal(Backend.zpad_rows(p.oN, p.N), g["zpad_rows"])


def test_interp_matrix_matches_reference(prob):
    p, g = prob
    G_ref = csr_from(g, "interp_")
    G = interp_mat(p.T, p.oN, p.width, p.table, p.coord.reshape(3, -1, order='F')).astype(np.float32).astype(C64).tocsr()
    G.sort_indices()
    np.testing.assert_array_equal(G.indptr, G_ref.indptr)
    np.testing.assert_array_equal(G.indices, G_ref.indices)
    np.testing.assert_allclose(G.data, G_ref.data, rtol=1e-6)
    # fast CSR path == COO path
    indptr, indices, w = interp_csr_arrays(p.T, p.oN, p.width, p.table, p.coord.reshape(3, -1, order='F'))
    np.testing.assert_array_equal(indptr, G_ref.indptr)
    np.testing.assert_array_equal(indices, G_ref.indices)
    np.testing.assert_allclose(w, G_ref.data.real, rtol=1e-6)


def test_nufft_apply(prob, oracle_backend):
    p, g = prob
    B = oracle_backend
    F1 = B.NUFFT(p.ksp_dims, p.N, p.coord, width=p.width, n=p.ntable, oversamp=p.oversamp, dtype=C64)
    y_d = B.zero_array((p.T, 2), C64)
    F1.eval(y_d, B.copy_array(g["nufft_x"]))
    assert rel_err(y_d.to_host(), g["nufft_fwd"]) < 1e-5
    x_d = B.zero_array(g["nufft_x"].shape, C64)
    F1.H.eval(x_d, B.copy_array(g["nufft_k"]))
    assert rel_err(x_d.to_host(), g["nufft_adj"]) < 1e-5


@pytest.mark.parametrize("level", [0, 1, 2, 3, "fused", "zpadfft", "zpadfft-xyz", "zpadfft-il", "zpadfft-chunked"])
def test_sense_forward_adjoint_normal(prob, oracle_backend, level):
    p, g = prob
    B = oracle_backend
    if hasattr(B, '_scratch'):
        B._scratch = None
    if level == "fused":
        A = p.build_fused(B)
    elif level == "zpadfft-chunked":
        # more coils than the chunk: VStack of coil-interleaved chunks sharing one gridding matrix (+ a left-over single coil)
        A = p.build_zpadfft(B, chunk=2)
        assert isinstance(A, op.VStack) and len(A.children) == (p.C + 1) // 2
        assert len({id(c.left.right) for c in A.children if getattr(c.left.right, '_grid_interleaved', False)}) <= 1
    elif level in ("zpadfft", "zpadfft-xyz", "zpadfft-il"):
        A = p.build_zpadfft(B, layout={"zpadfft": 1, "zpadfft-xyz": 0, "zpadfft-il": 2}[level])
    else:
        A = p.build_tree(B, level=level)
    x, k = g["sense_x"], g["sense_k"]
    assert rel_err(A * x, g["sense_Ax"]) < 1e-5
    assert rel_err(A.H * k, g["sense_AHk"]) < 1e-5
    if level in (3, "fused", "zpadfft", "zpadfft-xyz", "zpadfft-il", "zpadfft-chunked"):
        assert rel_err(A * x, g["sense_O3_Ax"]) < 1e-5
        assert rel_err(A.H * k, g["sense_O3_AHk"]) < 1e-5
    AHA = normal_operator(A, lamda=float(g["lamda"]))
    y_d = B.zero_array((A.shape[1], 1), C64)
    AHA.eval(y_d, B.copy_array(x))
    assert rel_err(y_d.to_host(), g["sense_AHAx"]) < 1e-5
    B._scratch = None


def test_O3_tree_shape_and_leaves(prob, oracle_backend):
    """our recipe produces the reference's -O3 tree: same node types, same fused matrices"""
    p, g = prob
    A = p.build_tree(oracle_backend, level=3)
    ref_types = [line.split(", ")[1] for line in str(g["sense_O3_dump"]).strip().split("\n")]
    our_types = [line.split(", ")[1] for line in A.dump().strip().split("\n")]
    assert our_types == ref_types

    def leaves(node, acc):
        if isinstance(node, op.SpMatrix):
            acc.append(node)
        for c in getattr(node, '_children', []):
            leaves(c, acc)
        return acc
    ours = leaves(A, [])
    assert len(ours) == int(g["O3_nleaves"])
    fused = [p.fused_interp(), p.fused_maps_T()]
    for j, leaf in enumerate(ours):
        ref = csr_from(g, "O3_leaf%d_" % j)
        for name, M in (("recipe", leaf._matrix.astype(C64).tocsr()), ("fused", fused[j].tocsr())):
            M.sort_indices()
            assert M.shape == ref.shape
            np.testing.assert_array_equal(M.indptr, ref.indptr, err_msg=name)
            np.testing.assert_array_equal(M.indices, ref.indices, err_msg=name)
            np.testing.assert_allclose(M.data, ref.data, rtol=2e-6, atol=1e-9, err_msg=name)


def test_cg_iterates(prob, oracle_backend):
    p, g = prob
    B = oracle_backend
    B._scratch = None
    A = p.build_tree(B, level=0)
    AHA = A.H * A + float(g["lamda"]) * B.Eye(A.shape[1])
    for it in (1, 2, 3):
        x0 = np.zeros((A.shape[1], 1), dtype=C64, order='F')
        B.cg(AHA, g["cg_b"].copy(order='F'), x0, maxiter=it)
        assert rel_err(x0, g["cg_it%d" % it]) < 1e-4


def test_coil_shards_sum_to_full_adjoint(prob, oracle_backend):
    """partial adjoint images of coil shards add up to A^H k (the single all-reduce of AHA)"""
    p, g = prob
    B = oracle_backend
    B._scratch = None
    k = g["sense_k"].reshape(p.T, p.C, order='F')
    total = 0
    for coils in ([0], [1, 2]):
        A = p.build_fused(B, coils=coils)
        ks = np.asfortranarray(k[:, coils]).reshape(-1, 1, order='F')
        total = total + A.H * ks
    assert rel_err(total, g["sense_AHk"]) < 1e-5


def test_coil_chunks_and_lazy_maps(oracle_backend):
    assert SenseProblem.coil_chunks(32, 8) == [(0, 8), (8, 16), (16, 24), (24, 32)]
    assert SenseProblem.coil_chunks(13, 8) == [(0, 8), (8, 12), (12, 13)]
    assert SenseProblem.coil_chunks(6, 4) == [(0, 4), (4, 6)]
    # per-coil generated maps (what a rank of a sharded run materialises) give the same operator as the stacked array
    lazy = SenseProblem.synthetic((12, 10, 8), 5, nspokes=7, nreadout=24, oversamp=1.5, seed=11, lazy_maps=True)
    full = SenseProblem(lazy.N, lazy.coord, np.stack([lazy.coil_map(c) for c in range(5)], axis=3), oversamp=1.5)
    B = oracle_backend
    B._scratch = None
    x = np.asfortranarray(lazy.coil_map(0).reshape(-1, 1, order='F'))
    A1, A2 = lazy.build_fused(B, coils=[1, 3]), full.build_fused(B, coils=[1, 3])
    np.testing.assert_array_equal(A1 * x, A2 * x)
    A3 = lazy.build_zpadfft(B, chunk=2)
    A4 = full.build_fused(B)
    assert rel_err(A3 * x, A4 * x) < 1e-5
    k = (A4 * x).astype(C64)
    assert rel_err(A3.H * k, A4.H * k) < 1e-5
    B._scratch = None


def test_fuse_zpadfft_transform_on_the_reference_recipe(prob, oracle_backend):
    """reference route: factories -> pics.py -O3 recipe -> FuseZpadFFT reaches the fused leaf and evaluates to the
    golden -O3 vectors; a tree without the S' structure is left alone"""
    from indigo_amd.transforms import FuseZpadFFT, sense_recipe
    p, g = prob
    B = oracle_backend
    B._scratch = None
    A = p.build_tree(B, level=0)
    for Step in sense_recipe(3) + [FuseZpadFFT]:
        A = Step().visit(A)
    assert A.has(op.ZpadFFT) and not A.has(op.UnscaledFFT)
    leaves = [line.split(", ")[1] for line in A.dump().strip().split("\n")]
    assert leaves.count("SpMatrix") == 1 and leaves.count("ZpadFFT") == 1
    x, k = g["sense_x"], g["sense_k"]
    assert rel_err(A * x, g["sense_O3_Ax"]) < 1e-5
    assert rel_err(A.H * k, g["sense_O3_AHk"]) < 1e-5
    AHA = normal_operator(A, lamda=float(g["lamda"]))
    y_d = B.zero_array((A.shape[1], 1), C64)
    AHA.eval(y_d, B.copy_array(x))
    assert rel_err(y_d.to_host(), g["sense_AHAx"]) < 1e-5
    B._scratch = None
    # the same leaves as the direct builder
    Ad = p.build_zpadfft(B)
    # (three coils: one 4-wide interleaved chunk whose last coil has zero weights, its k-space rows cut off by HeadRows)
    assert isinstance(A, op.HeadRows) and isinstance(Ad, op.HeadRows) and A.shape == Ad.shape == (3 * p.T, int(np.prod(p.N)))
    A, Ad = A.child, Ad.child
    assert type(Ad.right).__name__ == type(A.right).__name__ == "ZpadFFT" and A.right._C == 4
    assert not np.any(A.right._weights().to_host().reshape(-1, 4)[:, 3])
    assert Ad.right._lo == A.right._lo and Ad.right._box == A.right._box and Ad.right._layout == A.right._layout
    np.testing.assert_allclose(A.right._weights().to_host(), Ad.right._weights().to_host(), rtol=2e-6, atol=1e-9)
    Ga, Gd = A.left.right._matrix.tocsr(), Ad.left.right._matrix.tocsr()
    np.testing.assert_array_equal(Ga.indices, Gd.indices)
    np.testing.assert_allclose(Ga.data, Gd.data, rtol=2e-6, atol=1e-12)
    # no match: a plain product of two sparse matrices stays as it is
    import scipy.sparse as spp
    M = B.SpMatrix(spp.eye(6, dtype=C64).tocsr()) * B.SpMatrix(spp.eye(6, dtype=C64).tocsr())
    assert FuseZpadFFT().visit(M) is M
    B._scratch = None


def test_fuse_zpadfft_with_masked_maps(prob, oracle_backend):
    """coil maps that vanish outside the body (masked ESPIRiT maps): the recipe's scipy products drop the exact zeros, so
    S' has fewer than C entries in most rows and none in many -- including whole boundary planes of the image box.
    FuseZpadFFT must still recognise the factor (missing entry = weight 0) and evaluate to the unfused -O3 tree."""
    from indigo_amd import fused
    from indigo_amd.transforms import FuseZpadFFT, sense_recipe
    p, g = prob
    B = oracle_backend
    B._scratch = None
    maps = np.asfortranarray(g["maps"]).copy(order='F')
    n0, n1, n2, C = maps.shape
    xx, yy, zz = np.meshgrid(np.arange(n0), np.arange(n1), np.arange(n2), indexing='ij')
    ball = ((xx - n0 / 2) ** 2 + (yy - n1 / 2) ** 2 + (zz - n2 / 2) ** 2) < (0.36 * min(n0, n1, n2)) ** 2
    maps[~ball] = 0                                   # outside the body: every coil; boundary planes vanish entirely
    maps[:, : n1 // 2, :, 1] = 0                      # coil 1 sees half of it
    q = SenseProblem(p.N, p.coord, maps, width=p.width, ntable=p.ntable, oversamp=p.oversamp)
    def scipy_route():          # the recipe on plain scipy matrices, as the reference runs it (no factor knows what it is)
        from indigo_amd.transforms import sense_recipe
        A = q.build_tree(B, level=0)
        _strip_descriptions(A)
        for Step in sense_recipe(3):
            A = Step().visit(A)
        return A
    A3 = scipy_route()
    St = [n for n in _leaves(A3) if isinstance(n, op.SpMatrix)][-1]._matrix
    assert St.nnz < int(np.prod(p.N)) * C              # entries really are missing
    Af = FuseZpadFFT().visit(scipy_route())
    # ... and the realisation that composes the factors' descriptions keeps the zeros as zero weights: the same leaf
    Ag = FuseZpadFFT().visit(q.build_tree(B, level=3))
    xx, kk = g["sense_x"], g["sense_k"]
    assert Ag.has(op.ZpadFFT) and rel_err(Ag * xx, Af * xx) < 2e-6 and rel_err(Ag.H * kk, Af.H * kk) < 2e-6
    assert Af.has(op.ZpadFFT) and not Af.has(op.UnscaledFFT)
    core = Af.child if isinstance(Af, op.HeadRows) else Af      # (C = 3: a 4-wide chunk with a zero-weight coil)
    assert core.right._box == p.N
    x, k = g["sense_x"], g["sense_k"]
    assert rel_err(Af * x, A3 * x) < 2e-6
    assert rel_err(Af.H * k, A3.H * k) < 2e-6
    # not a zero-pad * diagonal factor: one entry moved to another grid point
    P = int(np.prod(p.oN))
    Ss = (St if St.shape[1] == C * P else St.conjugate().transpose()).tocsr().astype(C64)
    Ss.sort_indices()
    assert fused.decode_zpad_maps(Ss, C, P, p.oN) is not None
    r = int(np.flatnonzero(np.diff(Ss.indptr) > 1)[0])
    bad = Ss.copy()
    bad.indices[bad.indptr[r]] -= 1
    assert fused.decode_zpad_maps(bad, C, P, p.oN) is None
    B._scratch = None


def _leaves(node):
    kids = getattr(node, '_children', None)
    if not kids:
        return [node]
    return [l for c in kids for l in _leaves(c)]


def test_support_tables_of_every_granularity_cover_the_touched_cells():
    """fused.grid_support(tile): every grid cell a nonzero of G' touches lies in a flagged segment; the flagged set of a finer
    table is a subset of the coarser one's (what a gather route writes by the 16-point table is a superset of what a reader
    with the 8-point table reads); hulls contain the flagged rows; the finer the table, the fewer grid points it flags"""
    from indigo_amd import fused
    p = SenseProblem.synthetic((16, 16, 16), 2, nspokes=40, nreadout=32, width=2, oversamp=2.0, seed=3)      # grid 32^3
    G = p.fused_interp(1)
    n0, n1, n2 = p.oN
    cols = np.unique(G.indices)
    prev, prev_count = None, None
    for tile in (2, 4, 8, 16):
        table = fused.grid_support(G, p.oN, tile)
        np.testing.assert_array_equal(table, fused.grid_support_numpy(G, p.oN, tile))     # native builder == numpy formulation
        zr, yr, bits = fused.split_support(table, p.oN, tile)
        nt = n0 // tile
        m = np.arange(n2 // 16, dtype=np.uint32)
        seg = ((bits.reshape(n1, nt, 16)[:, :, :, None] >> m) & 1).astype(bool)      # (ky, tile, t, m): kz = t + 16 m
        seg = seg.transpose(0, 3, 2, 1).reshape(n1, n2, nt)                              # (ky, kz, tile)
        rows = np.repeat(seg, tile, axis=2).reshape(-1)                                  # kx + n0*(kz + n2*ky)
        assert rows[cols].all()
        if prev is not None:
            assert not (prev & ~rows).any() and prev_count <= rows.sum()                 # finer subset of coarser
        prev, prev_count = rows, rows.sum()
        ky, kz, tt = np.nonzero(seg)
        assert np.all(zr[ky * nt + tt, 0] <= kz) and np.all(kz < zr[ky * nt + tt, 1])  # z hull per (ky, tile)
        assert np.all(yr[tt, 0] <= ky) and np.all(ky < yr[tt, 1])                      # y hull per tile


def test_fused_weights_interleaved_is_the_same_array_in_another_memory_order():
    """SenseProblem.fused_weights(interleaved=True): same values and shape, a voxel's coils adjacent in memory -- what the
    coil-interleaved leaf uploads without a transposition (lazy and stored maps, threaded and unthreaded sizes)"""
    from indigo_amd.sense import SenseProblem
    for N, C, lazy in (((24, 20, 16), 3, True), ((112, 96, 100), 4, False)):
        p = SenseProblem.synthetic(N, C, nspokes=8, nreadout=16, lazy_maps=lazy)
        a = p.fused_weights()
        b = p.fused_weights(interleaved=True)
        assert a.shape == b.shape == N + (C,) and np.array_equal(a, b)
        assert a.flags['F_CONTIGUOUS'] and b.reshape((-1, C), order='F').flags['C_CONTIGUOUS']
        one = p.fused_weights([1], interleaved=True)
        assert np.array_equal(one[..., 0], a[..., 1])


@pytest.mark.parametrize("oN,tile,zw", [((32, 24, 48), 16, (16, 16)), ((32, 30, 40), 8, (8, 5)), ((64, 20, 100), 16, (10, 10)), ((48, 18, 70), 4, (10, 7))])
def test_support_table_builder_native_vs_numpy(oN, tile, zw):
    """ig_grid_support (threaded host routine) against the numpy formulation, for bitmaps of 16 words per entry (256- and
    512-point z axes) and of B / A words per entry (an axis the A x B kernel transforms): same table bit for bit, and the
    bitmaps' two forms flag the same (ky, kz, kx tile) segments"""
    import scipy.sparse as spp
    from indigo_amd import fused
    n0, n1, n2 = oN
    rng = np.random.default_rng(n2)
    P = n0 * n1 * n2
    cols = np.unique(rng.integers(0, P, size=P // 7))
    cols = cols[((cols // n0) % n2 > n2 // 5) | (cols % 3 == 0)]
    G = spp.csr_matrix((np.ones(cols.size, np.complex64), cols, np.array([0, cols.size])), shape=(1, P))
    a = fused.grid_support(G, oN, tile, zw)
    b = fused.grid_support_numpy(G, oN, tile, zw)
    np.testing.assert_array_equal(a, b)
    nt = n0 // tile
    zr, yr, bits_in = fused.split_support(a, oN, tile, zw[0])
    kz = np.arange(n2)
    seg_in = (bits_in.reshape(n1, nt, zw[0])[:, :, kz % zw[0]] >> (kz // zw[0]).astype(np.uint32)) & 1
    if zw[0] != zw[1]:
        off = 2 * (n1 * nt + nt) + 2 * n1 * nt * zw[0]
        bits_out = np.ascontiguousarray(a[off:]).view(np.uint32).reshape(n1, nt, zw[1])
        seg_out = (bits_out[:, :, kz % zw[1]] >> (kz // zw[1]).astype(np.uint32)) & 1
        np.testing.assert_array_equal(seg_in, seg_out)
    # every touched column lies in a flagged segment and inside the hulls
    kx, kzc, kyc = cols % n0, (cols // n0) % n2, cols // (n0 * n2)
    assert seg_in[kyc, kx // tile, kzc].all()
    z = zr.reshape(n1, nt, 2)
    assert (z[kyc, kx // tile, 0] <= kzc).all() and (kzc < z[kyc, kx // tile, 1]).all()
    assert (yr[kx // tile, 0] <= kyc).all() and (kyc < yr[kx // tile, 1]).all()


def test_modulated_gridding_matrix_of_an_even_grid_is_real_up_to_rounding_residue():
    """what HipBackend's real-weight formats rest on (indigo_amd/backends/hip.py:weights_are_real): on an even grid the centred
    transform's modulation is exp(i pi k) = +-1, so G' = G * mod * scale has real weights and its imaginary parts are the rounding
    residue of sin(pi k) -- orders of magnitude below the threshold; on an odd grid (the reference driver's 277-point axis) the
    weights are genuinely complex"""
    from indigo_amd.backends.hip import weights_are_real
    from indigo_amd.sense import SenseProblem
    even = SenseProblem.synthetic((32, 32, 32), 2, nspokes=60, nreadout=64, width=2, oversamp=2.0, seed=4).fused_interp(2)
    assert weights_are_real(even.data)
    assert np.abs(even.data.imag).max() < 1e-9 * np.abs(even.data.real).max()
    odd = SenseProblem.synthetic((30, 26, 22), 2, nspokes=60, nreadout=60, width=2, oversamp=1.25, seed=4)
    assert any(n % 2 for n in odd.oN)
    assert not weights_are_real(odd.fused_interp(2).data)
    assert weights_are_real(np.array([2.0, -1.0], dtype=np.complex64)) and not weights_are_real(np.zeros(0, np.complex64))
    assert not weights_are_real(np.array([1 + 1e-6j], dtype=np.complex64))


def test_cg_splits_a_tikhonov_term_off_the_operator(oracle_backend):
    """HipBackend.cg moves a real multiple of Eye at the root of the tree (examples/pics.py:195 writes the regularisation into the
    operator) into its own lamda; anything else is left alone"""
    import scipy.sparse as spp
    from indigo_amd.backends.hip import HipBackend
    B = oracle_backend
    M = B.SpMatrix(spp.identity(6, dtype=np.complex64, format='csr'))
    for tree in (M + 0.25 * B.Eye(6), 0.25 * B.Eye(6) + M):
        rest, lam = HipBackend._split_identity(tree, 0.5)
        assert rest is M and abs(lam - 0.75) < 1e-12
    for tree in (M, M + (0.25 + 1j) * B.Eye(6), M + M):
        rest, lam = HipBackend._split_identity(tree, 0.5)
        assert rest is tree and lam == 0.5


def even_grid_problem():
    """the problem of tests/golden/sense_even.npz, rebuilt from its seeds (image 64^3, 8 coils, grid 128^3, radial, width 2)"""
    from indigo_amd.sense import radial_trajectory
    from indigo_amd.util import rand64c
    g = golden("sense_even")
    C, width, ntab, osf, ro, nsp = g["params"]
    s_coord, s_maps, s_x, s_k, _ = (int(v) for v in g["seeds"])
    N = tuple(int(n) for n in g["N"])
    p = SenseProblem(N, radial_trajectory(int(nsp), int(ro), seed=s_coord), np.asfortranarray(rand64c(*N, int(C), seed=s_maps)),
                     width=int(width), ntable=int(ntab), oversamp=float(osf))
    assert p.oN == tuple(int(n) for n in g["oN"]) and p.T == int(ro * nsp)
    x = rand64c(int(np.prod(N)), 1, seed=s_x)
    k = rand64c(p.T * int(C), 1, seed=s_k)
    return p, g, x, k


def check_even_grid_products(A, g, x, k, tol=1e-5):
    """A x whole, A^H k and A^H A x on the fixture's sample of voxels -- errors relative to the norm of the reference's vector"""
    pick = g["pick"]
    Ax = A * x
    assert rel_err(Ax, g["sense_Ax"]) < tol
    assert rel_err(Ax, g["sense_O3_Ax"]) < tol
    scale = np.sqrt(pick.size / float(A.shape[1]))          # a sample of n of P entries carries ~sqrt(n / P) of the norm
    AHk = (A.H * k)[pick]
    assert np.linalg.norm(AHk - g["sense_AHk_pick"]) < tol * scale * float(g["sense_AHk_norm"])
    AHAx = (A.H * (A * x))[pick]
    assert np.linalg.norm(AHAx - g["sense_AHAx_pick"]) < tol * scale * float(g["sense_AHAx_norm"])
    assert abs(np.linalg.norm(g["sense_AHAx_pick"]) / (scale * float(g["sense_AHAx_norm"])) - 1) < 0.05     # (the sample is representative)


def test_even_grid_fixture_construction_and_gprime(oracle_backend):
    """OUR construction against the reference at a second size, on an even grid (64^3 x 8 coils, grid 128^3): the products of the
    -O3 tree and of the directly fused tree on the oracle backend, and the reference's own G' -- same nonzeros, same sum, and an
    imaginary part that is rounding residue (7e-14 of the real part in the reference's matrix): what the GPU's real-weight formats
    rest on"""
    from indigo_amd.backends.hip import weights_are_real
    p, g, x, k = even_grid_problem()
    Gp = p.fused_interp(2)
    # (a sample exactly on a grid point -- the centre of every spoke -- has four taps per axis of which the outermost weighs zero:
    # the reference's sparse products drop those 37 explicit zeros per spoke, our builder stores them)
    assert np.count_nonzero(Gp.data) == int(g["gprime_nnz"]) and Gp.nnz - int(g["gprime_nnz"]) == 37 * 48
    assert abs(Gp.data.astype(np.complex128).sum() - complex(g["gprime_sum"])) < 1e-5 * abs(complex(g["gprime_sum"]))
    assert float(g["gprime_max_abs_imag"]) < 1e-12 * float(g["gprime_max_abs_real"])
    assert abs(np.abs(Gp.data.real).max() / float(g["gprime_max_abs_real"]) - 1) < 1e-6 and weights_are_real(Gp.data)
    oracle_backend._scratch = None
    check_even_grid_products(p.build_tree(oracle_backend, level=3), g, x, k)
    oracle_backend._scratch = None
    check_even_grid_products(p.build_fused(oracle_backend), g, x, k)
    oracle_backend._scratch = None


def _strip_descriptions(node):
    """materialise every sparse leaf and forget what it is: the realisation passes then multiply with scipy, as the reference does"""
    if isinstance(node, op.SpMatrix):
        node._matrix
        node._struct = None
    for c in getattr(node, '_children', None) or []:
        _strip_descriptions(c)


@pytest.mark.parametrize("masked", [False, True])
def test_structured_realisation_equals_the_scipy_products(oracle_backend, masked):
    """`pics.py -O3` on the factories' tree (examples/pics.py:104-193) two ways: the realisation passes composing the factors'
    descriptions (indigo_amd.structured: no sparse-sparse product at all) and the same passes on plain scipy matrices, as the
    reference runs them.  G' = interp * mod * scale and S' = (I (x) mod * zpad * apod) * maps come out with the same pattern and
    the same values to float32 rounding; the trees evaluate alike; and the HIP backend's native G' (one pass over the
    trajectory, ig_interp3_fill_modulated, in the fused leaf's column order) equals the scipy product renumbered."""
    from indigo_amd import fused
    from indigo_amd.structured import AdjointS, InterpS, StackS
    from indigo_amd.transforms import sense_recipe
    B = oracle_backend
    B._scratch = None
    p = SenseProblem.synthetic((12, 10, 8), 3, nspokes=11, nreadout=24, oversamp=1.5, seed=17)
    if masked:
        p.maps[:, :5, :, 1] = 0
        p.maps[:3] = 0
    trees = []
    for strip in (False, True):
        A = p.build_tree(B, level=0)
        if strip:
            _strip_descriptions(A)
        for Step in sense_recipe(3):
            A = Step().visit(A)
        trees.append(A)
    At, As = trees
    Lt = [n for n in _leaves(At) if isinstance(n, op.SpMatrix)]
    Ls = [n for n in _leaves(As) if isinstance(n, op.SpMatrix)]
    assert len(Lt) == len(Ls) == 2
    assert isinstance(Lt[0]._struct, InterpS) and Ls[0]._struct is None and Ls[1]._struct is None
    # S' (stored transposed by MriGoodAdjoints): the blocks (mod * zpad * apod) * diag(map_c) over one shared pattern
    assert isinstance(Lt[1]._struct, AdjointS) and isinstance(Lt[1]._struct.inner, StackS) and Lt[1]._struct.inner.shared_pattern()
    for a, b in zip(Lt, Ls):
        Ma, Mb = a._matrix.tocsr().astype(C64), b._matrix.tocsr().astype(C64)
        Ma.sort_indices(), Mb.sort_indices()
        # (scipy's products drop exact zeros -- taps at the very edge of the kernel, masked maps --, the descriptions keep them as
        # zero weights: the matrices are compared as matrices, not pattern by pattern)
        assert Ma.shape == Mb.shape and Ma.nnz >= Mb.nnz and abs(Ma - Mb).max() <= 3e-7 * abs(Mb).max()
    from indigo_amd.util import rand64c
    x, k = rand64c(At.shape[1], 1, seed=1), rand64c(At.shape[0], 1, seed=2)
    assert rel_err(At * x, As * x) < 2e-6 and rel_err(At.H * k, As.H * k) < 2e-6
    # the native builder of the HIP backend's host library against the scipy product, in the fused leaf's (x, z, y) column order
    import types
    from indigo_amd.backends.hip import HipBackend
    Gn = HipBackend.gridding_from_struct(types.SimpleNamespace(), Lt[0]._struct, 1)
    Gs = fused.permute_grid_columns(Ls[0]._matrix.tocsr().astype(C64), p.oN)
    assert Gn.shape == Gs.shape and Gn.has_sorted_indices and abs(Gn - Gs).max() <= 3e-7 * abs(Gs).max()
    B._scratch = None


def test_head_rows_operator_alpha_beta_and_panels(oracle_backend):
    """operators.HeadRows (a coil chunk padded with zero-weight coils: the first rows of its tree): forward and adjoint with
    alpha / beta and a two-column panel against the explicit matrix rows"""
    import scipy.sparse as spp
    B = oracle_backend
    B._scratch = None
    rng = np.random.default_rng(5)
    M = (spp.random(9, 6, density=0.6, random_state=rng) + 1j * spp.random(9, 6, density=0.6, random_state=rng)).astype(C64).tocsr()
    H = op.HeadRows(B, B.SpMatrix(M), 5)
    assert H.shape == (5, 6)
    Md = M.toarray()[:5]
    for ncol in (1, 2):
        x = (rng.random((6, ncol)) + 1j * rng.random((6, ncol))).astype(C64)
        y0 = (rng.random((5, ncol)) + 1j * rng.random((5, ncol))).astype(C64)
        y = B.copy_array(np.asfortranarray(y0))
        H.eval(y, B.copy_array(np.asfortranarray(x)), alpha=0.5 - 1j, beta=2.0)
        np.testing.assert_allclose(y.to_host(), (0.5 - 1j) * (Md @ x) + 2.0 * y0, rtol=2e-6, atol=1e-6)
        k = (rng.random((5, ncol)) + 1j * rng.random((5, ncol))).astype(C64)
        z0 = (rng.random((6, ncol)) + 1j * rng.random((6, ncol))).astype(C64)
        z = B.copy_array(np.asfortranarray(z0))
        H.H.eval(z, B.copy_array(np.asfortranarray(k)), alpha=1j, beta=-0.5)
        np.testing.assert_allclose(z.to_host(), 1j * (Md.conj().T @ k) - 0.5 * z0, rtol=2e-6, atol=1e-6)
        # beta = 0 never reads the output
        yn = B.copy_array(np.full((5, ncol), np.nan + 1j * np.nan, dtype=C64, order='F'))
        H.eval(yn, B.copy_array(np.asfortranarray(x)))
        np.testing.assert_allclose(yn.to_host(), Md @ x, rtol=2e-6, atol=1e-6)
    B._scratch = None

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


@pytest.mark.parametrize("level", [0, 1, 2, 3, "fused", "zpadfft", "zpadfft-xyz", "zpadfft-il"])
def test_sense_forward_adjoint_normal(prob, oracle_backend, level):
    p, g = prob
    B = oracle_backend
    if hasattr(B, '_scratch'):
        B._scratch = None
    if level == "fused":
        A = p.build_fused(B)
    elif level in ("zpadfft", "zpadfft-xyz", "zpadfft-il"):
        A = p.build_zpadfft(B, layout={"zpadfft": 1, "zpadfft-xyz": 0, "zpadfft-il": 2}[level])
    else:
        A = p.build_tree(B, level=level)
    x, k = g["sense_x"], g["sense_k"]
    assert rel_err(A * x, g["sense_Ax"]) < 1e-5
    assert rel_err(A.H * k, g["sense_AHk"]) < 1e-5
    if level in (3, "fused", "zpadfft", "zpadfft-xyz", "zpadfft-il"):
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

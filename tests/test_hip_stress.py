"""Short version of tests/stress_adjoint.py in the GPU suite: the adjoint of the 2 + 2 + 2 + 1 coil-chunk tree (slot scatter
k_grid_slots<2> / <1>, chunk VStack) evaluated repeatedly, every stage checked bit for bit against its first evaluation over
NaN-poisoned memory (the one intermittent deviation of round 3).  The reference's scatter is race-free by
construction (indigo/backends/_customgpu.cu:49-81); so must this one be."""
import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def test_adjoint_of_coil_chunks_is_bitwise_repeatable(hip, oracle_backend):
    import stress_adjoint
    h = stress_adjoint.Harness(hip, log=lambda s: None)
    h.make_references(oracle_backend)
    findings = []
    for it in range(30):
        if it == 15:
            h.build()                       # all formats rebuilt (threaded host builders) half way
        findings += h.check(it)
    hip._scratch = None
    assert not findings, "\n".join(findings)


def test_adjoint_of_eight_coil_bricks_is_bitwise_repeatable(hip):
    """the same for the round format of the headline tree at reduced size: k_grid_bricks<8, 8, PAIR> writing by the 4-point support
    table (pairs of 4-cell segments per wave store), 8-byte real-weight entries"""
    import stress_adjoint
    h = stress_adjoint.Harness(hip, coils=8, chunk=8, log=lambda s: None)
    h.make_references(None)
    findings = []
    for it in range(10):
        findings += h.check(it)
    hip._scratch = None
    assert not findings, "\n".join(findings)


def test_adjoint_of_shares_on_the_matrix_cores_is_bitwise_repeatable(hip):
    """round 6: the same at kernel half-width 3 (the reference's default), where the 8-coil tree scatters (sample, brick) shares with computed
    taps on the MFMA pipe (k_grid_scatter_mfma, bricks of 16 x 4 x 4 cells in registers): bit for bit on every brick no shared piece touches,
    unflagged segments still NaN, the cropped transform of the poisoned grid bit for bit"""
    import stress_adjoint
    h = stress_adjoint.Harness(hip, coils=8, chunk=8, width=3, log=lambda s: None)
    assert [inf['fmt'] for inf in h.info] == ['shares']
    h.make_references(None)
    findings = []
    for it in range(10):
        if it == 5:
            h.build()
        findings += h.check(it)
    hip._scratch = None
    assert not findings, "\n".join(findings)


@pytest.mark.parametrize("coils,chunk,widths", [(6, 4, [4, 2]), (7, 4, [4, 4])])
def test_adjoint_of_chunks_of_different_widths_is_bitwise_repeatable(hip, oracle_backend, coils, chunk, widths):
    """Chunks of DIFFERENT widths under one VStack -- 6 coils as 4 + 2: the brick rounds of the 4-wide chunk write by the 8-point
    table, the slots of the 2-wide one by the 16-point table, both formats and both tables on ONE device matrix, the repacked-panel
    buffer of the library resized between them, the scratch arena reused chunk after chunk -- and a chunk padded with a zero-weight
    coil (7 coils as 4 + 4, operators.HeadRows).  Every stage of every chunk bit for bit against its first evaluation over
    NaN-poisoned memory, the chain against the oracle; a deviation names stage, chunk, bricks and whether they are shared."""
    import stress_adjoint
    h = stress_adjoint.Harness(hip, coils=coils, chunk=chunk, log=lambda s: None)
    assert [inf['nc'] for inf in h.info] == widths
    h.make_references(oracle_backend)
    findings = []
    for it in range(12):
        if it == 6:
            h.build()
        findings += h.check(it)
    hip._scratch = None
    assert not findings, "\n".join(findings)

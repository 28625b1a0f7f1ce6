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
    """the same for the round format (k_grid_bricks<8,8>, 8-point support table) of the headline tree, at reduced size"""
    import stress_adjoint
    h = stress_adjoint.Harness(hip, coils=8, chunk=8, log=lambda s: None)
    h.make_references(None)
    findings = []
    for it in range(10):
        findings += h.check(it)
    hip._scratch = None
    assert not findings, "\n".join(findings)

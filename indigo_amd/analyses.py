"""Tree analyses: memory estimate, scratch-arena sizing, class search.

`Memusage` keeps the reference's meaning (indigo/analyses.py:10-71: bytes of all
distinct matrices + the deepest stack of live Product temporaries) but fixes
its KronI handling -- the reference dispatches on a class named `KronI` that
does not exist (analyses.py:45 vs operators.py:357), so the coil factor is never
applied and the arena reserved by `Optimize` is too small for SENSE trees.
`ScratchUsage` is the part of that estimate that `Backend.scratch` actually
serves, in complex64 elements, including the arena's 32-element alignment.
"""
from contextlib import contextmanager

import numpy as np

from indigo_amd.transforms import Visitor
from indigo_amd import operators as op


class Memusage(Visitor):
    """Peak bytes needed to evaluate a tree on panels of `ncols` columns."""

    def measure(self, node, ncols=1):
        self._seen = set()
        self._base = 0            # matrices, counted once each
        self._live = [0]          # stack of temporaries alive at the current depth
        self._cols = [ncols]
        self._peak = 0
        self.visit(node)
        return self._base + self._peak

    @contextmanager
    def _push(self, stack, value):
        stack.append(value)
        try:
            yield
        finally:
            stack.pop()

    def _ncols(self):
        return int(np.prod(self._cols, dtype=np.int64))

    def _round(self, nbytes):
        return nbytes

    def visit(self, node):
        # pre-order bookkeeping with scoped pushes, so do not use Visitor's post-order dispatch
        self._peak = max(self._peak, sum(self._live))
        if isinstance(node, (op.Product, op.UnscaledFFT, op.ZpadFFT, op.HeadRows)):
            with self._push(self._live, self._round(node._mem_usage(self._ncols()))):
                self._peak = max(self._peak, sum(self._live))
                self.generic_visit(node)
        elif isinstance(node, op.Kron) and isinstance(node.left, op.Eye):
            with self._push(self._cols, node.left.shape[0]):
                self.visit(node.right)
        elif isinstance(node, op.SpMatrix):
            if id(node) not in self._seen:
                self._seen.add(id(node))
                self._base += (node.shape[0] + 1) * 4 + node.nnz * 4 + node.nnz * 8
        elif isinstance(node, op.DenseMatrix):
            if id(node) not in self._seen:
                self._seen.add(id(node))
                self._base += node._matrix.nbytes
        else:
            self.generic_visit(node)


class ScratchUsage(Memusage):
    """Peak demand on the scratch arena, in complex64 elements."""

    def measure(self, node, ncols=1):
        super().measure(node, ncols)
        return self._peak // 8

    def _round(self, nbytes):
        elems = (nbytes + 7) // 8
        return (elems + 31) // 32 * 32 * 8


class TreeHasOp(Visitor):
    def __init__(self, op_classes):
        self._op_classes = tuple(op_classes)

    def search(self, node):
        self._found = False
        self.visit(node)
        return self._found

    def visit(self, node):
        if isinstance(node, self._op_classes):
            self._found = True
        self.generic_visit(node)

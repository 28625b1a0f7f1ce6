"""Synthetic-data generators and the leaf-call trace.

`rand64c` / `randM` follow the value distributions of the reference's
generators (indigo/util.py:9-30: uniform[0,1) real and imaginary parts) but
take an explicit seed, because every parity test here runs on seeded inputs.

`Trace` replaces the reference's `profile` context manager
(indigo/util.py:33-80).  The reference brackets EVERY leaf call with two device
barriers unless logging is raised above DEBUG; here tracing is opt-in, records
the reference's own algorithmic-bytes model per leaf, and never synchronises.
"""
import numpy as np
import scipy.sparse as spp


def _rng(seed):
    if isinstance(seed, np.random.Generator):
        return seed
    return np.random.default_rng(seed)


def rand64c(*shape, order='F', seed=None):
    """complex64 array, uniform[0,1) + 1j*uniform[0,1), Fortran-ordered by default."""
    rng = _rng(seed)
    n = int(np.prod(shape, dtype=np.int64))
    flat = np.empty(n, dtype=np.complex64)
    rng.random(out=flat.view(np.float32), dtype=np.float32)       # interleaved (re, im) in one pass
    return flat.reshape(shape, order=order if order in ('F', 'C') else 'F')


def randM(M, N, density, seed=None):
    """Random sparse complex64 CSR matrix whose real and imaginary parts have independent patterns."""
    rng = _rng(seed)
    re = spp.random(M, N, density=density, format='csr', dtype=np.float32, random_state=rng)
    im = spp.random(M, N, density=density, format='csr', dtype=np.float32, random_state=rng)
    return (re.astype(np.complex64) + 1j * im).tocsr()


class Trace(object):
    """Per-leaf record of (event, algorithmic bytes, flops, shape).  Attach with `backend.trace = Trace()`."""

    def __init__(self):
        self.records = []

    def add(self, event, nbytes=0, nflops=0, **extra):
        rec = dict(event=event, nbytes=float(nbytes), nflops=float(nflops))
        rec.update(extra)
        self.records.append(rec)

    def clear(self):
        self.records = []

    def total_bytes(self, event=None):
        return sum(r['nbytes'] for r in self.records if event is None or r['event'] == event)

    def by_event(self):
        out = {}
        for r in self.records:
            e = out.setdefault(r['event'], dict(calls=0, nbytes=0.0, nflops=0.0))
            e['calls'] += 1
            e['nbytes'] += r['nbytes']
            e['nflops'] += r['nflops']
        return out

"""`HipBackend`: the MI355X backend.  Every leaf kernel is one ctypes call into
``libindigo_hip.so`` (hand-written HIP for gfx950, C ABI in include/indigo_hip.h).

Plays the role `CudaBackend` plays in the reference (indigo/backends/cuda.py:
device pointers in `dndarray._arr`, pitch copies, plan cache, one call per leaf)
but is not derived from it: there are no vendor BLAS/FFT/sparse libraries
underneath, no per-leaf device synchronisation, sizes are 64-bit, and the
adjoint SpMM uses a cached transposed CSR (gather) instead of a scatter.

There is no CPU fallback.  Constructing the backend without a usable GPU or
without the built library raises RuntimeError.
"""
import ctypes
import logging
import os

import numpy as np

from indigo_amd import _lib
from indigo_amd.backends.backend import Backend

log = logging.getLogger(__name__)
_C64 = np.dtype('complex64')


def weights_are_real(data):
    """True when the imaginary parts of a complex64 array are nothing but rounding residue: at most 2^-34 of the largest magnitude
    (the residue of exp(2 pi i phase) for a phase of some hundred half turns is ~1e-13 relative; 2^-34 = 5.8e-11 is still a thousand
    times below the float32 rounding of the products it would enter)"""
    if data.size == 0:
        return False
    im = float(np.abs(data.imag).max())
    return im == 0.0 or im <= float(np.abs(data.real).max()) * 2.0 ** -34


def brick_tasks(counts, ptr, chunk, run, max_bricks=64, longest_first=True):
    """Task list and brick table of ig_ccsrmm_t_bricks (include/indigo_hip.h) from the entries per brick `counts` and their
    prefix sums `ptr`: table = (brick, end of its entries) per non-empty brick; a brick with more than `chunk` entries is
    cut into shared tasks of at most `chunk`; the others are grouped into runs of consecutive table rows -- a new run
    starts when the entry offset crosses a multiple of `run`, after a heavy brick, and after `max_bricks` bricks.  Returns
    tasks (n, 4) int32 [lo, hi, first table row, rows | shared << 16] sorted longest first (or, longest_first=False, in
    brick order), table (nb, 2) int32, and the ids of the shared bricks."""
    bricks = np.flatnonzero(counts)
    if bricks.size == 0:
        return np.zeros((0, 4), np.int32), np.zeros((0, 2), np.int32), np.zeros(0, np.int32)
    cnt = counts[bricks].astype(np.int64)
    lo_b, hi_b = ptr[bricks], ptr[bricks + 1]
    table = np.stack([bricks, hi_b], axis=1).astype(np.int32)
    heavy = cnt > chunk
    # runs of light bricks
    light = np.flatnonzero(~heavy)
    key = np.cumsum(heavy)[light] * (int(ptr[-1]) // max(run, 1) + 2) + lo_b[light] // max(run, 1)
    new_run = np.ones(light.size, dtype=bool)
    new_run[1:] = key[1:] != key[:-1]
    run_id = np.cumsum(new_run) - 1
    first_of_run = np.flatnonzero(new_run)
    rank = np.arange(light.size) - first_of_run[run_id]
    new_run |= (rank % max_bricks) == 0
    starts = np.flatnonzero(new_run)
    ends = np.append(starts[1:], light.size) - 1
    t_run = np.stack([lo_b[light[starts]], hi_b[light[ends]], light[starts], ends - starts + 1], axis=1) if light.size else np.zeros((0, 4), np.int64)
    # pieces of heavy bricks
    hv = np.flatnonzero(heavy)
    npiece = (cnt[hv] + chunk - 1) // chunk
    rep = np.repeat(np.arange(hv.size), npiece)
    firstp = np.concatenate(([0], np.cumsum(npiece)[:-1])) if hv.size else np.zeros(0, np.int64)
    part = np.arange(rep.size) - firstp[rep]
    plo = lo_b[hv][rep] + part * chunk
    phi = np.minimum(plo + chunk, hi_b[hv][rep])
    t_hv = np.stack([plo, phi, hv[rep], np.full(rep.size, 1 | (1 << 16))], axis=1) if rep.size else np.zeros((0, 4), np.int64)
    tasks = np.concatenate([t_hv, t_run]).astype(np.int32)
    order = np.argsort(-(tasks[:, 1] - tasks[:, 0]), kind='stable') if longest_first else np.argsort(tasks[:, 0], kind='stable')
    tasks = np.ascontiguousarray(tasks[order])
    return tasks, np.ascontiguousarray(table), bricks[hv].astype(np.int32)


def _cplx(v):
    v = complex(v)
    return ctypes.c_float(v.real), ctypes.c_float(v.imag)


class HipBackend(Backend):

    def __init__(self, device_id=0, stream=None):
        """`stream`: optional raw hipStream_t (int) to adopt, e.g. a torch stream's `.cuda_stream`."""
        super().__init__(device_id)
        self._L = _lib.lib()
        ctx = ctypes.c_void_p()
        if stream is None:
            rc = self._L.ig_init(int(device_id), ctypes.byref(ctx))
        else:
            rc = self._L.ig_init_on_stream(int(device_id), ctypes.c_void_p(stream), ctypes.byref(ctx))
        _lib.check(rc, None, "ig_init")
        self._ctx = ctx
        self.device_id = int(device_id)
        self._plans = dict()
        # 'transpose': adjoint of a non-exwrite matrix gathers through a cached CSR of A^T (deterministic)
        # 'atomic'   : adjoint scatters with float atomics straight from A's CSR (no extra memory)
        self.adjoint_policy = 'transpose'
        # Format choices of this backend's matrices and fused trees (defaults = the measured best; tests switch routes off to
        # reach the fallback kernels):
        #   bricks        coil counts whose interleaved adjoint gridding is the brick-binned scatter (others: gather over G'^T)
        #   support_tile  kx points per entry of the fine k-space support table of coil-interleaved trees, by coil count (16: one table only)
        #   brick_shape   per coil count: (grid lines, slabs, heavy-brick piece, entries per run) of the binned format.  Four coils
        #                 pad a sample's share of a brick to 16 entries: bricks of 16 x 2 x 4 cells waste less than 16 x 2 x 2 (0.61
        #                 against 0.68 ms; profiles/r03_brick_shape_sweep.txt)
        #   xrows         wide panels (16..64 columns): repack only the panel rows the matrix touches (forward)
        #   wide_bricks   64-column column-major panels: brick scatter through LDS (adjoint)
        #   slots         coil counts whose adjoint gridding is the slot-format scatter (ig_ccsrmm_t_slots): the ranks of a coil-sharded
        #                 run with one or two coils
        #   placement_candidates / placement_min_bytes   arrays of at least that many bytes (scratch arenas, grids) are allocated that many
        #                 times, probed (ig_probe_placement) and the best-placed candidate kept; 1 = plain allocation
        #   cg_graph      HipBackend.cg replays a block of iterations as one HIP graph launch (ig_graph_*).  Off: measured on the headline
        #                 problem the replay saves 0.03 ms of a 6.89 ms iteration and recording costs 5 ms per solve (profiles/r05_cg_graph_ab.log)
        #   placement_window_gb   (round 6) a large array is a window of ONE allocation that many GB larger, placed at the best-probing 1 GB step
        #   separable / sep_gather / sep_scatter   the gridding matrix in separable form (one record per sample) and the kernels that compute
        #                 their taps from it: forward on every even grid; adjoint (shares) for `shares` coil counts from `shares_min_tw` taps
        #                 per axis on (kernel half-width > 2: below that the stored-tap bricks are faster)
        #   share_shape   per coil count: (grid lines, slabs, heavy-brick piece, shares per run).  The PIECE bounds how many samples a wave sums
        #                 into one float32 image before it adds it to the grid: pieces of 1024 shares put the evaluation 2.1e-5 from the float64
        #                 one on an ill-conditioned problem (oversampling 1.25, half-width 3: the k-space centre's sum leaks to the image's edge,
        #                 where the roll-off correction is large); 128: 4.7 ... 6.1e-6 over four runs, the stored-tap scatter 4.8 ... 7.6e-6 (the
        #                 order of the float atomics differs from run to run), 0.95 against 0.98 ms (profiles/r06_share_pieces.txt)
        self._placement_log = []          # (bytes, candidate probe times in ms, chosen) of every array placed by probing
        self.tuning = dict(placement_candidates=3, placement_min_bytes=1 << 31, placement_window_gb=24, placement_window_allocs=3, fold_odd_axes=True, real_gridding=True, gather_order=True, cg_graph=False, bricks=(4, 8), slots=(1, 2), slot_shape=(4, 4, 256, 64), support_tile={8: 4, 4: 8}, brick_shape={8: (2, 2, 4096, 4096), 4: (2, 4, 4096, 4096)}, xrows=True, runs=True, wide_bricks=True,
                           wide_brick_shape=(2, 2), wide_task_shape=(8192, 2048),
                           # round 6: gridding from the separable form of the matrix (one record per sample, taps computed)
                           separable=True, sep_gather=True, sep_scatter=True, shares=(4, 8), shares_min_tw=6, share_shape={8: (4, 4, 128, 1024), 4: (4, 4, 128, 1024)})

    def __del__(self):
        try:
            for entry in getattr(self, '_plans', {}).values():
                self._L.ig_fft_destroy(entry[0])
            if getattr(self, '_ctx', None):
                self._L.ig_destroy(self._ctx)
                self._ctx = None
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != 0:
            _lib.check(rc, self._ctx, what)

    # -- housekeeping ---------------------------------------------------------------
    def barrier(self):
        self._check(self._L.ig_sync(self._ctx), "ig_sync")

    @property
    def stream(self):
        """raw hipStream_t of this backend (int)"""
        return self._L.ig_stream(self._ctx) or 0

    def device_name(self):
        buf = ctypes.create_string_buffer(256)
        self._check(self._L.ig_device_name(self._ctx, buf, 256), "ig_device_name")
        return buf.value.decode()

    def set_option(self, name, value):
        """plan options of the library (ig_set_option): 'fft.kernels' = 0 all kernels, 1 no A x B passes, 2 generic stages only;
        takes effect for plans made afterwards (cached plans are dropped)"""
        self._check(self._L.ig_set_option(self._ctx, name.encode(), int(value)), "ig_set_option(%s)" % name)
        for entry in self._plans.values():
            self._L.ig_fft_destroy(entry[0])
        self._plans = dict()

    def mem_usage(self):
        """bytes of device memory in use through this backend: the arrays it handed out (Backend.mem_usage, backend.py:249)
        plus what the library holds on its own (the SpMM kernels' repacked-panel buffer, work lists, reduction scratch)"""
        own = ctypes.c_size_t()
        self._check(self._L.ig_library_bytes(self._ctx, ctypes.byref(own)), "ig_library_bytes")
        return super().mem_usage() + own.value

    def mem_info(self):
        free, total = ctypes.c_size_t(), ctypes.c_size_t()
        self._check(self._L.ig_mem_info(self._ctx, ctypes.byref(free), ctypes.byref(total)), "ig_mem_info")
        return free.value, total.value

    # events: (start, stop) pairs on the backend's stream, no host sync until `elapsed_ms`
    def event(self):
        ev = ctypes.c_void_p()
        self._check(self._L.ig_event_create(self._ctx, ctypes.byref(ev)), "ig_event_create")
        return ev

    def record(self, ev):
        self._check(self._L.ig_event_record(ev), "ig_event_record")

    def elapsed_ms(self, start, stop):
        ms = ctypes.c_float()
        self._check(self._L.ig_event_elapsed_ms(start, stop, ctypes.byref(ms)), "ig_event_elapsed_ms")
        return ms.value

    def event_destroy(self, ev):
        self._L.ig_event_destroy(ev)

    def profile(self, on=True):
        """bracket every kernel launch with stream events (no host sync) until switched off"""
        self._check(self._L.ig_prof_enable(self._ctx, 1 if on else 0), "ig_prof_enable")
        self._prof_on = bool(on)

    def profile_report(self):
        """{kernel: dict(launches, total_ms, avg_ms, bytes)} since the last report; synchronises"""
        buf = ctypes.create_string_buffer(1 << 16)
        self._check(self._L.ig_prof_report(self._ctx, buf, len(buf)), "ig_prof_report")
        out = {}
        for line in buf.value.decode().splitlines():
            name, n, ms, nbytes = line.split()
            out[name] = dict(launches=int(n), total_ms=float(ms), avg_ms=float(ms) / max(int(n), 1), bytes=float(nbytes))
        return out

    # -- arrays -----------------------------------------------------------------------
    class dndarray(Backend.dndarray):
        """`_arr` is the device address (Python int)."""

        def _malloc(self, shape, dtype):
            b = self._backend
            if self.nbytes >= int(b.tuning.get('placement_min_bytes', 1 << 31)):
                slack = int(b.tuning.get('placement_window_gb', 0))
                if slack > 0:
                    ptr = self._malloc_window(slack)
                    if ptr is not None:
                        return ptr
                ncand = int(b.tuning.get('placement_candidates', 1))
                if ncand > 1:
                    return self._malloc_best_placed(ncand)
            ptr = ctypes.c_void_p()
            b._check(b._L.ig_malloc(b._ctx, self.nbytes, ctypes.byref(ptr)), "ig_malloc(%d bytes)" % self.nbytes)
            return ptr.value

        def _malloc_window(self, slack_gb):
            """A large array as the best-placed WINDOW of an allocation (round 6).  The passes that step megabytes per element run
            3 ... 6 % faster or slower with where their array lies -- from allocation to allocation (every other
            process or so gets a slow first one: 3.6 - 3.7 ms in the placement probe against 3.2), and inside ONE allocation the
            probe's time changes smoothly with the offset (profiles/r06_placement_offsets.txt, r06_placement_slack.txt): a slow level
            (3.7 ms for the headline's arena) over the first 0 ... 13 GB of an allocation, a ramp down over the next 7 - 8 GB, a fast
            plateau (2.9 ms) behind; what looked like two kinds of allocation in round 5 is where an allocation starts relative to that
            ramp.  So: an allocation of nbytes + slack_gb GB (24: the plateau has been inside it on every box so far), the probe
            (ig_probe_placement) on the window at every GB step, the fastest window kept.  An allocation without a ramp (all windows
            alike: all slow) is followed by another, up to `placement_window_allocs`; the losers are freed at the end.  slack_gb GB stay
            held beyond the array.  None (the caller falls back) when the device has no room for the slack."""
            b = self._backend
            GBs = 1 << 30
            nalloc = max(1, int(b.tuning.get('placement_window_allocs', 3)))
            worst = b.tuning.get('placement_pick') == 'worst'
            cands = []          # (best time, base, pick, times)
            try:
                for _ in range(nalloc):
                    free, total = ctypes.c_size_t(), ctypes.c_size_t()
                    # (a further allocation only while the device has room for it and as much again: another process on the same GPU
                    # must not fail because of a transient candidate)
                    need = (self.nbytes + (slack_gb + 4) * GBs) if not cands else 2 * (self.nbytes + slack_gb * GBs)
                    if b._L.ig_mem_info(b._ctx, ctypes.byref(free), ctypes.byref(total)) != 0 or free.value < need:
                        break
                    base = ctypes.c_void_p()
                    if b._L.ig_malloc(b._ctx, self.nbytes + slack_gb * GBs, ctypes.byref(base)) != 0:
                        break
                    cands.append([None, base.value, 0, []])
                    times = []
                    for off in range(slack_gb + 1):
                        ms = ctypes.c_double(0.0)
                        b._check(b._L.ig_probe_placement(b._ctx, ctypes.c_void_p(base.value + off * GBs), self.nbytes, ctypes.byref(ms)), "ig_probe_placement")
                        times.append(ms.value)
                    if len(cands) == 1:
                        # (the first window may have been timed while the clocks were still coming up: once more)
                        ms = ctypes.c_double(0.0)
                        b._check(b._L.ig_probe_placement(b._ctx, base, self.nbytes, ctypes.byref(ms)), "ig_probe_placement")
                        times[0] = min(times[0], ms.value)
                    pick = int(np.argmax(times)) if worst else int(np.argmin(times))
                    cands[-1] = [times[pick], base.value, pick, times]
                    # enough once a RAMP has been seen -- probes 8 % apart: the slow level (~3.7 ms on the headline's arena) and the fast one
                    # (2.9) or the way down between them -- the best window is then on the fast side.  An allocation whose windows all
                    # probe alike is all slow (a fast one always starts with a ramp): another one is tried, up to `placement_window_allocs`
                    # (the losers are held until the choice is made: freed at once, the next allocation of the same size would get their
                    # pages back)
                    allt = [v for c in cands for v in c[3]]
                    if min(allt) <= 0.92 * max(allt):
                        break
            except Exception:
                for c in cands:
                    if c[1] is not None:
                        b._L.ig_free(b._ctx, ctypes.c_void_p(c[1]))
                raise
            for c in cands:
                if c[0] is None:
                    b._L.ig_free(b._ctx, ctypes.c_void_p(c[1]))
            cands = [c for c in cands if c[0] is not None]
            if not cands:
                return None
            keep = (max if worst else min)(range(len(cands)), key=lambda i: cands[i][0])
            for i, c in enumerate(cands):
                if i != keep:
                    b._L.ig_free(b._ctx, ctypes.c_void_p(c[1]))
            t, base, pick, times = cands[keep]
            self._alloc_base = base
            b._placement_log.append((self.nbytes, [[round(v, 4) for v in c[3]] for c in cands], round(t, 4)))
            log.debug("placement: %d bytes, windows at +0 .. +%d GB of %d allocation(s) %s ms -> allocation %d, +%d GB", self.nbytes, slack_gb, len(cands),
                      [[round(v, 4) for v in c[3]] for c in cands], keep, pick)
            return base + pick * GBs

        def _malloc_best_placed(self, ncand):
            """A large array (a scratch arena, a grid): allocate up to `ncand` candidates, time the library's placement probe on each
            (ig_probe_placement: the write pattern of a pass that steps megabytes per element) and keep the fastest -- allocations of
            one size made by one process differ by 3 ... 6 % in such passes, repeatably.  Candidates that no longer
            fit are simply not tried; the losers are freed before this returns."""
            b = self._backend
            cands = []
            keep = None
            try:
                for _ in range(ncand):
                    if cands:
                        # a further candidate only while the device has room for it AND as much again: another process on the same
                        # GPU (ranks sharing a device in a rehearsal, another tenant) must not fail because of transient candidates
                        free, total = ctypes.c_size_t(), ctypes.c_size_t()
                        if b._L.ig_mem_info(b._ctx, ctypes.byref(free), ctypes.byref(total)) != 0 or free.value < 2 * self.nbytes:
                            break
                    ptr = ctypes.c_void_p()
                    rc = b._L.ig_malloc(b._ctx, self.nbytes, ctypes.byref(ptr))
                    if rc != 0:
                        if not cands:
                            b._check(rc, "ig_malloc(%d bytes)" % self.nbytes)
                        break
                    cands.append([None, ptr.value])
                    ms = ctypes.c_double(0.0)
                    b._check(b._L.ig_probe_placement(b._ctx, ptr, self.nbytes, ctypes.byref(ms)), "ig_probe_placement")
                    cands[-1][0] = ms.value
                if len(cands) > 1:
                    # the first candidate may have been timed while the clocks were still coming up (the probe is often the first work of
                    # a process): time it once more, now behind the others, and keep its better figure
                    ms = ctypes.c_double(0.0)
                    b._check(b._L.ig_probe_placement(b._ctx, ctypes.c_void_p(cands[0][1]), self.nbytes, ctypes.byref(ms)), "ig_probe_placement")
                    cands[0][0] = min(cands[0][0], ms.value)
                best = max(cands) if b.tuning.get('placement_pick') == 'worst' else min(cands)       # ('worst': lab, to see what the probe's spread is worth)
                keep = best[1]
            finally:
                # the losers -- all candidates if a probe failed -- are freed whatever happened
                for _, ptr in cands:
                    if ptr != keep:
                        b._L.ig_free(b._ctx, ctypes.c_void_p(ptr))
            b._placement_log.append((self.nbytes, [round(ms, 4) for ms, _ in cands], round(best[0], 4)))
            log.debug("placement: %d bytes, candidates %s ms -> %.4f", self.nbytes, [round(ms, 4) for ms, _ in cands], best[0])
            return best[1]

        def _free(self):
            b = self._backend
            if getattr(b, '_ctx', None):
                b._L.ig_free(b._ctx, ctypes.c_void_p(getattr(self, '_alloc_base', None) or self._arr))

        def _zero(self):
            b = self._backend
            if self.ndim == 2 and not self.contiguous:
                # strided view: clear column by column through a scale by zero
                b.scale(self, 0)
            else:
                b._check(b._L.ig_memset0(b._ctx, ctypes.c_void_p(self._arr), self.nbytes), "ig_memset0")

        def _pitched(self, host_rows):
            """(device pitch, host pitch, row bytes, count) for a 2-d column-major transfer"""
            return (self._leading_dim * self.itemsize, host_rows * self.itemsize,
                    self.shape[0] * self.itemsize, self.shape[1])

        def _copy_from(self, arr):
            assert arr.flags['F_CONTIGUOUS']
            b = self._backend
            src = ctypes.c_void_p(arr.ctypes.data)
            dst = ctypes.c_void_p(self._arr)
            if self.size == 0:
                return
            if self.ndim == 2 and not self.contiguous:
                dpitch, spitch, width, height = self._pitched(self.shape[0])
                rc = b._L.ig_copy2d(b._ctx, dst, dpitch, src, spitch, width, height, _lib.IG_H2D)
            else:
                assert self.contiguous
                rc = b._L.ig_copy2d(b._ctx, dst, self.nbytes, src, self.nbytes, self.nbytes, 1, _lib.IG_H2D)
            b._check(rc, "ig_copy2d(H2D)")

        def _copy_to(self, arr):
            b = self._backend
            if self.size == 0:
                return
            out = arr if arr.flags['F_CONTIGUOUS'] else np.empty(self.shape, self.dtype, order='F')
            dst = ctypes.c_void_p(out.ctypes.data)
            src = ctypes.c_void_p(self._arr)
            if self.ndim == 2 and not self.contiguous:
                spitch, dpitch, width, height = self._pitched(self.shape[0])
                rc = b._L.ig_copy2d(b._ctx, dst, dpitch, src, spitch, width, height, _lib.IG_D2H)
            else:
                assert self.contiguous
                rc = b._L.ig_copy2d(b._ctx, dst, self.nbytes, src, self.nbytes, self.nbytes, 1, _lib.IG_D2H)
            b._check(rc, "ig_copy2d(D2H)")
            if out is not arr:
                arr[...] = out.reshape(arr.shape, order='F')

        def _copy(self, d_arr):
            """device -> device: self <- d_arr"""
            b = self._backend
            assert self.size == d_arr.size
            if self.size == 0:
                return
            dst, src = ctypes.c_void_p(self._arr), ctypes.c_void_p(d_arr._arr)
            if self.ndim == 2 and d_arr.ndim == 2 and not (self.contiguous and d_arr.contiguous):
                assert self.shape == d_arr.shape
                rc = b._L.ig_copy2d(b._ctx, dst, self._leading_dim * self.itemsize,
                                    src, d_arr._leading_dim * d_arr.itemsize,
                                    self.shape[0] * self.itemsize, self.shape[1], _lib.IG_D2D)
            else:
                assert self.contiguous and d_arr.contiguous
                rc = b._L.ig_copy2d(b._ctx, dst, self.nbytes, src, self.nbytes, self.nbytes, 1, _lib.IG_D2D)
            b._check(rc, "ig_copy2d(D2D)")

        def __getitem__(self, slc):
            """Contiguous-box slicing; returns a view (pointer + F-order offset, same leading dim)."""
            if not isinstance(slc, tuple):
                slc = (slc,)
            slc = slc + (slice(None),) * (self.ndim - len(slc))
            start, shape = [], []
            for s, n in zip(slc, self.shape):
                if isinstance(s, (int, np.integer)):
                    s = slice(int(s), int(s) + 1)
                assert s.step in (None, 1), "strided slices are not supported"
                b, e, _ = s.indices(n)
                if e < b:
                    e = b
                start.append(b)
                shape.append(e - b)
            if self.ndim == 1:
                offset = start[0]
            else:
                # element (i0, i1, ...) lives at i0 + ld*(i1 + shape[1]*(i2 + ...))
                offset, stride = start[0], self._leading_dim
                for d in range(1, self.ndim):
                    offset += start[d] * stride
                    stride *= self.shape[d]
            ptr = self._arr + offset * self.itemsize
            ld = shape[0] if self.ndim == 1 else self._leading_dim
            return self._view(tuple(shape), ld, ptr)

    # -- BLAS-1 ---------------------------------------------------------------------------
    def _flat(self, a):
        assert a.contiguous or a.ndim == 2, "unsupported layout"
        return a.contiguous

    def axpby(self, beta, y, alpha, x):
        """y = beta*y + alpha*x  (one fused pass; the CUDA reference takes two, cuda.py:239-248)"""
        assert isinstance(x, self.dndarray) and isinstance(y, self.dndarray)
        assert x.dtype == _C64 and y.dtype == _C64, "only complex64 is supported"
        assert x.size == y.size
        br, bi = _cplx(beta)
        ar, ai = _cplx(alpha)
        if self._flat(y) and self._flat(x):
            self._check(self._L.ig_caxpby(self._ctx, y.size, br, bi, ctypes.c_void_p(y._arr),
                                          ar, ai, ctypes.c_void_p(x._arr)), "ig_caxpby")
        else:
            x2 = x if x.ndim == 2 else x.reshape(y.shape)
            assert x2.shape == y.shape
            for j in range(y.shape[1]):
                yp = y._arr + j * y._leading_dim * 8
                xp = x2._arr + j * x2._leading_dim * 8
                self._check(self._L.ig_caxpby(self._ctx, y.shape[0], br, bi, ctypes.c_void_p(yp),
                                              ar, ai, ctypes.c_void_p(xp)), "ig_caxpby")

    def scale(self, x, alpha):
        assert isinstance(x, self.dndarray) and x.dtype == _C64
        ar, ai = _cplx(alpha)
        if self._flat(x):
            self._check(self._L.ig_cscal(self._ctx, x.size, ar, ai, ctypes.c_void_p(x._arr)), "ig_cscal")
        else:
            for j in range(x.shape[1]):
                xp = x._arr + j * x._leading_dim * 8
                self._check(self._L.ig_cscal(self._ctx, x.shape[0], ar, ai, ctypes.c_void_p(xp)), "ig_cscal")

    def dot(self, x, y):
        """Re(x^H y) as a Python float (device -> host sync point)"""
        assert x.dtype == _C64 and y.dtype == _C64 and x.size == y.size
        assert x.contiguous and y.contiguous
        out = (ctypes.c_double * 2)()
        self._check(self._L.ig_cdotc(self._ctx, x.size, ctypes.c_void_p(x._arr), ctypes.c_void_p(y._arr), out), "ig_cdotc")
        return out[0]

    def cdot(self, x, y):
        """full complex x^H y"""
        out = (ctypes.c_double * 2)()
        self._check(self._L.ig_cdotc(self._ctx, x.size, ctypes.c_void_p(x._arr), ctypes.c_void_p(y._arr), out), "ig_cdotc")
        return complex(out[0], out[1])

    def norm2(self, x):
        """||x||^2"""
        assert x.dtype == _C64 and x.contiguous
        out = ctypes.c_double()
        self._check(self._L.ig_scnrm2sq(self._ctx, x.size, ctypes.c_void_p(x._arr), ctypes.byref(out)), "ig_scnrm2sq")
        return out.value

    # -- CG with device-resident scalars --------------------------------------------------------
    def _slots(self):
        if getattr(self, '_scal', None) is None:
            ptr, n = ctypes.c_void_p(), ctypes.c_int()
            self._check(self._L.ig_scalars(self._ctx, ctypes.byref(ptr), ctypes.byref(n)), "ig_scalars")
            self._scal = (ptr.value, n.value)
        return self._scal

    def cg(self, A, b_h, x_h, lamda=0.0, tol=1e-10, maxiter=100, team=None, check_every=10):
        """Conjugate gradients with the iteration's scalars kept on the device (same update sequence and the same
        numbers as Backend.cg / the reference's backend.py:651-689), an iteration's vector work in three fused passes
        (ig_cg_dot, ig_cg_step_r, ig_cg_step_xp: 7 reads + 3 writes of a vector instead of 9 + 3, three launches instead of
        twelve): alpha = rr/<p,Ap> and beta = r2/rr are computed inside the update kernels from the block partials of the
        reductions, so an iteration enqueues without a host synchronisation.  The relative residuals are recorded on the device for
        EVERY iteration and fetched every `check_every` iterations (the only syncs).  The reference leaves its loop the
        moment resid < tol (backend.py:683-685); here up to check_every-1 further iterations are already enqueued by
        then, so the step length is gated on the device: once rr/r0 < tol^2 (or <p,Ap> == 0: an exactly converged
        system) alpha is 0 and those iterations leave x and r alone.  The returned history ends at the first residual
        below tol, like the reference's.  A `team` (host-scalar all-reduces per iteration, backend.py:469-479) uses the
        base implementation."""
        if team is not None or not (hasattr(A, 'eval')):
            return super().cg(A, b_h, x_h, lamda=lamda, tol=tol, maxiter=maxiter, team=team)
        A_in, lamda_in = A, lamda
        A, lamda = self._split_identity(A, lamda)
        if np.imag(lamda) != 0:
            # the fused passes carry a REAL regularisation weight; a complex one (which the reference's loop accepts,
            # backend.py:651-689) takes the base implementation with the caller's own operator -- decided before anything is allocated
            return super().cg(A_in, b_h, x_h, lamda=lamda_in, tol=tol, maxiter=maxiter, team=team)
        base, nslots = self._slots()
        S = lambda i: ctypes.c_void_p(base + 8 * i)          # slot i (a device double)
        RRA, R0, RRB, ALPHA, HIST = 0, 1, 2, 4, 8            # rr lives in two slots used in turn (ig_cg_step_xp writes the other one)
        cap = nslots - HIST                                  # history slots: a ring, fetched before it wraps
        L, ctx = self._L, self._ctx
        P = lambda a: ctypes.c_void_p(a._arr)
        x_dev = isinstance(x_h, self.dndarray)
        x = x_h if x_dev else self.copy_array(x_h, name='x')
        b = b_h.copy(name='b') if isinstance(b_h, self.dndarray) else self.copy_array(b_h, name='b')
        assert x.dtype == _C64 and b.dtype == _C64 and x.contiguous and b.contiguous
        n = x.size
        Ap = x.copy()
        r = b
        A.eval(Ap, x)
        self.axpby(1, r, -1, Ap)
        self.axpby(1, r, -lamda, x)
        p = r.copy(name='p')
        self._check(L.ig_scnrm2sq_dev(ctx, n, P(r), S(RRA)), "ig_scnrm2sq_dev")
        self._check(L.ig_scalar_copy(ctx, S(R0), S(RRA), 1), "ig_scalar_copy")
        history = []
        fetched = 0
        every = max(1, min(int(check_every), cap))
        host = (ctypes.c_double * every)()
        tol2 = float(tol) ** 2
        lam = ctypes.c_float(float(np.real(lamda)))
        it = 0
        done = False

        def iteration(i):
            rr, rr_next = (RRA, RRB) if i % 2 == 0 else (RRB, RRA)
            A.eval(Ap, p)
            # three fused passes (ig_blas.hip): Ap += lamda p and <p, Ap>;  alpha = rr / <p, Ap> (zero once rr / r0 < tol^2: the
            # reference has left its loop by then), r -= alpha Ap, ||r||^2;  beta = r2 / rr, x += alpha p, p = r + beta p,
            # rr <- r2, history[i] = r2 / r0
            self._check(L.ig_cg_dot(ctx, n, P(p), P(Ap), lam), "ig_cg_dot")
            self._check(L.ig_cg_step_r(ctx, n, P(r), P(Ap), S(rr), S(R0), tol2, S(ALPHA)), "ig_cg_step_r")
            self._check(L.ig_cg_step_xp(ctx, n, P(x), P(p), P(r), S(ALPHA), S(rr), S(rr_next), S(R0), S(HIST + i % every)), "ig_cg_step_xp")

        # A block of `every` iterations issues the same launches on the same buffers every time (an even `every` keeps the two
        # rr slots in step): the SECOND block is recorded as a HIP graph (the first ran plain: formats built, attributes set,
        # the library's buffers sized) and every later full block is one graph launch -- the gaps between ~25 dependent launches
        # per iteration shrink from the host's launch path to the device's own.  Needs the scratch arena (the evaluation's
        # temporaries must sit where they sat when recorded); anything that cannot be recorded falls back to plain launches.
        graph = None
        use_graph = (self.tuning.get('cg_graph', False) and getattr(self, '_scratch', None) is not None and every % 2 == 0
                     and maxiter >= 3 * every and getattr(self, 'trace', None) is None and not getattr(self, '_prof_on', False))
        try:
            while it < maxiter and not done:
                nblk = min(every, maxiter - it)
                if use_graph and nblk == every and it % every == 0 and it >= every:
                    if graph is None:
                        try:
                            self._check(L.ig_graph_begin(ctx), "ig_graph_begin")
                            for j in range(every):
                                iteration(it + j)
                            g = ctypes.c_void_p()
                            self._check(L.ig_graph_end(ctx, ctypes.byref(g)), "ig_graph_end")
                            graph = g
                        except Exception as e:          # noqa: BLE001 -- e.g. a leaf that synchronises: this solve runs on plain launches
                            L.ig_graph_abort(ctx)
                            log.info("cg: the iteration cannot be recorded as a graph (%s); plain launches", e)
                            use_graph = False
                            continue
                    self._check(L.ig_graph_launch(graph), "ig_graph_launch")
                else:
                    for j in range(nblk):
                        iteration(it + j)
                it += nblk
                self._check(L.ig_scalar_read(ctx, S(HIST), it - fetched, host), "ig_scalar_read")       # (the block's only synchronisation)
                for j in range(it - fetched):
                    history.append(float(np.sqrt(host[j])))
                    log.info("iter %d, residual %g", fetched + j, history[-1])
                    if history[-1] < tol:
                        log.info("cg reached tolerance")
                        done = True
                        break
                fetched = it
        finally:
            if graph is not None:
                self.barrier()
                L.ig_graph_destroy(graph)
        if not done:
            log.info("cg reached maxiter")
        if not x_dev:
            x.copy_to(x_h)
        return history

    @staticmethod
    def _split_identity(A, lamda):
        """(A', lamda') with A + lamda I = A' + lamda' I: examples/pics.py:195 puts the Tikhonov term INTO the operator
        ((A.H * A) + lamda * Eye), which costs an axpby per evaluation; a real multiple of Eye at the root of the tree is
        moved into cg's own lamda instead, where ig_cg_dot adds it on the pass that reads p and Ap anyway."""
        from indigo_amd.operators import Sum, Scale, Eye
        if isinstance(A, Sum):
            for k in (0, 1):
                c = A._children[k]
                if isinstance(c, Scale) and isinstance(c.child, Eye) and np.imag(c._val) == 0:
                    return A._children[1 - k], lamda + float(np.real(c._val))
        return A, lamda

    def max(self, val, arr):
        """elementwise max on the real and imaginary parts independently"""
        assert arr.dtype == _C64 and arr.contiguous
        self._check(self._L.ig_cmax(self._ctx, arr.size * 2, ctypes.c_float(val), ctypes.c_void_p(arr._arr)), "ig_cmax")

    # -- FFT --------------------------------------------------------------------------------
    def _get_or_create_plan(self, x_shape):
        x_shape = tuple(int(s) for s in x_shape)
        if x_shape not in self._plans:
            dims = x_shape[:-1]
            assert 1 <= len(dims) <= 3, "FFT rank must be 1, 2 or 3"
            c_dims = (ctypes.c_int64 * len(dims))(*dims)
            plan, ws = ctypes.c_void_p(), ctypes.c_size_t()
            self._check(self._L.ig_fft_plan(self._ctx, len(dims), c_dims, x_shape[-1],
                                            ctypes.byref(plan), ctypes.byref(ws)), "ig_fft_plan%s" % (x_shape,))
            ws_in = ctypes.c_size_t()
            self._check(self._L.ig_fft_inplace_workspace(plan, ctypes.byref(ws_in)), "ig_fft_inplace_workspace")
            self._plans[x_shape] = (plan, ws.value, ws_in.value)
        return self._plans[x_shape]

    def _fft_workspace_size(self, x_shape):
        """bytes of scratch an UnscaledFFT of this shape may take (operators.UnscaledFFT._mem_usage / ScratchUsage): the
        in-place figure, so that a tree sized by it never allocates inside an evaluation"""
        return self._get_or_create_plan(x_shape)[2]

    def fft_describe(self, x_shape):
        plan = self._get_or_create_plan(x_shape)[0]
        buf = ctypes.create_string_buffer(1024)
        self._check(self._L.ig_fft_describe(plan, buf, 1024), "ig_fft_describe")
        return buf.value.decode()

    def _fft(self, y, x, direction):
        assert x.dtype == _C64 and y.dtype == _C64, "only complex64 is supported"
        assert x.shape == y.shape and x.contiguous and y.contiguous
        plan, ws, ws_inplace = self._get_or_create_plan(x.shape)
        if x._arr == y._arr:
            ws = ws_inplace
        if ws and getattr(self, '_scratch', None) is None:
            # no arena reserved (a bare fftn / ifftn call, not an operator tree): a grow-only workspace kept by the backend --
            # allocating, zeroing and freeing 9 GB per call (the chirp-z columns of a 640 x 277 x 410 x 8 transform) cost 50x
            # the transform.  The kernels write every element of it before they read it.
            tmp = getattr(self, '_fft_ws', None)
            if tmp is None or tmp.nbytes < ws:
                self._fft_ws = None
                tmp = self._fft_ws = self.empty_array((int(ws) // 8,), _C64, name='fft workspace')
            rc = self._L.ig_fft_exec(plan, ctypes.c_void_p(x._arr), ctypes.c_void_p(y._arr), direction, ctypes.c_void_p(tmp._arr))
        elif ws:
            with self.scratch(nbytes=ws) as tmp:
                rc = self._L.ig_fft_exec(plan, ctypes.c_void_p(x._arr), ctypes.c_void_p(y._arr), direction,
                                         ctypes.c_void_p(tmp._arr))
        else:
            rc = self._L.ig_fft_exec(plan, ctypes.c_void_p(x._arr), ctypes.c_void_p(y._arr), direction, None)
        self._check(rc, "ig_fft_exec")

    # fused zero-pad / crop transforms (operators.ZpadFFT)
    PADDED_AXES_POW2 = (256, 512)

    def support_words(self, n):
        """(zw_in, zw_out) of an axis the library has a zero-pad-aware z pass for (ig_fft_support_words: 256 and 512 through the
        power-of-two kernel, every length 128 ... 640 with factors 2, 3, 5, 7 that splits as A x B, A, B <= 32, through the A x B
        kernel, lengths with a larger prime factor through the chirp-z kernel over an A x B length), else None."""
        if not self.tuning.get('support_chirp', True) and self.padded_axis_kind(n) == 5:
            return None          # (lab switch: a chirp-z grid without its table, as in round 4)
        zi, zo = ctypes.c_int(), ctypes.c_int()
        if self._L.ig_fft_support_words(int(n), ctypes.byref(zi), ctypes.byref(zo)) != 0:
            return None
        return zi.value, zo.value

    def padded_axis_kind(self, n):
        """3 = power-of-two kernel, 4 = A x B kernel, 5 = chirp-z over an A x B length, 0 = no zero-pad-aware pass (ig_fft_padded_axis_kind)"""
        k = ctypes.c_int(0)
        self._check(self._L.ig_fft_padded_axis_kind(int(n), ctypes.byref(k)), "ig_fft_padded_axis_kind")
        return k.value

    def supports_padded_fft(self, grid, ncoils=None):
        """256- and 512-point axes in every grid layout; the reference driver's own oversampled grids (320 ... 640,
        examples/pics.py:87-90), every other smooth length from 128 to 640 and chirp-z y / z axes in the coil-interleaved layout --
        for ANY coil count: indigo_amd.fused.plan_chunks cuts it into interleaved chunks of 8, 4 and 2 coils, the last one
        padded with zero-weight coils where the count does not divide"""
        if len(grid) != 3:
            return False
        if all(int(n) in self.PADDED_AXES_POW2 for n in grid):
            return True
        kinds = [self.padded_axis_kind(n) for n in grid]
        # (5 = chirp-z: lengths with a prime factor above 7 -- 277, 410: int(N * osf) of the reference's driver -- on the y and z axes)
        return kinds[0] in (3, 4) and all(k in (3, 4, 5) for k in kinds[1:])

    def fold_axis_shifts(self, grid, phases):
        """Which axes' modulation the zero-padded / cropped transform of `grid` can carry itself (round 6).  `phases`: per-axis arrays
        ph with the k-space modulation exp(2 pi i (ph_x[kx] + ph_y[ky] + ph_z[kz])) that the gridding matrix would otherwise hold
        (Backend.fftc_mod, indigo/backends/backend.py:352-366).  On an ODD y or z axis that is linear in k with slope c / n, c = n // 2:
        a circular shift by c on the image side, which a chirp-z axis takes into its tables for nothing (ig_fft_set_axis_shift).
        Returns (shifts, phases') -- the shift per axis and the phases with those axes' terms replaced by their constant, so that
        the matrix built from phases' has real weights times one complex constant -- or (None, None) when no axis qualifies."""
        if not self.tuning.get('fold_odd_axes', True) or phases is None or len(grid) != 3:
            return None, None
        shifts, out = [0, 0, 0], [np.asarray(ph, dtype=np.float64) for ph in phases]
        for a in (1, 2):
            n = int(grid[a])
            ph = out[a]
            kind = ctypes.c_int(0)
            if n % 2 == 0 or ph.size != n or self._L.ig_fft_padded_axis_kind(n, ctypes.byref(kind)) != 0 or kind.value != 5:
                continue
            slope = (ph[1:] - ph[:-1]) * n                    # c for a linear phase (any constant offset)
            c = int(round(float(slope[0]))) % n
            lin = ph[0] + np.arange(n) * (c / n)
            d = (ph - lin)
            if c == 0 or np.abs(d - np.round(d)).max() > 1e-9:          # (whole turns do not matter)
                continue
            shifts[a] = c
            out[a] = np.full(n, ph[0])
        return (tuple(shifts), out) if any(shifts) else (None, None)

    def split_gridding_constant(self, phases):
        """(g, phases') with exp(2 pi i sum phases) = g * exp(2 pi i sum phases'), |g| = 1, and exp(2 pi i phases'_d[k]) = +-1 on every
        axis -- when the modulation is a sign per axis times a constant (every even axis of a centred transform: the constant is 1 for
        lengths divisible by four, -+i otherwise; a folded odd axis: its constant phase), and the constant is not 1.  The fused leaf then
        builds its gridding matrix from phases' -- REAL weights: 8-byte entries, 4-byte gather values, records with gconst = 1 -- and
        multiplies g into the per-voxel weights of the transform to its right, which are complex anyway.  (1, None) otherwise."""
        from indigo_amd.interp import _axis_signs
        if not self.tuning.get('real_gridding', True) or phases is None:
            return 1.0, None
        g, out = 1.0 + 0.0j, []
        for ph in phases:
            gs = _axis_signs(ph)
            if gs is None:
                return 1.0, None
            g *= gs[0]
            ph = np.asarray(ph, dtype=np.float64)
            out.append(ph - ph[0])
        if abs(g - 1.0) < 1e-12:
            return 1.0, None
        return complex(g), out

    def supports_single_coil_layout(self, grid):
        """the per-coil grid layouts (one coil per panel column: a left-over single coil runs without a padding coil) exist for
        power-of-two grids only"""
        return len(grid) == 3 and all(int(n) in self.PADDED_AXES_POW2 for n in grid)

    supports_support_tile = True          # ZpadFFT / the brick scatter take support tables of 8 or 4 kx points per entry

    def _padded_plan(self, grid, box_lo, box_dims, batch, layout=0, support_tile=16, kshift=None):
        kshift = tuple(int(v) for v in kshift) if kshift is not None else (0, 0, 0)
        key = ('padded', tuple(grid), tuple(box_lo), tuple(box_dims), int(batch), int(layout), int(support_tile), kshift)
        if key not in self._plans:
            a3 = ctypes.c_int64 * 3
            plan, ws = ctypes.c_void_p(), ctypes.c_size_t()
            self._check(self._L.ig_fft_plan_padded(self._ctx, a3(*grid), a3(*box_lo), a3(*box_dims), int(batch),
                                                   int(layout), ctypes.byref(plan), ctypes.byref(ws)),
                        "ig_fft_plan_padded%s" % (key,))
            if int(support_tile) != 16:
                self._check(self._L.ig_fft_set_support_tile(plan, int(support_tile)), "ig_fft_set_support_tile")
            for axis, c in enumerate(kshift):          # (round 6) the centred transform's modulation of an odd chirp-z axis, carried by its passes
                if c:
                    self._check(self._L.ig_fft_set_axis_shift(plan, axis, int(c)), "ig_fft_set_axis_shift(axis %d, %d)" % (axis, c))
            self._plans[key] = (plan, ws.value)
        return self._plans[key]

    def _fft_padded_workspace(self, grid, box_lo, box_dims, batch, layout=0):
        return self._padded_plan(grid, box_lo, box_dims, batch, layout)[1]

    def fft_padded(self, y, x, w, grid, box_lo, box_dims, workspace=None, layout=0, support=None, support_tile=16, kshift=None):
        C = y.shape[1]
        assert y.dtype == _C64 and x.dtype == _C64 and y.contiguous and x.contiguous
        assert y.shape[0] == int(np.prod(grid)) and x.size == int(np.prod(box_dims))
        assert w is None or (w.contiguous and w.size == x.size * C)
        plan, ws = self._padded_plan(grid, box_lo, box_dims, C, layout, support_tile, kshift)
        assert layout == 0 or (workspace is not None and workspace.nbytes >= ws)
        self._check(self._L.ig_fft_exec_padded(plan, ctypes.c_void_p(x._arr), 0,
                                               ctypes.c_void_p(w._arr) if w is not None else None,
                                               ctypes.c_void_p(y._arr),
                                               ctypes.c_void_p(workspace._arr) if workspace is not None else None,
                                               ctypes.c_void_p(support._arr) if support is not None else None),
                    "ig_fft_exec_padded")

    def ifft_cropped(self, xc, y, w, grid, box_lo, box_dims, workspace, layout=0, support=None, support_tile=16, slab=None, kshift=None):
        """xc[:, c] = conj(w[:, c]) * crop(IFFT(y[:, c])).  slab (grid layout 1 only): 'z' = only the z pass; (z0, z1) = the y and x
        passes of the image planes z0..z1-1 (after one 'z' call) -- the one-coil ranks of a coil-sharded run all-reduce finished
        slabs while later ones are transformed"""
        C = y.shape[1]
        assert y.dtype == _C64 and xc.dtype == _C64 and y.contiguous and xc.contiguous
        assert xc.shape == (int(np.prod(box_dims)), C) and (layout != 2 or xc.contiguous)
        plan, ws = self._padded_plan(grid, box_lo, box_dims, C, layout, support_tile, kshift)
        assert workspace.nbytes >= ws
        if slab is not None:
            assert layout == 1
            phase, z0, z1 = (0, 0, 0) if slab == 'z' else (1, int(slab[0]), int(slab[1]))
            self._check(self._L.ig_fft_exec_cropped_slab(plan, ctypes.c_void_p(y._arr), ctypes.c_void_p(w._arr) if w is not None else None,
                                                         ctypes.c_void_p(xc._arr), xc.shape[0], ctypes.c_void_p(workspace._arr),
                                                         ctypes.c_void_p(support._arr) if support is not None else None, phase, z0, z1),
                        "ig_fft_exec_cropped_slab")
            return
        self._check(self._L.ig_fft_exec_cropped(plan, ctypes.c_void_p(y._arr),
                                                ctypes.c_void_p(w._arr) if w is not None else None,
                                                ctypes.c_void_p(xc._arr), xc.shape[0],
                                                ctypes.c_void_p(workspace._arr),
                                                ctypes.c_void_p(support._arr) if support is not None else None),
                    "ig_fft_exec_cropped")

    def ifft_cropped_sum(self, x, y, w, grid, box_lo, box_dims, workspace, support=None, slab=None, support_tile=16, kshift=None):
        """x = sum_c conj(w[:, c]) * crop(IFFT(y[:, c])) for a coil-interleaved grid panel y (layout 2): the cropped
        transform with the coil combination folded into its last pass.
        slab: None = everything; 'z' = only the z pass; (z0, z1) = the y and x passes of the image planes z0..z1-1
        (after one 'z' call; lets a multi-GPU caller all-reduce finished slabs while later ones are transformed)"""
        C = y.shape[1]
        assert y.dtype == _C64 and x.dtype == _C64 and y.contiguous and x.contiguous and w is not None
        assert x.size == int(np.prod(box_dims))
        plan, ws = self._padded_plan(grid, box_lo, box_dims, C, 2, support_tile, kshift)
        assert workspace.nbytes >= ws
        sup = ctypes.c_void_p(support._arr) if support is not None else None
        if slab is None:
            self._check(self._L.ig_fft_exec_cropped_sum(plan, ctypes.c_void_p(y._arr), ctypes.c_void_p(w._arr),
                                                        ctypes.c_void_p(x._arr), ctypes.c_void_p(workspace._arr), sup),
                        "ig_fft_exec_cropped_sum")
            return
        phase, z0, z1 = (0, 0, 0) if slab == 'z' else (1, int(slab[0]), int(slab[1]))
        self._check(self._L.ig_fft_exec_cropped_sum_slab(plan, ctypes.c_void_p(y._arr), ctypes.c_void_p(w._arr),
                                                         ctypes.c_void_p(x._arr), ctypes.c_void_p(workspace._arr), sup,
                                                         phase, z0, z1), "ig_fft_exec_cropped_sum_slab")

    def sum_columns(self, y, X, alpha=1, beta=0, interleaved=False):
        assert y.dtype == _C64 and X.dtype == _C64 and y.contiguous and y.size == X.shape[0]
        ar, ai = _cplx(alpha)
        br, bi = _cplx(beta)
        if interleaved:
            assert X.contiguous
            self._check(self._L.ig_csum_il(self._ctx, X.shape[0], X.shape[1], ctypes.c_void_p(X._arr),
                                           ar, ai, br, bi, ctypes.c_void_p(y._arr)), "ig_csum_il")
            return
        self._check(self._L.ig_csum_cols(self._ctx, X.shape[0], X.shape[1], ctypes.c_void_p(X._arr), X._leading_dim,
                                         ar, ai, br, bi, ctypes.c_void_p(y._arr)), "ig_csum_cols")

    def fftn(self, y, x):
        self._fft(y, x, -1)

    def ifftn(self, y, x):
        self._fft(y, x, +1)

    # -- SpMM -------------------------------------------------------------------------------
    def ccsrmm(self, y, A_shape, A_indx, A_ptr, A_vals, x, alpha=1, beta=0, adjoint=False, exwrite=False):
        m, k = A_shape
        n = x.shape[1]
        ar, ai = _cplx(alpha)
        br, bi = _cplx(beta)
        rc = self._L.ig_ccsrmm(self._ctx, 1 if adjoint else 0, 1 if exwrite else 0, m, k, n, A_vals.size,
                               ar, ai, ctypes.c_void_p(A_vals._arr), ctypes.c_void_p(A_indx._arr),
                               ctypes.c_void_p(A_ptr._arr),
                               ctypes.c_void_p(x._arr), x._leading_dim, br, bi,
                               ctypes.c_void_p(y._arr), y._leading_dim)
        self._check(rc, "ig_ccsrmm")

    def ccsrmm_t(self, y, A_shape, At_indx, At_ptr, At_vals, x, alpha=1, beta=0, support=None, xperm=None):
        """y = alpha * A^H x + beta*y through the CSR of A^T (gather); `support` = (table, n0, nm) restricts
        the output rows to a grid support region (rows outside are left untouched)"""
        m, k = A_shape
        ar, ai = _cplx(alpha)
        br, bi = _cplx(beta)
        if support is not None or xperm is not None:
            tab, n0, nm = support if support is not None else (None, 0, 0)
            rc = self._L.ig_ccsrmm_t_grid(self._ctx, m, k, x.shape[1], At_vals.size,
                                          ar, ai, ctypes.c_void_p(At_vals._arr), ctypes.c_void_p(At_indx._arr),
                                          ctypes.c_void_p(At_ptr._arr),
                                          ctypes.c_void_p(x._arr), x._leading_dim, br, bi,
                                          ctypes.c_void_p(y._arr), y._leading_dim,
                                          ctypes.c_void_p(tab._arr) if tab is not None else None, n0, nm,
                                          ctypes.c_void_p(xperm._arr) if xperm is not None else None)
            self._check(rc, "ig_ccsrmm_t_grid")
            return
        rc = self._L.ig_ccsrmm_t(self._ctx, m, k, x.shape[1], At_vals.size,
                                 ar, ai, ctypes.c_void_p(At_vals._arr), ctypes.c_void_p(At_indx._arr),
                                 ctypes.c_void_p(At_ptr._arr),
                                 ctypes.c_void_p(x._arr), x._leading_dim, br, bi,
                                 ctypes.c_void_p(y._arr), y._leading_dim)
        self._check(rc, "ig_ccsrmm_t")

    # -- ones / DIA / dense (outside the SENSE tree) ---------------------------------
    def onemm(self, y, x, alpha=1, beta=0):
        """y = beta*y + alpha * ones(M, K) * x"""
        assert x.dtype == _C64 and y.dtype == _C64 and x.shape[1] == y.shape[1]
        ar, ai = _cplx(alpha)
        br, bi = _cplx(beta)
        self._check(self._L.ig_conemm(self._ctx, y.shape[0], x.shape[0], x.shape[1], ar, ai, ctypes.c_void_p(x._arr), x._leading_dim,
                                      br, bi, ctypes.c_void_p(y._arr), y._leading_dim), "ig_conemm")

    def cdiamm(self, y, shape, offsets, data, x, alpha=1.0, beta=0.0, adjoint=True):
        assert x.dtype == _C64 and y.dtype == _C64 and data.dtype == _C64 and offsets.dtype == np.int32
        m, k = shape
        ar, ai = _cplx(alpha)
        br, bi = _cplx(beta)
        self._check(self._L.ig_cdiamm(self._ctx, 1 if adjoint else 0, m, k, x.shape[1], offsets.size, ctypes.c_void_p(offsets._arr),
                                      ctypes.c_void_p(data._arr), data._leading_dim, ar, ai, ctypes.c_void_p(x._arr), x._leading_dim,
                                      br, bi, ctypes.c_void_p(y._arr), y._leading_dim), "ig_cdiamm")

    def cgemm(self, y, M, x, alpha=1, beta=0, forward=True, left=True):
        """y = beta*y + alpha * op(M) * x   (left)   or   beta*y + alpha * x * op(M)   (not left); op = identity / ^H"""
        assert x.dtype == _C64 and y.dtype == _C64 and M.dtype == _C64 and M.ndim == 2
        ar, ai = _cplx(alpha)
        br, bi = _cplx(beta)
        r, c = M.shape if forward else M.shape[::-1]
        if left:
            x2, y2 = x.reshape((c, -1)), y.reshape((r, -1))
            p = x2.shape[1]
        else:
            x2, y2 = x.reshape((-1, r)), y.reshape((-1, c))
            p = x2.shape[0]
        self._check(self._L.ig_cgemm(self._ctx, 0 if forward else 1, 0 if left else 1, M.shape[0], M.shape[1], p, ar, ai,
                                     ctypes.c_void_p(M._arr), M._leading_dim, ctypes.c_void_p(x2._arr), x2._leading_dim,
                                     br, bi, ctypes.c_void_p(y2._arr), y2._leading_dim), "ig_cgemm")

    def csymm(self, y, M, x, alpha, beta, left=True):
        """the same product for a real symmetric M (the reference's cublasCsymm call, cuda.py:362-392)"""
        self.cgemm(y, M, x, alpha, beta, forward=True, left=left)

    # -- gridding matrices from their description (indigo_amd.structured.InterpS): the library's native host builder ------------
    def _interp_matrix(self, npts, N, width, table, coord, dtype):
        """Backend.Interp's matrix through ig_interp3_count / _fill (bit-identical to the numpy formulation, tests/test_sense_cpu.py)"""
        import scipy.sparse as spp
        from indigo_amd.interp import interp_csr_arrays
        indptr, indices, data = interp_csr_arrays(npts, N, width, table, coord, dtype=np.float32)
        return spp.csr_matrix((data.astype(dtype), indices, indptr), shape=(npts, int(np.prod(N, dtype=np.int64))))

    def gridding_from_struct(self, s, grid_order=0, phases=None):
        """G' = interp * diag(exp(2 pi i separable phase)) * real constant (what `pics.py -O3` folds into the gridding matrix,
        examples/pics.py:104-177) in ONE native pass over the trajectory, columns numbered for the fused leaf's grid order
        (ig_interp3_fill_modulated) -- instead of a scipy product of a 5e7-nonzero matrix with two 1.3e8-entry diagonals and a
        renumbering sort.  None for any other column scaling: the caller takes the scipy route.  `phases`: per-axis phases to use instead of
        the description's (fold_axis_shifts: the leaf's transform carries the rest)."""
        import scipy.sparse as spp
        from indigo_amd.interp import interp_csr_arrays, interp_csr_modulated
        shape = (s.npts, int(np.prod(s.N, dtype=np.int64)))
        if s.colscale is None:
            indptr, indices, data = interp_csr_arrays(s.npts, s.N, s.width, s.table, s.coord, dtype=np.float32, grid_order=grid_order)
            return spp.csr_matrix((data.astype(_C64), indices, indptr), shape=shape)
        sep = s.colscale.separable()
        if sep is None or tuple(sep[0].shape) != tuple(s.N):
            return None
        indptr, indices, data = interp_csr_modulated(s.npts, s.N, s.width, s.table, s.coord, sep[0].phases if phases is None else phases, sep[1], grid_order=grid_order)
        return spp.csr_matrix((data, indices, indptr), shape=shape)

    def gridding_sep_from_struct(self, s, grid_order=0, phases=None):
        """the same G' in SEPARABLE form -- one record per sample (indigo_amd.interp.interp_sep_records) -- or None when the column
        scaling is no sign per axis times a real constant (an odd grid axis) or the kernel is wider than 8 taps"""
        from indigo_amd.interp import interp_sep_records
        if not self.tuning.get('separable', True):
            return None
        if s.colscale is None:
            return interp_sep_records(s.npts, s.N, s.width, s.table, s.coord, None, 1.0, grid_order=grid_order)
        sep = s.colscale.separable()
        if sep is None or tuple(sep[0].shape) != tuple(s.N):
            return None
        return interp_sep_records(s.npts, s.N, s.width, s.table, s.coord, sep[0].phases if phases is None else phases, sep[1], grid_order=grid_order)

    def inspect(self, csr):
        indptr = np.ascontiguousarray(csr.indptr, dtype=np.int32)
        indices = np.ascontiguousarray(csr.indices, dtype=np.int32)
        nzrow, nzcol, exw = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int()
        rc = self._L.ig_csr_inspect(ctypes.c_void_p(indptr.ctypes.data), ctypes.c_void_p(indices.ctypes.data),
                                    csr.shape[0], csr.shape[1], ctypes.byref(nzrow), ctypes.byref(nzcol),
                                    ctypes.byref(exw))
        _lib.check(rc, None, "ig_csr_inspect")
        return nzrow.value, nzcol.value, bool(exw.value)

    def csr_transpose(self, csr):
        """(indptr_t, indices_t, data_t) of csr^T via the library's native counting sort"""
        M, K = csr.shape
        indptr = np.ascontiguousarray(csr.indptr, dtype=np.int32)
        indices = np.ascontiguousarray(csr.indices, dtype=np.int32)
        data = np.ascontiguousarray(csr.data, dtype=np.complex64)
        pt = np.empty(K + 1, dtype=np.int32)
        it = np.empty(csr.nnz, dtype=np.int32)
        dt = np.empty(csr.nnz, dtype=np.complex64)
        rc = self._L.ig_csr_transpose(M, K, csr.nnz, ctypes.c_void_p(indptr.ctypes.data),
                                      ctypes.c_void_p(indices.ctypes.data), ctypes.c_void_p(data.ctypes.data),
                                      ctypes.c_void_p(pt.ctypes.data), ctypes.c_void_p(it.ctypes.data),
                                      ctypes.c_void_p(dt.ctypes.data))
        _lib.check(rc, None, "ig_csr_transpose")
        return pt, it, dt

    class csr_matrix(Backend.csr_matrix):
        """Device CSR with an optional cached CSR of the transpose for gather-form adjoints."""

        def __init__(self, backend, A, name='mat'):
            super().__init__(backend, A, name)
            self._host_csr = A if A.dtype == _C64 else A.astype(_C64)   # kept until the transpose is built
            self._t = None

        def _transposed(self):
            if self._t is None:
                b = self._backend
                pt, it, dt = b.csr_transpose(self._host_csr)
                self._t = (b.copy_array(pt, name=self._name + ".T.rowPtrs"),
                           b.copy_array(it, name=self._name + ".T.colInds"),
                           b.copy_array(dt, name=self._name + ".T.data"))
                self._host_csr = None
            return self._t

        def set_grid_support(self, table, n0, nm, zw=16):
            """zw: words per entry of the table's bitmaps (the input-side form, ig_grid_support); 16 for 256- / 512-point nm"""
            self._support = (self._backend.copy_array(np.ascontiguousarray(table, dtype=np.int16).reshape(-1),
                                                      name=self._name + ".support"), int(n0), int(nm))
            self._support_zw = int(zw)
            self._support_host = table

        def set_grid_support_fine(self, table, tile, ncols=None):
            """a support table with `tile` (8 or 4) kx points per entry: what the brick scatter writes by (the gather routes keep
            the 16-point table of set_grid_support; a reader with the finer table reads a subset of what they write).
            ncols: the panel width whose adjoint writes by this table (a matrix shared by coil chunks of several widths carries
            one table per width); None = every width without a table of its own."""
            built = getattr(self, '_bricks_by', {})
            assert not (any(v is not None for v in built.values()) if ncols is None else built.get(int(ncols)) is not None), \
                "set_grid_support_fine must come before set_grid_bricks: the runs of bricks are sized for the table's segments"
            assert int(tile) in (4, 8, 16)
            self._support_fine = (self._backend.copy_array(np.ascontiguousarray(table, dtype=np.int16).reshape(-1),
                                                           name=self._name + ".supportFine"), int(tile))
            self.__dict__.setdefault('_support_fine_by', {})[None if ncols is None else int(ncols)] = self._support_fine
            self.__dict__.setdefault('_support_fine_host_by', {})[None if ncols is None else int(ncols)] = (table, int(tile))

        def _format(self, which, ncols, exact=False):
            """the binned format ('_bricks' / '_slots') or fine table ('_support_fine') registered for panels of `ncols` columns"""
            by = getattr(self, which + '_by', None)
            if by is None:
                return None
            key = None if ncols is None else int(ncols)
            if key in by or exact:
                return by.get(key)
            return by.get(None)

        def set_grid_bricks(self, n0, nm, ns, ncols=8, bm=2, bs=2, chunk=4096, run=4096):
            """Sort the nonzeros by the 16 x bm x bs brick of the n0 x nm x ns grid their column falls into (native host
            routine), padded so that a wave instruction (64/ncols entries x ncols panel columns) holds entries of one row
            only, and keep entries + brick table + task list on the device: the adjoint of an `ncols`-column interleaved
            panel then scatters brick by brick through LDS (ig_ccsrmm_t_bricks) and needs neither the transposed matrix nor
            its 4-bytes-per-grid-point row pointers.  A task (one wave) is a run of consecutive non-empty bricks of about
            `run` entries, or a piece of at most `chunk` entries of a heavy brick (more than `chunk` entries: shared)."""
            b = self._backend
            A = self._host_csr
            assert A is not None and A.shape[1] == n0 * nm * ns and ncols in (4, 8) and bm * bs <= 32
            unit = 64 // ncols
            chunk = max(unit, chunk // unit * unit)
            indptr = np.ascontiguousarray(A.indptr, dtype=np.int32)
            indices = np.ascontiguousarray(A.indices, dtype=np.int32)
            data = np.ascontiguousarray(A.data, dtype=_C64)
            nb = (n0 // 16) * (nm // bm) * (ns // bs)
            counts = np.zeros(nb, dtype=np.int32)
            if b._L.ig_grid_bricks_count(A.shape[0], indptr.ctypes.data, indices.ctypes.data, n0, nm, ns, bm, bs, unit,
                                         counts.ctypes.data) != 0:
                # e.g. a row that touches more than 64 bricks (very wide gridding kernels): the gather over the transpose serves it
                log.info("%s: no brick-binned format (%s); the adjoint keeps the gather route", self._name,
                         _lib.last_error(None) if hasattr(_lib, 'last_error') else "ig_grid_bricks_count failed")
                self._bricks = None
                self.__dict__.setdefault('_bricks_by', {})[int(ncols)] = None
                return
            ptr = np.zeros(nb + 1, dtype=np.int64)
            np.cumsum(counts, out=ptr[1:])
            assert ptr[-1] < 2**31, "brick entries are addressed with 32 bits"
            entries = np.empty((max(int(ptr[-1]), 1), 3), dtype=np.uint32)
            round_rows = np.empty(max(int(ptr[-1]) // unit, 1), dtype=np.uint32)
            _lib.check(b._L.ig_grid_bricks_fill(A.shape[0], indptr.ctypes.data, indices.ctypes.data, data.ctypes.data, n0, nm, ns,
                                                bm, bs, unit, ptr.ctypes.data, entries.ctypes.data, round_rows.ctypes.data),
                       None, "ig_grid_bricks_fill")
            fine = self._format('_support_fine', ncols)
            nseg = (16 // (fine[1] if fine is not None else 16)) * bm * bs          # segments per brick (the kernel looks up 512 per run)
            tasks, table, shared = brick_tasks(counts, ptr, chunk, run, max_bricks=min(64, 512 // nseg))
            # Real weights (a gridding matrix times the +-1 modulation of a centred transform on an even grid, whose imaginary parts
            # are the 1e-16 rounding residue of exp(i pi k)): 8-byte entries {cell, re}
            words = 3
            if self._real_weights(data):
                entries = np.ascontiguousarray(entries[:, :2])
                words = 2
            self._bricks = dict(n0=int(n0), nm=int(nm), bm=int(bm), bs=int(bs), ncols=int(ncols), ntasks=int(tasks.shape[0]), words=words,
                                nshared=int(shared.size), nentries=int(ptr[-1]),
                                tasks=b.copy_array(tasks.reshape(-1) if tasks.size else np.zeros(4, np.int32), name=self._name + ".brickTasks"),
                                table=b.copy_array(table.reshape(-1) if table.size else np.zeros(2, np.int32), name=self._name + ".brickTable"),
                                entries=b.copy_array(entries.reshape(-1), name=self._name + ".brickEntries"),
                                rounds=b.copy_array(round_rows, name=self._name + ".brickRoundRows"),
                                shared=b.copy_array(shared if shared.size else np.zeros(1, np.int32), name=self._name + ".sharedBricks"))
            self.__dict__.setdefault('_bricks_by', {})[int(ncols)] = self._bricks

        def set_grid_slots(self, n0, nm, ns, ncols=1, bm=2, bs=2, chunk=256, run=128):
            """The slot format of ig_ccsrmm_t_slots for an `ncols`-column panel (1, 2 or 4): the nonzeros binned by 16 x bm x bs
            bricks of the n0 x nm x ns grid WITHOUT padding (ig_grid_bricks_count / _fill, unit 1), reordered inside every brick so
            that a slot of at most 64 entries never holds a cell twice (ig_grid_slots_build), tasks = runs of about `run` slots
            of consecutive bricks, heavy bricks (more than `chunk` slots) cut into shared pieces."""
            b = self._backend
            A = self._host_csr
            assert A is not None and A.shape[1] == n0 * nm * ns and ncols in (1, 2, 4)
            indptr = np.ascontiguousarray(A.indptr, dtype=np.int32)
            indices = np.ascontiguousarray(A.indices, dtype=np.int32)
            data = np.ascontiguousarray(A.data, dtype=_C64)
            nb = (n0 // 16) * (nm // bm) * (ns // bs)
            counts = np.zeros(nb, dtype=np.int32)
            if b._L.ig_grid_bricks_count(A.shape[0], indptr.ctypes.data, indices.ctypes.data, n0, nm, ns, bm, bs, 1, counts.ctypes.data) != 0 \
                    or int(counts.sum(dtype=np.int64)) * 16 >= 2 ** 31:
                log.info("%s: no slot format; the adjoint keeps the gather route", self._name)
                self._slots = None
                self.__dict__.setdefault('_slots_by', {})[int(ncols)] = None
                return
            ptr = np.zeros(nb + 1, dtype=np.int64)
            np.cumsum(counts, out=ptr[1:])
            nent = int(ptr[-1])
            e12 = np.empty((max(nent, 1), 3), dtype=np.uint32)
            rows = np.empty(max(nent, 1), dtype=np.uint32)
            _lib.check(b._L.ig_grid_bricks_fill(A.shape[0], indptr.ctypes.data, indices.ctypes.data, data.ctypes.data, n0, nm, ns, bm, bs, 1,
                                                ptr.ctypes.data, e12.ctypes.data, rows.ctypes.data), None, "ig_grid_bricks_fill")
            e16 = np.empty((max(nent, 1), 4), dtype=np.uint32)
            brick_slots = np.zeros(nb, dtype=np.int32)
            slot_ptr = np.empty(nent + 1, dtype=np.int32)
            nslots = ctypes.c_int64()
            _lib.check(b._L.ig_grid_slots_build(nb, ptr.ctypes.data, e12.ctypes.data, rows.ctypes.data, 16 * bm * bs, e16.ctypes.data,
                                                brick_slots.ctypes.data, slot_ptr.ctypes.data, ctypes.byref(nslots)), None, "ig_grid_slots_build")
            del e12, rows
            sptr = np.zeros(nb + 1, dtype=np.int64)
            np.cumsum(brick_slots, out=sptr[1:])
            fine = self._format('_support_fine', ncols)
            nseg = (16 // (fine[1] if fine is not None else 16)) * bm * bs      # segments per brick (the kernel looks up 512 per run)
            tasks, table, shared = brick_tasks(brick_slots, sptr, chunk, run, max_bricks=min(64, 512 // nseg))
            words = 4
            if self._real_weights(data):
                e16 = np.ascontiguousarray(e16.reshape(-1, 4)[:, [0, 1, 3]])     # {cell, re, row}: 12 bytes per nonzero
                words = 3
            self._slots = dict(n0=int(n0), nm=int(nm), bm=int(bm), bs=int(bs), ncols=int(ncols), ntasks=int(tasks.shape[0]), words=words,
                               nshared=int(shared.size), nslots=int(nslots.value), nentries=nent,
                               tasks=b.copy_array(tasks.reshape(-1) if tasks.size else np.zeros(4, np.int32), name=self._name + ".slotTasks"),
                               table=b.copy_array(table.reshape(-1) if table.size else np.zeros(2, np.int32), name=self._name + ".slotTable"),
                               entries=b.copy_array(e16.reshape(-1), name=self._name + ".slotEntries"),
                               slot_ptr=b.copy_array(slot_ptr[:int(nslots.value) + 1].copy(), name=self._name + ".slotPtr"),
                               shared=b.copy_array(shared if shared.size else np.zeros(1, np.int32), name=self._name + ".slotSharedBricks"))
            self.__dict__.setdefault('_slots_by', {})[int(ncols)] = self._slots

        def set_grid_separable(self, sep):
            """The matrix in SEPARABLE form (indigo_amd.interp.interp_sep_records: one record per sample, columns numbered in the
            memory order of the coil-interleaved grid panel): the products with interleaved panels of 2, 4 or 8 columns compute their
            taps from the records (ig_grid_gather_sep / ig_grid_scatter_sep) instead of streaming the stored ones."""
            b = self._backend
            n0, nm, ns = (int(v) for v in sep['dims'])
            assert sep['records'].shape[0] == self.shape[0] and n0 * nm * ns == self.shape[1]
            rec = np.ascontiguousarray(sep['records'])
            # the forward reads the records as they are (16 or 32 words apart: one 64- or 128-byte line each); the share scatter wants every
            # record followed by room for the sample's panel row -- a second, wider copy that set_grid_shares uploads when it is needed
            rw = rec.shape[1]
            self._sep = dict(tw=int(sep['tw']), dims=(n0, nm, ns), gconst=complex(sep['gconst']), host=rec, stride=rw,
                             records=b.copy_array(rec.reshape(-1), name=self._name + ".sepRecords"))

        def _gather_order(self, ncols):
            """The order in which the workgroups of the record gather take their groups of consecutive samples (ig_grid_gather_sep's
            group_order) for an `ncols`-column panel: the groups sorted by the 32^3-cell block of the grid their first sample's first tap
            lies in, blocks in Morton order -- built once per panel width from the host copy of the records.  Measured
            (profiles/r06_gather_order.txt): 8 x the spokes 3.28 -> 2.79 ms, half-width 3 1.46 -> 1.36 ms, the headline 0.430 -> 0.416 ms."""
            sep = self._sep
            by = sep.setdefault('order', {})
            if ncols not in by:
                rec, tw = sep['host'], sep['tw']
                group = int(self._backend._L.ig_grid_gather_sep_group(int(ncols), int(tw)))
                by[ncols] = None
                if group > 0 and rec.shape[0] > 4 * group:
                    h0, h1 = rec[::group, 3 * tw], rec[::group, 3 * tw + 1]
                    j = [(h0 & 0xffff).astype(np.int64) >> 5, (h0 >> 16).astype(np.int64) >> 5, (h1 & 0xffff).astype(np.int64) >> 5]
                    key = np.zeros_like(j[0])
                    for bit in range(11):                       # axes of up to 65535 points: 11 bits of 32-cell blocks each
                        for a in range(3):
                            key |= ((j[a] >> bit) & 1) << (3 * bit + a)
                    by[ncols] = self._backend.copy_array(np.argsort(key, kind='stable').astype(np.uint32), name=self._name + ".gatherOrder%d" % ncols)
            return by[ncols]

        def set_grid_shares(self, ncols=8, bm=8, bs=2, chunk=1024, run=1024):
            """The adjoint of an `ncols`-column interleaved panel as a scatter of SHARES (ig_grid_scatter_sep): every (sample, brick of
            16 x bm x bs cells) pair the sample's footprint meets is one 8-byte share, binned by brick on the host (ig_grid_shares_count /
            _fill: brick order, sample order inside a brick); the taps come from the separable records (set_grid_separable, which must
            come first, as must set_grid_support_fine: a brick's flagged segments are looked up here, once).  A task (one wave) is a run
            of consecutive non-empty bricks of about `run` shares, or a piece of at most `chunk` shares of a heavy brick (shared)."""
            b = self._backend
            sep = getattr(self, '_sep', None)
            assert sep is not None and ncols in (4, 8)
            bm, bs = min(int(bm), 4), min(int(bs), 4)          # (the brick image is four MFMA blocks x four accumulator groups)
            n0, nm, ns = sep['dims']
            rec, tw = sep['host'], sep['tw']
            fine = self._format('_support_fine_host', ncols)
            sup = getattr(self, '_support_host', None)
            tab, tile, zw = (fine[0], fine[1], getattr(self, '_support_zw', 16)) if fine is not None else \
                            (sup, 16, getattr(self, '_support_zw', 16)) if sup is not None else (None, 16, 16)
            by = self.__dict__.setdefault('_shares_by', {})
            xs = 16 // tile
            while xs * bm * bs > 64 and bs > 1:
                bs //= 2
            while xs * bm * bs > 64 and bm > 1:
                bm //= 2
            # (bricks need not divide the middle and slow axes: the part of a last brick outside the grid is never flagged below)
            nbx, nbm, nbs = n0 // 16, -(-nm // bm), -(-ns // bs)
            nb = nbx * nbm * nbs
            counts = np.zeros(nb, dtype=np.int32)
            if n0 % 16 or b._L.ig_grid_shares_count(rec.shape[0], rec.ctypes.data, tw, n0, nm, ns, bm, bs, counts.ctypes.data) != 0:
                log.info("%s: no share format; the adjoint keeps the stored-tap routes", self._name)
                by[int(ncols)] = None
                return
            ptr = np.zeros(nb + 1, dtype=np.int64)
            np.cumsum(counts, out=ptr[1:])
            assert ptr[-1] < 2**31, "shares are addressed with 32 bits"
            shares = np.empty((max(int(ptr[-1]), 1), 2), dtype=np.uint32)
            _lib.check(b._L.ig_grid_shares_fill(rec.shape[0], rec.ctypes.data, tw, n0, nm, ns, bm, bs, ptr.ctypes.data, shares.ctypes.data),
                       None, "ig_grid_shares_fill")
            tasks, table, shared = brick_tasks(counts, ptr, int(chunk), int(run), max_bricks=64)
            # the flagged segments of every non-empty brick: bit xs + XS * (im + bm * is), from the support table's input-side bitmaps
            bricks = table[:, 0].astype(np.int64) if table.size else np.zeros(0, np.int64)
            mask = np.zeros(bricks.size, dtype=np.uint64)
            if bricks.size:
                bits = None
                if tab is not None:
                    nt = n0 // tile
                    tabh = np.ascontiguousarray(tab, dtype=np.int16).reshape(-1)
                    off = 2 * (ns * nt + nt)
                    bits = tabh[off:off + 2 * ns * nt * zw].view(np.uint32).reshape(ns * nt, zw)
                bx, bmi, bsi = bricks % nbx, (bricks // nbx) % nbm, bricks // (nbx * nbm)
                for is_ in range(bs):
                    for im in range(bm):
                        km, ks = bmi * bm + im, bsi * bs + is_
                        inside = (km < nm) & (ks < ns)                 # (a last brick may reach beyond the grid)
                        kmc, ksc = np.minimum(km, nm - 1), np.minimum(ks, ns - 1)
                        for x in range(xs):
                            bit = inside.astype(np.uint64) if bits is None else \
                                (((bits[ksc * nt + bx * xs + x, kmc % zw] >> (kmc // zw).astype(np.uint32)) & np.uint32(1)).astype(np.uint64) * inside)
                            mask |= bit << np.uint64(x + xs * (im + bm * is_))
            tab16 = np.empty((max(bricks.size, 1), 4), dtype=np.uint32)
            if bricks.size:
                tab16[:, 0:2] = table.astype(np.uint32)
                tab16[:, 2] = (mask & np.uint64(0xffffffff)).astype(np.uint32)
                tab16[:, 3] = (mask >> np.uint64(32)).astype(np.uint32)
            sh_rows = tab16[np.searchsorted(bricks, shared.astype(np.int64))] if shared.size else np.zeros((1, 4), np.uint32)
            if 'recx' not in sep:
                # (the MFMA scatter reads a share's record and panel row as ONE line: record, then the row k_sep_pack_recx writes there)
                rw = rec.shape[1]
                rs = 32 if rw == 16 else 64
                recx = np.zeros((rec.shape[0], rs), dtype=np.uint32)
                recx[:, :rw] = rec
                sep['recx'] = b.copy_array(recx.reshape(-1), name=self._name + ".sepRecordsWithRows")
                sep['stride_x'] = rs
            by[int(ncols)] = dict(bm=int(bm), bs=int(bs), tile=int(tile), ncols=int(ncols), ntasks=int(tasks.shape[0]), nshared=int(shared.size),
                                  nshares=int(ptr[-1]), nbricks=int(bricks.size),
                                  tasks=b.copy_array(tasks.reshape(-1) if tasks.size else np.zeros(4, np.int32), name=self._name + ".shareTasks"),
                                  table=b.copy_array(tab16.reshape(-1), name=self._name + ".shareTable"),
                                  shares=b.copy_array(shares.reshape(-1), name=self._name + ".shares"),
                                  shared=b.copy_array(np.ascontiguousarray(sh_rows).reshape(-1), name=self._name + ".shareSharedBricks"))

        def set_grid_dims(self, n0, nm, ns):
            """Hint: the columns of the matrix are the points of an n0 x nm x ns grid, n0 running fastest (a gridding matrix).
            The wide adjoint then bins by bricks of 16 x 2 x 2 points instead of 16 consecutive columns."""
            assert int(n0) * int(nm) * int(ns) == self.shape[1]
            self._grid_dims = (int(n0), int(nm), int(ns))
            self._wide = False

        def _guess_grid_dims(self):
            """(n, n, n) when the column count is a cube with n a multiple of 32, else None.  Only a grouping of the columns:
            a wrong guess costs speed, never correctness (rows that touch more than 64 bricks decline the format)."""
            k = self.shape[1]
            n = int(round(k ** (1.0 / 3.0)))
            return (n, n, n) if n > 0 and n ** 3 == k and n % 32 == 0 else None

        def _wide_bricks(self):
            """The matrix binned by bricks, 12-byte entries {column inside the brick, re, im} + their rows: the format of
            ig_ccsrmm_t_bricks_wide[_grid] (adjoint of a 64-column column-major panel as a scatter).  Bricks are 16 x bm x bs
            points of the grid the columns form (set_grid_dims, or a cube guessed from the column count; tuning
            'wide_brick_shape'), else 16 consecutive columns.  Built on first use; None when the matrix does not qualify
            (a row touching more than 64 bricks)."""
            wb = getattr(self, '_wide', False)
            if wb is not False:
                return wb
            b = self._backend
            m, k = self.shape
            if self._host_csr is not None:
                indptr, indices, data = self._host_csr.indptr, self._host_csr.indices, self._host_csr.data
            else:
                indptr, indices, data = self.rowPtrs.to_host(), self.colInds.to_host(), self.values.to_host()
            indptr = np.ascontiguousarray(indptr, dtype=np.int32)
            indices = np.ascontiguousarray(indices, dtype=np.int32)
            data = np.ascontiguousarray(data, dtype=_C64)
            dims = getattr(self, '_grid_dims', None) or self._guess_grid_dims()
            bm, bs = b.tuning.get('wide_brick_shape', (2, 2))
            geoms = []
            if dims is not None and bm * bs > 1 and dims[0] % 16 == 0 and dims[1] % bm == 0 and dims[2] % bs == 0:
                geoms.append((dims[0], dims[1], dims[2], bm, bs))
            geoms.append((k, 1, 1, 1, 1))
            for n0, nm, ns, bm, bs in geoms:
                nbx, nbm = n0 // 16, nm // bm
                # grid bricks: a row's share of a brick padded to QUADS (one panel row is loaded per four entries)
                unit = 4 if bm * bs > 1 else 1
                counts = np.zeros(nbx * nbm * (ns // bs), dtype=np.int32)
                if b._L.ig_grid_bricks_count(m, indptr.ctypes.data, indices.ctypes.data, n0, nm, ns, bm, bs, unit, counts.ctypes.data) == 0:
                    # a matrix whose rows do not cluster on the (guessed) grid would be mostly padding in quads: keep 16-row bricks
                    if unit == 1 or int(counts.sum(dtype=np.int64)) <= 1.6 * max(int(indptr[-1] - indptr[0]), 1):
                        break
            else:
                self._wide = None
                return None
            ptr = np.zeros(counts.size + 1, dtype=np.int64)
            np.cumsum(counts, out=ptr[1:])
            e12 = np.empty((max(int(ptr[-1]), 1), 3), dtype=np.uint32)
            rows = np.empty(max(int(ptr[-1]) // unit, 1), dtype=np.uint32)
            _lib.check(b._L.ig_grid_bricks_fill(m, indptr.ctypes.data, indices.ctypes.data, data.ctypes.data, n0, nm, ns, bm, bs, unit,
                                                ptr.ctypes.data, e12.ctypes.data, rows.ctypes.data), None, "ig_grid_bricks_fill")
            if unit > 1 and ptr[-1] > 0:
                # padding entries (weight zero) take the cell of the real entry before them: whatever the panel row holds
                # (an infinity times zero) lands on a cell the sample touches anyway
                cell = e12[:int(ptr[-1]), 0]
                last_real = np.maximum.accumulate(np.where(cell != 0xffffffff, np.arange(cell.size), 0))
                cell[:] = cell[last_real]
            # tasks: pieces of at most 4096 entries of a heavy brick, runs of about 1024 entries of consecutive bricks, longest first.
            # Measured on BASELINE config 3 and rejected (round 3, profiles/r03_cfg3_sweep_*.txt): bricks in index order or in a
            # (y, z)-blocked order of the grid, with chunks of 4..64 consecutive workgroups dealt to one XCD so that the bricks that
            # need the same rows of X meet behind one L2 -- 3.7..4.2 ms against 3.35 ms, and the same 9.7 GB of re-fetched rows by
            # the PMC counters: a brick takes a wave ~25 us, a line lives ~7 us in a 4 MB L2 that 0.5 TB/s stream through.
            chunk, run = (max(4, int(v) // 4 * 4) for v in b.tuning.get('wide_task_shape', (8192, 2048)))
            tasks, table, shared = brick_tasks(counts, ptr, chunk, run, max_bricks=64, longest_first=True)
            # tiles (16 rows of the result) some task stores in full: those of the non-empty bricks that are not cut into shared pieces
            owned_b = np.zeros(counts.size, dtype=bool)
            owned_b[table[:, 0]] = True
            owned_b[shared] = False
            ob = np.flatnonzero(owned_b).astype(np.int64)
            bx, bmi, bsi = ob % nbx, (ob // nbx) % nbm, ob // (nbx * nbm)
            owned = np.zeros(k // 16, dtype=bool)
            for im in range(bm):
                for is_ in range(bs):
                    owned[bx + nbx * ((bmi * bm + im) + nm * (bsi * bs + is_))] = True
            bits = np.packbits(np.concatenate([owned, np.zeros((-owned.size) % 32, dtype=bool)]), bitorder='little').view(np.uint32)
            words = 3
            if bm * bs > 1 and self._real_weights(data):
                e12 = np.ascontiguousarray(e12[:, :2])            # {cell, re}: the register-image kernel's real-weight form
                words = 2
            self._wide = dict(ntasks=int(tasks.shape[0]), geom=(n0, nm, bm, bs), words=words,
                              owned=b.copy_array(bits, name=self._name + ".wideOwnedTiles"),
                              tasks=b.copy_array(tasks.reshape(-1) if tasks.size else np.zeros(4, np.int32), name=self._name + ".wideTasks"),
                              table=b.copy_array(table.reshape(-1) if table.size else np.zeros(2, np.int32), name=self._name + ".wideTable"),
                              entries=b.copy_array(e12.reshape(-1), name=self._name + ".wideEntries"),
                              rows=b.copy_array(rows, name=self._name + ".wideEntryRows"))
            return self._wide

        def _runs(self, sub):
            """The run format of the matrix over its touched columns (ig_csr_runs_build; `sub` = the cached (touched columns,
            compact column indices) pair of the xrows route): built on first use, None when the matrix does not qualify."""
            r = getattr(self, '_runs_fmt', False)
            if r is not False:
                return r
            b = self._backend
            m = self.shape[0]
            indptr = np.ascontiguousarray(self._host_csr.indptr if self._host_csr is not None else self.rowPtrs.to_host(), dtype=np.int32)
            compact = np.ascontiguousarray(sub[1].to_host(), dtype=np.int32)
            data = np.ascontiguousarray(self._host_csr.data if self._host_csr is not None else self.values.to_host(), dtype=_C64)
            nruns = (m + 15) // 16
            dptr = np.zeros(nruns + 1, dtype=np.int32)
            K = int(sub[0].size)
            if indptr[0] != 0 or b._L.ig_csr_runs_build(m, K, indptr.ctypes.data, compact.ctypes.data, data.ctypes.data, dptr.ctypes.data, None, None, None) != 0:
                log.info("%s: no run format (%s); the forward product keeps the per-nonzero gather", self._name, _lib.last_error(None))
                self._runs_fmt = None
                return None
            dcols = np.empty(max(int(dptr[-1]), 1), dtype=np.uint32)
            entries = np.empty((max(int(indptr[-1]), 1), 3), dtype=np.uint32)
            real = ctypes.c_int(0)
            _lib.check(b._L.ig_csr_runs_build(m, K, indptr.ctypes.data, compact.ctypes.data, data.ctypes.data, dptr.ctypes.data,
                                              dcols.ctypes.data, entries.ctypes.data, ctypes.byref(real)), None, "ig_csr_runs_build")
            # order of the runs: by the 16 x 16 x 16 brick of the grid (set_grid_dims, or a cube guessed from the column count) their
            # first nonzero falls into -- runs that are neighbours in space share panel rows and then meet behind one L2.  Only a
            # grouping: any order gives the same product.
            order = None
            dims = getattr(self, '_grid_dims', None) or self._guess_grid_dims()
            if dims is not None and b.tuning.get('runs_order', True) and nruns > 1:
                touched = sub[0].to_host().astype(np.int64)
                first = np.minimum(indptr[np.minimum(np.arange(nruns, dtype=np.int64) * 16, m - 1)], max(int(indptr[-1]) - 1, 0))
                col = touched[compact[first]] if indptr[-1] > 0 else np.zeros(nruns, np.int64)
                n0, nm, ns = dims
                bx, bm_, bs_ = (col % n0) // 16, ((col // n0) % nm) // 16, (col // (n0 * nm)) // 16
                key = bx + ((n0 + 15) // 16) * (bm_ + ((nm + 15) // 16) * bs_)
                order = b.copy_array(np.argsort(key, kind='stable').astype(np.int32), name=self._name + ".runOrder")
            self._runs_fmt = dict(dptr=b.copy_array(dptr, name=self._name + ".runPtr"), dcols=b.copy_array(dcols, name=self._name + ".runCols"),
                                  entries=b.copy_array(entries.reshape(-1), name=self._name + ".runEntries"), all_real=int(real.value),
                                  ndistinct=int(dptr[-1]), order=order)
            return self._runs_fmt

        def set_row_order(self, perm):
            """Store the matrix with its rows in the order `perm` (stored row r = row perm[r] of A), e.g. gridding
            samples sorted by the grid cell they touch: neighbouring rows then gather neighbouring panel rows.
            The products are unchanged -- the forward result is written through the permutation and the
            adjoint reads its panel through it."""
            b = self._backend
            perm = np.ascontiguousarray(perm, dtype=np.int32)
            assert self._host_csr is not None and perm.shape == (self.shape[0],)
            Ap = self._host_csr[perm]
            Ap.sort_indices()
            self.rowPtrs = b.copy_array(Ap.indptr.astype(np.int32), name=self._name + ".rowPtrs")
            self.colInds = b.copy_array(Ap.indices.astype(np.int32), name=self._name + ".colInds")
            self.values = b.copy_array(Ap.data.astype(_C64), name=self._name + ".data")
            self._host_csr = Ap
            self._t = None
            self._values_re = False          # (built on first use from the reordered values)
            self._weights_real = None
            self._perm = b.copy_array(perm, name=self._name + ".rowOrder")

        def _real_weights(self, data):
            """are the matrix's weights real up to rounding residue (weights_are_real; one pass over the values, remembered) -- and
            does the backend's tuning allow the 4-byte forms?"""
            if not self._backend.tuning.get('real_entries', True):
                return False
            r = getattr(self, '_weights_real', None)
            if r is None:
                r = self._weights_real = weights_are_real(data)
            return r

        def _real_values(self):
            """the weights' real parts as a float32 device array when the matrix is real up to rounding residue (built on first
            use; None otherwise, or when the backend's tuning asks for complex entries)"""
            r = getattr(self, '_values_re', False)
            if r is False:
                r = None
                data = self._host_csr.data if self._host_csr is not None else self.values.to_host()
                if self._real_weights(data):
                    r = self._backend.copy_array(np.ascontiguousarray(data.real, dtype=np.float32), name=self._name + ".dataRe")
                self._values_re = r
            return r

        def forward(self, y, x, alpha=1, beta=0):
            perm = getattr(self, '_perm', None)
            if getattr(self, '_grid_il', False):
                assert perm is None and x.contiguous, "interleaved panels: no row order, contiguous grid panel"
                self._check_panels(y, x, self.values)
                b = self._backend
                ar, ai = _cplx(alpha)
                br, bi = _cplx(beta)
                m, k = self.shape
                sep = getattr(self, '_sep', None)
                if sep is not None and x.shape[1] in (2, 4, 8) and b.tuning.get('sep_gather', True):
                    # the taps computed from one 64-byte record per sample (ig_grid_gather_sep): no index or value stream
                    gr, gi = _cplx(complex(alpha) * sep['gconst'])
                    n0, nm, ns = sep['dims']
                    order = self._gather_order(x.shape[1]) if b.tuning.get('gather_order', True) else None
                    b._check(b._L.ig_grid_gather_sep(b._ctx, m, x.shape[1], sep['tw'], ctypes.c_void_p(sep['records']._arr), sep['stride'], ctypes.c_void_p(x._arr),
                                                     n0, nm, ns, gr, gi, br, bi, ctypes.c_void_p(y._arr), y._leading_dim,
                                                     ctypes.c_void_p(order._arr) if order is not None else None), "ig_grid_gather_sep")
                    return
                vre = self._real_values() if x.shape[1] in (2, 4, 8) else None
                if vre is not None:
                    # every weight real (see weights_are_real): the gather reads 4-byte values
                    b._check(b._L.ig_ccsrmm_il_rw(b._ctx, m, k, x.shape[1], self.values.size, ar, ai,
                                                  ctypes.c_void_p(self.values._arr), ctypes.c_void_p(vre._arr), ctypes.c_void_p(self.colInds._arr),
                                                  ctypes.c_void_p(self.rowPtrs._arr), ctypes.c_void_p(x._arr), br, bi,
                                                  ctypes.c_void_p(y._arr), y._leading_dim), "ig_ccsrmm_il_rw")
                    return
                b._check(b._L.ig_ccsrmm_il(b._ctx, m, k, x.shape[1], self.values.size, ar, ai,
                                           ctypes.c_void_p(self.values._arr), ctypes.c_void_p(self.colInds._arr),
                                           ctypes.c_void_p(self.rowPtrs._arr), ctypes.c_void_p(x._arr), br, bi,
                                           ctypes.c_void_p(y._arr), y._leading_dim), "ig_ccsrmm_il")
                return
            if perm is None and 16 <= x.shape[1] <= 64 and self._col_frac <= 0.6 \
                    and self.values.size >= self.shape[1] and self._backend.tuning['xrows']:
                # a wide panel of which the matrix touches a fraction of the rows (a gridding matrix: 30 % of its grid):
                # the panel is repacked row-major anyway -- repack only the touched rows (ig_ccsrmm_xrows)
                self._check_panels(y, x, self.values)
                b = self._backend
                sub = getattr(self, '_xrows', None)
                if sub is None:
                    indices = self._host_csr.indices if self._host_csr is not None else self.colInds.to_host()
                    mark = np.zeros(self.shape[1], dtype=bool)
                    mark[indices] = True
                    touched = np.flatnonzero(mark).astype(np.int32)
                    compact = np.searchsorted(touched, indices).astype(np.int32)
                    sub = self._xrows = (b.copy_array(touched, name=self._name + ".touchedCols"),
                                         b.copy_array(compact, name=self._name + ".compactColInds"))
                ar, ai = _cplx(alpha)
                br, bi = _cplx(beta)
                m, k = self.shape
                runs = self._runs(sub) if (x.shape[1] == 64 and b.tuning.get('runs', True) and sub[0].size * 512 < 2 ** 32) else None
                if runs is not None:
                    # 64 columns: the run format -- every panel row a run of 16 matrix rows touches is loaded once (ig_ccsrmm_xrows_runs)
                    b._check(b._L.ig_ccsrmm_xrows_runs(b._ctx, m, k, self.values.size, ar, ai, ctypes.c_void_p(self.rowPtrs._arr),
                                                       ctypes.c_void_p(runs['dptr']._arr), ctypes.c_void_p(runs['dcols']._arr),
                                                       ctypes.c_void_p(runs['entries']._arr), runs['all_real'],
                                                       ctypes.c_void_p(runs['order']._arr) if runs['order'] is not None else None,
                                                       ctypes.c_void_p(x._arr), x._leading_dim, br, bi, ctypes.c_void_p(y._arr), y._leading_dim,
                                                       ctypes.c_void_p(sub[0]._arr), sub[0].size), "ig_ccsrmm_xrows_runs")
                    return
                b._check(b._L.ig_ccsrmm_xrows(b._ctx, m, k, x.shape[1], self.values.size, ar, ai,
                                              ctypes.c_void_p(self.values._arr), ctypes.c_void_p(sub[1]._arr),
                                              ctypes.c_void_p(self.rowPtrs._arr), ctypes.c_void_p(x._arr), x._leading_dim,
                                              br, bi, ctypes.c_void_p(y._arr), y._leading_dim,
                                              ctypes.c_void_p(sub[0]._arr), sub[0].size), "ig_ccsrmm_xrows")
                return
            if perm is None:
                return super().forward(y, x, alpha=alpha, beta=beta)
            self._check_panels(y, x, self.values)
            b = self._backend
            ar, ai = _cplx(alpha)
            br, bi = _cplx(beta)
            m, k = self.shape
            b._check(b._L.ig_ccsrmm_rowperm(b._ctx, m, k, x.shape[1], self.values.size, ar, ai,
                                            ctypes.c_void_p(self.values._arr), ctypes.c_void_p(self.colInds._arr),
                                            ctypes.c_void_p(self.rowPtrs._arr),
                                            ctypes.c_void_p(x._arr), x._leading_dim, br, bi,
                                            ctypes.c_void_p(y._arr), y._leading_dim,
                                            ctypes.c_void_p(perm._arr)), "ig_ccsrmm_rowperm")

        def adjoint(self, y, x, alpha=1, beta=0):
            self._check_panels(y, x, self.values)
            b = self._backend
            sup = getattr(self, '_support', None)
            perm = getattr(self, '_perm', None)
            shf = self._format('_shares', x.shape[1], exact=True)
            if (shf is not None and perm is None and beta == 0 and y.contiguous and getattr(self, '_grid_il', False) and shf['ntasks'] > 0
                    and b.tuning.get('sep_scatter', True)):
                sep = self._sep
                if not (self._format('_support_fine', x.shape[1]) is not None or sup is not None):
                    y._zero()           # without a support table every row is defined: bricks no sample touches stay zero
                ar, ai = _cplx(complex(alpha) * np.conj(sep['gconst']))
                n0, nm, ns = sep['dims']
                b._check(b._L.ig_grid_scatter_sep(b._ctx, self.shape[0], x.shape[1], sep['tw'], ctypes.c_void_p(sep['recx']._arr), sep['stride_x'],
                                                  ctypes.c_void_p(shf['shares']._arr), ctypes.c_void_p(x._arr), x._leading_dim, ctypes.c_void_p(y._arr),
                                                  n0, nm, ns, shf['bm'], shf['bs'], ctypes.c_void_p(shf['tasks']._arr), shf['ntasks'],
                                                  ctypes.c_void_p(shf['table']._arr), ctypes.c_void_p(shf['shared']._arr), shf['nshared'], shf['tile'], ar, ai),
                         "ig_grid_scatter_sep")
                return
            br = self._format('_bricks', x.shape[1], exact=True)
            # (the gather routes over the transposed matrix read 16-word bitmaps only: with another table they compute every
            # row -- a superset of what any reader of the grid looks at)
            sup_gather = sup if getattr(self, '_support_zw', 16) == 16 else None
            if (br is not None and perm is None and beta == 0 and y.contiguous and getattr(self, '_grid_il', False)
                    and x.shape[1] == br['ncols']):
                fine = self._format('_support_fine', x.shape[1])
                tab, tile = (fine[0], fine[1]) if fine is not None else (sup[0] if sup is not None else None, 16)
                if tab is None:
                    y._zero()           # without a support table every row is defined: bricks no sample touches stay zero
                ar, ai = _cplx(alpha)
                m, k = self.shape
                b._check(b._L.ig_ccsrmm_t_bricks(b._ctx, m, k, x.shape[1], ar, ai,
                                                 ctypes.c_void_p(br['entries']._arr), ctypes.c_void_p(br['rounds']._arr), ctypes.c_void_p(x._arr), x._leading_dim,
                                                 ctypes.c_void_p(y._arr), ctypes.c_void_p(tab._arr) if tab is not None else None,
                                                 br['n0'], br['nm'], br['bm'], br['bs'], ctypes.c_void_p(br['tasks']._arr), br['ntasks'],
                                                 ctypes.c_void_p(br['table']._arr), ctypes.c_void_p(br['shared']._arr), br['nshared'], tile,
                                                 getattr(self, '_support_zw', 16), br['words']),
                         "ig_ccsrmm_t_bricks")
                return
            sl = self._format('_slots', x.shape[1], exact=True)
            if sl is not None and perm is None and beta == 0 and y.contiguous and x.shape[1] == sl['ncols'] and sl['ntasks'] > 0:
                fine = self._format('_support_fine', x.shape[1])
                tab, tile = (fine[0], fine[1]) if fine is not None else (sup[0] if sup is not None else None, 16)
                if tab is None:
                    y._zero()
                ar, ai = _cplx(alpha)
                m, k = self.shape
                b._check(b._L.ig_ccsrmm_t_slots(b._ctx, m, k, x.shape[1], ar, ai, ctypes.c_void_p(sl['entries']._arr), ctypes.c_void_p(sl['slot_ptr']._arr),
                                                ctypes.c_void_p(x._arr), x._leading_dim, ctypes.c_void_p(y._arr),
                                                ctypes.c_void_p(tab._arr) if tab is not None else None, sl['n0'], sl['nm'], sl['bm'], sl['bs'],
                                                ctypes.c_void_p(sl['tasks']._arr), sl['ntasks'], ctypes.c_void_p(sl['table']._arr),
                                                ctypes.c_void_p(sl['shared']._arr), sl['nshared'], tile, getattr(self, '_support_zw', 16), sl['words']),
                         "ig_ccsrmm_t_slots")
                return
            if (x.shape[1] == 64 and beta == 0 and perm is None and not getattr(self, '_grid_il', False) and self.shape[1] % 16 == 0
                    and self.shape[1] > 0 and self.shape[0] * 512 < 2 ** 31 and self.values.size >= self.shape[1] // 4
                    and b.tuning['wide_bricks']):
                # 64 columns at the reference boundary (BASELINE config 3): scatter through LDS brick images
                wb = self._wide_bricks()
                if wb is not None:
                    ar, ai = _cplx(alpha)
                    m, k = self.shape
                    n0, nm, bm, bs = wb['geom']
                    b._check(b._L.ig_ccsrmm_t_bricks_wide_grid(b._ctx, m, k, ar, ai, ctypes.c_void_p(wb['entries']._arr), ctypes.c_void_p(wb['rows']._arr),
                                                               ctypes.c_void_p(x._arr), x._leading_dim, ctypes.c_void_p(y._arr), y._leading_dim,
                                                               ctypes.c_void_p(wb['tasks']._arr), wb['ntasks'], ctypes.c_void_p(wb['table']._arr),
                                                               ctypes.c_void_p(wb['owned']._arr), n0, nm, bm, bs, wb['words']),
                             "ig_ccsrmm_t_bricks_wide_grid")
                    return
            if getattr(self, '_grid_il', False):
                assert perm is None and beta == 0 and y.contiguous, "interleaved panels: no row order, beta = 0"
                pt, it, dt = self._transposed()
                tab, n0, nm = sup_gather if sup_gather is not None else (None, 0, 0)
                ar, ai = _cplx(alpha)
                m, k = self.shape
                b._check(b._L.ig_ccsrmm_t_grid_il(b._ctx, m, k, x.shape[1], dt.size, ar, ai,
                                                  ctypes.c_void_p(dt._arr), ctypes.c_void_p(it._arr), ctypes.c_void_p(pt._arr),
                                                  ctypes.c_void_p(x._arr), x._leading_dim, ctypes.c_void_p(y._arr),
                                                  ctypes.c_void_p(tab._arr) if tab is not None else None, n0, nm),
                         "ig_ccsrmm_t_grid_il")
                return
            if perm is not None:
                assert not self._exwrite and b.adjoint_policy == 'transpose' and x.shape[1] <= 8, \
                    "row-ordered matrices use the packed transposed gather"
            if (sup_gather is not None or perm is not None) and not self._exwrite and b.adjoint_policy == 'transpose':
                pt, it, dt = self._transposed()
                b.ccsrmm_t(y, self.shape, it, pt, dt, x, alpha=alpha, beta=beta, support=sup_gather, xperm=perm)
                return
            if self._exwrite or b.adjoint_policy != 'transpose':
                b.ccsrmm(y, self.shape, self.colInds, self.rowPtrs, self.values,
                         x, alpha=alpha, beta=beta, adjoint=True, exwrite=self._exwrite)
            else:
                pt, it, dt = self._transposed()
                b.ccsrmm_t(y, self.shape, it, pt, dt, x, alpha=alpha, beta=beta)

"""Coil-sharded SENSE normal operator over several GPUs (one process per GPU).

`KronI(C, B)` never mixes coils, so rank g owns the coils `coil_range(C, g, G)`:
its rows of the maps matrix S', its columns of every intermediate panel, its
slice of k-space; the gridding matrix G' is replicated.  The forward operator
needs no communication.  The adjoint leaves a partial image per rank, and
A^H A x = sum_g A_g^H A_g x is ONE all-reduce (sum) of N complex64 voxels
per evaluation -- the only collective on the path.  x, and hence all
CG vectors, stay replicated, so `dot`/`norm2` need no collective.

Two providers of that all-reduce:

  * `RcclComm`   the library's own binding (`ig_comm_*` in include/indigo_hip.h: RCCL over xGMI, loaded at run
                 time by libindigo_hip.so).  No torch anywhere in the product path.  The 128-byte communicator
                 id travels from rank 0 to the others through a file in the node's temp directory (one node, one
                 launcher: the ranks are siblings, see `_rendezvous_path`).  With it the all-reduce is issued slab
                 by slab on the communicator's own stream while the cropped transform is still producing later
                 slabs of the image (`ShardedNormalOperator`, `operators.ZpadFFT._slab_hook`).
  * `TorchComm`  `torch.distributed.all_reduce` ("nccl" = RCCL on the GPU box, "gloo" in the CPU tests) on a tensor
                 that aliases the backend's buffer, enqueued on the backend's own stream.  Kept for the CPU tests
                 (the numpy oracle backend has no RCCL) and as the fallback if RCCL cannot be brought up directly.
"""
import ctypes
import hashlib
import os
import tempfile
import time

import numpy as np

_C64 = np.dtype('complex64')


def coil_range(C, rank, world):
    """Contiguous, balanced split of C coils over `world` ranks."""
    base, rem = divmod(C, world)
    lo = rank * base + min(rank, rem)
    return range(lo, lo + base + (1 if rank < rem else 0))


class _DevicePtr(object):
    """Minimal __cuda_array_interface__ carrier so torch can alias a raw device buffer."""

    def __init__(self, ptr, nfloats):
        self.__cuda_array_interface__ = dict(shape=(int(nfloats),), typestr='<f4', data=(int(ptr), False),
                                             version=2, strides=None)


def _rendezvous_path():
    """Where rank 0 leaves the communicator id.  All ranks of one launch are children of one launcher process
    (torch.distributed.run's agent, a test's parent): its pid + start time, the rendezvous port and the restart
    count name the launch uniquely on this node.  INDIGO_COMM_ID_FILE overrides."""
    if os.environ.get("INDIGO_COMM_ID_FILE"):
        return os.environ["INDIGO_COMM_ID_FILE"]
    ppid = os.getppid()
    try:
        with open("/proc/%d/stat" % ppid) as f:
            start = f.read().rsplit(")", 1)[1].split()[19]          # field 22: start time in clock ticks
    except OSError:
        start = "0"
    key = "%d_%s_%s_%s" % (ppid, start, os.environ.get("MASTER_PORT", "0"), os.environ.get("TORCHELASTIC_RESTART_COUNT", "0"))
    return os.path.join(tempfile.gettempdir(), "indigo_rccl_id_%s" % key)


def _nonce(raw):
    return hashlib.sha256(raw).hexdigest()[:32].encode()


def _publish(p, raw):
    """atomic, exclusive: a fresh temp file (O_EXCL | O_NOFOLLOW: never through a planted link), renamed over p"""
    tmp = "%s.%d.tmp" % (p, os.getpid())
    try:
        os.remove(tmp)
    except OSError:
        pass
    fd = os.open(tmp, os.O_WRONLY | os.O_CREAT | os.O_EXCL | getattr(os, "O_NOFOLLOW", 0), 0o600)
    with os.fdopen(fd, "wb") as f:
        f.write(raw)
    os.replace(tmp, p)                                       # readers never see a partial file


def _read(p, n=None):
    try:
        with open(p, "rb") as f:
            raw = f.read()
        return raw if (n is None or len(raw) == n) else None
    except OSError:
        return None


def _remove_rendezvous_files(path, world):
    for p in [path, path + ".go"] + ["%s.ack%d" % (path, r) for r in range(1, world)]:
        try:
            os.remove(p)
        except OSError:
            pass


def exchange_id(rank, world, make_id, path, nbytes=128, timeout=60.0, ready=True):
    """File rendezvous of one launch on one node: rank 0 publishes `make_id()` (nbytes), every other rank reads and
    acknowledges it, rank 0 waits for all acknowledgements and publishes the go-ahead.  Returns (id, all_ready) on every
    rank: `all_ready` is the AND over the ranks' `ready` flags (can this rank load RCCL at all?  ig_comm_preflight) -- the
    one decision the ranks must take TOGETHER before any of them enters ncclCommInitRank, which only returns when all have
    entered it.  (make_id is not called on a rank 0 that is not ready: a zero id travels instead.)

    The handshake completes BEFORE anyone enters the collective bring-up, so that a rendezvous problem (ranks that are
    not siblings, an unwritable temp directory) makes EVERY rank raise within `timeout` -- and fall back together --
    instead of leaving some ranks inside ncclCommInitRank forever.

    Left-overs of an earlier launch under the same name cannot be mistaken for this one: rank 0 removes them before it
    publishes; an acknowledgement carries the nonce of the id it answers, the reader's readiness and a fresh random TOKEN
    of that reader; the go-ahead echoes the nonce, the verdict and every token it was built from.  A reader only accepts a
    go-ahead that holds ITS token -- which no file written before this call can -- and keeps re-reading (and
    re-acknowledging a replaced id) until then.  A rank that fails removes its own files."""
    def wait_for(fn, what):
        t0 = time.time()
        while True:
            v = fn()
            if v is not None:
                return v
            if time.time() - t0 > timeout:
                raise RuntimeError("rendezvous: rank %d timed out after %.0f s waiting for %s (%s)" % (rank, timeout, what, path))
            time.sleep(0.02)

    if rank == 0:
        raw = make_id() if ready else bytes(nbytes)
        assert len(raw) == nbytes
        all_ready = bool(ready)
        if world > 1:
            _remove_rendezvous_files(path, world)
            nonce = _nonce(raw)
            try:
                _publish(path, raw)
                tokens = []
                for r in range(1, world):
                    def ack_of(r=r):
                        parts = (_read("%s.ack%d" % (path, r)) or b"").split(b":")
                        return parts if len(parts) == 3 and parts[0] == nonce and len(parts[1]) == 32 else None
                    parts = wait_for(ack_of, "the acknowledgement of rank %d" % r)
                    tokens.append(parts[1])
                    all_ready = all_ready and parts[2] == b"1"
                _publish(path + ".go", b":".join([nonce, b"1" if all_ready else b"0"] + tokens))
            except BaseException:
                _remove_rendezvous_files(path, world)
                raise
        return raw, all_ready
    ack = "%s.ack%d" % (path, rank)
    token = os.urandom(16).hex().encode()
    state = {}

    def step():
        raw = _read(path, nbytes)
        if raw is None:
            return None
        if state.get("raw") != raw:                          # first sight of an id, or rank 0 replaced a stale one
            state["raw"] = raw
            _publish(ack, b":".join([_nonce(raw), token, b"1" if ready else b"0"]))
        go = (_read(path + ".go") or b"").split(b":")
        if len(go) >= 3 and go[0] == _nonce(raw) and token in go[2:]:
            return raw, go[1] == b"1"
        return None
    try:
        return wait_for(step, "rank 0's communicator id and go-ahead")
    except BaseException:
        try:
            os.remove(ack)
        except OSError:
            pass
        raise


def cleanup_rendezvous(rank, world, path, timeout=10.0):
    """remove the rendezvous files: every other rank removes its acknowledgement once it has the go-ahead; rank 0 removes
    the id and the go-ahead once all acknowledgements are gone (so nobody is still looking for them)"""
    def rm(p):
        try:
            os.remove(p)
        except OSError:
            pass
    if rank != 0:
        rm("%s.ack%d" % (path, rank))
        return
    t0 = time.time()
    while any(os.path.exists("%s.ack%d" % (path, r)) for r in range(1, world)) and time.time() - t0 < timeout:
        time.sleep(0.02)
    _remove_rendezvous_files(path, world)


class RcclComm(object):
    """Sum all-reduce of a HipBackend array across the ranks through the library's RCCL binding (ig_comm_*)."""

    def __init__(self, backend, rank, world, timeout=60.0, overlap=True):
        from indigo_amd import _lib
        self._backend, self.rank, self.world = backend, int(rank), int(world)
        self._L = backend._L
        nbytes = 128
        path = _rendezvous_path() if world > 1 else None

        def make_id():
            buf = ctypes.create_string_buffer(nbytes)
            _lib.check(self._L.ig_comm_unique_id(buf), None, "ig_comm_unique_id")
            return buf.raw
        # what can fail on this rank alone (no RCCL to load) is tried first and VOTED on in the handshake: either every rank
        # enters ncclCommInitRank or none does (and all of them raise here -- `bench.py --comm auto` then falls back to
        # torch.distributed on all ranks together)
        ready = self._L.ig_comm_preflight() == 0
        why = None if ready else _lib.last_error(None)
        raw, all_ready = exchange_id(self.rank, self.world, make_id, path, nbytes, timeout, ready=ready)
        if not all_ready:
            if path:
                cleanup_rendezvous(self.rank, self.world, path, timeout=2.0)
            raise RuntimeError("ig_comm: %s; no rank enters the RCCL bring-up"
                               % ("this rank cannot load RCCL (%s)" % why if not ready else "another rank cannot load RCCL"))
        idbuf = ctypes.create_string_buffer(raw, nbytes)
        comm = ctypes.c_void_p()
        try:
            backend._check(self._L.ig_comm_init_rank(backend._ctx, self.world, self.rank, idbuf, ctypes.byref(comm)), "ig_comm_init_rank")
            self._comm = comm
            self.overlap = bool(overlap)     # slab-by-slab all-reduce on the communicator's own stream (ShardedNormalOperator)
            self._pending = False
            self.barrier()                                           # every rank is through the bring-up
        finally:
            if path:
                cleanup_rendezvous(self.rank, self.world, path, timeout=10.0 if getattr(self, '_comm', None) else 0.0)

    def describe(self):
        buf = ctypes.create_string_buffer(256)
        self._L.ig_comm_info(self._comm, None, None, buf, 256)
        return "ig_comm (C ABI) over %s, %d ranks, slab overlap %s" % (buf.value.decode(), self.world, "on" if self.overlap else "off")

    def allreduce_(self, arr, force=False):
        """in-place sum over ranks, in order on the backend's stream"""
        if self.world == 1 and not force:
            return
        assert arr.dtype == _C64 and arr.contiguous
        self._backend._check(self._L.ig_allreduce_sum_f32(self._comm, ctypes.c_void_p(arr._arr), arr.size * 2), "ig_allreduce_sum_f32")

    def allreduce_slab_(self, arr, lo, hi):
        """sum voxels [lo, hi) of arr over the ranks on the communicator's own stream, after the work enqueued so far"""
        ptr = arr._arr + lo * 8
        self._backend._check(self._L.ig_allreduce_sum_f32_side(self._comm, ctypes.c_void_p(ptr), (hi - lo) * 2), "ig_allreduce_sum_f32_side")
        self._pending = True

    def join(self):
        if self._pending:
            self._backend._check(self._L.ig_comm_join(self._comm), "ig_comm_join")
            self._pending = False

    def barrier(self):
        self._backend._check(self._L.ig_comm_barrier(self._comm), "ig_comm_barrier")

    def max(self, value):
        v = ctypes.c_double(float(value))
        self._backend._check(self._L.ig_allreduce_max_f64_host(self._comm, ctypes.byref(v)), "ig_allreduce_max_f64_host")
        return v.value

    def allreduce(self, value):
        """sum of a host float over ranks (the reference's team.allreduce in pdot/pnorm2, backend.py:469-479)"""
        v = ctypes.c_double(float(value))
        self._backend._check(self._L.ig_allreduce_sum_f64_host(self._comm, ctypes.byref(v)), "ig_allreduce_sum_f64_host")
        return v.value

    def close(self):
        if getattr(self, '_comm', None):
            self._L.ig_comm_destroy(self._comm)
            self._comm = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DirectComm(RcclComm):
    """The library's own all-reduce route, no RCCL (ig_comm_init_direct): every rank exposes a window of device memory to the other
    ranks of the node through HIP IPC and reduces its 1 / world slab of all windows -- a direct reduce-scatter + all-gather over all
    xGMI links where RCCL may pick a ring.  Same interface as RcclComm (the C ABI routes ig_allreduce_* by the communicator's kind);
    host-synchronous, so the slab overlap is off.  The ranks agree on the shared-memory name through the same file rendezvous as
    the RCCL id (a random 16-byte token of rank 0)."""

    def __init__(self, backend, rank, world, timeout=60.0, window_bytes=256 << 20, name=None):
        self._backend, self.rank, self.world = backend, int(rank), int(world)
        self._L = backend._L
        self.overlap = False
        self._pending = False
        path = (_rendezvous_path() + ".direct") if (world > 1 and name is None) else None
        if name is None:
            raw, _ = exchange_id(self.rank, self.world, lambda: os.urandom(16), path, 16, timeout, ready=True) if world > 1 else (os.urandom(16), True)
            name = "/indigo_direct_" + raw.hex()
        comm = ctypes.c_void_p()
        try:
            backend._check(self._L.ig_comm_init_direct(backend._ctx, self.world, self.rank, name.encode(), int(window_bytes) // 4096 * 4096,
                                                       float(timeout), ctypes.byref(comm)), "ig_comm_init_direct")
            self._comm = comm
        finally:
            if path:
                cleanup_rendezvous(self.rank, self.world, path, timeout=10.0 if getattr(self, '_comm', None) else 0.0)

    def describe(self):
        buf = ctypes.create_string_buffer(256)
        self._L.ig_comm_info(self._comm, None, None, buf, 256)
        return "ig_comm (C ABI), %s, %d ranks, host-synchronous" % (buf.value.decode(), self.world)


class TorchComm(object):
    """Sum all-reduce of a backend array across the default torch.distributed process group."""
    overlap = False

    def __init__(self, backend):
        import torch
        import torch.distributed as dist
        assert dist.is_initialized(), "call torch.distributed.init_process_group first"
        self._torch, self._dist, self._backend = torch, dist, backend
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self._on_gpu = hasattr(backend, 'stream')
        self._stream = None
        if self._on_gpu:
            self._device = torch.device('cuda', backend.device_id)
            self._stream = torch.cuda.ExternalStream(backend.stream, device=self._device)
        self._cache = {}

    def describe(self):
        return "torch.distributed (%s), %d ranks" % (self._dist.get_backend(), self.world)

    def _tensor(self, arr):
        assert arr.dtype == _C64 and arr.contiguous
        key = (id(arr._arr) if not self._on_gpu else arr._arr, arr.size)
        t = self._cache.get(key)
        if t is None:
            if self._on_gpu:
                t = self._torch.as_tensor(_DevicePtr(arr._arr, arr.size * 2), device=self._device)
            else:
                t = self._torch.from_numpy(arr._arr.reshape(-1, order='F').view(np.float32))
                assert t.data_ptr() == arr._arr.ctypes.data, "oracle array is not contiguous"
            self._cache = {key: t}          # keep only the latest alias
        return t

    def allreduce_(self, arr, force=False):
        """in-place sum over ranks (`force` issues the collective even for a single rank: used by tests)"""
        if self.world == 1 and not force:
            return
        t = self._tensor(arr)
        if self._on_gpu:
            with self._torch.cuda.stream(self._stream):
                self._dist.all_reduce(t, op=self._dist.ReduceOp.SUM)
        else:
            self._dist.all_reduce(t, op=self._dist.ReduceOp.SUM)

    def join(self):
        pass

    def barrier(self):
        if self.world > 1:
            self._dist.barrier()

    def max(self, value):
        """max of a host float over ranks"""
        if self.world == 1:
            return value
        t = self._torch.tensor([value], dtype=self._torch.float64,
                               device=self._device if self._on_gpu else 'cpu')
        self._dist.all_reduce(t, op=self._dist.ReduceOp.MAX)
        return float(t.item())

    def allreduce(self, value):
        """sum of a host float over ranks (the reference's team.allreduce, backend.py:469-479)"""
        if self.world == 1:
            return value
        t = self._torch.tensor([value], dtype=self._torch.float64, device=self._device if self._on_gpu else 'cpu')
        self._dist.all_reduce(t, op=self._dist.ReduceOp.SUM)
        return float(t.item())

    def close(self):
        if self._dist.is_initialized():
            self._dist.destroy_process_group()


class ShardedNormalOperator(object):
    """y = sum_g A_g^H A_g x (+ lamda x) with the sum taken by one all-reduce.

    `A_local` is the forward operator for this rank's coils (e.g.
    `SenseProblem.build_zpadfft(backend, coils=coil_range(C, rank, world))`).
    Exposes `eval(y, x)` and `shape` so `Backend.cg` can drive it like any operator.

    With a communicator that has its own stream (`RcclComm`) and a tree that ends in ONE coil-interleaved
    `ZpadFFT` leaf, the image is all-reduced in `nslabs` z-slabs: slab s crosses xGMI while the y and x passes
    of slab s+1 run (the z pass, gridding and everything before it cannot overlap: the image does not exist yet).
    """

    def __init__(self, A_local, comm, lamda=0.0, nslabs=4):
        self._A = A_local
        self._backend = A_local._backend
        self._comm = comm
        self._lamda = lamda
        n = A_local.shape[1]
        self.shape = (n, n)
        self.dtype = _C64
        self._ksp = None
        self._leaf = None
        self._nslabs = int(nslabs)
        # How the image is all-reduced -- in `nslabs` slabs behind the transform, or whole -- is ONE decision of all ranks:
        # their trees differ (8 coils on 3 ranks: 3 + 3 + 2 -- the 3-coil ranks have no coil-interleaved leaf, the 2-coil
        # rank has), and ranks that issue different sequences of collectives hang or mix up their buffers.  'undecided':
        # the first one-column evaluation reduces the whole image on every rank, records whether this rank's tree WOULD have
        # covered the image slab by slab (the hook then only takes notes), and the ranks vote (one host max-reduction);
        # 'slab' only if every rank can.  'full' without a vote where no rank can know otherwise (one rank, a communicator
        # without its own stream, nslabs <= 1: the same on all ranks).
        self._route = 'undecided' if (getattr(comm, 'overlap', False) and comm.world > 1 and self._nslabs > 1) else 'full'
        if self._route == 'undecided':
            from indigo_amd import operators as op
            # the leaf that writes y LAST in the adjoint: the rightmost factor of the tree, or -- for a VStack of coil chunks,
            # whose adjoint accumulates its children's images in order (operators.VStack._eval) -- of its last child
            r = A_local
            if isinstance(r, op.VStack):
                r = r.children[-1]
            if isinstance(r, op.HeadRows):          # a chunk padded with zero-weight coils: alpha and beta pass through to its tree
                r = r.child
            while isinstance(r, op.Product):
                r = r.right
            if isinstance(r, op.ZpadFFT) and hasattr(self._backend, 'ifft_cropped_sum') and (r._layout == 2 or (r._layout == 1 and r._C == 1)):
                self._leaf = r

    def _covers(self, done):
        done = sorted(done)
        return bool(done) and done[0][0] == 0 and done[-1][1] == self.shape[1] and all(a[1] == b[0] for a, b in zip(done[:-1], done[1:]))

    def eval(self, y, x, alpha=1, beta=0, forward=True, left=True):
        assert alpha == 1 and beta == 0, "ShardedNormalOperator computes y = AHA x only"
        B = self._backend
        ncols = x.size // self.shape[1]
        if self._ksp is None or self._ksp.shape != (self._A.shape[0], ncols):
            self._ksp = B.zero_array((self._A.shape[0], ncols), _C64, name='ksp(shard)')
        self._A.eval(self._ksp, x)
        if self._route == 'undecided' and ncols == 1:
            done = []
            if self._leaf is not None:
                self._leaf._slab_hook = (self._nslabs, lambda arr, lo, hi: done.append((lo, hi)))       # takes notes, sends nothing
            try:
                self._A.eval(y, self._ksp, forward=False)
            finally:
                if self._leaf is not None:
                    self._leaf._slab_hook = None
            self._comm.allreduce_(y)
            cannot = 0.0 if self._covers(done) else 1.0
            self._route = 'slab' if self._comm.max(cannot) == 0.0 else 'full'
        elif self._route == 'slab' and ncols == 1:
            # The leaf all-reduces the image slab by slab (ZpadFFT._slab_hook).  That it does was established -- on every rank --
            # by the first evaluation; it is still CHECKED: a tree whose last writer suddenly takes another branch must not
            # return a rank-local partial sum silently, and it cannot be repaired locally either (the other ranks have issued
            # their slab collectives), so it raises.
            done = []

            def hook(arr, lo, hi):
                self._comm.allreduce_slab_(arr, lo, hi)
                done.append((lo, hi))
            self._leaf._slab_hook = (self._nslabs, hook)
            try:
                self._A.eval(y, self._ksp, forward=False)
            finally:
                self._leaf._slab_hook = None
            self._comm.join()
            if not self._covers(done):
                raise RuntimeError("slab all-reduce covered %s of [0, %d): the ranks' collectives no longer match" % (sorted(done), self.shape[1]))
        else:
            self._A.eval(y, self._ksp, forward=False)
            self._comm.allreduce_(y)
        if self._lamda:
            B.axpby(1, y, self._lamda, x)

"""Coil-sharded SENSE normal operator over several GPUs (one process per GPU).

`KronI(C, B)` never mixes coils, so rank g owns the coils `coil_range(C, g, G)`:
its rows of the maps matrix S', its columns of every intermediate panel, its
slice of k-space; the gridding matrix G' is replicated.  The forward operator
needs no communication.  The adjoint leaves a partial image per rank, and
A^H A x = sum_g A_g^H A_g x is ONE all-reduce (sum) of N complex64 voxels
per evaluation -- the only collective on the path.  x, and hence all
CG vectors, stay replicated, so `dot`/`norm2` need no collective.

The collective is `torch.distributed.all_reduce` (backend "nccl" = RCCL over
xGMI on the GPU box, "gloo" in the CPU tests) on a tensor that aliases the
backend's device buffer; it is enqueued on the backend's own stream so no host
synchronisation is needed.  torch is imported lazily and only here.
"""
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


class TorchComm(object):
    """Sum all-reduce of a backend array across the default torch.distributed process group."""

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


class ShardedNormalOperator(object):
    """y = sum_g A_g^H A_g x (+ lamda x) with the sum taken by one all-reduce.

    `A_local` is the forward operator for this rank's coils (e.g.
    `SenseProblem.build_fused(backend, coils=coil_range(C, rank, world))`).
    Exposes `eval(y, x)` and `shape` so `Backend.cg` can drive it like any operator.
    """

    def __init__(self, A_local, comm, lamda=0.0):
        self._A = A_local
        self._backend = A_local._backend
        self._comm = comm
        self._lamda = lamda
        n = A_local.shape[1]
        self.shape = (n, n)
        self.dtype = _C64
        self._ksp = None

    def eval(self, y, x, alpha=1, beta=0, forward=True, left=True):
        assert alpha == 1 and beta == 0, "ShardedNormalOperator computes y = AHA x only"
        B = self._backend
        ncols = x.size // self.shape[1]
        if self._ksp is None or self._ksp.shape != (self._A.shape[0], ncols):
            self._ksp = B.zero_array((self._A.shape[0], ncols), _C64, name='ksp(shard)')
        self._A.eval(self._ksp, x)
        self._A.eval(y, self._ksp, forward=False)
        self._comm.allreduce_(y)
        if self._lamda:
            B.axpby(1, y, self._lamda, x)

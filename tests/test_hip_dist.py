"""GPU side of the multi-GPU path, exercised with ONE rank on the one GPU of the test box.

  * the library's own RCCL communicator (ig_comm_* through ctypes, no torch in the process): bring-up from a
    128-byte id, in-order all-reduce, slab-by-slab all-reduce on the communicator's stream overlapped with the
    cropped transform, host-scalar reductions, barrier;
  * the torch fallback: torch (RCCL) aliases the backend's device buffer through __cuda_array_interface__ and the
    all-reduce is enqueued on the backend's own stream.
Both run in subprocesses (the second because torch must be imported before libindigo_hip.so so that both share one
HIP runtime; the first so that RCCL is loaded by the library alone)."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

WORKER = r"""
import os, sys
sys.path.insert(0, os.environ["REPO_ROOT"])
os.environ["INDIGO_HIP_WITH_TORCH"] = "1"
import torch, torch.distributed as dist
import numpy as np
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from indigo_amd.backends import get_backend
from indigo_amd.dist import TorchComm, ShardedNormalOperator
from indigo_amd.sense import SenseProblem, normal_operator
from indigo_amd.util import rand64c
B = get_backend("hip")
comm = TorchComm(B)
x = rand64c(100003, seed=1)
d = B.copy_array(x)
t = comm._tensor(d)
assert t.data_ptr() == d._arr and t.numel() == 2 * x.size
B.scale(d, 2.0)                       # queued on the backend stream ...
comm.allreduce_(d, force=True)        # ... the collective must see it (same stream), sum over 1 rank = identity
B.axpby(1, d, 1, B.copy_array(x))     # ... and later work must see the collective's result
np.testing.assert_allclose(d.to_host(), 3 * x, rtol=1e-6)
# sharded operator with a single shard == plain normal operator
p = SenseProblem.synthetic((16, 16, 16), 2, nspokes=24, nreadout=32, seed=4)
A = p.build_fused(B)
xs = B.copy_array(rand64c(A.shape[1], 1, seed=2))
y1 = B.zero_array((A.shape[1], 1), np.dtype("complex64"))
y2 = B.zero_array((A.shape[1], 1), np.dtype("complex64"))
ShardedNormalOperator(A, comm, lamda=0.1).eval(y1, xs)
normal_operator(A, lamda=0.1).eval(y2, xs)
a, b = y1.to_host(), y2.to_host()
assert np.linalg.norm(a - b) <= 1e-6 * np.linalg.norm(b)
dist.destroy_process_group()
print("OK")
"""


def test_torch_aliasing_and_stream_ordered_allreduce():
    env = dict(os.environ, REPO_ROOT=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT="29517",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", WORKER], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       text=True, timeout=600)
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout[-4000:]


RCCL_WORKER = r"""
import os, sys
sys.path.insert(0, os.environ["REPO_ROOT"])
import numpy as np
if os.environ.get("TORCH_FIRST") == "1":
    # bench.py --comm auto: torch (with its own HIP runtime and RCCL copies) is in the process before the library loads;
    # ig_comm_* must then find and reuse the RCCL copy that is already there
    os.environ["INDIGO_HIP_WITH_TORCH"] = "1"
    import torch
else:
    assert "torch" not in sys.modules
from indigo_amd.backends import get_backend
from indigo_amd.dist import RcclComm, ShardedNormalOperator
from indigo_amd.sense import SenseProblem, normal_operator
from indigo_amd.util import rand64c
B = get_backend("hip")
comm = RcclComm(B, 0, 1)
print(comm.describe())
x = rand64c(100003, seed=1)
d = B.copy_array(x)
B.scale(d, 2.0)                       # queued on the backend stream ...
comm.allreduce_(d, force=True)        # ... the collective must see it (same stream); sum over 1 rank = identity
B.axpby(1, d, 1, B.copy_array(x))     # ... and later work must see the collective's result
np.testing.assert_allclose(d.to_host(), 3 * x, rtol=1e-6)
# side-stream all-reduce of two halves, joined before the next kernel
B.scale(d, 0.5)
comm.allreduce_slab_(d, 0, 50000)
comm.allreduce_slab_(d, 50000, 100003)
comm.join()
B.scale(d, 2.0)
np.testing.assert_allclose(d.to_host(), 3 * x, rtol=1e-6)
assert comm.max(2.5) == 2.5 and comm.allreduce(1.25) == 1.25
comm.barrier()
# sharded operator, slab path forced on although there is one rank: == the plain normal operator
p = SenseProblem.synthetic((128, 128, 128), 4, nspokes=200, nreadout=256, width=2, oversamp=2.0, seed=5)
A = p.build_zpadfft(B)
xs = B.copy_array(rand64c(A.shape[1], 1, seed=2))
c64 = np.dtype("complex64")
y1, y2 = B.zero_array((A.shape[1], 1), c64), B.zero_array((A.shape[1], 1), c64)
op = ShardedNormalOperator(A, comm, lamda=0.1, nslabs=5)
assert op._route == 'full'            # one rank: nothing to overlap, nothing to vote on
comm.world = 2                        # pretend, to take the slab route (the RCCL communicator still has one rank)
op = ShardedNormalOperator(A, comm, lamda=0.1, nslabs=5)
assert op._leaf is not None and op._route == 'undecided'
op.eval(y1, xs)                       # first evaluation: whole-image all-reduce, the ranks vote on the route
assert op._route == 'slab'
op.eval(y1, xs)                       # the slab route for real
comm.world = 1
B._scratch = None
normal_operator(A, lamda=0.1).eval(y2, xs)
a, b = y1.to_host(), y2.to_host()
assert np.linalg.norm(a - b) <= 1e-6 * np.linalg.norm(b), np.linalg.norm(a - b) / np.linalg.norm(b)
# a rank with more coils than one interleaved grid holds: a VStack of chunks.  The LAST chunk's leaf all-reduces slab by slab
# (its slabs are added to the earlier chunks' images first: beta = 1 inside the hook)
del A, op
B._scratch = None
A = p.build_zpadfft(B, chunk=2)
from indigo_amd import operators as ops
assert isinstance(A, ops.VStack) and len(A.children) == 2
comm.world = 2
op = ShardedNormalOperator(A, comm, lamda=0.1, nslabs=3)
assert op._leaf is A.children[-1].right
op.eval(y1, xs)
assert op._route == 'slab'
op.eval(y1, xs)
comm.world = 1
B._scratch = None
normal_operator(A, lamda=0.1).eval(y2, xs)
a, b = y1.to_host(), y2.to_host()
assert np.linalg.norm(a - b) <= 1e-6 * np.linalg.norm(b), np.linalg.norm(a - b) / np.linalg.norm(b)
# a rank with ONE coil (per-coil grid layout 1): the cropped transform writes the image slab by slab too
del A, op
B._scratch = None
A = p.build_zpadfft(B, coils=[2])
assert A.right._layout == 1 and A.right._C == 1
comm.world = 2
op = ShardedNormalOperator(A, comm, lamda=0.1, nslabs=3)
assert op._leaf is A.right
op.eval(y1, xs)
assert op._route == 'slab'
op.eval(y1, xs)
comm.world = 1
B._scratch = None
normal_operator(A, lamda=0.1).eval(y2, xs)
a, b = y1.to_host(), y2.to_host()
assert np.linalg.norm(a - b) <= 1e-6 * np.linalg.norm(b), np.linalg.norm(a - b) / np.linalg.norm(b)
# a rank whose coil count is no power of two (8 coils on 3 ranks: 3 + 3 + 2): ONE 4-wide interleaved chunk whose fourth coil has zero
# weights, cut to its three real coils by operators.HeadRows -- alpha and beta pass through to the leaf underneath, which reduces slab
# by slab like any other
del A, op
B._scratch = None
A = p.build_zpadfft(B, coils=[0, 1, 2])
assert isinstance(A, ops.HeadRows) and A.child.right._C == 4
comm.world = 2
op = ShardedNormalOperator(A, comm, lamda=0.1, nslabs=3)
assert op._leaf is A.child.right
op.eval(y1, xs)
assert op._route == 'slab'
op.eval(y1, xs)
comm.world = 1
B._scratch = None
normal_operator(A, lamda=0.1).eval(y2, xs)
a, b = y1.to_host(), y2.to_host()
assert np.linalg.norm(a - b) <= 1e-6 * np.linalg.norm(b), np.linalg.norm(a - b) / np.linalg.norm(b)
del A, op
B._scratch = None
A = p.build_zpadfft(B, chunk=2)
# a tree whose last writer does not take the slab branch: the first evaluation notices (nothing was sent slab by slab yet),
# the vote says 'full', and every evaluation is ONE plain all-reduce
calls = []
orig = comm.allreduce_
comm.allreduce_ = lambda arr, force=False: (calls.append(arr.size), orig(arr, force=True))
comm.world = 2
op = ShardedNormalOperator(A, comm, nslabs=3)
op._leaf = A.children[0].right           # (wrong on purpose: chunk 0's leaf runs with beta = 0 FIRST, then chunk 1 overwrites nothing it reduced)
op._leaf = type("NoHook", (), {"_slab_hook": None})()    # a leaf that never calls the hook
op.eval(y1, xs)
assert calls == [y1.size] and op._route == 'full', (calls, op._route)
op.eval(y1, xs)
comm.world = 1
assert calls == [y1.size] * 2, calls
a = y1.to_host()
B._scratch = None
normal_operator(A).eval(y2, xs)
b = y2.to_host()
assert np.linalg.norm(a - b) <= 1e-6 * np.linalg.norm(b)
comm.close()
print("OK")
"""


@pytest.mark.parametrize("torch_first", ["0", "1"])
def test_rccl_communicator_through_the_c_abi(torch_first):
    env = dict(os.environ, REPO_ROOT=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0", TORCH_FIRST=torch_first)
    env.pop("INDIGO_HIP_WITH_TORCH", None)
    r = subprocess.run([sys.executable, "-c", RCCL_WORKER], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       text=True, timeout=600)
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout[-4000:]
    if torch_first == "1":
        assert "already loaded" in r.stdout, r.stdout[-2000:]          # the library reused torch's RCCL, it did not load a second one


TWO_RANK_WORKER = r"""
import os, sys
sys.path.insert(0, os.environ["REPO_ROOT"])
import numpy as np
from indigo_amd.backends import get_backend
from indigo_amd.dist import RcclComm, ShardedNormalOperator, coil_range
from indigo_amd.sense import SenseProblem, normal_operator
from indigo_amd.util import rand64c
rank, world = int(os.environ["RANK"]), 2
B = get_backend("hip", device_id=rank)
comm = RcclComm(B, rank, world)
p = SenseProblem.synthetic((128, 128, 128), 8, nspokes=200, nreadout=256, width=2, oversamp=2.0, seed=5)
c64 = np.dtype("complex64")
xs = B.copy_array(rand64c(int(np.prod(p.N)), 1, seed=2))
A = p.build_zpadfft(B, coils=coil_range(8, rank, world))
y = B.zero_array((A.shape[1], 1), c64)
op = ShardedNormalOperator(A, comm, lamda=0.1, nslabs=4)
assert op._leaf is not None
op.eval(y, xs)                              # first evaluation: whole image, then the ranks vote
assert op._route == 'slab', op._route       # the slab route, for real this time
first = y.to_host()
op.eval(y, xs)
got = y.to_host()
assert np.linalg.norm(got - first) <= 1e-6 * np.linalg.norm(first)
B._scratch = None
del A, op
Afull = p.build_zpadfft(B)                  # the unsharded operator on this rank's GPU
y2 = B.zero_array((Afull.shape[1], 1), c64)
normal_operator(Afull, lamda=0.1).eval(y2, xs)
ref = y2.to_host()
err = np.linalg.norm(got - ref) / np.linalg.norm(ref)
assert err < 1e-5, err
comm.barrier()
comm.close()
print("OK rank", rank, err)
"""


def test_two_rccl_ranks_against_the_unsharded_operator(tmp_path):
    """two processes, two GPUs, the library's own RCCL communicator: the coil-sharded normal operator with the slab-overlapped
    all-reduce equals the unsharded one on every rank.  Needs two GPUs: skipped on the one-GPU boxes this suite usually runs on."""
    import ctypes
    from indigo_amd import _lib
    n = ctypes.c_int()
    _lib.lib().ig_device_count(ctypes.byref(n))
    if n.value < 2:
        pytest.skip("needs two GPUs (this box has %d)" % n.value)
    procs = []
    for rank in range(2):
        env = dict(os.environ, REPO_ROOT=ROOT, RANK=str(rank), WORLD_SIZE="2", HSA_ENABLE_IPC_MODE_LEGACY="0",
                   INDIGO_COMM_ID_FILE=str(tmp_path / "rccl_id"))
        env.pop("INDIGO_HIP_WITH_TORCH", None)
        procs.append(subprocess.Popen([sys.executable, "-c", TWO_RANK_WORKER], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            pytest.fail("two-rank RCCL workers timed out")
        outs.append(o)
    assert all(p.returncode == 0 and "OK rank" in o for p, o in zip(procs, outs)), "\n".join(o[-3000:] for o in outs)


DIRECT_WORKER = r"""
import os, sys, hashlib
sys.path.insert(0, os.environ["REPO_ROOT"])
import numpy as np
from indigo_amd.backends import get_backend
from indigo_amd.dist import DirectComm, ShardedNormalOperator, coil_range
from indigo_amd.sense import SenseProblem, normal_operator
from indigo_amd.util import rand64c
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
B = get_backend("hip", device_id=0)                  # every rank on the ONE GPU of the box: IPC works across processes on a shared device
comm = DirectComm(B, rank, world, name=os.environ["DIRECT_NAME"], window_bytes=int(os.environ["DIRECT_WINDOW"]), timeout=120.0)
assert "direct" in comm.describe()
digests = []
for n in (100003, 1 << 20, 5):                       # a ragged length, one larger than the window (pieces), one shorter than the rank count's slabs
    xs = [rand64c(n, seed=10 * n % 9973 + r) for r in range(world)]
    d = B.copy_array(xs[rank])
    B.scale(d, 2.0)                                  # queued on the backend's stream: the collective must see it ...
    comm.allreduce_(d)
    got = d.to_host()
    digests.append(hashlib.sha256(got.tobytes()).hexdigest()[:16])
    B.axpby(1, d, 1, B.copy_array(xs[rank]))         # ... and later work its result
    exp = 2.0 * np.sum(np.stack(xs).astype(np.complex128), axis=0)
    assert np.linalg.norm(got - exp) <= 1e-6 * np.linalg.norm(exp), (n, np.linalg.norm(got - exp) / np.linalg.norm(exp))
    # the sum in rank order, float32: bit for bit what numpy gives in that order
    ref = np.zeros(n, np.complex64)
    for x in xs:
        ref = (ref + (2.0 * x).astype(np.complex64)).astype(np.complex64)
    assert np.array_equal(got.view(np.float32), ref.view(np.float32)), n
    assert np.linalg.norm(d.to_host() - (exp + xs[rank])) <= 1e-6 * np.linalg.norm(exp)
assert comm.max(float(rank)) == world - 1 and comm.allreduce(float(rank + 1)) == world * (world + 1) / 2
comm.barrier()
# the coil-sharded normal operator against the unsharded one
p = SenseProblem.synthetic((64, 64, 64), 4 * world, nspokes=100, nreadout=128, width=2, oversamp=2.0, seed=5)
c64 = np.dtype("complex64")
xs = B.copy_array(rand64c(int(np.prod(p.N)), 1, seed=2))
A = p.build_zpadfft(B, coils=coil_range(4 * world, rank, world))
y = B.zero_array((A.shape[1], 1), c64)
op = ShardedNormalOperator(A, comm, lamda=0.1)
op.eval(y, xs)
op.eval(y, xs)
got = y.to_host()
B._scratch = None
del A, op
Afull = p.build_zpadfft(B)
y2 = B.zero_array((Afull.shape[1], 1), c64)
normal_operator(Afull, lamda=0.1).eval(y2, xs)
ref = y2.to_host()
err = np.linalg.norm(got - ref) / np.linalg.norm(ref)
assert err < 1e-5, err
comm.barrier()
comm.close()
print("OK rank", rank, "digests", " ".join(digests), "err %.2e" % err)
"""


@pytest.mark.parametrize("world", [2, 3])
def test_direct_allreduce_over_ipc_windows_on_one_gpu(world):
    """The library's own all-reduce route (ig_comm_init_direct: reduce-scatter + all-gather over peer-mapped IPC windows, no RCCL),
    rehearsed with `world` processes on the ONE GPU of the box -- protocol, piece-wise messages, bit-exactness (every rank holds the
    same bits: the float32 sum in rank order), stream ordering, host scalars, and the coil-sharded normal operator against the
    unsharded one.  What this cannot show is bandwidth over xGMI: that needs the GPUs of a node."""
    name = "/indigo_direct_test_%d_%d" % (os.getpid(), world)
    procs = []
    for rank in range(world):
        env = dict(os.environ, REPO_ROOT=ROOT, RANK=str(rank), WORLD_SIZE=str(world), HSA_ENABLE_IPC_MODE_LEGACY="0", DIRECT_NAME=name,
                   DIRECT_WINDOW=str(4 << 20))
        env.pop("INDIGO_HIP_WITH_TORCH", None)
        procs.append(subprocess.Popen([sys.executable, "-c", DIRECT_WORKER], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            pytest.fail("direct all-reduce workers timed out")
        outs.append(o)
    assert all(p.returncode == 0 and "OK rank" in o for p, o in zip(procs, outs)), "\n".join(o[-3000:] for o in outs)
    digests = {o.split("digests", 1)[1].split("err")[0].strip() for o in outs}
    assert len(digests) == 1, digests          # every rank holds the same bits


DIRECT_DEAD_WORKER = r"""
import os, sys, time
sys.path.insert(0, os.environ["REPO_ROOT"])
import numpy as np
from indigo_amd.backends import get_backend
from indigo_amd.dist import DirectComm
from indigo_amd.util import rand64c
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
B = get_backend("hip", device_id=0)
comm = DirectComm(B, rank, world, timeout=float(os.environ["DIRECT_TIMEOUT"]), window_bytes=4 << 20, name=os.environ["DIRECT_NAME"])
d = B.copy_array(rand64c(100000, seed=rank))
comm.allreduce_(d)                       # one good collective first
if rank == 1:
    os._exit(7)                          # dies without a word: no barrier, no clean-up
t0 = time.time()
try:
    comm.allreduce_(d)
except RuntimeError as e:
    print("FAILED AS IT SHOULD after %.1f s: %s" % (time.time() - t0, e))
    sys.exit(0)
print("the collective returned although a rank was dead")
sys.exit(3)
"""


def test_direct_allreduce_fails_instead_of_hanging_when_a_rank_dies():
    """a rank that dies between two collectives: the survivor's next all-reduce gives up at its barrier after the communicator's timeout
    (3 s here) with an error that says so -- it does not hang (round 3 saw a multi-rank rehearsal sit silent for seven minutes)"""
    name = "/indigo_direct_dead_%d" % os.getpid()
    procs = []
    for rank in range(2):
        env = dict(os.environ, REPO_ROOT=ROOT, RANK=str(rank), WORLD_SIZE="2", HSA_ENABLE_IPC_MODE_LEGACY="0", DIRECT_NAME=name, DIRECT_TIMEOUT="3")
        env.pop("INDIGO_HIP_WITH_TORCH", None)
        procs.append(subprocess.Popen([sys.executable, "-c", DIRECT_DEAD_WORKER], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=120)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            pytest.fail("the surviving rank hung")
        outs.append(o)
    assert procs[1].returncode == 7, outs[1][-2000:]
    assert procs[0].returncode == 0 and "FAILED AS IT SHOULD" in outs[0] and "waited" in outs[0], outs[0][-3000:]

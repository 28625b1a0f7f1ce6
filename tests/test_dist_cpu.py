"""Coil-sharded normal operator across two processes (gloo on CPU, world_size 2).

Each rank owns half of the coils, evaluates A_g^H A_g x on the numpy oracle
backend and the partial images meet in ONE all-reduce -- the same code path
bench.py runs with one rank per GPU over RCCL.  The result must equal the
single-process operator with all coils; CG driven through the sharded operator
must reproduce the single-process iterates (vectors stay replicated).
"""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, golden, rel_err

WORKER = r"""
import os, sys
import numpy as np
sys.path.insert(0, os.environ["REPO_ROOT"])
import torch.distributed as dist
from oracle.np_backend import NumpyBackend
from indigo_amd.dist import ShardedNormalOperator, TorchComm, coil_range
from indigo_amd.sense import SenseProblem

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
g = np.load(os.path.join(os.environ["REPO_ROOT"], "tests", "golden", "sense.npz"))
C, width, ntab, osf, ro, tr = g["params"]
p = SenseProblem(tuple(int(n) for n in g["N"]), g["coord"], np.asfortranarray(g["maps"]),
                 width=int(width), ntable=int(ntab), oversamp=float(osf))
B = NumpyBackend()
comm = TorchComm(B)
coils = list(coil_range(p.C, rank, world))
A = p.build_fused(B, coils=coils)
lam = float(g["lamda"])
AHA = ShardedNormalOperator(A, comm, lamda=lam)
x = B.copy_array(g["sense_x"])
y = B.zero_array(g["sense_x"].shape, np.dtype("complex64"))
AHA.eval(y, x)
x0 = np.zeros(g["sense_x"].shape, dtype=np.complex64, order="F")
B.cg(AHA, g["cg_b"].copy(order="F"), x0, maxiter=3)
np.savez(os.environ["OUT"] + ".%d.npz" % rank, y=y.to_host(), cg=x0, coils=np.array(coils))
dist.barrier()
dist.destroy_process_group()
"""


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_coil_range_partitions():
    from indigo_amd.dist import coil_range
    for C in (1, 3, 8, 32):
        for world in (1, 2, 3, 8):
            got = [c for r in range(world) for c in coil_range(C, r, world)]
            assert got == list(range(C))
            sizes = [len(coil_range(C, r, world)) for r in range(world)]
            assert max(sizes) - min(sizes) <= 1


def test_sharded_normal_operator_two_ranks(tmp_path):
    port = _free_port()
    out = str(tmp_path / "shard")
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   REPO_ROOT=ROOT, OUT=out, OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, "-c", WORKER], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    logs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            pytest.fail("distributed workers timed out")
        logs.append(o)
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)

    g = golden("sense")
    r0, r1 = np.load(out + ".0.npz"), np.load(out + ".1.npz")
    assert sorted(list(r0["coils"]) + list(r1["coils"])) == [0, 1, 2]
    # both ranks hold the same, complete result
    np.testing.assert_array_equal(r0["y"], r1["y"])
    assert rel_err(r0["y"], g["sense_AHAx"]) < 1e-5
    np.testing.assert_array_equal(r0["cg"], r1["cg"])
    assert rel_err(r0["cg"], g["cg_it3"]) < 1e-4


def _rendezvous_worker(rank, world, path, q, ready=True, delay=0.0):
    import time
    from indigo_amd.dist import cleanup_rendezvous, exchange_id
    try:
        time.sleep(delay)
        raw, all_ready = exchange_id(rank, world, lambda: bytes(range(128)), path, 128, timeout=20.0, ready=ready)
        q.put((rank, (raw, all_ready)))
        cleanup_rendezvous(rank, world, path)
    except Exception as e:          # noqa: BLE001
        q.put((rank, repr(e)))


def test_rccl_id_rendezvous_handshake(tmp_path):
    """the out-of-band channel of the C-ABI communicator (indigo_amd.dist.exchange_id): three sibling processes agree on
    rank 0's id; a rank whose peers never show up raises within the timeout instead of entering the collective"""
    import multiprocessing as mp
    from indigo_amd.dist import exchange_id
    ctx = mp.get_context("spawn")
    path = str(tmp_path / "id")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rendezvous_worker, args=(r, 3, path, q)) for r in range(3)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=60) for _ in procs)
    for p in procs:
        p.join(30)
    assert got == {r: (bytes(range(128)), True) for r in range(3)}
    assert not any(f.name.startswith("id") for f in tmp_path.iterdir()), "rendezvous files are cleaned up"
    import pytest
    with pytest.raises(RuntimeError, match="timed out"):
        exchange_id(1, 2, None, str(tmp_path / "nobody"), 128, timeout=0.3)
    with pytest.raises(RuntimeError, match="timed out"):
        exchange_id(0, 2, lambda: bytes(128), str(tmp_path / "alone"), 128, timeout=0.3)


def test_rendezvous_votes_on_readiness_and_ignores_stale_files(tmp_path):
    """(1) one rank that cannot load RCCL (ig_comm_preflight failed: ready=False) makes EVERY rank see all_ready == False --
    nobody enters ncclCommInitRank alone; (2) a complete set of left-over files of an earlier exchange under the same name
    (id + matching go-ahead + acknowledgements: two communicators built back to back, a reused INDIGO_COMM_ID_FILE) is not
    mistaken for this exchange: the go-ahead must carry the reader's fresh token"""
    import multiprocessing as mp
    from indigo_amd.dist import _nonce, _publish
    ctx = mp.get_context("spawn")
    path = str(tmp_path / "id")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rendezvous_worker, args=(r, 3, path, q, r != 2)) for r in range(3)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=60) for _ in procs)
    for p in procs:
        p.join(30)
    assert got == {r: (bytes(range(128)), False) for r in range(3)}, got
    # stale files: an old id, its go-ahead in the old (token-less) and in the new format, old acknowledgements
    stale = bytes(reversed(range(128)))
    _publish(path, stale)
    _publish(path + ".go", b":".join([_nonce(stale), b"1", b"0" * 32]))
    _publish(path + ".ack1", b":".join([_nonce(stale), b"0" * 32, b"1"]))
    procs = [ctx.Process(target=_rendezvous_worker, args=(r, 2, path, q, True, 0.5 if r == 0 else 0.0)) for r in range(2)]
    for p in procs:                     # rank 1 starts half a second BEFORE rank 0: it sees only the stale files at first
        p.start()
    got = dict(q.get(timeout=60) for _ in procs)
    for p in procs:
        p.join(30)
    assert got == {r: (bytes(range(128)), True) for r in range(2)}, got


ROUTE_WORKER = r"""
import os, sys
import numpy as np
sys.path.insert(0, os.environ["REPO_ROOT"])
import torch.distributed as dist
from oracle.np_backend import NumpyBackend
from indigo_amd.dist import ShardedNormalOperator, TorchComm, coil_range
from indigo_amd.sense import SenseProblem

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
g = np.load(os.path.join(os.environ["REPO_ROOT"], "tests", "golden", "sense.npz"))
C, width, ntab, osf, ro, tr = g["params"]
p = SenseProblem(tuple(int(n) for n in g["N"]), g["coord"], np.asfortranarray(g["maps"]),
                 width=int(width), ntable=int(ntab), oversamp=float(osf))
B = NumpyBackend()


class SlabComm(TorchComm):
    # a communicator that offers the slab route (as RcclComm does) on top of gloo: every collective is logged
    overlap = True

    def __init__(self, backend):
        super().__init__(backend)
        self.log = []

    def allreduce_(self, arr, force=False):
        self.log.append(("full", arr.size))
        super().allreduce_(arr, force)

    def allreduce_slab_(self, arr, lo, hi):
        self.log.append(("slab", hi - lo))
        super().allreduce_(arr[lo:hi])

    def max(self, value):
        self.log.append(("max", 1))
        return super().max(value)


comm = SlabComm(B)
coils = list(coil_range(p.C, rank, world))          # 3 coils on 3 ranks: one each
A = p.build_zpadfft(B, coils=coils, layout=1)       # a one-coil fused leaf in the per-coil layout: could go slab by slab
if rank == 1:
    A = p.build_fused(B, coils=coils)               # ... but rank 1's tree has no such leaf (as a 3-coil rank beside a 2-coil rank)
lam = float(g["lamda"])
AHA = ShardedNormalOperator(A, comm, lamda=lam, nslabs=2)
x = B.copy_array(g["sense_x"])
y = B.zero_array(g["sense_x"].shape, np.dtype("complex64"))
AHA.eval(y, x)
AHA.eval(y, x)
np.savez(os.environ["OUT"] + ".%d.npz" % rank, y=y.to_host(), route=np.array([AHA._route]), log=np.array([k for k, _ in comm.log]))
dist.barrier()
dist.destroy_process_group()
"""


def test_three_ranks_agree_on_one_allreduce_route(tmp_path):
    """ranks whose trees differ (two could all-reduce slab by slab, one cannot) must issue the SAME sequence of collectives:
    the first evaluation votes, every rank takes the whole-image route, and the result is the reference's"""
    port = _free_port()
    out = str(tmp_path / "route")
    procs = []
    for rank in range(3):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="3", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   REPO_ROOT=ROOT, OUT=out, OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, "-c", ROUTE_WORKER], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    logs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            pytest.fail("distributed workers timed out")
        logs.append(o)
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)
    g = golden("sense")
    res = [np.load(out + ".%d.npz" % r) for r in range(3)]
    for r in res:
        assert str(r["route"][0]) == "full"
        assert list(r["log"]) == ["full", "max", "full"], list(r["log"])
        assert rel_err(r["y"], g["sense_AHAx"]) < 1e-5


SPLIT_WORKER = r"""
import os, sys
import numpy as np
sys.path.insert(0, os.environ["REPO_ROOT"])
import torch.distributed as dist
from oracle.np_backend import NumpyBackend
from indigo_amd import operators as op
from indigo_amd.dist import ShardedNormalOperator, TorchComm, coil_range
from indigo_amd.sense import SenseProblem
from indigo_amd.util import rand64c

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
p = SenseProblem.synthetic((12, 10, 8), 8, nspokes=9, nreadout=24, oversamp=1.5, seed=21)
B = NumpyBackend()
coils = list(coil_range(p.C, rank, world))          # 8 coils on 3 ranks: 3 + 3 + 2
A = p.build_zpadfft(B, coils=coils)
leaves = []
def walk(n):
    if isinstance(n, op.ZpadFFT):
        leaves.append((n._C, n._layout))
    for c in getattr(n, "_children", None) or []:
        walk(c)
walk(A)
AHA = ShardedNormalOperator(A, TorchComm(B), lamda=0.25)
x = B.copy_array(rand64c(A.shape[1], 1, seed=3))
y = B.zero_array((A.shape[1], 1), np.dtype("complex64"))
AHA.eval(y, x)
np.savez(os.environ["OUT"] + ".%d.npz" % rank, y=y.to_host(), ncoils=len(coils), leaves=np.array(leaves), rows=A.shape[0])
dist.barrier()
dist.destroy_process_group()
"""


def test_eight_coils_on_three_ranks_take_interleaved_leaves(tmp_path, oracle_backend):
    """8 coils on 3 ranks are 3 + 3 + 2 (coil_range): a 3-coil rank evaluates ONE 4-wide coil-interleaved chunk whose fourth coil
    has zero weights -- the fused leaf with the binned adjoint and the fine table on the GPU -- instead of falling back to the
    per-coil layout; the 2-coil rank a 2-wide chunk.  The all-reduced normal operator equals the unsharded -O3 tree's."""
    from indigo_amd.sense import SenseProblem, normal_operator
    from indigo_amd.util import rand64c
    port = _free_port()
    out = str(tmp_path / "split")
    procs = []
    for rank in range(3):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="3", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   REPO_ROOT=ROOT, OUT=out, OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, "-c", SPLIT_WORKER], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    logs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            pytest.fail("distributed workers timed out")
        logs.append(o)
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)
    res = [np.load(out + ".%d.npz" % r) for r in range(3)]
    assert [int(r["ncoils"]) for r in res] == [3, 3, 2]
    assert [r["leaves"].tolist() for r in res] == [[[4, 2]], [[4, 2]], [[2, 2]]]          # (interleave width, grid layout) per chunk
    q = SenseProblem.synthetic((12, 10, 8), 8, nspokes=9, nreadout=24, oversamp=1.5, seed=21)
    assert [int(r["rows"]) for r in res] == [3 * q.T, 3 * q.T, 2 * q.T]                   # padding coils add no k-space rows
    B = oracle_backend
    B._scratch = None
    A = q.build_fused(B)
    x = rand64c(A.shape[1], 1, seed=3)
    y = B.zero_array((A.shape[1], 1), np.dtype("complex64"))
    normal_operator(A, lamda=0.25).eval(y, B.copy_array(x))
    for r in res:
        assert rel_err(r["y"], y.to_host()) < 1e-5
    B._scratch = None

"""The reference-side binding (integration/hip_backend_for_indigo.py = INTEGRATION.md section 2 as a file) is EXECUTABLE:

  * its ctypes prototype table equals indigo_amd/_lib.py:PROTOTYPES -- which tests/test_abi.py keeps equal to include/indigo_hip.h
    and to the exports of the built library -- so a stub that drifts from the header fails here;
  * the host-only shim of the ABI (tests/abi_shim/ig_shim.c: the same ig_* symbols as CPU loops, test infrastructure) compiles
    against the real header: a changed prototype is a compile error;
  * in the build container (where /root/reference exists; skipped elsewhere) the REFERENCE's own backend tests --
    indigo/backends/test_backends.py, read where it lies -- run on the binding under the reference's own Backend base class, on
    top of the shim: dndarray semantics, FFT, csr / exwrite csr, dia, BLAS-1, cgemm / csymm, cg, apgd, max, mem_usage.
No GPU, nothing of the reference is copied or written to."""
import ast
import ctypes
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STUB = os.path.join(ROOT, "integration", "hip_backend_for_indigo.py")
SHIM_SRC = os.path.join(ROOT, "tests", "abi_shim", "ig_shim.c")


def _stub_prototypes():
    """the stub's PROTOTYPES table, evaluated without importing the stub (it imports the reference's Backend)"""
    tree = ast.parse(open(STUB).read())
    env = {"C": ctypes}
    for node in tree.body:
        if isinstance(node, ast.Assign) and any(isinstance(t, (ast.Name, ast.Tuple)) for t in node.targets):
            names = [n.id for t in node.targets for n in ast.walk(t) if isinstance(n, ast.Name)]
            if "PROTOTYPES" in names or "c_f" in names:
                exec(compile(ast.Module([node], []), STUB, "exec"), env)
    return env["PROTOTYPES"]


def test_stub_prototypes_equal_the_abi():
    from indigo_amd import _lib
    protos = _stub_prototypes()
    assert len(protos) >= 20
    for name, (res, args) in protos.items():
        assert name in _lib.PROTOTYPES, name
        r2, a2 = _lib.PROTOTYPES[name]
        assert res == r2, (name, res, r2)
        assert len(args) == len(a2), (name, len(args), len(a2))
        for i, (u, v) in enumerate(zip(args, a2)):
            # (a POINTER(T) parameter may be declared as a bare void pointer on either side: same ABI)
            same = u == v or {u, v} <= {ctypes.c_void_p, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int)} and ctypes.c_void_p in (u, v)
            assert same, (name, i, u, v)


@pytest.fixture(scope="module")
def shim(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("abi_shim") / "libig_shim.so")
    r = subprocess.run(["gcc", "-std=c11", "-O2", "-Wall", "-Werror=implicit-function-declaration", "-fPIC", "-shared", "-I" + os.path.join(ROOT, "include"),
                        SHIM_SRC, "-o", out, "-lm"], capture_output=True, text=True)
    assert r.returncode == 0, "the shim no longer compiles against include/indigo_hip.h (a prototype drifted?):\n" + r.stderr
    return out


def test_shim_defines_every_symbol_the_stub_binds(shim):
    L = ctypes.CDLL(shim)
    for name in _stub_prototypes():
        assert hasattr(L, name), name
    assert L.ig_abi_version() == 1


@pytest.mark.skipif(not os.path.isdir("/root/reference/indigo"), reason="the reference tree exists in the build container only")
def test_reference_backend_tests_pass_on_the_binding(shim):
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "abi_shim", "run_reference_tests.py"), shim, "not get_backend and not bad_backend"],
                       capture_output=True, text=True, env=env, timeout=900)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    last = [ln for ln in r.stdout.splitlines() if " passed" in ln][-1]
    assert int(last.split(" passed")[0].split()[-1]) >= 4000 and " failed" not in last, last

#!/usr/bin/env python3
"""Runs the REFERENCE's own backend tests (indigo/backends/test_backends.py, read where it lies under /root/reference) with
integration/hip_backend_for_indigo.py registered as the reference's `hip` backend, on top of the host-only shim of the C ABI.

    python tests/abi_shim/run_reference_tests.py <path of the shim .so> [pytest -k expression]

Build container only: the reference tree does not exist on the GPU box, nothing of it is copied here, and nothing is written into it
(no bytecode, no pytest cache).  Third-party drift shims as in tests/golden/make_golden.py (this process only)."""
import importlib.util
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.dont_write_bytecode = True
os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
os.environ["INDIGO_HIP_LIB"] = os.path.abspath(sys.argv[1])
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

import make_golden                                   # noqa: E402  (install_shims: scipy .H, np.int, kaiser, numba / numexpr stand-ins)

make_golden.install_shims()
sys.path.insert(0, REF)
import indigo.backends                               # noqa: E402

spec = importlib.util.spec_from_file_location("indigo.backends.hip", os.path.join(ROOT, "integration", "hip_backend_for_indigo.py"))
mod = importlib.util.module_from_spec(spec)
sys.modules["indigo.backends.hip"] = mod
spec.loader.exec_module(mod)
indigo.backends.hip = mod
# the registry entry a maintainer would add (here patched in, the reference tree is read-only)
indigo.backends.available_backends = lambda: [mod.HipBackend]
_get = indigo.backends.get_backend
indigo.backends.get_backend = lambda name, **init: mod.HipBackend(**init) if name == "hip" else _get(name, **init)

import pytest                                        # noqa: E402

select = sys.argv[2] if len(sys.argv) > 2 else ""
args = [os.path.join(REF, "indigo", "backends", "test_backends.py"), "-q", "-p", "no:cacheprovider", "--rootdir", "/tmp", "-o", "addopts=",
        "--tb=line"]
if select:
    args += ["-k", select]
sys.exit(int(pytest.main(args)))

#!/usr/bin/env python3
"""Kernel-lab helper: run bench.py with entries of HipBackend.tuning overridden (the product reads no environment switches):
    python tools/run_with_tuning.py tiles64=64 xrows=False -- --config 3 --steps 5
    python tools/run_with_tuning.py opt:fft.kernels=1 -- --steps 20 --no-extras          (a plan option of the library)"""
import ast
import os
import runpy
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sep = sys.argv.index("--") if "--" in sys.argv else len(sys.argv)
over = {}
for kv in sys.argv[1:sep]:
    k, v = kv.split("=", 1)
    over[k] = ast.literal_eval(v)
import indigo_amd.backends.hip as H      # noqa: E402
_init = H.HipBackend.__init__


def patched(self, *a, **k):
    _init(self, *a, **k)
    self.tuning.update({k: v for k, v in over.items() if not k.startswith("opt:")})
    for k, v in over.items():          # opt:<name>=<value>: a plan option of the library (ig_set_option), e.g. opt:fft.kernels=1
        if k.startswith("opt:"):
            self.set_option(k[4:], v)


H.HipBackend.__init__ = patched
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[sep + 1:]
runpy.run_path(sys.argv[0], run_name="__main__")

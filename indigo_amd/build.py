"""Builds ``indigo_amd/lib/libindigo_hip.so`` (the C-ABI HIP library) in-tree.

    python -m indigo_amd.build [--force] [--verbose]

hipcc cross-compiles for gfx950 without a GPU present.  The shared object is
git-ignored but travels with the working tree to the GPU box.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(ROOT, "include")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(HERE, "lib", "obj")
LIBNAME = "libindigo_hip.so"

SOURCES = ["ig_fft_abd0.hip", "ig_fft_abd1.hip", "ig_fft_abd2.hip", "ig_fft_abd3.hip", "ig_fft.hip", "ig_spmm.hip", "ig_gridsep.hip", "ig_context.hip", "ig_blas.hip",
           "ig_comm.hip", "ig_interp.hip", "ig_dense.hip"]       # (the slow ones first: four compile side by side)
ARCH = "gfx950"
CXXFLAGS = [
    "--offload-arch=%s" % ARCH, "-O3", "-std=c++17", "-fPIC",
    "-munsafe-fp-atomics",          # float atomicAdd -> global_atomic_add_f32 (no CAS loop)
    "-Wall", "-Wno-unused-function",
    "-I" + INCLUDE, "-I" + CSRC,
]


# per-file additions
EXTRA_FLAGS = {
    # the SLP vectoriser pairs unrelated scalar butterfly operations into v_pk_* instructions and pays four v_mov per
    # pair to gather the operands (measured: 298 v_mov among 1942 instructions of the 512-point kernel)
    "ig_fft.hip": os.environ.get("INDIGO_FFT_FLAGS", "-fno-slp-vectorize").split(),
    "ig_fft_abd0.hip": ["-fno-slp-vectorize"], "ig_fft_abd1.hip": ["-fno-slp-vectorize"],
    "ig_fft_abd2.hip": ["-fno-slp-vectorize"], "ig_fft_abd3.hip": ["-fno-slp-vectorize"],
    # host-only double-precision arithmetic that must round like the reference's (numpy / numba): no fused multiply-add
    "ig_interp.hip": ["-ffp-contract=off"],
}


def source_hash():
    """sha256 (16 hex digits) over the kernel sources and headers the library is built from: recorded beside measurements
    that are only valid for these exact kernels (the rocprofv3 PMC summaries under profiles/, tools/pmc_summary.py)"""
    import hashlib
    h = hashlib.sha256()
    for name in sorted(os.listdir(CSRC)):
        if name.endswith((".hip", ".h", ".inc")):
            with open(os.path.join(CSRC, name), "rb") as f:
                h.update(name.encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


def lib_path():
    return os.path.join(LIBDIR, LIBNAME)


def _newer(path, deps):
    if not os.path.exists(path):
        return False
    t = os.path.getmtime(path)
    return all(os.path.getmtime(d) <= t for d in deps)


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def build(force=False, verbose=False):
    """Compile every HIP translation unit and link the shared library.  Returns its path."""
    os.makedirs(OBJDIR, exist_ok=True)
    headers = [os.path.join(INCLUDE, "indigo_hip.h")] + sorted(os.path.join(CSRC, h) for h in os.listdir(CSRC) if h.endswith((".h", ".inc")))
    hipcc = _hipcc()
    jobs = []
    objs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJDIR, src.replace(".hip", ".o"))
        objs.append(o)
        if force or not _newer(o, [s] + headers):
            jobs.append([hipcc] + CXXFLAGS + EXTRA_FLAGS.get(src, []) + ["-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError("build failed: %s\n%s" % (" ".join(cmd), r.stdout))
        if verbose and r.stdout.strip():
            print(r.stdout)

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))

    out = lib_path()
    if force or jobs or not _newer(out, objs):
        run([hipcc, "--offload-arch=%s" % ARCH, "-shared", "-fPIC", "-o", out] + objs + ["-ldl", "-lpthread"])
    return out


if __name__ == "__main__":
    p = build(force="--force" in sys.argv, verbose="--verbose" in sys.argv or "-v" in sys.argv)
    print(p)

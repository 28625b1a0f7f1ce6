"""ctypes binding of ``libindigo_hip.so`` (declared in ``include/indigo_hip.h``).

This module is the only place that knows symbol names and argument types; the
backend (``indigo_amd/backends/hip.py``) calls through :func:`lib`.  There is
no CPU fallback: if the shared object is missing, :func:`lib` raises.
"""
import ctypes
import os
import sys
from ctypes import (POINTER, c_char_p, c_double, c_float, c_int, c_int32, c_int64,
                    c_size_t, c_void_p)

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

IG_OK = 0
IG_H2D, IG_D2H, IG_D2D = 1, 2, 3

# name -> (restype, argtypes); mirrors include/indigo_hip.h one to one
PROTOTYPES = {
    "ig_abi_version":     (c_int, []),
    "ig_device_count":    (c_int, [POINTER(c_int)]),
    "ig_init":            (c_int, [c_int, POINTER(c_void_p)]),
    "ig_init_on_stream":  (c_int, [c_int, c_void_p, POINTER(c_void_p)]),
    "ig_destroy":         (None,  [c_void_p]),
    "ig_last_error":      (c_char_p, [c_void_p]),
    "ig_sync":            (c_int, [c_void_p]),
    "ig_stream":          (c_void_p, [c_void_p]),
    "ig_device_name":     (c_int, [c_void_p, c_char_p, c_size_t]),
    "ig_mem_info":        (c_int, [c_void_p, POINTER(c_size_t), POINTER(c_size_t)]),
    "ig_set_option":      (c_int, [c_void_p, c_char_p, c_int64]),
    "ig_library_bytes":   (c_int, [c_void_p, POINTER(c_size_t)]),
    "ig_malloc":          (c_int, [c_void_p, c_size_t, POINTER(c_void_p)]),
    "ig_free":            (c_int, [c_void_p, c_void_p]),
    "ig_probe_placement": (c_int, [c_void_p, c_void_p, c_size_t, POINTER(c_double)]),
    "ig_memset0":         (c_int, [c_void_p, c_void_p, c_size_t]),
    "ig_copy2d":          (c_int, [c_void_p, c_void_p, c_size_t, c_void_p, c_size_t, c_size_t, c_size_t, c_int]),
    "ig_event_create":    (c_int, [c_void_p, POINTER(c_void_p)]),
    "ig_event_record":    (c_int, [c_void_p]),
    "ig_event_elapsed_ms": (c_int, [c_void_p, c_void_p, POINTER(c_float)]),
    "ig_event_destroy":   (c_int, [c_void_p]),
    "ig_graph_begin":     (c_int, [c_void_p]),
    "ig_graph_end":       (c_int, [c_void_p, POINTER(c_void_p)]),
    "ig_graph_abort":     (c_int, [c_void_p]),
    "ig_graph_launch":    (c_int, [c_void_p]),
    "ig_graph_destroy":   (c_int, [c_void_p]),
    "ig_prof_enable":     (c_int, [c_void_p, c_int]),
    "ig_prof_report":     (c_int, [c_void_p, c_char_p, c_size_t]),
    "ig_caxpby":          (c_int, [c_void_p, c_int64, c_float, c_float, c_void_p, c_float, c_float, c_void_p]),
    "ig_cscal":           (c_int, [c_void_p, c_int64, c_float, c_float, c_void_p]),
    "ig_cdotc":           (c_int, [c_void_p, c_int64, c_void_p, c_void_p, POINTER(c_double)]),
    "ig_scnrm2sq":        (c_int, [c_void_p, c_int64, c_void_p, POINTER(c_double)]),
    "ig_cmax":            (c_int, [c_void_p, c_int64, c_float, c_void_p]),
    "ig_scalars":         (c_int, [c_void_p, POINTER(c_void_p), POINTER(c_int)]),
    "ig_cdotc_dev":       (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p]),
    "ig_scnrm2sq_dev":    (c_int, [c_void_p, c_int64, c_void_p, c_void_p]),
    "ig_scalar_ratio":    (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_double]),
    "ig_scalar_ratio_gated": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_double, c_void_p, c_void_p, c_double]),
    "ig_scalar_copy":     (c_int, [c_void_p, c_void_p, c_void_p, c_int64]),
    "ig_scalar_read":     (c_int, [c_void_p, c_void_p, c_int64, c_void_p]),
    "ig_cg_dot":          (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_float]),
    "ig_cg_step_r":       (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_double, c_void_p]),
    "ig_cg_step_xp":      (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "ig_caxpby_dev":      (c_int, [c_void_p, c_int64, c_void_p, c_float, c_void_p, c_void_p, c_float, c_void_p]),
    "ig_csum_cols":       (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_int64, c_float, c_float, c_float, c_float, c_void_p]),
    "ig_csum_il":         (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_float, c_float, c_float, c_float, c_void_p]),
    "ig_ccsrmm_il":       (c_int, [c_void_p, c_int64, c_int64, c_int64, c_int64,
                                   c_float, c_float, c_void_p, c_void_p, c_void_p,
                                   c_void_p, c_float, c_float, c_void_p, c_int64]),
    "ig_ccsrmm_il_rw":    (c_int, [c_void_p, c_int64, c_int64, c_int64, c_int64,
                                   c_float, c_float, c_void_p, c_void_p, c_void_p, c_void_p,
                                   c_void_p, c_float, c_float, c_void_p, c_int64]),
    "ig_ccsrmm_t_grid_il": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_int64,
                                    c_float, c_float, c_void_p, c_void_p, c_void_p,
                                    c_void_p, c_int64, c_void_p,
                                    c_void_p, c_int64, c_int64]),
    "ig_grid_bricks_count": (c_int, [c_int64, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int, c_int, c_int, c_void_p]),
    "ig_grid_bricks_fill": (c_int, [c_int64, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "ig_ccsrmm_t_bricks": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_float, c_float, c_void_p, c_void_p,
                                   c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_int64, c_int, c_int, c_void_p, c_int64,
                                   c_void_p, c_void_p, c_int64, c_int, c_int, c_int]),
    "ig_ccsrmm":          (c_int, [c_void_p, c_int, c_int, c_int64, c_int64, c_int64, c_int64,
                                   c_float, c_float, c_void_p, c_void_p, c_void_p,
                                   c_void_p, c_int64, c_float, c_float, c_void_p, c_int64]),
    "ig_ccsrmm_t":        (c_int, [c_void_p, c_int64, c_int64, c_int64, c_int64,
                                   c_float, c_float, c_void_p, c_void_p, c_void_p,
                                   c_void_p, c_int64, c_float, c_float, c_void_p, c_int64]),
    "ig_ccsrmm_t_grid":   (c_int, [c_void_p, c_int64, c_int64, c_int64, c_int64,
                                   c_float, c_float, c_void_p, c_void_p, c_void_p,
                                   c_void_p, c_int64, c_float, c_float, c_void_p, c_int64,
                                   c_void_p, c_int64, c_int64, c_void_p]),
    "ig_grid_slots_build": (c_int, [c_int64, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, POINTER(c_int64)]),
    "ig_ccsrmm_t_slots":  (c_int, [c_void_p, c_int64, c_int64, c_int64, c_float, c_float, c_void_p, c_void_p, c_void_p, c_int64, c_void_p,
                                   c_void_p, c_int64, c_int64, c_int, c_int, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_int, c_int, c_int]),
    "ig_ccsrmm_t_bricks_wide": (c_int, [c_void_p, c_int64, c_int64, c_float, c_float, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int64,
                                        c_void_p, c_int64, c_void_p, c_void_p]),
    "ig_ccsrmm_t_bricks_wide_grid": (c_int, [c_void_p, c_int64, c_int64, c_float, c_float, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int64,
                                             c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_int64, c_int, c_int, c_int]),
    "ig_fft_set_support_tile": (c_int, [c_void_p, c_int]),
    "ig_fft_set_axis_shift": (c_int, [c_void_p, c_int, c_int64]),
    "ig_csr_runs_build":  (c_int, [c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, POINTER(c_int)]),
    "ig_ccsrmm_xrows_runs": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_float, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p,
                                     c_void_p, c_int64, c_float, c_float, c_void_p, c_int64, c_void_p, c_int64]),
    "ig_ccsrmm_xrows":    (c_int, [c_void_p, c_int64, c_int64, c_int64, c_int64,
                                   c_float, c_float, c_void_p, c_void_p, c_void_p,
                                   c_void_p, c_int64, c_float, c_float, c_void_p, c_int64, c_void_p, c_int64]),
    "ig_ccsrmm_rowperm":  (c_int, [c_void_p, c_int64, c_int64, c_int64, c_int64,
                                   c_float, c_float, c_void_p, c_void_p, c_void_p,
                                   c_void_p, c_int64, c_float, c_float, c_void_p, c_int64, c_void_p]),
    "ig_csr_inspect":     (c_int, [c_void_p, c_void_p, c_int64, c_int64,
                                   POINTER(c_int64), POINTER(c_int64), POINTER(c_int)]),
    "ig_csr_transpose":   (c_int, [c_int64, c_int64, c_int64, c_void_p, c_void_p, c_void_p,
                                   c_void_p, c_void_p, c_void_p]),
    "ig_conemm":          (c_int, [c_void_p, c_int64, c_int64, c_int64, c_float, c_float, c_void_p, c_int64, c_float, c_float, c_void_p, c_int64]),
    "ig_cdiamm":          (c_int, [c_void_p, c_int, c_int64, c_int64, c_int64, c_int64, c_void_p, c_void_p, c_int64,
                                   c_float, c_float, c_void_p, c_int64, c_float, c_float, c_void_p, c_int64]),
    "ig_cgemm":           (c_int, [c_void_p, c_int, c_int, c_int64, c_int64, c_int64, c_float, c_float, c_void_p, c_int64,
                                   c_void_p, c_int64, c_float, c_float, c_void_p, c_int64]),
    "ig_interp3_count":   (c_int, [c_int64, POINTER(c_int64), c_double, c_void_p, c_void_p]),
    "ig_interp3_fill":    (c_int, [c_int64, POINTER(c_int64), c_double, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_int]),
    "ig_grid_support":    (c_int, [c_int64, c_void_p, c_int64, c_int64, c_int64, c_int, c_int, c_int, c_void_p]),
    "ig_fft_support_words": (c_int, [c_int64, POINTER(c_int), POINTER(c_int)]),
    "ig_fft_padded_axis_kind": (c_int, [c_int64, POINTER(c_int)]),
    "ig_interp3_sep_words": (c_int, [c_int]),
    "ig_interp3_sep":     (c_int, [c_int64, POINTER(c_int64), c_double, c_void_p, c_int64, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_double,
                                   c_int, c_void_p]),
    "ig_grid_gather_sep": (c_int, [c_void_p, c_int64, c_int64, c_int, c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int64, c_float, c_float, c_float, c_float,
                                   c_void_p, c_int64, c_void_p]),
    "ig_grid_gather_sep_group": (c_int, [c_int64, c_int]),
    "ig_grid_shares_count": (c_int, [c_int64, c_void_p, c_int, c_int64, c_int64, c_int64, c_int, c_int, c_void_p]),
    "ig_grid_shares_fill": (c_int, [c_int64, c_void_p, c_int, c_int64, c_int64, c_int64, c_int, c_int, c_void_p, c_void_p]),
    "ig_grid_scatter_sep": (c_int, [c_void_p, c_int64, c_int64, c_int, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int64,
                                    c_int, c_int, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_int, c_float, c_float]),
    "ig_interp3_fill_modulated": (c_int, [c_int64, POINTER(c_int64), c_double, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                          c_void_p, c_void_p, c_void_p, c_double]),
    "ig_fft_plan":        (c_int, [c_void_p, c_int, POINTER(c_int64), c_int64, POINTER(c_void_p), POINTER(c_size_t)]),
    "ig_fft_exec":        (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "ig_fft_describe":    (c_int, [c_void_p, c_char_p, c_size_t]),
    "ig_fft_inplace_workspace": (c_int, [c_void_p, POINTER(c_size_t)]),
    "ig_fft_destroy":     (c_int, [c_void_p]),
    "ig_fft_plan_padded": (c_int, [c_void_p, POINTER(c_int64), POINTER(c_int64), POINTER(c_int64), c_int64, c_int,
                                   POINTER(c_void_p), POINTER(c_size_t)]),
    "ig_fft_exec_padded": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p]),
    "ig_fft_exec_cropped": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p]),
    "ig_fft_exec_cropped_sum": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "ig_fft_exec_cropped_sum_slab": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int64, c_int64]),
    "ig_fft_exec_cropped_slab": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_int, c_int64, c_int64]),
    "ig_comm_preflight":  (c_int, []),
    "ig_comm_unique_id":  (c_int, [c_void_p]),
    "ig_comm_init_rank":  (c_int, [c_void_p, c_int, c_int, c_void_p, POINTER(c_void_p)]),
    "ig_comm_init_direct": (c_int, [c_void_p, c_int, c_int, c_char_p, c_size_t, c_double, POINTER(c_void_p)]),
    "ig_comm_info":       (c_int, [c_void_p, POINTER(c_int), POINTER(c_int), c_char_p, c_size_t]),
    "ig_allreduce_sum_f32": (c_int, [c_void_p, c_void_p, c_int64]),
    "ig_allreduce_sum_f32_side": (c_int, [c_void_p, c_void_p, c_int64]),
    "ig_comm_join":       (c_int, [c_void_p]),
    "ig_allreduce_max_f64_host": (c_int, [c_void_p, POINTER(c_double)]),
    "ig_allreduce_sum_f64_host": (c_int, [c_void_p, POINTER(c_double)]),
    "ig_comm_barrier":    (c_int, [c_void_p]),
    "ig_comm_destroy":    (c_int, [c_void_p]),
}


def lib_path():
    return os.environ.get("INDIGO_HIP_LIB") or os.path.join(_HERE, "lib", "libindigo_hip.so")


def lib():
    """Load (once) and return the ctypes handle.  Raises RuntimeError if the library is absent."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = lib_path()
    if not os.path.exists(path):
        raise RuntimeError(
            "indigo_amd: %s not found. Build it with `python -m indigo_amd.build` "
            "(needs hipcc); there is no CPU fallback." % path)
    # torch ships its own copy of the HIP runtime.  If the process is going to use torch
    # (torch.distributed for the multi-GPU all-reduce), torch must be imported FIRST so that
    # both share one runtime instance (same SONAME => the loader reuses it).
    if os.environ.get("INDIGO_HIP_WITH_TORCH", "0") == "1" and "torch" not in sys.modules:
        import torch  # noqa: F401
    handle = ctypes.CDLL(path)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(handle, name)      # AttributeError here means header and library disagree
        fn.restype = res
        fn.argtypes = args
    if handle.ig_abi_version() != 1:
        raise RuntimeError("indigo_amd: ABI version mismatch in %s" % path)
    _LIB = handle
    return _LIB


def last_error(ctx=None):
    msg = lib().ig_last_error(ctx)
    return msg.decode("utf-8", "replace") if msg else ""


def check(rc, ctx=None, what=""):
    """Status int -> RuntimeError carrying ig_last_error (reference: cuda.py:42-49)."""
    if rc != IG_OK:
        raise RuntimeError("%s failed (status %d): %s" % (what or "libindigo_hip call", rc, last_error(ctx)))


def device_count():
    n = c_int(0)
    lib().ig_device_count(ctypes.byref(n))
    return n.value

"""Backend registry (counterpart of indigo/backends/__init__.py:6-64).

The product ships exactly one backend, ``hip``.  `available_backends()` lists
the backend classes usable on this machine, filtered by the environment
variable INDIGO_TEST_BACKENDS like the reference; `get_backend(name, **init)`
instantiates by name and raises for unknown names.  The numpy oracle backend
lives under ``oracle/`` and is test infrastructure, not a registry entry.
"""
import logging
import os

log = logging.getLogger(__name__)


def available_backends():
    allow = os.environ.get("INDIGO_TEST_BACKENDS", "hip")
    backends = []
    if 'hip' in allow:
        try:
            from indigo_amd import _lib
            from indigo_amd.backends.hip import HipBackend
            if _lib.device_count() > 0:
                backends.append(HipBackend)
        except Exception as e:     # library not built / cannot load
            log.warning("couldn't find HIP backend: %s", e)
    return backends


def get_backend(name, **init):
    """Instantiate the requested backend (only 'hip' exists; no silent fallback)."""
    if name == 'hip':
        from indigo_amd.backends.hip import HipBackend
        return HipBackend(**init)
    raise ValueError("unrecognized backend: %s" % name)

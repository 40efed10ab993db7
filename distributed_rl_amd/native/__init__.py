"""ctypes loaders for the in-tree native libraries.

``host()`` → libdrl_host.so (built on demand with g++; works on any box).
``hip()``  → libdrl_hip.so (must have been built by ``__graft_entry__.build()``; raises loudly when
missing — GPU ops never fall back silently to PyTorch).
"""
from __future__ import annotations

import ctypes
import os
import threading
from ctypes import c_char_p, c_int, c_uint64, c_void_p, POINTER

from . import build as _build

_lock = threading.Lock()
_host = None
_hip = None


def host() -> ctypes.CDLL:
    global _host
    with _lock:
        if _host is None:
            path = _build.HOST_SO
            if not os.path.exists(path):
                path = _build.build_host()
            lib = ctypes.CDLL(path)
            try:
                _declare_host(lib)
            except AttributeError:                      # a library built from an older source: rebuild (seconds) and reload
                lib = ctypes.CDLL(_build.build_host(force=True))
                _declare_host(lib)
            _host = lib
    return _host


def hip_available() -> bool:
    return os.path.exists(_build.HIP_SO)


def hip() -> ctypes.CDLL:
    global _hip
    with _lock:
        if _hip is None:
            if not os.path.exists(_build.HIP_SO):
                raise RuntimeError("libdrl_hip.so is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                                   "(hipcc --offload-arch=gfx950) before using GPU ops")
            _hip = ctypes.CDLL(_build.HIP_SO)
            from . import hipdecl
            hipdecl.declare(_hip)
    return _hip


def _declare_host(lib) -> None:
    u64p = POINTER(c_uint64)
    sig = {
        "drl_ring_create": (c_void_p, [c_char_p, c_uint64, c_uint64]),
        "drl_ring_open": (c_void_p, [c_char_p]),
        "drl_ring_close": (None, [c_void_p, c_int]),
        "drl_ring_slot_bytes": (c_uint64, [c_void_p]),
        "drl_ring_n_slots": (c_uint64, [c_void_p]),
        "drl_ring_try_reserve": (c_void_p, [c_void_p, u64p]),
        "drl_ring_commit": (c_int, [c_void_p, c_uint64, c_uint64, c_uint64]),
        "drl_ring_push": (c_int, [c_void_p, c_void_p, c_uint64, c_uint64]),
        "drl_ring_try_acquire": (c_void_p, [c_void_p, u64p, u64p, u64p]),
        "drl_ring_release": (None, [c_void_p, c_uint64]),
        "drl_ring_size": (c_uint64, [c_void_p]),
        "drl_ring_pushed": (c_uint64, [c_void_p]),
        "drl_ring_rejected": (c_uint64, [c_void_p]),
        "drl_ring_drain": (c_uint64, [c_void_p]),
        "drl_ring_abandoned": (c_uint64, [c_void_p]),
        "drl_ring_set_abandon_grace": (None, [c_void_p, ctypes.c_double]),
        "drl_beat_free_pid": (c_int, [c_void_p, c_uint64]),
        "drl_param_create": (c_void_p, [c_char_p, c_uint64]),
        "drl_param_open": (c_void_p, [c_char_p]),
        "drl_param_close": (None, [c_void_p, c_int]),
        "drl_param_capacity": (c_uint64, [c_void_p]),
        "drl_param_version": (c_uint64, [c_void_p]),
        "drl_param_publish": (c_uint64, [c_void_p, c_void_p, c_uint64]),
        "drl_param_reset": (None, [c_void_p]),
        "drl_param_nbytes": (c_uint64, [c_void_p, c_uint64]),
        "drl_param_mirror_set": (None, [c_void_p, c_void_p, c_char_p, c_uint64, c_uint64]),
        "drl_param_mirror_clear": (None, [c_void_p]),
        "drl_param_mirror_get": (c_uint64, [c_void_p, c_void_p, c_void_p, u64p]),
        "drl_param_mirror_begin": (None, [c_void_p, c_uint64]),
        "drl_param_mirror_end": (None, [c_void_p, c_uint64]),
        "drl_param_mirror_latest": (c_uint64, [c_void_p]),
        "drl_param_mirror_check": (c_int, [c_void_p, c_uint64]),
        "drl_param_read": (c_uint64, [c_void_p, c_void_p, c_uint64, c_uint64, u64p]),
        "drl_beat_open": (c_void_p, [c_char_p, c_int]),
        "drl_beat_close": (None, [c_void_p, c_char_p]),
        "drl_beat_register": (c_int, [c_void_p, c_uint64, c_uint64, c_uint64]),
        "drl_beat_tick": (None, [c_void_p, c_int, c_uint64]),
        "drl_beat_unregister": (None, [c_void_p, c_int]),
        "drl_beat_snapshot": (c_int, [c_void_p, u64p, c_int]),
        "drl_shm_unlink": (c_int, [c_char_p]),
        "drl_planelog_observe": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
        "drl_host_version": (c_char_p, []),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args

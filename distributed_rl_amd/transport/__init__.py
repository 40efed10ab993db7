"""Experience queue + parameter register behind one interface.

Semantics preserved from the reference's two Redis keys:

* experience — unordered many-producer → single-consumer queue, at-most-once, atomic drain
  (``RPUSH`` / ``LRANGE+LTRIM``; ape_x/actor.py:97, ape_x/replay.py:26-29); fill level for
  back-pressure (``LLEN``; ape_x/learner.py:47-49); ``reset`` = ``DEL experience`` (replay.py:21).
* params — latest-value register: overwrite, readers poll, may skip versions, may read nothing before
  the first write (``SET``/``GET params``; ape_x/learner.py:83, ape_x/actor.py:52-56); ``reset`` = ``DEL params``.

Backends: ``shm`` (POSIX shared memory via libdrl_host — the default on one node), ``inproc``
(threads / tests), ``redis`` (optional plugin, only if the ``redis`` module is installed). Across GPUs
the parameter path is an RCCL broadcast (parallel/dist.py), not this module.
"""
from __future__ import annotations

import ctypes
import os
import threading
from collections import deque
from typing import Dict, List, Optional, Tuple

import numpy as np

from . import codec


class ExperienceChannel:
    def push(self, block: Dict[str, np.ndarray], kind: int, actor_id: int = 0, env_steps: int = 0) -> bool:
        raise NotImplementedError

    def drain(self, max_blocks: int = 1 << 30) -> List[Dict[str, np.ndarray]]:
        """Pop up to ``max_blocks`` blocks (decoded, owning their memory)."""
        raise NotImplementedError

    def size(self) -> int:
        raise NotImplementedError

    def reset(self) -> None:
        raise NotImplementedError

    def close(self) -> None:
        pass


class ParamChannel:
    # what ``publish2`` wants from a GPU learner: a host copy of the flat parameters, the device tensor itself, or both
    wants_host, wants_device = True, False

    def publish(self, flat: np.ndarray) -> int:
        raise NotImplementedError

    def publish2(self, host: Optional[np.ndarray], dev=None) -> int:
        """Publication by a learner that holds the parameters on a GPU (engine/learner.py): ``host`` is given when
        ``wants_host``, ``dev`` (a snapshot tensor the channel may read until it returns) when ``wants_device``."""
        return self.publish(host)

    def pull(self, have_version: int = 0) -> Optional[Tuple[int, np.ndarray]]:
        """Return ``(version, flat fp32 params)`` if something newer than ``have_version`` exists."""
        raise NotImplementedError

    def version(self) -> int:
        raise NotImplementedError

    def reset(self) -> None:
        raise NotImplementedError

    def close(self) -> None:
        pass


# ---- in-process -------------------------------------------------------------------------------------
class InProcExperience(ExperienceChannel):
    def __init__(self, max_blocks: int = 1024):
        self._q: deque = deque()
        self._lock = threading.Lock()
        self._max = max_blocks

    def push(self, block, kind, actor_id=0, env_steps=0):
        with self._lock:
            if len(self._q) >= self._max:
                return False
            self._q.append(bytes(codec.encode(block, kind, actor_id, env_steps)))
        return True

    def drain(self, max_blocks=1 << 30):
        with self._lock:
            n = min(max_blocks, len(self._q))
            raw = [self._q.popleft() for _ in range(n)]
        return [codec.decode(r) for r in raw]

    def size(self):
        return len(self._q)

    def reset(self):
        with self._lock:
            self._q.clear()


class InProcParams(ParamChannel):
    def __init__(self):
        self._lock = threading.Lock()
        self._v, self._flat = 0, None

    def publish(self, flat):
        with self._lock:
            self._v += 1
            self._flat = np.array(flat, dtype=np.float32, copy=True)
            return self._v

    def pull(self, have_version=0):
        with self._lock:
            if self._flat is None or self._v <= have_version:
                return None
            return self._v, self._flat.copy()

    def version(self):
        return self._v

    def reset(self):
        with self._lock:
            self._v, self._flat = 0, None


# ---- POSIX shared memory ----------------------------------------------------------------------------
class ShmExperience(ExperienceChannel):
    """Bounded MPSC ring of fixed-size slots in /dev/shm (csrc/host/drl_host.cpp)."""

    def __init__(self, name: str, create: bool = False, n_slots: int = 64, slot_bytes: int = 4 << 20):
        from .. import native
        self._lib = native.host()
        self.name = name
        self._owner = create
        if create:
            self._h = self._lib.drl_ring_create(name.encode(), n_slots, slot_bytes)
        else:
            self._h = self._lib.drl_ring_open(name.encode())
        if not self._h:
            raise OSError("cannot %s shm ring %r" % ("create" if create else "open", name))
        self.slot_bytes = int(self._lib.drl_ring_slot_bytes(self._h))
        self.n_slots = int(self._lib.drl_ring_n_slots(self._h))

    def push(self, block, kind, actor_id=0, env_steps=0):
        need = codec.block_bytes(block, kind)
        if need > self.slot_bytes:
            raise ValueError("experience block of %d bytes exceeds the ring slot size %d" % (need, self.slot_bytes))
        ticket = ctypes.c_uint64()
        ptr = self._lib.drl_ring_try_reserve(self._h, ctypes.byref(ticket))
        if not ptr:
            return False
        buf = (ctypes.c_ubyte * self.slot_bytes).from_address(ptr)
        used = codec.encode_into(buf, block, kind, actor_id, env_steps)
        self._lib.drl_ring_commit(self._h, ticket.value, used, kind)
        return True

    def acquire(self):
        """Zero-copy consumer access: returns ``(ticket, decoded views)`` or None. Call :meth:`release`."""
        t, nb, tag = ctypes.c_uint64(), ctypes.c_uint64(), ctypes.c_uint64()
        while True:
            ptr = self._lib.drl_ring_try_acquire(self._h, ctypes.byref(t), ctypes.byref(nb), ctypes.byref(tag))
            if not ptr:
                return None
            if nb.value >= codec.HEADER_BYTES:
                break
            self._lib.drl_ring_release(self._h, t.value)      # an empty commit (a producer that gave up its reservation)
        buf = (ctypes.c_ubyte * nb.value).from_address(ptr)
        return t.value, codec.decode(buf)

    def acquire_raw(self):
        """Zero-copy consumer access without decoding: ``(ticket, address, used bytes)`` of the oldest committed block, or
        None. The memory stays valid until :meth:`release` (engine/ingest.py copies it into pinned memory in one go)."""
        t, nb, tag = ctypes.c_uint64(), ctypes.c_uint64(), ctypes.c_uint64()
        while True:
            ptr = self._lib.drl_ring_try_acquire(self._h, ctypes.byref(t), ctypes.byref(nb), ctypes.byref(tag))
            if not ptr:
                return None
            if nb.value >= codec.HEADER_BYTES:
                return t.value, int(ptr), int(nb.value)
            self._lib.drl_ring_release(self._h, t.value)      # an empty commit (a producer that gave up its reservation)

    def release(self, ticket: int) -> None:
        self._lib.drl_ring_release(self._h, ticket)

    def drain(self, max_blocks=1 << 30):
        out = []
        while len(out) < max_blocks:
            got = self.acquire()
            if got is None:
                break
            t, blk = got
            out.append({k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in blk.items()})
            self.release(t)
        return out

    def size(self):
        return int(self._lib.drl_ring_size(self._h))

    def stats(self):
        return dict(size=self.size(), pushed=int(self._lib.drl_ring_pushed(self._h)),
                    rejected=int(self._lib.drl_ring_rejected(self._h)), abandoned=int(self._lib.drl_ring_abandoned(self._h)))

    def set_abandon_grace(self, seconds: float) -> None:
        """Consumer side: how long the head slot may stay reserved-but-uncommitted before its owner is checked for life
        (a producer killed between reserve and commit; csrc/host/drl_host.cpp drl_ring_try_acquire)."""
        self._lib.drl_ring_set_abandon_grace(self._h, float(seconds))

    def reset(self):
        self._lib.drl_ring_drain(self._h)

    def close(self):
        if self._h:
            self._lib.drl_ring_close(self._h, 1 if self._owner else 0)
            self._h = None


class ShmParams(ParamChannel):
    def __init__(self, name: str, create: bool = False, capacity: int = 64 << 20):
        from .. import native
        self._lib = native.host()
        self.name, self._owner = name, create
        self._h = self._lib.drl_param_create(name.encode(), capacity) if create else self._lib.drl_param_open(name.encode())
        if not self._h:
            raise OSError("cannot %s shm param register %r" % ("create" if create else "open", name))
        self.capacity = int(self._lib.drl_param_capacity(self._h))
        self._scratch = np.empty(self.capacity // 4, dtype=np.float32)

    def publish(self, flat):
        a = np.ascontiguousarray(flat, dtype=np.float32)
        v = int(self._lib.drl_param_publish(self._h, a.ctypes.data_as(ctypes.c_void_p), a.nbytes))
        if v == 0:
            raise ValueError("parameter blob of %d bytes exceeds register capacity %d" % (a.nbytes, self.capacity))
        return v

    def pull(self, have_version=0):
        nb = ctypes.c_uint64()
        v = int(self._lib.drl_param_read(self._h, self._scratch.ctypes.data_as(ctypes.c_void_p), self._scratch.nbytes,
                                         have_version, ctypes.byref(nb)))
        if v == 0 or v == (1 << 64) - 1:
            return None
        return v, self._scratch[:nb.value // 4].copy()

    def version(self):
        return int(self._lib.drl_param_version(self._h))

    def reset(self):
        self._lib.drl_param_reset(self._h)

    def close(self):
        if self._h:
            self._lib.drl_param_close(self._h, 1 if self._owner else 0)
            self._h = None


# ---- factory ----------------------------------------------------------------------------------------
_INPROC: Dict[str, Tuple[InProcExperience, InProcParams]] = {}


def shm_names(run_id: str) -> Tuple[str, str, str]:
    rid = "".join(c if c.isalnum() else "_" for c in run_id)
    return "/drl_%s_exp" % rid, "/drl_%s_par" % rid, "/drl_%s_hb" % rid


def connect(kind: str = "shm", run_id: str = "default", create: bool = False, host: str = "localhost",
            n_slots: int = 64, slot_bytes: int = 4 << 20, param_capacity: int = 64 << 20
            ) -> Tuple[ExperienceChannel, ParamChannel]:
    """``create=True`` is the learner side (owns and resets the channels, like the reference's
    ``DEL params`` / ``DEL experience`` at start-up, ape_x/learner.py:37, ape_x/replay.py:21)."""
    if kind == "inproc":
        if run_id not in _INPROC or create:
            _INPROC[run_id] = (InProcExperience(), InProcParams())
        return _INPROC[run_id]
    if kind == "shm":
        en, pn, _ = shm_names(run_id)
        return (ShmExperience(en, create, n_slots, slot_bytes), ShmParams(pn, create, param_capacity))
    if kind == "redis":
        from .redis_plugin import RedisExperience, RedisParams
        exp, par = RedisExperience(host), RedisParams(host)
        if create:
            exp.reset()
            par.reset()
        return exp, par
    raise ValueError("unknown transport %r" % kind)

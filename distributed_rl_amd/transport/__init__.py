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
        # 0: the consumer gave the reservation up while we were filling it (a stall longer than its grace period): not
        # published - the caller keeps the block and pushes it again, as after a full ring
        return bool(self._lib.drl_ring_commit(self._h, ticket.value, used, kind))

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
    """Latest-value parameter register in POSIX shared memory (seqlock double buffer, csrc/host/drl_host.cpp) - and, when
    the learner runs on a GPU, a DEVICE MIRROR of it: the two newest versions in a double buffer in the learner GPU's
    memory, exported by hipIpc handle through the register's header page. Actors whose inference engine sits on the same
    physical GPU copy device to device (``pull_tensor``) instead of D2H -> shm -> H2D (1.8 - 5 ms per
    pull on one GPU); everybody else reads the host register as before (``pull``). Reference semantics: ``SET params`` /
    ``GET params``, ape_x/learner.py:83, ape_x/actor.py:52-56."""

    device_mirror = True           # pull_tensor takes the reader's device

    def __init__(self, name: str, create: bool = False, capacity: int = 64 << 20):
        from .. import native
        self._lib = native.host()
        self.name, self._owner = name, create
        self._h = self._lib.drl_param_create(name.encode(), capacity) if create else self._lib.drl_param_open(name.encode())
        if not self._h:
            raise OSError("cannot %s shm param register %r" % ("create" if create else "open", name))
        self.capacity = int(self._lib.drl_param_capacity(self._h))
        self._scratch = np.empty(self.capacity // 4, dtype=np.float32)
        # device mirror: writer side (ptr of the double buffer) / reader side (opened handle, staging tensor)
        self._mir_ptr = None
        self._mir_bytes = 0
        self._mir_open = None          # (exporter pid, device ptr, slot bytes)
        self._mir_dev = None
        self._mir_stage = None
        self._mir_off = os.environ.get("DRL_PARAM_MIRROR", "1") == "0"
        self.mirror_pulls = 0
        self.host_pulls = 0

    def publish(self, flat):
        a = np.ascontiguousarray(flat, dtype=np.float32)
        v = int(self._lib.drl_param_publish(self._h, a.ctypes.data_as(ctypes.c_void_p), a.nbytes))
        if v == 0:
            raise ValueError("parameter blob of %d bytes exceeds register capacity %d" % (a.nbytes, self.capacity))
        return v

    # ---- device mirror, writer (the GPU learner's publisher) -------------------------------------------------------
    def enable_device_mirror(self, device, nbytes: int) -> bool:
        """Allocate + export the double buffer on ``device``; from now on ``publish2`` wants the device snapshot too."""
        if self._mir_off or self._mir_ptr is not None:
            return self._mir_ptr is not None
        import torch
        from .. import native
        hip = native.hip()
        dev = torch.device(device)
        dev = torch.device("cuda", dev.index if dev.index is not None else torch.cuda.current_device())
        slot = (int(nbytes) + 255) // 256 * 256
        ptr, handle, bus = ctypes.c_void_p(), (ctypes.c_ubyte * 64)(), ctypes.create_string_buffer(64)
        with torch.cuda.device(dev):
            if hip.drl_ipc_alloc(2 * slot, ctypes.byref(ptr), handle) != 0:
                return False
            hip.drl_device_bus_id(dev.index, bus)
        self._mir_ptr, self._mir_bytes, self._mir_dev = ptr.value, slot, dev
        self._lib.drl_param_mirror_set(self._h, handle, bus.value, slot, os.getpid())
        self.wants_device = True
        return True

    def publish2(self, host, dev=None):
        v = self.publish(host)
        if self._mir_ptr is not None and dev is not None:
            import torch
            from .. import native
            nb = dev.numel() * dev.element_size()
            assert dev.is_contiguous() and nb <= self._mir_bytes
            self._lib.drl_param_mirror_begin(self._h, v)
            st = torch.cuda.current_stream(dev.device)
            native.hip().drl_copy_d2d_async(ctypes.c_void_p(self._mir_ptr + (v & 1) * self._mir_bytes), ctypes.c_void_p(dev.data_ptr()),
                                            nb, ctypes.c_void_p(st.cuda_stream))
            st.synchronize()                                  # the version is announced when the slot holds it completely
            self._lib.drl_param_mirror_end(self._h, v)
        return v

    # ---- device mirror, reader (an actor with a GPU engine) ----------------------------------------------------------
    def _mirror_reader(self, device):
        """Open the exporter's double buffer if it lives on the physical GPU of ``device``; None if there is no usable one."""
        import torch
        from .. import native
        handle, bus, sb = (ctypes.c_ubyte * 64)(), ctypes.create_string_buffer(64), ctypes.c_uint64()
        pid = int(self._lib.drl_param_mirror_get(self._h, handle, bus, ctypes.byref(sb)))
        if not pid:
            return None
        if self._mir_open is not None and self._mir_open[0] == pid:
            return self._mir_open
        hip = native.hip()
        idx = torch.device(device).index
        mine = ctypes.create_string_buffer(64)
        hip.drl_device_bus_id(idx, mine)
        if mine.value != bus.value or pid == os.getpid():       # another GPU (or our own export): the host register serves
            return None
        ptr = ctypes.c_void_p()
        with torch.cuda.device(idx):
            if self._mir_open is not None:                       # the learner was restarted: a new export
                hip.drl_ipc_close(ctypes.c_void_p(self._mir_open[1]))
                self._mir_open = None
            if hip.drl_ipc_open(handle, ctypes.byref(ptr)) != 0:
                return None
        self._mir_open = (pid, ptr.value, int(sb.value))
        return self._mir_open

    def pull_tensor(self, have_version=0, device=None):
        """``(version, flat fp32 tensor)`` newer than ``have_version`` - a DEVICE tensor copied D2D out of the learner's
        mirror when that lives on this process' GPU, else the host register's array (``pull``)."""
        import torch
        if not self._mir_off and torch.cuda.is_available():
            dev = torch.device(device if device is not None else "cuda")
            if dev.type != "cuda":
                return self._host_pull(have_version)
            dev = torch.device("cuda", dev.index if dev.index is not None else torch.cuda.current_device())
            m = self._mirror_reader(dev)
            if m is not None:
                from .. import native
                _, ptr, slot = m
                for _ in range(4):
                    v = int(self._lib.drl_param_mirror_latest(self._h))
                    if v == 0 or v <= have_version:
                        return None
                    nb = int(self._lib.drl_param_nbytes(self._h, v))
                    if nb == 0 or nb > slot:
                        break
                    # two staging tensors, alternating: the one written now was handed out two pulls ago (its reader, the
                    # caller's copy into its own parameters on ITS stream, is long done)
                    if self._mir_stage is None or self._mir_stage[0].numel() * 4 != nb or self._mir_stage[0].device != dev:
                        self._mir_stage = [torch.empty(nb // 4, dtype=torch.float32, device=dev) for _ in range(2)]
                        torch.cuda.current_stream(dev).synchronize()
                    stage = self._mir_stage[self.mirror_pulls & 1]
                    # a stream of its own: the wait below is for this copy only, not for whatever inference work the caller
                    # has queued on its stream
                    if getattr(self, "_mir_stream", None) is None:
                        self._mir_stream = torch.cuda.Stream(dev, priority=-1)
                    st = self._mir_stream
                    native.hip().drl_copy_d2d_async(ctypes.c_void_p(stage.data_ptr()), ctypes.c_void_p(ptr + (v & 1) * slot),
                                                    nb, ctypes.c_void_p(st.cuda_stream))
                    st.synchronize()
                    if self._lib.drl_param_mirror_check(self._h, v):      # the writer did not start on this slot meanwhile
                        self.mirror_pulls += 1
                        return v, stage
        return self._host_pull(have_version)

    def _host_pull(self, have_version):
        got = self.pull(have_version)
        if got is not None:
            self.host_pulls += 1
        return got

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
            if self._mir_open is not None or self._mir_ptr is not None:
                try:
                    from .. import native
                    hip = native.hip()
                    if self._mir_open is not None:
                        hip.drl_ipc_close(ctypes.c_void_p(self._mir_open[1]))
                    if self._mir_ptr is not None:
                        self._lib.drl_param_mirror_clear(self._h)
                        hip.drl_ipc_free(ctypes.c_void_p(self._mir_ptr))
                except Exception:  # noqa: BLE001 - shutting down
                    pass
                self._mir_open = self._mir_ptr = None
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

"""Parameter sync by collective broadcast ("RCCL broadcast from learner rank 0 to actor GPUs").

On one node actors read the learner's weights from the shared-memory register (transport.ShmParams). Actors that own a
GPU of their own - other GPUs of the node, or other nodes - join ONE ``torch.distributed`` group with the learner rank
(backend ``nccl`` = RCCL over xGMI / the NIC on GPUs, ``gloo`` on CPU) and receive the flat fp32 parameter buffer by
broadcast straight into device memory: no host staging, no pickle, 13 MB (Ape-X) / 32 MB (R2D2) per publication.

The collective is blocking by nature and actors are asynchronous by design, so every receiver runs a small thread
that sits in ``broadcast`` waiting for the next publication (header = [version, numel]; version < 0 = shut down) and
swaps the payload in under a lock; ``pull(have_version)`` never blocks. The learner publishes every K steps, exactly
like the reference's ``SET params`` (ape_x/learner.py:83), but without touching the host.

Give it its OWN process group (``group=torch.distributed.new_group(...)``) when the learner captures its gradient
all-reduce into the step graph: eager collectives of a group whose RCCL stream is being captured make the group's
watchdog abort the process (``Dist.quiesce`` in parallel/dist.py explains the mechanism).

Next to data-parallel learners the channel is a LANE of the launcher's world (runtime/placement.py, parallel/world.py):
``group`` = the lane's sub-group, ``src_rank`` = the WORLD rank of the lane's learner. ``FanoutParams`` is the learner's
side of a whole placement: the shard's register for co-located actors plus every lane that starts at this rank.
"""
from __future__ import annotations

import threading
from typing import Optional, Tuple

import numpy as np
import torch
import torch.distributed as td

from . import ParamChannel


class CollectiveParams(ParamChannel):
    def __init__(self, numel: int, src_rank: int = 0, group=None, device: Optional[torch.device] = None,
                 start: bool = True):
        """``src_rank``: rank of the publisher in the DEFAULT group (what ``torch.distributed.broadcast`` expects)."""
        assert td.is_initialized(), "init the process group first (parallel.dist.Dist / parallel.world.join_world)"
        self.numel, self.src, self.group = int(numel), int(src_rank), group
        self.rank = td.get_rank()
        self.is_source = self.rank == self.src
        self.device = torch.device(device) if device is not None else torch.device("cpu")
        self.wants_device = self.device.type == "cuda"       # the GPU learner hands over its device snapshot: no D2H/H2D
        self.wants_host = not self.wants_device
        self.last_publish_us = 0.0
        self._hdr = torch.zeros(2, dtype=torch.int64, device=self.device)
        self._stage = torch.zeros(self.numel, dtype=torch.float32, device=self.device)
        self._cur = torch.zeros(self.numel, dtype=torch.float32, device=self.device)
        self._v = 0
        self._lock = threading.Lock()
        self._thread = None
        self._closed = False
        if not self.is_source and start:
            self._thread = threading.Thread(target=self._recv_loop, name="param-bcast", daemon=True)
            self._thread.start()

    # ---- learner side -----------------------------------------------------------------------------
    def publish(self, flat) -> int:
        """Source rank only: broadcast the header, then the payload (device tensor or numpy array)."""
        assert self.is_source, "only the source rank publishes"
        t = flat if isinstance(flat, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(flat, dtype=np.float32))
        assert t.numel() == self.numel
        self._v += 1
        self._hdr[0], self._hdr[1] = self._v, self.numel
        td.broadcast(self._hdr, src=self.src, group=self.group)
        self._stage.copy_(t.reshape(-1).to(self.device, torch.float32), non_blocking=True)
        td.broadcast(self._stage, src=self.src, group=self.group)
        return self._v

    def publish2(self, host, dev=None) -> int:
        return self.publish(dev if (self.wants_device and dev is not None) else host)

    # ---- actor side -------------------------------------------------------------------------------
    def _recv_loop(self) -> None:
        if self.device.type == "cuda":
            torch.cuda.set_device(self.device)
        while not self._closed:
            td.broadcast(self._hdr, src=self.src, group=self.group)
            v, n = int(self._hdr[0].item()), int(self._hdr[1].item())
            if v < 0:
                break
            assert n == self.numel, "parameter buffer size changed (%d != %d)" % (n, self.numel)
            td.broadcast(self._stage, src=self.src, group=self.group)
            if self.device.type == "cuda":
                torch.cuda.current_stream().synchronize()
            with self._lock:
                self._cur.copy_(self._stage)
                self._v = v

    def step(self) -> bool:
        """Receive ONE publication synchronously (receivers created with start=False; used by tests / lock-step runs)."""
        td.broadcast(self._hdr, src=self.src, group=self.group)
        v = int(self._hdr[0].item())
        if v < 0:
            return False
        td.broadcast(self._stage, src=self.src, group=self.group)
        with self._lock:
            self._cur.copy_(self._stage)
            self._v = v
        return True

    def pull(self, have_version: int = 0) -> Optional[Tuple[int, np.ndarray]]:
        with self._lock:
            if self._v <= have_version:
                return None
            return self._v, self._cur.detach().cpu().numpy().copy()

    def pull_tensor(self, have_version: int = 0) -> Optional[Tuple[int, torch.Tensor]]:
        """As ``pull`` but the payload stays where the broadcast put it (the actor's GPU with the nccl backend): a private
        copy on ``self.device``, no D2H + H2D round trip through numpy. Used by actors that run a device engine
        (reference: ape_x/actor.py:69-76 loads the pulled state_dict into the GPU model)."""
        with self._lock:
            if self._v <= have_version:
                return None
            return self._v, self._cur.detach().clone()

    def version(self) -> int:
        return self._v

    def reset(self) -> None:
        with self._lock:
            self._v = 0 if not self.is_source else self._v

    def close(self) -> None:
        """Source: tell the receivers to stop. Receivers: wait for the sentinel."""
        if self._closed:
            return
        self._closed = True
        if self.is_source:
            self._hdr[0], self._hdr[1] = -1, 0
            td.broadcast(self._hdr, src=self.src, group=self.group)
        elif self._thread is not None:
            self._thread.join(timeout=30)


class FanoutParams(ParamChannel):
    """Learner side of a placement: ONE publication reaches the shard's register (actors on this learner's GPU or on
    the CPU) and every broadcast lane that starts at this rank (actors on other GPUs). Versions are the register's when
    there is one (they are what co-located actors compare against), else the first lane's."""

    def __init__(self, register: Optional[ParamChannel], lanes):
        self.register, self.lanes = register, list(lanes)
        self.wants_host = register is not None or any(l.wants_host for l in self.lanes)
        self.wants_device = any(l.wants_device for l in self.lanes)

    def publish(self, flat) -> int:
        host = flat if not isinstance(flat, torch.Tensor) else flat.detach().float().cpu().numpy()
        return self.publish2(host, flat if isinstance(flat, torch.Tensor) and flat.is_cuda else None)

    def publish2(self, host, dev=None) -> int:
        v = 0
        if self.register is not None:
            v = self.register.publish(host)
        for lane in self.lanes:
            lv = lane.publish2(host, dev)
            v = v or lv
        return v

    def pull(self, have_version: int = 0):
        return self.register.pull(have_version) if self.register is not None else None

    def version(self) -> int:
        return self.register.version() if self.register is not None else (self.lanes[0].version() if self.lanes else 0)

    def reset(self) -> None:
        if self.register is not None:
            self.register.reset()

    def close(self) -> None:
        for lane in self.lanes:
            lane.close()
        if self.register is not None:
            self.register.close()

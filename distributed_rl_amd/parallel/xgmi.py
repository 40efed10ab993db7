"""Direct all-reduce over the xGMI mesh (csrc/hip/xgmi_ar.hip).

The reference has no gradient collective at all (single learner, ape_x/learner.py:11); SURVEY M7 asks for a
latency-optimal one: the gradients (13-32 MB) are small, an MI355X node is a full mesh, so every GPU exchanges 1/W of the
buffer with each of its peers at once instead of walking a ring. :class:`XgmiAllReduce` owns the gradient buffer (the
engines build their flat gradient ``ParamPack`` on top of it), exchanges the IPC handles over an existing
``torch.distributed`` group once, and afterwards a collective is ONE kernel launch on the caller's stream - it is
captured into the step's hipGraph like any compute kernel.

Every rank must call :meth:`all_reduce` with the same arguments in the same order per channel; collectives that may be in
flight at the same time (the engines' big FC slice under the conv backward, then the small conv slice) use different
channels.
"""
from __future__ import annotations

import ctypes
import os
from typing import Optional

import torch
import torch.distributed as td

from .. import native

N_CHANNELS = 4
MAX_WORLD = 8
MEM_KINDS = {"coarse": 0, "fine": 1, "uncached": 2}      # where the communication buffer lives (xgmi_ar.hip drl_xar_create2)


class _DevMem:
    """Zero-copy view of a raw device pointer for ``torch.as_tensor`` (CUDA array interface, version 2)."""

    def __init__(self, ptr: int, numel: int, owner):
        self.owner = owner
        self.__cuda_array_interface__ = {"shape": (numel,), "typestr": "<f4", "data": (ptr, False), "version": 2,
                                         "strides": None}


class XgmiAllReduce:
    def __init__(self, numel: int, device: torch.device, rank: int, world: int, group=None, n_blocks: int = 32,
                 timeout_s: float = 20.0, mem_kind: str = "coarse"):
        if world > MAX_WORLD:
            raise ValueError("xGMI all-reduce: at most %d ranks (one node)" % MAX_WORLD)
        self.lib = native.hip()
        device = torch.device(device)
        if device.index is None:
            device = torch.device("cuda", torch.cuda.current_device())
        self.rank, self.world, self.device = rank, world, device
        self.timeout_s = float(timeout_s)
        self.mem_kind, self.n_blocks = mem_kind, int(n_blocks)
        self.cap = (numel + 63) // 64 * 64
        self._ctx = ctypes.c_void_p()
        self.flat, self._base = None, 0
        # Set-up is a sequence of COLLECTIVE steps (handle exchange, agreement). A rank whose local part fails must still
        # take every collective step, or its peers would pair their next collective with a different one of ours: failures
        # are recorded, agreed on, and raised on ALL ranks together afterwards.
        ok, why = True, ""
        mine = ctypes.create_string_buffer(128)
        vis = os.environ.get("HIP_VISIBLE_DEVICES", "") + "|" + os.environ.get("ROCR_VISIBLE_DEVICES", "")
        with torch.cuda.device(self.device):
            try:
                self._ck(self.lib.drl_xar_create2(rank, world, self.cap, n_blocks, timeout_s, MEM_KINDS[mem_kind],
                                                      ctypes.byref(self._ctx)), "create")
                self._ck(self.lib.drl_xar_handles(self._ctx, mine), "handles")
            except Exception as e:                                 # noqa: BLE001
                ok, why = False, repr(e)
            every = [None] * world
            if world > 1:
                td.all_gather_object(every, (bytes(mine.raw) if ok else None, self.device.index, vis), group=group)
            if ok:
                try:
                    for p in range(world):
                        if p == rank:
                            continue
                        handles, dev_p, vis_p = every[p]
                        if handles is None:
                            raise RuntimeError("peer %d could not allocate its buffers" % p)
                        # the peer's device ordinal is only meaningful when both processes enumerate the same devices
                        self._ck(self.lib.drl_xar_open(self._ctx, p, ctypes.c_char_p(handles), dev_p if vis_p == vis else -1),
                                 "open(peer %d)" % p)
                    ptr = self.lib.drl_xar_data_ptr(self._ctx)
                    self.flat = torch.as_tensor(_DevMem(ptr, self.cap, self), device=self.device)
                    if self.flat.data_ptr() != ptr or self.flat.dtype != torch.float32:
                        raise RuntimeError("could not wrap the communication buffer as a tensor")
                    self._base = ptr
                except Exception as e:                             # noqa: BLE001
                    ok, why = False, repr(e)
            if world > 1:
                # agreement; it is also the barrier "nobody launches before every peer has mapped every buffer"
                votes = [None] * world
                td.all_gather_object(votes, (ok, why), group=group)
                bad = [(p, w) for p, (o, w) in enumerate(votes) if not o]
                if bad:
                    ok, why = False, "rank %d: %s" % bad[0]
        if not ok:
            self.close()
            raise RuntimeError("xgmi all-reduce set-up failed (%s)" % why)

    def _ck(self, rc: int, what: str) -> None:
        if rc != 0:
            raise RuntimeError("xgmi all-reduce %s failed: error %d" % (what, rc))

    def owns(self, t: torch.Tensor) -> bool:
        a = t.data_ptr()
        return (t.dtype == torch.float32 and t.is_contiguous() and self._base <= a
                and a + t.numel() * 4 <= self._base + self.cap * 4)

    def all_reduce(self, t: torch.Tensor, channel: int = 0) -> None:
        """In-place SUM over ranks of ``t`` (a contiguous slice of ``self.flat`` starting at a multiple of 4 elements), on
        the current stream."""
        assert self.owns(t), "tensor is not a slice of the xGMI communication buffer"
        off = (t.data_ptr() - self._base) // 4
        n = min((t.numel() + 3) // 4 * 4, self.cap - off)       # the padding is zero on every rank
        self._ck(self.lib.drl_xar_all_reduce(self._ctx, off, n, channel, torch.cuda.current_stream(self.device).cuda_stream),
                 "launch")

    def set_blocks(self, n_blocks: int) -> None:
        """Workgroups per collective from the next launch on (1..64). Every rank must switch at the same point of its
        launch sequence: block b pairs up with block b of every peer."""
        self._ck(self.lib.drl_xar_set_blocks(self._ctx, int(n_blocks)), "set_blocks")
        self.n_blocks = int(n_blocks)

    def set_timeout(self, seconds: float) -> None:
        """Bound of every flag poll inside the kernel (launches issued after this call)."""
        self._ck(self.lib.drl_xar_set_timeout(self._ctx, float(seconds)), "set_timeout")
        self.timeout_s = float(seconds)

    def err_ptr(self) -> int:
        """Device address of the timeout word (a guard of the optimizer launch, ops.hip.grad_sumsq)."""
        return int(self.lib.drl_xar_err_ptr(self._ctx) or 0)

    def error(self) -> int:
        """0 = fine; otherwise the barrier (1..3) whose poll timed out. Synchronises the device."""
        with torch.cuda.device(self.device):
            return int(self.lib.drl_xar_error(self._ctx))

    def close(self) -> None:
        if self._ctx:
            with torch.cuda.device(self.device):
                self.lib.drl_xar_destroy(self._ctx)
            self._ctx = ctypes.c_void_p()

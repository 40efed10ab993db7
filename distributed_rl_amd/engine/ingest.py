"""Asynchronous experience ingest for the GPU learner: a host thread + a dedicated HIP stream (replaces the
Replay thread of the reference, ape_x/replay.py:24-35, whose drain shares the GIL and a lock with the training loop and
which the learner has to ``_sleep`` for, ape_x/learner.py:47-49).

Pipeline of one experience block (plane-pool blocks: ~0.5 MB for 50 Ape-X transitions; 8 MB for 148 as full stacks):

  ring slot (shm) --thread: ONE memcpy--> pinned staging buffer --ingest stream: DMA--> replay ring in HBM
                                                                                        |
  learner stream:   ... graph k | RETIRE(slots) | graph k+1 | event | graph k+2 ... | COMMIT(slots, priorities) | ...

* The thread takes a slot zero-copy (``ShmExperience.acquire``), copies its used bytes into a pinned buffer and releases
  the slot at once, so producers never wait for the GPU. Other channels (in-process, Redis) hand decoded arrays, which
  are copied into the same pinned layout.
* RETIRE (learner stream, between two graph replays): the destination slots leave the sum-tree (priority 0, generation
  bumped), so no later draw returns them and stale priority write-backs are dropped. The Ape-X engine draws the batch
  of step k+1 inside step k, so a batch drawn BEFORE the retire may still read the old contents during graph k+1: the
  copy therefore waits for an event recorded after graph k+1 was launched.
* The DMA runs on the ingest stream, concurrently with the graphs that follow (it touches only retired slots).
* COMMIT (learner stream, once the copy's event has completed - polled, never waited for): priorities + generation +
  the live item count in ONE launch (no separate ``n_items.fill_``).

Everything that touches the learner stream happens in ``service()``, called by the learner loop between graph replays;
the thread never issues GPU work, so there is no cross-thread stream state.
"""
from __future__ import annotations

import ctypes
import threading
import time
from collections import deque
from typing import Deque, List, Optional

import numpy as np
import torch

from ..transport import codec


class _Staged:
    __slots__ = ("buf", "meta", "fields", "fmap", "blk", "K", "done", "retired_at", "need", "fast")

    def __init__(self, buf, meta, fields):
        self.buf, self.meta, self.fields = buf, meta, fields
        self.fmap = {name: (off, nb) for name, _, _, off, nb in fields}        # built on the ingest thread, not the learner's
        self.K = int(meta["K"])
        self.blk, self.done, self.retired_at, self.need, self.fast = None, None, -1, None, False


class AsyncIngest:
    def __init__(self, exp, replay, device, depth: int = 4, slot_bytes: Optional[int] = None, poll_s: float = 0.0005,
                 max_items: int = 4096):
        self.exp, self.replay = exp, replay
        self.device = torch.device(device)
        self.slot_bytes = int(slot_bytes or getattr(exp, "slot_bytes", 8 << 20))
        self.poll_s = poll_s
        self.stream = torch.cuda.Stream(self.device)
        # pinned staging buffers + one device priority buffer each (the commit kernel reads priorities from HBM)
        self._bufs = [torch.empty(self.slot_bytes, dtype=torch.uint8).pin_memory() for _ in range(depth)]
        self._prio = [torch.zeros(max_items, device=self.device) for _ in range(depth)]
        # device copy of a block's tables (everything behind its planes): the replay scatters them with one launch
        self._tail = [torch.zeros(1 << 16, dtype=torch.uint8, device=self.device) for _ in range(depth)]
        self.fast_path = hasattr(replay, "write_staged")
        self._free: Deque[int] = deque(range(depth))
        self._staged: Deque[_Staged] = deque()          # thread -> service: copied into pinned memory
        self._retired: Deque[_Staged] = deque()         # slots retired, waiting for one more step launch
        self._copying: Deque[_Staged] = deque()         # DMA enqueued on the ingest stream
        self._lock = threading.Lock()
        self._stop = threading.Event()
        self._tick = 0                                  # number of service() calls = learner steps launched
        self.items = 0                                  # committed
        self.blocks = 0
        self.bytes = 0
        self.stage_s = 0.0                              # thread time spent copying into pinned memory
        self.error: Optional[BaseException] = None
        self._zero_copy = hasattr(exp, "acquire")
        self._th = threading.Thread(target=self._run, name="drl-ingest", daemon=True)
        self._th.start()

    # ---- host thread -------------------------------------------------------------------------------------------
    def _take_free(self) -> Optional[int]:
        with self._lock:
            return self._free.popleft() if self._free else None

    def _run(self) -> None:
        try:
            while not self._stop.is_set():
                b = self._take_free()
                if b is None:
                    time.sleep(self.poll_s)
                    continue
                rec = self._stage_one(b)
                if rec is None:
                    with self._lock:
                        self._free.appendleft(b)
                    time.sleep(self.poll_s)
                    continue
                with self._lock:
                    self._staged.append(rec)
        except BaseException as e:  # noqa: BLE001 - surfaced by service()
            self.error = e

    def _stage_one(self, b: int) -> Optional[_Staged]:
        pin = self._bufs[b]
        if self._zero_copy:
            got = self.exp.acquire_raw()
            if got is None:
                return None
            ticket, ptr, nbytes = got
            t0 = time.perf_counter()
            try:
                if nbytes > self.slot_bytes:
                    raise ValueError("experience block of %d bytes exceeds the staging buffer" % nbytes)
                ctypes.memmove(pin.data_ptr(), ptr, nbytes)          # the ONE host copy: slot -> pinned
            finally:
                self.exp.release(ticket)                             # producers get the slot back before any GPU work
            self.stage_s += time.perf_counter() - t0
            meta, fields = codec.layout((ctypes.c_ubyte * codec.HEADER_BYTES).from_address(pin.data_ptr()))
        else:
            blocks = self.exp.drain(1)
            if not blocks:
                return None
            t0 = time.perf_counter()
            blk = blocks[0]
            view = pin.numpy()
            used = codec.encode_into(view, blk, int(blk["kind"]), int(blk.get("actor_id", 0)), int(blk.get("env_steps", 0)))
            meta, fields = codec.layout(view[:codec.HEADER_BYTES])
            assert used == meta["bytes"]
            self.stage_s += time.perf_counter() - t0
        rec = _Staged(b, meta, fields)
        # how much of the replay's rings the block takes - decided HERE, off the learner's thread: blocks of full stacks
        # are checked on their pixels for it (replay/device_seq.py stack_block_planes)
        rec.need = self.replay.block_need(int(meta["kind"]), rec.K, int(meta.get("M", 0)), self._views(rec))
        return rec

    # ---- learner thread ------------------------------------------------------------------------------------------
    def _views(self, rec: _Staged):
        pin = self._bufs[rec.buf]
        out = {}
        for name, dt, shape, off, nb in rec.fields:
            tdt = {np.uint8: torch.uint8, np.int32: torch.int32, np.float32: torch.float32}[dt]
            out[name] = pin[off:off + nb].view(tdt).view(*shape)
        return out

    def _tail_for(self, b: int, nbytes: int) -> torch.Tensor:
        if self._tail[b].numel() < nbytes:
            self._tail[b] = torch.zeros((nbytes + 4095) // 4096 * 4096, dtype=torch.uint8, device=self.device)
        return self._tail[b]

    def service(self) -> int:
        """Advance every block by at most one stage; returns the number of items committed by this call. Call once per
        learner step, BEFORE launching the step."""
        if self.error is not None:
            raise RuntimeError("ingest thread died") from self.error
        self._tick += 1
        r = self.replay
        n_commit = 0
        # 3. commit the blocks whose copy has finished (polled: the learner stream never waits for a copy); blocks written
        #    by the same service call share their event, neighbours in the slot ring share the commit launch
        fast, ev_ok = [], None
        while self._copying and (self._copying[0].done is ev_ok or self._copying[0].done.query()):
            rec = self._copying.popleft()
            ev_ok = rec.done
            if rec.fast:
                fast.append(rec.blk)
            else:
                r.commit_block(rec.blk, self._prio[rec.buf])
            with self._lock:
                self._free.append(rec.buf)
            n_commit += rec.K
            self.items += rec.K
            self.blocks += 1
            self.bytes += int(rec.meta["bytes"])
        if fast:
            r.commit_blocks(fast)
        # 2. blocks retired before the previous step launch: that step has been launched, start their copy behind it
        if self._retired and self._retired[0].retired_at < self._tick - 1:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.device))
            self.stream.wait_event(ev)
            done, raw = torch.cuda.Event(), self.stream.cuda_stream
            while self._retired and self._retired[0].retired_at < self._tick - 1:
                rec = self._retired.popleft()
                rec.fast = self.fast_path and r.write_staged(rec.blk, int(rec.meta["kind"]), self._bufs[rec.buf], rec.fmap,
                                                             lambda nb, b=rec.buf: self._tail_for(b, nb), raw)
                if not rec.fast:                                      # full-stack blocks: field by field (torch ops)
                    with torch.cuda.stream(self.stream):
                        v = self._views(rec)
                        r.write_views(rec.blk, int(rec.meta["kind"]), v)     # plane pool / stacks / sequences: the replay knows
                        self._prio[rec.buf][:rec.K].copy_(v["prio"], non_blocking=True)
                rec.done = done
                self._copying.append(rec)
            done.record(self.stream)
        # 1. newly staged blocks: reserve their destination ranges; what owned them leaves the tree on the learner stream
        #    (one retire launch per run of neighbouring evicted blocks)
        evicted = [] if hasattr(r, "retire_ranges") else None
        while True:
            with self._lock:
                rec = self._staged.popleft() if self._staged else None
            if rec is None:
                break
            if rec.K > self._prio[rec.buf].numel():
                self._prio[rec.buf] = torch.zeros(rec.K, device=self.device)
            K, M = rec.need
            rec.blk = r.reserve_block(K, M, evicted) if evicted is not None else r.reserve_block(K, M)
            rec.retired_at = self._tick
            self._retired.append(rec)
        if evicted:
            r.retire_ranges(evicted)
        return n_commit

    def pending(self) -> int:
        with self._lock:
            return len(self._staged) + len(self._retired) + len(self._copying)

    def flush(self, timeout: float = 10.0) -> int:
        """Drive everything that is already staged to the committed state (start-up fill, tests, shutdown)."""
        n, t0 = 0, time.time()
        while self.pending() and time.time() - t0 < timeout:
            k = self.service()
            n += k
            if not k:
                time.sleep(0.0002)
        return n

    def stats(self) -> dict:
        return dict(items=self.items, blocks=self.blocks, bytes=self.bytes, stage_s=self.stage_s, pending=self.pending())

    def close(self) -> None:
        self._stop.set()
        self._th.join(2.0)
        torch.cuda.synchronize(self.device)

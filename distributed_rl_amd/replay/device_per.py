"""HBM-resident prioritized replay for transitions (replaces PrioritizedMemory + Replay thread,
libs/replay_memory.py:51-81, ape_x/replay.py).

Everything lives in device memory, sized for the 288 GB of an MI355X:

* ``frames``  uint8 ``[2·cap, 84·84·4]`` — states in the first half, next-states in the second (2 M transitions =
  113 GB; the reference's float32 pickles would need 452 GB). HWC layout, so conv1's implicit-GEMM
  reads 32 contiguous bytes per kernel row straight out of this store through the sampled byte offsets — a
  sampled batch is never gathered or copied.
* ``act / rew / done`` per slot, ``gen`` per-slot generation, and the 64-ary sum-tree.

It is a ring: ingest overwrites the oldest slots (FIFO, like ``remove_to_fit`` — replay_memory.py:70-75 — but
immediate instead of every 100 steps: documented deviation). ``sample`` and ``update`` are pure
kernel launches on the current stream with preallocated outputs, so they can sit inside a captured hipGraph;
the number of live items is a device scalar for the same reason.
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np
import torch

from ..ops import hip as H

FRAME_BYTES = 84 * 84 * 4


class DevicePER:
    def __init__(self, capacity: int, device="cuda", batch_size: int = 512, seed: int = 0,
                 frame_bytes: int = FRAME_BYTES, max_ingest: int = 4096):
        self.capacity = int(capacity)
        self.device = torch.device(device)
        self.frame_bytes = frame_bytes
        self.batch_size = batch_size
        self.seed = seed
        self.lay = H.TreeLayout(self.capacity)
        dev = self.device
        self.tree = torch.zeros(self.lay.total, device=dev)
        self.gen = torch.zeros(self.capacity, dtype=torch.int32, device=dev)
        self.frames = torch.zeros(2 * self.capacity, frame_bytes, dtype=torch.uint8, device=dev)
        self.act = torch.zeros(self.capacity, dtype=torch.int32, device=dev)
        self.rew = torch.zeros(self.capacity, device=dev)
        self.done = torch.zeros(self.capacity, device=dev)
        self.n_items = torch.zeros(1, dtype=torch.int64, device=dev)
        self._scratch = torch.zeros(max(max_ingest, 64), dtype=torch.int32, device=dev)
        self.cursor = 0
        self.count = 0
        # sample outputs (fixed addresses: graph-capturable)
        B = batch_size
        self.idx = torch.zeros(B, dtype=torch.int32, device=dev)
        self.prio = torch.zeros(B, device=dev)
        self.gen_s = torch.zeros(B, dtype=torch.int32, device=dev)
        self.frame_off = torch.zeros(2 * B, dtype=torch.int64, device=dev)

    def __len__(self) -> int:
        return self.count

    def with_batch(self, batch_size: int) -> "DevicePER":
        """A second sampler over the SAME store (frames, tree, generations are shared, the sample buffers are its own):
        e.g. the strong-scaling measurement of bench.py draws 512 / world per rank from the shard the weak-scaling run
        filled. Ingest through either object, not both."""
        import copy
        o = copy.copy(self)
        B, dev = int(batch_size), self.device
        o.batch_size = B
        o.idx = torch.zeros(B, dtype=torch.int32, device=dev)
        o.prio = torch.zeros(B, device=dev)
        o.gen_s = torch.zeros(B, dtype=torch.int32, device=dev)
        o.frame_off = torch.zeros(2 * B, dtype=torch.int64, device=dev)
        o._nxt = None
        return o

    def total_prios(self) -> float:
        top = self.lay.offsets[-1]
        return float(self.tree[top:top + 64].sum())

    # ---- ingest -----------------------------------------------------------------------------------
    # Three stages, so that the bulk copy can run on its own stream while the learner keeps stepping (engine/ingest.py):
    # reserve + retire (the slots leave the tree: nothing samples them any more), write_slots (the data, any stream),
    # commit (priorities + generation + item count, one launch). ``ingest`` runs the three back to back on one stream.
    def _ring_write(self, dst: torch.Tensor, src: torch.Tensor, start: int) -> None:
        n = src.shape[0]
        first = min(n, self.capacity - start)
        dst[start:start + first].copy_(src[:first], non_blocking=True)       # pinned source: one DMA straight into the ring
        if first < n:
            dst[:n - first].copy_(src[first:], non_blocking=True)

    def reserve(self, K: int) -> int:
        if K > self.capacity:
            raise ValueError("block larger than the replay")
        if K > self._scratch.numel():
            self._scratch = torch.zeros(K, dtype=torch.int32, device=self.device)
        s = self.cursor
        self.cursor = (s + K) % self.capacity
        return s

    def retire(self, start: int, K: int) -> None:
        H.sumtree_ingest(self.tree, self.lay, start, None, self.gen, self._scratch, n=K)

    def write_slots(self, start: int, state: torch.Tensor, next_state: torch.Tensor, action: torch.Tensor,
                    reward: torch.Tensor, done: torch.Tensor) -> None:
        K = int(action.shape[0])
        self._ring_write(self.frames[:self.capacity], state.reshape(K, -1), start)
        self._ring_write(self.frames[self.capacity:], next_state.reshape(K, -1), start)
        self._ring_write(self.act, action, start)
        self._ring_write(self.rew, reward, start)
        self._ring_write(self.done, done, start)

    def commit(self, start: int, prio: torch.Tensor, K: int) -> None:
        """``prio``: float32 on the device, at least K elements. The item count rides along with the tree launch."""
        self.count = min(self.capacity, self.count + K)
        H.sumtree_ingest(self.tree, self.lay, start, prio, self.gen, self._scratch, n=K, n_items=self.n_items,
                         n_items_val=self.count)

    def ingest(self, state: torch.Tensor, next_state: torch.Tensor, action: torch.Tensor, reward: torch.Tensor,
               done: torch.Tensor, prio: torch.Tensor) -> int:
        """Append K transitions (tensors on any device; frames uint8 HWC, flattened or not)."""
        K = int(action.shape[0])
        if K == 0:
            return 0
        s = self.reserve(K)
        self.write_slots(s, state, next_state, action, reward, done)
        self.commit(s, prio.to(self.device, torch.float32).contiguous(), K)
        return K

    def ingest_block(self, block: Dict[str, np.ndarray]) -> int:
        """Consume a decoded experience block (transport/codec.py KIND_APEX)."""
        t = torch.from_numpy
        return self.ingest(t(np.ascontiguousarray(block["state"])), t(np.ascontiguousarray(block["next_state"])),
                           t(np.ascontiguousarray(block["action"])), t(np.ascontiguousarray(block["reward"])),
                           t(np.ascontiguousarray(block["done"])), t(np.ascontiguousarray(block["prio"])))

    def fill_synthetic(self, n: Optional[int] = None, n_action: int = 4, chunk: int = 65536, seed: int = 0) -> None:
        """Fill with random Atari-shaped transitions generated ON DEVICE (benchmarks: no dataset, no H2D)."""
        n = self.capacity if n is None else n
        g = torch.Generator(device=self.device).manual_seed(seed)
        done = 0
        while done < n:
            k = min(chunk, n - done)
            fr = torch.randint(0, 256, (2, k, self.frame_bytes), dtype=torch.uint8, device=self.device, generator=g)
            self.ingest(fr[0], fr[1], torch.randint(0, n_action, (k,), device=self.device, generator=g),
                        torch.randn(k, device=self.device, generator=g).clamp_(-1, 1),
                        (torch.rand(k, device=self.device, generator=g) < 0.01).float(),
                        torch.rand(k, device=self.device, generator=g) + 0.1)
            done += k

    # ---- sample / update (graph-capturable) -----------------------------------------------------
    def sample(self, st: torch.Tensor, begin: bool = False, is_n_scale: float = 0.0,
               n_is: Optional[torch.Tensor] = None) -> None:
        """Draw ``batch_size`` slots ∝ priority into ``self.idx / prio / gen_s / frame_off``. ``begin`` fuses
        step_begin (beta / n_items refresh, fresh Philox counter) into the sampling kernel; ``is_n_scale`` > 0 also
        accumulates the batch's max IS weight there (see ops.hip.sumtree_sample). ``n_is`` (int64 device scalar)
        replaces the live item count in the IS weights: sharded PER passes N_global / world (engine.shard_n_is)."""
        n = (n_is if n_is is not None else self.n_items) if begin else None
        H.sumtree_sample(self.tree, self.lay, st, self.seed, self.idx, self.prio, self.gen, self.gen_s, self.frame_off,
                         self.frame_bytes, begin_n_items=n, is_n_scale=is_n_scale)

    def _next_buffers(self):
        if getattr(self, "_nxt", None) is None:
            self._nxt = (torch.zeros_like(self.idx), torch.zeros_like(self.prio), torch.zeros_like(self.gen_s),
                         torch.zeros_like(self.frame_off))
        return self._nxt

    def sample_extra(self, st: torch.Tensor, is_n_scale: float, n_is: Optional[torch.Tensor] = None):
        """``sample(st, begin=True, is_n_scale=...)`` as a record that draws into the NEXT-batch buffers: the learner engine
        hands it to the conv1 backward launch (ops.hip.conv_bwd(1, ..., sample=...)), which still reads the current
        frame offsets, and ``handover_extra`` to the optimizer launch."""
        idx, prio, gen_s, foff = self._next_buffers()
        return H.sample_extra(self.tree, self.lay, st, self.seed, idx, prio, self.gen, gen_s, foff, self.frame_bytes,
                              n_is if n_is is not None else self.n_items, is_n_scale)

    def handover_extra(self):
        """Copies next-batch buffers -> live sample buffers (ops.hip.rmsprop_step(copies=...))."""
        idx, prio, gen_s, foff = self._next_buffers()
        return H.copy_extra([(self.idx, idx), (self.prio, prio), (self.gen_s, gen_s), (self.frame_off, foff)])

    def update(self, new_prio: torch.Tensor, end_state: Optional[torch.Tensor] = None) -> None:
        """Write back priorities of the last sample; slots overwritten since are skipped (generation check).
        ``end_state`` fuses step_end into the tree rebuild launch."""
        H.sumtree_update(self.tree, self.lay, self.idx, new_prio, self.gen, self.gen_s, end_state=end_state)

    def update_stage(self, kind: int, new_prio: torch.Tensor, end_state: Optional[torch.Tensor] = None):
        """The same write-back as ``update`` split into its two stages (kind 1 = leaves + level 1, kind 2 = upper levels +
        optional step_end) as ``TreeExtra`` records: the learner engine hands them to two consecutive fused backward
        launches (ops.hip.conv_bwd) instead of launching them at the tail of the step."""
        return H.tree_extra(kind, self.tree, self.lay, self.idx, new_prio, self.gen, self.gen_s, end_state=end_state)

    # host-API parity with PrioritizedMemory (tests / tools; not on the hot path)
    def update_priorities(self, indices, priorities) -> None:
        idx = torch.as_tensor(indices, dtype=torch.int32, device=self.device)
        val = torch.as_tensor(priorities, dtype=torch.float32, device=self.device)
        H.sumtree_update(self.tree, self.lay, idx.contiguous(), val.contiguous())

    def remove_to_fit(self) -> int:
        return 0     # the ring evicts on ingest

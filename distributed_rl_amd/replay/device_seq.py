"""HBM-resident prioritized *sequence* replay for R2D2 (reference: r2d2/actor.py:56-63 stores
``Sequence(transitions, recurrent_state)`` pickles in the same PrioritizedMemory as Ape-X).

Per slot: ``T = burn_in + seq_len`` uint8 frames (zero-padded past ``length``), per-step action / n-step reward /
done, the true length, and the stored recurrent state ``(h0, c0)`` that preceded frame 0 (r2d2/actor.py:39,57).
Sampling yields sequence indices plus the *time-major* byte offsets of all ``T·B`` frames, so the time-batched
torso reads its 7 680 frames straight from this store (no gather, no H2D). Capacity is whatever fits the GPU's
288 GB (``sequences_that_fit``): 3.4 MB per sequence at full frame stacks.
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np
import torch

from ..config import R2D2Config
from ..ops import hip as H

FRAME_BYTES = 84 * 84 * 4


def sequences_that_fit(free_bytes: int, cfg: R2D2Config, frac: float = 0.6) -> int:
    per = cfg.total_len * (FRAME_BYTES + 12) + 2 * cfg.hidden * 4 + 64
    return int(free_bytes * frac // per)


class DeviceSequenceReplay:
    def __init__(self, capacity: int, device, cfg: R2D2Config, seed: int = 0, frame_bytes: int = FRAME_BYTES):
        self.capacity, self.cfg = int(capacity), cfg
        self.device = dev = torch.device(device)
        self.T, self.Hd, self.B = cfg.total_len, cfg.hidden, cfg.batch_size
        self.frame_bytes, self.seed = frame_bytes, seed
        S, T = self.capacity, self.T
        self.lay = H.TreeLayout(S)
        self.tree = torch.zeros(self.lay.total, device=dev)
        self.gen = torch.zeros(S, dtype=torch.int32, device=dev)
        self.frames = torch.zeros(S * T, frame_bytes, dtype=torch.uint8, device=dev)
        self.act = torch.zeros(S, T, dtype=torch.int32, device=dev)
        self.rew = torch.zeros(S, T, device=dev)
        self.done = torch.zeros(S, T, device=dev)
        self.length = torch.zeros(S, dtype=torch.int32, device=dev)
        self.h0 = torch.zeros(S, self.Hd, device=dev)
        self.c0 = torch.zeros(S, self.Hd, device=dev)
        self.n_items = torch.zeros(1, dtype=torch.int64, device=dev)
        self._scratch = torch.zeros(max(1024, 64), dtype=torch.int32, device=dev)
        self.cursor = 0
        self.count = 0
        B = self.B
        self.idx = torch.zeros(B, dtype=torch.int32, device=dev)
        self.prio = torch.zeros(B, device=dev)
        self.gen_s = torch.zeros(B, dtype=torch.int32, device=dev)
        self.frame_off = torch.zeros(T * B, dtype=torch.int64, device=dev)

    def __len__(self) -> int:
        return self.count

    def total_prios(self) -> float:
        top = self.lay.offsets[-1]
        return float(self.tree[top:top + 64].sum())

    # ---- ingest: reserve + retire / write_slots / commit (see replay/device_per.py and engine/ingest.py) ------------
    def _ring_write(self, dst, src, start, per: int = 1) -> None:
        n = src.shape[0] // per
        first = min(n, self.capacity - start)
        dst[start * per:(start + first) * per].copy_(src[:first * per], non_blocking=True)
        if first < n:
            dst[:(n - first) * per].copy_(src[first * per:], non_blocking=True)

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

    def write_slots(self, start: int, frames, action, reward, done, length, h0, c0) -> None:
        K = int(length.shape[0])
        self._ring_write(self.frames, frames.reshape(K * self.T, -1), start, per=self.T)
        self._ring_write(self.act, action, start)
        self._ring_write(self.rew, reward, start)
        self._ring_write(self.done, done, start)
        self._ring_write(self.length, length, start)
        self._ring_write(self.h0, h0, start)
        self._ring_write(self.c0, c0, start)

    def commit(self, start: int, prio: torch.Tensor, K: int) -> None:
        self.count = min(self.capacity, self.count + K)
        H.sumtree_ingest(self.tree, self.lay, start, prio, self.gen, self._scratch, n=K, n_items=self.n_items,
                         n_items_val=self.count)

    def ingest(self, frames, action, reward, done, length, h0, c0, prio) -> int:
        """K sequences: frames (K,T,…) uint8, action/reward/done (K,T), length (K,), h0/c0 (K,H), prio (K,)."""
        K = int(length.shape[0])
        if K == 0:
            return 0
        s = self.reserve(K)
        self.write_slots(s, frames, action, reward, done, length, h0, c0)
        self.commit(s, prio.to(self.device, torch.float32).contiguous(), K)
        return K

    def ingest_block(self, block: Dict[str, np.ndarray]) -> int:
        t = lambda k: torch.from_numpy(np.ascontiguousarray(block[k]))  # noqa: E731
        return self.ingest(t("frames"), t("action"), t("reward"), t("done"), t("length"), t("h0"), t("c0"), t("prio"))

    def fill_synthetic(self, n: Optional[int] = None, n_action: int = 4, chunk: int = 512, seed: int = 0) -> None:
        n = self.capacity if n is None else n
        dev, T = self.device, self.T
        g = torch.Generator(device=dev).manual_seed(seed)
        done_n = 0
        while done_n < n:
            k = min(chunk, n - done_n)
            self.ingest(torch.randint(0, 256, (k, T, self.frame_bytes), dtype=torch.uint8, device=dev, generator=g),
                        torch.randint(0, n_action, (k, T), device=dev, generator=g),
                        torch.randn(k, T, device=dev, generator=g).clamp_(-1, 1),
                        torch.zeros(k, T, device=dev), torch.full((k,), T, dtype=torch.int32, device=dev),
                        torch.randn(k, self.Hd, device=dev, generator=g) * 0.1, torch.randn(k, self.Hd, device=dev, generator=g) * 0.1,
                        torch.rand(k, device=dev, generator=g) + 0.1)
            done_n += k

    # ---- graph-capturable ------------------------------------------------------------------------------
    def sample(self, st: torch.Tensor, begin: bool = False, n_is: Optional[torch.Tensor] = None) -> None:
        """``n_is``: item count used by the IS weights instead of this shard's own (sharded PER: N_global / world)."""
        H.sumtree_sample(self.tree, self.lay, st, self.seed, self.idx, self.prio, self.gen, self.gen_s,
                         begin_n_items=(n_is if n_is is not None else self.n_items) if begin else None)
        H.seq_offsets(self.idx, self.T, self.frame_bytes, self.frame_off)

    def update(self, new_prio: torch.Tensor, end_state=None, fault=None) -> None:
        H.sumtree_update(self.tree, self.lay, self.idx, new_prio, self.gen, self.gen_s, end_state=end_state, fault=fault)

    def remove_to_fit(self) -> int:
        return 0

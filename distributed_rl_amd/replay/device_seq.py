"""HBM-resident prioritized *sequence* replay for R2D2 (reference: r2d2/actor.py:56-63 stores
``Sequence(transitions, recurrent_state)`` pickles in the same PrioritizedMemory as Ape-X).

Every 84x84 frame is stored ONCE, across sequences too. The store is a ring of single uint8 PLANES; a
sequence is a row of ``T = burn_in + seq_len`` window starts into it (``win[slot, t]`` = byte offset of the first of the 4
consecutive planes that make up observation t) plus per-step action / n-step reward / done, its true length and the
stored recurrent state ``(h0, c0)`` that preceded observation 0 (r2d2/actor.py:39,57). Consecutive sequences of one
environment overlap by ``burn_in + overlap`` = 80 of 120 steps (r2d2/actor.py:56-60): an actor publishes the sequences of
ONE environment together, with the planes they span once (transport/codec.py kind 4), so a sequence costs ~44 new planes
instead of its own 123 (and instead of the 480 of full 4-stacks): 0.31 MB, 100 K paper-shape sequences in ~32 GB, ~280 K
in the 87 GB they took when every sequence kept its own planes. The windows are PER STEP, so nothing is assumed about how
an environment's stack moves: one that does not shift by exactly one frame per step is stored correctly, just not
compressed (the pixel check is the actor's, algo/vec_assembly.PlaneLog).

Two rings advance together in ingestion order (replay/rings.py): sequence slots (tables + the 64-ary sum-tree) and planes;
making room retires the OLDEST blocks whole - FIFO, like the reference's ``remove_to_fit`` (replay_memory.py:70-75).
Sampling yields sequence indices plus the *time-major* byte offsets of the plane windows of all ``T·B`` observations
(one gather from ``win``), so the time-batched torso reads its 7 680 frames straight from this store - no copy, no H2D.
Blocks of kind 2 (full HWC stacks per step: external producers, older tools) are accepted too and split on the device.
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np
import torch

from ..config import R2D2Config
from ..ops import hip as H
from .rings import Block as _Block, RingPair

PLANE_BYTES = 84 * 84
STACK = 4
FRAME_BYTES = PLANE_BYTES * STACK


def _stride(cfg: R2D2Config) -> int:
    """Frames between the starts of two consecutive sequences of an env (r2d2/actor.py:56-60)."""
    return cfg.seq_len - cfg.overlap if 0 <= cfg.overlap < cfg.seq_len else max(1, cfg.seq_len // 2)


def planes_per_sequence(cfg: R2D2Config, run: int = 20) -> float:
    """Planes per sequence when an env publishes ``run`` consecutive sequences together (the actors' ``actor_batch``): they
    span stride * (run - 1) + T + 3 planes - 44 each at the paper's shape; episode boundaries cost a little more."""
    stride = _stride(cfg)
    return (stride * (run - 1) + cfg.total_len + STACK - 1) / run


def sequences_that_fit(free_bytes: int, cfg: R2D2Config, frac: float = 0.6, planes_per_item: Optional[float] = None) -> int:
    ppi = planes_per_item if planes_per_item is not None else planes_per_sequence(cfg) * 1.08
    per = ppi * PLANE_BYTES + cfg.total_len * 20 + 2 * cfg.hidden * 4 + 64
    return int(free_bytes * frac // per)


class DeviceSequenceReplay(RingPair):
    def __init__(self, capacity: int, device, cfg: R2D2Config, seed: int = 0, planes_per_item: Optional[float] = None,
                 plane_capacity: Optional[int] = None):
        self.capacity, self.cfg = int(capacity), cfg
        self.device = dev = torch.device(device)
        self.T, self.Hd, self.B = cfg.total_len, cfg.hidden, cfg.batch_size
        self.seed = seed
        S, T = self.capacity, self.T
        self.P = T + STACK - 1                              # planes of ONE sequence on its own (no sharing)
        ppi = planes_per_item if planes_per_item is not None else planes_per_sequence(cfg) * 1.08
        # room for at least a few unshared blocks, so that small replays (tests) work with any producer
        self.plane_capacity = int(plane_capacity if plane_capacity is not None else max(S * ppi, min(S, 64) * self.P) + self.P)
        self.lay = H.TreeLayout(S)
        self.tree = torch.zeros(self.lay.total, device=dev)
        self.gen = torch.zeros(S, dtype=torch.int32, device=dev)
        self.frames = torch.zeros(self.plane_capacity, PLANE_BYTES, dtype=torch.uint8, device=dev)     # the plane ring
        self.win = torch.zeros(S, T, dtype=torch.int64, device=dev)                                     # byte offsets
        self.act = torch.zeros(S, T, dtype=torch.int32, device=dev)
        self.rew = torch.zeros(S, T, device=dev)
        self.done = torch.zeros(S, T, device=dev)
        self.length = torch.zeros(S, dtype=torch.int32, device=dev)
        self.h0 = torch.zeros(S, self.Hd, device=dev)
        self.c0 = torch.zeros(S, self.Hd, device=dev)
        self.n_items = torch.zeros(1, dtype=torch.int64, device=dev)
        self.prio_pending = torch.zeros(S, device=dev)                  # priorities of written, not yet committed blocks
        self._scratch = torch.zeros(1024, dtype=torch.int32, device=dev)
        self._init_rings()
        B = self.B
        self.idx = torch.zeros(B, dtype=torch.int32, device=dev)
        self.prio = torch.zeros(B, device=dev)
        self.gen_s = torch.zeros(B, dtype=torch.int32, device=dev)
        self.frame_off = torch.zeros(T * B, dtype=torch.int64, device=dev)
        self.unshifted_blocks = 0                            # kind-2 blocks whose stacks did not shift by one frame per step

    def __len__(self) -> int:
        return self.count

    def total_prios(self) -> float:
        top = self.lay.offsets[-1]
        return float(self.tree[top:top + 64].sum())

    def bytes_per_item(self) -> float:
        """HBM bytes per sequence of capacity (planes + tables)."""
        return (self.frames.numel() + self.capacity * (self.T * 20 + 8 * self.Hd + 16)) / self.capacity

    def planes_in_use(self) -> int:
        return sum(b.M for b in self._blocks if b.state == 1)

    # ---- ingest: reserve (+ retire what is in the way) / write / commit (engine/ingest.py drives the three stages) ------
    def _need_scratch(self, n: int) -> None:
        if n > self._scratch.numel():
            self._scratch = torch.zeros(n, dtype=torch.int32, device=self.device)

    def _retire(self, slot: int, K: int) -> None:
        self._need_scratch(K)
        H.sumtree_ingest(self.tree, self.lay, slot, None, self.gen, self._scratch, n=K, n_items=self.n_items,
                         n_items_val=self.count)

    def _commit_range(self, slot: int, n: int) -> None:
        self._need_scratch(n)
        H.sumtree_ingest(self.tree, self.lay, slot, self.prio_pending[slot:slot + n], self.gen, self._scratch, n=n,
                         n_items=self.n_items, n_items_val=self.count)

    def reserve_block(self, K: int, M: int, deferred=None) -> _Block:
        self._need_scratch(K)
        return RingPair.reserve_block(self, K, M, deferred)

    def write_block(self, blk: _Block, planes: torch.Tensor, win: torch.Tensor, action, reward, done, length, h0, c0) -> None:
        """planes u8 [M, 84*84] (or [M,84,84]); win int32 [K, T]: index, relative to the block's planes, of the first of
        the 4 consecutive planes of every observation (padding steps: any valid index). Tensors on any device."""
        s, K, p, M, dev = blk.slot, blk.K, blk.plane, blk.M, self.device
        self.frames[p:p + M].copy_(planes.reshape(M, PLANE_BYTES), non_blocking=True)
        w = win.to(dev, non_blocking=True).to(torch.int64)
        self.win[s:s + K] = (w + p) * PLANE_BYTES
        self._write_tables(s, K, action, reward, done, length, h0, c0)

    def _write_tables(self, s, K, action, reward, done, length, h0, c0) -> None:
        for dst, src in ((self.act, action), (self.rew, reward), (self.done, done), (self.length, length), (self.h0, h0),
                         (self.c0, c0)):
            dst[s:s + K].copy_(src, non_blocking=True)

    def write_block_stacks(self, blk: _Block, frames: torch.Tensor, action, reward, done, length, h0, c0) -> None:
        """A kind-2 block: full HWC stacks [K, T, 84, 84, 4]. ``blk.M`` says which form was reserved (``stack_block_planes``):
        K * (T + 3) = every stack is the one before it moved on by a frame (checked by the caller): the three older frames of
        the first stack + the newest frame of every stack; K * 4T = no such relation: all four planes of every stack."""
        K, T, p, dev = blk.K, self.T, blk.plane, self.device
        fr = frames.reshape(K, T, 84, 84, STACK).to(dev, non_blocking=True)
        t = torch.arange(T, device=dev, dtype=torch.int64)
        if blk.M == K * self.P:
            pl = torch.empty(K, self.P, 84, 84, dtype=torch.uint8, device=dev)
            pl[:, :STACK - 1] = fr[:, 0, :, :, :STACK - 1].permute(0, 3, 1, 2)
            pl[:, STACK - 1:] = fr[:, :, :, :, STACK - 1]
            per, step = self.P, 1
        else:
            assert blk.M == K * STACK * T
            pl = fr.permute(0, 1, 4, 2, 3)                                               # [K, T, 4, 84, 84]
            per, step = STACK * T, STACK
        self.frames[p:p + blk.M].view(K, per, PLANE_BYTES).copy_(pl.reshape(K, per, PLANE_BYTES))
        base = p + torch.arange(K, device=dev, dtype=torch.int64)[:, None] * per
        self.win[blk.slot:blk.slot + K] = (base + t[None, :] * step) * PLANE_BYTES
        self._write_tables(blk.slot, K, action, reward, done, length, h0, c0)

    def stack_block_planes(self, frames, length) -> int:
        """Planes a kind-2 block needs: T + 3 per sequence when, inside every sequence's true length, each stack is its
        predecessor shifted by one frame (frame-stacking envs: wrapped_env.py:37-42) - verified on the pixels, on the host
        copy the ingest thread holds anyway - else 4T (a sequence from any other env used to be stored wrongly)."""
        fr = frames.numpy() if isinstance(frames, torch.Tensor) else np.asarray(frames)
        ln = length.numpy() if isinstance(length, torch.Tensor) else np.asarray(length)
        K, T = int(ln.shape[0]), self.T
        fr = fr.reshape(K, T, 84 * 84, STACK)
        for k in range(K):
            L = int(ln[k])
            if L > 1 and not np.array_equal(fr[k, 1:L, :, :STACK - 1], fr[k, :L - 1, :, 1:]):
                self.unshifted_blocks += 1
                return K * STACK * T
        return K * self.P

    def commit_block(self, blk: _Block, prio: torch.Tensor) -> None:
        if blk.state == 2:                  # its ranges were handed to a younger block while this one was in flight
            return
        blk.state = 1
        self.count += blk.K
        H.sumtree_ingest(self.tree, self.lay, blk.slot, prio, self.gen, self._scratch, n=blk.K, n_items=self.n_items,
                         n_items_val=self.count)

    # block-kind dispatch used by engine/ingest.py (views of one staged block)
    def block_need(self, kind: int, K: int, M: int = 0, views=None):
        from ..transport import codec
        if kind == codec.KIND_R2D2_PLANES:
            return K, M
        if views is not None:
            return K, self.stack_block_planes(views["frames"], views["length"])
        return K, K * STACK * self.T

    def write_staged(self, blk: _Block, kind: int, pin: torch.Tensor, fields, tail_for,
                     stream: Optional[int] = None) -> bool:
        """As DevicePER.write_staged, for sequence blocks with a plane pool (codec kind 4): planes by one DMA, the tables by
        one DMA + one scatter launch, issued by one native call; priorities go to ``prio_pending``. False for other kinds."""
        from ..transport import codec
        if kind != codec.KIND_R2D2_PLANES:
            return False
        f = fields if isinstance(fields, dict) else {name: (off, nb) for name, _, _, off, nb in fields}
        s, K, p, M, T = blk.slot, blk.K, blk.plane, blk.M, self.T
        p_off, p_nb = f["planes"]
        t0, t1 = f["win"][0], f["prio"][0] + f["prio"][1]
        Hd = self.h0.shape[1]
        assert p_nb == M * PLANE_BYTES and f["win"][1] == 4 * K * T and f["h0"][1] == 4 * K * Hd and f["prio"][1] == 4 * K
        assert 0 <= p and p + M <= self.plane_capacity and 0 <= s and s + K <= self.capacity
        assert t1 <= pin.numel() and p_off + p_nb <= pin.numel()
        o = lambda name: f[name][0] - t0                                            # noqa: E731
        segs = [(o("action"), self.act.data_ptr() + 4 * s * T, K * T), (o("reward"), self.rew.data_ptr() + 4 * s * T, K * T),
                (o("done"), self.done.data_ptr() + 4 * s * T, K * T), (o("length"), self.length.data_ptr() + 4 * s, K),
                (o("h0"), self.h0.data_ptr() + 4 * s * Hd, K * Hd), (o("c0"), self.c0.data_ptr() + 4 * s * Hd, K * Hd),
                (o("prio"), self.prio_pending.data_ptr() + 4 * s, K)]
        wins = [(o("win"), self.win.data_ptr() + 8 * s * T, K * T)]
        dst = self.frames.data_ptr() + p * PLANE_BYTES
        src = pin.data_ptr()
        H.ingest_block(dst, src + p_off, p_nb, tail_for(t1 - t0), src + t0, t1 - t0, segs, wins, p, stream=stream)
        return True

    def write_views(self, blk: _Block, kind: int, v) -> None:
        from ..transport import codec
        if kind == codec.KIND_R2D2_PLANES:
            self.write_block(blk, v["planes"], v["win"], v["action"], v["reward"], v["done"], v["length"], v["h0"], v["c0"])
        else:
            self.write_block_stacks(blk, v["frames"], v["action"], v["reward"], v["done"], v["length"], v["h0"], v["c0"])

    def gather_stacks(self, idx: torch.Tensor) -> torch.Tensor:
        """The observations of the slots ``idx`` as HWC stacks [n, T, 84, 84, 4] (tests / tools)."""
        off = self.win[idx.long()]                                                        # [n, T] byte offsets
        ar = torch.arange(FRAME_BYTES, device=self.device)
        st = self.frames.view(-1)[off[:, :, None] + ar[None, None, :]].view(off.shape[0], self.T, STACK, 84, 84)
        return st.permute(0, 1, 3, 4, 2).contiguous()

    def ingest_planes(self, planes, win, action, reward, done, length, h0, c0, prio) -> int:
        K, M = int(length.shape[0]), int(planes.shape[0])
        if K == 0:
            return 0
        blk = self.reserve_block(K, M)
        self.write_block(blk, planes, win, action, reward, done, length, h0, c0)
        self.commit_block(blk, prio.to(self.device, torch.float32).contiguous())
        return K

    def ingest(self, frames, action, reward, done, length, h0, c0, prio) -> int:
        """K sequences as full stacks: frames (K,T,…) uint8, action/reward/done (K,T), length (K,), h0/c0 (K,H), prio (K,)."""
        K = int(length.shape[0])
        if K == 0:
            return 0
        host = frames.cpu() if isinstance(frames, torch.Tensor) else frames
        blk = self.reserve_block(K, self.stack_block_planes(host, length.cpu() if isinstance(length, torch.Tensor) else length))
        self.write_block_stacks(blk, torch.as_tensor(frames), action, reward, done, length, h0, c0)
        self.commit_block(blk, prio.to(self.device, torch.float32).contiguous())
        return K

    def ingest_block(self, block: Dict[str, np.ndarray]) -> int:
        t = lambda k: torch.from_numpy(np.ascontiguousarray(block[k]))  # noqa: E731
        if "planes" in block:
            return self.ingest_planes(t("planes"), t("win"), t("action"), t("reward"), t("done"), t("length"), t("h0"), t("c0"),
                                      t("prio"))
        return self.ingest(t("frames"), t("action"), t("reward"), t("done"), t("length"), t("h0"), t("c0"), t("prio"))

    # ---- snapshot / restore (checkpoint sidecar) ---------------------------------------------------------
    def snapshot(self) -> Dict[str, torch.Tensor]:
        """The live blocks in ingestion order as plain CPU tensors (``weights_only=True`` loadable): per block its planes and
        its windows RELATIVE to them, the per-step tables, stored state and the CURRENT priorities (tree leaves). 100 K
        sequences are ~32 GB: opt-in (``learner_node.py --save-replay``)."""
        torch.cuda.synchronize(self.device) if self.device.type == "cuda" else None
        live = [b for b in self._blocks if b.state == 1]
        dev = self.device
        K = torch.tensor([b.K for b in live], dtype=torch.int64)
        M = torch.tensor([b.M for b in live], dtype=torch.int64)
        if not live:
            sidx = pidx = first = torch.zeros(0, dtype=torch.int64, device=dev)
        else:
            sidx = torch.cat([torch.arange(b.slot, b.slot + b.K) for b in live]).to(dev)
            pidx = torch.cat([torch.arange(b.plane, b.plane + b.M) for b in live]).to(dev)
            first = torch.repeat_interleave(torch.tensor([b.plane for b in live], dtype=torch.int64), K).to(dev)
        return dict(kind=torch.tensor([4]), K=K, M=M, planes=self.frames[pidx].cpu(),
                    win=(self.win[sidx] // PLANE_BYTES - first[:, None]).to(torch.int32).cpu(),
                    action=self.act[sidx].cpu(), reward=self.rew[sidx].cpu(), done=self.done[sidx].cpu(),
                    length=self.length[sidx].cpu(), h0=self.h0[sidx].cpu(), c0=self.c0[sidx].cpu(), prio=self.tree[sidx].cpu())

    def restore(self, snap: Dict[str, torch.Tensor]) -> int:
        """Re-ingest a ``snapshot`` block by block (ordinary ingestion, oldest first, so a smaller replay keeps the newest)."""
        K, M = snap["K"].tolist(), snap["M"].tolist()
        assert not K or int(snap["win"].shape[1]) == self.T, "snapshot of another sequence length"
        n, k0, m0 = 0, 0, 0
        for kk, mm in zip(K, M):
            sl = slice(k0, k0 + kk)
            n += self.ingest_planes(snap["planes"][m0:m0 + mm], snap["win"][sl], snap["action"][sl], snap["reward"][sl],
                                    snap["done"][sl], snap["length"][sl], snap["h0"][sl], snap["c0"][sl], snap["prio"][sl])
            k0, m0 = k0 + kk, m0 + mm
        return n

    def fill_synthetic(self, n: Optional[int] = None, n_action: int = 4, chunk: int = 400, seed: int = 0, run: int = 20) -> None:
        """Fill with random Atari-shaped sequences generated ON DEVICE (benchmarks: no dataset, no H2D), shaped like the
        vectorised actors' blocks: runs of ``run`` consecutive sequences of one environment, each starting
        ``seq_len - overlap`` frames after the one before it and sharing its planes with its neighbours."""
        n = self.capacity if n is None else n
        dev, T, cfg = self.device, self.T, self.cfg
        stride = _stride(cfg)
        g = torch.Generator(device=dev).manual_seed(seed)
        t = torch.arange(T, device=dev, dtype=torch.int32)
        done_n = 0
        while done_n < n:
            k = min(chunk // run * run or run, n - done_n)
            runs = (k + run - 1) // run
            per = stride * (run - 1) + self.P                                          # planes of one run
            planes = torch.randint(0, 256, (runs * per, PLANE_BYTES), dtype=torch.uint8, device=dev, generator=g)
            i = torch.arange(k, device=dev, dtype=torch.int32)
            win = ((i // run) * per + (i % run) * stride)[:, None] + t[None, :]
            self.ingest_planes(planes, win, torch.randint(0, n_action, (k, T), device=dev, generator=g).to(torch.int32),
                               torch.randn(k, T, device=dev, generator=g).clamp_(-1, 1), torch.zeros(k, T, device=dev),
                               torch.full((k,), T, dtype=torch.int32, device=dev),
                               torch.randn(k, self.Hd, device=dev, generator=g) * 0.1,
                               torch.randn(k, self.Hd, device=dev, generator=g) * 0.1,
                               torch.rand(k, device=dev, generator=g) + 0.1)
            done_n += k

    # ---- graph-capturable ------------------------------------------------------------------------------
    def sample(self, st: torch.Tensor, begin: bool = False, n_is: Optional[torch.Tensor] = None) -> None:
        """``n_is``: item count used by the IS weights instead of this shard's own (sharded PER: N_global / world)."""
        H.sumtree_sample(self.tree, self.lay, st, self.seed, self.idx, self.prio, self.gen, self.gen_s,
                         begin_n_items=(n_is if n_is is not None else self.n_items) if begin else None)
        H.seq_win_offsets(self.idx, self.win, self.T, self.frame_off)

    def sample_with_state(self, st: torch.Tensor, hs_a, cs_a, hs_b=None, cs_b=None, sums=(),
                          n_is: Optional[torch.Tensor] = None) -> None:
        """``sample(st, begin=True)`` and what the learner does with the draw right away - the window offsets and the stored
        recurrent state of the drawn sequences into the initial-state rows of one or two networks (+ up to two vector sums
        ``out = a + b``) - as ONE launch (ops.hip.seq_sample)."""
        H.seq_sample(self.tree, self.lay, st, self.seed, self.idx, self.prio, self.gen, self.gen_s,
                     n_is if n_is is not None else self.n_items, self.win, self.frame_off, self.T, self.h0, self.c0,
                     hs_a, cs_a, hs_b, cs_b, sums=sums)

    def update(self, new_prio: torch.Tensor, end_state=None, fault=None) -> None:
        H.sumtree_update(self.tree, self.lay, self.idx, new_prio, self.gen, self.gen_s, end_state=end_state, fault=fault)

    def remove_to_fit(self) -> int:
        return 0     # the rings evict on ingest

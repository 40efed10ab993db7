"""HBM-resident prioritized replay for transitions (replaces PrioritizedMemory + Replay thread,
libs/replay_memory.py:51-81, ape_x/replay.py).

Every 84x84 frame is stored ONCE: the store is a ring of single uint8 PLANES, and a transition is two
window starts into it - s = the 4 consecutive planes from ``win[i]``, s' = the 4 from ``win[cap + i]`` (byte offsets).
An actor ships the planes of a run of consecutive steps of one environment once (transport/codec.py kind 3); inside such
a run s_t and s_{t+1} share 3 planes and s'_t = s_{t+n}, so a transition costs ~1.1-1.3 planes = 8-9 KB instead of the 56
KB of two full stacks: 2 M transitions fit in ~18 GB instead of 113 GB. Episode starts need nothing special: the
environment's reset stack (4 copies of the first frame, wrapped_env.py:32-33) is simply 3 extra planes in the pool.
conv1 reads its operand straight out of this store through the sampled window offsets (csrc/hip/conv_direct.hip, the
``OFF`` forms) - a sampled batch is never gathered or copied.

Two rings advance together, both FIFO: transition slots (``act / rew / done / win / gen`` + the 64-ary sum-tree) and
planes. A block is written to contiguous ranges of both (a range that does not fit before the end of a ring starts at 0
again). Before a range of planes or slots is overwritten, every older block that still owns part of it is RETIRED as a
whole: its slots leave the tree (priority 0, generation bumped), so nothing samples a transition whose planes are gone.
The reference evicts FIFO too (``remove_to_fit``, replay_memory.py:70-75), every 100 steps instead of immediately:
documented deviation, SURVEY §7.3. ``sample`` and ``update`` are pure kernel launches on the current stream with
preallocated outputs, so they can sit inside a captured hipGraph; the number of live items is a device scalar for the same
reason.

Blocks of kind 1 (full HWC stacks per transition, no plane sharing: the scalar actors, Redis tools) are accepted as well
and split into 8 planes per transition on the device.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from ..ops import hip as H
from .rings import Block as _Block, RingPair

PLANE_BYTES = 84 * 84
STACK = 4
FRAME_BYTES = PLANE_BYTES * STACK        # one stacked observation as the environments deliver it (HWC)


class DevicePER(RingPair):
    def __init__(self, capacity: int, device="cuda", batch_size: int = 512, seed: int = 0,
                 planes_per_item: float = 1.3, plane_capacity: Optional[int] = None, max_ingest: int = 4096):
        self.capacity = int(capacity)
        self.device = torch.device(device)
        self.batch_size = batch_size
        self.seed = seed
        # planes: 1 per transition inside a run + (n + 3) per run boundary; 8 per transition for kind-1 blocks
        self.plane_capacity = int(plane_capacity if plane_capacity is not None else self.capacity * planes_per_item + 4 * STACK)
        self.plane_bytes = PLANE_BYTES
        self.lay = H.TreeLayout(self.capacity)
        dev = self.device
        self.tree = torch.zeros(self.lay.total, device=dev)
        self.gen = torch.zeros(self.capacity, dtype=torch.int32, device=dev)
        self.frames = torch.zeros(self.plane_capacity, PLANE_BYTES, dtype=torch.uint8, device=dev)     # the plane ring
        self.win = torch.zeros(2 * self.capacity, dtype=torch.int64, device=dev)       # byte offsets: s | s'
        self.act = torch.zeros(self.capacity, dtype=torch.int32, device=dev)
        self.rew = torch.zeros(self.capacity, device=dev)
        self.done = torch.zeros(self.capacity, device=dev)
        self.n_items = torch.zeros(1, dtype=torch.int64, device=dev)
        self.prio_pending = torch.zeros(self.capacity, device=dev)      # priorities of written, not yet committed blocks
        self._scratch = torch.zeros(max(max_ingest, 64), dtype=torch.int32, device=dev)
        self._init_rings()           # cursor / pcursor / count (host view; n_items is the device copy) / block deque
        self._warned_planes = False
        # sample outputs (fixed addresses: graph-capturable)
        B = batch_size
        self.idx = torch.zeros(B, dtype=torch.int32, device=dev)
        self.prio = torch.zeros(B, device=dev)
        self.gen_s = torch.zeros(B, dtype=torch.int32, device=dev)
        self.frame_off = torch.zeros(2 * B, dtype=torch.int64, device=dev)

    def __len__(self) -> int:
        return self.count

    def with_batch(self, batch_size: int) -> "DevicePER":
        """A second sampler over the SAME store (planes, tree, generations are shared, the sample buffers are its own):
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

    def bytes_per_item(self) -> float:
        """HBM bytes per transition of capacity (planes + tables + tree leaves)."""
        return (self.frames.numel() + self.win.numel() * 8 + self.capacity * 20) / self.capacity

    # ---- ingest -----------------------------------------------------------------------------------
    # Three stages, so that the bulk copy can run on its own stream while the learner keeps stepping (engine/ingest.py):
    # reserve_block (ranges of both rings; everything that still owns part of them leaves the tree: nothing samples it any
    # more), write_block (the data, any stream), commit_block (priorities + generation + item count, one launch).
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
        if M > 2 * STACK * K - 1 and not self._warned_planes and M * self.capacity > 2 * K * self.plane_capacity:
            # producers that send full stacks (codec kind 1: 8 planes per transition) into a plane ring sized for plane-pool
            # blocks (~1.3): whole blocks are evicted early and the replay holds far fewer items than its capacity says
            import warnings
            self._warned_planes = True
            warnings.warn("replay: blocks carry %.1f planes per transition but the plane ring was sized for %.1f: at most ~%d "
                          "of the %d transitions stay resident (size it with planes_per_item / DRL_PLANES_PER_ITEM)"
                          % (M / K, self.plane_capacity / self.capacity, self.plane_capacity * K // M, self.capacity))
        return RingPair.reserve_block(self, K, M, deferred)

    def write_block(self, blk: _Block, planes: torch.Tensor, win_s: torch.Tensor, win_s2: torch.Tensor, action: torch.Tensor,
                    reward: torch.Tensor, done: torch.Tensor) -> None:
        """planes u8 [M, 84*84] (or [M,84,84]); win_s / win_s2 int32 [K]: plane index, relative to the block's planes, of the
        first of the 4 consecutive planes of s / s'. Tensors on any device (pinned host memory: one DMA each)."""
        s, K, p, M = blk.slot, blk.K, blk.plane, blk.M
        dev = self.device
        self.frames[p:p + M].copy_(planes.reshape(M, PLANE_BYTES), non_blocking=True)
        ws = win_s.to(dev, non_blocking=True).to(torch.int64)
        ws2 = win_s2.to(dev, non_blocking=True).to(torch.int64)
        self.win[s:s + K] = (ws + p) * PLANE_BYTES
        self.win[self.capacity + s:self.capacity + s + K] = (ws2 + p) * PLANE_BYTES
        self.act[s:s + K].copy_(action, non_blocking=True)
        self.rew[s:s + K].copy_(reward, non_blocking=True)
        self.done[s:s + K].copy_(done, non_blocking=True)

    def write_block_stacks(self, blk: _Block, state: torch.Tensor, next_state: torch.Tensor, action: torch.Tensor,
                           reward: torch.Tensor, done: torch.Tensor) -> None:
        """A kind-1 block: full HWC stacks [K,84,84,4] per transition, no sharing -> 8 planes per transition."""
        K, p = blk.K, blk.plane
        assert blk.M == 2 * STACK * K
        dev = self.device
        both = torch.stack([state.reshape(K, 84, 84, STACK).to(dev, non_blocking=True),
                            next_state.reshape(K, 84, 84, STACK).to(dev, non_blocking=True)], 1)      # [K, 2, 84, 84, 4]
        self.frames[p:p + blk.M].view(K, 2, STACK, 84, 84).copy_(both.permute(0, 1, 4, 2, 3))
        base = torch.arange(K, device=dev, dtype=torch.int64) * (2 * STACK)
        s = blk.slot
        self.win[s:s + K] = (base + p) * PLANE_BYTES
        self.win[self.capacity + s:self.capacity + s + K] = (base + STACK + p) * PLANE_BYTES
        self.act[s:s + K].copy_(action, non_blocking=True)
        self.rew[s:s + K].copy_(reward, non_blocking=True)
        self.done[s:s + K].copy_(done, non_blocking=True)

    def commit_block(self, blk: _Block, prio: torch.Tensor) -> None:
        """``prio``: float32 on the device, at least K elements. The item count rides along with the tree launch."""
        if blk.state == 2:                  # its ranges were handed to a younger block while this one was in flight
            return
        blk.state = 1
        self.count += blk.K
        H.sumtree_ingest(self.tree, self.lay, blk.slot, prio, self.gen, self._scratch, n=blk.K, n_items=self.n_items,
                         n_items_val=self.count)

    # block-kind dispatch used by engine/ingest.py (views of one staged block) and ingest_block (decoded arrays)
    @staticmethod
    def block_need(kind: int, K: int, M: int = 0, views=None) -> Tuple[int, int]:
        from ..transport import codec
        return (K, M) if kind == codec.KIND_APEX_PLANES else (K, 2 * STACK * K)

    def write_staged(self, blk: _Block, kind: int, pin: torch.Tensor, fields, tail_for,
                     stream: Optional[int] = None) -> bool:
        """The ingest pipeline's write of a plane-pool block straight out of its pinned staging buffer ``pin`` (fields =
        codec.layout, or its {name: (offset, bytes)} map) as ONE native call on ``stream``: a DMA puts the planes into the
        ring, a DMA brings the tables (everything behind the planes) into the device buffer ``tail_for(nbytes)``, one launch
        scatters them (window offsets, actions, rewards, dones; priorities into ``prio_pending`` at the block's slots, for
        ``commit_blocks``). False for a block kind it does not take (the caller falls back to ``write_views``)."""
        from ..transport import codec
        if kind != codec.KIND_APEX_PLANES:
            return False
        f = fields if isinstance(fields, dict) else {name: (off, nb) for name, _, _, off, nb in fields}
        s, K, p, M = blk.slot, blk.K, blk.plane, blk.M
        p_off, p_nb = f["planes"]
        t0, t1 = f["win_s"][0], f["prio"][0] + f["prio"][1]
        cap = self.capacity
        assert p_nb == M * PLANE_BYTES and 0 <= p and p + M <= self.plane_capacity and 0 <= s and s + K <= cap
        assert t1 <= pin.numel() and p_off + p_nb <= pin.numel() and f["prio"][1] == 4 * K
        o = lambda name: f[name][0] - t0                                            # noqa: E731
        segs = [(o("action"), self.act.data_ptr() + 4 * s, K), (o("reward"), self.rew.data_ptr() + 4 * s, K),
                (o("done"), self.done.data_ptr() + 4 * s, K), (o("prio"), self.prio_pending.data_ptr() + 4 * s, K)]
        wins = [(o("win_s"), self.win.data_ptr() + 8 * s, K), (o("win_s2"), self.win.data_ptr() + 8 * (cap + s), K)]
        dst = self.frames.data_ptr() + p * PLANE_BYTES
        src = pin.data_ptr()
        H.ingest_block(dst, src + p_off, p_nb, tail_for(t1 - t0), src + t0, t1 - t0, segs, wins, p, stream=stream)
        return True

    def write_views(self, blk: _Block, kind: int, v) -> None:
        from ..transport import codec
        if kind == codec.KIND_APEX_PLANES:
            self.write_block(blk, v["planes"], v["win_s"], v["win_s2"], v["action"], v["reward"], v["done"])
        else:
            self.write_block_stacks(blk, v["state"], v["next_state"], v["action"], v["reward"], v["done"])

    def ingest_planes(self, planes, win_s, win_s2, action, reward, done, prio) -> int:
        K, M = int(action.shape[0]), int(planes.shape[0])
        if K == 0:
            return 0
        blk = self.reserve_block(K, M)
        self.write_block(blk, planes, win_s, win_s2, action, reward, done)
        self.commit_block(blk, prio.to(self.device, torch.float32).contiguous())
        return K

    def ingest(self, state: torch.Tensor, next_state: torch.Tensor, action: torch.Tensor, reward: torch.Tensor,
               done: torch.Tensor, prio: torch.Tensor) -> int:
        """Append K transitions given as full stacks (tensors on any device; frames uint8 HWC, flattened or not)."""
        K = int(action.shape[0])
        if K == 0:
            return 0
        blk = self.reserve_block(K, 2 * STACK * K)
        self.write_block_stacks(blk, state, next_state, action, reward, done)
        self.commit_block(blk, prio.to(self.device, torch.float32).contiguous())
        return K

    def ingest_block(self, block: Dict[str, np.ndarray]) -> int:
        """Consume a decoded experience block (transport/codec.py kind 1 or 3)."""
        t = lambda k: torch.from_numpy(np.ascontiguousarray(block[k]))  # noqa: E731
        if "planes" in block:
            return self.ingest_planes(t("planes"), t("win_s"), t("win_s2"), t("action"), t("reward"), t("done"), t("prio"))
        return self.ingest(t("state"), t("next_state"), t("action"), t("reward"), t("done"), t("prio"))

    # ---- snapshot / restore (checkpoint sidecar; SURVEY 5.4: the reference saves no replay at all) ---------------------
    def snapshot(self) -> Dict[str, torch.Tensor]:
        """The live content in ingestion order as plain CPU tensors (loadable with ``weights_only=True``): the planes of
        every live block, its windows RELATIVE to its own planes, actions / rewards / dones and the CURRENT priorities (tree
        leaves). Synchronises. 2 M transitions are ~18 GB: opt-in (``learner_node.py --save-replay``)."""
        torch.cuda.synchronize(self.device) if self.device.type == "cuda" else None
        live = [b for b in self._blocks if b.state == 1]
        dev = self.device
        K = torch.tensor([b.K for b in live], dtype=torch.int64)
        M = torch.tensor([b.M for b in live], dtype=torch.int64)
        if not live:
            z = lambda dt: torch.zeros(0, dtype=dt)                                   # noqa: E731
            return dict(kind=torch.tensor([3]), K=K, M=M, planes=torch.zeros(0, PLANE_BYTES, dtype=torch.uint8), win_s=z(torch.int32),
                        win_s2=z(torch.int32), action=z(torch.int32), reward=z(torch.float32), done=z(torch.float32),
                        prio=z(torch.float32))
        sidx = torch.cat([torch.arange(b.slot, b.slot + b.K) for b in live]).to(dev)
        pidx = torch.cat([torch.arange(b.plane, b.plane + b.M) for b in live]).to(dev)
        first = torch.repeat_interleave(torch.tensor([b.plane for b in live], dtype=torch.int64), K).to(dev)
        return dict(kind=torch.tensor([3]), K=K, M=M, planes=self.frames[pidx].cpu(),
                    win_s=(self.win[sidx] // PLANE_BYTES - first).to(torch.int32).cpu(),
                    win_s2=(self.win[self.capacity + sidx] // PLANE_BYTES - first).to(torch.int32).cpu(),
                    action=self.act[sidx].cpu(), reward=self.rew[sidx].cpu(), done=self.done[sidx].cpu(),
                    prio=self.tree[sidx].cpu())

    def restore(self, snap: Dict[str, torch.Tensor], group_items: int = 4096) -> int:
        """Re-ingest a ``snapshot`` (into an empty or a running replay: ordinary ingestion, oldest first, so a smaller
        replay keeps the newest part). Neighbouring blocks are merged into groups of ~``group_items`` transitions."""
        K, M = snap["K"].tolist(), snap["M"].tolist()
        n, k0, m0, i = 0, 0, 0, 0
        while i < len(K):
            j, kk, mm, shift = i, 0, 0, []
            while j < len(K) and (j == i or kk + K[j] <= group_items):
                shift += [mm] * K[j]
                kk, mm, j = kk + K[j], mm + M[j], j + 1
            sh = torch.tensor(shift, dtype=torch.int32)
            sl, pl = slice(k0, k0 + kk), slice(m0, m0 + mm)
            n += self.ingest_planes(snap["planes"][pl], snap["win_s"][sl] + sh, snap["win_s2"][sl] + sh, snap["action"][sl],
                                    snap["reward"][sl], snap["done"][sl], snap["prio"][sl])
            k0, m0, i = k0 + kk, m0 + mm, j
        return n

    def fill_synthetic(self, n: Optional[int] = None, n_action: int = 4, chunk: int = 65536, seed: int = 0, nstep: int = 3,
                       run: int = 64) -> None:
        """Fill with random Atari-shaped transitions generated ON DEVICE (benchmarks: no dataset, no H2D), shaped like the
        vectorised actors' blocks: runs of ``run`` consecutive steps of one environment, s' = the stack ``nstep`` steps on."""
        n = self.capacity if n is None else n
        dev = self.device
        g = torch.Generator(device=dev).manual_seed(seed)
        done_n = 0
        while done_n < n:
            k = min(chunk, n - done_n)
            runs = (k + run - 1) // run
            per = run + nstep + STACK - 1                                   # planes of one run
            planes = torch.randint(0, 256, (runs * per, PLANE_BYTES), dtype=torch.uint8, device=dev, generator=g)
            i = torch.arange(k, device=dev)
            ws = ((i // run) * per + i % run).to(torch.int32)
            self.ingest_planes(planes, ws, ws + nstep, torch.randint(0, n_action, (k,), device=dev, generator=g).to(torch.int32),
                               torch.randn(k, device=dev, generator=g).clamp_(-1, 1),
                               (torch.rand(k, device=dev, generator=g) < 0.01).float(),
                               torch.rand(k, device=dev, generator=g) + 0.1)
            done_n += k

    def gather_stacks(self, idx: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """(s, s') of the slots ``idx`` as HWC stacks [n,84,84,4] (tests / tools: the kernels never materialise them)."""
        out = []
        flat = self.frames.view(-1)
        ar = torch.arange(FRAME_BYTES, device=self.device)
        for half in (0, self.capacity):
            off = self.win[idx.long() + half]
            st = flat[(off[:, None] + ar[None, :])].view(-1, STACK, 84, 84)
            out.append(st.permute(0, 2, 3, 1).contiguous())
        return out[0], out[1]

    # ---- sample / update (graph-capturable) -----------------------------------------------------
    def sample(self, st: torch.Tensor, begin: bool = False, is_n_scale: float = 0.0,
               n_is: Optional[torch.Tensor] = None) -> None:
        """Draw ``batch_size`` slots ∝ priority into ``self.idx / prio / gen_s / frame_off``. ``begin`` fuses
        step_begin (beta / n_items refresh, fresh Philox counter) into the sampling kernel; ``is_n_scale`` > 0 also
        accumulates the batch's max IS weight there (see ops.hip.sumtree_sample). ``n_is`` (int64 device scalar)
        replaces the live item count in the IS weights: sharded PER passes N_global / world (engine.shard_n_is)."""
        n = (n_is if n_is is not None else self.n_items) if begin else None
        H.sumtree_sample(self.tree, self.lay, st, self.seed, self.idx, self.prio, self.gen, self.gen_s, self.frame_off,
                         self.win, begin_n_items=n, is_n_scale=is_n_scale)

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
        return H.sample_extra(self.tree, self.lay, st, self.seed, idx, prio, self.gen, gen_s, foff, self.win,
                              n_is if n_is is not None else self.n_items, is_n_scale)

    def handover_extra(self):
        """Copies next-batch buffers -> live sample buffers (ops.hip.rmsprop_step(copies=...))."""
        idx, prio, gen_s, foff = self._next_buffers()
        return H.copy_extra([(self.idx, idx), (self.prio, prio), (self.gen_s, gen_s), (self.frame_off, foff)])

    def update(self, new_prio: torch.Tensor, end_state: Optional[torch.Tensor] = None) -> None:
        """Write back priorities of the last sample; slots overwritten since are skipped (generation check).
        ``end_state`` fuses step_end into the tree rebuild launch."""
        H.sumtree_update(self.tree, self.lay, self.idx, new_prio, self.gen, self.gen_s, end_state=end_state)

    def update_stage(self, kind: int, new_prio: torch.Tensor, end_state: Optional[torch.Tensor] = None, sumsq=None, zero=None):
        """The same write-back as ``update`` split into its two stages (kind 1 = leaves + level 1, kind 2 = upper levels +
        optional step_end) as ``TreeExtra`` records: the learner engine hands them to two consecutive fused backward
        launches (ops.hip.conv_bwd) instead of launching them at the tail of the step. ``sumsq`` / ``zero``: gradient-norm
        partials and a zeroing that ride along in the same record (ops.hip.tree_extra)."""
        return H.tree_extra(kind, self.tree, self.lay, self.idx, new_prio, self.gen, self.gen_s, end_state=end_state,
                            sumsq=sumsq, zero=zero)

    # host-API parity with PrioritizedMemory (tests / tools; not on the hot path)
    def update_priorities(self, indices, priorities) -> None:
        idx = torch.as_tensor(indices, dtype=torch.int32, device=self.device)
        val = torch.as_tensor(priorities, dtype=torch.float32, device=self.device)
        H.sumtree_update(self.tree, self.lay, idx.contiguous(), val.contiguous())

    def remove_to_fit(self) -> int:
        return 0     # the rings evict on ingest

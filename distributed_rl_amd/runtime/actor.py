"""Actors (reference: ape_x/actor.py:9-102, r2d2/actor.py:10-91).

One actor *process* drives ``E`` environments with one batched forward per step (E = 1 reproduces
the reference's one-env-per-process actors). Per-env ε comes from the reference's ladder
``0.4^(1+7·(i−1)/(N−1))`` evaluated over the global env index, so ``N`` processes × ``E`` envs
explore like ``N·E`` reference actors (actor.py:66-69).

Loop per env step (Ape-X): ε-greedy action → env.step → reward clip → n-step assembly → local buffer;
when ≥ ``actor_batch`` transitions are buffered, compute their initial priorities with the local
net (online == target, actor.py:93 — SURVEY Q3) and publish one binary block; every
``actor_param_period`` env steps poll the parameter register (actor.py:101-102).
R2D2 adds recurrent-state recording, overlapping sequence assembly and sequence priorities
(r2d2/actor.py:39-87); pulling params also refreshes the actor-side target net (r2d2/actor.py:89-91).

The inference function is pluggable: ``policy`` is either an ``nn.Module`` (CPU / oracle path) or an
``engine.ActorEngine`` / ``engine.R2D2ActorEngine`` (gfx950 path: fused torso → [LSTM step →] head → ε-greedy with
device-side RNG; R2D2 sequence priorities through the learner's time-batched forward + fused TD kernel).
"""
from __future__ import annotations

import os
import time
from typing import List, Optional

import numpy as np
import torch

from ..algo.nstep import SequenceItem
from ..algo.vec_assembly import (PlaneLog, VecNStepAssembler, VecSequenceAssembler, plane_block, sequence_plane_block,
                                 stacks_from_planes)
from ..config import ApexConfig, R2D2Config
from ..models.packing import ParamPack
from ..transport import ExperienceChannel, ParamChannel, codec
from ..utils import epsilon_ladder


class _LocalApexBuffer:
    """Actor-side staging of finished n-step transitions (the reference's ``ReplayMemory(1000)``, ape_x/actor.py:49):
    preallocated arrays, oldest first; overflow drops the oldest like a bounded deque."""

    def __init__(self, capacity: int):
        self.capacity = capacity
        self.n = 0
        self.state = self.next_state = None
        self.action = np.zeros(capacity, dtype=np.int32)
        self.reward = np.zeros(capacity, dtype=np.float32)
        self.done = np.zeros(capacity, dtype=np.float32)
        self.env = np.zeros(capacity, dtype=np.int64)          # source env + positions of s / s' in its plane stream
        self.qs = np.zeros(capacity, dtype=np.int64)
        self.qs2 = np.zeros(capacity, dtype=np.int64)

    def clear(self):
        self.n = 0

    def _drop(self, k: int) -> None:
        rest = self.n - k
        for a in (self.state, self.next_state, self.action, self.reward, self.done, self.env, self.qs, self.qs2):
            a[:rest] = a[k:self.n]
        self.n = rest

    def extend(self, b):
        k = int(b.action.shape[0])
        if self.state is None:
            self.state = np.zeros((self.capacity,) + tuple(b.state.shape[1:]), dtype=np.uint8)
            self.next_state = np.zeros_like(self.state)
        if k > self.capacity:
            b = type(b)(*[x[-self.capacity:] for x in b])
            k = self.capacity
        over = self.n + k - self.capacity
        if over > 0:                                      # bounded deque semantics (ReplayMemory(1000), actor.py:49)
            self._drop(over)
        n = self.n
        self.state[n:n + k], self.next_state[n:n + k] = b.state, b.next_state
        self.action[n:n + k], self.reward[n:n + k], self.done[n:n + k] = b.action, b.reward, b.done
        self.env[n:n + k] = b.env
        if b.qs is not None:
            self.qs[n:n + k], self.qs2[n:n + k] = b.qs, b.qs2
        self.n = n + k

    def __len__(self):
        return self.n

    def take(self, n: int):
        blk = dict(state=self.state[:n].copy(), next_state=self.next_state[:n].copy(), action=self.action[:n].copy(),
                   reward=self.reward[:n].copy(), done=self.done[:n].copy(), env=self.env[:n].copy(), qs=self.qs[:n].copy(),
                   qs2=self.qs2[:n].copy())
        self._drop(n)
        return blk


class _LocalPlaneBuffer:
    """Actor-side staging when every frame is kept once (PlaneLog): per env, the finished n-step transitions as two
    positions in the env's plane stream + (action, return, done) - no frames at all. A block is cut from ONE env's
    transitions (each env then publishes like one actor of the reference: ``actor_batch`` consecutive steps,
    ape_x/actor.py:86-99), so it spans one run of the stream: ``actor_batch + n + 3`` planes for ``actor_batch``
    transitions. A block that mixed all envs would carry n + 3 extra planes per env."""

    def __init__(self, num_envs: int, capacity: int):
        self.E, self.cap = num_envs, capacity
        self.n = np.zeros(num_envs, dtype=np.int64)
        self.qs = np.zeros((num_envs, capacity), dtype=np.int64)
        self.qs2 = np.zeros((num_envs, capacity), dtype=np.int64)
        self.action = np.zeros((num_envs, capacity), dtype=np.int32)
        self.reward = np.zeros((num_envs, capacity), dtype=np.float32)
        self.done = np.zeros((num_envs, capacity), dtype=np.float32)

    def _arrays(self):
        return self.qs, self.qs2, self.action, self.reward, self.done

    def _drop(self, e: int, k: int) -> None:
        rest = int(self.n[e]) - k
        for a in self._arrays():
            a[e, :rest] = a[e, k:k + rest]
        self.n[e] = rest

    def extend(self, b) -> None:
        env = np.asarray(b.env)
        full = np.unique(env[self.n[env] >= self.cap])
        for e in full:                                       # bounded deque semantics per env (ReplayMemory(1000), actor.py:49)
            self._drop(int(e), self.cap // 2)
        src = (b.qs, b.qs2, b.action, b.reward, b.done)
        if len(np.unique(env)) == len(env):                  # the usual step: one finished window per env
            at = self.n[env]
            for a, v in zip(self._arrays(), src):
                a[env, at] = v
            self.n[env] += 1
            return
        for i, e in enumerate(env):                          # episode tails: several windows of one env in one step
            if self.n[e] >= self.cap:
                self._drop(int(e), self.cap // 2)
            for a, v in zip(self._arrays(), src):
                a[e, self.n[e]] = v[i]
            self.n[e] += 1

    def __len__(self) -> int:
        return int(self.n.sum())

    def ready(self, k: int):
        r = np.nonzero(self.n >= k)[0]
        return int(r[0]) if len(r) else None

    def oldest_q(self) -> np.ndarray:
        """Stream position of every env's oldest buffered transition (a huge value where nothing is buffered)."""
        return np.where(self.n > 0, self.qs[:, 0], np.iinfo(np.int64).max)

    def take(self, e: int, k: int):
        out = dict(qs=self.qs[e, :k].copy(), qs2=self.qs2[e, :k].copy(), action=self.action[e, :k].copy(),
                   reward=self.reward[e, :k].copy(), done=self.done[e, :k].copy())
        self._drop(e, k)
        return out


class Actor:
    def __init__(self, actor_no: int, env, policy_net, cfg, exp: ExperienceChannel, params: ParamChannel,
                 metrics=None, num_total_actors: int = 4, device=torch.device("cpu"), target_net=None,
                 heartbeat=None, seed: Optional[int] = None, render_every: int = 0, engine=None):
        self.no, self.env, self.cfg = actor_no, env, cfg
        self.E = env.E
        self.name = "actor_%d" % actor_no
        self.net = policy_net
        self.net.eval()
        self.engine = engine          # optional engine.ActorEngine: gfx950 inference + on-device ε-greedy
        self.device = device
        self.exp, self.params, self.metrics = exp, params, metrics
        self.heartbeat = heartbeat
        self.is_r2d2 = isinstance(cfg, R2D2Config)
        self.eps = torch.from_numpy(epsilon_ladder(self.E, (actor_no - 1) * self.E + 1, num_total_actors * self.E))
        self.rng = np.random.default_rng(seed if seed is not None else (os.getpid() * 7919 + actor_no))
        self._pack = ParamPack.from_module(policy_net)
        self._version = 0
        self.target = target_net
        if self.is_r2d2:
            assert target_net is not None, "R2D2 actors carry a target net (r2d2/actor.py:21-23)"
            self.target.load_state_dict(self.net.state_dict())
            self.target.eval()
            self.net.hx = torch.zeros(self.E, self.net.hidden, device=device)
            self.net.cx = torch.zeros(self.E, self.net.hidden, device=device)
            # every frame once on the wire and in the replay, across the overlapping sequences of an env too: the frames
            # live in a per-env plane log, sequences are rows of positions, and an env publishes its finished sequences
            # together with the planes they span (codec kind 4). DRL_PLANE_BLOCKS=0: full stacks per step (kind 2).
            self.plane_log = None
            if os.environ.get("DRL_PLANE_BLOCKS", "1") != "0":
                stride = cfg.seq_len - cfg.overlap
                cap = stride * cfg.actor_batch + cfg.total_len + 64
                self.plane_log = PlaneLog(self.E, cap=cap)
            self.asm = VecSequenceAssembler(self.E, cfg.nstep, cfg.gamma, cfg.burn_in, cfg.seq_len, cfg.overlap,
                                            cfg.reference_nstep_order, flush_tail=not cfg.reference_nstep_order,
                                            hidden=self.net.hidden, store_states=self.plane_log is None)
            self.local: List[SequenceItem] = []
            self.local_env: List[List[SequenceItem]] = [[] for _ in range(self.E)]      # plane mode: finished sequences per env
        else:
            self.asm = VecNStepAssembler(self.E, cfg.nstep, cfg.gamma, cfg.reference_nstep_order,
                                         flush_tail=not cfg.reference_nstep_order)
            self.local = _LocalApexBuffer(cfg.actor_local_capacity)
            # every frame once on the wire and in the replay: blocks carry the planes their transitions span + two window
            # positions per transition (codec kind 3). DRL_PLANE_BLOCKS=0: full stacks per transition (kind 1).
            self.plane_log = None
            if os.environ.get("DRL_PLANE_BLOCKS", "1") != "0":
                self.plane_log = PlaneLog(self.E, cap=4 * (cfg.actor_batch + cfg.nstep + 8) + 64)
                self.local = _LocalPlaneBuffer(self.E, max(2 * cfg.actor_batch, min(cfg.actor_local_capacity, 256)))
                self.asm.store_states = False             # the frames live in the plane log: positions only from here on
        self.env_steps = 0
        self.published_blocks = 0
        self.dropped_blocks = 0
        self.lost_to_plane_log = 0        # transitions whose planes had left the plane log (must stay 0: _urgent_envs)
        self.episodes = 0
        self.render_every = render_every
        if metrics is not None:
            self.timers = metrics
        else:
            from ..obs import Metrics
            self.timers = Metrics([])
        self.report_interval = float(os.environ.get("DRL_ACTOR_REPORT_S", "2.0"))

    # ---- parameters -------------------------------------------------------------------------------
    def pull_params(self) -> bool:
        # a device engine takes the parameters as a tensor: channels that receive on the GPU (collective broadcast over
        # RCCL) hand it over without a round trip through host memory
        pull = getattr(self.params, "pull_tensor", None) if self.engine is not None else None
        if pull is not None and getattr(self.params, "device_mirror", False):
            got = pull(self._version, self.device)           # shm register: D2D out of the learner's mirror when it is on our GPU
        else:
            got = pull(self._version) if pull is not None else self.params.pull(self._version)
        if got is None:
            return False
        self._version, flat = got
        if self.metrics is not None:
            self.metrics.scalar("Params version %s" % self.name, self.env_steps, self._version)
        if self.engine is not None:
            self.engine.load_flat(flat)
            return True
        self._pack.flat.copy_(torch.from_numpy(flat))
        self.net.load_state_dict(self._pack.state_dict(), strict=True)
        if self.is_r2d2:
            self.target.load_state_dict(self.net.state_dict())
        return True

    # ---- acting -----------------------------------------------------------------------------------
    @torch.no_grad()
    def act(self, obs_u8: np.ndarray) -> np.ndarray:
        if self.engine is not None:
            return self.engine.act(obs_u8)
        x = torch.from_numpy(obs_u8).to(self.device)
        q = self.net(x).float().cpu()
        greedy = q.argmax(1)
        explore = torch.from_numpy(self.rng.random(self.E)) <= self.eps.double()     # random() > eps → greedy
        rand = torch.from_numpy(self.rng.integers(0, q.shape[1], self.E))
        return torch.where(explore, rand, greedy).numpy()

    def _clip(self, r: np.ndarray) -> np.ndarray:
        c = self.cfg.reward_clip
        return np.clip(r, -c, c) if c is not None else r

    # ---- publishing -------------------------------------------------------------------------------
    @torch.no_grad()
    def _publish_apex_planes(self, e: int, k: Optional[int] = None) -> None:
        """One env's next ``actor_batch`` (or ``k``) transitions as a plane-pool block: the planes they span once + two
        window positions each; initial priorities from the same pool (no stack is ever materialised on the engine path)."""
        cfg, t = self.cfg, self.local.take(e, self.cfg.actor_batch if k is None else k)
        K = len(t["action"])
        try:
            planes, win_s, win_s2 = plane_block(self.plane_log, np.full(K, e), t["qs"], t["qs2"])
        except IndexError:                  # outlived the plane log: _urgent_envs() publishes before this can happen
            self.dropped_blocks += 1
            self.lost_to_plane_log += K
            return
        if self.engine is not None:
            prio = self.engine.priorities_planes(planes, win_s, win_s2, t["action"], t["reward"], t["done"], cfg.gamma_n,
                                                 cfg.per_alpha, cfg.prio_eps)
        else:
            dev = self.device
            ar = np.arange(4)
            st = lambda w: torch.from_numpy(np.ascontiguousarray(np.transpose(planes[w[:, None] + ar], (0, 2, 3, 1)))).to(dev)  # noqa: E731
            td = self.net.td_batch(self.net, st(win_s), torch.from_numpy(t["action"]).long().to(dev),
                                   torch.from_numpy(t["reward"]).to(dev), st(win_s2), torch.from_numpy(t["done"]).to(dev), None,
                                   cfg.per_alpha, cfg.gamma_n)
            prio = td.prio.float().cpu().numpy()
        self._send(dict(planes=planes, win_s=win_s, win_s2=win_s2, action=t["action"], reward=t["reward"], done=t["done"],
                        prio=prio), codec.KIND_APEX_PLANES)

    @torch.no_grad()
    def _urgent_envs(self) -> np.ndarray:
        """Envs whose oldest buffered transition is about to leave the plane log. The log is sized for environments that
        shift their stack by one frame per step (1 new plane per step); one that does not (``SyntheticRandom-v0``, a
        non-stacking env, a frame-skip mismatch) appends up to 2 x stack planes per step, so a full ``actor_batch`` would
        outlive the ring. Such envs publish what they have NOW, as a shorter block - nothing is ever dropped for the
        log's sake, and the memory stays what the common case needs."""
        log = self.plane_log
        margin = 2 * log.stack * (self.cfg.nstep + 3)
        return np.nonzero(log.head - self.local.oldest_q() > log.cap - margin)[0]

    def _publish_apex(self) -> None:
        blk = self.local.take(self.cfg.actor_batch)
        dev = self.device
        if self.engine is not None:
            prio = self.engine.priorities(blk["state"], blk["next_state"], blk["action"], blk["reward"], blk["done"],
                                          self.cfg.gamma_n, self.cfg.per_alpha, self.cfg.prio_eps)
        else:
            td = self.net.td_batch(self.net, torch.from_numpy(blk["state"]).to(dev), torch.from_numpy(blk["action"]).to(dev),
                                   torch.from_numpy(blk["reward"]).to(dev), torch.from_numpy(blk["next_state"]).to(dev),
                                   torch.from_numpy(blk["done"]).to(dev), None, self.cfg.per_alpha, self.cfg.gamma_n)
            prio = td.prio.float().cpu().numpy()
        for k in ("env", "qs", "qs2"):
            blk.pop(k)
        blk["prio"] = prio
        self._send(blk, codec.KIND_APEX)

    @torch.no_grad()
    def _publish_r2d2(self) -> None:
        seqs, self.local = self.local[:self.cfg.actor_batch], self.local[self.cfg.actor_batch:]
        dev = self.device
        if self.engine is not None:       # gfx950 path: time-batched forward of both nets + fused TD kernel
            prio = self.engine.sequence_priorities(
                np.stack([s.frames for s in seqs], 1), np.stack([s.action for s in seqs], 1),
                np.stack([s.reward for s in seqs], 1), np.stack([s.done for s in seqs], 1),
                np.array([s.length for s in seqs], dtype=np.int32), np.stack([s.h0 for s in seqs]),
                np.stack([s.c0 for s in seqs]))
            self._send_sequences(seqs, prio)
            return
        frames = torch.from_numpy(np.stack([s.frames for s in seqs], 1)).to(dev)          # (T,K,…)
        action = torch.from_numpy(np.stack([s.action for s in seqs], 1)).to(dev)
        reward = torch.from_numpy(np.stack([s.reward for s in seqs], 1)).to(dev)
        done = torch.from_numpy(np.stack([s.done for s in seqs], 1)).to(dev)
        lengths = torch.tensor([s.length for s in seqs], device=dev)
        h0 = torch.from_numpy(np.stack([s.h0 for s in seqs])).to(dev)
        c0 = torch.from_numpy(np.stack([s.c0 for s in seqs])).to(dev)
        td = self.net.sequence_td(self.target, frames, action, reward, done, lengths, h0, c0, None, self.cfg.eta,
                                  self.cfg.gamma_n, self.cfg.rescale_eps, require_grad=False,
                                  done_offset=self.cfg.nstep if self.cfg.reference_nstep_order else 0)
        prio = td.prio.float().cpu().numpy()
        self._send_sequences(seqs, prio)

    @torch.no_grad()
    def _publish_r2d2_planes(self, e: int, k: int) -> None:
        """The next ``k`` finished sequences of env ``e`` as ONE plane-pool block: the planes they span once (consecutive
        sequences overlap by burn_in + overlap steps, r2d2/actor.py:56-60) + a row of window positions per sequence;
        priorities from the same pool (the engine reads plane windows in place, like the learner does)."""
        cfg = self.cfg
        seqs, self.local_env[e] = self.local_env[e][:k], self.local_env[e][k:]
        try:
            planes, win = sequence_plane_block(self.plane_log, e, seqs)
        except IndexError:                  # outlived the plane log: _urgent_seq_envs() publishes before this can happen
            self.dropped_blocks += 1
            self.lost_to_plane_log += len(seqs)
            return
        action = np.stack([s.action for s in seqs]).astype(np.int32)
        reward, done = np.stack([s.reward for s in seqs]), np.stack([s.done for s in seqs])
        length = np.array([s.length for s in seqs], dtype=np.int32)
        h0, c0 = np.stack([s.h0 for s in seqs]), np.stack([s.c0 for s in seqs])
        if self.engine is not None:
            prio = self.engine.sequence_priorities_planes(planes, win, action, reward, done, length, h0, c0)
        else:
            dev = self.device
            frames = torch.from_numpy(stacks_from_planes(planes, win.T)).to(dev)               # (T, K, 84, 84, 4)
            td = self.net.sequence_td(self.target, frames, torch.from_numpy(action.T.astype(np.int64)).to(dev),
                                      torch.from_numpy(reward.T.copy()).to(dev), torch.from_numpy(done.T.copy()).to(dev),
                                      torch.from_numpy(length).to(dev), torch.from_numpy(h0).to(dev), torch.from_numpy(c0).to(dev),
                                      None, cfg.eta, cfg.gamma_n, cfg.rescale_eps, require_grad=False,
                                      done_offset=cfg.nstep if cfg.reference_nstep_order else 0)
            prio = td.prio.float().cpu().numpy()
        blk = dict(planes=planes, win=win, action=action, reward=reward, done=done, length=length, h0=h0, c0=c0,
                   prio=prio.astype(np.float32))
        if codec.block_bytes(blk, codec.KIND_R2D2_PLANES) > getattr(self.exp, "slot_bytes", 1 << 62) and len(seqs) > 1:
            # too large for a ring slot (many episode boundaries in the span): two blocks, each with the planes of its half
            h = len(seqs) // 2
            for a, b in ((0, h), (h, len(seqs))):
                lo, hi = int(win[a:b].min()), int(win[a:b].max()) + self.plane_log.stack
                part = {n_: v[a:b] for n_, v in blk.items() if n_ != "planes"}
                part["planes"], part["win"] = planes[lo:hi], win[a:b] - lo
                self._send(part, codec.KIND_R2D2_PLANES)
            return
        self._send(blk, codec.KIND_R2D2_PLANES)

    def _urgent_seq_envs(self) -> np.ndarray:
        """Envs whose oldest sequence (finished or still being assembled) is about to leave the plane log - an env that
        adds more than one plane per step (see _urgent_envs): publish what is finished, now."""
        log = self.plane_log
        oldest = np.array([min([int(s.q[0]) for s in self.local_env[e]] or [np.iinfo(np.int64).max]) for e in range(self.E)])
        margin = log.stack * (self.cfg.total_len // 4 + self.cfg.nstep + 3)
        return np.nonzero((log.head - oldest > log.cap - margin) & (oldest < np.iinfo(np.int64).max))[0]

    def _send_sequences(self, seqs, prio) -> None:
        T, H = seqs[0].frames.shape[0], seqs[0].h0.shape[0]
        per = max(1, (getattr(self.exp, "slot_bytes", 1 << 62) - codec.HEADER_BYTES - 4096) //
                  codec.r2d2_block_bytes(1, T, H, tuple(seqs[0].frames.shape[1:])))
        for i in range(0, len(seqs), per):
            part = seqs[i:i + per]
            blk = dict(frames=np.stack([s.frames for s in part]), action=np.stack([s.action for s in part]).astype(np.int32),
                       reward=np.stack([s.reward for s in part]), done=np.stack([s.done for s in part]),
                       length=np.array([s.length for s in part], dtype=np.int32), h0=np.stack([s.h0 for s in part]),
                       c0=np.stack([s.c0 for s in part]), prio=prio[i:i + per])
            self._send(blk, codec.KIND_R2D2)

    def _send(self, blk, kind) -> None:
        for _ in range(200):                                 # bounded back-pressure wait (≈2 s), then drop
            if self.exp.push(blk, kind, self.no, self.env_steps):
                self.published_blocks += 1
                return
            time.sleep(0.01)
        self.dropped_blocks += 1

    # ---- main loop --------------------------------------------------------------------------------
    def _recurrent_state_before(self):
        """(h, c) before the next observation is consumed, for the envs whose step may start a sequence (need_state):
        a small gather instead of a (2, E, 512) device-to-host copy on every env step."""
        need = np.nonzero(self.asm.need_state())[0]
        if len(need) == 0:
            return None, None
        if self.engine is not None:
            return self.engine.get_state(rows=need)
        idx = torch.from_numpy(need).to(self.net.hx.device)
        return self.net.hx.detach()[idx].cpu().numpy(), self.net.cx.detach()[idx].cpu().numpy()

    def _report(self, t: int, now: float, t_prev: float, steps_prev: int) -> None:
        """Actor frames/s, dropped blocks, parameter lag and the per-phase wall clock -> metrics sinks."""
        m = self.metrics
        sps = (self.env_steps - steps_prev) / max(1e-9, now - t_prev)
        m.scalar("Env steps per second %s" % self.name, self.env_steps, sps, force=True)
        m.scalar("Env frames per second %s" % self.name, self.env_steps, sps * getattr(self.env, "frameskip", 4), force=True)
        m.scalar("Published blocks %s" % self.name, self.env_steps, self.published_blocks, force=True)
        m.scalar("Dropped blocks %s" % self.name, self.env_steps, self.dropped_blocks, force=True)
        try:
            m.scalar("Param lag %s" % self.name, self.env_steps, self.params.version() - self._version, force=True)
        except Exception:  # noqa: BLE001 - a channel without a cheap version counter
            pass
        for name, rep in self.timers.phase_report().items():
            if name.startswith("actor/"):
                m.scalar("Phase ms %s %s" % (name, self.name), self.env_steps, rep["mean_ms"], force=True)

    def run(self, max_steps: Optional[int] = None, stop_event=None) -> None:
        cfg = self.cfg
        ph = self.timers.phase
        self.env.reset()
        scores = np.zeros(self.E)
        t = 0
        log = self.plane_log
        q_state = q_next = None
        t_rep, steps_rep = time.time(), self.env_steps
        while max_steps is None or t < max_steps:
            if stop_event is not None and stop_event.is_set():
                break
            obs = self.env.current_obs()
            with ph("actor/act"):
                if self.is_r2d2:
                    h_before, c_before = self._recurrent_state_before()
                action = self.act(obs)
            with ph("actor/env"):
                next_obs, reward, done, info = self.env.step(action)
            scores += reward
            r = self._clip(reward)
            with ph("actor/assemble"):
                if self.is_r2d2:
                    if log is not None:
                        for sq in self.asm.step(obs, action, r, done, h_before, c_before, log.observe(obs)):
                            self.local_env[sq.env].append(sq)
                    else:
                        self.local.extend(self.asm.step(obs, action, r, done, h_before, c_before))
                    if done.any():
                        if self.engine is not None:
                            self.engine.reset(done)
                        else:
                            self.net.reset(torch.from_numpy(done))
                else:
                    if log is not None:
                        q_state = log.observe(obs)                   # position of obs in every env's plane stream ...
                        q_next = log.observe(next_obs)               # ... and of what the step led to (terminal frames too)
                    out = self.asm.step(obs, action, r, next_obs, done, q_state, q_next)
                    if out is not None:
                        self.local.extend(out)
            if done.any():
                for e in np.nonzero(done)[0]:
                    self.episodes += 1
                    if self.metrics is not None:
                        self.metrics.scalar("Score %s" % self.name, self.episodes, scores[e])
                    scores[e] = 0.0
            if self.metrics is not None and self.render_every and t % self.render_every == 0:
                rgb = getattr(self.env, "render_rgb", None)          # emulator screen when there is one, else the newest plane
                self.metrics.image("Frame %s" % self.name, rgb(0) if rgb is not None else obs[0][..., -1])
            with ph("actor/publish"):
                if log is not None and self.is_r2d2:
                    for e in range(self.E):
                        while len(self.local_env[e]) >= cfg.actor_batch:
                            self._publish_r2d2_planes(e, cfg.actor_batch)
                    for e in self._urgent_seq_envs():
                        self._publish_r2d2_planes(int(e), len(self.local_env[e]))
                elif log is not None:
                    while True:
                        e = self.local.ready(cfg.actor_batch)
                        if e is None:
                            break
                        self._publish_apex_planes(e)
                    for e in self._urgent_envs():
                        self._publish_apex_planes(int(e), int(self.local.n[e]))
                else:
                    while len(self.local) >= cfg.actor_batch:
                        self._publish_r2d2() if self.is_r2d2 else self._publish_apex()
            t += 1
            self.env_steps += self.E
            if t % cfg.actor_param_period == 0:
                with ph("actor/params"):
                    self.pull_params()
            if self.heartbeat is not None and t % 16 == 0:
                self.heartbeat.tick(self.env_steps)
            if self.metrics is not None and t % 16 == 0:
                now = time.time()
                if now - t_rep >= self.report_interval:
                    self._report(t, now, t_rep, steps_rep)
                    t_rep, steps_rep = now, self.env_steps

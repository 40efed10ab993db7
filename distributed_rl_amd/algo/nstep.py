"""n-step transition and R2D2 sequence assembly (reference: ape_x/actor.py:60-90, r2d2/actor.py:26-75).

Both assemblers are *vectorised over E environments* (one actor process drives many envs on the
GPU — SURVEY §2.3) and work on raw uint8 frames. With ``E == 1`` they reproduce the reference's
per-process buffering.

Semantics (switchable):

* ``reference_order``  — reversed discount weighting of the reference (Q1). Default: ``Σ γᵏ r_{t+k}``.
* ``flush_tail``       — at episode end also emit the last ``n−1`` truncated windows with ``done=1``
  (default). The reference drops them (``step_buffer.clear()``, actor.py:90), so only the single
  window ending on the terminal step carries ``done=1``.
"""
from __future__ import annotations

from collections import deque
from typing import List, NamedTuple, Optional

import numpy as np


def discounted(rewards, gamma: float, n: int, reference_order: bool) -> float:
    """Return of a window of ``len(rewards) ≤ n`` rewards, oldest first."""
    k = len(rewards)
    if reference_order:   # reference: oldest reward gets γ^{n-1}, newest γ^0 (actor.py:61,79)
        return float(sum(gamma ** (n - 1 - i) * float(r) for i, r in enumerate(rewards)))
    return float(sum(gamma ** i * float(r) for i, r in enumerate(rewards)))


class NStepBatch(NamedTuple):
    state: np.ndarray       # (K, …) uint8
    action: np.ndarray      # (K,) int64
    reward: np.ndarray      # (K,) float32  n-step return
    next_state: np.ndarray  # (K, …) uint8
    done: np.ndarray        # (K,) float32
    env: np.ndarray         # (K,) int64 source env id
    qs: Optional[np.ndarray] = None     # (K,) int64 position of ``state`` in its env's plane stream (vec_assembly.PlaneLog)
    qs2: Optional[np.ndarray] = None    # (K,) int64 position of ``next_state``


class NStepAssembler:
    def __init__(self, num_envs: int, nstep: int, gamma: float, reference_order: bool = False,
                 flush_tail: bool = True):
        self.E, self.n, self.gamma = num_envs, nstep, gamma
        self.reference_order = reference_order
        self.flush_tail = flush_tail
        self._buf: List[deque] = [deque(maxlen=nstep) for _ in range(num_envs)]

    def step(self, state, action, reward, next_state, done) -> Optional[NStepBatch]:
        """Feed one env step for every env (arrays with leading dim E). Returns the transitions that
        completed on this step (possibly none)."""
        out = []
        for e in range(self.E):
            buf = self._buf[e]
            buf.append((state[e], int(action[e]), float(reward[e])))
            if len(buf) == self.n:
                ret = discounted([b[2] for b in buf], self.gamma, self.n, self.reference_order)
                out.append((buf[0][0], buf[0][1], ret, next_state[e], float(done[e]), e))
            if done[e]:
                if self.flush_tail:
                    items = list(buf)
                    for j in range(1 if len(items) == self.n else 0, len(items)):
                        tail = items[j:]
                        ret = discounted([b[2] for b in tail], self.gamma, self.n, self.reference_order)
                        out.append((tail[0][0], tail[0][1], ret, next_state[e], 1.0, e))
                buf.clear()
        if not out:
            return None
        return NStepBatch(np.stack([o[0] for o in out]), np.array([o[1] for o in out], dtype=np.int64),
                          np.array([o[2] for o in out], dtype=np.float32), np.stack([o[3] for o in out]),
                          np.array([o[4] for o in out], dtype=np.float32), np.array([o[5] for o in out], dtype=np.int64))


class SequenceItem(NamedTuple):
    frames: np.ndarray     # (T_total, …) uint8, zero-padded past ``length``
    action: np.ndarray     # (T_total,) int64
    reward: np.ndarray     # (T_total,) float32 n-step returns
    done: np.ndarray       # (T_total,) float32
    length: int
    h0: np.ndarray         # (H,) recurrent state *before* frames[0]  (r2d2/actor.py:39,57)
    c0: np.ndarray
    env: int = -1          # source env of a vectorised actor
    q: Optional[np.ndarray] = None   # (T_total,) int64: positions of the observations in the env's plane stream (PlaneLog)
                                     # when the assembler keeps no frames (``frames`` is then None)


class SequenceAssembler:
    """Overlapping fixed-length sequences: ``burn_in + seq_len`` steps, consecutive sequences share the
    last ``burn_in + overlap`` steps (r2d2/actor.py:33-34,56-60); a shorter tail is emitted at episode
    end when it is longer than ``burn_in + overlap + n`` (r2d2/actor.py:61-63)."""

    def __init__(self, num_envs: int, nstep: int, gamma: float, burn_in: int = 40, seq_len: int = 80,
                 overlap: int = 40, reference_order: bool = False, flush_tail: bool = True):
        assert seq_len > 1, "n_sequence must be more than 1."
        assert overlap < seq_len, "n_overlap must be less than n_sequence."
        self.E, self.n, self.gamma = num_envs, nstep, gamma
        self.total = burn_in + seq_len
        self.keep = burn_in + overlap
        self.reference_order = reference_order
        self.flush_tail = flush_tail
        self._step: List[deque] = [deque(maxlen=nstep) for _ in range(num_envs)]
        self._seq: List[list] = [[] for _ in range(num_envs)]   # entries (frame, action, R, done, h, c)

    def _emit(self, seq) -> SequenceItem:
        f0 = seq[0][0]
        frames = np.zeros((self.total,) + f0.shape, dtype=f0.dtype)
        action = np.zeros(self.total, dtype=np.int64)
        reward = np.zeros(self.total, dtype=np.float32)
        done = np.zeros(self.total, dtype=np.float32)
        for t, (f, a, r, d, _, _) in enumerate(seq):
            frames[t], action[t], reward[t], done[t] = f, a, r, d
        return SequenceItem(frames, action, reward, done, len(seq),
                            np.asarray(seq[0][4], dtype=np.float32).reshape(-1).copy(),
                            np.asarray(seq[0][5], dtype=np.float32).reshape(-1).copy())

    def _push(self, e: int, entry, out: List[SequenceItem]) -> None:
        seq = self._seq[e]
        seq.append(entry)
        if len(seq) == self.total:
            out.append(self._emit(seq))
            self._seq[e] = seq[-self.keep:] if self.keep > 0 else []

    def step(self, state, action, reward, done, h, c) -> List[SequenceItem]:
        """``h``/``c``: (E,H) recurrent state *before* the net consumed ``state`` on this step."""
        out: List[SequenceItem] = []
        for e in range(self.E):
            sb = self._step[e]
            sb.append((state[e], int(action[e]), float(reward[e]), np.array(h[e], copy=True), np.array(c[e], copy=True)))
            if len(sb) == self.n:
                ret = discounted([b[2] for b in sb], self.gamma, self.n, self.reference_order)
                self._push(e, (sb[0][0], sb[0][1], ret, float(done[e]), sb[0][3], sb[0][4]), out)
            if done[e]:
                if self.flush_tail:
                    items = list(sb)
                    for j in range(1 if len(items) == self.n else 0, len(items)):
                        tail = items[j:]
                        ret = discounted([b[2] for b in tail], self.gamma, self.n, self.reference_order)
                        self._push(e, (tail[0][0], tail[0][1], ret, 1.0, tail[0][3], tail[0][4]), out)
                if len(self._seq[e]) > self.keep + self.n:
                    out.append(self._emit(self._seq[e]))
                sb.clear()
                self._seq[e] = []
        return out

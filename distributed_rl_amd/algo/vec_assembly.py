"""Array-based n-step / sequence assembly for vectorised actors (same semantics as algo/nstep.py, which stays as the
per-env oracle: tests/test_assembly_env.py compares the two step by step; reference: ape_x/actor.py:60-90,
r2d2/actor.py:26-75).

One actor process drives E environments. The per-env deques and Python tuples of algo/nstep.py cost O(E) interpreter work
per env step; here every env shares preallocated numpy rings and a step is a handful of fancy-indexed array operations:

* ``VecNStepAssembler``: ring ``[n, E, ...]`` of the last n (state, action, reward); the window that completes on a step
  is read at ring slot ``(t + 1) % n`` for every env whose buffer is full; episode tails are handled per finished env.
* ``VecSequenceAssembler``: the same ring feeds per-env sequence buffers ``[E, T_total, ...]``; a full buffer is emitted
  as one copy and its last ``burn_in + overlap`` entries slide to the front. The recurrent state is only needed for the
  entries that START a sequence (every ``seq_len - overlap`` steps of an episode): ``need_state()`` says for which envs
  the caller has to supply (h, c) on this step, instead of a (2, E, 512) device-to-host copy on every step.
"""
from __future__ import annotations

from typing import List, Optional

import numpy as np

from .nstep import NStepBatch, SequenceItem


def _weights(n: int, gamma: float, reference_order: bool) -> np.ndarray:
    k = np.arange(n, dtype=np.float64)
    return gamma ** (n - 1 - k) if reference_order else gamma ** k


class _StepRing:
    """Last n (state, action, reward) of every env."""

    def __init__(self, E: int, n: int):
        self.E, self.n = E, n
        self.S: Optional[np.ndarray] = None
        self.A = np.zeros((n, E), dtype=np.int64)
        self.R = np.zeros((n, E), dtype=np.float64)
        self.cnt = np.zeros(E, dtype=np.int64)
        self.t = 0

    def write(self, state, action, reward) -> int:
        if self.S is None:
            self.S = np.zeros((self.n,) + tuple(state.shape), dtype=state.dtype)
        h = self.t % self.n
        self.S[h] = state
        self.A[h] = action
        self.R[h] = reward
        np.minimum(self.cnt + 1, self.n, out=self.cnt)
        self.t += 1
        return h

    def slots_of(self, e: int, h: int) -> np.ndarray:
        """Ring slots of env ``e``'s buffered steps, oldest first."""
        L = int(self.cnt[e])
        return (h - L + 1 + np.arange(L)) % self.n


class VecNStepAssembler:
    def __init__(self, num_envs: int, nstep: int, gamma: float, reference_order: bool = False, flush_tail: bool = True):
        self.E, self.n, self.gamma = num_envs, nstep, gamma
        self.reference_order, self.flush_tail = reference_order, flush_tail
        self.ring = _StepRing(num_envs, nstep)
        self.w = _weights(nstep, gamma, reference_order)

    def step(self, state, action, reward, next_state, done) -> Optional[NStepBatch]:
        r, n = self.ring, self.n
        h = r.write(state, action, reward)
        done = np.asarray(done, dtype=bool)
        ready = np.nonzero(r.cnt == n)[0]
        old = (h + 1) % n
        parts = []
        if len(ready):
            order = (old + np.arange(n)) % n
            ret = (self.w[:, None] * r.R[order][:, ready]).sum(0)
            parts.append((r.S[old, ready], r.A[old, ready], ret, next_state[ready], done[ready].astype(np.float32), ready))
        didx = np.nonzero(done)[0]
        if len(didx):
            if self.flush_tail:                                   # the last n-1 truncated windows end the episode too
                for e in didx:
                    sl = r.slots_of(e, h)
                    L = len(sl)
                    for j in range(1 if L == n else 0, L):
                        tail = sl[j:]
                        k = np.arange(len(tail))
                        w = self.gamma ** (n - 1 - k) if self.reference_order else self.gamma ** k
                        parts.append((r.S[tail[0], e][None], r.A[tail[0], e][None], np.array([(w * r.R[tail, e]).sum()]),
                                      next_state[e][None], np.ones(1, np.float32), np.array([e])))
            r.cnt[didx] = 0
        if not parts:
            return None
        cat = (lambda i: parts[0][i]) if len(parts) == 1 else (lambda i: np.concatenate([p[i] for p in parts]))
        return NStepBatch(cat(0), cat(1).astype(np.int64), cat(2).astype(np.float32), cat(3), cat(4).astype(np.float32),
                          cat(5).astype(np.int64))


class VecSequenceAssembler:
    SLOTS = 4        # live sequence starts per env: ceil(T_total / stride) + 1 never exceeds this for overlap < seq_len/…

    def __init__(self, num_envs: int, nstep: int, gamma: float, burn_in: int = 40, seq_len: int = 80, overlap: int = 40,
                 reference_order: bool = False, flush_tail: bool = True, hidden: int = 512):
        assert seq_len > 1, "n_sequence must be more than 1."
        assert overlap < seq_len, "n_overlap must be less than n_sequence."
        self.E, self.n, self.gamma = num_envs, nstep, gamma
        self.total, self.keep = burn_in + seq_len, burn_in + overlap
        self.stride = self.total - self.keep
        self.slots = max(self.SLOTS, -(-(self.total + nstep) // self.stride) + 1)
        self.reference_order, self.flush_tail = reference_order, flush_tail
        self.ring = _StepRing(num_envs, nstep)
        self.w = _weights(nstep, gamma, reference_order)
        E, T = num_envs, self.total
        self.F: Optional[np.ndarray] = None                      # [E, T, ...] frames of the sequence being assembled
        self.A = np.zeros((E, T), dtype=np.int64)
        self.R = np.zeros((E, T), dtype=np.float32)
        self.D = np.zeros((E, T), dtype=np.float32)
        self.len = np.zeros(E, dtype=np.int64)
        self.ent0 = np.zeros(E, dtype=np.int64)                  # episode entry index of sequence position 0
        self.ep_step = np.zeros(E, dtype=np.int64)               # env steps since the episode began
        self.start_h = np.zeros((E, self.slots, hidden), dtype=np.float32)
        self.start_c = np.zeros((E, self.slots, hidden), dtype=np.float32)

    # ---- recurrent state ---------------------------------------------------------------------------------------
    def need_state(self) -> np.ndarray:
        """Envs whose NEXT ``step`` call consumes an observation that may start a sequence: the caller passes the
        recurrent state (before that observation) of exactly these envs, in env order."""
        return self.ep_step % self.stride == 0

    # ---- internals -----------------------------------------------------------------------------------------------
    def _emit(self, e: int, out: List[SequenceItem]) -> None:
        L = int(self.len[e])
        frames = np.zeros((self.total,) + self.F.shape[2:], dtype=self.F.dtype)
        frames[:L] = self.F[e, :L]
        pad = lambda a: np.concatenate([a[e, :L], np.zeros(self.total - L, a.dtype)])   # noqa: E731
        slot = (self.ent0[e] // self.stride) % self.slots
        out.append(SequenceItem(frames, pad(self.A), pad(self.R), pad(self.D), L, self.start_h[e, slot].copy(),
                                self.start_c[e, slot].copy()))

    def _slide(self, e: int) -> None:
        k = self.keep
        if k > 0:
            self.F[e, :k] = self.F[e, self.total - k:self.total]
            for a in (self.A, self.R, self.D):
                a[e, :k] = a[e, self.total - k:self.total]
        self.len[e] = k
        self.ent0[e] += self.stride

    def _push_one(self, e: int, frame, a, ret, d, out) -> None:
        p = int(self.len[e])
        self.F[e, p], self.A[e, p], self.R[e, p], self.D[e, p] = frame, a, ret, d
        self.len[e] = p + 1
        if p + 1 == self.total:
            self._emit(e, out)
            self._slide(e)

    def step(self, state, action, reward, done, h, c) -> List[SequenceItem]:
        """``h`` / ``c``: recurrent state BEFORE ``state`` was consumed - either for every env, shape (E, H), or only for
        the envs flagged by ``need_state()`` (rows in env order)."""
        r, n = self.ring, self.n
        if self.F is None:
            self.F = np.zeros((self.E, self.total) + tuple(state.shape[1:]), dtype=state.dtype)
        need = np.nonzero(self.need_state())[0]
        if len(need):
            h, c = np.asarray(h, dtype=np.float32), np.asarray(c, dtype=np.float32)
            if h.shape[0] == self.E and len(need) != self.E:
                h, c = h[need], c[need]
            slot = (self.ep_step[need] // self.stride) % self.slots
            self.start_h[need, slot] = h.reshape(len(need), -1)
            self.start_c[need, slot] = c.reshape(len(need), -1)
        hd = r.write(state, action, reward)
        done = np.asarray(done, dtype=bool)
        out: List[SequenceItem] = []
        ready = np.nonzero(r.cnt == n)[0]
        if len(ready):
            old = (hd + 1) % n
            order = (old + np.arange(n)) % n
            ret = (self.w[:, None] * r.R[order][:, ready]).sum(0)
            pos = self.len[ready]
            self.F[ready, pos] = r.S[old, ready]
            self.A[ready, pos] = r.A[old, ready]
            self.R[ready, pos] = ret
            self.D[ready, pos] = done[ready]
            self.len[ready] = pos + 1
            for e in ready[pos + 1 == self.total]:
                self._emit(e, out)
                self._slide(e)
        didx = np.nonzero(done)[0]
        for e in didx:
            if self.flush_tail:
                sl = r.slots_of(e, hd)
                L = len(sl)
                for j in range(1 if L == n else 0, L):
                    tail = sl[j:]
                    k = np.arange(len(tail))
                    w = self.gamma ** (n - 1 - k) if self.reference_order else self.gamma ** k
                    self._push_one(e, r.S[tail[0], e], r.A[tail[0], e], float((w * r.R[tail, e]).sum()), 1.0, out)
            if self.len[e] > self.keep + n:
                self._emit(e, out)                                 # the shorter tail sequence (r2d2/actor.py:61-63)
            self.len[e] = 0
            self.ent0[e] = 0
        self.ep_step += 1
        if len(didx):
            r.cnt[didx] = 0
            self.ep_step[didx] = 0
        return out

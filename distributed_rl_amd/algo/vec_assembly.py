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


def _load_native_observe():
    try:
        from .. import native
        return native.host().drl_planelog_observe
    except Exception:  # noqa: BLE001 - no compiler / library: the numpy form does the same
        return None


_native_observe = _load_native_observe()


def _weights(n: int, gamma: float, reference_order: bool) -> np.ndarray:
    k = np.arange(n, dtype=np.float64)
    return gamma ** (n - 1 - k) if reference_order else gamma ** k


class PlaneLog:
    """Every frame once: the observations of an environment are windows of a PLANE STREAM.

    A frame-stacking environment (wrapped_env.py:30-47) delivers the last ``stack`` frames at every step, so consecutive
    observations share all but one plane. Per env this log keeps the stream of distinct planes (a ring of the last ``cap``)
    and the position ``q`` of the current observation in it: observation = planes ``q .. q + stack - 1``. A step that
    shifts the stack by one appends ONE plane (q += 1); anything else - the first observation, a reset (4 copies of the
    first frame + no-ops), an env that does not stack at all - appends all ``stack`` planes (q jumps to them). Whether a
    step shifted is CHECKED on the pixels (the planes that should be shared are compared), never assumed, so any env is
    stored correctly; one that does not shift just does not compress. The replay stores transitions as two such positions
    (replay/device_per.py), the wire block carries the planes they span once (transport/codec.py kind 3)."""

    def __init__(self, num_envs: int, cap: int = 512, stack: int = 4, native: bool = True):
        self.E, self.cap, self.stack = num_envs, cap, stack
        self.native = bool(native) and _native_observe is not None
        self.P: Optional[np.ndarray] = None                   # [E, cap, H, W]
        self.head = np.zeros(num_envs, dtype=np.int64)        # planes appended so far, per env
        self.q = np.zeros(num_envs, dtype=np.int64)           # stream position of the last observation seen
        self._last: Optional[np.ndarray] = None               # that observation (HWC), to test the next one against
        self._ar = np.arange(num_envs)
        self._tmp: Optional[np.ndarray] = None                # scratch of _relation

    def _append(self, idx: np.ndarray, planes: np.ndarray) -> None:
        """planes [len(idx), H, W] -> position head[idx] of each env's ring."""
        self.P[idx, self.head[idx] % self.cap] = planes
        self.head[idx] += 1

    def restart(self, obs: np.ndarray, rows: Optional[np.ndarray] = None) -> None:
        """The observation of envs ``rows`` (default: all) is unrelated to what they showed before (first observation,
        reset): all its planes enter the stream."""
        if self.P is None:
            self.P = np.zeros((self.E, self.cap) + tuple(obs.shape[1:3]), dtype=obs.dtype)
            self._last = np.zeros_like(obs)
        rows = self._ar if rows is None else np.asarray(rows)
        if len(rows) == 0:
            return
        self.q[rows] = self.head[rows]
        for c in range(self.stack):
            self._append(rows, obs[rows, :, :, c])
        self._last[rows] = obs[rows]

    def _relation(self, obs: np.ndarray):
        """Per env: (obs is the last observation again, obs is the last observation moved on by one frame)."""
        E, k = self.E, self.stack
        if k == 4 and obs.dtype == np.uint8 and obs.flags.c_contiguous and (obs[0].size & 7) == 0:
            # one little-endian word per pixel (byte c = plane c), two pixels per uint64: "moved on by one frame" means
            # planes 0..2 of obs = planes 1..3 of the last one, i.e. ((last >> 8) ^ obs) has nothing in the low 3 bytes of
            # either word (the byte the shift drags across the word boundary lands in a masked one). Three in-place passes
            # over a scratch array for all envs at once: 0.4 ms for 64 envs where fancy-indexed temporaries took 2.2 ms.
            a = obs.reshape(E, -1).view("<u8")
            b = self._last.reshape(E, -1).view("<u8")
            same = (a == b).all(1)
            if same.all():
                return same, np.zeros(E, dtype=bool)
            if self._tmp is None or self._tmp.shape != a.shape:
                self._tmp = np.empty_like(a)
            t = self._tmp
            np.right_shift(b, 8, out=t)
            np.bitwise_xor(t, a, out=t)
            np.bitwise_and(t, 0x00FFFFFF00FFFFFF, out=t)
            return same, ~t.any(1) & ~same
        a, b = obs.reshape(E, -1, k), self._last.reshape(E, -1, k)
        same = (a == b).all((1, 2))
        return same, ~same & (a[:, :, :k - 1] == b[:, :, 1:]).all((1, 2))

    def observe(self, obs: np.ndarray) -> np.ndarray:
        """Register the observation [E, H, W, stack] every env shows now - the one seen last again (nothing happens), that
        one moved on by a frame (one new plane), or something else (all its planes) - and return its stream positions."""
        if self.P is None:
            self.restart(obs)
            return self.q.copy()
        if self.native and self.stack == 4 and obs.dtype == np.uint8 and obs.flags.c_contiguous and obs.shape == self._last.shape:
            # one native pass per env over the two stacks: relation, the new plane(s) into the ring, the copy of the stack
            # (csrc/host/drl_host.cpp drl_planelog_observe; the numpy form below is its oracle, tests/test_planes_cpu.py)
            rc = _native_observe(obs.ctypes.data, self._last.ctypes.data, self.P.ctypes.data, self.head.ctypes.data,
                                 self.q.ctypes.data, self.E, obs.shape[1] * obs.shape[2], self.cap, None)
            if rc < 0:
                raise RuntimeError("drl_planelog_observe failed")
            return self.q.copy()
        same, shift = self._relation(obs)
        i = np.nonzero(shift)[0]
        if len(i):
            self._append(i, obs[i, :, :, self.stack - 1])
            self.q[i] += 1
            self._last[i] = obs[i]
        self.restart(obs, np.nonzero(~same & ~shift)[0])
        return self.q.copy()

    def planes(self, e: int, q0: int, q1: int) -> np.ndarray:
        """Planes q0 .. q1 (inclusive) of env e."""
        if self.head[e] - q0 > self.cap or q1 >= self.head[e]:
            raise IndexError("plane %d..%d of env %d is no longer / not yet in the log (head %d, cap %d)" %
                             (q0, q1, e, self.head[e], self.cap))
        return self.P[e, np.arange(q0, q1 + 1) % self.cap]


def sequence_plane_block(log: PlaneLog, e: int, seqs):
    """(planes [M,H,W], win [K,T] int32) of sequences of ONE env assembled without frames (``SequenceItem.q``): the planes
    they span once - overlapping sequences share theirs - and, per step, the pool index of the first plane of its window."""
    lo = min(int(s.q[:s.length].min()) for s in seqs)
    hi = max(int(s.q[:s.length].max()) for s in seqs) + log.stack - 1
    win = np.stack([np.clip(s.q - lo, 0, hi - lo - log.stack + 1) for s in seqs]).astype(np.int32)
    return log.planes(e, lo, hi), win


def stacks_from_planes(planes: np.ndarray, win: np.ndarray, stack: int = 4) -> np.ndarray:
    """HWC observation stacks [..., H, W, stack] of the windows ``win`` (any shape) over a plane pool [M, H, W]."""
    idx = np.asarray(win)[..., None] + np.arange(stack)
    return np.ascontiguousarray(np.moveaxis(planes[idx], -3, -1))


def plane_block(log: PlaneLog, env: np.ndarray, qs: np.ndarray, qs2: np.ndarray):
    """(planes [M,H,W], win_s [K], win_s2 [K]) of the transitions (env, qs, qs2): per env the planes its transitions span,
    once; win = index of a window's first plane in the pool."""
    pools, base = [], 0
    win_s = np.zeros(len(env), dtype=np.int32)
    win_s2 = np.zeros(len(env), dtype=np.int32)
    for e in np.unique(env):
        m = env == e
        lo = int(min(qs[m].min(), qs2[m].min()))
        hi = int(max(qs[m].max(), qs2[m].max())) + log.stack - 1
        pools.append(log.planes(int(e), lo, hi))
        win_s[m] = base + (qs[m] - lo)
        win_s2[m] = base + (qs2[m] - lo)
        base += hi - lo + 1
    return np.concatenate(pools), win_s, win_s2


class _StepRing:
    """Last n (state, action, reward) of every env (+ the state's position in the env's plane stream, when given)."""

    def __init__(self, E: int, n: int):
        self.E, self.n = E, n
        self.S: Optional[np.ndarray] = None
        self.A = np.zeros((n, E), dtype=np.int64)
        self.R = np.zeros((n, E), dtype=np.float64)
        self.Q = np.zeros((n, E), dtype=np.int64)
        self.cnt = np.zeros(E, dtype=np.int64)
        self.t = 0

    def write(self, state, action, reward, q=None) -> int:
        h = self.t % self.n
        if state is not None:                                 # None: the caller keeps the frames elsewhere (PlaneLog)
            if self.S is None:
                self.S = np.zeros((self.n,) + tuple(state.shape), dtype=state.dtype)
            self.S[h] = state
        self.A[h] = action
        self.R[h] = reward
        if q is not None:
            self.Q[h] = q
        np.minimum(self.cnt + 1, self.n, out=self.cnt)
        self.t += 1
        return h

    def slots_of(self, e: int, h: int) -> np.ndarray:
        """Ring slots of env ``e``'s buffered steps, oldest first."""
        L = int(self.cnt[e])
        return (h - L + 1 + np.arange(L)) % self.n


class VecNStepAssembler:
    def __init__(self, num_envs: int, nstep: int, gamma: float, reference_order: bool = False, flush_tail: bool = True,
                 store_states: bool = True):
        """store_states=False: the caller keeps every frame in a PlaneLog and passes stream positions (q_state / q_next) -
        the emitted batches carry positions only (state / next_state None) and no observation is copied here (three
        1.8 MB copies per step of 64 envs otherwise)."""
        self.E, self.n, self.gamma = num_envs, nstep, gamma
        self.reference_order, self.flush_tail = reference_order, flush_tail
        self.ring = _StepRing(num_envs, nstep)
        self.w = _weights(nstep, gamma, reference_order)
        self.store_states = store_states

    def step(self, state, action, reward, next_state, done, q_state=None, q_next=None) -> Optional[NStepBatch]:
        """q_state / q_next (optional, [E] int64): positions of ``state`` / ``next_state`` in the envs' plane streams
        (PlaneLog.observe); the emitted batch then carries the positions of its states (``qs``, ``qs2``)."""
        r, n = self.ring, self.n
        keep = self.store_states
        assert keep or q_state is not None, "store_states=False needs the plane-stream positions"
        h = r.write(state if keep else None, action, reward, q_state)
        done = np.asarray(done, dtype=bool)
        ready = np.nonzero(r.cnt == n)[0]
        old = (h + 1) % n
        parts = []
        qn = q_next if q_next is not None else np.zeros(self.E, dtype=np.int64)
        if len(ready):
            order = (old + np.arange(n)) % n
            ret = (self.w[:, None] * r.R[order][:, ready]).sum(0)
            parts.append((r.S[old, ready] if keep else None, r.A[old, ready], ret, next_state[ready] if keep else None,
                          done[ready].astype(np.float32), ready, r.Q[old, ready], qn[ready]))
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
                        parts.append((r.S[tail[0], e][None] if keep else None, r.A[tail[0], e][None],
                                      np.array([(w * r.R[tail, e]).sum()]), next_state[e][None] if keep else None,
                                      np.ones(1, np.float32), np.array([e]), r.Q[tail[0], e][None], qn[e][None]))
            r.cnt[didx] = 0
        if not parts:
            return None
        cat = (lambda i: parts[0][i]) if len(parts) == 1 else (lambda i: np.concatenate([p[i] for p in parts]))
        pos = (cat(6).astype(np.int64), cat(7).astype(np.int64)) if q_state is not None else (None, None)
        return NStepBatch(cat(0) if keep else None, cat(1).astype(np.int64), cat(2).astype(np.float32),
                          cat(3) if keep else None, cat(4).astype(np.float32), cat(5).astype(np.int64), *pos)


class VecSequenceAssembler:
    SLOTS = 4        # live sequence starts per env: ceil(T_total / stride) + 1 never exceeds this for overlap < seq_len/…

    def __init__(self, num_envs: int, nstep: int, gamma: float, burn_in: int = 40, seq_len: int = 80, overlap: int = 40,
                 reference_order: bool = False, flush_tail: bool = True, hidden: int = 512, store_states: bool = True):
        """store_states=False: the caller keeps every frame in a PlaneLog and passes the observations' stream positions
        (``q_state``); sequences then carry positions (``SequenceItem.q``, ``.env``) and no frames - nothing is copied here
        (a 28 KB stack per env and step into the sequence buffers, and 3.4 MB per emitted sequence, otherwise)."""
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
        self.store_states = store_states
        self.F: Optional[np.ndarray] = None                      # [E, T, ...] frames of the sequence being assembled
        self.Q = np.zeros((E, T), dtype=np.int64)                # ... or their positions in the env's plane stream
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
        frames = q = None
        if self.store_states:
            frames = np.zeros((self.total,) + self.F.shape[2:], dtype=self.F.dtype)
            frames[:L] = self.F[e, :L]
        else:                                                     # padding steps point at the last real observation
            q = np.concatenate([self.Q[e, :L], np.full(self.total - L, self.Q[e, L - 1], np.int64)])
        pad = lambda a: np.concatenate([a[e, :L], np.zeros(self.total - L, a.dtype)])   # noqa: E731
        slot = (self.ent0[e] // self.stride) % self.slots
        out.append(SequenceItem(frames, pad(self.A), pad(self.R), pad(self.D), L, self.start_h[e, slot].copy(),
                                self.start_c[e, slot].copy(), int(e), q))

    def _slide(self, e: int) -> None:
        k = self.keep
        if k > 0:
            if self.store_states:
                self.F[e, :k] = self.F[e, self.total - k:self.total]
            for a in (self.A, self.R, self.D, self.Q):
                a[e, :k] = a[e, self.total - k:self.total]
        self.len[e] = k
        self.ent0[e] += self.stride

    def _push_one(self, e: int, frame, a, ret, d, out, q=0) -> None:
        p = int(self.len[e])
        if self.store_states:
            self.F[e, p] = frame
        self.Q[e, p], self.A[e, p], self.R[e, p], self.D[e, p] = q, a, ret, d
        self.len[e] = p + 1
        if p + 1 == self.total:
            self._emit(e, out)
            self._slide(e)

    def step(self, state, action, reward, done, h, c, q_state=None) -> List[SequenceItem]:
        """``h`` / ``c``: recurrent state BEFORE ``state`` was consumed - either for every env, shape (E, H), or only for
        the envs flagged by ``need_state()`` (rows in env order). ``q_state`` ([E] int64): positions of ``state`` in the
        envs' plane streams (PlaneLog.observe) - required with store_states=False."""
        r, n = self.ring, self.n
        keep = self.store_states
        assert keep or q_state is not None, "store_states=False needs the plane-stream positions"
        if keep and self.F is None:
            self.F = np.zeros((self.E, self.total) + tuple(state.shape[1:]), dtype=state.dtype)
        need = np.nonzero(self.need_state())[0]
        if len(need):
            h, c = np.asarray(h, dtype=np.float32), np.asarray(c, dtype=np.float32)
            if h.shape[0] == self.E and len(need) != self.E:
                h, c = h[need], c[need]
            slot = (self.ep_step[need] // self.stride) % self.slots
            self.start_h[need, slot] = h.reshape(len(need), -1)
            self.start_c[need, slot] = c.reshape(len(need), -1)
        hd = r.write(state if keep else None, action, reward, q_state)
        done = np.asarray(done, dtype=bool)
        out: List[SequenceItem] = []
        ready = np.nonzero(r.cnt == n)[0]
        if len(ready):
            old = (hd + 1) % n
            order = (old + np.arange(n)) % n
            ret = (self.w[:, None] * r.R[order][:, ready]).sum(0)
            pos = self.len[ready]
            if keep:
                self.F[ready, pos] = r.S[old, ready]
            self.Q[ready, pos] = r.Q[old, ready]
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
                    self._push_one(e, r.S[tail[0], e] if keep else None, r.A[tail[0], e], float((w * r.R[tail, e]).sum()), 1.0,
                                   out, r.Q[tail[0], e])
            if self.len[e] > self.keep + n:
                self._emit(e, out)                                 # the shorter tail sequence (r2d2/actor.py:61-63)
            self.len[e] = 0
            self.ent0[e] = 0
        self.ep_step += 1
        if len(didx):
            r.cnt[didx] = 0
            self.ep_step[didx] = 0
        return out

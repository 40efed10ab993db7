"""Synthetic Atari-shaped environments (no ROM, no gym): uint8 84×84 frames, 4-frame stack.

``CatchVecEnv`` is a vectorised "catch the ball" game rendered into 84×84 uint8 frames with the
same observation contract as the reference's ``MultiFrameAtariEnv`` (wrapped_env.py:7-47):
frameskip is folded into the dynamics, the observation is the stack of the last 4 frames, reset
fills the stack with 4 copies and plays a random number of no-ops, and episodes are capped at
``max_episode_steps`` (wrapped_env.py:66). It has a learnable signal (reward +1 catch / −1 miss),
so end-to-end tests can check that the loss actually goes down, and it is cheap enough that actor
throughput measures inference + transport rather than Python.

Observation layout is **HWC with the stack as channels**: ``(E, 84, 84, 4)`` uint8 — the layout the
replay store and the conv1 implicit-GEMM kernel consume (32 contiguous bytes per kernel row).
"""
from __future__ import annotations

from typing import Optional, Tuple

import numpy as np


class CatchVecEnv:
    H = W = 84
    BALL = 4
    PAD_W, PAD_H = 14, 3

    def __init__(self, num_envs: int = 1, n_action: int = 4, stack: int = 4, max_episode_steps: int = 10000,
                 lives: int = 5, seed: Optional[int] = None, noop_max: int = 30, frameskip: int = 4):
        self.E, self.n_action, self.stack = num_envs, n_action, stack
        self.max_episode_steps, self.lives0 = max_episode_steps, lives
        self.noop_max, self.frameskip = noop_max, frameskip
        self.rng = np.random.default_rng(seed)
        self.obs = np.zeros((num_envs, self.H, self.W, stack), dtype=np.uint8)
        self.ball_x = np.zeros(num_envs, dtype=np.int64)
        self.ball_y = np.zeros(num_envs, dtype=np.int64)
        self.ball_dx = np.zeros(num_envs, dtype=np.int64)
        self.pad_x = np.zeros(num_envs, dtype=np.int64)
        self.lives = np.zeros(num_envs, dtype=np.int64)
        self.t = np.zeros(num_envs, dtype=np.int64)
        self.score = np.zeros(num_envs, dtype=np.float64)
        self._ar = np.arange(num_envs)
        self.reset()

    # ---- helpers ----------------------------------------------------------------------------------
    @property
    def observation_shape(self) -> Tuple[int, int, int]:
        return (self.H, self.W, self.stack)

    def _new_ball(self, idx) -> None:
        n = len(idx)
        self.ball_x[idx] = self.rng.integers(0, self.W - self.BALL, n)
        self.ball_y[idx] = 0
        self.ball_dx[idx] = self.rng.integers(-2, 3, n)

    def _render(self, idx) -> np.ndarray:
        n = len(idx)
        fr = np.zeros((n, self.H, self.W), dtype=np.uint8)
        fr[:, 0:2, :] = 60                                   # a static top bar: non-trivial background
        by, bx = self.ball_y[idx], self.ball_x[idx]
        ar = np.arange(n)[:, None, None]
        yy = (by[:, None, None] + np.arange(self.BALL)[None, :, None]).clip(0, self.H - 1)
        xx = (bx[:, None, None] + np.arange(self.BALL)[None, None, :]).clip(0, self.W - 1)
        fr[ar, yy, xx] = 255
        py = (self.H - self.PAD_H + np.arange(self.PAD_H))[None, :, None]
        px = (self.pad_x[idx][:, None, None] + np.arange(self.PAD_W)[None, None, :]).clip(0, self.W - 1)
        fr[ar, py, px] = 180
        return fr

    def _push_frame(self, idx, fr) -> None:
        if idx is self._ar and self.stack == 4:
            # all envs, 4-stack: a pixel's stack is one little-endian uint32 - dropping the oldest frame and appending the
            # new one is a shift and an or over E x 7056 words (the strided byte copy below costs 10x as much)
            o32 = self.obs.view(np.uint32)[..., 0]
            np.right_shift(o32, 8, out=o32)
            np.bitwise_or(o32, fr.astype(np.uint32) << 24, out=o32)
            return
        self.obs[idx, :, :, :-1] = self.obs[idx, :, :, 1:]
        self.obs[idx, :, :, -1] = fr

    def _reset_envs(self, idx) -> None:
        if len(idx) == 0:
            return
        self.pad_x[idx] = (self.W - self.PAD_W) // 2
        self.lives[idx] = self.lives0
        self.t[idx] = 0
        self.score[idx] = 0
        self._new_ball(idx)
        fr = self._render(idx)
        self.obs[idx] = fr[..., None]                        # 4 copies (wrapped_env.py:32-33)
        noops = self.rng.integers(1, self.noop_max, len(idx)) // self.frameskip   # wrapped_env.py:34
        for k in range(int(noops.max()) if len(noops) else 0):
            sel = idx[noops > k]
            if len(sel):
                self._advance(sel, np.zeros(len(sel), dtype=np.int64))

    def _advance(self, idx, action):
        """One (frame-skipped) game tick for envs ``idx``. Returns (reward, life_lost_all)."""
        move = np.where(action == 2, 4, np.where(action == 3, -4, 0))
        self.pad_x[idx] = np.clip(self.pad_x[idx] + move, 0, self.W - self.PAD_W)
        self.ball_y[idx] += 3
        bx = self.ball_x[idx] + self.ball_dx[idx]
        bounce = (bx < 0) | (bx > self.W - self.BALL)
        self.ball_dx[idx] = np.where(bounce, -self.ball_dx[idx], self.ball_dx[idx])
        self.ball_x[idx] = np.clip(bx, 0, self.W - self.BALL)
        at_bottom = self.ball_y[idx] >= self.H - self.PAD_H - self.BALL
        caught = at_bottom & (self.ball_x[idx] + self.BALL > self.pad_x[idx]) & (self.ball_x[idx] < self.pad_x[idx] + self.PAD_W)
        missed = at_bottom & ~caught
        reward = caught.astype(np.float64) - missed.astype(np.float64)
        self.lives[idx] -= missed
        nb = idx[at_bottom]
        if len(nb):
            self._new_ball(nb)
        self._push_frame(idx, self._render(idx))
        return reward

    # ---- gym-like vector API ----------------------------------------------------------------------
    def reset(self) -> np.ndarray:
        self._reset_envs(self._ar)
        return self.obs.copy()

    def step(self, action: np.ndarray):
        """Returns (obs, reward, done, info). Finished envs are reset in place and ``obs`` already holds
        the first observation of the next episode; ``info['terminal_obs']`` has the pre-reset frames
        for the envs in ``info['done_idx']``."""
        action = np.asarray(action, dtype=np.int64).reshape(self.E)
        reward = self._advance(self._ar, action)
        self.t += 1
        self.score += reward
        done = (self.lives <= 0) | (self.t >= self.max_episode_steps)
        info = {}
        didx = np.nonzero(done)[0]
        if len(didx):
            info["done_idx"] = didx
            info["terminal_obs"] = self.obs[didx].copy()
            info["episode_return"] = self.score[didx].copy()
            info["episode_length"] = self.t[didx].copy()
            next_obs = self.obs.copy()
            self._reset_envs(didx)
            info["reset_obs"] = self.obs[didx].copy()
            # caller sees terminal frames as next_state of the final transition
            return next_obs, reward.astype(np.float32), done, info
        return self.obs.copy(), reward.astype(np.float32), done, info

    def current_obs(self) -> np.ndarray:
        return self.obs.copy()


class RandomFrameVecEnv:
    """Throughput-only env: random uint8 frames, random rewards, fixed episode length. Used by the
    actor-inference benchmark so that frames/s measures the GPU path, not game logic."""

    def __init__(self, num_envs: int, n_action: int = 4, episode_len: int = 1000, seed: Optional[int] = None,
                 pool: int = 64):
        self.E, self.n_action, self.episode_len = num_envs, n_action, episode_len
        self.rng = np.random.default_rng(seed)
        self._pool = self.rng.integers(0, 256, (pool, 84, 84, 4), dtype=np.uint8)
        self.t = np.zeros(num_envs, dtype=np.int64)
        self._k = 0
        self.observation_shape = (84, 84, 4)

    def _obs(self):
        ix = (np.arange(self.E) + self._k) % len(self._pool)
        self._k += 1
        return self._pool[ix]

    def reset(self):
        self.t[:] = 0
        return self._obs()

    def current_obs(self):
        return self._obs()

    def step(self, action):
        self.t += 1
        done = self.t >= self.episode_len
        self.t[done] = 0
        reward = (self.rng.random(self.E) < 0.05).astype(np.float32)
        info = {}
        if done.any():
            info["done_idx"] = np.nonzero(done)[0]
        return self._obs(), reward, done, info

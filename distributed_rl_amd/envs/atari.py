"""Optional gym/ALE adapter with the reference's ``MultiFrameAtariEnv`` semantics
(libs/wrapped_env.py:7-47): frameskip 4, sticky-action prob 0, gray 84×84, 4-frame stack,
max over the last two returned frames, reset = 4 copies + ``randint(1,30)//4`` no-ops, episode cap.

Only imported when gym + an Atari backend are installed (they are not in the MI355X image), so
this module must never be imported unconditionally. tests/test_plugins_fakes.py pins it to the reference wrapper's
arithmetic over a scripted in-tree AtariEnv (parity with the real emulator unpinned). Frames stay uint8 (the reference divides by
255 on the host, utils.py:34; here that is fused into the conv1 kernel).
"""
from __future__ import annotations

from typing import Optional

import numpy as np

from ..utils import preprocess


class AtariVecEnv:
    no_op_steps = 30

    def __init__(self, game: str, num_envs: int = 1, seed: Optional[int] = None, frameskip: int = 4,
                 max_episode_steps: int = 10000, stack: int = 4):
        from gym.envs.atari.atari_env import AtariEnv  # type: ignore
        self.E, self.stack, self.frameskip = num_envs, stack, frameskip
        self.max_episode_steps = max_episode_steps
        self.envs = [AtariEnv(game=game, obs_type='image', frameskip=frameskip, repeat_action_probability=0.)
                     for _ in range(num_envs)]
        self.n_action = self.envs[0].action_space.n
        self.rng = np.random.default_rng(seed)
        if seed is not None:
            for i, e in enumerate(self.envs):
                e.seed(seed + i)
        self.obs = np.zeros((num_envs, 84, 84, stack), dtype=np.uint8)
        self._last_raw = [None] * num_envs
        self.t = np.zeros(num_envs, dtype=np.int64)
        self.score = np.zeros(num_envs, dtype=np.float64)
        self.observation_shape = (84, 84, stack)
        self.reset()

    def _gray(self, rgb):
        return preprocess(rgb, (84, 84), gray=True, dtype=np.uint8)

    def _reset_one(self, i):
        raw = self.envs[i].reset()
        self._last_raw[i] = raw
        self.obs[i] = self._gray(raw)[..., None]
        self.t[i] = 0
        self.score[i] = 0
        for _ in range(int(self.rng.integers(1, self.no_op_steps)) // self.frameskip):
            self._step_one(i, 0)

    def _step_one(self, i, a):
        prev = self._last_raw[i]
        raw, r, d, _ = self.envs[i].step(int(a))
        self._last_raw[i] = raw
        fr = self._gray(np.maximum(raw, prev))             # wrapped_env.py:40
        self.obs[i, :, :, :-1] = self.obs[i, :, :, 1:]
        self.obs[i, :, :, -1] = fr
        return r, d

    def reset(self):
        for i in range(self.E):
            self._reset_one(i)
        return self.obs.copy()

    def current_obs(self):
        return self.obs.copy()

    def render_rgb(self, i: int = 0) -> np.ndarray:
        """The emulator's RGB screen of env ``i`` as CHW uint8 - what the reference's actors send to their visdom
        window every step (``env.env._get_image()``, ape_x/actor.py:46,82)."""
        return np.ascontiguousarray(np.asarray(self.envs[i]._get_image(), dtype=np.uint8).transpose(2, 0, 1))

    def step(self, action):
        reward = np.zeros(self.E, dtype=np.float32)
        done = np.zeros(self.E, dtype=bool)
        for i in range(self.E):
            r, d = self._step_one(i, action[i])
            self.t[i] += 1
            self.score[i] += r
            reward[i] = r
            done[i] = d or self.t[i] >= self.max_episode_steps
        out = self.obs.copy()
        info = {}
        didx = np.nonzero(done)[0]
        if len(didx):
            info["done_idx"] = didx
            info["episode_return"] = self.score[didx].copy()
            info["episode_length"] = self.t[didx].copy()
            for i in didx:
                self._reset_one(i)
        return out, reward, done, info

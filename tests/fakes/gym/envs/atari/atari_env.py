"""Scripted stand-in for ``gym.envs.atari.atari_env.AtariEnv`` (gym 0.10 API, the one the reference was written against:
libs/wrapped_env.py:7-14): ``reset() -> rgb``, ``step(a) -> (rgb, reward, done, info)``, ``seed``, ``action_space.n``.
Frame k of an episode is a deterministic function of (game, episode number, k, actions so far), every second frame has
a bright sprite the next one lacks (so the max over two consecutive raw frames is observable), reward and episode end
follow a fixed script."""
import zlib

import numpy as np

N_ACTION = {"breakout": 4, "pong": 6}


class _Discrete:
    def __init__(self, n):
        self.n = n


class AtariEnv:
    EPISODE_FRAMES = 37                      # raw frames (agent steps) per scripted episode

    def __init__(self, game="pong", obs_type="image", frameskip=4, repeat_action_probability=0.0):
        assert obs_type == "image"
        self.game, self.frameskip, self.repeat_action_probability = game, frameskip, repeat_action_probability
        self.action_space = _Discrete(N_ACTION.get(game, 18))
        self._seed, self._episode, self._k, self._acts = 0, -1, 0, 0
        self.calls = []                      # ("reset",) / ("step", a): lets tests check no-op counts and action passing

    def seed(self, s=None):
        self._seed = int(s or 0)
        return [self._seed]

    def _frame(self):
        h = zlib.crc32(("%s/%d/%d/%d/%d" % (self.game, self._seed, self._episode, self._k, self._acts)).encode())
        rng = np.random.RandomState(h & 0x7fffffff)
        img = rng.randint(0, 120, (210, 160, 3)).astype(np.uint8)
        if self._k % 2 == 0:                 # flickering sprite: present on even frames only
            y, x = 20 + (self._k * 7) % 150, 10 + (self._k * 11) % 120
            img[y:y + 24, x:x + 24] = 255
        return img

    def reset(self):
        self.calls.append(("reset",))
        self._episode += 1
        self._k, self._acts = 0, 0
        return self._frame()

    def step(self, a):
        self.calls.append(("step", int(a)))
        self._k += 1
        self._acts = (self._acts * 31 + int(a) + 1) % 1000003
        reward = float(self._k % 5 == 0) * (2.0 if a else 1.0)
        done = self._k >= self.EPISODE_FRAMES
        return self._frame(), reward, done, {"ale.lives": 3}

    def _get_image(self):
        return self._frame()

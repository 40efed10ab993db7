"""Environment registry (reference: libs/wrapped_env.py:49-76).

``make("MultiFrameBreakout-v0")`` works without gym: every ``MultiFrame<Game>-v0`` /
``SingleFrame<Game>-v0`` id of the reference's 60-game list is registered here. When gym +
atari-py/ale-py are importable the real ALE adapter (:mod:`.atari`) is used; otherwise the id
resolves to the synthetic Atari-shaped env with that game's action count, so the whole
actor/learner pipeline runs anywhere (neither gym nor ALE exist in the MI355X image).
"""
from __future__ import annotations

import warnings
from typing import Optional

import numpy as np

from .synthetic import CatchVecEnv, RandomFrameVecEnv

GAMES = ['air_raid', 'alien', 'amidar', 'assault', 'asterix', 'asteroids', 'atlantis',
         'bank_heist', 'battle_zone', 'beam_rider', 'berzerk', 'bowling', 'boxing', 'breakout', 'carnival',
         'centipede', 'chopper_command', 'crazy_climber', 'demon_attack', 'double_dunk',
         'elevator_action', 'enduro', 'fishing_derby', 'freeway', 'frostbite', 'gopher', 'gravitar',
         'hero', 'ice_hockey', 'jamesbond', 'journey_escape', 'kangaroo', 'krull', 'kung_fu_master',
         'montezuma_revenge', 'ms_pacman', 'name_this_game', 'phoenix', 'pitfall', 'pong', 'pooyan',
         'private_eye', 'qbert', 'riverraid', 'road_runner', 'robotank', 'seaquest', 'skiing',
         'solaris', 'space_invaders', 'star_gunner', 'tennis', 'time_pilot', 'tutankham', 'up_n_down',
         'venture', 'video_pinball', 'wizard_of_wor', 'yars_revenge', 'zaxxon']

#: minimal action-set sizes of ALE games (18 when not listed)
_N_ACTION = {'breakout': 4, 'pong': 6, 'space_invaders': 6, 'qbert': 6, 'freeway': 3, 'skiing': 3,
             'ms_pacman': 9, 'beam_rider': 9, 'enduro': 9, 'asterix': 9, 'atlantis': 4, 'assault': 7,
             'air_raid': 6, 'amidar': 10, 'bowling': 6, 'carnival': 6, 'demon_attack': 6, 'gopher': 8,
             'kung_fu_master': 14, 'name_this_game': 6, 'phoenix': 8, 'pooyan': 6, 'tutankham': 8,
             'up_n_down': 6, 'video_pinball': 9, 'wizard_of_wor': 10, 'journey_escape': 16,
             'asteroids': 14, 'crazy_climber': 9, 'star_gunner': 18, 'time_pilot': 10, 'krull': 18}


def camel(game: str) -> str:
    return ''.join(g.capitalize() for g in game.split('_'))


REGISTRY = {}
for _g in GAMES:
    REGISTRY['MultiFrame%s-v0' % camel(_g)] = dict(game=_g, buf_size=4, gray=True)
    # the reference passes the CamelCase name as `game` here (wrapped_env.py:73, latent bug); fixed
    REGISTRY['SingleFrame%s-v0' % camel(_g)] = dict(game=_g, buf_size=1, gray=False)
REGISTRY['SyntheticCatch-v0'] = dict(game='synthetic_catch', buf_size=4, gray=True)
REGISTRY['SyntheticRandom-v0'] = dict(game='synthetic_random', buf_size=4, gray=True)


def n_action_of(env_id: str) -> int:
    spec = REGISTRY[env_id]
    return _N_ACTION.get(spec['game'], 4 if spec['game'].startswith('synthetic') else 18)


def _ale_available() -> bool:
    try:
        import gym  # noqa: F401
        from gym.envs.atari.atari_env import AtariEnv  # noqa: F401
        return True
    except Exception:
        return False


def make_vec(env_id: str, num_envs: int = 1, seed: Optional[int] = None, backend: str = "auto",
             max_episode_steps: int = 10000):
    """Vectorised env with uint8 HWC-stacked observations ``(E,84,84,4)``."""
    if env_id not in REGISTRY:
        raise KeyError("unknown environment id %r" % env_id)
    spec = REGISTRY[env_id]
    if spec['game'] == 'synthetic_random':
        return RandomFrameVecEnv(num_envs, seed=seed)
    if backend in ("auto", "ale") and not spec['game'].startswith('synthetic') and _ale_available():
        from .atari import AtariVecEnv
        return AtariVecEnv(spec['game'], num_envs, seed=seed, max_episode_steps=max_episode_steps)
    if backend == "ale":
        raise RuntimeError("gym/ALE not importable; cannot create a real %s" % env_id)
    if not spec['game'].startswith('synthetic'):
        warnings.warn("%s: gym/ALE not installed, using the synthetic Atari-shaped env" % env_id, stacklevel=2)
    return CatchVecEnv(num_envs, n_action=n_action_of(env_id), stack=spec['buf_size'] if spec['gray'] else 4,
                       seed=seed, max_episode_steps=max_episode_steps)


class _Discrete:
    def __init__(self, n):
        self.n = n

    def sample(self):
        return int(np.random.randint(self.n))


class SingleEnv:
    """gym-style single-env facade with the reference's observation contract: ``reset() -> obs``,
    ``step(a) -> (obs, reward, done, info)``, obs float32 ``(4,84,84)`` in [0,1] (wrapped_env.py:37-47)
    or raw uint8 HWC when ``raw=True``."""

    def __init__(self, vec, raw: bool = False):
        assert vec.E == 1
        self.vec, self.raw = vec, raw
        self.action_space = _Discrete(vec.n_action)

    def _fmt(self, o):
        if self.raw:
            return o
        return np.ascontiguousarray(np.moveaxis(o, -1, 0), dtype=np.float32) / 255

    def reset(self):
        self.vec.reset()
        return self._fmt(self.vec.current_obs()[0])

    def step(self, a):
        obs, r, d, info = self.vec.step(np.array([a]))
        return self._fmt(obs[0]), float(r[0]), bool(d[0]), info

    def render_rgb(self):
        g = self.vec.current_obs()[0][..., -1]
        return np.repeat(g[None], 3, 0)


def make(env_id: str, seed: Optional[int] = None, raw: bool = False, **kw) -> SingleEnv:
    """``gym.make`` replacement (actor_node.py:18, learner_node.py:20)."""
    return SingleEnv(make_vec(env_id, 1, seed=seed, **kw), raw=raw)

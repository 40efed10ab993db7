"""Fake ``gym`` package (tests only): just enough for ``gym.envs.atari.atari_env.AtariEnv`` (see that module)."""
__fake__ = True
__version__ = "0.10.fake"

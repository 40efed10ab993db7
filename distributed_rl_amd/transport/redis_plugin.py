"""Optional Redis transport with the reference's key names and commands.

Only usable when the ``redis`` Python module and a server are present (neither is in the MI355X
image); kept so multi-node CPU actor fleets can still feed a learner over TCP exactly like the
reference deployment (README.md:14-19). Payloads are this framework's binary blocks, not pickles.
Executed in tests/test_plugins_fakes.py against an in-tree fake of the module (real-server parity unpinned).
"""
from __future__ import annotations

from typing import Optional, Tuple

import numpy as np

from . import ExperienceChannel, ParamChannel, codec


def _client(host: str):
    try:
        import redis  # type: ignore
    except ImportError as e:  # pragma: no cover - module absent in this image
        raise RuntimeError("transport=redis needs the `redis` python module, which is not installed; "
                           "use transport=shm on a single node") from e
    return redis.StrictRedis(host=host)


class RedisExperience(ExperienceChannel):
    KEY = "experience"

    def __init__(self, host: str = "localhost"):
        self._c = _client(host)

    def push(self, block, kind, actor_id=0, env_steps=0):
        self._c.rpush(self.KEY, bytes(codec.encode(block, kind, actor_id, env_steps)))   # actor.py:97
        return True

    def drain(self, max_blocks=1 << 30):
        # atomic drain (MULTI/EXEC), replay.py:26-29. The reference trims with LTRIM -1 0, which empties a list of n > 1
        # elements but KEEPS a single one (start = stop = 0), so a lone blob is delivered again on every poll until a second
        # one arrives; DEL inside the same transaction has the intended at-most-once semantics.
        pipe = self._c.pipeline()
        pipe.lrange(self.KEY, 0, -1)
        pipe.delete(self.KEY)
        data = pipe.execute()[0] or []
        return [codec.decode(d, copy=True) for d in data]

    def size(self):
        return int(self._c.llen(self.KEY))           # learner.py:48

    def reset(self):
        self._c.delete(self.KEY)                     # replay.py:21


class RedisParams(ParamChannel):
    KEY, VKEY = "params", "params_version"

    def __init__(self, host: str = "localhost"):
        self._c = _client(host)

    def publish(self, flat):
        a = np.ascontiguousarray(flat, dtype=np.float32)
        pipe = self._c.pipeline()
        pipe.set(self.KEY, a.tobytes())              # learner.py:83
        pipe.incr(self.VKEY)
        return int(pipe.execute()[1])

    def pull(self, have_version=0) -> Optional[Tuple[int, np.ndarray]]:
        v = self.version()
        if v <= have_version:
            return None
        raw = self._c.get(self.KEY)                  # actor.py:53
        if raw is None:
            return None
        return v, np.frombuffer(raw, dtype=np.float32).copy()

    def version(self):
        v = self._c.get(self.VKEY)
        return int(v) if v is not None else 0

    def reset(self):
        self._c.delete(self.KEY, self.VKEY)          # learner.py:37

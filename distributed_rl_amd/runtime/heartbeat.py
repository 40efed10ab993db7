"""Heartbeats in shared memory (the reference has none; a dead learner or actor is
never detected there). Workers ``tick(progress)``; the launcher's watchdog reads a snapshot."""
from __future__ import annotations

import ctypes
import os
from typing import Dict, List

ROLE = {"learner": 1, "actor": 2, "replay": 3, "other": 4}
ROLE_NAME = {v: k for k, v in ROLE.items()}


class HeartbeatTable:
    def __init__(self, name: str, create: bool = False):
        from .. import native
        self._lib = native.host()
        self.name, self._owner = name, create
        self._h = self._lib.drl_beat_open(name.encode(), 1 if create else 0)
        if not self._h:
            raise OSError("cannot %s heartbeat table %r" % ("create" if create else "open", name))

    def register(self, role: str, ident: int = 0) -> "Heartbeat":
        slot = self._lib.drl_beat_register(self._h, os.getpid(), ROLE[role], ident)
        if slot < 0:
            raise RuntimeError("heartbeat table full")
        return Heartbeat(self, slot)

    def snapshot(self) -> List[Dict]:
        buf = (ctypes.c_uint64 * (6 * 1024))()
        n = self._lib.drl_beat_snapshot(self._h, buf, 1024)
        return [dict(pid=int(buf[6 * i]), role=ROLE_NAME.get(int(buf[6 * i + 1]), "other"), ident=int(buf[6 * i + 2]),
                     beats=int(buf[6 * i + 3]), age_s=buf[6 * i + 4] / 1e9, progress=int(buf[6 * i + 5]))
                for i in range(n)]

    def free_pid(self, pid: int) -> int:
        """Supervisor side: free the slots a dead child registered (it could not unregister itself)."""
        return int(self._lib.drl_beat_free_pid(self._h, int(pid)))

    def close(self):
        if self._h:
            self._lib.drl_beat_close(self._h, self.name.encode() if self._owner else b"")
            self._h = None


class Heartbeat:
    def __init__(self, table: HeartbeatTable, slot: int):
        self._t, self.slot = table, slot

    def tick(self, progress: int = 0) -> None:
        self._t._lib.drl_beat_tick(self._t._h, self.slot, int(progress))

    def close(self) -> None:
        self._t._lib.drl_beat_unregister(self._t._h, self.slot)

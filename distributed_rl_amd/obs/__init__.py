"""Observability: metrics registry, phase timers, sinks.

The reference hard-wires visdom into both entrypoints and sends one HTTP request per learner step
and one RGB frame per actor env step (ape_x/learner.py:86, ape_x/actor.py:82). Here metrics go
through a :class:`Metrics` registry with pluggable sinks (stdout / JSONL by default; visdom only if
the module is importable) and are flushed at a bounded rate, never on the hot path.
"""
from __future__ import annotations

import json
import os
import sys
import time
from collections import defaultdict
from contextlib import contextmanager
from typing import Dict, List, Optional


class Sink:
    def scalar(self, name: str, x: float, y: float) -> None:
        raise NotImplementedError

    def image(self, name: str, img) -> None:
        pass

    def close(self) -> None:
        pass


class StdoutSink(Sink):
    def __init__(self, prefix: str = "", stream=None):
        self.prefix, self.stream = prefix, stream or sys.stdout

    def scalar(self, name, x, y):
        self.stream.write("%s%s[%s] = %.6g\n" % (self.prefix, name, _fmt_x(x), y))
        self.stream.flush()


class JsonlSink(Sink):
    def __init__(self, path: str):
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        self._fh = open(path, "a", buffering=1)

    def scalar(self, name, x, y):
        self._fh.write(json.dumps({"t": time.time(), "name": name, "x": x, "y": y}) + "\n")

    def close(self):
        self._fh.close()


class VisdomSink(Sink):
    """Line plots / image windows like the reference's dashboard ('Memory size', 'Q loss',
    'Score actor_i'; learner.py:39-42, actor.py:46-48). Requires the optional ``visdom`` module."""

    def __init__(self, server: str = "localhost"):
        import visdom  # type: ignore
        import numpy as np
        self._np = np
        self._vis = visdom.Visdom(server="http://" + server)
        self._wins: Dict[str, object] = {}

    def scalar(self, name, x, y):
        np = self._np
        if name not in self._wins:
            self._wins[name] = self._vis.line(X=np.array([x]), Y=np.array([y]), opts=dict(title=name))
        else:
            self._vis.line(X=np.array([x]), Y=np.array([y]), win=self._wins[name], update="append")

    def image(self, name, img):
        if name not in self._wins:
            self._wins[name] = self._vis.image(img, opts=dict(title=name))
        else:
            self._vis.image(img, win=self._wins[name])


def _fmt_x(x):
    return "%d" % x if float(x).is_integer() else "%.3f" % x


class Metrics:
    """Named scalar series + wall-clock phase timers."""

    def __init__(self, sinks: Optional[List[Sink]] = None, min_interval: float = 0.0):
        self.sinks = sinks if sinks is not None else []
        self.timers: Dict[str, List[float]] = defaultdict(lambda: [0.0, 0])
        self._last: Dict[str, float] = {}
        self.min_interval = min_interval

    def scalar(self, name: str, x: float, y: float, force: bool = False) -> None:
        now = time.time()
        if not force and self.min_interval > 0 and now - self._last.get(name, 0.0) < self.min_interval:
            return
        self._last[name] = now
        for s in self.sinks:
            s.scalar(name, x, float(y))

    def image(self, name: str, img) -> None:
        for s in self.sinks:
            s.image(name, img)

    @contextmanager
    def phase(self, name: str):
        t0 = time.perf_counter()
        try:
            yield
        finally:
            acc = self.timers[name]
            acc[0] += time.perf_counter() - t0
            acc[1] += 1

    def phase_report(self) -> Dict[str, Dict[str, float]]:
        return {k: dict(total_s=v[0], calls=v[1], mean_ms=1e3 * v[0] / max(1, v[1])) for k, v in self.timers.items()}

    def close(self):
        for s in self.sinks:
            s.close()


def make_metrics(role: str, visdom_server: str = "", jsonl_dir: str = "", quiet: bool = False,
                 min_interval: float = 1.0) -> Metrics:
    sinks: List[Sink] = []
    if not quiet:
        sinks.append(StdoutSink(prefix="[%s] " % role))
    if jsonl_dir:
        sinks.append(JsonlSink(os.path.join(jsonl_dir, "%s.jsonl" % role)))
    if visdom_server:
        try:
            sinks.append(VisdomSink(visdom_server))
        except Exception as e:  # visdom missing or server down: metrics must never kill a worker
            sys.stderr.write("[%s] visdom sink disabled: %s\n" % (role, e))
    return Metrics(sinks, min_interval=min_interval)

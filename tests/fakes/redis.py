"""Fake ``redis`` module (tests only): the subset of StrictRedis the reference uses (ape_x/replay.py:24-35,
ape_x/actor.py:96-98, ape_x/learner.py:36-37,48,83), with real Redis semantics for those commands, shared between
processes through a directory of files guarded by one flock (``FAKE_REDIS_DIR``; one directory = one server)."""
import fcntl
import os
import struct

__fake__ = True


def _root(host):
    d = os.environ.get("FAKE_REDIS_DIR")
    if not d:
        raise ConnectionError("fake redis: FAKE_REDIS_DIR is not set (no server)")
    d = os.path.join(d, "".join(c if c.isalnum() else "_" for c in str(host)))
    os.makedirs(d, exist_ok=True)
    return d


def _b(v):
    if isinstance(v, bytes):
        return v
    if isinstance(v, (bytearray, memoryview)):
        return bytes(v)
    return str(v).encode()


class _Store:
    """All state of one server. Values: bytes (strings) or list of bytes. One file per key: b'S' + data or b'L' + records."""

    def __init__(self, root):
        self.root = root
        self._lock_path = os.path.join(root, ".lock")

    def locked(self):
        store = self

        class _L:
            def __enter__(self_):
                self_.fh = open(store._lock_path, "a+")
                fcntl.flock(self_.fh, fcntl.LOCK_EX)

            def __exit__(self_, *a):
                fcntl.flock(self_.fh, fcntl.LOCK_UN)
                self_.fh.close()
        return _L()

    def _path(self, key):
        return os.path.join(self.root, "k_" + _b(key).hex())

    def load(self, key):
        try:
            with open(self._path(key), "rb") as fh:
                raw = fh.read()
        except FileNotFoundError:
            return None
        if raw[:1] == b"S":
            return raw[1:]
        out, o = [], 1
        while o < len(raw):
            n, = struct.unpack_from("<Q", raw, o)
            out.append(raw[o + 8:o + 8 + n])
            o += 8 + n
        return out

    def save(self, key, val):
        p = self._path(key)
        if val is None or (isinstance(val, list) and not val):       # Redis deletes a list that becomes empty
            try:
                os.unlink(p)
            except FileNotFoundError:
                pass
            return
        tmp = p + ".tmp"
        with open(tmp, "wb") as fh:
            if isinstance(val, list):
                fh.write(b"L")
                for v in val:
                    fh.write(struct.pack("<Q", len(v)))
                    fh.write(v)
            else:
                fh.write(b"S" + val)
        os.replace(tmp, p)

    # ---- commands (called with the lock held) ----
    def rpush(self, key, *vals):
        cur = self.load(key) or []
        if not isinstance(cur, list):
            raise ResponseError("WRONGTYPE")
        cur.extend(_b(v) for v in vals)
        self.save(key, cur)
        return len(cur)

    @staticmethod
    def _range(n, start, stop):
        if start < 0:
            start = max(n + start, 0)
        if stop < 0:
            stop = n + stop
        stop = min(stop, n - 1)
        return start, stop                                          # inclusive; empty when start > stop

    def lrange(self, key, start, stop):
        cur = self.load(key) or []
        a, b = self._range(len(cur), start, stop)
        return cur[a:b + 1] if a <= b else []

    def ltrim(self, key, start, stop):
        cur = self.load(key) or []
        a, b = self._range(len(cur), start, stop)
        self.save(key, cur[a:b + 1] if a <= b else [])
        return True

    def llen(self, key):
        cur = self.load(key)
        return len(cur) if isinstance(cur, list) else 0

    def set(self, key, val):
        self.save(key, _b(val))
        return True

    def get(self, key):
        cur = self.load(key)
        if isinstance(cur, list):
            raise ResponseError("WRONGTYPE")
        return cur

    def incr(self, key, amount=1):
        cur = self.load(key)
        v = int(cur or b"0") + amount
        self.save(key, str(v).encode())
        return v

    def delete(self, *keys):
        n = 0
        for k in keys:
            if self.load(k) is not None:
                n += 1
                self.save(k, None)
        return n


class ResponseError(Exception):
    pass


class ConnectionError(Exception):          # noqa: A001 - mirrors redis.ConnectionError
    pass


class Pipeline:
    """MULTI/EXEC: commands are queued and executed atomically (under one lock) by ``execute``."""

    def __init__(self, store):
        self._s, self._q = store, []

    def __getattr__(self, name):
        if name.startswith("_") or not hasattr(_Store, name):
            raise AttributeError(name)

        def queue(*a, **kw):
            self._q.append((name, a, kw))
            return self
        return queue

    def execute(self):
        with self._s.locked():
            out = [getattr(self._s, n)(*a, **kw) for n, a, kw in self._q]
        self._q = []
        return out


class StrictRedis:
    def __init__(self, host="localhost", port=6379, db=0, **kw):
        self._s = _Store(_root(host))

    def pipeline(self, transaction=True):
        return Pipeline(self._s)

    def _one(self, name, *a, **kw):
        with self._s.locked():
            return getattr(self._s, name)(*a, **kw)

    def rpush(self, key, *vals):
        return self._one("rpush", key, *vals)

    def lrange(self, key, start, stop):
        return self._one("lrange", key, start, stop)

    def ltrim(self, key, start, stop):
        return self._one("ltrim", key, start, stop)

    def llen(self, key):
        return self._one("llen", key)

    def set(self, key, val):
        return self._one("set", key, val)

    def get(self, key):
        return self._one("get", key)

    def incr(self, key, amount=1):
        return self._one("incr", key, amount)

    def delete(self, *keys):
        return self._one("delete", *keys)


Redis = StrictRedis

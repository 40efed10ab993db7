"""Fake ``visdom`` module (tests only): records what the sinks send (reference calls: ape_x/learner.py:39-42,86,91-92,
ape_x/actor.py:46-48,82,85-86)."""
__fake__ = True
CALLS = []          # (server, kind, kwargs) of every call made by any Visdom object of this process


class Visdom:
    def __init__(self, server="http://localhost", port=8097, **kw):
        self.server = server
        self._n = 0

    def _win(self, win):
        if win is None:
            self._n += 1
            return "win_%d" % self._n
        return win

    def line(self, Y=None, X=None, win=None, update=None, opts=None, **kw):
        w = self._win(win)
        CALLS.append((self.server, "line", dict(X=X, Y=Y, win=win, update=update, opts=opts)))
        return w

    def image(self, img, win=None, opts=None, **kw):
        w = self._win(win)
        CALLS.append((self.server, "image", dict(img=img, win=win, opts=opts)))
        return w

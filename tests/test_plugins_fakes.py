"""The optional plugins, executed against in-tree fakes of the modules the MI355X image lacks (tests/fakes/: ``redis``,
``gym`` + ALE, ``visdom``). No package is installed and no reference binary is loaded; the fakes are scripted stand-ins.

* ``AtariVecEnv`` is pinned to the reference's ``MultiFrameAtariEnv`` (libs/wrapped_env.py:30-47) through an oracle
  written here from those lines: max over the last two RAW frames, 4 copies on reset, ``randint(1,30)//4`` no-ops,
  oldest-first stack, episode cap outside the no-ops.
* The Redis transport runs BASELINE config 1 the way the reference deploys it (fabfile.py:12-23):
  ``run.sh 0 config/learner.conf`` + ``run.sh 2 config/actor.conf`` as separate launchers sharing one (fake) server.
* ``VisdomSink`` draws the reference's windows (learner.py:39-42,86; actor.py:46-48,82,85-86).
Parity with the real third-party modules stays unpinned (they cannot be imported here); what is pinned is the framework's
side of each protocol.
"""
import glob
import importlib
import os
import subprocess
import sys
import time

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAKES = os.path.join(ROOT, "tests", "fakes")


@pytest.fixture
def fakes(monkeypatch, tmp_path):
    """Put tests/fakes in front of sys.path for this test only and forget the fake modules afterwards."""
    monkeypatch.syspath_prepend(FAKES)
    monkeypatch.setenv("FAKE_REDIS_DIR", str(tmp_path / "redis"))
    for m in [k for k in sys.modules if k.split(".")[0] in ("redis", "gym", "visdom")]:
        monkeypatch.delitem(sys.modules, m)
    importlib.invalidate_caches()
    yield
    for m in [k for k in sys.modules if k.split(".")[0] in ("redis", "gym", "visdom")]:
        del sys.modules[m]


# ---------------------------------------------------------------------------------------------------------------------
# gym / ALE
class _OracleMultiFrame:
    """The reference wrapper's arithmetic (wrapped_env.py:30-47) over one fake AtariEnv, uint8 instead of /255."""

    def __init__(self, game, seed, n_noop):
        from gym.envs.atari.atari_env import AtariEnv
        from distributed_rl_amd.utils import preprocess
        self.pre = lambda im: preprocess(im, (84, 84), gray=True, dtype=np.uint8)
        self.env = AtariEnv(game=game, obs_type="image", frameskip=4, repeat_action_probability=0.)
        self.env.seed(seed)
        self.buf = []
        self.reset(n_noop)

    def reset(self, n_noop):
        self.nx = self.env.reset()
        self.buf = [self.pre(self.nx)] * 4
        for _ in range(n_noop):
            self.step(0)
        return np.stack(self.buf)

    def step(self, a):
        cur = self.nx.copy()
        self.nx, r, d, info = self.env.step(a)
        self.buf = self.buf[1:] + [self.pre(np.maximum(self.nx, cur))]
        return np.stack(self.buf), r, d, info


def _noops_after_last_reset(calls):
    k = len(calls) - 1 - max(i for i, c in enumerate(calls) if c[0] == "reset")
    assert all(c == ("step", 0) for c in calls[len(calls) - k:])
    return k


def test_atari_vec_env_matches_reference_wrapper(fakes):
    from distributed_rl_amd import envs
    assert envs._ale_available()
    with _nowarn():                                                          # no 'using the synthetic env' fallback warning
        env = envs.make_vec("MultiFrameBreakout-v0", 2, seed=5)
    from distributed_rl_amd.envs.atari import AtariVecEnv
    assert isinstance(env, AtariVecEnv) and env.n_action == 4 and env.observation_shape == (84, 84, 4)
    inner = env.envs
    noops = [_noops_after_last_reset(e.calls) for e in inner]
    assert all(0 <= k <= 7 for k in noops)                                   # randint(1,30)//4 ∈ [0,7]
    oracles = [_OracleMultiFrame("breakout", 5 + i, noops[i]) for i in range(2)]
    obs = env.current_obs()
    for i in range(2):
        assert np.array_equal(np.moveaxis(obs[i], -1, 0), np.stack(oracles[i].buf))
    rng = np.random.default_rng(0)
    seen_done = 0
    for t in range(90):                                                      # crosses two scripted episode ends (37 frames)
        a = rng.integers(0, 4, 2)
        nobs, r, d, info = env.step(a)
        for i in range(2):
            o, ro, do, _ = oracles[i].step(int(a[i]))
            assert np.array_equal(np.moveaxis(nobs[i], -1, 0), o), (t, i)    # max(raw_t, raw_{t-1}), oldest plane first
            assert r[i] == ro and bool(d[i]) == do
            if do:
                seen_done += 1
                k = _noops_after_last_reset(inner[i].calls)
                assert 0 <= k <= 7
                oracles[i].reset(k)
                assert np.array_equal(np.moveaxis(env.current_obs()[i], -1, 0), np.stack(oracles[i].buf))
                assert list(info["done_idx"]).count(i) == 1
    assert seen_done >= 4
    # the flicker sprite of the fake sits on even raw frames only: without the two-frame max every other observation would
    # lose it; with it every newest plane carries a saturated patch
    assert all((env.current_obs()[i][..., -1] == 255).sum() > 20 for i in range(2))


class _nowarn:
    def __enter__(self):
        import warnings
        self._c = warnings.catch_warnings()
        self._c.__enter__()
        warnings.simplefilter("error")

    def __exit__(self, *a):
        return self._c.__exit__(*a)


def test_atari_episode_cap_excludes_noops_and_render(fakes):
    from distributed_rl_amd import envs
    env = envs.make_vec("MultiFramePong-v0", 1, seed=1, max_episode_steps=10)
    assert env.n_action == 6
    for t in range(10):
        _, _, d, info = env.step(np.zeros(1, np.int64))
        assert bool(d[0]) == (t == 9)                                        # TimeLimit counts agent steps only
    assert info["episode_length"][0] == 10
    rgb = env.render_rgb(0)                                                  # actor.py:46,82: env.env._get_image()
    assert rgb.shape == (3, 210, 160) and rgb.dtype == np.uint8
    single = envs.make("MultiFramePong-v0", seed=1)                          # gym.make facade, float32 CHW in [0,1]
    o = single.reset()
    assert o.shape == (4, 84, 84) and o.dtype == np.float32 and 0 <= o.min() and o.max() <= 1


# ---------------------------------------------------------------------------------------------------------------------
# redis
def test_fake_redis_semantics(fakes):
    import redis
    assert getattr(redis, "__fake__", False)
    c, c2 = redis.StrictRedis(host="h"), redis.StrictRedis(host="h")
    assert c.llen("experience") == 0 and c.get("params") is None
    assert c.rpush("experience", b"a") == 1 and c2.rpush("experience", b"b", b"c") == 3
    assert c.lrange("experience", 0, -1) == [b"a", b"b", b"c"] and c.lrange("experience", 1, 1) == [b"b"]
    p = c.pipeline()
    p.lrange("experience", 0, -1)
    p.ltrim("experience", -1, 0)                                             # replay.py:26-29: empties a list of n > 1 ...
    assert p.execute() == [[b"a", b"b", b"c"], True] and c2.llen("experience") == 0
    c.rpush("experience", b"z")
    c.ltrim("experience", -1, 0)
    assert c.llen("experience") == 1                                         # ... but KEEPS a single element (real Redis too)
    assert c.set("params", b"\x00\x01") and c2.get("params") == b"\x00\x01"
    assert c.incr("v") == 1 and c2.incr("v") == 2 and c.get("v") == b"2"
    assert c.delete("params", "v", "nope") == 2 and c.get("params") is None
    assert redis.StrictRedis(host="other").llen("experience") == 0           # one directory per server


def test_redis_channels_roundtrip_no_duplicates(fakes):
    from distributed_rl_amd.transport import codec, connect
    exp, par = connect("redis", "ignored", create=True, host="localhost")
    aexp, apar = connect("redis", "ignored", create=False, host="localhost")
    rng = np.random.default_rng(0)
    blk = dict(state=rng.integers(0, 256, (3, 84, 84, 4), dtype=np.uint8), next_state=rng.integers(0, 256, (3, 84, 84, 4), dtype=np.uint8),
               action=np.arange(3, dtype=np.int32), reward=np.ones(3, np.float32), done=np.zeros(3, np.float32),
               prio=np.full(3, 0.5, np.float32))
    assert apar.pull(0) is None                                              # nothing before the first SET (actor.py:53-54)
    assert aexp.push(blk, codec.KIND_APEX, 3, 17) and exp.size() == 1
    got = exp.drain()
    assert len(got) == 1 and np.array_equal(got[0]["state"], blk["state"]) and got[0]["actor_id"] == 3
    assert exp.drain() == [] and exp.size() == 0                             # a single queued blob is NOT delivered twice
    for k in range(3):
        aexp.push(blk, codec.KIND_APEX, k, 0)
    assert len(exp.drain()) == 3 and exp.drain() == []
    flat = rng.standard_normal(1000).astype(np.float32)
    v = par.publish(flat)
    got = apar.pull(0)
    assert got is not None and got[0] == v and np.array_equal(got[1], flat) and apar.pull(v) is None
    assert par.publish(flat * 2) == v + 1 and np.array_equal(apar.pull(v)[1], flat * 2)
    connect("redis", "ignored", create=True, host="localhost")               # learner start-up: DEL params, DEL experience
    assert apar.pull(0) is None and aexp.size() == 0


def test_run_sh_learner_conf_plus_actor_conf_over_redis(fakes, tmp_path):
    """BASELINE config 1 in the reference's two-host form: a learner-only launcher and an actor-only launcher that only
    share the Redis server."""
    env = dict(os.environ, PYTHONPATH=FAKES + os.pathsep + ROOT, OMP_NUM_THREADS="1", DRL_NUM_ENVS="2",
               FAKE_REDIS_DIR=str(tmp_path / "redis"))
    save = tmp_path / "models"
    learner = subprocess.Popen(
        ["bash", "run.sh", "0", "config/learner.conf", "localhost", "--timeout", "280", "--learner-args",
         "--max-steps 30 --batch-size 16 --start-memory-size 200 --save-timing 10 --save-dir %s --quiet --device cpu -e SyntheticCatch-v0"
         % save], env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    actors = subprocess.Popen(
        ["bash", "run.sh", "2", "config/actor.conf", "localhost", "--timeout", "280", "--actor-args",
         "--quiet --device cpu --actor-batch 16 --param-period 20 -e SyntheticCatch-v0 --log-dir %s" % (tmp_path / "logs")],
        env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    try:
        lout, _ = learner.communicate(timeout=300)
    finally:
        for p in (learner, actors):
            if p.poll() is None:
                p.terminate()
    aout, _ = actors.communicate(timeout=30)
    assert learner.returncode == 0 and "learner exited with 0" in lout, lout[-3000:]
    assert "transport: redis" in lout and "learner" not in "".join(ln for ln in aout.splitlines() if "exited" in ln), aout[-3000:]
    assert sorted(os.path.basename(p) for p in glob.glob(str(save / "model_*.pth")) if "aux" not in p) == \
        ["model_0.pth", "model_10.pth", "model_20.pth"]
    import redis
    c = redis.StrictRedis(host="localhost")
    assert int(c.get("params_version")) >= 30 and len(c.get("params")) >= 4 * 3292837     # DuelingDQN(A=4), SURVEY §2.5 M3 (flat pack, padded)
    logs = "".join(open(p).read() for p in glob.glob(str(tmp_path / "logs" / "actor_*.jsonl")))
    assert "Params version actor_1" in logs and "Params version actor_2" in logs              # both actors pulled over GET


# ---------------------------------------------------------------------------------------------------------------------
# visdom
def test_visdom_sink_draws_reference_windows(fakes):
    import visdom
    from distributed_rl_amd.obs import VisdomSink, make_metrics
    visdom.CALLS.clear()
    m = make_metrics("learner", visdom_server="vis-host", quiet=True, min_interval=0.0)
    assert any(isinstance(s, VisdomSink) for s in m.sinks)
    m.scalar("Q loss", 0, 1.5)
    m.scalar("Q loss", 1, 1.25)
    m.scalar("Memory size", 0, 10000)
    m.image("Frame actor_1", np.zeros((3, 210, 160), np.uint8))
    m.image("Frame actor_1", np.ones((3, 210, 160), np.uint8))
    kinds = [(k, kw.get("update"), (kw.get("opts") or {}).get("title")) for _, k, kw in visdom.CALLS]
    assert kinds == [("line", None, "Q loss"), ("line", "append", None), ("line", None, "Memory size"),
                     ("image", None, "Frame actor_1"), ("image", None, None)]
    assert all(srv == "http://vis-host" for srv, _, _ in visdom.CALLS)
    assert visdom.CALLS[1][2]["win"] == "win_1" and visdom.CALLS[4][2]["win"] == "win_3"    # appended to the SAME windows
    assert float(visdom.CALLS[1][2]["Y"][0]) == 1.25


def test_actor_render_every_sends_frames(fakes, run_id):
    """--render-every N: the actor's live frame window (actor.py:46-48,82), rate-limited instead of one HTTP per step."""
    import visdom
    from distributed_rl_amd import envs
    from distributed_rl_amd.config import make_config
    from distributed_rl_amd.models import DuelingDQN
    from distributed_rl_amd.obs import make_metrics
    from distributed_rl_amd.runtime.actor import Actor
    from distributed_rl_amd.transport import connect
    visdom.CALLS.clear()
    torch.manual_seed(0)
    exp, par = connect("inproc", run_id, create=True)
    env = envs.make_vec("MultiFrameBreakout-v0", 1, seed=0)
    m = make_metrics("actor_1", visdom_server="v", quiet=True, min_interval=0.0)
    actor = Actor(1, env, DuelingDQN(4), make_config("ape_x", actor_batch=8), exp, par, m, 1, seed=0, render_every=5)
    actor.run(max_steps=21)
    imgs = [kw for _, k, kw in visdom.CALLS if k == "image"]
    assert len(imgs) == 5 and imgs[0]["img"].shape == (3, 210, 160)          # t = 0,5,10,15,20: the RGB screen
    assert exp.size() >= 1                                                   # and the env fed experience the usual way

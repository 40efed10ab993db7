#!/usr/bin/env python
"""Does it LEARN? SyntheticCatch end to end on one GPU: vectorised actor (HIP inference engine, epsilon ladder, n-step /
sequence assembly, initial priorities) -> in-process experience channel -> learner (graph-captured HIP engine, or the
PyTorch oracle learner with ``--learner torch``) -> parameter channel -> actor. Actor and learner alternate in one thread
(K env steps, then U updates), so a run is reproducible and its update-to-data ratio is fixed.

The score is the mean reward per ball (+1 caught, -1 missed) of the GREEDY policy on fresh environments: about -0.6 for an
untrained network, +1 for a perfect one.   python scripts/learn_catch.py --algo ape_x --rounds 40
(VERDICT r1 #6b; reference: the loop of ape_x/actor.py:58-102 + ape_x/learner.py:63-99, r2d2/actor.py + r2d2/learner.py.)
"""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_rl_amd import envs  # noqa: E402
from distributed_rl_amd.config import make_config  # noqa: E402
from distributed_rl_amd.models import DuelingDQN, DuelingLSTMDQN  # noqa: E402
from distributed_rl_amd.runtime.actor import Actor  # noqa: E402
from distributed_rl_amd.transport import connect  # noqa: E402

A = 4


def make_cfg(algo, lr=None):
    if algo == "ape_x":
        cfg = make_config("ape_x", batch_size=128, start_memory_size=2000, actor_batch=50, target_update=200,
                          actor_param_period=64, save_timing=10 ** 9, fit_timing=10 ** 9, lr=lr or 2.5e-4, gamma=0.99)
    else:
        cfg = make_config("r2d2", batch_size=32, start_memory_size=200, actor_batch=8, target_update=200,
                          actor_param_period=64, save_timing=10 ** 9, fit_timing=10 ** 9, lr=lr or 2.5e-4, gamma=0.99)
        cfg.burn_in, cfg.seq_len, cfg.overlap, cfg.nstep = 8, 16, 8, 3
    return cfg


def evaluate(algo, cfg, flat, n_envs=32, ticks=400, seed=123):
    """Mean reward per ball of the greedy policy with parameters ``flat`` (kernel-layout fp32) on fresh environments."""
    env = envs.make_vec("SyntheticCatch-v0", n_envs, seed=seed)
    eps = np.zeros(n_envs, dtype=np.float32)
    if algo == "ape_x":
        from distributed_rl_amd.engine.actor import ActorEngine
        eng = ActorEngine(A, n_envs, "cuda", eps=eps, seed=seed, use_graph=False)
    else:
        from distributed_rl_amd.engine.r2d2_actor import R2D2ActorEngine
        eng = R2D2ActorEngine(A, n_envs, cfg, "cuda", eps=eps, seed=seed, use_graph=False)
    eng.load_flat(flat)
    obs = env.reset()
    tot, balls = 0.0, 0
    for _ in range(ticks):
        act = eng.act(obs)
        obs, rew, done, info = env.step(act)
        if algo == "r2d2":
            eng.reset(done)
        tot += float(rew.sum())
        balls += int((rew != 0).sum())
    return tot / max(1, balls)


def train(algo="ape_x", learner_kind="hip", rounds=30, actor_steps=64, updates=100, n_envs=32, seed=0, lr=None, log=None,
          evals=6, autocast=False):
    torch.manual_seed(seed)
    cfg = make_cfg(algo, lr)
    run_id = "learn-%s-%s-%d" % (algo, learner_kind, os.getpid())
    exp, par = connect("inproc", run_id, create=True)
    env = envs.make_vec("SyntheticCatch-v0", n_envs, seed=seed)
    if algo == "ape_x":
        from distributed_rl_amd.engine.actor import ActorEngine
        net = DuelingDQN(A)
        engine = ActorEngine(A, n_envs, "cuda", init_model=net, seed=seed + 1)
        actor = Actor(1, env, net, cfg, exp, par, None, 1, torch.device("cuda"), seed=seed, engine=engine)
        mk = lambda: DuelingDQN(A)                                                            # noqa: E731
    else:
        from distributed_rl_amd.engine.r2d2_actor import R2D2ActorEngine
        net = DuelingLSTMDQN(A, n_envs, cfg.burn_in, cfg.nstep)
        engine = R2D2ActorEngine(A, n_envs, cfg, "cuda", seed=seed + 1, init_model=net)
        actor = Actor(1, env, net, cfg, exp, par, None, 1, torch.device("cpu"),
                      target_net=DuelingLSTMDQN(A, n_envs, cfg.burn_in, cfg.nstep), seed=seed, engine=engine)
        mk = lambda: DuelingLSTMDQN(A, cfg.batch_size, cfg.burn_in, cfg.nstep)               # noqa: E731
    tmp = tempfile.mkdtemp(prefix="learn_catch_")
    if learner_kind == "hip":
        from distributed_rl_amd.engine.learner import HipLearner
        learner = HipLearner(algo, A, cfg, exp, par, None, replay_size=20000 if algo == "ape_x" else 2000, device="cuda",
                             publish_interval=0.0, seed=seed)
        flat = lambda: learner.engine.P.flat                                                  # noqa: E731
        publish = lambda: learner.publish(wait=True)                                          # noqa: E731
    else:
        from distributed_rl_amd.models import ParamPack
        from distributed_rl_amd.runtime.learner import Learner
        m, t = mk().cuda(), mk().cuda()
        t.load_state_dict(m.state_dict())
        if algo == "ape_x":
            opt = torch.optim.RMSprop(m.parameters(), lr=cfg.lr, alpha=cfg.rms_alpha, eps=cfg.rms_eps)
        else:
            opt = torch.optim.Adam(m.parameters(), lr=cfg.lr, eps=cfg.adam_eps)
        cfg.publish_period = 10 ** 9                   # the loop below publishes once per round (the reference: every step)
        learner = Learner(m, t, opt, cfg, exp, par, None, 20000 if algo == "ape_x" else 2000, device=torch.device("cuda"))
        flat = lambda: ParamPack.from_module(m).flat                                          # noqa: E731
        publish = learner.publish
    publish()
    actor.pull_params()
    score0 = evaluate(algo, cfg, flat())
    curve, t0 = [], time.time()
    actor.run(max_steps=3 * actor_steps)                   # enough for the learn-start size before the first update
    for r in range(rounds):
        actor.run(max_steps=actor_steps)
        if learner_kind == "hip":
            learner.optimize_loop(max_steps=updates, save_model_dir=tmp, warmup_timeout=60)
        elif autocast:       # the PyTorch learner with bf16 matrix operands (fp32 master weights): what precision alone costs
            with torch.autocast(device_type="cuda", dtype=torch.bfloat16):
                learner.optimize_loop(max_steps=updates, save_model_dir=tmp, start_step=learner.step, warmup_timeout=60)
        else:
            learner.optimize_loop(max_steps=updates, save_model_dir=tmp, start_step=learner.step, warmup_timeout=60)
        publish()
        if (r + 1) % max(1, rounds // evals) == 0 or r + 1 == rounds:
            curve.append((r + 1, round(evaluate(algo, cfg, flat()), 3)))
            if log:
                log("%s/%s round %d: greedy reward per ball %.3f (%.0f s)" % (algo, learner_kind, r + 1, curve[-1][1], time.time() - t0))
    if hasattr(learner, "close"):
        learner.close()
    return dict(algo=algo, learner=learner_kind, start=round(score0, 3), curve=curve, final=curve[-1][1],
                updates=rounds * updates, env_steps=rounds * actor_steps * n_envs, seconds=round(time.time() - t0, 1))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--algo", default="ape_x")
    ap.add_argument("--learner", default="hip")
    ap.add_argument("--rounds", type=int, default=30)
    ap.add_argument("--actor-steps", type=int, default=64)
    ap.add_argument("--updates", type=int, default=100)
    ap.add_argument("--envs", type=int, default=32)
    ap.add_argument("--lr", type=float, default=None)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--evals", type=int, default=6, help="evaluations of the greedy policy over the run")
    ap.add_argument("--autocast", action="store_true", help="torch learner under torch.autocast(bfloat16)")
    a = ap.parse_args()
    print(json.dumps(train(a.algo, a.learner, a.rounds, a.actor_steps, a.updates, a.envs, a.seed, a.lr,
                           log=lambda s: print(s, flush=True), evals=a.evals, autocast=a.autocast)))

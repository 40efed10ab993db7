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


class _RoundBf16(torch.autograd.Function):
    """x -> bf16(x) forward, g -> bf16(g) backward (tests/test_production_gpu.py): where the kernels store bf16, fp32 rounds too."""

    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).float()

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).float()


def _emulate_engine_storage(m, t, opt, algo):
    """The fp32 PyTorch learner with the HIP engines' STORAGE precision and nothing else of them: activations and activation
    gradients of the layers the kernels keep in bf16 are rounded to bf16, the networks compute with bf16-rounded copies of
    fp32 master weights (the optimizer updates the masters). Arithmetic, loss, TD targets and accumulation stay fp32."""
    names = ["conv1", "conv2", "conv3", "adv1", "val1"] + (["lstm"] if algo == "r2d2" else [])
    for net in (m, t):
        for n in names:
            def hook(mod, inp, out):
                if isinstance(out, tuple):
                    return (_RoundBf16.apply(out[0]),) + tuple(out[1:])
                return _RoundBf16.apply(out)
            getattr(net, n).register_forward_hook(hook)
    params = [p for p in m.parameters()]
    master = [p.detach().clone() for p in params]
    with torch.no_grad():
        for p in params:
            p.copy_(p.to(torch.bfloat16).float())
    step0 = opt.step

    def step(*a, **k):
        with torch.no_grad():
            for p, w in zip(params, master):
                p.copy_(w)                              # the update applies to the fp32 masters ...
        out = step0(*a, **k)
        with torch.no_grad():
            for p, w in zip(params, master):
                w.copy_(p)
                p.copy_(p.to(torch.bfloat16).float())   # ... the next forward sees their bf16 rounding
        return out
    opt.step = step


def train(algo="ape_x", learner_kind="hip", rounds=30, actor_steps=64, updates=100, n_envs=32, seed=0, lr=None, log=None,
          evals=6, autocast=False, emulate=False, engine_args=None, replay=None, keep_all=False):
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
        if engine_args:                                  # --engine-arg KEY=VALUE: constructor arguments of the engine (A/B runs)
            import distributed_rl_amd.engine.apex as EA
            import distributed_rl_amd.engine.r2d2 as ER
            for mod, name in ((EA, "ApexEngine"), (ER, "R2D2Engine")):
                orig = getattr(mod, name)
                setattr(mod, name, (lambda o: lambda *a_, **k_: o(*a_, **{**k_, **engine_args}))(orig))
        learner = HipLearner(algo, A, cfg, exp, par, None, replay_size=replay or (20000 if algo == "ape_x" else 2000), device="cuda",
                             publish_interval=0.0, seed=seed)
        flat = lambda: learner.engine.P.flat                                                  # noqa: E731
        publish = lambda: learner.publish(wait=True)                                          # noqa: E731
    elif learner_kind == "hybrid":
        # HipLearner's HBM replay, ingest pipeline, sampler and priority write-back - with the fp32 PyTorch network, loss and
        # optimizer in place of the engine's kernels: which half does a difference between the two learners live in?
        assert algo == "ape_x"
        from distributed_rl_amd.engine.learner import HipLearner
        from distributed_rl_amd.models import ParamPack
        from distributed_rl_amd.ops import hip as H
        from distributed_rl_amd.ops import oracle
        hl = HipLearner(algo, A, cfg, exp, par, None, replay_size=replay or 20000, device="cuda", publish_interval=0.0, seed=seed,
                        use_graph=False)
        m, t = mk().cuda(), mk().cuda()
        t.load_state_dict(m.state_dict())
        opt = torch.optim.RMSprop(m.parameters(), lr=cfg.lr, alpha=cfg.rms_alpha, eps=cfg.rms_eps)
        pack = ParamPack.from_module(m)

        class _Hybrid:
            step = 0

            def optimize_loop(self, max_steps, **_):
                eng, rep = hl.engine, hl.replay
                while len(rep) <= max(cfg.batch_size, cfg.start_memory_size):
                    hl.ingest()
                for _i in range(max_steps):
                    hl.ingest()
                    rep.sample(eng.st, begin=True, is_n_scale=1.0, n_is=eng.n_is)
                    st = H.read_step_state(eng.st)
                    idx = rep.idx.long()
                    s, s2 = rep.gather_stacks(idx)
                    w = oracle.is_weights(rep.prio, st["total_prio"], st["n_items"], st["beta"])
                    opt.zero_grad()
                    td = m.td_batch(t, s, rep.act[idx].long(), rep.rew[idx], s2, rep.done[idx], w, cfg.per_alpha, cfg.gamma_n)
                    td.loss.backward()
                    torch.nn.utils.clip_grad_norm_(m.parameters(), cfg.max_grad_norm)
                    opt.step()
                    rep.update(td.prio.detach().float().contiguous(), end_state=eng.st)
                    self.step += 1
                    if self.step % cfg.target_update == 0:
                        t.load_state_dict(m.state_dict())

            def close(self):
                hl.close()
        learner = _Hybrid()

        def flat():
            pack.load_state_dict(m.state_dict())
            return pack.flat

        publish = lambda: par.publish(flat().cpu().numpy())                                   # noqa: E731
    else:
        from distributed_rl_amd.models import ParamPack
        from distributed_rl_amd.runtime.learner import Learner
        m, t = mk().cuda(), mk().cuda()
        t.load_state_dict(m.state_dict())
        if algo == "ape_x":
            opt = torch.optim.RMSprop(m.parameters(), lr=cfg.lr, alpha=cfg.rms_alpha, eps=cfg.rms_eps)
        else:
            opt = torch.optim.Adam(m.parameters(), lr=cfg.lr, eps=cfg.adam_eps)
        if emulate:
            _emulate_engine_storage(m, t, opt, algo)
        cfg.publish_period = 10 ** 9                   # the loop below publishes once per round (the reference: every step)
        # the reference's replay is trimmed to its capacity only every fit_timing learner steps (learner.py:88-92): never,
        # with the 10 ** 9 above - the learner would train on everything ever collected while the HIP learner's HBM ring holds
        # the newest ``replay`` transitions (60 000 updates: 0.98 reward per ball against 0.93; the HIP learner with a replay
        # that holds everything: 0.98, profiles/learning_long_and_system_r3.txt). Once per round keeps the two comparable.
        if not keep_all:
            cfg.fit_timing = updates
        learner = Learner(m, t, opt, cfg, exp, par, None, replay or (20000 if algo == "ape_x" else 2000), device=torch.device("cuda"))
        flat = lambda: ParamPack.from_module(m).flat                                          # noqa: E731
        publish = learner.publish
    publish()
    actor.pull_params()
    score0 = evaluate(algo, cfg, flat())
    curve, t0 = [], time.time()
    actor.run(max_steps=3 * actor_steps)                   # enough for the learn-start size before the first update
    for r in range(rounds):
        actor.run(max_steps=actor_steps)
        if learner_kind in ("hip", "hybrid"):
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
                hl_ = learner if learner_kind == "hip" else (hl if learner_kind == "hybrid" else None)
                if hl_ is not None:                          # is everything the actor published in the replay?
                    log("    published %d blocks (%d dropped), channel backlog %d, ingested %d items, replay holds %d, in flight %s"
                        % (actor.published_blocks, actor.dropped_blocks, exp.size(), hl_.ingested, len(hl_.replay),
                           hl_.pipe.stats() if hl_.pipe is not None else None))
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
    ap.add_argument("--emulate-bf16", action="store_true",
                    help="torch learner with the HIP engines' storage precision emulated (bf16 activations / weights copies)")
    ap.add_argument("--replay", type=int, default=None, help="replay capacity (default 20 000 transitions / 2 000 sequences)")
    ap.add_argument("--keep-all", action="store_true", help="torch learner: never trim the replay to its capacity")
    ap.add_argument("--engine-arg", action="append", default=[], help="KEY=VALUE constructor argument of the HIP engine")
    a = ap.parse_args()
    import ast
    eargs = {}
    for kv in a.engine_arg:
        k_, v_ = kv.split("=", 1)
        try:
            eargs[k_] = ast.literal_eval(v_)
        except (ValueError, SyntaxError):
            eargs[k_] = v_
    print(json.dumps(train(a.algo, a.learner, a.rounds, a.actor_steps, a.updates, a.envs, a.seed, a.lr,
                           log=lambda s: print(s, flush=True), evals=a.evals, autocast=a.autocast, emulate=a.emulate_bf16, engine_args=eargs, replay=a.replay, keep_all=a.keep_all)))

"""End-to-end plumbing (BASELINE config 1: `run.sh 2` = 1 learner + 2 actors on CPU): real processes, shm
transport, synthetic Atari-shaped env; asserts loss finite, params version advances, checkpoint written
and loadable into a reference-shaped state_dict, conf semantics (learner=false honoured)."""
import glob
import os
import subprocess
import sys
import threading
import time

import numpy as np
import pytest
import torch

from distributed_rl_amd import envs
from distributed_rl_amd.config import make_config, parse_conf
from distributed_rl_amd.models import DuelingDQN, DuelingLSTMDQN
from distributed_rl_amd.runtime.actor import Actor
from distributed_rl_amd.runtime.launcher import build_commands
from distributed_rl_amd.runtime.learner import Learner, load_checkpoint, save_checkpoint
from distributed_rl_amd.transport import connect

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_conf_parsing_and_commands():
    conf = parse_conf(os.path.join(ROOT, "config", "actor.conf"))
    assert conf.actor and not conf.learner and conf.algorithm == "ape_x"
    cmds = build_commands(conf, 3, "10.0.0.1", "rid", [], [])
    assert "learner" not in cmds and sorted(cmds) == ["actor_1", "actor_2", "actor_3"]     # reference bug Q5 fixed
    assert cmds["actor_2"][cmds["actor_2"].index("-r") + 1] == "10.0.0.1"                  # run.sh:7-9 default to $3
    conf = parse_conf(os.path.join(ROOT, "config", "all_r2d2.conf"))
    assert conf.learner and conf.algorithm == "r2d2"
    cmds = build_commands(conf, 2, "", "rid", ["--max-steps", "3"], [])
    assert cmds["learner"][-2:] == ["--max-steps", "3"] and "-t" in cmds["actor_1"]
    conf = parse_conf(os.path.join(ROOT, "config", "learner.conf"))
    assert not conf.actor and conf.actor_device == "cpu"
    assert "-d" in build_commands(conf, 0, "h", "rid", [], [])["learner"]


@pytest.mark.parametrize("algo", ["ape_x", "r2d2"])
def test_inproc_learning_loop(algo, tmp_path, run_id):
    torch.manual_seed(0)
    torch.set_num_threads(4)
    cfg = make_config(algo, batch_size=8, start_memory_size=16, actor_batch=8 if algo == "ape_x" else 2,
                      save_timing=4, target_update=3, actor_param_period=5)
    if algo == "r2d2":
        cfg.burn_in, cfg.seq_len, cfg.overlap = 4, 8, 4
    exp, par = connect("inproc", run_id, create=True)
    env = envs.make_vec("SyntheticCatch-v0", 4, seed=0)
    if algo == "ape_x":
        m, t = DuelingDQN(4), DuelingDQN(4)
        opt = torch.optim.RMSprop(m.parameters(), lr=cfg.lr, alpha=cfg.rms_alpha, eps=cfg.rms_eps)
        actor = Actor(1, env, DuelingDQN(4), cfg, exp, par, None, 2, seed=0)
    else:
        m, t = DuelingLSTMDQN(4, 8, 4, 5), DuelingLSTMDQN(4, 8, 4, 5)
        opt = torch.optim.Adam(m.parameters(), lr=cfg.lr, eps=cfg.adam_eps)
        actor = Actor(1, env, DuelingLSTMDQN(4, 4, 4, 5), cfg, exp, par, None, 2, target_net=DuelingLSTMDQN(4, 4, 4, 5), seed=0)
    learner = Learner(m, t, opt, cfg, exp, par, None, 64)
    w0 = m.conv1.weight.detach().clone()
    learner.publish()
    stop = threading.Event()
    th = threading.Thread(target=actor.run, kwargs=dict(stop_event=stop))
    th.start()
    try:
        n = learner.optimize_loop(max_steps=9, save_model_dir=str(tmp_path), warmup_timeout=120)
    finally:
        stop.set()
        th.join()
        learner.close()
    assert n == 9 and np.isfinite(learner.last_loss)
    assert not torch.equal(w0, m.conv1.weight)                       # the optimizer moved the weights
    assert par.version() >= 9 and actor._version >= 1                # actors saw new params
    assert len(learner.replay) <= 64 + cfg.fit_timing * 50           # eviction ran at t=0
    assert sorted(os.path.basename(p) for p in glob.glob(str(tmp_path / "model_*.pth")) if "aux" not in p) == \
        ["model_0.pth", "model_4.pth", "model_8.pth"]                # learner.py:96-98 incl. t=0
    sd = torch.load(tmp_path / "model_8.pth", weights_only=True)     # reference format: a bare state_dict
    assert set(sd) == set(m.state_dict())
    # resume path (absent upstream)
    m2 = type(m)(4) if algo == "ape_x" else DuelingLSTMDQN(4, 8, 4, 5)
    t2 = type(m)(4) if algo == "ape_x" else DuelingLSTMDQN(4, 8, 4, 5)
    opt2 = type(opt)(m2.parameters(), lr=cfg.lr)
    step = load_checkpoint(str(tmp_path / "model_8.pth"), m2, t2, opt2)
    assert step == 8 and len(opt2.state_dict()["state"]) > 0


def test_run_sh_two_actors(tmp_path, run_id):
    """`./run.sh 2` with bounded learner: 1 learner + 2 actor processes over shm; checkpoint must appear."""
    save = tmp_path / "models"
    env = dict(os.environ, PYTHONPATH=ROOT, OMP_NUM_THREADS="2", DRL_NUM_ENVS="2")
    cmd = [os.path.join(ROOT, "run.sh"), "2", "config/all.conf", "", "--run-id", run_id, "--timeout", "240",
           "--learner-args", "--max-steps 6 --batch-size 16 --start-memory-size 32 --save-timing 5 --save-dir %s --quiet --device cpu" % save,
           "--actor-args", "--quiet --device cpu --actor-batch 16"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert "learner exited with 0" in r.stdout, r.stdout[-3000:]
    assert os.path.exists(save / "model_0.pth") and os.path.exists(save / "model_5.pth"), r.stdout[-3000:]
    sd = torch.load(save / "model_5.pth", weights_only=True)
    assert tuple(sd["adv1.weight"].shape) == (512, 3136)
    assert not glob.glob("/dev/shm/drl_%s_*" % run_id)             # segments cleaned up


def test_run_sh_two_data_parallel_learners(tmp_path, run_id):
    """`learner_gpus=2` in the conf: the launcher starts the learner under torch.distributed.run (2 ranks, gloo on CPU),
    every rank owns its own experience ring + replay shard, actor i feeds rank (i-1) % 2, gradients are all-reduced and
    only rank 0 writes checkpoints."""
    conf = tmp_path / "dp.conf"
    conf.write_text("actor=true\nlearner=true\nvisdom=false\nalgorithm=ape_x\nredis_server=localhost\n"
                    "visdom_server=localhost\nactor_device=cpu\nlearner_gpus=2\n")
    save = tmp_path / "models"
    env = dict(os.environ, PYTHONPATH=ROOT, OMP_NUM_THREADS="2", DRL_NUM_ENVS="2")
    cmd = [os.path.join(ROOT, "run.sh"), "2", str(conf), "", "--run-id", run_id, "--timeout", "280",
           "--learner-args", "--max-steps 6 --batch-size 16 --start-memory-size 32 --save-timing 5 --save-dir %s --quiet --device cpu --engine torch" % save,
           "--actor-args", "--quiet --device cpu --actor-batch 16"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=330)
    assert "learner exited with 0" in r.stdout, r.stdout[-3000:]
    assert os.path.exists(save / "model_5.pth"), r.stdout[-3000:]
    assert not glob.glob("/dev/shm/drl_%s_*" % run_id)             # both shards' segments cleaned up


def test_run_sh_collective_param_channel(tmp_path, run_id):
    """`param_channel=collective` in the conf: learner and actors form one torch.distributed group (gloo on CPU, RCCL
    between GPUs) and the flat parameter buffer reaches the actors by broadcast instead of the shm register; experience
    still flows through the shm ring."""
    conf = tmp_path / "pc.conf"
    conf.write_text("actor=true\nlearner=true\nvisdom=false\nalgorithm=ape_x\nredis_server=localhost\n"
                    "visdom_server=localhost\nactor_device=cpu\nparam_channel=collective\nparam_backend=gloo\n")
    save, logs = tmp_path / "models", tmp_path / "logs"
    env = dict(os.environ, PYTHONPATH=ROOT, OMP_NUM_THREADS="2", DRL_NUM_ENVS="2")
    cmd = [os.path.join(ROOT, "run.sh"), "2", str(conf), "", "--run-id", run_id, "--timeout", "280",
           "--learner-args", "--max-steps 8 --batch-size 16 --start-memory-size 32 --save-timing 5 --save-dir %s --quiet --device cpu --engine torch" % save,
           "--actor-args", "--quiet --device cpu --actor-batch 16 --param-period 4 --log-dir %s" % logs]
    r = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=330)
    assert "learner exited with 0" in r.stdout, r.stdout[-3000:]
    assert os.path.exists(save / "model_5.pth"), r.stdout[-3000:]
    assert not glob.glob("/dev/shm/drl_%s_*" % run_id)
    import json
    versions = []
    for f in glob.glob(str(logs / "*.jsonl")):
        for line in open(f):
            rec = json.loads(line)
            if rec["name"].startswith("Params version"):
                versions.append(rec["y"])
    assert versions and max(versions) >= 1, (versions, r.stdout[-2000:])      # the broadcast reached the actors


def test_run_sh_data_parallel_learners_with_collective_param_channel(tmp_path, run_id):
    """BASELINE config 3 in one run: `learner_gpus=2` AND `param_channel=collective`. Learner ranks and
    actors form ONE torch.distributed world (gloo on CPU); the learners' gradient all-reduce runs in their own sub-group,
    every learner rank broadcasts parameters to the actors of ITS shard over its own lane (runtime/placement.py)."""
    conf = tmp_path / "dpc.conf"
    conf.write_text("actor=true\nlearner=true\nvisdom=false\nalgorithm=ape_x\nredis_server=localhost\n"
                    "visdom_server=localhost\nactor_device=cpu\nlearner_gpus=2\nparam_channel=collective\nparam_backend=gloo\n")
    save, logs = tmp_path / "models", tmp_path / "logs"
    env = dict(os.environ, PYTHONPATH=ROOT, OMP_NUM_THREADS="2", DRL_NUM_ENVS="2")
    cmd = [os.path.join(ROOT, "run.sh"), "4", str(conf), "", "--run-id", run_id, "--timeout", "280",
           "--learner-args", "--max-steps 8 --batch-size 16 --start-memory-size 32 --save-timing 5 --save-dir %s --quiet --device cpu --engine torch" % save,
           "--actor-args", "--quiet --device cpu --actor-batch 16 --param-period 4 --log-dir %s" % logs]
    r = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=330)
    assert "learner exited with 0" in r.stdout and "learner_r1 exited with 0" in r.stdout, r.stdout[-3000:]
    assert "lane 0 <- learner rank 0" in r.stdout and "lane 1 <- learner rank 1" in r.stdout, r.stdout[-3000:]
    assert os.path.exists(save / "model_5.pth"), r.stdout[-3000:]
    assert not glob.glob("/dev/shm/drl_%s*" % run_id)
    import json
    seen = {}
    for f in glob.glob(str(logs / "actor_*.jsonl")):
        for line in open(f):
            rec = json.loads(line)
            if rec["name"].startswith("Params version"):
                seen[rec["name"]] = max(seen.get(rec["name"], 0), rec["y"])
    assert len(seen) == 4 and min(seen.values()) >= 1, (seen, r.stdout[-2000:])   # every actor of BOTH shards was reached

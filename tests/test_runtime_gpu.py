"""GPU runtime: HipLearner (HBM replay + graph-captured engine) fed by real actors through the transport;
checkpoint in the reference format + resume; ActorEngine actions / priorities vs the PyTorch oracle."""
import glob
import os
import time

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from distributed_rl_amd import envs  # noqa: E402
from distributed_rl_amd.config import make_config  # noqa: E402
from distributed_rl_amd.models import DuelingDQN, DuelingLSTMDQN  # noqa: E402
from distributed_rl_amd.runtime.actor import Actor  # noqa: E402
from distributed_rl_amd.transport import connect  # noqa: E402


def test_actor_engine_matches_module():
    from distributed_rl_amd.engine.actor import ActorEngine
    torch.manual_seed(0)
    m = DuelingDQN(6)
    with torch.no_grad():
        for p in m.parameters():
            p.copy_(p.to(torch.bfloat16).float())
    eng = ActorEngine(6, 16, "cuda", eps=np.zeros(16, np.float32), init_model=m)
    obs = np.random.default_rng(0).integers(0, 256, (16, 84, 84, 4), dtype=np.uint8)
    a = eng.act(obs)
    with torch.no_grad():
        q = m(torch.from_numpy(obs))
    torch.testing.assert_close(eng.q_values().cpu(), q, rtol=3e-2, atol=3e-2)
    top2 = q.topk(2, 1).values
    clear = (top2[:, 0] - top2[:, 1]) > 0.05
    assert (torch.from_numpy(a)[clear] == q.argmax(1)[clear]).all()
    a2 = eng.act(obs)                                        # graph replay path
    assert (torch.from_numpy(a2)[clear] == q.argmax(1)[clear]).all()
    # initial priorities (online == target)
    rng = np.random.default_rng(1)
    K = 20
    s, s2 = rng.integers(0, 256, (2, K, 84, 84, 4), dtype=np.uint8)
    act, rew, done = rng.integers(0, 6, K), rng.random(K).astype(np.float32), (rng.random(K) < 0.2).astype(np.float32)
    pr = eng.priorities(s, s2, act, rew, done, 0.99, 0.6)
    td = m.td_batch(m, torch.from_numpy(s), torch.from_numpy(act), torch.from_numpy(rew), torch.from_numpy(s2),
                    torch.from_numpy(done), None, 0.6, 0.99)
    np.testing.assert_allclose(pr, td.prio.detach().numpy(), rtol=0.15, atol=0.03)
    # the same priorities from a plane-pool block (codec kind 3: every frame once + two window positions per transition, one
    # staging buffer / one DMA per block); blocks of different pool sizes reuse the buffers
    # (and shorter blocks - an env that publishes early - a prefix of them)
    for M, k in ((K + 9, K), (K + 5, K), (3 * K, K), (12, 7), (40, K)):
        planes = rng.integers(0, 256, (M, 84, 84), dtype=np.uint8)
        ws, ws2 = rng.integers(0, M - 3, k), rng.integers(0, M - 3, k)
        st = lambda w: np.ascontiguousarray(np.transpose(planes[w[:, None] + np.arange(4)], (0, 2, 3, 1)))    # noqa: E731
        pr_planes = eng.priorities_planes(planes, ws, ws2, act[:k], rew[:k], done[:k], 0.99, 0.6)
        pr_stacks = eng.priorities(st(ws), st(ws2), act[:k], rew[:k], done[:k], 0.99, 0.6)
        assert pr_planes.shape == (k,)
        np.testing.assert_allclose(pr_planes, pr_stacks, rtol=2e-2, atol=2e-3)
    # parameter refresh from a flat buffer
    m2 = DuelingDQN(6)
    from distributed_rl_amd.models import ParamPack
    eng.load_flat(ParamPack.from_module(m2).flat.numpy())
    eng.act(obs)
    with torch.no_grad():
        q_ref = m2(torch.from_numpy(obs))
        assert float((eng.q_values().cpu() - q_ref).abs().max()) <= 2e-2 * float(q_ref.abs().max())


def test_hip_learner_apex_end_to_end(tmp_path, run_id):
    from distributed_rl_amd.engine.actor import ActorEngine
    from distributed_rl_amd.engine.learner import HipLearner
    cfg = make_config("ape_x", batch_size=32, start_memory_size=64, actor_batch=16, save_timing=5, target_update=4,
                      actor_param_period=10, fit_timing=5)
    exp, par = connect("inproc", run_id, create=True)
    learner = HipLearner("ape_x", 4, cfg, exp, par, None, replay_size=256, device="cuda", publish_interval=0.0, save_replay=True)
    v0 = learner.publish(wait=True)
    env = envs.make_vec("SyntheticCatch-v0", 8, seed=0)
    net = DuelingDQN(4)
    actor = Actor(1, env, net, cfg, exp, par, None, 2, torch.device("cuda"), seed=0,
                  engine=ActorEngine(4, 8, "cuda", init_model=net, seed=1))
    assert actor.pull_params() and actor._version == v0
    actor.run(max_steps=60)
    assert actor.published_blocks > 4
    n = learner.optimize_loop(max_steps=11, save_model_dir=str(tmp_path), warmup_timeout=60)
    assert n == 11 and len(learner.replay) > 64 and learner.ingested >= 64
    st = learner.engine.stats()
    assert st["step"] == 11 + 2 and np.isfinite(st["loss"])            # +2 warm-up steps of the graph capture
    learner.publish(wait=True)
    assert par.version() > v0
    learner.close()
    files = sorted(os.path.basename(p) for p in glob.glob(str(tmp_path / "model_*.pth")) if "aux" not in p and "replay" not in p)
    assert files == ["model_0.pth", "model_10.pth", "model_5.pth"]
    ref = DuelingDQN(4)
    ref.load_state_dict(torch.load(tmp_path / "model_10.pth", weights_only=True))      # reference-format checkpoint
    # resume: parameters, optimizer moments, step counter
    exp2, par2 = connect("inproc", run_id + "b", create=True)
    l2 = HipLearner("ape_x", 4, cfg, exp2, par2, None, replay_size=256, device="cuda")
    assert l2.resume(str(tmp_path / "model_10.pth")) == 10
    torch.testing.assert_close(l2.engine.P.flat.cpu(), ref_flat(ref))
    assert float(l2.engine.SQ.flat.abs().sum()) > 0 and l2.engine.stats()["step"] >= 10
    # ... and the replay (--save-replay): the snapshot taken at step 10, re-ingested oldest first: same observations, tables
    # and priorities in ingestion order
    snap = torch.load(tmp_path / "model_10.replay.pth", weights_only=True)
    n = int(snap["K"].sum())
    assert len(l2.replay) == n > 64 and l2.ingested == n
    r2, o = l2.replay, torch.arange(n, device="cuda")
    s_, s2_ = r2.gather_stacks(o)
    pl = snap["planes"].view(-1, 84, 84).cuda()
    first = torch.repeat_interleave(torch.cumsum(snap["M"], 0) - snap["M"], snap["K"]).cuda()
    for got, win in ((s_, snap["win_s"]), (s2_, snap["win_s2"])):
        want = pl[(first + win.cuda())[:, None] + torch.arange(4, device="cuda")[None, :]].permute(0, 2, 3, 1)
        assert torch.equal(got.view(n, 84, 84, 4), want)
    assert torch.equal(r2.act[:n].cpu(), snap["action"]) and torch.equal(r2.rew[:n].cpu(), snap["reward"])
    torch.testing.assert_close(r2.tree[:n].cpu(), snap["prio"])
    assert abs(r2.total_prios() - float(snap["prio"].sum())) < 1e-3 * float(snap["prio"].sum())
    # actor picks up the newer parameters
    assert actor.pull_params() and actor._version == par.version()


def ref_flat(module):
    from distributed_rl_amd.models import ParamPack
    return ParamPack.from_module(module).flat


def test_hip_learner_r2d2_end_to_end(tmp_path, run_id):
    from distributed_rl_amd.engine.learner import HipLearner
    cfg = make_config("r2d2", batch_size=8, start_memory_size=10, actor_batch=2, save_timing=4, target_update=3, fit_timing=2)
    cfg.burn_in, cfg.seq_len, cfg.overlap, cfg.nstep = 4, 8, 4, 2
    exp, par = connect("inproc", run_id, create=True)
    learner = HipLearner("r2d2", 4, cfg, exp, par, None, replay_size=64, device="cuda", publish_interval=0.0, save_replay=True)
    learner.publish(wait=True)
    env = envs.make_vec("SyntheticCatch-v0", 4, seed=0)
    actor = Actor(1, env, DuelingLSTMDQN(4, 4, 4, 2), cfg, exp, par, None, 2, torch.device("cpu"),
                  target_net=DuelingLSTMDQN(4, 4, 4, 2), seed=0)
    actor.pull_params()
    actor.run(max_steps=60)
    n = learner.optimize_loop(max_steps=6, save_model_dir=str(tmp_path), warmup_timeout=60)
    assert n == 6 and np.isfinite(learner.engine.stats()["loss"])
    learner.close()
    sd = torch.load(tmp_path / "model_4.pth", weights_only=True)
    m = DuelingLSTMDQN(4, 8, 4, 2)
    m.load_state_dict(sd)
    assert tuple(sd["lstm.weight_hh"].shape) == (2048, 512)
    # resume with the replay snapshot: sequences, stored states and priorities come back in ingestion order
    snap = torch.load(tmp_path / "model_4.replay.pth", weights_only=True)
    n = int(snap["length"].shape[0])
    exp2, par2 = connect("inproc", run_id + "b", create=True)
    l2 = HipLearner("r2d2", 4, cfg, exp2, par2, None, replay_size=64, device="cuda")
    assert l2.resume(str(tmp_path / "model_4.pth")) == 4 and len(l2.replay) == n >= 10
    r2 = l2.replay
    m_tot = int(snap["M"].sum())
    assert int(snap["K"].sum()) == n and m_tot < n * r2.P                          # neighbouring sequences share planes
    assert torch.equal(r2.frames[:m_tot].cpu(), snap["planes"]) and torch.equal(r2.length[:n].cpu(), snap["length"])
    live = learner.replay.gather_stacks(torch.arange(n, device="cuda")).cpu()       # same observations through both stores
    assert torch.equal(r2.gather_stacks(torch.arange(n, device="cuda")).cpu(), live)
    torch.testing.assert_close(r2.h0[:n].cpu(), snap["h0"])
    torch.testing.assert_close(r2.tree[:n].cpu(), snap["prio"])
    l2.close()


def test_r2d2_actor_engine_matches_module():
    """Recurrent actor engine: step-by-step Q / state vs the PyTorch module (incl. reset), sequence priorities vs
    model.sequence_td, parameter refresh."""
    from distributed_rl_amd.config import R2D2Config
    from distributed_rl_amd.engine.r2d2_actor import R2D2ActorEngine
    torch.manual_seed(0)
    E, A = 16, 5
    cfg = R2D2Config(burn_in=4, seq_len=8, nstep=2)
    m = DuelingLSTMDQN(A, E, cfg.burn_in, cfg.nstep)
    with torch.no_grad():
        for p in m.parameters():
            p.copy_(p.to(torch.bfloat16).float())
    m.eval()
    m.hx, m.cx = torch.zeros(E, 512), torch.zeros(E, 512)
    eng = R2D2ActorEngine(A, E, cfg, "cuda", eps=np.zeros(E, np.float32), init_model=m)
    rng = np.random.default_rng(0)
    for step in range(6):
        obs = rng.integers(0, 256, (E, 84, 84, 4), dtype=np.uint8)
        h_before, c_before = eng.get_state()
        np.testing.assert_allclose(h_before, m.hx.numpy(), rtol=5e-2, atol=3e-2)
        np.testing.assert_allclose(c_before, m.cx.numpy(), rtol=5e-2, atol=3e-2)
        a = eng.act(obs)
        with torch.no_grad():
            q = m(torch.from_numpy(obs))
        assert float((eng.q_values().cpu() - q).abs().max()) <= 2e-2 * float(q.abs().max())
        top2 = q.topk(2, 1).values
        clear = (top2[:, 0] - top2[:, 1]) > 0.1
        assert (torch.from_numpy(a)[clear] == q.argmax(1)[clear]).all()
        if step == 2:
            done = np.zeros(E, bool)
            done[[1, 7]] = True
            eng.reset(done)
            m.reset(torch.from_numpy(done))
    # sequence priorities
    T, K = cfg.total_len, 5
    frames = rng.integers(0, 256, (T, K, 84, 84, 4), dtype=np.uint8)
    action = rng.integers(0, A, (T, K)).astype(np.int32)
    reward = rng.standard_normal((T, K)).astype(np.float32)
    done = np.zeros((T, K), np.float32)
    lengths = np.array([T, T, T - 2, T, T - 3], dtype=np.int32)
    done[T - 3, 2] = done[T - 4, 4] = 1.0
    h0 = (rng.standard_normal((K, 512)) * 0.2).astype(np.float32)
    c0 = (rng.standard_normal((K, 512)) * 0.2).astype(np.float32)
    pr = eng.sequence_priorities(frames, action, reward, done, lengths, h0, c0)
    tgt = DuelingLSTMDQN(A, K, cfg.burn_in, cfg.nstep)
    tgt.load_state_dict(m.state_dict())
    mk = DuelingLSTMDQN(A, K, cfg.burn_in, cfg.nstep)
    mk.load_state_dict(m.state_dict())
    with torch.no_grad():
        td = mk.sequence_td(tgt, torch.from_numpy(frames), torch.from_numpy(action).long(), torch.from_numpy(reward),
                            torch.from_numpy(done), torch.from_numpy(lengths), torch.from_numpy(h0).to(torch.bfloat16).float(),
                            torch.from_numpy(c0), None, cfg.eta, cfg.gamma_n, cfg.rescale_eps, require_grad=False)
    np.testing.assert_allclose(pr, td.prio.numpy(), rtol=0.1, atol=0.03)
    # the same from a plane pool + window positions (codec kind 4): one staging image / one DMA per block; pools of different
    # sizes and blocks of fewer sequences reuse the buffers
    for M, k in ((300, K), (1500, K), (200, 3)):
        planes = rng.integers(0, 256, (M, 84, 84), dtype=np.uint8)
        win = rng.integers(0, M - 3, (k, T))
        fr = np.ascontiguousarray(np.transpose(planes[win.T[:, :, None] + np.arange(4)], (0, 1, 3, 4, 2)))      # (T, k, 84, 84, 4)
        pr_planes = eng.sequence_priorities_planes(planes, win, np.ascontiguousarray(action.T[:k]), np.ascontiguousarray(reward.T[:k]),
                                                   np.ascontiguousarray(done.T[:k]), lengths[:k], h0[:k], c0[:k])
        pr_frames = eng.sequence_priorities(fr, action[:, :k], reward[:, :k], done[:, :k], lengths[:k], h0[:k], c0[:k])
        assert pr_planes.shape == (k,)
        np.testing.assert_allclose(pr_planes, pr_frames, rtol=2e-2, atol=2e-3)
    # parameter refresh also refreshes the actor-side target
    from distributed_rl_amd.models import ParamPack
    m2 = DuelingLSTMDQN(A, E, cfg.burn_in, cfg.nstep)
    eng.load_flat(ParamPack.from_module(m2).flat.numpy())
    assert torch.equal(eng.Tb.flat, eng.Pb.flat)


def test_r2d2_runtime_actor_with_engine_publishes_sequences(run_id):
    from distributed_rl_amd.engine.r2d2_actor import R2D2ActorEngine
    from distributed_rl_amd.transport import codec
    cfg = make_config("r2d2", burn_in=4, seq_len=8, overlap=4, nstep=2, actor_batch=4, actor_param_period=50)
    exp, par = connect("inproc", run_id, create=True)
    E = 4
    env = envs.make_vec("SyntheticCatch-v0", E, seed=0)
    net = DuelingLSTMDQN(env.n_action, E, cfg.burn_in, cfg.nstep)
    tgt = DuelingLSTMDQN(env.n_action, E, cfg.burn_in, cfg.nstep)
    eng = R2D2ActorEngine(env.n_action, E, cfg, "cuda", init_model=net)
    actor = Actor(1, env, net, cfg, exp, par, None, 1, "cpu", target_net=tgt, seed=0, engine=eng)
    actor.run(max_steps=60)
    assert actor.published_blocks > 0
    blocks = exp.drain()
    assert len(blocks) > 0
    b = blocks[0]
    assert b["kind"] == codec.KIND_R2D2_PLANES and b["win"].shape == (b["action"].shape[0], cfg.total_len) and b["h0"].shape[1] == 512
    assert b["planes"].shape[0] < b["action"].shape[0] * (cfg.total_len + 3)       # consecutive sequences share their planes
    assert np.isfinite(b["prio"]).all() and (b["prio"] >= 0).all()


def test_run_sh_two_data_parallel_hip_learners():
    """``learner_gpus=2`` with the gfx950 engines: ./run.sh starts two learner ranks (here sharing the one GPU, handles
    exchanged over gloo) + two actors; the gradient all-reduce is the direct xGMI kernel inside each rank's step graph
    (DRL_XGMI_AR=1: required, no fallback); rank 0 writes the checkpoints."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, DRL_DIST_BACKEND="gloo", DRL_SHARE_GPU="1", DRL_XGMI_AR="1",
               DRL_E2E_CONF="config/all_dp2.conf", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run(["bash", os.path.join(root, "scripts", "e2e_gpu.sh"), "ape_x"], cwd=root, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=170)
    assert r.returncode == 0 and "E2E ape_x OK" in r.stdout, r.stdout[-3000:]


_LANE_SCRIPT = r'''
import json, os, sys, time
import torch
sys.path.insert(0, os.environ["DRL_ROOT"])
from distributed_rl_amd.parallel.world import join_world
from distributed_rl_amd.runtime.nodes import _learner_fanout
from distributed_rl_amd.engine.learner import _AsyncPublisher
dev = torch.device("cuda:0")
w = join_world(dev)                                   # world of ONE: learner rank 0 + a lane that holds only itself
assert w is not None and w.size == 1 and w.my_lanes() == [0] and w.backend == "cpu:gloo,cuda:nccl"
n = 3292837
flat = torch.randn(n, device=dev)
fan = _learner_fanout(w, None, n, dev, keep_register=False)
lane = fan.lanes[0]
assert lane.device.type == "cuda" and fan.wants_device and not fan.wants_host
pub = _AsyncPublisher(flat, fan)
assert pub.host is None                               # no pinned host buffer: nothing of a publication crosses PCIe
v1 = pub.publish(wait=True)
assert torch.equal(lane._stage, flat)                 # what RCCL broadcast IS the learner's device snapshot
flat.mul_(2.0)
t0 = time.perf_counter()
v2 = pub.publish(wait=True)
dt = time.perf_counter() - t0
assert pub.error is None and v2 == v1 + 1 == 2 and torch.equal(lane._stage, flat)
pub.close(); fan.close(); w.close()
print(json.dumps({"ok": True, "publish_ms": dt * 1e3, "channel_us": pub.last_us}))
'''


def test_collective_param_lane_takes_the_device_snapshot_over_rccl(tmp_path):
    """CollectiveParams with a DEVICE tensor over the nccl (= RCCL) backend, driven by the HIP
    learner's asynchronous publisher, world size 1 (the one GPU of the box). The lane is built the way learner_node.py
    builds it from the launcher's world plan (parallel/world.py + runtime/nodes._learner_fanout)."""
    import json
    import socket
    import subprocess
    import sys
    from distributed_rl_amd.config import LaunchConf
    from distributed_rl_amd.runtime.placement import Plan, child_env
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    plan = Plan(["cuda:0"], [], [[0]], 1, "nccl", "cpu:gloo,cuda:nccl")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    script = tmp_path / "lane.py"
    script.write_text(_LANE_SCRIPT)
    env = dict(os.environ, DRL_ROOT=root, HSA_ENABLE_IPC_MODE_LEGACY="0", **child_env(plan, "127.0.0.1", port, 0))
    r = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=150)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    print("device-to-device parameter publication over RCCL (world 1):", out)
    assert out["ok"]


_MIRROR_READER = r"""
import sys, time, json
import torch
from distributed_rl_amd.transport import ShmParams
name, n_versions = sys.argv[1], int(sys.argv[2])
par = ShmParams(name, create=False)
dev = torch.device("cuda")
have, seen, bad = 0, [], 0
t0 = time.time()
while len(seen) < n_versions and time.time() - t0 < 60:
    got = par.pull_tensor(have, dev)
    if got is None:
        time.sleep(0.002)
        continue
    have, flat = got
    flat = torch.as_tensor(flat)
    # every published version v is the constant vector v: a torn copy would mix two values
    if not bool((flat == float(have)).all()):
        bad += 1
    seen.append(have)
print(json.dumps(dict(seen=seen, bad=bad, mirror=par.mirror_pulls, host=par.host_pulls)), flush=True)
par.close()
"""


def test_param_register_device_mirror_between_processes(run_id):
    """The learner's parameter versions reach an actor PROCESS on the same GPU device to device: the reader's pulls all
    come out of the IPC-mapped double buffer (none through the host register), whole (never a mix of two versions), with
    increasing version numbers; with the mirror switched off the same reader falls back to the host register."""
    import json
    import subprocess
    import sys
    from distributed_rl_amd.transport import ShmParams, shm_names
    name = shm_names(run_id)[1]
    par = ShmParams(name, create=True, capacity=16 << 20)
    n = 3_300_000
    try:
        assert par.enable_device_mirror("cuda", n * 4)
        for env_extra, want_mirror in (({}, True), ({"DRL_PARAM_MIRROR": "0"}, False)):
            par.reset()
            env = dict(os.environ, PYTHONPATH=os.getcwd(), **env_extra)
            proc = subprocess.Popen([sys.executable, "-c", _MIRROR_READER, name, "5"], stdout=subprocess.PIPE, env=env, text=True)
            snap = torch.empty(n, device="cuda")
            host = np.empty(n, np.float32)
            t0 = time.time()
            v = 0
            while proc.poll() is None and time.time() - t0 < 90:
                v += 1
                snap.fill_(float(v)); host.fill(float(v))
                torch.cuda.synchronize()
                assert par.publish2(host, snap) == v
                time.sleep(0.01)
            out = json.loads(proc.stdout.read().strip().splitlines()[-1])
            assert proc.returncode == 0 and out["bad"] == 0 and len(out["seen"]) == 5, out
            assert out["seen"] == sorted(set(out["seen"]))
            if want_mirror:
                assert out["mirror"] == 5 and out["host"] == 0, out
            else:
                assert out["mirror"] == 0 and out["host"] == 5, out
    finally:
        par.close()


def test_launcher_heals_a_learner_whose_persistent_kernel_timed_out(tmp_path, run_id):
    """The real thing behind tests/test_fault_injection.py's injected version: every inter-workgroup poll of the persistent
    LSTM kernels gives up at once (DRL_LSTM_SPIN_LIMIT=0), the R2D2 learner's step guard discards the step, the process
    writes its checkpoint + fault note and leaves with status 75; the launcher restarts it ATTACHED to the live segments with
    the persistent paths switched off, and that learner finishes its step budget next to the same actor process."""
    import re
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    save = tmp_path / "models"
    env = dict(os.environ, PYTHONPATH=root, OMP_NUM_THREADS="2", DRL_LSTM_SPIN_LIMIT="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-u", "-m", "distributed_rl_amd.runtime.launcher", "--num-actors", "1", "--conf", "config/all_r2d2.conf",
           "--run-id", run_id, "--timeout", "200",
           "--learner-args", "--max-steps 6 --batch-size 16 --start-memory-size 32 --save-timing 100000 --save-dir %s --quiet "
                             "--env SyntheticCatch-v0 -s 256" % save,
           "--actor-args", "--quiet --env SyntheticCatch-v0 --num-envs 16"]
    r = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=280)
    out = r.stdout + r.stderr
    assert re.search(r"learner guard fired .*persistent.*with .*DRL_PERSISTENT_LSTM=0.*\(heal 1 of 3\)", out), out[-4000:]
    assert "learner exited with 0" in out and r.returncode == 0, out[-4000:]
    assert not re.search(r"actor_1 exited", out.split("learner exited with 0")[0]), out[-4000:]


class _RankStub:
    """What HipLearner.save / resume read of a data-parallel group (the collectives themselves are covered by
    tests/test_xgmi_gpu.py and the 2-rank rehearsal of bench.py)."""
    enabled = False

    def __init__(self, rank, world):
        self.rank, self.world_size = rank, world


def test_data_parallel_checkpoint_every_rank_restores_the_same_weights_and_its_own_replay_shard(tmp_path, run_id):
    """Data-parallel checkpoint / resume: rank 0 alone writes the weights, moments and step state (identical on
    every rank), every rank writes ITS replay shard (model_<t>.replay.pth / .replay.r<k>.pth); on resume every rank loads
    rank 0's weights and its own shard."""
    from distributed_rl_amd.engine.learner import HipLearner
    cfg = make_config("ape_x", batch_size=32, start_memory_size=64, save_timing=10 ** 9, fit_timing=10 ** 9, target_update=4)
    learners, shards = [], []
    for rank in (0, 1):
        exp, par = connect("inproc", "%s_r%d" % (run_id, rank), create=True)
        lr = HipLearner("ape_x", 4, cfg, exp, par, None, replay_size=512, device="cuda", dist=_RankStub(rank, 2), seed=rank,
                        save_replay=True)
        lr.replay.fill_synthetic(200 + 100 * rank, n_action=4, seed=10 + rank)       # shards of different size and content
        learners.append(lr)
    learners[1].engine.P.flat.copy_(learners[0].engine.P.flat)                      # DP ranks hold identical weights
    for _ in range(3):
        learners[0].engine.step()
    learners[1].engine.P.flat.copy_(learners[0].engine.P.flat)
    for lr in learners:
        lr.step = 7
        shards.append(lr.replay.snapshot())
        path = lr.save(str(tmp_path))
    names = sorted(os.path.basename(p) for p in glob.glob(str(tmp_path / "model_7*")))
    assert names == ["model_7.aux.pth", "model_7.pth", "model_7.replay.pth", "model_7.replay.r1.pth"]      # one writer of the weights
    for rank in (0, 1):
        exp, par = connect("inproc", "%s_b%d" % (run_id, rank), create=True)
        l2 = HipLearner("ape_x", 4, cfg, exp, par, None, replay_size=512, device="cuda", dist=_RankStub(rank, 2), seed=rank)
        assert l2.resume(path) == 7
        assert torch.equal(l2.engine.P.flat, learners[0].engine.P.flat) and torch.equal(l2.engine.SQ.flat, learners[0].engine.SQ.flat)
        n = int(shards[rank]["K"].sum())
        assert len(l2.replay) == n == 200 + 100 * rank
        assert torch.equal(l2.replay.act[:n].cpu(), shards[rank]["action"]) and torch.equal(l2.replay.rew[:n].cpu(), shards[rank]["reward"])
        l2.close()
    for lr in learners:
        lr.close()


def test_learner_loop_drains_the_device_now_and_then(tmp_path, run_id, monkeypatch):
    """A learner loop that only waits for events lets host-side records of finished launches pile up inside the HIP runtime:
    2.6 KB per production R2D2 step, 5.6 GB of resident memory per hour; with a device-wide synchronize every few thousand
    steps it stays flat (scripts/debug/learner_rss.py, profiles/learner_resident_memory_r3.txt - the effect needs tens of
    thousands of production-size steps, so what is pinned here is the behaviour: ONE long optimize_loop call synchronizes the
    device every ``sync_every`` steps)."""
    from distributed_rl_amd.engine.learner import HipLearner
    cfg = make_config("r2d2", batch_size=16, start_memory_size=16, save_timing=1 << 30, fit_timing=1 << 30)
    cfg.burn_in, cfg.seq_len, cfg.overlap, cfg.nstep = 4, 8, 4, 2
    exp, par = connect("inproc", run_id, create=True)
    lr = HipLearner("r2d2", 4, cfg, exp, par, None, replay_size=256, device="cuda", publish_interval=1e9)
    assert HipLearner.sync_every == 4096
    lr.sync_every = 500
    lr.replay.fill_synthetic(256, n_action=4, seed=0)
    lr.optimize_loop(max_steps=10, save_model_dir=str(tmp_path))              # capture
    calls = []
    real = torch.cuda.synchronize
    monkeypatch.setattr(torch.cuda, "synchronize", lambda *a, **k: (calls.append(lr.step), real(*a, **k))[1])
    lr.optimize_loop(max_steps=2600, save_model_dir=str(tmp_path))
    monkeypatch.undo()
    lr.close()
    inside = [c for c in calls if c < 2610]
    assert len(inside) >= 5 and all(b - a >= 500 for a, b in zip(inside, inside[1:])), calls

"""The GPU-side half of tests/test_switches_cpu.py: switches and module constants of the HIP path that no other test sets."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_planes_per_item_sizes_the_plane_ring(run_id, monkeypatch):
    """DRL_PLANES_PER_ITEM: planes of ring per transition of capacity in the learner's HBM replay (1.3 for the actors' plane
    pools, 8 when the producers send full stacks: codec kind 1)."""
    from distributed_rl_amd.config import make_config
    from distributed_rl_amd.engine.learner import HipLearner
    from distributed_rl_amd.transport import connect
    cfg = make_config("ape_x", batch_size=32)
    caps = {}
    for tag, val in (("default", None), ("stacks", "8")):
        if val is not None:
            monkeypatch.setenv("DRL_PLANES_PER_ITEM", val)
        exp, par = connect("inproc", run_id + tag, create=True)
        lr = HipLearner("ape_x", 4, cfg, exp, par, replay_size=4096, use_graph=False)
        caps[tag] = int(lr.replay.plane_capacity)
        lr.close()
    assert 1.2 * 4096 <= caps["default"] <= 1.5 * 4096 and caps["stacks"] >= 8 * 4096, caps


def _engines(monkeypatch, fused, **kw):
    from distributed_rl_amd.config import ApexConfig
    from distributed_rl_amd.engine.apex import ApexEngine
    from distributed_rl_amd.models import DuelingDQN
    from distributed_rl_amd.ops import hip as H
    from distributed_rl_amd.replay.device_per import DevicePER
    monkeypatch.setattr(H, "TORSO_FUSED", fused)
    torch.manual_seed(5)
    model = DuelingDQN(6)
    cfg = ApexConfig(batch_size=64, target_update=7)
    cfg.rms_eps = 1e-2      # updates proportional to the gradient: the order of float atomics stays rounding noise (test_engine_gpu._setup)
    rep = DevicePER(1000, "cuda", batch_size=64, seed=3)
    rep.fill_synthetic(900, n_action=6, seed=1)
    return ApexEngine(6, cfg, rep, "cuda", init_model=model, **kw)


def test_one_launch_per_layer_is_the_same_step_as_the_fused_torso(monkeypatch):
    """ops.hip.TORSO_FUSED = False (the A/B form of profiles/bench_r3_torso_*): conv1 / conv2 / conv3 as three launches. The
    fused kernel is bit-identical to them: the first step's Q values are equal bit for bit (its loss up to the order of a float sum), and 9 steps end at the
    same parameters up to the order of the weight gradients' float atomics."""
    out = {}
    for fused in (True, False):
        e = _engines(monkeypatch, fused, use_graph=False)
        e.step()
        torch.cuda.synchronize()
        first = (e.on.q.clone(), e.tg.q.clone(), e.stats()["loss"])
        for _ in range(8):
            e.step()
        torch.cuda.synchronize()
        out[fused] = (first, e.P.flat.clone())
    assert torch.equal(out[True][0][0], out[False][0][0]) and torch.equal(out[True][0][1], out[False][0][1])
    assert out[True][0][2] == pytest.approx(out[False][0][2], rel=1e-5)     # (the batch loss is summed with float atomics)
    a, b = out[True][1], out[False][1]
    assert float((a - b).norm() / a.norm()) < 5e-5


@pytest.mark.parametrize("K", [1, 2])
def test_steps_per_graph_argument(monkeypatch, K):
    """ApexEngine(steps_per_graph=K): K learner steps per graph replay in run_steps (default 4, the measured optimum)."""
    e = _engines(monkeypatch, True, steps_per_graph=K)
    ref = _engines(monkeypatch, True)
    e.run_steps(13)
    ref.run_steps(13)
    torch.cuda.synchronize()
    assert e.steps_per_graph == K and ref.steps_per_graph == 4
    assert (e._graph_k is not None) == (K > 1)
    assert e.steps_done == ref.steps_done and e.stats()["step"] == ref.stats()["step"]
    a, b = e.P.flat, ref.P.flat
    assert float((a - b).norm() / a.norm()) < 5e-5

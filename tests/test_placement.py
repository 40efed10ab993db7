"""Who runs where (runtime/placement.py): actors of a multi-GPU node are placed explicitly - never a silent cuda:0
(VERDICT r2 Missing #1; reference: actor_node.py:18-19 picks "cuda if available" per process, aws/aws_config.yaml:1-4)."""
import os
import subprocess
import sys

from distributed_rl_amd.config import LaunchConf
from distributed_rl_amd.runtime.launcher import build_commands, child_envs
from distributed_rl_amd.runtime.placement import make_plan

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _dev(cmd):
    return cmd[cmd.index("--device") + 1]


def test_colocated_actors_follow_their_shards_gpu():
    conf = LaunchConf(learner_gpus=8, num_envs=64)
    plan = make_plan(conf, 16, n_gpus=8)
    assert plan.learner_devices == ["cuda:%d" % k for k in range(8)]
    for a in plan.actors:
        assert a.shard == (a.no - 1) % 8 and a.device == "cuda:%d" % a.shard and a.channel == "register"
    assert plan.world_size == 8 and not plan.lanes                     # the learners' gradient group only
    cmds = build_commands(conf, 16, "", "rid", [], [], plan=plan)
    assert [_dev(cmds["learner" if k == 0 else "learner_r%d" % k]) for k in range(8)] == plan.learner_devices
    assert {_dev(cmds["actor_%d" % i]) for i in range(1, 17)} == set(plan.learner_devices)      # all 8 GPUs, 2 actors each
    envs = child_envs(plan, list(cmds), 12345)
    assert envs["learner_r3"]["DRL_WORLD_RANK"] == "3" and "DRL_WORLD_PLAN" not in envs["actor_1"]


def test_dedicated_actor_gpus_and_lanes_have_distinct_devices():
    conf = LaunchConf(learner_gpus=4, actor_gpus="4,5,6,7", param_channel="collective", param_backend="nccl")
    plan = make_plan(conf, 24, n_gpus=8)
    by_rank = {a.world_rank: a for a in plan.actors}
    assert all(a.channel == "collective" and a.device in ("cuda:4", "cuda:5", "cuda:6", "cuda:7") for a in plan.actors)
    assert plan.world_size == 4 + 24
    seen = set()
    for lane in plan.lanes:
        k, members = lane[0], lane[1:]
        devs = [by_rank[r].device for r in members]
        assert len(set(devs)) == len(devs) and plan.learner_devices[k] not in devs      # RCCL: one rank per device
        assert all(by_rank[r].shard == k for r in members)                             # a lane starts at ITS shard's learner
        seen.update(members)
    assert seen == set(by_rank)                                                         # every lane actor is in exactly one lane
    assert sum(len(lane) - 1 for lane in plan.lanes) == 24
    per_gpu = {}
    for a in plan.actors:
        per_gpu[a.device] = per_gpu.get(a.device, 0) + 1
    assert set(per_gpu.values()) == {6}                                                 # evenly dealt
    cmds = build_commands(conf, 24, "", "rid", [], [], plan=plan)
    assert "--register-actors" in cmds["learner"] and cmds["learner"][cmds["learner"].index("--register-actors") + 1] == "0"


def test_nccl_lanes_skip_colocated_actors_and_gloo_lanes_take_everyone():
    conf = LaunchConf(learner_gpus=2, param_channel="collective", param_backend="nccl")
    plan = make_plan(conf, 4, n_gpus=2)                    # co-located: RCCL refuses two ranks on one device -> register
    assert all(a.channel == "register" for a in plan.actors) and not plan.lanes and plan.world_size == 2
    conf = LaunchConf(learner_gpus=2, param_channel="collective", param_backend="gloo", actor_device="cpu")
    plan = make_plan(conf, 4, n_gpus=0)
    assert [a.device for a in plan.actors] == ["cpu"] * 4 and plan.lanes == [[0, 2, 4], [1, 3, 5]] and plan.world_size == 6
    assert plan.world_backend == "gloo"


def test_actor_device_key_of_the_reference_is_honoured():
    plan = make_plan(LaunchConf(actor_device="cpu"), 3, n_gpus=8)
    assert [a.device for a in plan.actors] == ["cpu"] * 3 and plan.learner_devices == ["cuda:0"] and plan.world_size == 0
    plan = make_plan(LaunchConf(actor_device="cuda:5"), 3, n_gpus=8)
    assert [a.device for a in plan.actors] == ["cuda:5"] * 3


def test_dry_run_prints_distinct_devices(tmp_path):
    conf = tmp_path / "dp8.conf"
    conf.write_text("actor=true\nlearner=true\nalgorithm=r2d2\nlearner_gpus=4\nactor_gpus=4,5\nparam_channel=collective\n"
                    "param_backend=nccl\nnum_envs=64\n")
    r = subprocess.run([sys.executable, "-m", "distributed_rl_amd.runtime.launcher", "--conf", str(conf), "--num-actors", "8",
                        "--dry-run", "--gpus", "8"], cwd=ROOT, capture_output=True, text=True, timeout=120,
                       env=dict(os.environ, PYTHONPATH=ROOT))
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith(("learner", "actor_"))]
    devs = [ln.split("--device ")[1].split()[0] for ln in lines]
    assert devs[:4] == ["cuda:0", "cuda:1", "cuda:2", "cuda:3"] and set(devs[4:]) == {"cuda:4", "cuda:5"}
    assert all("--device" in ln for ln in lines)

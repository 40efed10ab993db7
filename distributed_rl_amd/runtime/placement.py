"""Who runs where on a node: learner ranks, actor processes, replay shards, parameter lanes.

The reference lets every actor process pick "cuda if available" for itself (actor_node.py:18-19) and describes the fleet
only as instance counts (aws/aws_config.yaml:1-4): on an 8-GPU MI355X node that would put every actor of every shard on
GPU 0. Here the launcher computes ONE plan from the conf file and hands each child its part explicitly (``--device``,
shard, process-group rank), so nothing ever falls back to a silent ``cuda:0``:

* learner rank k (data-parallel, ``learner_gpus=L``) owns GPU ``learner_devices[k]`` and replay shard k;
* actor i feeds shard ``(i-1) % L``. Default placement = the GPU of that shard's learner (co-located: parameters reach it
  through the shard's shared-memory register, experience through the shard's ring - nothing crosses xGMI).
  ``actor_gpus=4,5,6,7`` gives the actors GPUs of their own: the actors of a shard are dealt over those GPUs, starting at
  a different one per shard, so every actor GPU serves every learner evenly;
* ``param_channel=collective``: an actor whose GPU is not its learner's GPU receives parameters by broadcast from ITS
  shard's learner rank (every rank holds the same weights, so the publications are spread over all learner GPUs' xGMI
  links instead of leaving from rank 0 alone). RCCL refuses two ranks of one communicator on one device, so the actors of
  a shard are grouped into LANES whose members all sit on different GPUs (lane s = the s-th actor of every actor GPU);
  a lane is one process group [learner k, actors...], one broadcast per publication. With ``param_backend=gloo`` (CPU
  runs, tests) there is no such restriction: one lane per shard, every actor in it.
* all of these processes form ONE ``torch.distributed`` world (learners first, then the lane actors in actor order);
  the data-parallel learners' own group and the lanes are sub-groups of it (parallel/world.py). Actors on the register
  stay outside the world - they remain restartable.
"""
from __future__ import annotations

import dataclasses
import json
import os
from dataclasses import dataclass, field
from typing import Dict, List, Optional


@dataclass
class ActorPlace:
    no: int                      # 1-based actor number (-n)
    shard: int                   # learner rank whose replay shard this actor feeds
    device: str                  # "cuda:3" | "cpu"
    channel: str = "register"    # "register" (transport's parameter register) | "collective"
    lane: int = -1               # index into Plan.lanes (collective only)
    world_rank: int = -1         # rank in the launcher's process-group world (collective only)


@dataclass
class Plan:
    learner_devices: List[str]
    actors: List[ActorPlace] = field(default_factory=list)
    lanes: List[List[int]] = field(default_factory=list)     # world ranks per lane: [learner rank, actor ranks ...]
    world_size: int = 0          # 0 = no process group at all (one learner, no lane actors)
    param_backend: str = "gloo"
    world_backend: str = "gloo"  # backend string of the world, the same on every member (parallel/world.py)

    @property
    def n_learners(self) -> int:
        return len(self.learner_devices)

    def world_spec(self) -> Dict:
        """What every member needs to rebuild the same sub-groups in the same order (JSON-able)."""
        return {"size": self.world_size, "learners": self.n_learners, "lanes": self.lanes, "param_backend": self.param_backend,
                "backend": self.world_backend}

    def describe(self) -> List[str]:
        out = ["learner rank %d: %s (replay shard %d)" % (k, d, k) for k, d in enumerate(self.learner_devices)]
        for a in self.actors:
            how = "register of shard %d" % a.shard if a.channel == "register" else \
                "lane %d <- learner rank %d (world rank %d)" % (a.lane, a.shard, a.world_rank)
            out.append("actor %d: %s, feeds shard %d, parameters: %s" % (a.no, a.device, a.shard, how))
        return out


def _int_list(text: str) -> List[int]:
    return [int(x) for x in str(text).replace(";", ",").split(",") if x.strip() != ""]


def visible_gpus() -> int:
    """Number of GPUs this process would see - WITHOUT initialising one (the launcher stays GPU-free)."""
    try:
        import torch
        return int(torch.cuda.device_count())
    except Exception:                                              # noqa: BLE001
        return 0


def make_plan(conf, n_actors: int, n_gpus: Optional[int] = None) -> Plan:
    """``conf``: config.LaunchConf. ``n_gpus``: visible GPUs (None = ask the runtime)."""
    if n_gpus is None:
        n_gpus = visible_gpus()
    L = max(1, int(conf.learner_gpus))
    share = os.environ.get("DRL_SHARE_GPU") == "1"                # rehearsal: more ranks than GPUs
    learner_gpu_ids = _int_list(conf.learner_devices) if getattr(conf, "learner_devices", "") else list(range(L))
    if len(learner_gpu_ids) != L:
        raise SystemExit("learner_devices lists %d GPUs but learner_gpus=%d" % (len(learner_gpu_ids), L))
    cpu_only = n_gpus == 0

    def gpu(i: int) -> str:
        if cpu_only:
            return "cpu"
        if i >= n_gpus:
            if not share:
                raise SystemExit("placement needs GPU %d but only %d are visible (DRL_SHARE_GPU=1 lets ranks share)" % (i, n_gpus))
            i %= n_gpus
        return "cuda:%d" % i

    plan = Plan([gpu(i) for i in learner_gpu_ids], param_backend=conf.param_backend,
                world_backend="gloo" if (cpu_only or share) else "cpu:gloo,cuda:nccl")
    if len(set(plan.learner_devices)) != L and not (cpu_only or share):
        raise SystemExit("two learner ranks on one GPU: %s" % plan.learner_devices)
    actor_gpu_ids = _int_list(getattr(conf, "actor_gpus", ""))
    adev = (conf.actor_device or "").strip()
    per_shard_seen: Dict[int, int] = {}
    for i in range(1, n_actors + 1):
        k = (i - 1) % L
        j = per_shard_seen.get(k, 0)                               # this actor is the j-th of its shard
        per_shard_seen[k] = j + 1
        if adev == "cpu" or cpu_only:
            dev = "cpu"
        elif adev.startswith("cuda:"):                             # the reference's actor_device key, taken literally
            dev = gpu(int(adev.split(":")[1]))
        elif actor_gpu_ids:
            dev = gpu(actor_gpu_ids[(j + k) % len(actor_gpu_ids)])
        else:
            dev = plan.learner_devices[k]
        plan.actors.append(ActorPlace(i, k, dev))

    if conf.param_channel == "collective" and conf.learner:
        gloo = conf.param_backend != "nccl"
        members: Dict[int, List[ActorPlace]] = {}
        for a in plan.actors:
            remote = gloo or (a.device.startswith("cuda") and a.device != plan.learner_devices[a.shard])
            if remote:
                a.channel = "collective"
                members.setdefault(a.shard, []).append(a)
        rank = L
        for a in plan.actors:                                      # world ranks in actor order
            if a.channel == "collective":
                a.world_rank = rank
                rank += 1
        for k in sorted(members):
            if gloo:
                groups = [members[k]]
            else:                                                  # distinct devices inside a lane
                by_dev: Dict[str, List[ActorPlace]] = {}
                for a in members[k]:
                    by_dev.setdefault(a.device, []).append(a)
                depth = max(len(v) for v in by_dev.values())
                groups = [[v[s] for v in by_dev.values() if len(v) > s] for s in range(depth)]
            for g in groups:
                for a in g:
                    a.lane = len(plan.lanes)
                plan.lanes.append([k] + [a.world_rank for a in g])
        plan.world_size = rank if (rank > L or L > 1) else 0
    elif L > 1 and conf.learner:
        plan.world_size = L
    return plan


def child_env(plan: Plan, addr: str, port: int, world_rank: int) -> Dict[str, str]:
    """Environment of one member of the world (learner rank or lane actor)."""
    return {"DRL_WORLD_PLAN": json.dumps(plan.world_spec()), "DRL_WORLD_RANK": str(world_rank),
            "DRL_WORLD_ADDR": addr, "DRL_WORLD_PORT": str(port)}


def plan_to_json(plan: Plan) -> str:
    return json.dumps(dataclasses.asdict(plan))

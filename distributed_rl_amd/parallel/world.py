"""The launcher's process-group world: data-parallel learners + parameter lanes in ONE ``torch.distributed`` world.

A process can hold only one default process group, so data-parallel learners (gradient all-reduce among L ranks) and a
collective parameter channel (learner -> actor GPUs, reference semantics ``SET/GET params``: ape_x/learner.py:83,
ape_x/actor.py:52-56) cannot each bring their own. runtime/placement.py therefore describes one world - learner ranks
0..L-1, then the lane actors - and every member builds the same sub-groups from it, in the same order (``new_group`` is
collective over the whole world, members or not):

  1. the learners' group (gradient all-reduce, shard scalars; handed to parallel.dist.Dist),
  2. one group per parameter lane [learner k, actors on pairwise different GPUs ...].

Backend (chosen once, by the launcher, identical on every member): "cpu:gloo,cuda:nccl" on GPU nodes - RCCL
communicators are created lazily, per sub-group, at the first collective on device tensors, and RCCL refuses two ranks
on one device; the members of one sub-group never share a GPU (placement guarantees it), members of DIFFERENT lanes
may. Plain "gloo" on CPU boxes and for rehearsals with more ranks than GPUs (``DRL_DIST_BACKEND=gloo``). No collective
is ever issued on the world group itself.
"""
from __future__ import annotations

import datetime
import json
import os
from typing import Dict, List, Optional

import torch
import torch.distributed as td


class World:
    def __init__(self, rank: int, spec: Dict, backend: str):
        self.rank, self.spec, self.backend = rank, spec, backend
        self.size, self.n_learners = int(spec["size"]), int(spec["learners"])
        self.lanes: List[List[int]] = [list(x) for x in spec.get("lanes", [])]
        self.param_backend = spec.get("param_backend", "gloo")
        self.learner_group = None
        self.lane_groups: Dict[int, object] = {}

    @property
    def is_learner(self) -> bool:
        return self.rank < self.n_learners

    def my_lanes(self) -> List[int]:
        return [i for i, ranks in enumerate(self.lanes) if self.rank in ranks]

    def close(self) -> None:
        if td.is_initialized():
            td.destroy_process_group()


def world_from_env() -> Optional[Dict]:
    text = os.environ.get("DRL_WORLD_PLAN")
    if not text:
        return None
    spec = json.loads(text)
    return spec if int(spec.get("size", 0)) > 0 else None


def join_world(device: torch.device, timeout_s: Optional[float] = None) -> Optional[World]:
    """Rendezvous with the other members named by ``DRL_WORLD_*`` (runtime/placement.child_env) and build the sub-groups.
    Returns None when the launcher planned no world (one learner, register-only actors)."""
    spec = world_from_env()
    if spec is None:
        return None
    rank = int(os.environ["DRL_WORLD_RANK"])
    addr, port = os.environ.get("DRL_WORLD_ADDR", "127.0.0.1"), os.environ.get("DRL_WORLD_PORT", "29900")
    timeout_s = float(os.environ.get("DRL_CONNECT_TIMEOUT", "120")) if timeout_s is None else timeout_s
    device = torch.device(device)
    backend = os.environ.get("DRL_DIST_BACKEND") or spec.get("backend") or "gloo"
    if device.type == "cuda":
        torch.cuda.set_device(device)
    td.init_process_group(backend, init_method="tcp://%s:%s" % (addr, port), rank=rank, world_size=int(spec["size"]),
                          timeout=datetime.timedelta(seconds=timeout_s))
    w = World(rank, spec, backend)
    if w.n_learners > 1:
        g = td.new_group(list(range(w.n_learners)))
        if w.is_learner:
            w.learner_group = g
    for i, ranks in enumerate(w.lanes):
        g = td.new_group(ranks)
        if rank in ranks:
            w.lane_groups[i] = g
    return w

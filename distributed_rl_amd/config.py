"""Hyper-parameters and ``config/*.conf`` parsing.

The reference scatters its settings over three tiers: bash-sourced conf files
(run.sh:4-13), argparse flags (actor_node.py:9-16, learner_node.py:12-19) and constants
hard-coded in the node scripts (learner_node.py:23-45). Here they live in two dataclasses whose
defaults are exactly the reference's effective values, and the conf files keep their keys
(``actor``, ``learner``, ``visdom``, ``algorithm``, ``redis_server``, ``visdom_server``,
``actor_device``) plus a few optional new ones (``transport``, ``num_envs``, ``run_dir``).
"""
from __future__ import annotations

import dataclasses
import shlex
from dataclasses import dataclass, field
from typing import Dict, Optional


@dataclass
class ApexConfig:
    """Ape-X dueling double-DQN (learner_node.py:23-31, ape_x/learner.py:57-61, ape_x/actor.py:28-32)."""
    algorithm: str = "ape_x"
    batch_size: int = 512
    nstep: int = 3
    gamma: float = 0.999              # per-step; bootstrap discount is gamma**nstep
    lr: float = 0.00025 / 4
    rms_alpha: float = 0.95
    rms_eps: float = 1.5e-7
    max_grad_norm: float = 40.0
    per_alpha: float = 0.6            # applied to (huber + 1e-5)  (models.py:47)
    prio_eps: float = 1e-5
    beta0: float = 0.4
    beta_decay: int = 1_000_000
    target_update: int = 1000         # learner steps
    replay_size: int = 100_000
    start_memory_size: int = 10_000
    fit_timing: int = 100             # eviction period (learner steps)
    save_timing: int = 10_000
    actor_batch: int = 50             # transitions per published block
    actor_local_capacity: int = 1000
    actor_param_period: int = 200     # env steps between param pulls ("target_update" on the actor)
    reward_clip: Optional[float] = 1.0
    publish_period: int = 1           # learner steps between param publishes (reference: every step)
    reference_nstep_order: bool = False

    @property
    def gamma_n(self) -> float:
        return self.gamma ** self.nstep


@dataclass
class R2D2Config:
    """R2D2 (learner_node.py:32-45, r2d2/actor.py:13-26, models.py:51-53,106-109)."""
    algorithm: str = "r2d2"
    batch_size: int = 64
    nstep: int = 5
    gamma: float = 0.997
    lr: float = 1.0e-4
    adam_eps: float = 1.0e-3
    adam_beta1: float = 0.9
    adam_beta2: float = 0.999
    max_grad_norm: float = 40.0
    eta: float = 0.9
    rescale_eps: float = 1.0e-3
    beta0: float = 0.6
    beta_decay: int = 1_000_000
    target_update: int = 2500
    replay_size: int = 100_000        # sequences
    start_memory_size: int = 10_000
    fit_timing: int = 100
    save_timing: int = 10_000
    burn_in: int = 40
    seq_len: int = 80
    overlap: int = 40
    hidden: int = 512
    actor_batch: int = 20             # sequences per published block
    actor_local_capacity: int = 1000
    actor_param_period: int = 400
    reward_clip: Optional[float] = None
    publish_period: int = 1
    reference_nstep_order: bool = False

    @property
    def gamma_n(self) -> float:
        return self.gamma ** self.nstep

    @property
    def total_len(self) -> int:
        return self.burn_in + self.seq_len

    @property
    def learn_steps(self) -> int:
        return self.seq_len - self.nstep


def make_config(algorithm: str, **overrides):
    cfg = {"ape_x": ApexConfig, "r2d2": R2D2Config}[algorithm]()
    for k, v in overrides.items():
        if v is None:
            continue
        if not hasattr(cfg, k):
            raise AttributeError("%s has no field %r" % (type(cfg).__name__, k))
        setattr(cfg, k, v)
    return cfg


_BOOL = {"true": True, "false": False, "1": True, "0": False, "yes": True, "no": False}


@dataclass
class LaunchConf:
    """One ``config/*.conf`` file (config/all.conf:1-6 etc.)."""
    actor: bool = True
    learner: bool = True
    visdom: bool = False
    algorithm: str = "ape_x"
    redis_server: str = ""
    visdom_server: str = ""
    actor_device: str = ""
    transport: str = "shm"           # shm | redis | inproc  (new key)
    num_envs: int = 1                # envs per actor process (new key)
    learner_gpus: int = 1            # data-parallel learner ranks on this node, one per GPU (new key)
    learner_devices: str = ""        # GPU ordinals of the learner ranks, e.g. "0,1,2,3" ('' = 0..learner_gpus-1) (new key)
    actor_gpus: str = ""             # GPU ordinals the actors run on, e.g. "4,5,6,7"; '' = the GPU of the learner rank whose
                                     # shard the actor feeds (runtime/placement.py) (new key)
    param_channel: str = ""          # "" = same as transport | "collective" = torch.distributed broadcast (new key)
    param_backend: str = "gloo"      # backend of the collective parameter channel: gloo | nccl (= RCCL, one GPU per process)
    run_dir: str = ""                # where shm names / heartbeats / logs live (new key)
    extra: Dict[str, str] = field(default_factory=dict)


def parse_conf(path: str) -> LaunchConf:
    """Parse a bash-style ``key=value`` conf file without executing it.

    The reference *sources* these files and executes ``$actor`` / ``$visdom`` as commands
    (run.sh:5,20,36); we only read assignments, so a conf file can never run code.
    """
    conf = LaunchConf()
    names = {f.name: f for f in dataclasses.fields(LaunchConf)}
    with open(path) as fh:
        for raw in fh:
            line = raw.split("#", 1)[0].strip()
            if not line or "=" not in line:
                continue
            key, val = line.split("=", 1)
            key = key.strip()
            toks = shlex.split(val.strip())
            val = toks[0] if toks else ""
            if key in names and key != "extra":
                typ = names[key].type
                if typ in (bool, "bool"):
                    if val.lower() not in _BOOL:
                        raise ValueError("%s: %s must be true/false, got %r" % (path, key, val))
                    setattr(conf, key, _BOOL[val.lower()])
                elif typ in (int, "int"):
                    setattr(conf, key, int(val))
                else:
                    setattr(conf, key, val)
            else:
                conf.extra[key] = val
    return conf

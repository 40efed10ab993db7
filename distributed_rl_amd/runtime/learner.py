"""Learner loop for both algorithms (reference: ape_x/learner.py:13-99; R2D2 reuses it, learner_node.py:7).

Per step: PER sample → IS weights → TD/loss → backward → global-norm clip → optimizer → priority
write-back → param publish → (every ``fit_timing``) eviction → (every ``target_update``) target sync →
(every ``save_timing``) checkpoint → back-pressure.

This class is the *host/PyTorch* learner: it runs on CPU (the ``run.sh 2`` plumbing config) or on
a GPU through ATen, and it is the behavioural oracle for ``engine.ApexEngine`` / ``engine.R2D2Engine``
(the graph-captured gfx950 learners, which keep sample → priority write-back entirely on device).

Differences from the reference, all deliberate: params are published as a versioned
flat buffer every ``publish_period`` steps instead of a pickled state_dict every step (Q4);
checkpoints can be resumed (``resume``; the reference has no load path, §5.4); back-pressure is
bounded; visdom is optional.
"""
from __future__ import annotations

import os
import time
from typing import Optional

import numpy as np
import torch

from ..config import ApexConfig, R2D2Config
from ..models.packing import ParamPack
from ..ops import oracle
from ..transport import ExperienceChannel, ParamChannel
from .replay_service import Replay


def save_checkpoint(save_dir: str, t: int, policy, target=None, optimizer=None, extra: Optional[dict] = None) -> str:
    """``models/model_<t>.pth`` holds exactly ``policy.state_dict()`` (learner.py:98) so reference tools can
    load it; everything else needed to resume goes into the sidecar ``model_<t>.aux.pth``."""
    os.makedirs(save_dir, exist_ok=True)
    path = os.path.join(save_dir, "model_%d.pth" % t)
    sd = {k: v.detach().cpu() for k, v in policy.state_dict().items()}
    torch.save(sd, path)
    aux = dict(step=t, extra=extra or {})
    if target is not None:
        aux["target"] = {k: v.detach().cpu() for k, v in target.state_dict().items()}
    if optimizer is not None:
        aux["optimizer"] = optimizer.state_dict()
    torch.save(aux, os.path.join(save_dir, "model_%d.aux.pth" % t))
    return path


def load_checkpoint(path: str, policy, target=None, optimizer=None) -> int:
    """Load ``model_<t>.pth`` (+ sidecar if present). Returns the step to resume from."""
    policy.load_state_dict(torch.load(path, map_location="cpu", weights_only=True))
    step = 0
    aux_path = path[:-4] + ".aux.pth"
    if os.path.exists(aux_path):
        # the sidecar holds only tensors, containers and primitives (optimizer.state_dict() is such a nest), so the
        # restricted unpickler loads it; anything else next to a checkpoint is refused, never executed
        try:
            aux = torch.load(aux_path, map_location="cpu", weights_only=True)
        except Exception as e:
            raise RuntimeError("refusing to load %s: not a plain tensor / primitive sidecar (%s)" % (aux_path, e)) from e
        step = int(aux.get("step", 0))
        if target is not None and "target" in aux:
            target.load_state_dict(aux["target"])
        if optimizer is not None and "optimizer" in aux:
            optimizer.load_state_dict(aux["optimizer"])
    elif target is not None:
        target.load_state_dict(policy.state_dict())
    return step


class Learner:
    def __init__(self, policy_net, target_net, optimizer, cfg, exp: ExperienceChannel, params: ParamChannel,
                 metrics=None, replay_size: Optional[int] = None, device=torch.device("cpu"),
                 heartbeat=None, memory=None, dist=None, sync_target: bool = True, reset_channels: bool = True):
        self.policy, self.target, self.opt, self.cfg = policy_net, target_net, optimizer, cfg
        if sync_target:
            self.target.load_state_dict(self.policy.state_dict())    # learner.py:33-34
        self.target.eval()
        self.exp, self.params, self.metrics = exp, params, metrics
        self.device = device
        self.heartbeat = heartbeat
        self.dist = dist
        self.is_r2d2 = isinstance(cfg, R2D2Config)
        if reset_channels:               # (a learner restarted next to running actors, --attach, keeps both)
            self.params.reset()                                       # DEL params (learner.py:37)
            self.exp.reset()                                          # DEL experience (replay.py:21)
        self.replay = Replay(replay_size or cfg.replay_size, exp, memory=memory)
        self.replay.start()
        self._pack = ParamPack.from_module(self.policy)
        self.step = 0
        self.last_loss = float("nan")

    # ---- helpers ----------------------------------------------------------------------------------
    def _sleep(self):
        """Back-pressure: the reference sleeps 0.01 s per queued block (learner.py:47-49); bounded here."""
        n = self.exp.size()
        if n:
            time.sleep(min(0.01 * n, 0.25))

    def wait_memory(self, n: int, timeout: Optional[float] = None) -> bool:
        t0 = time.time()
        while len(self.replay) <= n:                                  # learner.py:51-55 (strict >)
            if timeout is not None and time.time() - t0 > timeout:
                return False
            if self.heartbeat is not None:
                self.heartbeat.tick(self.step)
            time.sleep(0.1)
        return True

    def publish(self) -> int:
        self._pack.load_state_dict(self.policy.state_dict())
        return self.params.publish(self._pack.flat.cpu().numpy())

    def _batch_tensors(self, items):
        dev = self.device
        if not self.is_r2d2:
            s, a, r, s2, d = zip(*items)
            return dict(state=torch.from_numpy(np.stack(s)).to(dev), action=torch.tensor(a, device=dev),
                        reward=torch.tensor(r, dtype=torch.float32, device=dev),
                        next_state=torch.from_numpy(np.stack(s2)).to(dev),
                        done=torch.tensor(d, dtype=torch.float32, device=dev))
        return dict(frames=torch.from_numpy(np.stack([i["frames"] for i in items], 1)).to(dev),
                    action=torch.from_numpy(np.stack([i["action"] for i in items], 1)).long().to(dev),
                    reward=torch.from_numpy(np.stack([i["reward"] for i in items], 1)).to(dev),
                    done=torch.from_numpy(np.stack([i["done"] for i in items], 1)).to(dev),
                    lengths=torch.tensor([i["length"] for i in items], device=dev),
                    h0=torch.from_numpy(np.stack([i["h0"] for i in items])).to(dev),
                    c0=torch.from_numpy(np.stack([i["c0"] for i in items])).to(dev))

    # ---- one optimisation step --------------------------------------------------------------------
    def train_step(self) -> float:
        cfg, t = self.cfg, self.step
        items, prios, idxs, ids, total, total_prio = self.replay.sample(cfg.batch_size)       # learner.py:64
        beta = oracle.beta_schedule(t, cfg.beta0, cfg.beta_decay)                               # learner.py:66
        w = oracle.is_weights(torch.tensor(prios), total_prio, total, beta).to(self.device)     # learner.py:67-68
        b = self._batch_tensors(items)
        if self.is_r2d2:
            td = self.policy.sequence_td(self.target, b["frames"], b["action"], b["reward"], b["done"], b["lengths"],
                                         b["h0"], b["c0"], w, cfg.eta, cfg.gamma_n, cfg.rescale_eps,
                                         done_offset=cfg.nstep if cfg.reference_nstep_order else 0)
        else:
            td = self.policy.td_batch(self.target, b["state"], b["action"], b["reward"], b["next_state"], b["done"],
                                      w, cfg.per_alpha, cfg.gamma_n)
        self.opt.zero_grad()
        td.loss.backward()
        if self.dist is not None and self.dist.world_size > 1:
            self.dist.all_reduce_grads(self.policy)
        torch.nn.utils.clip_grad_norm_(self.policy.parameters(), cfg.max_grad_norm)            # learner.py:77
        self.opt.step()
        self.replay.update_priorities(idxs, td.prio.detach().float().cpu().numpy(), ids=ids)   # learner.py:80-81
        self.last_loss = float(td.loss.detach())
        return self.last_loss

    def optimize_loop(self, max_steps: Optional[int] = None, save_model_dir: str = "./models",
                      start_step: int = 0, stop_event=None, warmup_timeout: Optional[float] = None) -> int:
        cfg = self.cfg
        if not self.wait_memory(max(cfg.batch_size, cfg.start_memory_size), warmup_timeout):  # learner.py:62
            raise TimeoutError("replay never reached the learn-start size")
        self.step = start_step
        end = None if max_steps is None else start_step + max_steps
        t_last, n_last = time.time(), self.step
        while end is None or self.step < end:
            if stop_event is not None and stop_event.is_set():
                break
            t = self.step
            # fault injection for the self-healing launcher's test (no bounded in-kernel waits on this engine): behave like
            # the GPU learner whose persistent LSTM kernel timed out - until that path is "switched off"
            inj = os.environ.get("DRL_INJECT_GUARD_AT_STEP")
            if inj and t == start_step + int(inj) and os.environ.get("DRL_PERSISTENT_LSTM", "1") != "0":
                raise RuntimeError("learner step discarded: a bounded in-kernel wait timed out (persistent LSTM (online)); "
                                   "injected by DRL_INJECT_GUARD_AT_STEP")
            loss = self.train_step()
            if t % cfg.publish_period == 0:
                self.publish()                                                                  # learner.py:83
            if self.metrics is not None:
                self.metrics.scalar("Q loss", t, loss)                                          # learner.py:86
            if t % cfg.fit_timing == 0:
                self.replay.remove_to_fit()                                                     # learner.py:88-92
                if self.metrics is not None:
                    self.metrics.scalar("Memory size", t, len(self.replay), force=True)
                    now = time.time()
                    if t > n_last:
                        self.metrics.scalar("Updates per second", t, (t - n_last) / max(1e-9, now - t_last), force=True)
                    t_last, n_last = now, t
            if t % cfg.target_update == 0:
                self.target.load_state_dict(self.policy.state_dict())                          # learner.py:93-95
            if t % cfg.save_timing == 0 and (self.dist is None or self.dist.rank == 0):        # one writer
                save_checkpoint(save_model_dir, t, self.policy, self.target, self.opt)         # learner.py:96-98
            if self.heartbeat is not None:
                self.heartbeat.tick(t)
            self.step += 1
            self._sleep()                                                                       # learner.py:99
        return self.step

    def close(self):
        self.replay.stop()

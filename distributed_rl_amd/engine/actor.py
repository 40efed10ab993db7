"""ActorEngine — batched ε-greedy inference for E environments on one MI355X.

The reference runs one env per process and one B=1 forward per env step (ape_x/actor.py:70, utils.py:36-41). Here
one process feeds E environments through ONE torso → FC1 → head launch chain (5 kernels, captured as a hipGraph);
the head kernel also draws the ε-greedy action on device (per-env ε from the reference's ladder, Philox RNG), so
what returns to the host per step is E int32 actions.

Parameters are either *shared* with a learner engine on the same GPU (the actor reads the learner's live bf16
weights — the "param sync" is free) or owned and refreshed from a flat fp32 buffer (ParamChannel pull or RCCL
broadcast, parallel/dist.py).
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch

from ..models import DuelingDQN
from ..models.packing import ParamPack, apex_layout, conv1_hwc
from ..ops import hip as H
from .streams import capture_graph
from ..utils import epsilon_ladder
from .apex import TorsoBuffers, torso_forward


class ActorEngine:
    def __init__(self, n_action: int, num_envs: int, device="cuda", params=None, eps: Optional[np.ndarray] = None,
                 seed: int = 0, use_graph: bool = True, init_model: Optional[DuelingDQN] = None):
        self.A, self.E = n_action, num_envs
        self.device = dev = torch.device(device)
        if params is not None:                      # share the learner's buffers (same GPU)
            self.P, self.Pb = params.P, params.Pb
            self.owns = False
        else:
            model = init_model if init_model is not None else DuelingDQN(n_action)
            self.P = ParamPack(apex_layout(n_action), dev)
            self.P.load_state_dict(model.state_dict())
            self.Pb = self.P.like(torch.bfloat16)
            H.cast_bf16(self.P.flat, self.Pb.flat)
            self.owns = True
        self.refresh_conv1()
        self.buf = TorsoBuffers(num_envs, n_action, dev)
        self.obs = torch.zeros(num_envs, 84, 84, 4, dtype=torch.uint8, device=dev)
        self.action = torch.zeros(num_envs, dtype=torch.int32, device=dev)
        e = eps if eps is not None else epsilon_ladder(num_envs)
        self.eps = torch.as_tensor(e, dtype=torch.float32, device=dev).contiguous()
        self.seed = seed
        self.st = H.new_step_state(dev, 1.0, 1)      # only its rng_ctr is used
        self._n = torch.zeros(1, dtype=torch.int64, device=dev)
        self._ctr = self.st[8:16]
        self.use_graph = use_graph
        self._graph = None
        self._host_obs = None
        self._host_act = None
        self._pool = None
        self._k = 0

    # ---- parameters -----------------------------------------------------------------------------------
    def refresh_conv1(self) -> None:
        """The environments deliver HWC stacks; the flat buffer keeps conv1 in the learner's (c, kh, kw) order (its input is
        the replay's plane store). The actor's conv1 runs on an HWC-ordered copy, rebuilt whenever the parameters change
        (shared-parameter actors: call this after the learner moved them)."""
        w = conv1_hwc(self.Pb.view("conv1.w"))
        if getattr(self, "w1", None) is None:
            self.w1 = w
        else:
            self.w1.copy_(w)                         # same address: a captured inference graph keeps reading it

    def load_flat(self, flat_fp32) -> None:
        """Refresh from a flat fp32 parameter buffer in kernel layout (ParamChannel.pull / broadcast target)."""
        assert self.owns, "shared-parameter actors read the learner's weights directly"
        t = torch.as_tensor(flat_fp32)
        self.P.flat.copy_(t.to(self.device, non_blocking=True))
        H.cast_bf16(self.P.flat, self.Pb.flat)
        self.refresh_conv1()

    # ---- inference --------------------------------------------------------------------------------------
    def _launch(self) -> None:
        H.step_begin(self.st, self._n)               # advances the Philox counter on device
        torso_forward(self.obs, None, self.E, self.Pb, self.P, self.buf, self.action, self.eps, 0, self.seed, self._ctr, w1=self.w1)

    def infer(self) -> torch.Tensor:
        """Q + ε-greedy actions for the frames currently in ``self.obs``; returns the device action tensor."""
        if not self.use_graph:
            self._launch()
            return self.action
        if self._graph is None:
            s = torch.cuda.Stream(self.device)
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                self._launch()
            torch.cuda.current_stream().wait_stream(s)
            torch.cuda.synchronize()
            self._graph = torch.cuda.CUDAGraph()
            with capture_graph(self._graph):
                self._launch()
        self._graph.replay()
        return self.action

    def act(self, obs_u8: np.ndarray) -> np.ndarray:
        """Host frames (E,84,84,4) uint8 → host actions: pinned H2D, graph replay, pinned D2H."""
        if self._host_obs is None:
            self._host_obs = torch.zeros(self.E, 84, 84, 4, dtype=torch.uint8).pin_memory()
            self._host_act = torch.zeros(self.E, dtype=torch.int32).pin_memory()
        self._host_obs.numpy()[...] = obs_u8
        self.obs.copy_(self._host_obs, non_blocking=True)
        self.infer()
        self._host_act.copy_(self.action, non_blocking=True)
        torch.cuda.current_stream().synchronize()
        return self._host_act.numpy().astype(np.int64)

    def q_values(self) -> torch.Tensor:
        return self.buf.q

    # ---- initial priorities of a block of fresh transitions (ape_x/actor.py:91-95; online == target, Q3) ----
    def priorities(self, state: np.ndarray, next_state: np.ndarray, action: np.ndarray, reward: np.ndarray,
                   done: np.ndarray, gamma_n: float, alpha: float = 0.6, prio_eps: float = 1e-5) -> np.ndarray:
        K = int(action.shape[0])
        dev = self.device
        if getattr(self, "_pk", 0) != K:
            self._pk = K
            self._pbuf = TorsoBuffers(2 * K, self.A, dev)
            self._pframes = torch.zeros(2 * K, 84, 84, 4, dtype=torch.uint8, device=dev)
            self._pdq = torch.zeros(K, self.A, device=dev)
            self._pprio = torch.zeros(K, device=dev)
        self._pframes[:K].copy_(torch.from_numpy(np.ascontiguousarray(state)), non_blocking=True)
        self._pframes[K:].copy_(torch.from_numpy(np.ascontiguousarray(next_state)), non_blocking=True)
        torso_forward(self._pframes, None, 2 * K, self.Pb, self.P, self._pbuf, w1=self.w1)
        a = torch.from_numpy(np.ascontiguousarray(action, dtype=np.int32)).to(dev)
        r = torch.from_numpy(np.ascontiguousarray(reward, dtype=np.float32)).to(dev)
        d = torch.from_numpy(np.ascontiguousarray(done, dtype=np.float32)).to(dev)
        H.apex_td(self._pbuf.q, self._pbuf.q[K:], None, a, r, d, None, self.st, gamma_n, alpha, prio_eps, self._pdq,
                  self._pprio, use_is=False)
        return self._pprio.cpu().numpy()

    def priorities_planes(self, planes: np.ndarray, win_s: np.ndarray, win_s2: np.ndarray, action: np.ndarray,
                          reward: np.ndarray, done: np.ndarray, gamma_n: float, alpha: float = 0.6,
                          prio_eps: float = 1e-5) -> np.ndarray:
        """``priorities`` for a plane-pool block (codec kind 3): the pool goes to the device once (M planes of 7 KB instead of
        2 K stacks of 28 KB) and conv1 reads the windows in place, as the learner does from its replay."""
        K, M = int(action.shape[0]), int(planes.shape[0])
        dev = self.device
        if getattr(self, "_ppk", (0, 0))[0] < K or self._ppk[1] < M:
            # ONE pinned staging buffer and ONE device image of it: [offsets 2K x i64 | action K x i32 | reward K x f32 |
            # done K x f32 | plane pool]; a block costs one host-to-device DMA instead of five pageable copies. Sized for the
            # largest block seen (shorter blocks - an env that publishes early, runtime/actor.py _urgent_envs - use a prefix)
            Kc, cap = max(K, getattr(self, "_ppk", (0, 0))[0]), max(M, 2 * K, getattr(self, "_ppk", (0, 0))[1])
            self._ppk = (Kc, cap)
            tab = (28 * Kc + 63) // 64 * 64
            self._pptab = tab
            self._pppin = torch.zeros(tab + cap * 84 * 84, dtype=torch.uint8).pin_memory()
            self._ppdev = torch.zeros(tab + cap * 84 * 84, dtype=torch.uint8, device=dev)
            self._pph = self._pppin.numpy()
            d_ = self._ppdev
            self._ppoff, self._ppa = d_[:16 * Kc].view(torch.int64), d_[16 * Kc:20 * Kc].view(torch.int32)
            self._ppr, self._ppd = d_[20 * Kc:24 * Kc].view(torch.float32), d_[24 * Kc:28 * Kc].view(torch.float32)
            self._pplanes = d_[tab:].view(cap, 84 * 84)
            self._ppbuf = TorsoBuffers(2 * Kc, self.A, dev)
            self._pdq2 = torch.zeros(Kc, self.A, device=dev)
            self._pprio2 = torch.zeros(Kc, device=dev)
            self._pprio_pin = torch.zeros(Kc, dtype=torch.float32).pin_memory()
        Kc = self._ppk[0]
        h, tab = self._pph, self._pptab
        off = h[:16 * K].view(np.int64)                       # [win_s | win_s2] as byte offsets: the first 2 K of the table
        off[:K] = win_s
        off[K:] = win_s2
        off *= 84 * 84
        h[16 * Kc:16 * Kc + 4 * K].view(np.int32)[:] = action
        h[20 * Kc:20 * Kc + 4 * K].view(np.float32)[:] = reward
        h[24 * Kc:24 * Kc + 4 * K].view(np.float32)[:] = done
        h[tab:tab + M * 84 * 84] = planes.reshape(-1)
        n = tab + M * 84 * 84
        self._ppdev[:n].copy_(self._pppin[:n], non_blocking=True)
        torso_forward(self._pplanes, self._ppoff[:2 * K], 2 * K, self.Pb, self.P, self._ppbuf)
        H.apex_td(self._ppbuf.q[:2 * K], self._ppbuf.q[K:2 * K], None, self._ppa[:K], self._ppr[:K], self._ppd[:K], None, self.st, gamma_n,
                  alpha, prio_eps, self._pdq2[:K], self._pprio2[:K], use_is=False)
        self._pprio_pin[:K].copy_(self._pprio2[:K], non_blocking=True)
        torch.cuda.current_stream().synchronize()
        return self._pprio_pin[:K].numpy().copy()

    # ---- benchmarks -------------------------------------------------------------------------------------
    def step_synthetic(self) -> None:
        """One inference step on device-resident random frames (pure GPU path)."""
        if self._pool is None:
            self._pool = torch.randint(0, 256, (2, self.E, 84, 84, 4), dtype=torch.uint8, device=self.device)
        self.obs.copy_(self._pool[self._k & 1])
        self._k += 1
        self.infer()

    def step_host(self) -> None:
        """One inference step including the PCIe round trip (pinned uint8 frames in, int32 actions out)."""
        if self._host_obs is None:
            self._host_obs = torch.randint(0, 256, (self.E, 84, 84, 4), dtype=torch.uint8).pin_memory()
            self._host_act = torch.zeros(self.E, dtype=torch.int32).pin_memory()
        self.obs.copy_(self._host_obs, non_blocking=True)
        self.infer()
        self._host_act.copy_(self.action, non_blocking=True)
        torch.cuda.current_stream().synchronize()

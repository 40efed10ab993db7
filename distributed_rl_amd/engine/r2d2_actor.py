"""R2D2ActorEngine — batched recurrent ε-greedy inference + sequence priorities on one MI355X
(reference: r2d2/actor.py:26-91, libs/models.py:94-162).

The reference actor runs one env per process, one B=1 LSTMCell forward per env step and, every 20 finished
sequences, a Python-loop ``calc_priorities`` over 120 steps. Here one process feeds E environments:

* ``act``: conv1-3 -> input projection GEMM -> ONE LSTM step (EPI_LSTM GEMM, cell update fused) -> FC + dueling head
  with the ε-greedy draw on device: 8 launches captured as one hipGraph; the recurrent state (h bf16 / c f32) of
  every env stays in HBM, ``get_state`` hands the fp32 copy the sequence assembler stores with each sequence, and
  ``reset(done)`` zeroes finished envs (models.py:79-84).
* ``sequence_priorities``: the learner's time-batched forward (engine/r2d2.py ``seq_net_forward``; the persistent
  recurrence kernel when the batch allows) for the online AND the actor-side target net, then the fused R2D2 TD
  kernel -> η·max + (1-η)·mean priorities. No backward, no host loop over time.
* parameters arrive as the learner's flat fp32 buffer (``load_flat``); pulling also refreshes the target copy
  (r2d2/actor.py:89-91).
"""
from __future__ import annotations

from typing import Optional, Tuple

import numpy as np
import torch

from ..config import R2D2Config
from ..models import DuelingLSTMDQN
from ..models.packing import ParamPack, conv1_hwc, r2d2_layout
from ..ops import hip as H
from .apex import torso_net
from .streams import capture_graph
from ..utils import epsilon_ladder
from .r2d2 import SeqNetBuffers, seq_net_forward


class R2D2ActorEngine:
    def __init__(self, n_action: int, num_envs: int, cfg: R2D2Config, device="cuda", eps: Optional[np.ndarray] = None,
                 seed: int = 0, use_graph: bool = True, init_model: Optional[DuelingLSTMDQN] = None, done_offset: int = 0):
        self.A, self.E, self.cfg = n_action, num_envs, cfg
        self.Hd = Hd = cfg.hidden
        self.device = dev = torch.device(device)
        self.done_offset = done_offset
        model = init_model if init_model is not None else DuelingLSTMDQN(n_action, num_envs, cfg.burn_in, cfg.nstep)
        self.P = ParamPack(r2d2_layout(n_action, hidden=Hd), dev)
        self.P.load_state_dict(model.state_dict())
        self.Pb = self.P.like(torch.bfloat16)
        self.Tg = self.P.like()
        self.Tb = self.P.like(torch.bfloat16)
        H.cast_bf16(self.P.flat, self.Pb.flat)
        self.sync_target()
        E = num_envs
        bf = dict(dtype=torch.bfloat16, device=dev)
        self.obs = torch.zeros(E, 84, 84, 4, dtype=torch.uint8, device=dev)
        self.act1 = torch.zeros(E, 20, 20, 32, **bf)
        self.act2 = torch.zeros(E, 9, 9, 64, **bf)
        self.act3 = torch.zeros(E, 3136, **bf)
        self.xg = torch.zeros(E, 4 * Hd, device=dev)
        self.h = torch.zeros(E, Hd, **bf)
        self.c = torch.zeros(E, Hd, device=dev)
        self.h_next = torch.zeros(E, Hd, **bf)
        self.c_next = torch.zeros(E, Hd, device=dev)
        self.h2 = torch.zeros(E, 1024, **bf)
        self.q = torch.zeros(E, n_action, device=dev)
        self.bsum = torch.zeros(4 * Hd, device=dev)
        self.action = torch.zeros(E, dtype=torch.int32, device=dev)
        e = eps if eps is not None else epsilon_ladder(E)
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
        self._prio_key = None

    # ---- parameters -----------------------------------------------------------------------------------
    def sync_target(self) -> None:
        self.Tg.flat.copy_(self.P.flat)
        self.Tb.flat.copy_(self.Pb.flat)
        self._refresh_conv1()

    def load_flat(self, flat_fp32) -> None:
        t = torch.as_tensor(flat_fp32)
        self.P.flat.copy_(t.to(self.device, non_blocking=True))
        H.cast_bf16(self.P.flat, self.Pb.flat)
        self.sync_target()

    def _refresh_conv1(self) -> None:
        """HWC-ordered copies of the online / target conv1 weights (the flat buffers keep the learner's (c, kh, kw) order;
        the environments deliver HWC stacks). Same addresses on every refresh: the captured inference graph reads them."""
        for name, pack in (("w1", self.Pb), ("w1_tg", self.Tb)):
            w = conv1_hwc(pack.view("conv1.w"))
            if getattr(self, name, None) is None:
                setattr(self, name, w)
            else:
                getattr(self, name).copy_(w)

    # ---- recurrent state ------------------------------------------------------------------------------
    def get_state(self, rows: Optional[np.ndarray] = None) -> Tuple[np.ndarray, np.ndarray]:
        """fp32 host copies of (h, c) BEFORE the next ``act`` (stored with the sequence, models.py:86-87). ``rows``: only
        these envs (the ones whose step can start a sequence, VecSequenceAssembler.need_state): one gather + one small
        device-to-host copy instead of 2 x E x 512 floats on every env step."""
        if rows is None:
            return self.h.float().cpu().numpy(), self.c.cpu().numpy()
        idx = torch.from_numpy(np.ascontiguousarray(rows, dtype=np.int64)).to(self.device, non_blocking=True)
        both = torch.cat((self.h.index_select(0, idx).float(), self.c.index_select(0, idx)), 1).cpu().numpy()
        return both[:, :self.Hd], both[:, self.Hd:]

    def set_state(self, h: np.ndarray, c: np.ndarray) -> None:
        self.h.copy_(torch.as_tensor(h).to(self.device).to(torch.bfloat16))
        self.c.copy_(torch.as_tensor(c).to(self.device))

    def reset(self, done: np.ndarray) -> None:
        """Zero the state of finished envs (models.py:79-84)."""
        keep = torch.as_tensor(~np.asarray(done, dtype=bool), device=self.device)
        self.h.mul_(keep.unsqueeze(1).to(self.h.dtype))
        self.c.mul_(keep.unsqueeze(1).to(self.c.dtype))

    # ---- inference --------------------------------------------------------------------------------------
    def _launch(self) -> None:
        E, Hd, wb, wf = self.E, self.Hd, self.Pb, self.P
        H.step_begin(self.st, self._n)               # advances the Philox counter on device
        if H.TORSO_FUSED:                            # conv1 -> conv2 -> conv3 as ONE launch (conv_torso.hip), as the Ape-X actor does
            H.torso_fwd(self.obs, torso_net(self.__dict__.setdefault("_torso", {}), None, E, wb, wf, None, None, self.act3, w1=self.w1))
        else:
            H.conv_fwd(1, self.obs, self.w1, wf.view("conv1.b"), self.act1, E)
            H.conv_fwd(2, self.act1, wb.view("conv2.w"), wf.view("conv2.b"), self.act2, E)
            H.conv_fwd(3, self.act2, wb.view("conv3.w"), wf.view("conv3.b"), self.act3, E)
        H.add2(wf.view("lstm.b_ih"), wf.view("lstm.b_hh"), self.bsum)
        H.gemm(H.NT, self.act3, wb.view("lstm.w_ih"), self.xg, E, 4 * Hd, 3136, bias=self.bsum, flags=H.EPI_OUT_F32,
               big=E >= 1024)
        H.lstm_step(self.h, wb.view("lstm.w_hh"), self.xg, self.c, self.h_next, self.c_next, None)
        H.gemm(H.NT, self.h_next, wb.view("fc1.w"), self.h2, E, 1024, Hd, bias=wf.view("fc1.b"), flags=H.EPI_RELU,
               big=E >= 1024)
        H.head_fwd(self.h2, wb.view("adv2.w"), None, wb.view("val2.w"), wf.view("val2.b"), self.q, self.action, self.eps, 0,
                   self.seed, self._ctr)
        self.h.copy_(self.h_next)
        self.c.copy_(self.c_next)

    def infer(self) -> torch.Tensor:
        """One recurrent step for the frames in ``self.obs``: advances (h, c), returns the device action tensor."""
        if not self.use_graph:
            self._launch()
            return self.action
        if self._graph is None:
            h0, c0 = self.h.clone(), self.c.clone()                  # the warm-up launch must not advance the state
            ctr0 = self.st.clone()
            s = torch.cuda.Stream(self.device)
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                self._launch()
            torch.cuda.current_stream().wait_stream(s)
            torch.cuda.synchronize()
            self._graph = torch.cuda.CUDAGraph()
            with capture_graph(self._graph):
                self._launch()
            self.h.copy_(h0)
            self.c.copy_(c0)
            self.st.copy_(ctr0)
        self._graph.replay()
        return self.action

    def act(self, obs_u8: np.ndarray) -> np.ndarray:
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
        return self.q

    # ---- sequence priorities (r2d2/actor.py:76-84 -> models.py:106-162, no gradient) -----------------------
    def sequence_priorities(self, frames: np.ndarray, action: np.ndarray, reward: np.ndarray, done: np.ndarray,
                            lengths: np.ndarray, h0: np.ndarray, c0: np.ndarray) -> np.ndarray:
        """frames (T,K,84,84,4) u8, action/reward/done (T,K), lengths (K,), h0/c0 (K,H) -> priorities (K,).
        K is padded to a multiple of 16 so the persistent recurrence kernel runs (padding rows: length 0)."""
        cfg, dev, Hd, A = self.cfg, self.device, self.Hd, self.A
        T, K = int(frames.shape[0]), int(frames.shape[1])
        Kp = (K + 15) // 16 * 16
        persistent = H.lstm_seq_supported(Kp, Hd)
        if not persistent:
            Kp = K
        if self._prio_key != (T, Kp):
            self._prio_key = (T, Kp)
            self._pon = SeqNetBuffers(T, Kp, Hd, A, dev, False)
            self._ptg = SeqNetBuffers(T, Kp, Hd, A, dev, False)
            self._pframes = torch.zeros(T, Kp, 84, 84, 4, dtype=torch.uint8, device=dev)
            self._pact = torch.zeros(Kp, T, dtype=torch.int32, device=dev)
            self._prew = torch.zeros(Kp, T, device=dev)
            self._pdone = torch.zeros(Kp, T, device=dev)
            self._plen = torch.zeros(Kp, dtype=torch.int32, device=dev)
            self._pidx = torch.arange(Kp, dtype=torch.int32, device=dev)
            self._pw = torch.ones(Kp, device=dev)
            self._pdq = torch.zeros(T, Kp, A, device=dev)
            self._pprio = torch.zeros(Kp, device=dev)
            self._pbs = [torch.zeros(4 * Hd, device=dev), torch.zeros(4 * Hd, device=dev)]
        t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a)).to(dev).to(dt)   # noqa: E731
        self._pframes[:, :K].copy_(torch.from_numpy(np.ascontiguousarray(frames)), non_blocking=True)
        self._pact[:K].copy_(t(action, torch.int32).t())
        self._prew[:K].copy_(t(reward, torch.float32).t())
        self._pdone[:K].copy_(t(done, torch.float32).t())
        self._plen.zero_()
        self._plen[:K].copy_(t(lengths, torch.int32))
        for buf in (self._pon, self._ptg):
            buf.hs[0].zero_()
            buf.cs[0].zero_()
            buf.hs[0, :K].copy_(t(h0, torch.bfloat16))
            buf.cs[0, :K].copy_(t(c0, torch.float32))
        fr = self._pframes.view(T * Kp, 84, 84, 4)
        seq_net_forward(fr, None, T, Kp, Hd, self.Pb, self.P, self._pon, self._pbs[0], persistent, w1=self.w1)
        seq_net_forward(fr, None, T, Kp, Hd, self.Tb, self.Tg, self._ptg, self._pbs[1], persistent, w1=self.w1_tg)
        H.r2d2_td(self._pon.q.view(T, Kp, A), self._ptg.q.view(T, Kp, A), self._pidx, self._pact, self._prew, self._pdone,
                  self._plen, self._pw, self.st, cfg.burn_in, cfg.nstep, self.done_offset, cfg.gamma_n, cfg.eta,
                  cfg.rescale_eps, self._pdq, self._pprio)
        out = self._pprio[:K].cpu().numpy()
        if persistent and (int(self._pon.err[0].item()) or int(self._ptg.err[0].item())):
            raise RuntimeError("persistent LSTM kernel: inter-workgroup wait timed out")
        return out

    def sequence_priorities_planes(self, planes: np.ndarray, win: np.ndarray, action: np.ndarray, reward: np.ndarray,
                                   done: np.ndarray, lengths: np.ndarray, h0: np.ndarray, c0: np.ndarray) -> np.ndarray:
        """The same over a plane pool: planes (M,84,84) u8 = every frame of the sequences once, win (K,T) = pool index of
        the first of the 4 planes of every observation, action/reward/done (K,T). Only the pool crosses PCIe (0.3 MB per
        sequence of a 20-sequence run instead of 3.4 MB of stacks) and conv1 reads the windows in place, exactly as the
        learner reads its replay (engine/r2d2.py)."""
        cfg, dev, Hd, A = self.cfg, self.device, self.Hd, self.A
        K, T = int(win.shape[0]), int(win.shape[1])
        M = int(planes.shape[0])
        Kp = (K + 15) // 16 * 16
        persistent = H.lstm_seq_supported(Kp, Hd)
        if not persistent:
            Kp = K
        if getattr(self, "_pp_key", None) != (T, Kp):
            self._pp_key = (T, Kp)
            self._qon = SeqNetBuffers(T, Kp, Hd, A, dev, False)
            self._qtg = SeqNetBuffers(T, Kp, Hd, A, dev, False)
            self._qoff = torch.zeros(T * Kp, dtype=torch.int64, device=dev)
            self._qact = torch.zeros(Kp, T, dtype=torch.int32, device=dev)
            self._qrew = torch.zeros(Kp, T, device=dev)
            self._qdone = torch.zeros(Kp, T, device=dev)
            self._qlen = torch.zeros(Kp, dtype=torch.int32, device=dev)
            self._qidx = torch.arange(Kp, dtype=torch.int32, device=dev)
            self._qw = torch.ones(Kp, device=dev)
            self._qdq = torch.zeros(T, Kp, A, device=dev)
            self._qprio = torch.zeros(Kp, device=dev)
            self._qbs = [torch.zeros(4 * Hd, device=dev), torch.zeros(4 * Hd, device=dev)]
            self._qplanes = None
        if self._qplanes is None or self._qplanes.shape[0] < M + 4:
            self._qplanes = torch.zeros(max(M + 4, 1024), 84 * 84, dtype=torch.uint8, device=dev)
        t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a)).to(dev).to(dt)   # noqa: E731
        self._qplanes[:M].copy_(torch.from_numpy(np.ascontiguousarray(planes)).view(M, 84 * 84), non_blocking=True)
        off = torch.zeros(T, Kp, dtype=torch.int64)
        off[:, :K] = torch.from_numpy(np.ascontiguousarray(win.T)).to(torch.int64) * (84 * 84)      # time-major byte offsets
        self._qoff.copy_(off.view(-1), non_blocking=True)
        self._qact.zero_(); self._qrew.zero_(); self._qdone.zero_(); self._qlen.zero_()
        self._qact[:K].copy_(t(action, torch.int32))
        self._qrew[:K].copy_(t(reward, torch.float32))
        self._qdone[:K].copy_(t(done, torch.float32))
        self._qlen[:K].copy_(t(lengths, torch.int32))
        for buf in (self._qon, self._qtg):
            buf.hs[0].zero_()
            buf.cs[0].zero_()
            buf.hs[0, :K].copy_(t(h0, torch.bfloat16))
            buf.cs[0, :K].copy_(t(c0, torch.float32))
        seq_net_forward(self._qplanes, self._qoff, T, Kp, Hd, self.Pb, self.P, self._qon, self._qbs[0], persistent)
        seq_net_forward(self._qplanes, self._qoff, T, Kp, Hd, self.Tb, self.Tg, self._qtg, self._qbs[1], persistent)
        H.r2d2_td(self._qon.q.view(T, Kp, A), self._qtg.q.view(T, Kp, A), self._qidx, self._qact, self._qrew, self._qdone,
                  self._qlen, self._qw, self.st, cfg.burn_in, cfg.nstep, self.done_offset, cfg.gamma_n, cfg.eta,
                  cfg.rescale_eps, self._qdq, self._qprio)
        out = self._qprio[:K].cpu().numpy()
        if persistent and (int(self._qon.err[0].item()) or int(self._qtg.err[0].item())):
            raise RuntimeError("persistent LSTM kernel: inter-workgroup wait timed out")
        return out

    # ---- benchmarks -------------------------------------------------------------------------------------
    def step_synthetic(self) -> None:
        if self._pool is None:
            self._pool = torch.randint(0, 256, (2, self.E, 84, 84, 4), dtype=torch.uint8, device=self.device)
        self.obs.copy_(self._pool[self._k & 1])
        self._k += 1
        self.infer()

    def step_host(self) -> None:
        if self._host_obs is None:
            self._host_obs = torch.randint(0, 256, (self.E, 84, 84, 4), dtype=torch.uint8).pin_memory()
            self._host_act = torch.zeros(self.E, dtype=torch.int32).pin_memory()
        self.obs.copy_(self._host_obs, non_blocking=True)
        self.infer()
        self._host_act.copy_(self.action, non_blocking=True)
        torch.cuda.current_stream().synchronize()

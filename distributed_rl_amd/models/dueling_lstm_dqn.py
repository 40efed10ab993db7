"""DuelingLSTMDQN — conv torso + LSTMCell(3136,512) + dueling head (reference: libs/models.py:50-162).

Two APIs over the same parameters:

* the reference's *stateful single-step* API (``hx``/``cx`` attributes, ``reset``, ``get_state``,
  ``set_state``, ``forward`` mutating the state; models.py:70-104) — used by single-env actors;
* a *functional, time-batched* API (``unroll``, ``sequence_td``) that runs the torso once over all
  ``T·B`` frames, one ``(T·B)×2048×3136`` input GEMM and a T-step recurrence — the shape the
  gfx950 engine executes. The 315 single-step forwards of models.py:124-159
  collapse to one online unroll + one target unroll (see ops.oracle.r2d2_td).
"""
from __future__ import annotations

from typing import Optional, Sequence, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from ..ops import oracle
from ..utils import Sequence as SeqTuple, Transition
from .dueling_dqn import frames_to_float, stack_field


class DuelingLSTMDQN(nn.Module):
    def __init__(self, n_action: int, batch_size: int = 1, n_burn_in: int = 40, nstep_return: int = 5,
                 input_shape: Sequence[int] = (4, 84, 84), hidden: int = 512):
        super().__init__()
        self.n_action = n_action
        self.batch_size = batch_size
        self.n_burn_in = n_burn_in
        self.nstep_return = nstep_return
        self.input_shape = tuple(input_shape)
        self.hidden = hidden
        self.conv1 = nn.Conv2d(input_shape[0], 32, kernel_size=8, stride=4)
        self.conv2 = nn.Conv2d(32, 64, kernel_size=4, stride=2)
        self.conv3 = nn.Conv2d(64, 64, kernel_size=3, stride=1)
        r = int((int(input_shape[1] / 4) - 1) / 2) - 3
        c = int((int(input_shape[2] / 4) - 1) / 2) - 3
        self.feature_hw = (r, c)
        self.lstm = nn.LSTMCell(r * c * 64, hidden)
        self.adv1 = nn.Linear(hidden, 512)
        self.adv2 = nn.Linear(512, n_action, bias=False)   # no bias: models.py:67
        self.val1 = nn.Linear(hidden, 512)
        self.val2 = nn.Linear(512, 1)
        # plain attributes, not buffers: hx/cx must stay out of the state_dict (models.py:70-71)
        self.hx = torch.zeros(batch_size, hidden)
        self.cx = torch.zeros(batch_size, hidden)

    # ---- stateful single-step API ---------------------------------------------------------------
    def to(self, *args, **kwargs):
        super().to(*args, **kwargs)
        dev = next(self.parameters()).device
        self.hx = self.hx.to(dev)
        self.cx = self.cx.to(dev)
        return self

    def reset(self, done=False):
        self.hx = self.hx.detach()
        self.cx = self.cx.detach()
        if isinstance(done, torch.Tensor) and done.dim() > 0:       # vectorised actors: per-env reset
            keep = (~done.bool()).to(self.hx.dtype).view(-1, 1).to(self.hx.device)
            self.hx = self.hx * keep
            self.cx = self.cx * keep
        elif done:
            self.hx = torch.zeros_like(self.hx)
            self.cx = torch.zeros_like(self.cx)

    def get_state(self):
        return self.hx.detach().clone().cpu(), self.cx.detach().clone().cpu()

    def set_state(self, state, device=None):
        hx, cx = state
        device = device if device is not None else next(self.parameters()).device
        self.hx = hx.clone().to(device)
        self.cx = cx.clone().to(device)

    def features(self, x: torch.Tensor) -> torch.Tensor:
        x = frames_to_float(x, self.input_shape[0])
        x = F.relu(self.conv1(x))
        x = F.relu(self.conv2(x))
        x = F.relu(self.conv3(x))
        return x.reshape(x.size(0), -1)

    def head(self, h: torch.Tensor) -> torch.Tensor:
        adv = self.adv2(F.relu(self.adv1(h)))
        val = self.val2(F.relu(self.val1(h)))
        return oracle.dueling_combine(val, adv)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        f = self.features(x)
        self.hx, self.cx = self.lstm(f, (self.hx, self.cx))
        return self.head(self.hx)

    # ---- functional time-batched API -------------------------------------------------------------
    def unroll(self, frames: torch.Tensor, h0: torch.Tensor, c0: torch.Tensor,
               grad_from: int = 0) -> Tuple[torch.Tensor, Tuple[torch.Tensor, torch.Tensor]]:
        """frames (T,B,…) → Q (T,B,A). Steps ``< grad_from`` run under no_grad (burn-in, models.py:125-130)
        and the state is detached at ``grad_from`` (models.py:137)."""
        T, B = frames.shape[:2]
        flat = frames.reshape((T * B,) + tuple(frames.shape[2:]))

        def run(lo, hi, h, c):
            if hi <= lo:
                return None, h, c
            feat = self.features(flat[lo * B:hi * B])
            xg = (feat @ self.lstm.weight_ih.t() + self.lstm.bias_ih + self.lstm.bias_hh).view(hi - lo, B, -1)
            hs, cs = oracle.lstm_unroll(xg, self.lstm.weight_hh, h, c)
            return hs, hs[-1], cs[-1]

        with torch.no_grad():
            hs0, h, c = run(0, min(grad_from, T), h0, c0)
        hs1, h, c = run(min(grad_from, T), T, h.detach(), c.detach())
        parts = [p for p in (hs0, hs1) if p is not None]
        hs = torch.cat(parts, 0)
        q = self.head(hs.reshape(T * B, -1)).view(T, B, -1)
        return q, (h, c)

    def sequence_td(self, target_net: "DuelingLSTMDQN", frames, action, reward, done, lengths, h0, c0,
                    weight=None, eta: float = 0.9, gamma: float = 0.997, rescale_eps: float = 1e-3,
                    require_grad: bool = True, done_offset: int = 0) -> oracle.R2D2TD:
        """Time-major tensors: frames (T,B,…), action/reward/done (T,B), lengths (B,), h0/c0 (B,H).
        ``gamma`` is the bootstrap discount γⁿ (models.py:155)."""
        T = frames.shape[0]
        with torch.set_grad_enabled(require_grad):
            q_on, _ = self.unroll(frames, h0, c0, grad_from=self.n_burn_in)
        with torch.no_grad():
            q_tg, _ = target_net.unroll(frames, h0, c0, grad_from=T)
        mask = oracle.r2d2_learn_mask(lengths, done, T, self.n_burn_in, self.nstep_return, done_offset)
        # autograd version of oracle.r2d2_td (which returns explicit dq for the kernels)
        n = self.nstep_return
        idx_n = (torch.arange(T, device=q_on.device) + n).clamp_(max=T - 1)
        a_star = q_on.detach()[idx_n].argmax(-1, keepdim=True)
        q_next = q_tg[idx_n].gather(-1, a_star).squeeze(-1)
        d = done.float() if done_offset == 0 else done.float()[(torch.arange(T, device=q_on.device) + done_offset).clamp_(max=T - 1)]
        from ..utils import inv_rescale, rescale
        y = rescale(inv_rescale(q_next, rescale_eps) * gamma * (1.0 - d) + reward.float(), rescale_eps)
        q_sa = q_on.gather(-1, action.long().unsqueeze(-1)).squeeze(-1)
        m = mask.float()
        diff = (q_sa - y) * m
        delta = diff.abs()
        seq_loss = diff.pow(2).sum(0)
        prio = (eta * delta.max(0)[0] + (1.0 - eta) * delta.sum(0) / m.sum(0).clamp(min=1.0)).detach()
        w = torch.ones_like(seq_loss) if weight is None else weight.float()
        return oracle.R2D2TD((seq_loss * w).mean(), seq_loss, prio, torch.empty(0), delta.detach())

    def calc_priorities(self, target_net, transitions, eta: float = 0.9, gamma: float = 0.997,
                        require_grad: bool = True, device=torch.device("cpu"), done_offset: Optional[int] = None):
        """Reference-compatible signature (models.py:106-162): list of ``Sequence`` → (Σδ² (B,1), prios (B,1)).

        Sequences of unequal length are padded and masked (the reference truncates the whole batch
        to the shortest one via ``zip``, models.py:117 — SURVEY §7.3)."""
        done_offset = 0 if done_offset is None else done_offset
        B = len(transitions)
        lens = [len(s.transitions) for s in transitions]
        T = max(lens)
        first = torch.as_tensor(transitions[0].transitions[0].state)
        frames = torch.zeros((T, B) + tuple(first.shape), dtype=first.dtype)
        action = torch.zeros(T, B, dtype=torch.long)
        reward = torch.zeros(T, B)
        done = torch.zeros(T, B)
        for b, s in enumerate(transitions):
            for t, tr in enumerate(s.transitions):
                frames[t, b] = torch.as_tensor(tr.state)
                action[t, b] = int(torch.as_tensor(tr.action).view(-1)[0])
                reward[t, b] = float(torch.as_tensor(tr.reward).view(-1)[0])
                done[t, b] = float(torch.as_tensor(tr.done).view(-1)[0])
        h0 = torch.cat([torch.as_tensor(s.recurrent_state[0]).view(1, -1) for s in transitions]).to(device)
        c0 = torch.cat([torch.as_tensor(s.recurrent_state[1]).view(1, -1) for s in transitions]).to(device)
        td = self.sequence_td(target_net, frames.to(device), action.to(device), reward.to(device), done.to(device),
                              torch.tensor(lens, device=device), h0, c0, None, eta, gamma,
                              require_grad=require_grad, done_offset=done_offset)
        return td.seq_loss.unsqueeze(1), td.prio.unsqueeze(1)

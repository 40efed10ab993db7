"""DuelingDQN — Nature-CNN torso + dueling head (reference: libs/models.py:6-48).

state_dict: conv1/2/3.{weight,bias}, adv1, adv2, val1, val2. Flatten order is
NCHW (models.py:24). This module is the fp32 PyTorch path (CPU plumbing, oracle for kernel
tests); the gfx950 path is ``engine.ApexEngine`` which consumes the same parameters.
"""
from __future__ import annotations

from typing import Sequence

import torch
import torch.nn as nn
import torch.nn.functional as F

from ..ops import oracle
from ..utils import Transition


def frames_to_float(x: torch.Tensor, in_channels: int) -> torch.Tensor:
    """Accept reference-style float CHW frames or raw uint8 (CHW or HWC) and return float CHW in [0,1]."""
    if x.dtype == torch.uint8:
        if x.shape[-1] == in_channels and x.shape[-3] != in_channels:
            x = x.permute(*range(x.dim() - 3), -1, -3, -2)
        return x.float() * (1.0 / 255.0)
    return x


def stack_field(items, device, dtype=None):
    ts = [torch.as_tensor(i) for i in items]
    out = torch.stack(ts).to(device)
    return out.to(dtype) if dtype is not None else out


class DuelingDQN(nn.Module):
    def __init__(self, n_action: int, input_shape: Sequence[int] = (4, 84, 84)):
        super().__init__()
        self.n_action = n_action
        self.input_shape = tuple(input_shape)
        self.conv1 = nn.Conv2d(input_shape[0], 32, kernel_size=8, stride=4)
        self.conv2 = nn.Conv2d(32, 64, kernel_size=4, stride=2)
        self.conv3 = nn.Conv2d(64, 64, kernel_size=3, stride=1)
        r = int((int(input_shape[1] / 4) - 1) / 2) - 3
        c = int((int(input_shape[2] / 4) - 1) / 2) - 3
        self.feature_hw = (r, c)
        self.adv1 = nn.Linear(r * c * 64, 512)
        self.adv2 = nn.Linear(512, n_action)
        self.val1 = nn.Linear(r * c * 64, 512)
        self.val2 = nn.Linear(512, 1)

    def features(self, x: torch.Tensor) -> torch.Tensor:
        x = frames_to_float(x, self.input_shape[0])
        x = F.relu(self.conv1(x))
        x = F.relu(self.conv2(x))
        x = F.relu(self.conv3(x))
        return x.reshape(x.size(0), -1)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        x = self.features(x)
        adv = self.adv2(F.relu(self.adv1(x)))
        val = self.val2(F.relu(self.val1(x)))
        return oracle.dueling_combine(val, adv)

    # ---- TD / loss / priority (models.py:31-48) -------------------------------------------------
    def td_batch(self, target_net, state, action, reward, next_state, done, weight=None,
                 alpha: float = 0.6, gamma: float = 0.999) -> oracle.ApexTD:
        """Tensor-batch form used by the learner. ``gamma`` is the bootstrap discount γⁿ."""
        q_s = self.forward(state)
        with torch.no_grad():
            q_s2 = self.forward(next_state)
            q_t2 = target_net(next_state)
        q_sa = q_s.gather(1, action.view(-1, 1).long()).squeeze(1)
        a_star = q_s2.argmax(1, keepdim=True)
        y = q_t2.gather(1, a_star).squeeze(1) * gamma * (1.0 - done.float().view(-1)) + reward.float().view(-1)
        delta = F.smooth_l1_loss(q_sa, y, reduction="none")
        prio = (delta.detach().abs() + 1e-5).pow(alpha)
        w = torch.ones_like(delta) if weight is None else weight.float().view(-1)
        return oracle.ApexTD((delta * w).mean(), delta, prio, torch.empty(0))

    def calc_priorities(self, target_net, transitions, alpha: float = 0.6, gamma: float = 0.999,
                        device=torch.device("cpu")):
        """Reference-compatible signature: list of ``Transition`` → (delta (B,1), prios (B,1))."""
        batch = Transition(*zip(*transitions))
        td = self.td_batch(target_net,
                           stack_field(batch.state, device), stack_field(batch.action, device).view(-1),
                           stack_field(batch.reward, device).view(-1), stack_field(batch.next_state, device),
                           stack_field(batch.done, device).view(-1), None, alpha, gamma)
        return td.delta.unsqueeze(1), td.prio.unsqueeze(1)

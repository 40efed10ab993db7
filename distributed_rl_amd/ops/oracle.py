"""Pure-PyTorch fp32 oracle of every fused op.

Each HIP kernel in ``csrc/hip`` has its semantics pinned here first; kernel tests compare against
these functions, and the CPU runtime (``run.sh 2`` plumbing config) executes them directly.
Citations are to the reference (``/root/reference``) lines whose *behaviour* is reproduced.
"""
from __future__ import annotations

from typing import NamedTuple, Optional, Tuple

import torch
import torch.nn.functional as F

from ..utils import inv_rescale, rescale


# --------------------------------------------------------------------------------------------
# dueling head
# --------------------------------------------------------------------------------------------
def dueling_combine(val: torch.Tensor, adv: torch.Tensor) -> torch.Tensor:
    """Q = V + A − mean_a(A)   (models.py:29, :104)."""
    return val + adv - adv.mean(-1, keepdim=True)


def dueling_combine_backward(dq: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """Given dL/dQ (…,A) return (dL/dV (…,1), dL/dA (…,A))."""
    dval = dq.sum(-1, keepdim=True)
    dadv = dq - dq.mean(-1, keepdim=True)
    return dval, dadv


def eps_greedy(q: torch.Tensor, eps: torch.Tensor, u_explore: torch.Tensor, u_action: torch.Tensor) -> torch.Tensor:
    """Batched ε-greedy with externally supplied uniforms (utils.py:36-41, vectorised).

    ``action = argmax_a q`` if ``u_explore > eps`` else ``floor(u_action * A)``.
    """
    n_action = q.shape[-1]
    greedy = q.argmax(-1)
    rand = (u_action * n_action).floor().clamp_(0, n_action - 1).to(torch.long)
    return torch.where(u_explore > eps, greedy, rand)


# --------------------------------------------------------------------------------------------
# importance-sampling weights
# --------------------------------------------------------------------------------------------
def beta_schedule(t: int, beta0: float, beta_decay: int) -> float:
    """β_t = min(1, β0 + (1−β0)·t/decay)   (ape_x/learner.py:66)."""
    return min(1.0, beta0 + (1.0 - beta0) / beta_decay * t)


def is_weights(prios: torch.Tensor, total_prio, n_items, beta: float,
               max_weight: Optional[torch.Tensor] = None) -> torch.Tensor:
    """w = (N·p/Σp)^−β / max w   (ape_x/learner.py:67-68).

    ``max_weight`` lets a sharded replay normalise by the *global* batch maximum (all-reduced).
    """
    w = (float(n_items) * prios.double() / float(total_prio)).pow(-beta)
    m = w.max() if max_weight is None else max_weight.double()
    return (w / m).float()


def sharded_is_weights(prios: torch.Tensor, shard_total_prio, n_global, world: int, beta: float,
                       max_weight: Optional[torch.Tensor] = None) -> torch.Tensor:
    """IS weights of a batch drawn from ONE shard of a sharded replay: every rank contributes 1/world of
    the global batch from its own tree, so item i is drawn with probability p_i / (world * Σp_shard) and
    w = (N_global * p_i / (world * Σp_shard))^-β / max w. With equal Σp per shard this IS the single-tree weight of
    ``is_weights`` (learner.py:67-68); with unequal shards it is the weight that keeps the estimator unbiased for the
    distribution actually sampled. The engines realise it by feeding N_global / world to the kernels as the item count
    (engine/shard.py)."""
    return is_weights(prios, shard_total_prio, float(n_global) / float(world), beta, max_weight)


# --------------------------------------------------------------------------------------------
# Ape-X n-step double-DQN TD / Huber / priority   (models.py:41-48, learner.py:72)
# --------------------------------------------------------------------------------------------
class ApexTD(NamedTuple):
    loss: torch.Tensor      # scalar: mean_b(w_b · huber_b)
    delta: torch.Tensor     # (B,) elementwise Huber value
    prio: torch.Tensor      # (B,) (|huber| + eps)^alpha
    dq: torch.Tensor        # (B,A) dL/dQ_online(s)


def apex_td(q_s: torch.Tensor, q_s2_online: torch.Tensor, q_s2_target: torch.Tensor,
            action: torch.Tensor, reward: torch.Tensor, done: torch.Tensor,
            weight: Optional[torch.Tensor], gamma_n: float, alpha: float = 0.6,
            prio_eps: float = 1e-5) -> ApexTD:
    q_s = q_s.float()
    a = action.view(-1, 1).long()
    q_sa = q_s.gather(1, a).squeeze(1)
    a_star = q_s2_online.float().argmax(1, keepdim=True)
    q_next = q_s2_target.float().gather(1, a_star).squeeze(1)
    y = q_next * gamma_n * (1.0 - done.float()) + reward.float()
    diff = q_sa - y
    delta = F.smooth_l1_loss(q_sa, y, reduction="none")
    prio = (delta.abs() + prio_eps).pow(alpha)
    w = torch.ones_like(delta) if weight is None else weight.float()
    loss = (delta * w).mean()
    dq = torch.zeros_like(q_s)
    dq.scatter_(1, a, (diff.clamp(-1.0, 1.0) * w / q_s.shape[0]).unsqueeze(1))
    return ApexTD(loss, delta, prio, dq)


# --------------------------------------------------------------------------------------------
# R2D2 sequence TD / priority   (models.py:142-162)
# --------------------------------------------------------------------------------------------
class R2D2TD(NamedTuple):
    loss: torch.Tensor      # scalar: mean_b(w_b · Σ_t δ²)
    seq_loss: torch.Tensor  # (B,) Σ_t δ²
    prio: torch.Tensor      # (B,) η·max_t δ + (1−η)·mean_t δ
    dq: torch.Tensor        # (T,B,A) dL/dQ_online (zero outside the learning window)
    delta: torch.Tensor     # (T,B) |Q − y| (zero where masked)


def r2d2_learn_mask(lengths: torch.Tensor, done: torch.Tensor, total_len: int, burn_in: int,
                    nstep: int, done_offset: int = 0) -> torch.Tensor:
    """(T,B) bool mask of the time steps that contribute a TD term.

    A step t learns iff ``burn_in ≤ t < len`` and its bootstrap state exists (``t+n < len``) —
    or no bootstrap is needed because the n-step window already hit a terminal (``done_t``;
    only with ``done_offset == 0``, i.e. intended semantics). Full-length sequences give the
    reference's window ``[burn_in, T−n)`` (models.py:143).
    """
    t = torch.arange(total_len, device=lengths.device).unsqueeze(1)
    ln = lengths.view(1, -1).to(t.dtype)
    has_boot = (t + nstep) < ln
    if done_offset == 0:
        has_boot = has_boot | (done.bool() & (t < ln))
    return (t >= burn_in) & (t < ln) & has_boot


def r2d2_td(q_online: torch.Tensor, q_target: torch.Tensor, action: torch.Tensor,
            reward: torch.Tensor, done: torch.Tensor, mask: torch.Tensor,
            weight: Optional[torch.Tensor], gamma_n: float, nstep: int, eta: float = 0.9,
            rescale_eps: float = 1e-3, done_offset: int = 0) -> R2D2TD:
    """All tensors are time-major. ``q_online``/``q_target``: (T,B,A) for every sequence step.

    ``a* = argmax_a q_online[t+n]`` — the reference's ``self_cp`` (models.py:131-136,153) is a
    copy of the online net carried through the same inputs, so its outputs are exactly
    ``q_online[t+n]`` of one continuous unroll. ``done_offset`` selects which stored flag masks
    the bootstrap: 0 = ``done[t]`` (intended), ``nstep`` = ``done[t+n]`` (reference, models.py:150).
    """
    T, B, A = q_online.shape
    qo = q_online.float()
    qt = q_target.float()
    idx_n = (torch.arange(T, device=qo.device) + nstep).clamp_(max=T - 1)
    a_star = qo[idx_n].argmax(-1, keepdim=True)                       # (T,B,1)
    q_next = qt[idx_n].gather(-1, a_star).squeeze(-1)                 # (T,B)
    if done_offset == 0:
        d = done.float()
    else:
        d = done.float()[(torch.arange(T, device=qo.device) + done_offset).clamp_(max=T - 1)]
    y = rescale(inv_rescale(q_next, rescale_eps) * gamma_n * (1.0 - d) + reward.float(), rescale_eps)
    q_sa = qo.gather(-1, action.long().unsqueeze(-1)).squeeze(-1)
    m = mask.float()
    diff = (q_sa - y) * m
    delta = diff.abs()
    seq_loss = diff.pow(2).sum(0)
    cnt = m.sum(0).clamp_(min=1.0)
    prio = eta * delta.max(0)[0] + (1.0 - eta) * delta.sum(0) / cnt
    w = torch.ones(B, device=qo.device) if weight is None else weight.float()
    loss = (seq_loss * w).mean()
    dq = torch.zeros_like(qo)
    dq.scatter_(-1, action.long().unsqueeze(-1), (2.0 * diff * w.view(1, B) / B).unsqueeze(-1))
    return R2D2TD(loss, seq_loss, prio, dq, delta)


# --------------------------------------------------------------------------------------------
# LSTM cell (PyTorch gate order i,f,g,o; models.py:99)
# --------------------------------------------------------------------------------------------
def lstm_pointwise(gates: torch.Tensor, c: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    i, f, g, o = gates.chunk(4, -1)
    c2 = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g)
    h2 = torch.sigmoid(o) * torch.tanh(c2)
    return h2, c2


def lstm_unroll(xg: torch.Tensor, w_hh: torch.Tensor, h0: torch.Tensor, c0: torch.Tensor):
    """xg: (T,B,4H) = x·W_ihᵀ + b_ih + b_hh, precomputed for all T. Returns h (T,B,H), c (T,B,H)."""
    hs, cs = [], []
    h, c = h0, c0
    for t in range(xg.shape[0]):
        h, c = lstm_pointwise(xg[t] + h @ w_hh.t(), c)
        hs.append(h)
        cs.append(c)
    return torch.stack(hs), torch.stack(cs)


# --------------------------------------------------------------------------------------------
# optimizers with fused global-norm clip   (learner.py:77-78; learner_node.py:28,40)
# --------------------------------------------------------------------------------------------
def clip_coef(grad_flat: torch.Tensor, max_norm: float) -> torch.Tensor:
    """torch.nn.utils.clip_grad_norm_ coefficient: min(1, max_norm / (‖g‖₂ + 1e-6))."""
    return (max_norm / (grad_flat.float().norm(2) + 1e-6)).clamp(max=1.0)


def rmsprop_step(p: torch.Tensor, g: torch.Tensor, sq: torch.Tensor, lr: float, alpha: float,
                 eps: float, coef=1.0) -> None:
    """torch.optim.RMSprop (no momentum, not centered) with the gradient pre-scaled by ``coef``."""
    g = g * coef
    sq.mul_(alpha).addcmul_(g, g, value=1.0 - alpha)
    p.addcdiv_(g, sq.sqrt().add_(eps), value=-lr)


def adam_step(p: torch.Tensor, g: torch.Tensor, m: torch.Tensor, v: torch.Tensor, step: int,
              lr: float, beta1: float, beta2: float, eps: float, coef=1.0) -> None:
    """torch.optim.Adam (no weight decay, no amsgrad); ``step`` is 1-based."""
    g = g * coef
    m.mul_(beta1).add_(g, alpha=1.0 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1.0 - beta2)
    bc1 = 1.0 - beta1 ** step
    bc2 = 1.0 - beta2 ** step
    denom = (v.sqrt() / (bc2 ** 0.5)).add_(eps)
    p.addcdiv_(m, denom, value=-lr / bc1)

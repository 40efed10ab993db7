"""Small shared helpers (reference: distributed_rl/libs/utils.py:14-62).

What is kept from the reference is the *API surface*: ``Transition``, ``Sequence``,
``preprocess``, ``epsilon_greedy``, ``dumps``/``loads``, ``rescale``/``inv_rescale``.
What changes: frames are uint8 end to end (the reference ships float32 frames,
utils.py:34), and the wire format is a flat header + raw arrays instead of a pickle
(see transport/codec.py); ``dumps``/``loads`` stay for small control messages.
"""
from __future__ import annotations

import pickle
import random
import zlib
from collections import namedtuple
from typing import Optional, Sequence as _Seq

import numpy as np
import torch

try:  # optional, not in this image
    import lz4.frame as _lz4  # type: ignore
except Exception:  # pragma: no cover
    _lz4 = None

_USE_COMPRESS = True

EPS_BASE = 0.4
EPS_ALPHA = 7.0


class Transition(namedtuple("Transition", ("state", "action", "reward", "next_state", "done"))):
    """One (n-step) transition. ``next_state`` / ``done`` default to None (utils.py:14-17)."""

    __slots__ = ()

    def __new__(cls, state, action, reward, next_state=None, done=None):
        return super().__new__(cls, state, action, reward, next_state, done)


#: A stored R2D2 sequence: list of Transitions + the recurrent state *before* the first one (utils.py:19).
Sequence = namedtuple("Sequence", ("transitions", "recurrent_state"))




def preprocess(img: np.ndarray, shape: Optional[_Seq[int]] = None, gray: bool = False,
               dtype=np.float32) -> np.ndarray:
    """Resize / grayscale / CHW-transpose an HWC uint8 image (utils.py:26-34).

    ``dtype=np.float32`` reproduces the reference (values in [0,1]); ``dtype=np.uint8``
    keeps the raw bytes, which is what the replay store and the conv1 kernel consume
    (the ``/255`` is fused into the kernel's load).
    """
    from PIL import Image

    pil_img = Image.fromarray(img)
    if shape is not None:
        pil_img = pil_img.resize(tuple(shape))
    if gray:
        ary = np.asarray(pil_img.convert("L"))
    else:
        ary = np.asarray(pil_img).transpose((2, 0, 1))
    if dtype == np.uint8:
        return np.ascontiguousarray(ary, dtype=np.uint8)
    return np.ascontiguousarray(ary, dtype=np.float32) / 255


def epsilon_greedy(state: torch.Tensor, policy_net, eps: float = 0.1) -> torch.Tensor:
    """Single-env ε-greedy (utils.py:36-41). The batched GPU version is ops.head.eps_greedy."""
    if random.random() > eps:
        with torch.no_grad():
            return policy_net(state).max(1)[1].view(1).cpu()
    return torch.tensor([random.randrange(policy_net.n_action)], dtype=torch.long)


def actor_epsilon(actor_no: int, num_total_actors: int, base: float = EPS_BASE,
                  alpha: float = EPS_ALPHA) -> float:
    """Per-actor ε ladder ε_i = base^(1 + α·(i−1)/(N−1)), i 1-based (ape_x/actor.py:66-69)."""
    if num_total_actors <= 1:
        return base
    return base ** (1.0 + (actor_no - 1.0) / (num_total_actors - 1.0) * alpha)


def epsilon_ladder(num_envs: int, first_actor_no: int = 1, num_total_actors: Optional[int] = None) -> np.ndarray:
    """Vector of ε for a block of ``num_envs`` consecutive actor ids (vectorised actors)."""
    total = num_total_actors if num_total_actors is not None else num_envs
    return np.array([actor_epsilon(first_actor_no + i, total) for i in range(num_envs)], dtype=np.float32)


def _compress(b: bytes) -> bytes:
    if _lz4 is not None:
        return b"L" + _lz4.compress(b)
    return b"Z" + zlib.compress(b, 1)


def _decompress(b: bytes) -> bytes:
    tag, body = b[:1], b[1:]
    if tag == b"L":
        if _lz4 is None:
            raise RuntimeError("blob is lz4-compressed but lz4 is not installed")
        return _lz4.decompress(body)
    if tag == b"Z":
        return zlib.decompress(body)
    if tag == b"R":
        return body
    raise ValueError("unknown blob tag %r" % tag)


def dumps(data) -> bytes:
    """Serialise a small python object (utils.py:43-47). lz4 when importable, else zlib-1."""
    raw = pickle.dumps(data, protocol=pickle.HIGHEST_PROTOCOL)
    return _compress(raw) if _USE_COMPRESS else b"R" + raw


def loads(packed: bytes):
    """Inverse of :func:`dumps` (utils.py:49-53). Only ever called on blobs this framework wrote."""
    return pickle.loads(_decompress(packed))


def rescale(x: torch.Tensor, eps: float = 1.0e-3) -> torch.Tensor:
    """R2D2 value rescaling h(x) = sign(x)(√(|x|+1) − 1) + εx (utils.py:55-56)."""
    return x.sign() * ((x.abs() + 1.0).sqrt() - 1.0) + eps * x


def inv_rescale(x: torch.Tensor, eps: float = 1.0e-3) -> torch.Tensor:
    """Closed-form h⁻¹ (utils.py:58-62)."""
    if eps == 0:
        return x.sign() * (x * x + 2.0 * x.abs())
    return x.sign() * ((((1.0 + 4.0 * eps * (x.abs() + 1.0 + eps)).sqrt() - 1.0) / (2.0 * eps)).pow(2) - 1.0)


def nstep_return(rewards, gamma: float, reference_order: bool = False) -> float:
    """n-step discounted return over ``rewards`` (oldest first).

    Standard ``Σ_k γ^k r_{t+k}`` by default. ``reference_order=True`` reproduces the
    reference's reversed weighting ``Σ_i γ^{n−1−i} r_i`` (ape_x/actor.py:61,79 — SURVEY Q1).
    """
    n = len(rewards)
    if reference_order:
        return float(sum(gamma ** (n - 1 - i) * float(r) for i, r in enumerate(rewards)))
    return float(sum(gamma ** i * float(r) for i, r in enumerate(rewards)))




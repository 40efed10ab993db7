"""Flat parameter buffers in *kernel layout*  ⇄  reference-shaped ``state_dict``.

The gfx950 kernels want NHWC activations, so the conv2 / conv3 weights are stored ``[C_out][kh][kw][C_in]``
(implicit-GEMM K order); conv1 reads single uint8 planes of the replay store (every frame kept once) and keeps
PyTorch's own ``[C_out][C_in][kh][kw]`` order (K = plane, patch row, pixel: csrc/hip/conv_direct.hip); the actors,
whose environments deliver HWC stacks, keep an HWC-ordered copy (``conv1_hwc``). The two 3136→512 heads are concatenated into one 1024-row GEMM
 and every matrix that consumes the flattened conv3 output has its columns
permuted from the reference's NCHW flatten order (models.py:24) to HWC. One flat fp32 buffer
holds all parameters back to back (16-byte aligned segments) so that:

* the gradient all-reduce is a single RCCL collective over one contiguous buffer,
* clip + optimizer + bf16 re-cast is one multi-tensor kernel over one range,
* a parameter broadcast / publish is one copy.

``ParamPack.load_state_dict`` / ``state_dict`` do the (exactly invertible) permutations, so
checkpoints stay interchangeable with the reference's ``model_<t>.pth`` files.
"""
from __future__ import annotations

from collections import OrderedDict
from dataclasses import dataclass
from typing import Callable, Dict, List, Tuple

import torch


@dataclass
class Segment:
    name: str
    shape: Tuple[int, ...]
    offset: int = 0

    @property
    def numel(self) -> int:
        n = 1
        for s in self.shape:
            n *= s
        return n


def _align(n: int, a: int = 64) -> int:
    return (n + a - 1) // a * a


def _chw_cols_to_hwc(w: torch.Tensor, c: int, h: int, wd: int) -> torch.Tensor:
    """(N, c*h*w) columns in CHW order → HWC order."""
    n = w.shape[0]
    return w.reshape(n, c, h, wd).permute(0, 2, 3, 1).reshape(n, h * wd * c)


def _hwc_cols_to_chw(w: torch.Tensor, c: int, h: int, wd: int) -> torch.Tensor:
    n = w.shape[0]
    return w.reshape(n, h, wd, c).permute(0, 3, 1, 2).reshape(n, c * h * wd)


def _conv_to_kernel(w: torch.Tensor) -> torch.Tensor:      # (O,I,kh,kw) → (O,kh,kw,I)
    return w.permute(0, 2, 3, 1).contiguous()


def _conv_from_kernel(w: torch.Tensor) -> torch.Tensor:    # (O,kh,kw,I) → (O,I,kh,kw)
    return w.permute(0, 3, 1, 2).contiguous()


def conv1_hwc(w: torch.Tensor) -> torch.Tensor:
    """conv1 weights of a flat buffer, (O, I, kh, kw) -> (O, kh, kw, I): the K order of conv1 over dense HWC stacks (actors)."""
    return w.reshape(32, -1, 8, 8).permute(0, 2, 3, 1).contiguous()


def _gate_interleave(w: torch.Tensor, H: int) -> torch.Tensor:
    """PyTorch LSTM row order [i(H); f(H); g(H); o(H)] -> kernel order n = 4*unit + gate, so one lane of the MFMA
    accumulator (4 consecutive columns) holds all four gates of a unit (csrc/hip/gemm.hip EPI_LSTM)."""
    return w.reshape(4, H, *w.shape[1:]).transpose(0, 1).reshape(4 * H, *w.shape[1:]).contiguous()


def _gate_deinterleave(w: torch.Tensor, H: int) -> torch.Tensor:
    return w.reshape(H, 4, *w.shape[1:]).transpose(0, 1).reshape(4 * H, *w.shape[1:]).contiguous()


class Layout:
    """Ordered list of kernel-layout segments + pack/unpack rules against a reference state_dict."""

    def __init__(self, segments: List[Segment], pack: Callable, unpack: Callable, meta: Dict):
        off = 0
        self.segments: "OrderedDict[str, Segment]" = OrderedDict()
        for s in segments:
            s.offset = off
            off = _align(off + s.numel)
            self.segments[s.name] = s
        self.total = off
        self.pack = pack      # state_dict -> {segment name: tensor}
        self.unpack = unpack  # {segment name: tensor} -> state_dict
        self.meta = meta

    def offset(self, name: str) -> int:
        return self.segments[name].offset


def apex_layout(n_action: int, in_ch: int = 4, feat_hw: Tuple[int, int] = (7, 7)) -> Layout:
    fh, fw = feat_hw
    F = fh * fw * 64
    segs = [
        Segment("conv1.w", (32, in_ch, 8, 8)), Segment("conv1.b", (32,)),
        Segment("conv2.w", (64, 4, 4, 32)), Segment("conv2.b", (64,)),
        Segment("conv3.w", (64, 3, 3, 64)), Segment("conv3.b", (64,)),
        Segment("fc1.w", (1024, F)), Segment("fc1.b", (1024,)),      # rows 0..511 adv1, 512..1023 val1
        Segment("adv2.w", (n_action, 512)), Segment("adv2.b", (n_action,)),
        Segment("val2.w", (1, 512)), Segment("val2.b", (1,)),
    ]

    def pack(sd):
        return {
            "conv1.w": sd["conv1.weight"], "conv1.b": sd["conv1.bias"],
            "conv2.w": _conv_to_kernel(sd["conv2.weight"]), "conv2.b": sd["conv2.bias"],
            "conv3.w": _conv_to_kernel(sd["conv3.weight"]), "conv3.b": sd["conv3.bias"],
            "fc1.w": _chw_cols_to_hwc(torch.cat([sd["adv1.weight"], sd["val1.weight"]], 0), 64, fh, fw),
            "fc1.b": torch.cat([sd["adv1.bias"], sd["val1.bias"]], 0),
            "adv2.w": sd["adv2.weight"], "adv2.b": sd["adv2.bias"],
            "val2.w": sd["val2.weight"], "val2.b": sd["val2.bias"],
        }

    def unpack(k):
        fc1 = _hwc_cols_to_chw(k["fc1.w"], 64, fh, fw)
        return OrderedDict([
            ("conv1.weight", k["conv1.w"].clone()), ("conv1.bias", k["conv1.b"].clone()),
            ("conv2.weight", _conv_from_kernel(k["conv2.w"])), ("conv2.bias", k["conv2.b"].clone()),
            ("conv3.weight", _conv_from_kernel(k["conv3.w"])), ("conv3.bias", k["conv3.b"].clone()),
            ("adv1.weight", fc1[:512].contiguous()), ("adv1.bias", k["fc1.b"][:512].clone()),
            ("adv2.weight", k["adv2.w"].clone()), ("adv2.bias", k["adv2.b"].clone()),
            ("val1.weight", fc1[512:].contiguous()), ("val1.bias", k["fc1.b"][512:].clone()),
            ("val2.weight", k["val2.w"].clone()), ("val2.bias", k["val2.b"].clone()),
        ])

    return Layout(segs, pack, unpack, dict(algorithm="ape_x", n_action=n_action, in_ch=in_ch, feat=F))


def r2d2_layout(n_action: int, in_ch: int = 4, feat_hw: Tuple[int, int] = (7, 7), hidden: int = 512) -> Layout:
    fh, fw = feat_hw
    F = fh * fw * 64
    H = hidden
    segs = [
        Segment("conv1.w", (32, in_ch, 8, 8)), Segment("conv1.b", (32,)),
        Segment("conv2.w", (64, 4, 4, 32)), Segment("conv2.b", (64,)),
        Segment("conv3.w", (64, 3, 3, 64)), Segment("conv3.b", (64,)),
        Segment("lstm.w_ih", (4 * H, F)), Segment("lstm.w_hh", (4 * H, H)),
        Segment("lstm.b_ih", (4 * H,)), Segment("lstm.b_hh", (4 * H,)),
        Segment("fc1.w", (1024, H)), Segment("fc1.b", (1024,)),
        Segment("adv2.w", (n_action, 512)),
        Segment("val2.w", (1, 512)), Segment("val2.b", (1,)),
    ]

    def pack(sd):
        return {
            "conv1.w": sd["conv1.weight"], "conv1.b": sd["conv1.bias"],
            "conv2.w": _conv_to_kernel(sd["conv2.weight"]), "conv2.b": sd["conv2.bias"],
            "conv3.w": _conv_to_kernel(sd["conv3.weight"]), "conv3.b": sd["conv3.bias"],
            "lstm.w_ih": _gate_interleave(_chw_cols_to_hwc(sd["lstm.weight_ih"], 64, fh, fw), H),
            "lstm.w_hh": _gate_interleave(sd["lstm.weight_hh"], H),
            "lstm.b_ih": _gate_interleave(sd["lstm.bias_ih"], H), "lstm.b_hh": _gate_interleave(sd["lstm.bias_hh"], H),
            "fc1.w": torch.cat([sd["adv1.weight"], sd["val1.weight"]], 0),
            "fc1.b": torch.cat([sd["adv1.bias"], sd["val1.bias"]], 0),
            "adv2.w": sd["adv2.weight"], "val2.w": sd["val2.weight"], "val2.b": sd["val2.bias"],
        }

    def unpack(k):
        return OrderedDict([
            ("conv1.weight", k["conv1.w"].clone()), ("conv1.bias", k["conv1.b"].clone()),
            ("conv2.weight", _conv_from_kernel(k["conv2.w"])), ("conv2.bias", k["conv2.b"].clone()),
            ("conv3.weight", _conv_from_kernel(k["conv3.w"])), ("conv3.bias", k["conv3.b"].clone()),
            ("lstm.weight_ih", _hwc_cols_to_chw(_gate_deinterleave(k["lstm.w_ih"], H), 64, fh, fw).contiguous()),
            ("lstm.weight_hh", _gate_deinterleave(k["lstm.w_hh"], H)),
            ("lstm.bias_ih", _gate_deinterleave(k["lstm.b_ih"], H)), ("lstm.bias_hh", _gate_deinterleave(k["lstm.b_hh"], H)),
            ("adv1.weight", k["fc1.w"][:512].contiguous()), ("adv1.bias", k["fc1.b"][:512].clone()),
            ("adv2.weight", k["adv2.w"].clone()),
            ("val1.weight", k["fc1.w"][512:].contiguous()), ("val1.bias", k["fc1.b"][512:].clone()),
            ("val2.weight", k["val2.w"].clone()), ("val2.bias", k["val2.b"].clone()),
        ])

    return Layout(segs, pack, unpack, dict(algorithm="r2d2", n_action=n_action, in_ch=in_ch, feat=F, hidden=H))


def layout_for(model) -> Layout:
    from .dueling_dqn import DuelingDQN
    from .dueling_lstm_dqn import DuelingLSTMDQN
    if isinstance(model, DuelingLSTMDQN):
        return r2d2_layout(model.n_action, model.input_shape[0], model.feature_hw, model.hidden)
    if isinstance(model, DuelingDQN):
        return apex_layout(model.n_action, model.input_shape[0], model.feature_hw)
    raise TypeError(type(model))


class ParamPack:
    """A flat buffer (any dtype/device) laid out by a :class:`Layout`, with named views."""

    def __init__(self, layout: Layout, device="cpu", dtype=torch.float32, flat: torch.Tensor = None):
        self.layout = layout
        if flat is not None:                                   # adopt an existing buffer (e.g. a communication buffer)
            assert flat.numel() == layout.total and flat.is_contiguous()
            self.flat = flat
        else:
            self.flat = torch.zeros(layout.total, device=device, dtype=dtype)

    def view(self, name: str) -> torch.Tensor:
        s = self.layout.segments[name]
        return self.flat[s.offset:s.offset + s.numel].view(s.shape)

    def views(self) -> Dict[str, torch.Tensor]:
        return {n: self.view(n) for n in self.layout.segments}

    def like(self, dtype=None, flat: torch.Tensor = None) -> "ParamPack":
        return ParamPack(self.layout, self.flat.device, dtype or self.flat.dtype, flat=flat)

    @torch.no_grad()
    def load_state_dict(self, sd) -> None:
        packed = self.layout.pack({k: v.detach() for k, v in sd.items()})
        for n, t in packed.items():
            self.view(n).copy_(t.reshape(self.layout.segments[n].shape))

    @torch.no_grad()
    def state_dict(self, device="cpu") -> "OrderedDict[str, torch.Tensor]":
        k = {n: v.detach().float().to(device) for n, v in self.views().items()}
        return self.layout.unpack(k)

    @classmethod
    def from_module(cls, model, device="cpu", dtype=torch.float32) -> "ParamPack":
        p = cls(layout_for(model), device, dtype)
        p.load_state_dict(model.state_dict())
        return p

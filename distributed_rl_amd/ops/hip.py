"""Thin Python entry points of the gfx950 kernels (csrc/hip). Tensors must already live on the GPU, be
contiguous and of the documented dtype; launches go to torch's *current* stream, so they can be captured into
a ``torch.cuda.CUDAGraph`` (a hipGraph) together with RCCL collectives. There is deliberately no PyTorch
fallback here: if ``libdrl_hip.so`` is missing these raise (oracle implementations live in ops/oracle.py).
"""
from __future__ import annotations

import ctypes
import threading
import os
from typing import Optional, Sequence

import torch

from .. import native

EPI_RELU, EPI_OUT_F32, EPI_ATOMIC, EPI_MASK = 1, 2, 4, 8
NT, NN, TN = 0, 1, 2
# tile variants of drl_gemm: 0 = 64x64x64 | 1 = 128x128x64 | 2 = 64x64x128 | 3 = 64x64x256 | 4 = 128x64x128
GEMM_SMALL_VARIANT = 0
FC1_FWD_VARIANT = 10        # dense-GEMM variant of the Ape-X FC1 forward (see drl_gemm: 10 = 64x64 tiles, LDS-DMA, 4 stages)


def _p(t: Optional[torch.Tensor]):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _s():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ck(rc: int, what: str) -> None:
    if rc != 0:
        raise RuntimeError("%s failed with code %d" % (what, rc))


def _chk(t: torch.Tensor, dtype, name: str) -> None:
    if t.dtype != dtype or not t.is_contiguous() or not t.is_cuda:
        raise TypeError("%s must be a contiguous CUDA %s tensor (got %s, contiguous=%s, device=%s)"
                        % (name, dtype, t.dtype, t.is_contiguous(), t.device))


# ---- step state ---------------------------------------------------------------------------------------
STEP_STATE_BYTES = 64


def new_step_state(device, beta0: float, beta_decay: int) -> torch.Tensor:
    """64-byte device struct (csrc/hip/common.h StepState): step, rng_ctr, beta, beta0, beta_inc, total_prio,
    n_items, loss, grad_norm, max_w."""
    import struct
    raw = struct.pack("<QQffff", 0, 0, beta0, beta0, (1.0 - beta0) / beta_decay, 0.0)
    host = bytearray(STEP_STATE_BYTES)
    host[:len(raw)] = raw
    return torch.frombuffer(host, dtype=torch.uint8).clone().to(device)


def read_step_state(st: torch.Tensor) -> dict:
    import struct
    b = bytes(st.cpu().numpy().tobytes())
    step, ctr, beta, beta0, beta_inc, total = struct.unpack_from("<QQffff", b, 0)
    n_items, = struct.unpack_from("<q", b, 32)
    loss_even, gnorm, max_w, loss_odd = struct.unpack_from("<ffff", b, 40)
    # two loss slots by step parity: "loss" = the last COMPLETED step (step already advanced by step_end), "loss_cur" = the
    # step in progress (raw-kernel tests that read before step_end)
    slots = (loss_even, loss_odd)
    return dict(step=step, rng_ctr=ctr, beta=beta, beta0=beta0, beta_inc=beta_inc, total_prio=total, n_items=n_items,
                loss=slots[(step - 1) & 1] if step > 0 else loss_even, loss_cur=slots[step & 1], grad_norm=gnorm, max_w=max_w)


def step_begin(st: torch.Tensor, n_items: torch.Tensor) -> None:
    _ck(native.hip().drl_step_begin(_p(st), _p(n_items), _s()), "step_begin")


def step_end(st: torch.Tensor) -> None:
    _ck(native.hip().drl_step_end(_p(st), _s()), "step_end")


# ---- sum-tree -----------------------------------------------------------------------------------------
class TreeLayout:
    """64-ary sum-tree over ``capacity`` leaves inside one flat fp32 buffer."""

    def __init__(self, capacity: int):
        self.capacity = int(capacity)
        sizes, n = [], self.capacity
        while True:
            padded = (n + 63) // 64 * 64
            sizes.append(padded)
            if n <= 64:
                break
            n = (n + 63) // 64
        self.sizes = sizes
        self.n_levels = len(sizes)
        offs, o = [], 0
        for s in sizes:
            offs.append(o)
            o += s
        self.offsets = offs
        self.total = o
        self.c_off = (ctypes.c_longlong * 6)(*(offs + [0] * (6 - len(offs))))


def sumtree_sample(tree: torch.Tensor, lay: TreeLayout, st: torch.Tensor, seed: int, idx_out: torch.Tensor,
                   prio_out: torch.Tensor, gen: Optional[torch.Tensor] = None, gen_out: Optional[torch.Tensor] = None,
                   frame_off: Optional[torch.Tensor] = None, win: Optional[torch.Tensor] = None,
                   begin_n_items: Optional[torch.Tensor] = None, is_n_scale: float = 0.0) -> None:
    """``frame_off`` (int64 [2B]) receives the byte offsets of the plane windows of s and s' of every draw, looked up in
    ``win`` (int64 [2 cap]: s in the first half, s' in the second; written at ingest).
    ``begin_n_items`` (int64 device scalar) fuses step_begin into the sampling kernel: beta / n_items of the step
    state are refreshed and the Philox counter of step t is rng_ctr + t + 1 (do not also call step_begin then).
    ``is_n_scale`` > 0 (with begin_n_items) additionally accumulates the batch's max un-normalised IS weight
    (is_n_scale * N * p / total)^-beta in the step state and clears its loss, as ``apex_td_dh`` expects."""
    B = idx_out.numel()
    _chk(tree, torch.float32, "tree"); _chk(idx_out, torch.int32, "idx_out"); _chk(prio_out, torch.float32, "prio_out")
    if frame_off is not None:
        _chk(frame_off, torch.int64, "frame_off"); _chk(win, torch.int64, "win")
        assert frame_off.numel() >= 2 * B and win.numel() == 2 * lay.capacity
    _ck(native.hip().drl_sumtree_sample(_p(tree), lay.n_levels, lay.c_off, lay.capacity, _p(st), seed, B, _p(idx_out),
                                        _p(prio_out), _p(gen), _p(gen_out), _p(frame_off), _p(win), _p(begin_n_items),
                                        float(is_n_scale), _s()), "sumtree_sample")


def seq_sample(tree: torch.Tensor, lay: TreeLayout, st: torch.Tensor, seed: int, idx_out: torch.Tensor, prio_out: torch.Tensor,
               gen: Optional[torch.Tensor], gen_out: Optional[torch.Tensor], begin_n_items: Optional[torch.Tensor],
               win: torch.Tensor, frame_off: torch.Tensor, T: int, h0: torch.Tensor, c0: torch.Tensor, hs_a, cs_a, hs_b=None,
               cs_b=None, sums=()) -> None:
    """``sumtree_sample`` of B sequences + ``seq_win_offsets`` + ``gather_state2`` (+ its vector sums) as ONE launch
    (sumtree.hip k_seq_sample): the R2D2 learner's first launch of a step."""
    B = idx_out.numel()
    _chk(tree, torch.float32, "tree"); _chk(idx_out, torch.int32, "idx_out"); _chk(prio_out, torch.float32, "prio_out")
    _chk(win, torch.int64, "win"); _chk(frame_off, torch.int64, "frame_off")
    _chk(h0, torch.float32, "h0"); _chk(c0, torch.float32, "c0"); _chk(hs_a, torch.bfloat16, "hs_a"); _chk(cs_a, torch.float32, "cs_a")
    Hd = h0.shape[-1]
    assert win.numel() == lay.capacity * T and frame_off.numel() >= T * B and h0.shape == c0.shape and h0.shape[0] == lay.capacity
    assert hs_a.numel() >= B * Hd and cs_a.numel() >= B * Hd
    if hs_b is not None:
        _chk(hs_b, torch.bfloat16, "hs_b"); _chk(cs_b, torch.float32, "cs_b")
        assert hs_b.numel() >= B * Hd and cs_b.numel() >= B * Hd
    assert len(sums) <= 2
    n_add, ptrs = 0, []
    for a, b, out in sums:
        _chk(a, torch.float32, "sum a"); _chk(b, torch.float32, "sum b"); _chk(out, torch.float32, "sum out")
        assert a.numel() == b.numel() == out.numel() and (n_add == 0 or n_add == a.numel())
        n_add = a.numel()
        ptrs += [_p(a), _p(b), _p(out)]
    ptrs += [None] * (6 - len(ptrs))
    _ck(native.hip().drl_seq_sample(_p(tree), lay.n_levels, lay.c_off, lay.capacity, _p(st), seed, B, _p(idx_out), _p(prio_out),
                                    _p(gen), _p(gen_out), _p(begin_n_items), _p(win), _p(frame_off), T, _p(h0), _p(c0), Hd,
                                    _p(hs_a), _p(cs_a), _p(hs_b), _p(cs_b), *ptrs, n_add, _s()), "seq_sample")


def sumtree_update(tree: torch.Tensor, lay: TreeLayout, idx: torch.Tensor, val: torch.Tensor,
                   gen: Optional[torch.Tensor] = None, gen_expected: Optional[torch.Tensor] = None,
                   end_state: Optional[torch.Tensor] = None, fault: Optional[torch.Tensor] = None) -> None:
    """Two launches for any depth: leaves + level 1 (one wave per index), then every level >= 2 rebuilt from level 1.
    ``end_state`` fuses step_end (step += 1) into the second launch. ``fault`` (int32 device word, see grad_sumsq):
    non-zero = the priorities of a discarded step are not written."""
    _chk(idx, torch.int32, "idx"); _chk(val, torch.float32, "val")
    assert idx.numel() == val.numel()
    _ck(native.hip().drl_sumtree_update(_p(tree), lay.n_levels, lay.c_off, lay.capacity, _p(idx), _p(val), idx.numel(),
                                        _p(gen), _p(gen_expected), _p(end_state), _p(fault), _s()), "sumtree_update")


def sumtree_ingest(tree: torch.Tensor, lay: TreeLayout, start: int, prio: Optional[torch.Tensor], gen: Optional[torch.Tensor],
                   idx_scratch: torch.Tensor, n: Optional[int] = None, n_items: Optional[torch.Tensor] = None,
                   n_items_val: int = 0) -> None:
    """Ring slots [start, start + n) get new priorities and a new generation. ``prio=None`` retires them (priority 0)
    while their data is being replaced (engine/ingest.py). ``n_items`` (int64 device scalar) is set to ``n_items_val``
    by the same launch."""
    n = prio.numel() if n is None else int(n)
    if prio is not None:
        _chk(prio, torch.float32, "prio")
        assert prio.numel() >= n
    _chk(idx_scratch, torch.int32, "idx_scratch")
    assert idx_scratch.numel() >= n
    if n_items is not None:
        _chk(n_items, torch.int64, "n_items")
    _ck(native.hip().drl_sumtree_ingest(_p(tree), lay.n_levels, lay.c_off, lay.capacity, int(start), n, _p(prio),
                                        _p(gen), _p(idx_scratch), _p(n_items), int(n_items_val), _s()), "sumtree_ingest")


def gather_rows(src: torch.Tensor, off: torch.Tensor, dst: torch.Tensor, row_bytes: int) -> None:
    _chk(off, torch.int64, "off")
    assert dst.numel() * dst.element_size() >= off.numel() * row_bytes
    _ck(native.hip().drl_gather_rows(_p(src), _p(off), _p(dst), row_bytes, off.numel(), _s()), "gather_rows")


# ---- TD ------------------------------------------------------------------------------------------------
def apex_td(q_on, q_tg, idx, act, rew, done, p_sampled, st, gamma_n, alpha, prio_eps, dq, new_prio,
            delta_out=None, w_out=None, n_scale: float = 1.0, use_is: bool = True) -> None:
    B, A = q_tg.shape
    assert q_on.shape == (2 * B, A) and dq.shape == (B, A)
    for t, n in ((q_on, "q_on"), (q_tg, "q_tg"), (rew, "rew"), (done, "done"), (dq, "dq"), (new_prio, "new_prio")):
        _chk(t, torch.float32, n)
    _chk(act, torch.int32, "act")
    _ck(native.hip().drl_apex_td(_p(q_on), _p(q_tg), B, A, _p(idx), _p(act), _p(rew), _p(done), _p(p_sampled), _p(st),
                                 gamma_n, alpha, prio_eps, n_scale, 1 if use_is else 0, _p(dq), _p(new_prio), _p(delta_out),
                                 _p(w_out), _s()), "apex_td")


def apex_td_dh(q_on, q_tg, idx, act, rew, done, p_sampled, st, gamma_n, alpha, prio_eps, dq, new_prio, h, w_adv, w_val, dh,
               n_scale: float = 1.0) -> None:
    """``apex_td`` fused with the dueling head's input gradient (head.hip k_apex_td_dh): one wave per transition, no
    batch-wide reduction - the max IS weight comes from the sampling kernel (sumtree_sample(..., is_n_scale=n_scale))."""
    B, A = q_tg.shape
    assert q_on.shape == (2 * B, A) and dq.shape == (B, A)
    for t, n in ((q_on, "q_on"), (q_tg, "q_tg"), (rew, "rew"), (done, "done"), (dq, "dq"), (new_prio, "new_prio")):
        _chk(t, torch.float32, n)
    _chk(act, torch.int32, "act"); _chk(h, torch.bfloat16, "h"); _chk(dh, torch.bfloat16, "dh")
    _ck(native.hip().drl_apex_td_dh(_p(q_on), _p(q_tg), B, A, _p(idx), _p(act), _p(rew), _p(done), _p(p_sampled), _p(st),
                                    gamma_n, alpha, prio_eps, n_scale, _p(dq), _p(new_prio), _p(h), _p(w_adv), _p(w_val),
                                    _p(dh), _s()), "apex_td_dh")


def apex_head_td_dh(h_on, h_tg, head_on, head_tg, q_on, q_tg, idx, act, rew, done, p_sampled, st, gamma_n, alpha, prio_eps, dq,
                    new_prio, dh, n_scale: float = 1.0) -> None:
    """The dueling heads of Q_online(s), Q_online(s'), Q_target(s') + ``apex_td_dh`` as ONE launch (head.hip
    k_apex_head_td_dh), one wave per transition. h_on bf16 [2B, 1024] (rows [0,B) = s, [B,2B) = s'), h_tg bf16 [B, 1024];
    head_* = (w_adv, b_adv | None, w_val, b_val). q_on / q_tg are written too."""
    B, A = q_tg.shape
    assert q_on.shape == (2 * B, A) and dq.shape == (B, A) and h_on.shape == (2 * B, 1024) and h_tg.shape[0] >= B
    for t, n in ((q_on, "q_on"), (q_tg, "q_tg"), (rew, "rew"), (done, "done"), (dq, "dq"), (new_prio, "new_prio")):
        _chk(t, torch.float32, n)
    _chk(act, torch.int32, "act"); _chk(h_on, torch.bfloat16, "h_on"); _chk(h_tg, torch.bfloat16, "h_tg"); _chk(dh, torch.bfloat16, "dh")
    (wa1, ba1, wv1, bv1), (wa2, ba2, wv2, bv2) = head_on, head_tg
    assert wa1.shape == (A, 512) and wa2.shape == (A, 512)
    _ck(native.hip().drl_apex_head_td_dh(_p(h_on), _p(h_tg), _p(wa1), _p(ba1), _p(wv1), _p(bv1), _p(wa2), _p(ba2), _p(wv2), _p(bv2),
                                         _p(q_on), _p(q_tg), B, A, _p(idx), _p(act), _p(rew), _p(done), _p(p_sampled), _p(st),
                                         gamma_n, alpha, prio_eps, n_scale, _p(dq), _p(new_prio), _p(dh), _s()), "apex_head_td_dh")


def wmax_put(st, mailbox: torch.Tensor, rank: int, world: int) -> None:
    """This rank's two max-IS-weight slots -> its place in the mailbox (zeros elsewhere); td.hip k_wmax_put."""
    _chk(mailbox, torch.float32, "mailbox")
    _ck(native.hip().drl_wmax_put(_p(st), _p(mailbox), rank, world, mailbox.numel(), _s()), "wmax_put")


def wmax_take(st, mailbox: torch.Tensor, world: int) -> None:
    """Maximum over the ranks' mailbox slots (after the SUM all-reduce that carried them) -> StepState::wmax_bits."""
    _ck(native.hip().drl_wmax_take(_p(st), _p(mailbox), world, _s()), "wmax_take")


def is_weights(p_sampled, st, w_out, n_scale: float = 1.0, mailbox: Optional[torch.Tensor] = None, world: int = 1) -> None:
    """IS weights of the sampled batch, normalised by their maximum - or, with ``mailbox`` (the ranks' batch maxima, gathered
    by a SUM all-reduce after ``is_wmax_put``), by the maximum over all ranks' batches."""
    if mailbox is not None:
        _chk(mailbox, torch.float32, "mailbox")
        assert mailbox.numel() >= world
    _ck(native.hip().drl_is_weights(_p(p_sampled), p_sampled.numel(), _p(st), n_scale, _p(w_out), _p(mailbox), world, _s()),
        "is_weights")


def is_wmax_put(p_sampled, st, mailbox: torch.Tensor, rank: int, world: int, n_scale: float = 1.0) -> None:
    """The maximum IS weight of this rank's sampled batch -> mailbox[rank], zeros in the other slots (td.hip k_is_wmax_put)."""
    _chk(mailbox, torch.float32, "mailbox")
    _ck(native.hip().drl_is_wmax_put(_p(p_sampled), p_sampled.numel(), _p(st), n_scale, _p(mailbox), rank, world, mailbox.numel(),
                                     _s()), "is_wmax_put")


def r2d2_td(q_on, q_tg, idx, act, rew, done, lengths, w, st, burn_in, nstep, done_offset, gamma_n, eta, rs_eps, dq,
            new_prio, seq_loss=None) -> None:
    T, B, A = q_on.shape
    for t, n in ((q_on, "q_on"), (q_tg, "q_tg"), (rew, "rew"), (done, "done"), (dq, "dq"), (new_prio, "new_prio")):
        _chk(t, torch.float32, n)
    _chk(act, torch.int32, "act"); _chk(lengths, torch.int32, "lengths")
    _ck(native.hip().drl_r2d2_td(_p(q_on), _p(q_tg), T, B, A, _p(idx), _p(act), _p(rew), _p(done), _p(lengths), _p(w), _p(st),
                                 burn_in, nstep, done_offset, gamma_n, eta, rs_eps, _p(dq), _p(new_prio), _p(seq_loss), _s()),
        "r2d2_td")


# ---- optimizer -----------------------------------------------------------------------------------------
def n_partials(n: int) -> int:
    return max(1, min(1024, (n // 4 + 255) // 256))


def _gp(x):
    """Guard word: an int32 device tensor, a raw device address, or None."""
    if x is None:
        return None
    return ctypes.c_void_p(x) if isinstance(x, int) else _p(x)


def grad_sumsq(g: torch.Tensor, partial: torch.Tensor, guards=(), fault: Optional[torch.Tensor] = None) -> int:
    """Partial sums of squares of the flat gradient. ``guards``: up to four int32 device words (tensors or raw addresses)
    that are non-zero when a bounded in-kernel wait of this step gave up; they are folded into the sticky ``fault`` word
    the update kernel (``rmsprop_step`` / ``adam_step`` with the same ``fault``) checks before it touches a parameter."""
    _chk(g, torch.float32, "g"); _chk(partial, torch.float32, "partial")
    assert partial.numel() >= 1024 and len(guards) <= 4
    if fault is not None:
        _chk(fault, torch.int32, "fault")
    gs = [_gp(x) for x in guards] + [None] * (4 - len(guards))
    n = ctypes.c_int()
    _ck(native.hip().drl_grad_sumsq(_p(g), g.numel(), _p(partial), ctypes.byref(n), gs[0], gs[1], gs[2], gs[3], _p(fault), _s()),
        "grad_sumsq")
    return n.value


def lstm_set_spin_limit(n: int) -> None:
    """Bound (poll iterations) of the inter-workgroup waits of the persistent LSTM / BPTT kernels; default 1 << 17. Tests
    shrink it to drive the give-up path."""
    _ck(native.hip().drl_lstm_set_spin_limit(int(n)), "lstm_set_spin_limit")


def rmsprop_step(p, g, sq, p_bf16, partial, n_partial, grad_scale, max_norm, lr, alpha, eps, st,
                 zero_grad: bool = False, copies: Optional["CopyExtra"] = None, fault: Optional[torch.Tensor] = None,
                 keep=(0, 0), late: int = 0) -> None:
    """``zero_grad`` leaves ``g`` zeroed for the next step's atomic accumulation (no separate memset launch) except the
    elements ``keep`` = [a, b), which the caller knows the next step OVERWRITES (plain GEMM outputs). ``copies``
    (see copy_extra) are done by one extra block of the same launch. ``fault`` (see grad_sumsq): non-zero = the step is
    discarded (nothing but the gradient zeroing and the copies happens). ``late`` (elements): g[0 .. late) is NOT covered
    by ``partial`` (gradient-norm partials that rode along with earlier launches, ``tree_extra(sumsq=...)``): every
    workgroup adds its squares itself, and the range is left un-zeroed (``tree_extra(zero=...)`` does that later)."""
    _ck(native.hip().drl_rmsprop_step(_p(p), _p(g), _p(sq), _p(p_bf16), p.numel(), _p(partial), n_partial, grad_scale,
                                      max_norm, lr, alpha, eps, _p(st), 1 if zero_grad else 0,
                                      ctypes.byref(copies) if copies is not None else None, _p(fault), keep[0], keep[1], int(late),
                                      _s()), "rmsprop_step")


def adam_step(p, g, m, v, p_bf16, partial, n_partial, grad_scale, max_norm, lr, b1, b2, eps, st,
              zero_grad: bool = False, fault: Optional[torch.Tensor] = None, keep=(0, 0)) -> None:
    _ck(native.hip().drl_adam_step(_p(p), _p(g), _p(m), _p(v), _p(p_bf16), p.numel(), _p(partial), n_partial, grad_scale,
                                   max_norm, lr, b1, b2, eps, _p(st), 1 if zero_grad else 0, _p(fault), keep[0], keep[1], _s()),
        "adam_step")


def cast_bf16(src: torch.Tensor, dst: torch.Tensor) -> None:
    _chk(src, torch.float32, "src"); _chk(dst, torch.bfloat16, "dst")
    _ck(native.hip().drl_cast_bf16(_p(src), _p(dst), src.numel(), _s()), "cast_bf16")


# ---- head ----------------------------------------------------------------------------------------------
def head_fwd(h, w_adv, b_adv, w_val, b_val, q_out, action_out=None, eps=None, eps_mod: int = 0, seed: int = 0,
             ctr: Optional[torch.Tensor] = None) -> None:
    M, A = q_out.shape
    _chk(h, torch.bfloat16, "h"); _chk(q_out, torch.float32, "q_out")
    assert h.shape == (M, 1024) and w_adv.shape == (A, 512)
    _ck(native.hip().drl_head_fwd(_p(h), M, A, _p(w_adv), _p(b_adv), _p(w_val), _p(b_val), _p(q_out), _p(action_out), _p(eps),
                                  eps_mod, seed, _p(ctr), _s()), "head_fwd")


def head_bwd(dq, h, w_adv, w_val, dh, gw_adv, gb_adv, gw_val, gb_val, parts: int = 3) -> None:
    """parts: 1 = input gradient dh only, 2 = weight / bias gradients only, 3 = both."""
    M, A = dq.shape
    _chk(dq, torch.float32, "dq"); _chk(h, torch.bfloat16, "h"); _chk(dh, torch.bfloat16, "dh")
    _ck(native.hip().drl_head_bwd(_p(dq), _p(h), M, A, _p(w_adv), _p(w_val), _p(dh), _p(gw_adv), _p(gb_adv), _p(gw_val),
                                  _p(gb_val), parts, _s()), "head_bwd")


# ---- GEMM / conv ---------------------------------------------------------------------------------------
def gemm(mode: int, a: torch.Tensor, b: torch.Tensor, c: torch.Tensor, M: int, N: int, K: int, bias=None, mask=None,
         scale: float = 1.0, flags: int = 0, splits: int = 1, big: Optional[bool] = None, lda=None, ldb=None, ldc=None) -> None:
    """mode NT: c[M,N] = a[M,K]·b[N,K]ᵀ; NN: c = a[M,K]·b[K,N]; TN: c = a[K,M]ᵀ·b[K,N]. a/b bf16; c bf16 or f32 (flag)."""
    _chk(a, torch.bfloat16, "a"); _chk(b, torch.bfloat16, "b")
    want = torch.float32 if flags & EPI_OUT_F32 else torch.bfloat16
    _chk(c, want, "c")
    lda = a.shape[-1] if lda is None else lda
    ldb = b.shape[-1] if ldb is None else ldb
    ldc = c.shape[-1] if ldc is None else ldc
    if big is None:
        big = (N % 128 == 0) and (M >= 2048) and (mode != TN or M % 128 == 0)
    variant = int(big) if not isinstance(big, bool) else (1 if big else GEMM_SMALL_VARIANT)
    ldm = mask.shape[-1] if mask is not None else 0
    _ck(native.hip().drl_gemm(mode, variant, _p(a), lda, _p(b), ldb, M, N, K, _p(c), ldc, _p(bias), _p(mask), ldm,
                              scale, flags, splits, _s()), "gemm")


# ---- large-tile GEMM (csrc/hip/gemm_big.hip): the big dense products of the R2D2 step -------------------------------
# tile ids: 0 = 256x256 (2 LDS stages) | 1 = 256x128 (3) | 2 = 128x256 (3) | 3 = 128x128 (4) | 4 = 256x128 (2) | 5 = 128x128 (3)
BIG_TILES = {0: (256, 256), 1: (256, 128), 2: (128, 256), 3: (128, 128), 4: (256, 128), 5: (128, 128)}
BIG_GEMM = True
BIG_GEMM_MIN_FLOP = 2.0e10


def big_tile_for(M: int, N: int, nz: int = 1, cus: int = 256) -> int:
    """Tile with the least (rounds over the CUs) x (tile area) x (measured inefficiency of the smaller tiles: one
    256 x 128 tile costs 0.64 of a 256 x 256 one, a 128 x 128 tile 0.36 - scripts/gemm_big_bench.py). ``cus``: the CUs
    that can take a workgroup - next to a persistent recurrence its 64 KB workgroups hold one CU each, and no tile of this
    kernel (96-144 KB of LDS) fits beside them."""
    best, best_cost = 3, None
    for t, ineff in ((0, 1.0), (1, 1.27), (2, 1.27), (3, 1.44)):
        bm, bn = BIG_TILES[t]
        blocks = -(-M // bm) * -(-N // bn) * nz
        cost = -(-blocks // max(1, cus)) * bm * bn * ineff
        if best_cost is None or cost < best_cost:
            best, best_cost = t, cost
    return best


def use_big_gemm(M: int, N: int, K: int) -> bool:
    return BIG_GEMM and 2.0 * M * N * K >= BIG_GEMM_MIN_FLOP


def gemm_big(mode: int, a: torch.Tensor, b: torch.Tensor, c: torch.Tensor, bias: Optional[torch.Tensor] = None,
             mask: Optional[torch.Tensor] = None, tile: Optional[int] = None, second=None, relu: bool = False,
             cus: int = 256) -> None:
    """c = a.b^T (NT: a [M,K], b [N,K]) | a.b (NN: a [M,K], b [K,N]) | a^T.b (TN: a [K,M], b [K,N]); a, b bf16 row-major
    2-D (row stride = shape[1]); c fp32 or bf16 [M,N]; ``bias`` fp32 [N] added, ``mask`` bf16 [M,N]: c = 0 where mask <= 0
    (ReLU backward), ``relu``: c = max(c, 0) after the bias - all in the epilogue, no second pass. ``second`` = (a2, b2, c2, bias2, mask2): a second problem of the
    same shape in the same launch (two networks fill the 256 CUs together where each alone would not). ``cus``: CUs free
    for this launch (the automatic tile choice counts its rounds over them, :func:`big_tile_for`)."""
    _chk(a, torch.bfloat16, "a"); _chk(b, torch.bfloat16, "b")
    assert a.dim() == 2 and b.dim() == 2 and c.dim() == 2
    if mode == NT:
        M, K = a.shape; N = b.shape[0]; assert b.shape[1] == K
    elif mode == NN:
        M, K = a.shape; N = b.shape[1]; assert b.shape[0] == K
    else:
        K, M = a.shape; N = b.shape[1]; assert b.shape[0] == K
    assert tuple(c.shape) == (M, N) and c.dtype in (torch.float32, torch.bfloat16) and c.is_contiguous()
    if bias is not None:
        _chk(bias, torch.float32, "bias"); assert bias.numel() == N
    if mask is not None:
        _chk(mask, torch.bfloat16, "mask"); assert tuple(mask.shape) == (M, N)
    a2 = b2 = c2 = bias2 = mask2 = None
    if second is not None:
        a2, b2, c2, bias2, mask2 = second
        _chk(a2, torch.bfloat16, "a2"); _chk(b2, torch.bfloat16, "b2")
        assert a2.shape == a.shape and b2.shape == b.shape and c2.shape == c.shape and c2.dtype == c.dtype and c2.is_contiguous()
        assert (bias2 is None) == (bias is None) and (mask2 is None) == (mask is None)
    t = big_tile_for(M, N, 2 if second is not None else 1, cus) if tile is None else tile
    _ck(native.hip().drl_gemm_big(mode, t, _p(a), a.shape[1], _p(b), b.shape[1], M, N, K, _p(c), N, _p(bias), _p(mask),
                                  N if mask is not None else 0, (1 if c.dtype == torch.float32 else 0) | (2 if relu else 0),
                                  _p(a2), _p(b2), _p(c2), _p(bias2), _p(mask2), _s()), "gemm_big")


# tiled conv2 / conv3 forward: 0 = register ring, 1..3 = LDS-DMA mainloop (128x64 x 3 stages / 128x64 x 4 / 64x64 x 4).
# Measured at 1024 frames: conv2 17.5 us ring vs 19.2 / 25.3 / 19.1; conv3 14.3 us ring vs 12.3 / 17.9 / 13.7.
CONV_FWD_TILED_VARIANT = {2: 0, 3: 1}


# conv2 / conv3 forward through the LDS-halo kernels (conv_halo.hip: whole input frames staged in LDS once, weights in
# registers); the implicit-GEMM tiles of gemm.hip remain for the no-ReLU / explicitly tiled calls (tests, kbench)
CONV_HALO = True
# ... and their input gradients (zero-padded dy through the same kernel, pre-transformed weights written by extra
# workgroups of the conv2 forward launch); the weight / bias gradients stay in the k_multi launch without its dgrad body
CONV_HALO_BWD = True
# variant 0 = two workgroups per CU, 1 = one per CU. Measured (scripts/conv_halo_bench.py, us, v0 / v1 / implicit GEMM):
#   conv2 1024+512 frames 19.2 / 17.9 / 24.5, 2 x 3840 frames 48.7 / 54.6 / 95.0;  conv3 15.5 (v1) / 20.3, 34.2 (v1) / 62.8
CONV_HALO_VARIANT = {2: -1, 3: 1}           # -1: by frame count (conv_halo_fwd)


import contextlib


# conv1 -> conv2 -> conv3 of a forward pass as ONE launch (conv_torso.hip). False: one launch per layer - the form the fused
# kernel is bit-compared with (tests/test_hip_kernels.py, tests/test_engine_gpu.py set it) and the A/B of profiles/bench_r3_torso_*.
TORSO_FUSED = True


@contextlib.contextmanager
def halo_busy(cus: int, lds_bytes: int = 66 * 1024):
    """The LDS-halo launches issued inside run next to a PERSISTENT kernel of another stream that holds ``cus`` CUs with
    ``lds_bytes`` of LDS each (the LSTM recurrences): their persistent grids are sized to what is resident at once
    (conv_halo.hip drl_halo_set_busy; conv_wgrad_halo takes the same through max_wgs). Host-side state, read at launch
    (capture) time."""
    lib = native.hip()
    if cus > 0:
        _ck(lib.drl_halo_set_busy(cus, lds_bytes), "halo_set_busy")
        _ck(lib.drl_torso_set_busy(cus), "torso_set_busy")      # the fused torso fills a CU's registers: it leaves those CUs out
    try:
        yield 256 - cus
    finally:
        if cus > 0:
            lib.drl_halo_set_busy(0, 0)
            lib.drl_torso_set_busy(0)


def conv_halo_fwd(layer: int, x1, f1: int, w1, b1, out1, x2=None, f2: int = 0, w2=None, b2=None, out2=None,
                  variant: Optional[int] = None, prep=None) -> None:
    """conv2 / conv3 (+bias+ReLU) of one or two networks in one launch of the LDS-halo kernel. prep (layer 2 only) = (w2, w3,
    wt2, wt3): extra workgroups of the launch also write the pre-transformed weights of the input-gradient kernels
    (conv_halo_wprep without a launch of its own)."""
    assert layer in (2, 3)
    ci, co = ((20 * 20 * 32, 81 * 64) if layer == 2 else (81 * 64, 49 * 64))
    for x, f, w, b, o in ((x1, f1, w1, b1, out1),) + (((x2, f2, w2, b2, out2),) if f2 else ()):
        _chk(x, torch.bfloat16, "x"); _chk(w, torch.bfloat16, "w"); _chk(b, torch.float32, "bias"); _chk(o, torch.bfloat16, "out")
        assert x.numel() >= f * ci and o.numel() >= f * co and w.numel() == 64 * (512 if layer == 2 else 576) and b.numel() == 64
    v = CONV_HALO_VARIANT[layer] if variant is None else variant
    if v < 0:
        v = 1 if f1 + f2 <= 2048 else 0
    pw2, pw3, pwt2, pwt3 = prep if prep is not None else (None, None, None, None)
    if prep is not None:
        assert layer == 2
        for t, n in ((pw2, 64 * 512), (pw3, 64 * 576), (pwt2, 64 * 512), (pwt3, 64 * 576)):
            _chk(t, torch.bfloat16, "prep"); assert t.numel() == n
    _ck(native.hip().drl_conv_halo_fwd(layer, _p(x1), f1, _p(w1), _p(b1), _p(out1), _p(x2) if f2 else None, f2,
                                       _p(w2) if f2 else None, _p(b2) if f2 else None, _p(out2) if f2 else None, v,
                                       _p(pw2), _p(pw3), _p(pwt2), _p(pwt3), _s()), "conv_halo_fwd")


def ingest_scatter(tail: torch.Tensor, segs, wins, plane_base: int) -> None:
    """One launch (ingest.hip) that scatters a staged experience block's tables out of ``tail`` (its device copy, uint8):
    segs = [(byte offset in tail, dst tensor of 32-bit elements, element count)], wins = [(byte offset of an int32 plane
    index table, dst int64 tensor, count)]: dst[i] = (src[i] + plane_base) * 7056."""
    ns, nw = len(segs), len(wins)
    assert ns <= 8 and nw <= 2 and tail.dtype == torch.uint8 and tail.is_cuda
    for off, dst, n in segs:
        assert dst.element_size() == 4 and dst.is_contiguous() and dst.numel() >= n and off + 4 * n <= tail.numel()
    for off, dst, n in wins:
        assert dst.dtype == torch.int64 and dst.is_contiguous() and dst.numel() >= n and off + 4 * n <= tail.numel()
    LL, VP, CI = ctypes.c_longlong, ctypes.c_void_p, ctypes.c_int
    _ck(native.hip().drl_ingest_scatter(
        _p(tail), ns, (LL * max(ns, 1))(*[s[0] for s in segs]), (VP * max(ns, 1))(*[s[1].data_ptr() for s in segs]),
        (CI * max(ns, 1))(*[s[2] for s in segs]), nw, (LL * max(nw, 1))(*[w[0] for w in wins]),
        (VP * max(nw, 1))(*[w[1].data_ptr() for w in wins]), (CI * max(nw, 1))(*[w[2] for w in wins]), int(plane_base), _s()),
        "ingest_scatter")


class _IngestArgs:
    """Reused ctypes argument arrays of ``ingest_block`` (building six arrays per block cost more than the launch)."""
    def __init__(self):
        LL, VP, CI = ctypes.c_longlong, ctypes.c_void_p, ctypes.c_int
        self.so, self.sd, self.sn = (LL * 8)(), (VP * 8)(), (CI * 8)()
        self.wo, self.wd, self.wn = (LL * 2)(), (VP * 2)(), (CI * 2)()


_ingest_args = threading.local()


def ingest_block(planes_dst: int, planes_src: int, planes_bytes: int, tail: torch.Tensor, tail_src: int, tail_bytes: int,
                 segs, wins, plane_base: int, stream: Optional[int] = None) -> None:
    """The device side of one staged experience block as ONE host call (ingest.hip) on ``stream`` (a raw handle; None: the
    current stream): DMA of the planes from pinned memory (``planes_src``, a host address) into the plane ring
    (``planes_dst``), DMA of the tables (``tail_src``) into the device buffer ``tail``, one scatter launch. Addresses are
    raw (``data_ptr()`` + byte offset: no tensor views are built per block); segs = [(byte offset in the tables, dst address
    of 32-bit words, count)], wins = [(byte offset of an int32 plane-index table, dst address of int64, count)]: dst[i] =
    (src[i] + plane_base) * 7056. The CALLER has checked the destination ranges against its tables."""
    ns, nw = len(segs), len(wins)
    assert ns <= 8 and nw <= 2 and tail.dtype == torch.uint8 and tail.is_cuda and 0 < tail_bytes <= tail.numel()
    a = getattr(_ingest_args, "a", None)
    if a is None:
        a = _ingest_args.a = _IngestArgs()
    for i, (off, dst, n) in enumerate(segs):
        a.so[i], a.sd[i], a.sn[i] = off, dst, n
    for i, (off, dst, n) in enumerate(wins):
        a.wo[i], a.wd[i], a.wn[i] = off, dst, n
    _ck(native.hip().drl_ingest_block(planes_dst, planes_src, planes_bytes, tail.data_ptr(), tail_src, tail_bytes, ns, a.so, a.sd,
                                      a.sn, nw, a.wo, a.wd, a.wn, plane_base, _s() if stream is None else stream), "ingest_block")


class TorsoNet:
    """One network's operands of the fused torso forward (``torso_fwd``): plane-store byte offsets (None = dense HWC
    stacks), frame count, the three conv weights (bf16; conv1 in the input form's k order) and biases (fp32), act3 out,
    and - only where a backward pass will read them - act1 / act2 arrays + the frame range ``save`` to write."""

    def __init__(self, frame_off, frames: int, w, b, act3, act1=None, act2=None, save=(0, 0)):
        assert len(w) == 3 and len(b) == 3
        for t, n in zip(w, (32 * 256, 64 * 512, 64 * 576)):
            _chk(t, torch.bfloat16, "w"); assert t.numel() == n
        for t, n in zip(b, (32, 64, 64)):
            _chk(t, torch.float32, "bias"); assert t.numel() == n
        _chk(act3, torch.bfloat16, "act3"); assert act3.numel() >= frames * 3136
        lo, hi = save
        if hi > lo:
            _chk(act1, torch.bfloat16, "act1"); _chk(act2, torch.bfloat16, "act2")
            assert 0 <= lo and hi <= frames and act1.numel() >= hi * 12800 and act2.numel() >= hi * 5184
        if frame_off is not None:
            _chk(frame_off, torch.int64, "frame_off"); assert frame_off.numel() >= frames
        self.frame_off, self.frames, self.save = frame_off, int(frames), (int(lo), int(hi))
        self._keep = (w, b, act1, act2, act3)
        self.w = (ctypes.c_void_p * 3)(*[t.data_ptr() for t in w])
        self.b = (ctypes.c_void_p * 3)(*[t.data_ptr() for t in b])
        self.a = (ctypes.c_void_p * 3)(act1.data_ptr() if hi > lo else None, act2.data_ptr() if hi > lo else None, act3.data_ptr())


def torso_fwd(x: torch.Tensor, net1: TorsoNet, net2: Optional[TorsoNet] = None, prep=None) -> None:
    """conv1 -> conv2 -> conv3 (+bias+ReLU each) of one or two networks over frames of ``x`` as ONE launch (conv_torso.hip):
    act1 / act2 live in LDS and reach HBM only for each network's ``save`` range. prep = (w2, w3, wt2, wt3): as
    ``conv_halo_fwd``."""
    _chk(x, torch.uint8, "x")
    if net2 is not None:
        assert (net1.frame_off is None) == (net2.frame_off is None)
    if net1.frame_off is None:
        assert x.numel() >= max(net1.frames, net2.frames if net2 else 0) * 28224
    pw2, pw3, pwt2, pwt3 = prep if prep is not None else (None, None, None, None)
    if prep is not None:
        for t, n in ((pw2, 64 * 512), (pw3, 64 * 576), (pwt2, 64 * 512), (pwt3, 64 * 576)):
            _chk(t, torch.bfloat16, "prep"); assert t.numel() == n
    n2 = net2
    _ck(native.hip().drl_torso_fwd(_p(x), _p(net1.frame_off), net1.frames, net1.w, net1.b, net1.a, net1.save[0], net1.save[1],
                                   _p(n2.frame_off) if n2 else None, n2.frames if n2 else 0, n2.w if n2 else None,
                                   n2.b if n2 else None, n2.a if n2 else None, n2.save[0] if n2 else 0, n2.save[1] if n2 else 0,
                                   _p(pw2), _p(pw3), _p(pwt2), _p(pwt3), _s()), "torso_fwd")


def torso_set_busy(cus: int) -> None:
    """CUs a concurrently running persistent kernel of another stream occupies: the fused torso leaves them out of its grid."""
    _ck(native.hip().drl_torso_set_busy(int(cus)), "torso_set_busy")


def conv_halo_wprep(w2: torch.Tensor, w3: torch.Tensor, wt2: torch.Tensor, wt3: torch.Tensor) -> None:
    """Flipped / transposed copies of the conv2 and conv3 weights for the LDS-halo input-gradient kernels (one tiny launch;
    the learners run it once per optimizer step): wt2 bf16 [4, 32, 2, 2, 64], wt3 bf16 [64, 3, 3, 64]."""
    for t, n in ((w2, 64 * 512), (w3, 64 * 576), (wt2, 64 * 512), (wt3, 64 * 576)):
        _chk(t, torch.bfloat16, "w"); assert t.numel() == n
    _ck(native.hip().drl_conv_halo_wprep(_p(w2), _p(w3), _p(wt2), _p(wt3), _s()), "conv_halo_wprep")


def conv_halo_dgrad(layer: int, dy: torch.Tensor, wt: torch.Tensor, act_in: torch.Tensor, dx: torch.Tensor, frames: int) -> None:
    """dx = relu'(act_in) * conv_transpose(dy, w) of conv2 / conv3 as a stride-1 convolution of the zero-padded dy through
    the LDS-halo kernel; wt = the layer's pre-transformed weights (conv_halo_wprep)."""
    assert layer in (2, 3)
    ny, nx = ((81 * 64, 400 * 32) if layer == 2 else (49 * 64, 81 * 64))
    _chk(dy, torch.bfloat16, "dy"); _chk(wt, torch.bfloat16, "wt"); _chk(act_in, torch.bfloat16, "act_in"); _chk(dx, torch.bfloat16, "dx")
    assert dy.numel() >= frames * ny and act_in.numel() >= frames * nx and dx.numel() >= frames * nx
    assert wt.numel() == 64 * (512 if layer == 2 else 576)
    _ck(native.hip().drl_conv_halo_dgrad(layer, _p(dy), frames, _p(wt), _p(act_in), _p(dx), _s()), "conv_halo_dgrad")


# conv2 / conv3 weight gradients through the LDS-halo kernel where the input gradients use theirs
CONV_HALO_WGRAD = True


def conv_wgrad_ws(layer: int, device) -> torch.Tensor:
    """Scratch of the LDS-halo weight-gradient kernel: one row of fragment-ordered accumulators per workgroup (32 / 40 MB).
    Launches that may run concurrently (different streams) need a buffer each."""
    lib = native.hip()
    lib.drl_conv_wgrad_halo_ws_floats.restype = ctypes.c_longlong
    return torch.empty(int(lib.drl_conv_wgrad_halo_ws_floats(layer)), device=device)


def conv_wgrad_halo(layer: int, x: torch.Tensor, dy: torch.Tensor, dw: torch.Tensor, db: torch.Tensor, frames: int,
                    ws: torch.Tensor, max_wgs: int = 0) -> None:
    """Weight + bias gradient of conv2 / conv3 through the LDS-halo kernel (conv_wgrad_halo.hip): dw [64, kh, kw, ci] fp32 and
    db [64] fp32 are ACCUMULATED into (the optimizer leaves them zeroed). x = the layer's forward input, dy = the gradient of
    its pre-activation output. ws = scratch (conv_wgrad_ws), max_wgs = workgroups (CUs) to use, 0 = all 256."""
    assert layer in (2, 3)
    nx, ny, K = ((400 * 32, 81 * 64, 512) if layer == 2 else (81 * 64, 49 * 64, 576))
    _chk(x, torch.bfloat16, "x"); _chk(dy, torch.bfloat16, "dy"); _chk(dw, torch.float32, "dw"); _chk(db, torch.float32, "db")
    assert x.numel() >= frames * nx and dy.numel() >= frames * ny and dw.numel() == 64 * K and db.numel() == 64
    _chk(ws, torch.float32, "ws")
    _ck(native.hip().drl_conv_wgrad_halo(layer, _p(x), _p(dy), _p(dw), _p(db), _p(ws), frames, max_wgs, _s()),
        "conv_wgrad_halo")


def conv1_wgrad_halo(store: torch.Tensor, frame_off: torch.Tensor, dy: torch.Tensor, dw: torch.Tensor, db: torch.Tensor,
                     frames: int, ws: torch.Tensor, max_wgs: int = 0) -> None:
    """Weight + bias gradient of conv1 over the replay's plane store (frame b = 4 consecutive 84 x 84 u8 planes from byte
    frame_off[b]) through the LDS-halo kernel: space-to-depth turns the 8 x 8 stride-4 convolution into a 2 x 2 stride-1 one
    over a 21 x 21 x 64 image, built in LDS while staging. dw [32, 4, 8, 8] fp32 (incl. the forward's 1 / 255) and db [32]
    are ACCUMULATED into. ws = conv_wgrad_ws(1, device)."""
    _chk(store, torch.uint8, "store"); _chk(frame_off, torch.int64, "frame_off"); _chk(dy, torch.bfloat16, "dy")
    _chk(dw, torch.float32, "dw"); _chk(db, torch.float32, "db"); _chk(ws, torch.float32, "ws")
    assert frame_off.numel() >= frames and dy.numel() >= frames * 400 * 32 and dw.numel() == 32 * 256 and db.numel() == 32
    _ck(native.hip().drl_conv1_wgrad_halo(_p(store), _p(frame_off), _p(dy), _p(dw), _p(db), _p(ws), frames, max_wgs, _s()),
        "conv1_wgrad_halo")


def conv_fwd(layer: int, x: torch.Tensor, w: torch.Tensor, bias: torch.Tensor, out: torch.Tensor, frames: int,
             frame_off: Optional[torch.Tensor] = None, relu: bool = True, tiled: Optional[int] = None, prep=None) -> None:
    """One conv layer (+bias, +ReLU). Default kernels: conv1 = the barrier-free direct kernel (conv_direct.hip: 20.6 us at
    1024 frames against 35.4 us tiled), conv2 / conv3 = the LDS-halo kernels (conv_halo.hip). ``tiled`` = variant of the
    LDS-tiled implicit GEMM of gemm.hip instead (0 = register ring, 1..3 = LDS-DMA mainloops): the A/B baseline, and the
    only form without ReLU."""
    _chk(w, torch.bfloat16, "w"); _chk(bias, torch.float32, "bias"); _chk(out, torch.bfloat16, "out")
    _chk(x, torch.uint8 if layer == 1 else torch.bfloat16, "x")
    if layer != 1 and CONV_HALO and relu and tiled is None:
        conv_halo_fwd(layer, x, frames, w, bias, out, prep=prep)
        return
    assert prep is None
    if layer == 1 and relu and tiled is None:
        _ck(native.hip().drl_conv1_fwd(_p(x), _p(w), _p(bias), _p(out), frames, _p(frame_off), _s()), "conv1_fwd")
        return
    if tiled is None:
        tiled = CONV_FWD_TILED_VARIANT.get(layer, 0)
    _ck(native.hip().drl_conv_fwd(layer, _p(x), _p(frame_off), frames, _p(w), _p(bias), _p(out),
                                  (1 if relu else 0) | (tiled << 8), _s()), "conv_fwd")


def conv_fwd_dual(layer: int, x1, f1: int, w1, b1, out1, x2, f2: int, w2, b2, out2, frame_off1=None, frame_off2=None,
                  prep=None) -> None:
    """Forward of one conv layer (+bias+ReLU) of TWO networks in one launch (Ape-X learner: online net on [s ; s'] and
    target net on s'). layer 1: x1 is the frame store, x2 is ignored, the two nets gather through frame_off1 / frame_off2."""
    for w, b, o in ((w1, b1, out1), (w2, b2, out2)):
        _chk(w, torch.bfloat16, "w"); _chk(b, torch.float32, "bias"); _chk(o, torch.bfloat16, "out")
    if layer == 1:
        _chk(x1, torch.uint8, "x")
        _ck(native.hip().drl_conv1_fwd_dual(_p(x1), _p(frame_off1), f1, _p(w1), _p(b1), _p(out1), _p(frame_off2), f2, _p(w2),
                                            _p(b2), _p(out2), _s()), "conv1_fwd_dual")
        return
    _chk(x1, torch.bfloat16, "x1"); _chk(x2, torch.bfloat16, "x2")
    if CONV_HALO:
        conv_halo_fwd(layer, x1, f1, w1, b1, out1, x2, f2, w2, b2, out2, prep=prep)
        return
    assert prep is None
    _ck(native.hip().drl_conv_fwd_dual(layer, _p(x1), f1, _p(w1), _p(b1), _p(out1), _p(x2), f2, _p(w2), _p(b2), _p(out2), _s()),
        "conv_fwd_dual")


def conv1_fwd_pair(x, frame_off, frames: int, w1, b1, out1, w2, b2, out2) -> None:
    """conv1 (+bias+ReLU) of TWO networks over the SAME frames of the store ``x`` (addressed through ``frame_off``): every
    input tile is read and converted once and multiplied with both weight sets (R2D2 learner: online + target)."""
    _chk(x, torch.uint8, "x"); _chk(frame_off, torch.int64, "frame_off")
    for w, b, o in ((w1, b1, out1), (w2, b2, out2)):
        _chk(w, torch.bfloat16, "w"); _chk(b, torch.float32, "bias"); _chk(o, torch.bfloat16, "out")
        assert o.numel() == frames * 400 * 32
    assert frame_off.numel() >= frames
    _ck(native.hip().drl_conv1_fwd_pair(_p(x), _p(frame_off), frames, _p(w1), _p(b1), _p(out1), _p(w2), _p(b2), _p(out2), _s()),
        "conv1_fwd_pair")


def gemm_nt_dual(a1, w1, b1, c1, a2, w2, b2, c2, N: int, K: int, flags: int = 0) -> None:
    """c_i = epi(a_i[M_i,K] . w_i[N,K]^T + b_i) for two (a, w, b, c) sets in one launch (64x64 LDS-DMA tiles)."""
    for a, w, c in ((a1, w1, c1), (a2, w2, c2)):
        _chk(a, torch.bfloat16, "a"); _chk(w, torch.bfloat16, "w")
        _chk(c, torch.float32 if flags & EPI_OUT_F32 else torch.bfloat16, "c")
        assert a.shape[-1] == K and w.shape == (N, K) and c.shape == (a.shape[0], N)
    _ck(native.hip().drl_gemm_nt_dual(_p(a1), a1.shape[0], _p(w1), _p(b1), _p(c1), _p(a2), a2.shape[0], _p(w2), _p(b2), _p(c2),
                                      N, K, flags, _s()), "gemm_nt_dual")


def head_fwd_dual(h1, wa1, ba1, wv1, bv1, q1, h2, wa2, ba2, wv2, bv2, q2) -> None:
    """Dueling head (Q values only) of two networks in one launch."""
    (M1, A), (M2, A2) = q1.shape, q2.shape
    assert A == A2 and h1.shape == (M1, 1024) and h2.shape == (M2, 1024) and wa1.shape == (A, 512) and wa2.shape == (A, 512)
    for h, q in ((h1, q1), (h2, q2)):
        _chk(h, torch.bfloat16, "h"); _chk(q, torch.float32, "q")
    _ck(native.hip().drl_head_fwd_dual(_p(h1), M1, _p(wa1), _p(ba1), _p(wv1), _p(bv1), _p(q1), _p(h2), M2, _p(wa2), _p(ba2),
                                       _p(wv2), _p(bv2), _p(q2), A, _s()), "head_fwd_dual")


def conv_wgrad(layer: int, x: torch.Tensor, dy: torch.Tensor, dw: torch.Tensor, frames: int,
               frame_off: Optional[torch.Tensor] = None, splits: int = 64) -> None:
    _chk(dy, torch.bfloat16, "dy"); _chk(dw, torch.float32, "dw")
    _ck(native.hip().drl_conv_wgrad(layer, _p(x), _p(frame_off), frames, _p(dy), _p(dw), splits, _s()), "conv_wgrad")


def conv_dgrad(layer: int, dy: torch.Tensor, w: torch.Tensor, act_in: Optional[torch.Tensor], dx: torch.Tensor,
               frames: int) -> None:
    """Implicit input gradient of conv2 / conv3 in ONE launch (no dCol matrix, no col2im pass):
    dx = relu'(act_in) * conv_transpose(dy, w); act_in None = no mask."""
    _chk(dy, torch.bfloat16, "dy"); _chk(w, torch.bfloat16, "w"); _chk(dx, torch.bfloat16, "dx")
    _ck(native.hip().drl_conv_dgrad(layer, _p(dy), frames, _p(w), _p(act_in), _p(dx), _s()), "conv_dgrad")


# body configuration of the fused launches (see drl_fc_bwd / drl_conv_bwd), measured at B = 512 (scripts/kbench.py --only
# fused): fc1 38.6 us with both ring-4 bodies -> 29.7 us with ring-2 bodies, dW issued first (fewer VGPRs: more
# resident workgroups for BOTH bodies); conv2 40.5 -> 29.8 us with a 64-deep ring-2 wgrad body issued first (32 KB LDS
# instead of 64 KB for every workgroup of the launch); conv3 25.2 -> 24.6 us. Round 3, with the riders of the FC launch in
# front of its GEMM bodies: the dX tiles (K = 1024, the longer loop) before the dW tiles (K = 512) - 0.1899 -> 0.1868 ms per
# step; the conv launches keep their order (scripts/debug/fc_order_ab.py, profiles/apex_riders_first_r3.txt).
FC_BWD_VARIANT = 2
CONV_BWD_VARIANT = {1: 0, 2: 4, 3: 2}


class CopyExtra(ctypes.Structure):
    """ctypes mirror of ``CopyExtra`` (csrc/hip/optim.hip): up to four small device copies done by one extra block of the
    optimizer launch (hand-over of the pre-drawn PER batch)."""
    _fields_ = [("n", ctypes.c_int), ("pad", ctypes.c_int), ("dst", ctypes.c_void_p * 4), ("src", ctypes.c_void_p * 4),
                ("bytes", ctypes.c_longlong * 4)]


def copy_extra(pairs) -> CopyExtra:
    """pairs = [(dst, src), ...] (<= 4, same shapes / dtypes, byte sizes multiples of 16)."""
    e = CopyExtra()
    e.n = len(pairs)
    assert e.n <= 4
    for i, (d, s_) in enumerate(pairs):
        nb = d.numel() * d.element_size()
        assert d.dtype == s_.dtype and d.numel() == s_.numel() and nb % 16 == 0 and d.is_contiguous() and s_.is_contiguous()
        e.dst[i], e.src[i], e.bytes[i] = d.data_ptr(), s_.data_ptr(), nb
    return e


class SampleExtra(ctypes.Structure):
    """ctypes mirror of ``SampleExtra`` (csrc/hip/sumtree_body.h): the PER sample of the NEXT step, drawn by extra blocks
    of the conv1 backward launch (conv_bwd(1, ..., sample=...))."""
    _fields_ = [("kind", ctypes.c_int), ("n_levels", ctypes.c_int), ("base", ctypes.c_void_p),
                ("off", ctypes.c_longlong * 6), ("cap", ctypes.c_longlong), ("st", ctypes.c_void_p),
                ("seed", ctypes.c_ulonglong), ("B", ctypes.c_int), ("pad", ctypes.c_int), ("idx_out", ctypes.c_void_p),
                ("prio_out", ctypes.c_void_p), ("gen", ctypes.c_void_p), ("gen_out", ctypes.c_void_p),
                ("frame_off", ctypes.c_void_p), ("win", ctypes.c_void_p), ("begin_n_items", ctypes.c_void_p),
                ("is_n_scale", ctypes.c_float), ("pad2", ctypes.c_float)]


def sample_extra(tree: torch.Tensor, lay: TreeLayout, st: torch.Tensor, seed: int, idx_out: torch.Tensor,
                 prio_out: torch.Tensor, gen, gen_out, frame_off, win: torch.Tensor, begin_n_items: torch.Tensor,
                 is_n_scale: float) -> SampleExtra:
    """Arguments of ``sumtree_sample(..., begin_n_items=..., is_n_scale=...)`` as a record for ``conv_bwd(1, ..., sample=...)``."""
    _chk(tree, torch.float32, "tree"); _chk(idx_out, torch.int32, "idx_out"); _chk(prio_out, torch.float32, "prio_out")
    e = SampleExtra()
    e.kind, e.n_levels, e.base, e.cap = 1, lay.n_levels, tree.data_ptr(), lay.capacity
    for i in range(6):
        e.off[i] = int(lay.offsets[i]) if i < lay.n_levels else 0
    e.st, e.seed, e.B = st.data_ptr(), int(seed), idx_out.numel()
    e.idx_out, e.prio_out = idx_out.data_ptr(), prio_out.data_ptr()
    e.gen = gen.data_ptr() if gen is not None else None
    e.gen_out = gen_out.data_ptr() if gen_out is not None else None
    e.frame_off = frame_off.data_ptr() if frame_off is not None else None
    _chk(win, torch.int64, "win"); assert win.numel() == 2 * lay.capacity
    e.win, e.begin_n_items, e.is_n_scale = win.data_ptr(), begin_n_items.data_ptr(), float(is_n_scale)
    return e


class TreeExtra(ctypes.Structure):
    """ctypes mirror of ``TreeExtra`` (csrc/hip/gemm.hip): one stage of the PER priority write-back that rides along with
    a fused conv-backward launch. kind 1 = leaves + level 1, 2 = levels >= 2 (+ step_end), 3 = step_end only."""
    _fields_ = [("kind", ctypes.c_int), ("n_levels", ctypes.c_int), ("base", ctypes.c_void_p),
                ("off", ctypes.c_longlong * 6), ("cap", ctypes.c_longlong), ("idx", ctypes.c_void_p),
                ("val", ctypes.c_void_p), ("n", ctypes.c_int), ("pad", ctypes.c_int), ("gen", ctypes.c_void_p),
                ("gen_expected", ctypes.c_void_p), ("st_end", ctypes.c_void_p),
                ("sq_g", ctypes.c_void_p), ("sq_n", ctypes.c_longlong), ("sq_partial", ctypes.c_void_p),
                ("zero_g", ctypes.c_void_p), ("zero_n", ctypes.c_longlong), ("sq_blocks", ctypes.c_int), ("pad2", ctypes.c_int)]


def tree_extra(kind: int, tree: Optional[torch.Tensor] = None, lay: Optional[TreeLayout] = None, idx: Optional[torch.Tensor] = None,
               val: Optional[torch.Tensor] = None, gen: Optional[torch.Tensor] = None, gen_expected: Optional[torch.Tensor] = None,
               end_state: Optional[torch.Tensor] = None, sumsq=None, zero: Optional[torch.Tensor] = None) -> TreeExtra:
    """kind 0: no tree stage (only the ride-along work below). ``sumsq`` = (g, partial): partial[b] = sum of squares of
    workgroup b's share of the fp32 gradient range ``g``, for b < partial.numel() (gradient-norm partials without a launch
    of their own; the optimizer reads them, ``rmsprop_step(partial, n_partial, late=...)``); ``zero``: an fp32 range the
    same workgroups set to zero."""
    e = TreeExtra()
    e.kind = kind
    if kind:
        _chk(idx, torch.int32, "idx"); _chk(val, torch.float32, "val")
        assert idx.numel() == val.numel()
        e.n_levels, e.base, e.cap = lay.n_levels, tree.data_ptr(), lay.capacity
        for i in range(6):
            e.off[i] = int(lay.offsets[i]) if i < lay.n_levels else 0
        e.idx, e.val, e.n = idx.data_ptr(), val.data_ptr(), idx.numel()
        e.gen = gen.data_ptr() if gen is not None else None
        e.gen_expected = gen_expected.data_ptr() if gen_expected is not None else None
        e.st_end = end_state.data_ptr() if end_state is not None else None
    if sumsq is not None:
        g, partial = sumsq
        _chk(g, torch.float32, "sumsq g"); _chk(partial, torch.float32, "sumsq partial")
        assert g.numel() % 4 == 0 and g.numel() > 0 and 0 < partial.numel() <= 1024 and g.data_ptr() % 16 == 0
        e.sq_g, e.sq_n, e.sq_partial, e.sq_blocks = g.data_ptr(), g.numel(), partial.data_ptr(), partial.numel()
        if zero is not None:
            _chk(zero, torch.float32, "zero")
            assert zero.numel() % 4 == 0 and zero.data_ptr() % 16 == 0
            e.zero_g, e.zero_n = zero.data_ptr(), zero.numel()
    else:
        assert zero is None
    return e


def conv_bwd(layer: int, x: torch.Tensor, dy: torch.Tensor, w: Optional[torch.Tensor], dw: torch.Tensor, db: torch.Tensor,
             dx: Optional[torch.Tensor], frames: int, frame_off: Optional[torch.Tensor] = None, splits: int = 32,
             variant: Optional[int] = None, extra: Optional["TreeExtra"] = None,
             sample: Optional["SampleExtra"] = None) -> None:
    """Whole backward of a conv layer as ONE launch (k_multi in gemm.hip): implicit input gradient dx (layers 2/3, masked
    by relu'(x)), split-K weight gradient accumulated into the ZEROED dw, bias gradient accumulated into the zeroed db.
    x = the layer's forward input (layer 1: uint8 frames, optionally via frame_off byte offsets)."""
    _chk(dy, torch.bfloat16, "dy"); _chk(dw, torch.float32, "dw"); _chk(db, torch.float32, "db")
    if layer != 1:
        _chk(x, torch.bfloat16, "x")
        if dx is not None:                       # dx None: weight + bias gradient only (input gradient: conv_halo_dgrad)
            _chk(w, torch.bfloat16, "w"); _chk(dx, torch.bfloat16, "dx")
    variant = CONV_BWD_VARIANT[layer] if variant is None else variant
    _ck(native.hip().drl_conv_bwd(layer, _p(x), _p(frame_off), frames, _p(dy), _p(w), _p(dw), _p(db), _p(dx),
                                  splits | (variant << 16), ctypes.byref(extra) if extra is not None else None,
                                  ctypes.byref(sample) if sample is not None else None, _s()), "conv_bwd")


def fc_bwd(dy: torch.Tensor, x: torch.Tensor, w: torch.Tensor, dw: torch.Tensor, db: torch.Tensor, dx: torch.Tensor,
           M: int, N: int, K: int, head=None, variant: Optional[int] = None) -> None:
    """Whole backward of y = relu-input FC layer as ONE launch: dx[M,K] = relu'(x) * (dy[M,N] @ w[N,K]), dw[N,K] = dy^T @ x
    (overwritten), db[N] += colsum(dy). head = (dq, h, gw_adv, gb_adv, gw_val, gb_val) additionally runs the dueling
    head's weight-gradient kernel inside the same launch (it only depends on dq and h)."""
    _chk(dy, torch.bfloat16, "dy"); _chk(x, torch.bfloat16, "x"); _chk(w, torch.bfloat16, "w")
    _chk(dw, torch.float32, "dw"); _chk(db, torch.float32, "db"); _chk(dx, torch.bfloat16, "dx")
    if head is not None:
        dq, h, gw_adv, gb_adv, gw_val, gb_val = head
        _chk(dq, torch.float32, "dq"); _chk(h, torch.bfloat16, "h")
        hargs = (_p(dq), _p(h), dq.shape[1], _p(gw_adv), _p(gb_adv), _p(gw_val), _p(gb_val))
    else:
        hargs = (None, None, 0, None, None, None, None)
    _ck(native.hip().drl_fc_bwd(_p(dy), _p(x), _p(w), _p(dw), _p(db), _p(dx), M, N, K, *hargs,
                                FC_BWD_VARIANT if variant is None else variant, _s()), "fc_bwd")


def colsum(dy: torch.Tensor, db: torch.Tensor, M: int, N: int) -> None:
    _chk(dy, torch.bfloat16, "dy"); _chk(db, torch.float32, "db")
    _ck(native.hip().drl_colsum(_p(dy), M, N, _p(db), _s()), "colsum")


# ---- LSTM ----------------------------------------------------------------------------------------------
def lstm_step(h_prev, w_hh, xg_t, c_prev, h_out, c_out, gates_out=None) -> None:
    """One recurrent step on gate-interleaved weights: (h,c) <- cell(xg_t + h_prev·W_hhᵀ, c_prev)."""
    M, Hd = h_prev.shape
    _chk(h_prev, torch.bfloat16, "h_prev"); _chk(w_hh, torch.bfloat16, "w_hh"); _chk(xg_t, torch.float32, "xg_t")
    _chk(c_prev, torch.float32, "c_prev"); _chk(h_out, torch.bfloat16, "h_out"); _chk(c_out, torch.float32, "c_out")
    assert w_hh.shape == (4 * Hd, Hd) and xg_t.shape == (M, 4 * Hd)
    _ck(native.hip().drl_lstm_step(_p(h_prev), _p(w_hh), _p(xg_t), _p(c_prev), _p(h_out), _p(c_out), _p(gates_out), M, Hd, _s()),
        "lstm_step")


def lstm_hx(device) -> torch.Tensor:
    """Scratch of one network for :func:`lstm_seq_fwd_x`: 256 KB (two 64 KB parity slots + the arrival counter)."""
    return torch.zeros(32768, dtype=torch.int64, device=device)


def lstm_seq_fwd_x(nets, T: int, M: int, Hd: int, err: torch.Tensor, groups: int = 1) -> None:
    """Persistent recurrence (csrc/hip/lstm_seq.hip): ONE launch runs all T steps of 1 or 2 networks; h is handed between
    the workgroups through an exchange buffer laid out in consumer order. ``nets`` = list of (xg [T,M,4H] f32, w_hh [4H,H]
    bf16 gate-interleaved, hs [T+1,M,H] bf16, cs [T+1,M,H] f32, gates [T,M,4H] bf16 | None, hx) or (..., hx, bias);
    hs[0] / cs[0] hold the initial state; ``err`` (1 x int32) turns non-zero if an inter-workgroup wait expired. hx from :func:`lstm_hx` (one per network, not shared between launches
    that may run at the same time); ``bias`` ([4H] fp32, gate-interleaved like xg's columns) is added to xg on the fly.
    ``groups`` (1, 2, 4, dividing M/16): independent row groups of 32 workgroups per network - same numbers, the h loads
    of a step spread over more CUs (for launches that have the chip to themselves)."""
    assert 1 <= len(nets) <= 2
    _chk(err, torch.int32, "err")
    args = []
    for net in nets:
        xg, w_hh, hs, cs, gates, hx = net[:6]
        bias = net[6] if len(net) > 6 else None
        _chk(xg, torch.float32, "xg"); _chk(w_hh, torch.bfloat16, "w_hh"); _chk(hs, torch.bfloat16, "hs"); _chk(cs, torch.float32, "cs")
        _chk(hx, torch.int64, "hx")
        assert xg.numel() == T * M * 4 * Hd and hs.numel() == (T + 1) * M * Hd and cs.numel() == (T + 1) * M * Hd
        assert w_hh.shape == (4 * Hd, Hd) and (gates is None or gates.numel() == T * M * 4 * Hd) and hx.numel() == 32768
        if bias is not None:
            _chk(bias, torch.float32, "bias")
            assert bias.numel() == 4 * Hd
        args += [_p(xg), _p(w_hh), _p(hs), _p(cs), _p(gates), _p(bias), _p(hx)]
    if len(nets) == 1:
        args += [None] * 7
    _ck(native.hip().drl_lstm_seq_fwd_x(*args, len(nets), T, M, Hd, int(groups), _p(err), _s()), "lstm_seq_fwd_x")


def lstm_bx(device) -> torch.Tensor:
    """Scratch of :func:`lstm_bptt`: 768 KB (two 256 KB parity slots of gate gradients + arrival flags)."""
    return torch.zeros(98304, dtype=torch.int64, device=device)


def lstm_bptt(dhs, gates, cs, dG, w_hh, dh_rec, dc, bx, err, k_hi: int, steps: int, first: bool, carry: bool,
              groups: int = 0) -> None:
    """Persistent BPTT (csrc/hip/lstm_seq.hip k_lstm_bptt): window steps k_hi .. k_hi - steps + 1 in ONE launch.
    dhs [L,M,H] bf16 (dL/dh from the heads), gates [L,M,4H] bf16, cs [L+1,M,H] f32 (cs[k] = c before step k), dG [L,M,4H]
    bf16 out, w_hh [4H,H] bf16; dh_rec / dc [M,H] f32 carry the chain between launches (``first``: start from zero,
    ``carry``: write them for the launch that continues). ``groups``: independent row groups of 32 workgroups each
    (1, 2 or 4, dividing M/16; 0 = the library default)."""
    L, M, Hd = dhs.shape
    _chk(dhs, torch.bfloat16, "dhs"); _chk(gates, torch.bfloat16, "gates"); _chk(cs, torch.float32, "cs"); _chk(dG, torch.bfloat16, "dG")
    _chk(w_hh, torch.bfloat16, "w_hh"); _chk(dh_rec, torch.float32, "dh_rec"); _chk(dc, torch.float32, "dc")
    _chk(bx, torch.int64, "bx"); _chk(err, torch.int32, "err")
    assert gates.shape == (L, M, 4 * Hd) and cs.shape == (L + 1, M, Hd) and dG.shape == (L, M, 4 * Hd) and w_hh.shape == (4 * Hd, Hd)
    assert dh_rec.shape == (M, Hd) and dc.shape == (M, Hd) and bx.numel() >= 98304 and 0 <= k_hi - steps + 1 and k_hi < L
    _ck(native.hip().drl_lstm_bptt(_p(dhs), _p(gates), _p(cs), _p(dG), _p(w_hh), _p(dh_rec), _p(dc), _p(bx), k_hi, steps, M, Hd,
                                   int(first), int(carry), int(groups), _p(err), _s()), "lstm_bptt")


def lstm_seq_supported(M: int, Hd: int) -> bool:
    return Hd == 512 and 0 < M <= 64 and M % 16 == 0


def lstm_bwd_point(dh_head, dh_rec, dc, gates, c_t, c_prev, dgates) -> None:
    B, Hd = c_t.shape
    _chk(dh_rec, torch.float32, "dh_rec"); _chk(dc, torch.float32, "dc"); _chk(gates, torch.bfloat16, "gates")
    _chk(dgates, torch.bfloat16, "dgates")
    _ck(native.hip().drl_lstm_bwd_point(_p(dh_head), _p(dh_rec), _p(dc), _p(gates), _p(c_t), _p(c_prev), _p(dgates), B, Hd, _s()),
        "lstm_bwd_point")


def lstm_bwd_step_supported(B: int, Hd: int) -> bool:
    return Hd == 512 and 0 < B <= 64


def lstm_bwd_step(dh_head, dh_rec_in, dh_rec_out, dh_rec_zero, dc_in, dc_out, gates, c_t, c_prev, dgates, w_hh) -> None:
    """One BPTT step in ONE launch (gemm.hip k_lstm_bwd_step): dgates = pointwise(dh_head + dh_rec_in, dc_in, ...),
    dc_out, and dh_rec_out += dgates @ w_hh. dh_rec_out must be zero on entry; dh_rec_zero (a third buffer) is zeroed for
    the step after; the three dh_rec buffers and the two dc buffers must be distinct."""
    B, Hd = c_t.shape
    for t, n in ((dh_rec_in, "dh_rec_in"), (dh_rec_out, "dh_rec_out"), (dh_rec_zero, "dh_rec_zero"), (dc_in, "dc_in"), (dc_out, "dc_out")):
        _chk(t, torch.float32, n)
    _chk(gates, torch.bfloat16, "gates"); _chk(dgates, torch.bfloat16, "dgates"); _chk(w_hh, torch.bfloat16, "w_hh")
    _ck(native.hip().drl_lstm_bwd_step(_p(dh_head), _p(dh_rec_in), _p(dh_rec_out), _p(dh_rec_zero), _p(dc_in), _p(dc_out),
                                       _p(gates), _p(c_t), _p(c_prev), _p(dgates), _p(w_hh), B, Hd, _s()), "lstm_bwd_step")


def seq_offsets(idx, T: int, seq_bytes: int, plane_bytes: int, off) -> None:
    """off[t * B + b] = idx[b] * seq_bytes + t * plane_bytes: plane-window offsets of a sampled sequence batch, time-major."""
    _chk(idx, torch.int32, "idx"); _chk(off, torch.int64, "off")
    assert off.numel() >= idx.numel() * T
    _ck(native.hip().drl_seq_offsets(_p(idx), idx.numel(), T, seq_bytes, plane_bytes, _p(off), _s()), "seq_offsets")


def seq_win_offsets(idx, win, T: int, off) -> None:
    """off[t * B + b] = win[idx[b], t]: time-major plane-window byte offsets of a sampled sequence batch from the replay's
    per-slot window table (sequences that share planes: replay/device_seq.py)."""
    _chk(idx, torch.int32, "idx"); _chk(win, torch.int64, "win"); _chk(off, torch.int64, "off")
    assert win.shape[-1] == T and off.numel() >= idx.numel() * T
    _ck(native.hip().drl_seq_win_offsets(_p(idx), _p(win), idx.numel(), T, _p(off), _s()), "seq_win_offsets")


def gather_state(src, idx, dst_f32=None, dst_bf16=None) -> None:
    _chk(src, torch.float32, "src"); _chk(idx, torch.int32, "idx")
    _ck(native.hip().drl_gather_state(_p(src), _p(idx), idx.numel(), src.shape[-1], _p(dst_f32), _p(dst_bf16), _s()), "gather_state")


def gather_state2(h0, c0, idx, hs_a, cs_a, hs_b=None, cs_b=None, sums=()) -> None:
    """Stored recurrent state of the sampled sequences -> initial-state rows (h as bf16, c as fp32) of one or two
    networks, one launch. ``sums``: up to two (a, b, out) triples of equally long fp32 vectors, out = a + b by the same
    launch (the LSTM's b_ih + b_hh of both networks)."""
    _chk(h0, torch.float32, "h0"); _chk(c0, torch.float32, "c0"); _chk(idx, torch.int32, "idx")
    _chk(hs_a, torch.bfloat16, "hs_a"); _chk(cs_a, torch.float32, "cs_a")
    if hs_b is not None:
        _chk(hs_b, torch.bfloat16, "hs_b"); _chk(cs_b, torch.float32, "cs_b")
    assert len(sums) <= 2
    n_add, ptrs = 0, []
    for a, b, out in sums:
        _chk(a, torch.float32, "sum a"); _chk(b, torch.float32, "sum b"); _chk(out, torch.float32, "sum out")
        assert a.numel() == b.numel() == out.numel() and (n_add == 0 or n_add == a.numel())
        n_add = a.numel()
        ptrs += [_p(a), _p(b), _p(out)]
    ptrs += [None] * (6 - len(ptrs))
    _ck(native.hip().drl_gather_state2(_p(h0), _p(c0), _p(idx), idx.numel(), h0.shape[-1], _p(hs_a), _p(cs_a), _p(hs_b), _p(cs_b),
                                       *ptrs, n_add, _s()), "gather_state2")


def add2(a, b, out) -> None:
    _ck(native.hip().drl_add2(_p(a), _p(b), _p(out), a.numel(), _s()), "add2")

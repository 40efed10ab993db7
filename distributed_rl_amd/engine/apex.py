"""ApexEngine — the graph-captured gfx950 Ape-X learner step (replaces the hot loop of ape_x/learner.py:63-99).

One step = one hipGraph replay (``torch.cuda.CUDAGraph``) of hand-written kernels; nothing returns to the host
between PER sampling and the priority write-back:

  step_begin → sum-tree sample (idx, prio, gen, frame byte offsets)
  online  torso on [s ; s'] (2B frames, conv1 reads uint8 straight from the replay store) → FC1 → head → Q
  target  torso on s' → Q_target
  fused TD (IS weights, double-DQN target, Huber, loss, dL/dQ, new priorities)
  backward: head + FC1 in one launch (dW, db, dX·ReLU') → conv3, conv2, conv1: one launch each (dX·ReLU' as a convolution
  of dY with the flipped weights + split-K dW + bias column sum; from 1024 frames dX runs in the LDS-halo kernel)
  [RCCL all-reduce of the flat gradient buffer when world_size > 1]
  global-norm clip + RMSprop + bf16 re-cast (2 launches over the flat buffers)
  sum-tree priority write-back → step_end → (every target_update steps) target ← online (one flat copy)

Parameters, gradients and optimizer state are flat fp32 buffers in kernel layout (models/packing.py); the MFMA
kernels read a bf16 shadow that the optimizer kernel refreshes. Checkpoints go through ``state_dict()`` in the
reference's format.
"""
from __future__ import annotations

from typing import Optional

import torch

from ..config import ApexConfig
from ..models import DuelingDQN
from ..models.packing import ParamPack, apex_layout
from ..ops import hip as H
from ..replay.device_per import DevicePER
from .guard import StepGuard
from .shard import ShardedIS
from .streams import capture_graph


class TorsoBuffers:
    """Activations of the Nature-CNN torso + FC1 + head for ``n`` frames (NHWC bf16)."""

    def __init__(self, n: int, n_action: int, device):
        bf = dict(dtype=torch.bfloat16, device=device)
        self.n = n
        self.act1 = torch.zeros(n, 20, 20, 32, **bf)
        self.act2 = torch.zeros(n, 9, 9, 64, **bf)
        self.act3 = torch.zeros(n, 3136, **bf)
        self.h = torch.zeros(n, 1024, **bf)
        self.q = torch.zeros(n, n_action, device=device)


def torso_net(cache: dict, frame_off: Optional[torch.Tensor], n: int, wb: ParamPack, wf: ParamPack, act1, act2, act3,
              save=(0, 0), w1=None) -> "H.TorsoNet":
    """The fused torso kernel's operand block for one network (ops.hip.TorsoNet), built once per distinct set of buffers:
    ``cache`` (any dict that lives as long as the buffers) is keyed by the addresses involved."""
    key = (frame_off.data_ptr() if frame_off is not None else 0, n, wb.flat.data_ptr(), wf.flat.data_ptr(), act3.data_ptr(),
           act1.data_ptr() if act1 is not None else 0, tuple(save), w1.data_ptr() if w1 is not None else 0)
    net = cache.get(key)
    if net is None:
        net = cache[key] = H.TorsoNet(frame_off, n, (w1 if w1 is not None else wb.view("conv1.w"), wb.view("conv2.w"), wb.view("conv3.w")),
                                      (wf.view("conv1.b"), wf.view("conv2.b"), wf.view("conv3.b")), act3, act1, act2, save)
    return net


def torso_forward(x_u8: torch.Tensor, frame_off: Optional[torch.Tensor], n: int, wb: ParamPack, wf: ParamPack,
                  buf: TorsoBuffers, action_out=None, eps=None, eps_mod: int = 0, seed: int = 0, ctr=None, w1=None) -> None:
    """wb = bf16 weights, wf = fp32 params (biases). 3 launches: the fused torso (conv1-3, activations in LDS),
    FC1(+bias+ReLU), head(+dueling[+ε-greedy]). frame_off given: x_u8 is a replay store of single planes and conv1 uses the
    pack's (c, kh, kw) weights; dense HWC stacks (actors, frame_off None) need ``w1`` = the HWC-ordered copy
    (models.packing.conv1_hwc)."""
    assert (frame_off is None) == (w1 is not None), "dense HWC input <-> HWC-ordered conv1 weights"
    if H.TORSO_FUSED:
        H.torso_fwd(x_u8, torso_net(buf.__dict__.setdefault("_torso", {}), frame_off, n, wb, wf, None, None, buf.act3, w1=w1))
    else:
        H.conv_fwd(1, x_u8, w1 if w1 is not None else wb.view("conv1.w"), wf.view("conv1.b"), buf.act1, n, frame_off)
        H.conv_fwd(2, buf.act1, wb.view("conv2.w"), wf.view("conv2.b"), buf.act2, n)
        H.conv_fwd(3, buf.act2, wb.view("conv3.w"), wf.view("conv3.b"), buf.act3, n)
    # FC1: 64x64 tiles, LDS-DMA mainloop with 4 stages (variant 10): 17.6 us vs 20.9 us for the register ring at n = 1024
    H.gemm(H.NT, buf.act3, wb.view("fc1.w"), buf.h, n, 1024, 3136, bias=wf.view("fc1.b"), flags=H.EPI_RELU, big=H.FC1_FWD_VARIANT)
    H.head_fwd(buf.h, wb.view("adv2.w"), wf.view("adv2.b"), wb.view("val2.w"), wf.view("val2.b"), buf.q, action_out,
               eps, eps_mod, seed, ctr)


def torso_forward_dual(x_u8: torch.Tensor, off1: torch.Tensor, n1: int, wb1: ParamPack, wf1: ParamPack, b1: TorsoBuffers,
                       off2: torch.Tensor, n2: int, wb2: ParamPack, wf2: ParamPack, b2: TorsoBuffers, prep=None,
                       save1: Optional[int] = None, heads: bool = True) -> None:
    """Forward of TWO networks (online on n1 frames, target on n2 frames), 3 launches in total: the torso, FC1 and the head
    of both networks are one launch each, so the learner's step needs no second stream (no fork / join) for the target
    network. ``save1``: how many of network 1's leading frames a backward pass will read (default: all). (Sharing the s'
    tiles of conv1 between the two networks inside that launch was measured and dropped: 0.2465 vs 0.2400 ms per step, the
    heavier pair blocks set the launch's registers and its tail.)"""
    if H.TORSO_FUSED:
        # ONE launch for conv1-3 of both networks; act1 / act2 reach HBM only for network 1's frames [0, save1): the ones
        # a backward pass reads (Ape-X: the B online-s frames of 3 B forwards)
        cache = b1.__dict__.setdefault("_torso", {})
        H.torso_fwd(x_u8, torso_net(cache, off1, n1, wb1, wf1, b1.act1, b1.act2, b1.act3, save=(0, n1 if save1 is None else save1)),
                    torso_net(cache, off2, n2, wb2, wf2, None, None, b2.act3), prep=prep)
    else:
        H.conv_fwd_dual(1, x_u8, n1, wb1.view("conv1.w"), wf1.view("conv1.b"), b1.act1, None, n2, wb2.view("conv1.w"),
                        wf2.view("conv1.b"), b2.act1, frame_off1=off1, frame_off2=off2)
        for l, src, dst in ((2, "act1", "act2"), (3, "act2", "act3")):
            H.conv_fwd_dual(l, getattr(b1, src), n1, wb1.view(f"conv{l}.w"), wf1.view(f"conv{l}.b"), getattr(b1, dst),
                            getattr(b2, src), n2, wb2.view(f"conv{l}.w"), wf2.view(f"conv{l}.b"), getattr(b2, dst),
                            prep=prep if l == 2 else None)
    H.gemm_nt_dual(b1.act3, wb1.view("fc1.w"), wf1.view("fc1.b"), b1.h, b2.act3, wb2.view("fc1.w"), wf2.view("fc1.b"), b2.h,
                   1024, 3136, flags=H.EPI_RELU)
    if heads:        # (the learner's step computes them inside its TD launch: ops.hip.apex_head_td_dh)
        H.head_fwd_dual(b1.h, wb1.view("adv2.w"), wf1.view("adv2.b"), wb1.view("val2.w"), wf1.view("val2.b"), b1.q,
                        b2.h, wb2.view("adv2.w"), wf2.view("adv2.b"), wb2.view("val2.w"), wf2.view("val2.b"), b2.q)


class ApexEngine(ShardedIS):
    MAILBOX = 32         # floats in front of the flat gradient in the communication buffer (2 per rank, 128-byte aligned)

    def __init__(self, n_action: int, cfg: ApexConfig, replay: DevicePER, device="cuda", dist=None,
                 init_model: Optional[DuelingDQN] = None, use_graph: bool = True, wgrad_splits=(16, 16, 128),
                 halo_bwd: Optional[bool] = None, overlap_comm: bool = True, presample: bool = True,
                 ride_norm: bool = True, steps_per_graph: int = 4, ride_blocks: int = 512):
        self.cfg, self.replay, self.dist = cfg, replay, dist
        self.device = torch.device(device)
        self.A = n_action
        self.B = B = cfg.batch_size
        assert replay.batch_size == B
        dev = self.device
        model = init_model if init_model is not None else DuelingDQN(n_action)
        lay = apex_layout(n_action)
        self.P = ParamPack(lay, dev)                    # fp32 master (online)
        self.P.load_state_dict(model.state_dict())
        self.Pb = self.P.like(torch.bfloat16)           # bf16 shadow read by the MFMA kernels
        self.T = self.P.like()                          # target fp32 (biases) ...
        self.Tb = self.P.like(torch.bfloat16)           # ... and bf16
        # flat gradient; multi-GPU: the direct xGMI all-reduce's communication buffer when that path is usable
        # data-parallel: the flat gradient sits behind a small MAILBOX inside ONE communication buffer, so the ranks' max IS
        # weights ride in the gradient all-reduce (global normalisation of the batch, td.hip k_wmax_put / k_wmax_take)
        self.comm = self.mailbox = None
        if dist is not None and getattr(dist, "enabled", False) and hasattr(dist, "grad_buffer"):
            comm = dist.grad_buffer(lay.total + self.MAILBOX)
            self.comm = comm if comm is not None else torch.zeros(lay.total + self.MAILBOX, device=dev)
            self.mailbox = self.comm[:self.MAILBOX]
        self.G = self.P.like(flat=self.comm[self.MAILBOX:] if self.comm is not None else None)
        self.SQ = self.P.like()                         # RMSprop second moment
        H.cast_bf16(self.P.flat, self.Pb.flat)
        self.sync_target()
        self.st = H.new_step_state(dev, cfg.beta0, cfg.beta_decay)
        self.partial = torch.zeros(1024, device=dev)
        self.n_partial = H.n_partials(self.P.flat.numel())
        self.on = TorsoBuffers(2 * B, n_action, dev)
        self.tg = TorsoBuffers(B, n_action, dev)
        bf = dict(dtype=torch.bfloat16, device=dev)
        self.dq = torch.zeros(B, n_action, device=dev)
        self.new_prio = torch.zeros(B, device=dev)
        self.dh = torch.zeros(B, 1024, **bf)
        self.d3 = torch.zeros(B * 49, 64, **bf)          # d conv3 pre-activation
        self.d2 = torch.zeros(B * 81, 64, **bf)          # d conv2 pre-activation
        self.d1 = torch.zeros(B * 400, 32, **bf)         # d conv1 pre-activation
        # input gradients of conv3 / conv2 through the LDS-halo kernel: flipped / transposed weight copies, rewritten by
        # extra workgroups of every step's conv2 forward launch (ops.hip.conv_halo_fwd(prep=...))
        # At B = 512 the fused launch (dX + dW + db bodies share the chip) still wins: 0.2278 ms per step against 0.2297 ms
        # with the halo dX + a dW launch per layer, and 0.2362 ms with the dW launches on a side stream (fork / join edges
        # of the replayed graph cost more than the overlap gains). The halo dX kernels pay from ~1000 frames per launch
        # (R2D2: 2.18 -> 2.13 ms), so: by batch size.
        self.halo_bwd = H.CONV_HALO_BWD and (B >= 1024 if halo_bwd is None else bool(halo_bwd))
        self.wt2 = torch.zeros(4, 32, 2, 2, 64, **bf)
        self.wt3 = torch.zeros(64, 3, 3, 64, **bf)
        # split-K of the conv3 / conv2 / conv1 weight-gradient bodies (one int: conv3 = conv2 = n, conv1 = 4 n). At B = 512 with
        # the riders of the launches in front: 32 / 32 / 128 (round 2) 0.1868 ms per step, 16 / 16 / 64: 0.1845, 16 / 16 / 128:
        # 0.1810, 8 or 24 for conv3 / conv2 and 32 or 96 / 160+ for conv1 are worse (profiles/apex_riders_first_r3.txt)
        self.wgrad_splits = wgrad_splits
        seg = self.G.layout.segments["fc1.w"]
        self.keep = (seg.offset, seg.offset + seg.numel) if seg.offset % 4 == 0 and seg.numel % 4 == 0 else (0, 0)
        self.world = dist.world_size if dist is not None else 1
        self.use_graph = use_graph
        self._init_shard(replay, dist, dev)              # self.n_is: device scalar N_global / world (engine/shard.py)
        self.overlap_comm = overlap_comm
        # software-pipelined sampler: the next step's PER batch is drawn inside this step's conv1 backward launch
        self.presample = B % 4 == 0 and presample
        if self.presample:
            replay._next_buffers()                                        # allocate now, never inside a graph capture
        self._primed = False                                              # True once a pre-drawn batch is waiting
        self._grad_cut = lay.offset("fc1.w")             # flat[cut:] = FC1 + head gradients, final after _fb_head
        self._graphs = None
        self._graph_k = None
        self.steps_per_graph = max(1, int(steps_per_graph))      # K = 4 measured best (2 / 4 / 8 / 16: DESIGN section 2)
        self.steps_done = 0
        # a timed-out poll of the direct xGMI all-reduce must not reach the parameters (engine/guard.py)
        self.guard = StepGuard(dev)
        if getattr(dist, "xgmi", None) is not None:
            self.guard.add("direct xGMI all-reduce", dist.xgmi.err_ptr())
        # One rank, no guarded waits: the gradient-norm partials need no launch of their own. A slice of the flat gradient
        # is final as soon as the launch that produces it has ended, so its sum of squares rides along with the NEXT
        # backward launch (extra workgroups that fill its tail: ops.hip.tree_extra(sumsq=...)): [conv3.w, end) inside the
        # conv2 launch, [conv2.w, conv3.w) inside the conv1 launch; what the conv1 launch itself produces ([0, conv2.w) =
        # 33 KB) every workgroup of the optimizer launch sums itself (``late``). Data-parallel runs keep the norm launch:
        # their gradients are final only after the all-reduce.
        o2, o3 = lay.offset("conv2.w"), lay.offset("conv3.w")
        self._ride_ok = (ride_norm and not self._dist_on() and o2 % 4 == 0 and o3 % 4 == 0 and lay.total % 4 == 0
                         and (self.G.flat.data_ptr() + 4 * o3) % 16 == 0)
        nb, ns = int(ride_blocks), 16                      # workgroups that square [conv3.w, end) / [conv2.w, conv3.w)
        assert 1 <= nb <= self.partial.numel() - ns
        self._sq_big = (self.G.flat[o3:], self.partial[:nb])
        self._sq_small = (self.G.flat[o2:o3], self.partial[nb:nb + ns])
        self._late = o2
        self._n_partial_ride = nb + ns

    # ---- pieces -------------------------------------------------------------------------------------
    @property
    def ride_norm(self) -> bool:
        """Gradient-norm partials inside the backward launches (see __init__)? Not once a guarded wait has been registered:
        the norm launch is the one that folds the guards into the fault word."""
        return self._ride_ok and not self.guard.sources

    def sync_target(self) -> None:
        self.T.flat.copy_(self.P.flat)
        self.Tb.flat.copy_(self.Pb.flat)

    def _forward_backward(self) -> None:
        self._fb_head()
        self._fb_torso()

    def _fb_head(self) -> None:
        """Sampling, both forwards, TD, and the backward of head + FC1. After it the gradient of everything from
        ``fc1.w`` to the end of the flat buffer (97.6 % of the parameters) is final - multi-GPU runs start its
        all-reduce here, under the conv backward that follows."""
        B, A, r, cfg = self.B, self.A, self.replay, self.cfg
        # step_begin fused; G was zeroed by the optimizer; the sampler also accumulates the max IS weight.
        # presample: from the second step on the batch was already drawn by the previous step's optimizer launch.
        if not (self.presample and self._primed):
            r.sample(self.st, begin=True, is_n_scale=1.0, n_is=self.n_is)
        prep = (self.Pb.view("conv2.w"), self.Pb.view("conv3.w"), self.wt2, self.wt3) if self.halo_bwd else None
        torso_forward_dual(r.frames, r.frame_off, 2 * B, self.Pb, self.P, self.on, r.frame_off[B:], B, self.Tb, self.T, self.tg,
                           prep=prep, save1=B, heads=False)
        G, Pb, on = self.G, self.Pb, self.on
        h, a3 = on.h[:B], on.act3[:B]
        # the three dueling heads + TD + head input gradient in ONE launch, one wave per transition (the max IS weight came
        # from the sampler)
        T, Tb = self.T, self.Tb
        H.apex_head_td_dh(on.h, self.tg.h, (Pb.view("adv2.w"), self.P.view("adv2.b"), Pb.view("val2.w"), self.P.view("val2.b")),
                          (Tb.view("adv2.w"), T.view("adv2.b"), Tb.view("val2.w"), T.view("val2.b")), on.q, self.tg.q, r.idx, r.act,
                          r.rew, r.done, r.prio, self.st, cfg.gamma_n, cfg.per_alpha, cfg.prio_eps, self.dq, self.new_prio, self.dh)
        # ONE launch: fc1 dX + fc1 dW + fc1 db + head dW/db (no fork / join: each costs 5-11 us in a replayed graph)
        H.fc_bwd(self.dh, a3, Pb.view("fc1.w"), G.view("fc1.w"), G.view("fc1.b"), self.d3, B, 1024, 3136,
                 head=(self.dq, h, G.view("adv2.w"), G.view("adv2.b"), G.view("val2.w"), G.view("val2.b")))

    def _fb_torso(self) -> None:
        """Backward of conv3 -> conv1: dX + dW + db of a layer are bodies of ONE launch (k_multi, gemm.hip). The forked
        side-stream form it replaced (0.318 -> 0.280 ms per step) and the dCol GEMM + col2im form of the input gradient
        (conv3 24.4 -> 15.8 us, conv2 39.3 -> 20.0 us as an implicit GEMM) are gone."""
        B, r = self.B, self.replay
        G, Pb, on = self.G, self.Pb, self.on
        a2, a1 = on.act2[:B], on.act1[:B]
        ws = self.wgrad_splits                                 # split-K of the weight-gradient bodies: one count (conv3 = conv2,
        sp3, sp2, sp1 = ws if isinstance(ws, (tuple, list)) else (ws, ws, 4 * ws)     # conv1 4 x) or (conv3, conv2, conv1)
        # the PER priority write-back (needs only the TD kernel's output) rides along as an extra body: leaves + level 1
        # inside the conv3 launch, upper levels + step_end inside the conv2 launch
        halo = self.halo_bwd                                   # dX in the LDS-halo kernel, dW + db (+ extra) in k_multi
        if halo:
            H.conv_halo_dgrad(3, self.d3, self.wt3, a2, self.d2, B)
        H.conv_bwd(3, a2, self.d3, Pb.view("conv3.w"), G.view("conv3.w"), G.view("conv3.b"), None if halo else self.d2,
                   B, splits=sp3, extra=r.update_stage(1, self.new_prio))
        if halo:
            H.conv_halo_dgrad(2, self.d2, self.wt2, a1, self.d1, B)
        H.conv_bwd(2, a1, self.d2, Pb.view("conv2.w"), G.view("conv2.w"), G.view("conv2.b"), None if halo else self.d1,
                   B, splits=sp2, extra=r.update_stage(2, self.new_prio, end_state=self.st,
                                                      sumsq=self._sq_big if self.ride_norm else None,
                                                      zero=G.flat[:self._late] if self.ride_norm else None))
        # presample: the PER batch of the NEXT step is drawn by extra blocks of this launch (the tree update and
        # step_end ran in the two launches before it) into the replay's next-batch buffers; the sampler's chain of
        # dependent tree reads hides here instead of opening the step. The optimizer launch hands the batch over.
        H.conv_bwd(1, r.frames, self.d1, None, G.view("conv1.w"), G.view("conv1.b"), None, B, frame_off=r.frame_off,
                   splits=sp1, sample=r.sample_extra(self.st, 1.0, self.n_is) if self.presample else None,
                   extra=H.tree_extra(0, sumsq=self._sq_small) if self.ride_norm else None)
        if self.mailbox is not None:      # this rank's max IS weight(s) -> mailbox, in front of the conv slice's all-reduce
            H.wmax_put(self.st, self.mailbox, self.dist.rank, self.world)

    def _optimize(self) -> None:
        cfg = self.cfg
        if self.mailbox is not None:      # the all-reduces are done: the batch's maximum over ALL ranks
            H.wmax_take(self.st, self.mailbox, self.world)
        if not self.ride_norm:
            H.grad_sumsq(self.G.flat, self.partial, self.guard.sources, self.guard.fault)
        H.rmsprop_step(self.P.flat, self.G.flat, self.SQ.flat, self.Pb.flat, self.partial,
                       self._n_partial_ride if self.ride_norm else self.n_partial, 1.0 / self.world, cfg.max_grad_norm, cfg.lr, cfg.rms_alpha, cfg.rms_eps, self.st, zero_grad=True,
                       copies=self.replay.handover_extra() if self.presample else None, fault=self.guard.fault,
                       keep=self.keep,                       # fc1.w (96 % of the buffer) is overwritten by fc_bwd: no zero store
                       late=self._late if self.ride_norm else 0)
        self.guard.mirror()
        self._primed = self.presample

    def redraw(self) -> None:
        """Draw the waiting (pre-sampled) batch again, eagerly. For callers that turned the ingest machinery over WITHOUT
        a learner step in between (``AsyncIngest.flush`` before a replay snapshot): slots of the waiting batch may have
        been retired and rewritten meanwhile, and the captured step starts from the batch it finds in the live buffers."""
        if self.presample and self._primed:
            self.replay.sample(self.st, begin=True, is_n_scale=1.0, n_is=self.n_is)

    def _all_reduce(self) -> None:
        if self.dist is not None and self.dist.enabled:
            self.dist.all_reduce_flat(self.comm if self.comm is not None else self.G.flat)

    def _dist_on(self) -> bool:
        return self.dist is not None and self.dist.enabled

    # ---- step ---------------------------------------------------------------------------------------
    def _eager_step(self) -> None:
        if self._dist_on() and self.overlap_comm:
            self._overlapped_step(self._fb_head, self._fb_torso, self._optimize)
            return
        self._forward_backward()
        self._all_reduce()
        self._optimize()

    def _overlapped_step(self, head, torso, optimize) -> None:
        """Data-parallel step with the gradient all-reduce hidden under the backward pass: the FC1/head slice of the
        flat gradient (12.9 of 13.2 MB) is reduced on RCCL's stream WHILE the conv backward runs on the compute
        stream; only the 0.3 MB conv slice is reduced after it. xGMI is point-to-point, the payload is latency-bound:
        two collectives per step, never per-tensor buckets."""
        cut = self._grad_cut
        head()
        w_big = self.dist.all_reduce_async(self.G.flat[cut:], channel=0)
        torso()
        small = self.comm[:self.MAILBOX + cut] if self.comm is not None else self.G.flat[:cut]     # (+ the max-w mailbox)
        w_small = self.dist.all_reduce_async(small, channel=1)
        w_big.wait()
        w_small.wait()
        optimize()

    def capture(self) -> None:
        """Warm up eagerly, then capture. world==1: one graph. world>1: try one graph including the RCCL
        all-reduce; if the collective cannot be captured fall back to graph / eager all-reduce / graph."""
        self.refresh_shard_scalars()
        s = torch.cuda.Stream(self.device)
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(2):
                self._eager_step()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        import os
        collective = self.dist is not None and self.dist.enabled
        in_graph = collective and self.dist.graph_collectives_for(self.G.flat)
        if in_graph:
            self.dist.quiesce()        # the watchdog must have retired the warm-up steps' eager collectives (see Dist.quiesce)
        if not collective or in_graph:
            # single graph; with DRL_GRAPH_COLLECTIVES=1 the RCCL all-reduces (overlapped schedule included) are
            # captured into it as well: no host work at all between sampling and the priority write-back
            g = torch.cuda.CUDAGraph()
            with capture_graph(g):
                self._eager_step()
            self._graphs = ("one", g)
            # ... and K consecutive steps as ONE graph for run_steps(): the boundary between two replays costs ~8 us on
            # the GPU (0.2297 ms per step replayed one by one where its kernels take 0.2213), everything a step needs
            # from the one before it is on the device (StepState, presampled batch, priorities)
            if self.steps_per_graph > 1:
                try:
                    gk = torch.cuda.CUDAGraph()
                    with capture_graph(gk):
                        for _ in range(self.steps_per_graph):
                            self._eager_step()
                    self._graph_k = gk
                except Exception as e:  # noqa: BLE001 - the one-step graph above is complete; run_steps() falls back to it
                    self._graph_k = None
                    if self.dist is not None:
                        self.dist.fallbacks.append("steps_per_graph->1: %r" % (e,))
        elif self.overlap_comm:
            # default for multi-GPU: three graphs, the two RCCL all-reduces issued between them on RCCL's own stream
            # (the big one overlaps the conv-backward graph)
            g1, g2, g3 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
            with capture_graph(g1):
                self._fb_head()
            with capture_graph(g2):
                self._fb_torso()
            with capture_graph(g3):
                self._optimize()
            self._graphs = ("overlap", g1, g2, g3)
        else:
            # two graphs with ONE flat all-reduce issued eagerly between them
            g1, g2 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
            with capture_graph(g1):
                self._forward_backward()
            with capture_graph(g2):
                self._optimize()
            self._graphs = ("split", g1, g2)
        self.steps_done += 2

    def step(self) -> None:
        self.guard.check()                               # raises once a discarded step has shown up in the pinned mirror
        self._maybe_refresh_shard()                      # outside the graph: the replay below reads the device scalar
        if self.use_graph:
            if self._graphs is None:
                self.capture()
            if self._graphs[0] == "one":
                self._graphs[1].replay()
            elif self._graphs[0] == "overlap":
                _, g1, g2, g3 = self._graphs
                self._overlapped_step(g1.replay, g2.replay, g3.replay)
            else:
                self._graphs[1].replay()
                self._all_reduce()
                self._graphs[2].replay()
        else:
            self._eager_step()
        self.steps_done += 1
        if self.steps_done % self.cfg.target_update == 0:
            self.sync_target()

    def run_steps(self, n: int) -> None:
        """``n`` learner steps, ``steps_per_graph`` at a time as one graph replay wherever no host-side duty (target sync,
        shard-scalar refresh) falls strictly inside such a run; the rest through :meth:`step`. Same steps, same order."""
        K = self.steps_per_graph
        while n > 0:
            gk = self._graph_k if (self.use_graph and self._graphs is not None) else None
            s = self.steps_done
            room = min(self.cfg.target_update - s % self.cfg.target_update,
                       (self.shard_refresh - s % self.shard_refresh) if self._shard_on else K)
            if gk is None or n < K or room < K:
                self.step()                                  # (the first call captures)
                n -= 1
                continue
            self.guard.check()
            self._maybe_refresh_shard()
            gk.replay()
            self.steps_done += K
            n -= K
            if self.steps_done % self.cfg.target_update == 0:
                self.sync_target()

    # ---- host-visible state ---------------------------------------------------------------------------
    def stats(self) -> dict:
        torch.cuda.synchronize(self.device)
        self.guard.check()
        if getattr(self.dist, "xgmi", None) is not None:
            self.dist.check_health()                     # a timed-out poll of the direct all-reduce must not go unnoticed
        return H.read_step_state(self.st)

    def state_dict(self):
        return self.P.state_dict()

    def load_state_dict(self, sd) -> None:
        self.P.load_state_dict(sd)
        H.cast_bf16(self.P.flat, self.Pb.flat)

    def flat_params(self) -> torch.Tensor:
        return self.P.flat

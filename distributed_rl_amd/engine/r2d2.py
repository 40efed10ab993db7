"""R2D2Engine — graph-captured gfx950 recurrent learner step (replaces DuelingLSTMDQN.calc_priorities,
libs/models.py:106-162, and the R2D2 use of the Learner loop, learner_node.py:32-45).

The reference issues 315 single-step batch-64 forwards per update from Python. Here the time axis is
batched wherever there is no recurrence:

  sample B sequences → time-major frame offsets, stored (h0, c0)
  per net (online, target):
      torso over ALL T·B = 7 680 frames at once (3 implicit-GEMM convs reading uint8 straight from the replay)
      ONE input GEMM  xg[T·B, 2048] = feat · W_ihᵀ + (b_ih + b_hh)
      T recurrent steps: h·W_hhᵀ MFMA GEMM whose epilogue IS the LSTM cell (gate-interleaved weights → the four
          gates of a unit land in one lane), saving post-activation gates for BPTT
      heads over all T·B rows (FC 512→1024 + ReLU, dueling head)
  one unroll of the online net yields both Q(s_t,·) and the n-steps-ahead argmax (the reference's ``self_cp``)
  fused R2D2 TD (value rescale, η-mix priorities, masks for short sequences) → dL/dQ
  backward over t ∈ [burn_in, T): heads → BPTT (split-K dG·W_hh GEMM + pointwise cell backward per step) →
      dW_hh, dW_ih, dfeat as three big GEMMs over all learning steps → conv backward
  [one flat RCCL all-reduce] → global-norm clip + Adam + bf16 re-cast → priority write-back

Burn-in steps run the same kernels; gradients simply never flow past t = burn_in (models.py:125-137).
"""
from __future__ import annotations

import os
from typing import Optional, Tuple

import torch

from ..config import R2D2Config
from ..models import DuelingLSTMDQN
from ..models.packing import ParamPack, r2d2_layout
from ..ops import hip as H
from .guard import StepGuard
from .shard import ShardedIS
from .apex import torso_net
from .streams import ForkJoin, capture_graph
from ..replay.device_seq import DeviceSequenceReplay


class SeqNetBuffers:
    def __init__(self, T: int, B: int, Hd: int, n_action: int, device, save_gates: bool):
        bf = dict(dtype=torch.bfloat16, device=device)
        N = T * B
        self.act1 = torch.zeros(N, 20, 20, 32, **bf)
        self.act2 = torch.zeros(N, 9, 9, 64, **bf)
        self.act3 = torch.zeros(N, 3136, **bf)
        self.xg = torch.zeros(T, B, 4 * Hd, device=device)
        self.hs = torch.zeros(T + 1, B, Hd, **bf)            # hs[0] = h0, hs[t+1] = h after step t
        self.cs = torch.zeros(T + 1, B, Hd, device=device)
        self.gates = torch.zeros(T, B, 4 * Hd, **bf) if save_gates else None
        self.h2 = torch.zeros(N, 1024, **bf)
        self.q = torch.zeros(N, n_action, device=device)
        self.err = torch.zeros(4, dtype=torch.int32, device=device)    # ... and its timeout flag
        self.hx = H.lstm_hx(device)                                     # exchange buffer of the persistent recurrence


def _save_range(save_from: Optional[int], B: int, r0: int, n: int) -> Tuple[int, int]:
    """Frames [lo, n) of a chunk that starts at row r0 whose act1 / act2 a backward pass reads: time steps >= save_from."""
    if save_from is None:
        return (0, 0)
    lo = min(max(save_from * B - r0, 0), n)
    return (lo, n) if lo < n else (0, 0)


def seq_torso_forward(frames: torch.Tensor, frame_off: Optional[torch.Tensor], T: int, B: int, Hd: int, wb: ParamPack,
                      wf: ParamPack, buf: SeqNetBuffers, bsum: torch.Tensor, steps: Optional[Tuple[int, int]] = None,
                      persistent: bool = False, prep=None, w1=None, save_from: Optional[int] = 0) -> None:
    """Time-batched part of a DuelingLSTMDQN forward: 3 conv launches over the frames of time steps ``steps`` (default:
    all T) + ONE input-projection GEMM for those steps (throughput-bound). ``persistent``: the recurrence that follows is
    the persistent kernel."""
    t0, t1 = steps if steps is not None else (0, T)
    r0, r1, n = t0 * B, t1 * B, (t1 - t0) * B
    x = frames if frame_off is not None else frames[r0:r1]
    fo = frame_off[r0:r1] if frame_off is not None else None
    a1, a2, a3 = buf.act1[r0:r1], buf.act2[r0:r1], buf.act3[r0:r1]
    # frame_off given: the frames are plane windows of the replay store, conv1 weights in the pack's (c, kh, kw) order; dense
    # HWC stacks (the actors' sequence-priority pass) need w1 = the HWC-ordered copy (models.packing.conv1_hwc)
    assert (frame_off is None) == (w1 is not None), "dense HWC input <-> HWC-ordered conv1 weights"
    if H.TORSO_FUSED:
        # ONE launch; act1 / act2 reach HBM only from time step ``save_from`` on (None: never): what a backward pass reads
        H.torso_fwd(x, torso_net(buf.__dict__.setdefault("_torso", {}), fo, n, wb, wf, a1, a2, a3,
                                 save=_save_range(save_from, B, r0, n), w1=w1), prep=prep if t0 == 0 else None)
    else:
        H.conv_fwd(1, x, w1 if w1 is not None else wb.view("conv1.w"), wf.view("conv1.b"), a1, n, fo)
        H.conv_fwd(2, a1, wb.view("conv2.w"), wf.view("conv2.b"), a2, n, prep=prep if t0 == 0 else None)
        H.conv_fwd(3, a2, wb.view("conv3.w"), wf.view("conv3.b"), a3, n)
    if t0 == 0:
        H.add2(wf.view("lstm.b_ih"), wf.view("lstm.b_hh"), bsum)
    xg = buf.xg.view(T * B, 4 * Hd)[r0:r1]
    if H.use_big_gemm(n, 4 * Hd, 3136):        # 8-wave large-tile kernel (gemm_big.hip), bias in the epilogue
        H.gemm_big(H.NT, a3, wb.view("lstm.w_ih"), xg, bias=bsum)
    else:
        H.gemm(H.NT, a3, wb.view("lstm.w_ih"), xg, n, 4 * Hd, 3136, bias=bsum, flags=H.EPI_OUT_F32, big=n >= 1024)


def seq_torso_forward_dual(frames: torch.Tensor, frame_off: torch.Tensor, T: int, B: int, Hd: int, net1, net2,
                           steps: Tuple[int, int], persistent: bool, prep=None, free_cus: int = 256,
                           save_from: Optional[int] = 0, bias_ready: bool = False) -> None:
    """seq_torso_forward of TWO networks over the SAME frames (R2D2 learner: online + target): the torso of both as one
    launch (the second network's read of a frame finds it in the Infinity Cache); only network 1 keeps act1 / act2, from
    time step ``save_from`` on. net = (wb, wf, buf, bsum)."""
    t0, t1 = steps
    r0, r1, n = t0 * B, t1 * B, (t1 - t0) * B
    fo = frame_off[r0:r1]
    (wb1, wf1, b1, bs1), (wb2, wf2, b2, bs2) = net1, net2
    if H.TORSO_FUSED:
        cache = b1.__dict__.setdefault("_torso", {})
        H.torso_fwd(frames, torso_net(cache, fo, n, wb1, wf1, b1.act1[r0:r1], b1.act2[r0:r1], b1.act3[r0:r1],
                                      save=_save_range(save_from, B, r0, n)),
                    torso_net(cache, fo, n, wb2, wf2, None, None, b2.act3[r0:r1]), prep=prep if t0 == 0 else None)
    else:                                                # per-layer launches: every frame read and converted once for both nets
        H.conv1_fwd_pair(frames, fo, n, wb1.view("conv1.w"), wf1.view("conv1.b"), b1.act1[r0:r1], wb2.view("conv1.w"),
                         wf2.view("conv1.b"), b2.act1[r0:r1])
    for l, src, dst in ((2, "act1", "act2"), (3, "act2", "act3")) if not H.TORSO_FUSED else ():
        H.conv_fwd_dual(l, getattr(b1, src)[r0:r1], n, wb1.view(f"conv{l}.w"), wf1.view(f"conv{l}.b"), getattr(b1, dst)[r0:r1],
                        getattr(b2, src)[r0:r1], n, wb2.view(f"conv{l}.w"), wf2.view(f"conv{l}.b"), getattr(b2, dst)[r0:r1],
                        prep=prep if (l == 2 and t0 == 0) else None)
    if t0 == 0 and not bias_ready:                       # (the learner adds them inside its first launch of the step)
        for wb, wf, buf, bsum in (net1, net2):
            H.add2(wf.view("lstm.b_ih"), wf.view("lstm.b_hh"), bsum)
    if H.use_big_gemm(n, 4 * Hd, 3136):
        # the input projections of BOTH networks as one launch: a forward chunk of one network alone leaves the 256 CUs
        # half empty (72 x 64 rows = 144 tiles of 256 x 256), two of them fill the chip
        H.gemm_big(H.NT, b1.act3[r0:r1], wb1.view("lstm.w_ih"), b1.xg.view(T * B, 4 * Hd)[r0:r1], bias=bs1,
                   second=(b2.act3[r0:r1], wb2.view("lstm.w_ih"), b2.xg.view(T * B, 4 * Hd)[r0:r1], bs2, None),
                   cus=free_cus)
        return
    for wb, wf, buf, bsum in (net1, net2):
        xg = buf.xg.view(T * B, 4 * Hd)[r0:r1]
        H.gemm(H.NT, buf.act3[r0:r1], wb.view("lstm.w_ih"), xg, n, 4 * Hd, 3136, bias=bsum, flags=H.EPI_OUT_F32, big=n >= 1024)


def seq_recurrence(T: int, B: int, Hd: int, wb: ParamPack, buf: SeqNetBuffers, persistent: bool,
                   steps: Optional[Tuple[int, int]] = None, bsum: Optional[torch.Tensor] = None) -> None:
    """The recurrence over time steps ``steps`` (default: all T) from the state in buf.hs[t0] / buf.cs[t0]: one persistent
    launch, or one launch of the EPI_LSTM GEMM per step; latency-bound."""
    t0, t1 = steps if steps is not None else (0, T)
    w_hh = wb.view("lstm.w_hh")
    if persistent:
        gates = buf.gates[t0:t1] if buf.gates is not None else None
        H.lstm_seq_fwd_x([(buf.xg[t0:t1], w_hh, buf.hs[t0:t1 + 1], buf.cs[t0:t1 + 1], gates, buf.hx)], t1 - t0, B, Hd, buf.err)
    else:
        for t in range(t0, t1):
            H.lstm_step(buf.hs[t], w_hh, buf.xg[t], buf.cs[t], buf.hs[t + 1], buf.cs[t + 1],
                        buf.gates[t] if buf.gates is not None else None)


def seq_heads(T: int, B: int, Hd: int, wb: ParamPack, wf: ParamPack, buf: SeqNetBuffers) -> None:
    """Time-batched FC + dueling head over all T*B LSTM outputs."""
    N = T * B
    H.gemm(H.NT, buf.hs[1:].view(N, Hd), wb.view("fc1.w"), buf.h2, N, 1024, Hd, bias=wf.view("fc1.b"),
           flags=H.EPI_RELU, big=N >= 1024)
    H.head_fwd(buf.h2, wb.view("adv2.w"), None, wb.view("val2.w"), wf.view("val2.b"), buf.q)


def seq_recurrent_forward(T: int, B: int, Hd: int, wb: ParamPack, wf: ParamPack, buf: SeqNetBuffers,
                          persistent: bool, bsum: Optional[torch.Tensor] = None) -> None:
    """Recurrence over all T steps + heads (``bsum``: kept for call-site symmetry with seq_torso_forward)."""
    seq_recurrence(T, B, Hd, wb, buf, persistent, bsum=bsum)
    seq_heads(T, B, Hd, wb, wf, buf)


def seq_net_forward(frames: torch.Tensor, frame_off: Optional[torch.Tensor], T: int, B: int, Hd: int, wb: ParamPack,
                    wf: ParamPack, buf: SeqNetBuffers, bsum: torch.Tensor, persistent: bool, w1=None) -> None:
    """Q values of one DuelingLSTMDQN over T x B frames (time-major), starting from the state in buf.hs[0] / buf.cs[0].
    Shared by the learner (frames addressed in the replay store through ``frame_off``) and the actor-side
    sequence-priority pass (dense frames, ``frame_off`` None)."""
    seq_torso_forward(frames, frame_off, T, B, Hd, wb, wf, buf, bsum, persistent=persistent, w1=w1, save_from=None)
    seq_recurrent_forward(T, B, Hd, wb, wf, buf, persistent, bsum=bsum)


class R2D2Engine(ShardedIS):
    MAILBOX = 32         # floats in front of the flat gradient in the communication buffer (one per rank, 128-byte aligned)

    def __init__(self, n_action: int, cfg: R2D2Config, replay: DeviceSequenceReplay, device="cuda", dist=None,
                 init_model: Optional[DuelingLSTMDQN] = None, use_graph: bool = True, wgrad_splits: int = 128,
                 done_offset: int = 0, overlap: bool = True, bwd_chunks: int = 2, fwd_chunks=3, bwd_cuts=None,
                 bptt_groups: int = 2, fwd_groups_last: int = 2):
        """Tuning arguments (measured defaults; scripts/r2d2_bench.py --engine-arg sweeps them): ``fwd_chunks`` = number of
        equal time chunks of the pipelined forward or explicit cut points "a,b,.."; ``bwd_chunks`` / ``bwd_cuts`` (chunk starts
        in window steps, e.g. "48,64") of the backward; row groups of the persistent BPTT and of the last forward chunk's
        recurrence (1, 2, 4)."""
        self.cfg, self.replay, self.dist = cfg, replay, dist
        self.device = dev = torch.device(device)
        self.A, self.B, self.T, self.Hd = n_action, cfg.batch_size, cfg.total_len, cfg.hidden
        self.burn = cfg.burn_in
        self.L = self.T - self.burn                       # backward window [burn, T)
        self.done_offset = done_offset
        B, T, Hd, L = self.B, self.T, self.Hd, self.L
        model = init_model if init_model is not None else DuelingLSTMDQN(n_action, B, cfg.burn_in, cfg.nstep)
        lay = r2d2_layout(n_action, hidden=Hd)
        self.P = ParamPack(lay, dev)
        self.P.load_state_dict(model.state_dict())
        self.Pb = self.P.like(torch.bfloat16)
        self.Tg = self.P.like()
        self.Tb = self.P.like(torch.bfloat16)
        # data-parallel: the flat gradient sits behind a small MAILBOX inside ONE communication buffer; a 128-byte all-reduce
        # of the mailbox under the forward pass gathers the ranks' batch maxima (global IS-weight normalisation, td.hip
        # k_is_wmax_put / k_is_weights)
        self.comm = self.mailbox = self._gw = None
        if dist is not None and getattr(dist, "enabled", False) and hasattr(dist, "grad_buffer"):
            comm = dist.grad_buffer(lay.total + self.MAILBOX)
            self.comm = comm if comm is not None else torch.zeros(lay.total + self.MAILBOX, device=dev)
            self.mailbox = self.comm[:self.MAILBOX]
        self.G = self.P.like(flat=self.comm[self.MAILBOX:] if self.comm is not None else None)
        self.M1 = self.P.like()
        self.M2 = self.P.like()
        H.cast_bf16(self.P.flat, self.Pb.flat)
        self.sync_target()
        self.st = H.new_step_state(dev, cfg.beta0, cfg.beta_decay)
        self._init_shard(replay, dist, dev)               # self.n_is: device scalar N_global / world (engine/shard.py)
        self.partial = torch.zeros(1024, device=dev)
        self.n_partial = H.n_partials(self.P.flat.numel())
        self.on = SeqNetBuffers(T, B, Hd, n_action, dev, True)
        self.tg = SeqNetBuffers(T, B, Hd, n_action, dev, False)
        self.bsum_on = torch.zeros(4 * Hd, device=dev)
        self.bsum_tg = torch.zeros(4 * Hd, device=dev)
        bf = dict(dtype=torch.bfloat16, device=dev)
        NB = L * B
        self.NB = NB
        self.w = torch.zeros(B, device=dev)
        self.dq = torch.zeros(T, B, n_action, device=dev)
        self.new_prio = torch.zeros(B, device=dev)
        self.dh2 = torch.zeros(NB, 1024, **bf)
        self.dhs = torch.zeros(L, B, Hd, **bf)
        self.dG = torch.zeros(L, B, 4 * Hd, **bf)
        self.dh_rec = torch.zeros(B, Hd, device=dev)
        self.dc = torch.zeros(B, Hd, device=dev)
        self.dh_rec3 = torch.zeros(3, B, Hd, device=dev)     # fused BPTT step: read / accumulate / being-zeroed rotation
        self.dc2 = torch.zeros(2, B, Hd, device=dev)
        self.d3 = torch.zeros(NB * 49, 64, **bf)
        self.d2 = torch.zeros(NB * 81, 64, **bf)
        self.d1 = torch.zeros(NB * 400, 32, **bf)
        self.halo_bwd = H.CONV_HALO_BWD                  # conv3 / conv2 input gradients through the LDS-halo kernel
        self.halo_wgrad = self.halo_bwd and H.CONV_HALO_WGRAD    # ... and their weight gradients (conv_wgrad_halo.hip)
        if self.halo_wgrad:                              # both backward chunks run on one stream: one scratch per layer
            self.ws1, self.ws2, self.ws3 = (H.conv_wgrad_ws(l, device) for l in (1, 2, 3))
        self.wt2 = torch.zeros(4, 32, 2, 2, 64, **bf)
        self.wt3 = torch.zeros(64, 3, 3, 64, **bf)
        self.wgrad_splits = wgrad_splits
        s0, s1 = self.G.layout.segments["lstm.w_ih"], self.G.layout.segments["lstm.w_hh"]
        contiguous = s1.offset == s0.offset + s0.numel and s0.offset % 4 == 0 and (s0.numel + s1.numel) % 4 == 0
        self.keep = (s0.offset, s1.offset + s1.numel) if contiguous else (0, 0)
        _os = os
        self.busy_aware = True                            # size persistent conv grids around the recurrences
        # row groups of the persistent BPTT (32 workgroups each; lstm_seq.hip): 80 steps alone take 366 / 270 / 271 us
        # with 1 / 2 / 4 groups
        # the big products that run next to a recurrence count their tile rounds over the CUs it leaves free: 64 CUs hold
        # a 64 KB workgroup each, no gemm_big tile (96-144 KB of LDS) fits beside one - 240 tiles of 256 x 256 take two
        # rounds on 192 CUs (93 -> 171 us), 128 x 128 tiles spread better (profiles/r2d2_busy_tile_sweep_r2.txt)
        self.big_busy_aware = True
        # the recurrence of the LAST forward chunk has the chip to itself: row groups (lstm_seq.hip) spread its h loads
        self.fwd_groups_last = max(1, min(4, int(fwd_groups_last)))
        while self.fwd_groups_last > 1 and (self.B // 16) % self.fwd_groups_last:
            self.fwd_groups_last //= 2
        self.bptt_groups = max(1, min(4, int(bptt_groups)))
        while self.bptt_groups > 1 and (self.B // 16) % self.bptt_groups:
            self.bptt_groups //= 2
        self.big_tn = H.BIG_GEMM and NB >= 1024
        self.bwd_chunks = max(1, int(bwd_chunks))
        self.bwd_cuts = bwd_cuts
        # both recurrences of a chunk in one launch: equal with the plain hand-off, 2.93 vs 2.96 ms with the exchange form
        self.dual_lstm = self.dual_torso = self.dual_heads = True    # both networks per launch wherever the kernels take two
        self.fused_bptt = H.lstm_bwd_step_supported(self.B, self.Hd)
        # the whole backward recurrence of a time chunk as ONE persistent launch (4.5 us per step instead of ~10)
        self.persistent_bptt = H.lstm_seq_supported(self.B, self.Hd) and _os.environ.get("DRL_PERSISTENT_BPTT", "1") != "0"
        if self.persistent_bptt:
            self.bx = H.lstm_bx(dev)
            self.bptt_rec = torch.zeros(self.B, self.Hd, device=dev)
            self.bptt_dc = torch.zeros(self.B, self.Hd, device=dev)
            self.bptt_err = torch.zeros(1, dtype=torch.int32, device=dev)
        # int = number of equal time chunks of the pipelined forward; "a,b,.." = explicit cut points (time steps)
        fc = str(fwd_chunks)
        if "," in fc:
            cuts = sorted(set(int(x) for x in fc.split(",") if x.strip() and 0 < int(x) < self.T))
        else:
            n_ = max(1, int(fc))
            csz_ = (self.T + n_ - 1) // n_
            cuts = list(range(csz_, self.T, csz_))
            if n_ == 3 and self.T >= 10:
                # three chunks whose input projections (both networks, 256 x 256 tiles) each take ONE round of tiles on the
                # CUs a running recurrence leaves free: 1.625 ms per step with two chunks (cut at 64 of 120: the second
                # chunk's projection, 128 x 128 tiles next to the recurrence, ended 33 us after the first recurrence) ->
                # 1.601-1.609 with cuts at 44 and 84 (profiles/r2d2_fwd_cut_sweep_r3.txt)
                cuts = [(11 * self.T) // 30, (7 * self.T) // 10]
            if n_ == 2 and self.T >= 10:
                # the torso, not the recurrence, bounds the pipelined forward since the recurrence got faster: a larger first
                # chunk leaves a shorter recurrence after the last torso (2.36 / 2.33 / 2.33 / 2.37 / 2.43 ms per step for a
                # cut at 60 / 72 / 80 / 90 / 100 of 120)
                cuts = [(3 * self.T) // 5]
                # the input projections of both networks run as ONE large-tile launch per chunk (gemm_big.hip): keep each
                # chunk within one round of 256 tiles of 256 x 256 (2 networks x rows/256 x 4H/256 tiles)
                cap_rows = 256 * 256 // (2 * max(1, 4 * self.Hd // 256))
                if H.BIG_GEMM and cuts[0] * self.B > cap_rows >= self.B and self.T * self.B <= 2 * cap_rows:
                    cuts = [max(self.T - cap_rows // self.B, min(cuts[0], cap_rows // self.B))]
        self.fwd_bounds = list(zip([0] + cuts, cuts + [self.T]))
        self.fwd_chunks = len(self.fwd_bounds)
        self.world = dist.world_size if dist is not None else 1
        self.use_graph = use_graph
        self.fj = ForkJoin(dev, 3, overlap)               # target forward / weight gradients on side streams
        self.persistent_lstm = H.lstm_seq_supported(B, Hd) and os.environ.get("DRL_PERSISTENT_LSTM", "1") != "0"
        if os.environ.get("DRL_LSTM_SPIN_LIMIT"):        # fault injection (tests of the self-healing launcher): polls give up early
            H.lstm_set_spin_limit(int(os.environ["DRL_LSTM_SPIN_LIMIT"]))
        self._graphs = None
        self.steps_done = 0
        # bounded inter-workgroup / inter-GPU waits of this step: when one gives up the step is discarded (engine/guard.py)
        self.guard = StepGuard(dev)
        if self.persistent_lstm:
            self.guard.add("persistent LSTM (online)", self.on.err)
            self.guard.add("persistent LSTM (target)", self.tg.err)
        if self.persistent_bptt:
            self.guard.add("persistent BPTT", self.bptt_err)
        if getattr(dist, "xgmi", None) is not None:
            self.guard.add("direct xGMI all-reduce", dist.xgmi.err_ptr())

    def sync_target(self) -> None:
        self.Tg.flat.copy_(self.P.flat)
        self.Tb.flat.copy_(self.Pb.flat)

    # ---- forward of one net over the whole sampled batch ---------------------------------------------------
    def _net_forward(self, wb: ParamPack, wf: ParamPack, buf: SeqNetBuffers, bsum: torch.Tensor) -> None:
        r = self.replay
        H.gather_state2(r.h0, r.c0, r.idx, buf.hs[0], buf.cs[0])
        seq_net_forward(r.frames, r.frame_off, self.T, self.B, self.Hd, wb, wf, buf, bsum, self.persistent_lstm)

    def _forward_backward(self) -> None:
        cfg, r, T, B, Hd, A, L, NB, burn = self.cfg, self.replay, self.T, self.B, self.Hd, self.A, self.L, self.NB, self.burn
        fj = self.fj
        pipelined = self.fwd_chunks > 1 and fj.enabled
        if pipelined:
            # ONE launch: the draw (step_begin fused; G was zeroed by the optimizer), the window offsets of the drawn
            # sequences, their stored state into the initial-state rows of both networks, and b_ih + b_hh of both networks
            # (the bias of the input projections) - five launches in front of the first conv before; worth 0-10 us of the step
            r.sample_with_state(self.st, self.on.hs[0], self.on.cs[0], self.tg.hs[0], self.tg.cs[0],
                                sums=((self.P.view("lstm.b_ih"), self.P.view("lstm.b_hh"), self.bsum_on),
                                      (self.Tg.view("lstm.b_ih"), self.Tg.view("lstm.b_hh"), self.bsum_tg)), n_is=self.n_is)
        else:
            r.sample(self.st, begin=True, n_is=self.n_is)
            H.gather_state2(r.h0, r.c0, r.idx, self.on.hs[0], self.on.cs[0])
        wmax_work = None
        if self._global_wmax():             # this rank's batch maximum -> mailbox; gathered while the forward pass runs
            H.is_wmax_put(r.prio, self.st, self.mailbox, self.dist.rank, self.world)
            wmax_work = self.dist.all_reduce_async(self.mailbox, channel=1)
        # forward. The two nets are STAGGERED, not run side by side: the torso is throughput-bound, the recurrence is
        # latency-bound (32 workgroups), so the target torso runs under the online recurrence and the two recurrences
        # overlap only partly.
        # pre-transformed conv2 / conv3 weights of the LDS-halo input-gradient kernels: written by extra workgroups of the
        # first conv2 forward launch of the step (or by a launch of their own when that forward is not the halo kernel)
        prep = (self.Pb.view("conv2.w"), self.Pb.view("conv3.w"), self.wt2, self.wt3) if self.halo_bwd else None
        if prep is not None and not H.CONV_HALO:
            H.conv_halo_wprep(*prep)
            prep = None
        if pipelined:
            # PIPELINED over time chunks: the torso of chunk k+1 (throughput-bound, all CUs) runs while the recurrences
            # of chunk k (latency-bound, 32 workgroups per net) chase it on two side streams, so the forward costs about
            # one recurrence + one torso chunk instead of two torsos + one recurrence. Measured (64 x 120, MI355X): 4.03 /
            # 3.76 / 3.87 / 4.49 / 4.47 ms per step for 1 / 2 / 4 / 6 / 8 chunks (every chunk restarts the persistent
            # kernel - W_hh slice back into LDS - and the recurrence steps slow down under the torso's memory traffic)
            dual = self.persistent_lstm and self.dual_lstm
            for st_ in self.fwd_bounds:
                pl = self.persistent_lstm
                if dual and self.dual_torso:
                    # from the second chunk on the torso runs next to the recurrence of the chunk before (64 workgroups)
                    with H.halo_busy(64 if (st_[0] > 0 and self.busy_aware) else 0):
                        seq_torso_forward_dual(r.frames, r.frame_off, T, B, Hd, (self.Pb, self.P, self.on, self.bsum_on),
                                               (self.Tb, self.Tg, self.tg, self.bsum_tg), st_, pl, prep=prep,
                                               free_cus=256 - 64 * self.fwd_groups_busy(st_), save_from=burn, bias_ready=True)
                else:
                    seq_torso_forward(r.frames, r.frame_off, T, B, Hd, self.Pb, self.P, self.on, self.bsum_on, steps=st_,
                                      persistent=pl, prep=prep, save_from=burn)
                    if not dual:
                        with fj.fork(0):
                            seq_recurrence(T, B, Hd, self.Pb, self.on, pl, steps=st_, bsum=self.bsum_on)
                    seq_torso_forward(r.frames, r.frame_off, T, B, Hd, self.Tb, self.Tg, self.tg, self.bsum_tg, steps=st_,
                                      persistent=pl, save_from=None)
                if dual:
                    # both recurrences of the chunk in ONE persistent launch (64 workgroups): 685 us for 120 steps of two
                    # nets against 655 us for one, so the second network rides along almost for free
                    t0, t1 = st_
                    with fj.fork(0):
                        on_net = (self.on.xg[t0:t1], self.Pb.view("lstm.w_hh"), self.on.hs[t0:t1 + 1], self.on.cs[t0:t1 + 1],
                                  self.on.gates[t0:t1])
                        tg_net = (self.tg.xg[t0:t1], self.Tb.view("lstm.w_hh"), self.tg.hs[t0:t1 + 1], self.tg.cs[t0:t1 + 1], None)
                        H.lstm_seq_fwd_x([on_net + (self.on.hx,), tg_net + (self.tg.hx,)], t1 - t0, B, Hd, self.on.err,
                                         groups=self.fwd_groups_last if st_ is self.fwd_bounds[-1] else 1)
                else:
                    with fj.fork(2):
                        seq_recurrence(T, B, Hd, self.Tb, self.tg, pl, steps=st_, bsum=self.bsum_tg)
            if dual and self.dual_heads:
                # both recurrences ended in the same launch: FC + dueling head of both networks as two launches, not four
                with fj.fork(0):
                    # only the rows the loss reads (t >= burn-in: the TD kernel looks at q[t] and q[t + n] of learning steps)
                    lo_ = burn * B if self.big_tn else 0
                    hs_on, hs_tg = self.on.hs[1:].view(T * B, Hd)[lo_:], self.tg.hs[1:].view(T * B, Hd)[lo_:]
                    if self.big_tn:                                 # two problems per launch, bias + ReLU in the epilogue
                        H.gemm_big(H.NT, hs_on, self.Pb.view("fc1.w"), self.on.h2[lo_:], bias=self.P.view("fc1.b"), relu=True,
                                   second=(hs_tg, self.Tb.view("fc1.w"), self.tg.h2[lo_:], self.Tg.view("fc1.b"), None))
                    else:
                        H.gemm_nt_dual(hs_on, self.Pb.view("fc1.w"), self.P.view("fc1.b"), self.on.h2,
                                       hs_tg, self.Tb.view("fc1.w"), self.Tg.view("fc1.b"), self.tg.h2, 1024, Hd, flags=H.EPI_RELU)
                    H.head_fwd_dual(self.on.h2[lo_:], self.Pb.view("adv2.w"), None, self.Pb.view("val2.w"), self.P.view("val2.b"),
                                    self.on.q[lo_:], self.tg.h2[lo_:], self.Tb.view("adv2.w"), None, self.Tb.view("val2.w"),
                                    self.Tg.view("val2.b"), self.tg.q[lo_:])
            else:
                with fj.fork(0):
                    seq_heads(T, B, Hd, self.Pb, self.P, self.on)
                with fj.fork(2 if not dual else 0):
                    seq_heads(T, B, Hd, self.Tb, self.Tg, self.tg)
        else:
            seq_torso_forward(r.frames, r.frame_off, T, B, Hd, self.Pb, self.P, self.on, self.bsum_on,
                              persistent=self.persistent_lstm, prep=prep, save_from=burn)
            with fj.fork(0):                                        # target net: queued behind the online torso
                self._net_forward(self.Tb, self.Tg, self.tg, self.bsum_tg)
            seq_recurrent_forward(T, B, Hd, self.Pb, self.P, self.on, self.persistent_lstm, bsum=self.bsum_on)
        if wmax_work is not None:
            wmax_work.wait()
            H.is_weights(r.prio, self.st, self.w, mailbox=self.mailbox, world=self.world)
        else:
            H.is_weights(r.prio, self.st, self.w)
        fj.join()
        H.r2d2_td(self.on.q.view(T, B, A), self.tg.q.view(T, B, A), r.idx, r.act, r.rew, r.done, r.length, self.w, self.st,
                  burn, cfg.nstep, self.done_offset, cfg.gamma_n, cfg.eta, cfg.rescale_eps, self.dq, self.new_prio)
        # ---- backward over rows [burn*B, T*B) ----
        G, Pb, on = self.G, self.Pb, self.on
        r0 = burn * B
        h2, hs_out = on.h2[r0:], on.hs[burn + 1:].view(NB, Hd)
        hb = (self.dq.view(T * B, A)[r0:], h2, Pb.view("adv2.w"), Pb.view("val2.w"), self.dh2, G.view("adv2.w"), None,
              G.view("val2.w"), G.view("val2.b"))
        with fj.fork(1):                                            # head dW, db: off the critical path
            H.head_bwd(*hb, parts=2)
        H.head_bwd(*hb, parts=1)
        with fj.fork(0):                                            # fc1 dW, db || the BPTT chain
            if self.big_tn:                                         # long-K, small-N products: 128 x 128 tiles of gemm_big.hip
                H.gemm_big(H.TN, self.dh2, hs_out, G.view("fc1.w"), tile=3)
            else:
                H.gemm(H.TN, self.dh2, hs_out, G.view("fc1.w"), 1024, Hd, NB, flags=H.EPI_OUT_F32, big=False)
            H.colsum(self.dh2, G.view("fc1.b"), NB, 1024)
        if self.big_tn:
            H.gemm_big(H.NN, self.dh2, Pb.view("fc1.w"), self.dhs.view(NB, Hd), tile=3)
        else:
            H.gemm(H.NN, self.dh2, Pb.view("fc1.w"), self.dhs.view(NB, Hd), NB, Hd, 1024, big=False)
        # BPTT: a latency-bound chain of one small launch per step (64 workgroups) that leaves 3/4 of the chip idle.
        # Everything below the LSTM is row-wise independent (weight gradients accumulate with atomics), so the window is
        # cut into time chunks and the torso backward of a chunk runs on a side stream as soon as the chain has passed
        # it. The heavy kernels slow the chain down, so few chunks are best: 3.26 / 3.12 / 3.13 / 3.18 / 3.38 ms per step
        # for 1 / 2 / 3 / 4 / 8 chunks (with the two-launch chain of earlier versions chunking did not pay at all).
        w_hh = Pb.view("lstm.w_hh")
        csz = (L + self.bwd_chunks - 1) // self.bwd_chunks
        starts = set(range(0, L, csz))
        if self.persistent_bptt and self.bwd_chunks == 2:
            # the persistent chain is faster than the torso backward it feeds: a smaller newest chunk starts the torso
            # earlier (2.49 / 2.44 / 2.43 / 2.45 ms for a cut at window step 64 / 56 / 48 / 40 of 80; more cuts: 2.59-2.61)
            # (round 3, faster chain in two row groups: 1.706 / 1.683 / 1.658 / 1.688 / 1.820 / 1.738 ms for 40 / 48 / 52 / 56 / 60 /
            # 64, three chunks 1.88: profiles/r2d2_bwd_cut_sweep_r3.txt)
            starts = {0, (13 * L) // 20}
        cuts_env = self.bwd_cuts                                  # explicit chunk starts in window steps: 48, "48,64", (48, 64)
        if cuts_env:
            cl = cuts_env.split(",") if isinstance(cuts_env, str) else (cuts_env if isinstance(cuts_env, (tuple, list)) else [cuts_env])
            starts = {0} | {int(c) for c in cl if str(c).strip() and 0 < int(c) < L}
        order = sorted(starts)
        end_of = {k0: (order[i + 1] if i + 1 < len(order) else L) for i, k0 in enumerate(order)}
        fused_step = self.fused_bptt
        if self.persistent_bptt:
            k_hi = L - 1
            for i, k0 in enumerate(sorted(starts, reverse=True)):    # one launch per time chunk, newest chunk first
                H.lstm_bptt(self.dhs, on.gates[burn:T], on.cs[burn:T + 1], self.dG, w_hh, self.bptt_rec, self.bptt_dc, self.bx,
                            self.bptt_err, k_hi=k_hi, steps=k_hi - k0 + 1, first=i == 0, carry=k0 > 0, groups=self.bptt_groups)
                with fj.fork(1):
                    self._torso_backward(k0, end_of[k0])
                k_hi = k0 - 1
        elif fused_step:
            R, DC = self.dh_rec3, self.dc2
            R.zero_()
            DC[0].zero_()
        else:
            self.dh_rec.zero_()
            self.dc.zero_()
        for s_, t in enumerate(range(T - 1, burn - 1, -1) if not self.persistent_bptt else ()):
            k = t - burn
            if fused_step and t > burn:                             # pointwise + recurrent GEMM of the step: ONE launch
                H.lstm_bwd_step(self.dhs[k], R[s_ % 3], R[(s_ + 1) % 3], R[(s_ + 2) % 3], DC[s_ % 2], DC[(s_ + 1) % 2],
                                on.gates[t], on.cs[t + 1], on.cs[t], self.dG[k], w_hh)
            elif fused_step:                                        # last step: nothing recurrent left to compute
                H.lstm_bwd_point(self.dhs[k], R[s_ % 3], DC[s_ % 2], on.gates[t], on.cs[t + 1], on.cs[t], self.dG[k])
            else:
                H.lstm_bwd_point(self.dhs[k], self.dh_rec, self.dc, on.gates[t], on.cs[t + 1], on.cs[t], self.dG[k])
                if t > burn:
                    H.gemm(H.NN, self.dG[k], w_hh, self.dh_rec, B, Hd, 4 * Hd, flags=H.EPI_OUT_F32 | H.EPI_ATOMIC, splits=8, big=5)
            if k in starts:
                with fj.fork(1):
                    self._torso_backward(k, end_of[k])
        dG = self.dG.view(NB, 4 * Hd)
        with fj.fork(0):                                            # LSTM weight gradients: need the whole window
            if self.big_tn:
                H.gemm_big(H.TN, dG, on.hs[burn:T].view(NB, Hd), G.view("lstm.w_hh"), tile=3)
            else:
                H.gemm(H.TN, dG, on.hs[burn:T].view(NB, Hd), G.view("lstm.w_hh"), 4 * Hd, Hd, NB, flags=H.EPI_OUT_F32, big=False)
            H.colsum(dG, G.view("lstm.b_ih"), NB, 4 * Hd)
            G.view("lstm.b_hh").copy_(G.view("lstm.b_ih"))
        if H.use_big_gemm(4 * Hd, 3136, NB):
            H.gemm_big(H.TN, dG, on.act3[r0:], G.view("lstm.w_ih"))
        else:
            H.gemm(H.TN, dG, on.act3[r0:], G.view("lstm.w_ih"), 4 * Hd, 3136, NB, flags=H.EPI_OUT_F32, big=False)
        fj.join()

    def fwd_groups_busy(self, st_) -> int:
        """1 if the forward chunk ``st_`` runs next to the dual recurrence of the chunk before it (64 workgroups), else 0."""
        return 1 if (st_[0] > 0 and self.big_busy_aware and self.persistent_lstm and self.dual_lstm) else 0

    def _torso_backward(self, k0: int, k1: int) -> None:
        """Backward of W_ih (input side) + conv3..conv1 for the time steps [k0, k1) of the backward window."""
        B, Hd, NB, r = self.B, self.Hd, self.NB, self.replay
        G, Pb, on = self.G, self.Pb, self.on
        r0 = self.burn * B
        lo, hi, n = k0 * B, k1 * B, (k1 - k0) * B
        dG = self.dG.view(NB, 4 * Hd)[lo:hi]
        a3, a2, a1 = on.act3[r0 + lo:r0 + hi], on.act2[r0 + lo:r0 + hi], on.act1[r0 + lo:r0 + hi]
        d3, d2, d1 = self.d3[lo * 49:hi * 49], self.d2[lo * 81:hi * 81], self.d1[lo * 400:hi * 400]
        sp = max(8, self.wgrad_splits * n // NB)
        if H.use_big_gemm(n, 3136, 4 * Hd):
            busy_g = 32 * self.bptt_groups if (k0 > 0 and self.persistent_bptt and self.big_busy_aware) else 0
            H.gemm_big(H.NN, dG, Pb.view("lstm.w_ih"), d3.view(n, 3136), mask=a3, cus=256 - busy_g)   # ReLU mask in the epilogue
        else:
            H.gemm(H.NN, dG, Pb.view("lstm.w_ih"), d3.view(n, 3136), n, 3136, 4 * Hd, mask=a3, flags=H.EPI_MASK, big=False)
        # dX + dW + db of a conv layer = one launch (k_multi); from ~1000 frames the dX body is the LDS-halo kernel
        # body variants / split counts measured at n = 5120 frames (scripts/kbench.py --frames 10240 --only bwd_fused)
        halo = self.halo_bwd                                    # dX in the LDS-halo kernel, dW + db in k_multi
        # every chunk but the oldest runs next to the persistent BPTT launch of the chunk after it (32 workgroups per row group)
        busy = 32 * self.bptt_groups if (k0 > 0 and self.persistent_bptt and self.busy_aware) else 0
        if self.halo_wgrad:                                     # ... or in the LDS-halo weight-gradient kernel
            with H.halo_busy(busy) as cus:
                H.conv_halo_dgrad(3, d3, self.wt3, a2, d2, n)
                H.conv_wgrad_halo(3, a2, d3, G.view("conv3.w"), G.view("conv3.b"), n, self.ws3, max_wgs=cus)
                H.conv_halo_dgrad(2, d2, self.wt2, a1, d1, n)
                H.conv_wgrad_halo(2, a1, d2, G.view("conv2.w"), G.view("conv2.b"), n, self.ws2, max_wgs=cus)
        else:
            if halo:
                H.conv_halo_dgrad(3, d3, self.wt3, a2, d2, n)
            H.conv_bwd(3, a2, d3, Pb.view("conv3.w"), G.view("conv3.w"), G.view("conv3.b"), None if halo else d2, n, splits=sp,
                       variant=4 if n >= 2048 else None)
            if halo:
                H.conv_halo_dgrad(2, d2, self.wt2, a1, d1, n)
            H.conv_bwd(2, a1, d2, Pb.view("conv2.w"), G.view("conv2.w"), G.view("conv2.b"), None if halo else d1, n, splits=sp)
        if self.halo_wgrad:
            H.conv1_wgrad_halo(r.frames, r.frame_off[r0 + lo:r0 + hi], d1, G.view("conv1.w"), G.view("conv1.b"), n, self.ws1,
                               max_wgs=256 - busy)
        else:
            H.conv_bwd(1, r.frames, d1, None, G.view("conv1.w"), G.view("conv1.b"), None, n,
                       frame_off=r.frame_off[r0 + lo:r0 + hi], splits=max(2 * sp, n // 10))

    def _optimize(self) -> None:
        cfg = self.cfg
        H.grad_sumsq(self.G.flat, self.partial, self.guard.sources, self.guard.fault)
        H.adam_step(self.P.flat, self.G.flat, self.M1.flat, self.M2.flat, self.Pb.flat, self.partial, self.n_partial,
                    1.0 / self.world, cfg.max_grad_norm, cfg.lr, cfg.adam_beta1, cfg.adam_beta2, cfg.adam_eps, self.st,
                    zero_grad=True, fault=self.guard.fault, keep=self.keep)   # lstm.w_ih / w_hh: plain GEMM outputs
        # step_end fused into the tree rebuild; a discarded step writes no priorities (they come from garbage Q values)
        self.replay.update(self.new_prio, end_state=self.st, fault=self.guard.fault)
        self.guard.mirror()

    def _all_reduce(self) -> None:
        if self.dist is not None and self.dist.enabled:
            self.dist.all_reduce_flat(self.G.flat)

    def _global_wmax(self) -> bool:
        """IS weights normalised by the maximum over ALL ranks' batches: needs a collective in the middle of the forward
        pass, i.e. one that can sit inside the step graph (the direct xGMI kernel, or RCCL where the start-up trial showed
        it can be captured) - or no graph at all. With the two-graph schedule (collectives issued between graphs) each
        rank normalises by its own batch maximum (a per-rank constant scale of its gradient share)."""
        if self._gw is None:
            d = self.dist
            if self.mailbox is None or d is None or not d.enabled:
                self._gw = False
            else:
                self._gw = (not self.use_graph) or bool(d.graph_collectives_for(self.G.flat))
        return self._gw

    def _eager_step(self) -> None:
        self._forward_backward()
        self._all_reduce()
        self._optimize()

    def capture(self) -> None:
        self.refresh_shard_scalars()
        s = torch.cuda.Stream(self.device)
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            self._eager_step()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        collective = self.dist is not None and self.dist.enabled
        in_graph = collective and self.dist.graph_collectives_for(self.G.flat)
        if in_graph:
            self.dist.quiesce()        # the watchdog must have retired the warm-up steps' eager collectives (see Dist.quiesce)
        if not collective or in_graph:
            # single graph; with DRL_GRAPH_COLLECTIVES=1 the RCCL all-reduce is captured into it as well
            g = torch.cuda.CUDAGraph()
            with capture_graph(g):
                self._eager_step()
            self._graphs = ("one", g)
        else:
            # default for multi-GPU: two graphs with the flat all-reduce issued eagerly between them
            g1, g2 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
            with capture_graph(g1):
                self._forward_backward()
            with capture_graph(g2):
                self._optimize()
            self._graphs = ("split", g1, g2)
        self.steps_done += 1

    def step(self) -> None:
        self.guard.check()                               # raises once a discarded step has shown up in the pinned mirror
        self._maybe_refresh_shard()                      # outside the graph: the replay below reads the device scalar
        if self.use_graph:
            if self._graphs is None:
                self.capture()
            if self._graphs[0] == "one":
                self._graphs[1].replay()
            else:
                self._graphs[1].replay()
                self._all_reduce()
                self._graphs[2].replay()
        else:
            self._eager_step()
        self.steps_done += 1
        if self.steps_done % self.cfg.target_update == 0:
            self.sync_target()

    def stats(self) -> dict:
        if getattr(self.dist, "xgmi", None) is not None:
            self.dist.check_health()
        st = H.read_step_state(self.st)
        self.guard.check()
        if self.persistent_bptt and int(self.bptt_err[0].item()):
            raise RuntimeError("persistent BPTT kernel: an inter-workgroup wait timed out (results invalid); "
                               "set DRL_PERSISTENT_BPTT=0 to use per-step launches")
        if self.persistent_lstm and (int(self.on.err[0].item()) or int(self.tg.err[0].item())):
            raise RuntimeError("persistent LSTM kernel: an inter-workgroup wait timed out (results invalid); "
                               "set DRL_PERSISTENT_LSTM=0 to use per-step launches")
        return st

    def state_dict(self):
        return self.P.state_dict()

    def load_state_dict(self, sd) -> None:
        self.P.load_state_dict(sd)
        H.cast_bf16(self.P.flat, self.Pb.flat)

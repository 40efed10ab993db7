"""Process-group glue: one process per GPU, ``torch.distributed`` backend "nccl" (= RCCL over xGMI on ROCm),
"gloo" on CPU for tests.

Design for xGMI (point-to-point links, latency-bound 13-32 MB payloads):

* **one flat bucket** — all gradients are contiguous in one buffer (models/packing.py), so a step issues exactly
  ONE all-reduce (sum); the 1/world averaging is folded into the fused optimizer kernel's ``grad_scale``.
  No per-tensor buckets, no DDP hooks.
* **param broadcast** of the flat fp32/bf16 parameter buffer from the learner rank to actor GPUs, versioned by a
  step counter (replaces ``SET/GET params``, ape_x/learner.py:83, ape_x/actor.py:52-56).
* **sharded PER**: every rank owns a replay shard in its own HBM and samples locally; the scalars needed for
  single-tree-equivalent IS weights (Σp, N, max w) travel in one tiny all-reduce / all-gather (M8).
"""
from __future__ import annotations

import os
from typing import Optional

import torch
import torch.distributed as td


class _Done:
    def wait(self):
        return True


class _StreamWork:
    """Handle of a collective that runs as a kernel on a side stream: ``wait()`` orders the current stream after it."""

    def __init__(self, stream):
        self.stream = stream

    def wait(self):
        torch.cuda.current_stream().wait_stream(self.stream)
        return True


class Dist:
    def __init__(self, backend: Optional[str] = None, device: Optional[torch.device] = None, timeout_s: int = 600,
                 group=None):
        """``group`` = an existing sub-group of an initialised world (parallel/world.py: the data-parallel learners of a
        launcher run that also holds actor ranks): rank / size are then the group's, and every collective of this object
        stays inside it. Without it the default group is used, initialised here from RANK / WORLD_SIZE if need be."""
        import datetime
        self.group = group
        if group is not None:
            self.rank, self.world_size = td.get_rank(group), td.get_world_size(group)
            self.local_rank = int(os.environ.get("LOCAL_RANK", str(self.rank)))
            backend = backend or td.get_backend(group)
        else:
            self.rank = int(os.environ.get("RANK", "0"))
            self.world_size = int(os.environ.get("WORLD_SIZE", "1"))
            self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        if device is None:
            if torch.cuda.is_available():
                # DRL_SHARE_GPU=1: rehearse W ranks on fewer GPUs (ranks share devices round-robin; RCCL refuses that, so
                # combine it with DRL_DIST_BACKEND=gloo)
                n = torch.cuda.device_count() if os.environ.get("DRL_SHARE_GPU") == "1" else 1 << 30
                device = torch.device("cuda", self.local_rank % n)
            else:
                device = torch.device("cpu")
        self.device = device
        if device.type == "cuda":
            torch.cuda.set_device(device)
        self.backend = backend or os.environ.get("DRL_DIST_BACKEND") or ("nccl" if device.type == "cuda" else "gloo")
        # DRL_FORCE_DIST=1 initialises the process group even for one rank (exercises the RCCL path on a 1-GPU box)
        self.enabled = self.world_size > 1 or os.environ.get("DRL_FORCE_DIST") == "1"
        self.xgmi = None                                       # direct xGMI all-reduce, see grad_buffer()
        self._xgmi_busy, self._xgmi_tried = False, False
        self.xgmi_reserve_numel = 0                            # make the direct all-reduce's buffer at least this large
        self.fallbacks = []                                    # every degraded path taken, for the bench's JSON line
        self.all_reduce_us = None                              # {"direct_xgmi": us, "process_group": us} measured by grad_buffer()
        self._owns_world = group is None                       # close() tears the default group down, never a caller's
        if self.enabled and group is None and not td.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29511")
            kw = {}
            if self.backend == "nccl":
                kw["device_id"] = device
            td.init_process_group(self.backend, rank=self.rank, world_size=self.world_size,
                                  timeout=datetime.timedelta(seconds=timeout_s), **kw)

    # ---- direct xGMI all-reduce (parallel/xgmi.py) -----------------------------------------------------------
    def grad_buffer(self, numel: int) -> Optional[torch.Tensor]:
        """Flat fp32 gradient buffer for an engine. On a multi-GPU node this is the communication buffer of the direct
        xGMI all-reduce (every peer maps it; the collective is one in-place kernel on it, csrc/hip/xgmi_ar.hip) - when that
        path is available AND has just reproduced RCCL's result on this very set of GPUs; otherwise ``None`` (the
        engine allocates an ordinary tensor and the collectives go through RCCL).

        The buffer is created once per process (``xgmi_reserve_numel`` makes it large enough for every engine the process
        will build); one engine owns it at a time and hands it back with :meth:`release_grad_buffer`.

        ``DRL_XGMI_AR`` = 0: never; 1: required (failures raise); unset / "auto": try, verify, fall back. Collective call:
        every rank must make it at the same point."""
        mode = os.environ.get("DRL_XGMI_AR", "auto")
        if self.xgmi is not None:                               # reuse by the next engine of this process
            if self._xgmi_busy or numel > self.xgmi.cap:
                self.fallbacks.append("xgmi->process_group: buffer busy or too small for %d elements" % numel)
                return None
            self._xgmi_busy = True
            self.xgmi.flat.zero_()
            return self.xgmi.flat[:numel]
        usable = (self.enabled and mode != "0" and self.device.type == "cuda" and 2 <= self.world_size <= 8
                  and not self._xgmi_tried)
        if not usable:
            return None
        self._xgmi_tried = True
        cap = max(int(numel), int(self.xgmi_reserve_numel))
        # the communication buffer lives in ordinary (cached) device memory; if THAT form does not reproduce the process
        # group's sums on this node (every other failure - no IPC, no peer access - is independent of the memory kind),
        # the same protocol is tried once more on fine-grained memory, which needs no cache maintenance to be coherent
        kinds = ["coarse", "fine"]
        ar, why, ok = None, "", False
        for kind in kinds:
            ar, why, ok, verified = None, "", False, False
            try:
                import socket
                hosts = [None] * self.world_size
                td.all_gather_object(hosts, socket.gethostname(), group=self.group)
                if len(set(hosts)) != 1:
                    raise RuntimeError("ranks on different hosts")
                from .xgmi import XgmiAllReduce
                ar = XgmiAllReduce(cap, self.device, self.rank, self.world_size, group=self.group,
                                   n_blocks=32, mem_kind=kind)
                verified = True
                ok = self._verify_xgmi(ar, numel)               # keeps the 1 s poll bound: see _agree below
                if not ok:
                    why = "sums differ from the process group's (%s memory)" % kind
            except Exception as e:                              # noqa: BLE001 - any failure means "use RCCL"
                ok, why = False, repr(e)
            # the verdict of the verification is agreed on BEFORE anything else is launched on the direct path: on a
            # broken IPC / xGMI path every further launch would burn its poll bound in three timed-out barriers
            ok = self._agree(ok)
            if ok:
                try:
                    ar.set_timeout(float(os.environ.get("DRL_XGMI_TIMEOUT_S", "60")))
                    # both paths are correct here: keep the direct kernel unless the process group's all-reduce is
                    # clearly faster on THIS node (every rank times both - collective calls - and the slowest rank's
                    # numbers decide); the direct kernel is timed at several grid sizes and keeps the best
                    t_x, t_pg = self._time_all_reduces(ar, numel)
                    self.all_reduce_us = {"direct_xgmi": round(t_x, 1), "process_group": round(t_pg, 1),
                                          "direct_xgmi_by_blocks": self._xgmi_sweep, "blocks": ar.n_blocks, "memory": kind}
                    if mode != "1" and t_x > 1.25 * t_pg:
                        ok, why = False, "slower than the process group here: %.0f vs %.0f us" % (t_x, t_pg)
                except Exception as e:                          # noqa: BLE001
                    ok, why = False, repr(e)
                ok = self._agree(ok)
                if ok:
                    break
            if ar is not None:
                try:
                    ar.close()
                except Exception:                               # noqa: BLE001
                    pass
                ar = None
            self.fallbacks.append("xgmi(%s memory) rejected: %s" % (kind, why or "verification failed on some rank"))
            if not self._agree(verified and "sums differ" in why):
                break                                           # not a coherence problem: another memory kind changes nothing
        if not ok:
            if mode == "1":
                raise RuntimeError("DRL_XGMI_AR=1 but the direct xGMI all-reduce is not usable: %s" % (why or "mismatch"))
            self.fallbacks.append("xgmi->process_group: %s" % (why or "verification failed on some rank"))
            if self.rank == 0:
                import sys
                print("[dist] direct xGMI all-reduce not used (%s); gradients go through RCCL" % (why or "verification failed"),
                      file=sys.stderr, flush=True)
            return None
        self.xgmi = ar
        self._xgmi_busy = True
        self._xgmi_stream = torch.cuda.Stream(self.device)
        ar.flat.zero_()
        return ar.flat[:numel]

    def release_grad_buffer(self) -> None:
        """The engine that owned the direct all-reduce's buffer is gone: the next ``grad_buffer`` call may have it."""
        self._xgmi_busy = False

    def _device_collectives(self) -> bool:
        """Can this group reduce DEVICE tensors natively (RCCL)? gloo accepts them too, but its CUDA path has deadlocked on
        back-to-back collectives when the ranks of a rehearsal share one GPU: such groups reduce through host tensors."""
        return self.device.type == "cuda" and "nccl" in str(self.backend)

    def _reduce(self, t: torch.Tensor, op) -> torch.Tensor:
        """In-place all-reduce of a small / set-up-time tensor that works on every backend (see _device_collectives)."""
        if t.is_cuda and not self._device_collectives():
            h = t.detach().cpu()
            td.all_reduce(h, op=op, group=self.group)
            t.copy_(h)
        else:
            td.all_reduce(t, op=op, group=self.group)
        return t

    def _agree(self, ok: bool) -> bool:
        """Same verdict on every rank (MIN); a rank that cannot even take the collective votes no."""
        try:
            v = torch.tensor([1.0 if ok else 0.0], device=self.device if self._device_collectives() else "cpu")
            td.all_reduce(v, op=td.ReduceOp.MIN, group=self.group)
            return float(v.item()) > 0.5
        except Exception:                                       # noqa: BLE001
            return False

    def _verify_xgmi(self, ar, numel: int, rounds: Optional[int] = None) -> bool:
        """The kernel against the process group on rank-dependent data that CHANGES between collectives (every round the
        buffer is rewritten by kernels, as the backward pass does, so a peer that kept stale lines of it - or a flag that
        overtook the data - shows up as a wrong sum): whole buffer, an engine-like split on two channels, alternating.
        Any mismatch, launch error or timed-out poll rejects the path. The loop contains collectives of the process group:
        every rank takes all of them, whatever happens to its own kernel launches (a local failure is only recorded)."""
        rounds = 50 if rounds is None else rounds
        g = torch.Generator(device=self.device).manual_seed(1234 + self.rank)
        cut = (numel // 41) // 64 * 64
        ar.set_timeout(1.0)                                     # a broken path must fail fast, not look like a hang
        noise = torch.empty(numel, device=self.device).normal_(generator=g)
        want = torch.empty_like(ar.flat)
        good = True
        for k in range(rounds):
            ar.flat.zero_()
            # new values in every element, every round, cheap: a rolled copy of the rank's noise scaled by the round
            sh = (k * 7919 + 13 * self.rank) % max(1, numel)
            ar.flat[:numel - sh].copy_(noise[sh:])
            ar.flat[numel - sh:numel].copy_(noise[:sh])
            ar.flat[:numel].mul_(1.0 + 0.125 * (k % 5))
            want.copy_(ar.flat)
            self._reduce(want, td.ReduceOp.SUM)
            torch.cuda.synchronize(self.device)
            try:
                if k % 2 == 0:
                    ar.all_reduce(ar.flat[:numel], channel=0)
                else:
                    ar.all_reduce(ar.flat[cut:numel], channel=0)
                    ar.all_reduce(ar.flat[:cut], channel=1)
                torch.cuda.synchronize(self.device)
                good = good and ar.error() == 0 and bool(torch.allclose(ar.flat, want, rtol=1e-4, atol=1e-4))
            except Exception:                                   # noqa: BLE001
                good = False
            if not self._agree(good):                           # stop early, TOGETHER (the body has a collective)
                return False
        return good                                             # the caller restores the production poll bound

    def _time_all_reduces(self, ar, numel: int, iters: int = 20) -> tuple:
        """us per all-reduce of the gradient buffer: (direct kernel at its best grid size, process group), max over
        ranks. The direct kernel is timed with 16 / 32 / 64 workgroups
        and left at the fastest - identical on every rank, because the decision is taken on all-reduced times."""
        import time
        tmp = torch.zeros(numel, device=self.device)

        def clock(fn):
            for _ in range(3):
                fn()
                torch.cuda.synchronize(self.device)
            t0 = time.perf_counter()
            for _ in range(iters):                              # one at a time (the same for both paths): gloo's CUDA path,
                fn()                                            # used for rehearsals, stalls on back-to-back collectives
                torch.cuda.synchronize(self.device)
            return (time.perf_counter() - t0) / iters * 1e6

        cands = [16, 32, 64]
        out = []
        for nb in cands:
            ar.set_blocks(nb)
            out.append(clock(lambda: ar.all_reduce(ar.flat[:numel], channel=2)))
        out.append(clock(lambda: self._reduce(tmp, td.ReduceOp.SUM)))
        v = torch.tensor(out, dtype=torch.float64)
        if self._device_collectives():
            v = v.to(self.device)
        td.all_reduce(v, op=td.ReduceOp.MAX, group=self.group)
        ts = [float(x) for x in v.tolist()]
        best = min(range(len(cands)), key=lambda i: ts[i])
        ar.set_blocks(cands[best])
        self._xgmi_sweep = {str(nb): round(t, 1) for nb, t in zip(cands, ts)}
        return ts[best], ts[-1]

    def check_health(self) -> None:
        """Raises if a poll of the direct all-reduce timed out since the last call (a peer did not arrive). Synchronises."""
        if self.xgmi is not None:
            e = self.xgmi.error()
            if e:
                raise RuntimeError("direct xGMI all-reduce: barrier %d timed out on rank %d" % (e - 1, self.rank))

    # ---- collectives --------------------------------------------------------------------------------
    def all_reduce_flat(self, flat: torch.Tensor) -> None:
        """SUM over ranks, in place, one collective for the whole gradient."""
        if not self.enabled:
            return
        if self.xgmi is not None and self.xgmi.owns(flat):
            self.xgmi.all_reduce(flat, channel=0)
            return
        td.all_reduce(flat, op=td.ReduceOp.SUM, group=self.group)

    def all_reduce_async(self, flat: torch.Tensor, channel: int = 0):
        """SUM over ranks, in place, asynchronous: the collective runs on a side stream (RCCL's own, or ours for the
        direct xGMI kernel) after the work already queued on the current stream; ``.wait()`` on the returned handle
        orders the current stream after it. Collectives that may be in flight together use different channels."""
        if not self.enabled:
            return _Done()
        if self.xgmi is not None and self.xgmi.owns(flat):
            side = self._xgmi_stream
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                self.xgmi.all_reduce(flat, channel=channel)
            return _StreamWork(side)
        return td.all_reduce(flat, op=td.ReduceOp.SUM, async_op=True, group=self.group)

    def all_reduce_grads(self, module: torch.nn.Module) -> None:
        """PyTorch-learner path: flatten → one all-reduce → average → scatter back."""
        if not self.enabled:
            return
        grads = [p.grad for p in module.parameters() if p.grad is not None]
        flat = torch.cat([g.reshape(-1) for g in grads])
        td.all_reduce(flat, op=td.ReduceOp.SUM, group=self.group)
        flat.div_(self.world_size)
        o = 0
        for g in grads:
            n = g.numel()
            g.copy_(flat[o:o + n].view_as(g))
            o += n

    # ---- collectives inside hipGraphs -------------------------------------------------------------------
    @property
    def graph_collectives(self) -> bool:
        return self.graph_collectives_for(None)

    def graph_collectives_for(self, flat: Optional[torch.Tensor] = None) -> bool:
        """May an engine put the collectives on its gradient buffer ``flat`` INTO the step graph (no host work between graph
        pieces)? (``flat=None``: on the direct all-reduce's buffer, if there is one.)

        * ``flat`` lives in the direct xGMI all-reduce's buffer: yes - the collective is an ordinary kernel launch
          (``DRL_GRAPH_COLLECTIVES=0`` still says no). An engine that did NOT get that buffer (a second engine of the
          process while the first holds it, or a buffer too small) goes through the process group like any other;
        * RCCL: only on request (``DRL_GRAPH_COLLECTIVES=1``, checked by a captured trial all-reduce that every rank must
          pass). Capturing a library collective races with the process group's watchdog thread (docs/DESIGN.md section 5:
          an eager collective retired less than a polling period before the capture aborts the process), and the only
          guard is a sleep; the default for RCCL is therefore the split schedule - graphs with the collectives issued
          between them - which has no such window."""
        env = os.environ.get("DRL_GRAPH_COLLECTIVES")
        if self.xgmi is not None and env != "0" and (flat is None or self.xgmi.owns(flat)):
            return True
        if env != "1":
            if self.enabled and "graph->split" not in " ".join(self.fallbacks):
                self.fallbacks.append("one graph->split graphs: RCCL collectives are issued between graph pieces")
            return False
        if getattr(self, "_graph_ok", None) is None:
            self._graph_ok = self._trial_graph_collective()
        return self._graph_ok

    def quiesce(self) -> None:
        """Call before capturing a hipGraph that CONTAINS collectives. The process group's watchdog thread polls the end
        events of eager collectives every 100 ms; HIP refuses ``hipEventQuery`` on an event whose stream is capturing
        (hipErrorCapturedEvent), which the watchdog turns into a process abort. A captured collective pulls the group's
        internal RCCL stream into the capture, so every eager collective issued before must have been RETIRED by the
        watchdog first: finish the device work, then leave the watchdog a few polling periods."""
        if not self.enabled or self.backend != "nccl":
            return
        import time
        torch.cuda.synchronize(self.device)
        time.sleep(0.35)

    def _trial_graph_collective(self) -> bool:
        if not self.enabled or self.device.type != "cuda" or self.backend != "nccl":
            return False
        ok = 1.0
        try:
            x = torch.ones(4096, device=self.device)
            td.all_reduce(x, group=self.group)                                   # communicator warm-up outside capture
            self.quiesce()
            buf = torch.ones(4096, device=self.device)
            side = torch.cuda.Stream(self.device)
            side.wait_stream(torch.cuda.current_stream())
            g = torch.cuda.CUDAGraph()
            with torch.cuda.stream(side):
                with torch.cuda.graph(g, capture_error_mode="thread_local"):
                    w = td.all_reduce(buf, op=td.ReduceOp.SUM, async_op=True, group=self.group)
                    w.wait()
                    buf.mul_(1.0 / self.world_size)            # back to ones: replays are idempotent
            torch.cuda.current_stream().wait_stream(side)
            for _ in range(2):
                g.replay()
            torch.cuda.synchronize(self.device)
            if not bool(torch.allclose(buf, torch.ones_like(buf), rtol=1e-5)):
                ok = 0.0
        except Exception:
            ok = 0.0
        try:                                                   # agree across ranks (eager collective)
            v = torch.tensor([ok], device=self.device)
            td.all_reduce(v, op=td.ReduceOp.MIN, group=self.group)
            ok = float(v.item())
        except Exception:
            ok = 0.0
        return ok > 0.5

    def broadcast_flat(self, flat: torch.Tensor, src: int = 0) -> None:
        if self.enabled:       # ``src`` is a rank of THIS object's group
            td.broadcast(flat, src=src if self.group is None else td.get_global_rank(self.group, src), group=self.group)

    def shard_scalars(self, total_prio: float, n_items: int) -> tuple:
        """→ (Σp over all shards, N over all shards)."""
        if not self.enabled:
            return total_prio, n_items
        t = torch.tensor([total_prio, float(n_items)], dtype=torch.float64, device=self.device)
        td.all_reduce(t, op=td.ReduceOp.SUM, group=self.group)
        return float(t[0]), int(t[1])

    def mean_count(self, n_local: torch.Tensor, out: torch.Tensor, tmp: torch.Tensor) -> None:
        """out <- round(sum over ranks of n_local / world) for int64 device scalars (sharded PER, engine/shard.py).
        ``out`` keeps its address: captured graphs read it. RCCL: stream-ordered, no host synchronisation. gloo on GPU
        tensors (rehearsals with ranks sharing a device) blocks the host inside the collective while step graphs with
        spinning inter-process kernels are still queued - observed to deadlock the shared device's queues - so that
        path drains the stream first and reduces on the host."""
        if not self.enabled:
            out.copy_(n_local)
            return
        if self.backend != "nccl" and n_local.is_cuda:
            torch.cuda.synchronize(self.device)
            t = n_local.cpu()
            td.all_reduce(t, op=td.ReduceOp.SUM, group=self.group)
            out.copy_((t + self.world_size // 2) // self.world_size)
            return
        tmp.copy_(n_local)
        td.all_reduce(tmp, op=td.ReduceOp.SUM, group=self.group)
        tmp.add_(self.world_size // 2)
        out.copy_(torch.div(tmp, self.world_size, rounding_mode="floor"))

    def max_scalar(self, x: float) -> float:
        if not self.enabled:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=self.device)
        td.all_reduce(t, op=td.ReduceOp.MAX, group=self.group)
        return float(t[0])

    def any_true(self, flag: bool) -> bool:
        """True on every rank iff ``flag`` is set on at least one (a stop request must take all data-parallel ranks out
        of their loops at the SAME step: the step contains collectives). Travels over gloo on a host tensor when the
        group has a CPU backend - no device synchronisation - else as a device scalar."""
        if not self.enabled:
            return bool(flag)
        on_cpu = "gloo" in str(self.backend)
        t = torch.tensor([1.0 if flag else 0.0], device="cpu" if on_cpu else self.device)
        td.all_reduce(t, op=td.ReduceOp.MAX, group=self.group)
        return float(t.item()) > 0.5

    def barrier(self) -> None:
        if self.enabled:
            if self.backend == "nccl":
                td.barrier(group=self.group, device_ids=[self.device.index])
            else:
                td.barrier(group=self.group)

    def close(self) -> None:
        if self.xgmi is not None:
            self.xgmi.close()
            self.xgmi = None
        if self.enabled and self._owns_world and td.is_initialized():
            td.destroy_process_group()

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


class Dist:
    def __init__(self, backend: Optional[str] = None, device: Optional[torch.device] = None, timeout_s: int = 600):
        import datetime
        self.rank = int(os.environ.get("RANK", "0"))
        self.world_size = int(os.environ.get("WORLD_SIZE", "1"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        if device is None:
            device = torch.device("cuda", self.local_rank) if torch.cuda.is_available() else torch.device("cpu")
        self.device = device
        if device.type == "cuda":
            torch.cuda.set_device(device)
        self.backend = backend or ("nccl" if device.type == "cuda" else "gloo")
        # DRL_FORCE_DIST=1 initialises the process group even for one rank (exercises the RCCL path on a 1-GPU box)
        self.enabled = self.world_size > 1 or os.environ.get("DRL_FORCE_DIST") == "1"
        if self.enabled and not td.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29511")
            kw = {}
            if self.backend == "nccl":
                kw["device_id"] = device
            td.init_process_group(self.backend, rank=self.rank, world_size=self.world_size,
                                  timeout=datetime.timedelta(seconds=timeout_s), **kw)

    # ---- collectives --------------------------------------------------------------------------------
    def all_reduce_flat(self, flat: torch.Tensor) -> None:
        """SUM over ranks, in place, one collective for the whole gradient."""
        if self.enabled:
            td.all_reduce(flat, op=td.ReduceOp.SUM)

    def all_reduce_async(self, flat: torch.Tensor):
        """SUM over ranks, in place, asynchronous: the collective runs on the backend's own stream after the work
        already queued on the current stream; ``.wait()`` on the returned handle orders the current stream after it."""
        if not self.enabled:
            return _Done()
        return td.all_reduce(flat, op=td.ReduceOp.SUM, async_op=True)

    def all_reduce_grads(self, module: torch.nn.Module) -> None:
        """PyTorch-learner path: flatten → one all-reduce → average → scatter back."""
        if not self.enabled:
            return
        grads = [p.grad for p in module.parameters() if p.grad is not None]
        flat = torch.cat([g.reshape(-1) for g in grads])
        td.all_reduce(flat, op=td.ReduceOp.SUM)
        flat.div_(self.world_size)
        o = 0
        for g in grads:
            n = g.numel()
            g.copy_(flat[o:o + n].view_as(g))
            o += n

    # ---- collectives inside hipGraphs -------------------------------------------------------------------
    @property
    def graph_collectives(self) -> bool:
        """May the engines capture RCCL collectives INTO their step graph (no host work between graph pieces: 0.32 vs
        0.38 ms per Ape-X step at one rank)? ``DRL_GRAPH_COLLECTIVES`` = 0 / 1 decides when set; otherwise a trial
        decides once: a small all-reduce is captured, replayed twice and checked on every rank, and the ranks agree
        on the verdict (MIN) so that all of them build the same kind of step."""
        env = os.environ.get("DRL_GRAPH_COLLECTIVES")
        if env in ("0", "1"):
            return env == "1"
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
        time.sleep(float(os.environ.get("DRL_QUIESCE_S", "0.35")))

    def _trial_graph_collective(self) -> bool:
        if not self.enabled or self.device.type != "cuda" or self.backend != "nccl":
            return False
        ok = 1.0
        try:
            x = torch.ones(4096, device=self.device)
            td.all_reduce(x)                                   # communicator warm-up outside capture
            self.quiesce()
            buf = torch.ones(4096, device=self.device)
            side = torch.cuda.Stream(self.device)
            side.wait_stream(torch.cuda.current_stream())
            g = torch.cuda.CUDAGraph()
            with torch.cuda.stream(side):
                with torch.cuda.graph(g, capture_error_mode="thread_local"):
                    w = td.all_reduce(buf, op=td.ReduceOp.SUM, async_op=True)
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
            td.all_reduce(v, op=td.ReduceOp.MIN)
            ok = float(v.item())
        except Exception:
            ok = 0.0
        return ok > 0.5

    def broadcast_flat(self, flat: torch.Tensor, src: int = 0) -> None:
        if self.enabled:
            td.broadcast(flat, src=src)

    def shard_scalars(self, total_prio: float, n_items: int) -> tuple:
        """→ (Σp over all shards, N over all shards)."""
        if not self.enabled:
            return total_prio, n_items
        t = torch.tensor([total_prio, float(n_items)], dtype=torch.float64, device=self.device)
        td.all_reduce(t, op=td.ReduceOp.SUM)
        return float(t[0]), int(t[1])

    def max_scalar(self, x: float) -> float:
        if not self.enabled:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=self.device)
        td.all_reduce(t, op=td.ReduceOp.MAX)
        return float(t[0])

    def barrier(self) -> None:
        if self.enabled:
            if self.backend == "nccl":
                td.barrier(device_ids=[self.device.index])
            else:
                td.barrier()

    def close(self) -> None:
        if self.enabled and td.is_initialized():
            td.destroy_process_group()

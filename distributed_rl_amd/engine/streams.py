"""Fork/join of independent kernel chains onto side HIP streams.

Most kernels of a learner step are too small to fill 256 CUs (a 64x64-tile GEMM with 32-256 workgroups, an LSTM time
step with 32), and the step has plenty of independent work: the target-network forward next to the online forward,
every weight-gradient GEMM next to the input-gradient chain of its layer. ``ForkJoin`` issues such branches on side
streams; under ``torch.cuda.graph`` capture the ``wait_stream`` edges become parallel branches of the hipGraph, in eager
mode they are ordinary event waits. ``enabled=False`` runs everything on the current stream.
"""
from __future__ import annotations

import contextlib
import os

import torch


class ForkJoin:
    def __init__(self, device, n_side: int = 2, enabled: bool = True):
        self.enabled = enabled
        self.side = [torch.cuda.Stream(device) for _ in range(n_side)] if self.enabled else []
        self._open = set()           # side streams forked since the last join

    def fork(self, i: int):
        """Context manager: the body runs on side stream ``i`` after everything issued so far on the current stream."""
        if not self.enabled:
            return contextlib.nullcontext()
        side = self.side[i % len(self.side)]
        side.wait_stream(torch.cuda.current_stream())
        self._open.add(i % len(self.side))
        return torch.cuda.stream(side)

    def join(self) -> None:
        """The current stream waits for every side stream forked since the last join (a wait on an idle stream would
        still be a cross-queue edge of the replayed graph)."""
        cur = torch.cuda.current_stream()
        for i in sorted(self._open):
            cur.wait_stream(self.side[i])
        self._open.clear()


def capture_graph(graph: "torch.cuda.CUDAGraph"):
    """``torch.cuda.graph`` with thread-local capture errors: other threads of the process (the RCCL watchdog polling
    its events, the ingest thread's H2D copies, metrics) may touch the HIP runtime while a step is being captured;
    only calls of the CAPTURING thread may invalidate the capture."""
    return torch.cuda.graph(graph, capture_error_mode="thread_local")

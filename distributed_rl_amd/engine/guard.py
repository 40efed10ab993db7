"""Discarding a poisoned learner step (the reference has no in-kernel waits to fail).

The persistent LSTM / BPTT kernels and the direct xGMI all-reduce hand data between workgroups / GPUs inside a launch
with BOUNDED polls: a poll that gives up sets a device word and the kernel finishes with wrong numbers instead of
hanging the GPU. Those words are the ``guards`` of the step. The first optimizer launch folds them into a sticky
``fault`` word; the update launch and the priority write-back read it and do nothing when it is set, so a poisoned
gradient never reaches the parameters, the optimizer moments or the replay priorities - this step and every later one
are discarded. A pinned host mirror of the word is written by the same launch (a system-scope store to device-mapped host
memory: a memcpy node of the step graph for these 4 bytes cost 4 us per step), and ``check()`` - called at the top of ``step()`` without synchronising - raises as soon as
the mirror shows it, i.e. at most the host's run-ahead (a few steps) later.
"""
from __future__ import annotations

from typing import List

import torch


class StepGuard:
    NAMES_MAX = 4

    def __init__(self, device):
        # [0] = sticky bit mask of the guards that fired, [2:4] = address of the pinned host mirror (written by the kernel
        # that folds the guards: k_sumsq_partial, optim.hip)
        self.fault = torch.zeros(4, dtype=torch.int32, device=device)
        self.host = torch.zeros(4, dtype=torch.int32).pin_memory()
        if torch.device(device).type == "cuda":
            self.fault[2:4].copy_(torch.tensor([self.host.data_ptr()], dtype=torch.int64).view(torch.int32))
        self.sources: List = []
        self.names: List[str] = []

    def add(self, name: str, word) -> None:
        """``word``: int32 device tensor or raw device address that is non-zero after a timed-out wait."""
        if word is None or (isinstance(word, int) and word == 0):
            return
        assert len(self.sources) < self.NAMES_MAX
        self.sources.append(word)
        self.names.append(name)

    def mirror(self) -> None:
        """Nothing to launch: the kernel that folds the guards also stores the fault word to the pinned host mirror."""

    def fired(self) -> List[str]:
        m = int(self.host[0])
        return [n for k, n in enumerate(self.names) if m >> k & 1]

    def check(self) -> None:
        if int(self.host[0]):
            raise RuntimeError("learner step discarded: a bounded in-kernel wait timed out (%s); no update has been applied "
                               "since, the parameters are those of the last good step" % ", ".join(self.fired()))

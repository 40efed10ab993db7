"""Sharded prioritized replay: the scalars that make per-shard sampling equivalent to ONE tree (SURVEY §2.5 M8, §5.8;
the reference has a single replay, ape_x/learner.py:64-68).

Rank r draws its batch from its own shard, so item i is drawn with probability ``p_i / (W * Σp_r)`` (every rank
contributes 1/W of the global batch) and its importance weight is ``(N_global * p_i / (W * Σp_r))^-β``. The kernels compute
``(n * p_i / Σp_local)^-β`` with ``n`` read from an int64 DEVICE scalar at sampling time, so all a sharded run has to do is
keep ``n = N_global / W`` in that scalar: ``ShardedIS.n_is``. It is refreshed OUTSIDE the captured step graph every
``shard_refresh`` (64) steps (one 8-byte all-reduce, no host synchronisation), so shards that fill at different rates
stay correct without re-capturing anything; with one rank it simply aliases the replay's own item counter.

The max-weight normalisation (``weights / weights.max()``, ape_x/learner.py:66-68) is GLOBAL in the Ape-X engine: the
batch of step t + 1 is drawn inside step t, every rank's maximum rides in a small mailbox in front of the flat gradient
through step t's gradient all-reduce, and the maximum over the ranks is in place one step before the TD kernel reads it
(engine/apex.py MAILBOX, td.hip k_wmax_put / k_wmax_take; the very first step, drawn at its own start, still uses the
rank's own maximum). The R2D2 engine draws at the start of the step that uses the batch, so its ranks' maxima are
gathered by a 128-byte all-reduce of the same mailbox that runs UNDER the forward pass (~0.9 ms; the weights are needed
only by the TD kernel at its end): engine/r2d2.py ``_global_wmax``, td.hip k_is_wmax_put / k_is_weights. That collective
has to sit inside the step graph - the direct xGMI kernel, or RCCL where the start-up trial allowed capturing it; with the
two-graph schedule (RCCL issued between graphs) the R2D2 ranks normalise by their own batch maximum, a per-rank constant
scale of their gradient share.
"""
from __future__ import annotations

import torch


class ShardedIS:
    def _init_shard(self, replay, dist, device) -> None:
        self._shard_on = dist is not None and dist.enabled and dist.world_size > 1
        self.shard_refresh = 64
        if self._shard_on:
            self.n_is = torch.zeros(1, dtype=torch.int64, device=device)
            self._n_tmp = torch.zeros(1, dtype=torch.int64, device=device)
        else:
            self.n_is = replay.n_items
        self._shard_fresh = False

    def _maybe_refresh_shard(self) -> None:
        """Top of every ``step()``: the first step and every ``shard_refresh``-th one re-read the global item count. All
        ranks step in lockstep, so they take this collective at the same points."""
        if self._shard_on and (not self._shard_fresh or self.steps_done % self.shard_refresh == 0):
            self.refresh_shard_scalars()
            self._shard_fresh = True

    def refresh_shard_scalars(self) -> None:
        """Collective (every rank calls it at the same step): n_is <- round(sum_r N_r / W). Device-side, asynchronous."""
        if self._shard_on:
            self.dist.mean_count(self.replay.n_items, self.n_is, self._n_tmp)

    def shard_n_is(self) -> int:
        """Host copy of the item count the IS weights use (tests / stats; synchronises)."""
        return int(self.n_is.item())

#!/usr/bin/env python
"""Loss of the R2D2 engine every 10 steps (fixed synthetic replay, fixed seeds): how much one step's loss scatters."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from distributed_rl_amd.config import R2D2Config  # noqa: E402
from distributed_rl_amd.engine.r2d2 import R2D2Engine  # noqa: E402
from distributed_rl_amd.replay.device_seq import DeviceSequenceReplay  # noqa: E402

cfg = R2D2Config()
rep = DeviceSequenceReplay(2000, "cuda", cfg, seed=0)
rep.fill_synthetic(2000, n_action=4, seed=0)
eng = R2D2Engine(4, cfg, rep, "cuda")
out = []
for i in range(300):
    eng.step()
    if i % 10 == 9:
        out.append("%.3f" % eng.stats()["loss"])
print(" ".join(out))

#!/usr/bin/env python
"""R2D2 learner step alone (64 sequences x 120 steps): timing, and a target for rocprofv3 --kernel-trace --stats."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_rl_amd.config import R2D2Config  # noqa: E402
from distributed_rl_amd.engine.r2d2 import R2D2Engine  # noqa: E402
from distributed_rl_amd.replay.device_seq import DeviceSequenceReplay  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--cap", type=int, default=2000)
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--engine-arg", action="append", default=[], metavar="KEY=VALUE",
                    help="tuning argument of R2D2Engine (fwd_chunks, bwd_chunks, bwd_cuts, bptt_groups, fwd_groups_last); repeatable")
    args = ap.parse_args()
    kw = {}
    for item in args.engine_arg:
        k, v = item.split("=", 1)
        kw[k] = int(v) if v.lstrip("-").isdigit() else v
    cfg = R2D2Config()
    rep = DeviceSequenceReplay(args.cap, "cuda", cfg, seed=0)
    rep.fill_synthetic(args.cap, n_action=4, seed=0)
    eng = R2D2Engine(4, cfg, rep, "cuda", use_graph=not args.no_graph, **kw)
    for _ in range(args.warmup):
        eng.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        eng.step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("r2d2: %.3f ms/step  %.1f updates/s  loss %.4f" % (1e3 * dt / args.steps, args.steps / dt, eng.stats()["loss"]))


if __name__ == "__main__":
    main()

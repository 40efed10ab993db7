#!/usr/bin/env python
"""The three conv layers (forward, bias + ReLU) through MIOpen (torch conv2d, bf16 channels_last) against the hand-written
kernels, at the Ape-X learner's forward size (1536 frames) and at an R2D2 chunk (3840 frames)."""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from distributed_rl_amd.ops import hip as H  # noqa: E402

DEV = "cuda"
LAYERS = {1: (4, 32, 8, 4, 84), 2: (32, 64, 4, 2, 20), 3: (64, 64, 3, 1, 9)}


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        with torch.cuda.graph(g):
            for _ in range(reps):
                fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    g.replay(); torch.cuda.synchronize()
    a.record(); g.replay(); b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / reps


torch.backends.cudnn.benchmark = True
for frames in (1536, 3840):
    for layer, (cin, cout, ks, st, hw) in LAYERS.items():
        oh = (hw - ks) // st + 1
        w = (torch.randn(cout, ks, ks, cin, device=DEV) * 0.05).to(torch.bfloat16)
        b = torch.randn(cout, device=DEV)
        if layer == 1:
            x = torch.randint(0, 256, (frames, hw, hw, cin), dtype=torch.uint8, device=DEV)
            xl = (x.float() / 255).to(torch.bfloat16).permute(0, 3, 1, 2)          # NCHW view of NHWC memory = channels_last
        else:
            x = torch.relu(torch.randn(frames, hw, hw, cin, device=DEV)).to(torch.bfloat16)
            xl = x.permute(0, 3, 1, 2)
        wl = w.permute(0, 3, 1, 2)
        bl = b.to(torch.bfloat16)
        out = torch.zeros(frames, oh, oh, cout, dtype=torch.bfloat16, device=DEV)
        t_own = timeit(lambda: H.conv_fwd(layer, x, w, b, out, frames))
        try:
            t_lib = timeit(lambda: F.relu(F.conv2d(xl, wl, bl, stride=st)))
            ref = F.relu(F.conv2d(xl, wl, bl, stride=st)).permute(0, 2, 3, 1)
            err = float((ref.float() - out.float()).abs().max())
        except Exception as e:  # noqa: BLE001
            t_lib, err = float("nan"), repr(e)[:80]
        fl = 2.0 * frames * oh * oh * cout * ks * ks * cin
        print("frames %d conv%d: own %.1f us (%.0f TFLOP/s)   MIOpen conv2d+relu %.1f us (%.0f TFLOP/s)   max diff %s"
              % (frames, layer, t_own, fl / t_own / 1e6, t_lib, fl / t_lib / 1e6, err))

#!/usr/bin/env python
"""Ape-X FC1 products (B = 512) through gemm_big.hip tiles vs the k_multi launches the step uses (scripts/kbench.py cases)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from distributed_rl_amd.ops import hip as H  # noqa: E402

dev = torch.device("cuda")
B = 512
bf = lambda *s: ((torch.rand(*s, device=dev) * 2 - 1) * 0.1).to(torch.bfloat16)  # noqa: E731
dh, a3, w = bf(B, 1024), bf(B, 3136), bf(1024, 3136)
a3f, hf = bf(3 * B, 3136), torch.zeros(3 * B, 1024, dtype=torch.bfloat16, device=dev)
g = torch.zeros(1024, 3136, device=dev)
d3 = torch.zeros(B, 3136, dtype=torch.bfloat16, device=dev)
bias = torch.zeros(1024, device=dev)


def t(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for tile in (0, 1, 2, 3, 4, 5):
    r = {}
    r["dW TN 1024x3136x512"] = t(lambda: H.gemm_big(H.TN, dh, a3, g, tile=tile))
    r["dX NN 512x3136x1024 +mask"] = t(lambda: H.gemm_big(H.NN, dh, w, d3, mask=a3, tile=tile))
    r["fwd NT 1536x1024x3136 +bias+relu"] = t(lambda: H.gemm_big(H.NT, a3f, w, hf, bias=bias, relu=True, tile=tile))
    print("tile", tile, {k: round(v, 1) for k, v in r.items()})

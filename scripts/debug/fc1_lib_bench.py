#!/usr/bin/env python
"""FC1 forward (bias + ReLU) through hipBLASLt (torch._addmm_activation) against the hand-written LDS-DMA GEMM."""
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from distributed_rl_amd.ops import hip as H  # noqa: E402

DEV = "cuda"


def timeit(fn, reps=50):
    for _ in range(5):
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


w = (torch.randn(1024, 3136, device=DEV) * 0.02).to(torch.bfloat16)
bias_f = torch.randn(1024, device=DEV)
bias_b = bias_f.to(torch.bfloat16)
for M in (512, 1024, 1536):
    a = torch.relu(torch.randn(M, 3136, device=DEV)).to(torch.bfloat16)
    out = torch.empty(M, 1024, dtype=torch.bfloat16, device=DEV)
    mine = torch.empty_like(out)
    t_mine = timeit(lambda: H.gemm(H.NT, a, w, mine, M, 1024, 3136, bias=bias_f, flags=H.EPI_RELU, big=H.FC1_FWD_VARIANT))
    try:
        t_lib = timeit(lambda: torch._addmm_activation(bias_b, a, w.t(), use_gelu=False, out=out))
        err = float((out.float() - mine.float()).abs().max())
    except Exception as e:  # noqa: BLE001
        t_lib, err = float("nan"), repr(e)[:80]
    t_mm = timeit(lambda: torch.mm(a, w.t(), out=out))
    print("M=%d  own %.1f us   _addmm_activation %.1f us (max diff %s)   plain mm %.1f us" % (M, t_mine, t_lib, err, t_mm))

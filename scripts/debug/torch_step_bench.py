#!/usr/bin/env python
"""Same-hardware baseline: the reference's learner step written the reference's way - PyTorch autograd on the DuelingDQN
module (MIOpen / hipBLASLt kernels), batch 512 already resident on the GPU (the reference additionally unpickles and
uploads 115 MB per step, ape_x/learner.py:64-72), smooth-L1 TD loss, clip_grad_norm_(40), RMSprop step - in fp32 as the
reference runs it and under bf16 autocast. No replay sampling, no priority write-back: a lower bound of what that design
costs on an MI355X, to put the graph-captured engine's 0.24 ms per step next to."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from distributed_rl_amd.models import DuelingDQN  # noqa: E402

DEV = "cuda"
B, A = 512, 4
torch.backends.cudnn.benchmark = True


def run(autocast: bool, steps: int = 200, warm: int = 20) -> float:
    torch.manual_seed(0)
    model, target = DuelingDQN(A).to(DEV), DuelingDQN(A).to(DEV)
    target.load_state_dict(model.state_dict())
    opt = torch.optim.RMSprop(model.parameters(), lr=0.00025 / 4, alpha=0.95, eps=1.5e-7)
    s = torch.randint(0, 256, (B, 84, 84, 4), dtype=torch.uint8, device=DEV)
    s2 = torch.randint(0, 256, (B, 84, 84, 4), dtype=torch.uint8, device=DEV)
    a = torch.randint(0, A, (B,), device=DEV)
    r, d, w = torch.randn(B, device=DEV), torch.zeros(B, device=DEV), torch.rand(B, device=DEV)

    def step():
        opt.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
            td = model.td_batch(target, s, a, r, s2, d, w, 0.6, 0.999 ** 3)
        td.loss.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), 40.0)
        opt.step()

    for _ in range(warm):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


for ac in (False, True):
    ms = run(ac)
    print("PyTorch autograd learner step, B=512, %s: %.3f ms/step = %.0f updates/s = %.0f transitions/s"
          % ("bf16 autocast" if ac else "fp32", ms, 1e3 / ms, 1e3 / ms * B))


def run_r2d2(steps: int = 5, warm: int = 2) -> float:
    """The R2D2 step the same way: DuelingLSTMDQN.sequence_td (time loop in Python over LSTMCell, as models.py:124-159
    does), 64 sequences x 120 steps resident on the GPU, Adam."""
    from distributed_rl_amd.models import DuelingLSTMDQN
    Bq, T, burn, n = 64, 120, 40, 5
    torch.manual_seed(0)
    model, target = DuelingLSTMDQN(A, Bq, burn, n).to(DEV), DuelingLSTMDQN(A, Bq, burn, n).to(DEV)
    target.load_state_dict(model.state_dict())
    opt = torch.optim.Adam(model.parameters(), lr=1e-4, eps=1e-3)
    frames = torch.randint(0, 256, (T, Bq, 84, 84, 4), dtype=torch.uint8, device=DEV)
    a = torch.randint(0, A, (T, Bq), device=DEV)
    r, d = torch.randn(T, Bq, device=DEV), torch.zeros(T, Bq, device=DEV)
    lens = torch.full((Bq,), T, dtype=torch.long, device=DEV)
    h0, c0, w = torch.zeros(Bq, 512, device=DEV), torch.zeros(Bq, 512, device=DEV), torch.rand(Bq, device=DEV)

    def step():
        opt.zero_grad(set_to_none=True)
        td = model.sequence_td(target, frames, a, r, d, lens, h0, c0, w, 0.9, 0.997 ** n, 1e-3)
        td.loss.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), 40.0)
        opt.step()

    for _ in range(warm):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


try:
    ms = run_r2d2()
    print("PyTorch autograd R2D2 step, 64 x 120, fp32: %.1f ms/step = %.1f updates/s" % (ms, 1e3 / ms))
except Exception as e:  # noqa: BLE001
    print("R2D2 autograd baseline failed: %r" % (e,))

#!/usr/bin/env python
"""What the memory system sustains for pure writes, pure reads and copies (1 GiB buffers, torch kernels)."""
import torch

dev = torch.device("cuda")
n = 1 << 28                                   # 1 GiB of float32
a = torch.empty(n, device=dev)
b = torch.empty(n, device=dev)


def t(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


gb = n * 4 / 1e9
print("fill  (write only)      %.2f TB/s" % (gb / t(lambda: a.zero_()) / 1e3))
print("sum   (read only)       %.2f TB/s" % (gb / t(lambda: a.sum()) / 1e3))
s = t(lambda: b.copy_(a))
print("copy  (read + write)    %.2f TB/s total (%.2f each way)" % (2 * gb / s / 1e3, gb / s / 1e3))
s = t(lambda: torch.add(a, b, out=b))
print("add   (2 reads 1 write) %.2f TB/s total" % (3 * gb / s / 1e3))

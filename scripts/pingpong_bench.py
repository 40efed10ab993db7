"""Inter-workgroup hand-off latency on one MI355X: flag + payload ping-pong between two workgroups (csrc/hip/pingpong.hip)
per memory kind, payload size and XCD placement. One HOP = one direction (store payload, flag, poll, read payload).
Usage: python scripts/pingpong_bench.py > profiles/pingpong_floor_r3.txt"""
import ctypes
import torch
from distributed_rl_amd import native

torch.cuda.init()
torch.zeros(1, device="cuda")
hip = native.hip()
ROUNDS = 2000
print("kind            payload  blocks   xcc(a,b)  us/hop   cycles/hop  mismatches  gave_up")
for kind, kname in ((0, "coarse+sc1"), (1, "fine-grained"), (2, "uncached")):
    for payload in (4, 1024, 4096):
        for a, b, grid in ((0, 8, 16), (0, 1, 16), (0, 4, 16)):
            out = (ctypes.c_ulonglong * 5)()
            us = ctypes.c_float()
            best = None
            for _ in range(3):
                rc = hip.drl_pingpong(kind, payload, ROUNDS, a, b, grid, out, ctypes.byref(us))
                if rc != 0:
                    break
                rec = (us.value / (2 * ROUNDS), out[0] / (2 * ROUNDS), int(out[1]), (int(out[2]), int(out[3])), int(out[4]))
                best = rec if best is None or rec[0] < best[0] else best
            if best is None:
                print("%-15s %6d   %d,%d   launch failed rc=%d" % (kname, payload, a, b, rc))
                continue
            print("%-15s %6d   %2d,%-2d    %d,%d      %6.2f   %8.0f    %d           %d" %
                  (kname, payload, a, b, best[3][0], best[3][1], best[0], best[1], best[2], best[4]), flush=True)

print()
print("all-gather among n workgroups (the recurrence's hand-off: every round each workgroup stores its slice, raises its flag,")
print("waits for all flags, reads all slices). R2D2 recurrence: 32 workgroups x 2 KB slices of h (64 x 512 bf16) = 64 KB per step.")
print("kind            n_wg  slice   placement        us/round  mismatches  gave_up")
for kind, kname in ((0, "coarse+sc1"), (1, "fine-grained"), (2, "uncached")):
    for n_wg, slice_b, stride, place in ((32, 2048, 1, "spread over XCDs"), (32, 2048, 8, "one XCD"), (16, 4096, 1, "spread over XCDs"),
                                          (8, 8192, 1, "one per XCD"), (8, 8192, 8, "one XCD"), (32, 64, 1, "spread (flags only)")):
        out = (ctypes.c_ulonglong * 5)()
        us = ctypes.c_float()
        best = None
        for _ in range(3):
            rc = hip.drl_allgather_bench(kind, slice_b, n_wg, ROUNDS, stride, out, ctypes.byref(us))
            if rc != 0:
                break
            rec = (us.value / ROUNDS, int(out[1]), int(out[4]))
            best = rec if best is None or rec[0] < best[0] else best
        if best is None:
            print("%-15s %3d  %5d   %-16s launch failed rc=%d" % (kname, n_wg, slice_b, place, rc))
            continue
        print("%-15s %3d  %5d   %-16s %7.2f   %d           %d" % (kname, n_wg, slice_b, place, best[0], best[1], best[2]), flush=True)

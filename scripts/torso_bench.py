"""Fused torso forward (conv_torso.hip) against the three per-layer launches, at the learners' shapes.
Usage: python scripts/torso_bench.py [--reps 200]"""
import argparse
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import json
import torch
from distributed_rl_amd.ops import hip as H

DEV = "cuda"


def net(seed, frames, off, save):
    g = torch.Generator().manual_seed(seed)
    bf = dict(dtype=torch.bfloat16, device=DEV)
    w = [(torch.randn(n, generator=g) * s).to(DEV).to(torch.bfloat16) for n, s in ((32 * 256, 0.06), (64 * 512, 0.04), (64 * 576, 0.04))]
    b = [(torch.randn(n, generator=g) * 0.05).to(DEV) for n in (32, 64, 64)]
    a1, a2, a3 = torch.zeros(frames, 20, 20, 32, **bf), torch.zeros(frames, 9, 9, 64, **bf), torch.zeros(frames, 7, 7, 64, **bf)
    return dict(w=w, b=b, a=(a1, a2, a3), off=off, frames=frames, t=H.TorsoNet(off, frames, w, b, a3, a1, a2, save))


def timeit(fn, reps):
    for _ in range(10):
        fn()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g):
            for _ in range(10):
                fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps // 10):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000.0 / (reps // 10 * 10)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=200)
    ap.add_argument("--only", default="", help="substring of the case name")
    ap.add_argument("--eager", action="store_true", help="a few eager launches of the fused kernel only (for rocprofv3 --pmc)")
    ap.add_argument("--dbg", type=int, default=0, help="skip roles of the fused kernel (1: conv1, 2: conv2, 4: conv3): timing only")
    args = ap.parse_args()
    from distributed_rl_amd import native
    native.hip().drl_torso_set_debug(args.dbg)
    planes = 60000
    store = torch.randint(0, 256, (planes, 7056), dtype=torch.uint8, device=DEV)
    for name, f1, f2, save in (("apex B=512 (online 1024 + target 512)", 1024, 512, (0, 512)),
                               ("r2d2 chunk (online 4608 + target 4608)", 4608, 4608, (2560, 4608)),
                               ("actor 64 envs", 64, 0, (0, 0)), ("one net 1536", 1536, 0, (0, 0))):
        if args.only and args.only not in name:
            continue
        off = (torch.randint(0, planes - 4, (f1 + f2,), device=DEV) * 7056)
        n1 = net(1, f1, off[:f1].contiguous(), save)
        n2 = net(2, f2, off[f1:].contiguous(), (0, 0)) if f2 else None

        def three():
            if n2 is None:
                H.conv_fwd(1, store, n1["w"][0], n1["b"][0], n1["a"][0], f1, frame_off=n1["off"])
                H.conv_fwd(2, n1["a"][0], n1["w"][1], n1["b"][1], n1["a"][1], f1)
                H.conv_fwd(3, n1["a"][1], n1["w"][2], n1["b"][2], n1["a"][2], f1)
                return
            H.conv_fwd_dual(1, store, f1, n1["w"][0], n1["b"][0], n1["a"][0], None, f2, n2["w"][0], n2["b"][0], n2["a"][0],
                            frame_off1=n1["off"], frame_off2=n2["off"])
            for l in (2, 3):
                H.conv_fwd_dual(l, n1["a"][l - 2], f1, n1["w"][l - 1], n1["b"][l - 1], n1["a"][l - 1],
                                n2["a"][l - 2], f2, n2["w"][l - 1], n2["b"][l - 1], n2["a"][l - 1])

        def fused():
            H.torso_fwd(store, n1["t"], n2["t"] if n2 else None)

        if args.eager:
            for _ in range(5):
                fused()
            torch.cuda.synchronize()
            continue
        t3, tf = timeit(three, args.reps), timeit(fused, args.reps)
        flops = (f1 + f2) * 2 * (400 * 32 * 256 + 81 * 64 * 512 + 49 * 64 * 576)
        print(json.dumps({"case": name, "three_launch_us": round(t3, 2), "fused_us": round(tf, 2),
                          "fused_tflops": round(flops / tf / 1e6, 1), "speedup": round(t3 / tf, 3)}), flush=True)


if __name__ == "__main__":
    main()

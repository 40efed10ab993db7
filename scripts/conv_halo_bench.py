#!/usr/bin/env python
"""conv2 / conv3 forward: LDS-halo kernels (conv_halo.hip) against the implicit-GEMM tiles of gemm.hip, at the learners'
launch shapes (Ape-X: online 1024 + target 512 frames per launch; R2D2: 2 x 3840 / 2 x 4096 frames). Interleaved timing
with events, median of rounds; prints one JSON line per shape.   python scripts/conv_halo_bench.py [out.jsonl]"""
import json
import sys

import torch

sys.path.insert(0, ".")
from distributed_rl_amd.ops import hip as H  # noqa: E402

DEV = "cuda"
SH = {2: ((20, 20, 32), (9, 9, 64), 512), 3: ((9, 9, 64), (7, 7, 64), 576)}


def timeit(fns, rounds=7, iters=40):
    res = {k: [] for k in fns}
    for _ in range(rounds):
        for k, fn in fns.items():
            fn()
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(iters):
                fn()
            b.record()
            torch.cuda.synchronize()
            res[k].append(a.elapsed_time(b) * 1e3 / iters)
    return {k: round(sorted(v)[len(v) // 2], 2) for k, v in res.items()}


def main():
    out = open(sys.argv[1], "w") if len(sys.argv) > 1 else None
    if len(sys.argv) > 2 and sys.argv[2] == "wgrad":
        return wgrad(out)
    g = torch.Generator().manual_seed(0)
    for layer in (2, 3):
        ishape, oshape, K = SH[layer]
        for f1, f2 in ((1024, 512), (512, 0), (3840, 3840), (4096, 4096)):
            nets = []
            for f in (f1, f2):
                if not f:
                    nets.append((None, 0, None, None, None))
                    continue
                x = torch.relu(torch.randn(f, *ishape, generator=g)).to(torch.bfloat16).to(DEV)
                w = (torch.randn(64, K, generator=g) * K ** -0.5).to(torch.bfloat16).to(DEV)
                b = (torch.randn(64, generator=g) * 0.1).to(DEV)
                o = torch.zeros(f, *oshape, dtype=torch.bfloat16, device=DEV)
                nets.append((x, f, w, b, o))
            (x1, _, w1, b1, o1), (x2, _, w2, b2, o2) = nets
            fns = {}
            for v in (0, 1):
                fns["halo v%d" % v] = (lambda v=v: H.conv_halo_fwd(layer, x1, f1, w1, b1, o1, x2, f2, w2, b2, o2, variant=v))
            if f2:
                def tiled():
                    H.native.hip().drl_conv_fwd_dual(layer, H._p(x1), f1, H._p(w1), H._p(b1), H._p(o1), H._p(x2), f2, H._p(w2),
                                                     H._p(b2), H._p(o2), H._s())
            else:
                def tiled():
                    H.conv_fwd(layer, x1, w1, b1, o1, f1, tiled=H.CONV_FWD_TILED_VARIANT[layer])
            fns["tiled"] = tiled
            t = timeit(fns)
            flop = 2.0 * (f1 + f2) * oshape[0] * oshape[1] * 64 * K
            row = {"layer": layer, "frames": [f1, f2], "us": t, "tflops": {k: round(flop / v / 1e6, 1) for k, v in t.items()}}
            print(json.dumps(row), flush=True)
            if out:
                out.write(json.dumps(row) + "\n")
    dgrad(out)
    wgrad(out)


def wgrad1(out):
    g = torch.Generator().manual_seed(3)
    for f in (512, 2048, 3072, 5120):
        store = torch.randint(0, 256, (f + 3, 7056), dtype=torch.uint8, generator=g).to(DEV)
        off = (torch.randperm(f, generator=g) * 7056).to(DEV)                          # overlapping windows, sampled order
        dy = (torch.randn(f, 20, 20, 32, generator=g) * 0.1).to(torch.bfloat16).to(DEV)
        dw, db, ws = torch.zeros(32, 256, device=DEV), torch.zeros(32, device=DEV), H.conv_wgrad_ws(1, DEV)
        sp = max(16, 2 * max(8, 128 * f // 5120), f // 10)
        fns = {"halo": lambda: H.conv1_wgrad_halo(store, off, dy, dw, db, f, ws),
               "halo 128 wgs": lambda: H.conv1_wgrad_halo(store, off, dy, dw, db, f, ws, max_wgs=128),
               "split-K implicit gemm": lambda: H.conv_bwd(1, store, dy, None, dw, db, None, f, frame_off=off, splits=sp)}
        t = timeit(fns)
        flop = 2.0 * f * 400 * 32 * 256
        row = {"wgrad layer": 1, "frames": f, "us": t, "tflops": {k: round(flop / v / 1e6, 1) for k, v in t.items()}}
        print(json.dumps(row), flush=True)
        if out:
            out.write(json.dumps(row) + "\n")


def wgrad(out):
    wgrad1(out)
    g = torch.Generator().manual_seed(2)
    for layer in (2, 3):
        ishape, oshape, K = SH[layer]
        for f in (512, 2560, 5120):
            dy = (torch.randn(f, *oshape, generator=g) * 0.1).to(torch.bfloat16).to(DEV)
            act = torch.relu(torch.randn(f, *ishape, generator=g)).to(torch.bfloat16).to(DEV)
            dw, db = torch.zeros(64, K, device=DEV), torch.zeros(64, device=DEV)
            ws = H.conv_wgrad_ws(layer, DEV)
            sp = max(8, 128 * f // 5120)
            fns = {"halo": lambda: H.conv_wgrad_halo(layer, act, dy, dw, db, f, ws),
                   "halo 128 wgs": lambda: H.conv_wgrad_halo(layer, act, dy, dw, db, f, ws, max_wgs=128),
                   "split-K implicit gemm": lambda: H.conv_bwd(layer, act, dy, None, dw, db, None, f, splits=sp)}
            t = timeit(fns)
            flop = 2.0 * f * oshape[0] * oshape[1] * 64 * K
            row = {"wgrad layer": layer, "frames": f, "us": t, "tflops": {k: round(flop / v / 1e6, 1) for k, v in t.items()}}
            print(json.dumps(row), flush=True)
            if out:
                out.write(json.dumps(row) + "\n")


def dgrad(out):
    g = torch.Generator().manual_seed(1)
    wt2 = torch.zeros(4, 32, 2, 2, 64, dtype=torch.bfloat16, device=DEV)
    wt3 = torch.zeros(64, 3, 3, 64, dtype=torch.bfloat16, device=DEV)
    w2 = (torch.randn(64, 512, generator=g) * 512 ** -0.5).to(torch.bfloat16).to(DEV)
    w3 = (torch.randn(64, 576, generator=g) * 576 ** -0.5).to(torch.bfloat16).to(DEV)
    H.conv_halo_wprep(w2, w3, wt2, wt3)
    for layer in (2, 3):
        ishape, oshape, K = SH[layer]
        for f in (512, 2048, 3072):
            dy = (torch.randn(f, *oshape, generator=g) * 0.1).to(torch.bfloat16).to(DEV)
            act = torch.relu(torch.randn(f, *ishape, generator=g)).to(torch.bfloat16).to(DEV)
            dx = torch.zeros_like(act)
            w, wt = (w2, wt2) if layer == 2 else (w3, wt3)
            fns = {"halo": lambda: H.conv_halo_dgrad(layer, dy, wt, act, dx, f),
                   "implicit gemm": lambda: H.conv_dgrad(layer, dy, w, act, dx, f),
                   "wprep": lambda: H.conv_halo_wprep(w2, w3, wt2, wt3)}
            t = timeit(fns)
            flop = 2.0 * f * oshape[0] * oshape[1] * 64 * K
            row = {"dgrad layer": layer, "frames": f, "us": t, "tflops": {k: round(flop / v / 1e6, 1) for k, v in t.items() if k != "wprep"}}
            print(json.dumps(row), flush=True)
            if out:
                out.write(json.dumps(row) + "\n")


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""Large-tile GEMM (csrc/hip/gemm_big.hip) against hipBLASLt (torch.mm) and the 4-wave kernels of gemm.hip on the big dense
products of the R2D2 step, random data, interleaved rounds in one process (median / min), correctness vs an fp32 product."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from distributed_rl_amd.ops import hip as H  # noqa: E402

dev = torch.device("cuda")
torch.manual_seed(0)



def lib_gemm(mode, a, b, c):
    """The vendor library's product (hipBLASLt through torch.mm, fp32 accumulation) - the yardstick of this script only; the
    engines run gemm_big.hip."""
    lhs = a.t() if mode == H.TN else a
    rhs = b.t() if mode == H.NT else b
    if c.dtype == torch.float32:
        torch.mm(lhs, rhs, out_dtype=torch.float32, out=c)
    else:
        torch.mm(lhs, rhs, out=c)


def rnd(*shape):
    return (torch.rand(*shape, device=dev) * 2 - 1).to(torch.bfloat16)


def timeit(fns, rounds=12, inner=6):
    for f in fns.values():
        for _ in range(2):
            f()
    torch.cuda.synchronize()
    res = {k: [] for k in fns}
    for _ in range(rounds):
        for k, f in fns.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(inner):
                f()
            e1.record()
            e1.synchronize()
            res[k].append(e0.elapsed_time(e1) * 1e3 / inner)
    return {k: (sorted(v)[len(v) // 2], min(v)) for k, v in res.items()}


def case(mode, M, N, K, tiles, epi):
    if mode == H.NT:
        a, b = rnd(M, K), rnd(N, K)
        ref = a.float() @ b.float().t()
    elif mode == H.NN:
        a, b = rnd(M, K), rnd(K, N)
        ref = a.float() @ b.float()
    else:
        a, b = rnd(K, M), rnd(K, N)
        ref = a.float().t() @ b.float()
    bias = torch.randn(N, device=dev) if epi == "bias" else None
    mask = rnd(M, N) if epi == "mask" else None
    out_dtype = torch.bfloat16 if epi == "mask" else torch.float32
    want = ref + bias if bias is not None else ref
    if mask is not None:
        want = want * (mask.float() > 0)
    c = torch.zeros(M, N, dtype=out_dtype, device=dev)
    errs = {}
    for t in tiles:
        c.zero_()
        H.gemm_big(mode, a, b, c, bias=bias, mask=mask, tile=t)
        torch.cuda.synchronize()
        errs[t] = float((c.float() - want).norm() / want.norm())
    c_lib = torch.zeros(M, N, dtype=out_dtype, device=dev)

    def lib():
        lib_gemm(mode, a, b, c_lib)
        if bias is not None:
            c_lib.add_(bias)
        if mask is not None:
            c_lib.mul_((mask > 0).to(c_lib.dtype))

    def lib_plain():
        lib_gemm(mode, a, b, c_lib)

    def old():
        flags = (H.EPI_OUT_F32 if out_dtype == torch.float32 else 0) | (H.EPI_MASK if mask is not None else 0)
        H.gemm(mode, a, b, c, M, N, K, bias=bias, mask=mask, flags=flags, big=(N % 128 == 0 and mode != H.TN))
    fns = {"lib+epilogue passes": lib, "lib (plain product)": lib_plain, "gemm.hip": old}
    for t in tiles:
        fns["big tile %d" % t] = (lambda t=t: H.gemm_big(mode, a, b, c, bias=bias, mask=mask, tile=t))
    tm = timeit(fns)
    flop = 2.0 * M * N * K
    row = {"mode": {H.NT: "NT", H.NN: "NN", H.TN: "TN"}[mode], "M": M, "N": N, "K": K, "epilogue": epi, "rel_err": errs,
           "auto_tile": H.big_tile_for(M, N),
           "us_median_min": {k: (round(v[0], 1), round(v[1], 1)) for k, v in tm.items()},
           "tflops_median": {k: round(flop / v[0] / 1e6, 0) for k, v in tm.items()}}
    print(json.dumps(row), flush=True)
    return row


if __name__ == "__main__":
    tiles = [int(x) for x in os.environ.get("TILES", "0,1,2,3").split(",")]
    rows = []
    for M in (7680, 4608, 3072):
        rows.append(case(H.NT, M, 2048, 3136, tiles, "bias"))
    for M in (5120, 3072, 2048):
        rows.append(case(H.NN, M, 3136, 2048, tiles, "mask"))
    rows.append(case(H.TN, 2048, 3136, 5120, tiles, "none"))
    rows.append(case(H.TN, 2048, 512, 5120, tiles, "none"))
    rows.append(case(H.NT, 7680, 1024, 512, tiles, "bias"))
    # two networks per launch (online + target input projection of one forward chunk)
    for M in (4096, 3584, 4608, 3072):
        a, a2, b, b2 = rnd(M, 3136), rnd(M, 3136), rnd(2048, 3136), rnd(2048, 3136)
        bias, bias2 = torch.randn(2048, device=dev), torch.randn(2048, device=dev)
        c, c2 = torch.zeros(M, 2048, device=dev), torch.zeros(M, 2048, device=dev)
        H.gemm_big(H.NT, a, b, c, bias=bias, second=(a2, b2, c2, bias2, None))
        torch.cuda.synchronize()
        e = [float((x - (p.float() @ q.float().t() + bb)).norm() / x.norm()) for x, p, q, bb in ((c, a, b, bias), (c2, a2, b2, bias2))]

        def lib2():
            for p, q, x, bb in ((a, b, c, bias), (a2, b2, c2, bias2)):
                lib_gemm(H.NT, p, q, x)
                x.add_(bb)
        fns = {"lib x2 + bias passes": lib2}
        for t in tiles:
            fns["big dual tile %d" % t] = (lambda t=t: H.gemm_big(H.NT, a, b, c, bias=bias, second=(a2, b2, c2, bias2, None), tile=t))
        tm = timeit(fns)
        print(json.dumps({"mode": "NT x2", "M": M, "rel_err": e, "auto_tile": H.big_tile_for(M, 2048, 2),
                          "us_median_min": {k: (round(v[0], 1), round(v[1], 1)) for k, v in tm.items()}}), flush=True)

#!/usr/bin/env python
"""Per-kernel micro-benchmark at the Ape-X (B=512) / R2D2 (T*B=7680) shapes: HIP-event timing of every hot
kernel in isolation, with achieved TFLOP/s and GB/s. Use it under rocprofv3 --pmc to collect counters for one
kernel family:   rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU ... -- python scripts/kbench.py --only conv1_fwd
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_rl_amd.ops import hip as H  # noqa: E402

DEV = "cuda"


def bf(*shape, scale=1.0):
    return (torch.randn(*shape, device=DEV) * scale).to(torch.bfloat16)


EAGER = False


def timeit(fn, reps=20, warm=3):
    """`reps` back-to-back launches captured in one hipGraph (no host launch overhead in the number)."""
    if EAGER:                      # plain launches (for rocprofv3 --pmc runs: counters are attributed per dispatch)
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(reps):
            fn()
        e.record()
        torch.cuda.synchronize()
        return s.elapsed_time(e) * 1e3 / reps
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(warm):
            fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    g.replay()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / reps     # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", type=str, default="")
    ap.add_argument("--frames", type=int, default=1024)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--eager", action="store_true")
    ap.add_argument("--splits", type=int, default=32, help="split-K count of the fused conv backward cases")
    args = ap.parse_args()
    global EAGER
    EAGER = args.eager
    F = args.frames
    B = F // 2
    cases = {}
    x = torch.randint(0, 256, (F, 84, 84, 4), dtype=torch.uint8, device=DEV)
    w1, b1 = bf(32, 8, 8, 4, scale=0.05), torch.randn(32, device=DEV)
    w2, b2 = bf(64, 4, 4, 32, scale=0.05), torch.randn(64, device=DEV)
    w3, b3 = bf(64, 3, 3, 64, scale=0.05), torch.randn(64, device=DEV)
    a1 = torch.zeros(F, 20, 20, 32, dtype=torch.bfloat16, device=DEV)
    a2 = torch.zeros(F, 9, 9, 64, dtype=torch.bfloat16, device=DEV)
    a3 = torch.zeros(F, 3136, dtype=torch.bfloat16, device=DEV)
    wfc, bfc = bf(1024, 3136, scale=0.02), torch.randn(1024, device=DEV)
    h = torch.zeros(F, 1024, dtype=torch.bfloat16, device=DEV)
    H.conv_fwd(1, x, w1, b1, a1, F)
    H.conv_fwd(2, a1, w2, b2, a2, F)
    H.conv_fwd(3, a2, w3, b3, a3, F)
    cases["conv1_fwd_direct"] = (lambda: H.conv_fwd(1, x, w1, b1, a1, F), 2 * F * 400 * 32 * 256, F * (28224 + 25600))
    cases["conv1_fwd_tiled"] = (lambda: H.conv_fwd(1, x, w1, b1, a1, F, tiled=0), 2 * F * 400 * 32 * 256, F * (28224 + 25600))
    cases["conv2_fwd_halo"] = (lambda: H.conv_fwd(2, a1, w2, b2, a2, F), 2 * F * 81 * 64 * 512, F * (25600 + 10368))
    cases["conv3_fwd_halo"] = (lambda: H.conv_fwd(3, a2, w3, b3, a3, F), 2 * F * 49 * 64 * 576, F * (10368 + 6272))
    for v in (0, 1, 2, 3):                            # LDS-tiled implicit GEMM: 0 = register ring, 1..3 = LDS-DMA mainloops
        cases["conv2_fwd_tiled%d" % v] = (lambda v=v: H.conv_fwd(2, a1, w2, b2, a2, F, tiled=v), 2 * F * 81 * 64 * 512, F * (25600 + 10368))
        cases["conv3_fwd_tiled%d" % v] = (lambda v=v: H.conv_fwd(3, a2, w3, b3, a3, F, tiled=v), 2 * F * 49 * 64 * 576, F * (10368 + 6272))
    for v in (0, 2, 3, 5, 6, 10, 11, 13, 14):
        cases["fc1_fwd_v%d" % v] = (lambda v=v: H.gemm(H.NT, a3, wfc, h, F, 1024, 3136, bias=bfc, flags=H.EPI_RELU, big=v), 2 * F * 1024 * 3136, 2 * (F * 3136 + 1024 * 3136 + F * 1024))
    # two networks per launch (Ape-X learner: online net on F frames + target net on F/2 frames)
    xoff = (torch.arange(F, device=DEV) * 28224).long()
    t1, t2 = torch.zeros_like(a1[:B]), torch.zeros_like(a2[:B])
    t3, th = torch.zeros_like(a3[:B]), torch.zeros_like(h[:B])
    q1, q2 = torch.zeros(F, 4, device=DEV), torch.zeros(B, 4, device=DEV)
    wa, wv, ba, bv = bf(4, 512, scale=0.05), bf(1, 512, scale=0.05), torch.randn(4, device=DEV), torch.randn(1, device=DEV)
    FB = F + B
    cases["conv1_fwd_dual"] = (lambda: H.conv_fwd_dual(1, x, F, w1, b1, a1, None, B, w1, b1, t1, frame_off1=xoff, frame_off2=xoff[B:]),
                               2 * FB * 400 * 32 * 256, FB * (28224 + 25600))
    cases["conv2_fwd_dual"] = (lambda: H.conv_fwd_dual(2, a1, F, w2, b2, a2, a1[:B], B, w2, b2, t2), 2 * FB * 81 * 64 * 512, FB * (25600 + 10368))
    cases["conv3_fwd_dual"] = (lambda: H.conv_fwd_dual(3, a2, F, w3, b3, a3, a2[:B], B, w3, b3, t3), 2 * FB * 49 * 64 * 576, FB * (10368 + 6272))
    cases["fc1_fwd_dual"] = (lambda: H.gemm_nt_dual(a3, wfc, bfc, h, a3[:B], wfc, bfc, th, 1024, 3136, flags=H.EPI_RELU),
                             2 * FB * 1024 * 3136, 2 * (FB * 3136 + 2 * 1024 * 3136 + FB * 1024))
    cases["head_fwd_dual"] = (lambda: H.head_fwd_dual(h, wa, ba, wv, bv, q1, th, wa, ba, wv, bv, q2), 2 * FB * 5 * 512, FB * 2048)
    hf = torch.zeros(F, 1024, device=DEV)
    for sp in (2, 4, 8):
        cases["fc1_fwd_v1_sk%d" % sp] = (lambda sp=sp: H.gemm(H.NT, a3, wfc, hf, F, 1024, 3136, flags=H.EPI_OUT_F32 | H.EPI_ATOMIC, splits=sp, big=1), 2 * F * 1024 * 3136, 0)
        cases["fc1_fwd_v4_sk%d" % sp] = (lambda sp=sp: H.gemm(H.NT, a3, wfc, hf, F, 1024, 3136, flags=H.EPI_OUT_F32 | H.EPI_ATOMIC, splits=sp, big=4), 2 * F * 1024 * 3136, 0)
    dh = bf(B, 1024, scale=0.01)
    gfc = torch.zeros(1024, 3136, device=DEV)
    d3 = torch.zeros(B, 3136, dtype=torch.bfloat16, device=DEV)
    for v in (0, 2, 5, 6, 10, 11, 13):
        cases["fc1_dw_tn_v%d" % v] = (lambda v=v: H.gemm(H.TN, dh, a3[:B], gfc, 1024, 3136, B, flags=H.EPI_OUT_F32, big=v), 2 * B * 1024 * 3136, 4 * 1024 * 3136)
        cases["fc1_dx_nn_v%d" % v] = (lambda v=v: H.gemm(H.NN, dh, wfc, d3, B, 3136, 1024, mask=a3[:B], flags=H.EPI_MASK, big=v), 2 * B * 1024 * 3136, 2 * 1024 * 3136)
    d3v = bf(B * 49, 64, scale=0.01)
    d2v = bf(B * 81, 64, scale=0.01)
    d1v = bf(B * 400, 32, scale=0.01)
    g3, g2, g1 = torch.zeros(64, 576, device=DEV), torch.zeros(64, 512, device=DEV), torch.zeros(32, 256, device=DEV)
    for sp in (16, 32, 64):
        cases["conv3_wgrad_s%d" % sp] = (lambda sp=sp: H.conv_wgrad(3, a2[:B], d3v, g3, B, splits=sp), 2 * B * 49 * 64 * 576, B * 10368)
        cases["conv2_wgrad_s%d" % sp] = (lambda sp=sp: H.conv_wgrad(2, a1[:B], d2v, g2, B, splits=sp), 2 * B * 81 * 64 * 512, B * 25600)
        cases["conv1_wgrad_s%d" % sp] = (lambda sp=sp: H.conv_wgrad(1, x[:B], d1v, g1, B, splits=2 * sp), 2 * B * 400 * 32 * 256, B * 28224)
    dx2 = torch.zeros(B, 9, 9, 64, dtype=torch.bfloat16, device=DEV)
    dx1 = torch.zeros(B, 20, 20, 32, dtype=torch.bfloat16, device=DEV)
    foff = (torch.arange(B, device=DEV, dtype=torch.int64) * 28224)
    for v in (0, 1, 2):
        for sp in (64, 128, 256):
            cases["conv1_wgrad_off_v%d_s%d" % (v, sp)] = (lambda v=v, sp=sp: H.conv_wgrad(1, x, d1v, g1, B, frame_off=foff, splits=sp | (v << 16)), 2 * B * 400 * 32 * 256, B * 28224)
    cases["conv3_dgrad_impl"] = (lambda: H.conv_dgrad(3, d3v, w3, a2[:B], dx2, B), 2 * B * 81 * 64 * 576, 2 * B * (49 + 81) * 64)
    cases["conv2_dgrad_impl"] = (lambda: H.conv_dgrad(2, d2v, w2, a1[:B], dx1, B), 2 * B * 400 * 32 * 256, 2 * B * (81 * 64 + 400 * 32))
    db = torch.zeros(64, device=DEV)
    cases["colsum_d2"] = (lambda: H.colsum(d2v, db, B * 81, 64), 0, 2 * B * 81 * 64)
    cases["colsum_d1"] = (lambda: H.colsum(d1v, db, B * 400, 32), 0, 2 * B * 400 * 32)
    dbh = torch.zeros(1024, device=DEV)
    cases["colsum_dh"] = (lambda: H.colsum(dh, dbh, B, 1024), 0, 2 * B * 1024)
    # horizontally fused backward launches (dX + dW + db in one launch) and their body variants
    dq = torch.randn(B, 4, device=DEV) * 0.01
    gwa, gba, gwv, gbv = torch.zeros(4, 512, device=DEV), torch.zeros(4, device=DEV), torch.zeros(1, 512, device=DEV), torch.zeros(1, device=DEV)
    hq = bf(B, 1024)
    for v in range(6):
        cases["fc1_bwd_fused_v%d" % v] = (lambda v=v: H.fc_bwd(dh, a3[:B], wfc, gfc, dbh, d3, B, 1024, 3136, head=(dq, hq, gwa, gba, gwv, gbv), variant=v), 4 * B * 1024 * 3136, 0)
    SP = args.splits
    for v in range(6):
        cases["conv3_bwd_fused_v%d" % v] = (lambda v=v: H.conv_bwd(3, a2[:B], d3v, w3, g3, db, dx2, B, splits=SP, variant=v), 2 * B * (49 + 81) * 64 * 576, 0)
        cases["conv2_bwd_fused_v%d" % v] = (lambda v=v: H.conv_bwd(2, a1[:B], d2v, w2, g2, db, dx1, B, splits=SP, variant=v), 2 * B * 81 * 64 * 512 + 2 * B * 400 * 32 * 256, 0)
    db1 = torch.zeros(32, device=DEV)
    cases["conv1_bwd_fused"] = (lambda: H.conv_bwd(1, x, d1v, None, g1, db1, None, B, frame_off=foff, splits=4 * SP), 2 * B * 400 * 32 * 256, 0)
    # R2D2-sized GEMMs
    N = 7680
    feat = bf(N, 3136)
    wih = bf(2048, 3136, scale=0.02)
    xg = torch.zeros(N, 2048, device=DEV)
    for v in (1, 4, 12, 13):
        cases["lstm_igemm_v%d" % v] = (lambda v=v: H.gemm(H.NT, feat, wih, xg, N, 2048, 3136, flags=H.EPI_OUT_F32, big=v), 2 * N * 2048 * 3136, 0)
    NBk = 5120                                        # backward window rows (80 steps x 64 sequences)
    dGk_all = bf(NBk, 2048, scale=0.01)
    gih = torch.zeros(2048, 3136, device=DEV)
    dx3 = torch.zeros(NBk, 3136, dtype=torch.bfloat16, device=DEV)
    for v in (0, 2, 4):                               # N = 3136 = 49 x 64: only 64-wide tiles divide it
        cases["r2_dwih_tn_v%d" % v] = (lambda v=v: H.gemm(H.TN, dGk_all, feat[:NBk], gih, 2048, 3136, NBk, flags=H.EPI_OUT_F32, big=v), 2 * NBk * 2048 * 3136, 0)
        cases["r2_dx_nn_v%d" % v] = (lambda v=v: H.gemm(H.NN, dGk_all, wih, dx3, NBk, 3136, 2048, mask=feat[:NBk], flags=H.EPI_MASK, big=v), 2 * NBk * 2048 * 3136, 0)
    hs_all = bf(NBk, 512)
    ghh = torch.zeros(2048, 512, device=DEV)
    for v in (0, 1, 2, 4):
        cases["r2_dwhh_tn_v%d" % v] = (lambda v=v: H.gemm(H.TN, dGk_all, hs_all, ghh, 2048, 512, NBk, flags=H.EPI_OUT_F32, big=v), 2 * NBk * 2048 * 512, 0)
    for v in (0, 4):
        for sp in (2, 4):
            cases["r2_dwih_tn_v%d_sk%d" % (v, sp)] = (lambda v=v, sp=sp: H.gemm(H.TN, dGk_all, feat[:NBk], gih, 2048, 3136, NBk, flags=H.EPI_OUT_F32 | H.EPI_ATOMIC, splits=sp, big=v), 2 * NBk * 2048 * 3136, 0)
    hp, whh = bf(64, 512), bf(2048, 512, scale=0.05)
    xg_t, c0 = torch.randn(64, 2048, device=DEV), torch.zeros(64, 512, device=DEV)
    ho, co = torch.zeros(64, 512, dtype=torch.bfloat16, device=DEV), torch.zeros(64, 512, device=DEV)
    cases["lstm_step"] = (lambda: H.lstm_step(hp, whh, xg_t, c0, ho, co, None), 2 * 64 * 2048 * 512, 0)
    dGk = bf(64, 2048, scale=0.01)
    dhr = torch.zeros(64, 512, device=DEV)
    for v in (0, 2, 5):
        for sp in (8, 16, 32):
            cases["bptt_nn_v%d_s%d" % (v, sp)] = (lambda v=v, sp=sp: H.gemm(H.NN, dGk, whh, dhr, 64, 512, 2048, flags=H.EPI_OUT_F32 | H.EPI_ATOMIC, splits=sp, big=v), 2 * 64 * 512 * 2048, 0)
    Ts = 120
    xgs = torch.randn(Ts, 64, 2048, device=DEV) * 0.3
    hss = torch.zeros(Ts + 1, 64, 512, dtype=torch.bfloat16, device=DEV)
    css = torch.zeros(Ts + 1, 64, 512, device=DEV)
    gts = torch.zeros(Ts, 64, 2048, dtype=torch.bfloat16, device=DEV)
    hss2, css2 = hss.clone(), css.clone()
    err = torch.zeros(1, dtype=torch.int32, device=DEV)

    def steps120():
        for t in range(Ts):
            H.lstm_step(hss[t], whh, xgs[t], css[t], hss[t + 1], css[t + 1], gts[t])
    cases["lstm_120_steps"] = (steps120, 2 * 64 * 2048 * 512 * Ts, 0)
    hx1, hx2 = H.lstm_hx(DEV), H.lstm_hx(DEV)
    cases["lstm_seq_x"] = (lambda: H.lstm_seq_fwd_x([(xgs, whh, hss, css, gts, hx1)], Ts, 64, 512, err), 2 * 64 * 2048 * 512 * Ts, 0)
    cases["lstm_seq_x_2net"] = (lambda: H.lstm_seq_fwd_x([(xgs, whh, hss, css, gts, hx1), (xgs, whh, hss2, css2, None, hx2)], Ts, 64,
                                                            512, err), 2 * 2 * 64 * 2048 * 512 * Ts, 0)
    # BPTT: 80 steps as the engine issues them - two launches per step vs the fused one-launch step
    Tb = 80
    dhs_b = bf(Tb, 64, 512, scale=0.05)
    gts_b = (torch.rand(Tb, 64, 2048, device=DEV) * 0.9 + 0.05).to(torch.bfloat16)
    css_b = torch.randn(Tb + 1, 64, 512, device=DEV) * 0.5
    dG_b = torch.zeros(Tb, 64, 2048, dtype=torch.bfloat16, device=DEV)
    rec1, dc1 = torch.zeros(64, 512, device=DEV), torch.zeros(64, 512, device=DEV)
    R3, DC2 = torch.zeros(3, 64, 512, device=DEV), torch.zeros(2, 64, 512, device=DEV)

    def bptt_two():
        for s_ in range(Tb):
            t = Tb - 1 - s_
            H.lstm_bwd_point(dhs_b[t], rec1, dc1, gts_b[t], css_b[t + 1], css_b[t], dG_b[t])
            H.gemm(H.NN, dG_b[t], whh, rec1, 64, 512, 2048, flags=H.EPI_OUT_F32 | H.EPI_ATOMIC, splits=8, big=5)

    def bptt_fused():
        for s_ in range(Tb):
            t = Tb - 1 - s_
            H.lstm_bwd_step(dhs_b[t], R3[s_ % 3], R3[(s_ + 1) % 3], R3[(s_ + 2) % 3], DC2[s_ % 2], DC2[(s_ + 1) % 2], gts_b[t],
                            css_b[t + 1], css_b[t], dG_b[t], whh)
    bx_b, rec_b, dc_b = H.lstm_bx(DEV), torch.zeros(64, 512, device=DEV), torch.zeros(64, 512, device=DEV)
    cases["lstm_bptt80_persistent"] = (lambda: H.lstm_bptt(dhs_b, gts_b, css_b, dG_b, whh, rec_b, dc_b, bx_b, err, k_hi=Tb - 1, steps=Tb,
                                                           first=True, carry=False), 2 * 64 * 2048 * 512 * Tb, 0)

    def bptt_two_chunks():
        H.lstm_bptt(dhs_b, gts_b, css_b, dG_b, whh, rec_b, dc_b, bx_b, err, k_hi=Tb - 1, steps=Tb // 2, first=True, carry=True)
        H.lstm_bptt(dhs_b, gts_b, css_b, dG_b, whh, rec_b, dc_b, bx_b, err, k_hi=Tb - 1 - Tb // 2, steps=Tb - Tb // 2, first=False,
                    carry=False)
    cases["lstm_bptt80_persistent_2chunks"] = (bptt_two_chunks, 2 * 64 * 2048 * 512 * Tb, 0)
    cases["lstm_bptt80_two_launches"] = (bptt_two, 2 * 64 * 2048 * 512 * Tb, 0)
    cases["lstm_bptt80_fused_step"] = (bptt_fused, 2 * 64 * 2048 * 512 * Tb, 0)
    print("%-18s %10s %10s %10s" % ("kernel", "us", "TFLOP/s", "GB/s"))
    for name, (fn, flops, byts) in cases.items():
        if args.only and not any(o in name for o in args.only.split(",")):
            continue
        us = timeit(fn, 3 if name.startswith("lstm_1") or name.startswith("lstm_seq") or name.startswith("lstm_bptt") else args.reps)
        print("%-18s %10.1f %10.1f %10.1f" % (name, us, flops / us / 1e6, byts / us / 1e3))


if __name__ == "__main__":
    main()

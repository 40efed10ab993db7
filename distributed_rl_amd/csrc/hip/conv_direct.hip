// Direct (barrier-free) forward convolution conv1 of the Nature-CNN torso on gfx950 (forward).
//
// One MFMA k-step (32 contraction elements) of conv1 is a CONTIGUOUS run of the input: one patch row = 8 px x 4 ch u8 =
// 32 bytes of an HWC stack, or 8 px of one plane x 4 patch rows of a plane store; the 8 consecutive k a lane needs for
// its 16x16x32 operand fragment are one 8-byte global load. So the activation operand goes global -> VGPR directly in
// MFMA fragment layout: no LDS staging, no __syncthreads, no shared K-loop between waves. Every wave walks its own run
// of 16-row output tiles with the next tile's loads in flight, and all 32x256 weights live in 64 VGPRs per lane for the
// life of the wave. Operands are passed swapped (acc = mfma(W, X)) so a lane ends up with consecutive output channels
// of one output pixel: bias + ReLU + v_cvt_pk_bf16_f32 and a 16-byte store.
// (conv2 / conv3 had the same form with the weights in LDS; 21.3 / 16.5 us at 1024 frames against 17.0 / 13.4 us for the
// LDS-tiled implicit GEMM, and the LDS-halo kernels of conv_halo.hip beat both: it is gone.)
#include "common.h"
#include "conv_frag.h"

typedef __attribute__((ext_vector_type(8))) __bf16 mfma_bf16x8;

__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(mfma_bf16x8, a), __builtin_bit_cast(mfma_bf16x8, b), c, 0, 0, 0);
}

// 8 u8 -> 8 bf16 (exact): v_cvt_f32_ubyteN, then one v_perm_b32 per pair keeps the two high halves
__device__ __forceinline__ bf16x8 u8x8_to_bf16(uint2 v) {
    uint32_t o[4];
    const uint32_t w[2] = {v.x, v.y};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t s = w[i >> 1];
        const float f0 = (i & 1) ? (float)((s >> 16) & 0xffu) : (float)(s & 0xffu);     // -> v_cvt_f32_ubyte{2,0}
        const float f1 = (i & 1) ? (float)(s >> 24) : (float)((s >> 8) & 0xffu);          // -> v_cvt_f32_ubyte{3,1}
        o[i] = __builtin_amdgcn_perm(__float_as_uint(f1), __float_as_uint(f0), 0x07060302u);
    }
    return __builtin_bit_cast(bf16x8, make_uint4(o[0], o[1], o[2], o[3]));
}

// conv1 in f16 instead of bf16: the kernel is vector-issue bound (59 % of its wave cycles wait for an
// instruction slot) and most of its vector work is the u8 -> bf16 conversion: 12 instructions per 8 pixels. In f16 the
// conversion is ONE v_perm_b32 per 2 pixels: the half 0x6400 | b is exactly 1024 + b (ulp 1 at 2^10), so a byte next to a
// constant 0x64 byte already is a usable operand. The accumulators start at -1024 * sum_k w[n][k], which cancels the
// offset exactly up to fp32 rounding at |1024 * sum w| ~ 1e3 (1e-4 pixel units, 5e-7 after the /255). The weights are
// converted bf16 -> f16 once per wave (exact: |w| < 65504, f16 has more mantissa bits; |w| < 6e-5 goes denormal).
static int g_conv1_waves = 8;   // wave slots the tiles are dealt to (per CU): fewer waves = fewer weight set-ups; drl_conv1_set_waves
// ---------------------------------------------------------------------------------------------------------------
// conv1: u8 frames -> bf16 [n,20,20,32]. Two input forms:
//   dense (OFF = false):  [n,84,84,4] HWC stacks as the environments deliver them (actors); weights [32][kh][kw][c].
//   store (OFF = true):   a replay store of single 84x84 PLANES (each frame kept once: replay/device_per.py); frame b =
//                         the 4 consecutive planes starting at byte offset frame_off[b]; weights [32][c][kh][kw].
// In both forms the contraction index is k = 32 s + 8 g + j with j = 8 consecutive bytes of the input: dense s = kh, (g, j)
// = the 8 px x 4 ch of a patch row; store s = 2 c + kh / 4, g = kh % 4, j = kw. The weight fragment address (row * 256 +
// 32 s + 8 g) is the same expression for both orders; only the input address differs.
// ---------------------------------------------------------------------------------------------------------------
// A second network (its own frame offsets / weights / bias / output / row count, same frame store x) may ride along in
// the same launch: blocks [nblk1, gridDim.x) work on it (Ape-X learner: online net on [s ; s'] + target net on s' as ONE
// launch instead of two launches on two streams).
struct Conv1Net {
    const long long* frame_off;
    const bf16_t* w;
    const float* bias;
    bf16_t* out;
    int M, per;
};

// PAIR = 1: every block also multiplies its tiles with a second weight set, n3 (weights / bias / output only; the frames
// are those of n1): a tile is loaded and converted once for both networks. R2D2 learner: online + target over one
// sequence batch. (A form for the Ape-X learner - online on s, online AND target on s' - measured no better than the
// two-network launch below and is gone.)
template <bool OFF, int PAIR = 0>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void k_conv1_direct(const uint8_t* __restrict__ x, Conv1Net n1, Conv1Net n2, Conv1Net n3, int nblk1) {
    const bool second = PAIR != 1 && (int)blockIdx.x >= nblk1;         // block-uniform
    constexpr bool pair_blk = PAIR == 1;
    const long long* __restrict__ frame_off = second ? n2.frame_off : n1.frame_off;
    const bf16_t* __restrict__ w = second ? n2.w : n1.w;
    const float* __restrict__ bias = second ? n2.bias : n1.bias;
    bf16_t* __restrict__ out = second ? n2.out : n1.out;
    const int M = second ? n2.M : n1.M, per = second ? n2.per : n1.per;
    const int blk = second ? (int)blockIdx.x - nblk1 : (int)blockIdx.x;
    const int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
    const int ntiles = (M + 15) >> 4;
    // every wave owns `per` CONSECUTIVE 16-row tiles (neighbouring output pixels overlap in the input: L1/L2 hits)
    const int t0 = (blk * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6)) * per, t1 = min(ntiles, t0 + per);   // SGPRs
    if (t0 >= ntiles) return;                       // wave-uniform: no barriers in this kernel
    uint4 wf[2][8];
    float acc0[2][4];                               // accumulator start: -1024 * (sum over k of this lane's channels' weights)
    {
        float csum[2] = {0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int s = 0; s < 8; ++s)
                wf[t][s] = bf16x8_to_f16(*reinterpret_cast<const bf16x8*>(w + ((r >> 2) * 8 + t * 4 + (r & 3)) * 256 + s * 32 + g * 8),
                                         csum[t]);
#pragma unroll
        for (int t = 0; t < 2; ++t) {               // lanes r, r+16, r+32, r+48 hold the four k-quarters of MFMA row r
            csum[t] += __shfl_xor(csum[t], 16, WAVE);
            csum[t] += __shfl_xor(csum[t], 32, WAVE);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc0[t][i] = -1024.0f * __shfl(csum[t], g * 4 + i, WAVE);   // accumulator row g*4+i
        }
    }
    uint4 wf2[PAIR ? 2 : 1][PAIR ? 8 : 1];
    float acc0b[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    if (pair_blk) {
        float csum[2] = {0.f, 0.f};
#pragma unroll
        for (int t = 0; t < (PAIR ? 2 : 1); ++t)
#pragma unroll
            for (int s = 0; s < (PAIR ? 8 : 1); ++s)
                wf2[t][s] = bf16x8_to_f16(*reinterpret_cast<const bf16x8*>(n3.w + ((r >> 2) * 8 + t * 4 + (r & 3)) * 256 + s * 32 + g * 8),
                                          csum[t]);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            csum[t] += __shfl_xor(csum[t], 16, WAVE);
            csum[t] += __shfl_xor(csum[t], 32, WAVE);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc0b[t][i] = -1024.0f * __shfl(csum[t], g * 4 + i, WAVE);
        }
    }
    // MFMA row i of weight tile t is output channel (i/4)*8 + t*4 + i%4 (a free permutation of the weight rows): the
    // lane that ends up with accumulator rows g*4..g*4+3 of both tiles then owns the 8 CONSECUTIVE channels g*8..g*8+7
    // of its pixel: one 16-byte store per lane, 1 KB contiguous per wave and tile.
    float4 b4[2], b4b[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        b4[t] = *reinterpret_cast<const float4*>(bias + g * 8 + t * 4);
        b4b[t] = pair_blk ? *reinterpret_cast<const float4*>(n3.bias + g * 8 + t * 4) : b4[t];
    }

    // Byte offset of this lane's first fragment (patch row 0) in tile `tile` = frame base + pixel offset; rows past M
    // re-read row M-1. The frame base of a replay-store frame is itself a (dependent) global load: it is issued two
    // tiles ahead and BEFORE the data loads of the next tile, so waiting for it (in-order vmcnt) never drains the
    // data prefetch.
    auto locate = [&](int tile, int& b) -> int {
        const int m = min(min(tile, ntiles - 1) * 16 + r, M - 1);
        b = m / 400;
        const int rem = m - b * 400, oy = rem / 20, ox = rem - oy * 20;
        if constexpr (OFF) return (4 * oy + g) * 84 + 4 * ox;          // plane store: row 4 oy + kh % 4 of plane c
        else return oy * (4 * 336) + ox * 16 + g * 8;
    };
    // byte offset of k-step s relative to locate(): dense = patch row s; store = plane s / 2, patch rows 4 (s % 2) ..
    auto koff = [](int s) -> int { return OFF ? (s >> 1) * 7056 + (s & 1) * 336 : s * 336; };
    auto frame_base = [&](int b) -> long long {
        if constexpr (OFF) return frame_off[b];
        else return (long long)b * 28224;
    };
    // THREE tile buffers: a wave computes tile t with the loads of t + 1 and t + 2 in flight (two buffers, one tile ahead:
    // 21.2 -> 20.8 us at 1536 dense frames, 45.2 -> 40.1 us at 3840 = 5.2 TB/s; the two-weight-set launches of the R2D2
    // step do not move: there 163 vector instructions + 30 MFMAs per tile keep the issue ports busy 55 % of the time,
    // profiles/conv1_pair_pmc_r2.txt).
    uint2 bufa[8], bufb[8], bufc[8];
    int b0, b1, b2_;
    const int p0 = locate(t0, b0), p1 = locate(t0 + 1, b1), p2_ = locate(t0 + 2, b2_);
    const long long f0 = frame_base(b0), f1 = frame_base(b1);
    long long fnext = frame_base(b2_);
#pragma unroll
    for (int s = 0; s < 8; ++s) bufa[s] = *reinterpret_cast<const uint2*>(x + f0 + p0 + koff(s));
#pragma unroll
    for (int s = 0; s < 8; ++s) bufb[s] = *reinterpret_cast<const uint2*>(x + f1 + p1 + koff(s));
    long long bnext = fnext + p2_;
    // one tile: prefetch tile+2 into `nxt`, compute `cur`. Called with the three buffers rotated (no register copies,
    // and the compiler cannot sink the prefetch below the compute because all buffers are live).
    auto body = [&](uint2 (&cur)[8], uint2 (&nxt)[8], int tile) __attribute__((always_inline)) {
        int b2;
        const int p2 = locate(tile + 3, b2);
        const long long f2 = frame_base(b2);                                                         // tile + 3 (first)
#pragma unroll
        for (int s = 0; s < 8; ++s) nxt[s] = *reinterpret_cast<const uint2*>(x + bnext + koff(s));    // tile + 2
        __builtin_amdgcn_sched_barrier(0);     // keep the prefetch ABOVE the compute (the scheduler sinks it otherwise)
        f32x4 acc[2] = {{acc0[0][0], acc0[0][1], acc0[0][2], acc0[0][3]}, {acc0[1][0], acc0[1][1], acc0[1][2], acc0[1][3]}};
        f32x4 accb[2] = {{acc0b[0][0], acc0b[0][1], acc0b[0][2], acc0b[0][3]}, {acc0b[1][0], acc0b[1][1], acc0b[1][2], acc0b[1][3]}};
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const uint4 xf = u8x8_to_f16_biased(cur[s]);
            acc[0] = mfma16h(wf[0][s], xf, acc[0]);
            acc[1] = mfma16h(wf[1][s], xf, acc[1]);
            if constexpr (PAIR) {
                if (pair_blk) {
                    accb[0] = mfma16h(wf2[0][s], xf, accb[0]);
                    accb[1] = mfma16h(wf2[1][s], xf, accb[1]);
                }
            }
        }
        const int m = tile * 16 + r;
        if (m < M) {
            constexpr float SC = 1.0f / 255.0f;          // pixel normalisation x/255 folded into the epilogue
            uint32_t o[4];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                o[2 * t] = pack_bf2(fmaxf(acc[t][0] * SC + b4[t].x, 0.f), fmaxf(acc[t][1] * SC + b4[t].y, 0.f));
                o[2 * t + 1] = pack_bf2(fmaxf(acc[t][2] * SC + b4[t].z, 0.f), fmaxf(acc[t][3] * SC + b4[t].w, 0.f));
            }
            *reinterpret_cast<uint4*>(out + (long long)m * 32 + g * 8) = make_uint4(o[0], o[1], o[2], o[3]);
            if (pair_blk) {
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    o[2 * t] = pack_bf2(fmaxf(accb[t][0] * SC + b4b[t].x, 0.f), fmaxf(accb[t][1] * SC + b4b[t].y, 0.f));
                    o[2 * t + 1] = pack_bf2(fmaxf(accb[t][2] * SC + b4b[t].z, 0.f), fmaxf(accb[t][3] * SC + b4b[t].w, 0.f));
                }
                *reinterpret_cast<uint4*>(n3.out + (long long)m * 32 + g * 8) = make_uint4(o[0], o[1], o[2], o[3]);
            }
        }
        bnext = f2 + p2;
    };
    int tile = t0;
    for (; tile + 2 < t1; tile += 3) {      // no skip path inside the loop: the back edge keeps exact vmcnt counts
        body(bufa, bufc, tile);
        body(bufb, bufa, tile + 1);
        body(bufc, bufb, tile + 2);
    }
    if (tile < t1) body(bufa, bufc, tile);          // tail of one or two tiles
    if (tile + 1 < t1) body(bufb, bufa, tile + 1);
}

// tiles per wave so that at most `slots` waves are needed; returns the grid in *grid
static int plan(int ntiles, int slots, int nwv, int* grid) {
    int per = (ntiles + slots - 1) / slots;
    if (per < 1) per = 1;
    *grid = (ntiles + per * nwv - 1) / (per * nwv);
    return per;
}

// conv1: x u8 HWC stacks, or a plane store + per-frame byte offsets. out bf16 NHWC, bias + ReLU fused.
DRL_API int drl_conv1_fwd(const void* x, const void* w, const float* bias, void* out, int frames, const long long* frame_off,
                          hipStream_t s) {
    if (frames <= 0) return -1;
    const int M = frames * 400;
    int grid;
    const int per = plan((M + 15) / 16, 256 * g_conv1_waves, 4, &grid);
    const Conv1Net n1{frame_off, (const bf16_t*)w, bias, (bf16_t*)out, M, per};
    if (frame_off) k_conv1_direct<true><<<grid, 256, 0, s>>>((const uint8_t*)x, n1, n1, n1, grid);
    else k_conv1_direct<false><<<grid, 256, 0, s>>>((const uint8_t*)x, n1, n1, n1, grid);
    return (int)hipGetLastError();
}

// conv1 of TWO networks over frames of one store in one launch (both addressed through byte offsets).
DRL_API int drl_conv1_fwd_dual(const void* x, const long long* frame_off1, int frames1, const void* w1, const float* bias1,
                               void* out1, const long long* frame_off2, int frames2, const void* w2, const float* bias2,
                               void* out2, hipStream_t s) {
    if (frames1 <= 0 || frames2 <= 0 || !frame_off1 || !frame_off2) return -1;
    int g1, g2;
    const int slots = 256 * g_conv1_waves;
    const int M1 = frames1 * 400, M2 = frames2 * 400;
    // split the resident wave slots in proportion to the work so both networks finish together
    const int s1 = (int)((long long)slots * M1 / (M1 + M2)), s2 = slots - s1;
    const int per1 = plan((M1 + 15) / 16, s1 > 0 ? s1 : 1, 4, &g1), per2 = plan((M2 + 15) / 16, s2 > 0 ? s2 : 1, 4, &g2);
    const Conv1Net n1{frame_off1, (const bf16_t*)w1, bias1, (bf16_t*)out1, M1, per1};
    const Conv1Net n2{frame_off2, (const bf16_t*)w2, bias2, (bf16_t*)out2, M2, per2};
    k_conv1_direct<true><<<g1 + g2, 256, 0, s>>>((const uint8_t*)x, n1, n2, n2, g1);
    return (int)hipGetLastError();
}

// conv1 of TWO networks over the SAME frames (one frame-offset list): every tile is read and converted once.
DRL_API int drl_conv1_fwd_pair(const void* x, const long long* frame_off, int frames, const void* w1, const float* bias1, void* out1,
                               const void* w2, const float* bias2, void* out2, hipStream_t s) {
    if (frames <= 0 || !frame_off) return -1;
    const int M = frames * 400;
    int grid;
    const int per = plan((M + 15) / 16, 256 * g_conv1_waves, 4, &grid);
    const Conv1Net n1{frame_off, (const bf16_t*)w1, bias1, (bf16_t*)out1, M, per};
    const Conv1Net n3{frame_off, (const bf16_t*)w2, bias2, (bf16_t*)out2, M, per};
    k_conv1_direct<true, 1><<<grid, 256, 0, s>>>((const uint8_t*)x, n1, n1, n3, grid);
    return (int)hipGetLastError();
}

DRL_API int drl_conv1_set_waves(int waves_per_cu) {
    if (waves_per_cu < 1 || waves_per_cu > 32) return -1;
    g_conv1_waves = waves_per_cu;
    return 0;
}

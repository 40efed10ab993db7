// Direct (barrier-free) forward convolutions for the Nature-CNN torso on gfx950 (forward).
//
// The K ordering (kh, kw, c) of the NHWC implicit GEMM has a property the generic LDS-tiled kernel in gemm.hip does not
// use: one MFMA k-step (32 contraction elements) is a CONTIGUOUS run of the input —
//     conv1: one patch row  = 8 px x 4 ch u8   = 32 bytes,
//     conv2: one patch pixel = 32 ch bf16       = 64 bytes,
//     conv3: half a patch pixel (64 ch bf16)    = 64 bytes,
// and the 8 consecutive k a lane needs for its 16x16x32 operand fragment are one 8-byte (u8) / 16-byte (bf16) global
// load. So the activation operand goes global -> VGPR directly in MFMA fragment layout: no LDS staging, no
// __syncthreads, no shared K-loop between waves. Every wave walks its own run of 16- or 32-row output tiles with the
// next tile's loads in flight, and the weights are resident for the life of the wave:
//     conv1: all 32x256 weights live in 64 VGPRs per lane;
//     conv2/3: 64x512 / 64x576 weights are copied once per workgroup into LDS in fragment order (conflict-free
//              ds_read_b128), then only read.
// Operands are passed swapped (acc = mfma(W, X)) so a lane ends up with 4 consecutive output channels of one output
// pixel: bias + ReLU + v_cvt_pk_bf16_f32 and an 8-byte store.
#include "common.h"

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

// conv1 in f16 instead of bf16 (DRL_CONV1_F16): the kernel is vector-issue bound (59 % of its wave cycles wait for an
// instruction slot) and most of its vector work is the u8 -> bf16 conversion: 12 instructions per 8 pixels. In f16 the
// conversion is ONE v_perm_b32 per 2 pixels: the half 0x6400 | b is exactly 1024 + b (ulp 1 at 2^10), so a byte next to a
// constant 0x64 byte already is a usable operand. The accumulators start at -1024 * sum_k w[n][k], which cancels the
// offset exactly up to fp32 rounding at |1024 * sum w| ~ 1e3 (1e-4 pixel units, 5e-7 after the /255). The weights are
// converted bf16 -> f16 once per wave (exact: |w| < 65504, f16 has more mantissa bits; |w| < 6e-5 goes denormal).
#ifndef DRL_CONV1_F16
#define DRL_CONV1_F16 1
#endif
#ifndef DRL_CONV1_WAVES_PER_CU
#define DRL_CONV1_WAVES_PER_CU 8     // wave slots the tiles are dealt to (per CU): fewer waves = fewer weight set-ups
#endif
typedef _Float16 mfma_f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f32x4 mfma16h(uint4 a, uint4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(mfma_f16x8, a), __builtin_bit_cast(mfma_f16x8, b), c, 0, 0, 0);
}
// 8 u8 -> 8 f16 holding 1024 + pixel: 4 x v_perm_b32
__device__ __forceinline__ uint4 u8x8_to_f16_biased(uint2 v) {
    const uint32_t k = 0x64646464u;
    return make_uint4(__builtin_amdgcn_perm(k, v.x, 0x04010400u), __builtin_amdgcn_perm(k, v.x, 0x04030402u),
                      __builtin_amdgcn_perm(k, v.y, 0x04010400u), __builtin_amdgcn_perm(k, v.y, 0x04030402u));
}
// 8 bf16 -> 8 f16, and the sum of the 8 values as the MFMA will see them
__device__ __forceinline__ uint4 bf16x8_to_f16(bf16x8 v, float& sum) {
    typedef __fp16 h2_t __attribute__((ext_vector_type(2)));
    uint32_t o[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float a = bf2f((bf16_t)v[2 * i]), b = bf2f((bf16_t)v[2 * i + 1]);
        const h2_t h = __builtin_amdgcn_cvt_pkrtz(a, b);
        sum += (float)h[0] + (float)h[1];
        o[i] = __builtin_bit_cast(uint32_t, h);
    }
    return make_uint4(o[0], o[1], o[2], o[3]);
}

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

// PAIR: the blocks of the SECOND range [nblk1, gridDim.x) also multiply every tile with a second weight set, n3 (weights /
// bias / output only; the frames are those of n2): the tile is loaded and converted once for both networks. R2D2 learner:
// online + target over one sequence batch (nblk1 = 0); Ape-X learner: online on s in the first range, online AND target
// on s' in the second.
// PAIR = 1: every block is a pair block (primary network n1, second weight set n3; decided at compile time);
// PAIR = 2: the blocks of the second range are.
template <bool OFF, int PAIR = 0>
__global__ __launch_bounds__(256) void k_conv1_direct(const uint8_t* __restrict__ x, Conv1Net n1, Conv1Net n2, Conv1Net n3, int nblk1) {
    static_assert(!PAIR || DRL_CONV1_F16, "the two-weight-set form exists for the f16 path only");
    const bool second = PAIR != 1 && (int)blockIdx.x >= nblk1;         // block-uniform
    const bool pair_blk = PAIR == 1 || (PAIR == 2 && second);
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
#if DRL_CONV1_F16
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
#else
    bf16x8 wf[2][8];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int s = 0; s < 8; ++s)
            wf[t][s] = *reinterpret_cast<const bf16x8*>(w + ((r >> 2) * 8 + t * 4 + (r & 3)) * 256 + s * 32 + g * 8);
#endif
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
    uint2 bufa[8], bufb[8];
    int b0, b1;
    const int p0 = locate(t0, b0), p1 = locate(t0 + 1, b1);
    const long long f0 = frame_base(b0);
    long long fnext = frame_base(b1);
#pragma unroll
    for (int s = 0; s < 8; ++s) bufa[s] = *reinterpret_cast<const uint2*>(x + f0 + p0 + koff(s));
    long long bnext = fnext + p1;
    // one tile: prefetch tile+1 into `nxt`, compute `cur`. Called with the two buffers swapped (no register copies,
    // and the compiler cannot sink the prefetch below the compute because both buffers are live).
    auto body = [&](uint2 (&cur)[8], uint2 (&nxt)[8], int tile) __attribute__((always_inline)) {
        int b2;
        const int p2 = locate(tile + 2, b2);
        const long long f2 = frame_base(b2);                                                         // tile + 2 (first)
#pragma unroll
        for (int s = 0; s < 8; ++s) nxt[s] = *reinterpret_cast<const uint2*>(x + bnext + koff(s));    // tile + 1
        __builtin_amdgcn_sched_barrier(0);     // keep the prefetch ABOVE the compute (the scheduler sinks it otherwise)
#if DRL_CONV1_F16
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
#else
        f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const bf16x8 xf = u8x8_to_bf16(cur[s]);
            acc[0] = mfma16(wf[0][s], xf, acc[0]);
            acc[1] = mfma16(wf[1][s], xf, acc[1]);
        }
#endif
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
#if DRL_CONV1_F16
            if (pair_blk) {
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    o[2 * t] = pack_bf2(fmaxf(accb[t][0] * SC + b4b[t].x, 0.f), fmaxf(accb[t][1] * SC + b4b[t].y, 0.f));
                    o[2 * t + 1] = pack_bf2(fmaxf(accb[t][2] * SC + b4b[t].z, 0.f), fmaxf(accb[t][3] * SC + b4b[t].w, 0.f));
                }
                *reinterpret_cast<uint4*>(n3.out + (long long)m * 32 + g * 8) = make_uint4(o[0], o[1], o[2], o[3]);
            }
#endif
        }
        bnext = f2 + p2;
    };
    int tile = t0;
    for (; tile + 1 < t1; tile += 2) {      // no skip path inside the loop: the back edge keeps exact vmcnt counts
        body(bufa, bufb, tile);
        body(bufb, bufa, tile + 1);
    }
    if (tile < t1) body(bufa, bufb, tile);  // odd tail
}

// ---------------------------------------------------------------------------------------------------------------
// conv2 / conv3: bf16 NHWC -> bf16 NHWC, 64 output channels, weights resident in LDS in fragment order.
// A k-step is 64 contiguous input bytes (C = 32: one patch pixel; C = 64: half a patch pixel). Each wave owns 32-row
// tiles (2 MFMA row blocks share every weight fragment read from LDS) and runs KS phases (one patch row each) with
// the next phase's loads in flight.
// ---------------------------------------------------------------------------------------------------------------
template <int H, int W, int C, int KS, int S, int RT, int NWV>
__global__ __launch_bounds__(NWV * 64, 2) void k_conv_direct(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
                                                            const float* __restrict__ bias, bf16_t* __restrict__ out,
                                                            int M, int per) {
    constexpr int OH = (H - KS) / S + 1, OW = (W - KS) / S + 1, K = KS * KS * C, NS = K / 32;   // NS k-steps
    constexpr int SPR = KS * C / 32;          // k-steps per patch row (conv2: 4, conv3: 6)
    constexpr int TR = RT * 16;               // rows per wave tile
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];                            // 4 * NS * 1 KB
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 15, g = lane >> 4;
    // weights -> LDS, fragment order: frag (t, s) holds for lane l the 8 k of W[t*16 + l%16][s*32 + (l/16)*8 ..]
    for (int i = tid; i < 4 * NS * 64; i += NWV * 64) {
        const int l = i & 63, f = i >> 6, s = f % NS, t = f / NS;
        *reinterpret_cast<uint4*>(smem + (size_t)i * 16) =
            *reinterpret_cast<const uint4*>(w + (size_t)(t * 16 + (l & 15)) * K + s * 32 + (l >> 4) * 8);
    }
    __syncthreads();                          // the only barrier: weights are read-only from here on
    const int ntiles = (M + TR - 1) / TR;
    // every wave owns `per` CONSECUTIVE tiles: neighbouring output pixels share input pixels, so the re-reads hit L1/L2
    const int t0 = (blockIdx.x * NWV + __builtin_amdgcn_readfirstlane(tid >> 6)) * per, t1 = min(ntiles, t0 + per);   // SGPRs
    float4 b4[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) b4[t] = *reinterpret_cast<const float4*>(bias + t * 16 + g * 4);

    // element offset of patch (row m) origin + this lane's 8-channel group
    auto base_of = [&](int tile, int sub) -> unsigned {
        const int m = min(min(tile, ntiles - 1) * TR + sub * 16 + r, M - 1);
        const int b = m / (OH * OW), rem = m - b * (OH * OW), oy = rem / OW, ox = rem - oy * OW;
        return (unsigned)(((b * H + oy * S) * W + ox * S) * C + g * 8);
    };
    auto load_row = [&](uint4 (&dst)[RT][SPR], const unsigned (&base)[RT], int ky) {
#pragma unroll
        for (int u = 0; u < RT; ++u)
#pragma unroll
            for (int j = 0; j < SPR; ++j) dst[u][j] = *reinterpret_cast<const uint4*>(x + base[u] + ky * (W * C) + j * 32);
    };
    unsigned base[RT], nbase[RT];
#pragma unroll
    for (int u = 0; u < RT; ++u) base[u] = base_of(t0, u);
    uint4 cur[RT][SPR], nxt[RT][SPR];
    load_row(cur, base, 0);
    for (int tile = t0; tile < t1; ++tile) {
#pragma unroll
        for (int u = 0; u < RT; ++u) nbase[u] = base_of(tile + 1, u);
        f32x4 acc[RT][4];
#pragma unroll
        for (int u = 0; u < RT; ++u)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[u][t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ky = 0; ky < KS; ++ky) {
            // compiler fence: the weight fragments are loop-invariant LDS reads and LICM would hoist all 4*NS of them
            // (256+ VGPRs, spills) out of the tile loop; they must be re-read from LDS in every phase
            asm volatile("" ::: "memory");
            if (ky + 1 < KS) load_row(nxt, base, ky + 1);
            else load_row(nxt, nbase, 0);                         // first patch row of the wave's next tile
#pragma unroll
            for (int j = 0; j < SPR; ++j) {
                const int s = ky * SPR + j;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const bf16x8 wf = *reinterpret_cast<const bf16x8*>(smem + ((size_t)(t * NS + s) * 64 + lane) * 16);
#pragma unroll
                    for (int u = 0; u < RT; ++u) acc[u][t] = mfma16(wf, __builtin_bit_cast(bf16x8, cur[u][j]), acc[u][t]);
                }
            }
#pragma unroll
            for (int u = 0; u < RT; ++u)
#pragma unroll
                for (int j = 0; j < SPR; ++j) cur[u][j] = nxt[u][j];
        }
#pragma unroll
        for (int u = 0; u < RT; ++u) {
            const int m = tile * TR + u * 16 + r;
            if (m < M) {
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const uint32_t lo = pack_bf2(fmaxf(acc[u][t][0] + b4[t].x, 0.f), fmaxf(acc[u][t][1] + b4[t].y, 0.f));
                    const uint32_t hi = pack_bf2(fmaxf(acc[u][t][2] + b4[t].z, 0.f), fmaxf(acc[u][t][3] + b4[t].w, 0.f));
                    *reinterpret_cast<uint2*>(out + (long long)m * 64 + t * 16 + g * 4) = make_uint2(lo, hi);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < RT; ++u) base[u] = nbase[u];
    }
}

// tiles per wave so that at most `slots` waves are needed; returns the grid in *grid
static int plan(int ntiles, int slots, int nwv, int* grid) {
    int per = (ntiles + slots - 1) / slots;
    if (per < 1) per = 1;
    *grid = (ntiles + per * nwv - 1) / (per * nwv);
    return per;
}

template <int H, int W, int C, int KS, int S, int RT, int NWV>
static void launch_direct(const void* x, const void* w, const float* bias, void* out, int M, hipStream_t s) {
    constexpr int LDS = 4 * (KS * KS * C / 32) * 1024;
    auto kern = k_conv_direct<H, W, C, KS, S, RT, NWV>;
    static bool once = (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS), true);
    (void)once;
    int grid;
    const int per = plan((M + RT * 16 - 1) / (RT * 16), 256 * 2 * NWV, NWV, &grid);   // 2 workgroups resident per CU
    kern<<<grid, NWV * 64, LDS, s>>>((const bf16_t*)x, (const bf16_t*)w, bias, (bf16_t*)out, M, per);
}

// layer 1: x u8 (+ optional per-frame byte offsets) ; layers 2/3: x bf16 NHWC. out bf16 NHWC, bias + ReLU fused.
// variant (layers 2/3): 0 = 32-row tiles, 4 waves | 1 = 16-row tiles, 8 waves | 2 = 32-row, 8 waves | 3 = 16-row, 4 waves
DRL_API int drl_conv_fwd_direct(int layer, const void* x, const void* w, const float* bias, void* out, int frames,
                                const long long* frame_off, int variant, hipStream_t s) {
    if (frames <= 0) return -1;
    if (layer == 1) {
        const int M = frames * 400;
        int grid;
        const int per = plan((M + 15) / 16, 256 * DRL_CONV1_WAVES_PER_CU, 4, &grid);
        const Conv1Net n1{frame_off, (const bf16_t*)w, bias, (bf16_t*)out, M, per};
        if (frame_off) k_conv1_direct<true><<<grid, 256, 0, s>>>((const uint8_t*)x, n1, n1, n1, grid);
        else k_conv1_direct<false><<<grid, 256, 0, s>>>((const uint8_t*)x, n1, n1, n1, grid);
    } else if (layer == 2) {
        if ((long long)frames * 20 * 20 * 32 >= (1LL << 32)) return -8;
        const int M = frames * 81;
        switch (variant) {
            case 0: launch_direct<20, 20, 32, 4, 2, 2, 4>(x, w, bias, out, M, s); break;
            case 1: launch_direct<20, 20, 32, 4, 2, 1, 8>(x, w, bias, out, M, s); break;
            case 2: launch_direct<20, 20, 32, 4, 2, 2, 8>(x, w, bias, out, M, s); break;
            default: launch_direct<20, 20, 32, 4, 2, 1, 4>(x, w, bias, out, M, s); break;
        }
    } else if (layer == 3) {
        if ((long long)frames * 9 * 9 * 64 >= (1LL << 32)) return -8;
        const int M = frames * 49;
        switch (variant) {
            case 0: launch_direct<9, 9, 64, 3, 1, 2, 4>(x, w, bias, out, M, s); break;
            case 1: launch_direct<9, 9, 64, 3, 1, 1, 8>(x, w, bias, out, M, s); break;
            case 2: launch_direct<9, 9, 64, 3, 1, 2, 8>(x, w, bias, out, M, s); break;
            default: launch_direct<9, 9, 64, 3, 1, 1, 4>(x, w, bias, out, M, s); break;
        }
    } else return -1;
    return (int)hipGetLastError();
}

// conv1 of TWO networks over frames of one store in one launch (both addressed through byte offsets).
DRL_API int drl_conv1_fwd_dual(const void* x, const long long* frame_off1, int frames1, const void* w1, const float* bias1,
                               void* out1, const long long* frame_off2, int frames2, const void* w2, const float* bias2,
                               void* out2, hipStream_t s) {
    if (frames1 <= 0 || frames2 <= 0 || !frame_off1 || !frame_off2) return -1;
    int g1, g2;
    const int slots = 256 * DRL_CONV1_WAVES_PER_CU;
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
#if DRL_CONV1_F16
    if (frames <= 0 || !frame_off) return -1;
    const int M = frames * 400;
    int grid;
    const int per = plan((M + 15) / 16, 256 * DRL_CONV1_WAVES_PER_CU, 4, &grid);
    const Conv1Net n1{frame_off, (const bf16_t*)w1, bias1, (bf16_t*)out1, M, per};
    const Conv1Net n3{frame_off, (const bf16_t*)w2, bias2, (bf16_t*)out2, M, per};
    k_conv1_direct<true, 1><<<grid, 256, 0, s>>>((const uint8_t*)x, n1, n1, n3, grid);
    return (int)hipGetLastError();
#else
    return -1;
#endif
}

// Ape-X learner: network 1 on frames1 (through frame_off1 -> out1a) AND on frames2 (frame_off2 -> out1b), network 2 on
// frames2 only (-> out2): the blocks of the second range multiply every tile of frames2 with both weight sets.
DRL_API int drl_conv1_fwd_apex(const void* x, const long long* frame_off1, int frames1, const long long* frame_off2, int frames2,
                               const void* w1, const float* bias1, void* out1a, void* out1b, const void* w2, const float* bias2,
                               void* out2, hipStream_t s) {
#if DRL_CONV1_F16
    if (frames1 <= 0 || frames2 <= 0 || !frame_off1 || !frame_off2) return -1;
    int g1, g2;
    const int slots = 256 * DRL_CONV1_WAVES_PER_CU;
    const int M1 = frames1 * 400, M2 = frames2 * 400;
    // a pair tile costs about 1.6 single tiles (two MFMA sets, one load + conversion): split the wave slots accordingly
    const long long wgt1 = 10LL * M1, wgt2 = 16LL * M2;
    const int s1 = (int)((long long)slots * wgt1 / (wgt1 + wgt2)), s2 = slots - s1;
    const int per1 = plan((M1 + 15) / 16, s1 > 0 ? s1 : 1, 4, &g1), per2 = plan((M2 + 15) / 16, s2 > 0 ? s2 : 1, 4, &g2);
    const Conv1Net n1{frame_off1, (const bf16_t*)w1, bias1, (bf16_t*)out1a, M1, per1};
    const Conv1Net n2{frame_off2, (const bf16_t*)w1, bias1, (bf16_t*)out1b, M2, per2};
    const Conv1Net n3{frame_off2, (const bf16_t*)w2, bias2, (bf16_t*)out2, M2, per2};
    k_conv1_direct<true, 2><<<g1 + g2, 256, 0, s>>>((const uint8_t*)x, n1, n2, n3, g1);
    return (int)hipGetLastError();
#else
    return -1;
#endif
}

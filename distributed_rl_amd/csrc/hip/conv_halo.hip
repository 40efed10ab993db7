// LDS-halo direct convolutions for conv2 / conv3 of the Nature-CNN torso on gfx950.
//
// The implicit-GEMM kernels of gemm.hip move the im2col VIEW of the input through L2 -> LDS: every input pixel travels
// (KS / S)^2 = 4 (conv2) or 9 (conv3) times, K-tile by K-tile, with a barrier per K-tile. Here a workgroup stages whole
// input FRAMES in LDS once (conv2: 1 frame = 25.6 KB, conv3: 3 frames = 31 KB) and all 4 waves read their MFMA operand
// fragments from that image with the patch offset as the instruction's immediate: one `ds_read_b128` per 16 pixels x 32
// contraction elements, no address arithmetic in the K loop, two barriers per frame group instead of one per K-tile.
//
//   * Weights live in REGISTERS for the life of the (persistent) workgroup: wave (wm, wn) owns output channels
//     [32 wn, 32 wn + 32) = two 16-row MFMA tiles x K / 32 k-steps = 128 (conv2) / 144 (conv3) VGPRs, so an activation
//     fragment read from LDS feeds two MFMAs and the LDS array (256 B/clk) runs at half of what would bind it.
//   * Operands are passed swapped (acc = mfma(W, X)) and the weight rows are permuted (MFMA row i of tile t = channel
//     (i/4)*8 + t*4 + i%4), so a lane ends up with 8 CONSECUTIVE output channels of one pixel: bias + ReLU +
//     v_cvt_pk_bf16_f32 and one 16-byte store.
//   * LDS image: pixel pitch / row pitch / frame pitch are padded so that EVERY fragment read (any tile, any tap, tiles
//     that straddle output rows and frames included) is conflict-free under the ds_read_b128 banking rule (four 16-lane
//     groups x 64 banks); found by brute force: scripts/debug/halo_lds_layout.py.
//       conv2: pixel 5 slots (80 B), row 101 slots, frame 2020 slots = 32 320 B
//       conv3: pixel 10 slots (160 B), row 102 slots, frame 922 slots = 14 752 B
//   * The next group's frames are fetched global -> VGPR (coalesced, 16 B per lane) BEFORE the MFMAs of the current group
//     and written to the second LDS image after them: the HBM / L2 latency hides under the compute, one barrier a round.
//   * Two networks per launch (online + target of a learner step): the first nblk workgroups work on network 1.
//
// The INPUT GRADIENTS of conv3 / conv2 are convolutions too and run through the same kernel (MODE 1 / 2):
//   conv3 dgrad: dx[9x9x64]  = 3x3 stride-1 convolution of dy[7x7x64] zero-padded by 2, flipped + transposed weights;
//   conv2 dgrad: dx[20x20x32] = four 2x2 stride-1 convolutions (one per parity class (py, px) of the input pixel: the taps
//                ky = py + 2 ty, kx = px + 2 tx) of dy[9x9x64] zero-padded by 1; wave w of the workgroup computes class w,
//                so the staged frame is read once for all four classes.
// The border of the LDS image is zeroed once per workgroup (the staging only ever writes the interior), the transformed
// weights come from a tiny pre-pass (k_halo_wprep, once per optimizer step) and the epilogue multiplies with the ReLU mask
// of the layer's input activation instead of adding a bias.
#include "common.h"
#include "conv_frag.h"

typedef __attribute__((ext_vector_type(8))) __bf16 halo_bf16x8;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));   // a VECTOR, not HIP's struct uint4: struct copies become memcpy and pin the array in scratch

__device__ __forceinline__ f32x4 halo_mfma(bf16x8 a, bf16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(halo_bf16x8, a), __builtin_bit_cast(halo_bf16x8, b), c, 0, 0, 0);
}

__device__ uint4 g_halo_dump[512 * 256];      // where the lanes of rows without a frame store (never read)

struct HaloNet {
    const bf16_t* x;       // [frames][SH][SW][CI] (MODE 0: the layer's input; MODE 1 / 2: dy)
    const bf16_t* w;       // MODE 0: [64][KS][KS][CI]; MODE 1: wt3 [64 ci][3][3][64 co]; MODE 2: wt2 [4 classes][32 ci][2][2][64 co]
    const float* bias;     // [64] (MODE 0)
    const bf16_t* mask;    // MODE 1 / 2: the layer's input activation (ReLU mask of dx), laid out like out
    bf16_t* out;           // MODE 0: [frames][OH][OW][64]; MODE 1: [frames][9][9][64]; MODE 2: [frames][20][20][32]
    int frames, nblk;      // workgroups that work on this network
};

enum { HALO_FWD = 0, HALO_DGRAD = 1, HALO_DGRAD_S2 = 2 };

// (weight pre-pass of the input-gradient kernels: HaloPrep / halo_wprep_elem, conv_frag.h)
// SH x SW x CI: source frame; PAD: zero border of the LDS image; P / RP / FP: pixel, row and frame pitch of the image in
// 16-byte slots. F frames per group, TC tiles per wave pass.
template <int SH, int SW, int CI, int PAD, int KS, int S, int P, int RP, int FP, int F, int TC, int OCC, int MODE>
__global__ __launch_bounds__(256, OCC) void k_conv_halo(HaloNet n1, HaloNet n2, HaloPrep prep) {
    constexpr int IH = SH + 2 * PAD, IW = SW + 2 * PAD;
    constexpr int OH = (IH - KS) / S + 1, OW = (IW - KS) / S + 1, OPX = OH * OW, K = KS * KS * CI, NS = K / 32;
    constexpr int CH = CI / 8, SPP = CI / 32;                  // 16-byte chunks per pixel, k-steps per tap
    constexpr int NCHUNK = F * SH * SW * CH, NLD = (NCHUNK + 255) / 256;
    constexpr int WM = MODE == HALO_DGRAD_S2 ? 1 : 2;          // waves that share the tiles of a group
    constexpr int CO = MODE == HALO_DGRAD_S2 ? 32 : 64;        // output channels per pixel
    constexpr int ROWS = F * OPX, NT = (ROWS + 15) / 16, NPASS = (NT + WM * TC - 1) / (WM * TC);
    constexpr unsigned BUF = F * FP * 16;                      // bytes of one LDS image
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];                  // two images of F * FP * 16 bytes
    if (MODE == HALO_FWD && (int)blockIdx.x >= n1.nblk + n2.nblk) {                 // the weight pre-pass workgroups
        const int b = (int)blockIdx.x - n1.nblk - n2.nblk;
        for (int i = b * 256 + (int)threadIdx.x; i < HALO_PREP_ELEMS; i += prep.nblk * 256) halo_wprep_elem(prep, i);
        return;
    }
    const bool second = (int)blockIdx.x >= n1.nblk;            // block-uniform
    const bf16_t* __restrict__ x = second ? n2.x : n1.x;
    const bf16_t* __restrict__ w = second ? n2.w : n1.w;
    const float* __restrict__ bias = second ? n2.bias : n1.bias;
    const bf16_t* __restrict__ mask = second ? n2.mask : n1.mask;
    bf16_t* __restrict__ out = second ? n2.out : n1.out;
    const int frames = second ? n2.frames : n1.frames, nblk = second ? n2.nblk : n1.nblk;
    const int blk = second ? (int)blockIdx.x - n1.nblk : (int)blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = MODE == HALO_DGRAD_S2 ? 0 : wave >> 1, wn = MODE == HALO_DGRAD_S2 ? 0 : wave & 1;
    const int py = wave >> 1, px = wave & 1;                   // MODE 2: the wave's parity class
    const int ngroups = (frames + F - 1) / F;
    if (blk >= ngroups) return;                                 // block-uniform, before any barrier

    // next group's frames: global -> registers (chunks past the last frame re-read its last chunk; their rows are never stored)
    u32x4 st[NLD];
    auto gload = [&](int grp) {
        const long long base = (long long)grp * F * (SH * SW * CI);
        const int nvalid = min(F, frames - grp * F) * (SH * SW * CH);
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int j = tid + i * 256;
            st[i] = *reinterpret_cast<const u32x4*>(x + base + (long long)min(j, nvalid - 1) * 8);
        }
    };
    auto lstore = [&](uint8_t* img) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int j = tid + i * 256;
            if (NCHUNK % 256 == 0 || j < NCHUNK) {
                const int pix = j / CH, c = j - pix * CH, f = pix / (SH * SW), rem = pix - f * (SH * SW), y = rem / SW, xx = rem - y * SW;
                *reinterpret_cast<u32x4*>(img + (size_t)(f * FP + (y + PAD) * RP + (xx + PAD) * P + c) * 16) = st[i];
            }
        }
    };
    int grp = blk;
    gload(grp);

    // weights -> registers (A operand: lane supplies W[row l%16][k-step s, k = (l/16)*8 ..])
    bf16x8 wf[2][NS];
    {
        const bf16_t* wb = w + (size_t)(MODE == HALO_DGRAD_S2 ? wave * 32 : 32 * wn) * K;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int s = 0; s < NS; ++s)
                wf[t][s] = *reinterpret_cast<const bf16x8*>(wb + (size_t)((r >> 2) * 8 + t * 4 + (r & 3)) * K + s * 32 + g * 8);
    }
    float4 b4[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    if (MODE == HALO_FWD) {
#pragma unroll
        for (int t = 0; t < 2; ++t) b4[t] = *reinterpret_cast<const float4*>(bias + 32 * wn + g * 8 + t * 4);
    }
    if (PAD > 0) {                                              // zero both images once: the staging never writes the border
        for (unsigned i = tid; i < 2 * BUF / 16; i += 256) *reinterpret_cast<u32x4*>(smem + (size_t)i * 16) = u32x4{0u, 0u, 0u, 0u};
        __syncthreads();
    }
    // Land weights and bias HERE, once: left to the compiler their waits sit at the first use inside the group loop as
    // counted vmcnt(N) that ignore the prefetch below, so every later round would wait for most of its prefetch in the
    // middle of the MFMAs.
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
        for (int s = 0; s < NS; ++s) asm volatile("" : "+v"(wf[t][s]));
        asm volatile("" : "+v"(b4[t].x), "+v"(b4[t].y), "+v"(b4[t].z), "+v"(b4[t].w));
    }

    // Round = one frame group. Two LDS images: the MFMAs of round i read image i % 2 while the frames of round i + 1,
    // fetched into registers at the top of the round, are written to the other image right after the K loop and BEFORE
    // the round's output stores are issued: the only stores in flight at the staging wait are a whole K loop old.
    // One barrier per round.
    lstore(smem);
    unsigned cur = 0;
    for (; grp < ngroups; grp += nblk) {
        __syncthreads();                                        // image `cur` is complete; nobody reads the other one any more
        // in flight under the MFMAs below. NO branch around the prefetch or its LDS write-out (the last round fetches the
        // last group again and writes an image nobody reads): with either under a condition the compiler has to assume a
        // path on which the staging registers still have loads pending and guards their re-use with counted waits that,
        // on the real path, drain the previous round's output stores.
        gload(min(grp + nblk, ngroups - 1));
        __builtin_amdgcn_sched_barrier(0);                      // keep the prefetch ABOVE the compute
        const int rows = F == 1 ? OPX : min(F, frames - grp * F) * OPX;
        f32x4 acc[NPASS][TC][2];
#pragma unroll
        for (int pass = 0; pass < NPASS; ++pass) {
            const int tile0 = wm * TC + pass * WM * TC;
            unsigned xb[TC];
            long long oaddr[TC];                                // element offset of this lane's 8 output channels
            bool live[TC];
#pragma unroll
            for (int u = 0; u < TC; ++u) {
                // rows past the end of the tile range are computed from the inputs of row ROWS - 1 and store the same
                // values to the same place; rows of frames that do not exist go to the dump page
                const int m = min((tile0 + u) * 16 + r, ROWS - 1);
                const int f = m / OPX, rem = m - f * OPX, oy = rem / OW, ox = rem - oy * OW;
                xb[u] = cur * BUF + (unsigned)(f * FP + (S * oy) * RP + (S * ox) * P + g) * 16u;
                live[u] = F == 1 || m < rows;
                if (MODE == HALO_DGRAD_S2)
                    oaddr[u] = ((long long)(grp * F + f) * (4 * OPX) + (2 * oy + py) * (2 * OW) + 2 * ox + px) * CO + g * 8;
                else
                    oaddr[u] = ((long long)grp * ROWS + m) * CO + 32 * wn + g * 8;
            }
            u32x4 mk[TC];                                       // ReLU mask of dx: loaded before the K loop, used after it
            if (MODE != HALO_FWD) {
#pragma unroll
                for (int u = 0; u < TC; ++u) mk[u] = *reinterpret_cast<const u32x4*>(mask + (live[u] ? oaddr[u] : 0));
            }
#pragma unroll
            for (int u = 0; u < TC; ++u) acc[pass][u][0] = acc[pass][u][1] = f32x4{0.f, 0.f, 0.f, 0.f};
            // software pipeline: the fragments of k-step s + 1 are read while the MFMAs of k-step s run (the scheduler
            // alone issues every read right in front of its first use and exposes the LDS latency once per k-step)
            bf16x8 xf[2][TC];
            auto rd = [&](bf16x8 (&dst)[TC], int s) __attribute__((always_inline)) {
                const int tap = s / SPP, h = s - tap * SPP, ky = tap / KS, kx = tap - ky * KS;
                const int off = (ky * RP + kx * P + h * 4) * 16;     // immediate offset of the ds_read
#pragma unroll
                for (int u = 0; u < TC; ++u) dst[u] = *reinterpret_cast<const bf16x8*>(smem + xb[u] + off);
            };
            rd(xf[0], 0);
            __builtin_amdgcn_sched_group_barrier(0x100, TC, 0);                  // every instruction of the pass is in a
#pragma unroll                                                                   // group, or the scheduler shifts the pattern
            for (int s = 0; s < NS; ++s) {
                if (s + 1 < NS) {
                    rd(xf[(s + 1) & 1], s + 1);
                    __builtin_amdgcn_sched_group_barrier(0x100, TC, 0);          // TC ds reads of k-step s + 1 ...
                }
#pragma unroll
                for (int u = 0; u < TC; ++u) {
                    acc[pass][u][0] = halo_mfma(wf[0][s], xf[s & 1][u], acc[pass][u][0]);
                    acc[pass][u][1] = halo_mfma(wf[1][s], xf[s & 1][u], acc[pass][u][1]);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 2 * TC, 0);          // ... then the 2 TC MFMAs of k-step s
            }
            if (pass + 1 == NPASS) {
                __builtin_amdgcn_sched_barrier(0);
                lstore(smem + (cur ^ 1u) * BUF);                    // next round's image, before this round's stores
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int u = 0; u < TC; ++u) {
                u32x4 o;
                if (MODE == HALO_FWD) {
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        o[2 * t] = pack_bf2(fmaxf(acc[pass][u][t][0] + b4[t].x, 0.f), fmaxf(acc[pass][u][t][1] + b4[t].y, 0.f));
                        o[2 * t + 1] = pack_bf2(fmaxf(acc[pass][u][t][2] + b4[t].z, 0.f), fmaxf(acc[pass][u][t][3] + b4[t].w, 0.f));
                    }
                } else {                                            // dx = relu'(act_in) * acc: a positive bf16 has a non-zero
#pragma unroll                                                     // word and a clear sign bit
                    for (int t = 0; t < 2; ++t)
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const uint32_t mw = mk[u][2 * t + h];
                            const bool lo = (mw & 0xffffu) != 0 && !(mw & 0x8000u), hi = (mw >> 16) != 0 && !(mw & 0x80000000u);
                            o[2 * t + h] = pack_bf2(lo ? acc[pass][u][t][2 * h] : 0.f, hi ? acc[pass][u][t][2 * h + 1] : 0.f);
                        }
                }
                bf16_t* dst = out + oaddr[u];
                if (F > 1) dst = live[u] ? dst : reinterpret_cast<bf16_t*>(g_halo_dump + ((blockIdx.x & 511) * 256 + tid));
                *reinterpret_cast<u32x4*>(dst) = o;
            }
        }
        cur ^= 1u;
    }
}

// CUs that a concurrently running PERSISTENT kernel of another stream occupies (the LSTM recurrences: one workgroup with
// 64 KB of LDS per CU), set by the engine around the launches that overlap with it. The workgroups here are persistent with
// a static share of the work each: one that does not fit next to the other kernel would start when the first ones END and
// double the launch's duration, so the grid is sized to what is resident at once.
static int g_busy_cus = 0, g_busy_lds = 0;
DRL_API int drl_halo_set_busy(int cus, int lds_bytes) {
    if (cus < 0 || cus > 192 || lds_bytes < 0) return -1;
    g_busy_cus = cus;
    g_busy_lds = lds_bytes;
    return 0;
}

template <int SH, int SW, int CI, int PAD, int KS, int S, int P, int RP, int FP, int F, int TC, int OCC, int MODE>
static int launch_halo(HaloNet n1, HaloNet n2, hipStream_t s, HaloPrep prep = HaloPrep{nullptr, nullptr, nullptr, nullptr, 0}) {
    constexpr int LDS = 2 * F * FP * 16;
    static_assert(LDS * OCC <= 160 * 1024, "LDS images of the resident workgroups");
    auto kern = k_conv_halo<SH, SW, CI, PAD, KS, S, P, RP, FP, F, TC, OCC, MODE>;
    static bool once = ((void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS), true);
    (void)once;
    // persistent workgroups: OCC per CU, dealt to the two networks in proportion to their frames
    int fit = OCC;                                     // workgroups of this kernel that fit on a busy CU
    if (g_busy_cus) { fit = (160 * 1024 - g_busy_lds) / LDS; fit = fit < 0 ? 0 : (fit > OCC ? OCC : fit); }
    const int slots = (256 - g_busy_cus) * OCC + g_busy_cus * fit;
    const int g1 = (n1.frames + F - 1) / F, g2 = n2.frames > 0 ? (n2.frames + F - 1) / F : 0;
    int b1, b2 = 0;
    if (g1 + g2 <= slots) { b1 = g1; b2 = g2; }
    else if (g2 == 0) b1 = slots;
    else {
        b1 = (int)((long long)slots * g1 / (g1 + g2));
        b1 = b1 < 1 ? 1 : (b1 > slots - 1 ? slots - 1 : b1);
        b2 = slots - b1;
    }
    // equal rounds per workgroup: ceil(groups / rounds) workgroups do the same work in the same number of rounds
    auto even = [](int groups, int blocks) { const int rounds = (groups + blocks - 1) / blocks; return (groups + rounds - 1) / rounds; };
    n1.nblk = even(g1, b1);
    n2.nblk = g2 ? even(g2, b2) : 0;
    kern<<<n1.nblk + n2.nblk + prep.nblk, 256, LDS, s>>>(n1, n2, prep);
    return (int)hipGetLastError();
}

// layer 2: x bf16 [n,20,20,32] -> out bf16 [n,9,9,64]; layer 3: x bf16 [n,9,9,64] -> out bf16 [n,7,7,64]; bias + ReLU fused.
// frames2 = 0: one network. variant: 0 = two workgroups per CU, 1 = one workgroup per CU (wider tile pass)
DRL_API int drl_conv_halo_fwd(int layer, const void* x1, int frames1, const void* w1, const float* bias1, void* out1,
                              const void* x2, int frames2, const void* w2, const float* bias2, void* out2, int variant,
                              const void* prep_w2, const void* prep_w3, void* prep_wt2, void* prep_wt3, hipStream_t s) {
    if (frames1 <= 0 || frames2 < 0) return -1;
    // prep_wt2 != null (layer 2 only): also write the flipped / transposed weights of the input-gradient kernels
    HaloPrep prep{(const bf16_t*)prep_w2, (const bf16_t*)prep_w3, (bf16_t*)prep_wt2, (bf16_t*)prep_wt3, prep_wt2 ? HALO_PREP_BLOCKS : 0};
    if (prep.nblk && (layer != 2 || !prep_w2 || !prep_w3 || !prep_wt3)) return -1;
    HaloNet n1{(const bf16_t*)x1, (const bf16_t*)w1, bias1, nullptr, (bf16_t*)out1, frames1, 0};
    HaloNet n2{(const bf16_t*)x2, (const bf16_t*)w2, bias2, nullptr, (bf16_t*)out2, frames2, 0};
    if (layer == 2) {
        if (variant == 1) return launch_halo<20, 20, 32, 0, 4, 2, 5, 101, 2020, 1, 3, 1, HALO_FWD>(n1, n2, s, prep);
        return launch_halo<20, 20, 32, 0, 4, 2, 5, 101, 2020, 1, 3, 2, HALO_FWD>(n1, n2, s, prep);
    }
    if (layer == 3) {
        if (variant == 1) return launch_halo<9, 9, 64, 0, 3, 1, 10, 102, 922, 3, 5, 1, HALO_FWD>(n1, n2, s);
        return launch_halo<9, 9, 64, 0, 3, 1, 10, 102, 922, 2, 2, 2, HALO_FWD>(n1, n2, s);
    }
    return -1;
}

// ---- input gradients ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_halo_wprep(HaloPrep p) {
    halo_wprep_elem(p, blockIdx.x * 256 + threadIdx.x);
}

DRL_API int drl_conv_halo_wprep(const void* w2, const void* w3, void* wt2, void* wt3, hipStream_t s) {
    k_halo_wprep<<<(HALO_PREP_ELEMS + 255) / 256, 256, 0, s>>>(HaloPrep{(const bf16_t*)w2, (const bf16_t*)w3, (bf16_t*)wt2, (bf16_t*)wt3, 0});
    return (int)hipGetLastError();
}

// layer 3: dy bf16 [n,7,7,64], wt = wt3, act_in / dx bf16 [n,9,9,64]; layer 2: dy bf16 [n,9,9,64], wt = wt2, act_in / dx bf16
// [n,20,20,32]. dx = relu'(act_in) * conv_transpose(dy, w).
DRL_API int drl_conv_halo_dgrad(int layer, const void* dy, int frames, const void* wt, const void* act_in, void* dx, hipStream_t s) {
    if (frames <= 0 || !act_in) return -1;
    HaloNet n1{(const bf16_t*)dy, (const bf16_t*)wt, nullptr, (const bf16_t*)act_in, (bf16_t*)dx, frames, 0};
    HaloNet n2{nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0};
    // image pitches from scripts/debug/halo_lds_layout.py (conflict-free for every tile and tap)
    if (layer == 3) return launch_halo<7, 7, 64, 2, 3, 1, 10, 122, 1354, 3, 4, 1, HALO_DGRAD>(n1, n2, s);
    if (layer == 2) return launch_halo<9, 9, 64, 1, 2, 1, 10, 116, 1276, 1, 7, 2, HALO_DGRAD_S2>(n1, n2, s);
    return -1;
}

// LDS-halo weight gradients of conv2 / conv3 for gfx950 (SURVEY G2, G3 backward).
//
//   dW[co][kh][kw][ci] = sum over (frame, oy, ox) of dy[frame][oy][ox][co] * x[frame][oy*S + kh][ox*S + kw][ci]
//   db[co]             = sum over (frame, oy, ox) of dy[frame][oy][ox][co]
//
// As a GEMM this is 64 x K (K = 512 / 576) with a CONTRACTION OVER PIXELS (frames * 81 / 49 of them, 100 K - 400 K). The
// split-K implicit GEMM of gemm.hip re-gathers every input pixel (KS/S)^2 times through L2 -> LDS, runs 64 x 64 tiles and
// pays an atomic round per split: 6-7 % MFMA-busy, 52 - 82 us per launch of 2560 frames. Here:
//   * a persistent workgroup (one per CU, 8 waves) stages whole frames of x AND dy in LDS once (double buffered, the loads of
//     the next frames in flight under the MFMAs of the current ones) and keeps its 64 x K / 8 slice of dW in registers for
//     its whole life. Float atomics execute at the memory side at ~1.3 TB/s chip-wide whatever their shape: 256 workgroups
//     x 128 KB of them took 40 us, more than the whole main loop. So every workgroup STORES its accumulators once, in
//     fragment order (256 contiguous bytes per instruction, 6 TB/s), into its own row of a workspace, and a second small
//     kernel sums the rows (8 row groups in parallel) and adds 8 values per element into dW;
//   * both operands need the contraction (pixel) index along the MFMA k dimension while memory is channel-major: that is
//     what ds_read_b64_tr_b16 does (a 4 pixel x 16 channel block, delivered column-major). Every lane supplies its own row
//     address, so a "row" can be any pixel: the LDS images are PLANAR per 16-channel block (32 bytes per pixel and block),
//     pixels of a stride-parity class next to each other, and one k-step of 32 contraction slots is 4 OCTETS of 8 pixels:
//     a half-wave instruction reads 8 pixels that are neighbours in the image (an output row's ox = 0..7 -> 8 consecutive
//     32-byte rows = all 64 banks once), or 8 pixels of one column (row pitch odd -> 8 different bank groups). No bank
//     conflicts, no swizzle, no padding between pixels;
//   * 9 x 9 outputs (conv2) = 9 row octets + the ox = 8 column as one octet + the corner = 11 octets -> 3 k-steps (84 % of
//     the slots carry a pixel); 7 x 7 (conv3) = 7 row octets with one empty slot each -> 2 k-steps (77 %). Empty slots hold
//     dy = 0 (written once); what x holds there does not matter as long as it is finite (LDS is zeroed at start);
//   * db: the thread that stages a 16-byte chunk of dy always sees the same 8 channels, sums them in registers and the
//     workgroup reduces once at the end.
#include "common.h"

typedef uint32_t wg_u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((ext_vector_type(8))) __bf16 wg_mfma_bf16x8;
typedef __attribute__((ext_vector_type(4))) short wg_s16x4;

#define WG_T 512

// b rounded up to 32 (mod 128)
constexpr int wg_pad128(int b) { return b + ((32 - b % 128) + 128) % 128; }

__device__ wg_u32x4 g_wg_zero;             // 16 zero bytes: the source of every staging load that has no frame behind it

template <int H_, int C_, int KS_, int S_, int F_>
struct WgCfg {
    static constexpr int H = H_, C = C_, KS = KS_, S = S_, F = F_;
    static constexpr int OH = (H - KS) / S + 1;                     // 9 (conv2) / 7 (conv3), square
    static constexpr int CL = S * S, HC = H / S;                    // stride-parity classes, size of a class image
    static constexpr int RP = (HC & 1) ? HC : HC + 1;               // odd row pitch of a class image, in pixels
    static constexpr int CB = C / 16;                               // 16-channel blocks of the input
    static constexpr int NOCT = OH <= 8 ? 8 : 12;                   // octets per frame, multiple of 4
    static constexpr int KSTEPS = NOCT / 4;
    // plane strides = 32 (mod 128): the four planes a staging wave writes side by side land on four different bank groups
    static constexpr int PLANE_X = wg_pad128((HC + 2) * RP * 32);      // two spare rows: reads of empty slots stay inside
    static constexpr int FRAME_X = CL * CB * PLANE_X;
    static constexpr int PLANE_A = wg_pad128(NOCT * 8 * 32);
    static constexpr int FRAME_A = 4 * PLANE_A;                     // 64 output channels = 4 blocks
    static constexpr int BUF = F * (FRAME_X + FRAME_A);
    static constexpr int K = KS * KS * C, NT = K / 16;              // n-tiles of 16 k
    static constexpr int NTW = (NT + 7) / 8;                        // n-tiles of a wave (some waves one less)
    static constexpr int PART = 8 * 4 * NTW * 4 * 64;               // floats of a workgroup's accumulators in fragment order
    static constexpr int XCH = H * H * C / 8, ACH = OH * OH * 8;    // 16-byte chunks of a frame
    static constexpr int NXP = (F * XCH + WG_T - 1) / WG_T, NAP = (F * ACH + WG_T - 1) / WG_T;
    static_assert(OH <= 9 && H % S == 0 && KS % S == 0, "row octets + one column octet + corner");
    static_assert(WG_T % 8 == 0, "a thread stages the same 8 channels of dy every time");
};

// byte offset of 16-byte chunk j of an x frame (NHWC, chunk = 8 channels) inside its LDS frame image
template <class Cf>
__device__ __forceinline__ int wg_xoff(int j) {
    constexpr int CPP = Cf::C / 8;
    const int pix = j / CPP, cc = j - pix * CPP;
    const int y = pix / Cf::H, x = pix - y * Cf::H;
    const int cls = (y % Cf::S) * Cf::S + (x % Cf::S);
    return (cls * Cf::CB + (cc >> 1)) * Cf::PLANE_X + ((y / Cf::S) * Cf::RP + (x / Cf::S)) * 32 + (cc & 1) * 16;
}
// ... of a dy frame (64 channels = 8 chunks per pixel): pixel -> contraction slot
template <class Cf>
__device__ __forceinline__ int wg_aoff(int j) {
    const int pix = j >> 3, cc = j & 7;
    const int oy = pix / Cf::OH, ox = pix - oy * Cf::OH;
    const int slot = ox < 8 ? oy * 8 + ox : (oy < 8 ? 72 + oy : 80);
    return (cc >> 1) * Cf::PLANE_A + slot * 32 + (cc & 1) * 16;
}
// pixel (inside a class image, before the tap offset) of element e of octet o
template <class Cf>
__device__ __forceinline__ int wg_oct_pix(int o, int e) {
    if (Cf::OH <= 8) return o * Cf::RP + e;                         // row octet oy = o (o = OH .. 7: empty)
    return o < 9 ? o * Cf::RP + e : (o == 9 ? e * Cf::RP + 8 : 8 * Cf::RP + 8);
}

__device__ __forceinline__ wg_s16x4 wg_tr_read(uint32_t addr) {
    wg_s16x4 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(addr));
    return v;
}

template <class Cf>
__global__ __launch_bounds__(WG_T, 2) void k_conv_wgrad_halo(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy,
                                                             float* __restrict__ ws, float* __restrict__ db, int frames) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const uint8_t*)smem;
    // everything any read can touch is finite from the start (empty dy slots must stay exactly zero)
    for (int i = tid; i < 2 * Cf::BUF / 16; i += WG_T) reinterpret_cast<wg_u32x4*>(smem)[i] = wg_u32x4{0, 0, 0, 0};

    // lane roles in a transposed read: half h, group g of the half, row q of the group's 4 x 16 block, 8-byte piece p
    const int h = lane >> 5, e = ((lane >> 4) & 1) * 4 + ((lane >> 2) & 3), p = lane & 3;
    uint32_t vA[2], vB[Cf::KSTEPS][2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        vA[s] = (uint32_t)(((2 * h + s) * 8 + e) * 32 + p * 8);
#pragma unroll
        for (int ks = 0; ks < Cf::KSTEPS; ++ks) vB[ks][s] = (uint32_t)(wg_oct_pix<Cf>(4 * ks + 2 * h + s, e) * 32 + p * 8);
    }
    // this wave's n-tiles (16 k each): NT / 8 per wave, the first NT % 8 waves ... no: waves w and w + 4 share a SIMD, so
    // the waves with one tile more are 0..3 and every SIMD gets the same total
    constexpr int LO = Cf::NT / 8, EXTRA = Cf::NT - 8 * LO;         // conv2: 4, 0   conv3: 4, 4
    static_assert(EXTRA == 0 || EXTRA == 4, "one extra tile on the first four waves");
    const int cnt = wave < EXTRA ? LO + 1 : LO;
    const int nt0 = wave < EXTRA ? wave * (LO + 1) : EXTRA * (LO + 1) + (wave - EXTRA) * LO;
    uint32_t woff[Cf::NTW];
#pragma unroll
    for (int t = 0; t < Cf::NTW; ++t) {
        const int nt = min(nt0 + t, Cf::NT - 1), tap = nt / Cf::CB, cb = nt - tap * Cf::CB;
        const int kh = tap / Cf::KS, kw = tap - kh * Cf::KS;
        const int cls = (kh % Cf::S) * Cf::S + (kw % Cf::S);
        woff[t] = (uint32_t)__builtin_amdgcn_readfirstlane((cls * Cf::CB + cb) * Cf::PLANE_X + ((kh / Cf::S) * Cf::RP + kw / Cf::S) * 32);
    }

    f32x4 acc[4][Cf::NTW];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int t = 0; t < Cf::NTW; ++t) acc[m][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bsum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    const int ngroups = (frames + Cf::F - 1) / Cf::F;
    // staging plan of this thread, fixed for the life of the workgroup: LDS byte offset of every chunk it copies (-1: none)
    int xo[Cf::NXP], ao[Cf::NAP], xf[Cf::NXP], af[Cf::NAP];
#pragma unroll
    for (int i = 0; i < Cf::NXP; ++i) {
        const int j = tid + i * WG_T, f = j / Cf::XCH;
        xf[i] = f;
        xo[i] = j < Cf::F * Cf::XCH ? f * Cf::FRAME_X + wg_xoff<Cf>(j - f * Cf::XCH) : -1;
    }
#pragma unroll
    for (int i = 0; i < Cf::NAP; ++i) {
        const int j = tid + i * WG_T, f = j / Cf::ACH;
        af[i] = f;
        ao[i] = j < Cf::F * Cf::ACH ? Cf::F * Cf::FRAME_X + f * Cf::FRAME_A + wg_aoff<Cf>(j - f * Cf::ACH) : -1;
    }
    uint8_t* const buf0 = smem;
    uint8_t* const buf1 = smem + Cf::BUF;
    uint8_t* const dump = smem + 2 * Cf::BUF + tid * 16;
    typedef wg_u32x4 (&XSet)[Cf::NXP];
    typedef wg_u32x4 (&ASet)[Cf::NAP];
    // every load is issued unconditionally (chunks without a frame behind them read a zero page): with a branch around a
    // load the compiler's wait counting gives up and drains the whole prefetch queue at every use
    auto fetch = [&](XSet sx, ASet sa, int g) __attribute__((always_inline)) {
        const long long f0 = (long long)g * Cf::F;
#pragma unroll
        for (int i = 0; i < Cf::NXP; ++i) {
            const bool ok = xo[i] >= 0 && g < ngroups && f0 + xf[i] < frames;
            const wg_u32x4* src = ok ? reinterpret_cast<const wg_u32x4*>(x + (f0 * Cf::XCH + tid + i * WG_T) * 8) : &g_wg_zero;
            sx[i] = *src;
        }
#pragma unroll
        for (int i = 0; i < Cf::NAP; ++i) {
            const bool ok = ao[i] >= 0 && g < ngroups && f0 + af[i] < frames;
            const wg_u32x4* src = ok ? reinterpret_cast<const wg_u32x4*>(dy + (f0 * Cf::ACH + tid + i * WG_T) * 8) : &g_wg_zero;
            sa[i] = *src;
        }
    };
    // ... and every LDS store too: chunks this thread does not own go to a dump area behind the two buffers
    auto stash = [&](uint8_t* buf, XSet sx, ASet sa) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < Cf::NXP; ++i)
            *reinterpret_cast<wg_u32x4*>(xo[i] >= 0 ? buf + xo[i] : dump) = sx[i];
#pragma unroll
        for (int i = 0; i < Cf::NAP; ++i) {
            *reinterpret_cast<wg_u32x4*>(ao[i] >= 0 ? buf + ao[i] : dump) = sa[i];
#pragma unroll
            for (int c = 0; c < 4; ++c) {                           // bias gradient: this chunk holds channels 8 (tid % 8) .. + 7
                bsum[2 * c] += __uint_as_float(sa[i][c] << 16);     // (chunks without a frame are zero)
                bsum[2 * c + 1] += __uint_as_float(sa[i][c] & 0xffff0000u);
            }
        }
    };
    auto compute = [&](uint32_t cur) __attribute__((always_inline)) {
        // opaque: the fragment addresses below are recomputed every round (one v_add per read). Left visible, the compiler
        // hoists all of them out of the round loop for both buffers: ~100 more live registers, spills
        asm volatile("" : "+v"(cur));
#pragma unroll
        for (int f = 0; f < Cf::F; ++f)
#pragma unroll
            for (int ks = 0; ks < Cf::KSTEPS; ++ks) {
                bf16x8 fa[4], fb[Cf::NTW];
                const uint32_t a0 = cur + Cf::F * Cf::FRAME_X + f * Cf::FRAME_A + ks * 1024;
                const uint32_t b0 = cur + f * Cf::FRAME_X;
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const wg_s16x4 lo = wg_tr_read(a0 + m * Cf::PLANE_A + vA[0]), hi = wg_tr_read(a0 + m * Cf::PLANE_A + vA[1]);
                    fa[m] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                }
#pragma unroll
                for (int t = 0; t < Cf::NTW; ++t) {
                    if (t < LO || cnt > LO) {                       // wave-uniform: EXEC stays all ones for the transposed reads
                        const wg_s16x4 lo = wg_tr_read(b0 + woff[t] + vB[ks][0]), hi = wg_tr_read(b0 + woff[t] + vB[ks][1]);
                        fb[t] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    } else fb[t] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
                }
                // the asm reads are invisible to the compiler's counters: wait for all of them, then make every fragment
                // opaque HERE so that no MFMA can be scheduled above the wait
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int m = 0; m < 4; ++m) asm volatile("" : "+v"(fa[m]));
#pragma unroll
                for (int t = 0; t < Cf::NTW; ++t) asm volatile("" : "+v"(fb[t]));
#pragma unroll
                for (int t = 0; t < Cf::NTW; ++t)
                    if (t < LO || cnt > LO) {
#pragma unroll
                        for (int m = 0; m < 4; ++m)
                            acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(wg_mfma_bf16x8, fa[m]),
                                                                                __builtin_bit_cast(wg_mfma_bf16x8, fb[t]), acc[m][t], 0, 0, 0);
                    }
            }
    };

    // Two register sets: the frames of round r + 1 were fetched during round r - 1 and are written to LDS at the end of
    // round r, the fetch of round r + 2 is issued at the start of round r - a whole round of MFMAs hides each load.
    wg_u32x4 sx0[Cf::NXP], sa0[Cf::NAP], sx1[Cf::NXP], sa1[Cf::NAP];
    const int G_ = gridDim.x;
    int grp = blockIdx.x;
    fetch(sx0, sa0, grp);
    fetch(sx1, sa1, grp + G_);
    __syncthreads();                                                // the zero fill is complete
    stash(buf0, sx0, sa0);
    __syncthreads();
    for (; grp < ngroups; grp += 2 * G_) {                          // two rounds per trip: the register sets stay static
        fetch(sx0, sa0, grp + 2 * G_);
        compute(lds0);
        stash(buf1, sx1, sa1);                                      // buf1 was last read a round ago: everybody is past that barrier
        __syncthreads();
        fetch(sx1, sa1, grp + 3 * G_);
        if (grp + G_ < ngroups) compute(lds0 + Cf::BUF);
        stash(buf0, sx0, sa0);
        __syncthreads();
    }

    // ---- the workgroup's accumulators, in fragment order: 256 contiguous bytes per store instruction ----
    float* out = ws + (long long)blockIdx.x * Cf::PART + (wave * 4 * Cf::NTW * 4) * 64 + lane;
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int t = 0; t < Cf::NTW; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) out[((m * Cf::NTW + t) * 4 + j) * 64] = acc[m][t][j];
    // bias gradient: thread tid summed channels 8 (tid % 8) .. + 7
    float* red = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int c = 0; c < 8; ++c) red[c * WG_T + tid] = bsum[c];
    __syncthreads();
    if (tid < 64) {
        const int blk = tid >> 3, c = tid & 7;                      // channel 8 blk + c: threads with tid % 8 == blk
        float s = 0.f;
        for (int k = 0; k < WG_T / 8; ++k) s += red[c * WG_T + 8 * k + blk];
        atomicAdd(db + tid, s);
    }
}

// dw[co][k] += sum over the workgroups' rows of ws; slot = (((wave * 4 + m) * NTW + t) * 4 + j) * 64 + lane holds element
// (co = 16 m + 4 (lane / 16) + j, k = 16 (nt0(wave) + t) + lane % 16). grid = (PART / 256, row groups).
template <class Cf>
__global__ __launch_bounds__(256) void k_conv_wgrad_reduce(const float* __restrict__ ws, float* __restrict__ dw, int rows) {
    constexpr int LO = Cf::NT / 8, EXTRA = Cf::NT - 8 * LO;
    const int slot = blockIdx.x * 256 + threadIdx.x, lane = slot & 63;
    int q = slot >> 6;
    const int j = q & 3; q >>= 2;
    const int t = q % Cf::NTW; q /= Cf::NTW;
    const int m = q & 3, wave = q >> 2;
    const int cnt = wave < EXTRA ? LO + 1 : LO;
    const int nt0 = wave < EXTRA ? wave * (LO + 1) : EXTRA * (LO + 1) + (wave - EXTRA) * LO;
    if (t >= cnt) return;
    const int r0 = (int)((long long)rows * blockIdx.y / gridDim.y), r1 = (int)((long long)rows * (blockIdx.y + 1) / gridDim.y);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int r = r0;
    for (; r + 4 <= r1; r += 4) {
        s0 += ws[(long long)r * Cf::PART + slot];
        s1 += ws[(long long)(r + 1) * Cf::PART + slot];
        s2 += ws[(long long)(r + 2) * Cf::PART + slot];
        s3 += ws[(long long)(r + 3) * Cf::PART + slot];
    }
    for (; r < r1; ++r) s0 += ws[(long long)r * Cf::PART + slot];
    atomicAdd(dw + (16 * m + 4 * (lane >> 4) + j) * Cf::K + (nt0 + t) * 16 + (lane & 15), (s0 + s1) + (s2 + s3));
}

#define WG_MAX_GRID 256
template <class Cf>
static int launch_wgrad(const void* x, const void* dy, float* dw, float* db, float* ws, int frames, int max_wgs, hipStream_t s) {
    auto kern = k_conv_wgrad_halo<Cf>;
    constexpr int LDS = 2 * Cf::BUF + WG_T * 16;
    static_assert(LDS <= 160 * 1024 && LDS >= 8 * WG_T * 4, "two frame buffers + dump; the bias reduction reuses them");
    static_assert(Cf::PART % 256 == 0, "reduce grid");
    static bool once = ((void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS), true);
    (void)once;
    const int ngroups = (frames + Cf::F - 1) / Cf::F;
    if (max_wgs <= 0 || max_wgs > WG_MAX_GRID) max_wgs = WG_MAX_GRID;
    const int grid = ngroups < max_wgs ? ngroups : max_wgs;
    kern<<<grid, WG_T, LDS, s>>>((const bf16_t*)x, (const bf16_t*)dy, ws, db, frames);
    const int groups = grid >= 64 ? 8 : (grid >= 8 ? 2 : 1);
    k_conv_wgrad_reduce<Cf><<<dim3(Cf::PART / 256, groups), 256, 0, s>>>(ws, dw, grid);
    return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------------
// conv1: dW1[co][c][kh][kw] = 1/255 * sum dy[frame][oy][ox][co] * plane[c][4 oy + kh][4 ox + kw]  (8 x 8, stride 4, u8 planes)
// ---------------------------------------------------------------------------------------------------------------------
// Space-to-depth makes it the same problem as above: with kh = 4 a + r, kw = 4 b + t the 84 x 84 x 4 stack is a 21 x 21 image
// of 64-vectors (c, r, t) - position (yy, xx) holds the 4 x 4 pixel block of each plane - and conv1 is a 2 x 2, stride-1
// convolution over it with 64 input channels. A 16-channel block = plane c, so the LDS image is planar per plane: staging
// converts 16 u8 pixels of a plane row (one 16-byte load) into four 8-byte pieces (4 pixels = 4 consecutive t) of four
// neighbouring positions. The work unit is HALF a frame (output rows 10 hf .. 10 hf + 9: 11 position rows = 44 contiguous
// plane rows = 3696 bytes per plane, 231 whole chunks), two units double-buffered in 88 KB. A 20-wide output row = two
// full octets + 4 pixels; the leftovers of two rows share an octet: 25 octets per unit -> 7 k-steps (89 % of the slots).
// 32 output channels = 2 m-tiles; a wave owns 2 of the 16 n-tiles ((a, b) x plane).
struct Wg1 {
    static constexpr int RP = 21, NOCT = 28, KSTEPS = 7, F = 1;
    static constexpr int PLANE_X = wg_pad128(11 * RP * 32), FRAME_X = 4 * PLANE_X;
    static constexpr int PLANE_A = wg_pad128(NOCT * 8 * 32), FRAME_A = 2 * PLANE_A;
    static constexpr int BUF = FRAME_X + FRAME_A;
    static constexpr int XCH = 4 * 231, ACH = 200 * 4;              // 16-byte chunks of a unit: u8 planes / dy
    static constexpr int NXP = (XCH + WG_T - 1) / WG_T, NAP = (ACH + WG_T - 1) / WG_T;
    static constexpr int PART = 8 * 2 * 2 * 4 * 64;                 // floats of a workgroup's accumulators
    static constexpr int K = 256;
};

__device__ __forceinline__ int wg1_oct_pix(int o, int e) {
    if (o >= 25) return 0;
    const int rp = o / 5, j = o - rp * 5;
    if (j < 4) return (2 * rp + (j >> 1)) * Wg1::RP + (j & 1) * 8 + e;
    return e < 4 ? (2 * rp) * Wg1::RP + 16 + e : (2 * rp + 1) * Wg1::RP + 12 + e;
}
__device__ __forceinline__ int wg1_slot(int oyl, int ox) {
    const int rp = oyl >> 1, par = oyl & 1;
    return ox < 16 ? (5 * rp + 2 * par + (ox >> 3)) * 8 + (ox & 7) : (5 * rp + 4) * 8 + par * 4 + (ox - 16);
}
// 4 u8 -> 4 bf16 (exact): v_cvt_f32_ubyteN x 4, one v_perm_b32 per pair keeps the two high halves
__device__ __forceinline__ uint2 wg1_u8x4_to_bf16(uint32_t v) {
    const float f0 = (float)(v & 0xffu), f1 = (float)((v >> 8) & 0xffu), f2 = (float)((v >> 16) & 0xffu), f3 = (float)(v >> 24);
    return make_uint2(__builtin_amdgcn_perm(__float_as_uint(f1), __float_as_uint(f0), 0x07060302u),
                      __builtin_amdgcn_perm(__float_as_uint(f3), __float_as_uint(f2), 0x07060302u));
}

__global__ __launch_bounds__(WG_T, 2) void k_conv1_wgrad_halo(const uint8_t* __restrict__ store, const long long* __restrict__ frame_off,
                                                              const bf16_t* __restrict__ dy, float* __restrict__ ws,
                                                              float* __restrict__ db, int frames) {
    typedef Wg1 Cf;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const uint8_t*)smem;
    for (int i = tid; i < 2 * Cf::BUF / 16; i += WG_T) reinterpret_cast<wg_u32x4*>(smem)[i] = wg_u32x4{0, 0, 0, 0};

    const int h = lane >> 5, e = ((lane >> 4) & 1) * 4 + ((lane >> 2) & 3), p = lane & 3;
    uint32_t vA[2], vB[Cf::KSTEPS][2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        vA[s] = (uint32_t)(((2 * h + s) * 8 + e) * 32 + p * 8);
#pragma unroll
        for (int ks = 0; ks < Cf::KSTEPS; ++ks) vB[ks][s] = (uint32_t)(wg1_oct_pix(4 * ks + 2 * h + s, e) * 32 + p * 8);
    }
    uint32_t woff[2];                                               // n-tile nt = 4 (2 a + b) + c
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int nt = 2 * wave + t, tap = nt >> 2, c = nt & 3;
        woff[t] = (uint32_t)__builtin_amdgcn_readfirstlane(c * Cf::PLANE_X + ((tap >> 1) * Cf::RP + (tap & 1)) * 32);
    }
    f32x4 acc[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int t = 0; t < 2; ++t) acc[m][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bsum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    const int nunits = 2 * frames;
    // staging plan of this thread: x chunk j = plane j / 231, 16 pixels from byte 16 (j % 231) of the unit's 44 plane rows
    int xsrc[Cf::NXP], xdst[Cf::NXP][4], ao[Cf::NAP];
#pragma unroll
    for (int i = 0; i < Cf::NXP; ++i) {
        const int j = tid + i * WG_T, c = j / 231, jj = j - c * 231;
        xsrc[i] = j < Cf::XCH ? c * 7056 + jj * 16 : -1;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            // store instruction g4 of a lane writes its pixel group (g4 + lane) % 4: the 8-byte pieces of one instruction
            // then spread over the four positions (32 B apart) x four plane rows (8 B apart) = all 32 banks. With every lane
            // on group g4 the lanes are 128 B apart - ONE bank pair, a 32-way conflict: half of the kernel's LDS cycles
            const int gg = (g4 + lane) & 3;
            const int lp = jj * 16 + 4 * gg, yl = lp / 84, xx = (lp - yl * 84) >> 2;
            xdst[i][g4] = j < Cf::XCH ? c * Cf::PLANE_X + ((yl >> 2) * Cf::RP + xx) * 32 + (yl & 3) * 8 : -1;
        }
    }
#pragma unroll
    for (int i = 0; i < Cf::NAP; ++i) {
        const int j = tid + i * WG_T, pix = j >> 2, cc = j & 3, oyl = pix / 20, ox = pix - oyl * 20;
        ao[i] = j < Cf::ACH ? Cf::FRAME_X + (cc >> 1) * Cf::PLANE_A + wg1_slot(oyl, ox) * 32 + (cc & 1) * 16 : -1;
    }
    uint8_t* const buf0 = smem;
    uint8_t* const buf1 = smem + Cf::BUF;
    uint8_t* const dump = smem + 2 * Cf::BUF + tid * 16;
    typedef wg_u32x4 (&XSet)[Cf::NXP];
    typedef wg_u32x4 (&ASet)[Cf::NAP];
    auto fetch = [&](XSet sx, ASet sa, int u) __attribute__((always_inline)) {
        const bool uok = u < nunits;
        const int f = min(u, nunits - 1) >> 1, hf = u & 1;
        const long long base = frame_off[f] + hf * 3360;            // 40 plane rows
#pragma unroll
        for (int i = 0; i < Cf::NXP; ++i) {
            const wg_u32x4* src = (xsrc[i] >= 0 && uok) ? reinterpret_cast<const wg_u32x4*>(store + base + xsrc[i]) : &g_wg_zero;
            sx[i] = *src;
        }
#pragma unroll
        for (int i = 0; i < Cf::NAP; ++i) {
            const wg_u32x4* src = (ao[i] >= 0 && uok) ? reinterpret_cast<const wg_u32x4*>(dy + ((long long)u * Cf::ACH + tid + i * WG_T) * 8)
                                                     : &g_wg_zero;
            sa[i] = *src;
        }
    };
    auto stash = [&](uint8_t* buf, XSet sx, ASet sa) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < Cf::NXP; ++i) {
            // the chunk's four dwords rotated left by lane % 4 (two conditional swaps), so that word g4 is pixel group
            // (g4 + lane) % 4 - the order xdst is in
            uint32_t w0 = sx[i][0], w1 = sx[i][1], w2 = sx[i][2], w3 = sx[i][3];
            if (lane & 1) { const uint32_t t = w0; w0 = w1; w1 = w2; w2 = w3; w3 = t; }
            if (lane & 2) { uint32_t t = w0; w0 = w2; w2 = t; t = w1; w1 = w3; w3 = t; }
            const uint32_t wr[4] = {w0, w1, w2, w3};
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4)
                *reinterpret_cast<uint2*>(xdst[i][g4] >= 0 ? buf + xdst[i][g4] : dump) = wg1_u8x4_to_bf16(wr[g4]);
        }
#pragma unroll
        for (int i = 0; i < Cf::NAP; ++i) {
            *reinterpret_cast<wg_u32x4*>(ao[i] >= 0 ? buf + ao[i] : dump) = sa[i];
#pragma unroll
            for (int c = 0; c < 4; ++c) {                           // bias gradient: this chunk holds channels 8 (tid % 4) .. + 7
                bsum[2 * c] += __uint_as_float(sa[i][c] << 16);
                bsum[2 * c + 1] += __uint_as_float(sa[i][c] & 0xffff0000u);
            }
        }
    };
    auto compute = [&](uint32_t cur) __attribute__((always_inline)) {
        asm volatile("" : "+v"(cur));                               // see k_conv_wgrad_halo
#pragma unroll
        for (int ks = 0; ks < Cf::KSTEPS; ++ks) {
            bf16x8 fa[2], fb[2];
            const uint32_t a0 = cur + Cf::FRAME_X + ks * 1024;
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const wg_s16x4 lo = wg_tr_read(a0 + m * Cf::PLANE_A + vA[0]), hi = wg_tr_read(a0 + m * Cf::PLANE_A + vA[1]);
                fa[m] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            }
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const wg_s16x4 lo = wg_tr_read(cur + woff[t] + vB[ks][0]), hi = wg_tr_read(cur + woff[t] + vB[ks][1]);
                fb[t] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int m = 0; m < 2; ++m) asm volatile("" : "+v"(fa[m]));
#pragma unroll
            for (int t = 0; t < 2; ++t) asm volatile("" : "+v"(fb[t]));
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int m = 0; m < 2; ++m)
                    acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(wg_mfma_bf16x8, fa[m]),
                                                                        __builtin_bit_cast(wg_mfma_bf16x8, fb[t]), acc[m][t], 0, 0, 0);
        }
    };

    wg_u32x4 sx0[Cf::NXP], sa0[Cf::NAP], sx1[Cf::NXP], sa1[Cf::NAP];
    const int G_ = gridDim.x;
    int u = blockIdx.x;
    fetch(sx0, sa0, u);
    fetch(sx1, sa1, u + G_);
    __syncthreads();                                                // the zero fill is complete
    stash(buf0, sx0, sa0);
    __syncthreads();
    for (; u < nunits; u += 2 * G_) {
        fetch(sx0, sa0, u + 2 * G_);
        compute(lds0);
        stash(buf1, sx1, sa1);
        __syncthreads();
        fetch(sx1, sa1, u + 3 * G_);
        if (u + G_ < nunits) compute(lds0 + Cf::BUF);
        stash(buf0, sx0, sa0);
        __syncthreads();
    }

    float* out = ws + (long long)blockIdx.x * Cf::PART + (wave * 2 * 2 * 4) * 64 + lane;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) out[((m * 2 + t) * 4 + j) * 64] = acc[m][t][j];
    float* red = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int c = 0; c < 8; ++c) red[c * WG_T + tid] = bsum[c];
    __syncthreads();
    if (tid < 32) {
        const int blk = tid >> 3, c = tid & 7;                      // channel 8 blk + c: threads with tid % 4 == blk
        float s = 0.f;
        for (int k = 0; k < WG_T / 4; ++k) s += red[c * WG_T + 4 * k + blk];
        atomicAdd(db + tid, s);
    }
}

// slot = (((wave * 2 + m) * 2 + t) * 4 + j) * 64 + lane -> co = 16 m + 4 (lane / 16) + j; n-tile 2 wave + t = 4 (2 a + b) + c,
// column lane % 16 = 4 r + tt -> dw[co][c][4 a + r][4 b + tt]. grid = (PART / 256, row groups).
__global__ __launch_bounds__(256) void k_conv1_wgrad_reduce(const float* __restrict__ ws, float* __restrict__ dw, int rows) {
    const int slot = blockIdx.x * 256 + threadIdx.x, lane = slot & 63;
    int q = slot >> 6;
    const int j = q & 3; q >>= 2;
    const int t = q & 1; q >>= 1;
    const int m = q & 1, wave = q >> 1;
    const int nt = 2 * wave + t, tap = nt >> 2, c = nt & 3, i = lane & 15;
    const int r0 = (int)((long long)rows * blockIdx.y / gridDim.y), r1 = (int)((long long)rows * (blockIdx.y + 1) / gridDim.y);
    float s0 = 0.f, s1 = 0.f;
    int r = r0;
    for (; r + 2 <= r1; r += 2) {
        s0 += ws[(long long)r * Wg1::PART + slot];
        s1 += ws[(long long)(r + 1) * Wg1::PART + slot];
    }
    for (; r < r1; ++r) s0 += ws[(long long)r * Wg1::PART + slot];
    const int co = 16 * m + 4 * (lane >> 4) + j, kh = 4 * (tap >> 1) + (i >> 2), kw = 4 * (tap & 1) + (i & 3);
    atomicAdd(dw + co * 256 + c * 64 + kh * 8 + kw, (s0 + s1) * (1.0f / 255.0f));
}

// floats of workspace the two layers need (rows of fragment-ordered accumulators, one per workgroup)
DRL_API long long drl_conv_wgrad_halo_ws_floats(int layer) {
    if (layer == 2) return (long long)WG_MAX_GRID * WgCfg<20, 32, 4, 2, 1>::PART;
    if (layer == 3) return (long long)WG_MAX_GRID * WgCfg<9, 64, 3, 1, 2>::PART;
    if (layer == 1) return (long long)WG_MAX_GRID * Wg1::PART;
    return -1;
}

// conv1: store = replay plane store (u8), frame_off[frames] = byte offset of each frame's 4 consecutive 84 x 84 planes,
// dy [frames][20][20][32] bf16 -> dw [32][4][8][8] (+=, incl. the 1/255 of the forward), db [32] (+=). ws: ..._ws_floats(1).
DRL_API int drl_conv1_wgrad_halo(const void* store, const long long* frame_off, const void* dy, float* dw, float* db, float* ws,
                                 int frames, int max_wgs, hipStream_t s) {
    if (frames <= 0 || !store || !frame_off || !dy || !dw || !db || !ws) return -1;
    constexpr int LDS = 2 * Wg1::BUF + WG_T * 16;
    static_assert(LDS <= 160 * 1024 && LDS >= 8 * WG_T * 4 && Wg1::PART % 256 == 0, "two unit buffers + dump");
    static bool once = ((void)hipFuncSetAttribute((const void*)k_conv1_wgrad_halo, hipFuncAttributeMaxDynamicSharedMemorySize, LDS), true);
    (void)once;
    if (max_wgs <= 0 || max_wgs > WG_MAX_GRID) max_wgs = WG_MAX_GRID;
    const int nunits = 2 * frames, grid = nunits < max_wgs ? nunits : max_wgs;
    k_conv1_wgrad_halo<<<grid, WG_T, LDS, s>>>((const uint8_t*)store, frame_off, (const bf16_t*)dy, ws, db, frames);
    const int groups = grid >= 64 ? 8 : (grid >= 8 ? 2 : 1);
    k_conv1_wgrad_reduce<<<dim3(Wg1::PART / 256, groups), 256, 0, s>>>(ws, dw, grid);
    return (int)hipGetLastError();
}

// layer 2: x [frames][20][20][32], dy [frames][9][9][64] -> dw [64][4][4][32] (+=), db [64] (+=)
// layer 3: x [frames][9][9][64],   dy [frames][7][7][64] -> dw [64][3][3][64] (+=), db [64] (+=)        (all accumulate)
// ws: drl_conv_wgrad_halo_ws_floats(layer) floats of scratch, overwritten. max_wgs: workgroups (= CUs) to use, 0 = 256.
DRL_API int drl_conv_wgrad_halo(int layer, const void* x, const void* dy, float* dw, float* db, float* ws, int frames, int max_wgs,
                                hipStream_t s) {
    if (frames <= 0 || !x || !dy || !dw || !db || !ws) return -1;
    if (layer == 2) return launch_wgrad<WgCfg<20, 32, 4, 2, 1>>(x, dy, dw, db, ws, frames, max_wgs, s);
    if (layer == 3) return launch_wgrad<WgCfg<9, 64, 3, 1, 2>>(x, dy, dw, db, ws, frames, max_wgs, s);
    return -1;
}

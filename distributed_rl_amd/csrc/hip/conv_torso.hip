// Fused forward of the Nature-CNN torso on gfx950 (SURVEY G1-G3 as ONE kernel; VERDICT r2 "do this" #2):
//   u8 planes / stacks -> conv1 + ReLU -> conv2 + ReLU -> conv3 + ReLU -> act3 [n, 7, 7, 64] bf16,
// with act1 (25.6 KB / frame) and act2 (10.4 KB / frame) living in LDS. They are written to HBM only for the frames a
// backward pass will read (a per-network frame range): the Ape-X learner needs them for B of its 3 B forwards, R2D2 for the
// learning window of the online network only.
//
// Why one kernel: at the learner's batch (1536 frame-forwards on 256 CUs = 6 frames per CU) the three per-layer launches
// are fixed cost: 27 + 17.4 + 15.9 us where their steady-state rates (R2D2, 60 frames per CU) would need 18 us. Each launch
// pays its own ramp (weights -> registers, first frame group through an empty pipeline) and tail, and layer l + 1 cannot
// start before the slowest workgroup of layer l has finished.
//
// Structure: persistent 512-thread workgroups, one per CU, one FRAME per round, three roles (the two waves of a SIMD get
// complementary work - MI355X_MICROARCH "two waves per SIMD" item 9), a three-stage pipeline over frames:
//   waves 0-3 "P":  conv1 of frame j + 1 (biased-f16 trick, all 32 x 256 weights in 64 VGPRs) out of the raw frame that an
//                LDS-DMA staged one round earlier; the 16-byte result goes to the LDS image conv2 reads. Tiles 0..23 of a
//                frame are dealt round-robin to the four waves (6 slots each), tile 24 is wave 2's seventh slot.
//   waves 4-5 "C2": conv2 of frame j (fragments by ds_read_b128 with the tap as the instruction's immediate, weights in
//                registers, conflict-free padded images as in conv_halo.hip), each wave owning 32 of the 64 output channels:
//                every fragment read feeds TWO MFMAs.
//   waves 6-7 "C3": conv3 of frame j - 1, 32 channels per wave, out of the act2 image C2 wrote one round earlier.
// (Round 3, first form: waves 4-7 each did 16 channels of conv2 AND conv3 of the same frame - 64 + 72 weight VGPRs, two
// barriers per round, every fragment read by all four waves: the LDS pipe was active 53 % of the kernel's time, 0.67 MB
// of fragment reads per frame and CU. Splitting by LAYER halves those reads, needs one barrier per round and a second act2
// image.)
// Round j = 0 .. nf (one barrier each):  P: frame j + 1 -> image1[(j + 1) & 1]   C2: image1[j & 1] -> image2[j & 1]
//                                        C3: image2[(j - 1) & 1] -> act3 (HBM)
// P's LDS-DMAs and global stores stay in flight across the barriers (raw s_barrier behind an lgkmcnt-only wait).
// LDS: 2 x 32 320 (act1 images, pixel pitch 80 B, row pitch 101 slots) + 2 x 14 752 (act2 images, pixel pitch 160 B, row
// pitch 102 slots) + 2 x 28 672 (raw frames) = 151 488 B: the image layouts of conv_halo.hip (scripts/debug/halo_lds_layout.py).
#include "common.h"
#include "conv_frag.h"
#include <type_traits>

typedef __attribute__((ext_vector_type(8))) __bf16 torso_bf16x8;
__device__ __forceinline__ f32x4 torso_mfma(bf16x8 a, bf16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(torso_bf16x8, a), __builtin_bit_cast(torso_bf16x8, b), c, 0, 0, 0);
}

struct TorsoNet {
    const long long* frame_off;     // plane store: byte offset of every frame's first plane (null: dense HWC stacks)
    const bf16_t *w1, *w2, *w3;     // conv1 [32][256] (k order of the input form, conv_direct.hip), conv2 [64][4][4][32], conv3 [64][3][3][64]
    const float *b1, *b2, *b3;
    bf16_t *act1, *act2, *act3;     // [frames][20][20][32], [frames][9][9][64] (rows [save_lo, save_hi) written), [frames][7][7][64]
    int frames, save_lo, save_hi, nblk;
};

constexpr int T_P1 = 5, T_RP1 = 101, T_IMG1 = 2020 * 16;     // conv2's input image: slots of 16 bytes
constexpr int T_P2 = 10, T_RP2 = 102, T_IMG2 = 922 * 16;     // conv3's input image
constexpr int T_I2 = 2 * T_IMG1;                              // act2 images [2] behind the act1 images [2]
constexpr int T_RAW = T_I2 + 2 * T_IMG2, T_RAWSZ = 28672;     // two raw-frame buffers (28 224 bytes used) behind the images
constexpr int T_LDS = T_RAW + 2 * T_RAWSZ;

// every LDS access of this wave has completed, then the workgroup barrier; vector-memory operations stay in flight
__device__ __forceinline__ void torso_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
}

template <bool OFF>
__global__ __launch_bounds__(512, 1) void k_torso_fwd(const uint8_t* __restrict__ x, TorsoNet n1, TorsoNet n2, HaloPrep prep, int dbg) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];     // image1[0..1], image2[0..1], raw[0..1]
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 15, g = lane >> 4;
    if ((int)blockIdx.x >= n1.nblk + n2.nblk) {                         // the weight pre-pass workgroups (conv_frag.h)
        const int b = (int)blockIdx.x - n1.nblk - n2.nblk;
        for (int i = b * 512 + tid; i < HALO_PREP_ELEMS; i += prep.nblk * 512) halo_wprep_elem(prep, i);
        return;
    }
    const bool second = (int)blockIdx.x >= n1.nblk;                     // block-uniform
    const int frames = second ? n2.frames : n1.frames, nblk = second ? n2.nblk : n1.nblk;
    const int save_lo = second ? n2.save_lo : n1.save_lo, save_hi = second ? n2.save_hi : n1.save_hi;
    const int blk = second ? (int)blockIdx.x - n1.nblk : (int)blockIdx.x;
    const int nf = (frames - blk + nblk - 1) / nblk;                    // frames blk, blk + nblk, ...
    if (nf <= 0) return;                                                // block-uniform, before any barrier
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    if (wave < 4) {
        // ------------------------------------------------ P: conv1 ------------------------------------------------
        const long long* __restrict__ frame_off = second ? n2.frame_off : n1.frame_off;
        const bf16_t* __restrict__ w = second ? n2.w1 : n1.w1;
        const float* __restrict__ bias = second ? n2.b1 : n1.b1;
        bf16_t* __restrict__ act1 = second ? n2.act1 : n1.act1;
        uint4 wf[2][8];
        float acc0[2][4];                           // accumulator start: -1024 * (sum over k of this lane's channels' weights)
        {
            float csum[2] = {0.f, 0.f};
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int s = 0; s < 8; ++s)
                    wf[t][s] = bf16x8_to_f16(*reinterpret_cast<const bf16x8*>(w + ((r >> 2) * 8 + t * 4 + (r & 3)) * 256 + s * 32 + g * 8),
                                             csum[t]);
#pragma unroll
            for (int t = 0; t < 2; ++t) {           // lanes r, r+16, r+32, r+48 hold the four k-quarters of MFMA row r
                csum[t] += __shfl_xor(csum[t], 16, WAVE);
                csum[t] += __shfl_xor(csum[t], 32, WAVE);
#pragma unroll
                for (int i = 0; i < 4; ++i) acc0[t][i] = -1024.0f * __shfl(csum[t], g * 4 + i, WAVE);   // accumulator row g*4+i
            }
        }
        float4 b4[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) b4[t] = *reinterpret_cast<const float4*>(bias + g * 8 + t * 4);
        // slot j < 6 of this wave = output tile wave + 4 j; slot 6 = tile 24, computed by wave 2 only: byte offset of the
        // lane's first fragment inside a frame, of its 16-byte result inside an act1 image, and inside a frame of the act1 array
        int pofs[7], lofs[7];
#pragma unroll
        for (int j = 0; j < 7; ++j) {
            const int m = min(wave + 4 * j, 24) * 16 + r, oy = m / 20, ox = m - oy * 20;
            pofs[j] = OFF ? (4 * oy + g) * 84 + 4 * ox : oy * (4 * 336) + ox * 16 + g * 8;
            lofs[j] = (oy * T_RP1 + ox * T_P1 + g) * 16;
        }
        const int gofs0 = (wave * 16 + r) * 64 + g * 16, gofs6 = (24 * 16 + r) * 64 + g * 16;   // slot j: pixel m0 + 64 j; slot 6: tile 24
        auto koff = [](int s) -> int { return OFF ? (s >> 1) * 7056 + (s & 1) * 336 : s * 336; };
        auto frame_of = [&](int i) -> int { return blk + min(i, nf - 1) * nblk; };          // wave-uniform
        // the frame's base is a SCALAR load (constant address space: the list is read-only for the launch)
        typedef const long long __attribute__((address_space(4)))* const_ll_ptr;
        auto fbase = [&](int f) -> long long {
            if constexpr (OFF) return ((const_ll_ptr)frame_off)[f];
            else return (long long)f * 28224;
        };
        // The raw frame (4 planes / one HWC stack = 28 224 contiguous bytes) goes global -> LDS by LDS-DMA, 16 bytes per
        // lane, 7 instructions per P wave, one whole round before conv1 reads it: HBM is read once and coalesced (the
        // direct global -> VGPR form of conv_direct.hip reads every byte 4 times through L1 and needs 16 VGPRs per tile in
        // flight), and no register waits for memory. Chunks past the frame's end re-read its last chunk.
        typedef const __attribute__((address_space(1))) void* t_gptr;
        typedef __attribute__((address_space(3))) void* t_lptr;
        auto dma = [&](long long fb, unsigned raw) __attribute__((always_inline)) {
            const uint8_t* __restrict__ fp = x + fb;
#pragma unroll
            for (int i = 0; i < 7; ++i) {
                const int chunk0 = (i * 4 + wave) * 64;                                  // wave-uniform
                const void* g_ = fp + (unsigned)min(chunk0 + lane, 1763) * 16u;
                __builtin_amdgcn_global_load_lds((t_gptr)g_, (t_lptr)(smem + raw + chunk0 * 16), 16, 0, 0);
            }
        };
        // LDS accesses of the P waves are INLINE ASM on purpose: hipcc (ROCm 7.2) orders every ds instruction it can see
        // behind the pending LDS-DMAs with an s_waitcnt vmcnt(0) - the DMA of the NEXT frame would be drained right after
        // its issue. The compiler does not count these reads: every use is behind an explicit s_waitcnt lgkmcnt(0).
        typedef uint32_t t_u32x2 __attribute__((ext_vector_type(2)));
        typedef uint32_t t_u32x4 __attribute__((ext_vector_type(4)));
        // fragments of one slot (8 k-steps x 8 bytes per lane). Plane store: k-step s = plane s / 2, patch rows 4 (s % 2) ..
        // (4-byte aligned: two dwords per read); dense stacks: patch row s (8-byte aligned)
        auto rd_slot = [&](t_u32x2 (&f)[8], unsigned raw, int j) __attribute__((always_inline)) {
            if constexpr (OFF) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const unsigned a_ = raw + (unsigned)pofs[j] + (unsigned)c * 7056u;
                    asm volatile("ds_read2_b32 %0, %1 offset1:1" : "=v"(f[2 * c]) : "v"(a_));
                    asm volatile("ds_read2_b32 %0, %1 offset0:84 offset1:85" : "=v"(f[2 * c + 1]) : "v"(a_));
                }
            } else {
                const unsigned a_ = raw + (unsigned)pofs[j];
#pragma unroll
                for (int s_ = 0; s_ < 8; ++s_) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(f[s_]) : "v"(a_), "n"(s_ * 336));
            }
        };
        // one group = N slots (j0 ..) at once: 2 N independent accumulator chains (an MFMA result is ready ~36 cycles after
        // issue: one slot alone, 2 chains of 8 dependent MFMAs, runs at half the pipe's rate). Then bias + ReLU, 16 bytes into
        // the image (SAVE: and into HBM, where a backward will read them; a template parameter, not a branch per slot).
        auto compute = [&](auto n_tag, auto save_tag, t_u32x2 (&fr)[2][8], int j0, unsigned img, int frame) __attribute__((always_inline)) {
            constexpr int N = decltype(n_tag)::value;
            constexpr bool SAVE = decltype(save_tag)::value;
            f32x4 acc[N][2];
#pragma unroll
            for (int q = 0; q < N; ++q)
#pragma unroll
                for (int t = 0; t < 2; ++t) acc[q][t] = f32x4{acc0[t][0], acc0[t][1], acc0[t][2], acc0[t][3]};
#pragma unroll
            for (int s_ = 0; s_ < 8; ++s_) {
#pragma unroll
                for (int q = 0; q < N; ++q) {
                    const uint4 xf = u8x8_to_f16_biased(make_uint2(fr[q][s_][0], fr[q][s_][1]));
                    acc[q][0] = mfma16h(wf[0][s_], xf, acc[q][0]);
                    acc[q][1] = mfma16h(wf[1][s_], xf, acc[q][1]);
                }
            }
            constexpr float SC = 1.0f / 255.0f;          // pixel normalisation x/255 folded into the epilogue
#pragma unroll
            for (int q = 0; q < N; ++q) {
                const int j = j0 + q;
                t_u32x4 ov;
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    ov[2 * t] = pack_bf2(fmaxf(acc[q][t][0] * SC + b4[t].x, 0.f), fmaxf(acc[q][t][1] * SC + b4[t].y, 0.f));
                    ov[2 * t + 1] = pack_bf2(fmaxf(acc[q][t][2] * SC + b4[t].z, 0.f), fmaxf(acc[q][t][3] * SC + b4[t].w, 0.f));
                }
                const unsigned la = img + (unsigned)lofs[j];
                asm volatile("ds_write_b128 %0, %1" : : "v"(la), "v"(ov) : "memory");
                if constexpr (SAVE)
                    *reinterpret_cast<t_u32x4*>(reinterpret_cast<uint8_t*>(act1) + (long long)frame * 25600 +
                                                (unsigned)(j < 6 ? gofs0 + 4096 * j : gofs6)) = ov;
            }
        };
        using one_t = std::integral_constant<int, 1>;
        using two_t = std::integral_constant<int, 2>;
        t_u32x2 fa[2][8], fb_[2][8];                     // fragment registers of two groups: one computing, one being read
        // the reads of group f have landed: the wait takes the group's registers as operands, so no use of them can be
        // scheduled in front of it (the compiler does not know what the ds_read asm statements leave in flight)
        auto land = [](t_u32x2 (&f)[2][8]) __attribute__((always_inline)) {
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(f[0][0]), "+v"(f[0][1]), "+v"(f[0][2]), "+v"(f[0][3]), "+v"(f[0][4]), "+v"(f[0][5]), "+v"(f[0][6]), "+v"(f[0][7]),
                           "+v"(f[1][0]), "+v"(f[1][1]), "+v"(f[1][2]), "+v"(f[1][3]), "+v"(f[1][4]), "+v"(f[1][5]), "+v"(f[1][6]), "+v"(f[1][7])
                         :: "memory");
        };
        // groups of a frame: (0, 1) (2, 3) (4, 5) and, on wave 2, (6). The reads of a group are issued before the group in
        // front of it is computed.
        const bool tail_slot = wave == 2;                // wave-uniform
        auto conv1_frame = [&](auto save_tag, unsigned raw, unsigned img, int f) __attribute__((always_inline)) {
            rd_slot(fa[0], raw, 0); rd_slot(fa[1], raw, 1);
            land(fa);
            rd_slot(fb_[0], raw, 2); rd_slot(fb_[1], raw, 3);
            compute(two_t{}, save_tag, fa, 0, img, f);
            land(fb_);
            rd_slot(fa[0], raw, 4); rd_slot(fa[1], raw, 5);
            compute(two_t{}, save_tag, fb_, 2, img, f);
            land(fa);
            if (tail_slot) rd_slot(fb_[0], raw, 6);
            compute(two_t{}, save_tag, fa, 4, img, f);
            if (tail_slot) {
                land(fb_);
                compute(one_t{}, save_tag, fb_, 6, img, f);
            }
        };
        // land weights and biases BEFORE the first DMA is issued (their first use would otherwise wait behind the DMAs)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
            for (int s_ = 0; s_ < 8; ++s_) asm volatile("" : "+v"(wf[t][s_].x), "+v"(wf[t][s_].y), "+v"(wf[t][s_].z), "+v"(wf[t][s_].w));
            asm volatile("" : "+v"(b4[t].x), "+v"(b4[t].y), "+v"(b4[t].z), "+v"(b4[t].w));
            asm volatile("" : "+v"(acc0[t][0]), "+v"(acc0[t][1]), "+v"(acc0[t][2]), "+v"(acc0[t][3]));
        }
        dma(fbase(frame_of(0)), T_RAW);
        dma(fbase(frame_of(1)), T_RAW + T_RAWSZ);
        asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
        torso_barrier();                                 // frame 0 is in raw[0]
        {                                                // frame 0 -> image1[0], before the first round
            const int f = frame_of(0);
            if (f >= save_lo && f < save_hi) conv1_frame(std::true_type{}, T_RAW, 0u, f);
            else conv1_frame(std::false_type{}, T_RAW, 0u, f);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        torso_barrier();                                 // frame 1 is in raw[1], frame 0 in image1[0]
        for (int j = 0; j <= nf; ++j) {
            if (j + 1 < nf) {                            // wave-uniform: frame j + 1 exists
                const int f = blk + (j + 1) * nblk;
                const unsigned img = ((j + 1) & 1) * T_IMG1, raw = T_RAW + ((j + 1) & 1) * T_RAWSZ;
                const bool save = f >= save_lo && f < save_hi;      // wave-uniform
                dma(fbase(frame_of(j + 2)), T_RAW + (j & 1) * T_RAWSZ);   // raw[j & 1] held frame j: read in round j - 1
                if (!(dbg & 1)) {
                    if (save) conv1_frame(std::true_type{}, raw, img, f);
                    else conv1_frame(std::false_type{}, raw, img, f);
                }
                if (save && !(dbg & 1)) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(6) : "memory");   // 6 act1 stores may stay in flight
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            torso_barrier();
        }
    } else if (wave < 6) {
        // ------------------------------------- C2: conv2, 32 channels per wave ------------------------------------
        const int h = wave - 4;                                          // output channels [32 h, 32 h + 32)
        const bf16_t* __restrict__ w2 = second ? n2.w2 : n1.w2;
        const float* __restrict__ bias2 = second ? n2.b2 : n1.b2;
        bf16_t* __restrict__ act2 = second ? n2.act2 : n1.act2;
        bf16x8 wf2[2][16];                          // A operand: lane supplies W[channel 32 h + 16 t + l % 16][k-step s, k = (l / 16) * 8 ..]
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int s = 0; s < 16; ++s)
                wf2[t][s] = *reinterpret_cast<const bf16x8*>(w2 + (size_t)(32 * h + 16 * t + r) * 512 + s * 32 + g * 8);
        // the lane ends up with channels 32 h + 16 t + 4 g .. + 3 of pixel r of every tile
        float4 bb2[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) bb2[t] = *reinterpret_cast<const float4*>(bias2 + 32 * h + 16 * t + 4 * g);
        // rows past the last pixel are computed from the last pixel's inputs and store the same values to the same place
        unsigned xb2[6], wo2[6];                    // (the HBM offset of a tile's result is recomputed where it is stored: registers)
#pragma unroll
        for (int u = 0; u < 6; ++u) {
            const int m = min(u * 16 + r, 80), oy = m / 9, ox = m - oy * 9;
            xb2[u] = (unsigned)((2 * oy) * T_RP1 + (2 * ox) * T_P1 + g) * 16u;
            wo2[u] = (unsigned)T_I2 + (unsigned)(oy * T_RP2 + ox * T_P2) * 16u + (unsigned)(64 * h + 8 * g);
        }
        const unsigned go2b = (unsigned)(64 * h + 8 * g);
        // land weights and biases once, here (left to the compiler the waits sit inside the round loop)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
            for (int s = 0; s < 16; ++s) asm volatile("" : "+v"(wf2[t][s]));
            asm volatile("" : "+v"(bb2[t].x), "+v"(bb2[t].y), "+v"(bb2[t].z), "+v"(bb2[t].w));
        }
        // (no s_setprio: C at priority 1 / P at 2 / both / neither measured 148.0 / 144.9 / 145.7 / 143.4 us on the R2D2 chunk)
        torso_barrier();                            // (P: frame 0 is in raw[0])
        torso_barrier();                            // frame 0 is in image1[0]
        for (int j = 0; j <= nf; ++j) {
            if (j < nf && !(dbg & 2)) {
                // ---- conv2 of frame j: 6 tiles x 16 k-steps (k-step = one tap of the 4x4 kernel x 32 channels) ----
                const int frame = blk + j * nblk;
                const unsigned cur = (j & 1) * T_IMG1, out = (j & 1) * T_IMG2;
                f32x4 acc[2][6];
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int u = 0; u < 6; ++u) acc[t][u] = f32x4{0.f, 0.f, 0.f, 0.f};
                bf16x8 xf[6];
#pragma unroll
                for (int u = 0; u < 6; ++u) xf[u] = *reinterpret_cast<const bf16x8*>(smem + cur + xb2[u]);
                __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
#pragma unroll
                for (int s = 0; s < 16; ++s) {
                    const int s1 = s + 1, off = ((s1 >> 2) * T_RP1 + (s1 & 3) * T_P1) * 16;   // immediate offset of the ds_read
#pragma unroll
                    for (int u = 0; u < 6; ++u) {
                        acc[0][u] = torso_mfma(wf2[0][s], xf[u], acc[0][u]);
                        acc[1][u] = torso_mfma(wf2[1][s], xf[u], acc[1][u]);
                        // the fragment of k-step s + 1 replaces the one just consumed: one ds read behind every pair of MFMAs
                        if (s1 < 16) xf[u] = *reinterpret_cast<const bf16x8*>(smem + cur + xb2[u] + off);
                        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                        if (s1 < 16) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    }
                }
                const bool save = frame >= save_lo && frame < save_hi;              // wave-uniform, ONE branch per frame
                uint8_t* __restrict__ dst = reinterpret_cast<uint8_t*>(act2) + (long long)frame * 10368;
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int u = 0; u < 6; ++u) {
                        const uint2 o = make_uint2(pack_bf2(fmaxf(acc[t][u][0] + bb2[t].x, 0.f), fmaxf(acc[t][u][1] + bb2[t].y, 0.f)),
                                                   pack_bf2(fmaxf(acc[t][u][2] + bb2[t].z, 0.f), fmaxf(acc[t][u][3] + bb2[t].w, 0.f)));
                        *reinterpret_cast<uint2*>(smem + out + wo2[u] + 32 * t) = o;
                        if (save) *reinterpret_cast<uint2*>(dst + (unsigned)min(u * 16 + r, 80) * 128u + go2b + 32 * t) = o;
                    }
            }
            torso_barrier();
        }
    } else {
        // ------------------------------------- C3: conv3, 32 channels per wave ------------------------------------
        const int h = wave - 6;
        const bf16_t* __restrict__ w3 = second ? n2.w3 : n1.w3;
        const float* __restrict__ bias3 = second ? n2.b3 : n1.b3;
        bf16_t* __restrict__ act3 = second ? n2.act3 : n1.act3;
        bf16x8 wf3[2][18];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int s = 0; s < 18; ++s)
                wf3[t][s] = *reinterpret_cast<const bf16x8*>(w3 + (size_t)(32 * h + 16 * t + r) * 576 + s * 32 + g * 8);
        float4 bb3[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) bb3[t] = *reinterpret_cast<const float4*>(bias3 + 32 * h + 16 * t + 4 * g);
        unsigned xb3[4], go3[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int m = min(u * 16 + r, 48), oy = m / 7, ox = m - oy * 7;
            xb3[u] = (unsigned)T_I2 + (unsigned)(oy * T_RP2 + ox * T_P2 + g) * 16u;
            go3[u] = (unsigned)(m * 128 + 64 * h + 8 * g);
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
            for (int s = 0; s < 18; ++s) asm volatile("" : "+v"(wf3[t][s]));
            asm volatile("" : "+v"(bb3[t].x), "+v"(bb3[t].y), "+v"(bb3[t].z), "+v"(bb3[t].w));
        }
        torso_barrier();
        torso_barrier();
        for (int j = 0; j <= nf; ++j) {
            if (j >= 1 && !(dbg & 4)) {
                // ---- conv3 of frame j - 1: 4 tiles x 18 k-steps (k-step = half the 64 channels of one tap of the 3x3 kernel) ----
                const int frame = blk + (j - 1) * nblk;
                const unsigned cur = ((j - 1) & 1) * T_IMG2;
                f32x4 acc[2][4];
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int u = 0; u < 4; ++u) acc[t][u] = f32x4{0.f, 0.f, 0.f, 0.f};
                bf16x8 xf[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) xf[u] = *reinterpret_cast<const bf16x8*>(smem + cur + xb3[u]);
                __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
                for (int s = 0; s < 18; ++s) {
                    const int s1 = s + 1, tap = s1 >> 1, ky = tap / 3, kx = tap - ky * 3;
                    const int off = (ky * T_RP2 + kx * T_P2 + (s1 & 1) * 4) * 16;
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        acc[0][u] = torso_mfma(wf3[0][s], xf[u], acc[0][u]);
                        acc[1][u] = torso_mfma(wf3[1][s], xf[u], acc[1][u]);
                        if (s1 < 18) xf[u] = *reinterpret_cast<const bf16x8*>(smem + cur + xb3[u] + off);
                        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                        if (s1 < 18) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    }
                }
                uint8_t* __restrict__ dst = reinterpret_cast<uint8_t*>(act3) + (long long)frame * 6272;
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const uint2 o = make_uint2(pack_bf2(fmaxf(acc[t][u][0] + bb3[t].x, 0.f), fmaxf(acc[t][u][1] + bb3[t].y, 0.f)),
                                                   pack_bf2(fmaxf(acc[t][u][2] + bb3[t].z, 0.f), fmaxf(acc[t][u][3] + bb3[t].w, 0.f)));
                        *reinterpret_cast<uint2*>(dst + go3[u] + 32 * t) = o;
                    }
            }
            torso_barrier();
        }
    }
}

// CUs that a concurrently running persistent kernel of another stream occupies (conv_halo.hip: drl_halo_set_busy). A torso
// workgroup fills a CU's registers: it cannot share one, so the grid leaves those CUs out.
static int g_torso_busy_cus = 0, g_torso_dbg = 0;
DRL_API int drl_torso_set_debug(int d) { g_torso_dbg = d; return 0; }
DRL_API int drl_torso_set_busy(int cus) {
    if (cus < 0 || cus > 192) return -1;
    g_torso_busy_cus = cus;
    return 0;
}

// One launch: the torso of network 1 over frames1 frames and (frames2 > 0) of network 2 over frames2 frames of the same
// store x. off = null: dense HWC u8 stacks [n, 84, 84, 4] (conv1 weights in HWC order); else byte offsets into a plane store
// (conv1 weights in (c, kh, kw) order). save range [lo, hi): frames whose act1 / act2 are also written to HBM.
// prep_wt2 != null: extra workgroups write the flipped / transposed conv2 / conv3 weights of the input-gradient kernels.
DRL_API int drl_torso_fwd(const void* x,
                          const long long* off1, int frames1, const void* const* w1, const float* const* b1, void* const* act1,
                          int save_lo1, int save_hi1,
                          const long long* off2, int frames2, const void* const* w2, const float* const* b2, void* const* act2,
                          int save_lo2, int save_hi2,
                          const void* prep_w2, const void* prep_w3, void* prep_wt2, void* prep_wt3, hipStream_t s) {
    if (frames1 <= 0 || frames2 < 0 || !x) return -1;
    if ((off1 == nullptr) != (frames2 > 0 ? off2 == nullptr : off1 == nullptr)) return -1;      // one input form per launch
    auto net = [](const long long* off, int frames, const void* const* w, const float* const* b, void* const* a, int lo, int hi) {
        TorsoNet n{};
        if (frames <= 0) return n;
        n.frame_off = off;
        n.w1 = (const bf16_t*)w[0]; n.w2 = (const bf16_t*)w[1]; n.w3 = (const bf16_t*)w[2];
        n.b1 = b[0]; n.b2 = b[1]; n.b3 = b[2];
        n.act1 = (bf16_t*)a[0]; n.act2 = (bf16_t*)a[1]; n.act3 = (bf16_t*)a[2];
        n.frames = frames;
        n.save_lo = lo < 0 ? 0 : lo;
        n.save_hi = hi > frames ? frames : hi;
        if (!n.act1 || !n.act2) n.save_lo = n.save_hi = 0;
        return n;
    };
    TorsoNet n1 = net(off1, frames1, w1, b1, act1, save_lo1, save_hi1);
    TorsoNet n2 = net(off2, frames2, frames2 > 0 ? w2 : nullptr, frames2 > 0 ? b2 : nullptr, frames2 > 0 ? act2 : nullptr, save_lo2, save_hi2);
    if (!n1.w1 || !n1.w2 || !n1.w3 || !n1.b1 || !n1.b2 || !n1.b3 || !n1.act3) return -1;
    if (frames2 > 0 && (!n2.w1 || !n2.w2 || !n2.w3 || !n2.b1 || !n2.b2 || !n2.b3 || !n2.act3)) return -1;
    HaloPrep prep{(const bf16_t*)prep_w2, (const bf16_t*)prep_w3, (bf16_t*)prep_wt2, (bf16_t*)prep_wt3, prep_wt2 ? HALO_PREP_BLOCKS / 2 : 0};
    if (prep.nblk && (!prep_w2 || !prep_w3 || !prep_wt3)) return -1;
    // persistent workgroups, one per free CU, dealt to the two networks in proportion to their frames; equal rounds per
    // workgroup of a network (ceil(frames / rounds) workgroups do the same work in the same number of rounds)
    const int slots = 256 - g_torso_busy_cus;
    int b1n, b2n = 0;
    if (frames1 + frames2 <= slots) { b1n = frames1; b2n = frames2; }
    else if (frames2 == 0) b1n = slots;
    else {
        b1n = (int)((long long)slots * frames1 / (frames1 + frames2));
        b1n = b1n < 1 ? 1 : (b1n > slots - 1 ? slots - 1 : b1n);
        b2n = slots - b1n;
    }
    auto even = [](int frames, int blocks) { const int rounds = (frames + blocks - 1) / blocks; return (frames + rounds - 1) / rounds; };
    n1.nblk = even(frames1, b1n);
    n2.nblk = frames2 > 0 ? even(frames2, b2n) : 0;
    const int grid = n1.nblk + n2.nblk + prep.nblk;
    if (off1) {
        static bool once = ((void)hipFuncSetAttribute((const void*)k_torso_fwd<true>, hipFuncAttributeMaxDynamicSharedMemorySize, T_LDS), true);
        (void)once;
        k_torso_fwd<true><<<grid, 512, T_LDS, s>>>((const uint8_t*)x, n1, n2, prep, g_torso_dbg);
    } else {
        static bool once = ((void)hipFuncSetAttribute((const void*)k_torso_fwd<false>, hipFuncAttributeMaxDynamicSharedMemorySize, T_LDS), true);
        (void)once;
        k_torso_fwd<false><<<grid, 512, T_LDS, s>>>((const uint8_t*)x, n1, n2, prep, g_torso_dbg);
    }
    return (int)hipGetLastError();
}

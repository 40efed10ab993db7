// Large-tile bf16 MFMA GEMM for gfx950: the three big dense products of the R2D2 step (reference
// libs/models.py:99 nn.LSTMCell input projection and its backward), which round 1 left to hipBLASLt:
//
//   NT  xg[n, 2048]      = feat[n, 3136] . W_ih[2048, 3136]^T + (b_ih + b_hh)     n = 3072 .. 7680   (fp32 out, bias fused)
//   NN  dfeat[n, 3136]   = relu'(feat) * (dG[n, 2048] . W_ih[2048, 3136])        (bf16 out, ReLU mask fused)
//   TN  dW_ih[2048,3136] = dG[5120, 2048]^T . feat[5120, 3136]                    (fp32 out, contraction over rows)
//
// gemm.hip's tiles are 4 waves and at most 128 x 128: at these sizes they leave half of the machine's LDS bandwidth
// and register file unused (154 / 159 / 147 us against 93 / 102 / 95 us for the library). This kernel is the
// wave64-first form the hardware wants for a big plain product:
//   * 8 waves (512 threads, 2 per SIMD), tile 256 x 256 (wave tile 128 x 64, 128 accumulator registers per lane) or
//     256 x 128 / 128 x 256 / 128 x 128 when the grid would otherwise not fill 256 CUs;
//   * both operands go global -> LDS by LDS-DMA (global_load_lds_dwordx4, no register staging), NS stages of BK = 64,
//     a COUNTED s_waitcnt vmcnt(N) + ONE raw s_barrier per K-tile, so NS - 1 tiles stay in flight across the barrier
//     (guide: "Pipelining across barriers"); all LDS in one dynamic array;
//   * the LDS image is linear in the DMA's lane order; the XOR swizzle that makes ds_read_b128 (row operands) and
//     ds_read_b64_tr_b16 (operands whose contraction index is the memory row: NN's B, both of TN's) conflict-free is
//     applied to the per-lane SOURCE address and to the read address (guide rule 21);
//   * operands are passed swapped to v_mfma_f32_16x16x32_bf16, so a lane holds 4 consecutive output columns: 16-byte
//     fp32 / 8-byte bf16 stores, vector loads of bias and ReLU mask; s_setprio around the MFMA cluster;
//   * out-of-range rows / columns / k are fetched from a zeroed page: branch-free, constant DMA count per tile;
//   * XCD-aware tile order (8 XCDs, private L2s), bijective for any grid.
#include "common.h"

typedef __attribute__((ext_vector_type(8))) __bf16 big_bf16x8;
typedef __attribute__((ext_vector_type(4))) short big_s16x4;
typedef __attribute__((address_space(1))) const void* big_gptr;
typedef __attribute__((address_space(3))) void* big_lptr;

#define BIG_T 512
#define BIG_BK 64

__device__ uint4 g_big_zero[4];            // 64 zero bytes: the source of every masked DMA

struct BigEpi {
    void* C;
    long long ldc;
    const float* bias;          // [N] or null
    const bf16_t* mask;         // [M][ldmask] forward activation (ReLU backward: out = 0 where mask <= 0) or null
    long long ldmask;
    int out_f32;                // bit 0: C is fp32 (else bf16); bit 1: ReLU after the bias
    // second problem of a two-problem launch (blockIdx.z == 1: same shapes, other operands - e.g. the online and the
    // target network's input projection, which fill the 256 CUs together where each alone would not)
    void* C2;
    const float* bias2;
    const bf16_t* mask2;
};

// row operand tile [R][64] bf16 (128-B rows): 16-B chunk c of row r lives at chunk c ^ ((r >> 1) & 7)
template <int R>
__device__ __forceinline__ int big_nrm_off(int r, int c) { return r * 128 + ((c ^ ((r >> 1) & 7)) << 4); }
// transposed-read tile [64][R] bf16 (2R-byte rows of U = R/16 units of 32 B): unit u of row k lives at u ^ swz(k).
// One ds_read_b64_tr_b16 takes, per 32-lane half, the rows 8g + q (g in {0,1} or {2,3}, q = 0..3; the second read of a
// fragment the same + 4): they differ in bits 0, 1 and 3 of k. With 16 units per row (512-B rows) k & 15 keeps the 8 rows
// on 8 different 32-B bank groups; with 8 units (256-B rows = exactly the 64 banks) bit 3 must be folded into bit 2, or
// rows k and k + 8 collide (2-way conflict on every transposed read of a 128-wide tile).
template <int U>
__device__ __forceinline__ int big_tr_swz(int k) {
    if constexpr (U >= 16) return k & 15;
    else return ((k & 3) | ((k >> 1) & 4)) & (U - 1);
}
template <int R>
__device__ __forceinline__ int big_tr_off(int k, int unit) {
    constexpr int U = R / 16;
    return k * (2 * R) + (((unit ^ big_tr_swz<U>(k)) & (U - 1)) << 5);
}

// One operand: DMA issue (source-side swizzle) and fragment reads.
//   TR = false: memory [rows = out index][cols = k], ld elements per row
//   TR = true : memory [rows = k][cols = out index]
template <int OUT, bool TR>
struct BigOperand {
    static constexpr int ROWS = TR ? BIG_BK : OUT, CPR = (TR ? OUT : BIG_BK) / 8;     // 16-B chunks per LDS row
    static constexpr int NCH = ROWS * CPR / BIG_T, BYTES = ROWS * CPR * 16;
    static_assert(ROWS * CPR % BIG_T == 0 && NCH >= 1, "tile too small for 512 threads");
    const bf16_t* p;
    long long ld;
    int n_out, K;
    const bf16_t* p2;           // blockIdx.z == 1
    __device__ __forceinline__ void issue(int out0, int k0, int tid, int wave, uint8_t* lds, bool active) const {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int id = tid + i * BIG_T, r = id / CPR, s = id % CPR;
            const bf16_t* src;
            bool v;
            if constexpr (TR) {
                constexpr int U = OUT / 16;
                const int c = ((((s >> 1) ^ big_tr_swz<U>(r)) & (U - 1)) << 1) | (s & 1);   // source chunk of LDS slot (r, s)
                const int k = k0 + r, o = out0 + c * 8;
                v = active && k < K && o < n_out;
                src = p + (long long)k * ld + o;
            } else {
                const int c = s ^ ((r >> 1) & 7);
                const int o = out0 + r, k = k0 + c * 8;
                v = active && o < n_out && k < K;
                src = p + (long long)o * ld + k;
            }
            const void* g = v ? (const void*)src : (const void*)g_big_zero;
            uint8_t* dst = lds + (i * BIG_T + wave * 64) * 16;                     // wave-uniform; the DMA adds lane * 16
            __builtin_amdgcn_global_load_lds((big_gptr)g, (big_lptr)dst, 16, 0, 0);
        }
    }
    // 16 out indices [o0, o0 + 16) x 32 contraction indices of k-step ks: lane l gets out o0 + (l & 15), k = 32 ks + 8 (l >> 4) + j
    static __device__ __forceinline__ bf16x8 frag(const uint8_t* lds, int o0, int ks, int lane) {
        if constexpr (!TR) {
            return *reinterpret_cast<const bf16x8*>(lds + big_nrm_off<OUT>(o0 + (lane & 15), ks * 4 + (lane >> 4)));
        } else {
            // INLINE ASM on purpose: with the __builtin_amdgcn_ds_read_tr16_b64 form hipcc (ROCm 7.2) puts an
            // s_waitcnt vmcnt(0) in front of the first transposed read of every K-tile - it drains the LDS-DMAs of the
            // NEXT tiles that were issued a few instructions earlier, i.e. the whole prefetch pipeline (NN / TN ran 1.4 x
            // slower per K-tile than NT, whose ds_read_b128 gets no such wait). The asm reads are invisible to the
            // compiler's counters: the caller waits lgkmcnt(0) and pins every fragment (big_land) before the MFMAs.
            const int g = lane >> 4, w = lane & 15, q = w >> 2, pp = w & 3;
            const int k1 = ks * 32 + 8 * g + q, k2 = k1 + 4, unit = o0 >> 4;
            const uint32_t base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const uint8_t*)lds;
            const uint32_t a1 = base + big_tr_off<OUT>(k1, unit) + pp * 8, a2 = base + big_tr_off<OUT>(k2, unit) + pp * 8;
            big_s16x4 lo, hi;
            asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"(a1));
            asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(hi) : "v"(a2));
            bf16x8 f = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            return f;
        }
    }
};

// after asm LDS reads: one wait for everything in flight, then every fragment is made opaque at this point so that no
// MFMA consuming it can be scheduled above the wait (guide 5.7: form (ii) / rule 18)
template <int N>
__device__ __forceinline__ void big_land(bf16x8 (&f)[N]) {
#pragma unroll
    for (int i = 0; i < N; ++i) asm volatile("" : "+v"(f[i]));
}

__device__ __forceinline__ void big_tile_remap(int& bx, int& by, int gx, int gy) {
    if (gx < 2 || gy < 2) return;
    constexpr int NXCD = 8, GROUP_M = 4;
    const int n = gx * gy, id = by * gx + bx;
    const int xcd = id % NXCD, q = n / NXCD, r = n % NXCD;
    const int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + id / NXCD;
    const int gsz = GROUP_M * gx, grp = t / gsz, first = grp * GROUP_M, gm = min(gy - first, GROUP_M), in = t - grp * gsz;
    by = first + in % gm;
    bx = in / gm;
}

template <int BM, int BN, int WM, int WN, bool A_TR, bool B_TR, int NS>
__global__ __launch_bounds__(BIG_T) void k_gemm_big(BigOperand<BM, A_TR> A, BigOperand<BN, B_TR> B, int M, int N, int K, BigEpi ep) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    static_assert(WM * WN == 8 && NS >= 2, "8 waves, >= 2 LDS stages");
    typedef BigOperand<BM, A_TR> OA;
    typedef BigOperand<BN, B_TR> OB;
    constexpr int TM = BM / WM / 16, TN = BN / WN / 16;
    constexpr int STAGE = OA::BYTES + OB::BYTES, LPT = OA::NCH + OB::NCH;
    static_assert(LPT * (NS - 2) <= 63, "vmcnt field");
    int bx = blockIdx.x, by = blockIdx.y;
    big_tile_remap(bx, by, gridDim.x, gridDim.y);
    if (blockIdx.z) { A.p = A.p2; B.p = B.p2; ep.C = ep.C2; ep.bias = ep.bias2; ep.mask = ep.mask2; }
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int m0 = by * BM, n0 = bx * BN;
    const int nkt = (K + BIG_BK - 1) / BIG_BK;

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll
    for (int s = 0; s < NS - 1; ++s) {
        A.issue(m0, s * BIG_BK, tid, wave, smem + s * STAGE, s < nkt);
        B.issue(n0, s * BIG_BK, tid, wave, smem + s * STAGE + OA::BYTES, s < nkt);
    }
    int st = 0, st_fill = NS - 1;
    for (int kt = 0; kt < nkt; ++kt) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPT * (NS - 2)) : "memory");     // this wave's DMAs of tile kt have landed
        __builtin_amdgcn_s_barrier();                                              // ... and everybody else's; stage st_fill is free
        A.issue(m0, (kt + NS - 1) * BIG_BK, tid, wave, smem + st_fill * STAGE, kt + NS - 1 < nkt);
        B.issue(n0, (kt + NS - 1) * BIG_BK, tid, wave, smem + st_fill * STAGE + OA::BYTES, kt + NS - 1 < nkt);
        const uint8_t* As = smem + st * STAGE;
        const uint8_t* Bs = As + OA::BYTES;
#pragma unroll
        for (int ks = 0; ks < BIG_BK / 32; ++ks) {
            bf16x8 af[TM], bfr[TN];
#pragma unroll
            for (int j = 0; j < TN; ++j) bfr[j] = OB::frag(Bs, wn * (BN / WN) + j * 16, ks, lane);
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = OA::frag(As, wm * (BM / WM) + i * 16, ks, lane);
            if constexpr (A_TR || B_TR) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                big_land<TN>(bfr);
                big_land<TM>(af);
                __builtin_amdgcn_sched_barrier(0);
            }
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(big_bf16x8, bfr[j]),
                                                                        __builtin_bit_cast(big_bf16x8, af[i]), acc[i][j], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
        }
        st = st + 1 == NS ? 0 : st + 1;
        st_fill = st_fill + 1 == NS ? 0 : st_fill + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the zero-page DMAs of the tail still target LDS

    // epilogue: lane holds C[m][n .. n+3], m = tile row (lane & 15), n = tile column 4 * (lane >> 4)
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = m0 + wm * (BM / WM) + i * 16 + (lane & 15);
        if (m >= M) continue;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn * (BN / WN) + j * 16 + (lane >> 4) * 4;
            if (n >= N) continue;
            float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
            if (ep.bias) {
                const float4 b = *reinterpret_cast<const float4*>(ep.bias + n);
                v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
            }
            if (ep.out_f32 & 2) {
#pragma unroll
                for (int t = 0; t < 4; ++t) v[t] = fmaxf(v[t], 0.f);
            }
            if (ep.mask) {
                const uint2 mk = *reinterpret_cast<const uint2*>(ep.mask + (long long)m * ep.ldmask + n);
                const uint32_t e[4] = {mk.x & 0xffffu, mk.x >> 16, mk.y & 0xffffu, mk.y >> 16};
#pragma unroll
                for (int t = 0; t < 4; ++t) if (e[t] == 0 || (e[t] & 0x8000u)) v[t] = 0.f;     // bf16 > 0 <=> non-zero, sign clear
            }
            if (ep.out_f32 & 1) {
                *reinterpret_cast<float4*>(reinterpret_cast<float*>(ep.C) + (long long)m * ep.ldc + n) = make_float4(v[0], v[1], v[2], v[3]);
            } else {
                *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(ep.C) + (long long)m * ep.ldc + n) =
                    make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]));
            }
        }
    }
}

struct BigArgs { const void *a, *a2, *b, *b2; long long lda, ldb; int M, N, K, nz; };

template <int BM, int BN, int WM, int WN, bool A_TR, bool B_TR, int NS>
static int big_launch(const BigArgs& g, const BigEpi& ep, hipStream_t s) {
    const int M = g.M, N = g.N, K = g.K;
    BigOperand<BM, A_TR> A{(const bf16_t*)g.a, g.lda, M, K, (const bf16_t*)g.a2};
    BigOperand<BN, B_TR> B{(const bf16_t*)g.b, g.ldb, N, K, (const bf16_t*)g.b2};
    const size_t lds = (size_t)NS * (BM + BN) * BIG_BK * 2;
    auto kern = k_gemm_big<BM, BN, WM, WN, A_TR, B_TR, NS>;
    static bool once = false;
    if (!once) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return -6;
        once = true;
    }
    dim3 grid((N + BN - 1) / BN, (M + BM - 1) / BM, g.nz);
    kern<<<grid, BIG_T, lds, s>>>(A, B, M, N, K, ep);
    return (int)hipGetLastError();
}

// tile: 0 = 256 x 256 x 2 stages | 1 = 256 x 128 x 3 | 2 = 128 x 256 x 3 | 3 = 128 x 128 x 4 | 4 = 256 x 128 x 2 | 5 = 128 x 128 x 3
template <bool A_TR, bool B_TR>
static int big_dispatch(int tile, const BigArgs& g, const BigEpi& ep, hipStream_t s) {
    switch (tile) {
        case 0: return big_launch<256, 256, 2, 4, A_TR, B_TR, 2>(g, ep, s);
        case 1: return big_launch<256, 128, 4, 2, A_TR, B_TR, 3>(g, ep, s);
        case 2: return big_launch<128, 256, 2, 4, A_TR, B_TR, 3>(g, ep, s);
        case 3: return big_launch<128, 128, 2, 4, A_TR, B_TR, 4>(g, ep, s);
        case 4: return big_launch<256, 128, 4, 2, A_TR, B_TR, 2>(g, ep, s);
        case 5: return big_launch<128, 128, 2, 4, A_TR, B_TR, 3>(g, ep, s);
    }
    return -7;
}

// mode 0 (NT): C[M][N] = A[M][K] . B[N][K]^T | 1 (NN): C = A[M][K] . B[K][N] | 2 (TN): C = A[K][M]^T . B[K][N]
// C fp32 (out_f32 bit 0) or bf16, ReLU after the bias (bit 1); bias [N] fp32 and mask [M][ldmask] bf16 optional. lda / ldb / K multiples of 8, N / ldc of 4.
// a2 != null: TWO problems of the same shape in one launch (a2, b2, C2, bias2, mask2 = operands of the second).
DRL_API int drl_gemm_big(int mode, int tile, const void* a, long long lda, const void* b, long long ldb, int M, int N, int K,
                         void* C, long long ldc, const float* bias, const void* mask, long long ldmask, int out_f32,
                         const void* a2, const void* b2, void* C2, const float* bias2, const void* mask2, hipStream_t s) {
    if ((lda & 7) || (ldb & 7) || (ldc & 3) || (N & 3) || (K & 7) || M <= 0 || N <= 0 || K <= 0) return -4;
    if (mask && (ldmask & 3)) return -4;
    if (mode == 2 && (M & 7)) return -4;
    if (mode != 0 && (N & 7)) return -4;
    if (a2 && (!b2 || !C2 || (bias != nullptr) != (bias2 != nullptr) || (mask != nullptr) != (mask2 != nullptr))) return -4;
    BigEpi ep{C, ldc, bias, (const bf16_t*)mask, ldmask, out_f32, C2, bias2, (const bf16_t*)mask2};
    BigArgs g{a, a2, b, b2, lda, ldb, M, N, K, a2 ? 2 : 1};
    switch (mode) {
        case 0: return big_dispatch<false, false>(tile, g, ep, s);
        case 1: return big_dispatch<false, true>(tile, g, ep, s);
        case 2: return big_dispatch<true, true>(tile, g, ep, s);
    }
    return -5;
}

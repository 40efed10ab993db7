// bf16 MFMA GEMM family for gfx950: one LDS-tiled kernel template covers every GEMM-shaped op of both models
// (SURVEY G1-G4, G6a, G7 and all their backward passes):
//
//   C[Mo][No] = sum_k  Aop(m,k) * Bop(n,k)          Mo/No = output dims, k = contraction
//
// Each operand is a row-major *memory matrix* read through an "Op" functor (dense, or implicit im2col of an NHWC
// activation / uint8 frame), and is either
//   normal : out index = memory row,  contraction = memory column   (LDS tile [out][64], ds_read_b128 fragments)
//   TR     : contraction = memory row, out index = memory column    (LDS tile [64][out], fragments through the
//            gfx950 transposing LDS read ds_read_b64_tr_b16)
// so forward (NT), input-gradient (NN: B is TR) and weight-gradient (TN: A and B are TR, contraction over the batch
// rows) all run on the same code with NO explicit transposes and NO materialised im2col / col^T buffers.
//
// CDNA4 specifics:
//   * v_mfma_f32_16x16x32_bf16, 4 waves per workgroup, wave grid WM x WN, BK = 64 (two MFMA k-steps per LDS tile).
//   * Operands are passed SWAPPED to the MFMA (acc = mfma(Bfrag, Afrag)): the accumulator then holds, per lane,
//     4 consecutive output COLUMNS of one output row, so the epilogue issues 8-byte (bf16) / 16-byte (f32) stores
//     instead of four scalar ones, and bias / ReLU-mask loads are vector loads too.
//   * LDS tiles are XOR-swizzled at 16-B (normal) / 32-B (TR) granularity so ds_read_b128 and ds_read_b64_tr_b16
//     are bank-conflict free for the 16-row / 8-row groups a wave reads (guide §2, T2, T10).
//   * Global->LDS staging goes through registers, issued one K-tile ahead of the MFMAs (T14 split), with two LDS
//     buffers and a single barrier per K-tile.
//   * Epilogue fuses: x scale (uint8 frames: 1/255), + bias, ReLU, ReLU-backward mask, bf16 / f32 store, or f32
//     atomic accumulation for split-K weight gradients.
#include "common.h"

typedef __attribute__((ext_vector_type(8))) __bf16 mfma_bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;

#define BK 64
#define NTHREADS 256

enum { EPI_RELU = 1, EPI_OUT_F32 = 2, EPI_ATOMIC = 4, EPI_MASK = 8, EPI_LSTM = 16 };

struct Epilogue {
    void* C;
    long long ldc;
    const float* bias;      // per output column (No), may be null
    const bf16_t* mask;     // [Mo][ldmask] forward activation for ReLU backward, may be null
    long long ldmask;
    float scale;
    int flags;
    // EPI_LSTM: the GEMM is h_prev . W_hh^T with gate-interleaved columns (n = 4*unit + {i,f,g,o}); the lane's 4
    // consecutive outputs are exactly one unit's gates, so the whole cell update is lane-local.
    const float* xg;        // [Mo][ldxg] x.W_ih^T + b_ih + b_hh for this time step (interleaved)
    long long ldxg;
    const float* c_prev;    // [Mo][No/4]
    float* c_out;           // [Mo][No/4]
    bf16_t* h_out;          // [Mo][No/4]
    bf16_t* gates_out;      // [Mo][No] post-activation i,f,g,o (for BPTT), may be null
};

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }
__device__ __forceinline__ float tanhf_(float x) { return 2.0f / (1.0f + __expf(-2.0f * x)) - 1.0f; }

// ---------------------------------------------------------------------------------------------------
// Operand functors: element address = row_base(r) + col_off(c)   (c multiple of 8; 8 elements never straddle
// a discontinuity). row_base < 0  =>  row out of range, tile is zero-filled.
// ---------------------------------------------------------------------------------------------------
struct DenseOp {
    const bf16_t* p;
    long long ld;
    int rows;
    __device__ __forceinline__ long long row_base(int r) const { return r < rows ? (long long)r * ld : -1; }
    __device__ __forceinline__ int col_off(int c) const { return c; }
    __device__ __forceinline__ uint4 load8(long long off) const { return *reinterpret_cast<const uint4*>(p + off); }
};

// Implicit im2col of an NHWC tensor: logical matrix col[M = frames*OH*OW][K = KS*KS*C], K ordered (kh, kw, c).
// U8 = source is raw uint8 frames (conv1): 8 elements = 8 bytes, widened to bf16 on load (exact for 0..255; the
// 1/255 lives in the epilogue scale). frame_off (optional) = byte/element offset of each frame: conv1 reads straight
// out of the replay store through the sampled indices — no gather copy.
template <int H_, int W_, int C_, int KS_, int S_, bool U8_>
struct ConvOp {
    static constexpr int H = H_, W = W_, C = C_, KS = KS_, S = S_;
    static constexpr int OH = (H_ - KS_) / S_ + 1, OW = (W_ - KS_) / S_ + 1, KWC = KS_ * C_, K = KS_ * KS_ * C_;
    const void* x;
    const long long* frame_off;
    int rows;    // frames*OH*OW
    __device__ __forceinline__ long long row_base(int m) const {
        if (m >= rows) return -1;
        int b = m / (OH * OW), rem = m - b * (OH * OW);
        int oy = rem / OW, ox = rem - oy * OW;
        long long fb = frame_off ? frame_off[b] : (long long)b * (H * W * C);
        return fb + ((long long)(oy * S) * W + ox * S) * C;
    }
    __device__ __forceinline__ int col_off(int kc) const {
        int kh = kc / KWC;
        return kh * (W * C) + (kc - kh * KWC);
    }
    __device__ __forceinline__ uint4 load8(long long off) const {
        if (U8_) {
            uint2 v = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint8_t*>(x) + off);
            uint32_t o[4];
            uint32_t w[2] = {v.x, v.y};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                uint32_t b0 = (w[i >> 1] >> ((i & 1) * 16)) & 0xff, b1 = (w[i >> 1] >> ((i & 1) * 16 + 8)) & 0xff;
                o[i] = (__float_as_uint((float)b0) >> 16) | (__float_as_uint((float)b1) & 0xffff0000u);
            }
            return make_uint4(o[0], o[1], o[2], o[3]);
        }
        return *reinterpret_cast<const uint4*>(reinterpret_cast<const bf16_t*>(x) + off);
    }
};

// ---------------------------------------------------------------------------------------------------
// LDS tile layouts
// ---------------------------------------------------------------------------------------------------
// normal tile [R][64] bf16: 128-B rows, 16-B chunk c of row r lives at chunk c ^ ((r>>1)&7)
__device__ __forceinline__ int nrm_off(int r, int c) { return r * 128 + ((c ^ ((r >> 1) & 7)) << 4); }
// TR tile [64][R] bf16: rows of 2R bytes = U=R/16 units of 32 B; unit u of row k lives at u ^ f(k)
template <int R>
__device__ __forceinline__ int tr_off(int k, int unit) {
    constexpr int U = R / 16;
    constexpr int SH = (U >= 8) ? 0 : (U == 4 ? 1 : (U == 2 ? 2 : 3));
    return k * (2 * R) + (((unit ^ ((k >> SH) & (U - 1))) & (U - 1)) << 5);
}

template <int OUT, bool TR, class Op>
struct Stage {
    static constexpr int ROWS = TR ? BK : OUT, COLS = TR ? OUT : BK, CPR = COLS / 8;
    static constexpr int NCH = (ROWS * CPR) / NTHREADS;
    static_assert(NCH >= 1 && (ROWS * CPR) % NTHREADS == 0, "tile too small for 256 threads");
    uint4 reg[NCH];
    // out0 = first out index of the tile, k0 = first contraction index
    __device__ __forceinline__ void load(const Op& op, int out0, int k0, int tid) {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            int id = tid + i * NTHREADS, r = id / CPR, c = id % CPR;
            long long rb = op.row_base((TR ? k0 : out0) + r);
            reg[i] = rb >= 0 ? op.load8(rb + op.col_off((TR ? out0 : k0) + c * 8)) : make_uint4(0, 0, 0, 0);
        }
    }
    __device__ __forceinline__ void store(uint8_t* lds, int tid) const {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            int id = tid + i * NTHREADS, r = id / CPR, c = id % CPR;
            int off = TR ? (tr_off<OUT>(r, c >> 1) + ((c & 1) << 4)) : nrm_off(r, c);
            *reinterpret_cast<uint4*>(lds + off) = reg[i];
        }
    }
};

// fragment of 16 out-indices [o0, o0+16) x 32 contraction indices of k-step ks, as the MFMA wants it:
// lane l holds out index o0 + (l&15), contraction 8*(l>>4) + j, j = 0..7
template <int OUT, bool TR>
__device__ __forceinline__ bf16x8 read_frag(const uint8_t* lds, int o0, int ks, int lane) {
    if (!TR) {
        int r = o0 + (lane & 15), c = ks * 4 + (lane >> 4);
        return *reinterpret_cast<const bf16x8*>(lds + nrm_off(r, c));
    } else {
        int g = lane >> 4, w = lane & 15, q = w >> 2, p = w & 3;
        int k1 = ks * 32 + 8 * g + q, k2 = k1 + 4, unit = o0 >> 4;
        typedef __attribute__((address_space(3))) s16x4* lptr;
        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(lds + tr_off<OUT>(k1, unit) + p * 8));
        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(lds + tr_off<OUT>(k2, unit) + p * 8));
        bf16x8 f = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return f;
    }
}

// ---------------------------------------------------------------------------------------------------
// The kernel. grid = (tiles_n, tiles_m, splits). Contraction length Kc is processed in BK tiles; split z handles
// tiles [z*kt_per, min((z+1)*kt_per, nkt)).
// ---------------------------------------------------------------------------------------------------
template <int BM, int BN, int WM, int WN, bool A_TR, bool B_TR, class AOp, class BOp>
__global__ __launch_bounds__(NTHREADS) void k_gemm(AOp aop, BOp bop, int Mo, int No, int Kc, int kt_per, Epilogue ep) {
    static_assert(WM * WN == 4, "4 waves");
    constexpr int TM = BM / WM / 16, TN = BN / WN / 16;    // MFMA tiles per wave
    static_assert(TM >= 1 && TN >= 1, "wave tile");
    constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    auto As = [&](int i) -> uint8_t* { return smem + i * (A_BYTES + B_BYTES); };
    auto Bs = [&](int i) -> uint8_t* { return smem + i * (A_BYTES + B_BYTES) + A_BYTES; };

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int nkt = (Kc + BK - 1) / BK;
    const int kt0 = blockIdx.z * kt_per, kt1 = min(kt0 + kt_per, nkt);
    if (kt0 >= kt1) return;

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    Stage<BM, A_TR, AOp> sa;
    Stage<BN, B_TR, BOp> sb;
    sa.load(aop, m0, kt0 * BK, tid);
    sb.load(bop, n0, kt0 * BK, tid);
    sa.store(As(0), tid);
    sb.store(Bs(0), tid);
    __syncthreads();

    for (int kt = kt0; kt < kt1; ++kt) {
        const int cur = (kt - kt0) & 1;
        const bool more = kt + 1 < kt1;
        if (more) {
            sa.load(aop, m0, (kt + 1) * BK, tid);
            sb.load(bop, n0, (kt + 1) * BK, tid);
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = read_frag<BM, A_TR>(As(cur), wm * (BM / WM) + i * 16, ks, lane);
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[j] = read_frag<BN, B_TR>(Bs(cur), wn * (BN / WN) + j * 16, ks, lane);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(mfma_bf16x8, bf[j]),
                                                                        __builtin_bit_cast(mfma_bf16x8, af[i]), acc[i][j], 0, 0, 0);
        }
        if (more) {
            sa.store(As(cur ^ 1), tid);
            sb.store(Bs(cur ^ 1), tid);
        }
        __syncthreads();
    }

    // epilogue: lane holds C[m][n..n+3], m = tile row (lane&15), n = tile col 4*(lane>>4)
    const int flags = ep.flags;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = m0 + wm * (BM / WM) + i * 16 + (lane & 15);
        if (m >= Mo) continue;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn * (BN / WN) + j * 16 + (lane >> 4) * 4;
            if (n >= No) continue;
            float v[4] = {acc[i][j][0] * ep.scale, acc[i][j][1] * ep.scale, acc[i][j][2] * ep.scale, acc[i][j][3] * ep.scale};
            if (ep.bias) {
                float4 b = *reinterpret_cast<const float4*>(ep.bias + n);
                v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
            }
            if (flags & EPI_LSTM) {
                const float4 x = *reinterpret_cast<const float4*>(ep.xg + (long long)m * ep.ldxg + n);
                const int H = No >> 2, u = n >> 2;
                const float ig = sigmoidf_(v[0] + x.x), fg = sigmoidf_(v[1] + x.y), gg = tanhf_(v[2] + x.z), og = sigmoidf_(v[3] + x.w);
                const float c = fg * ep.c_prev[(long long)m * H + u] + ig * gg;
                ep.c_out[(long long)m * H + u] = c;
                ep.h_out[(long long)m * H + u] = f2bf(og * tanhf_(c));
                if (ep.gates_out)
                    *reinterpret_cast<uint2*>(ep.gates_out + (long long)m * No + n) = make_uint2(pack_bf2(ig, fg), pack_bf2(gg, og));
                continue;
            }
            if (flags & EPI_RELU) {
#pragma unroll
                for (int t = 0; t < 4; ++t) v[t] = fmaxf(v[t], 0.f);
            }
            if (flags & EPI_MASK) {
                uint2 mk = *reinterpret_cast<const uint2*>(ep.mask + (long long)m * ep.ldmask + n);
                // bf16 > 0  <=>  sign bit clear and not zero
                uint32_t e[4] = {mk.x & 0xffffu, mk.x >> 16, mk.y & 0xffffu, mk.y >> 16};
#pragma unroll
                for (int t = 0; t < 4; ++t) if (e[t] == 0 || (e[t] & 0x8000u)) v[t] = 0.f;
            }
            if (flags & EPI_OUT_F32) {
                float* c = reinterpret_cast<float*>(ep.C) + (long long)m * ep.ldc + n;
                if (flags & EPI_ATOMIC) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) atomicAdd(c + t, v[t]);
                } else {
                    *reinterpret_cast<float4*>(c) = make_float4(v[0], v[1], v[2], v[3]);
                }
            } else {
                bf16_t* c = reinterpret_cast<bf16_t*>(ep.C) + (long long)m * ep.ldc + n;
                *reinterpret_cast<uint2*>(c) = make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]));
            }
        }
    }
}

template <int BM, int BN, int WM, int WN, bool A_TR, bool B_TR, class AOp, class BOp>
static int launch(const AOp& a, const BOp& b, int Mo, int No, int Kc, int splits, const Epilogue& ep, hipStream_t s) {
    if (No % BN != 0 || Mo <= 0 || Kc <= 0) return -2;
    const int nkt = (Kc + BK - 1) / BK;
    if (splits < 1) splits = 1;
    if (splits > nkt) splits = nkt;
    const int kt_per = (nkt + splits - 1) / splits;
    splits = (nkt + kt_per - 1) / kt_per;
    if (splits > 1 && !(ep.flags & EPI_ATOMIC)) return -3;
    dim3 grid(No / BN, (Mo + BM - 1) / BM, splits);
    const size_t lds = 2 * (BM + BN) * BK * 2;
    auto kern = k_gemm<BM, BN, WM, WN, A_TR, B_TR, AOp, BOp>;
    if (lds > 64 * 1024) {
        static bool once = false;
        if (!once) { hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); once = true; }
    }
    kern<<<grid, NTHREADS, lds, s>>>(a, b, Mo, No, Kc, kt_per, ep);
    return (int)hipGetLastError();
}

typedef ConvOp<84, 84, 4, 8, 4, true> Conv1Op;
typedef ConvOp<20, 20, 32, 4, 2, false> Conv2Op;
typedef ConvOp<9, 9, 64, 3, 1, false> Conv3Op;

static Epilogue make_epi(void* C, long long ldc, const float* bias, const void* mask, long long ldmask, float scale, int flags) {
    Epilogue e;
    e.C = C; e.ldc = ldc; e.bias = bias; e.mask = (const bf16_t*)mask; e.ldmask = ldmask; e.scale = scale; e.flags = flags;
    e.xg = nullptr; e.ldxg = 0; e.c_prev = nullptr; e.c_out = nullptr; e.h_out = nullptr; e.gates_out = nullptr;
    return e;
}

// ---- dense GEMMs ------------------------------------------------------------------------------------
// mode 0 (NT): C[M][N] = A[M][K] . W[N][K]^T          a = A (ld lda), b = W (ld ldb)
// mode 1 (NN): C[M][N] = A[M][K] . B[K][N]            a = A, b = B [K][N]
// mode 2 (TN): C[M][N] = A[K][M]^T . B[K][N]          a = A [K][M], b = B [K][N]   (contraction over memory rows)
// big != 0 selects the 128x128 tile (needs N % 128 == 0), else 64x64.
DRL_API int drl_gemm(int mode, int big, const void* a, long long lda, const void* b, long long ldb, int M, int N, int K,
                     void* C, long long ldc, const float* bias, const void* mask, long long ldmask, float scale, int flags,
                     int splits, hipStream_t s) {
    if ((lda & 7) || (ldb & 7) || (ldc & 3) || (N & 3)) return -4;
    Epilogue ep = make_epi(C, ldc, bias, mask, ldmask, scale, flags);
    const bf16_t* pa = (const bf16_t*)a;
    const bf16_t* pb = (const bf16_t*)b;
    if (mode == 0) {
        if (K % BK) return -5;
        DenseOp A{pa, lda, M}, B{pb, ldb, N};
        return big ? launch<128, 128, 2, 2, false, false>(A, B, M, N, K, splits, ep, s)
                   : launch<64, 64, 2, 2, false, false>(A, B, M, N, K, splits, ep, s);
    } else if (mode == 1) {
        if (K % BK) return -5;
        DenseOp A{pa, lda, M}, B{pb, ldb, K};
        return big ? launch<128, 128, 2, 2, false, true>(A, B, M, N, K, splits, ep, s)
                   : launch<64, 64, 2, 2, false, true>(A, B, M, N, K, splits, ep, s);
    } else if (mode == 2) {
        if (M % (big ? 128 : 64)) return -5;
        DenseOp A{pa, lda, K}, B{pb, ldb, K};
        return big ? launch<128, 128, 2, 2, true, true>(A, B, M, N, K, splits, ep, s)
                   : launch<64, 64, 2, 2, true, true>(A, B, M, N, K, splits, ep, s);
    }
    return -1;
}

// ---- conv forward: out[frames*OH*OW][Cout] = relu(scale * col . W^T + bias), NHWC bf16 ------------------
// layer 1: x = uint8 frames (optionally through frame_off byte offsets), W [32][256];  layer 2: W [64][512];  layer 3: W [64][576]
DRL_API int drl_conv_fwd(int layer, const void* x, const long long* frame_off, int frames, const void* w, const float* bias,
                         void* out, int relu, hipStream_t s) {
    const int fl = relu ? EPI_RELU : 0;
    if (layer == 1) {
        Conv1Op A{x, frame_off, frames * Conv1Op::OH * Conv1Op::OW};
        DenseOp B{(const bf16_t*)w, Conv1Op::K, 32};
        return launch<128, 32, 4, 1, false, false>(A, B, A.rows, 32, Conv1Op::K, 1, make_epi(out, 32, bias, nullptr, 0, 1.0f / 255.0f, fl), s);
    } else if (layer == 2) {
        Conv2Op A{x, frame_off, frames * Conv2Op::OH * Conv2Op::OW};
        DenseOp B{(const bf16_t*)w, Conv2Op::K, 64};
        return launch<128, 64, 2, 2, false, false>(A, B, A.rows, 64, Conv2Op::K, 1, make_epi(out, 64, bias, nullptr, 0, 1.0f, fl), s);
    } else if (layer == 3) {
        Conv3Op A{x, frame_off, frames * Conv3Op::OH * Conv3Op::OW};
        DenseOp B{(const bf16_t*)w, Conv3Op::K, 64};
        return launch<128, 64, 2, 2, false, false>(A, B, A.rows, 64, Conv3Op::K, 1, make_epi(out, 64, bias, nullptr, 0, 1.0f, fl), s);
    }
    return -1;
}

// ---- conv weight gradient: dW[Cout][K] (+)= scale * dY[M][Cout]^T . col[M][K]  (f32, split over M, atomic) ----
// dW must be zeroed by the caller when splits > 1 (the flat gradient buffer is zeroed once per step anyway).
DRL_API int drl_conv_wgrad(int layer, const void* x, const long long* frame_off, int frames, const void* dy, float* dw,
                           int splits, hipStream_t s) {
    const int fl = EPI_OUT_F32 | EPI_ATOMIC;
    if (layer == 1) {
        Conv1Op B{x, frame_off, frames * Conv1Op::OH * Conv1Op::OW};
        DenseOp A{(const bf16_t*)dy, 32, B.rows};
        return launch<32, 64, 1, 4, true, true>(A, B, 32, Conv1Op::K, B.rows, splits, make_epi(dw, Conv1Op::K, nullptr, nullptr, 0, 1.0f / 255.0f, fl), s);
    } else if (layer == 2) {
        Conv2Op B{x, frame_off, frames * Conv2Op::OH * Conv2Op::OW};
        DenseOp A{(const bf16_t*)dy, 64, B.rows};
        return launch<64, 64, 2, 2, true, true>(A, B, 64, Conv2Op::K, B.rows, splits, make_epi(dw, Conv2Op::K, nullptr, nullptr, 0, 1.0f, fl), s);
    } else if (layer == 3) {
        Conv3Op B{x, frame_off, frames * Conv3Op::OH * Conv3Op::OW};
        DenseOp A{(const bf16_t*)dy, 64, B.rows};
        return launch<64, 64, 2, 2, true, true>(A, B, 64, Conv3Op::K, B.rows, splits, make_epi(dw, Conv3Op::K, nullptr, nullptr, 0, 1.0f, fl), s);
    }
    return -1;
}

// ---- conv input gradient, step 1: dCol[M][K] = dY[M][Cout] . W[Cout][K]   (bf16; col2im gathers it, conv_bwd.hip) ----
DRL_API int drl_conv_dcol(int layer, const void* dy, int frames, const void* w, void* dcol, hipStream_t s) {
    if (layer == 2) {
        const int M = frames * Conv2Op::OH * Conv2Op::OW;
        DenseOp A{(const bf16_t*)dy, 64, M}, B{(const bf16_t*)w, Conv2Op::K, 64};
        return launch<128, 64, 2, 2, false, true>(A, B, M, Conv2Op::K, 64, 1, make_epi(dcol, Conv2Op::K, nullptr, nullptr, 0, 1.0f, 0), s);
    } else if (layer == 3) {
        const int M = frames * Conv3Op::OH * Conv3Op::OW;
        DenseOp A{(const bf16_t*)dy, 64, M}, B{(const bf16_t*)w, Conv3Op::K, 64};
        return launch<128, 64, 2, 2, false, true>(A, B, M, Conv3Op::K, 64, 1, make_epi(dcol, Conv3Op::K, nullptr, nullptr, 0, 1.0f, 0), s);
    }
    return -1;
}

// ---- LSTM forward step: (h, c) <- cell(xg_t + h_prev . W_hh^T, c_prev), gate-interleaved layout -----------------
// h_prev bf16 [M][H], w_hh bf16 [4H][H] (row n = 4*unit + gate), xg f32 [M][4H], c f32 [M][H].
DRL_API int drl_lstm_step(const void* h_prev, const void* w_hh, const float* xg, const float* c_prev, void* h_out, float* c_out,
                          void* gates_out, int M, int H, hipStream_t s) {
    if (H % 64 != 0 || M <= 0) return -1;
    Epilogue ep = make_epi(nullptr, 0, nullptr, nullptr, 0, 1.0f, EPI_LSTM);
    ep.xg = xg; ep.ldxg = 4LL * H; ep.c_prev = c_prev; ep.c_out = c_out; ep.h_out = (bf16_t*)h_out; ep.gates_out = (bf16_t*)gates_out;
    DenseOp A{(const bf16_t*)h_prev, H, M}, B{(const bf16_t*)w_hh, H, 4 * H};
    return launch<64, 64, 2, 2, false, false>(A, B, M, 4 * H, H, 1, ep, s);
}

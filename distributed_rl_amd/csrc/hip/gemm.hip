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
#include "colsum_body.h"
#include "head_bwd_body.h"
#include "sumtree_body.h"
#include <type_traits>

typedef __attribute__((ext_vector_type(8))) __bf16 mfma_bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;


template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}
#define NTHREADS 256
constexpr bool GEMM_STORE_FIRST = true;

enum { EPI_RELU = 1, EPI_OUT_F32 = 2, EPI_ATOMIC = 4, EPI_MASK = 8, EPI_LSTM = 16, EPI_REMAP = 32 };

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
    // EPI_REMAP (implicit dgrad): GEMM row m = (b, yy, xx) of parity class (py, px) -> NHWC pixel row
    // (b*rm_h + rm_s*yy + py)*rm_w + rm_s*xx + px, used for the C store and the mask load
    int rm_hc, rm_wc, rm_s, rm_h, rm_w, rm_py, rm_px;
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
    typedef unsigned off_t;  // element offsets of a dense operand fit 32 bits (checked on the host): 1 VGPR per address
    __device__ __forceinline__ bool row_ok(int r) const { return r < rows; }
    __device__ __forceinline__ off_t row_base(int r) const { return (off_t)r * (off_t)ld; }
    __device__ __forceinline__ int col_off(int c) const { return c; }
    __device__ __forceinline__ off_t addr(off_t rb, int c) const { return rb + (off_t)c; }
    __device__ __forceinline__ bool pair_ok(off_t, int) const { return true; }
    static constexpr bool HAS_CLASS = false;
    static constexpr bool ROW_PREFETCH = false;
    typedef uint4 raw_t;     // 8 bf16
    __device__ __forceinline__ raw_t load_raw(off_t off) const { return *reinterpret_cast<const uint4*>(p + off); }
    __device__ __forceinline__ const void* gptr(off_t off) const { return p + off; }
    __device__ static __forceinline__ raw_t zero_raw() { return make_uint4(0, 0, 0, 0); }
    __device__ static __forceinline__ uint4 to_bf16x8(raw_t v) { return v; }
};

// Implicit im2col of an NHWC tensor: logical matrix col[M = frames*OH*OW][K = KS*KS*C], K ordered (kh, kw, c).
// U8 = source is raw uint8 frames (conv1): 8 elements = 8 bytes, widened to bf16 on load (exact for 0..255; the
// 1/255 lives in the epilogue scale). OFF_ (conv1 only) = the source is a replay store of single H x W PLANES (every
// frame kept once, replay/device_per.py): frame b = the C consecutive planes from byte offset frame_off[b], and K is
// ordered (c, kh, kw) - PyTorch's own weight order - so that 8 consecutive k are still 8 consecutive bytes (one patch
// row of one plane). conv1 reads straight out of the store through the sampled offsets - no gather copy.
template <int H_, int W_, int C_, int KS_, int S_, bool U8_, bool OFF_ = false>
struct ConvOp {
    static constexpr int H = H_, W = W_, C = C_, KS = KS_, S = S_;
    static constexpr int OH = (H_ - KS_) / S_ + 1, OW = (W_ - KS_) / S_ + 1, KWC = KS_ * C_, K = KS_ * KS_ * C_;
    const void* x;
    const long long* frame_off;
    int rows;    // frames*OH*OW
    // frames addressed through frame_off live in a >4 GB replay store: 64-bit offsets; dense batches fit 32 bits
    typedef typename std::conditional<OFF_, long long, unsigned>::type off_t;
    __device__ __forceinline__ bool row_ok(int m) const { return m < rows; }
    __device__ __forceinline__ off_t row_base(int m) const {      // branch-free (see Stage::load)
        const int mm = m < rows ? m : 0;
        const int b = mm / (OH * OW), rem = mm - b * (OH * OW);
        const int oy = rem / OW, ox = rem - oy * OW;
        off_t fb;
        if constexpr (OFF_) return (off_t)frame_off[b] + (off_t)((oy * S) * W + ox * S);
        else return (off_t)b * (off_t)(H * W * C) + (off_t)(((oy * S) * W + ox * S) * C);
    }
    // OFF_: the frame base is a dependent global load (frame_off[b]). When the rows of a tile change every K-tile (TR
    // operand of the weight-gradient GEMM) that load would sit in front of every data load and serialise the prefetch
    // ring, so the mainloop fetches the frame bases of a stage's NEXT tile right after issuing its current one
    // (Stage::load) and hands them back here.
    static constexpr bool ROW_PREFETCH = OFF_;
    __device__ __forceinline__ long long row_pref(int m) const {
        if constexpr (OFF_) return frame_off[(m < rows ? m : 0) / (OH * OW)];
        else return 0;
    }
    __device__ __forceinline__ off_t row_base_fb(int m, long long fb) const {
        const int mm = m < rows ? m : 0;
        const int b = mm / (OH * OW), rem = mm - b * (OH * OW);
        const int oy = rem / OW, ox = rem - oy * OW;
        if constexpr (OFF_) return (off_t)fb + (off_t)((oy * S) * W + ox * S);
        else return (off_t)fb + (off_t)(((oy * S) * W + ox * S) * C);
    }
    __device__ __forceinline__ int col_off(int kc) const {
        if constexpr (OFF_) {                     // k = (c, kh, kw): plane c, patch row kh
            static_assert(!OFF_ || (U8_ && KS_ == 8), "the plane store form is conv1's");
            const int c = kc / (KS * KS), rem = kc - c * (KS * KS), kh = rem / KS;
            return c * (H * W) + kh * W + (rem - kh * KS);
        } else {
            const int kh = kc / KWC;
            return kh * (W * C) + (kc - kh * KWC);
        }
    }
    __device__ __forceinline__ off_t addr(off_t rb, int kc) const { return rb + (off_t)col_off(kc); }
    __device__ __forceinline__ bool pair_ok(off_t, int) const { return true; }
    static constexpr bool HAS_CLASS = false;
    // raw 8 elements as they sit in memory (8 bytes for uint8 frames, 16 for bf16); widened to bf16 only when the
    // tile is written to LDS, so the global loads stay in flight across the MFMA phase instead of being consumed
    // by the conversion right after issue.
    struct raw_u8 { uint2 v; };
    typedef typename std::conditional<U8_, uint2, uint4>::type raw_t;
    __device__ __forceinline__ raw_t load_raw(off_t off) const {
        if constexpr (U8_) return *reinterpret_cast<const uint2*>(reinterpret_cast<const uint8_t*>(x) + off);
        else return *reinterpret_cast<const uint4*>(reinterpret_cast<const bf16_t*>(x) + off);
    }
    __device__ __forceinline__ const void* gptr(off_t off) const {
        if constexpr (U8_) return reinterpret_cast<const uint8_t*>(x) + off;
        else return reinterpret_cast<const bf16_t*>(x) + off;
    }
    static constexpr bool IS_U8 = U8_;
    __device__ static __forceinline__ raw_t zero_raw() {
        if constexpr (U8_) return make_uint2(0, 0);
        else return make_uint4(0, 0, 0, 0);
    }
    __device__ static __forceinline__ uint4 to_bf16x8(raw_t v) {
        if constexpr (U8_) {
            // v_cvt_f32_ubyteN per byte, then ONE v_perm_b32 per pair takes the two high halves (exact: 0..255 fit bf16)
            uint32_t o[4];
            const uint32_t w[2] = {v.x, v.y};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint32_t ww = w[i >> 1];
                const float f0 = (i & 1) ? (float)((ww >> 16) & 0xffu) : (float)(ww & 0xffu);
                const float f1 = (i & 1) ? (float)(ww >> 24) : (float)((ww >> 8) & 0xffu);
                o[i] = __builtin_amdgcn_perm(__float_as_uint(f1), __float_as_uint(f0), 0x07060302u);
            }
            return make_uint4(o[0], o[1], o[2], o[3]);
        } else {
            return v;
        }
    }
};

// ---------------------------------------------------------------------------------------------------
// Implicit input-gradient (dgrad) operands: dX[b,y,x,ci] = sum_{ky,kx,co} dY[b,(y-ky)/S,(x-kx)/S,co] * W[co,ky,kx,ci]
// as ONE GEMM per stride-parity class (py, px) = (y % S, x % S): only the taps ky = py + S*ty, kx = px + S*tx touch a
// pixel of that class, so M = frames*(H/S)*(W/S) pixels, K = (KS/S)^2 * CO ordered (ty, tx, co), N = CI. No dCol matrix
// is materialised and no col2im pass runs (the dCol GEMM + col2im pair moved 58 / 84 MB per step for conv3 / conv2 at
// B=512; this form moves 8.5 / 18 MB). blockIdx.z selects the class (HAS_CLASS), the epilogue maps the class-local row
// back to the NHWC pixel and applies the ReLU mask of the layer input.
// ---------------------------------------------------------------------------------------------------
template <int H_, int W_, int KS_, int S_, int CO_>
struct DgradOp {          // A operand (normal orientation): rows = pixels of one parity class, k = (ty, tx, co)
    static constexpr int H = H_, W = W_, KS = KS_, S = S_, CO = CO_;
    static constexpr int OH = (H_ - KS_) / S_ + 1, OW = (W_ - KS_) / S_ + 1, TS = KS_ / S_, HC = H_ / S_, WC = W_ / S_;
    static constexpr int K = TS * TS * CO_;
    static_assert(KS_ % S_ == 0 && H_ % S_ == 0 && W_ % S_ == 0, "parity classes must tile the input");
    static constexpr bool HAS_CLASS = true;
    static constexpr bool ROW_PREFETCH = false;
    const bf16_t* p;      // dY [frames][OH][OW][CO]
    int rows;             // frames*HC*WC (per class)
    int py, px;           // set from blockIdx.z in the kernel
    __device__ __forceinline__ void set_class(int z) { py = z / S; px = z % S; }
    // row handle: bits 0..39 element offset of dY[b][yy][xx][0] (may point past the valid window), 40..47 yy, 48..55 xx
    typedef unsigned long long off_t;
    __device__ __forceinline__ bool row_ok(int m) const { return m < rows; }
    __device__ __forceinline__ off_t row_base(int m) const {
        const int mm = m < rows ? m : 0;
        const int b = mm / (HC * WC), rem = mm - b * (HC * WC), yy = rem / WC, xx = rem - yy * WC;
        const off_t o = (off_t)((long long)(b * OH + yy) * OW + xx) * CO;
        return o | ((off_t)yy << 40) | ((off_t)xx << 48);
    }
    __device__ __forceinline__ int col_off(int c) const { return c; }
    __device__ __forceinline__ bool pair_ok(off_t rb, int k) const {
        const int tap = k / CO, ty = tap / TS, tx = tap - ty * TS;
        const int oy = (int)((rb >> 40) & 0xff) - ty, ox = (int)((rb >> 48) & 0xff) - tx;
        return oy >= 0 && oy < OH && ox >= 0 && ox < OW;
    }
    __device__ __forceinline__ off_t addr(off_t rb, int k) const {
        const int tap = k / CO, ty = tap / TS, tx = tap - ty * TS;
        return (rb & 0xffffffffffull) - (off_t)((ty * OW + tx) * CO) + (off_t)(k - tap * CO);
    }
    typedef uint4 raw_t;
    __device__ __forceinline__ raw_t load_raw(off_t off) const { return *reinterpret_cast<const uint4*>(p + off); }
    __device__ __forceinline__ const void* gptr(off_t off) const { return p + off; }
    __device__ static __forceinline__ raw_t zero_raw() { return make_uint4(0, 0, 0, 0); }
    __device__ static __forceinline__ uint4 to_bf16x8(raw_t v) { return v; }
};

template <int KS_, int S_, int CI_, int CO_>
struct WdgradOp {         // B operand (transposed orientation, memory rows = k): W[co][ky][kx][ci], k = (ty, tx, co)
    static constexpr int TS = KS_ / S_, K = TS * TS * CO_;
    static constexpr bool HAS_CLASS = true;
    static constexpr bool ROW_PREFETCH = false;
    const bf16_t* p;
    int py, px;
    __device__ __forceinline__ void set_class(int z) { py = z / S_; px = z % S_; }
    typedef unsigned off_t;
    __device__ __forceinline__ bool row_ok(int k) const { return k < K; }
    __device__ __forceinline__ off_t row_base(int k) const {
        const int kk = k < K ? k : 0;
        const int tap = kk / CO_, co = kk - tap * CO_, ty = tap / TS, tx = tap - ty * TS;
        return (off_t)(co * (KS_ * KS_ * CI_) + ((py + S_ * ty) * KS_ + (px + S_ * tx)) * CI_);
    }
    __device__ __forceinline__ int col_off(int c) const { return c; }
    __device__ __forceinline__ off_t addr(off_t rb, int c) const { return rb + (off_t)c; }
    __device__ __forceinline__ bool pair_ok(off_t, int) const { return true; }
    typedef uint4 raw_t;
    __device__ __forceinline__ raw_t load_raw(off_t off) const { return *reinterpret_cast<const uint4*>(p + off); }
    __device__ __forceinline__ const void* gptr(off_t off) const { return p + off; }
    __device__ static __forceinline__ raw_t zero_raw() { return make_uint4(0, 0, 0, 0); }
    __device__ static __forceinline__ uint4 to_bf16x8(raw_t v) { return v; }
};

// ---------------------------------------------------------------------------------------------------
// LDS tile layouts
// ---------------------------------------------------------------------------------------------------
// normal tile [R][BKT] bf16, stored as BKT/64 sub-tiles of [R][64] (128-B rows); inside a sub-tile the 16-B chunk c
// of row r lives at chunk c ^ ((r>>1)&7)  -> ds_read_b128 of 16 consecutive rows at one chunk is conflict-free.
template <int R>
__device__ __forceinline__ int nrm_off(int r, int c) {
    return (c >> 3) * (R * 128) + r * 128 + (((c & 7) ^ ((r >> 1) & 7)) << 4);
}
// TR tile [BKT][R] bf16: rows of 2R bytes = U=R/16 units of 32 B; unit u of row k lives at u ^ f(k)
template <int R>
__device__ __forceinline__ int tr_off(int k, int unit) {
    constexpr int U = R / 16;
    constexpr int SH = (U >= 8) ? 0 : (U == 4 ? 1 : (U == 2 ? 2 : 3));
    return k * (2 * R) + (((unit ^ ((k >> SH) & (U - 1))) & (U - 1)) << 5);
}

// Rows of a normal-orientation tile never change along K: their base offsets (and validity) are computed once per
// workgroup and shared by every register set of the prefetch ring.
template <int OUT, bool TR, int BKT, class Op>
struct RowCache {
    static constexpr int ROWS = TR ? BKT : OUT, COLS = TR ? OUT : BKT, CPR = COLS / 8;
    static constexpr int NCH = (ROWS * CPR) / NTHREADS;
    typename Op::off_t base[TR ? 1 : NCH];
    unsigned ok;
    __device__ __forceinline__ void init(const Op& op, int out0, int tid) {
        ok = 0;
        if constexpr (!TR) {
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                const int r = out0 + (tid + i * NTHREADS) / CPR;
                base[i] = op.row_base(r);
                ok |= (op.row_ok(r) ? 1u : 0u) << i;
            }
        }
    }
};

template <int OUT, bool TR, int BKT, class Op>
struct Stage {
    static constexpr int ROWS = TR ? BKT : OUT, COLS = TR ? OUT : BKT, CPR = COLS / 8;
    static constexpr int NCH = (ROWS * CPR) / NTHREADS;
    static_assert(NCH >= 1 && (ROWS * CPR) % NTHREADS == 0, "tile too small for 256 threads");
    typename Op::raw_t reg[NCH];
    unsigned valid;
    static constexpr bool PRE = TR && Op::ROW_PREFETCH;
    long long pre[PRE ? NCH : 1];          // frame bases of the rows of the NEXT tile this stage will load (PRE only)
    __device__ __forceinline__ void prefetch(const Op& op, int k0, int tid) {
        if constexpr (PRE) {
#pragma unroll
            for (int i = 0; i < NCH; ++i) pre[i] = op.row_pref(k0 + (tid + i * NTHREADS) / CPR);
        }
    }
    // out0 = first out index of the tile, k0 = first contraction index, Kc = contraction length. BRANCH-FREE on
    // purpose: out-of-range rows / k (and whole inactive tiles past the end of the K range) read element 0 of the
    // operand and are zeroed when the tile is written to LDS. With no control flow around the loads hipcc keeps
    // counted s_waitcnt vmcnt(N) and the register ring really stays in flight (conditional loads made it fall back
    // to vmcnt(0) everywhere).
    // next_k0 (PRE only): first contraction index of the tile this stage loads NEXT time; its frame bases are fetched
    // right behind this tile's data loads, a whole ring revolution before they are needed.
    __device__ __forceinline__ void load(const Op& op, const RowCache<OUT, TR, BKT, Op>& rc, int out0, int k0, int Kc, int tid,
                                         bool active, int next_k0 = 0) {
        valid = 0;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int id = tid + i * NTHREADS, r = id / CPR, c = id % CPR;
            const int kidx = TR ? k0 + r : k0 + c * 8;
            typename Op::off_t rb;
            bool ok = active && kidx < Kc;
            if constexpr (PRE) { rb = op.row_base_fb(kidx, pre[i]); ok = ok && op.row_ok(kidx); }
            else if constexpr (TR) { rb = op.row_base(kidx); ok = ok && op.row_ok(kidx); }
            else { rb = rc.base[i]; ok = ok && ((rc.ok >> i) & 1u) && op.pair_ok(rb, kidx); }
            typename Op::off_t off;
            if constexpr (TR) off = ok ? rb + (typename Op::off_t)op.col_off(out0 + c * 8) : 0;
            else off = ok ? op.addr(rb, kidx) : 0;
            reg[i] = op.load_raw(off);
            valid |= (ok ? 1u : 0u) << i;
        }
        prefetch(op, next_k0, tid);
    }
    __device__ __forceinline__ void store(uint8_t* lds, int tid) const {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int id = tid + i * NTHREADS, r = id / CPR, c = id % CPR;
            const int off = TR ? (tr_off<OUT>(r, c >> 1) + ((c & 1) << 4)) : nrm_off<OUT>(r, c);
            uint4 v = Op::to_bf16x8(reg[i]);
            if (!((valid >> i) & 1u)) v = make_uint4(0, 0, 0, 0);
            *reinterpret_cast<uint4*>(lds + off) = v;
        }
    }
};

// fragment of 16 out-indices [o0, o0+16) x 32 contraction indices of k-step ks, as the MFMA wants it:
// lane l holds out index o0 + (l&15), contraction 32*ks + 8*(l>>4) + j, j = 0..7
template <int OUT, bool TR>
__device__ __forceinline__ bf16x8 read_frag(const uint8_t* lds, int o0, int ks, int lane) {
    if (!TR) {
        int r = o0 + (lane & 15), c = ks * 4 + (lane >> 4);
        return *reinterpret_cast<const bf16x8*>(lds + nrm_off<OUT>(r, c));
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

// XCD-aware tile order. Workgroups are dealt round-robin to the 8 XCDs (linear id % 8) and every XCD has its own 4 MB L2:
// with the plain (bx, by) order the 32 tiles an XCD works on at a time are spread over the whole output and pull (almost)
// ALL of A and B through that L2 (FC1 forward, 16 x 16 tiles: 7.2 MB per XCD). Re-ordered, the tiles of one XCD are
// consecutive ids of a grouped order (GROUP_M tile rows, column-major inside the group - 4 x 8 tiles for a 16 x 16 grid),
// so tiles that share operand panels share an L2. Bijective for any grid (guide 5.3 "XCD swizzle must be bijective").
constexpr bool GEMM_XCD_REMAP = true;
__device__ __forceinline__ void tile_remap(int& bx, int& by, int gx, int gy) {
    if (!GEMM_XCD_REMAP || gx < 2 || gy < 2) return;
    constexpr int NXCD = 8, GROUP_M = 4;
    const int n = gx * gy, id = by * gx + bx;
    const int xcd = id % NXCD, q = n / NXCD, r = n % NXCD;
    const int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + id / NXCD;
    const int gsz = GROUP_M * gx, grp = t / gsz, first = grp * GROUP_M, gm = min(gy - first, GROUP_M), in = t - grp * gsz;
    by = first + in % gm;
    bx = in / gm;
}

// Epilogue shared by the register-ring and the LDS-DMA mainloops: lane holds C[m][n..n+3] of each of its MFMA tiles.
template <int BM, int BN, int WM, int WN>
__device__ __forceinline__ void gemm_epilogue(f32x4 (&acc)[BM / WM / 16][BN / WN / 16], const Epilogue& ep, int m0, int n0, int Mo,
                                              int No, uint8_t* smem, int tid, int lane, int wm, int wn) {
    constexpr int TM = BM / WM / 16, TN = BN / WN / 16;
    const int flags = ep.flags;
    if (flags & EPI_ATOMIC) {
        // split-K accumulation: stage the f32 tile through LDS so every atomic wave-instruction adds 256 CONTIGUOUS
        // bytes of one output row (full-rate shape, guide Guideline 12); the fragment layout itself (16 rows x 4
        // scattered dwords per instruction) measured ~6x slower and dominated the weight-gradient kernels.
        constexpr int LDC = BN + 4;
        float* tile = reinterpret_cast<float*>(smem);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int ml = wm * (BM / WM) + i * 16 + (lane & 15), nl = wn * (BN / WN) + j * 16 + (lane >> 4) * 4;
                *reinterpret_cast<float4*>(tile + ml * LDC + nl) =
                    make_float4(acc[i][j][0] * ep.scale, acc[i][j][1] * ep.scale, acc[i][j][2] * ep.scale, acc[i][j][3] * ep.scale);
            }
        __syncthreads();
        float* C = reinterpret_cast<float*>(ep.C);
        for (int idx = tid; idx < BM * BN; idx += NTHREADS) {
            const int ml = idx / BN, nl = idx - ml * BN;
            if (m0 + ml < Mo) atomicAdd(C + (long long)(m0 + ml) * ep.ldc + n0 + nl, tile[ml * LDC + nl]);
        }
        return;
    }

    // epilogue: lane holds C[m][n..n+3], m = tile row (lane&15), n = tile col 4*(lane>>4)
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = m0 + wm * (BM / WM) + i * 16 + (lane & 15);
        if (m >= Mo) continue;
        long long mrow = m;                  // row of C / mask in memory
        if (flags & EPI_REMAP) {
            const int b = m / (ep.rm_hc * ep.rm_wc), rem = m - b * (ep.rm_hc * ep.rm_wc), yy = rem / ep.rm_wc, xx = rem - yy * ep.rm_wc;
            mrow = ((long long)b * ep.rm_h + ep.rm_s * yy + ep.rm_py) * ep.rm_w + ep.rm_s * xx + ep.rm_px;
        }
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
                uint2 mk = *reinterpret_cast<const uint2*>(ep.mask + mrow * ep.ldmask + n);
                // bf16 > 0  <=>  sign bit clear and not zero
                uint32_t e[4] = {mk.x & 0xffffu, mk.x >> 16, mk.y & 0xffffu, mk.y >> 16};
#pragma unroll
                for (int t = 0; t < 4; ++t) if (e[t] == 0 || (e[t] & 0x8000u)) v[t] = 0.f;
            }
            if (flags & EPI_OUT_F32) {
                float* c = reinterpret_cast<float*>(ep.C) + mrow * ep.ldc + n;
                *reinterpret_cast<float4*>(c) = make_float4(v[0], v[1], v[2], v[3]);
            } else {
                bf16_t* c = reinterpret_cast<bf16_t*>(ep.C) + mrow * ep.ldc + n;
                *reinterpret_cast<uint2*>(c) = make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]));
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// The kernel. grid = (tiles_n, tiles_m, splits). Contraction length Kc is processed in BKT tiles; split z handles
// tiles [z*kt_per, min((z+1)*kt_per, nkt)).
//   BKT : K-tile (64 / 128 / 256). The small-grid GEMMs of the Ape-X step have ~1 workgroup per CU, so the fixed
//         per-tile chain (ds_read latency -> MFMA -> ds_write -> barrier, ~500 cycles measured) is amortised over
//         more MFMAs per barrier instead of being hidden by occupancy.
//   D   : depth of the register prefetch ring (even): while tile k is multiplied out of LDS, the global loads of
//         tiles k+1 .. k+D-1 are in flight in D-1 register sets.
// ---------------------------------------------------------------------------------------------------
// The body is a device function of the tile coordinates (bx, by, bz) so that it runs both as its own kernel (k_gemm) and
// as one of several bodies of a horizontally fused launch (k_multi, below).
template <int BM, int BN, int BKT, int WM, int WN, bool A_TR, bool B_TR, int D, class AOp, class BOp>
__device__ __forceinline__ void gemm_tile(AOp aop, BOp bop, int Mo, int No, int Kc, int kt_per, Epilogue ep, int bx, int by,
                                          int bz, int gx, int gy, uint8_t* smem) {
    tile_remap(bx, by, gx, gy);
    constexpr bool CLASSES = AOp::HAS_CLASS;     // bz = stride-parity class of an implicit dgrad (no split-K)
    static_assert(AOp::HAS_CLASS == BOp::HAS_CLASS, "both operands must agree on classes");
    if constexpr (CLASSES) {
        aop.set_class(bz);
        bop.set_class(bz);
        ep.rm_py = aop.py;
        ep.rm_px = aop.px;
    }
    static_assert(WM * WN == 4, "4 waves");
    static_assert(D >= 2 && D % 2 == 0, "ring depth");
    constexpr int TM = BM / WM / 16, TN = BN / WN / 16;    // MFMA tiles per wave
    constexpr bool STORE_FIRST = GEMM_STORE_FIRST;
    static_assert(TM >= 1 && TN >= 1, "wave tile");
    constexpr int A_BYTES = BM * BKT * 2, B_BYTES = BN * BKT * 2;
    auto As = [&](int i) -> uint8_t* { return smem + i * (A_BYTES + B_BYTES); };
    auto Bs = [&](int i) -> uint8_t* { return smem + i * (A_BYTES + B_BYTES) + A_BYTES; };

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int m0 = by * BM, n0 = bx * BN;
    const int nkt = (Kc + BKT - 1) / BKT;
    const int kt0 = CLASSES ? 0 : bz * kt_per, kt1 = min(kt0 + kt_per, nkt);
    if (kt0 >= kt1) return;

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    RowCache<BM, A_TR, BKT, AOp> ra;
    RowCache<BN, B_TR, BKT, BOp> rb;
    ra.init(aop, m0, tid);
    rb.init(bop, n0, tid);
    Stage<BM, A_TR, BKT, AOp> sa[D];
    Stage<BN, B_TR, BKT, BOp> sb[D];
#pragma unroll
    for (int s = 0; s < D; ++s) {
        sa[s].prefetch(aop, (kt0 + s) * BKT, tid);
        sb[s].prefetch(bop, (kt0 + s) * BKT, tid);
    }
#pragma unroll
    for (int s = 0; s < D; ++s) {
        sa[s].load(aop, ra, m0, (kt0 + s) * BKT, Kc, tid, kt0 + s < kt1, (kt0 + s + D) * BKT);
        sb[s].load(bop, rb, n0, (kt0 + s) * BKT, Kc, tid, kt0 + s < kt1, (kt0 + s + D) * BKT);
    }
    sa[0].store(As(0), tid);
    sb[0].store(Bs(0), tid);
    __syncthreads();

    // one K-tile: refill register set S with tile cur+D, multiply tile cur out of LDS buffer S&1 (D is even), move
    // tile cur+1 from its register set into the other LDS buffer, barrier.
    auto substep = [&](auto sc, int cur) __attribute__((always_inline)) {
        constexpr int S = decltype(sc)::value;
        sa[S].load(aop, ra, m0, (cur + D) * BKT, Kc, tid, cur + D < kt1, (cur + 2 * D) * BKT);
        sb[S].load(bop, rb, n0, (cur + D) * BKT, Kc, tid, cur + D < kt1, (cur + 2 * D) * BKT);
        if (STORE_FIRST) {   // tile cur+1 -> the other LDS buffer BEFORE the MFMA phase: its write latency hides under it
            sa[(S + 1) % D].store(As((S + 1) & 1), tid);
            sb[(S + 1) % D].store(Bs((S + 1) & 1), tid);
        }
#pragma unroll
        for (int ks = 0; ks < BKT / 32; ++ks) {
            bf16x8 af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = read_frag<BM, A_TR>(As(S & 1), wm * (BM / WM) + i * 16, ks, lane);
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[j] = read_frag<BN, B_TR>(Bs(S & 1), wn * (BN / WN) + j * 16, ks, lane);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(mfma_bf16x8, bf[j]),
                                                                        __builtin_bit_cast(mfma_bf16x8, af[i]), acc[i][j], 0, 0, 0);
        }
        if (!STORE_FIRST) {
            sa[(S + 1) % D].store(As((S + 1) & 1), tid);   // tile cur+1 (zeros past the end: never read)
            sb[(S + 1) % D].store(Bs((S + 1) & 1), tid);
        }
        __syncthreads();
    };
    int kt = kt0;
    // steady state: whole groups of D tiles, no control flow inside (keeps hipcc's vmcnt counting exact)
    for (; kt + D <= kt1; kt += D) static_for<0, D>([&](auto sc) __attribute__((always_inline)) { substep(sc, kt + decltype(sc)::value); });
    // tail: fewer than D tiles left
    static_for<0, D>([&](auto sc) __attribute__((always_inline)) { if (kt + decltype(sc)::value < kt1) substep(sc, kt + decltype(sc)::value); });

    gemm_epilogue<BM, BN, WM, WN>(acc, ep, m0, n0, Mo, No, smem, tid, lane, wm, wn);
}

// ---------------------------------------------------------------------------------------------------
// LDS-DMA mainloop (global_load_lds_dwordx4): tiles go global -> LDS with NO register staging.
// ---------------------------------------------------------------------------------------------------
// The register ring above keeps D-1 tiles in flight at a cost of 16 VGPRs per tile and ~46 vector instructions per
// K-tile (address selects, validity masks, ds_write). Here a K-tile is NCH_A + NCH_B `global_load_lds_dwordx4` per
// thread into one of NS LDS stages; NS - 1 tiles are in flight whatever the register budget.
//   * The LDS destination of an LDS-DMA is wave-uniform base + lane * 16, i.e. the LDS image is LINEAR in the chunk id
//     (tid + i * 256). The XOR swizzles of nrm_off / tr_off permute chunks inside a row only, so they are applied on the
//     SOURCE side: LDS slot (r, s) is fed from global chunk (r, s ^ f(r)); the fragment reads are unchanged.
//   * A masked chunk (row / k out of range, dgrad border tap, tile past the end of the K range) is fetched from a
//     zeroed 64-byte page instead of its real address: still branch-free, and the number of DMAs per tile is CONSTANT,
//     so one counted `s_waitcnt vmcnt(LPT * (NS - 2))` is right for every iteration, tail included.
//   * Ordering (guide "Read a staged buffer one phase AFTER the wait that retires it"): every wave waits for ITS DMAs of
//     tile kt, then a raw s_barrier makes all of them visible; the DMAs of tile kt + NS - 1 reuse the stage read in
//     iteration kt - 1 and are issued after that barrier, which every wave reaches only after its MFMAs of kt - 1.
__device__ uint4 g_zero_page[4];

typedef __attribute__((address_space(1))) const void* gas_cptr;
typedef __attribute__((address_space(3))) void* las_ptr;

template <int OUT, bool TR, class Op>
struct GldsTile {
    static constexpr int BKT = 64;
    static constexpr int ROWS = TR ? BKT : OUT, COLS = TR ? OUT : BKT, CPR = COLS / 8;
    static constexpr int NCH = (ROWS * CPR) / NTHREADS, BYTES = ROWS * COLS * 2;
    static_assert(NCH >= 1 && (ROWS * CPR) % NTHREADS == 0, "tile too small for 256 threads");
    typename Op::off_t base[TR ? 1 : NCH];
    unsigned ok;
    __device__ __forceinline__ void init(const Op& op, int out0, int tid) {
        ok = 0;
        if constexpr (!TR) {
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                const int r = out0 + (tid + i * NTHREADS) / CPR;
                base[i] = op.row_base(r);
                ok |= (op.row_ok(r) ? 1u : 0u) << i;
            }
        }
    }
    __device__ __forceinline__ void issue(const Op& op, int out0, int k0, int Kc, int tid, int wave, uint8_t* lds, bool active) const {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int id = tid + i * NTHREADS, r = id / CPR, s = id % CPR;
            int c;                                   // source chunk column of LDS slot (r, s)
            if constexpr (TR) {
                constexpr int U = OUT / 16;
                constexpr int SH = (U >= 8) ? 0 : (U == 4 ? 1 : (U == 2 ? 2 : 3));
                c = ((((s >> 1) ^ ((r >> SH) & (U - 1))) & (U - 1)) << 1) | (s & 1);
            } else {
                c = s ^ ((r >> 1) & 7);
            }
            const int kidx = TR ? k0 + r : k0 + c * 8;
            bool v = active && kidx < Kc;
            typename Op::off_t rb, off;
            if constexpr (TR) { rb = op.row_base(kidx); v = v && op.row_ok(kidx); off = rb + (typename Op::off_t)op.col_off(out0 + c * 8); }
            else { rb = base[i]; v = v && ((ok >> i) & 1u) && op.pair_ok(rb, kidx); off = op.addr(rb, kidx); }
            const void* src = v ? op.gptr(off) : (const void*)g_zero_page;
            uint8_t* dst = lds + (i * NTHREADS + wave * 64) * 16;          // wave-uniform; the DMA adds lane * 16
            __builtin_amdgcn_global_load_lds((gas_cptr)src, (las_ptr)dst, 16, 0, 0);
        }
    }
};

template <int BM, int BN, int WM, int WN, bool A_TR, bool B_TR, int NS, class AOp, class BOp>
__device__ __forceinline__ void gemm_tile_glds(AOp aop, BOp bop, int Mo, int No, int Kc, int kt_per, Epilogue ep, int bx, int by,
                                               int bz, int gx, int gy, uint8_t* smem) {
    tile_remap(bx, by, gx, gy);
    constexpr int BKT = 64;
    constexpr bool CLASSES = AOp::HAS_CLASS;
    static_assert(AOp::HAS_CLASS == BOp::HAS_CLASS, "both operands must agree on classes");
    static_assert(WM * WN == 4 && NS >= 3, "4 waves, >= 3 LDS stages");
    if constexpr (CLASSES) {
        aop.set_class(bz);
        bop.set_class(bz);
        ep.rm_py = aop.py;
        ep.rm_px = aop.px;
    }
    constexpr int TM = BM / WM / 16, TN = BN / WN / 16;
    typedef GldsTile<BM, A_TR, AOp> GA;
    typedef GldsTile<BN, B_TR, BOp> GB;
    constexpr int STAGE = GA::BYTES + GB::BYTES, LPT = GA::NCH + GB::NCH;
    static_assert(LPT * (NS - 2) <= 63, "vmcnt field");
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int m0 = by * BM, n0 = bx * BN;
    const int nkt = (Kc + BKT - 1) / BKT;
    const int kt0 = CLASSES ? 0 : bz * kt_per, kt1 = min(kt0 + kt_per, nkt);
    if (kt0 >= kt1) return;

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    GA ga;
    GB gb;
    ga.init(aop, m0, tid);
    gb.init(bop, n0, tid);
#pragma unroll
    for (int s = 0; s < NS - 1; ++s) {
        ga.issue(aop, m0, (kt0 + s) * BKT, Kc, tid, wave, smem + s * STAGE, kt0 + s < kt1);
        gb.issue(bop, n0, (kt0 + s) * BKT, Kc, tid, wave, smem + s * STAGE + GA::BYTES, kt0 + s < kt1);
    }
    int st = 0, st_fill = NS - 1;                 // stage of tile kt / stage the tile kt + NS - 1 goes to
    for (int kt = kt0; kt < kt1; ++kt) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPT * (NS - 2)) : "memory");     // this wave's DMAs of tile kt have landed
        __builtin_amdgcn_s_barrier();                                              // ... and everybody else's
        ga.issue(aop, m0, (kt + NS - 1) * BKT, Kc, tid, wave, smem + st_fill * STAGE, kt + NS - 1 < kt1);
        gb.issue(bop, n0, (kt + NS - 1) * BKT, Kc, tid, wave, smem + st_fill * STAGE + GA::BYTES, kt + NS - 1 < kt1);
        const uint8_t* As = smem + st * STAGE;
        const uint8_t* Bs = As + GA::BYTES;
#pragma unroll
        for (int ks = 0; ks < BKT / 32; ++ks) {
            bf16x8 af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = read_frag<BM, A_TR>(As, wm * (BM / WM) + i * 16, ks, lane);
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[j] = read_frag<BN, B_TR>(Bs, wn * (BN / WN) + j * 16, ks, lane);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(mfma_bf16x8, bf[j]),
                                                                        __builtin_bit_cast(mfma_bf16x8, af[i]), acc[i][j], 0, 0, 0);
        }
        st = st + 1 == NS ? 0 : st + 1;
        st_fill = st_fill + 1 == NS ? 0 : st_fill + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the zero-page DMAs of the tail still target LDS
    __syncthreads();
    gemm_epilogue<BM, BN, WM, WN>(acc, ep, m0, n0, Mo, No, smem, tid, lane, wm, wn);
}

template <int BM, int BN, int WM, int WN, bool A_TR, bool B_TR, int NS, class AOp, class BOp>
__global__ __launch_bounds__(NTHREADS) void k_gemm_glds(AOp aop, BOp bop, int Mo, int No, int Kc, int kt_per, Epilogue ep) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    gemm_tile_glds<BM, BN, WM, WN, A_TR, B_TR, NS, AOp, BOp>(aop, bop, Mo, No, Kc, kt_per, ep, blockIdx.x, blockIdx.y, blockIdx.z, gridDim.x, gridDim.y, smem);
}

template <int BM, int BN, int BKT, int WM, int WN, bool A_TR, bool B_TR, int D, class AOp, class BOp>
__global__ __launch_bounds__(NTHREADS) void k_gemm(AOp aop, BOp bop, int Mo, int No, int Kc, int kt_per, Epilogue ep) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    gemm_tile<BM, BN, BKT, WM, WN, A_TR, B_TR, D, AOp, BOp>(aop, bop, Mo, No, Kc, kt_per, ep, blockIdx.x, blockIdx.y, blockIdx.z, gridDim.x, gridDim.y, smem);
}

// grid / split / LDS plan shared by the stand-alone launch and the fused launches; returns 0 or a negative error
template <int BM, int BN, int BKT>
static int plan_gemm(int Mo, int No, int Kc, int splits, const Epilogue& ep, int classes, int* kt_per_out, dim3* grid, size_t* lds) {
    if (No % BN != 0 || Mo <= 0 || Kc <= 0) return -2;
    const int nkt = (Kc + BKT - 1) / BKT;
    if (splits < 1) splits = 1;
    if (splits > nkt) splits = nkt;
    int kt_per = (nkt + splits - 1) / splits;
    splits = (nkt + kt_per - 1) / kt_per;
    if (classes > 0) { kt_per = nkt; splits = classes; }     // grid.z = parity classes, each runs the whole K range
    if (splits > 1 && !classes && !(ep.flags & EPI_ATOMIC)) return -3;
    *grid = dim3(No / BN, (Mo + BM - 1) / BM, splits);
    *lds = 2 * (size_t)(BM + BN) * BKT * 2;
    if ((ep.flags & EPI_ATOMIC) && *lds < (size_t)BM * (BN + 4) * 4) *lds = (size_t)BM * (BN + 4) * 4;
    *kt_per_out = kt_per;
    return 0;
}

template <int BM, int BN, int BKT, int WM, int WN, bool A_TR, bool B_TR, int D, class AOp, class BOp>
static int launch(const AOp& a, const BOp& b, int Mo, int No, int Kc, int splits, const Epilogue& ep, hipStream_t s,
                  int classes = 0) {
    int kt_per;
    dim3 grid;
    size_t lds;
    if (int rc = plan_gemm<BM, BN, BKT>(Mo, No, Kc, splits, ep, classes, &kt_per, &grid, &lds)) return rc;
    auto kern = k_gemm<BM, BN, BKT, WM, WN, A_TR, B_TR, D, AOp, BOp>;
    if (lds > 64 * 1024) {
        static bool once = false;
        if (!once) {
            if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return -6;
            once = true;
        }
    }
    kern<<<grid, NTHREADS, lds, s>>>(a, b, Mo, No, Kc, kt_per, ep);
    return (int)hipGetLastError();
}

template <int BM, int BN, int WM, int WN, bool A_TR, bool B_TR, int NS, class AOp, class BOp>
static int launch_glds(const AOp& a, const BOp& b, int Mo, int No, int Kc, int splits, const Epilogue& ep, hipStream_t s,
                       int classes = 0) {
    int kt_per;
    dim3 grid;
    size_t lds;
    if (int rc = plan_gemm<BM, BN, 64>(Mo, No, Kc, splits, ep, classes, &kt_per, &grid, &lds)) return rc;
    lds = (size_t)NS * (BM + BN) * 64 * 2;
    if ((ep.flags & EPI_ATOMIC) && lds < (size_t)BM * (BN + 4) * 4) lds = (size_t)BM * (BN + 4) * 4;
    auto kern = k_gemm_glds<BM, BN, WM, WN, A_TR, B_TR, NS, AOp, BOp>;
    if (lds > 64 * 1024) {
        static bool once = false;
        if (!once) {
            if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return -6;
            once = true;
        }
    }
    kern<<<grid, NTHREADS, lds, s>>>(a, b, Mo, No, Kc, kt_per, ep);
    return (int)hipGetLastError();
}

typedef ConvOp<84, 84, 4, 8, 4, true, false> Conv1Op;
typedef ConvOp<84, 84, 4, 8, 4, true, true> Conv1OffOp;
typedef ConvOp<20, 20, 32, 4, 2, false> Conv2Op;
typedef ConvOp<9, 9, 64, 3, 1, false> Conv3Op;

__device__ __forceinline__ Epilogue make_epi_dev(void* C, long long ldc, float scale, int flags) {
    Epilogue e;
    e.C = C; e.ldc = ldc; e.bias = nullptr; e.mask = nullptr; e.ldmask = 0; e.scale = scale; e.flags = flags;
    e.xg = nullptr; e.ldxg = 0; e.c_prev = nullptr; e.c_out = nullptr; e.h_out = nullptr; e.gates_out = nullptr;
    e.rm_hc = e.rm_wc = e.rm_s = e.rm_h = e.rm_w = 1; e.rm_py = e.rm_px = 0;
    return e;
}

static Epilogue make_epi(void* C, long long ldc, const float* bias, const void* mask, long long ldmask, float scale, int flags) {
    Epilogue e;
    e.C = C; e.ldc = ldc; e.bias = bias; e.mask = (const bf16_t*)mask; e.ldmask = ldmask; e.scale = scale; e.flags = flags;
    e.xg = nullptr; e.ldxg = 0; e.c_prev = nullptr; e.c_out = nullptr; e.h_out = nullptr; e.gates_out = nullptr;
    e.rm_hc = e.rm_wc = e.rm_s = e.rm_h = e.rm_w = 1; e.rm_py = e.rm_px = 0;
    return e;
}

// ---- dense GEMMs ------------------------------------------------------------------------------------
// mode 0 (NT): C[M][N] = A[M][K] . W[N][K]^T          a = A (ld lda), b = W (ld ldb)
// mode 1 (NN): C[M][N] = A[M][K] . B[K][N]            a = A, b = B [K][N]
// mode 2 (TN): C[M][N] = A[K][M]^T . B[K][N]          a = A [K][M], b = B [K][N]   (contraction over memory rows)
// variant: 0 = 64x64x64 ring 4 | 1 = 128x128x64 ring 2 | 2 = 64x64x128 ring 2 | 3 = 64x64x256 ring 2 | 4 = 128x64x128 ring 2
//          5 = 64x32x64 ring 4 | 6 = 32x64x64 ring 4 (twice the workgroups of variant 0: two per CU for the 1024x1024 FC)
template <bool A_TR, bool B_TR>
static int dense_dispatch(int variant, const DenseOp& A, const DenseOp& B, int M, int N, int K, int splits, const Epilogue& ep,
                          hipStream_t s) {
    switch (variant) {
        case 0: return launch<64, 64, 64, 2, 2, A_TR, B_TR, 4>(A, B, M, N, K, splits, ep, s);
        case 1: return launch<128, 128, 64, 2, 2, A_TR, B_TR, 2>(A, B, M, N, K, splits, ep, s);
        case 2: return launch<64, 64, 128, 2, 2, A_TR, B_TR, 2>(A, B, M, N, K, splits, ep, s);
        case 3: return launch<64, 64, 256, 2, 2, A_TR, B_TR, 2>(A, B, M, N, K, splits, ep, s);
        case 4: return launch<128, 64, 128, 2, 2, A_TR, B_TR, 2>(A, B, M, N, K, splits, ep, s);
        case 5: return launch<64, 32, 64, 4, 1, A_TR, B_TR, 4>(A, B, M, N, K, splits, ep, s);
        case 6: return launch<32, 64, 64, 1, 4, A_TR, B_TR, 4>(A, B, M, N, K, splits, ep, s);
        // LDS-DMA mainloop (no register staging): tile / number of LDS stages
        case 10: return launch_glds<64, 64, 2, 2, A_TR, B_TR, 4>(A, B, M, N, K, splits, ep, s);
        case 11: return launch_glds<64, 64, 2, 2, A_TR, B_TR, 6>(A, B, M, N, K, splits, ep, s);
        case 12: return launch_glds<128, 128, 2, 2, A_TR, B_TR, 3>(A, B, M, N, K, splits, ep, s);
        case 13: return launch_glds<128, 64, 2, 2, A_TR, B_TR, 4>(A, B, M, N, K, splits, ep, s);
        case 14: return launch_glds<64, 64, 2, 2, A_TR, B_TR, 8>(A, B, M, N, K, splits, ep, s);
    }
    return -7;
}

DRL_API int drl_gemm(int mode, int variant, const void* a, long long lda, const void* b, long long ldb, int M, int N, int K,
                     void* C, long long ldc, const float* bias, const void* mask, long long ldmask, float scale, int flags,
                     int splits, hipStream_t s) {
    if ((lda & 7) || (ldb & 7) || (ldc & 3) || (N & 3)) return -4;
    if ((long long)M * lda >= (1LL << 32) || (long long)K * lda >= (1LL << 32) || (long long)N * ldb >= (1LL << 32) || (long long)K * ldb >= (1LL << 32)) return -8;
    Epilogue ep = make_epi(C, ldc, bias, mask, ldmask, scale, flags);
    const bf16_t* pa = (const bf16_t*)a;
    const bf16_t* pb = (const bf16_t*)b;
    const int bm = (variant == 1 || variant == 4 || variant == 12 || variant == 13) ? 128 : (variant == 6 ? 32 : 64);
    if (mode == 0) {
        if (K % 8) return -5;
        DenseOp A{pa, lda, M}, B{pb, ldb, N};
        return dense_dispatch<false, false>(variant, A, B, M, N, K, splits, ep, s);
    } else if (mode == 1) {
        if (K % 8) return -5;
        DenseOp A{pa, lda, M}, B{pb, ldb, K};
        return dense_dispatch<false, true>(variant, A, B, M, N, K, splits, ep, s);
    } else if (mode == 2) {
        if (M % bm) return -5;
        DenseOp A{pa, lda, K}, B{pb, ldb, K};
        return dense_dispatch<true, true>(variant, A, B, M, N, K, splits, ep, s);
    }
    return -1;
}

// ---- conv forward: out[frames*OH*OW][Cout] = relu(scale * col . W^T + bias), NHWC bf16 ------------------
// layer 1: x = uint8 frames (optionally through frame_off byte offsets), W [32][256];  layer 2: W [64][512];  layer 3: W [64][576]
DRL_API int drl_conv_fwd(int layer, const void* x, const long long* frame_off, int frames, const void* w, const float* bias,
                         void* out, int relu, hipStream_t s) {
    const int fl = (relu & 1) ? EPI_RELU : 0;
    const int variant = relu >> 8;     // layers 2 / 3: 0 = register ring | 1..3 = LDS-DMA mainloop (tile / stages), tuning hook
    if (layer == 1) {
        DenseOp B{(const bf16_t*)w, Conv1Op::K, 32};
        const int rows = frames * Conv1Op::OH * Conv1Op::OW;
        const Epilogue ep = make_epi(out, 32, bias, nullptr, 0, 1.0f / 255.0f, fl);
        if (frame_off) {
            Conv1OffOp A{x, frame_off, rows};
            return launch<128, 32, 64, 4, 1, false, false, 4>(A, B, rows, 32, Conv1Op::K, 1, ep, s);
        }
        Conv1Op A{x, nullptr, rows};
        return launch<128, 32, 64, 4, 1, false, false, 4>(A, B, rows, 32, Conv1Op::K, 1, ep, s);
    } else if (layer == 2) {
        Conv2Op A{x, frame_off, frames * Conv2Op::OH * Conv2Op::OW};
        DenseOp B{(const bf16_t*)w, Conv2Op::K, 64};
        const Epilogue ep = make_epi(out, 64, bias, nullptr, 0, 1.0f, fl);
        if (variant == 1) return launch_glds<128, 64, 2, 2, false, false, 3>(A, B, A.rows, 64, Conv2Op::K, 1, ep, s);
        if (variant == 2) return launch_glds<128, 64, 2, 2, false, false, 4>(A, B, A.rows, 64, Conv2Op::K, 1, ep, s);
        if (variant == 3) return launch_glds<64, 64, 2, 2, false, false, 4>(A, B, A.rows, 64, Conv2Op::K, 1, ep, s);
        return launch<128, 64, 64, 2, 2, false, false, 2>(A, B, A.rows, 64, Conv2Op::K, 1, ep, s);
    } else if (layer == 3) {
        Conv3Op A{x, frame_off, frames * Conv3Op::OH * Conv3Op::OW};
        DenseOp B{(const bf16_t*)w, Conv3Op::K, 64};
        const Epilogue ep = make_epi(out, 64, bias, nullptr, 0, 1.0f, fl);
        if (variant == 1) return launch_glds<128, 64, 2, 2, false, false, 3>(A, B, A.rows, 64, Conv3Op::K, 1, ep, s);
        if (variant == 2) return launch_glds<128, 64, 2, 2, false, false, 4>(A, B, A.rows, 64, Conv3Op::K, 1, ep, s);
        if (variant == 3) return launch_glds<64, 64, 2, 2, false, false, 4>(A, B, A.rows, 64, Conv3Op::K, 1, ep, s);
        return launch<128, 64, 64, 2, 2, false, false, 2>(A, B, A.rows, 64, Conv3Op::K, 1, ep, s);
    }
    return -1;
}

// ---- conv weight gradient: dW[Cout][K] (+)= scale * dY[M][Cout]^T . col[M][K]  (f32, split over M, atomic) ----
// dW must be zeroed by the caller when splits > 1 (the flat gradient buffer is zeroed once per step anyway).
DRL_API int drl_conv_wgrad(int layer, const void* x, const long long* frame_off, int frames, const void* dy, float* dw,
                           int splits, hipStream_t s) {
    const int fl = EPI_OUT_F32 | EPI_ATOMIC;
    const int variant = splits >> 16;          // tile variant in the high half (tuning hook), split count in the low half
    splits &= 0xffff;
    if (layer == 1) {
        const int rows = frames * Conv1Op::OH * Conv1Op::OW;
        DenseOp A{(const bf16_t*)dy, 32, rows};
        const Epilogue ep = make_epi(dw, Conv1Op::K, nullptr, nullptr, 0, 1.0f / 255.0f, fl);
        if (frame_off) {
            Conv1OffOp B{x, frame_off, rows};
            // wide-N tiles read every dY row once per workgroup instead of once per 64 weight columns
            if (variant == 1) return launch<32, 256, 128, 1, 4, true, true, 2>(A, B, 32, Conv1Op::K, rows, splits, ep, s);
            if (variant == 2) return launch<32, 128, 128, 1, 4, true, true, 2>(A, B, 32, Conv1Op::K, rows, splits, ep, s);
            return launch<32, 64, 128, 1, 4, true, true, 4>(A, B, 32, Conv1Op::K, rows, splits, ep, s);
        }
        Conv1Op B{x, nullptr, rows};
        return launch<32, 64, 128, 1, 4, true, true, 4>(A, B, 32, Conv1Op::K, rows, splits, ep, s);
    } else if (layer == 2) {
        Conv2Op B{x, frame_off, frames * Conv2Op::OH * Conv2Op::OW};
        DenseOp A{(const bf16_t*)dy, 64, B.rows};
        return launch<64, 64, 128, 2, 2, true, true, 2>(A, B, 64, Conv2Op::K, B.rows, splits, make_epi(dw, Conv2Op::K, nullptr, nullptr, 0, 1.0f, fl), s);
    } else if (layer == 3) {
        Conv3Op B{x, frame_off, frames * Conv3Op::OH * Conv3Op::OW};
        DenseOp A{(const bf16_t*)dy, 64, B.rows};
        return launch<64, 64, 128, 2, 2, true, true, 2>(A, B, 64, Conv3Op::K, B.rows, splits, make_epi(dw, Conv3Op::K, nullptr, nullptr, 0, 1.0f, fl), s);
    }
    return -1;
}

// ---- conv input gradient, implicit form: dX = relu'(act_in) * dgrad(dY, W), one launch, no dCol / col2im ----
// layer 2: dY [frames][9][9][64], W [64][4][4][32] -> dX [frames][20][20][32];  layer 3: dY [frames][7][7][64],
// W [64][3][3][64] -> dX [frames][9][9][64]. act_in = forward input of the layer (ReLU mask), same shape as dX.
template <int H, int W, int KS, int S, int CI, int CO, int BM, int BN, int WM, int WN>
static int launch_dgrad(const void* dy, const void* w, const void* act_in, void* dx, int frames, hipStream_t s) {
    typedef DgradOp<H, W, KS, S, CO> AOp;
    typedef WdgradOp<KS, S, CI, CO> BOp;
    const int M = frames * AOp::HC * AOp::WC;
    if ((long long)frames * AOp::OH * AOp::OW * CO >= (1LL << 40)) return -8;
    AOp A{(const bf16_t*)dy, M, 0, 0};
    BOp B{(const bf16_t*)w, 0, 0};
    Epilogue ep = make_epi(dx, CI, nullptr, act_in, CI, 1.0f, (act_in ? EPI_MASK : 0) | EPI_REMAP);
    ep.rm_hc = AOp::HC; ep.rm_wc = AOp::WC; ep.rm_s = S; ep.rm_h = H; ep.rm_w = W;
    return launch<BM, BN, 64, WM, WN, false, true, 2>(A, B, M, CI, AOp::K, 1, ep, s, S * S);
}

DRL_API int drl_conv_dgrad(int layer, const void* dy, int frames, const void* w, const void* act_in, void* dx, hipStream_t s) {
    if (frames <= 0) return -1;
    if (layer == 2) return launch_dgrad<20, 20, 4, 2, 32, 64, 128, 32, 4, 1>(dy, w, act_in, dx, frames, s);
    if (layer == 3) return launch_dgrad<9, 9, 3, 1, 64, 64, 128, 64, 2, 2>(dy, w, act_in, dx, frames, s);
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
    return launch<64, 64, 256, 2, 2, false, false, 2>(A, B, M, 4 * H, H, 1, ep, s);
}

// ---------------------------------------------------------------------------------------------------
// Horizontally fused backward launches
// ---------------------------------------------------------------------------------------------------
// The backward of a layer is two or three INDEPENDENT kernels (input gradient, weight gradient, bias gradient), each
// too small to fill 256 CUs. Run on side streams they overlap, but every fork / join of a replayed hipGraph costs
// 5-11 us of cross-queue signalling on MI355X (profiles/apex_step_timeline_v3.txt) - a third of the kernels' own time.
// k_multi runs up to four such bodies as ONE launch: workgroup ranges [0,e0) [e0,e1) [e1,e2) [e2,grid) execute body
// 0..3; the launch allocates max(VGPR) / max(LDS) of its bodies. Dependent launches in one queue start back to back.
template <int BM, int BN, int BKT, int WM, int WN, bool A_TR, bool B_TR, int D, class AOp, class BOp>
struct GemmBody {
    struct Args { AOp a; BOp b; int Mo, No, Kc, kt_per, gx, gy; Epilogue ep; };
    static __device__ __forceinline__ void run(const Args& g, int block, uint8_t* smem) {
        const int bx = block % g.gx, r = block / g.gx, by = r % g.gy, bz = r / g.gy;
        gemm_tile<BM, BN, BKT, WM, WN, A_TR, B_TR, D, AOp, BOp>(g.a, g.b, g.Mo, g.No, g.Kc, g.kt_per, g.ep, bx, by, bz, g.gx, g.gy, smem);
    }
    // fills ``g``; returns the number of workgroups (or a negative error) and raises *lds to what this body needs
    static int plan(Args& g, const AOp& a, const BOp& b, int Mo, int No, int Kc, int splits, const Epilogue& ep, int classes, size_t* lds) {
        dim3 grid;
        size_t need;
        int kt_per;
        if (int rc = plan_gemm<BM, BN, BKT>(Mo, No, Kc, splits, ep, classes, &kt_per, &grid, &need)) return rc;
        g.a = a; g.b = b; g.Mo = Mo; g.No = No; g.Kc = Kc; g.kt_per = kt_per; g.gx = grid.x; g.gy = grid.y; g.ep = ep;
        if (need > *lds) *lds = need;
        return (int)(grid.x * grid.y * grid.z);
    }
};

template <int BM, int BN, int WM, int WN, bool A_TR, bool B_TR, int NS, class AOp, class BOp>
struct GldsBody {             // like GemmBody, LDS-DMA mainloop
    struct Args { AOp a; BOp b; int Mo, No, Kc, kt_per, gx, gy; Epilogue ep; };
    static __device__ __forceinline__ void run(const Args& g, int block, uint8_t* smem) {
        const int bx = block % g.gx, r = block / g.gx, by = r % g.gy, bz = r / g.gy;
        gemm_tile_glds<BM, BN, WM, WN, A_TR, B_TR, NS, AOp, BOp>(g.a, g.b, g.Mo, g.No, g.Kc, g.kt_per, g.ep, bx, by, bz, g.gx, g.gy, smem);
    }
    static int plan(Args& g, const AOp& a, const BOp& b, int Mo, int No, int Kc, int splits, const Epilogue& ep, int classes, size_t* lds) {
        dim3 grid;
        size_t need;
        int kt_per;
        if (int rc = plan_gemm<BM, BN, 64>(Mo, No, Kc, splits, ep, classes, &kt_per, &grid, &need)) return rc;
        need = (size_t)NS * (BM + BN) * 64 * 2;
        g.a = a; g.b = b; g.Mo = Mo; g.No = No; g.Kc = Kc; g.kt_per = kt_per; g.gx = grid.x; g.gy = grid.y; g.ep = ep;
        if (need > *lds) *lds = need;
        return (int)(grid.x * grid.y * grid.z);
    }
};

struct ColsumBody {
    struct Args { const bf16_t* dy; long long total; int NC8, cpb; float* db; };
    static __device__ __forceinline__ void run(const Args& g, int block, uint8_t* smem) {
        colsum_block(g.dy, g.total, g.NC8, g.cpb, g.db, block, reinterpret_cast<float*>(smem));
    }
    static int plan(Args& g, const void* dy, long long M, int N, float* db, size_t* lds) {
        if (M <= 0 || N < 8 || N > 2048 || (N & (N - 1))) return -1;
        g.dy = (const bf16_t*)dy; g.total = M * (N / 8); g.NC8 = N / 8; g.cpb = (int)colsum_cpb(g.total); g.db = db;
        if (*lds < 256 * 9 * 4) *lds = 256 * 9 * 4;
        return (int)((g.total + g.cpb - 1) / g.cpb);
    }
};

struct HeadDwBody {
    struct Args { const float* dq; const bf16_t* h; int M, A; float *gw_adv, *gb_adv, *gw_val, *gb_val; };
    static __device__ __forceinline__ void run(const Args& g, int block, uint8_t* smem) {
        head_dw_block(g.dq, g.h, g.M, g.A, g.gw_adv, g.gb_adv, g.gw_val, g.gb_val, block & 3, block >> 2, reinterpret_cast<float*>(smem));
    }
    static int plan(Args& g, const float* dq, const void* h, int M, int A, float* gw_adv, float* gb_adv, float* gw_val, float* gb_val,
                    size_t* lds) {
        if (!dq) return 0;
        if (M <= 0 || A <= 0 || A > HEAD_MAX_A) return -1;
        g.dq = dq; g.h = (const bf16_t*)h; g.M = M; g.A = A; g.gw_adv = gw_adv; g.gb_adv = gb_adv; g.gw_val = gw_val; g.gb_val = gb_val;
        if (*lds < DW_ROWS * (HEAD_MAX_A + 1) * 4) *lds = DW_ROWS * (HEAD_MAX_A + 1) * 4;
        return 4 * ((M + DW_ROWS - 1) / DW_ROWS);
    }
};

// One stage of the PER priority write-back (sumtree.hip) riding along with a backward launch: the write-back depends
// only on the TD kernel's output, so its two stages run inside two CONSECUTIVE fused launches of the conv backward.
struct TreeExtra {        // host-side description (ctypes mirror: ops/hip.py TreeExtra)
    int kind;             // 0 none | 1 leaves + level 1 | 2 levels >= 2 (+ st_end: step += 1) | 3 step += 1 only
    int n_levels;
    float* base;
    long long off[MAX_LEVELS];
    long long cap;
    const int* idx;
    const float* val;
    int n, pad;
    const int* gen;
    const int* gen_expected;
    void* st_end;
    // gradient-norm partials riding along (single-GPU Ape-X: no launch of their own): sq_blocks more workgroups write the
    // partial sums of squares of sq_g[0 .. sq_n) to sq_partial[0 .. sq_blocks) and zero zero_g[0 .. zero_n) (a gradient range
    // the optimizer launch leaves alone because all of its workgroups read it, optim.hip ``late``)
    const float* sq_g;
    long long sq_n;
    float* sq_partial;
    float* zero_g;
    long long zero_n;
    int sq_blocks, pad2;
};

static_assert(sizeof(TreeExtra) == 168, "TreeExtra is mirrored field by field by ops/hip.py (ctypes)");

__device__ __forceinline__ void sumsq_block(const float* __restrict__ g, long long n4, float* __restrict__ partial,
                                            float* __restrict__ zero_g, long long zero_n4, int block, int n_blocks, float* red) {
    const float4* g4 = reinterpret_cast<const float4*>(g);
    const long long nt = (long long)n_blocks * NTHREADS;
    long long i = (long long)block * NTHREADS + threadIdx.x;
    float acc = 0.f;
    for (; i + 3 * nt < n4; i += 4 * nt) {                      // four loads in flight per lane
        const float4 a = g4[i], b = g4[i + nt], c = g4[i + 2 * nt], d = g4[i + 3 * nt];
        acc += a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w + b.x * b.x + b.y * b.y + b.z * b.z + b.w * b.w;
        acc += c.x * c.x + c.y * c.y + c.z * c.z + c.w * c.w + d.x * d.x + d.y * d.y + d.z * d.z + d.w * d.w;
    }
    for (; i < n4; i += nt) {
        const float4 a = g4[i];
        acc += a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w;
    }
    for (long long z = (long long)block * NTHREADS + threadIdx.x; z < zero_n4; z += nt)
        reinterpret_cast<float4*>(zero_g)[z] = make_float4(0.f, 0.f, 0.f, 0.f);
    acc = block_sum(acc, red);
    if (threadIdx.x == 0) partial[block] = acc;
}

struct TreeBody {
    struct Args {
        TreeDesc tr; const int* idx; const float* val; int n, kind, n2_blocks; const int* gen; const int* gen_expected; StepState* st_end;
        const float* sq_g; long long sq_n4; float* sq_partial; float* zero_g; long long zero_n4; int sq_blocks, tree_blocks;
    };
    static __device__ __forceinline__ void run(const Args& g, int block, uint8_t* smem) {
        if (block >= g.tree_blocks) {
            sumsq_block(g.sq_g, g.sq_n4, g.sq_partial, g.zero_g, g.zero_n4, block - g.tree_blocks, g.sq_blocks, reinterpret_cast<float*>(smem));
            return;
        }
        if (g.kind == 1) tree_update_l01_block(g.tr, g.idx, g.val, g.n, g.gen, g.gen_expected, block);
        else if (g.kind == 2) tree_rebuild_upper_block(g.tr, g.n2_blocks, g.st_end, block, reinterpret_cast<float*>(smem));
        else if (block == 0 && threadIdx.x == 0 && g.st_end) g.st_end->step += 1;
    }
    static int plan_tree(Args& g, const TreeExtra* e, size_t* lds) {
        g.kind = 0;
        if (!e || e->kind == 0) return 0;
        if (e->n_levels < 1 || e->n_levels > MAX_LEVELS) return -1;
        g.tr = make_desc(e->base, e->n_levels, e->off, e->cap);
        g.idx = e->idx; g.val = e->val; g.n = e->n; g.kind = e->kind; g.gen = e->gen; g.gen_expected = e->gen_expected;
        g.st_end = (StepState*)e->st_end; g.n2_blocks = 0;
        if (*lds < 17 * 4) *lds = 17 * 4;
        if (e->kind == 1) return e->n > 0 ? (e->n + 3) / 4 : -1;
        if (e->kind == 2) {
            const long long blocks = tree_upper_blocks(g.tr, &g.n2_blocks);
            if (blocks > 0) return (int)blocks;
            g.kind = 3;
        }
        return g.st_end ? 1 : 0;
    }
    static int plan(Args& g, const TreeExtra* e, size_t* lds) {
        g.sq_blocks = 0; g.sq_g = nullptr; g.sq_partial = nullptr; g.zero_g = nullptr; g.sq_n4 = g.zero_n4 = 0;
        const int nt = plan_tree(g, e, lds);
        if (nt < 0) return nt;
        g.tree_blocks = nt;
        if (!e || e->sq_blocks <= 0) { g.tree_blocks = 1 << 30; return nt; }       // every block of this body is a tree block
        if (!e->sq_g || !e->sq_partial || e->sq_n <= 0 || (e->sq_n & 3) || e->sq_blocks > 1024 || e->zero_n < 0 || (e->zero_n & 3) ||
            (e->zero_n && !e->zero_g) || ((uintptr_t)e->sq_g & 15) || ((uintptr_t)e->zero_g & 15))
            return -1;
        g.sq_g = e->sq_g; g.sq_n4 = e->sq_n >> 2; g.sq_partial = e->sq_partial; g.zero_g = e->zero_g; g.zero_n4 = e->zero_n >> 2;
        g.sq_blocks = e->sq_blocks;
        if (*lds < 32 * 4) *lds = 32 * 4;
        return nt + e->sq_blocks;
    }
};

// PER sample of the NEXT step as a body (4 draws per 256-thread block): rides along with the conv1 backward launch and
// writes into the replay's "next" buffers, which the optimizer launch hands over (optim.hip CopyExtra) - the conv1
// weight gradient of the same launch still reads the CURRENT frame offsets.
struct SampleBody {
    typedef SampleArgs Args;
    static __device__ __forceinline__ void run(const Args& g, int block, uint8_t* smem) {
        sample_block(g.tr, g.st, g.seed, g.B, g.idx_out, g.prio_out, g.gen, g.gen_out, g.frame_off, g.win, g.begin_n_items,
                     g.is_n_scale, block, reinterpret_cast<float*>(smem));
    }
    static int plan(Args& g, const void* extra, size_t* lds) {
        const int n = fill_sample_args(extra, &g);
        if (n > 0 && *lds < 64) *lds = 64;
        return n;
    }
};

struct NoBody {
    struct Args { int unused; };
    static __device__ __forceinline__ void run(const Args&, int, uint8_t*) {}
};

template <class B0, class B1, class B2, class B3>
__global__ __launch_bounds__(NTHREADS) void k_multi(typename B0::Args a0, typename B1::Args a1, typename B2::Args a2,
                                                    typename B3::Args a3, int e0, int e1, int e2) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int b = blockIdx.x;
    if (b < e0) B0::run(a0, b, smem);
    else if (b < e1) B1::run(a1, b - e0, smem);
    else if (b < e2) B2::run(a2, b - e1, smem);
    else B3::run(a3, b - e2, smem);
}

template <class B0, class B1, class B2, class B3>
static int launch_multi_big(const typename B0::Args& a0, int n0, const typename B1::Args& a1, int n1, size_t lds, hipStream_t s) {
    // two-body launch whose LDS may exceed 64 KB (the LDS-DMA bodies)
    if (n0 < 0) return n0;
    if (n1 < 0) return n1;
    auto kern = k_multi<B0, B1, B2, B3>;
    if (lds > 64 * 1024) {
        static bool once = false;
        if (!once) {
            if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return -6;
            once = true;
        }
    }
    typename B2::Args z2{};
    typename B3::Args z3{};
    kern<<<n0 + n1, NTHREADS, lds, s>>>(a0, a1, z2, z3, n0, n0 + n1, n0 + n1);
    return (int)hipGetLastError();
}

template <class B0, class B1, class B2, class B3>
static int launch_multi(const typename B0::Args& a0, int n0, const typename B1::Args& a1, int n1, const typename B2::Args& a2, int n2,
                        const typename B3::Args& a3, int n3, size_t lds, hipStream_t s) {
    if (n0 < 0) return n0;
    if (n1 < 0) return n1;
    if (n2 < 0) return n2;
    if (n3 < 0) return n3;
    if (lds > 64 * 1024) return -6;
    const int total = n0 + n1 + n2 + n3;
    if (total <= 0) return -1;
    k_multi<B0, B1, B2, B3><<<total, NTHREADS, lds, s>>>(a0, a1, a2, a3, n0, n0 + n1, n0 + n1 + n2);
    return (int)hipGetLastError();
}

// ---- fused FC backward: dX = relu'(x) * (dY . W), dW = dY^T . x, db = colsum(dY) [+ dueling-head weight grads] ----------
// dY [M][N] bf16, x [M][K] bf16 (the layer's input: operand of dW and ReLU mask of dX), W [N][K] bf16, dW [N][K] f32
// (overwritten), db [N] f32 (accumulated: zeroed by the caller), dX [M][K] bf16. dq != null adds the weight gradients of
// the dueling head (head.hip, k_head_bwd_dw) on hq = the head's input [M][1024] as a fourth body.
template <int DXD, int DWD, bool DW_FIRST>
static int fc_bwd_launch(const void* dy, const void* x, const void* w, float* dw, float* db, void* dx, int M, int N, int K,
                         const float* dq, const void* hq, int A, float* gw_adv, float* gb_adv, float* gw_val, float* gb_val,
                         hipStream_t s) {
    typedef GemmBody<64, 64, 64, 2, 2, false, true, DXD, DenseOp, DenseOp> DX;
    typedef GemmBody<64, 64, 64, 2, 2, true, true, DWD, DenseOp, DenseOp> DW;
    size_t lds = 0;
    typename DX::Args ax; typename DW::Args aw; ColsumBody::Args ac; HeadDwBody::Args ah;
    const bf16_t* pdy = (const bf16_t*)dy;
    const int nx = DX::plan(ax, DenseOp{pdy, N, M}, DenseOp{(const bf16_t*)w, K, N}, M, K, N, 1,
                            make_epi(dx, K, nullptr, x, K, 1.0f, EPI_MASK), 0, &lds);
    const int nw = DW::plan(aw, DenseOp{pdy, N, M}, DenseOp{(const bf16_t*)x, K, M}, N, K, M, 1,
                            make_epi(dw, K, nullptr, nullptr, 0, 1.0f, EPI_OUT_F32), 0, &lds);
    const int nc = ColsumBody::plan(ac, dy, M, N, db, &lds);
    const int nh = HeadDwBody::plan(ah, dq, hq, M, A, gw_adv, gb_adv, gw_val, gb_val, &lds);
    // The small riders go FIRST: issued behind the GEMM tiles they start when the last tiles are already running and are a
    // tail of their own (dependent loads, a barrier, atomics: ~4 us for very little work) - 28.2 -> 24.0 us for this
    // launch alone, 0.1945 -> 0.1905 ms per step; with the conv2 launch's riders first as well 0.1893
    // (profiles/apex_riders_first_r3.txt; the conv3 / conv1 launches do not gain).
    if (DW_FIRST) return launch_multi<HeadDwBody, ColsumBody, DW, DX>(ah, nh, ac, nc, aw, nw, ax, nx, lds, s);
    return launch_multi<HeadDwBody, ColsumBody, DX, DW>(ah, nh, ac, nc, ax, nx, aw, nw, lds, s);
}

DRL_API int drl_fc_bwd(const void* dy, const void* x, const void* w, float* dw, float* db, void* dx, int M, int N, int K,
                       const float* dq, const void* hq, int A, float* gw_adv, float* gb_adv, float* gw_val, float* gb_val,
                       int variant, hipStream_t s) {
    if ((N & 7) || (K & 7) || (N % 64) || M <= 0) return -4;
    if ((long long)M * K >= (1LL << 32) || (long long)N * K >= (1LL << 32) || (long long)M * N >= (1LL << 32)) return -8;
#define FCB(a, b, c) return fc_bwd_launch<a, b, c>(dy, x, w, dw, db, dx, M, N, K, dq, hq, A, gw_adv, gb_adv, gw_val, gb_val, s)
    switch (variant) {        // ring depths of the dX / dW bodies and their order (tuning hook, see docs/DESIGN.md)
        case 0: FCB(4, 4, false);
        case 1: FCB(4, 2, false);
        case 2: FCB(2, 2, false);
        case 3: FCB(4, 4, true);
        case 4: FCB(4, 2, true);
        case 5: FCB(2, 2, true);
    }
#undef FCB
    return -7;
}

// ---- fused conv backward: dX (implicit dgrad, layers 2/3), dW (split-K atomics into a ZEROED buffer), db ---------------
// x = the layer's forward input (layer 1: uint8 frames, optionally through frame_off; layers 2/3: NHWC bf16 activation,
// which is also the ReLU mask of dX). dY = d(pre-activation) of this layer's output, NHWC bf16.
template <class WG, class BOp>
static int conv_bwd_12(const BOp& B, int cout, int kdim, float scale, const void* dy, float* dw, float* db, int splits,
                       const TreeExtra* extra, const void* sample, hipStream_t s) {
    size_t lds = 0;
    typename WG::Args aw; ColsumBody::Args ac; TreeBody::Args at; SampleBody::Args as;
    const int nt = TreeBody::plan(at, extra, &lds);
    const int ns = SampleBody::plan(as, sample, &lds);
    const int nw = WG::plan(aw, DenseOp{(const bf16_t*)dy, cout, B.rows}, B, cout, kdim, B.rows, splits,
                            make_epi(dw, kdim, nullptr, nullptr, 0, scale, EPI_OUT_F32 | EPI_ATOMIC), 0, &lds);
    const int nc = ColsumBody::plan(ac, dy, B.rows, cout, db, &lds);
    return launch_multi<WG, ColsumBody, TreeBody, SampleBody>(aw, nw, ac, nc, at, nt, as, ns, lds, s);
}

template <int H, int W, int KS, int S, int CI, int CO, int BM, int BN, int WM, int WN, class COp, int WBK, int WD, bool WG_FIRST>
static int conv_bwd_3(const void* x, int frames, const void* dy, const void* w, float* dw, float* db, void* dx, int splits,
                      const TreeExtra* extra, hipStream_t s) {
    typedef DgradOp<H, W, KS, S, CO> AOp;
    typedef WdgradOp<KS, S, CI, CO> BOp;
    typedef GemmBody<BM, BN, 64, WM, WN, false, true, 2, AOp, BOp> DG;
    typedef GemmBody<64, 64, WBK, 2, 2, true, true, WD, DenseOp, COp> WG;
    const int M = frames * AOp::HC * AOp::WC;
    if ((long long)frames * AOp::OH * AOp::OW * CO >= (1LL << 40)) return -8;
    size_t lds = 0;
    typename DG::Args ad; typename WG::Args aw; ColsumBody::Args ac; TreeBody::Args at;
    const int nt = TreeBody::plan(at, extra, &lds);
    Epilogue ep = make_epi(dx, CI, nullptr, x, CI, 1.0f, EPI_MASK | EPI_REMAP);
    ep.rm_hc = AOp::HC; ep.rm_wc = AOp::WC; ep.rm_s = S; ep.rm_h = H; ep.rm_w = W;
    const int nd = DG::plan(ad, AOp{(const bf16_t*)dy, M, 0, 0}, BOp{(const bf16_t*)w, 0, 0}, M, CI, AOp::K, 1, ep, S * S, &lds);
    COp B{x, nullptr, frames * COp::OH * COp::OW};
    const int nw = WG::plan(aw, DenseOp{(const bf16_t*)dy, CO, B.rows}, B, CO, COp::K, B.rows, splits,
                            make_epi(dw, COp::K, nullptr, nullptr, 0, 1.0f, EPI_OUT_F32 | EPI_ATOMIC), 0, &lds);
    const int nc = ColsumBody::plan(ac, dy, B.rows, CO, db, &lds);
    if constexpr (CI == 32) {      // conv2: riders first (see fc_bwd_launch); conv3's launch does not gain from it
        if (WG_FIRST) return launch_multi<TreeBody, ColsumBody, WG, DG>(at, nt, ac, nc, aw, nw, ad, nd, lds, s);
        return launch_multi<TreeBody, ColsumBody, DG, WG>(at, nt, ac, nc, ad, nd, aw, nw, lds, s);
    }
    if (WG_FIRST) return launch_multi<WG, DG, ColsumBody, TreeBody>(aw, nw, ad, nd, ac, nc, at, nt, lds, s);
    return launch_multi<DG, WG, ColsumBody, TreeBody>(ad, nd, aw, nw, ac, nc, at, nt, lds, s);
}

// extra (optional, host pointer to a TreeExtra): a stage of the PER priority write-back to run inside the same launch.
DRL_API int drl_conv_bwd(int layer, const void* x, const long long* frame_off, int frames, const void* dy, const void* w, float* dw,
                         float* db, void* dx, int splits, const void* extra_, const void* sample, hipStream_t s) {
    // sample (optional, layer 1 only; host pointer to a SampleExtra): the next step's PER batch drawn inside this launch
    const TreeExtra* extra = (const TreeExtra*)extra_;
    if (sample && layer != 1) return -1;
    if (frames <= 0) return -1;
    const int variant = splits >> 16;          // tile variant in the high half (tuning hook), split count in the low half
    splits &= 0xffff;
    if (layer == 1) {
        const int rows = frames * Conv1Op::OH * Conv1Op::OW;
        if (frame_off)
            return conv_bwd_12<GemmBody<32, 64, 128, 1, 4, true, true, 4, DenseOp, Conv1OffOp>>(Conv1OffOp{x, frame_off, rows}, 32, Conv1Op::K,
                                                                                              1.0f / 255.0f, dy, dw, db, splits, extra, sample, s);
        return conv_bwd_12<GemmBody<32, 64, 128, 1, 4, true, true, 4, DenseOp, Conv1Op>>(Conv1Op{x, nullptr, rows}, 32, Conv1Op::K,
                                                                                       1.0f / 255.0f, dy, dw, db, splits, extra, sample, s);
    }
    // dx == null (layers 2 / 3): weight + bias gradient (+ extra) only; the input gradient runs in the LDS-halo kernel
    // (conv_halo.hip, drl_conv_halo_dgrad)
    if (!dx) {
        const int wv = variant < 3 ? variant : (variant == 3 ? 1 : (variant == 4 ? 2 : 0));   // wgrad body of the fused variant
#define CW(COp, bk, d) return conv_bwd_12<GemmBody<64, 64, bk, 2, 2, true, true, d, DenseOp, COp>>(COp{x, nullptr, frames * COp::OH * COp::OW}, \
                                                                                                   64, COp::K, 1.0f, dy, dw, db, splits, extra, nullptr, s)
        if (layer == 2) { if (wv == 0) CW(Conv2Op, 128, 2); if (wv == 1) CW(Conv2Op, 64, 4); CW(Conv2Op, 64, 2); }
        if (layer == 3) { if (wv == 0) CW(Conv3Op, 128, 2); if (wv == 1) CW(Conv3Op, 64, 4); CW(Conv3Op, 64, 2); }
#undef CW
        return -1;
    }
#define CB2(bk, d, f) return conv_bwd_3<20, 20, 4, 2, 32, 64, 128, 32, 4, 1, Conv2Op, bk, d, f>(x, frames, dy, w, dw, db, dx, splits, extra, s)
#define CB3(bk, d, f) return conv_bwd_3<9, 9, 3, 1, 64, 64, 128, 64, 2, 2, Conv3Op, bk, d, f>(x, frames, dy, w, dw, db, dx, splits, extra, s)
    // variant (tuning hook): K-tile / ring depth of the weight-gradient body and the order of the bodies. The launch
    // allocates max(LDS) and max(VGPR) of its bodies for EVERY workgroup, so a fat wgrad body halves the number of
    // resident dgrad workgroups.
    if (layer == 2) switch (variant) {
        case 0: CB2(128, 2, false);
        case 1: CB2(64, 4, false);
        case 2: CB2(64, 2, false);
        case 3: CB2(64, 4, true);
        case 4: CB2(64, 2, true);
        case 5: CB2(128, 2, true);
    }
    if (layer == 3) switch (variant) {
        case 0: CB3(128, 2, false);
        case 1: CB3(64, 4, false);
        case 2: CB3(64, 2, false);
        case 3: CB3(64, 4, true);
        case 4: CB3(64, 2, true);
        case 5: CB3(128, 2, true);
    }
#undef CB2
#undef CB3
    return -1;
}

// ---------------------------------------------------------------------------------------------------
// One BPTT step of the LSTM in ONE launch (was: pointwise kernel + split-K GEMM, ~10 us per step of pure latency).
// ---------------------------------------------------------------------------------------------------
// dh_rec_{t-1}[m][h] = sum_n dG_t[m][n] * W_hh[n][h]  with  dG_t = pointwise(dh_head_t + dh_rec_t, dc_t, gates_t, c_t, c_{t-1}).
// grid = (H/64 output tiles j, H/64 K-slices z). Workgroup (j, z) recomputes the pointwise part for the 64 units of
// slice z (all 8 workgroups of a slice do the same cheap work: nobody has to wait for anybody inside the launch),
// writes the bf16 dG tile [64][256] straight into LDS in the A-operand layout, multiplies it with W_hh[256 z .., 64 j ..]
// (prefetched into registers before the pointwise phase) and adds its 64 x 64 partial sum into dh_rec_out through the
// LDS-staged 256-byte atomics of gemm_epilogue. The j == 0 workgroups also store dG (weight-gradient GEMMs later) and
// the new dc. Three rotating dh_rec buffers (read / accumulate / being zeroed) and two dc buffers make the launch free
// of any intra-launch ordering.
__global__ __launch_bounds__(NTHREADS) void k_lstm_bwd_step(const bf16_t* __restrict__ dh_head, const float* __restrict__ dh_rec_in,
                                                           float* __restrict__ dh_rec_out, float* __restrict__ dh_rec_zero,
                                                           const float* __restrict__ dc_in, float* __restrict__ dc_out,
                                                           const bf16_t* __restrict__ gates, const float* __restrict__ c_t,
                                                           const float* __restrict__ c_prev, bf16_t* __restrict__ dgates,
                                                           DenseOp w_hh, int M) {
    constexpr int HD = 512, UN = 64, KW = 4 * UN;              // units per slice, gate columns per slice
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint8_t* As = smem;                                         // [64][256] bf16, nrm layout (4 sub-tiles of 64 k)
    uint8_t* Bs = smem + 64 * KW * 2;                           // 4 TR tiles [64 k][64 n] bf16
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const int j = blockIdx.x, z = blockIdx.y;
    // W_hh tile loads first: they land while the pointwise phase runs
    RowCache<64, true, 64, DenseOp> rb;
    rb.init(w_hh, 64 * j, tid);
    Stage<64, true, 64, DenseOp> sb[4];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) sb[kt].load(w_hh, rb, 64 * j, KW * z + 64 * kt, 4 * HD, tid, true);
    // zero this workgroup's 2 KB piece of the buffer the NEXT step accumulates into
    {
        float* zp = dh_rec_zero + (size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 512;
        if ((blockIdx.y * gridDim.x + blockIdx.x) * 512 < M * HD)
            *reinterpret_cast<float2*>(zp + 2 * tid) = make_float2(0.f, 0.f);
    }
    // pointwise: thread = 4 consecutive units x 4 rows, every load 16 bytes wide and ALL of them issued before the first
    // use (one memory round trip for the whole phase: the launch is latency-bound, not bandwidth-bound)
    const int u4 = (tid & 15) * 4, ug = UN * z + u4;
    float4 v_rec[4], v_dc[4], v_ct[4], v_cp[4];
    uint4 v_g[4][2];
    uint2 v_dh[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = min((tid >> 4) + 16 * i, M - 1);           // rows >= M read row M-1 (results dropped below)
        const size_t e = (size_t)m * HD + ug;
        v_rec[i] = *reinterpret_cast<const float4*>(dh_rec_in + e);
        v_dc[i] = *reinterpret_cast<const float4*>(dc_in + e);
        v_ct[i] = *reinterpret_cast<const float4*>(c_t + e);
        v_cp[i] = *reinterpret_cast<const float4*>(c_prev + e);
        v_g[i][0] = *reinterpret_cast<const uint4*>(gates + e * 4);
        v_g[i][1] = *reinterpret_cast<const uint4*>(gates + e * 4 + 8);
        v_dh[i] = dh_head ? *reinterpret_cast<const uint2*>(dh_head + e) : make_uint2(0u, 0u);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = (tid >> 4) + 16 * i;
        const bool live = m < M;
        const size_t e = (size_t)min(m, M - 1) * HD + ug;
        const float rec[4] = {v_rec[i].x, v_rec[i].y, v_rec[i].z, v_rec[i].w}, dcv[4] = {v_dc[i].x, v_dc[i].y, v_dc[i].z, v_dc[i].w};
        const float ct[4] = {v_ct[i].x, v_ct[i].y, v_ct[i].z, v_ct[i].w}, cp[4] = {v_cp[i].x, v_cp[i].y, v_cp[i].z, v_cp[i].w};
        const uint32_t gw[8] = {v_g[i][0].x, v_g[i][0].y, v_g[i][0].z, v_g[i][0].w, v_g[i][1].x, v_g[i][1].y, v_g[i][1].z, v_g[i][1].w};
        const float dhh[4] = {__uint_as_float(v_dh[i].x << 16), __uint_as_float(v_dh[i].x & 0xffff0000u),
                              __uint_as_float(v_dh[i].y << 16), __uint_as_float(v_dh[i].y & 0xffff0000u)};
        uint32_t o[8];
        float dcn[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            dcn[q] = dcv[q];
            const uint2 dg = lstm_bwd_unit(rec[q] + dhh[q], dcn[q], gw[2 * q], gw[2 * q + 1], ct[q], cp[q]);
            o[2 * q] = live ? dg.x : 0u;
            o[2 * q + 1] = live ? dg.y : 0u;
        }
        if (j == 0 && live) {
            *reinterpret_cast<float4*>(dc_out + e) = make_float4(dcn[0], dcn[1], dcn[2], dcn[3]);
            *reinterpret_cast<uint4*>(dgates + e * 4) = make_uint4(o[0], o[1], o[2], o[3]);
            *reinterpret_cast<uint4*>(dgates + e * 4 + 8) = make_uint4(o[4], o[5], o[6], o[7]);
        }
        // A tile: row m, local gate columns 4 u4 .. 4 u4 + 15 = the two 16-B chunks u4/2, u4/2 + 1
        *reinterpret_cast<uint4*>(As + nrm_off<64>(m, u4 >> 1)) = make_uint4(o[0], o[1], o[2], o[3]);
        *reinterpret_cast<uint4*>(As + nrm_off<64>(m, (u4 >> 1) + 1)) = make_uint4(o[4], o[5], o[6], o[7]);
    }
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) sb[kt].store(Bs + kt * 8192, tid);
    __syncthreads();
    f32x4 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KW / 32; ++ks) {
        bf16x8 af[2], bf[2];
#pragma unroll
        for (int a = 0; a < 2; ++a) af[a] = read_frag<64, false>(As, wm * 32 + a * 16, ks, lane);
#pragma unroll
        for (int b = 0; b < 2; ++b) bf[b] = read_frag<64, true>(Bs + (ks >> 1) * 8192, wn * 32 + b * 16, ks & 1, lane);
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
                acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(mfma_bf16x8, bf[b]),
                                                                    __builtin_bit_cast(mfma_bf16x8, af[a]), acc[a][b], 0, 0, 0);
    }
    __syncthreads();
    const Epilogue ep = make_epi_dev(dh_rec_out, HD, 1.0f, EPI_OUT_F32 | EPI_ATOMIC);
    gemm_epilogue<64, 64, 2, 2>(acc, ep, 0, 64 * j, M, HD, smem, tid, lane, wm, wn);
}

// dh_head bf16 [M][H] (or null), dh_rec_* f32 [M][H] (three distinct buffers), dc_* f32 [M][H] (two distinct buffers),
// gates bf16 [M][4H] (post-activation, interleaved), c_t / c_prev f32 [M][H], dgates bf16 [M][4H], w_hh bf16 [4H][H].
DRL_API int drl_lstm_bwd_step(const void* dh_head, const float* dh_rec_in, float* dh_rec_out, float* dh_rec_zero,
                              const float* dc_in, float* dc_out, const void* gates, const float* c_t, const float* c_prev,
                              void* dgates, const void* w_hh, int M, int H, hipStream_t s) {
    if (H != 512 || M <= 0 || M > 64) return -1;
    if (dh_rec_in == dh_rec_out || dh_rec_out == dh_rec_zero || dh_rec_in == dh_rec_zero || dc_in == dc_out) return -2;
    DenseOp W{(const bf16_t*)w_hh, H, 4 * H};
    constexpr int LDS = 64 * 256 * 2 + 4 * 8192;
    k_lstm_bwd_step<<<dim3(H / 64, H / 64), NTHREADS, LDS, s>>>((const bf16_t*)dh_head, dh_rec_in, dh_rec_out, dh_rec_zero, dc_in,
                                                                dc_out, (const bf16_t*)gates, c_t, c_prev, (bf16_t*)dgates, W, M);
    return (int)hipGetLastError();
}

// ---- forward of TWO networks per launch (Ape-X learner: online net + target net, same layer, different weights) -------
// Same bodies as the single-network launches (conv2: register ring 128x64, conv3: LDS-DMA 128x64 x 3 stages, FC: LDS-DMA
// 64x64 x 4 stages); no fork / join between two streams is needed any more.
DRL_API int drl_conv_fwd_dual(int layer, const void* x1, int frames1, const void* w1, const float* bias1, void* out1,
                              const void* x2, int frames2, const void* w2, const float* bias2, void* out2, hipStream_t s) {
    if (frames1 <= 0 || frames2 <= 0) return -1;
    size_t lds = 0;
    if (layer == 2) {
        typedef GemmBody<128, 64, 64, 2, 2, false, false, 2, Conv2Op, DenseOp> BD;
        BD::Args a1, a2;
        Conv2Op A1{x1, nullptr, frames1 * Conv2Op::OH * Conv2Op::OW}, A2{x2, nullptr, frames2 * Conv2Op::OH * Conv2Op::OW};
        const int n1 = BD::plan(a1, A1, DenseOp{(const bf16_t*)w1, Conv2Op::K, 64}, A1.rows, 64, Conv2Op::K, 1,
                                make_epi(out1, 64, bias1, nullptr, 0, 1.0f, EPI_RELU), 0, &lds);
        const int n2 = BD::plan(a2, A2, DenseOp{(const bf16_t*)w2, Conv2Op::K, 64}, A2.rows, 64, Conv2Op::K, 1,
                                make_epi(out2, 64, bias2, nullptr, 0, 1.0f, EPI_RELU), 0, &lds);
        return launch_multi_big<BD, BD, NoBody, NoBody>(a1, n1, a2, n2, lds, s);
    }
    if (layer == 3) {
        typedef GldsBody<128, 64, 2, 2, false, false, 3, Conv3Op, DenseOp> BD;
        BD::Args a1, a2;
        Conv3Op A1{x1, nullptr, frames1 * Conv3Op::OH * Conv3Op::OW}, A2{x2, nullptr, frames2 * Conv3Op::OH * Conv3Op::OW};
        const int n1 = BD::plan(a1, A1, DenseOp{(const bf16_t*)w1, Conv3Op::K, 64}, A1.rows, 64, Conv3Op::K, 1,
                                make_epi(out1, 64, bias1, nullptr, 0, 1.0f, EPI_RELU), 0, &lds);
        const int n2 = BD::plan(a2, A2, DenseOp{(const bf16_t*)w2, Conv3Op::K, 64}, A2.rows, 64, Conv3Op::K, 1,
                                make_epi(out2, 64, bias2, nullptr, 0, 1.0f, EPI_RELU), 0, &lds);
        return launch_multi_big<BD, BD, NoBody, NoBody>(a1, n1, a2, n2, lds, s);
    }
    return -1;
}

// c_i[M_i][N] = relu(a_i[M_i][K] . w_i[N][K]^T + bias_i), i = 1, 2, one launch.
DRL_API int drl_gemm_nt_dual(const void* a1, int M1, const void* w1, const float* bias1, void* c1, const void* a2, int M2,
                             const void* w2, const float* bias2, void* c2, int N, int K, int flags, hipStream_t s) {
    if ((K & 7) || (N % 64) || M1 <= 0 || M2 <= 0) return -4;
    if ((long long)M1 * K >= (1LL << 32) || (long long)M2 * K >= (1LL << 32) || (long long)N * K >= (1LL << 32)) return -8;
    typedef GldsBody<64, 64, 2, 2, false, false, 4, DenseOp, DenseOp> BD;
    size_t lds = 0;
    BD::Args g1, g2;
    const int n1 = BD::plan(g1, DenseOp{(const bf16_t*)a1, K, M1}, DenseOp{(const bf16_t*)w1, K, N}, M1, N, K, 1,
                            make_epi(c1, N, bias1, nullptr, 0, 1.0f, flags), 0, &lds);
    const int n2 = BD::plan(g2, DenseOp{(const bf16_t*)a2, K, M2}, DenseOp{(const bf16_t*)w2, K, N}, M2, N, K, 1,
                            make_epi(c2, N, bias2, nullptr, 0, 1.0f, flags), 0, &lds);
    return launch_multi_big<BD, BD, NoBody, NoBody>(g1, n1, g2, n2, lds, s);
}

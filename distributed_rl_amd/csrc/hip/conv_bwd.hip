// Conv backward helpers (SURVEY G1-G3 backward): col2im gather for the input gradient and bias gradients.
// The heavy parts (dCol = dY.W, dW = dY^T.col) are MFMA GEMMs in gemm.hip; what is left is memory-bound:
//   * col2im as a GATHER: every input pixel sums the <= (KS/S)^2 dCol entries that touch it (no atomics, each
//     output written once, 16-B vector accesses over 8 channels), with the ReLU-backward mask of the layer input
//     fused so the result is directly d(pre-activation) of the previous layer;
//   * column sums of dY for the bias gradients.
#include "common.h"
#include "colsum_body.h"

template <int H, int W, int C, int KS, int S>
__global__ __launch_bounds__(256) void k_col2im(const bf16_t* __restrict__ dcol, const bf16_t* __restrict__ act_in,
                                               bf16_t* __restrict__ dx, int frames) {
    constexpr int OH = (H - KS) / S + 1, OW = (W - KS) / S + 1, K = KS * KS * C, C8 = C / 8;
    const long long total = (long long)frames * H * W * C8;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c8 = (int)(i % C8);
        long long pix = i / C8;
        const int x = (int)(pix % W);
        pix /= W;
        const int y = (int)(pix % H);
        const int b = (int)(pix / H);
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int kh = 0; kh < KS; ++kh) {
            const int ty = y - kh;
            if (ty < 0 || ty % S != 0) continue;
            const int oy = ty / S;
            if (oy >= OH) continue;
#pragma unroll
            for (int kw = 0; kw < KS; ++kw) {
                const int tx = x - kw;
                if (tx < 0 || tx % S != 0) continue;
                const int ox = tx / S;
                if (ox >= OW) continue;
                const uint4 v = *reinterpret_cast<const uint4*>(dcol + ((long long)(b * OH + oy) * OW + ox) * K + (kh * KS + kw) * C + c8 * 8);
                const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int t = 0; t < 4; ++t) { acc[2 * t] += __uint_as_float(w[t] << 16); acc[2 * t + 1] += __uint_as_float(w[t] & 0xffff0000u); }
            }
        }
        const long long o = (((long long)b * H + y) * W + x) * C + c8 * 8;
        const uint4 m = *reinterpret_cast<const uint4*>(act_in + o);
        const uint32_t mw[4] = {m.x, m.y, m.z, m.w};
        uint32_t out[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const uint32_t lo = mw[t] & 0xffffu, hi = mw[t] >> 16;
            const float a = (lo != 0 && !(lo & 0x8000u)) ? acc[2 * t] : 0.f;
            const float c = (hi != 0 && !(hi & 0x8000u)) ? acc[2 * t + 1] : 0.f;
            out[t] = pack_bf2(a, c);
        }
        *reinterpret_cast<uint4*>(dx + o) = make_uint4(out[0], out[1], out[2], out[3]);
    }
}

// db[n] += sum_m dY[m][n]: see colsum_body.h. grid = ceil(chunks / cpb), block = 256.
__global__ __launch_bounds__(256) void k_colsum(const bf16_t* __restrict__ dy, long long total, int NC8, int cpb,
                                               float* __restrict__ db) {
    __shared__ float red[256 * 9];
    colsum_block(dy, total, NC8, cpb, db, blockIdx.x, red);
}

// out[r][c] = a[r][c] * (mask[r][c] > 0)   (standalone ReLU backward, bf16)
__global__ __launch_bounds__(256) void k_relu_mask(const bf16_t* __restrict__ a, const bf16_t* __restrict__ mask,
                                                  bf16_t* __restrict__ out, long long n8) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long long)gridDim.x * blockDim.x) {
        uint4 v = reinterpret_cast<const uint4*>(a)[i], m = reinterpret_cast<const uint4*>(mask)[i];
        uint32_t vw[4] = {v.x, v.y, v.z, v.w}, mw[4] = {m.x, m.y, m.z, m.w};
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            uint32_t lo = mw[t] & 0xffffu, hi = mw[t] >> 16, keep = 0;
            if (lo != 0 && !(lo & 0x8000u)) keep |= 0xffffu;
            if (hi != 0 && !(hi & 0x8000u)) keep |= 0xffff0000u;
            vw[t] &= keep;
        }
        reinterpret_cast<uint4*>(out)[i] = make_uint4(vw[0], vw[1], vw[2], vw[3]);
    }
}

static int grid1d(long long n) {
    long long b = (n + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

// layer 2: dCol2 [frames*81][512] -> d(act1 pre-activation) [frames][20][20][32], masked by act1
// layer 3: dCol3 [frames*49][576] -> d(act2 pre-activation) [frames][9][9][64],  masked by act2
DRL_API int drl_col2im(int layer, const void* dcol, const void* act_in, void* dx, int frames, hipStream_t s) {
    if (frames <= 0) return -1;
    if (layer == 2) {
        k_col2im<20, 20, 32, 4, 2><<<grid1d((long long)frames * 20 * 20 * 4), 256, 0, s>>>((const bf16_t*)dcol, (const bf16_t*)act_in, (bf16_t*)dx, frames);
    } else if (layer == 3) {
        k_col2im<9, 9, 64, 3, 1><<<grid1d((long long)frames * 9 * 9 * 8), 256, 0, s>>>((const bf16_t*)dcol, (const bf16_t*)act_in, (bf16_t*)dx, frames);
    } else return -1;
    return (int)hipGetLastError();
}

DRL_API int drl_colsum(const void* dy, int M, int N, float* db, hipStream_t s) {
    if (M <= 0 || N < 8 || N > 2048 || (N & (N - 1))) return -1;   // N/8 must divide 256
    const long long total = (long long)M * (N / 8);
    const long long cpb = colsum_cpb(total);
    k_colsum<<<(unsigned)((total + cpb - 1) / cpb), 256, 0, s>>>((const bf16_t*)dy, total, N / 8, (int)cpb, db);
    return (int)hipGetLastError();
}

DRL_API int drl_relu_mask(const void* a, const void* mask, void* out, long long n, hipStream_t s) {
    if (n <= 0 || n % 8) return -1;
    k_relu_mask<<<grid1d(n / 8), 256, 0, s>>>((const bf16_t*)a, (const bf16_t*)mask, (bf16_t*)out, n / 8);
    return (int)hipGetLastError();
}

// Column-sum (bias gradient) block body, shared by k_colsum (conv_bwd.hip) and the fused backward launches (gemm.hip).
#pragma once
#include "common.h"

// db[n] += sum_m dY[m][n]. The matrix is walked as a flat stream of 16-B chunks (8 bf16 columns): with N/8 dividing
// 256 every thread stays on the same 8 columns, so it accumulates 8 floats in registers over its whole slice, the
// block folds its 256 partial rows through LDS and issues N atomics. ``red`` = 256*9 floats of LDS; ``block`` = index of
// this block among ceil(total / cpb).
__device__ __forceinline__ void colsum_block(const bf16_t* __restrict__ dy, long long total, int NC8, int cpb,
                                             float* __restrict__ db, int block, float* red) {
    const int tid = threadIdx.x;
    const long long c0 = (long long)block * cpb, c1 = min(total, c0 + cpb);
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    auto add = [&](const uint4& v) {
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int t = 0; t < 4; ++t) { acc[2 * t] += __uint_as_float(w[t] << 16); acc[2 * t + 1] += __uint_as_float(w[t] & 0xffff0000u); }
    };
    long long i = c0 + tid;
    for (; i + 768 < c1; i += 1024) {          // 4 independent 16-B loads in flight per thread
        const uint4 v0 = reinterpret_cast<const uint4*>(dy)[i], v1 = reinterpret_cast<const uint4*>(dy)[i + 256];
        const uint4 v2 = reinterpret_cast<const uint4*>(dy)[i + 512], v3 = reinterpret_cast<const uint4*>(dy)[i + 768];
        add(v0); add(v1); add(v2); add(v3);
    }
    for (; i < c1; i += 256) add(reinterpret_cast<const uint4*>(dy)[i]);
#pragma unroll
    for (int t = 0; t < 8; ++t) red[tid * 9 + t] = acc[t];
    __syncthreads();
    const int N = NC8 * 8, reps = 256 / NC8;
    for (int col = tid; col < N; col += 256) {
        const int g = col >> 3, e = col & 7;
        float sum = 0.f;
        for (int j = 0; j < reps; ++j) sum += red[(j * NC8 + g) * 9 + e];
        atomicAdd(&db[col], sum);
    }
}

// few blocks on purpose: same-address fp32 atomics serialise in L2 (~35 ns each on MI355X), so the number of blocks,
// not the bytes read, sets the time of the column sum. Returns chunks per block for ``total`` 16-B chunks.
static inline long long colsum_cpb(long long total) {
    long long cpb = ((total + 127) / 128 + 255) / 256 * 256;
    return cpb < 1024 ? 1024 : cpb;
}

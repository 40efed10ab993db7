// Dueling-head weight-gradient block body, shared by k_head_bwd_dw (head.hip) and the fused FC backward launch (gemm.hip).
#pragma once
#include "common.h"

#define DW_ROWS 16
#define HEAD_HID 512
#define HEAD_MAX_A 32

// Weight grads of the dueling head, accumulated with fp32 atomics into ZEROED buffers. Logical grid = (4, ceil(M/DW_ROWS));
// block = 256: thread owns hidden column k = bx*256 + tid of the 1024. ``s_d`` = DW_ROWS*(HEAD_MAX_A+1) floats of LDS.
__device__ __forceinline__ void head_dw_block(const float* __restrict__ dq, const bf16_t* __restrict__ h, int M, int A,
                                              float* __restrict__ gw_adv, float* __restrict__ gb_adv /* may be null */,
                                              float* __restrict__ gw_val, float* __restrict__ gb_val, int bx, int by, float* s_d) {
    constexpr int LD = HEAD_MAX_A + 1;           // [r][0..A-1] = dadv, [r][A] = dval
    const int r0 = by * DW_ROWS;
    const int nr = min(DW_ROWS, M - r0);
    for (int i = threadIdx.x; i < DW_ROWS * (A + 1); i += 256) {
        int r = i / (A + 1), j = i % (A + 1);
        float v = 0.f;
        if (r < nr) {
            const float* q = dq + (long long)(r0 + r) * A;
            float s = 0.f;
            for (int t = 0; t < A; ++t) s += q[t];
            v = (j < A) ? q[j] - s / (float)A : s;
        }
        s_d[r * LD + j] = v;
    }
    __syncthreads();
    const int k = bx * 256 + threadIdx.x;     // 0..1023
    const bool adv_half = k < HEAD_HID;
    float acc[HEAD_MAX_A];
#pragma unroll
    for (int j = 0; j < HEAD_MAX_A; ++j) acc[j] = 0.f;
    float accv = 0.f;
    for (int r = 0; r < nr; ++r) {
        const float hv = bf2f(h[(long long)(r0 + r) * (2 * HEAD_HID) + k]);
        if (adv_half) {
#pragma unroll
            for (int j = 0; j < HEAD_MAX_A; ++j) if (j < A) acc[j] += s_d[r * LD + j] * hv;
        } else {
            accv += s_d[r * LD + A] * hv;
        }
    }
    if (adv_half) {
#pragma unroll
        for (int j = 0; j < HEAD_MAX_A; ++j) if (j < A) atomicAdd(&gw_adv[(long long)j * HEAD_HID + k], acc[j]);
    } else {
        atomicAdd(&gw_val[k - HEAD_HID], accv);
    }
    if (bx == 0 && threadIdx.x <= A) {
        float s = 0.f;
        for (int r = 0; r < nr; ++r) s += s_d[r * LD + threadIdx.x];
        if (threadIdx.x < A) { if (gb_adv) atomicAdd(&gb_adv[threadIdx.x], s); }
        else atomicAdd(&gb_val[0], s);
    }
}

// Operand helpers shared by the conv1 kernels (conv_direct.hip, conv_torso.hip) and the weight pre-pass of the LDS-halo
// input-gradient kernels (conv_halo.hip; it rides along in whichever forward launch runs conv2).
#pragma once
#include "common.h"

typedef _Float16 mfma_f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f32x4 mfma16h(uint4 a, uint4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(mfma_f16x8, a), __builtin_bit_cast(mfma_f16x8, b), c, 0, 0, 0);
}
// 8 u8 -> 8 f16 holding 1024 + pixel: 4 x v_perm_b32 (the half 0x6400 | b is exactly 1024 + b)
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

// Weight pre-pass of the input-gradient kernels (conv_halo.hip MODE 1 / 2), once per optimizer step:
//   wt3[ci][a][b][co] = w3[co][2 - a][2 - b][ci]                                            (64 x 576)
//   wt2[py * 2 + px][ci][a][b][co] = w2[co][py + 2 (1 - a)][px + 2 (1 - b)][ci]              (4 x 32 x 256)
struct HaloPrep {
    const bf16_t* w2;
    const bf16_t* w3;
    bf16_t* wt2;
    bf16_t* wt3;
    int nblk;
};
constexpr int HALO_PREP_ELEMS = 64 * 576 + 4 * 32 * 256, HALO_PREP_BLOCKS = 34;      // 34 x 256 threads x 8 elements

__device__ __forceinline__ void halo_wprep_elem(const HaloPrep& p, int i) {
    if (i < 64 * 576) {
        const int co = i & 63, b = (i >> 6) % 3, a = (i / 192) % 3, ci = i / 576;
        p.wt3[i] = p.w3[((co * 3 + (2 - a)) * 3 + (2 - b)) * 64 + ci];
    } else if (i < HALO_PREP_ELEMS) {
        const int j = i - 64 * 576;
        const int co = j & 63, b = (j >> 6) & 1, a = (j >> 7) & 1, ci = (j >> 8) & 31, cls = j >> 13, py = cls >> 1, px = cls & 1;
        p.wt2[j] = p.w2[((co * 4 + py + 2 * (1 - a)) * 4 + px + 2 * (1 - b)) * 32 + ci];
    }
}

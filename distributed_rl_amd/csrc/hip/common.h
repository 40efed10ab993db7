// Shared device helpers for the gfx950 kernels. Wave = 64 lanes everywhere.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define DRL_API extern "C" __attribute__((visibility("default")))
#define WAVE 64

typedef __attribute__((ext_vector_type(8))) short bf16x8;   // 8 bf16 = 4 VGPRs (MFMA A/B fragment)
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;    // 16x16 MFMA accumulator
typedef __attribute__((ext_vector_type(16))) float f32x16;  // 32x32 MFMA accumulator
typedef unsigned short bf16_t;

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }

// round-to-nearest-even f32 -> bf16 in hardware: gfx950 has v_cvt_pk_bf16_f32 (two conversions per instruction)
typedef __bf16 hw_bf16x2 __attribute__((ext_vector_type(2)));
typedef float hw_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) {
    const hw_f32x2 f = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, hw_bf16x2));
}
__device__ __forceinline__ bf16_t f2bf(float f) { return (bf16_t)(pack_bf2(f, 0.f) & 0xffffu); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, WAVE);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, WAVE));
    return v;
}
// inclusive prefix sum across the 64 lanes
__device__ __forceinline__ float wave_scan_incl(float v, int lane) {
#pragma unroll
    for (int o = 1; o < WAVE; o <<= 1) {
        float t = __shfl_up(v, o, WAVE);
        if (lane >= o) v += t;
    }
    return v;
}

// block-wide sum for blockDim.x <= 1024 (result valid in every thread)
__device__ __forceinline__ float block_sum(float v, float* smem /* >= 17 floats */) {
    int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    v = wave_sum(v);
    __syncthreads();
    if (lane == 0) smem[w] = v;
    __syncthreads();
    float t = (threadIdx.x < nw) ? smem[threadIdx.x] : 0.f;
    if (w == 0) { t = wave_sum(t); if (lane == 0) smem[16] = t; }
    __syncthreads();
    return smem[16];
}
__device__ __forceinline__ float block_max(float v, float* smem) {
    int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    v = wave_max(v);
    __syncthreads();
    if (lane == 0) smem[w] = v;
    __syncthreads();
    float t = (threadIdx.x < nw) ? smem[threadIdx.x] : -3.0e38f;
    if (w == 0) { t = wave_max(t); if (lane == 0) smem[16] = t; }
    __syncthreads();
    return smem[16];
}

// Backward of one LSTM unit at one time step, shared by the BPTT kernels (lstm.hip, gemm.hip k_lstm_bwd_step,
// lstm_seq.hip k_lstm_bptt). g_if / g_go = the saved post-activation gates packed as bf16 pairs (i, f) and (g, o);
// dh = dL/dh_t (heads + recurrent part), dc = dL/dc_t carried from step t+1 (in) -> dL/dc_{t-1} (out).
// Returns the four gate gradients packed the same way.
__device__ __forceinline__ uint2 lstm_bwd_unit(float dh, float& dc, uint32_t g_if, uint32_t g_go, float c_t, float c_prev) {
    const float ig = __uint_as_float(g_if << 16), fg = __uint_as_float(g_if & 0xffff0000u);
    const float gg = __uint_as_float(g_go << 16), og = __uint_as_float(g_go & 0xffff0000u);
    const float tc = 2.0f / (1.0f + __expf(-2.0f * c_t)) - 1.0f;
    const float dcell = dh * og * (1.0f - tc * tc) + dc;
    const float d_o = dh * tc * og * (1.0f - og);
    const float d_i = dcell * gg * ig * (1.0f - ig);
    const float d_g = dcell * ig * (1.0f - gg * gg);
    const float d_f = dcell * c_prev * fg * (1.0f - fg);
    dc = dcell * fg;
    return make_uint2(pack_bf2(d_i, d_f), pack_bf2(d_g, d_o));
}

// Philox-4x32-10 counter RNG (device-side ε-greedy and PER sampling; no host RNG on the hot path)
struct Philox {
    uint32_t k0, k1;
    __device__ __forceinline__ Philox(uint64_t seed) : k0((uint32_t)seed), k1((uint32_t)(seed >> 32)) {}
    __device__ __forceinline__ uint4 operator()(uint64_t ctr, uint64_t stream) const {
        uint32_t c0 = (uint32_t)ctr, c1 = (uint32_t)(ctr >> 32), c2 = (uint32_t)stream, c3 = (uint32_t)(stream >> 32);
        uint32_t a = k0, b = k1;
#pragma unroll
        for (int r = 0; r < 10; ++r) {
            uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
            uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ a, n1 = (uint32_t)p1;
            uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ b, n3 = (uint32_t)p0;
            c0 = n0; c1 = n1; c2 = n2; c3 = n3;
            a += 0x9E3779B9u; b += 0xBB67AE85u;
        }
        return make_uint4(c0, c1, c2, c3);
    }
};
// uint32 -> uniform float in [0,1)
__device__ __forceinline__ float u01(uint32_t x) { return (float)(x >> 8) * (1.0f / 16777216.0f); }

// Device-resident per-step scalars, advanced on device so a captured graph needs no host arguments.
struct StepState {
    unsigned long long step;     // learner step t
    unsigned long long rng_ctr;  // Philox counter
    float beta;                  // IS exponent at this step
    float beta0, beta_inc;       // beta = min(1, beta0 + beta_inc * t)
    float total_prio;            // root of the tree at sample time
    long long n_items;           // live items at sample time
    float loss;                  // loss of EVEN steps (written / atomically summed by the TD kernels, see loss_slot)
    float grad_norm;             // last global grad norm (written by the optimizer)
    float max_w;                 // max IS weight of the last batch (pre-normalisation)
    float loss_odd;              // loss of ODD steps: two slots, so the sampler of step t+1 may run (and clear its slot) before
                                 // the host has read the loss of step t
    // max IS weight of the batch being sampled, as float bits (positive floats order like unsigned ints), accumulated
    // with atomicMax by the sampling kernel: slot [step & 1] belongs to this step, the other one is cleared for the next
    unsigned wmax_bits[2];
};
__device__ __forceinline__ float* loss_slot(StepState* st, unsigned long long step) { return (step & 1) ? &st->loss_odd : &st->loss; }
static_assert(sizeof(StepState) == 64, "StepState is mirrored by ops/hip.py (STEP_STATE_BYTES)");


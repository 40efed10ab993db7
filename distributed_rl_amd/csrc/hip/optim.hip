// Fused global-norm clip + optimizer + bf16 re-cast over ONE flat parameter buffer (SURVEY G11, G12, G17;
// reference: clip_grad_norm_(…,40) ape_x/learner.py:77, RMSprop learner_node.py:28, Adam learner_node.py:40).
//
// All parameters / gradients / optimizer moments live back to back in flat fp32 buffers (models/packing.py), so
// the 14-17 per-tensor launches of a framework optimizer collapse to two: a partial sum-of-squares pass and an
// update pass that (a) finishes the norm reduction in-block, (b) derives the clip coefficient, (c) applies the
// update to the fp32 master weights and (d) writes the bf16 copy the MFMA kernels read — so there is no separate
// cast pass and no host synchronisation for the norm.
#include "common.h"
#include "sumtree_body.h"

#define NPART 1024

// Up to four device words that turn non-zero when a bounded in-kernel wait of this step gave up (persistent LSTM / BPTT
// hand-offs, the direct xGMI all-reduce): block 0 folds them into the STICKY fault word the update kernel checks.
struct Guards { const int* g[4]; };

__global__ __launch_bounds__(256) void k_sumsq_partial(const float* __restrict__ g, long long n, float* __restrict__ partial,
                                                      Guards gd, int* __restrict__ fault) {
    __shared__ float red[32];
    if (fault && blockIdx.x == 0 && threadIdx.x == 0) {
        int f = *fault;
#pragma unroll
        for (int k = 0; k < 4; ++k) if (gd.g[k] && *gd.g[k]) f |= 1 << k;
        *fault = f;
        // host mirror: fault[2..3] hold the address of a pinned, device-mapped host word (engine/guard.py). Written from
        // here - a memcpy node of the step graph for these 4 bytes costs 4 us per step
        int* mirror = reinterpret_cast<int*>(*reinterpret_cast<const unsigned long long*>(fault + 2));
        if (mirror) __hip_atomic_store(mirror, f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    float acc = 0.f;
    const long long n4 = n >> 2;
    const float4* g4 = reinterpret_cast<const float4*>(g);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        float4 v = g4[i];
        acc += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    if (blockIdx.x == 0) for (long long i = (n4 << 2) + threadIdx.x; i < n; i += blockDim.x) acc += g[i] * g[i];
    acc = block_sum(acc, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = acc;
}

// ``late``: the first late_n4 float4 of the gradient are not covered by the partials (they became final only in the launch
// before this one): every workgroup adds their squares itself (a few KB out of L2).
__device__ __forceinline__ float clip_coef_from_partials(const float* __restrict__ partial, int n_partial, float grad_scale,
                                                         float max_norm, float* red, float* norm_out,
                                                         const float4* __restrict__ late = nullptr, long long late_n4 = 0) {
    float acc = 0.f;
    for (int i = threadIdx.x; i < n_partial; i += blockDim.x) acc += partial[i];
    for (long long i = threadIdx.x; i < late_n4; i += blockDim.x) {
        const float4 v = late[i];
        acc += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    acc = block_sum(acc, red);
    const float norm = sqrtf(acc) * grad_scale;
    *norm_out = norm;
    return max_norm > 0.f ? fminf(1.0f, max_norm / (norm + 1e-6f)) : 1.0f;
}

// Hand-over of the pre-drawn PER batch (see SampleBody in gemm.hip): up to four small device-to-device copies done by ONE
// extra block of the optimizer launch. Host-side description, mirrored by ops/hip.py (ctypes).
struct CopyExtra {
    int n, pad;
    void* dst[4];
    const void* src[4];
    long long bytes[4];            // multiples of 16
};
static_assert(sizeof(CopyExtra) == 104, "CopyExtra is mirrored by ops/hip.py (ctypes)");

// mode 0: RMSprop (alpha, eps; no momentum, not centered).  mode 1: Adam (beta1, beta2, eps, bias correction).
// The last n_opt blocks update the parameters; one optional block in front of them copies the pre-drawn PER batch of the
// next step into the live sample buffers (by the time the optimizer runs nothing of the current step reads them any more).
template <int MODE>
__global__ __launch_bounds__(256) void k_opt_step(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m1,
                                                 float* __restrict__ m2, bf16_t* __restrict__ p_bf16, long long n,
                                                 const float* __restrict__ partial, int n_partial, float grad_scale,
                                                 float max_norm, float lr, float a1, float a2, float eps, StepState* st,
                                                 int zero_grad, int n_opt, CopyExtra cp, const int* __restrict__ fault,
                                                 long long keep0, long long keep1, long long late_n4) {
    __shared__ float red[32];
    // optional extra block in front of the update blocks: hands the pre-drawn batch of the next step over (copies)
    const int n_smp = (int)gridDim.x - n_opt;
    if ((int)blockIdx.x < n_smp) {
        for (int c = 0; c < cp.n; ++c) {
            const uint4* src = reinterpret_cast<const uint4*>(cp.src[c]);
            uint4* dst = reinterpret_cast<uint4*>(cp.dst[c]);
            for (long long i = threadIdx.x; i < cp.bytes[c] / 16; i += blockDim.x) dst[i] = src[i];
        }
        return;
    }
    const int ob = (int)blockIdx.x - n_smp;          // index among the update blocks
    if (fault && *fault) {
        // a wait inside this step timed out: its gradients are garbage. Discard the step - parameters, moments and the
        // bf16 shadow stay as they are - but still hand the next step a zeroed gradient buffer. The word is sticky: every
        // later step is discarded too until the host (which sees a pinned mirror of it) raises.
        if (zero_grad) {
            float4* g4z = reinterpret_cast<float4*>(g);
            for (long long i = (long long)ob * blockDim.x + threadIdx.x; i < (n >> 2); i += (long long)n_opt * blockDim.x)
                g4z[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        return;
    }
    const long long n4 = n >> 2, stride = (long long)n_opt * blockDim.x;
    float4* p4 = reinterpret_cast<float4*>(p);
    float4* g4 = reinterpret_cast<float4*>(g);
    float4* m14 = reinterpret_cast<float4*>(m1);
    float4* m24 = reinterpret_cast<float4*>(m2);
    uint2* pb4 = reinterpret_cast<uint2*>(p_bf16);
    // the first element's operands are requested BEFORE the norm is reduced (partials, ``late`` range, two block-wide
    // reductions: ~1.5 us that every workgroup would otherwise spend with nothing in flight), and inside the loop the next
    // element's operands are requested before the current one is computed
    long long i = (long long)ob * blockDim.x + threadIdx.x;
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 pv = z4, gv = z4, mv = z4, vv = z4;
    if (i < n4) {
        pv = p4[i]; gv = g4[i]; mv = m14[i];
        if (MODE == 1) vv = m24[i];
    }
    float norm;
    const float coef = clip_coef_from_partials(partial, n_partial, grad_scale, max_norm, red, &norm,
                                               reinterpret_cast<const float4*>(g), late_n4) * grad_scale;
    if (ob == 0 && threadIdx.x == 0 && st) st->grad_norm = norm;
    float bc1 = 1.f, bc2s = 1.f;
    if (MODE == 1) {
        const float t = (float)(st->step + 1);
        bc1 = 1.0f - __powf(a1, t);
        bc2s = sqrtf(1.0f - __powf(a2, t));
    }
    for (; i < n4; i += stride) {
        const long long nx = i + stride;
        float4 pn = z4, gn = z4, mn = z4, vn = z4;
        if (nx < n4) {
            pn = p4[nx]; gn = g4[nx]; mn = m14[nx];
            if (MODE == 1) vn = m24[nx];
        }
        // the split-K / bias-gradient kernels of the NEXT step accumulate with atomics: hand them a zeroed buffer from
        // this pass (the line is in flight anyway) instead of a separate memset launch. Elements [keep0, keep1) are
        // OVERWRITTEN by the next step (plain GEMM outputs: fc1.w, lstm.w_ih / w_hh - 96 % / 92 % of the buffer): no store
        // ... and the ``late`` range, which ALL workgroups of this launch read for the norm, is zeroed by a later launch
        if (zero_grad && !(i >= keep0 && i < keep1) && i >= late_n4) g4[i] = z4;
        float pe[4] = {pv.x, pv.y, pv.z, pv.w}, ge[4] = {gv.x * coef, gv.y * coef, gv.z * coef, gv.w * coef};
        if (MODE == 0) {
            float se[4] = {mv.x, mv.y, mv.z, mv.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                se[k] = a1 * se[k] + (1.0f - a1) * ge[k] * ge[k];
                pe[k] -= lr * ge[k] / (sqrtf(se[k]) + eps);
            }
            m14[i] = make_float4(se[0], se[1], se[2], se[3]);
        } else {
            float me[4] = {mv.x, mv.y, mv.z, mv.w}, ve[4] = {vv.x, vv.y, vv.z, vv.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                me[k] = a1 * me[k] + (1.0f - a1) * ge[k];
                ve[k] = a2 * ve[k] + (1.0f - a2) * ge[k] * ge[k];
                pe[k] -= (lr / bc1) * me[k] / (sqrtf(ve[k]) / bc2s + eps);
            }
            m14[i] = make_float4(me[0], me[1], me[2], me[3]);
            m24[i] = make_float4(ve[0], ve[1], ve[2], ve[3]);
        }
        p4[i] = make_float4(pe[0], pe[1], pe[2], pe[3]);
        if (p_bf16) pb4[i] = make_uint2(pack_bf2(pe[0], pe[1]), pack_bf2(pe[2], pe[3]));
        pv = pn; gv = gn; mv = mn; vv = vn;
    }
}

__global__ __launch_bounds__(256) void k_cast_bf16(const float* __restrict__ src, bf16_t* __restrict__ dst, long long n) {
    const long long n4 = n >> 2;
    const float4* s4 = reinterpret_cast<const float4*>(src);
    uint2* d4 = reinterpret_cast<uint2*>(dst);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        float4 v = s4[i];
        d4[i] = make_uint2(pack_bf2(v.x, v.y), pack_bf2(v.z, v.w));
    }
    if (blockIdx.x == 0) for (long long i = (n4 << 2) + threadIdx.x; i < n; i += blockDim.x) dst[i] = f2bf(src[i]);
}

static int grid_for(long long n) {
    long long b = (n / 4 + 255) / 256;
    if (b < 1) b = 1;
    if (b > NPART) b = NPART;
    return (int)b;
}

DRL_API int drl_grad_sumsq(const float* g, long long n, float* partial /* NPART floats */, int* n_partial_out,
                           const int* g0, const int* g1, const int* g2, const int* g3, int* fault, hipStream_t s) {
    if (n <= 0 || (n & 3)) return -1;       // flat buffers are padded to 64 elements
    int grid = grid_for(n);
    Guards gd = {{g0, g1, g2, g3}};
    k_sumsq_partial<<<grid, 256, 0, s>>>(g, n, partial, gd, fault);
    if (n_partial_out) *n_partial_out = grid;
    return (int)hipGetLastError();
}

// copies (optional, host pointer to a CopyExtra): done by one extra block of this launch. [keep0, keep1): elements of g that
// zero_grad leaves alone (multiples of 4; the caller guarantees that the next step overwrites them).
DRL_API int drl_rmsprop_step(float* p, float* g, float* sq, void* p_bf16, long long n, const float* partial,
                             int n_partial, float grad_scale, float max_norm, float lr, float alpha, float eps, void* st,
                             int zero_grad, const void* copies, const int* fault, long long keep0, long long keep1, long long late,
                             hipStream_t s) {
    // late (elements, multiple of 4): g[0 .. late) is not covered by ``partial``; the launch sums its squares itself and
    // leaves it un-zeroed (the caller zeroes it before the next accumulation: TreeExtra::zero_g in gemm.hip)
    if (n <= 0 || (n & 3) || (keep0 & 3) || (keep1 & 3) || late < 0 || (late & 3) || late > n || late > (1 << 16)) return -1;
    CopyExtra cp;
    cp.n = 0;
    if (copies) {
        cp = *(const CopyExtra*)copies;
        if (cp.n < 0 || cp.n > 4) return -2;
        for (int c = 0; c < cp.n; ++c) if (cp.bytes[c] & 15) return -2;
    }
    const int no = grid_for(n), ns = cp.n > 0 ? 1 : 0;       // 512 .. 2048 update workgroups: the same 16.3 us
    k_opt_step<0><<<no + ns, 256, 0, s>>>(p, g, sq, nullptr, (bf16_t*)p_bf16, n, partial, n_partial, grad_scale, max_norm,
                                          lr, alpha, 0.f, eps, (StepState*)st, zero_grad, no, cp, fault, keep0 >> 2, keep1 >> 2, late >> 2);
    return (int)hipGetLastError();
}

DRL_API int drl_adam_step(float* p, float* g, float* m, float* v, void* p_bf16, long long n, const float* partial,
                          int n_partial, float grad_scale, float max_norm, float lr, float beta1, float beta2, float eps,
                          void* st, int zero_grad, const int* fault, long long keep0, long long keep1, hipStream_t s) {
    if (n <= 0 || (n & 3) || !st || (keep0 & 3) || (keep1 & 3)) return -1;
    CopyExtra cp;
    cp.n = 0;
    const int no = grid_for(n);
    k_opt_step<1><<<no, 256, 0, s>>>(p, g, m, v, (bf16_t*)p_bf16, n, partial, n_partial, grad_scale, max_norm, lr,
                                     beta1, beta2, eps, (StepState*)st, zero_grad, no, cp, fault, keep0 >> 2, keep1 >> 2, 0);
    return (int)hipGetLastError();
}

DRL_API int drl_cast_bf16(const float* src, void* dst, long long n, hipStream_t s) {
    if (n <= 0) return -1;
    k_cast_bf16<<<grid_for(n), 256, 0, s>>>(src, (bf16_t*)dst, n);
    return (int)hipGetLastError();
}

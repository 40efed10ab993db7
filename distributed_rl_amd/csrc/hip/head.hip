// Dueling head: FC 512->A (advantage) + FC 512->1 (value) + combine + (optionally) batched epsilon-greedy,
// and its backward (reference: libs/models.py:26-29,101-104, libs/utils.py:36-41).
//
// Input is the post-ReLU hidden row h[1024] (bf16) written by the fused FC1 GEMM epilogue: h[0:512] = adv
// branch, h[512:1024] = val branch. N = A+1 <= 19 outputs is far too narrow for an MFMA tile, so one wave owns
// one row: 16 elements per lane, A+1 running dot products, wave-wide reductions, and the combine / argmax /
// epsilon-greedy draw in the same wave — Q never takes a second kernel, and actor inference ends here with the
// action already selected on device (per-env epsilon vector, Philox RNG; no host `random`).
#include "common.h"
#include "head_bwd_body.h"

#define MAX_A 32
#define HID 512

__device__ __forceinline__ void load16_bf16(const bf16_t* p, float* out) {
    const uint4* q = reinterpret_cast<const uint4*>(p);
    uint4 a = q[0], b = q[1];
    uint32_t w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
    for (int i = 0; i < 8; ++i) { out[2 * i] = __uint_as_float(w[i] << 16); out[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u); }
}

// one wave = one row: Q_j of the row in lane j (j < A); hv = the lane's 16 elements of the hidden row
__device__ __forceinline__ float head_q_row(const bf16_t* __restrict__ h, int row, int A, const bf16_t* __restrict__ w_adv,
                                            const float* __restrict__ b_adv /* may be null */, const bf16_t* __restrict__ w_val,
                                            const float* __restrict__ b_val, float (&hv)[16]) {
    const int lane = threadIdx.x & 63;
    load16_bf16(h + (long long)row * (2 * HID) + lane * 16, hv);
    const bool adv_half = lane < 32;
    const int k0 = (lane & 31) * 16;
    float wv[16];
    float val = 0.f;
    if (!adv_half) {
        load16_bf16(w_val + k0, wv);
#pragma unroll
        for (int i = 0; i < 16; ++i) val += hv[i] * wv[i];
    }
    val = wave_sum(val) + b_val[0];
    float my_q = 0.f, adv_sum = 0.f;       // lane j keeps adv_j
    for (int j = 0; j < A; ++j) {
        float acc = 0.f;
        if (adv_half) {
            load16_bf16(w_adv + (long long)j * HID + k0, wv);
#pragma unroll
            for (int i = 0; i < 16; ++i) acc += hv[i] * wv[i];
        }
        acc = wave_sum(acc) + (b_adv ? b_adv[j] : 0.f);
        adv_sum += acc;
        if (lane == j) my_q = acc;
    }
    return val + my_q - adv_sum / (float)A;
}

// one wave = one row
__device__ __forceinline__ void head_fwd_row(const bf16_t* __restrict__ h, int row, int A, const bf16_t* __restrict__ w_adv,
                                             const float* __restrict__ b_adv /* may be null */, const bf16_t* __restrict__ w_val,
                                             const float* __restrict__ b_val, float* __restrict__ q_out,
                                             int* __restrict__ action_out, const float* __restrict__ eps, int eps_mod,
                                             unsigned long long seed, const unsigned long long* __restrict__ ctr_ptr) {
    const int lane = threadIdx.x & 63;
    float hv[16];
    load16_bf16(h + (long long)row * (2 * HID) + lane * 16, hv);
    const bool adv_half = lane < 32;
    const int k0 = (lane & 31) * 16;
    float wv[16];
    float val = 0.f;
    if (!adv_half) {
        load16_bf16(w_val + k0, wv);
#pragma unroll
        for (int i = 0; i < 16; ++i) val += hv[i] * wv[i];
    }
    val = wave_sum(val) + b_val[0];
    float my_q = 0.f, adv_sum = 0.f;       // lane j keeps adv_j
    for (int j = 0; j < A; ++j) {
        float acc = 0.f;
        if (adv_half) {
            load16_bf16(w_adv + (long long)j * HID + k0, wv);
#pragma unroll
            for (int i = 0; i < 16; ++i) acc += hv[i] * wv[i];
        }
        acc = wave_sum(acc) + (b_adv ? b_adv[j] : 0.f);
        adv_sum += acc;
        if (lane == j) my_q = acc;
    }
    my_q = val + my_q - adv_sum / (float)A;
    if (lane < A) q_out[(long long)row * A + lane] = my_q;
    if (action_out) {
        // argmax (first max wins, like torch.argmax) over lanes [0,A)
        float best = lane < A ? my_q : -3.0e38f;
        int bi = lane < A ? lane : MAX_A + 64;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            float ob = __shfl_xor(best, o, WAVE);
            int oi = __shfl_xor(bi, o, WAVE);
            if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
        }
        if (lane == 0) {
            int act = bi;
            if (eps) {
                Philox ph(seed);
                uint4 r = ph(ctr_ptr ? *ctr_ptr : 0ull, (unsigned long long)row);
                float e = eps[eps_mod > 0 ? row % eps_mod : row];
                if (!(u01(r.x) > e)) { int ra = (int)(u01(r.y) * (float)A); act = ra < A ? ra : A - 1; }   // utils.py:37
            }
            action_out[row] = act;
        }
    }
}

// grid = ceil(M/4), block = 256 (4 rows per block)
__global__ __launch_bounds__(256) void k_head_fwd(const bf16_t* __restrict__ h, int M, int A, const bf16_t* __restrict__ w_adv,
                                                 const float* __restrict__ b_adv /* may be null */, const bf16_t* __restrict__ w_val,
                                                 const float* __restrict__ b_val, float* __restrict__ q_out,
                                                 int* __restrict__ action_out, const float* __restrict__ eps, int eps_mod,
                                                 unsigned long long seed, const unsigned long long* __restrict__ ctr_ptr) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    head_fwd_row(h, row, A, w_adv, b_adv, w_val, b_val, q_out, action_out, eps, eps_mod, seed, ctr_ptr);
}

// Two networks in one launch (rows [0, M1) = network 1, [M1, M1 + M2) = network 2; Q values only).
struct HeadNet { const bf16_t* h; const bf16_t* w_adv; const float* b_adv; const bf16_t* w_val; const float* b_val; float* q_out; int M; };
__global__ __launch_bounds__(256) void k_head_fwd_dual(HeadNet n1, HeadNet n2, int A) {
    int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n1.M + n2.M) return;
    const bool second = row >= n1.M;                       // wave-uniform
    const HeadNet n = second ? n2 : n1;
    if (second) row -= n1.M;
    head_fwd_row(n.h, row, A, n.w_adv, n.b_adv, n.w_val, n.b_val, n.q_out, nullptr, nullptr, 0, 0ull, nullptr);
}

// dh[row][k] = relu'(h) * (k<512 ? sum_j dadv_j W_adv[j][k] : dval W_val[k-512]);  one wave per row.
__global__ __launch_bounds__(256) void k_head_bwd_dh(const float* __restrict__ dq, const bf16_t* __restrict__ h, int M, int A,
                                                    const bf16_t* __restrict__ w_adv, const bf16_t* __restrict__ w_val,
                                                    bf16_t* __restrict__ dh) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    float g = lane < A ? dq[(long long)row * A + lane] : 0.f;
    const float dval = wave_sum(g);
    const float dadv_mine = g - dval / (float)A;     // lane j holds dadv_j
    float hv[16], wv[16], acc[16];
    load16_bf16(h + (long long)row * (2 * HID) + lane * 16, hv);
    const bool adv_half = lane < 32;
    const int k0 = (lane & 31) * 16;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    for (int j = 0; j < A; ++j) {
        const float dj = __shfl(dadv_mine, j, WAVE);
        if (adv_half) {
            load16_bf16(w_adv + (long long)j * HID + k0, wv);
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] += dj * wv[i];
        }
    }
    if (!adv_half) {
        load16_bf16(w_val + k0, wv);
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = dval * wv[i];
    }
    uint32_t o[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = pack_bf2(hv[2 * i] > 0.f ? acc[2 * i] : 0.f, hv[2 * i + 1] > 0.f ? acc[2 * i + 1] : 0.f);
    uint4* d = reinterpret_cast<uint4*>(dh + (long long)row * (2 * HID) + lane * 16);
    d[0] = make_uint4(o[0], o[1], o[2], o[3]);
    d[1] = make_uint4(o[4], o[5], o[6], o[7]);
}

// Ape-X TD (td.hip k_apex_td semantics) FUSED with the head's input gradient: one wave per sampled transition computes
// the double-DQN target, Huber loss, IS weight, dL/dQ row and new priority, then immediately its dh row. The batch-wide
// max IS weight was accumulated by the sampling kernel (StepState::wmax_bits), so no block-wide reduction and no
// single-block kernel sits between the forward and the backward any more. grid = ceil(B/16), block = 1024.
// q_on rows [0,B) = Q_online(s), [B,2B) = Q_online(s'); q_tg rows [0,B) = Q_target(s'); h = online hidden rows of s.
__global__ __launch_bounds__(1024) void k_apex_td_dh(const float* __restrict__ q_on, const float* __restrict__ q_tg, int B, int A,
                                                    const int* __restrict__ idx, const int* __restrict__ act,
                                                    const float* __restrict__ rew, const float* __restrict__ done,
                                                    const float* __restrict__ p_sampled, StepState* st, float gamma_n, float alpha,
                                                    float prio_eps, float n_scale, float* __restrict__ dq,
                                                    float* __restrict__ new_prio, const bf16_t* __restrict__ h,
                                                    const bf16_t* __restrict__ w_adv, const bf16_t* __restrict__ w_val,
                                                    bf16_t* __restrict__ dh) {
    __shared__ float s_loss[16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = blockIdx.x * 16 + wave;
    float lw = 0.f;
    if (row < B) {
        const int slot = idx ? idx[row] : row;
        const int a = act[slot];
        const float r = rew[slot], d = done[slot], p = p_sampled[row];
        const float beta = st->beta, total = st->total_prio, N = (float)st->n_items * n_scale;
        const float wmax = __uint_as_float(st->wmax_bits[st->step & 1]);
        const float qs = lane < A ? q_on[(long long)row * A + lane] : 0.f;
        const float qn = lane < A ? q_on[(long long)(B + row) * A + lane] : -3.0e38f;
        const float qt = lane < A ? q_tg[(long long)row * A + lane] : 0.f;
        const float mx = wave_max(qn);
        const int best = __ffsll((long long)__ballot(lane < A && qn == mx)) - 1;          // first max, like torch.argmax
        const float y = __shfl(qt, best, WAVE) * gamma_n * (1.0f - d) + r;
        const float diff = __shfl(qs, a, WAVE) - y;
        const float ad = fabsf(diff);
        const float hub = ad < 1.0f ? 0.5f * diff * diff : ad - 0.5f;
        const float w = (p > 0.f && total > 0.f && wmax > 0.f) ? __powf(N * p / total, -beta) / wmax : 0.f;
        lw = hub * w;
        const float gsel = fminf(fmaxf(diff, -1.0f), 1.0f) * w / (float)B;
        const float g = (lane == a) ? gsel : 0.f;
        if (lane < A) dq[(long long)row * A + lane] = g;
        if (lane == 0) new_prio[row] = __powf(hub + prio_eps, alpha);
        // dh row: dval = sum_j dq_j = gsel, dadv_j = dq_j - dval / A   (k_head_bwd_dh)
        const float dval = gsel;
        const float dadv_mine = g - dval / (float)A;
        float hv[16], wv[16], acc[16];
        load16_bf16(h + (long long)row * (2 * HID) + lane * 16, hv);
        const bool adv_half = lane < 32;
        const int k0 = (lane & 31) * 16;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        for (int j = 0; j < A; ++j) {
            const float dj = __shfl(dadv_mine, j, WAVE);
            if (adv_half) {
                load16_bf16(w_adv + (long long)j * HID + k0, wv);
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[i] += dj * wv[i];
            }
        }
        if (!adv_half) {
            load16_bf16(w_val + k0, wv);
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = dval * wv[i];
        }
        uint32_t o[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = pack_bf2(hv[2 * i] > 0.f ? acc[2 * i] : 0.f, hv[2 * i + 1] > 0.f ? acc[2 * i + 1] : 0.f);
        uint4* dst = reinterpret_cast<uint4*>(dh + (long long)row * (2 * HID) + lane * 16);
        dst[0] = make_uint4(o[0], o[1], o[2], o[3]);
        dst[1] = make_uint4(o[4], o[5], o[6], o[7]);
    }
    if (lane == 0) s_loss[wave] = lw;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) t += s_loss[i];
        atomicAdd(loss_slot(st, st->step), t / (float)B);       // slot zeroed by the sampling kernel of this step
        if (blockIdx.x == 0) st->max_w = __uint_as_float(st->wmax_bits[st->step & 1]);
    }
}

// The same with the three dueling heads in front (Q_online(s), Q_online(s'), Q_target(s') of the wave's transition): the
// Ape-X step needs no head launch of its own (k_head_fwd_dual: 5.8 us of a 205 us step for 1536 x 5 dot products).
// grid = ceil(B/4), block = 256. h_on rows [0,B) = s, [B,2B) = s'; h_tg rows [0,B) = s'. Q values are still written
// (statistics, tests).
struct HeadW { const bf16_t* w_adv; const float* b_adv; const bf16_t* w_val; const float* b_val; };
// block = 768 threads = 4 transitions x 3 waves: wave 3 t + k computes head k of transition t (0: Q_online(s), 1:
// Q_online(s'), 2: Q_target(s')) - the three chains of dot products + wave reductions run side by side, not one after the
// other (one wave per transition for all three: 14.6 us; this form: see profiles/apex_step_timeline_r3.txt) - the Q rows
// meet in LDS, then wave 3 t does the TD and the dh row of its transition.
__global__ __launch_bounds__(768) void k_apex_head_td_dh(const bf16_t* __restrict__ h_on, const bf16_t* __restrict__ h_tg, HeadW on, HeadW tg,
                                                        float* __restrict__ q_on, float* __restrict__ q_tg, int B, int A,
                                                        const int* __restrict__ idx, const int* __restrict__ act,
                                                        const float* __restrict__ rew, const float* __restrict__ done,
                                                        const float* __restrict__ p_sampled, StepState* st, float gamma_n, float alpha,
                                                        float prio_eps, float n_scale, float* __restrict__ dq,
                                                        float* __restrict__ new_prio, bf16_t* __restrict__ dh) {
    __shared__ float s_q[4][3][MAX_A];
    __shared__ float s_loss[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int t = wave / 3, k = wave - 3 * t;
    const int row = blockIdx.x * 4 + t;
    float lw = 0.f;
    float hv[16];
    if (row < B) {
        const bool tgt = k == 2;                                        // wave-uniform
        const HeadW& hw = tgt ? tg : on;
        const float q = head_q_row(tgt ? h_tg : h_on, k == 1 ? B + row : row, A, hw.w_adv, hw.b_adv, hw.w_val, hw.b_val, hv);
        if (lane < A) {
            s_q[t][k][lane] = q;
            (tgt ? q_tg : q_on)[(long long)(k == 1 ? B + row : row) * A + lane] = q;
        }
    }
    __syncthreads();
    if (row < B && k == 0) {
        const int slot = idx ? idx[row] : row;
        const int a = act[slot];
        const float r = rew[slot], d = done[slot], p = p_sampled[row];
        const float beta = st->beta, total = st->total_prio, N = (float)st->n_items * n_scale;
        const float wmax = __uint_as_float(st->wmax_bits[st->step & 1]);
        const float qs = lane < A ? s_q[t][0][lane] : 0.f, qn = lane < A ? s_q[t][1][lane] : -3.0e38f, qt = lane < A ? s_q[t][2][lane] : 0.f;
        const float mx = wave_max(qn);
        const int best = __ffsll((long long)__ballot(lane < A && qn == mx)) - 1;          // first max, like torch.argmax
        const float y = __shfl(qt, best, WAVE) * gamma_n * (1.0f - d) + r;
        const float diff = __shfl(qs, a, WAVE) - y;
        const float ad = fabsf(diff);
        const float hub = ad < 1.0f ? 0.5f * diff * diff : ad - 0.5f;
        const float w = (p > 0.f && total > 0.f && wmax > 0.f) ? __powf(N * p / total, -beta) / wmax : 0.f;
        lw = hub * w;
        const float gsel = fminf(fmaxf(diff, -1.0f), 1.0f) * w / (float)B;
        const float g = (lane == a) ? gsel : 0.f;
        if (lane < A) dq[(long long)row * A + lane] = g;
        if (lane == 0) new_prio[row] = __powf(hub + prio_eps, alpha);
        const float dval = gsel;
        const float dadv_mine = g - dval / (float)A;
        float wv[16], acc[16];
        const bool adv_half = lane < 32;
        const int k0 = (lane & 31) * 16;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        for (int j = 0; j < A; ++j) {
            const float dj = __shfl(dadv_mine, j, WAVE);
            if (adv_half) {
                load16_bf16(on.w_adv + (long long)j * HID + k0, wv);
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[i] += dj * wv[i];
            }
        }
        if (!adv_half) {
            load16_bf16(on.w_val + k0, wv);
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = dval * wv[i];
        }
        uint32_t o[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = pack_bf2(hv[2 * i] > 0.f ? acc[2 * i] : 0.f, hv[2 * i + 1] > 0.f ? acc[2 * i + 1] : 0.f);
        uint4* dst = reinterpret_cast<uint4*>(dh + (long long)row * (2 * HID) + lane * 16);
        dst[0] = make_uint4(o[0], o[1], o[2], o[3]);
        dst[1] = make_uint4(o[4], o[5], o[6], o[7]);
    }
    if (lane == 0 && k == 0) s_loss[t] = lw;
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(loss_slot(st, st->step), (s_loss[0] + s_loss[1] + s_loss[2] + s_loss[3]) / (float)B);
        if (blockIdx.x == 0) st->max_w = __uint_as_float(st->wmax_bits[st->step & 1]);
    }
}

DRL_API int drl_apex_head_td_dh(const void* h_on, const void* h_tg, const void* w_adv1, const float* b_adv1, const void* w_val1,
                                const float* b_val1, const void* w_adv2, const float* b_adv2, const void* w_val2, const float* b_val2,
                                float* q_on, float* q_tg, int B, int A, const int* idx, const int* act, const float* rew,
                                const float* done, const float* p_sampled, void* st, float gamma_n, float alpha, float prio_eps,
                                float n_scale, float* dq, float* new_prio, void* dh, hipStream_t s) {
    if (B <= 0 || A <= 0 || A > MAX_A) return -1;
    const HeadW on{(const bf16_t*)w_adv1, b_adv1, (const bf16_t*)w_val1, b_val1}, tg{(const bf16_t*)w_adv2, b_adv2, (const bf16_t*)w_val2, b_val2};
    k_apex_head_td_dh<<<(B + 3) / 4, 768, 0, s>>>((const bf16_t*)h_on, (const bf16_t*)h_tg, on, tg, q_on, q_tg, B, A, idx, act, rew, done,
                                                  p_sampled, (StepState*)st, gamma_n, alpha, prio_eps, n_scale, dq, new_prio, (bf16_t*)dh);
    return (int)hipGetLastError();
}

// Weight grads: see head_bwd_body.h. grid = (4, ceil(M/DW_ROWS)); block = 256.
__global__ __launch_bounds__(256) void k_head_bwd_dw(const float* __restrict__ dq, const bf16_t* __restrict__ h, int M, int A,
                                                    float* __restrict__ gw_adv, float* __restrict__ gb_adv /* may be null */,
                                                    float* __restrict__ gw_val, float* __restrict__ gb_val) {
    __shared__ float s_d[DW_ROWS * (HEAD_MAX_A + 1)];
    head_dw_block(dq, h, M, A, gw_adv, gb_adv, gw_val, gb_val, blockIdx.x, blockIdx.y, s_d);
}

DRL_API int drl_head_fwd(const void* h, int M, int A, const void* w_adv, const float* b_adv, const void* w_val, const float* b_val,
                         float* q_out, int* action_out, const float* eps, int eps_mod, unsigned long long seed,
                         const unsigned long long* ctr_ptr, hipStream_t s) {
    if (M <= 0 || A <= 0 || A > MAX_A) return -1;
    k_head_fwd<<<(M + 3) / 4, 256, 0, s>>>((const bf16_t*)h, M, A, (const bf16_t*)w_adv, b_adv, (const bf16_t*)w_val, b_val, q_out,
                                           action_out, eps, eps_mod, seed, ctr_ptr);
    return (int)hipGetLastError();
}

DRL_API int drl_head_bwd(const float* dq, const void* h, int M, int A, const void* w_adv, const void* w_val, void* dh,
                         float* gw_adv, float* gb_adv, float* gw_val, float* gb_val, int parts, hipStream_t s) {
    // parts: bit 0 = input gradient dh (on the critical path), bit 1 = weight / bias gradients (independent of it:
    // the engines issue it on a side stream)
    if (M <= 0 || A <= 0 || A > MAX_A) return -1;
    if (parts & 1)
        k_head_bwd_dh<<<(M + 3) / 4, 256, 0, s>>>(dq, (const bf16_t*)h, M, A, (const bf16_t*)w_adv, (const bf16_t*)w_val, (bf16_t*)dh);
    if (parts & 2)
        k_head_bwd_dw<<<dim3(4, (M + DW_ROWS - 1) / DW_ROWS), 256, 0, s>>>(dq, (const bf16_t*)h, M, A, gw_adv, gb_adv, gw_val, gb_val);
    return (int)hipGetLastError();
}

DRL_API int drl_apex_td_dh(const float* q_on, const float* q_tg, int B, int A, const int* idx, const int* act, const float* rew,
                           const float* done, const float* p_sampled, void* st, float gamma_n, float alpha, float prio_eps,
                           float n_scale, float* dq, float* new_prio, const void* h, const void* w_adv, const void* w_val, void* dh,
                           hipStream_t s) {
    if (B <= 0 || A <= 0 || A > MAX_A) return -1;
    k_apex_td_dh<<<(B + 15) / 16, 1024, 0, s>>>(q_on, q_tg, B, A, idx, act, rew, done, p_sampled, (StepState*)st, gamma_n, alpha,
                                               prio_eps, n_scale, dq, new_prio, (const bf16_t*)h, (const bf16_t*)w_adv,
                                               (const bf16_t*)w_val, (bf16_t*)dh);
    return (int)hipGetLastError();
}

DRL_API int drl_head_fwd_dual(const void* h1, int M1, const void* w_adv1, const float* b_adv1, const void* w_val1, const float* b_val1,
                              float* q1, const void* h2, int M2, const void* w_adv2, const float* b_adv2, const void* w_val2,
                              const float* b_val2, float* q2, int A, hipStream_t s) {
    if (M1 <= 0 || M2 <= 0 || A <= 0 || A > MAX_A) return -1;
    const HeadNet n1{(const bf16_t*)h1, (const bf16_t*)w_adv1, b_adv1, (const bf16_t*)w_val1, b_val1, q1, M1};
    const HeadNet n2{(const bf16_t*)h2, (const bf16_t*)w_adv2, b_adv2, (const bf16_t*)w_val2, b_val2, q2, M2};
    k_head_fwd_dual<<<(M1 + M2 + 3) / 4, 256, 0, s>>>(n1, n2, A);
    return (int)hipGetLastError();
}

// Fused TD-error / loss / priority kernels (reference semantics: libs/models.py:41-48,
// 142-162 and ape_x/learner.py:65-72). One launch turns Q matrices + (a, r, d, sampled priorities) into
//   * the scalar loss,
//   * dL/dQ for the online net, so backward starts without an autograd graph,
//   * the new priorities, written where the sum-tree update kernel reads them,
// with the importance-sampling weights (N·p/Σp)^-β / max computed in the same kernel from the device-resident
// StepState (no host round trip between sampling and priority write-back).
#include "common.h"

// ---------------------------------------------------------------------------------------------------
// Ape-X: n-step double-DQN, Huber, priority = (|huber| + eps)^alpha.   grid = 1 block, any B.
// q_on: rows [0,B) = Q_online(s), rows [B,2B) = Q_online(s');  q_tg: rows [0,B) = Q_target(s').
// a/r/d are read through idx (replay slot) when idx != nullptr, else directly.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_apex_td(const float* __restrict__ q_on, const float* __restrict__ q_tg, int B, int A,
                                                 const int* __restrict__ idx, const int* __restrict__ act,
                                                 const float* __restrict__ rew, const float* __restrict__ done,
                                                 const float* __restrict__ p_sampled, StepState* st, float gamma_n,
                                                 float alpha, float prio_eps, float n_scale, int use_is,
                                                 float* __restrict__ dq, float* __restrict__ new_prio,
                                                 float* __restrict__ delta_out, float* __restrict__ w_out) {
    __shared__ float red[32];
    const float beta = st->beta, total = st->total_prio, N = (float)st->n_items * n_scale;
    // pass 1: max IS weight
    float wmax = 0.f;
    if (use_is) {
        for (int b = threadIdx.x; b < B; b += blockDim.x) {
            float p = p_sampled[b];
            float w = (p > 0.f && total > 0.f) ? __powf(N * p / total, -beta) : 0.f;
            wmax = fmaxf(wmax, w);
        }
        wmax = block_max(wmax, red);
    }
    const float inv_wmax = (use_is && wmax > 0.f) ? 1.0f / wmax : 1.0f;
    float loss = 0.f;
    for (int b = threadIdx.x; b < B; b += blockDim.x) {
        const int slot = idx ? idx[b] : b;
        const int a = act[slot];
        const float r = rew[slot], d = done[slot];
        const float* qs = q_on + (long long)b * A;
        const float* qn = q_on + (long long)(B + b) * A;
        const float* qt = q_tg + (long long)b * A;
        int best = 0;
        float bv = qn[0];
        for (int j = 1; j < A; ++j) { float v = qn[j]; if (v > bv) { bv = v; best = j; } }   // first max, like torch.argmax
        const float y = qt[best] * gamma_n * (1.0f - d) + r;
        const float diff = qs[a] - y;
        const float ad = fabsf(diff);
        const float hub = ad < 1.0f ? 0.5f * diff * diff : ad - 0.5f;
        float w = 1.0f;
        if (use_is) {
            float p = p_sampled[b];
            w = (p > 0.f && total > 0.f) ? __powf(N * p / total, -beta) * inv_wmax : 0.f;
        }
        loss += hub * w;
        const float g = fminf(fmaxf(diff, -1.0f), 1.0f) * w / (float)B;
        for (int j = 0; j < A; ++j) dq[(long long)b * A + j] = (j == a) ? g : 0.f;
        new_prio[b] = __powf(hub + prio_eps, alpha);
        if (delta_out) delta_out[b] = hub;
        if (w_out) w_out[b] = w;
    }
    loss = block_sum(loss, red);
    if (threadIdx.x == 0) { *loss_slot(st, st->step) = loss / (float)B; st->max_w = wmax; }
}

// ---------------------------------------------------------------------------------------------------
// IS weights alone (R2D2 path / tests): w = (N p / total)^-beta / max.
// ---------------------------------------------------------------------------------------------------
// ``mailbox`` (data-parallel R2D2, engine/r2d2.py): the ranks' batch maxima, gathered by a small SUM all-reduce that ran
// under the forward pass (k_is_wmax_put wrote this rank's slot, zeros elsewhere) - the weights are then normalised by the
// maximum over ALL ranks' batches, as one big batch would be (reference: weights / weights.max(), r2d2/learner.py).
__global__ __launch_bounds__(1024) void k_is_weights(const float* __restrict__ p_sampled, int B, StepState* st, float n_scale,
                                                    float* __restrict__ w_out, const float* __restrict__ mailbox, int world) {
    __shared__ float red[32];
    const float beta = st->beta, total = st->total_prio, N = (float)st->n_items * n_scale;
    float wmax = 0.f;
    if (mailbox) {
        for (int r = 0; r < world; ++r) wmax = fmaxf(wmax, mailbox[r]);
    } else {
        for (int b = threadIdx.x; b < B; b += blockDim.x) {
            float p = p_sampled[b];
            wmax = fmaxf(wmax, (p > 0.f && total > 0.f) ? __powf(N * p / total, -beta) : 0.f);
        }
        wmax = block_max(wmax, red);
    }
    const float inv = wmax > 0.f ? 1.0f / wmax : 1.0f;
    for (int b = threadIdx.x; b < B; b += blockDim.x) {
        float p = p_sampled[b];
        w_out[b] = (p > 0.f && total > 0.f) ? __powf(N * p / total, -beta) * inv : 0.f;
    }
    if (threadIdx.x == 0) { st->max_w = wmax; *loss_slot(st, st->step) = 0.f; }
}

// This rank's batch maximum -> mailbox[rank], zeros in the other n - 1 slots (the SUM all-reduce then gathers the ranks' values).
__global__ __launch_bounds__(1024) void k_is_wmax_put(const float* __restrict__ p_sampled, int B, const StepState* st, float n_scale,
                                                     float* __restrict__ mailbox, int rank, int n) {
    __shared__ float red[32];
    const float beta = st->beta, total = st->total_prio, N = (float)st->n_items * n_scale;
    float wmax = 0.f;
    for (int b = threadIdx.x; b < B; b += blockDim.x) {
        float p = p_sampled[b];
        wmax = fmaxf(wmax, (p > 0.f && total > 0.f) ? __powf(N * p / total, -beta) : 0.f);
    }
    wmax = block_max(wmax, red);
    if ((int)threadIdx.x < n) mailbox[threadIdx.x] = (int)threadIdx.x == rank ? wmax : 0.f;
}

// ---------------------------------------------------------------------------------------------------
// R2D2: value-rescaled n-step double-Q over a sequence; loss_b = sum_t delta^2, prio = eta*max + (1-eta)*mean.
// grid = B blocks (one sequence each), blockDim = 128 (T <= 1024 handled by stride).
// Q tensors are time-major [T][B][A]; a/r/d are [B_store][T] rows addressed through idx (or b).
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ float h_rescale(float x, float eps) {
    return copysignf(sqrtf(fabsf(x) + 1.0f) - 1.0f, x) + eps * x;
}
__device__ __forceinline__ float h_inv(float x, float eps) {
    float ax = fabsf(x);
    float s = (sqrtf(1.0f + 4.0f * eps * (ax + 1.0f + eps)) - 1.0f) / (2.0f * eps);
    return copysignf(s * s - 1.0f, x);
}

__global__ __launch_bounds__(128) void k_r2d2_td(const float* __restrict__ q_on, const float* __restrict__ q_tg, int T, int B, int A,
                                                const int* __restrict__ idx, const int* __restrict__ act,
                                                const float* __restrict__ rew, const float* __restrict__ done,
                                                const int* __restrict__ lengths, const float* __restrict__ w_in, StepState* st,
                                                int burn_in, int nstep, int done_offset, float gamma_n, float eta, float rs_eps,
                                                float* __restrict__ dq, float* __restrict__ new_prio, float* __restrict__ seq_loss_out) {
    __shared__ float red[32];
    const int b = blockIdx.x;
    const int slot = idx ? idx[b] : b;
    const int len = lengths[slot];
    const float w = w_in ? w_in[b] : 1.0f;
    float s2 = 0.f, smax = 0.f, ssum = 0.f, cnt = 0.f;
    for (int t = threadIdx.x; t < T; t += blockDim.x) {
        const long long row = (long long)t * B + b;
        const int a = act[(long long)slot * T + t];
        const float dt = done[(long long)slot * T + t];
        bool learn = (t >= burn_in) && (t < len) && ((t + nstep < len) || (done_offset == 0 && dt != 0.f));
        float g = 0.f;
        if (learn) {
            const int tn = min(t + nstep, T - 1);
            const float* qn = q_on + ((long long)tn * B + b) * A;
            int best = 0;
            float bv = qn[0];
            for (int j = 1; j < A; ++j) { float v = qn[j]; if (v > bv) { bv = v; best = j; } }
            const float qnext = q_tg[((long long)tn * B + b) * A + best];
            const float d = done_offset == 0 ? dt : done[(long long)slot * T + min(t + done_offset, T - 1)];
            const float y = h_rescale(h_inv(qnext, rs_eps) * gamma_n * (1.0f - d) + rew[(long long)slot * T + t], rs_eps);
            const float diff = q_on[row * A + a] - y;
            const float ad = fabsf(diff);
            s2 += diff * diff; ssum += ad; smax = fmaxf(smax, ad); cnt += 1.0f;
            g = 2.0f * diff * w / (float)B;
        }
        for (int j = 0; j < A; ++j) dq[row * A + j] = (j == a) ? g : 0.f;
    }
    s2 = block_sum(s2, red);
    ssum = block_sum(ssum, red);
    cnt = block_sum(cnt, red);
    smax = block_max(smax, red);
    if (threadIdx.x == 0) {
        new_prio[b] = eta * smax + (1.0f - eta) * ssum / fmaxf(cnt, 1.0f);
        if (seq_loss_out) seq_loss_out[b] = s2;
        atomicAdd(loss_slot(st, st->step), s2 * w / (float)B);
    }
}

DRL_API int drl_apex_td(const float* q_on, const float* q_tg, int B, int A, const int* idx, const int* act, const float* rew,
                        const float* done, const float* p_sampled, void* st, float gamma_n, float alpha, float prio_eps,
                        float n_scale, int use_is, float* dq, float* new_prio, float* delta_out, float* w_out, hipStream_t s) {
    if (B <= 0 || A <= 0) return -1;
    k_apex_td<<<1, 1024, 0, s>>>(q_on, q_tg, B, A, idx, act, rew, done, p_sampled, (StepState*)st, gamma_n, alpha, prio_eps,
                                 n_scale, use_is, dq, new_prio, delta_out, w_out);
    return (int)hipGetLastError();
}

DRL_API int drl_is_weights(const float* p_sampled, int B, void* st, float n_scale, float* w_out, const float* mailbox, int world,
                           hipStream_t s) {
    if (B <= 0 || (mailbox && (world < 1 || world > 32))) return -1;
    k_is_weights<<<1, 1024, 0, s>>>(p_sampled, B, (StepState*)st, n_scale, w_out, mailbox, world);
    return (int)hipGetLastError();
}
DRL_API int drl_is_wmax_put(const float* p_sampled, int B, const void* st, float n_scale, float* mailbox, int rank, int world,
                            int mailbox_floats, hipStream_t s) {
    if (B <= 0 || !mailbox || world < 1 || rank < 0 || rank >= world || world > mailbox_floats || mailbox_floats > 1024) return -1;
    k_is_wmax_put<<<1, 1024, 0, s>>>(p_sampled, B, (const StepState*)st, n_scale, mailbox, rank, mailbox_floats);
    return (int)hipGetLastError();
}

DRL_API int drl_r2d2_td(const float* q_on, const float* q_tg, int T, int B, int A, const int* idx, const int* act,
                        const float* rew, const float* done, const int* lengths, const float* w_in, void* st, int burn_in,
                        int nstep, int done_offset, float gamma_n, float eta, float rs_eps, float* dq, float* new_prio,
                        float* seq_loss_out, hipStream_t s) {
    if (B <= 0 || A <= 0 || T <= 0) return -1;
    k_r2d2_td<<<B, 128, 0, s>>>(q_on, q_tg, T, B, A, idx, act, rew, done, lengths, w_in, (StepState*)st, burn_in, nstep,
                                done_offset, gamma_n, eta, rs_eps, dq, new_prio, seq_loss_out);
    return (int)hipGetLastError();
}

// ---- global max IS weight of a data-parallel batch (reference: weights / weights.max(), ape_x/learner.py:66-68) ----
// Every rank's sampler accumulates the maximum weight of ITS share of the batch in StepState::wmax_bits. The maxima travel
// in a small mailbox in front of the flat gradient inside the communication buffer: rank r writes its two slots at [2 r,
// 2 r + 1] and zeros elsewhere, the gradient all-reduce (a SUM) leaves every rank's values in place, and the maximum over the
// ranks goes back into wmax_bits - one step before the TD kernel of the batch it belongs to reads it (the batch of step
// t + 1 is drawn inside step t).
__global__ void k_wmax_put(const StepState* st, float* __restrict__ mailbox, int rank, int n) {
    const int t = threadIdx.x;
    if (t < n) mailbox[t] = (t >> 1) == rank ? __uint_as_float(st->wmax_bits[t & 1]) : 0.f;
}
__global__ void k_wmax_take(StepState* st, const float* __restrict__ mailbox, int world) {
    const int t = threadIdx.x;
    if (t < 2) {
        float m = 0.f;
        for (int r = 0; r < world; ++r) m = fmaxf(m, mailbox[2 * r + t]);
        st->wmax_bits[t] = __float_as_uint(m);
    }
}
DRL_API int drl_wmax_put(const void* st, float* mailbox, int rank, int world, int mailbox_floats, hipStream_t s) {
    if (world < 1 || rank < 0 || rank >= world || 2 * world > mailbox_floats || mailbox_floats > 64) return -1;
    k_wmax_put<<<1, 64, 0, s>>>((const StepState*)st, mailbox, rank, mailbox_floats);
    return (int)hipGetLastError();
}
DRL_API int drl_wmax_take(void* st, const float* mailbox, int world, hipStream_t s) {
    if (world < 1 || world > 32) return -1;
    k_wmax_take<<<1, 64, 0, s>>>((StepState*)st, mailbox, world);
    return (int)hipGetLastError();
}

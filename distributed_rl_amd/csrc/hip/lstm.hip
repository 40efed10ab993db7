// LSTM backward-through-time pointwise kernel + small helpers (SURVEY G6 backward).
// Forward steps and all GEMMs of the recurrence run on the MFMA kernel of gemm.hip (EPI_LSTM epilogue / NN split-K);
// what is left per time step is elementwise over (batch, unit): combine the gradient arriving from the heads with
// the recurrent one, push it through the cell, and emit d(gate pre-activations) in the gate-interleaved layout
// the weight-gradient GEMMs consume.
#include "common.h"

// one thread per (b, unit).  gates = post-activation (i,f,g,o) saved by the forward epilogue.
// dh_rec (f32) is read AND zeroed: the next (earlier) time step's split-K GEMM accumulates into it with atomics.
__global__ __launch_bounds__(256) void k_lstm_bwd_point(const bf16_t* __restrict__ dh_head /* may be null */,
                                                       float* __restrict__ dh_rec, float* __restrict__ dc,
                                                       const bf16_t* __restrict__ gates, const float* __restrict__ c_t,
                                                       const float* __restrict__ c_prev, bf16_t* __restrict__ dgates,
                                                       int n /* B*H */, int H) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int b = i / H, u = i - b * H;
    float dh = dh_rec[i];
    dh_rec[i] = 0.f;
    if (dh_head) dh += bf2f(dh_head[i]);
    const uint2 gw = *reinterpret_cast<const uint2*>(gates + ((long long)b * H + u) * 4);
    float dcv = dc[i];
    const uint2 dg = lstm_bwd_unit(dh, dcv, gw.x, gw.y, c_t[i], c_prev[i]);
    dc[i] = dcv;
    *reinterpret_cast<uint2*>(dgates + ((long long)b * H + u) * 4) = dg;
}

// byte offsets of the plane windows of a sampled sequence batch, time-major: off[t*B + b] = idx[b] * seq_bytes + t *
// plane_bytes (a sequence slot holds T + 3 single planes: step t is the 4 planes from plane t, replay/device_seq.py)
__global__ void k_seq_offsets(const int* __restrict__ idx, int B, int T, long long seq_bytes, long long plane_bytes,
                              long long* __restrict__ off) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * T) return;
    const int t = i / B, b = i - t * B;
    off[i] = (long long)idx[b] * seq_bytes + (long long)t * plane_bytes;
}

// the same for a store that shares planes between sequences (replay/device_seq.py): every slot keeps a row of T window
// offsets, off[t*B + b] = win[idx[b]][t]
__global__ void k_seq_win_offsets(const int* __restrict__ idx, const long long* __restrict__ win, int B, int T,
                                  long long* __restrict__ off) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * T) return;
    const int b = i / T, t = i - b * T;                 // consecutive threads read consecutive words of a row
    off[t * B + b] = win[(long long)idx[b] * T + t];
}

// dst[b][:] = src[idx[b]][:]  (f32 rows -> f32 and bf16 copies): stored recurrent state of the sampled sequences
__global__ void k_gather_state(const float* __restrict__ src, const int* __restrict__ idx, int B, int H,
                               float* __restrict__ dst_f32, bf16_t* __restrict__ dst_bf16) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * H) return;
    int b = i / H, u = i - b * H;
    float v = src[(long long)idx[b] * H + u];
    if (dst_f32) dst_f32[i] = v;
    if (dst_bf16) dst_bf16[i] = f2bf(v);
}

// the stored (h0, c0) of the sampled sequences into the initial-state rows of TWO networks, one launch; the same launch
// adds up to two pairs of bias vectors (b_ih + b_hh of the online and the target LSTM: they were two launches of their own
// on the critical path of the R2D2 step, 11 us)
struct Add2 { const float* a; const float* b; float* out; };
__global__ void k_gather_state2(const float* __restrict__ h0, const float* __restrict__ c0, const int* __restrict__ idx, int B, int H,
                                bf16_t* __restrict__ hs_a, float* __restrict__ cs_a, bf16_t* __restrict__ hs_b,
                                float* __restrict__ cs_b, Add2 s0, Add2 s1, int n_add) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_add) {
        if (s0.out) s0.out[i] = s0.a[i] + s0.b[i];
        if (s1.out) s1.out[i] = s1.a[i] + s1.b[i];
    }
    if (i >= B * H) return;
    int b = i / H, u = i - b * H;
    const long long o = (long long)idx[b] * H + u;
    const float h = h0[o], c = c0[o];
    const bf16_t hb = f2bf(h);
    hs_a[i] = hb; cs_a[i] = c;
    if (hs_b) { hs_b[i] = hb; cs_b[i] = c; }
}

__global__ void k_add2(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = a[i] + b[i];
}

DRL_API int drl_lstm_bwd_point(const void* dh_head, float* dh_rec, float* dc, const void* gates, const float* c_t,
                               const float* c_prev, void* dgates, int B, int H, hipStream_t s) {
    if (B <= 0 || H <= 0) return -1;
    int n = B * H;
    k_lstm_bwd_point<<<(n + 255) / 256, 256, 0, s>>>((const bf16_t*)dh_head, dh_rec, dc, (const bf16_t*)gates, c_t, c_prev,
                                                     (bf16_t*)dgates, n, H);
    return (int)hipGetLastError();
}
DRL_API int drl_seq_offsets(const int* idx, int B, int T, long long seq_bytes, long long plane_bytes, long long* off, hipStream_t s) {
    if (B <= 0 || T <= 0) return -1;
    k_seq_offsets<<<(B * T + 255) / 256, 256, 0, s>>>(idx, B, T, seq_bytes, plane_bytes, off);
    return (int)hipGetLastError();
}
DRL_API int drl_seq_win_offsets(const int* idx, const long long* win, int B, int T, long long* off, hipStream_t s) {
    if (B <= 0 || T <= 0) return -1;
    k_seq_win_offsets<<<(B * T + 255) / 256, 256, 0, s>>>(idx, win, B, T, off);
    return (int)hipGetLastError();
}
DRL_API int drl_gather_state(const float* src, const int* idx, int B, int H, float* dst_f32, void* dst_bf16, hipStream_t s) {
    if (B <= 0 || H <= 0) return -1;
    k_gather_state<<<(B * H + 255) / 256, 256, 0, s>>>(src, idx, B, H, dst_f32, (bf16_t*)dst_bf16);
    return (int)hipGetLastError();
}
DRL_API int drl_gather_state2(const float* h0, const float* c0, const int* idx, int B, int H, void* hs_a, float* cs_a, void* hs_b,
                              float* cs_b, const float* a0, const float* b0, float* o0, const float* a1, const float* b1, float* o1,
                              int n_add, hipStream_t s) {
    if (B <= 0 || H <= 0 || !hs_a || !cs_a || (!hs_b) != (!cs_b) || n_add < 0) return -1;
    if ((o0 && (!a0 || !b0)) || (o1 && (!a1 || !b1))) return -1;
    if (!o0 && !o1) n_add = 0;
    const int n = B * H > n_add ? B * H : n_add;
    k_gather_state2<<<(n + 255) / 256, 256, 0, s>>>(h0, c0, idx, B, H, (bf16_t*)hs_a, cs_a, (bf16_t*)hs_b, cs_b, Add2{a0, b0, o0},
                                                    Add2{a1, b1, o1}, n_add);
    return (int)hipGetLastError();
}
DRL_API int drl_add2(const float* a, const float* b, float* out, int n, hipStream_t s) {
    if (n <= 0) return -1;
    k_add2<<<(n + 255) / 256, 256, 0, s>>>(a, b, out, n);
    return (int)hipGetLastError();
}

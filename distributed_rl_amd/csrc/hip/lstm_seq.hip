// Persistent LSTM recurrence for gfx950 (SURVEY G6 "(b) persistent recurrence kernel").
//
// One launch runs ALL T time steps of one or two networks (online + target of the R2D2 learner) instead of T small
// GEMM launches whose 32 workgroups cannot hide their own latency (9.4 us per step measured with the tiled kernel).
//   * 4H/64 workgroups per network; workgroup j owns hidden units [16j, 16j+16) = 64 gate columns (the gate-interleaved
//     layout n = 4*unit + gate keeps the 4 gates of a unit together). Its 64 x H slice of W_hh is copied ONCE into LDS in
//     MFMA-fragment order (conflict-free ds_read_b128) and stays there for all T steps.
//   * h_{t-1} (M x H bf16) goes global -> VGPR directly in fragment layout (16-byte loads, 8 consecutive k per lane);
//     wave w owns batch rows [16w, 16w+16). acc = mfma(W, h): a lane ends with the 4 gates of one unit per MFMA tile, and
//     the weight rows are permuted so that its 4 tiles are 4 CONSECUTIVE units: cell update in registers (c never leaves
//     them), 8-byte h stores, 16-byte c stores, 32-byte gate stores, 64-byte xg loads.
//   * Between steps the workgroups of one network meet at arrival flags in global memory (write-through hand-off, see
//     grid_wait_fl / grid_arrive_fl below). Every spin is BOUNDED: if it expires an error flag is raised and the kernel stops
//     waiting (wrong numbers, reported by the host wrapper, but never a hung GPU). 32-64 workgroups of 256 threads /
//     64 KB LDS are always co-resident on 256 CUs.
#include "common.h"

typedef __attribute__((ext_vector_type(8))) __bf16 mfma_bf16x8_t;

__device__ __forceinline__ f32x4 mfma16_(bf16x8 a, bf16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(mfma_bf16x8_t, a), __builtin_bit_cast(mfma_bf16x8_t, b), c, 0, 0, 0);
}
__device__ __forceinline__ float sigm_(float x) { return 1.0f / (1.0f + __expf(-x)); }
__device__ __forceinline__ float tanh_(float x) { return 2.0f / (1.0f + __expf(-2.0f * x)) - 1.0f; }

// bound of every inter-workgroup poll (iterations of load + s_sleep); a device global so that a test can shrink it and
// drive the give-up path for real (drl_lstm_set_spin_limit)
__device__ int g_spin_limit = 1 << 17;
#define SPIN_LIMIT g_spin_limit
#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT
typedef unsigned long long u64_t;

// Hand-off of h_t between the workgroups of one network, write-through form (MI355X: per-CU L1 and per-XCD L2 are not
// coherent inside a launch): every h store is an agent-scope 8-byte atomic store (global_store_dwordx2 sc1), every
// storing wave drains (s_waitcnt vmcnt(0)), workgroup barrier, one lane signals; the consumer polls with sc1 loads,
// workgroup barrier, and EVERY load of h is an agent-scope load (sc1, bypasses L1) - so neither an L2 write-back nor an
// L1 invalidate fence is paid per step (the fence form measured 9.4 us per step, no better than one launch per step).
// A first form handed h over in its plain [M][H] layout with one arrival counter per network: there a wave's load
// instruction touches 16 rows x 64 bytes, half of 16 different 128-byte lines, and the other halves are fetched again by
// the next instruction (sc1 loads are kept by no cache on the way): 5.5 us per step. What runs is the coalesced exchange
// form (4.6 us per step): h_t travels through an exchange buffer laid out in CONSUMER order, [parity][wave][ks][lane][8 x bf16]: one load instruction
// of a wave is one contiguous kilobyte (8 whole lines, each fetched once). Producers scatter their 8-byte pieces into it
// (write-through stores as before); hs keeps its plain layout for the kernels outside (ordinary stores, nobody inside the
// launch reads them). Same counter protocol. Two parity slots: slot (t+1)&1 last held h_{t-1}, which every workgroup had
// finished reading before it arrived at the barrier this workgroup passed to start step t.
struct LstmXNet {
    const float* xg; const bf16_t* w_hh; bf16_t* hs; float* cs; bf16_t* gates;
    const float* bias;    // [4H] added to xg on the fly (nullptr: xg already holds it) - saves a pass over xg after a library GEMM
    bf16_t* hx;           // [2][4][H/32][64][8] bf16 = 2 x 64 KB, then the arrival flags at byte 192 K
};
#define HX_SYNC_BYTE_OFFSET (192 * 1024)

// Arrival flags instead of a counter: atomics on one address serialise (~35 ns each, so 32 simultaneous arrivals take
// > 1 us to land). Every workgroup owns one flag word (FLAG_STRIDE ints apart) and STORES the number of the step it has
// finished (a write-through store, no read-modify-write); lanes 0..NWG-1 of the polling wave watch one flag each.
constexpr int LSTM_FLAG_STRIDE = 32;
__device__ __forceinline__ void grid_wait_fl(int* flags, int nwg, int target, int* err, int* s_dead) {
    if (threadIdx.x < 64 && !*s_dead) {
        const bool mine = (int)threadIdx.x < nwg;
        int it = 0;
        while (true) {
            const int v = mine ? __hip_atomic_load(flags + threadIdx.x * LSTM_FLAG_STRIDE, RLX_AGENT) : target;
            if (__all((int)((unsigned)v - (unsigned)target) >= 0)) break;          // wrap-safe: flags carry a launch epoch
            if (++it > SPIN_LIMIT) { if (threadIdx.x == 0) { atomicExch(err, 1); *s_dead = 1; } break; }
            __builtin_amdgcn_s_sleep(1);
        }
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ void grid_arrive_fl(int* flags, int wg, int step_done) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // every storing wave's h stores are acknowledged ...
    __syncthreads();                                          // ... before the one lane that raises the flag
    if (threadIdx.x == 0) __hip_atomic_store(flags + wg * LSTM_FLAG_STRIDE, step_done, RLX_AGENT);
}

template <int HD>
__global__ __launch_bounds__(256) void k_lstm_seq_fwd_x(LstmXNet n0, LstmXNet n1, int T, int M, int groups, int* err) {
    constexpr int NWG = 4 * HD / 64, NKS = HD / 32;
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    // row groups as in k_lstm_bptt below: the M/16 row blocks are dealt to `groups` independent sets of NWG workgroups
    // (blockDim.x / 64 row blocks each, own flags) - for launches that have the chip to themselves
    const int net = blockIdx.x / (NWG * groups), grp = (blockIdx.x / NWG) % groups, wg = blockIdx.x % NWG;
    const LstmXNet a = net ? n1 : n0;
    int* sync = reinterpret_cast<int*>(reinterpret_cast<uint8_t*>(a.hx) + HX_SYNC_BYTE_OFFSET) + grp * (2 * NWG * LSTM_FLAG_STRIDE);
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 15, g4 = lane >> 4;
    const int wave = grp * (int)(blockDim.x >> 6) + (tid >> 6);      // row block [16 wave, 16 wave + 16) of the batch
    for (int i = tid; i < 4 * NKS * 64; i += (int)blockDim.x) {
        const int l = i & 63, f = i >> 6, ks = f % NKS, nt = f / NKS, ii = l & 15;
        const int n = 64 * wg + 4 * (4 * (ii >> 2) + nt) + (ii & 3);
        *reinterpret_cast<uint4*>(smem + (size_t)i * 16) = *reinterpret_cast<const uint4*>(a.w_hh + (size_t)n * HD + ks * 32 + (l >> 4) * 8);
    }
    __syncthreads();
    const bool active = wave * 16 < M;
    const int m = min(wave * 16 + r, M - 1);
    const int u0 = 16 * wg + 4 * g4;                 // this lane owns units u0 .. u0+3 of row m
    float c[4];
    {
        const float4 c0 = *reinterpret_cast<const float4*>(a.cs + (size_t)m * HD + u0);
        c[0] = c0.x; c[1] = c0.y; c[2] = c0.z; c[3] = c0.w;
    }
    float4 bq[4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
        bq[q] = a.bias ? *reinterpret_cast<const float4*>(a.bias + 4 * u0 + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
    const auto hx_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.hx, 0, 2 * 4 * NKS * 64 * 16, 0x00020000);
    // where this lane's 4 units live in the consumer-ordered buffer: fragment u0/32, piece (u0%32)/8 of row r, half (u0%8)/4
    const unsigned st_off = (unsigned)((((wave * NKS + u0 / 32) * 64 + ((u0 & 31) >> 3) * 16 + r) * 8 + (u0 & 4)) * sizeof(bf16_t));
    const unsigned ld_off = (unsigned)(((wave * NKS) * 64 + lane) * 16);
    constexpr unsigned SLOT = 4 * NKS * 64 * 16;     // bytes per parity slot
    __shared__ int s_dead;
    if (tid == 0) s_dead = 0;
    __syncthreads();
    // No memset between launches: a flag holds (launch epoch << 12) + steps finished. Every workgroup counts the launches
    // of this buffer in a word of its own (all of them run in every launch, so the counts agree without communication).
    int* my_epoch = sync + (NWG + wg) * LSTM_FLAG_STRIDE;
    const int base = (int)((unsigned)*my_epoch << 12);
    for (int t = 0; t < T; ++t) {
        float4 xq[4];
        const float* xrow = a.xg + ((size_t)t * M + m) * (4 * HD) + 4 * u0;
#pragma unroll
        for (int q = 0; q < 4; ++q) xq[q] = *reinterpret_cast<const float4*>(xrow + 4 * q);
        bf16x8 hf[NKS];
        if (t == 0) {                                          // initial state: written before the launch, plain loads
            const uint4* hrow = reinterpret_cast<const uint4*>(a.hs + (size_t)m * HD + g4 * 8);
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) hf[ks] = __builtin_bit_cast(bf16x8, hrow[ks * 4]);
        } else {
            grid_wait_fl(sync, NWG, base + t, err, &s_dead);
            const unsigned base = (t & 1) * SLOT + ld_off;
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {                 // one contiguous KB per instruction, sc1 (never an L1 hit)
                const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(hx_rsrc, base + ks * 1024, 0, 16);
                hf[ks] = __builtin_bit_cast(bf16x8, v);
            }
        }
        f32x4 acc[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const bf16x8 wf = *reinterpret_cast<const bf16x8*>(smem + ((size_t)(nt * NKS + ks) * 64 + lane) * 16);
                acc[nt] = mfma16_(wf, hf[ks], acc[nt]);
            }
        float h[4];
        uint32_t gp[8];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const float ig = sigm_(acc[nt][0] + xq[nt].x + bq[nt].x), fg = sigm_(acc[nt][1] + xq[nt].y + bq[nt].y);
            const float gg = tanh_(acc[nt][2] + xq[nt].z + bq[nt].z), og = sigm_(acc[nt][3] + xq[nt].w + bq[nt].w);
            c[nt] = fg * c[nt] + ig * gg;
            h[nt] = og * tanh_(c[nt]);
            gp[2 * nt] = pack_bf2(ig, fg);
            gp[2 * nt + 1] = pack_bf2(gg, og);
        }
        if (active) {
            const uint32_t h01 = pack_bf2(h[0], h[1]), h23 = pack_bf2(h[2], h[3]);
            if (t + 1 < T)                                     // the hand-off first: it is what the other workgroups wait for
                __hip_atomic_store(reinterpret_cast<u64_t*>(reinterpret_cast<uint8_t*>(a.hx) + ((t + 1) & 1) * SLOT + st_off),
                                   (u64_t)h01 | ((u64_t)h23 << 32), RLX_AGENT);
            const size_t o = ((size_t)(t + 1) * M + m) * HD + u0;
            *reinterpret_cast<uint2*>(a.hs + o) = make_uint2(h01, h23);
            *reinterpret_cast<float4*>(a.cs + o) = make_float4(c[0], c[1], c[2], c[3]);
            if (a.gates) {
                uint4* gout = reinterpret_cast<uint4*>(a.gates + ((size_t)t * M + m) * (4 * HD) + 4 * u0);
                gout[0] = make_uint4(gp[0], gp[1], gp[2], gp[3]);
                gout[1] = make_uint4(gp[4], gp[5], gp[6], gp[7]);
            }
        }
        if (t + 1 < T) grid_arrive_fl(sync, wg, base + t + 1);
    }
    if (tid == 0) *my_epoch = (int)(((unsigned)base >> 12) + 1u);
}

// hx0 / hx1: exchange buffers of 256 KB each (lstm_hx), zero when first used; nothing is reset between launches.
// groups: 1, 2 or 4 independent row groups (must divide M / 16; 0 = 1).
DRL_API int drl_lstm_seq_fwd_x(const float* xg0, const void* w0, void* hs0, float* cs0, void* gates0, const float* bias0, void* hx0,
                                const float* xg1, const void* w1, void* hs1, float* cs1, void* gates1, const float* bias1, void* hx1, int n_nets,
                                int T, int M, int H, int groups, int* err, hipStream_t s) {
    if (H != 512 || M <= 0 || M > 64 || (M & 15) || T <= 0 || n_nets < 1 || n_nets > 2 || !hx0 || (n_nets == 2 && !hx1)) return -1;
    if (T >= 4096) return -1;                                  // flags carry the step in 12 bits
    const int rbs = M / 16;
    if (groups <= 0) groups = 1;
    if (groups > 4 || rbs % groups) return -1;
    LstmXNet a{xg0, (const bf16_t*)w0, (bf16_t*)hs0, cs0, (bf16_t*)gates0, bias0, (bf16_t*)hx0};
    LstmXNet b{xg1, (const bf16_t*)w1, (bf16_t*)hs1, cs1, (bf16_t*)gates1, bias1, (bf16_t*)hx1};
    constexpr int LDS = 4 * (512 / 32) * 1024;
    auto k0 = k_lstm_seq_fwd_x<512>;
    static bool once = (hipFuncSetAttribute((const void*)k0, hipFuncAttributeMaxDynamicSharedMemorySize, LDS), true);
    (void)once;
    k0<<<n_nets * groups * (4 * 512 / 64), 64 * (rbs / groups), LDS, s>>>(a, b, T, M, groups, err);
    return (int)hipGetLastError();
}


// ---- persistent BPTT ------------------------------------------------------------------------------------------------
// The backward recurrence as ONE launch per time chunk (was: one launch per step, 9.8 us each, a third of the R2D2 step).
// Same decomposition as the forward: workgroup j owns hidden units [16j, 16j+16), wave w batch rows [16w, 16w+16), a lane
// 4 consecutive units of one row - for those it keeps dc in registers, reads gates / c / dh_head of the step (plain loads,
// issued before the wait: they do not depend on the recurrence), and produces the 16 gate gradients dG (bf16, as the
// weight-gradient GEMMs read them). What it needs from the others is dh_rec[m][own units] = sum_n dG_{t+1}[m][n] W_hh[n][u]:
// the whole dG of the previous step (64 x 2048 bf16), handed over through an exchange buffer in consumer order
// ([parity][wave][ks = n/32][lane][8 bf16]: one load instruction = one contiguous KB) exactly like h in the forward, and
// multiplied with the workgroup's 2048 x 16 column slice of W_hh, which sits in LDS in MFMA-fragment order for the whole
// launch. K = 2048 is accumulated in one fp32 MFMA accumulator in a fixed order: no atomics, bit-reproducible.
struct BpttArgs {
    const bf16_t* dhs;    // [L][M][H]    dL/dh_t from the heads (window index k = t - burn)
    const bf16_t* gates;  // [L][M][4H]   post-activation gates of the window
    const float* cs;      // [L+1][M][H]  cs[k] = c_{t-1}, cs[k+1] = c_t
    bf16_t* dG;           // [L][M][4H]   out: gate gradients
    const bf16_t* w_hh;   // [4H][H]
    float* dh_rec;        // [M][H] carry between launches (in unless `first`, out if `carry`)
    float* dc;            // [M][H] carry
    uint8_t* bx;          // 2 x 256 KB exchange slots, arrival flags + launch counters at byte 512 K
};
#define BX_SYNC_BYTE_OFFSET (512 * 1024)

template <int HD>
__global__ __launch_bounds__(256) void k_lstm_bptt(BpttArgs a, int k_hi, int steps, int M, int first, int carry, int* err) {
    constexpr int NWG = HD / 16, NKS = 4 * HD / 32;              // 32 workgroups, 64 k-steps of 32 gate columns
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];   // NKS KB: W_hh[:, 16 wg .. +15] in fragment order
    // Row groups: batch rows never mix in the recurrence, so the M/16 row blocks are dealt to gridDim.x / NWG independent
    // groups of NWG workgroups (blockDim.x / 64 row blocks each, own flags). What bounds a step is the 4 KB per lane of dG
    // a wave has to pull in after the wait; with one or two waves per workgroup that load is spread over 2-4x as many CUs.
    const int grp = blockIdx.x / NWG, wg = blockIdx.x % NWG, tid = threadIdx.x, lane = tid & 63, r = lane & 15, g4 = lane >> 4;
    const int wave = grp * (int)(blockDim.x >> 6) + (tid >> 6);      // row block [16 wave, 16 wave + 16) of the batch
    int* sync = reinterpret_cast<int*>(a.bx + BX_SYNC_BYTE_OFFSET) + grp * (2 * NWG * LSTM_FLAG_STRIDE);
    // fragment (ks, lane (ii, g)) element e = W_hh[ks*32 + g*8 + e][16 wg + ii]: row n of the slice is 32 contiguous bytes
    for (int n = tid; n < 4 * HD; n += (int)blockDim.x) {
        const uint4* src = reinterpret_cast<const uint4*>(a.w_hh + (size_t)n * HD + 16 * wg);
        const uint4 lo = src[0], hi = src[1];
        const uint32_t v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
        bf16_t* dst = reinterpret_cast<bf16_t*>(smem) + ((size_t)(n >> 5) * 64 + ((n & 31) >> 3) * 16) * 8 + (n & 7);
#pragma unroll
        for (int ii = 0; ii < 16; ++ii) dst[ii * 8] = (bf16_t)((ii & 1) ? (v[ii >> 1] >> 16) : (v[ii >> 1] & 0xffffu));
    }
    __shared__ int s_dead;
    if (tid == 0) s_dead = 0;
    __syncthreads();
    const bool active = wave * 16 < M;
    const int m = min(wave * 16 + r, M - 1);
    const int u0 = 16 * wg + 4 * g4;                             // this lane owns units u0 .. u0+3 of row m
    const size_t e0 = (size_t)m * HD + u0;
    float dc[4] = {0.f, 0.f, 0.f, 0.f}, rec[4] = {0.f, 0.f, 0.f, 0.f};
    if (!first) {
        const float4 c4 = *reinterpret_cast<const float4*>(a.dc + e0), r4 = *reinterpret_cast<const float4*>(a.dh_rec + e0);
        dc[0] = c4.x; dc[1] = c4.y; dc[2] = c4.z; dc[3] = c4.w;
        rec[0] = r4.x; rec[1] = r4.y; rec[2] = r4.z; rec[3] = r4.w;
    }
    const auto bx_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.bx, 0, 2 * 4 * NKS * 64 * 16, 0x00020000);
    constexpr unsigned SLOT = 4 * NKS * 64 * 16;                 // 256 KB per parity slot
    const unsigned ld_off = (unsigned)(((wave * NKS) * 64 + lane) * 16);
    // this lane's 16 gate columns n = 4 u0 .. 4 u0 + 15 in consumer order: k-step n/32, pieces 2 (g4 & 1) and 2 (g4 & 1) + 1
    const unsigned st_off = (unsigned)((((wave * NKS + (u0 >> 3)) * 64 + (2 * (g4 & 1)) * 16 + r) * 8) * sizeof(bf16_t));
    int* my_epoch = sync + (NWG + wg) * LSTM_FLAG_STRIDE;
    const int base = (int)((unsigned)*my_epoch << 12);
    // dh_rec of this lane's units from the dG all workgroups published for `stamp` steps
    auto recurrent = [&](int stamp) {
        grid_wait_fl(sync, NWG, base + stamp, err, &s_dead);
        const unsigned b0 = (stamp & 1) * SLOT + ld_off;
        bf16x8 gf[NKS];
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(bx_rsrc, b0 + ks * 1024, 0, 16);
            gf[ks] = __builtin_bit_cast(bf16x8, v);
        }
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            const bf16x8 wf = *reinterpret_cast<const bf16x8*>(smem + ((size_t)ks * 64 + lane) * 16);
            acc = mfma16_(wf, gf[ks], acc);
        }
        rec[0] = acc[0]; rec[1] = acc[1]; rec[2] = acc[2]; rec[3] = acc[3];
    };
    for (int s = 0; s < steps; ++s) {
        const int k = k_hi - s;
        const size_t e = (size_t)k * M * HD + e0;
        // inputs of the pointwise part: independent of the recurrence, in flight while this workgroup waits
        const uint2 v_dh = *reinterpret_cast<const uint2*>(a.dhs + e);
        const uint4 v_g0 = *reinterpret_cast<const uint4*>(a.gates + e * 4), v_g1 = *reinterpret_cast<const uint4*>(a.gates + e * 4 + 8);
        const float4 v_ct = *reinterpret_cast<const float4*>(a.cs + e + (size_t)M * HD), v_cp = *reinterpret_cast<const float4*>(a.cs + e);
        if (s > 0) recurrent(s);
        const float ct[4] = {v_ct.x, v_ct.y, v_ct.z, v_ct.w}, cp[4] = {v_cp.x, v_cp.y, v_cp.z, v_cp.w};
        const uint32_t gw[8] = {v_g0.x, v_g0.y, v_g0.z, v_g0.w, v_g1.x, v_g1.y, v_g1.z, v_g1.w};
        const float dhh[4] = {__uint_as_float(v_dh.x << 16), __uint_as_float(v_dh.x & 0xffff0000u),
                              __uint_as_float(v_dh.y << 16), __uint_as_float(v_dh.y & 0xffff0000u)};
        uint32_t o[8];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint2 dg = lstm_bwd_unit(rec[q] + dhh[q], dc[q], gw[2 * q], gw[2 * q + 1], ct[q], cp[q]);
            o[2 * q] = dg.x;
            o[2 * q + 1] = dg.y;
        }
        const bool more = s + 1 < steps || carry;
        if (active) {
            if (more) {                                            // the hand-off first: it is what the others wait for
                u64_t* x = reinterpret_cast<u64_t*>(a.bx + ((s + 1) & 1) * SLOT + st_off);
                __hip_atomic_store(x, (u64_t)o[0] | ((u64_t)o[1] << 32), RLX_AGENT);
                __hip_atomic_store(x + 1, (u64_t)o[2] | ((u64_t)o[3] << 32), RLX_AGENT);
                __hip_atomic_store(x + 32, (u64_t)o[4] | ((u64_t)o[5] << 32), RLX_AGENT);      // next piece: 16 lanes x 16 B on
                __hip_atomic_store(x + 33, (u64_t)o[6] | ((u64_t)o[7] << 32), RLX_AGENT);
            }
            uint4* gout = reinterpret_cast<uint4*>(a.dG + e * 4);
            gout[0] = make_uint4(o[0], o[1], o[2], o[3]);
            gout[1] = make_uint4(o[4], o[5], o[6], o[7]);
        }
        if (more) grid_arrive_fl(sync, wg, base + s + 1);
    }
    if (carry) {                                                   // state for the launch that continues the chain
        recurrent(steps);
        if (active) {
            *reinterpret_cast<float4*>(a.dh_rec + e0) = make_float4(rec[0], rec[1], rec[2], rec[3]);
            *reinterpret_cast<float4*>(a.dc + e0) = make_float4(dc[0], dc[1], dc[2], dc[3]);
        }
    }
    if (tid == 0) *my_epoch = (int)(((unsigned)base >> 12) + 1u);
}

// Window steps k_hi, k_hi - 1, ..., k_hi - steps + 1 (k = t - burn). first: the chain starts here (dh_rec = dc = 0);
// carry: another launch continues below k_hi - steps + 1 (dh_rec / dc are written for it). bx: 768 KB scratch, zero when
// first used. gates / cs are the window parts of the forward buffers (cs has one more row than gates).
// groups: 0 = the default (g_bptt_groups, capped to the row blocks there are); else it must divide M / 16.
static int g_bptt_groups = 4;
DRL_API int drl_lstm_bptt(const void* dhs, const void* gates, const float* cs, void* dG, const void* w_hh, float* dh_rec, float* dc,
                          void* bx, int k_hi, int steps, int M, int H, int first, int carry, int groups, int* err, hipStream_t s) {
    if (H != 512 || M <= 0 || M > 64 || (M & 15) || steps <= 0 || steps >= 4095 || k_hi - steps + 1 < 0) return -1;
    const int rbs = M / 16;
    if (groups <= 0) { groups = g_bptt_groups; while (groups > 1 && (groups > rbs || rbs % groups)) groups >>= 1; }
    if (groups < 1 || groups > 4 || rbs % groups) return -1;
    BpttArgs a{(const bf16_t*)dhs, (const bf16_t*)gates, cs, (bf16_t*)dG, (const bf16_t*)w_hh, dh_rec, dc, (uint8_t*)bx};
    constexpr int LDS = (4 * 512 / 32) * 1024;
    auto kern = k_lstm_bptt<512>;
    static bool once = (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS), true);
    (void)once;
    kern<<<groups * (512 / 16), 64 * (rbs / groups), LDS, s>>>(a, k_hi, steps, M, first, carry, err);
    return (int)hipGetLastError();
}
DRL_API int drl_lstm_set_bptt_groups(int g) { g_bptt_groups = g < 1 ? 1 : g > 4 ? 4 : g; return 0; }

DRL_API int drl_lstm_set_spin_limit(int n) {
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_spin_limit), &n, sizeof(int));
}

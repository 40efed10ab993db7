// Inter-workgroup hand-off latency floor (VERDICT r2 "do this" #5): two workgroups bounce a flag + payload back and forth
// through device memory, the way the persistent LSTM recurrences hand h over between workgroups (lstm_seq.hip), for the
// memory kinds a kernel can choose from: coarse-grained hipMalloc (sc1 = agent-scope stores / loads), fine-grained
// (hipDeviceMallocFinegrained) and uncached (hipDeviceMallocUncached) allocations. Diagnostic entry point: scripts/
// pingpong_bench.py prints the table under profiles/.
//
// One hop: the sender's 256 lanes store `payload_bytes` (sc1), every wave waits for its stores (vmcnt(0)), workgroup
// barrier, lane 0 stores the flag (sc1). The receiver's lane 0 polls the flag (sc1 loads), barrier, then all lanes load the
// payload with sc1 loads (which bypass the CU's L1, MI355X_MICROARCH "valid forms") and check it.
#include "common.h"

__device__ __forceinline__ void pp_store(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ uint32_t pp_load(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// buf: [2 directions][payload words]; flags: [2] (one 128-byte line each). out[0] = cycles (s_memtime) of the whole exchange as
// seen by block a, out[1] = payload mismatches, out[2] = XCC id of block a, out[3] = XCC id of block b.
__global__ __launch_bounds__(256) void k_pingpong(uint32_t* buf, uint32_t* flags, int payload_words, int rounds, int blk_a, int blk_b,
                                                  unsigned long long* out, long long spin_limit) {
    const int me = (int)blockIdx.x == blk_a ? 0 : ((int)blockIdx.x == blk_b ? 1 : -1);
    if (me < 0) return;
    const int tid = threadIdx.x;
    uint32_t* tx = buf + me * payload_words;            // what I write
    const uint32_t* rx = buf + (1 - me) * payload_words; // what the peer writes
    uint32_t* my_flag = flags + me * 32;
    const uint32_t* peer_flag = flags + (1 - me) * 32;
    unsigned long long bad = 0;
    __shared__ int give_up;
    if (tid == 0) give_up = 0;
    __syncthreads();
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int r = 1; r <= rounds; ++r) {
        if (me == 0) {              // send r, then wait for the echo
            for (int i = tid; i < payload_words; i += 256) pp_store(tx + i, (uint32_t)(r * 131 + i));
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) pp_store(my_flag, (uint32_t)r);
        }
        if (tid == 0) {             // wait for the peer's round r
            long long spins = 0;
            while (pp_load(peer_flag) < (uint32_t)r) {
                if (++spins > spin_limit) { give_up = 1; break; }
            }
        }
        __syncthreads();
        if (give_up) break;         // every wave reaches this: the grid drains
        for (int i = tid; i < payload_words; i += 256) bad += pp_load(rx + i) != (uint32_t)(r * 131 + i + (me == 0 ? 7 : 0));
        if (me == 1) {              // echo: payload + 7
            for (int i = tid; i < payload_words; i += 256) pp_store(tx + i, (uint32_t)(r * 131 + i + 7));
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) pp_store(my_flag, (uint32_t)r);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    bad = (unsigned long long)wave_sum((float)bad);
    if (tid == 0) {
        if (me == 0) { out[0] = t1 - t0; out[2] = xcc & 0xf; out[4] = give_up; }
        else out[3] = xcc & 0xf;
    }
    if ((tid & 63) == 0 && bad) atomicAdd(out + 1, bad);
}

// The recurrence's real pattern: n_wg workgroups, every round each one stores its slice (slice_words), raises its flag, waits
// for ALL flags (lane i polls flag i) and reads ALL slices. Two slice buffers by round parity (a workgroup can be at most
// one round ahead of the slowest). out[0] = cycles seen by block 0, out[1] = mismatches, out[4] = gave up.
__global__ __launch_bounds__(256) void k_allgather(uint32_t* buf, uint32_t* flags, int slice_words, int n_wg, int rounds, int stride,
                                                   unsigned long long* out, long long spin_limit) {
    if ((int)blockIdx.x % stride != 0 || (int)blockIdx.x / stride >= n_wg) return;     // stride 8: all on one XCD; 1: spread
    const int me = (int)blockIdx.x / stride, tid = threadIdx.x;
    __shared__ int give_up;
    if (tid == 0) give_up = 0;
    __syncthreads();
    unsigned long long bad = 0;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int r = 1; r <= rounds; ++r) {
        uint32_t* base = buf + (size_t)(r & 1) * n_wg * slice_words;
        for (int i = tid; i < slice_words; i += 256) pp_store(base + me * slice_words + i, (uint32_t)(r * 131 + me * 17 + i));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) pp_store(flags + me * 32, (uint32_t)r);
        if (tid < 64) {
            long long spins = 0;
            bool ok = false;
            while (!ok) {
                const bool mine = tid >= n_wg || pp_load(flags + tid * 32) >= (uint32_t)r;
                ok = __all(mine);
                if (!ok && ++spins > spin_limit) { if (tid == 0) give_up = 1; break; }
            }
        }
        __syncthreads();
        if (give_up) break;
        // 16-byte sc1 loads (L1 bypassed), all issued before the first is waited for - as lstm_seq.hip reads h
        const int total16 = n_wg * slice_words / 4;
        typedef uint32_t pp_u32x4 __attribute__((ext_vector_type(4)));
        for (int i0 = 0; i0 < total16; i0 += 256 * 8) {
            pp_u32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = min(i0 + u * 256 + tid, total16 - 1);
                asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v[u]) : "v"(base + (size_t)i * 4) : "memory");
            }
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]));
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = min(i0 + u * 256 + tid, total16 - 1) * 4, w = i / slice_words, j = i - w * slice_words;
                bad += v[u][0] != (uint32_t)(r * 131 + w * 17 + j) || v[u][3] != (uint32_t)(r * 131 + w * 17 + j + 3);
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    bad = (unsigned long long)wave_sum((float)bad);
    if (tid == 0 && me == 0) { out[0] = t1 - t0; out[4] = give_up; }
    if ((tid & 63) == 0 && bad) atomicAdd(out + 1, bad);
}

DRL_API int drl_allgather_bench(int kind, int slice_bytes, int n_wg, int rounds, int stride, unsigned long long* out5, float* us_out) {
    if (slice_bytes < 4 || slice_bytes % 4 || n_wg < 2 || n_wg > 64 || rounds < 1 || stride < 1 || n_wg * stride > 256) return -1;
    const int words = slice_bytes / 4;
    uint32_t *buf = nullptr, *flags = nullptr;
    unsigned long long* out = nullptr;
    const size_t nb = 2 * (size_t)n_wg * words * 4, nf = 64 * 128;
    hipError_t e;
    if (kind == 0) { e = hipMalloc(&buf, nb); if (e == hipSuccess) e = hipMalloc(&flags, nf); }
    else {
        const unsigned fl = kind == 1 ? hipDeviceMallocFinegrained : hipDeviceMallocUncached;
        e = hipExtMallocWithFlags((void**)&buf, nb, fl);
        if (e == hipSuccess) e = hipExtMallocWithFlags((void**)&flags, nf, fl);
    }
    if (e != hipSuccess) return (int)e;
    if ((e = hipMalloc(&out, 64)) != hipSuccess) return (int)e;
    (void)hipMemset(buf, 0, nb); (void)hipMemset(flags, 0, nf); (void)hipMemset(out, 0, 64);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0, 0);
    k_allgather<<<n_wg * stride, 256, 0, 0>>>(buf, flags, words, n_wg, rounds, stride, out, 1ll << 22);
    (void)hipEventRecord(e1, 0);
    e = hipDeviceSynchronize();
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    *us_out = ms * 1000.f;
    (void)hipMemcpy(out5, out, 40, hipMemcpyDeviceToHost);
    (void)hipFree(buf); (void)hipFree(flags); (void)hipFree(out);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return (int)e;
}

// kind: 0 = hipMalloc (coarse-grained), 1 = fine-grained, 2 = uncached. Returns 0 and fills out[5] =
// {cycles, mismatches, xcc of a, xcc of b, gave up}; us_out = wall time of the kernel (events).
DRL_API int drl_pingpong(int kind, int payload_bytes, int rounds, int blk_a, int blk_b, int grid, unsigned long long* out5, float* us_out) {
    if (payload_bytes < 4 || payload_bytes % 4 || rounds < 1 || blk_a == blk_b || blk_a >= grid || blk_b >= grid || grid > 1024) return -1;
    const int words = payload_bytes / 4;
    uint32_t *buf = nullptr, *flags = nullptr;
    unsigned long long* out = nullptr;
    const size_t nb = 2 * (size_t)words * 4, nf = 2 * 128;
    hipError_t e;
    if (kind == 0) { e = hipMalloc(&buf, nb); if (e == hipSuccess) e = hipMalloc(&flags, nf); }
    else {
        const unsigned fl = kind == 1 ? hipDeviceMallocFinegrained : hipDeviceMallocUncached;
        e = hipExtMallocWithFlags((void**)&buf, nb, fl);
        if (e == hipSuccess) e = hipExtMallocWithFlags((void**)&flags, nf, fl);
    }
    if (e != hipSuccess) return (int)e;
    if ((e = hipMalloc(&out, 64)) != hipSuccess) return (int)e;
    (void)hipMemset(buf, 0, nb); (void)hipMemset(flags, 0, nf); (void)hipMemset(out, 0, 64);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0, 0);
    k_pingpong<<<grid, 256, 0, 0>>>(buf, flags, words, rounds, blk_a, blk_b, out, 1ll << 22);
    (void)hipEventRecord(e1, 0);
    e = hipDeviceSynchronize();
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    *us_out = ms * 1000.f;
    (void)hipMemcpy(out5, out, 40, hipMemcpyDeviceToHost);
    (void)hipFree(buf); (void)hipFree(flags); (void)hipFree(out);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return (int)e;
}

// Direct all-reduce over xGMI for the flat gradient buffer.
//
// An MI355X node is a full mesh: every GPU has its own link to each of its 7 peers. A ring all-reduce drives one link in
// and one link out per GPU; the 13-32 MB gradients of this framework are then bound by a single link plus 2(W-1) hops
// of latency. This kernel uses the mesh: a two-shot (reduce-scatter + all-gather) all-reduce in which every GPU pulls
// 1/W of the buffer from ALL peers at once (phase 1, sums it, writes the result in place into its own slice) and then
// pulls the W-1 reduced slices from their owners (phase 2), again over all links at once: two hops of latency, all
// 7 links busy in both phases, no staging copy (the collective works in place on the engine's gradient buffer, which
// is allocated here and IPC-mapped by every peer), and it is ONE kernel launch per collective - it captures into the
// step's hipGraph like any other kernel (no host thread, no watchdog, no second stream owned by a library).
//
// Synchronisation: per (channel, phase, source rank, block) 32-bit epoch flags in uncached device memory, written with
// system-scope release stores into the PEER's flag block and polled locally with system-scope acquire loads. Block b of
// every rank owns the same chunk b of every slice, so blocks pair up across ranks and no grid-wide barrier is needed:
//   barrier 0  "my kernel has started" = everything this stream ran before (the backward pass) is complete and, being
//              past a kernel boundary, written back from the XCD L2s: peers may read my buffer;
//   barrier 1  "chunk b of my slice is reduced" (release: L2 write-back, then the flag);
//   barrier 2  "I have read your reduced chunks": the owner's next kernels may overwrite the buffer.
// Epochs are kept on the device (one counter per block and channel, advanced by the kernel itself), so a replayed graph
// needs no host arguments. Every poll loop is bounded by the constant 100 MHz clock: a peer that never arrives sets the
// error flag (drl_xar_error) and the kernel exits - it cannot hang the GPU.
#include "common.h"
#include <string.h>

#define XAR_MAX_W 8
#define XAR_MAX_B 64          // blocks per collective
#define XAR_CH 4              // independent channels (collectives that may be in flight at the same time)
#define XAR_THREADS 512
#define XAR_FLAG_WORDS (XAR_CH * 3 * XAR_MAX_W * XAR_MAX_B)

struct XarPeers {
    float* data[XAR_MAX_W];           // [rank] = local buffer, others IPC-mapped
    uint32_t* flags[XAR_MAX_W];
};

struct XarCtx {
    int rank, world, n_blocks, device;
    long long cap_elems;
    XarPeers peers;
    uint32_t* epochs;                 // [XAR_CH][XAR_MAX_B], local
    int* err;                         // local
    long long timeout_ticks;
    void* opened[XAR_MAX_W][2];
};

// One inter-GPU barrier of block b with block b of every peer. What makes it a hand-off of DATA (not only of control),
// in the order the instructions must appear in the ISA (checked in the .s; hipcc leaves out both waits on its own):
//   1. EVERY wave waits for its own stores (s_waitcnt vmcnt(0)): a workgroup-scope barrier does not wait for vector
//      memory on gfx950 (one CU, one L1), so without this the signalling wave could release before its siblings' stores
//      have reached the L2 it writes back;
//   2. block barrier, then ONE wave: system-scope release fence (buffer_wbl2 sc0 sc1: the XCD's L2 goes to memory) and
//      an explicit s_waitcnt vmcnt(0) - the compiler drops the fence's own wait when the counter is provably zero, and
//      the flag would overtake the write-back;
//   3. relaxed system-scope flag stores into the peers' flag blocks (uncached memory), relaxed polls of the local block,
//      then ONE system-scope acquire fence (buffer_inv sc0 sc1: L1 and the L2's lines of peer memory) + its wait, and the
//      block barrier that lets the other waves load.
__device__ __forceinline__ void xar_barrier(const XarPeers& P, int ch, int phase, int rank, int W, uint32_t epoch, int* err,
                                            long long timeout_ticks) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int t = threadIdx.x;
    if (t < WAVE) {                                    // wave 0 (wave-uniform branch): lanes 0..W-1 each serve one peer
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (t < W && t != rank) {
            const int b = blockIdx.x;
            uint32_t* remote = P.flags[t] + ((ch * 3 + phase) * XAR_MAX_W + rank) * XAR_MAX_B + b;
            __hip_atomic_store(remote, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            const uint32_t* local = P.flags[rank] + ((ch * 3 + phase) * XAR_MAX_W + t) * XAR_MAX_B + b;
            const long long t0 = wall_clock64();
            while ((int)(__hip_atomic_load(local, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - epoch) < 0) {
                if (wall_clock64() - t0 > timeout_ticks) { *err = 1 + phase; break; }
                __builtin_amdgcn_s_sleep(4);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
}

// grid = n_blocks (the same on every rank), block = XAR_THREADS. off4 / n4 in float4 units.
__global__ __launch_bounds__(XAR_THREADS) void k_xgmi_all_reduce(XarPeers P, int rank, int W, long long off4, long long n4, int ch,
                                                                 uint32_t* __restrict__ epochs, int* __restrict__ err,
                                                                 long long timeout_ticks) {
    const int b = blockIdx.x, nb = gridDim.x, tid = threadIdx.x;
    const uint32_t epoch = epochs[ch * XAR_MAX_B + b] + 1u;
    const long long s4 = (n4 + W - 1) / W;             // float4 per slice
    const long long c4 = (s4 + nb - 1) / nb;           // float4 per block and slice
    xar_barrier(P, ch, 0, rank, W, epoch, err, timeout_ticks);
    // ---- phase 1: reduce chunk b of slice `rank` over all ranks, in place ----
    {
        const long long base = off4 + (long long)rank * s4;
        const long long lim = min(s4, n4 - (long long)rank * s4);      // slice length (the last one may be short / empty)
        const long long lo = (long long)b * c4, hi = min(lo + c4, lim);
        for (long long i = lo + tid; i < hi; i += XAR_THREADS) {
            float4 v[XAR_MAX_W];
#pragma unroll
            for (int p = 0; p < XAR_MAX_W; ++p)
                if (p < W) v[p] = reinterpret_cast<const float4*>(P.data[p])[base + i];
            float4 a = v[0];
#pragma unroll
            for (int p = 1; p < XAR_MAX_W; ++p)
                if (p < W) { a.x += v[p].x; a.y += v[p].y; a.z += v[p].z; a.w += v[p].w; }
            reinterpret_cast<float4*>(P.data[rank])[base + i] = a;
        }
    }
    xar_barrier(P, ch, 1, rank, W, epoch, err, timeout_ticks);
    // ---- phase 2: fetch chunk b of every other slice from its owner ----
    for (int k = 1; k < W; ++k) {
        const int p = (rank + k) % W;                   // every rank starts at a different peer: all links busy
        const long long base = off4 + (long long)p * s4;
        const long long lim = min(s4, n4 - (long long)p * s4);
        const long long lo = (long long)b * c4, hi = min(lo + c4, lim);
        const float4* __restrict__ src = reinterpret_cast<const float4*>(P.data[p]) + base;
        float4* __restrict__ dst = reinterpret_cast<float4*>(P.data[rank]) + base;
        for (long long i = lo + tid; i < hi; i += XAR_THREADS) dst[i] = src[i];
    }
    xar_barrier(P, ch, 2, rank, W, epoch, err, timeout_ticks);
    if (tid == 0) epochs[ch * XAR_MAX_B + b] = epoch;
}

// ---- host side -------------------------------------------------------------------------------------------
#define XAR_CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return (int)e_; } while (0)

// Allocates the communication buffer (cap_elems floats, zeroed), the flag block and the epoch counters on the current device.
// mem_kind: where the DATA buffer lives. 0 = ordinary (coarse-grained) device memory: cached in the owner's L2, peers see
// it after the release / acquire pairs of xar_barrier; 1 = fine-grained, 2 = uncached device memory: coherent without
// any cache maintenance - the fallback parallel/dist.py switches to when the default fails its verification on a node
// (the engines' own kernels then write their gradients through the L2: slower steps, same numbers).
DRL_API int drl_xar_create2(int rank, int world, long long cap_elems, int n_blocks, double timeout_s, int mem_kind, void** out) {
    if (world < 1 || world > XAR_MAX_W || rank < 0 || rank >= world || n_blocks < 1 || n_blocks > XAR_MAX_B || cap_elems <= 0 ||
        (cap_elems & 3))
        return -1;
    XarCtx* c = new XarCtx();
    c->rank = rank; c->world = world; c->n_blocks = n_blocks; c->cap_elems = cap_elems;
    c->timeout_ticks = (long long)(timeout_s * 1.0e8);                 // wall_clock64: constant 100 MHz
    XAR_CK(hipGetDevice(&c->device));
    void* d = nullptr; void* f = nullptr;
    if (mem_kind == 1) { XAR_CK(hipExtMallocWithFlags(&d, (size_t)cap_elems * 4, hipDeviceMallocFinegrained)); }
    else if (mem_kind == 2) { XAR_CK(hipExtMallocWithFlags(&d, (size_t)cap_elems * 4, hipDeviceMallocUncached)); }
    else if (mem_kind == 0) { XAR_CK(hipMalloc(&d, (size_t)cap_elems * 4)); }
    else return -1;
    XAR_CK(hipMemset(d, 0, (size_t)cap_elems * 4));
    XAR_CK(hipExtMallocWithFlags(&f, XAR_FLAG_WORDS * 4, hipDeviceMallocUncached));
    XAR_CK(hipMemset(f, 0, XAR_FLAG_WORDS * 4));
    XAR_CK(hipMalloc((void**)&c->epochs, XAR_CH * XAR_MAX_B * 4));
    XAR_CK(hipMemset(c->epochs, 0, XAR_CH * XAR_MAX_B * 4));
    XAR_CK(hipMalloc((void**)&c->err, 4));
    XAR_CK(hipMemset(c->err, 0, 4));
    XAR_CK(hipDeviceSynchronize());
    c->peers.data[rank] = (float*)d;
    c->peers.flags[rank] = (uint32_t*)f;
    *out = c;
    return 0;
}

DRL_API int drl_xar_create(int rank, int world, long long cap_elems, int n_blocks, double timeout_s, void** out) {
    return drl_xar_create2(rank, world, cap_elems, n_blocks, timeout_s, 0, out);
}

// Workgroups per collective from the next launch on. Every rank must switch at the same point of its launch sequence
// (block b pairs with block b of the peers); the per-block epochs stay consistent because all ranks always launch the
// same grid for the same collective.
DRL_API int drl_xar_set_blocks(void* ctx, int n_blocks) {
    if (n_blocks < 1 || n_blocks > XAR_MAX_B) return -1;
    ((XarCtx*)ctx)->n_blocks = n_blocks;
    return 0;
}

// bound of every flag poll (seconds): short while the path is being verified, generous in production
DRL_API int drl_xar_set_timeout(void* ctx, double timeout_s) {
    ((XarCtx*)ctx)->timeout_ticks = (long long)(timeout_s * 1.0e8);
    return 0;
}

DRL_API void* drl_xar_data_ptr(void* ctx) { return ((XarCtx*)ctx)->peers.data[((XarCtx*)ctx)->rank]; }

// out: 2 x 64 bytes (data buffer, flag block) for the peers' drl_xar_open
DRL_API int drl_xar_handles(void* ctx, void* out128) {
    XarCtx* c = (XarCtx*)ctx;
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "handle size");
    hipIpcMemHandle_t h[2];
    XAR_CK(hipIpcGetMemHandle(&h[0], c->peers.data[c->rank]));
    XAR_CK(hipIpcGetMemHandle(&h[1], c->peers.flags[c->rank]));
    memcpy(out128, h, 128);
    return 0;
}

// peer_device: the peer's device ordinal as seen by THIS process (-1 = unknown, e.g. remapped by HIP_VISIBLE_DEVICES).
// Known and different from ours: peer access is checked and switched on before anything is mapped.
DRL_API int drl_xar_open(void* ctx, int peer, const void* handles128, int peer_device) {
    XarCtx* c = (XarCtx*)ctx;
    if (peer < 0 || peer >= c->world || peer == c->rank) return -1;
    if (peer_device >= 0 && peer_device != c->device) {
        int can = 0;
        XAR_CK(hipDeviceCanAccessPeer(&can, c->device, peer_device));
        if (!can) return -3;
        const hipError_t e = hipDeviceEnablePeerAccess(peer_device, 0);
        if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) return (int)e;
        (void)hipGetLastError();
    }
    hipIpcMemHandle_t h[2];
    memcpy(h, handles128, 128);
    void* d = nullptr; void* f = nullptr;
    XAR_CK(hipIpcOpenMemHandle(&d, h[0], hipIpcMemLazyEnablePeerAccess));
    XAR_CK(hipIpcOpenMemHandle(&f, h[1], hipIpcMemLazyEnablePeerAccess));
    c->opened[peer][0] = d; c->opened[peer][1] = f;
    // probe: a mapping this device cannot reach shows up here as an error code, not later as a fault inside the kernel
    uint32_t probe[2] = {1u, 1u};
    XAR_CK(hipMemcpy(&probe[0], d, 4, hipMemcpyDeviceToHost));
    XAR_CK(hipMemcpy(&probe[1], f, 4, hipMemcpyDeviceToHost));
    c->peers.data[peer] = (float*)d; c->peers.flags[peer] = (uint32_t*)f;
    return 0;
}

// In place SUM over ranks of data[off, off + n) (elements; both multiples of 4) - every rank calls it with the same
// arguments, in the same order per channel.
DRL_API int drl_xar_all_reduce(void* ctx, long long off, long long n, int channel, hipStream_t s) {
    XarCtx* c = (XarCtx*)ctx;
    if ((off & 3) || (n & 3) || off < 0 || n <= 0 || off + n > c->cap_elems || channel < 0 || channel >= XAR_CH) return -1;
    for (int p = 0; p < c->world; ++p)
        if (!c->peers.data[p] || !c->peers.flags[p]) return -2;
    k_xgmi_all_reduce<<<c->n_blocks, XAR_THREADS, 0, s>>>(c->peers, c->rank, c->world, off / 4, n / 4, channel, c->epochs, c->err,
                                                          c->timeout_ticks);
    return (int)hipGetLastError();
}

// 0 = fine; 1..3 = a poll of barrier 0..2 timed out at least once since the last call (synchronises the device)
// device address of the timeout word (the optimizer launch reads it as a guard: optim.hip)
DRL_API void* drl_xar_err_ptr(void* ctx) { return ctx ? (void*)((XarCtx*)ctx)->err : nullptr; }

DRL_API int drl_xar_error(void* ctx) {
    XarCtx* c = (XarCtx*)ctx;
    int e = 0;
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpy(&e, c->err, 4, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    if (e) hipMemset(c->err, 0, 4);
    return e;
}

DRL_API int drl_xar_destroy(void* ctx) {
    XarCtx* c = (XarCtx*)ctx;
    if (!c) return 0;
    hipDeviceSynchronize();
    for (int p = 0; p < c->world; ++p)
        for (int k = 0; k < 2; ++k)
            if (p != c->rank && c->opened[p][k]) hipIpcCloseMemHandle(c->opened[p][k]);
    hipFree(c->peers.data[c->rank]);
    hipFree(c->peers.flags[c->rank]);
    hipFree(c->epochs);
    hipFree(c->err);
    delete c;
    return 0;
}

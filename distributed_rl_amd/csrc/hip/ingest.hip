// Experience ingest: ONE launch scatters the small tables of a staged block (everything except the frame planes, which a
// DMA puts straight into the plane ring) from the block's device-side tail copy into the replay's arrays (VERDICT r2 weak
// #2: the learner thread issued ~12 small torch ops per block - copies, dtype casts, index arithmetic).
//   plain segments:  dst[i] = src[i]                      (32-bit words: actions, rewards, dones, lengths, h0, c0)
//   window segments: dst[i] = (src[i] + plane_base) * 7056  (int32 plane index inside the block -> int64 byte offset in the ring)
#include "common.h"

constexpr int ING_MAX_SEG = 8, ING_MAX_WIN = 2;
struct IngestArgs {
    const uint32_t* seg_src[ING_MAX_SEG];
    uint32_t* seg_dst[ING_MAX_SEG];
    int seg_words[ING_MAX_SEG];
    const int* win_src[ING_MAX_WIN];
    long long* win_dst[ING_MAX_WIN];
    int win_n[ING_MAX_WIN];
    int n_seg, n_win;
    long long plane_base;
};

__global__ __launch_bounds__(256) void k_ingest_scatter(IngestArgs a) {
    const int tid = blockIdx.x * 256 + threadIdx.x, nt = gridDim.x * 256;
    for (int s = 0; s < a.n_seg; ++s)
        for (int i = tid; i < a.seg_words[s]; i += nt) a.seg_dst[s][i] = a.seg_src[s][i];
    for (int w = 0; w < a.n_win; ++w)
        for (int i = tid; i < a.win_n[w]; i += nt) a.win_dst[w][i] = ((long long)a.win_src[w][i] + a.plane_base) * 7056ll;
}

DRL_API int drl_ingest_scatter(const void* tail, int n_seg, const long long* seg_src_off, void* const* seg_dst, const int* seg_words,
                               int n_win, const long long* win_src_off, void* const* win_dst, const int* win_n, long long plane_base,
                               hipStream_t s) {
    if (!tail || n_seg < 0 || n_seg > ING_MAX_SEG || n_win < 0 || n_win > ING_MAX_WIN) return -1;
    IngestArgs a{};
    int most = 1;
    for (int i = 0; i < n_seg; ++i) {
        if (!seg_dst[i] || seg_words[i] < 0 || (seg_src_off[i] & 3)) return -1;
        a.seg_src[i] = (const uint32_t*)((const uint8_t*)tail + seg_src_off[i]);
        a.seg_dst[i] = (uint32_t*)seg_dst[i];
        a.seg_words[i] = seg_words[i];
        most = seg_words[i] > most ? seg_words[i] : most;
    }
    for (int i = 0; i < n_win; ++i) {
        if (!win_dst[i] || win_n[i] < 0 || (win_src_off[i] & 3)) return -1;
        a.win_src[i] = (const int*)((const uint8_t*)tail + win_src_off[i]);
        a.win_dst[i] = (long long*)win_dst[i];
        a.win_n[i] = win_n[i];
        most = win_n[i] > most ? win_n[i] : most;
    }
    a.n_seg = n_seg; a.n_win = n_win; a.plane_base = plane_base;
    int grid = (most + 255) / 256;
    grid = grid > 64 ? 64 : grid;
    k_ingest_scatter<<<grid, 256, 0, s>>>(a);
    return (int)hipGetLastError();
}

// The device side of one staged block as ONE host call: planes pinned -> ring and tables pinned -> tail
// by DMA, then the scatter launch (priorities included: one more plain segment, into the replay's pending-priority
// array). A one-launch form that pulled planes and tables over PCIe from the GPU-addressable staging buffer was measured
// and rejected: same host time in the running system, and the learner next to it fell into a slow mode in 2 runs of 3
// (docs/DESIGN.md section 7).
DRL_API int drl_ingest_block(void* planes_dst, const void* planes_src, long long planes_bytes, void* tail, const void* tail_src,
                             long long tail_bytes, int n_seg, const long long* seg_src_off, void* const* seg_dst, const int* seg_words,
                             int n_win, const long long* win_src_off, void* const* win_dst, const int* win_n, long long plane_base,
                             hipStream_t s) {
    if (!planes_dst || !planes_src || !tail || !tail_src || planes_bytes < 0 || tail_bytes <= 0) return -1;
    for (int i = 0; i < n_seg && i < ING_MAX_SEG; ++i)
        if (seg_src_off[i] < 0 || seg_src_off[i] + 4ll * seg_words[i] > tail_bytes) return -1;
    for (int i = 0; i < n_win && i < ING_MAX_WIN; ++i)
        if (win_src_off[i] < 0 || win_src_off[i] + 4ll * win_n[i] > tail_bytes) return -1;
    hipError_t e = hipSuccess;
    if (planes_bytes) e = hipMemcpyAsync(planes_dst, planes_src, (size_t)planes_bytes, hipMemcpyHostToDevice, s);
    if (e != hipSuccess) return (int)e;
    e = hipMemcpyAsync(tail, tail_src, (size_t)tail_bytes, hipMemcpyHostToDevice, s);
    if (e != hipSuccess) return (int)e;
    return drl_ingest_scatter(tail, n_seg, seg_src_off, seg_dst, seg_words, n_win, win_src_off, win_dst, win_n, plane_base, s);
}

// HBM-resident prioritized-replay sum-tree for gfx950 (reference semantics: libs/sumtree.py).
//
// Design (wave64-first, not a translation of the reference's pointer-based binary tree):
//   * 64-ary tree: one wave loads a node's 64 children with one coalesced 256-B read, so 2 M leaves are
//     4 levels (2^21 -> 2^15 -> 2^9 -> 8) instead of 21 dependent loads of a binary descent.
//   * Sampling draws a FRESH uniform per level (u_l ~ U[0, S_node)) and picks the child by a wave-wide
//     inclusive scan + ballot. P(leaf) = prod_l (child/parent) = p_leaf / total exactly, and no residual is
//     carried down, so fp32 sums of 2 M priorities lose nothing at the leaf level. Draws are independent
//     and with replacement, like the reference (sumtree.py:133-137). A zero-priority leaf is never returned
//     (the reference can crash on that float edge, sumtree.py:57-61).
//   * Updates scatter leaves, then recompute every touched parent FROM ITS 64 CHILDREN level by level
//     (never +=delta), so there is no float drift (sumtree.py:20-26). Duplicate indices recompute the same
//     node to the same value -> deterministic.
//   * The tree is a ring: ingest overwrites the oldest slots (FIFO eviction, replay_memory.py:70-75) and
//     bumps a per-slot generation; priority write-backs carry the generation they sampled and are dropped
//     if the slot was overwritten meanwhile.
#include "common.h"
#include "sumtree_body.h"

__global__ void k_step_begin(StepState* st, const long long* n_items) {
    if (threadIdx.x == 0) {
        unsigned long long t = st->step;
        st->beta = fminf(1.0f, st->beta0 + st->beta_inc * (float)t);
        st->n_items = *n_items;
        st->rng_ctr += 1;
    }
}
__global__ void k_step_end(StepState* st) {
    if (threadIdx.x == 0) st->step += 1;
}

// One wave per draw: see sample_block (sumtree_body.h).
__global__ __launch_bounds__(1024) void k_sample(TreeDesc tr, StepState* st, unsigned long long seed, int B,
                                               int* __restrict__ idx_out, float* __restrict__ prio_out,
                                               const int* __restrict__ gen, int* __restrict__ gen_out,
                                               long long* __restrict__ frame_off, const long long* __restrict__ win,
                                               const long long* __restrict__ begin_n_items, float is_n_scale) {
    __shared__ float s_w[16];
    sample_block(tr, st, seed, B, idx_out, prio_out, gen, gen_out, frame_off, win, begin_n_items, is_n_scale, blockIdx.x, s_w);
}

// R2D2 learner: the draw and everything that only depends on it as ONE launch - they were three (k_sample, the window
// offsets of the drawn sequences, their stored recurrent state into the initial-state rows of both networks) plus two
// bias-sum launches, 17 + 11 us in front of the first conv of the step. A 256-thread block draws 4 sequences (one wave
// each, sample_block) and then serves those 4: off[t][b] = win[idx[b]][t] (time-major window offsets for conv1), hs / cs
// rows of one or two networks from h0 / c0, and - by global thread index - up to two vector sums out = a + b.
struct SeqAfter {
    const long long* win; long long* off; int T, H;
    const float* h0; const float* c0;
    bf16_t* hs_a; float* cs_a; bf16_t* hs_b; float* cs_b;
    const float* a0; const float* b0; float* o0; const float* a1; const float* b1; float* o1; int n_add;
};
__global__ __launch_bounds__(256) void k_seq_sample(TreeDesc tr, StepState* st, unsigned long long seed, int B,
                                                   int* __restrict__ idx_out, float* __restrict__ prio_out,
                                                   const int* __restrict__ gen, int* __restrict__ gen_out,
                                                   const long long* __restrict__ begin_n_items, SeqAfter a) {
    __shared__ float s_w[16];
    __shared__ int s_idx[4];
    const int gi = blockIdx.x * 256 + threadIdx.x;
    if (gi < a.n_add) {
        if (a.o0) a.o0[gi] = a.a0[gi] + a.b0[gi];
        if (a.o1) a.o1[gi] = a.a1[gi] + a.b1[gi];
    }
    if ((int)blockIdx.x * 4 >= B) return;                       // blocks that exist for the vector sums only
    sample_block(tr, st, seed, B, idx_out, prio_out, gen, gen_out, nullptr, nullptr, begin_n_items, 0.f, blockIdx.x, s_w);
    // the four draws of this block: lane 0 of wave w wrote idx_out[4 blk + w]; read it back past the L1 (same address, other wave)
    __threadfence_block();
    __syncthreads();
    if (threadIdx.x < 4) {
        const int b = blockIdx.x * 4 + threadIdx.x;
        s_idx[threadIdx.x] = b < B ? __hip_atomic_load(idx_out + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : -1;
    }
    __syncthreads();
    for (int q = 0; q < 4; ++q) {
        const int b = blockIdx.x * 4 + q, i = s_idx[q];
        if (i < 0) break;
        for (int t = threadIdx.x; t < a.T; t += 256) a.off[(long long)t * B + b] = a.win[(long long)i * a.T + t];
        for (int u = threadIdx.x; u < a.H; u += 256) {
            const long long o = (long long)i * a.H + u, d = (long long)b * a.H + u;
            const float h = a.h0[o], c = a.c0[o];
            const bf16_t hb = f2bf(h);
            a.hs_a[d] = hb; a.cs_a[d] = c;
            if (a.hs_b) { a.hs_b[d] = hb; a.cs_b[d] = c; }
        }
    }
}

// Leaves <- new priorities (dropped when the slot generation changed since sampling).
__global__ void k_set_leaves(TreeDesc tr, const int* __restrict__ idx, const float* __restrict__ val, int n,
                             const int* __restrict__ gen, const int* __restrict__ gen_expected) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int s = idx[i];
    if (s < 0 || s >= tr.cap) return;
    if (gen && gen_expected && gen[s] != gen_expected[i]) return;
    tr.base[tr.off[0] + s] = val[i];
}

// Recompute parents at `level` (>=1) of the given leaf indices from their 64 children. One wave per index.
__global__ __launch_bounds__(256) void k_fix_level(TreeDesc tr, int level, const int* __restrict__ idx, int n) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (i >= n) return;
    int s = idx[i];
    if (s < 0 || s >= tr.cap) return;
    long long node = (long long)s >> (6 * level);
    float v = tr.base[tr.off[level - 1] + node * 64 + lane];
    v = wave_sum(v);
    if (lane == 0) tr.base[tr.off[level] + node] = v;
}

// ---- fused priority write-back: 2 launches for any tree depth ---------------------------------------------------
// K1 (one wave per index): leaves + level 1. A wave applies EVERY batch entry that falls into its level-1 node (it
// scans the index list), so the 64 children it sums are final no matter which other waves run when: no inter-block
// dependency, duplicates resolve to the entry with the highest batch position (deterministic), all waves of a node
// write identical values.
__global__ __launch_bounds__(256) void k_update_l01(TreeDesc tr, const int* __restrict__ idx, const float* __restrict__ val,
                                                   int n, const int* __restrict__ gen, const int* __restrict__ gen_expected,
                                                   const int* __restrict__ fault) {
    if (fault && *fault) return;          // a discarded step (optim.hip): its priorities come from garbage Q values
    tree_update_l01_block(tr, idx, val, n, gen, gen_expected, blockIdx.x);
}

// K2: every node of every level >= 2 is rebuilt directly FROM LEVEL 1 (a level-l node covers 64^(l-1) level-1 values),
// so the levels do not depend on each other and one launch rebuilds them all, exactly (no += delta drift). Level 2:
// one wave per node; levels >= 3: one block per node. Block 0 optionally ends the step (step += 1).
__global__ __launch_bounds__(256) void k_rebuild_upper(TreeDesc tr, int n2_blocks, StepState* st_end) {
    __shared__ float red[17];
    tree_rebuild_upper_block(tr, n2_blocks, st_end, blockIdx.x, red);
}

// Ring ingest: slots [start, start+n) mod cap get new priorities and a new generation. prio == nullptr RETIRES the
// slots (priority 0: they cannot be drawn any more while their data is overwritten asynchronously; engine/ingest.py).
// The live item count (a device scalar read by the sampler) rides along instead of costing a launch of its own.
__global__ void k_ingest(TreeDesc tr, long long start, int n, const float* __restrict__ prio,
                         int* __restrict__ gen, int* __restrict__ idx_out, long long* __restrict__ n_items_out,
                         long long n_items_val) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0 && n_items_out) *n_items_out = n_items_val;
    if (i >= n) return;
    long long s = (start + i) % tr.cap;
    tr.base[tr.off[0] + s] = prio ? prio[i] : 0.f;
    if (gen) gen[s] += 1;
    idx_out[i] = (int)s;
}

// Generic row gather: dst[i] = src[byte offset off[i]], rows of row_bytes (multiple of 16).
__global__ __launch_bounds__(256) void k_gather_rows(const uint8_t* __restrict__ src, const long long* __restrict__ off,
                                                    uint8_t* __restrict__ dst, int row_bytes) {
    const uint4* s = reinterpret_cast<const uint4*>(src + off[blockIdx.x]);
    uint4* d = reinterpret_cast<uint4*>(dst + (long long)blockIdx.x * row_bytes);
    for (int k = threadIdx.x; k < row_bytes / 16; k += blockDim.x) d[k] = s[k];
}

template <typename T>
__global__ void k_gather_scalar(const T* __restrict__ src, const int* __restrict__ idx, T* __restrict__ dst, int n, int inner) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * inner) return;
    dst[i] = src[(long long)idx[i / inner] * inner + (i % inner)];
}

DRL_API int drl_step_begin(void* st, const long long* n_items, hipStream_t s) {
    k_step_begin<<<1, 64, 0, s>>>((StepState*)st, n_items);
    return (int)hipGetLastError();
}
DRL_API int drl_step_end(void* st, hipStream_t s) {
    k_step_end<<<1, 64, 0, s>>>((StepState*)st);
    return (int)hipGetLastError();
}

DRL_API int drl_sumtree_sample(float* base, int n_levels, const long long* off, long long cap, void* st,
                               unsigned long long seed, int B, int* idx_out, float* prio_out, const int* gen,
                               int* gen_out, long long* frame_off, const long long* win, const long long* begin_n_items,
                               float is_n_scale, hipStream_t s) {
    if (n_levels < 1 || n_levels > MAX_LEVELS || B <= 0) return -1;
    TreeDesc d = make_desc(base, n_levels, off, cap);
    const int wpb = (begin_n_items && is_n_scale > 0.f) ? 16 : 4;           // waves per block (one atomic per block, see kernel)
    k_sample<<<(B + wpb - 1) / wpb, wpb * 64, 0, s>>>(d, (StepState*)st, seed, B, idx_out, prio_out, gen, gen_out, frame_off,
                                                     win, begin_n_items, is_n_scale);
    return (int)hipGetLastError();
}

DRL_API int drl_seq_sample(float* base, int n_levels, const long long* off, long long cap, void* st, unsigned long long seed, int B,
                           int* idx_out, float* prio_out, const int* gen, int* gen_out, const long long* begin_n_items,
                           const long long* win, long long* frame_off, int T, const float* h0, const float* c0, int H, void* hs_a,
                           float* cs_a, void* hs_b, float* cs_b, const float* a0, const float* b0, float* o0, const float* a1,
                           const float* b1, float* o1, int n_add, hipStream_t s) {
    if (n_levels < 1 || n_levels > MAX_LEVELS || B <= 0 || T <= 0 || H <= 0 || n_add < 0) return -1;
    if (!idx_out || !prio_out || !win || !frame_off || !h0 || !c0 || !hs_a || !cs_a || (!hs_b) != (!cs_b)) return -1;
    if ((o0 && (!a0 || !b0)) || (o1 && (!a1 || !b1))) return -1;
    if (!o0 && !o1) n_add = 0;
    TreeDesc d = make_desc(base, n_levels, off, cap);
    SeqAfter a{win, frame_off, T, H, h0, c0, (bf16_t*)hs_a, cs_a, (bf16_t*)hs_b, cs_b, a0, b0, o0, a1, b1, o1, n_add};
    const int nb = (B + 3) / 4, na = (n_add + 255) / 256;
    k_seq_sample<<<nb > na ? nb : na, 256, 0, s>>>(d, (StepState*)st, seed, B, idx_out, prio_out, gen, gen_out, begin_n_items, a);
    return (int)hipGetLastError();
}

static void rebuild_upper(const TreeDesc& d, void* st_end, hipStream_t s) {
    int n2_blocks;
    const long long blocks = tree_upper_blocks(d, &n2_blocks);
    if (blocks == 0) {
        if (st_end) k_step_end<<<1, 64, 0, s>>>((StepState*)st_end);
        return;
    }
    k_rebuild_upper<<<(unsigned)blocks, 256, 0, s>>>(d, n2_blocks, (StepState*)st_end);
}

// set leaves + recompute all ancestor levels
DRL_API int drl_sumtree_update(float* base, int n_levels, const long long* off, long long cap, const int* idx,
                               const float* val, int n, const int* gen, const int* gen_expected, void* st_end,
                               const int* fault, hipStream_t s) {
    if (n_levels < 1 || n_levels > MAX_LEVELS || n <= 0) return -1;
    TreeDesc d = make_desc(base, n_levels, off, cap);
    k_update_l01<<<(n + 3) / 4, 256, 0, s>>>(d, idx, val, n, gen, gen_expected, fault);
    rebuild_upper(d, st_end, s);          // levels >= 2 (+ optional fused step_end)
    return (int)hipGetLastError();
}

DRL_API int drl_sumtree_ingest(float* base, int n_levels, const long long* off, long long cap, long long start, int n,
                               const float* prio, int* gen, int* idx_scratch, long long* n_items_out,
                               long long n_items_val, hipStream_t s) {
    if (n_levels < 1 || n_levels > MAX_LEVELS || n <= 0 || n > cap) return -1;
    TreeDesc d = make_desc(base, n_levels, off, cap);
    k_ingest<<<(n + 255) / 256, 256, 0, s>>>(d, start, n, prio, gen, idx_scratch, n_items_out, n_items_val);
    if (n_levels > 1) k_fix_level<<<(n + 3) / 4, 256, 0, s>>>(d, 1, idx_scratch, n);
    rebuild_upper(d, nullptr, s);
    return (int)hipGetLastError();
}

DRL_API int drl_gather_rows(const void* src, const long long* off, void* dst, int row_bytes, int n, hipStream_t s) {
    if (row_bytes % 16 != 0 || n <= 0) return -1;
    k_gather_rows<<<n, 256, 0, s>>>((const uint8_t*)src, off, (uint8_t*)dst, row_bytes);
    return (int)hipGetLastError();
}
DRL_API int drl_gather_f32(const float* src, const int* idx, float* dst, int n, int inner, hipStream_t s) {
    k_gather_scalar<float><<<(n * inner + 255) / 256, 256, 0, s>>>(src, idx, dst, n, inner);
    return (int)hipGetLastError();
}
DRL_API int drl_gather_i32(const int* src, const int* idx, int* dst, int n, int inner, hipStream_t s) {
    k_gather_scalar<int><<<(n * inner + 255) / 256, 256, 0, s>>>(src, idx, dst, n, inner);
    return (int)hipGetLastError();
}

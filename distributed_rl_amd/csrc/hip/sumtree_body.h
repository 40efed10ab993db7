// Sum-tree descriptor and the block bodies of the priority write-back, shared by sumtree.hip (stand-alone kernels) and
// the fused backward launches of gemm.hip (the write-back only depends on the TD kernel's output, so it rides along
// with the conv backward instead of extending the tail of the step).
#pragma once
#include "common.h"

#define MAX_LEVELS 6

struct TreeDesc {
    float* base;
    int n_levels;               // level 0 = leaves ... level n_levels-1 = top (<= 64 nodes)
    long long off[MAX_LEVELS];  // element offset of each level inside base
    long long cap;              // number of leaf slots (ring capacity)
};

static inline TreeDesc make_desc(float* base, int n_levels, const long long* off, long long cap) {
    TreeDesc d;
    d.base = base; d.n_levels = n_levels; d.cap = cap;
    for (int i = 0; i < MAX_LEVELS; ++i) d.off[i] = i < n_levels ? off[i] : 0;
    return d;
}

// number of blocks of the upper-level rebuild (0 = the tree has no level >= 2)
static inline long long tree_upper_blocks(const TreeDesc& d, int* n2_blocks) {
    *n2_blocks = 0;
    if (d.n_levels < 3) return 0;
    const long long n1 = d.off[2] - d.off[1];
    *n2_blocks = (int)(((n1 + 63) / 64 + 3) / 4);
    long long blocks = *n2_blocks, span = 64LL * 64;
    for (int l = 3; l < d.n_levels; ++l, span *= 64) blocks += (n1 + span - 1) / span;
    return blocks;
}

// K1 (one wave per index, 4 per block): leaves + level 1. A wave applies EVERY batch entry that falls into its level-1
// node (it scans the index list), so the 64 children it sums are final no matter which other waves run when: no
// inter-block dependency, duplicates resolve to the entry with the highest batch position (deterministic), all waves
// of a node write identical values.
__device__ __forceinline__ void tree_update_l01_block(const TreeDesc& tr, const int* __restrict__ idx, const float* __restrict__ val,
                                                      int n, const int* __restrict__ gen, const int* __restrict__ gen_expected,
                                                      int block) {
    const int lane = threadIdx.x & 63;
    const int i = block * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (i >= n) return;
    const int s = idx[i];
    if (s < 0 || s >= tr.cap) return;
    if (gen && gen_expected && gen[s] != gen_expected[i]) return;
    const long long node = (long long)s >> 6;
    float v = tr.base[tr.off[0] + node * 64 + lane];
    for (int c = 0; c < n; c += 64) {
        const int j = c + lane;
        int sj = -1;
        if (j < n) {
            sj = idx[j];
            if (sj < 0 || sj >= tr.cap || ((long long)sj >> 6) != node) sj = -1;
            else if (gen && gen_expected && gen[sj] != gen_expected[j]) sj = -1;
        }
        unsigned long long m = __ballot(sj >= 0);
        while (m) {
            const int bpos = __ffsll((long long)m) - 1;
            m &= m - 1;
            const int t = __shfl(sj, bpos, WAVE);
            const float x = val[c + bpos];
            if (lane == (t & 63)) v = x;
        }
    }
    tr.base[tr.off[0] + node * 64 + lane] = v;
    if (tr.n_levels > 1) {
        const float sum = wave_sum(v);
        if (lane == 0) tr.base[tr.off[1] + node] = sum;
    }
}

// K2: every node of every level >= 2 is rebuilt directly FROM LEVEL 1 (a level-l node covers 64^(l-1) level-1 values),
// so the levels do not depend on each other and one launch rebuilds them all, exactly (no += delta drift). Level 2:
// one wave per node; levels >= 3: one block per node. Block 0 optionally ends the step (step += 1). red = 17 floats.
__device__ __forceinline__ void tree_rebuild_upper_block(const TreeDesc& tr, int n2_blocks, StepState* st_end, int block, float* red) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (block == 0 && threadIdx.x == 0 && st_end) st_end->step += 1;
    if (tr.n_levels < 3) return;
    const long long n1 = tr.off[2] - tr.off[1];                    // padded level-1 size (zeros past the live nodes)
    const float* l1 = tr.base + tr.off[1];
    if (block < n2_blocks) {
        const long long node = (long long)block * 4 + wave;
        const long long n2 = (n1 + 63) / 64;
        if (node >= n2) return;
        const long long k = node * 64 + lane;
        const float v = wave_sum(k < n1 ? l1[k] : 0.f);
        if (lane == 0) tr.base[tr.off[2] + node] = v;
        return;
    }
    // levels >= 3: find (level, node) of this block
    long long rest = (long long)block - n2_blocks, span = 64LL * 64;
    int lvl = 3;
    for (; lvl < tr.n_levels; ++lvl, span *= 64) {
        const long long nl = (n1 + span - 1) / span;
        if (rest < nl) break;
        rest -= nl;
    }
    if (lvl >= tr.n_levels) return;
    const long long k0 = rest * span, k1 = min(n1, k0 + span);
    float acc = 0.f;
    for (long long k = k0 + threadIdx.x; k < k1; k += 256) acc += l1[k];
    acc = block_sum(acc, red);
    if (threadIdx.x == 0) tr.base[tr.off[lvl] + rest] = acc;
}

// PER sampling, one wave per draw; ``block`` = index of this block among ceil(B / waves-per-block), s_w = 16 floats of LDS.
// Shared by k_sample (sumtree.hip) and the optimizer launch of the Ape-X learner (optim.hip), where the sample of the
// NEXT step rides along: by then nothing of the current step reads the sample buffers any more.
__device__ __forceinline__ void sample_block(const TreeDesc& tr, StepState* st, unsigned long long seed, int B,
                                             int* __restrict__ idx_out, float* __restrict__ prio_out,
                                             const int* __restrict__ gen, int* __restrict__ gen_out,
                                             long long* __restrict__ frame_off, const long long* __restrict__ win,
                                             const long long* __restrict__ begin_n_items, float is_n_scale, int block, float* s_w) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = block * (blockDim.x >> 6) + wave;
    float w_is = 0.f;                                      // this draw's un-normalised IS weight (lane 0)
    if (b < B) {                                           // whole waves take the branch together
    Philox ph(seed);
    // fused step_begin (learner engines): the Philox counter of step t is rng_ctr + t + 1 (nobody writes rng_ctr or step
    // while this kernel runs), and wave 0 publishes beta / n_items for the kernels that follow
    unsigned long long ctr = st->rng_ctr;
    const unsigned long long t = st->step;
    if (begin_n_items) {
        ctr += t + 1;
        if (b == 0 && lane == 0) {
            st->beta = fminf(1.0f, st->beta0 + st->beta_inc * (float)t);
            st->n_items = *begin_n_items;
            // clear the NEXT step's max-weight slot and THIS step's loss slot (atomically summed by the TD kernel)
            if (is_n_scale > 0.f) { st->wmax_bits[(t + 1) & 1] = 0u; *loss_slot(st, t) = 0.f; }
        }
    }
    float top_sum = 0.f;
    uint4 r = ph(ctr, (unsigned long long)b);
    uint32_t rnd[4] = {r.x, r.y, r.z, r.w};
    long long node = 0;
    float leaf_p = 0.f;
    for (int lvl = tr.n_levels - 1, k = 0; lvl >= 0; --lvl, ++k) {
        if (k == 4) { r = ph(ctr, (unsigned long long)b + ((unsigned long long)1 << 40)); rnd[0] = r.x; rnd[1] = r.y; rnd[2] = r.z; rnd[3] = r.w; k = 0; }
        float v = tr.base[tr.off[lvl] + node * 64 + lane];
        float incl = wave_scan_incl(v, lane);
        float S = __shfl(incl, 63, WAVE);
        if (lvl == tr.n_levels - 1) { top_sum = S; if (b == 0 && lane == 0) st->total_prio = S; }
        float u = u01(rnd[k]) * S;
        unsigned long long hit = __ballot(incl > u && v > 0.f);
        unsigned long long nz = __ballot(v > 0.f);
        int pick;
        if (hit) pick = __ffsll((long long)hit) - 1;
        else if (nz) pick = 63 - __clzll((long long)nz);          // float edge: take the last non-empty child
        else pick = 0;                                             // empty tree
        leaf_p = __shfl(v, pick, WAVE);
        node = node * 64 + pick;
    }
    if (lane == 0) {
        int i = (int)(node < tr.cap ? node : tr.cap - 1);
        idx_out[b] = i;
        prio_out[b] = leaf_p;
        if (begin_n_items && is_n_scale > 0.f && leaf_p > 0.f && top_sum > 0.f) {
            // un-normalised IS weight (N p / total)^-beta of this draw -> running max of the batch (consumed by the fused
            // TD kernel, head.hip k_apex_td_dh, which therefore needs no batch-wide reduction of its own)
            const float beta = fminf(1.0f, st->beta0 + st->beta_inc * (float)t);
            w_is = __powf((float)(*begin_n_items) * is_n_scale * leaf_p / top_sum, -beta);
        }
        if (gen_out) gen_out[b] = gen ? gen[i] : 0;
        if (frame_off) {                                   // byte offsets of the plane windows of s and s' in the frame store
            frame_off[b] = win[i];                         // (per-slot table written at ingest: replay/device_per.py)
            frame_off[B + b] = win[tr.cap + i];
        }
    }
    }
    // one atomicMax per BLOCK: same-address atomics serialise in L2 (~35 ns each), 512 of them doubled this kernel's time
    if (begin_n_items && is_n_scale > 0.f) {
        if (lane == 0) s_w[wave] = w_is;
        __syncthreads();
        if (threadIdx.x == 0) {
            float m = 0.f;
            for (int i = 0; i < (int)(blockDim.x >> 6); ++i) m = fmaxf(m, s_w[i]);
            if (m > 0.f) atomicMax(&st->wmax_bits[st->step & 1], __float_as_uint(m));
        }
    }
}

// PER sample of the NEXT step riding along with the conv1 backward launch (Ape-X learner, gemm.hip SampleBody): host-side
// description, mirrored field by field by ops/hip.py (ctypes). kind 0 = none.
struct SampleExtra {
    int kind, n_levels;
    float* base;
    long long off[MAX_LEVELS];
    long long cap;
    void* st;
    unsigned long long seed;
    int B, pad;
    int* idx_out;
    float* prio_out;
    const int* gen;
    int* gen_out;
    long long* frame_off;
    const long long* win;
    const long long* begin_n_items;
    float is_n_scale, pad2;
};
static_assert(sizeof(SampleExtra) == 160, "SampleExtra is mirrored by ops/hip.py (ctypes)");

struct SampleArgs {           // device-side copy
    TreeDesc tr;
    StepState* st;
    unsigned long long seed;
    int B;
    int* idx_out;
    float* prio_out;
    const int* gen;
    int* gen_out;
    long long* frame_off;
    const long long* win;
    const long long* begin_n_items;
    float is_n_scale;
};

static inline int fill_sample_args(const void* extra_, SampleArgs* sa) {     // returns the number of sampler blocks (0 = none)
    const SampleExtra* e = (const SampleExtra*)extra_;
    sa->B = 0;
    if (!e || e->kind == 0) return 0;
    if (e->n_levels < 1 || e->n_levels > MAX_LEVELS || e->B <= 0) return -1;
    sa->tr = make_desc(e->base, e->n_levels, e->off, e->cap);
    sa->st = (StepState*)e->st; sa->seed = e->seed; sa->B = e->B; sa->idx_out = e->idx_out; sa->prio_out = e->prio_out;
    sa->gen = e->gen; sa->gen_out = e->gen_out; sa->frame_off = e->frame_off; sa->win = e->win;
    sa->begin_n_items = e->begin_n_items; sa->is_n_scale = e->is_n_scale;
    return (e->B + 3) / 4;
}


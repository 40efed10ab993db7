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

// Conv / FC backward helpers (SURVEY G1-G3 backward): the memory-bound leftovers next to the MFMA kernels of gemm.hip
// and conv_halo.hip - column sums of dY for bias gradients and a standalone ReLU-backward mask (used after a library
// GEMM, which has no masking epilogue). The input gradient itself is an implicit GEMM (gemm.hip, DgradOp) or an LDS-halo
// convolution (conv_halo.hip); the dCol GEMM + col2im gather pair it replaced (58 / 84 MB of HBM traffic per Ape-X step
// for conv3 / conv2) is gone.
#include "common.h"
#include "colsum_body.h"

// db[n] += sum_m dY[m][n]: see colsum_body.h. grid = ceil(chunks / cpb), block = 256.
__global__ __launch_bounds__(256) void k_colsum(const bf16_t* __restrict__ dy, long long total, int NC8, int cpb,
                                               float* __restrict__ db) {
    __shared__ float red[256 * 9];
    colsum_block(dy, total, NC8, cpb, db, blockIdx.x, red);
}


[[maybe_unused]] static int grid1d(long long n) {
    long long b = (n + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

DRL_API int drl_colsum(const void* dy, int M, int N, float* db, hipStream_t s) {
    if (M <= 0 || N < 8 || N > 2048 || (N & (N - 1))) return -1;   // N/8 must divide 256
    const long long total = (long long)M * (N / 8);
    const long long cpb = colsum_cpb(total);
    k_colsum<<<(unsigned)((total + cpb - 1) / cpb), 256, 0, s>>>((const bf16_t*)dy, total, N / 8, (int)cpb, db);
    return (int)hipGetLastError();
}


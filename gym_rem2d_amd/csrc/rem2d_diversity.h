// rem2d_diversity.h -- pairwise tree distance of a population (the reference's diversity metric,
// DataAnalysis/AdvancedDataAnalysis.py:291-313,367-381).  Part of the single translation unit rem2d.hip.
//
// out[c] = sum over t != c of  #{i in c : pos_c[i] not in t} + #{j in t : pos_t[j] not in c},  positions
// compared as binary64 with == (so -0 == +0 and NaN != NaN, like the Python floats of the reference).
// One workgroup = one tree c (its positions in LDS); its 256 lanes stride over the other trees t, each lane
// streaming t's positions from HBM/L2 (16 B per node, read once per (c, t) pair: N^2 * n * 16 B in total, all
// L2 hits -- the population is N * 1 KB) and keeping a 64-bit "found" mask of t's nodes in a register pair.
// Integer result, so the reduction order does not matter.  Bound: VALU compares (4 per node pair).
#ifndef REM2D_DIVERSITY_H
#define REM2D_DIVERSITY_H

#define DIV_THREADS 256
#define DIV_MAX_NODES 64

__global__ __launch_bounds__(DIV_THREADS) void rem2d_tree_diversity_kernel(const double *__restrict__ pos,
                                                                           const int *__restrict__ cnt, int n, int stride,
                                                                           long long *__restrict__ out) {
    __shared__ double cx[DIV_MAX_NODES], cy[DIV_MAX_NODES];
    __shared__ long long partial[DIV_THREADS / WAVE];
    const int c = blockIdx.x;
    const int nc = cnt[c];
    if (threadIdx.x < nc) {
        cx[threadIdx.x] = pos[((size_t)c * stride + threadIdx.x) * 2 + 0];
        cy[threadIdx.x] = pos[((size_t)c * stride + threadIdx.x) * 2 + 1];
    }
    __syncthreads();
    long long acc = 0;
    for (int t = threadIdx.x; t < n; t += DIV_THREADS) {
        if (t == c) continue;
        const int nt = cnt[t];
        const double *pt = pos + (size_t)t * stride * 2;
        unsigned long long matchedC = 0ull; // bit i: node i of c sits on some node of t
        int foundT = 0;                     // nodes of t that some node of c sits on
        for (int j = 0; j < nt; ++j) {
            const double tx = pt[2 * j], ty = pt[2 * j + 1];
            bool any = false;
            for (int i = 0; i < nc; ++i) {
                const bool eq = (cx[i] == tx) && (cy[i] == ty);
                any |= eq;
                matchedC |= (unsigned long long)eq << i;
            }
            foundT += any ? 1 : 0;
        }
        acc += (nc - __popcll(matchedC)) + (nt - foundT);
    }
    // workgroup reduction (integers: order-free)
#pragma unroll
    for (int o = 1; o < WAVE; o <<= 1) acc += __shfl_xor(acc, o);
    if ((threadIdx.x & (WAVE - 1)) == 0) partial[threadIdx.x / WAVE] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        long long s = 0;
        for (int k = 0; k < DIV_THREADS / WAVE; ++k) s += partial[k];
        out[c] = s;
    }
}

#endif

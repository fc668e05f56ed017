// mfma_util.h -- the software-pipelined MFMA loop shared by the convolution kernels.
#pragma once
#include "dm_common.h"

// The MFMA loop of MP M-tiles x NT N-tiles.  off(s) is the compile-time LDS offset of K step s.
// Software pipeline over chunks of CH K-steps: the LDS operands of chunk c+1 are requested before the
// MFMAs of chunk c are issued (two register buffers); sched_barrier keeps hipcc from sinking the reads
// back next to their uses, which would expose the LDS latency once per MFMA group.
template <int MP, int NT, int KS, int CH, class OFF>
__device__ __forceinline__ void mfma_tiles(const float *const (&ap)[MP], const float (&wreg)[NT][KS],
                                           f32x4 (&acc)[MP][NT], OFF off)
{
    constexpr int NC = (KS + CH - 1) / CH;
    float av[2][MP][CH];
#pragma unroll
    for (int j = 0; j < CH; ++j)
        if (j < KS) {
#pragma unroll
            for (int i = 0; i < MP; ++i) av[0][i][j] = ap[i][off(j)];
        }
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        if (c + 1 < NC) {
#pragma unroll
            for (int j = 0; j < CH; ++j) {
                const int s = (c + 1) * CH + j;
                if (s < KS) {
#pragma unroll
                    for (int i = 0; i < MP; ++i) av[(c + 1) & 1][i][j] = ap[i][off(s)];
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            const int s = c * CH + j;
            if (s < KS) {
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int i = 0; i < MP; ++i)
                        acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[c & 1][i][j], wreg[t][s], acc[i][t], 0, 0, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}


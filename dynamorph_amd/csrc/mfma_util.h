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


// ---- split-bf16 form (tile.h: split_pack4): the LDS tile and the weight registers hold (hi, lo) bf16 pairs in the f32
// slots; four K-steps make the 8-slot operand of one v_mfma_f32_16x16x32_bf16, and the product with the operand's halves
// swapped (one v_alignbit per register) supplies the cross terms.  KS % 4 == 0; K-steps in quads, the LDS operands of
// quad q+1 requested before the products of quad q.
typedef unsigned dm_u32x4_t __attribute__((ext_vector_type(4)));
typedef __bf16 dm_bf16x8_t __attribute__((ext_vector_type(8)));
__device__ __forceinline__ dm_u32x4_t dm_rot16(dm_u32x4_t v)
{
    return (dm_u32x4_t){__builtin_amdgcn_alignbit(v.x, v.x, 16), __builtin_amdgcn_alignbit(v.y, v.y, 16),
                        __builtin_amdgcn_alignbit(v.z, v.z, 16), __builtin_amdgcn_alignbit(v.w, v.w, 16)};
}
__device__ __forceinline__ f32x4 dm_mfma_split(dm_u32x4_t a, dm_u32x4_t ar, dm_u32x4_t b, f32x4 acc)
{
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(dm_bf16x8_t, a), __builtin_bit_cast(dm_bf16x8_t, b), acc, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(dm_bf16x8_t, ar), __builtin_bit_cast(dm_bf16x8_t, b), acc, 0, 0, 0);
}

template <int MP, int NT, int KS, class OFF>
__device__ __forceinline__ void mfma_tiles_split(const float *const (&ap)[MP], const float (&wreg)[NT][KS],
                                                 f32x4 (&acc)[MP][NT], OFF off)
{
    static_assert(KS % 4 == 0, "K-steps in quads");
    constexpr int NQ = KS / 4;
    float av[2][MP][4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < MP; ++i) av[0][i][j] = ap[i][off(j)];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        if (q + 1 < NQ) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < MP; ++i) av[(q + 1) & 1][i][j] = ap[i][off(4 * (q + 1) + j)];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < MP; ++i) {
            const float(&a)[4] = av[q & 1][i];
            const dm_u32x4_t a4 = {__builtin_bit_cast(unsigned, a[0]), __builtin_bit_cast(unsigned, a[1]),
                                   __builtin_bit_cast(unsigned, a[2]), __builtin_bit_cast(unsigned, a[3])};
            const dm_u32x4_t ar = dm_rot16(a4);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const dm_u32x4_t b4 = {__builtin_bit_cast(unsigned, wreg[t][4 * q]), __builtin_bit_cast(unsigned, wreg[t][4 * q + 1]),
                                       __builtin_bit_cast(unsigned, wreg[t][4 * q + 2]), __builtin_bit_cast(unsigned, wreg[t][4 * q + 3])};
                acc[i][t] = dm_mfma_split(a4, ar, b4, acc[i][t]);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// pairwise.hip -- the pairwise mean-squared latent distance of the time-matching loss.
//
// Reference: HiddenStateExtractor/vq_vae.py:324-329 and vae.py:322-326
//     z = z_before.reshape(B, -1);  sim_mat = pow(z.reshape(1, B, n) - z.reshape(B, 1, n), 2).mean(2)
// which materialises a (B, B, n) tensor (n = 4096: 268 MB at B = 128, 68 GB at B = 2048).  Here the (B, B)
// result is formed directly; differences are taken before squaring (no |a|^2 + |b|^2 - 2ab cancellation).
// The weighting / hinge / mean of the (B, B) matrix stays in torch (it is B*B elements); its autograd hands
// g = dL/dsim back to dm_pair_msd_backward:
//     dz[i] = (2/n) * sum_j (g[i][j] + g[j][i]) * (z[i] - z[j])
#include "dm_common.h"

namespace {

constexpr int PM_T = 16;          // pairs block: 16 x 16
constexpr int PM_CH = 64;         // latent elements staged per step

// sim[i][j] = (1/n) sum_d (z[i][d] - z[j][d])^2 ; one workgroup per 16 x 16 block of pairs, thread = one pair
__global__ __launch_bounds__(256) void pair_msd_kernel(const float *__restrict__ z, float *__restrict__ sim, int B, int n)
{
    __shared__ float sI[PM_T][PM_CH + 1], sJ[PM_T][PM_CH + 1];
    const int ti = threadIdx.x >> 4, tj = threadIdx.x & 15;
    const int i0 = blockIdx.y * PM_T, j0 = blockIdx.x * PM_T;
    double acc = 0.0;
    for (int d0 = 0; d0 < n; d0 += PM_CH) {
        __syncthreads();
        for (int e = threadIdx.x; e < PM_T * PM_CH; e += 256) {
            const int r = e / PM_CH, c = e % PM_CH;
            const int d = d0 + c;
            sI[r][c] = (i0 + r < B && d < n) ? z[(long long)(i0 + r) * n + d] : 0.f;
            sJ[r][c] = (j0 + r < B && d < n) ? z[(long long)(j0 + r) * n + d] : 0.f;
        }
        __syncthreads();
        float part = 0.f;
#pragma unroll 16
        for (int c = 0; c < PM_CH; ++c) {
            const float df = sJ[tj][c] - sI[ti][c];
            part += df * df;
        }
        acc += (double)part;
    }
    if (i0 + ti < B && j0 + tj < B) sim[(long long)(i0 + ti) * B + j0 + tj] = (float)(acc / (double)n);
}

// dz[i][d] = (2/n) * sum_j s_ij * (z[i][d] - z[j][d]),  s_ij = g[i][j] + g[j][i]
// one workgroup per (16 rows i) x (256 elements d); j walks in chunks of 16 through LDS
__global__ __launch_bounds__(256) void pair_msd_backward_kernel(const float *__restrict__ z, const float *__restrict__ g,
                                                                float *__restrict__ dz, int B, int n)
{
    __shared__ float sS[PM_T][PM_T + 1];            // s_ij for the block's 16 rows i and the current 16 columns j
    __shared__ float sZ[PM_T][256];                 // z[j][d chunk]
    const int i0 = blockIdx.y * PM_T, d = blockIdx.x * 256 + threadIdx.x;
    float zi[PM_T];
    double acc[PM_T];
#pragma unroll
    for (int r = 0; r < PM_T; ++r) {
        zi[r] = (i0 + r < B && d < n) ? z[(long long)(i0 + r) * n + d] : 0.f;
        acc[r] = 0.0;
    }
    for (int j0 = 0; j0 < B; j0 += PM_T) {
        __syncthreads();
        {
            const int r = threadIdx.x >> 4, c = threadIdx.x & 15;
            const int i = i0 + r, j = j0 + c;
            sS[r][c] = (i < B && j < B) ? g[(long long)i * B + j] + g[(long long)j * B + i] : 0.f;
        }
#pragma unroll
        for (int r = 0; r < PM_T; ++r) sZ[r][threadIdx.x] = (j0 + r < B && d < n) ? z[(long long)(j0 + r) * n + d] : 0.f;
        __syncthreads();
#pragma unroll
        for (int r = 0; r < PM_T; ++r) {
            float part = 0.f;
#pragma unroll
            for (int c = 0; c < PM_T; ++c) part += sS[r][c] * (zi[r] - sZ[c][threadIdx.x]);
            acc[r] += (double)part;
        }
    }
    if (d < n) {
        const double sc = 2.0 / (double)n;
#pragma unroll
        for (int r = 0; r < PM_T; ++r)
            if (i0 + r < B) dz[(long long)(i0 + r) * n + d] = (float)(acc[r] * sc);
    }
}

}  // namespace

extern "C" int dm_pair_msd(const float *z, float *sim, int B, int n, void *stream)
{
    DM_REQUIRE(z && sim && B > 0 && n > 0, "dm_pair_msd: bad argument");
    DM_REQUIRE((long long)B * n < (1LL << 31), "dm_pair_msd: tensor too large for 32-bit offsets");
    const int nb = (B + PM_T - 1) / PM_T;
    hipLaunchKernelGGL(pair_msd_kernel, dim3(nb, nb), dim3(256), 0, (hipStream_t)stream, z, sim, B, n);
    return dm_launch_status("dm_pair_msd");
}

extern "C" int dm_pair_msd_backward(const float *z, const float *g_sim, float *dz, int B, int n, void *stream)
{
    DM_REQUIRE(z && g_sim && dz && B > 0 && n > 0, "dm_pair_msd_backward: bad argument");
    DM_REQUIRE((long long)B * n < (1LL << 31), "dm_pair_msd_backward: tensor too large for 32-bit offsets");
    hipLaunchKernelGGL(pair_msd_backward_kernel, dim3((n + 255) / 256, (B + PM_T - 1) / PM_T), dim3(256), 0,
                       (hipStream_t)stream, z, g_sim, dz, B, n);
    return dm_launch_status("dm_pair_msd_backward");
}

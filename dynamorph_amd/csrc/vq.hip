// vq.hip -- VectorQuantizer kernels (reference: HiddenStateExtractor/vq_vae.py:52-116).
//
// Compiled with -ffp-contract=off: the reference materialises (z - e) and (z - e)^2 in
// fp32 before summing over d (no FMA), and ATen's CPU reduction adds the squares
// sequentially inside blocks of 16 consecutive d, block sums again sequentially.  The
// kernels reproduce that order so argmin indices are bit-identical to the CPU path.
//
// Layout: one thread per latent position (b,h,w); the D-vector of a position is strided
// by H*W in NCHW, so for every d a wave reads 64 consecutive floats (256 B, coalesced)
// and keeps its D values in registers.  Codes are processed two at a time as packed fp32
// (v_pk_add_f32 / v_pk_mul_f32, IEEE round-to-nearest, same results as scalar ops); the
// codebook is first re-laid out as [K/2][D][2] so a pair's operands are adjacent, staged in
// LDS per workgroup and read with wave-uniform (broadcast) ds_read_b128.
#include "dm_common.h"

namespace {

constexpr int VQ_BLOCK = 256;
constexpr int VQ_MAX_LDS_HIST = 4096;

__global__ void vq_prep_kernel(const float *__restrict__ cb, float *__restrict__ cbT, int K, int D,
                               int *__restrict__ hist, double *__restrict__ slabs, int nslabs)
{
    // also clears the outputs the forward kernel accumulates into (no separate memset launches)
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < K; i += gridDim.x * blockDim.x) hist[i] = 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nslabs; i += gridDim.x * blockDim.x) slabs[i] = 0.0;
    // cbT[p][d][j] = cb[2p + j][d]; for odd K the missing partner repeats code K-1 (never selected).
    const int npairs = (K + 1) >> 1;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < npairs * D * 2; i += gridDim.x * blockDim.x) {
        const int j = i & 1, d = (i >> 1) % D, p = (i >> 1) / D;
        int k = 2 * p + j;
        if (k >= K) k = K - 1;
        cbT[i] = cb[(long long)k * D + d];
    }
}

// first-minimum with torch.argmax(-dist) NaN semantics: a NaN distance beats any number,
// the first NaN wins (vq_vae.py:68).
__device__ __forceinline__ bool vq_better(float cand, float best)
{
    // best is a number AND (cand is NaN OR cand < best); bitwise so that no branch splits the loop body
    return (best == best) & !(cand >= best);
}

// PP positions per lane (pos, pos + 256, ...): every codebook operand fetched from LDS is used PP times.
// The pair-interleaved codebook is staged in LDS in chunks of CHUNK_PAIRS pairs; all lanes of a wave read the
// same address (broadcast ds_read_b128, conflict free), and because DS reads return in order hipcc can keep the
// next pair's operands in flight behind a counted lgkmcnt while the current pair's packed math runs -- scalar
// (s_load) operands cannot be pipelined that way (SMEM returns out of order: every wait is lgkmcnt(0)), which
// left the first version of this kernel 58 % parked on s_waitcnt.
template <int D, int PP>
__global__ __launch_bounds__(VQ_BLOCK, (D <= 16 ? 4 : (D <= 64 ? 2 : 1))) void vq_forward_kernel(
    const float *__restrict__ z, const float *__restrict__ cb, const float *__restrict__ cbT,
    long long *__restrict__ idx, float *__restrict__ out, double *__restrict__ sse_slabs,
    int *__restrict__ hist, int K, int HW, long long P)
{
    constexpr int CHUNK_PAIRS = 2048 / D;                 // 16 KB of LDS per chunk
    __shared__ __attribute__((aligned(16))) float s_cb[CHUNK_PAIRS * D * 2];
    __shared__ int s_hist[VQ_MAX_LDS_HIST];
    __shared__ double s_red[4];
    const bool lds_hist = K <= VQ_MAX_LDS_HIST;
    if (lds_hist)
        for (int k = threadIdx.x; k < K; k += VQ_BLOCK) s_hist[k] = 0;

    long long pos[PP], base[PP];
    bool active[PP];
    float zr[PP][D];
#pragma unroll
    for (int q = 0; q < PP; ++q) {
        pos[q] = ((long long)blockIdx.x * PP + q) * VQ_BLOCK + threadIdx.x;
        active[q] = pos[q] < P;
        const long long b = active[q] ? pos[q] / HW : 0;
        const long long p = active[q] ? pos[q] - b * HW : 0;
        base[q] = b * (long long)D * HW + p;
#pragma unroll
        for (int d = 0; d < D; ++d) zr[q][d] = active[q] ? z[base[q] + (long long)d * HW] : 0.f;
    }

    float bestd[PP];
    int bi[PP];
#pragma unroll
    for (int q = 0; q < PP; ++q) { bestd[q] = __builtin_inff(); bi[q] = 0; }
    const int npairs = K >> 1;
    for (int c0 = 0; c0 < npairs; c0 += CHUNK_PAIRS) {
        const int cn = min(CHUNK_PAIRS, npairs - c0);
        __syncthreads();                                   // previous chunk consumed (and s_hist zeroed)
        for (int i = threadIdx.x; i < cn * D * 2 / 4; i += VQ_BLOCK)
            reinterpret_cast<f32x4 *>(s_cb)[i] = reinterpret_cast<const f32x4 *>(cbT + (long long)c0 * D * 2)[i];
        __syncthreads();
        for (int kp = 0; kp < cn; ++kp) {
            const float *e = s_cb + kp * (2 * D);          // wave-uniform LDS address: broadcast reads
            f32x2 total[PP];
#pragma unroll
            for (int d0 = 0; d0 < D; d0 += 16) {
                f32x2 acc[PP];
#pragma unroll
                for (int q = 0; q < PP; ++q) acc[q] = (f32x2){0.f, 0.f};
                // the additions of a block are a sequential chain (ATen's order); the subtractions and squares are
                // not, so they are formed eight d at a time ahead of the chain to keep independent work in the pipe
#pragma unroll
                for (int h = d0; h < d0 + 16 && h < D; h += 8) {
                    // (the empty asm statements pin each stage: hipcc otherwise sinks every sub and mul next to its add
                    // and runs sub -> mul -> add per d on two registers, each instruction waiting for the previous one)
                    f32x2 sq[PP][8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const f32x2 e2 = *reinterpret_cast<const f32x2 *>(e + 2 * (h + j));
#pragma unroll
                        for (int q = 0; q < PP; ++q) {
                            const f32x2 zz = {zr[q][h + j], zr[q][h + j]};
                            sq[q][j] = zz - e2;
                        }
                    }
#pragma unroll
                    for (int j = 0; j < 8; ++j)
#pragma unroll
                        for (int q = 0; q < PP; ++q) asm volatile("" : "+v"(sq[q][j]));
#pragma unroll
                    for (int j = 0; j < 8; ++j)
#pragma unroll
                        for (int q = 0; q < PP; ++q) sq[q][j] = sq[q][j] * sq[q][j];
#pragma unroll
                    for (int j = 0; j < 8; ++j)
#pragma unroll
                        for (int q = 0; q < PP; ++q) asm volatile("" : "+v"(sq[q][j]));
#pragma unroll
                    for (int j = 0; j < 8; ++j)
#pragma unroll
                        for (int q = 0; q < PP; ++q)
                            acc[q] = (h == d0 && j == 0) ? sq[q][0] : acc[q] + sq[q][j];   // 0 + s == s for a square (never -0)
                }
#pragma unroll
                for (int q = 0; q < PP; ++q) total[q] = (d0 == 0) ? acc[q] : total[q] + acc[q];
            }
            const int k0 = 2 * (c0 + kp);
#pragma unroll
            for (int q = 0; q < PP; ++q) {
                // bestd starts at +inf with index 0: code 0 wins unless a later code is strictly smaller (or NaN)
                const bool b0 = vq_better(total[q].x, bestd[q]);
                bestd[q] = b0 ? total[q].x : bestd[q]; bi[q] = b0 ? k0 : bi[q];
                const bool b1 = vq_better(total[q].y, bestd[q]);
                bestd[q] = b1 ? total[q].y : bestd[q]; bi[q] = b1 ? k0 + 1 : bi[q];
            }
        }
    }
    if (K & 1) {   // odd tail, scalar
        const int k = K - 1;
        const float *__restrict__ e = cb + (long long)k * D;
#pragma unroll
        for (int q = 0; q < PP; ++q) {
            float total = 0.f;
#pragma unroll
            for (int d0 = 0; d0 < D; d0 += 16) {
                float acc = 0.f;
#pragma unroll
                for (int d = d0; d < d0 + 16 && d < D; ++d) {
                    const float diff = zr[q][d] - e[d];
                    acc = acc + diff * diff;
                }
                total = (d0 == 0) ? acc : total + acc;
            }
            const bool bt = vq_better(total, bestd[q]);
            bestd[q] = bt ? total : bestd[q]; bi[q] = bt ? k : bi[q];
        }
    }

    if (npairs == 0) __syncthreads();                      // s_hist zeroing visible even when the pair loop is empty
    double sse = 0.0;
#pragma unroll
    for (int q = 0; q < PP; ++q) {
        if (active[q]) {
            if (idx) idx[pos[q]] = (long long)bi[q];
            const float *__restrict__ qv = cb + (long long)bi[q] * D;
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const float diff = qv[d] - zr[q][d];
                if (out) out[base[q] + (long long)d * HW] = zr[q][d] + diff;     // z + (q - z), vq_vae.py:71
                const float sq = diff * diff;
                sse += (double)sq;
            }
            if (lds_hist) atomicAdd(&s_hist[bi[q]], 1);
            else atomicAdd(&hist[bi[q]], 1);
        }
    }
    const double tot = block_sum(sse, s_red);
    if (threadIdx.x == 0) sse_slabs[blockIdx.x] = tot;
    if (lds_hist) {
        __syncthreads();
        for (int k = threadIdx.x; k < K; k += VQ_BLOCK) {
            const int c = s_hist[k];
            if (c) atomicAdd(&hist[k], c);
        }
    }
}

__global__ void vq_decode_kernel(const long long *__restrict__ idx, const float *__restrict__ cb,
                                 float *__restrict__ q, int D, int K, int HW, long long P)
{
    const long long pos = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (pos >= P) return;
    long long k = idx[pos];
    if (k < 0) k = 0;
    if (k >= K) k = K - 1;
    const long long b = pos / HW, p = pos - b * HW;
    for (int d = 0; d < D; ++d) q[(b * D + d) * HW + p] = cb[k * D + d];
}

__global__ void vq_finalize_kernel(const double *__restrict__ sse_slabs, int nslabs,
                                   const int *__restrict__ hist, int K, long long P, int D,
                                   float cc, float *__restrict__ scalars)
{
    __shared__ double s_red[4];
    double s = 0.0;
    for (int i = threadIdx.x; i < nslabs; i += blockDim.x) s += sse_slabs[i];
    const double sse = block_sum(s, s_red);
    double e = 0.0;
    for (int k = threadIdx.x; k < K; k += blockDim.x) {
        const float pk = (float)hist[k] / (float)P;
        e += (double)(pk * logf(pk + 1e-10f));
    }
    const double ent = block_sum(e, s_red);
    if (threadIdx.x == 0) {
        const float mse = (float)(sse / ((double)P * (double)D));
        scalars[0] = mse + cc * mse;           // q_latent_loss + commitment_cost * e_latent_loss
        scalars[1] = expf(-(float)ent);
        scalars[2] = mse;
    }
}

// 1024-thread workgroups walking the positions with a grid stride: the codebook gradient is accumulated in LDS
// over ALL of a workgroup's positions and flushed once, so the global float atomics (K*D addresses that every
// workgroup hits) number grid*K*D instead of (P/256)*K*D -- at B = 2048 that flush, not the streaming, was the cost.
// Codebooks larger than the LDS window (512 x 64, 4096 x 16) are split into windows of codes over grid.y.
constexpr int VQ_BWD_BLOCK = 1024;
constexpr int VQ_BWD_LDS = 128 * 1024;     // codebook-gradient window per workgroup (gfx950: 160 KB of LDS per CU)
// grid (x: positions, grid stride; y: windows of Kc codes).  A workgroup only touches the positions whose code falls
// into its window, so z / g_out / dz are still streamed once; idx is read once per window.
template <int D>
__global__ __launch_bounds__(VQ_BWD_BLOCK) void vq_backward_kernel(
    const float *__restrict__ z, const float *__restrict__ cb, const long long *__restrict__ idx,
    const float *__restrict__ g_out, const float *__restrict__ g_loss_dev, float cc,
    float *__restrict__ dz, float *__restrict__ dw, float *__restrict__ dw_slabs, int K, int HW, long long P,
    int Kc)
{
    extern __shared__ float s_dw[];    // [Kc][D]
    const long long k_lo = (long long)blockIdx.y * Kc;
    const int kn = K - k_lo < Kc ? (int)(K - k_lo) : Kc;
    for (int i = threadIdx.x; i < kn * D; i += VQ_BWD_BLOCK) s_dw[i] = 0.f;
    __syncthreads();
    const float g_loss = g_loss_dev ? g_loss_dev[0] : 1.f;
    const double N = (double)P * (double)D;
    const float sz = (float)(2.0 * (double)cc / N) * g_loss;   // d/dz of cc * mse(q.detach(), z)
    const float sw = (float)(2.0 / N) * g_loss;                // d/dq of mse(q, z.detach())
    for (long long pos = (long long)blockIdx.x * VQ_BWD_BLOCK + threadIdx.x; pos < P;
         pos += (long long)gridDim.x * VQ_BWD_BLOCK) {
        const long long b = pos / HW, p = pos - b * HW;
        const long long base = b * (long long)D * HW + p;
        const long long k = idx[pos];
        if (k < k_lo || k >= k_lo + kn) continue;
        const float *__restrict__ q = cb + k * D;
        const int kl = (int)(k - k_lo);
        constexpr int DC = D < 16 ? D : 16;                // d in chunks of 16 (register budget of a 1024-thread group)
#pragma unroll
        for (int d0 = 0; d0 < D; d0 += DC) {
            float zv[DC], gv[DC], qv[DC];
#pragma unroll
            for (int j = 0; j < DC; ++j) {                 // all loads of the chunk first: one round trip, not DC
                const long long o = base + (long long)(d0 + j) * HW;
                zv[j] = z[o]; qv[j] = q[d0 + j];
                gv[j] = g_out ? g_out[o] : 0.f;
            }
#pragma unroll
            for (int j = 0; j < DC; ++j) {
                if (dz) dz[base + (long long)(d0 + j) * HW] = gv[j] + sz * (zv[j] - qv[j]);
                atomicAdd(&s_dw[kl * D + d0 + j], sw * (qv[j] - zv[j]));
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < kn * D; i += VQ_BWD_BLOCK) {
        const float v = s_dw[i];
        if (dw_slabs) dw_slabs[(long long)blockIdx.x * K * D + k_lo * D + i] = v;   // dm_reduce_slabs adds them in slab order
        else if (v != 0.f) atomicAdd(&dw[k_lo * D + i], v);
    }
}

bool vq_dim_supported(int D) { return D == 8 || D == 16 || D == 32 || D == 64 || D == 128; }

}  // namespace

extern "C" size_t dm_vq_workspace_bytes(int K, int D)
{
    return (size_t)((K + 1) / 2) * 2 * (size_t)D * sizeof(float);
}

constexpr int VQ_PP = 2;      // positions per lane in the forward kernel (1 for embedding_dim 64: registers)

extern "C" int dm_vq_num_blocks(int64_t positions)
{
    return (int)((positions + VQ_BLOCK - 1) / VQ_BLOCK);     // upper bound over all variants; unused slabs are zeroed
}

extern "C" int dm_vq_forward(const float *z, const float *codebook, int64_t *idx, float *out,
                             double *sse_slabs, int32_t *hist, int B, int D, int K, int H, int W,
                             void *workspace, size_t workspace_bytes, void *stream)
{
    DM_REQUIRE(z && codebook && sse_slabs && hist, "dm_vq_forward: NULL pointer");
    DM_REQUIRE(B > 0 && H > 0 && W > 0 && K > 0, "dm_vq_forward: bad shape B=%d K=%d H=%d W=%d", B, K, H, W);
    DM_REQUIRE(vq_dim_supported(D), "dm_vq_forward: embedding_dim %d not built (8/16/32/64/128)", D);
    DM_REQUIRE(workspace && workspace_bytes >= dm_vq_workspace_bytes(K, D), "dm_vq_forward: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    const long long P = (long long)B * H * W;
    float *cbT = (float *)workspace;
    const int n = ((K + 1) / 2) * 2 * D;
    const int nslabs = dm_vq_num_blocks(P);
    hipLaunchKernelGGL(vq_prep_kernel, dim3((n + 255) / 256), dim3(256), 0, s, codebook, cbT, K, D, (int *)hist, sse_slabs, nslabs);
    // (vq_prep_kernel cleared hist and all slabs: with PP positions per lane there are fewer workgroups than slabs)
#define DM_VQ_FWD(DD, PP_)                                                                                   \
    hipLaunchKernelGGL((vq_forward_kernel<DD, PP_>), dim3((unsigned)((P + VQ_BLOCK * PP_ - 1) / (VQ_BLOCK * PP_))), \
                       dim3(VQ_BLOCK), 0, s, z, codebook, cbT, (long long *)idx, out, sse_slabs, (int *)hist, K, H * W, P)
    switch (D) {
    case 8: DM_VQ_FWD(8, VQ_PP); break;
    case 16: DM_VQ_FWD(16, VQ_PP); break;
    case 32: DM_VQ_FWD(32, VQ_PP); break;
    case 64: DM_VQ_FWD(64, 1); break;
    default: DM_VQ_FWD(128, 1); break;        // VectorQuantizer's own default embedding_dim (vq_vae.py:35)
    }
#undef DM_VQ_FWD
    return dm_launch_status("dm_vq_forward");
}

extern "C" int dm_vq_decode(const int64_t *idx, const float *codebook, float *q,
                            int B, int D, int K, int H, int W, void *stream)
{
    DM_REQUIRE(idx && codebook && q, "dm_vq_decode: NULL pointer");
    DM_REQUIRE(B > 0 && D > 0 && K > 0 && H > 0 && W > 0, "dm_vq_decode: bad shape");
    const long long P = (long long)B * H * W;
    hipLaunchKernelGGL(vq_decode_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const long long *)idx, codebook, q, D, K, H * W, P);
    return dm_launch_status("dm_vq_decode");
}

extern "C" int dm_vq_finalize(const double *sse_slabs, int nslabs, const int32_t *hist, int K,
                              int64_t positions, int D, float commitment_cost, float *scalars, void *stream)
{
    DM_REQUIRE(sse_slabs && hist && scalars && nslabs > 0 && K > 0 && positions > 0, "dm_vq_finalize: bad argument");
    hipLaunchKernelGGL(vq_finalize_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream,
                       sse_slabs, nslabs, (const int *)hist, K, (long long)positions, D, commitment_cost, scalars);
    return dm_launch_status("dm_vq_finalize");
}

namespace {
int vq_backward_grid(long long P)
{
    const long long want = (P + VQ_BWD_BLOCK - 1) / VQ_BWD_BLOCK;
    return (int)(want < 512 ? want : 512);                 // two 1024-thread workgroups per CU
}

int vq_backward_launch(const char *who, const float *z, const float *codebook, const int64_t *idx, const float *g_out,
                       const float *g_loss_dev, float commitment_cost, float *dz, float *dw, float *dw_slabs,
                       int B, int D, int K, int H, int W, void *stream)
{
    DM_REQUIRE(z && codebook && idx && (dw || dw_slabs), "%s: NULL pointer", who);
    DM_REQUIRE(vq_dim_supported(D), "%s: embedding_dim %d not built (8/16/32/64/128)", who, D);
    const long long P = (long long)B * H * W;
    int Kc = VQ_BWD_LDS / (D * (int)sizeof(float));
    if (Kc > K) Kc = K;
    const size_t lds = (size_t)Kc * D * sizeof(float);
    const int grid = vq_backward_grid(P);
    const dim3 g3((unsigned)grid, (unsigned)((K + Kc - 1) / Kc));
    hipStream_t s = (hipStream_t)stream;
#define DM_VQ_BWD(DD)                                                                                          \
    if (lds > 48 * 1024)                                                                                       \
        (void)hipFuncSetAttribute((const void *)vq_backward_kernel<DD>, hipFuncAttributeMaxDynamicSharedMemorySize,  \
                                  (int)lds);                                                                   \
    hipLaunchKernelGGL(vq_backward_kernel<DD>, g3, dim3(VQ_BWD_BLOCK), lds, s, z, codebook,                    \
                       (const long long *)idx, g_out, g_loss_dev, commitment_cost, dz, dw, dw_slabs, K, H * W, P, Kc)
    switch (D) {
    case 8: DM_VQ_BWD(8); break;
    case 16: DM_VQ_BWD(16); break;
    case 32: DM_VQ_BWD(32); break;
    case 64: DM_VQ_BWD(64); break;
    default: DM_VQ_BWD(128); break;
    }
#undef DM_VQ_BWD
    return dm_launch_status(who);
}
}  // namespace

extern "C" int dm_vq_backward(const float *z, const float *codebook, const int64_t *idx,
                              const float *g_out, const float *g_loss_dev, float commitment_cost,
                              float *dz, float *dw, int B, int D, int K, int H, int W, void *stream)
{
    return vq_backward_launch("dm_vq_backward", z, codebook, idx, g_out, g_loss_dev, commitment_cost, dz, dw, nullptr,
                              B, D, K, H, W, stream);
}

extern "C" int dm_vq_backward_num_slabs(int64_t positions) { return vq_backward_grid(positions); }

extern "C" int dm_vq_backward_slabs(const float *z, const float *codebook, const int64_t *idx,
                                    const float *g_out, const float *g_loss_dev, float commitment_cost,
                                    float *dz, float *dw_slabs, int B, int D, int K, int H, int W, void *stream)
{
    return vq_backward_launch("dm_vq_backward_slabs", z, codebook, idx, g_out, g_loss_dev, commitment_cost, dz, nullptr,
                              dw_slabs, B, D, K, H, W, stream);
}

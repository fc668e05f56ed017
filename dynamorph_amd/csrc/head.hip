// head.hip -- decoder output conv (dec.6, 1x1) fused with the masked reconstruction loss,
// its gradient, and the final loss assembly.
//
// Reference: HiddenStateExtractor/vq_vae.py:298 (nn.Conv2d(num_hiddens//4, num_inputs, 1)),
// :320-323 (recon_loss = mean(mse(decoded*mask, inputs*mask, 'none') / channel_var);
// total = weight_recon*recon + weight_commitment*c_loss).  These are thin, HBM-bound layers
// (Cout = num_inputs): plain VALU, float4 accesses, one thread per 4 consecutive pixels.
#include "dm_common.h"

namespace {

constexpr int HEAD_MAX_BLOCKS = 2048;

template <int C4, int NIN>
__global__ __launch_bounds__(256) void head_forward_kernel(
    const float *__restrict__ d4, const float *__restrict__ w6, const float *__restrict__ b6,
    const float *__restrict__ x, const float *__restrict__ mask, int MC, const float *__restrict__ cvar,
    float *__restrict__ dec, double *__restrict__ loss_slabs, int HW4, long long total4)
{
    __shared__ double s_red[4];
    float w[NIN][C4], bias[NIN], inv_var[NIN];
#pragma unroll
    for (int co = 0; co < NIN; ++co) {
        bias[co] = b6 ? b6[co] : 0.f;
        inv_var[co] = cvar[co];
#pragma unroll
        for (int ci = 0; ci < C4; ++ci) w[co][ci] = w6[co * C4 + ci];
    }
    double loss = 0.0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total4;
         i += (long long)gridDim.x * blockDim.x) {
        const long long b = i / HW4, p = i - b * HW4;
        f32x4 dv[C4];
#pragma unroll
        for (int ci = 0; ci < C4; ++ci) dv[ci] = *reinterpret_cast<const f32x4 *>(d4 + ((b * C4 + ci) * HW4 + p) * 4);
#pragma unroll
        for (int co = 0; co < NIN; ++co) {
            f32x4 o = {bias[co], bias[co], bias[co], bias[co]};
#pragma unroll
            for (int ci = 0; ci < C4; ++ci) o += w[co][ci] * dv[ci];
            const long long off = ((b * NIN + co) * HW4 + p) * 4;
            *reinterpret_cast<f32x4 *>(dec + off) = o;
            if (!x) continue;                       // decoder-only call: no loss
            const f32x4 xv = *reinterpret_cast<const f32x4 *>(x + off);
            f32x4 t;
            if (mask) {
                const f32x4 mv = *reinterpret_cast<const f32x4 *>(mask + ((b * MC + (MC == 1 ? 0 : co)) * HW4 + p) * 4);
                t = o * mv - xv * mv;
            } else {
                t = o - xv;
            }
            const f32x4 sq = t * t;
            const float iv = inv_var[co];
            loss += (double)(sq.x / iv) + (double)(sq.y / iv) + (double)(sq.z / iv) + (double)(sq.w / iv);
        }
    }
    const double tot = block_sum(loss, s_red);
    if (threadIdx.x == 0 && loss_slabs) loss_slabs[blockIdx.x] = tot;
}

template <int C4, int NIN>
__global__ __launch_bounds__(256) void head_backward_kernel(
    const float *__restrict__ dec, const float *__restrict__ x, const float *__restrict__ mask, int MC,
    const float *__restrict__ cvar, const float *__restrict__ d4, const float *__restrict__ w6,
    const float *__restrict__ gscale_dev, const float *__restrict__ gdec_ext, float *__restrict__ g4,
    double *__restrict__ part, int HW4, long long total4, double inv_count)
{
    __shared__ double s_red[4];
    constexpr int NP = NIN * C4 + NIN + C4;
    float w[NIN][C4], cv[NIN];
#pragma unroll
    for (int co = 0; co < NIN; ++co) {
        cv[co] = cvar[co];
#pragma unroll
        for (int ci = 0; ci < C4; ++ci) w[co][ci] = w6[co * C4 + ci];
    }
    const float gs = gscale_dev ? (float)(2.0 * inv_count) * gscale_dev[0] : 0.f;
    float acc[NP];
#pragma unroll
    for (int k = 0; k < NP; ++k) acc[k] = 0.f;

    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total4;
         i += (long long)gridDim.x * blockDim.x) {
        const long long b = i / HW4, p = i - b * HW4;
        f32x4 dv[C4], gd[NIN];
#pragma unroll
        for (int ci = 0; ci < C4; ++ci) dv[ci] = *reinterpret_cast<const f32x4 *>(d4 + ((b * C4 + ci) * HW4 + p) * 4);
#pragma unroll
        for (int co = 0; co < NIN; ++co) {
            const long long off = ((b * NIN + co) * HW4 + p) * 4;
            f32x4 g = {0.f, 0.f, 0.f, 0.f};
            if (gscale_dev) {
                const f32x4 o = *reinterpret_cast<const f32x4 *>(dec + off);
                const f32x4 xv = *reinterpret_cast<const f32x4 *>(x + off);
                if (mask) {
                    const f32x4 mv = *reinterpret_cast<const f32x4 *>(mask + ((b * MC + (MC == 1 ? 0 : co)) * HW4 + p) * 4);
                    g = (o * mv - xv * mv) * mv;
                } else {
                    g = o - xv;
                }
                g = g * (gs / cv[co]);
            }
            if (gdec_ext) g += *reinterpret_cast<const f32x4 *>(gdec_ext + off);
            gd[co] = g;
            acc[NIN * C4 + co] += g.x + g.y + g.z + g.w;
#pragma unroll
            for (int ci = 0; ci < C4; ++ci) {
                const f32x4 pr = g * dv[ci];
                acc[co * C4 + ci] += pr.x + pr.y + pr.z + pr.w;
            }
        }
#pragma unroll
        for (int ci = 0; ci < C4; ++ci) {
            f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int co = 0; co < NIN; ++co) s += w[co][ci] * gd[co];
            s.x = dv[ci].x > 0.f ? s.x : 0.f; s.y = dv[ci].y > 0.f ? s.y : 0.f;
            s.z = dv[ci].z > 0.f ? s.z : 0.f; s.w = dv[ci].w > 0.f ? s.w : 0.f;
            *reinterpret_cast<f32x4 *>(g4 + ((b * C4 + ci) * HW4 + p) * 4) = s;
            acc[NIN * C4 + NIN + ci] += s.x + s.y + s.z + s.w;
        }
    }
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        const double t = block_sum((double)acc[k], s_red);
        if (threadIdx.x == 0) {
            part[((long long)blockIdx.x * NP + k) * 2 + 0] = t;
            part[((long long)blockIdx.x * NP + k) * 2 + 1] = 0.0;
        }
    }
}

// ---- plain masked reconstruction loss on a decoder output (VQ_VAE_z32: vae.py:450-452, the decoder ends in a
//      ConvTranspose2d, there is no 1x1 head to fuse with) ------------------------------------------------------------
// loss_slabs[block] = sum ((dec*m - x*m)^2 / var[c]);  one thread per 4 consecutive pixels of one channel plane
__global__ __launch_bounds__(256) void recon_loss_kernel(
    const float *__restrict__ dec, const float *__restrict__ x, const float *__restrict__ mask, int MC,
    const float *__restrict__ cvar, double *__restrict__ loss_slabs, int NIN, int HW4, long long total4)
{
    __shared__ double s_red[4];
    double loss = 0.0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total4;
         i += (long long)gridDim.x * blockDim.x) {
        const long long plane = i / HW4, p = i - plane * HW4;
        const long long b = plane / NIN;
        const int c = (int)(plane - b * NIN);
        const f32x4 o = *reinterpret_cast<const f32x4 *>(dec + i * 4);
        const f32x4 xv = *reinterpret_cast<const f32x4 *>(x + i * 4);
        f32x4 t = o - xv;
        if (mask) {
            const f32x4 mv = *reinterpret_cast<const f32x4 *>(mask + ((b * MC + (MC == 1 ? 0 : c)) * HW4 + p) * 4);
            t = o * mv - xv * mv;
        }
        const f32x4 sq = t * t;
        const float v = cvar[c];
        loss += (double)(sq.x / v) + (double)(sq.y / v) + (double)(sq.z / v) + (double)(sq.w / v);
    }
    const double tot = block_sum(loss, s_red);
    if (threadIdx.x == 0) loss_slabs[blockIdx.x] = tot;
}

// g_dec = gscale * 2/N * (dec*m - x*m) * m / var[c]; bias_slabs[block][c][2] = (sum g_dec over channel c, 0)
// (blocks walk whole channel planes so that a block's partial sum belongs to one channel at a time)
__global__ __launch_bounds__(256) void recon_loss_backward_kernel(
    const float *__restrict__ dec, const float *__restrict__ x, const float *__restrict__ mask, int MC,
    const float *__restrict__ cvar, const float *__restrict__ gscale_dev, float *__restrict__ gdec,
    double *__restrict__ bias_slabs, int NIN, int HW4, long long nplanes, double inv_count)
{
    __shared__ double s_red[4];
    const float gs = (float)(2.0 * inv_count) * gscale_dev[0];
    for (int c = threadIdx.x; c < NIN; c += blockDim.x) {
        bias_slabs[((long long)blockIdx.x * NIN + c) * 2 + 0] = 0.0;
        bias_slabs[((long long)blockIdx.x * NIN + c) * 2 + 1] = 0.0;
    }
    __syncthreads();
    for (long long plane = blockIdx.x; plane < nplanes; plane += gridDim.x) {
        const long long b = plane / NIN;
        const int c = (int)(plane - b * NIN);
        const float sc = gs / cvar[c];
        double part = 0.0;
        for (int p = threadIdx.x; p < HW4; p += blockDim.x) {
            const long long i = plane * HW4 + p;
            const f32x4 o = *reinterpret_cast<const f32x4 *>(dec + i * 4);
            const f32x4 xv = *reinterpret_cast<const f32x4 *>(x + i * 4);
            f32x4 t = o - xv;
            if (mask) {
                const f32x4 mv = *reinterpret_cast<const f32x4 *>(mask + ((b * MC + (MC == 1 ? 0 : c)) * HW4 + p) * 4);
                t = (o * mv - xv * mv) * mv;
            }
            const f32x4 g = t * sc;
            *reinterpret_cast<f32x4 *>(gdec + i * 4) = g;
            part += (double)((g.x + g.y) + (g.z + g.w));
        }
        const double tot = block_sum(part, s_red);
        if (threadIdx.x == 0) bias_slabs[((long long)blockIdx.x * NIN + c) * 2 + 0] += tot;
        __syncthreads();
    }
}

__global__ void loss_finalize_kernel(const double *__restrict__ loss_slabs, int nslabs, long long count,
                                     const float *__restrict__ vq_scalars, float w_recon, float w_commit,
                                     float *__restrict__ out)
{
    __shared__ double s_red[4];
    double s = 0.0;
    for (int i = threadIdx.x; i < nslabs; i += blockDim.x) s += loss_slabs[i];
    const double tot = block_sum(s, s_red);
    if (threadIdx.x == 0) {
        const float recon = (float)(tot / (double)count);
        const float commit = vq_scalars[0];
        out[0] = recon;
        out[1] = commit;
        out[2] = w_recon * recon + w_commit * commit;
        out[3] = vq_scalars[1];
    }
}

}  // namespace

extern "C" int dm_head_num_blocks(int B, int H, int W)
{
    const long long total4 = (long long)B * H * W / 4;
    const long long g = (total4 + 255) / 256;
    return (int)(g < HEAD_MAX_BLOCKS ? g : HEAD_MAX_BLOCKS);
}

#define DM_HEAD_NIN(KERNEL, C4V, ...)                                                                   \
    switch (NIN) {                                                                                      \
    case 1: hipLaunchKernelGGL((KERNEL<C4V, 1>), dim3(grid), dim3(256), 0, st, __VA_ARGS__); break;     \
    case 2: hipLaunchKernelGGL((KERNEL<C4V, 2>), dim3(grid), dim3(256), 0, st, __VA_ARGS__); break;     \
    case 3: hipLaunchKernelGGL((KERNEL<C4V, 3>), dim3(grid), dim3(256), 0, st, __VA_ARGS__); break;     \
    default: hipLaunchKernelGGL((KERNEL<C4V, 4>), dim3(grid), dim3(256), 0, st, __VA_ARGS__); break;    \
    }
#define DM_HEAD_DISPATCH(KERNEL, ...)                                                                   \
    switch (C4) {                                                                                       \
    case 4: DM_HEAD_NIN(KERNEL, 4, __VA_ARGS__) break;                                                  \
    case 8: DM_HEAD_NIN(KERNEL, 8, __VA_ARGS__) break;                                                  \
    default: DM_HEAD_NIN(KERNEL, 16, __VA_ARGS__) break;                                                \
    }

// num_hiddens 16/32/64 x num_inputs 1..4 are instantiated; anything else goes through the generic 1x1 convolution
// (dm_conv3x3 taps=1) + dm_recon_loss(_backward) on the host side (engine.py)
extern "C" int dm_head_supported(int C4, int NIN)
{
    return (C4 == 4 || C4 == 8 || C4 == 16) && NIN >= 1 && NIN <= 4;
}

extern "C" int dm_head_forward(const float *d4, const float *w6, const float *b6, const float *x, const float *mask,
                               int mask_channels, const float *channel_var, float *decoded, double *loss_slabs,
                               int B, int C4, int NIN, int H, int W, void *stream)
{
    DM_REQUIRE(d4 && w6 && channel_var && decoded, "dm_head_forward: NULL pointer");
    DM_REQUIRE(!x || loss_slabs, "dm_head_forward: loss_slabs required when x is given");
    DM_REQUIRE(C4 == 4 || C4 == 8 || C4 == 16, "dm_head_forward: num_hiddens//4 = %d not built (4, 8, 16)", C4);
    DM_REQUIRE(NIN >= 1 && NIN <= 4, "dm_head_forward: num_inputs %d not built (1..4)", NIN);
    DM_REQUIRE((H * W) % 4 == 0 && B > 0, "dm_head_forward: bad shape");
    DM_REQUIRE(!mask || mask_channels == 1 || mask_channels == NIN, "dm_head_forward: mask channels %d", mask_channels);
    const int HW4 = H * W / 4;
    const long long total4 = (long long)B * HW4;
    const int grid = dm_head_num_blocks(B, H, W);
    hipStream_t st = (hipStream_t)stream;
    DM_HEAD_DISPATCH(head_forward_kernel, d4, w6, b6, x, mask, mask_channels, channel_var, decoded, loss_slabs, HW4, total4)
    return dm_launch_status("dm_head_forward");
}

extern "C" int dm_head_backward(const float *decoded, const float *x, const float *mask, int mask_channels,
                                const float *channel_var, const float *d4, const float *w6, const float *gscale_dev,
                                const float *gdec_ext, float *g4, double *part_slabs, int B, int C4, int NIN, int H,
                                int W, void *stream)
{
    DM_REQUIRE(channel_var && d4 && w6 && g4 && part_slabs, "dm_head_backward: NULL pointer");
    DM_REQUIRE(gscale_dev || gdec_ext, "dm_head_backward: need gscale_dev and/or gdec_ext");
    DM_REQUIRE(!gscale_dev || (decoded && x), "dm_head_backward: loss gradient needs decoded and x");
    DM_REQUIRE(C4 == 4 || C4 == 8 || C4 == 16, "dm_head_backward: num_hiddens//4 = %d not built (4, 8, 16)", C4);
    DM_REQUIRE(NIN >= 1 && NIN <= 4, "dm_head_backward: num_inputs %d not built (1..4)", NIN);
    DM_REQUIRE((H * W) % 4 == 0 && B > 0, "dm_head_backward: bad shape");
    DM_REQUIRE(!mask || mask_channels == 1 || mask_channels == NIN, "dm_head_backward: mask channels %d", mask_channels);
    const int HW4 = H * W / 4;
    const long long total4 = (long long)B * HW4;
    const int grid = dm_head_num_blocks(B, H, W);
    const double inv_count = 1.0 / ((double)B * NIN * H * W);
    hipStream_t st = (hipStream_t)stream;
    DM_HEAD_DISPATCH(head_backward_kernel, decoded, x, mask, mask_channels, channel_var, d4, w6, gscale_dev, gdec_ext,
                     g4, part_slabs, HW4, total4, inv_count)
    return dm_launch_status("dm_head_backward");
}

extern "C" int dm_loss_finalize(const double *loss_slabs, int nslabs, int64_t count, const float *vq_scalars,
                                float weight_recon, float weight_commitment, float *scalars_out, void *stream)
{
    DM_REQUIRE(loss_slabs && vq_scalars && scalars_out && nslabs > 0 && count > 0, "dm_loss_finalize: bad argument");
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, loss_slabs, nslabs,
                       (long long)count, vq_scalars, weight_recon, weight_commitment, scalars_out);
    return dm_launch_status("dm_loss_finalize");
}

extern "C" int dm_recon_loss_num_blocks(int B, int NIN, int H, int W)
{
    (void)H; (void)W;
    const long long planes = (long long)B * NIN;
    return (int)(planes < 1024 ? planes : 1024);
}

extern "C" int dm_recon_loss(const float *decoded, const float *x, const float *mask, int mask_channels,
                             const float *channel_var, double *loss_slabs, int B, int NIN, int H, int W, void *stream)
{
    DM_REQUIRE(decoded && x && channel_var && loss_slabs && B > 0 && NIN > 0, "dm_recon_loss: bad argument");
    DM_REQUIRE((H * W) % 4 == 0, "dm_recon_loss: H*W must be a multiple of 4");
    DM_REQUIRE(!mask || mask_channels == 1 || mask_channels == NIN, "dm_recon_loss: mask channels %d", mask_channels);
    const int HW4 = H * W / 4;
    const long long total4 = (long long)B * NIN * HW4;
    hipLaunchKernelGGL(recon_loss_kernel, dim3(dm_recon_loss_num_blocks(B, NIN, H, W)), dim3(256), 0, (hipStream_t)stream,
                       decoded, x, mask, mask_channels, channel_var, loss_slabs, NIN, HW4, total4);
    return dm_launch_status("dm_recon_loss");
}

extern "C" int dm_recon_loss_backward(const float *decoded, const float *x, const float *mask, int mask_channels,
                                      const float *channel_var, const float *gscale_dev, float *g_decoded,
                                      double *bias_slabs, int B, int NIN, int H, int W, void *stream)
{
    DM_REQUIRE(decoded && x && channel_var && gscale_dev && g_decoded && bias_slabs && B > 0 && NIN > 0,
               "dm_recon_loss_backward: bad argument");
    DM_REQUIRE((H * W) % 4 == 0, "dm_recon_loss_backward: H*W must be a multiple of 4");
    DM_REQUIRE(!mask || mask_channels == 1 || mask_channels == NIN, "dm_recon_loss_backward: mask channels %d", mask_channels);
    const int HW4 = H * W / 4;
    const double inv_count = 1.0 / ((double)B * NIN * H * W);
    hipLaunchKernelGGL(recon_loss_backward_kernel, dim3(dm_recon_loss_num_blocks(B, NIN, H, W)), dim3(256), 0,
                       (hipStream_t)stream, decoded, x, mask, mask_channels, channel_var, gscale_dev, g_decoded, bias_slabs,
                       NIN, HW4, (long long)B * NIN, inv_count);
    return dm_launch_status("dm_recon_loss_backward");
}

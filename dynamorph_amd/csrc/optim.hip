// optim.hip -- fused Adam, the enc.0 o enc.1 weight composition with its chain rule,
// and the on-device flip/rot90 augmentation.
//
// Reference: run_training.py:485 (t.optim.Adam(model.parameters(), lr, betas=(.9,.999)),
// eps 1e-8, no weight decay), run_training.py:396-403 (per-sample flip + rot90),
// HiddenStateExtractor/vq_vae.py:277-278 (enc.0 1x1 conv feeding enc.1 4x4/s2 conv).
#include "dm_common.h"

namespace {

// One launch over the flat parameter buffer (all 43 trainable tensors are views of it).
// Same operation order as torch.optim.Adam's default path:
//   m.lerp_(g, 1-b1); v.mul_(b2).addcmul_(g, g, 1-b2);
//   denom = sqrt(v)/sqrt(1-b2^t) + eps; p.addcdiv_(m, denom, value=-lr/(1-b1^t))
__global__ __launch_bounds__(256) void adam_kernel(float *__restrict__ p, const float *__restrict__ g,
                                                   float *__restrict__ m, float *__restrict__ v, long long n,
                                                   float lr, float b1, float b2, float eps,
                                                   const float *__restrict__ step_dev, float *__restrict__ step_next,
                                                   float grad_scale)
{
    // grad_scale: the bucket holds the SUM over ranks of the data-parallel exchange, 1 / world turns it into the mean here
    // (the same single float product the separate "x 1/world" launch did; 1 in one process: g * 1.f == g bit for bit)
    // step_next given: step_dev holds the number of COMPLETED steps, this is step t = that + 1, and t is written to
    // step_next (a different word: the caller ping-pongs two counters, so no launch is spent on "step += 1")
    const double t = (double)step_dev[0] + (step_next ? 1.0 : 0.0);
    if (step_next && blockIdx.x == 0 && threadIdx.x == 0) step_next[0] = (float)t;
    const double bc1 = 1.0 - pow((double)b1, t);
    const double bc2 = 1.0 - pow((double)b2, t);
    const float step_size = (float)((double)lr / bc1);
    const float bc2_sqrt = (float)sqrt(bc2);
    const float w1 = 1.f - b1, w2 = 1.f - b2;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float gi = g[i] * grad_scale;
        const float mi = m[i] + w1 * (gi - m[i]);
        const float vi = v[i] * b2 + w2 * (gi * gi);
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        p[i] = p[i] - step_size * (mi / denom);
    }
}

// Weff[c1][ci][t], ci in [0, NIN] (ci == NIN is the ones channel), t = ky*4+kx.
__global__ void e1_compose_kernel(const float *__restrict__ w0, const float *__restrict__ b0,
                                  const float *__restrict__ w1, const float *__restrict__ b1, float *__restrict__ weff,
                                  float *__restrict__ bias_border, int NIN, int C0, int C1)
{
    // bias_border[ry][rx][c1] = b1[c1] + sum over the taps inside the image of the ones-channel weights
    // (row class 0 = first output row: ky = 0 falls into the padding; class 2 = last row: ky = 3 does)
    if (bias_border) {
        // 16 lanes per table entry, one per tap: its ones-channel weight (if the tap stays inside the image for that
        // row / column class), then a 16-lane sum
        const int gt = blockIdx.x * blockDim.x + threadIdx.x, t = gt & 15;
        const int ky = t >> 2, kx = t & 3;
        for (int i = gt >> 4; i < ((9 * C1 + 15) & ~15); i += (gridDim.x * blockDim.x) >> 4) {
            const bool live = i < 9 * C1;
            const int c1 = live ? i % C1 : 0, rx = live ? (i / C1) % 3 : 1, ry = live ? i / (3 * C1) : 1;
            const bool inside = !(ry == 0 && ky == 0) && !(ry == 2 && ky == 3) && !(rx == 0 && kx == 0) && !(rx == 2 && kx == 3);
            double s = 0.0;
            if (live && inside)
                for (int c = 0; c < C0; ++c) s += (double)w1[(c1 * C0 + c) * 16 + t] * (double)b0[c];
            s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64); s += __shfl_xor(s, 8, 64);
            if (live && t == 0) bias_border[i] = (float)(s + (b1 ? (double)b1[c1] : 0.0));
        }
    }
    const int total = C1 * (NIN + 1) * 16;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int t = i & 15, ci = (i >> 4) % (NIN + 1), c1 = (i >> 4) / (NIN + 1);
        double s = 0.0;
        for (int c = 0; c < C0; ++c) {
            const double a = (double)w1[(c1 * C0 + c) * 16 + t];
            s += a * (ci < NIN ? (double)w0[c * NIN + ci] : (double)b0[c]);
        }
        weff[i] = (float)s;
    }
}

__global__ void e1_chain_kernel(const float *__restrict__ dweff, const float *__restrict__ w0,
                                const float *__restrict__ b0, const float *__restrict__ w1,
                                float *__restrict__ dw0, float *__restrict__ db0, float *__restrict__ dw1,
                                int NIN, int C0, int C1)
{
    const int n_w1 = C1 * C0 * 16, n_w0 = C0 * NIN, total = n_w1 + n_w0 + C0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        if (i < n_w1) {
            const int t = i & 15, c = (i >> 4) % C0, c1 = (i >> 4) / C0;
            double s = (double)dweff[(c1 * (NIN + 1) + NIN) * 16 + t] * (double)b0[c];
            for (int ci = 0; ci < NIN; ++ci)
                s += (double)dweff[(c1 * (NIN + 1) + ci) * 16 + t] * (double)w0[c * NIN + ci];
            dw1[i] = (float)s;
        } else {
            const int j = i - n_w1;
            const bool is_bias = j >= n_w0;
            const int c = is_bias ? j - n_w0 : j / NIN, ci = is_bias ? NIN : j % NIN;
            double s = 0.0;
            for (int c1 = 0; c1 < C1; ++c1)
                for (int t = 0; t < 16; ++t)
                    s += (double)w1[(c1 * C0 + c) * 16 + t] * (double)dweff[(c1 * (NIN + 1) + ci) * 16 + t];
            if (is_bias) db0[c] = (float)s;
            else dw0[j] = (float)s;
        }
    }
}

// out[b] = rot90(flip(in[b], flip), k) on (C, H, H) patches, gather form.
__global__ __launch_bounds__(256) void augment_kernel(const float *__restrict__ in, float *__restrict__ out,
                                                      const int *__restrict__ flip_code, const int *__restrict__ rot_code,
                                                      int C, int H, long long total)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(i % H), y = (int)((i / H) % H);
        const long long plane = i / ((long long)H * H);
        const int b = (int)(plane / C);
        const int k = rot_code[b] & 3, f = flip_code[b];
        int sy, sx;
        switch (k) {                      // torch.rot90(img, k, dims=[1, 2])
        case 0: sy = y; sx = x; break;
        case 1: sy = x; sx = H - 1 - y; break;
        case 2: sy = H - 1 - y; sx = H - 1 - x; break;
        default: sy = H - 1 - x; sx = y; break;
        }
        if (f == 1) sy = H - 1 - sy;      // torch.flip(img, dims=(1,))
        else if (f == 2) sx = H - 1 - sx; // torch.flip(img, dims=(2,))
        out[i] = in[plane * H * H + (long long)sy * H + sx];
    }
}

}  // namespace

// Per-patch, per-channel z-score (pipeline/train_utils.py:252-274 zscore_patch, applied at patch_VAE.py:413-419 on a
// float64 array that is then cast to float32): out = float((x - mean) / (std + eps)) with the population std over
// H x W, all in double.  One workgroup per (patch, channel) plane; the plane is read twice (mean, then variance about
// the mean -- numpy's np.std is the same two-pass form), the second and third read come from L2.
template <typename T>
__global__ __launch_bounds__(256) void zscore_patch_kernel(const T *__restrict__ in, float *__restrict__ out, int HW)
{
    __shared__ double s_red[4];
    __shared__ double s_bc[2];
    const T *p = in + (long long)blockIdx.x * HW;
    float *q = out + (long long)blockIdx.x * HW;
    double s = 0.0;
    for (int i = threadIdx.x; i < HW; i += blockDim.x) s += (double)p[i];
    const double tot = block_sum(s, s_red);
    if (threadIdx.x == 0) s_bc[0] = tot / (double)HW;
    __syncthreads();
    const double mean = s_bc[0];
    double v = 0.0;
    for (int i = threadIdx.x; i < HW; i += blockDim.x) {
        const double d = (double)p[i] - mean;
        v += d * d;
    }
    const double var = block_sum(v, s_red);
    if (threadIdx.x == 0) s_bc[1] = sqrt(var / (double)HW) + 2.220446049250313e-16;
    __syncthreads();
    const double denom = s_bc[1];
    for (int i = threadIdx.x; i < HW; i += blockDim.x) q[i] = (float)(((double)p[i] - mean) / denom);
}

// Dataset-wide per-channel z-score with GIVEN statistics (pipeline/train_utils.py:228-250 zscore as run_training.py:880 calls it,
// followed by .astype(np.float32)): out = float((x - mean[c]) / denom[c]), denom = std + eps formed by the caller the way numpy
// forms it.  TD / TQ: the types numpy's promotion rules give the difference and the quotient -- double throughout for a float64
// dataset; for a float32 dataset and Python-float statistics the difference stays float32 and the quotient, divided by
// `std + np.finfo(float).eps` (a float64 scalar), is float64 under NumPy 2 and float32 under NumPy 1: the caller asks numpy.
// 16 bytes of output per lane; an element's channel from its plane index.
template <typename TIN, typename TD, typename TQ>
__global__ __launch_bounds__(256) void zscore_channels_kernel(const TIN *__restrict__ in, float *__restrict__ out,
                                                              const double *__restrict__ mean, const double *__restrict__ denom,
                                                              long long total4, int C, long long HW4)
{
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total4; i += (long long)gridDim.x * 256) {
        const int c = (int)((i / HW4) % C);
        const TD m = (TD)mean[c];
        const TQ d = (TQ)denom[c];
        const TIN *p = in + 4 * i;
        // (IEEE operations, by name: no compiler setting turns these into a reciprocal and a multiplication)
        auto q = [m, d](TIN v) {
            TD diff;
            if constexpr (sizeof(TD) == 8) diff = __dsub_rn((double)v, (double)m); else diff = __fsub_rn((float)v, (float)m);
            if constexpr (sizeof(TQ) == 8) return (float)__ddiv_rn((double)diff, (double)d);
            else return __fdiv_rn((float)diff, (float)d);
        };
        f32x4 o;
        o.x = q(p[0]); o.y = q(p[1]); o.z = q(p[2]); o.w = q(p[3]);
        *reinterpret_cast<f32x4 *>(out + 4 * i) = o;
    }
}

extern "C" int dm_zscore_channels(const void *in, int in_is_f64, int diff_f64, int quot_f64, float *out, const double *mean,
                                  const double *denom, int64_t N, int C, int64_t HW, void *stream)
{
    DM_REQUIRE(in && out && mean && denom && N > 0 && C > 0 && HW > 0, "dm_zscore_channels: bad argument");
    DM_REQUIRE(HW % 4 == 0, "dm_zscore_channels: H * W must be a multiple of 4");
    DM_REQUIRE((diff_f64 || !in_is_f64) && (quot_f64 || !diff_f64), "dm_zscore_channels: types only widen (input <= difference <= quotient)");
    const long long total4 = (long long)N * C * (HW / 4), HW4 = HW / 4;
    const int grid = (int)((total4 + 255) / 256 < 8192 ? (total4 + 255) / 256 : 8192);
    hipStream_t s = (hipStream_t)stream;
#define DM_ZS(TIN, TD, TQ) hipLaunchKernelGGL((zscore_channels_kernel<TIN, TD, TQ>), dim3(grid), dim3(256), 0, s, (const TIN *)in, out, \
                                              mean, denom, total4, C, HW4)
    if (in_is_f64) DM_ZS(double, double, double);
    else if (diff_f64) DM_ZS(float, double, double);
    else if (quot_f64) DM_ZS(float, float, double);
    else DM_ZS(float, float, float);
#undef DM_ZS
    return dm_launch_status("dm_zscore_channels");
}

extern "C" int dm_adam(float *param, const float *grad, float *m, float *v, int64_t n,
                       float lr, float beta1, float beta2, float eps, const float *step_dev, void *stream)
{
    DM_REQUIRE(param && grad && m && v && step_dev && n > 0, "dm_adam: bad argument");
    const int grid = (int)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024);
    hipLaunchKernelGGL(adam_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, param, grad, m, v, (long long)n, lr,
                       beta1, beta2, eps, step_dev, (float *)nullptr, 1.f);
    return dm_launch_status("dm_adam");
}

extern "C" int dm_adam_counted(float *param, const float *grad, float *m, float *v, int64_t n, float lr, float beta1,
                               float beta2, float eps, const float *steps_done, float *steps_done_next, void *stream)
{
    DM_REQUIRE(param && grad && m && v && steps_done && steps_done_next && steps_done != steps_done_next && n > 0,
               "dm_adam_counted: bad argument");
    const int grid = (int)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024);
    hipLaunchKernelGGL(adam_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, param, grad, m, v, (long long)n, lr,
                       beta1, beta2, eps, steps_done, steps_done_next, 1.f);
    return dm_launch_status("dm_adam_counted");
}

extern "C" int dm_adam_counted_scaled(float *param, const float *grad, float *m, float *v, int64_t n, float lr, float beta1,
                                      float beta2, float eps, float grad_scale, const float *steps_done,
                                      float *steps_done_next, void *stream)
{
    DM_REQUIRE(param && grad && m && v && steps_done && steps_done_next && steps_done != steps_done_next && n > 0,
               "dm_adam_counted_scaled: bad argument");
    const int grid = (int)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024);
    hipLaunchKernelGGL(adam_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, param, grad, m, v, (long long)n, lr,
                       beta1, beta2, eps, steps_done, steps_done_next, grad_scale);
    return dm_launch_status("dm_adam_counted_scaled");
}

extern "C" int dm_e1_compose(const float *w0, const float *b0, const float *w1, float *weff,
                             int NIN, int C0, int C1, void *stream)
{
    DM_REQUIRE(w0 && b0 && w1 && weff && NIN > 0 && C0 > 0 && C1 > 0, "dm_e1_compose: bad argument");
    hipLaunchKernelGGL(e1_compose_kernel, dim3(2), dim3(256), 0, (hipStream_t)stream, w0, b0, w1, (const float *)nullptr,
                       weff, (float *)nullptr, NIN, C0, C1);
    return dm_launch_status("dm_e1_compose");
}

extern "C" int dm_e1_compose_border(const float *w0, const float *b0, const float *w1, const float *b1, float *weff,
                                    float *bias_border, int NIN, int C0, int C1, void *stream)
{
    DM_REQUIRE(w0 && b0 && w1 && weff && bias_border && NIN > 0 && C0 > 0 && C1 > 0, "dm_e1_compose_border: bad argument");
    hipLaunchKernelGGL(e1_compose_kernel, dim3(8), dim3(256), 0, (hipStream_t)stream, w0, b0, w1, b1, weff, bias_border,
                       NIN, C0, C1);
    return dm_launch_status("dm_e1_compose_border");
}

extern "C" int dm_e1_chain(const float *dweff, const float *w0, const float *b0, const float *w1,
                           float *dw0, float *db0, float *dw1, int NIN, int C0, int C1, void *stream)
{
    DM_REQUIRE(dweff && w0 && b0 && w1 && dw0 && db0 && dw1, "dm_e1_chain: NULL pointer");
    hipLaunchKernelGGL(e1_chain_kernel, dim3(8), dim3(256), 0, (hipStream_t)stream, dweff, w0, b0, w1, dw0, db0, dw1,
                       NIN, C0, C1);
    return dm_launch_status("dm_e1_chain");
}

extern "C" int dm_augment(const float *in, float *out, const int32_t *flip_code, const int32_t *rot_code,
                          int B, int C, int H, void *stream)
{
    DM_REQUIRE(in && out && flip_code && rot_code && B > 0 && C > 0 && H > 0, "dm_augment: bad argument");
    DM_REQUIRE(in != out, "dm_augment: in-place not supported");
    const long long total = (long long)B * C * H * H;
    const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(augment_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, in, out, (const int *)flip_code,
                       (const int *)rot_code, C, H, total);
    return dm_launch_status("dm_augment");
}

extern "C" int dm_zscore_patch(const void *in, int in_is_f64, float *out, int planes, int HW, void *stream)
{
    DM_REQUIRE(in && out && planes > 0 && HW > 0, "dm_zscore_patch: bad argument");
    if (in_is_f64)
        hipLaunchKernelGGL(zscore_patch_kernel<double>, dim3(planes), dim3(256), 0, (hipStream_t)stream, (const double *)in, out, HW);
    else
        hipLaunchKernelGGL(zscore_patch_kernel<float>, dim3(planes), dim3(256), 0, (hipStream_t)stream, (const float *)in, out, HW);
    return dm_launch_status("dm_zscore_patch");
}

// bn.hip -- training-mode BatchNorm2d finalisation (forward statistics -> affine
// coefficients, backward reductions -> AFFINE2 coefficients), the elementwise
// residual/affine apply, generic slab reductions, and the library's error plumbing.
//
// Reference: nn.BatchNorm2d as instantiated at HiddenStateExtractor/vq_vae.py:206,209,
// 279,282,285,288 with PyTorch defaults (eps 1e-5, momentum 0.1, affine, running stats);
// the path never calls .eval(), so batch statistics are always used.
// The heavy per-element work (normalise, ReLU, BN backward formula) is NOT done here:
// it is folded into the loads of the consuming convolution kernels via dm_operand.
#include "dm_common.h"
#include <string.h>

// ------------------------------------------------------------------ error plumbing
static thread_local char g_err[512] = "";

void dm_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char *dm_last_error(void) { return g_err; }
extern "C" int dm_version(void) { return DM_VERSION; }

namespace {

// This thread's share of the (sum, sum of products) slabs of channel c: slabs tid, tid + 256, ... -- four 16-byte loads in
// flight per round (one round for up to 1024 slabs) instead of one dependent load pair per slab.
__device__ __forceinline__ void slab_partial(const double *__restrict__ stats, int nslabs, int C, int c, double &s1, double &s2)
{
    typedef double f64x2 __attribute__((ext_vector_type(2)));
    const f64x2 *__restrict__ st2 = reinterpret_cast<const f64x2 *>(stats);
    s1 = 0.0; s2 = 0.0;
    for (int i0 = threadIdx.x; i0 < nslabs; i0 += 4 * (int)blockDim.x) {
        f64x2 v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = i0 + j * (int)blockDim.x;
            v[j] = (f64x2){0.0, 0.0};
            if (i < nslabs) v[j] = st2[(long long)i * C + c];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) { s1 += v[j].x; s2 += v[j].y; }
    }
}

// Batch mode: one block per channel, all slabs form one group.
__global__ __launch_bounds__(256) void bn_finalize_kernel(
    const double *__restrict__ stats, int nslabs, int C, long long count,
    const float *__restrict__ gamma, const float *__restrict__ beta,
    float *__restrict__ running_mean, float *__restrict__ running_var, long long *__restrict__ nbt,
    float momentum, float eps, float *__restrict__ coef, float *__restrict__ saved)
{
    __shared__ double s_red[4];
    const int c = blockIdx.x;
    const float g = gamma ? gamma[c] : 1.f, bt = beta ? beta[c] : 0.f;
    const double n = (double)count;
    const double unbias = count > 1 ? n / (n - 1.0) : 1.0;
    // (the old running statistics are requested with the slabs: they are only needed after the two block sums)
    float rm0 = 0.f, rv0 = 0.f;
    if (threadIdx.x == 0) { if (running_mean) rm0 = running_mean[c]; if (running_var) rv0 = running_var[c]; }
    {
        double s1, s2;
        slab_partial(stats, nslabs, C, c, s1, s2);
        const double t1 = block_sum(s1, s_red);
        const double t2 = block_sum(s2, s_red);
        if (threadIdx.x == 0) {
            const double mean = t1 / n;
            double var = t2 / n - mean * mean;
            if (var < 0.0) var = 0.0;
            const float mean_f = (float)mean;
            const float invstd = (float)(1.0 / sqrt(var + (double)eps));
            const float scale = g * invstd;
            coef[c * 4 + 0] = scale;
            coef[c * 4 + 1] = 0.f;
            coef[c * 4 + 2] = bt - mean_f * scale;
            coef[c * 4 + 3] = 0.f;
            saved[c * 2 + 0] = mean_f;
            saved[c * 2 + 1] = invstd;
            // ATen's CPU kernel evaluates these two updates in double (acc type) before the fp32 store
            const double mom = (double)momentum;
            if (running_mean) running_mean[c] = (float)(mom * mean + (1.0 - mom) * (double)rm0);
            if (running_var) running_var[c] = (float)(mom * (var * unbias) + (1.0 - mom) * (double)rv0);
            if (nbt && c == 0) nbt[0] += 1;
        }
        return;
    }
}

// base^e for a small non-negative integer exponent: ~2 log2(e) multiplications (pow() in double costs an order more)
__device__ __forceinline__ double int_pow(double base, int e)
{
    double r = 1.0;
    while (e > 0) {
        if (e & 1) r *= base;
        base *= base;
        e >>= 1;
    }
    return r;
}

// Running statistics of channel c after B successive batch-of-one calls (one workgroup; see the closed form below).
__device__ __forceinline__ void running_replay_channel(const double *__restrict__ stats, int B, int spg, int C, long long count, int c,
                                                       float *__restrict__ running_mean, float *__restrict__ running_var,
                                                       long long *__restrict__ nbt, float momentum, double *s_red)
{
    const double n = (double)count;
    const double unbias = count > 1 ? n / (n - 1.0) : 1.0;
    const bool track = running_mean && running_var;
    const double m = (double)momentum, keep = 1.0 - m;
    double sm = 0.0, sv = 0.0;
    // (the old values are requested now, next to the slabs: they are only needed after the two block sums)
    float rm0 = 0.f, rv0 = 0.f;
    if (track && threadIdx.x == 0) { rm0 = running_mean[c]; rv0 = running_var[c]; }
    const double kB = int_pow(keep, B);
    if (track) {
        // four samples per thread and eight slabs per sample requested together (one memory round trip for B <= 1024,
        // <= 8 slabs per sample); each sample's slabs are still added in slab order
        typedef double f64x2 __attribute__((ext_vector_type(2)));
        const f64x2 *__restrict__ st2 = reinterpret_cast<const f64x2 *>(stats);
        for (int b0 = threadIdx.x; b0 < B; b0 += 4 * (int)blockDim.x) {
            double s1[4] = {0.0, 0.0, 0.0, 0.0}, s2[4] = {0.0, 0.0, 0.0, 0.0};
            for (int i0 = 0; i0 < spg; i0 += 8) {
                f64x2 v[4][8];
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const int b = b0 + j * (int)blockDim.x, i = i0 + k;
                        v[j][k] = (f64x2){0.0, 0.0};
                        if (b < B && i < spg) v[j][k] = st2[(long long)(b * spg + i) * C + c];
                    }
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int k = 0; k < 8; ++k) { s1[j] += v[j][k].x; s2[j] += v[j][k].y; }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int b = b0 + j * (int)blockDim.x;
                if (b >= B) continue;
                const double mean = s1[j] / n;
                double var = s2[j] / n - mean * mean;
                if (var < 0.0) var = 0.0;
                const double w = m * int_pow(keep, B - 1 - b);
                sm += w * (double)(float)mean;                   // the fp32 values a batch-of-one call would feed
                sv += w * (double)(float)(var * unbias);
            }
        }
    }
    const double tm = block_sum(sm, s_red);
    const double tv = block_sum(sv, s_red);
    if (threadIdx.x == 0) {
        if (track) {
            running_mean[c] = (float)(kB * (double)rm0 + tm);
            running_var[c] = (float)(kB * (double)rv0 + tv);
        }
        if (nbt && c == 0) nbt[0] += B;
    }
}

// The same replay for several BatchNorm layers in ONE launch (dm_bn_running_replay): the per-sample inference path
// (process_VAE) needs the coefficients of a layer before the next convolution, but its running statistics only as
// state -- they are taken off the critical path and brought up to date together at the end of the encoder.
struct ReplaySegs {
    const double *stats[16];
    float *running_mean[16], *running_var[16];
    long long *nbt[16];
    long long count[16];
    int B[16], spg[16], C[16], first_block[17];
    float momentum[16];
    int nseg;
};
__global__ __launch_bounds__(256) void bn_running_replay_kernel(ReplaySegs rs)
{
    __shared__ double s_red[4];
    const int k = blockIdx.y;                      // grid (widest layer, layers): no search through the segment table
    if ((int)blockIdx.x >= rs.C[k]) return;
    running_replay_channel(rs.stats[k], rs.B[k], rs.spg[k], rs.C[k], rs.count[k], blockIdx.x,
                           rs.running_mean[k], rs.running_var[k], rs.nbt[k], rs.momentum[k], s_red);
}

// Per-sample mode (process_VAE: every patch is its own batch), ONE launch:
//   blocks [0, nb): one thread per (sample, channel) sums that sample's slabs_per_group slabs and writes its coefficients;
//   blocks [nb, nb + C): channel c = blockIdx - nb replays the running statistics of B successive batch-of-one calls.
//     The recurrence r <- m*x_b + (1-m)*r unrolls to r = (1-m)^B r0 + sum_b m (1-m)^(B-1-b) x_b: every thread takes
//     samples b, b+256, ... (statistics recomputed from the slabs, weight evaluated in double), one block sum per
//     channel.  (The sequential fp32 form took 17 us at B = 1024; the closed form differs from it by fp32 rounding.)
__global__ __launch_bounds__(256) void bn_finalize_per_sample_kernel(
    const double *__restrict__ stats, int B, int spg, int C, long long count, const float *__restrict__ gamma,
    const float *__restrict__ beta, float eps, float *__restrict__ coef, float *__restrict__ saved, int nb,
    float *__restrict__ running_mean, float *__restrict__ running_var, long long *__restrict__ nbt, float momentum)
{
    const double n = (double)count;
    if ((int)blockIdx.x >= nb) {
        __shared__ double s_red[4];
        running_replay_channel(stats, B, spg, C, count, blockIdx.x - nb, running_mean, running_var, nbt, momentum, s_red);
        return;
    }
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * C) return;
    const int b = idx / C, c = idx - b * C;
    const float g = gamma ? gamma[c] : 1.f, bt = beta ? beta[c] : 0.f;
    double s1 = 0.0, s2 = 0.0;
    for (int i = 0; i < spg; ++i) {
        s1 += stats[((long long)(b * spg + i) * C + c) * 2 + 0];
        s2 += stats[((long long)(b * spg + i) * C + c) * 2 + 1];
    }
    const double mean = s1 / n;
    double var = s2 / n - mean * mean;
    if (var < 0.0) var = 0.0;
    const float mean_f = (float)mean;
    const float invstd = (float)(1.0 / sqrt(var + (double)eps));
    const float scale = g * invstd;
    *reinterpret_cast<f32x4 *>(coef + (long long)idx * 4) = (f32x4){scale, 0.f, bt - mean_f * scale, 0.f};
    saved[(long long)idx * 2 + 0] = mean_f;
    saved[(long long)idx * 2 + 1] = invstd;
}

// slabs hold (sum dy, sum dy*a).  x_hat = (a - mean)*invstd, so
//   sum dy*x_hat = invstd * (sum dy*a - mean * sum dy).
// da = scale*(dy - c1 - x_hat*c2) with c1 = mean(dy), c2 = mean(dy*x_hat)  ==>
// da = A*dy + Bc*a + Cc,  A = scale, Bc = -scale*invstd*c2, Cc = -scale*c1 - Bc*mean.
__global__ __launch_bounds__(256) void bn_backward_finalize_kernel(
    const double *__restrict__ stats, int nslabs, int C, long long count,
    const float *__restrict__ gamma, const float *__restrict__ saved,
    float *__restrict__ dgamma, float *__restrict__ dbeta, float *__restrict__ coef_bwd)
{
    __shared__ double s_red[4];
    const int c = blockIdx.x;
    // (requested with the slabs, used after the block sums)
    float mean_f = 0.f, invstd_f = 0.f, g_f = 1.f;
    if (threadIdx.x == 0) { mean_f = saved[c * 2 + 0]; invstd_f = saved[c * 2 + 1]; if (gamma) g_f = gamma[c]; }
    double s1, s2;
    slab_partial(stats, nslabs, C, c, s1, s2);
    const double sum_dy = block_sum(s1, s_red);
    const double sum_dya = block_sum(s2, s_red);
    if (threadIdx.x == 0) {
        // count == 0: fixed statistics (eval() mode: `saved` holds the running mean and 1 / sqrt(running_var + eps)) -- the
        // batch-mean terms of the gradient vanish, da = scale * dy
        const double inv_n = count > 0 ? 1.0 / (double)count : 0.0;
        const double mean = (double)mean_f, invstd = (double)invstd_f;
        const double g = (double)g_f;
        const double sum_dyxh = invstd * (sum_dya - mean * sum_dy);
        if (dgamma) dgamma[c] = (float)sum_dyxh;
        if (dbeta) dbeta[c] = (float)sum_dy;
        const double scale = g * invstd;
        const double c1 = sum_dy * inv_n, c2 = sum_dyxh * inv_n;
        const double Bc = -scale * invstd * c2;
        coef_bwd[c * 4 + 0] = (float)scale;
        coef_bwd[c * 4 + 1] = (float)Bc;
        coef_bwd[c * 4 + 2] = (float)(-scale * c1 - Bc * mean);
        coef_bwd[c * 4 + 3] = 0.f;
    }
}

__global__ __launch_bounds__(256) void apply_kernel(Operand in, const float *__restrict__ resid,
                                                    float *__restrict__ out, int C, int HW4, long long total4)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total4;
         i += (long long)gridDim.x * blockDim.x) {
        const long long plane = i / HW4;          // b*C + c
        const int c = (int)(plane % C), b = (int)(plane / C);
        f32x4 v = operand_load4(in, i * 4, b, c);
        if (resid) v += *reinterpret_cast<const f32x4 *>(resid + i * 4);
        *reinterpret_cast<f32x4 *>(out + i * 4) = v;
    }
}

// One block per (chunk of samples, channel): stats[chunk][c] = (sum p, sum p*q).
__global__ __launch_bounds__(256) void channel_stats_kernel(const float *__restrict__ p, const float *__restrict__ q,
                                                            double *__restrict__ stats, int B, int C, int HW4, int bchunk)
{
    __shared__ double s_red[4];
    const int c = blockIdx.x % C, chunk = blockIdx.x / C;
    const int b0 = chunk * bchunk, b1 = min(B, b0 + bchunk);
    double s1 = 0.0, s2 = 0.0;
    // the chunk's (sample, position) pairs spread over all threads (a 16 x 16 latent has 64 float4 per plane: a loop over
    // the plane alone would leave three waves idle); four elements in fp32, then one promotion, as in the conv epilogues
    const int total = (b1 - b0) * HW4;
    for (int e = threadIdx.x; e < total; e += blockDim.x) {
        const int bb = e / HW4, i = e - bb * HW4;
        const long long off = ((((long long)(b0 + bb)) * C + c) * HW4 + i) * 4;
        const f32x4 a = *reinterpret_cast<const f32x4 *>(p + off);
        const f32x4 w = q ? *reinterpret_cast<const f32x4 *>(q + off) : a;
        s1 += (double)((a.x + a.y) + (a.z + a.w));
        s2 += (double)((a.x * w.x + a.y * w.y) + (a.z * w.z + a.w * w.w));
    }
    const double t1 = block_sum(s1, s_red);
    const double t2 = block_sum(s2, s_red);
    if (threadIdx.x == 0) {
        stats[((long long)chunk * C + c) * 2 + 0] = t1;
        stats[((long long)chunk * C + c) * 2 + 1] = t2;
    }
}

__global__ __launch_bounds__(256) void sum_slabs_kernel(const double *__restrict__ stats, int nslabs, int N,
                                                        float scale, float *__restrict__ dst)
{
    __shared__ double s_red[4];
    const int n = blockIdx.x;
    double s = 0.0;
    for (int i = threadIdx.x; i < nslabs; i += blockDim.x) s += stats[((long long)i * N + n) * 2];
    const double t = block_sum(s, s_red);
    if (threadIdx.x == 0) dst[n] = (float)(t * (double)scale);
}

struct Scatter {
    int nseg;
    int end[8];
    float *dst[8];
};

__global__ __launch_bounds__(256) void sum_slabs_scatter_kernel(const double *__restrict__ stats, int nslabs, int N,
                                                                float scale, Scatter sc)
{
    __shared__ double s_red[4];
    const int n = blockIdx.x;
    double s = 0.0;
    for (int i = threadIdx.x; i < nslabs; i += blockDim.x) s += stats[((long long)i * N + n) * 2];
    const double t = block_sum(s, s_red);
    if (threadIdx.x == 0) {
        int start = 0;
        for (int k = 0; k < sc.nseg; ++k) {
            if (n < sc.end[k]) { sc.dst[k][n - start] = (float)(t * (double)scale); break; }
            start = sc.end[k];
        }
    }
}

constexpr int STATS_BCHUNK = 32;

}  // namespace

extern "C" int dm_bn_finalize(const double *stats, int nslabs, int slabs_per_group, int C, int64_t count_per_group,
                              const float *gamma, const float *beta, float *running_mean, float *running_var,
                              int64_t *num_batches_tracked, float momentum, float eps,
                              float *coef, float *saved, int per_sample, void *stream)
{
    DM_REQUIRE(stats && coef && saved, "dm_bn_finalize: NULL pointer");
    DM_REQUIRE(nslabs > 0 && C > 0 && count_per_group > 0, "dm_bn_finalize: bad sizes");
    DM_REQUIRE(!per_sample || (slabs_per_group > 0 && nslabs % slabs_per_group == 0),
               "dm_bn_finalize: nslabs %d not a multiple of slabs_per_group %d", nslabs, slabs_per_group);
    if (per_sample) {
        const int B = nslabs / slabs_per_group;
        DM_REQUIRE((long long)B * C < (1LL << 30), "dm_bn_finalize: too many (sample, channel) pairs");
        const int nb = (B * C + 255) / 256;
        // (no running statistics and no counter given: the caller replays them later, dm_bn_running_replay)
        const int rb = (running_mean && running_var) || num_batches_tracked ? C : 0;
        hipLaunchKernelGGL(bn_finalize_per_sample_kernel, dim3(nb + rb), dim3(256), 0, (hipStream_t)stream, stats, B,
                           slabs_per_group, C, (long long)count_per_group, gamma, beta, eps, coef, saved, nb, running_mean,
                           running_var, (long long *)num_batches_tracked, momentum);
        return dm_launch_status("dm_bn_finalize");
    }
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(C), dim3(256), 0, (hipStream_t)stream, stats, nslabs, C,
                       (long long)count_per_group, gamma, beta, running_mean, running_var,
                       (long long *)num_batches_tracked, momentum, eps, coef, saved);
    return dm_launch_status("dm_bn_finalize");
}

extern "C" int dm_bn_running_replay(const dm_bn_replay_seg *segs, int nseg, void *stream)
{
    DM_REQUIRE(segs && nseg >= 1 && nseg <= 16, "dm_bn_running_replay: 1..16 segments");
    ReplaySegs rs;
    rs.nseg = nseg;
    int blocks = 0;
    for (int k = 0; k < 16; ++k) {
        const bool on = k < nseg;
        if (on)
            DM_REQUIRE(segs[k].stats && segs[k].nslabs > 0 && segs[k].slabs_per_group > 0 && segs[k].C > 0 &&
                           segs[k].count_per_group > 0 && segs[k].nslabs % segs[k].slabs_per_group == 0,
                       "dm_bn_running_replay: bad segment %d", k);
        rs.stats[k] = on ? segs[k].stats : nullptr;
        rs.running_mean[k] = on ? segs[k].running_mean : nullptr; rs.running_var[k] = on ? segs[k].running_var : nullptr;
        rs.nbt[k] = on ? (long long *)segs[k].num_batches_tracked : nullptr;
        rs.count[k] = on ? (long long)segs[k].count_per_group : 1;
        rs.B[k] = on ? segs[k].nslabs / segs[k].slabs_per_group : 0; rs.spg[k] = on ? segs[k].slabs_per_group : 1;
        rs.C[k] = on ? segs[k].C : 0; rs.momentum[k] = on ? segs[k].momentum : 0.f;
        rs.first_block[k] = blocks;
        if (on) blocks += segs[k].C;
    }
    rs.first_block[16] = blocks;
    int widest = 1;
    for (int k = 0; k < nseg; ++k) widest = segs[k].C > widest ? segs[k].C : widest;
    hipLaunchKernelGGL(bn_running_replay_kernel, dim3(widest, nseg), dim3(256), 0, (hipStream_t)stream, rs);
    return dm_launch_status("dm_bn_running_replay");
}

extern "C" int dm_bn_backward_finalize(const double *stats, int nslabs, int C, int64_t count,
                                       const float *gamma, const float *saved, float *dgamma, float *dbeta,
                                       float *coef_bwd, void *stream)
{
    DM_REQUIRE(stats && saved && coef_bwd, "dm_bn_backward_finalize: NULL pointer");
    DM_REQUIRE(nslabs > 0 && C > 0 && count >= 0, "dm_bn_backward_finalize: bad sizes");
    hipLaunchKernelGGL(bn_backward_finalize_kernel, dim3(C), dim3(256), 0, (hipStream_t)stream, stats, nslabs, C,
                       (long long)count, gamma, saved, dgamma, dbeta, coef_bwd);
    return dm_launch_status("dm_bn_backward_finalize");
}

extern "C" int dm_apply(const dm_operand *in, const float *resid, float *out, int B, int C, int H, int W, void *stream)
{
    if (dm_check_operand(in, "dm_apply")) return -1;
    DM_REQUIRE(out && B > 0 && C > 0 && H > 0 && W > 0, "dm_apply: bad argument");
    DM_REQUIRE((H * W) % 4 == 0, "dm_apply: H*W must be a multiple of 4");
    DM_REQUIRE(!in->ones_channel, "dm_apply: ones_channel not supported");
    const int HW4 = H * W / 4;
    const long long total4 = (long long)B * C * HW4;
    const int grid = (int)((total4 + 255) / 256 < 2048 ? (total4 + 255) / 256 : 2048);
    hipLaunchKernelGGL(apply_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, to_dev(in), resid, out, C, HW4, total4);
    return dm_launch_status("dm_apply");
}

extern "C" int dm_channel_stats_num_blocks(int B, int C, int H, int W)
{
    (void)C; (void)H; (void)W;
    return (B + STATS_BCHUNK - 1) / STATS_BCHUNK;
}

extern "C" int dm_channel_stats(const float *p, const float *q, double *stats, int B, int C, int H, int W, void *stream)
{
    DM_REQUIRE(p && stats && B > 0 && C > 0, "dm_channel_stats: bad argument");
    DM_REQUIRE((H * W) % 4 == 0, "dm_channel_stats: H*W must be a multiple of 4");
    const int chunks = dm_channel_stats_num_blocks(B, C, H, W);
    hipLaunchKernelGGL(channel_stats_kernel, dim3(chunks * C), dim3(256), 0, (hipStream_t)stream, p, q, stats, B, C,
                       H * W / 4, STATS_BCHUNK);
    return dm_launch_status("dm_channel_stats");
}

extern "C" int dm_sum_slabs_scatter(const double *stats, int nslabs, int N, float scale, const dm_scatter *sc, void *stream)
{
    DM_REQUIRE(stats && sc && nslabs > 0 && N > 0, "dm_sum_slabs_scatter: bad argument");
    DM_REQUIRE(sc->nseg >= 1 && sc->nseg <= 8 && sc->end[sc->nseg - 1] == N, "dm_sum_slabs_scatter: segments must cover N");
    Scatter d;
    d.nseg = sc->nseg;
    for (int k = 0; k < 8; ++k) { d.end[k] = k < sc->nseg ? sc->end[k] : N; d.dst[k] = k < sc->nseg ? sc->dst[k] : nullptr; }
    for (int k = 0; k < sc->nseg; ++k) DM_REQUIRE(d.dst[k], "dm_sum_slabs_scatter: NULL destination %d", k);
    hipLaunchKernelGGL(sum_slabs_scatter_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, stats, nslabs, N, scale, d);
    return dm_launch_status("dm_sum_slabs_scatter");
}

extern "C" int dm_sum_slabs(const double *stats, int nslabs, int N, float scale, float *dst, void *stream)
{
    DM_REQUIRE(stats && dst && nslabs > 0 && N > 0, "dm_sum_slabs: bad argument");
    hipLaunchKernelGGL(sum_slabs_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, stats, nslabs, N, scale, dst);
    return dm_launch_status("dm_sum_slabs");
}

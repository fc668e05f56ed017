// conv1x1_bwd.hip -- backward of a 1x1 convolution that feeds a train-mode BatchNorm, data AND weight gradient in one pass.
//
// Reference: the second convolution of a ResidualBlock layer, Conv2d(num_residual_hiddens -> num_hiddens, 1)
// (HiddenStateExtractor/vq_vae.py:207-209), as autograd differentiates it for total_loss.backward()
// (run_training.py:406): aten::convolution_backward (input and weight) with BatchNorm's backward in front of it and the
// ReLU mask / BatchNorm-backward reductions of the layer below behind it.
//
// As two kernels (rounds 1-3: conv3x3_kernel<16, 2, 1, 1, ..> + wgrad_kernel<16, 32, 1, ..>) the pair read the output
// gradient (two tensors: BatchNorm's backward is folded into the load) and the layer input twice: 335 MB for 201 MB of
// tensors per 2048 patches, and both kernels sit at the HBM roof (6.2 / 5.0 TB/s, 32 + 27 us).  Here one staging of
//     da = c0*dy + c1*y + c2        (CD channels: BatchNorm backward of the conv's output y, dm_operand AFFINE2)
//     x                             (CX channels, RAW: the layer input before its BatchNorm + ReLU)
// per tile of 256 positions feeds both products on v_mfma_f32_16x16x4_f32:
//     data gradient    dx[ci][p] = (t[ci][p] > 0) * sum_co W[co][ci] * da[co][p],   t = c0x*x + c2x   (M = positions, K = co)
//     weight gradient  dW[co][ci] += sum_p da[co][p] * relu(t[ci][p])                                  (M = co, K = positions)
// plus the (sum dx, sum dx*x) slabs BatchNorm's backward of the layer below needs.  201 MB moved once.
#include "dm_common.h"

// 64 -> 64 channels (the wide residual blocks): wide_stream.hip
bool dm_stream_conv1x1_bwd_shape(int CD, int CX, int H, int W);
int dm_stream_conv1x1_bwd_slabs(int B, int H, int W);
int dm_stream_conv1x1_bwd(const Operand &dy, const float *x, const float *xcoef, const float *w, float *dx, double *stats,
                          float *wslabs, int B, int H, int W, hipStream_t st);

namespace {

constexpr int C1_TP = 256;                  // positions per tile (one 16 x 16 latent)
constexpr int C1_PS = C1_TP + 4;            // LDS plane stride: == 4 (mod 32) dwords
constexpr int C1_MAX_GRID = 512;            // two workgroups per CU

template <int CD, int CX>
__global__ __launch_bounds__(DM_BLOCK, 2)
void conv1x1_bwd_kernel(Operand dy, const float *__restrict__ x, const float *__restrict__ xcoef,
                        const float *__restrict__ w, float *__restrict__ dx, double *__restrict__ stats,
                        float *__restrict__ wslabs, int HW, int ntiles)
{
    static_assert(CD == 16 && CX % 16 == 0 && CX <= 64, "one M tile of output-gradient channels, 1..4 N tiles of input channels");
    constexpr int NT = CX / 16;
    constexpr int ED = CD * (C1_TP / 4) / DM_BLOCK, EX = CX * (C1_TP / 4) / DM_BLOCK;      // float4 per thread and tile
    __shared__ __attribute__((aligned(16))) float sD[CD * C1_PS];
    __shared__ __attribute__((aligned(16))) float sX[CX * C1_PS];
    __shared__ double s_stat[4][CX][2];

    const int lane = threadIdx.x & 63, m = lane & 15, kq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tps = HW / C1_TP;                                   // tiles per sample
    const bool two = dy.p1 != nullptr;

    // weights of the data gradient: B operand W[co = 4 ks + kq][ci = 16 nt + m]
    float wreg[4][NT];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) wreg[ks][nt] = w[(4 * ks + kq) * CX + 16 * nt + m];
    // BatchNorm + ReLU of the layer input, for this lane's channels ci = 16 nt + m
    float xc0[NT], xc2[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) { xc0[nt] = xcoef[(16 * nt + m) * 4]; xc2[nt] = xcoef[(16 * nt + m) * 4 + 2]; }
    // BatchNorm backward of the output gradient: element e of a thread belongs to channel 4 e + wave (wave-uniform)
    float dc0[ED], dc1[ED], dc2[ED];
#pragma unroll
    for (int e = 0; e < ED; ++e) {
        const int c = 4 * e + wave;
        dc0[e] = dy.coef ? dy.coef[c * 4] : 1.f;
        dc1[e] = (dy.coef && two) ? dy.coef[c * 4 + 1] : 0.f;
        dc2[e] = dy.coef ? dy.coef[c * 4 + 2] : 0.f;
    }

    f32x4 wacc[NT];
    double s1[NT], s2[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) { wacc[nt] = (f32x4){0.f, 0.f, 0.f, 0.f}; s1[nt] = 0.0; s2[nt] = 0.0; }

    // register staging of the next tile: element e of thread tid = float4 (e * 256 + tid) of the [channel][64] tile image
    f32x4 rv[ED], ru[ED], rx[EX];
    auto issue = [&](int t) {
        const int b = t / tps, p0 = (t - b * tps) * C1_TP;
        const long long dbase = (long long)b * CD * HW + p0, xbase = (long long)b * CX * HW + p0;
        const int q = (threadIdx.x & 63) * 4;
#pragma unroll
        for (int e = 0; e < ED; ++e) {
            const long long off = dbase + (long long)(4 * e + wave) * HW + q;
            rv[e] = *reinterpret_cast<const f32x4 *>(dy.p0 + off);
            if (two) ru[e] = *reinterpret_cast<const f32x4 *>(dy.p1 + off);
        }
#pragma unroll
        for (int e = 0; e < EX; ++e)
            rx[e] = *reinterpret_cast<const f32x4 *>(x + xbase + (long long)(4 * e + wave) * HW + q);
    };
    int tile = blockIdx.x;
    if (tile < ntiles) issue(tile);

    while (tile < ntiles) {
        __syncthreads();                                         // the previous tile has been consumed
        {
            const int q = (threadIdx.x & 63) * 4;
#pragma unroll
            for (int e = 0; e < ED; ++e) {
                f32x4 v = dc0[e] * rv[e] + dc2[e];               // (the operand transform of tile.h: two fused multiply-adds)
                if (two) v += dc1[e] * ru[e];
                *reinterpret_cast<f32x4 *>(sD + (4 * e + wave) * C1_PS + q) = v;
            }
#pragma unroll
            for (int e = 0; e < EX; ++e) *reinterpret_cast<f32x4 *>(sX + (4 * e + wave) * C1_PS + q) = rx[e];
        }
        __syncthreads();
        const int cur = tile;
        tile += gridDim.x;
        if (tile < ntiles) issue(tile);                          // in flight during the products below

        const int b = cur / tps, p0 = (cur - b * tps) * C1_TP;
        const int wbase = 64 * wave;                             // this wave's 64 positions of the tile
        // ---- weight gradient: K = this wave's positions, four per step
        {
            const float *pa = sD + m * C1_PS + wbase + kq;
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const float a = pa[4 * s];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const float xv = sX[(16 * nt + m) * C1_PS + wbase + 4 * s + kq];
                    const float t = dm_relu(xc0[nt] * xv + xc2[nt]);
                    wacc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, t, wacc[nt], 0, 0, 0);
                }
            }
        }
        // ---- data gradient: four M tiles of 16 positions, K = the CD output-gradient channels
        float *__restrict__ dxs = dx + (long long)b * CX * HW + p0;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            f32x4 acc[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
            const float *pa = sD + kq * C1_PS + wbase + 16 * mt + m;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const float a = pa[4 * ks * C1_PS];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, wreg[ks][nt], acc[nt], 0, 0, 0);
            }
            // lane (m, kq) holds positions 4 kq .. 4 kq + 3 of the M tile for channel ci = 16 nt + m
            const int pl = wbase + 16 * mt + 4 * kq;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const f32x4 xr = *reinterpret_cast<const f32x4 *>(sX + (16 * nt + m) * C1_PS + pl);
                const f32x4 t = xc0[nt] * xr + xc2[nt];
                f32x4 v = acc[nt];
                v.x = t.x > 0.f ? v.x : 0.f; v.y = t.y > 0.f ? v.y : 0.f;
                v.z = t.z > 0.f ? v.z : 0.f; v.w = t.w > 0.f ? v.w : 0.f;
                *reinterpret_cast<f32x4 *>(dxs + (long long)(16 * nt + m) * HW + pl) = v;
                // four elements in fp32, then one promotion (as the convolution epilogues do)
                s1[nt] += (double)((v.x + v.y) + (v.z + v.w));
                s2[nt] += (double)((v.x * xr.x + v.y * xr.y) + (v.z * xr.z + v.w * xr.w));
            }
        }
    }

    // ---- statistics slab of this workgroup: lanes of one channel (the four kq groups), then the waves in wave order
    __syncthreads();
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        double a = s1[nt], c = s2[nt];
        a += __shfl_xor(a, 16, 64); c += __shfl_xor(c, 16, 64);
        a += __shfl_xor(a, 32, 64); c += __shfl_xor(c, 32, 64);
        if (lane < 16) { s_stat[wave][16 * nt + lane][0] = a; s_stat[wave][16 * nt + lane][1] = c; }
    }
    // ---- weight-gradient slab: the four waves' accumulators through LDS in wave order (sD is free now)
    float *red = sD;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) *reinterpret_cast<f32x4 *>(red + ((wave * NT + nt) * 64 + lane) * 4) = wacc[nt];
    __syncthreads();
    if (threadIdx.x < CX) {
        double ta = 0.0, tc = 0.0;
#pragma unroll
        for (int wv = 0; wv < 4; ++wv) { ta += s_stat[wv][threadIdx.x][0]; tc += s_stat[wv][threadIdx.x][1]; }
        stats[((long long)blockIdx.x * CX + threadIdx.x) * 2 + 0] = ta;
        stats[((long long)blockIdx.x * CX + threadIdx.x) * 2 + 1] = tc;
    }
    // element i of the slab = dW[co][ci], co = i / CX: accumulator row 4 kq + r of lane (m, kq) in N tile nt
    for (int i = threadIdx.x; i < CD * CX; i += DM_BLOCK) {
        const int co = i / CX, ci = i - co * CX;
        const int nt = ci >> 4, ln = (co >> 2) * 16 + (ci & 15), r = co & 3;
        float sum = 0.f;
#pragma unroll
        for (int wv = 0; wv < 4; ++wv) sum += red[((wv * NT + nt) * 64 + ln) * 4 + r];
        wslabs[(long long)blockIdx.x * CD * CX + i] = sum;
    }
}

bool conv1x1_bwd_shape(int CD, int CX, int H, int W)
{
    return CD == 16 && CX == 32 && H > 0 && W > 0 && (H * W) % C1_TP == 0;
}

}  // namespace

extern "C" int dm_conv1x1_bwd_fused_supported(int CD, int CX, int H, int W)
{
    return (conv1x1_bwd_shape(CD, CX, H, W) || dm_stream_conv1x1_bwd_shape(CD, CX, H, W)) ? 1 : 0;
}

extern "C" int dm_conv1x1_bwd_fused_num_blocks(int B, int CD, int CX, int H, int W)
{
    if (B > 0 && dm_stream_conv1x1_bwd_shape(CD, CX, H, W)) return dm_stream_conv1x1_bwd_slabs(B, H, W);
    if (B <= 0 || !conv1x1_bwd_shape(CD, CX, H, W)) return -1;
    const long long ntiles = (long long)B * (H * W / C1_TP);
    return (int)(ntiles < C1_MAX_GRID ? ntiles : C1_MAX_GRID);
}

extern "C" int dm_conv1x1_bwd_fused(const dm_operand *dy, const float *x, const float *xcoef, const float *w, float *dx,
                                    double *stats, float *wslabs, int B, int CD, int CX, int H, int W, void *stream)
{
    DM_REQUIRE(dy && dy->p0 && x && xcoef && w && dx && stats && wslabs, "dm_conv1x1_bwd_fused: NULL pointer");
    DM_REQUIRE(B > 0 && (conv1x1_bwd_shape(CD, CX, H, W) || dm_stream_conv1x1_bwd_shape(CD, CX, H, W)),
               "dm_conv1x1_bwd_fused: shape %d -> %d channels on %dx%d not built", CX, CD, H, W);
    DM_REQUIRE(dy->mode == DM_LOAD_IDENT || dy->mode == DM_LOAD_AFFINE2, "dm_conv1x1_bwd_fused: dy operand must be IDENT or AFFINE2");
    DM_REQUIRE(dy->mode == DM_LOAD_IDENT || dy->coef, "dm_conv1x1_bwd_fused: AFFINE2 needs coefficients");
    DM_REQUIRE(dy->coef_bstride == 0 && !dy->ones_channel, "dm_conv1x1_bwd_fused: shared coefficients only");
    DM_REQUIRE((long long)B * CX * H * W < (1LL << 31), "dm_conv1x1_bwd_fused: tensor too large");
    Operand d = to_dev(dy);
    if (d.mode == DM_LOAD_IDENT) { d.coef = nullptr; d.p1 = nullptr; }
    if (dm_stream_conv1x1_bwd_shape(CD, CX, H, W)) {
        dm_stream_conv1x1_bwd(d, x, xcoef, w, dx, stats, wslabs, B, H, W, (hipStream_t)stream);
        return dm_launch_status("dm_conv1x1_bwd_fused");
    }
    const int grid = dm_conv1x1_bwd_fused_num_blocks(B, CD, CX, H, W);
    const int ntiles = B * (H * W / C1_TP);
    hipLaunchKernelGGL((conv1x1_bwd_kernel<16, 32>), dim3(grid), dim3(DM_BLOCK), 0, (hipStream_t)stream, d, x, xcoef, w, dx, stats,
                       wslabs, H * W, ntiles);
    return dm_launch_status("dm_conv1x1_bwd_fused");
}

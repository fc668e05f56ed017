// conv4x4s2_patch.hip -- Conv2d(16 -> 16, 4, stride 2, padding 1) from a 32 x 32 to a 16 x 16 grid with ONE PATCH per workgroup
// pass: the backward kernel (data AND weight gradient from one staging) and, below it, the forward kernel.
//
// Reference: enc.7 of VQ_VAE.enc (HiddenStateExtractor/vq_vae.py:284) as autograd differentiates it for
// total_loss.backward() (run_training.py:406): aten::convolution_backward (input and weight) with BatchNorm's backward
// folded into the output gradient's load, the ReLU mask of enc.6 and the reductions of enc.5's BatchNorm backward in the
// epilogue.  Rounds 1-3: convT_phase_kernel<16, 16, ..> (79.8 us) + wgrad_kernel<16, 16, 4, ..> (58.4 us) at B = 2048, each staging
// the output gradient (two tensors) and the layer input on its own.
//
// One workgroup of 512 threads per CU keeps a patch in LDS,
//     da  [16][18][24]    = c0*dy + c1*y + c2 inside the 16 x 16 image, 0 in the padding   (dm_operand AFFINE2)
//     T   [16][34][40]    = relu(c0x*x + c2x) inside the 32 x 32 image, 0 in the padding    (what the forward multiplied)
// and runs both products on v_mfma_f32_16x16x4_f32:
//     weight gradient  dW[co][ci][ky][kx] = sum_{y,x} da[co][y][x] * T[ci][2y+ky-1][2x+kx-1]
//                      M = co, N tile t = input channel ci with its 16 taps as columns, K = positions, four consecutive x per step
//     data gradient    the transposed convolution by output phase: row u = 2Y + pu, column v = 2X + pv take the taps
//                      ky = 2a + 1 - pu at y = Y + pu - a, kx = 2b + 1 - pv at x = X + pv - b (a, b in {0, 1}):
//                      M = the 16 X of a phase row, N = ci, K = (a, b, co): 16 steps per phase; the two column phases of a
//                      lane interleave in registers into 8 consecutive output columns
// plus the (sum dx, sum dx * x) slabs of the BatchNorm below.
#include "dm_common.h"

namespace {

constexpr int S2_RS = 24, S2_PS = 18 * 24 + 20;             // da image: 452 == 4 (mod 32) dwords per plane
constexpr int S2_RST = 40, S2_PST = 34 * 40 + 20;           // T image: 1380 == 4 (mod 32)
constexpr int S2_NTH = 512, S2_NW = 8;
constexpr int S2_LDS_FLOATS = 16 * S2_PS + 16 * S2_PST + 64 * 64;      // da + T + the transposed weights in operand order
constexpr size_t S2_LDS_BYTES = (size_t)S2_LDS_FLOATS * sizeof(float);

__global__ __launch_bounds__(S2_NTH, 1)
void conv4x4s2_bwd_kernel(Operand dy, const float *__restrict__ x, const float *__restrict__ xcoef, const float *__restrict__ w,
                          float *__restrict__ dx, double *__restrict__ stats, float *__restrict__ wslabs, int ntiles)
{
    constexpr int C = 16, NW = S2_NW, PS = S2_PS, RS = S2_RS, PST = S2_PST, RST = S2_RST;
    extern __shared__ __attribute__((aligned(16))) float lds4[];
    float *sD = lds4, *sT = lds4 + C * PS, *sW = sT + C * PST;
    __shared__ double s_stat[NW][C][2];
    __shared__ float s_tc[C][2];

    const int lane = threadIdx.x & 63, m = lane & 15, kq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool two = dy.p1 != nullptr;

    // the padding never changes: zero both images once (the commits write the interior only)
    for (int i = threadIdx.x; i < (C * PS + C * PST) / 4; i += S2_NTH) reinterpret_cast<f32x4 *>(lds4)[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // data-gradient weights in operand order: step (phase = 2 pu + pv, a, b, channel group cg): B[k = kq][n = ci = m] =
    // W[co = 4 cg + kq][ci][2a + 1 - pu][2b + 1 - pv]
    for (int s = wave; s < 64; s += NW) {
        const int ph = s >> 4, pu = ph >> 1, pv = ph & 1, a = (s >> 3) & 1, b = (s >> 2) & 1, cg = s & 3;
        sW[s * 64 + lane] = w[((4 * cg + kq) * C + m) * 16 + (2 * a + 1 - pu) * 4 + (2 * b + 1 - pv)];
    }
    if (threadIdx.x < C) { s_tc[threadIdx.x][0] = xcoef[threadIdx.x * 4]; s_tc[threadIdx.x][1] = xcoef[threadIdx.x * 4 + 2]; }
    // BatchNorm backward of the output gradient for the planes this wave stages (2 e + ..: planes e * NW + wave)
    float dc0[2], dc1[2], dc2[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int c = e * NW + wave;
        dc0[e] = dy.coef ? dy.coef[c * 4] : 1.f;
        dc1[e] = (dy.coef && two) ? dy.coef[c * 4 + 1] : 0.f;
        dc2[e] = dy.coef ? dy.coef[c * 4 + 2] : 0.f;
    }
    f32x4 wacc[C];
#pragma unroll
    for (int t = 0; t < C; ++t) wacc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    double s1 = 0.0, s2 = 0.0;

    // staging.  da: float4 `lane` of plane e * NW + wave (row lane >> 2, columns 4 (lane & 3) ..).  T: float4 `lane` of quarter
    // plane qp = e * NW + wave (channel qp >> 2, rows 8 (qp & 3) + (lane >> 3), columns 4 (lane & 7) ..)
    const int dq = (lane >> 2) * 16 + 4 * (lane & 3), dl = ((lane >> 2) + 1) * RS + 4 + 4 * (lane & 3);
    const int tq = (lane >> 3) * 32 + 4 * (lane & 7), tl = ((lane >> 3) + 1) * RST + 4 + 4 * (lane & 7);
    f32x4 rv[2], ru[2], rx[8];
    auto issue = [&](int b) {
        const long long db = (long long)b * C * 256, xb = (long long)b * C * 1024;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const long long off = db + (long long)(e * NW + wave) * 256 + dq;
            rv[e] = *reinterpret_cast<const f32x4 *>(dy.p0 + off);
            if (two) ru[e] = *reinterpret_cast<const f32x4 *>(dy.p1 + off);
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int qp = e * NW + wave;
            rx[e] = *reinterpret_cast<const f32x4 *>(x + xb + (long long)(qp >> 2) * 1024 + (qp & 3) * 256 + tq);
        }
    };
    int tile = blockIdx.x;
    if (tile < ntiles) issue(tile);
    __syncthreads();                                             // zero fill, weights and coefficient table complete

    // weight-gradient B column of this lane inside N tile t = ci: tap (ky, kx) = (m >> 2, m & 3):
    // T[ci][2y + ky - 1][2x + kx - 1] <-> sT[ci*PST + (2y + ky)*RST + 2x + kx + 3], x = 4 s + kq
    const int wb = (m >> 2) * RST + (m & 3) + 3 + 2 * kq;

    while (tile < ntiles) {
        if (tile != (int)blockIdx.x) __syncthreads();            // the previous patch has been consumed
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            f32x4 v = dc0[e] * rv[e] + dc2[e];
            if (two) v += dc1[e] * ru[e];
            *reinterpret_cast<f32x4 *>(sD + (e * NW + wave) * PS + dl) = v;
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int qp = e * NW + wave, ci = qp >> 2;
            const f32x4 v = s_tc[ci][0] * rx[e] + s_tc[ci][1];
            *reinterpret_cast<f32x4 *>(sT + ci * PST + (qp & 3) * 8 * RST + tl) = dm_relu4(v);
        }
        __syncthreads();
        const int b = tile;
        tile += gridDim.x;
        if (tile < ntiles) issue(tile);                          // in flight during the products below

        // ---- weight gradient: rows y = wave, wave + 8; four positions per step
        // (the row loops of both products stay rolled: unrolled, hipcc hoists the LDS operands of several rows and spills)
#pragma unroll 1
        for (int rr = 0; rr < 2; ++rr) {
            const int y = wave + NW * rr;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const float a = sD[m * PS + (y + 1) * RS + 4 * s + kq + 4];
                const float *pb = sT + wb + 2 * y * RST + 8 * s;
#pragma unroll
                for (int t = 0; t < C; ++t) wacc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, pb[t * PST], wacc[t], 0, 0, 0);
            }
        }
        // ---- data gradient: phase rows (Y, pu), Y = wave, wave + 8; both column phases of a row in flight
        const float *xq = x + (long long)b * C * 1024 + (long long)m * 1024 + 8 * kq;
        float *oq = dx + (long long)b * C * 1024 + (long long)m * 1024 + 8 * kq;
#pragma unroll 1
        for (int it = 0; it < 4; ++it) {
            const int Y = wave + NW * (it >> 1), pu = it & 1, u = 2 * Y + pu;
            // the statistics' second factor (the raw input) for this lane's 8 output columns: requested now
            const f32x4 q0 = *reinterpret_cast<const f32x4 *>(xq + u * 32), q1 = *reinterpret_cast<const f32x4 *>(xq + u * 32 + 4);
            f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int s = 0; s < 16; ++s) {                       // (a, b, cg)
                const int a = s >> 3, bb = (s >> 2) & 1, cg = s & 3;
#pragma unroll
                for (int pv = 0; pv < 2; ++pv) {
                    // A[m = X][k = kq] = da[4 cg + kq][Y + pu - a][X + pv - b]
                    const float av = sD[(4 * cg + kq) * PS + (Y + pu - a + 1) * RS + m + pv - bb + 4];
                    acc[pv] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, sW[((2 * pu + pv) * 16 + s) * 64 + lane], acc[pv], 0, 0, 0);
                }
            }
            // lane (m = ci, kq): X = 4 kq .. 4 kq + 3 -> columns v = 8 kq .. 8 kq + 7 of row u
            f32x4 o0 = {acc[0].x, acc[1].x, acc[0].y, acc[1].y}, o1 = {acc[0].z, acc[1].z, acc[0].w, acc[1].w};
            const float *pt = sT + m * PST + (u + 1) * RST + 4 + 8 * kq;
            const f32x4 t0 = *reinterpret_cast<const f32x4 *>(pt), t1 = *reinterpret_cast<const f32x4 *>(pt + 4);
            o0.x = t0.x > 0.f ? o0.x : 0.f; o0.y = t0.y > 0.f ? o0.y : 0.f; o0.z = t0.z > 0.f ? o0.z : 0.f; o0.w = t0.w > 0.f ? o0.w : 0.f;
            o1.x = t1.x > 0.f ? o1.x : 0.f; o1.y = t1.y > 0.f ? o1.y : 0.f; o1.z = t1.z > 0.f ? o1.z : 0.f; o1.w = t1.w > 0.f ? o1.w : 0.f;
            *reinterpret_cast<f32x4 *>(oq + u * 32) = o0;
            *reinterpret_cast<f32x4 *>(oq + u * 32 + 4) = o1;
            s1 += (double)(((o0.x + o0.y) + (o0.z + o0.w)) + ((o1.x + o1.y) + (o1.z + o1.w)));
            s2 += (double)(((o0.x * q0.x + o0.y * q0.y) + (o0.z * q0.z + o0.w * q0.w)) +
                           ((o1.x * q1.x + o1.y * q1.y) + (o1.z * q1.z + o1.w * q1.w)));
        }
    }

    // ---- statistics slab: the four kq groups of a channel, then the waves in wave order
    __syncthreads();
    {
        double a = s1, c = s2;
        a += __shfl_xor(a, 16, 64); c += __shfl_xor(c, 16, 64);
        a += __shfl_xor(a, 32, 64); c += __shfl_xor(c, 32, 64);
        if (lane < 16) { s_stat[wave][lane][0] = a; s_stat[wave][lane][1] = c; }
    }
    // ---- weight-gradient slab: the eight waves' accumulators through LDS in wave order, 8 input channels at a time
    float *red = lds4;                                           // [wave][8][64 lanes][4] = 16 384 floats (the images are free now)
    static_assert(NW * 8 * 256 <= 16 * S2_PS + 16 * S2_PST, "the slab combine reuses the tile images");
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        if (half) __syncthreads();
#pragma unroll
        for (int t = 0; t < 8; ++t) *reinterpret_cast<f32x4 *>(red + ((wave * 8 + t) * 64 + lane) * 4) = wacc[8 * half + t];
        __syncthreads();
        if (half == 0 && stats && threadIdx.x < C) {
            double ta = 0.0, tc = 0.0;
#pragma unroll
            for (int wv = 0; wv < NW; ++wv) { ta += s_stat[wv][threadIdx.x][0]; tc += s_stat[wv][threadIdx.x][1]; }
            stats[((long long)blockIdx.x * C + threadIdx.x) * 2 + 0] = ta;
            stats[((long long)blockIdx.x * C + threadIdx.x) * 2 + 1] = tc;
        }
        // element e = dW[co][ci = 8 half + t][tap]: accumulator row co = 4 kq + r of lane (m = tap, kq) in N tile t
        for (int e = threadIdx.x; e < 16 * 8 * 16; e += S2_NTH) {
            const int co = e >> 7, t = (e >> 4) & 7, tap = e & 15;
            const int ln = (co >> 2) * 16 + tap, r = co & 3;
            float sum = 0.f;
#pragma unroll
            for (int wv = 0; wv < NW; ++wv) sum += red[((wv * 8 + t) * 64 + ln) * 4 + r];
            wslabs[(long long)blockIdx.x * 4096 + (co * 16 + 8 * half + t) * 16 + tap] = sum;
        }
    }
}

bool conv4x4s2_bwd_shape(int CD, int CX, int H, int W) { return CD == 16 && CX == 16 && H == 16 && W == 16; }

}  // namespace

extern "C" int dm_conv4x4s2_bwd_fused_supported(int CD, int CX, int H, int W) { return conv4x4s2_bwd_shape(CD, CX, H, W) ? 1 : 0; }

extern "C" int dm_conv4x4s2_bwd_fused_num_blocks(int B, int CD, int CX, int H, int W)
{
    if (B <= 0 || !conv4x4s2_bwd_shape(CD, CX, H, W)) return -1;
    return B < 256 ? B : 256;                                    // one resident workgroup per CU
}

extern "C" int dm_conv4x4s2_bwd_fused(const dm_operand *dy, const float *x, const float *xcoef, const float *w, float *dx,
                                      double *stats, float *wslabs, int B, int CD, int CX, int H, int W, void *stream)
{
    DM_REQUIRE(dy && dy->p0 && x && xcoef && w && dx && stats && wslabs, "dm_conv4x4s2_bwd_fused: NULL pointer");
    DM_REQUIRE(B > 0 && conv4x4s2_bwd_shape(CD, CX, H, W), "dm_conv4x4s2_bwd_fused: shape %d -> %d channels, %dx%d output grid not built",
               CX, CD, H, W);
    DM_REQUIRE(dy->mode == DM_LOAD_IDENT || dy->mode == DM_LOAD_AFFINE2, "dm_conv4x4s2_bwd_fused: dy operand must be IDENT or AFFINE2");
    DM_REQUIRE(dy->mode == DM_LOAD_IDENT || dy->coef, "dm_conv4x4s2_bwd_fused: AFFINE2 needs coefficients");
    DM_REQUIRE(dy->coef_bstride == 0 && !dy->ones_channel, "dm_conv4x4s2_bwd_fused: shared coefficients only");
    Operand d = to_dev(dy);
    if (d.mode == DM_LOAD_IDENT) { d.coef = nullptr; d.p1 = nullptr; }
    static DmPerDeviceOnce attr_done;
    if (attr_done.need()) {
        const hipError_t e = hipFuncSetAttribute((const void *)conv4x4s2_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                 (int)S2_LDS_BYTES);
        if (e != hipSuccess) {
            dm_set_error("dm_conv4x4s2_bwd_fused: hipFuncSetAttribute: %s", hipGetErrorString(e));
            return (int)e;
        }
        attr_done.mark();
    }
    const int grid = dm_conv4x4s2_bwd_fused_num_blocks(B, CD, CX, H, W);
    hipLaunchKernelGGL(conv4x4s2_bwd_kernel, dim3(grid), dim3(S2_NTH), S2_LDS_BYTES, (hipStream_t)stream, d, x, xcoef, w, dx, stats,
                       wslabs, B);
    return dm_launch_status("dm_conv4x4s2_bwd_fused");
}

// ================================================================================================================ forward
// enc.7 forward on whole patches.  Rounds 1-3 ran conv4x4s2_kernel<16, 1, 8, 16, ..> on 8 x 16 output tiles: 67 us at
// B = 2048 in the training step (a 27 us matrix floor, 34 us of traffic) and 45 us at B = 1024 in the inference path.  With the
// transformed input patch (16 x 34 x 40 floats with its zero padding) in LDS once, every output row is 64 matrix instructions
// with both operands on immediate LDS offsets: M = the 16 x of a row, N = co, K step (ci, ky) with k = kx.
// Statistics: one slab per workgroup (batch statistics), or -- per_tile, the per-sample BatchNorm statistics of process_VAE --
// per patch, written to the first of the two tile slabs the caller sized for conv4x4s2_kernel's 8-row tiles (the second: zeros).
namespace {

constexpr int F2_LDS_FLOATS = 16 * S2_PST + 64 * 64;
constexpr size_t F2_LDS_BYTES = (size_t)F2_LDS_FLOATS * sizeof(float);

__global__ __launch_bounds__(S2_NTH, 1)
void conv4x4s2_patch_forward_kernel(Operand in, WeightView wv, const float *__restrict__ bias, int relu_out,
                                    float *__restrict__ out, double *__restrict__ stats, int per_tile, int nslabs, int ntiles)
{
    constexpr int C = 16, NW = S2_NW, PST = S2_PST, RST = S2_RST;
    extern __shared__ __attribute__((aligned(16))) float ldsf[];
    float *sT = ldsf, *sW = ldsf + C * PST;
    __shared__ double s_stat[NW][C][2];

    const int lane = threadIdx.x & 63, m = lane & 15, kq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < C * PST / 4; i += S2_NTH) reinterpret_cast<f32x4 *>(ldsf)[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // weights in operand order: step s = (ci, ky): B[k = kx = kq][n = co = m] = W[co][ci][ky][kx] through the caller's view
    for (int s = wave; s < 64; s += NW)
        sW[s * 64 + lane] = wv.w[wv.off + m * wv.sn + (s >> 2) * wv.sc + (s & 3) * wv.sky + kq * wv.skx];
    const float bco = bias ? bias[m] : 0.f;
    const bool affine = in.mode == DM_LOAD_AFFINE || in.mode == DM_LOAD_AFFINE_RELU;
    const bool relu_in = in.mode == DM_LOAD_RELU || in.mode == DM_LOAD_AFFINE_RELU;
    double s1 = 0.0, s2 = 0.0;

    const int tq = (lane >> 3) * 32 + 4 * (lane & 7), tl = ((lane >> 3) + 1) * RST + 4 + 4 * (lane & 7);
    f32x4 rx[8];
    float c0[8], c2[8];                                          // coefficients of the planes this wave stages, for the patch in rx
    auto issue = [&](int b) {
        const long long xb = (long long)b * C * 1024;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int qp = e * NW + wave, ci = qp >> 2;
            rx[e] = *reinterpret_cast<const f32x4 *>(in.p0 + xb + (long long)ci * 1024 + (qp & 3) * 256 + tq);
            if (affine) {
                const float *cf = in.coef + (long long)b * in.coef_bstride + ci * 4;
                c0[e] = cf[0]; c2[e] = cf[2];
            }
        }
    };
    int tile = blockIdx.x;
    if (tile < ntiles) issue(tile);
    __syncthreads();

    while (tile < ntiles) {
        if (tile != (int)blockIdx.x) __syncthreads();            // the previous patch has been consumed
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int qp = e * NW + wave, ci = qp >> 2;
            f32x4 v = rx[e];
            if (affine) v = c0[e] * v + c2[e];
            if (relu_in) v = dm_relu4(v);
            *reinterpret_cast<f32x4 *>(sT + ci * PST + (qp & 3) * 8 * RST + tl) = v;
        }
        __syncthreads();
        const int b = tile;
        tile += gridDim.x;
        if (tile < ntiles) issue(tile);                          // in flight during the products below

        // rows y = wave, wave + 8: A[m = x][k = kx = kq] = T[ci][2y + ky - 1][2x + kx - 1] <-> sT[ci*PST + (2y + ky)*RST + 2x + kx + 3]
        f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        const float *pa = sT + 2 * wave * RST + 2 * m + kq + 3;
        // the LDS operands of the next group of four K steps (one input channel) are requested before the products of this
        // one: left to hipcc every read sinks next to its matrix instruction and its latency is paid per step
        float ga[2][4], gb[2][4], gw[2][4];
        auto fetch = [&](int g, int buf) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int off = g * PST + j * RST;
                ga[buf][j] = pa[off]; gb[buf][j] = pa[off + 16 * RST]; gw[buf][j] = sW[(4 * g + j) * 64 + lane];
            }
        };
        fetch(0, 0);
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            if (g + 1 < 16) fetch(g + 1, (g + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(ga[g & 1][j], gw[g & 1][j], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(gb[g & 1][j], gw[g & 1][j], acc[1], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        float *ob = out + (long long)b * C * 256 + (long long)m * 256 + 4 * kq;
        double p1 = 0.0, p2 = 0.0;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            f32x4 v = acc[j] + bco;
            if (relu_out) v = dm_relu4(v);
            *reinterpret_cast<f32x4 *>(ob + (wave + NW * j) * 16) = v;
            p1 += (double)((v.x + v.y) + (v.z + v.w));
            p2 += (double)((v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w));
        }
        if (stats && per_tile) {
            // per-sample statistics: this patch's sums to its first slab, zeros to its second
            p1 += __shfl_xor(p1, 16, 64); p2 += __shfl_xor(p2, 16, 64);
            p1 += __shfl_xor(p1, 32, 64); p2 += __shfl_xor(p2, 32, 64);
            if (lane < 16) { s_stat[wave][lane][0] = p1; s_stat[wave][lane][1] = p2; }
            __syncthreads();
            if (threadIdx.x < C) {
                double ta = 0.0, tc = 0.0;
#pragma unroll
                for (int w8 = 0; w8 < NW; ++w8) { ta += s_stat[w8][threadIdx.x][0]; tc += s_stat[w8][threadIdx.x][1]; }
                const int spp = nslabs / ntiles;                 // slabs the caller holds per patch
                double *dst = stats + ((long long)b * spp * C + threadIdx.x) * 2;
                dst[0] = ta; dst[1] = tc;
                for (int k = 1; k < spp; ++k) { dst[(long long)k * C * 2] = 0.0; dst[(long long)k * C * 2 + 1] = 0.0; }
            }
        } else {
            s1 += p1; s2 += p2;
        }
    }
    if (stats && !per_tile) {
        __syncthreads();
        double a = s1, c = s2;
        a += __shfl_xor(a, 16, 64); c += __shfl_xor(c, 16, 64);
        a += __shfl_xor(a, 32, 64); c += __shfl_xor(c, 32, 64);
        if (lane < 16) { s_stat[wave][lane][0] = a; s_stat[wave][lane][1] = c; }
        __syncthreads();
        if (threadIdx.x < C) {
            double ta = 0.0, tc = 0.0;
#pragma unroll
            for (int w8 = 0; w8 < NW; ++w8) { ta += s_stat[w8][threadIdx.x][0]; tc += s_stat[w8][threadIdx.x][1]; }
            stats[((long long)blockIdx.x * C + threadIdx.x) * 2 + 0] = ta;
            stats[((long long)blockIdx.x * C + threadIdx.x) * 2 + 1] = tc;
        }
        for (int t2 = blockIdx.x + gridDim.x; t2 < nslabs; t2 += gridDim.x)         // slabs no workgroup owns
            for (int i = threadIdx.x; i < C * 2; i += S2_NTH) stats[(long long)t2 * C * 2 + i] = 0.0;
    }
}

}  // namespace

// Called by dm_conv4x4s2 (conv_mfma.hip) for the shape this kernel is built for; returns false when it does not apply.
// DM_PATCH_CONV=0 in the environment keeps the tiled kernel (A/B runs).
bool dm_conv4x4s2_patch_forward(const Operand &in, const WeightView &wv, float *out, const Epilogue &ep, int B, int Cphys, int CIN,
                                int NOUT, int H, int W, int per_tile, int nslabs, hipStream_t stream, int *rc)
{
    static const bool off = [] { const char *e = getenv("DM_PATCH_CONV"); return e && e[0] == '0'; }();
    if (off || CIN != 16 || Cphys != 16 || NOUT != 16 || H != 32 || W != 32) return false;
    if (in.mode == DM_LOAD_AFFINE2 || in.ones || ep.mask.p0 || ep.resid || ep.stat_q || ep.bias_border) return false;
    if (per_tile && (!ep.stats || nslabs % B != 0)) return false;
    static DmPerDeviceOnce attr_done;
    if (attr_done.need()) {
        const hipError_t e = hipFuncSetAttribute((const void *)conv4x4s2_patch_forward_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                 (int)F2_LDS_BYTES);
        if (e != hipSuccess) return false;
        attr_done.mark();
    }
    const int grid = B < 256 ? B : 256;
    hipLaunchKernelGGL(conv4x4s2_patch_forward_kernel, dim3(grid), dim3(S2_NTH), F2_LDS_BYTES, stream, in, wv, ep.bias, ep.relu, out,
                       ep.stats, per_tile, nslabs, B);
    *rc = dm_launch_status("dm_conv4x4s2");
    return true;
}

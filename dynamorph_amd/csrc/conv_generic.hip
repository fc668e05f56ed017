// conv_generic.hip -- fallback kernels for channel families / spatial sizes the MFMA kernels are not built for.
//
// The MFMA kernels (conv_mfma.hip, wgrad_mfma.hip) are instantiated for the default VQ-VAE family (num_hiddens 16,
// num_residual_hiddens 32) and tile-aligned spatial sizes.  The reference's example configuration trains a wider
// network (config_example.yml: num_hiddens 64, num_residual_hiddens 64, 512 codes); rather than refuse it, dm_conv4x4s2 /
// dm_conv3x3 / dm_wgrad fall through to these kernels: same operands (dm_operand transforms, dm_weight_view,
// dm_epilogue with bias / ReLU / mask / residual / statistics slabs), plain fp32 FMAs on the VALU, one workgroup per
// sample, no tiling constraints.  They are several times slower than the MFMA kernels and exist for coverage, not speed
// (still the GPU: nothing here is a CPU fallback).
#include "dm_common.h"

namespace {

enum { GEN_S2 = 0, GEN_S1 = 1, GEN_PIX = 2 };

// operand value at (b, c, y, x) with zero padding; ones channel (c == Cphys) is 1 inside the image
__device__ __forceinline__ float gen_load(const Operand &op, int b, int c, int y, int x, int Cphys, int H, int W)
{
    if (y < 0 || y >= H || x < 0 || x >= W) return 0.f;
    if (c >= Cphys) return 1.f;
    const long long off = (((long long)b * Cphys + c) * H + y) * W + x;
    float v = op.p0[off];
    if (op.mode == DM_LOAD_IDENT) return v;
    if (op.mode == DM_LOAD_RELU) return dm_relu(v);                // a NaN stays NaN
    const float *cf = op.coef + (long long)b * op.coef_bstride + c * 4;
    if (op.mode == DM_LOAD_AFFINE2) return cf[0] * v + (cf[1] * op.p1[off] + cf[2]);
    v = cf[0] * v + cf[2];
    if (op.mode == DM_LOAD_AFFINE_RELU) v = dm_relu(v);
    return v;
}

// One workgroup per sample (grid-stride over samples); channel by channel, 256 output pixels at a time.
//   GEN_S2 : out (B, NOUT, H/2, W/2) = conv 4x4 stride 2 pad 1
//   GEN_S1 : out (B, NOUT, H, W)     = conv 3x3 pad 1 (taps 9) or 1x1 (taps 1)
//   GEN_PIX: out (B, NOUT/4, 2H, 2W) = ConvTranspose2d(4,2,1) in dm_conv3x3's pixel-shuffle weight convention
template <int FORM>
__global__ __launch_bounds__(256) void conv_generic_kernel(Operand in, WeightView wv, float *__restrict__ out, Epilogue ep,
                                                           int B, int Cphys, int CIN, int NOUT, int H, int W, int taps,
                                                           int nslabs, int per_tile)
{
    __shared__ double s_red[4];
    const int CO = FORM == GEN_PIX ? NOUT >> 2 : NOUT;
    const int OH = FORM == GEN_S2 ? H >> 1 : (FORM == GEN_PIX ? 2 * H : H);
    const int OW = FORM == GEN_S2 ? W >> 1 : (FORM == GEN_PIX ? 2 * W : W);
    const int npix = OH * OW;
    const int spg = per_tile ? nslabs / B : 1;                       // slabs per sample when grouped per sample
    for (int b = blockIdx.x; b < B; b += gridDim.x) {
        for (int co = 0; co < CO; ++co) {
            float mc0 = ep.mask.p0 ? 1.f : 0.f, mc2 = ep.mask.p0 ? 0.f : 1.f;
            if (ep.mask.p0 && ep.mask.mode >= DM_LOAD_AFFINE) {
                const float *cf = ep.mask.coef + (long long)b * ep.mask.coef_bstride + co * 4;
                mc0 = cf[0]; mc2 = cf[2];
            }
            const float bias = ep.bias ? ep.bias[co] : 0.f;
            float tb[9];                                           // bias_border[row class][column class][co] (GEN_S2 only)
            if (FORM == GEN_S2 && ep.bias_border)
                for (int q = 0; q < 9; ++q) tb[q] = ep.bias_border[q * CO + co];
            double s1 = 0.0, s2 = 0.0;
            for (int p = threadIdx.x; p < npix; p += blockDim.x) {
                const int oy = p / OW, ox = p - oy * OW;
                float acc = 0.f;
                if (FORM == GEN_S2) {
                    for (int c = 0; c < CIN; ++c)
                        for (int ky = 0; ky < 4; ++ky)
                            for (int kx = 0; kx < 4; ++kx)
                                acc += gen_load(in, b, c, 2 * oy - 1 + ky, 2 * ox - 1 + kx, Cphys, H, W) *
                                       wv.w[wv.off + co * wv.sn + c * wv.sc + ky * wv.sky + kx * wv.skx];
                } else if (FORM == GEN_S1) {
                    const int r = taps == 9 ? 1 : 0;
                    for (int c = 0; c < CIN; ++c)
                        for (int ty = 0; ty <= 2 * r; ++ty)
                            for (int tx = 0; tx <= 2 * r; ++tx)
                                acc += gen_load(in, b, c, oy - r + ty, ox - r + tx, Cphys, H, W) *
                                       wv.w[wv.off + co * wv.sn + c * wv.sc + ty * wv.sky + tx * wv.skx];
                } else {
                    const int py = oy & 1, px = ox & 1, y = oy >> 1, x = ox >> 1;
                    for (int c = 0; c < CIN; ++c)
                        for (int a = 0; a < 2; ++a)
                            for (int bb = 0; bb < 2; ++bb) {
                                const int tyy = py + a, txx = px + bb;
                                const int ky = py + 3 - 2 * tyy, kx = px + 3 - 2 * txx;
                                acc += gen_load(in, b, c, y - 1 + tyy, x - 1 + txx, Cphys, H, W) *
                                       wv.w[wv.off + co * wv.sn + c * wv.sc + ky * wv.sky + kx * wv.skx];
                            }
                }
                float v = acc + bias;
                if (FORM == GEN_S2 && ep.bias_border) {
                    const int ry = oy == 0 ? 0 : (oy == OH - 1 ? 2 : 1), rx = ox == 0 ? 0 : (ox == OW - 1 ? 2 : 1);
                    v = acc + tb[ry * 3 + rx];
                }
                if (ep.relu) v = dm_relu(v);
                const long long o = (((long long)b * CO + co) * OH + oy) * OW + ox;
                float mval = 1.f;
                if (ep.mask.p0) {
                    mval = ep.mask.p0[o];
                    v = (mc0 * mval + mc2) > 0.f ? v : 0.f;
                }
                if (ep.resid) v += ep.resid[o];
                out[o] = v;
                const float q = ep.stat_q ? ep.stat_q[o] : v;
                s1 += (double)v;
                s2 += (double)v * (double)q;
            }
            if (ep.stats) {
                const double t1 = block_sum(s1, s_red);
                const double t2 = block_sum(s2, s_red);
                if (threadIdx.x == 0) {
                    if (per_tile) {
                        double *st = ep.stats + ((long long)b * spg * CO + co) * 2;
                        st[0] = t1; st[1] = t2;
                        for (int k = 1; k < spg; ++k) { st[(long long)k * CO * 2] = 0.0; st[(long long)k * CO * 2 + 1] = 0.0; }
                    } else {
                        double *st = ep.stats + ((long long)blockIdx.x * CO + co) * 2;
                        if (b == (int)blockIdx.x) { st[0] = t1; st[1] = t2; }
                        else { st[0] += t1; st[1] += t2; }
                    }
                }
                __syncthreads();
            }
        }
    }
    if (ep.stats && !per_tile) {                                     // slabs no workgroup owns
        for (int t2 = blockIdx.x + gridDim.x; t2 < nslabs; t2 += gridDim.x)
            for (int i = threadIdx.x; i < CO * 2; i += blockDim.x) ep.stats[(long long)t2 * CO * 2 + i] = 0.0;
    }
}

// R[cs][ct][ky][kx] = sum_{b,y,x} S'[b,cs,y,x] * T'[b,ct, y*s+ky-p, x*s+kx-p]; workgroup k takes samples k, k+grid, ...
// and writes slab k (E = CS*CT*KK*KK floats); thread = one output element (grid-stride over the E elements).
__global__ __launch_bounds__(256) void wgrad_generic_kernel(Operand S, Operand T, float *__restrict__ slabs, int B, int CS,
                                                            int CT, int CTphys, int Hs, int Ws, int KK, int nslabs)
{
    const int stride = KK == 4 ? 2 : 1, pad = KK == 1 ? 0 : 1;
    const int Ht = Hs * stride, Wt = Ws * stride;
    const int E = CS * CT * KK * KK;
    for (int e = threadIdx.x; e < E; e += blockDim.x) {
        const int kx = e % KK, ky = (e / KK) % KK, ct = (e / (KK * KK)) % CT, cs = e / (KK * KK * CT);
        // rows are summed in fp32, rows into a double: the gradients of weights that see a BatchNorm-backward operand
        // are small differences of large sums, and one long fp32 chain loses them
        double acc = 0.0;
        for (int b = blockIdx.x; b < B; b += gridDim.x)
            for (int y = 0; y < Hs; ++y) {
                float row = 0.f;
                for (int x = 0; x < Ws; ++x)
                    row += gen_load(S, b, cs, y, x, CS, Hs, Ws) *
                           gen_load(T, b, ct, y * stride + ky - pad, x * stride + kx - pad, CTphys, Ht, Wt);
                acc += (double)row;
            }
        slabs[(long long)blockIdx.x * E + e] = (float)acc;
    }
    for (int t2 = blockIdx.x + gridDim.x; t2 < nslabs; t2 += gridDim.x)
        for (int e = threadIdx.x; e < E; e += blockDim.x) slabs[(long long)t2 * E + e] = 0.f;
}

}  // namespace

// ---- entry points used by the dispatchers in conv_mfma.hip / wgrad_mfma.hip (not part of the public header) ----------
int dm_generic_conv_slabs(int B, int per_tile) { return per_tile ? B : (B < 768 ? B : 768); }

int dm_generic_conv(int form, const Operand &in, const WeightView &wv, float *out, const Epilogue &ep, int B, int Cphys,
                    int CIN, int NOUT, int H, int W, int taps, int nslabs, int per_tile, hipStream_t st)
{
    const int grid = per_tile ? B : (B < nslabs || !ep.stats ? (B < 2048 ? B : 2048) : nslabs);
    if (form == GEN_S2)
        hipLaunchKernelGGL((conv_generic_kernel<GEN_S2>), dim3(grid), dim3(256), 0, st, in, wv, out, ep, B, Cphys, CIN, NOUT,
                           H, W, taps, nslabs, per_tile);
    else if (form == GEN_S1)
        hipLaunchKernelGGL((conv_generic_kernel<GEN_S1>), dim3(grid), dim3(256), 0, st, in, wv, out, ep, B, Cphys, CIN, NOUT,
                           H, W, taps, nslabs, per_tile);
    else
        hipLaunchKernelGGL((conv_generic_kernel<GEN_PIX>), dim3(grid), dim3(256), 0, st, in, wv, out, ep, B, Cphys, CIN, NOUT,
                           H, W, taps, nslabs, per_tile);
    return 0;
}

int dm_generic_wgrad(const Operand &S, const Operand &T, float *slabs, int B, int CS, int CT, int CTphys, int Hs, int Ws,
                     int KK, int nslabs, hipStream_t st)
{
    const int grid = B < nslabs ? B : nslabs;
    hipLaunchKernelGGL(wgrad_generic_kernel, dim3(grid), dim3(256), 0, st, S, T, slabs, B, CS, CT, CTphys, Hs, Ws, KK, nslabs);
    return 0;
}

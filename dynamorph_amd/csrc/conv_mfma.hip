// conv_mfma.hip -- implicit-GEMM convolutions on v_mfma_f32_16x16x4_f32 (exact fp32 FMA chains).
//
// Replaces aten::convolution / aten::conv_transpose2d / aten::convolution_backward(data)
// for every layer of VQ_VAE.enc / VQ_VAE.dec (HiddenStateExtractor/vq_vae.py:276-298, the
// ResidualBlock convs at :203-209).
//
// GEMM view per workgroup: M = 16 consecutive output pixels of one row (one MFMA tile),
// N = output channels (16 per tile), K = input channels x taps.  The input tile (with its
// halo, after the on-load operand transform: BatchNorm apply / ReLU / BatchNorm backward)
// is staged once in LDS; weights for the wave's N tiles live in registers for the whole
// workgroup, so each MFMA needs exactly one 4-byte LDS read (the A operand).
//
//  kernel A  conv4x4s2: K step = (ci, ky), the 4 k-lanes of the MFMA are the 4 taps kx.
//            lane (m = lane&15, kq = lane>>4) reads tile[ci][2r+ky][2m+kq+..]: the 32 lanes
//            of a ds_read_b32 group touch 32 consecutive dwords -> conflict free.
//  kernel B  conv3x3 / 1x1: K step = (4-channel group, tap), the 4 k-lanes are 4 channels;
//            plane stride == 16 (mod 32) dwords makes the two channel planes of a 32-lane
//            group hit disjoint bank halves -> conflict free.
//            pixel_shuffle: the 3x3 neighbourhood formulation of ConvTranspose2d(4,2,1)
//            with N = 4 phases x Cout; lane pairs exchange values so every lane still
//            stores 16 contiguous bytes.
//
// Epilogue (dm_epilogue): bias, ReLU, ReLU-backward mask, residual add, store, and per-channel
// partial sums (sum v, sum v*q) in double -- one slab per workgroup, reduced deterministically
// by the finalize kernels in bn.hip.
#include "dm_common.h"
#include "tile.h"

namespace {

// ----------------------------------------------------------------------------- epilogue
// Side inputs of one output float4 (ReLU-backward mask tensor, residual, second-moment partner) are
// loaded first for every tile of a pass and only then consumed, so the loads overlap.
struct EpiIn {
    f32x4 m, r, q;
};

__device__ __forceinline__ EpiIn epilogue_loads(const Epilogue &ep, long long off, bool valid)
{
    EpiIn e;
    e.m = (f32x4){1.f, 1.f, 1.f, 1.f};
    e.r = (f32x4){0.f, 0.f, 0.f, 0.f};
    e.q = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (valid) {
        if (ep.mask.p0) e.m = *reinterpret_cast<const f32x4 *>(ep.mask.p0 + off);
        if (ep.resid) e.r = *reinterpret_cast<const f32x4 *>(ep.resid + off);
        if (ep.stat_q) e.q = *reinterpret_cast<const f32x4 *>(ep.stat_q + off);
    }
    return e;
}

// mask coefficients (c0, c2) of the lane's output channel: keep v where c0*m + c2 > 0
__device__ __forceinline__ void mask_coef(const Epilogue &ep, int b, int chan, float &c0, float &c2)
{
    c0 = 1.f; c2 = 0.f;
    if (ep.mask.p0 && ep.mask.mode >= DM_LOAD_AFFINE) {
        const float *cf = ep.mask.coef + (long long)b * ep.mask.coef_bstride + chan * 4;
        c0 = cf[0]; c2 = cf[2];
    }
}

// v: 4 consecutive output elements (along x) at element offset `off`.
__device__ __forceinline__ void epilogue_tail(f32x4 v, const Epilogue &ep, const EpiIn &e, float mc0, float mc2,
                                              float *__restrict__ out, long long off, double &s1, double &s2)
{
    if (ep.mask.p0) {
        const f32x4 mv = mc0 * e.m + mc2;
        v.x = mv.x > 0.f ? v.x : 0.f; v.y = mv.y > 0.f ? v.y : 0.f;
        v.z = mv.z > 0.f ? v.z : 0.f; v.w = mv.w > 0.f ? v.w : 0.f;
    }
    v += e.r;
    *reinterpret_cast<f32x4 *>(out + off) = v;
    if (ep.stats) {
        const f32x4 q = ep.stat_q ? e.q : v;
        s1 += (double)v.x + (double)v.y + (double)v.z + (double)v.w;
        s2 += (double)(v.x * q.x) + (double)(v.y * q.y) + (double)(v.z * q.z) + (double)(v.w * q.w);
    }
}

__device__ __forceinline__ f32x4 bias_relu(f32x4 v, const Epilogue &ep, float bias)
{
    v += bias;
    if (ep.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    return v;
}

// Per-workgroup reduction of the per-lane channel partials -> stats[block][NCH][2].
// Lane layout: channel n = 16*t + (lane & 15) (PIX: channel = n >> 2); partials of the 4 lane
// quarters (lane >> 4) and, with PIX, of the 4 phase lanes are summed with shuffles.
template <int NTT, bool PIX>
__device__ __forceinline__ void stats_reduce(double (&s1)[NTT], double (&s2)[NTT], double (*s_stat)[2],
                                             const Epilogue &ep, int NCH)
{
    // s_stat: [4 waves][NTT*16][2]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int t = 0; t < NTT; ++t) {
        double a = s1[t], c = s2[t];
        a += __shfl_xor(a, 16, 64); c += __shfl_xor(c, 16, 64);
        a += __shfl_xor(a, 32, 64); c += __shfl_xor(c, 32, 64);
        if (PIX) {
            a += __shfl_xor(a, 1, 64); c += __shfl_xor(c, 1, 64);
            a += __shfl_xor(a, 2, 64); c += __shfl_xor(c, 2, 64);
        }
        if (lane < 16) {
            s_stat[(wave * NTT + t) * 16 + lane][0] = a;
            s_stat[(wave * NTT + t) * 16 + lane][1] = c;
        }
    }
    __syncthreads();
    for (int ch = threadIdx.x; ch < NCH; ch += DM_BLOCK) {
        const int n = PIX ? ch * 4 : ch;      // lane slot that holds the channel total
        const int t = n >> 4, l = n & 15;
        double a = 0.0, c = 0.0;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            a += s_stat[(w * NTT + t) * 16 + l][0];
            c += s_stat[(w * NTT + t) * 16 + l][1];
        }
        ep.stats[((long long)blockIdx.x * NCH + ch) * 2 + 0] = a;
        ep.stats[((long long)blockIdx.x * NCH + ch) * 2 + 1] = c;
    }
}

// ============================================================================ kernel A
template <int CIN, int NT, int TH, int TW, bool TWO>
__global__ __launch_bounds__(DM_BLOCK) void conv4x4s2_kernel(Operand in, WeightView wv, float *__restrict__ out,
                                                             Epilogue ep, int Cphys, int NOUT, int H, int W)
{
    constexpr int IH = 2 * TH + 2, RS = 2 * TW + 8, COLS4 = RS / 4, PS = IH * RS;
    constexpr int KS = CIN * 4, CG = TW / 16, MT = TH * CG, MTW = MT / 4;
    static_assert(MTW % 2 == 0 && MTW >= 2, "need an even number of M tiles per wave");
    __shared__ __attribute__((aligned(16))) float tile[CIN * PS];
    __shared__ __attribute__((aligned(16))) float s_coef[DM_COEF_MAX_C * 4];
    __shared__ double s_stat[4 * NT * 16][2];

    const int Ho = H >> 1, Wo = W >> 1;
    const int tiles_x = Wo / TW, tiles_y = Ho / TH;
    int bid = blockIdx.x;
    const int tx = bid % tiles_x; bid /= tiles_x;
    const int ty = bid % tiles_y;
    const int b = bid / tiles_y;
    const int oy0 = ty * TH, ox0 = tx * TW;

    TileStage<CIN, IH, COLS4, RS, PS, TWO> stage;
    stage.issue(in, b, Cphys, H, W, 2 * oy0 - 1, 2 * ox0 - 4);
    stage_coef(s_coef, in, b, Cphys);

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, m = lane & 15, kq = lane >> 4;
    float wreg[NT][KS], bias[NT], mc0[NT], mc2[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int n = 16 * t + m;
        const int nc = n < NOUT ? n : 0;
        bias[t] = ep.bias ? ep.bias[nc] : 0.f;
        mask_coef(ep, b, nc, mc0[t], mc2[t]);
#pragma unroll
        for (int s = 0; s < KS; ++s)
            wreg[t][s] = n < NOUT ? wv.w[wv.off + n * wv.sn + (s >> 2) * wv.sc + (s & 3) * wv.sky + kq * wv.skx] : 0.f;
    }
    __syncthreads();
    stage.commit(tile, s_coef, Cphys, H, W, 2 * oy0 - 1, 2 * ox0 - 4);
    __syncthreads();

    double s1[NT], s2[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) { s1[t] = 0.0; s2[t] = 0.0; }

    const int abase = 2 * m + kq + 3;
    for (int p = 0; p < MTW / 2; ++p) {
        const int t0 = wave + 8 * p, t1 = t0 + 4;
        const int r0 = t0 / CG, c0 = t0 % CG, r1 = t1 / CG, c1 = t1 % CG;
        const float *a0p = tile + (2 * r0) * RS + 32 * c0 + abase;
        const float *a1p = tile + (2 * r1) * RS + 32 * c1 + abase;
        // side inputs of the epilogue first: they are in flight during the MFMA loop
        EpiIn e0[NT], e1[NT];
        long long o0[NT], o1[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int n = 16 * t + m;
            o0[t] = (((long long)b * NOUT + n) * Ho + (oy0 + r0)) * Wo + ox0 + 16 * c0 + 4 * kq;
            o1[t] = (((long long)b * NOUT + n) * Ho + (oy0 + r1)) * Wo + ox0 + 16 * c1 + 4 * kq;
            e0[t] = epilogue_loads(ep, o0[t], n < NOUT);
            e1[t] = epilogue_loads(ep, o1[t], n < NOUT);
        }
        f32x4 acc0[NT], acc1[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) { acc0[t] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc1[t] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int o = (s >> 2) * PS + (s & 3) * RS;
            const float a0 = a0p[o], a1 = a1p[o];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                acc0[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, wreg[t][s], acc0[t], 0, 0, 0);
                acc1[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, wreg[t][s], acc1[t], 0, 0, 0);
            }
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            if (16 * t + m < NOUT) {
                epilogue_tail(bias_relu(acc0[t], ep, bias[t]), ep, e0[t], mc0[t], mc2[t], out, o0[t], s1[t], s2[t]);
                epilogue_tail(bias_relu(acc1[t], ep, bias[t]), ep, e1[t], mc0[t], mc2[t], out, o1[t], s1[t], s2[t]);
            }
        }
    }
    if (ep.stats) stats_reduce<NT, false>(s1, s2, s_stat, ep, NOUT);
}

// ============================================================================ kernel B
template <int CIN, int NT, int NPASS, int TAPS, bool PIX, int TH, int TW, bool TWO>
__global__ __launch_bounds__(DM_BLOCK) void conv3x3_kernel(Operand in, WeightView wv, float *__restrict__ out,
                                                           Epilogue ep, int Cphys, int NOUT, int H, int W)
{
    constexpr int PADR = TAPS == 9 ? 1 : 0;
    constexpr int IH = TH + 2 * PADR, RS = TAPS == 9 ? TW + 8 : TW, COLS4 = RS / 4;
    constexpr int PSRAW = IH * RS;
    constexpr int PS = PSRAW + ((16 - (PSRAW % 32)) + 32) % 32;      // PS == 16 (mod 32)
    constexpr int KS = (CIN / 4) * TAPS, CG = TW / 16, MT = TH * CG, MTW = MT / 4;
    constexpr int NTT = NT * NPASS;
    static_assert(CIN % 4 == 0, "channel groups of 4");
    static_assert(MTW % 2 == 0 && MTW >= 2, "need an even number of M tiles per wave");
    static_assert(!PIX || TAPS == 9, "pixel shuffle is the 3x3 formulation");
    __shared__ __attribute__((aligned(16))) float tile[CIN * PS];
    __shared__ __attribute__((aligned(16))) float s_coef[DM_COEF_MAX_C * 4];
    __shared__ double s_stat[4 * NTT * 16][2];

    const int tiles_x = W / TW, tiles_y = H / TH;
    int bid = blockIdx.x;
    const int tx = bid % tiles_x; bid /= tiles_x;
    const int ty = bid % tiles_y;
    const int b = bid / tiles_y;
    const int y0 = ty * TH, x0 = tx * TW;

    TileStage<CIN, IH, COLS4, RS, PS, TWO> stage;
    stage.issue(in, b, Cphys, H, W, y0 - PADR, x0 - 4 * PADR);
    stage_coef(s_coef, in, b, Cphys);

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, m = lane & 15, kq = lane >> 4;
    const int CO = PIX ? NOUT >> 2 : NOUT;           // physical output channels
    const int OH = PIX ? 2 * H : H, OW = PIX ? 2 * W : W;

    double s1[NTT], s2[NTT];
#pragma unroll
    for (int t = 0; t < NTT; ++t) { s1[t] = 0.0; s2[t] = 0.0; }

    const int abase = kq * PS + m + 3 * PADR;
#pragma unroll
    for (int pass = 0; pass < NPASS; ++pass) {
        float wreg[NT][KS], bias[NT], mc0[NT], mc2[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int n = 16 * (pass * NT + t) + m;
            const int nb = n < NOUT ? (PIX ? n >> 2 : n) : 0;
            bias[t] = ep.bias ? ep.bias[nb] : 0.f;
            mask_coef(ep, b, nb, mc0[t], mc2[t]);
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const int cg4 = s / TAPS, tap = s % TAPS;
                const int c = 4 * cg4 + kq;
                const int tyy = TAPS == 9 ? tap / 3 : 0, txx = TAPS == 9 ? tap % 3 : 0;
                float wvl = 0.f;
                if (n < NOUT) {
                    if (PIX) {
                        const int co = n >> 2, py = (n >> 1) & 1, px = n & 1;
                        const int ky = py + 3 - 2 * tyy, kx = px + 3 - 2 * txx;
                        if (ky >= 0 && ky <= 3 && kx >= 0 && kx <= 3)
                            wvl = wv.w[wv.off + co * wv.sn + c * wv.sc + ky * wv.sky + kx * wv.skx];
                    } else {
                        wvl = wv.w[wv.off + n * wv.sn + c * wv.sc + tyy * wv.sky + txx * wv.skx];
                    }
                }
                wreg[t][s] = wvl;
            }
        }
        if (pass == 0) {
            __syncthreads();
            stage.commit(tile, s_coef, Cphys, H, W, y0 - PADR, x0 - 4 * PADR);
            __syncthreads();
        }

        for (int p = 0; p < MTW / 2; ++p) {
            const int t0 = wave + 8 * p, t1 = t0 + 4;
            const int r0 = t0 / CG, c0 = t0 % CG, r1 = t1 / CG, c1 = t1 % CG;
            const float *a0p = tile + r0 * RS + 16 * c0 + abase;
            const float *a1p = tile + r1 * RS + 16 * c1 + abase;
            EpiIn e0[NT], e1[NT];
            long long o0[NT], o1[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int n = 16 * (pass * NT + t) + m;
                if (PIX) {
                    const int nb = n >> 2, py = (n >> 1) & 1, px = n & 1;
                    o0[t] = (((long long)b * CO + nb) * OH + 2 * (y0 + r0) + py) * OW + 2 * (x0 + 16 * c0 + 4 * kq) + 4 * px;
                    o1[t] = (((long long)b * CO + nb) * OH + 2 * (y0 + r1) + py) * OW + 2 * (x0 + 16 * c1 + 4 * kq) + 4 * px;
                } else {
                    o0[t] = (((long long)b * CO + n) * OH + (y0 + r0)) * OW + x0 + 16 * c0 + 4 * kq;
                    o1[t] = (((long long)b * CO + n) * OH + (y0 + r1)) * OW + x0 + 16 * c1 + 4 * kq;
                }
                e0[t] = epilogue_loads(ep, o0[t], n < NOUT);
                e1[t] = epilogue_loads(ep, o1[t], n < NOUT);
            }
            f32x4 acc0[NT], acc1[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) { acc0[t] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc1[t] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const int cg4 = s / TAPS, tap = s % TAPS;
                const int tyy = TAPS == 9 ? tap / 3 : 0, txx = TAPS == 9 ? tap % 3 : 0;
                const int o = 4 * cg4 * PS + tyy * RS + txx;
                const float a0 = a0p[o], a1 = a1p[o];
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    acc0[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, wreg[t][s], acc0[t], 0, 0, 0);
                    acc1[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, wreg[t][s], acc1[t], 0, 0, 0);
                }
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int tt = pass * NT + t;
                const int n = 16 * tt + m;
                const bool valid = n < NOUT;
#pragma unroll
                for (int which = 0; which < 2; ++which) {
                    f32x4 v = bias_relu(which ? acc1[t] : acc0[t], ep, bias[t]);
                    if (PIX) {
                        // partner lane (n ^ 1) holds the other x-phase of the same output row
                        f32x4 pv;
                        pv.x = __shfl_xor(v.x, 1, 64); pv.y = __shfl_xor(v.y, 1, 64);
                        pv.z = __shfl_xor(v.z, 1, 64); pv.w = __shfl_xor(v.w, 1, 64);
                        v = (n & 1) ? (f32x4){pv.z, v.z, pv.w, v.w} : (f32x4){v.x, pv.x, v.y, pv.y};
                    }
                    if (valid)
                        epilogue_tail(v, ep, which ? e1[t] : e0[t], mc0[t], mc2[t], out, which ? o1[t] : o0[t],
                                      s1[tt], s2[tt]);
                }
            }
        }
    }
    if (ep.stats) stats_reduce<NTT, PIX>(s1, s2, s_stat, ep, CO);
}

// ------------------------------------------------------------------------------ dispatch
struct ConvArgs {
    Operand in; WeightView wv; float *out; Epilogue ep;
    int B, Cphys, CIN, NOUT, H, W;
    hipStream_t stream;
};

int conv4_tw(int CIN, int Wo)
{
    const int cap = CIN <= 5 ? 64 : (CIN <= 8 ? 32 : 16);
    return Wo < cap ? Wo : cap;
}

template <int CIN, int TW>
int launch_conv4(const ConvArgs &a)
{
    constexpr int TH = 8;
    const int grid = a.B * ((a.H / 2) / TH) * ((a.W / 2) / TW);
    if (a.in.mode == DM_LOAD_AFFINE2)
        hipLaunchKernelGGL((conv4x4s2_kernel<CIN, 1, TH, TW, true>), dim3(grid), dim3(DM_BLOCK), 0, a.stream,
                           a.in, a.wv, a.out, a.ep, a.Cphys, a.NOUT, a.H, a.W);
    else
        hipLaunchKernelGGL((conv4x4s2_kernel<CIN, 1, TH, TW, false>), dim3(grid), dim3(DM_BLOCK), 0, a.stream,
                           a.in, a.wv, a.out, a.ep, a.Cphys, a.NOUT, a.H, a.W);
    return 0;
}

int conv3_tw(int W) { return W < 64 ? W : 64; }
int conv3_th(int TW) { return TW == 16 ? 16 : 8; }

template <int CIN, int NT, int NPASS, int TAPS, bool PIX, int TW>
int launch_conv3(const ConvArgs &a)
{
    constexpr int TH = TW == 16 ? 16 : 8;
    const int grid = a.B * (a.H / TH) * (a.W / TW);
    if (a.in.mode == DM_LOAD_AFFINE2)
        hipLaunchKernelGGL((conv3x3_kernel<CIN, NT, NPASS, TAPS, PIX, TH, TW, true>), dim3(grid), dim3(DM_BLOCK), 0,
                           a.stream, a.in, a.wv, a.out, a.ep, a.Cphys, a.NOUT, a.H, a.W);
    else
        hipLaunchKernelGGL((conv3x3_kernel<CIN, NT, NPASS, TAPS, PIX, TH, TW, false>), dim3(grid), dim3(DM_BLOCK), 0,
                           a.stream, a.in, a.wv, a.out, a.ep, a.Cphys, a.NOUT, a.H, a.W);
    return 0;
}

}  // namespace

// =============================================================================== C ABI
static int conv_common_checks(const char *who, const dm_operand *in, const dm_weight_view *w, float *out,
                              const dm_epilogue *ep, int B, int CIN, int NOUT, int H, int W)
{
    if (dm_check_operand(in, who)) return -1;
    DM_REQUIRE(w && w->w && out, "%s: NULL weight or output", who);
    DM_REQUIRE(B > 0 && CIN > 0 && NOUT > 0 && H > 0 && W > 0, "%s: bad shape", who);
    DM_REQUIRE(CIN <= DM_COEF_MAX_C, "%s: more than %d input channels", who, DM_COEF_MAX_C);
    DM_REQUIRE(CIN - (in->ones_channel ? 1 : 0) > 0, "%s: no physical input channel", who);
    if (ep && ep->mask.p0 && dm_check_operand(&ep->mask, who)) return -1;
    DM_REQUIRE(!(ep && ep->mask.p0 && ep->mask.ones_channel), "%s: mask operand cannot have a ones channel", who);
    return 0;
}

extern "C" int dm_conv4x4s2_num_blocks(int B, int CIN, int NOUT, int H, int W)
{
    (void)NOUT;
    const int Wo = W / 2, Ho = H / 2;
    const int TW = conv4_tw(CIN, Wo);
    if (TW <= 0 || Ho % 8 || Wo % TW) return -1;
    return B * (Ho / 8) * (Wo / TW);
}

extern "C" int dm_conv4x4s2(const dm_operand *in, const dm_weight_view *w, float *out, const dm_epilogue *ep,
                            int B, int CIN, int NOUT, int H, int W, void *stream)
{
    if (conv_common_checks("dm_conv4x4s2", in, w, out, ep, B, CIN, NOUT, H, W)) return -1;
    DM_REQUIRE(H % 16 == 0 && W % 32 == 0, "dm_conv4x4s2: H must be a multiple of 16 and W of 32 (got %dx%d)", H, W);
    DM_REQUIRE(NOUT <= 16, "dm_conv4x4s2: NOUT %d > 16 not built", NOUT);
    const int Wo = W / 2;
    const int TW = conv4_tw(CIN, Wo);
    DM_REQUIRE(TW == 16 || TW == 32 || TW == 64, "dm_conv4x4s2: output width %d not tileable", Wo);
    DM_REQUIRE(Wo % TW == 0, "dm_conv4x4s2: output width %d not a multiple of tile %d", Wo, TW);
    ConvArgs a{to_dev(in), to_dev(w), out, to_dev(ep), B, CIN - (in->ones_channel ? 1 : 0), CIN, NOUT, H, W,
               (hipStream_t)stream};
#define DM_C4(C, T) if (CIN == C && TW == T) { launch_conv4<C, T>(a); return dm_launch_status("dm_conv4x4s2"); }
    DM_C4(3, 64) DM_C4(3, 32) DM_C4(3, 16)
    DM_C4(4, 64) DM_C4(4, 32) DM_C4(4, 16)
    DM_C4(5, 64) DM_C4(5, 32) DM_C4(5, 16)
    DM_C4(2, 64) DM_C4(2, 32) DM_C4(2, 16)
    DM_C4(8, 32) DM_C4(8, 16)
    DM_C4(16, 16)
#undef DM_C4
    dm_set_error("dm_conv4x4s2: no kernel built for CIN=%d (tile width %d)", CIN, TW);
    return -1;
}

extern "C" int dm_conv3x3_num_blocks(int B, int CIN, int NOUT, int H, int W, int taps, int pixel_shuffle)
{
    (void)CIN; (void)NOUT; (void)taps; (void)pixel_shuffle;
    const int TW = conv3_tw(W), TH = conv3_th(TW);
    if (TW <= 0 || H % TH || W % TW) return -1;
    return B * (H / TH) * (W / TW);
}

extern "C" int dm_conv3x3(const dm_operand *in, const dm_weight_view *w, float *out, const dm_epilogue *ep,
                          int B, int CIN, int NOUT, int H, int W, int taps, int pixel_shuffle, void *stream)
{
    if (conv_common_checks("dm_conv3x3", in, w, out, ep, B, CIN, NOUT, H, W)) return -1;
    DM_REQUIRE(taps == 9 || taps == 1, "dm_conv3x3: taps must be 9 or 1");
    DM_REQUIRE(!pixel_shuffle || (taps == 9 && NOUT % 4 == 0), "dm_conv3x3: pixel_shuffle needs taps=9, NOUT%%4==0");
    DM_REQUIRE(!in->ones_channel, "dm_conv3x3: ones_channel not supported");
    const int TW = conv3_tw(W), TH = conv3_th(TW);
    DM_REQUIRE((TW == 16 || TW == 32 || TW == 64) && W % TW == 0 && H % TH == 0,
               "dm_conv3x3: spatial size %dx%d not tileable", H, W);
    ConvArgs a{to_dev(in), to_dev(w), out, to_dev(ep), B, CIN, CIN, NOUT, H, W, (hipStream_t)stream};
    const int NTT = (NOUT + 15) / 16;
    const bool pix = pixel_shuffle != 0;
#define DM_C3(C, NTOT, NT_, NP_, TP, PX, T)                                                \
    if (CIN == C && NTT == NTOT && taps == TP && pix == PX && TW == T) {                   \
        launch_conv3<C, NT_, NP_, TP, PX, T>(a);                                           \
        return dm_launch_status("dm_conv3x3");                                             \
    }
    // 3x3 plain: enc.10, residual 3x3 and their data gradients
    DM_C3(16, 1, 1, 1, 9, false, 16) DM_C3(16, 1, 1, 1, 9, false, 32)
    DM_C3(16, 2, 2, 1, 9, false, 16) DM_C3(16, 2, 2, 1, 9, false, 32)
    DM_C3(32, 1, 1, 1, 9, false, 16) DM_C3(32, 1, 1, 1, 9, false, 32)
    // 1x1: residual 1x1 and its data gradient
    DM_C3(32, 1, 1, 1, 1, false, 16) DM_C3(32, 1, 1, 1, 1, false, 32)
    DM_C3(16, 2, 2, 1, 1, false, 16) DM_C3(16, 2, 2, 1, 1, false, 32)
    // pixel shuffle: ConvTranspose2d forward (dec.0/2/4) and data gradients of enc.4 / enc.7
    DM_C3(16, 2, 2, 1, 9, true, 16) DM_C3(16, 2, 2, 1, 9, true, 32) DM_C3(16, 2, 2, 1, 9, true, 64)
    DM_C3(16, 4, 2, 2, 9, true, 16) DM_C3(16, 4, 2, 2, 9, true, 32)
    DM_C3(8, 1, 1, 1, 9, true, 32) DM_C3(8, 1, 1, 1, 9, true, 64)
    DM_C3(4, 1, 1, 1, 9, true, 64)
#undef DM_C3
    dm_set_error("dm_conv3x3: no kernel built for CIN=%d NOUT=%d taps=%d pixel_shuffle=%d width=%d", CIN, NOUT, taps,
                 pixel_shuffle, W);
    return -1;
}

// conv_wide.hip -- MFMA convolution / weight-gradient kernels for arbitrary channel counts.
//
// conv_mfma.hip / wgrad_mfma.hip keep a layer's whole weight tensor in registers, which only works for the thin
// default family (num_hiddens 16).  The reference's example configuration (config_example.yml: VQ_VAE_z32 with
// num_hiddens 64, num_residual_hiddens 64, 512 codes) has 64 -> 64 channel 3x3 layers: 147 KB of weights, MFMA bound.
// These kernels are the classic implicit GEMM for that regime:
//   convolution     M = 8 x 16 pixels of one sample, N = up to 128 output channels, K = (tap, channel) in chunks of
//                   8 channels; the input chunk (operand transform, zero padding, ones channel applied) and the weight
//                   chunk (re-laid as [tap][channel][n]) are staged in LDS, every wave owns 2 pixel rows x all N;
//   weight gradient M = 64 S channels (one 16-row tile per wave), N = (T channel, tap) flattened, K = the 128 pixels
//                   of a tile; both operand tiles in LDS, deterministic slabs as everywhere else.
// Same operands / epilogue / statistics-slab semantics as the thin kernels (dm_operand, dm_weight_view, dm_epilogue);
// the fp32 MFMA (v_mfma_f32_16x16x4_f32) keeps full precision.  LDS strides are chosen so that the 4 x 16 lane groups
// of an MFMA operand read fall into distinct banks (channel stride = 16 mod 64 floats for conv, 4 mod 64 for wgrad).
#include "dm_common.h"

namespace {

enum { W_S2 = 0, W_S1 = 1, W_PIX = 2 };
constexpr int WKC = 8;                 // input channels per K chunk
constexpr int WIDE_MAX_BLOCKS = 768;   // persistent grid cap (x dimension)

// operand value at (b, c, y, x): zero outside the image, ones channel (c >= Cphys) is 1 inside it
__device__ __forceinline__ float wide_load(const Operand &op, int b, int c, int y, int x, int Cphys, int H, int W)
{
    if (y < 0 || y >= H || x < 0 || x >= W) return 0.f;
    if (c >= Cphys) return 1.f;
    const long long off = (((long long)b * Cphys + c) * H + y) * W + x;
    float v = op.p0[off];
    if (op.mode == DM_LOAD_IDENT) return v;
    if (op.mode == DM_LOAD_RELU) return v < 0.f ? 0.f : v;
    const float *cf = op.coef + (long long)b * op.coef_bstride + c * 4;
    if (op.mode == DM_LOAD_AFFINE2) return cf[0] * v + (cf[1] * op.p1[off] + cf[2]);
    v = cf[0] * v + cf[2];
    if (op.mode == DM_LOAD_AFFINE_RELU) v = v < 0.f ? 0.f : v;
    return v;
}

template <int FORM, int TAPS>
struct WideGeom {
    static constexpr int S = FORM == W_S2 ? 2 : 1;                       // stride
    static constexpr int R = FORM == W_S2 ? 1 : (TAPS == 9 ? 1 : 0);     // halo
    static constexpr int T = FORM == W_S2 ? 16 : TAPS;                   // taps of the K loop
    static constexpr int ROWS = FORM == W_S2 ? 18 : 8 + 2 * R;           // input rows of a tile
    static constexpr int LCOLS = FORM == W_S2 ? 34 : 16 + 2 * R;         // logical input columns of a tile
    static constexpr int COLS = FORM == W_S2 ? 17 : LCOLS;               // columns of an LDS row (S2: one parity plane)
    static constexpr int PLS = ROWS * COLS;                              // plane size (S2)
    static constexpr int RAW = FORM == W_S2 ? 2 * PLS : PLS;
    static constexpr int CHS = ((RAW - 16 + 63) / 64) * 64 + 16;         // channel stride, = 16 (mod 64)
};

// ---------------------------------------------------------------------------------------------- convolution
// grid (x: persistent over tiles or samples, y: passes of 16*NPW output channels).
//   per_tile == 0: workgroup x walks tiles x, x+gx, ...; statistics of all of them -> slab x; slabs >= gx are zeroed.
//   per_tile != 0: workgroup x walks samples x, x+gx, ...; statistics of a sample -> slab b*(nslabs/B), the sample's
//                  other slabs are zeroed (per-sample BatchNorm sums the slabs of a sample).
template <int FORM, int TAPS, int NPW>
__global__ __launch_bounds__(256) void conv_wide_kernel(Operand in, WeightView wv, float *__restrict__ out, Epilogue ep,
                                                        int B, int Cphys, int CIN, int NOUT, int H, int W, int nslabs,
                                                        int per_tile)
{
    using G = WideGeom<FORM, TAPS>;
    constexpr int NS = NPW == 1 ? 16 : 16 * NPW + 16;     // weight row stride (floats), distinct banks for the 4 k lanes
    constexpr int NPASS = 16 * NPW;
    __shared__ __attribute__((aligned(16))) float s_in[WKC * G::CHS];
    __shared__ __attribute__((aligned(16))) float s_w[G::T * WKC * NS];
    __shared__ double s_red[4 * NPW * 16 * 2];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, p = lane & 15, kq = lane >> 4;
    const int CO = FORM == W_PIX ? NOUT >> 2 : NOUT;
    const int BH = FORM == W_S2 ? H >> 1 : H, BW = FORM == W_S2 ? W >> 1 : W;       // base grid the tiles cover
    const int OH = FORM == W_PIX ? 2 * H : BH, OW = FORM == W_PIX ? 2 * W : BW;
    const int tx_n = BW >> 4, tps = (BH >> 3) * tx_n;
    const int n0 = blockIdx.y * NPASS;
    const int nchunks = (CIN + WKC - 1) / WKC;
    const int spg = per_tile ? nslabs / B : 1;
    const int ngroups = per_tile ? B : 1;

    for (int g = per_tile ? blockIdx.x : 0; g < ngroups; g += per_tile ? gridDim.x : 1) {
        double st1[NPW], st2[NPW];
#pragma unroll
        for (int t = 0; t < NPW; ++t) { st1[t] = 0.0; st2[t] = 0.0; }
        const int t_begin = per_tile ? g * tps : blockIdx.x, t_end = per_tile ? (g + 1) * tps : B * tps;
        const int t_step = per_tile ? 1 : gridDim.x;
        for (int tile = t_begin; tile < t_end; tile += t_step) {
            const int b = tile / tps, r = tile - b * tps;
            const int y0 = (r / tx_n) << 3, x0 = (r - (r / tx_n) * tx_n) << 4;
            f32x4 acc[2][NPW];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int t = 0; t < NPW; ++t) acc[i][t] = (f32x4){0.f, 0.f, 0.f, 0.f};

            for (int ch = 0; ch < nchunks; ++ch) {
                const int c0 = ch * WKC;
                __syncthreads();
                // ---- input chunk: WKC channels x ROWS x LCOLS, transform + padding applied
                for (int idx = tid; idx < WKC * G::ROWS * G::LCOLS; idx += 256) {
                    const int c = idx / (G::ROWS * G::LCOLS), rem = idx - c * (G::ROWS * G::LCOLS);
                    const int iy = rem / G::LCOLS, ix = rem - iy * G::LCOLS;
                    float v = 0.f;
                    if (c0 + c < CIN) v = wide_load(in, b, c0 + c, G::S * y0 - G::R + iy, G::S * x0 - G::R + ix, Cphys, H, W);
                    const int a = FORM == W_S2 ? c * G::CHS + (ix & 1) * G::PLS + iy * G::COLS + (ix >> 1)
                                               : c * G::CHS + iy * G::COLS + ix;
                    s_in[a] = v;
                }
                // ---- weight chunk as [tap][channel][n]
                for (int idx = tid; idx < G::T * WKC * NPASS; idx += 256) {
                    const int tap = idx % G::T, cl = (idx / G::T) % WKC, nl = idx / (G::T * WKC);
                    const int n = n0 + nl, c = c0 + cl;
                    float v = 0.f;
                    if (n < NOUT && c < CIN) {
                        if (FORM == W_PIX) {
                            const int co = n >> 2, py = (n >> 1) & 1, px = n & 1, tyy = tap / 3, txx = tap - tyy * 3;
                            const int da = tyy - py, db = txx - px;
                            if (da >= 0 && da <= 1 && db >= 0 && db <= 1)
                                v = wv.w[wv.off + co * wv.sn + c * wv.sc + (py + 3 - 2 * tyy) * wv.sky + (px + 3 - 2 * txx) * wv.skx];
                        } else {
                            constexpr int KW = FORM == W_S2 ? 4 : (TAPS == 9 ? 3 : 1);
                            const int ky = tap / KW, kx = tap - ky * KW;
                            v = wv.w[wv.off + n * wv.sn + c * wv.sc + ky * wv.sky + kx * wv.skx];
                        }
                    }
                    s_w[(tap * WKC + cl) * NS + nl] = v;
                }
                __syncthreads();
                // ---- MFMAs: wave owns base rows 2*wave, 2*wave+1
#pragma unroll
                for (int tap = 0; tap < G::T; ++tap) {
                    int toff;
                    if (FORM == W_S2) toff = (tap & 1) * G::PLS + (tap >> 2) * G::COLS + ((tap & 3) >> 1);
                    else toff = TAPS == 9 ? (tap / 3) * G::COLS + (tap % 3) : 0;
#pragma unroll
                    for (int cq = 0; cq < WKC / 4; ++cq) {
                        float av[2], bv[NPW];
#pragma unroll
                        for (int i = 0; i < 2; ++i)
                            av[i] = s_in[(cq * 4 + kq) * G::CHS + toff + (G::S * (2 * wave + i)) * G::COLS + p];
#pragma unroll
                        for (int t = 0; t < NPW; ++t) bv[t] = s_w[(tap * WKC + cq * 4 + kq) * NS + t * 16 + p];
#pragma unroll
                        for (int t = 0; t < NPW; ++t)
#pragma unroll
                            for (int i = 0; i < 2; ++i)
                                acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bv[t], acc[i][t], 0, 0, 0);
                    }
                }
            }

            // ---- epilogue: lane holds pixels (row 2*wave+i, columns 4*kq .. 4*kq+3) of channel n0 + 16*t + p
#pragma unroll
            for (int t = 0; t < NPW; ++t) {
                const int n = n0 + t * 16 + p;
                const bool live = n < NOUT;
                const int chn = FORM == W_PIX ? n >> 2 : n;
                const float bias = (live && ep.bias) ? ep.bias[chn] : 0.f;
                float mc0 = 1.f, mc2 = 0.f;
                if (live && ep.mask.p0 && ep.mask.mode >= DM_LOAD_AFFINE) {
                    const float *cf = ep.mask.coef + (long long)b * ep.mask.coef_bstride + chn * 4;
                    mc0 = cf[0]; mc2 = cf[2];
                }
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    f32x4 v = acc[i][t];
                    int oy, ox;
                    if (FORM == W_PIX) {
                        // pair (px = 0, 1) lanes exchange so that each writes 4 consecutive output pixels
                        const f32x4 o = lane_xor1(v);
                        const bool odd = n & 1;
                        v = odd ? (f32x4){o.z, v.z, o.w, v.w} : (f32x4){v.x, o.x, v.y, o.y};
                        oy = 2 * (y0 + 2 * wave + i) + ((n >> 1) & 1);
                        ox = 2 * (x0 + 4 * kq) + (odd ? 4 : 0);
                    } else {
                        oy = y0 + 2 * wave + i;
                        ox = x0 + 4 * kq;
                    }
                    if (!live) continue;
                    if (FORM == W_S2 && ep.bias_border) {
                        const int ry = oy == 0 ? 0 : (oy == OH - 1 ? 2 : 1);
                        const float *tb = ep.bias_border + (ry * 3) * CO + chn;
                        const float mid = tb[CO];
                        v.x += ox == 0 ? tb[0] : mid;
                        v.y += mid;
                        v.z += mid;
                        v.w += ox + 3 == OW - 1 ? tb[2 * CO] : mid;
                    } else {
                        v += bias;
                    }
                    if (ep.relu) {
                        v.x = v.x < 0.f ? 0.f : v.x; v.y = v.y < 0.f ? 0.f : v.y;
                        v.z = v.z < 0.f ? 0.f : v.z; v.w = v.w < 0.f ? 0.f : v.w;
                    }
                    const long long o = (((long long)b * CO + chn) * OH + oy) * OW + ox;
                    if (ep.mask.p0) {
                        const f32x4 m = *reinterpret_cast<const f32x4 *>(ep.mask.p0 + o);
                        v.x = (mc0 * m.x + mc2) > 0.f ? v.x : 0.f; v.y = (mc0 * m.y + mc2) > 0.f ? v.y : 0.f;
                        v.z = (mc0 * m.z + mc2) > 0.f ? v.z : 0.f; v.w = (mc0 * m.w + mc2) > 0.f ? v.w : 0.f;
                    }
                    if (ep.resid) v += *reinterpret_cast<const f32x4 *>(ep.resid + o);
                    *reinterpret_cast<f32x4 *>(out + o) = v;
                    if (ep.stats) {
                        f32x4 q = v;
                        if (ep.stat_q) q = *reinterpret_cast<const f32x4 *>(ep.stat_q + o);
                        st1[t] += (double)((v.x + v.y) + (v.z + v.w));
                        st2[t] += (double)((v.x * q.x + v.y * q.y) + (v.z * q.z + v.w * q.w));
                    }
                }
            }
        }

        // ---- statistics of this group -> one slab
        if (ep.stats) {
            __syncthreads();
#pragma unroll
            for (int t = 0; t < NPW; ++t) {
                double a = st1[t], c = st2[t];
                a += __shfl_xor(a, 16, 64); a += __shfl_xor(a, 32, 64);
                c += __shfl_xor(c, 16, 64); c += __shfl_xor(c, 32, 64);
                if (kq == 0) {
                    s_red[((wave * NPW + t) * 16 + p) * 2 + 0] = a;
                    s_red[((wave * NPW + t) * 16 + p) * 2 + 1] = c;
                }
            }
            __syncthreads();
            const long long slab = per_tile ? (long long)g * spg : blockIdx.x;
            constexpr int CPP = FORM == W_PIX ? 4 * NPW : 16 * NPW;        // statistics channels of one pass
            for (int i = tid; i < CPP * 2; i += 256) {
                const int cl = i >> 1, k = i & 1;
                const int chn = (FORM == W_PIX ? n0 >> 2 : n0) + cl;
                if (chn >= CO) continue;
                double s = 0.0;
                for (int w = 0; w < 4; ++w) {
                    if (FORM == W_PIX) {
                        const int t = cl >> 2, pp = (cl & 3) * 4;
                        for (int j = 0; j < 4; ++j) s += s_red[((w * NPW + t) * 16 + pp + j) * 2 + k];
                    } else {
                        s += s_red[((w * NPW + (cl >> 4)) * 16 + (cl & 15)) * 2 + k];
                    }
                }
                ep.stats[(slab * CO + chn) * 2 + k] = s;
                if (per_tile)
                    for (int e = 1; e < spg; ++e) ep.stats[((slab + e) * CO + chn) * 2 + k] = 0.0;
            }
        }
    }
    if (ep.stats && !per_tile) {                                      // slabs no workgroup owns
        constexpr int CPP = FORM == W_PIX ? 4 * NPW : 16 * NPW;
        const int cbase = FORM == W_PIX ? n0 >> 2 : n0;
        for (int sl = blockIdx.x + gridDim.x; sl < nslabs; sl += gridDim.x)
            for (int i = tid; i < CPP * 2; i += 256)
                if (cbase + (i >> 1) < CO) ep.stats[((long long)sl * CO + cbase + (i >> 1)) * 2 + (i & 1)] = 0.0;
    }
}

// ---------------------------------------------------------------------------------------------- weight gradient
template <int KK>
struct WgGeom {
    static constexpr int S = KK == 4 ? 2 : 1;
    static constexpr int R = KK == 1 ? 0 : 1;
    static constexpr int T2 = KK * KK;
    static constexpr int ROWS = KK == 4 ? 18 : 8 + 2 * R;
    static constexpr int LCOLS = KK == 4 ? 34 : 16 + 2 * R;
    static constexpr int COLS = KK == 4 ? 17 : LCOLS;
    static constexpr int PLS = ROWS * COLS;
    static constexpr int RAW = KK == 4 ? 2 * PLS : PLS;
    static constexpr int CTS = ((RAW - 4 + 63) / 64) * 64 + 4;           // T channel stride, = 4 (mod 64)
    static constexpr int NTW = KK == 1 ? 4 : 8;                          // N tiles (of 16) per pass
    static constexpr int NCTP = 16 * NTW / T2;                           // T channels per pass: 64 / 14 / 8
    static constexpr int ROWSTEP = S * COLS;                             // LDS step of one S-grid row
};
constexpr int WG_CSS = 132;            // S channel stride (128 pixels, = 4 mod 64)

// grid (x: persistent over (sample, tile) units -> slab x, y: passes of NCTP T channels, z: passes of 64 S channels)
template <int KK>
__global__ __launch_bounds__(256) void wgrad_wide_kernel(Operand S, Operand T, float *__restrict__ slabs, int B, int CS,
                                                         int CT, int CTphys, int Hs, int Ws, int nslabs)
{
    using G = WgGeom<KK>;
    __shared__ __attribute__((aligned(16))) float s_S[64 * WG_CSS];
    __shared__ __attribute__((aligned(16))) float s_T[G::NCTP * G::CTS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, p = lane & 15, kq = lane >> 4;
    const int ct0 = blockIdx.y * G::NCTP, cs0 = blockIdx.z * 64;
    const int nct = CT - ct0 < G::NCTP ? CT - ct0 : G::NCTP;           // T channels of this pass
    const int ntw = (nct * G::T2 + 15) >> 4;                          // N tiles in use
    const int Ht = Hs * G::S, Wt = Ws * G::S;
    const int tx_n = Ws >> 4, tps = (Hs >> 3) * tx_n;
    const long long E = (long long)CS * CT * G::T2;

    int boff[G::NTW];
#pragma unroll
    for (int nt = 0; nt < G::NTW; ++nt) {
        const int n = nt * 16 + p, ctl = n / G::T2, tap = n - ctl * G::T2;
        const int ky = tap / KK, kx = tap - ky * KK;
        int o = ctl * G::CTS + (KK == 4 ? (kx & 1) * G::PLS + ky * G::COLS + (kx >> 1) : ky * G::COLS + kx);
        boff[nt] = ctl < nct ? o : 0;
    }
    f32x4 acc[G::NTW];
#pragma unroll
    for (int nt = 0; nt < G::NTW; ++nt) acc[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int units = B * tps;
    for (int u = blockIdx.x; u < units; u += gridDim.x) {
        const int b = u / tps, r = u - b * tps;
        const int y0 = (r / tx_n) << 3, x0 = (r - (r / tx_n) * tx_n) << 4;
        __syncthreads();
        // S tile: 64 channels x 8 rows x 16 columns as float4
        for (int idx = tid; idx < 64 * 32; idx += 256) {
            const int cl = idx >> 5, rr = (idx >> 2) & 7, c4 = idx & 3;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (cs0 + cl < CS) {
                const long long off = (((long long)b * CS + cs0 + cl) * Hs + y0 + rr) * Ws + x0 + c4 * 4;
                v = operand_load4(S, off, b, cs0 + cl);
            }
            *reinterpret_cast<f32x4 *>(&s_S[cl * WG_CSS + rr * 16 + c4 * 4]) = v;
        }
        // T tile with halo
        for (int idx = tid; idx < nct * G::ROWS * G::LCOLS; idx += 256) {
            const int c = idx / (G::ROWS * G::LCOLS), rem = idx - c * (G::ROWS * G::LCOLS);
            const int iy = rem / G::LCOLS, ix = rem - iy * G::LCOLS;
            const float v = wide_load(T, b, ct0 + c, G::S * y0 - G::R + iy, G::S * x0 - G::R + ix, CTphys, Ht, Wt);
            const int a = KK == 4 ? c * G::CTS + (ix & 1) * G::PLS + iy * G::COLS + (ix >> 1) : c * G::CTS + iy * G::COLS + ix;
            s_T[a] = v;
        }
        __syncthreads();
#pragma unroll 4
        for (int ks = 0; ks < 32; ++ks) {
            const float av = s_S[(wave * 16 + p) * WG_CSS + ks * 4 + kq];
            const int po = (ks >> 2) * G::ROWSTEP + (ks & 3) * 4 + kq;
#pragma unroll
            for (int nt = 0; nt < G::NTW; ++nt)
                if (nt < ntw) {
                    const float bv = s_T[boff[nt] + po];
                    acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[nt], 0, 0, 0);
                }
        }
    }

    // lane holds R[cs = cs0 + 16*wave + 4*kq + j][n = 16*nt + p]
    float *row = slabs + (long long)blockIdx.x * E;
#pragma unroll
    for (int nt = 0; nt < G::NTW; ++nt) {
        const int n = nt * 16 + p, ctl = n / G::T2, tap = n - ctl * G::T2;
        if (nt >= ntw || ctl >= nct) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int cs = cs0 + wave * 16 + kq * 4 + j;
            if (cs < CS) row[((long long)cs * CT + ct0 + ctl) * G::T2 + tap] = acc[nt][j];
        }
    }
    if (blockIdx.y == 0 && blockIdx.z == 0)
        for (int sl = blockIdx.x + gridDim.x; sl < nslabs; sl += gridDim.x)
            for (long long e = tid; e < E; e += 256) slabs[(long long)sl * E + e] = 0.f;
}

}  // namespace

// ---- entry points used by the dispatchers in conv_mfma.hip / wgrad_mfma.hip (not part of the public header) ----------
// base grid (output pixels for the strided / plain forms, input pixels for the transposed form) must tile by 8 x 16
bool dm_wide_conv_ok(int form, int H, int W)
{
    const int BH = form == W_S2 ? H / 2 : H, BW = form == W_S2 ? W / 2 : W;
    return BH > 0 && BW > 0 && BH % 8 == 0 && BW % 16 == 0 && (form != W_S2 || (H % 2 == 0 && W % 2 == 0));
}

int dm_wide_conv_slabs(int form, int B, int H, int W, int per_tile)
{
    if (per_tile) return B;
    const int BH = form == W_S2 ? H / 2 : H, BW = form == W_S2 ? W / 2 : W;
    const long long nt = (long long)B * (BH / 8) * (BW / 16);
    return (int)(nt < WIDE_MAX_BLOCKS ? nt : WIDE_MAX_BLOCKS);
}

int dm_wide_conv(int form, const Operand &in, const WeightView &wv, float *out, const Epilogue &ep, int B, int Cphys, int CIN,
                 int NOUT, int H, int W, int taps, int nslabs, int per_tile, hipStream_t st)
{
    const int BH = form == W_S2 ? H / 2 : H, BW = form == W_S2 ? W / 2 : W;
    const long long ntiles = (long long)B * (BH / 8) * (BW / 16);
    long long gx = per_tile ? B : ntiles;
    if (gx > WIDE_MAX_BLOCKS) gx = WIDE_MAX_BLOCKS;
    if (ep.stats && !per_tile && gx > nslabs) gx = nslabs;
    const int maxnp = form == W_S2 ? 4 : 8;
    int np = NOUT <= 16 ? 1 : (NOUT <= 32 ? 2 : (NOUT <= 64 ? 4 : 8));
    if (np > maxnp) np = maxnp;
    const dim3 grid((unsigned)gx, (unsigned)((NOUT + 16 * np - 1) / (16 * np)));
#define DM_WL(F, TP, NP_)                                                                                           \
    hipLaunchKernelGGL((conv_wide_kernel<F, TP, NP_>), grid, dim3(256), 0, st, in, wv, out, ep, B, Cphys, CIN, NOUT, \
                       H, W, nslabs, per_tile)
#define DM_WN(F, TP)                                                              \
    switch (np) {                                                                 \
    case 1: DM_WL(F, TP, 1); break;                                               \
    case 2: DM_WL(F, TP, 2); break;                                               \
    case 4: DM_WL(F, TP, 4); break;                                               \
    default: DM_WL(F, TP, 8); break;                                              \
    }
    if (form == W_S2) {
        switch (np) {
        case 1: DM_WL(W_S2, 16, 1); break;
        case 2: DM_WL(W_S2, 16, 2); break;
        default: DM_WL(W_S2, 16, 4); break;
        }
    } else if (form == W_PIX) {
        DM_WN(W_PIX, 9)
    } else if (taps == 9) {
        DM_WN(W_S1, 9)
    } else {
        DM_WN(W_S1, 1)
    }
#undef DM_WN
#undef DM_WL
    return 0;
}

bool dm_wide_wgrad_ok(int Hs, int Ws) { return Hs > 0 && Ws > 0 && Hs % 8 == 0 && Ws % 16 == 0; }

static void wide_wgrad_grid(int CS, int CT, int k, int &gy, int &gz, int &cap)
{
    const int nctp = k == 4 ? WgGeom<4>::NCTP : (k == 3 ? WgGeom<3>::NCTP : WgGeom<1>::NCTP);
    gy = (CT + nctp - 1) / nctp;
    gz = (CS + 63) / 64;
    cap = 1024 / (gy * gz);
    if (cap < 32) cap = 32;
    if (cap > 512) cap = 512;
}

int dm_wide_wgrad_slabs(int B, int CS, int CT, int Hs, int Ws, int k)
{
    int gy, gz, cap;
    wide_wgrad_grid(CS, CT, k, gy, gz, cap);
    const long long units = (long long)B * (Hs / 8) * (Ws / 16);
    return (int)(units < cap ? units : cap);
}

int dm_wide_wgrad(const Operand &S, const Operand &T, float *slabs, int B, int CS, int CT, int CTphys, int Hs, int Ws,
                  int k, int nslabs, hipStream_t st)
{
    int gy, gz, cap;
    wide_wgrad_grid(CS, CT, k, gy, gz, cap);
    int gx = dm_wide_wgrad_slabs(B, CS, CT, Hs, Ws, k);
    if (gx > nslabs) gx = nslabs;
    const dim3 grid((unsigned)gx, (unsigned)gy, (unsigned)gz);
    if (k == 4)
        hipLaunchKernelGGL((wgrad_wide_kernel<4>), grid, dim3(256), 0, st, S, T, slabs, B, CS, CT, CTphys, Hs, Ws, nslabs);
    else if (k == 3)
        hipLaunchKernelGGL((wgrad_wide_kernel<3>), grid, dim3(256), 0, st, S, T, slabs, B, CS, CT, CTphys, Hs, Ws, nslabs);
    else
        hipLaunchKernelGGL((wgrad_wide_kernel<1>), grid, dim3(256), 0, st, S, T, slabs, B, CS, CT, CTphys, Hs, Ws, nslabs);
    return 0;
}

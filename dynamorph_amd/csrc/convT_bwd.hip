// convT_bwd.hip -- backward of a thin ConvTranspose2d(CI -> CO, 4, stride 2, padding 1): input AND weight gradient in one pass.
//
// Reference: dec.0 = ConvTranspose2d(16, 8, 4, 2, 1) and dec.2 = ConvTranspose2d(8, 4, 4, 2, 1) of VQ_VAE.dec
// (HiddenStateExtractor/vq_vae.py:291-296) as autograd differentiates them for total_loss.backward()
// (run_training.py:406): aten::convolution_backward (input and weight) + the ReLU mask of the layer below + the channel sums
// that are the previous layer's bias gradient.
//
// As two kernels (rounds 1-3: conv4x4s2_kernel for the input gradient, wgrad_kernel / wgrad_ys_kernel for the weights) each
// layer read its output gradient G and its input S twice, and all four kernels sit at the HBM roof (dec.2: 268 MB in 56 us +
// 201 MB in 42 us).  Here one staging per tile of 8 x TW input positions,
//     G   [CO][18][2 TW + 8]   the output gradient with its halo (rows 2 y0 - 1 .., columns 2 x0 - 4 ..; zero outside the image)
//     S   [CI][8][TW]          the layer input (what the forward multiplied: post-ReLU where there is one)
// feeds both products on v_mfma_f32_16x16x4_f32:
//     input gradient   gin[ci][y][x] = [S > 0] * sum_{co,ky,kx} G[co][2y+ky-1][2x+kx-1] * W[ci][co][ky][kx]
//                      M = 16 positions of a row, N = ci, K = (co, ky | kx): the four kx of one (co, ky) are one K step
//     weight gradient  dW[ci][co][ky][kx] = sum_{y,x} S[ci][y][x] * G[co][2y+ky-1][2x+kx-1]
//                      M = ci, N = (co, ky, kx) = CO tiles of 16, K = positions, four consecutive x per step
// With CI = 8 half of every 16 x 16 tile would be empty and the f32 matrix pipe -- not HBM -- would set the time (measured:
// 89 us for dec.2 at B = 2048 with the plain mapping, 55 us of it matrix instructions).  Both products then fold a factor
// two of the tap index into the 16-wide dimension (HALF):
//     input gradient   N = (ci, dy): TWO output rows y, y + 1 per M tile; K = (co, rho), rho = 0..5 the six G rows the pair
//                      reads, B[(ci, dy)] = W[ci][co][rho - 2 dy][kx] or 0 -- 6 K steps per co instead of 2 x 4
//     weight gradient  M = (ci, sy) with ky = 2 sy + py, summed over position rows Y = y + sy: A = S shifted by sy rows,
//                      N = (co, py, kx) = CO / 2 tiles -- half the instructions (the y-shift of wgrad_ys_kernel)
#include "dm_common.h"

namespace {

constexpr int CT_TH = 8;                    // input rows per tile

template <int CI, int CO, int TW>
struct ConvTBwdGeom {
    static constexpr int GROWS = 2 * CT_TH + 2, RSG = 2 * TW + 8, C4G = RSG / 4;
    static constexpr int PSG_RAW = GROWS * RSG, PSG = PSG_RAW + ((4 - (PSG_RAW % 32)) + 32) % 32;     // == 4 (mod 32)
    static constexpr int PSS_RAW = CT_TH * TW, PSS = PSS_RAW + ((4 - (PSS_RAW % 32)) + 32) % 32;
    static constexpr int NG4 = CO * GROWS * C4G, NS4 = CI * CT_TH * (TW / 4);          // float4 per tile
    static constexpr int EG = (NG4 + DM_BLOCK - 1) / DM_BLOCK, ES = (NS4 + DM_BLOCK - 1) / DM_BLOCK;
    static constexpr int TILE_FLOATS = CO * PSG + CI * PSS;
    static constexpr int RED_FLOATS = 4 * CO * 256;                                     // the four waves' weight accumulators
    static constexpr int LDS_FLOATS = TILE_FLOATS > RED_FLOATS ? TILE_FLOATS : RED_FLOATS;   // (the slab combine reuses the tile buffers)
};

constexpr int convT_bwd_wgs(int CI) { return CI == 8 ? 3 : 2; }      // resident workgroups per CU (registers): the grid is exactly that

template <int CI, int CO, int TW>
__global__ __launch_bounds__(DM_BLOCK, convT_bwd_wgs(CI))
void convT_bwd_kernel(const float *__restrict__ S, const float *__restrict__ G, const float *__restrict__ w,
                      float *__restrict__ gin, double *__restrict__ stats, float *__restrict__ wslabs, int mask_relu, int H,
                      int W, int ntiles)
{
    using Geo = ConvTBwdGeom<CI, CO, TW>;
    static_assert(CI <= 16 && TW % 16 == 0, "one tile of input channels; rows of whole 16-position groups");
    constexpr int GROWS = Geo::GROWS, RSG = Geo::RSG, C4G = Geo::C4G, PSG = Geo::PSG, PSS = Geo::PSS;
    constexpr bool HALF = CI == 8;
    constexpr int EG = Geo::EG, ES = Geo::ES, CG = TW / 16;
    constexpr int MTW = (HALF ? CT_TH / 2 : CT_TH) * CG / 4;       // M tiles of the input gradient per wave
    constexpr int KS = HALF ? CO * 6 : CO * 4;                      // its K steps
    constexpr int NWT = HALF ? CO / 2 : CO;                         // N tiles of the weight gradient
    static_assert(!HALF || CO % 2 == 0, "HALF: (co, py, kx) tiles of 16");
    __shared__ __attribute__((aligned(16))) float lds[Geo::LDS_FLOATS];
    __shared__ double s_stat[4][16];
    float *sG = lds, *sS = lds + CO * PSG;

    const int lane = threadIdx.x & 63, m = lane & 15, kq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int OH = 2 * H, OW = 2 * W, tiles_x = W / TW, tiles_y = H / CT_TH;

    // weights of the input gradient: K step (co, ky), B[k = kx = kq][n = ci = m] = W[ci][co][ky][kx];
    // HALF: K step (co, rho), n = (ci = m & 7, dy = m >> 3): W[ci][co][rho - 2 dy][kx], zero outside the 4 taps
    float wreg[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        if constexpr (HALF) {
            const int co = s / 6, ky = s % 6 - 2 * (m >> 3);
            wreg[s] = (ky >= 0 && ky <= 3) ? w[((m & 7) * CO + co) * 16 + ky * 4 + kq] : 0.f;
        } else {
            wreg[s] = m < CI ? w[(m * CO + (s >> 2)) * 16 + (s & 3) * 4 + kq] : 0.f;
        }
    }

    // staging: element e of thread tid = float4 number e * 256 + tid of the tile image (G: [co][row][col4]; S: [ci][row][col4])
    int g_lds[EG], g_row[EG], g_col[EG], g_ch[EG];
#pragma unroll
    for (int e = 0; e < EG; ++e) {
        const int i = e * DM_BLOCK + threadIdx.x;
        const int co = i / (GROWS * C4G), rem = i - co * (GROWS * C4G), lr = rem / C4G, c4 = rem - lr * C4G;
        g_ch[e] = i < Geo::NG4 ? co : -1;
        g_row[e] = lr; g_col[e] = 4 * c4;
        g_lds[e] = co * PSG + lr * RSG + 4 * c4;
    }
    int s_lds[ES], s_off[ES];
    bool s_ok[ES];
#pragma unroll
    for (int e = 0; e < ES; ++e) {
        const int i = e * DM_BLOCK + threadIdx.x;
        const int ci = i / (CT_TH * (TW / 4)), rem = i - ci * (CT_TH * (TW / 4)), lr = rem / (TW / 4), c4 = rem - lr * (TW / 4);
        s_ok[e] = i < Geo::NS4;
        s_lds[e] = ci * PSS + lr * TW + 4 * c4;
        s_off[e] = (ci * H + lr) * W + 4 * c4;
    }
    f32x4 rg[EG], rs[ES];
    auto coords = [&](int t, int &b, int &y0, int &x0) {
        x0 = (t % tiles_x) * TW; t /= tiles_x;
        y0 = (t % tiles_y) * CT_TH; b = t / tiles_y;
    };
    auto issue = [&](int t) {
        int b, y0, x0;
        coords(t, b, y0, x0);
        const float *gb = G + (long long)b * CO * OH * OW;
        const float *sb = S + (long long)b * CI * H * W + (long long)y0 * W + x0;
#pragma unroll
        for (int e = 0; e < EG; ++e) {
            const int gr = 2 * y0 - 1 + g_row[e], gc = 2 * x0 - 4 + g_col[e];
            const bool ok = g_ch[e] >= 0 && (unsigned)gr < (unsigned)OH && (unsigned)gc < (unsigned)OW;
            rg[e] = ok ? *reinterpret_cast<const f32x4 *>(gb + ((long long)g_ch[e] * OH + gr) * OW + gc) : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int e = 0; e < ES; ++e)
            rs[e] = s_ok[e] ? *reinterpret_cast<const f32x4 *>(sb + s_off[e]) : (f32x4){0.f, 0.f, 0.f, 0.f};
    };

    f32x4 wacc[NWT];
#pragma unroll
    for (int t = 0; t < NWT; ++t) wacc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    double s1 = 0.0;
    // weight-gradient operand offsets: A[m = ci][k = kq] = S[ci][row][4 s + kq]; B[k = kq][n = (co = t, ky, kx)] =
    // G[t][2 row + ky][2 (4 s + kq) + kx + 3]   (column j of the G tile <-> image column 2 x0 - 4 + j)
    // HALF: position rows Yr = 0..TH; A[m = (ci, sy)][k] = S[ci][Yr - sy][4 s + kq] (zero when that row is not the tile's);
    //       B[k][n = 16 t + m = (co, py, kx)] = G[co][2 Yr + py][2 (4 s + kq) + kx + 3]
    const int wsy = HALF ? (m >> 3) : 0;
    const int wa = HALF ? (m & 7) * PSS + kq - wsy * TW : (m < CI ? m : 0) * PSS + kq;
    const float wa_on = (HALF || m < CI) ? 1.f : 0.f;
    int wbt[NWT];
#pragma unroll
    for (int t = 0; t < NWT; ++t) {
        if constexpr (HALF) {
            const int n = 16 * t + m;
            wbt[t] = (n >> 3) * PSG + ((n >> 2) & 1) * RSG + (n & 3) + 3 + 2 * kq;
        } else {
            wbt[t] = t * PSG + (m >> 2) * RSG + (m & 3) + 3 + 2 * kq;
        }
    }

    int tile = blockIdx.x;
    if (tile < ntiles) issue(tile);
    while (tile < ntiles) {
        __syncthreads();                                         // the previous tile has been consumed
#pragma unroll
        for (int e = 0; e < EG; ++e)
            if (g_ch[e] >= 0) *reinterpret_cast<f32x4 *>(sG + g_lds[e]) = rg[e];
#pragma unroll
        for (int e = 0; e < ES; ++e)
            if (s_ok[e]) *reinterpret_cast<f32x4 *>(sS + s_lds[e]) = rs[e];
        __syncthreads();
        int b, y0, x0;
        coords(tile, b, y0, x0);
        tile += gridDim.x;
        if (tile < ntiles) issue(tile);                          // in flight during the products below

        // ---- weight gradient, four positions per step
        if constexpr (HALF) {
            // position rows Yr = 0..TH: this wave takes rows wave and wave + 4 whole and a quarter of row TH
            auto steps = [&](int Yr, int s0, int ns) {
                const float on = (Yr - wsy >= 0 && Yr - wsy < CT_TH) ? 1.f : 0.f;
                const float *pa = sS + wa + (on != 0.f ? Yr * TW : wsy * TW);     // (a row that is not the tile's: read row 0, times zero)
                for (int s = s0; s < s0 + ns; ++s) {
                    const float a = pa[4 * s] * on;
#pragma unroll
                    for (int t = 0; t < NWT; ++t)
                        wacc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, sG[wbt[t] + 2 * Yr * RSG + 8 * s], wacc[t], 0, 0, 0);
                }
            };
            steps(wave, 0, TW / 4);
            steps(wave + 4, 0, TW / 4);
            steps(CT_TH, wave * (TW / 16), TW / 16);
        } else {
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
                const int row = wave + 4 * rr;
                const float *pa = sS + wa + row * TW;
#pragma unroll
                for (int s = 0; s < TW / 4; ++s) {
                    const float a = pa[4 * s] * wa_on;
#pragma unroll
                    for (int t = 0; t < NWT; ++t)
                        wacc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, sG[wbt[t] + 2 * row * RSG + 8 * s], wacc[t], 0, 0, 0);
                }
            }
        }
        // ---- input gradient: M tiles (row or row pair, 16-column group) wave, wave + 4, ..
        float *__restrict__ gb = gin + (long long)b * CI * H * W + (long long)y0 * W + x0;
#pragma unroll
        for (int i = 0; i < MTW; ++i) {
            const int ti = wave + 4 * i, rp = ti / CG, cg = ti - rp * CG;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            // A[m = x][k = kx = kq] of K step (co, ky): G[co][2 r + ky][2 (16 cg + m) + kq + 3];  HALF: rows 4 rp + rho
            const float *pa = sG + (HALF ? 4 : 2) * rp * RSG + 2 * (16 * cg + m) + kq + 3;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const int off = HALF ? (s / 6) * PSG + (s % 6) * RSG : (s >> 2) * PSG + (s & 3) * RSG;
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[off], wreg[s], acc, 0, 0, 0);
            }
            // lane (m, kq): positions 16 cg + 4 kq .. + 3 of channel ci = m, row rp (HALF: ci = m & 7, row 2 rp + (m >> 3))
            const int ci = HALF ? (m & 7) : m, r = HALF ? 2 * rp + (m >> 3) : rp;
            if (HALF || m < CI) {
                const int off = r * TW + 16 * cg + 4 * kq;
                if (mask_relu) {
                    const f32x4 sv = *reinterpret_cast<const f32x4 *>(sS + ci * PSS + off);
                    acc.x = sv.x > 0.f ? acc.x : 0.f; acc.y = sv.y > 0.f ? acc.y : 0.f;
                    acc.z = sv.z > 0.f ? acc.z : 0.f; acc.w = sv.w > 0.f ? acc.w : 0.f;
                }
                *reinterpret_cast<f32x4 *>(gb + ((long long)ci * H + r) * W + 16 * cg + 4 * kq) = acc;
                s1 += (double)((acc.x + acc.y) + (acc.z + acc.w));
            }
        }
    }

    // ---- channel sums of the input gradient (the previous layer's bias gradient): kq groups, then waves in wave order
    __syncthreads();
    if (stats) {
        double a = s1;
        a += __shfl_xor(a, 16, 64);
        a += __shfl_xor(a, 32, 64);
        if (HALF) a += __shfl_xor(a, 8, 64);                     // the two output rows of a channel
        if (lane < 16) s_stat[wave][lane] = a;
    }
    // ---- weight-gradient slab: the four waves' accumulators through LDS in wave order
    float *red = lds;
#pragma unroll
    for (int t = 0; t < NWT; ++t) *reinterpret_cast<f32x4 *>(red + ((wave * NWT + t) * 64 + lane) * 4) = wacc[t];
    __syncthreads();
    if (stats && threadIdx.x < CI) {
        double ta = 0.0;
#pragma unroll
        for (int wv = 0; wv < 4; ++wv) ta += s_stat[wv][threadIdx.x];
        stats[((long long)blockIdx.x * CI + threadIdx.x) * 2 + 0] = ta;
        stats[((long long)blockIdx.x * CI + threadIdx.x) * 2 + 1] = 0.0;
    }
    // element i of the slab = dW[ci][co][ky][kx]: accumulator row 4 kq + r of lane (m, kq) in an N tile --
    // plain: row = ci, tile co, column m = (ky, kx);  HALF: row = (ci, sy) = ci + 8 sy, column 16 t + m = (co, py, kx), ky = 2 sy + py
    for (int i = threadIdx.x; i < CI * CO * 16; i += DM_BLOCK) {
        const int ci = i / (CO * 16), rem = i - ci * (CO * 16), co = rem >> 4, ky = (rem >> 2) & 3, kx = rem & 3;
        int row, t, n;
        if constexpr (HALF) { row = ci + 8 * (ky >> 1); const int col = co * 8 + (ky & 1) * 4 + kx; t = col >> 4; n = col & 15; }
        else { row = ci; t = co; n = ky * 4 + kx; }
        const int ln = (row >> 2) * 16 + n, r = row & 3;
        float sum = 0.f;
#pragma unroll
        for (int wv = 0; wv < 4; ++wv) sum += red[((wv * NWT + t) * 64 + ln) * 4 + r];
        wslabs[(long long)blockIdx.x * CI * CO * 16 + i] = sum;
    }
}

// (CI, CO, W): dec.2 of the default model is (8, 4, 32); dec.0 (16, 8, 16); wider images have more tiles per row
int convT_bwd_tw(int CI, int CO, int W)
{
    if (CI == 8 && CO == 4) return W % 32 == 0 ? 32 : 0;
    if (CI == 16 && CO == 8) return W % 16 == 0 ? 16 : 0;        // (32-column tiles: 256 VGPRs and spills)
    return 0;
}
bool convT_bwd_shape(int CI, int CO, int H, int W) { return H > 0 && W > 0 && H % CT_TH == 0 && convT_bwd_tw(CI, CO, W) > 0; }

}  // namespace

extern "C" int dm_convt_bwd_fused_supported(int CI, int CO, int H, int W) { return convT_bwd_shape(CI, CO, H, W) ? 1 : 0; }

extern "C" int dm_convt_bwd_fused_num_blocks(int B, int CI, int CO, int H, int W)
{
    if (B <= 0 || !convT_bwd_shape(CI, CO, H, W)) return -1;
    const long long ntiles = (long long)B * (H / CT_TH) * (W / convT_bwd_tw(CI, CO, W));
    const long long cap = 256LL * convT_bwd_wgs(CI);             // exactly what is resident: a larger grid would run in two rounds
    return (int)(ntiles < cap ? ntiles : cap);
}

extern "C" int dm_convt_bwd_fused(const float *S, const float *G, const float *w, float *gin, double *stats, float *wslabs,
                                  int mask_relu, int B, int CI, int CO, int H, int W, void *stream)
{
    DM_REQUIRE(S && G && w && gin && wslabs, "dm_convt_bwd_fused: NULL pointer");
    DM_REQUIRE(B > 0 && convT_bwd_shape(CI, CO, H, W), "dm_convt_bwd_fused: ConvTranspose2d(%d -> %d) on %dx%d not built", CI, CO, H, W);
    DM_REQUIRE((long long)B * CO * 4 * H * W < (1LL << 31), "dm_convt_bwd_fused: tensor too large");
    const int ntiles = B * (H / CT_TH) * (W / convT_bwd_tw(CI, CO, W));
    const int grid = dm_convt_bwd_fused_num_blocks(B, CI, CO, H, W);
    hipStream_t st = (hipStream_t)stream;
#define DM_CTB(CI_, CO_, TW_)                                                                                          \
    hipLaunchKernelGGL((convT_bwd_kernel<CI_, CO_, TW_>), dim3(grid), dim3(DM_BLOCK), 0, st, S, G, w, gin, stats, wslabs, \
                       mask_relu, H, W, ntiles)
    if (CI == 8) DM_CTB(8, 4, 32);
    else DM_CTB(16, 8, 16);
#undef DM_CTB
    return dm_launch_status("dm_convt_bwd_fused");
}

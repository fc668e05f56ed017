// dec_tail.hip -- the decoder tail fused around its two 128x128x4 tensors.
//
// Reference: HiddenStateExtractor/vq_vae.py:296-298 (dec.4 ConvTranspose2d(4,4,4,2,1), dec.5 ReLU,
// dec.6 Conv2d(4, num_inputs, 1)) and :320-323 (masked reconstruction loss), plus their backward as
// autograd derives it for total_loss.backward() (run_training.py:406).
//
// Unfused, d4 = relu(dec.4(d2)) and its gradient g4 are 262 144 B/patch each and are touched by five
// kernels (29 % of the training step).  Here:
//   forward   d2 -> [MFMA: dec.4 as a 3x3-neighbourhood conv, N = 4 phases x 4 channels] -> ReLU ->
//             [shuffle-reduce over the 4 channel lanes: dec.6] -> decoded + loss partials.  d4 is never stored.
//   backward  d2 -> recompute d4 the same way -> g_dec from (decoded, x) -> g4 = (W6^T g_dec)*(d4>0) written to an
//             LDS tile (with the halo rows the strided conv needs) -> from LDS: data gradient of dec.4
//             (4x4/s2 conv, masked by d2>0) and weight gradient of dec.4; dW6/db6/db4/db2 partial sums on the way.
//             g4 never exists in HBM.
// Algorithmic bytes/patch: forward 65 536 (d2) + 131 072 (x) + 131 072 (decoded) = 327 680;
// backward 65 536 + 131 072 + 131 072 + 65 536 (g2) = 393 216  (unfused: 851 968 and 1 507 328).
//
// Built for the default decoder family: num_hiddens//4 = 4 channels, d2 exactly 64 wide (128x128 patches);
// other shapes use the unfused kernels (dynamorph_amd/engine.py decides).
#include "dm_common.h"
#include "tile.h"
#include "mfma_util.h"

namespace {

constexpr int TT_C = 4;        // channels of d2 / d4
constexpr int TT_TH = 8;       // d2 rows per tile
constexpr int TT_W = 64;       // d2 width (one tile spans the full row)
constexpr int TT_RS = TT_W + 8;             // d2 LDS row: col j <-> x = j - 4
constexpr int TT_MAX_GRID = 512;            // 2 workgroups per CU (LDS-limited)

// dec.4 weights for the pixel-shuffle formulation: lane n = co*4 + py*2 + px, K lane kq = input channel,
// K step = tap (ty,tx) of the 3x3 neighbourhood; kernel element ky = py+3-2ty, kx = px+3-2tx when in 0..3.
__device__ __forceinline__ void load_convT_weights(float (&wreg)[1][9], const float *__restrict__ w4, int m, int kq)
{
    const int co = m >> 2, py = (m >> 1) & 1, px = m & 1;
#pragma unroll
    for (int s = 0; s < 9; ++s) {
        const int ty = s / 3, tx = s % 3;
        const int ky = py + 3 - 2 * ty, kx = px + 3 - 2 * tx;
        float v = 0.f;
        if (ky >= 0 && ky <= 3 && kx >= 0 && kx <= 3) v = w4[(kq * TT_C + co) * 16 + ky * 4 + kx];
        wreg[0][s] = v;
    }
}

// swap halves with the x-phase partner lane (n ^ 1): afterwards the lane holds 4 CONSECUTIVE output columns
// starting at 2*(x of its first element) + 4*px, of output row 2y+py, channel co
__device__ __forceinline__ f32x4 pixel_interleave(f32x4 v, int px)
{
    const f32x4 pv = lane_xor1(v);
    return px ? (f32x4){pv.z, v.z, pv.w, v.w} : (f32x4){v.x, pv.x, v.y, pv.y};
}

__device__ __forceinline__ f32x4 relu4(f32x4 v)
{
    v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
    return v;
}

__device__ __forceinline__ float hsum4(f32x4 v) { return (v.x + v.y) + (v.z + v.w); }

// =================================================================================== forward
template <int NIN>
__global__ __launch_bounds__(DM_BLOCK, 3)
void dec_tail_forward_kernel(const float *__restrict__ d2, const float *__restrict__ w4, const float *__restrict__ b4,
                             const float *__restrict__ w6, const float *__restrict__ b6, const float *__restrict__ x,
                             const float *__restrict__ mask, int MC, const float *__restrict__ cvar,
                             float *__restrict__ dec, double *__restrict__ loss_slabs, int H2, int ntiles)
{
    constexpr int ROWS = TT_TH + 2, PS = ROWS * TT_RS;           // 720 == 16 (mod 32)
    static_assert(PS % 32 == 16, "conflict-free plane stride");
    __shared__ __attribute__((aligned(16))) float tile[TT_C * PS];
    __shared__ __attribute__((aligned(16))) float s_coef[DM_COEF_MAX_C * 4];
    __shared__ double s_red[4];

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, m = lane & 15, kq = lane >> 4;
    const int co = m >> 2, py = (m >> 1) & 1, px = m & 1;
    const int tiles_y = H2 / TT_TH, OH = 2 * H2, OW = 2 * TT_W;
    Operand in;
    in.p0 = d2; in.p1 = nullptr; in.coef = nullptr; in.coef_bstride = 0; in.mode = DM_LOAD_IDENT; in.ones = 0;

    TileStage<TT_C, ROWS, TT_RS / 4, TT_RS, PS, false> stage;
    stage.init();
    int tidx = blockIdx.x, b = 0, y0 = 0;
    if (tidx < ntiles) {
        y0 = (tidx % tiles_y) * TT_TH; b = tidx / tiles_y;
        stage.issue(in, b, TT_C, H2, TT_W, y0 - 1, -4);
    }
    if (threadIdx.x < TT_C) *reinterpret_cast<f32x4 *>(s_coef + threadIdx.x * 4) = (f32x4){1.f, 0.f, 0.f, -__builtin_inff()};

    float wreg[1][9];
    load_convT_weights(wreg, w4, m, kq);
    const float bias4 = b4[co];
    float w6c[NIN], b6c[NIN];
    float ivar = 1.f / cvar[0];                     // 1/channel_var of the channel this lane stores (c == co)
#pragma unroll
    for (int c = 0; c < NIN; ++c) {
        w6c[c] = w6[c * TT_C + co]; b6c[c] = b6 ? b6[c] : 0.f;
        if (co == c) ivar = 1.f / cvar[c];
    }

    const int abase = kq * PS + m + 3;
    auto off = [](int s) { return (s / 3) * TT_RS + s % 3; };
    double loss = 0.0;

    while (tidx < ntiles) {
        __syncthreads();
        stage.commit(tile, s_coef, TT_C, H2, TT_W, y0 - 1, -4);
        __syncthreads();
        const int cb = b, cy0 = y0;
        const int next = tidx + gridDim.x;
        if (next < ntiles) {
            y0 = (next % tiles_y) * TT_TH; b = next / tiles_y;
            stage.issue(in, b, TT_C, H2, TT_W, y0 - 1, -4);
        }
        float tl = 0.f;                                               // this tile's loss partial (fp32)
        for (int p = 0; p < (TT_TH * 4 / 4) / 2; ++p) {              // 32 M tiles, 8 per wave, 2 in flight
            const float *ap[2];
            int r[2], cg[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int ti = wave + 4 * (2 * p + i);
                r[i] = ti >> 2; cg[i] = ti & 3;
                ap[i] = tile + r[i] * TT_RS + 16 * cg[i] + abase;
            }
            // x / mask rows of the outputs this lane will own (channel c == co): in flight during the MFMAs
            f32x4 xv[2], mv[2];
            int oo[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int oy = 2 * (cy0 + r[i]) + py, ox = 2 * (16 * cg[i] + 4 * kq) + 4 * px;
                oo[i] = ((cb * NIN + co) * OH + oy) * OW + ox;
                xv[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
                mv[i] = (f32x4){1.f, 1.f, 1.f, 1.f};
                if (co < NIN && x) {
                    xv[i] = *reinterpret_cast<const f32x4 *>(x + oo[i]);
                    if (mask) mv[i] = *reinterpret_cast<const f32x4 *>(mask + ((cb * MC + (MC == 1 ? 0 : co)) * OH + oy) * OW + ox);
                }
            }
            f32x4 acc[2][1];
            acc[0][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc[1][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
            mfma_tiles<2, 1, 9, 3>(ap, wreg, acc, off);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const f32x4 d4 = pixel_interleave(relu4(acc[i][0] + bias4), px);
                // dec.6: sum over the 4 channel lanes (n differs in bits 2,3) of W6[c][co]*d4
                f32x4 mine = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int c = 0; c < NIN; ++c) {
                    f32x4 pc = w6c[c] * d4;
                    pc += lane_xor4(pc);
                    pc += lane_xor8(pc);
                    if (co == c) mine = pc + b6c[c];
                }
                if (co < NIN) {
                    *reinterpret_cast<f32x4 *>(dec + oo[i]) = mine;
                    if (x) {
                        const f32x4 t = mask ? mine * mv[i] - xv[i] * mv[i] : mine - xv[i];
                        tl += hsum4(t * t) * ivar;
                    }
                }
            }
        }
        loss += (double)tl;
        tidx = next;
    }
    const double tot = block_sum(loss, s_red);
    if (threadIdx.x == 0 && loss_slabs) loss_slabs[blockIdx.x] = tot;
}

// ================================================================================== backward
template <int NIN>
__global__ __launch_bounds__(DM_BLOCK, 2)
void dec_tail_backward_kernel(const float *__restrict__ d2, const float *__restrict__ w4, const float *__restrict__ b4,
                              const float *__restrict__ w6, const float *__restrict__ decp, const float *__restrict__ x,
                              const float *__restrict__ mask, int MC, const float *__restrict__ cvar,
                              const float *__restrict__ gscale_dev, float *__restrict__ g2, double *__restrict__ part,
                              float *__restrict__ wslabs, int H2, int ntiles, double inv_count)
{
    constexpr int AROWS = TT_TH + 4, APS_RAW = AROWS * TT_RS, APS = APS_RAW + ((16 - (APS_RAW % 32)) + 32) % 32;
    constexpr int GROWS = 2 * TT_TH + 2, GRS = 2 * TT_W + 8, GPS = GROWS * GRS;       // g4 tile: col j <-> ox = j - 4
    constexpr int NP = NIN * TT_C + NIN + TT_C + TT_C;      // dW6 | db6 | db4 | db2
    constexpr int NV = 2 * NIN + 1;                         // per-lane partials in phase 2
    static_assert(APS % 32 == 16, "conflict-free plane stride");
    __shared__ __attribute__((aligned(16))) float sA[TT_C * APS];
    __shared__ __attribute__((aligned(16))) float sG[TT_C * GPS];
    __shared__ __attribute__((aligned(16))) float s_coef[DM_COEF_MAX_C * 4];
    __shared__ double s_part[4][TT_C][NV + 1];

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, m = lane & 15, kq = lane >> 4;
    const int co = m >> 2, py = (m >> 1) & 1, px = m & 1;
    const int tiles_y = H2 / TT_TH, OH = 2 * H2, OW = 2 * TT_W;
    Operand in;
    in.p0 = d2; in.p1 = nullptr; in.coef = nullptr; in.coef_bstride = 0; in.mode = DM_LOAD_IDENT; in.ones = 0;

    TileStage<TT_C, AROWS, TT_RS / 4, TT_RS, APS, false> stage;
    stage.init();
    int tidx = blockIdx.x, b = 0, y0 = 0;
    if (tidx < ntiles) {
        y0 = (tidx % tiles_y) * TT_TH; b = tidx / tiles_y;
        stage.issue(in, b, TT_C, H2, TT_W, y0 - 2, -4);
    }
    if (threadIdx.x < TT_C) *reinterpret_cast<f32x4 *>(s_coef + threadIdx.x * 4) = (f32x4){1.f, 0.f, 0.f, -__builtin_inff()};
    // the padding columns of the g4 tile (ox = -4..-1 and 128..131) are zero for every tile
    for (int i = threadIdx.x; i < TT_C * GROWS * 2; i += DM_BLOCK) {
        const int row = i >> 1, side = i & 1;
        *reinterpret_cast<f32x4 *>(sG + row * GRS + (side ? GRS - 4 : 0)) = (f32x4){0.f, 0.f, 0.f, 0.f};
    }

    float wT[1][9];                                  // dec.4 forward (recompute), pixel-shuffle form
    load_convT_weights(wT, w4, m, kq);
    float wD[1][16];                                 // dec.4 data gradient: 4x4/s2 conv over g4, W[n=ci][c=co][ky][kx=kq]
#pragma unroll
    for (int s = 0; s < 16; ++s) wD[0][s] = m < TT_C ? w4[(m * TT_C + (s >> 2)) * 16 + (s & 3) * 4 + kq] : 0.f;
    const float bias4 = b4[co];
    const float gs = (float)(2.0 * inv_count) * gscale_dev[0];
    float w6c[NIN], gsv[NIN];
#pragma unroll
    for (int c = 0; c < NIN; ++c) { w6c[c] = w6[c * TT_C + co]; gsv[c] = gs / cvar[c]; }

    // weight-gradient operands: S = d2 (rows of A), T = g4 tile; lane column n = 16*t + m <-> (ct = t, ky = m>>2, kx = m&3)
    int boff[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) boff[t] = t * GPS + (m >> 2) * GRS + (m & 3) + 3 + 2 * kq;
    const int aoffw = (m < TT_C ? m : TT_C - 1) * APS + kq + 4;
    f32x4 wacc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) wacc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};

    double pv[NV], pb2 = 0.0;                         // dW6[c][co], sum g_dec[c], db4[co] ; db2[m]
#pragma unroll
    for (int k = 0; k < NV; ++k) pv[k] = 0.0;

    const int abaseT = kq * APS + m + 3;
    auto offT = [](int s) { return (s / 3) * TT_RS + s % 3; };
    const int abaseD = 2 * m + kq + 3;
    auto offD = [](int s) { return (s >> 2) * GPS + (s & 3) * GRS; };

    while (tidx < ntiles) {
        __syncthreads();                                   // previous tile done with sA and sG
        stage.commit(sA, s_coef, TT_C, H2, TT_W, y0 - 2, -4);
        __syncthreads();
        const int cb = b, cy0 = y0;
        const int next = tidx + gridDim.x;
        if (next < ntiles) {
            y0 = (next % tiles_y) * TT_TH; b = next / tiles_y;
            stage.issue(in, b, TT_C, H2, TT_W, y0 - 2, -4);
        }

        // ---- phase 2: recompute d4 on position rows y0-1 .. y0+TH, build g4 in LDS ------------------------
        float tv[NV], tb2 = 0.f;                                      // this tile's partial sums (fp32)
#pragma unroll
        for (int k = 0; k < NV; ++k) tv[k] = 0.f;
        for (int p = 0; p < ((TT_TH + 2) * 4 / 4) / 2; ++p) {         // 40 M tiles, 10 per wave, 2 in flight
            const float *ap[2];
            int pr[2], cg[2], oy[2], ox[2];
            bool live[2];
            f32x4 dv[2][NIN], xv[2][NIN], mv[2][NIN];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int ti = wave + 4 * (2 * p + i);
                pr[i] = ti >> 2; cg[i] = ti & 3;
                ap[i] = sA + pr[i] * TT_RS + 16 * cg[i] + abaseT;     // LDS rows pr..pr+2 <-> d2 rows y-1..y+1
                oy[i] = 2 * (cy0 - 1 + pr[i]) + py;
                ox[i] = 2 * (16 * cg[i] + 4 * kq) + 4 * px;
                const int gr = 2 * pr[i] + py - 1;
                live[i] = gr >= 0 && gr < GROWS && oy[i] >= 0 && oy[i] < OH;
#pragma unroll
                for (int c = 0; c < NIN; ++c) {
                    dv[i][c] = (f32x4){0.f, 0.f, 0.f, 0.f}; xv[i][c] = dv[i][c];
                    mv[i][c] = (f32x4){1.f, 1.f, 1.f, 1.f};
                    if (live[i]) {
                        const int o = ((cb * NIN + c) * OH + oy[i]) * OW + ox[i];
                        dv[i][c] = *reinterpret_cast<const f32x4 *>(decp + o);
                        xv[i][c] = *reinterpret_cast<const f32x4 *>(x + o);
                        if (mask) mv[i][c] = *reinterpret_cast<const f32x4 *>(mask + ((cb * MC + (MC == 1 ? 0 : c)) * OH + oy[i]) * OW + ox[i]);
                    }
                }
            }
            f32x4 acc[2][1];
            acc[0][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc[1][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
            mfma_tiles<2, 1, 9, 3>(ap, wT, acc, offT);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const f32x4 d4 = pixel_interleave(relu4(acc[i][0] + bias4), px);
                const int gr = 2 * pr[i] + py - 1;
                const bool owned = live[i] && gr >= 1 && gr <= 2 * TT_TH;      // rows 2*y0 .. 2*y0+2*TH-1
                f32x4 g4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int c = 0; c < NIN; ++c) {
                    f32x4 g = mask ? (dv[i][c] * mv[i][c] - xv[i][c] * mv[i][c]) * mv[i][c] : dv[i][c] - xv[i][c];
                    g = g * gsv[c];
                    g4 += w6c[c] * g;
                    if (owned) {
                        tv[c] += hsum4(g * d4);           // dW6[c][co]
                        tv[NIN + c] += hsum4(g);          // db6[c] (taken from the co == 0 lanes)
                    }
                }
                g4.x = d4.x > 0.f ? g4.x : 0.f; g4.y = d4.y > 0.f ? g4.y : 0.f;
                g4.z = d4.z > 0.f ? g4.z : 0.f; g4.w = d4.w > 0.f ? g4.w : 0.f;
                if (!live[i]) g4 = (f32x4){0.f, 0.f, 0.f, 0.f};               // rows outside the image: zero padding
                if (owned) tv[2 * NIN] += hsum4(g4);                           // db4[co]
                if (gr >= 0 && gr < GROWS) *reinterpret_cast<f32x4 *>(sG + co * GPS + gr * GRS + ox[i] + 4) = g4;
            }
        }
        __syncthreads();

        // ---- phase 3: data gradient of dec.4 = 4x4/s2 conv over the g4 tile, masked by d2 > 0 -------------
        for (int p = 0; p < (TT_TH * 4 / 4) / 2; ++p) {               // 32 M tiles, 8 per wave
            const float *ap[2];
            int r[2], cg[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int ti = wave + 4 * (2 * p + i);
                r[i] = ti >> 2; cg[i] = ti & 3;
                ap[i] = sG + (2 * r[i]) * GRS + 32 * cg[i] + abaseD;
            }
            f32x4 acc[2][1];
            acc[0][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc[1][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
            mfma_tiles<2, 1, 16, 4>(ap, wD, acc, offD);
            if (m < TT_C) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const f32x4 dd = *reinterpret_cast<const f32x4 *>(sA + m * APS + (r[i] + 2) * TT_RS + 16 * cg[i] + 4 * kq + 4);
                    f32x4 v = acc[i][0];
                    v.x = dd.x > 0.f ? v.x : 0.f; v.y = dd.y > 0.f ? v.y : 0.f;
                    v.z = dd.z > 0.f ? v.z : 0.f; v.w = dd.w > 0.f ? v.w : 0.f;
                    *reinterpret_cast<f32x4 *>(g2 + ((cb * TT_C + m) * H2 + cy0 + r[i]) * TT_W + 16 * cg[i] + 4 * kq) = v;
                    tb2 += hsum4(v);
                }
            }
        }

#pragma unroll
        for (int k = 0; k < NV; ++k) pv[k] += (double)tv[k];
        pb2 += (double)tb2;

        // ---- phase 4: weight gradient of dec.4: R[ci][co][ky][kx] += sum d2[ci,y,x] * g4[co,2y-1+ky,2x-1+kx] ---
        for (int r = wave; r < TT_TH; r += 4) {
            float a[2], bv[2][4];
            const int ra = (r + 2) * TT_RS, rb = 2 * r * GRS;
            a[0] = sA[aoffw + ra];
#pragma unroll
            for (int t = 0; t < 4; ++t) bv[0][t] = sG[boff[t] + rb];
#pragma unroll
            for (int x4 = 0; x4 < TT_W / 4; ++x4) {
                if (x4 + 1 < TT_W / 4) {
                    a[(x4 + 1) & 1] = sA[aoffw + ra + 4 * (x4 + 1)];
#pragma unroll
                    for (int t = 0; t < 4; ++t) bv[(x4 + 1) & 1][t] = sG[boff[t] + rb + 8 * (x4 + 1)];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    wacc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[x4 & 1], bv[x4 & 1][t], wacc[t], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        tidx = next;
    }

    // ---- partial sums: reduce over the lanes of a channel, then over the 4 waves ---------------------------
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        double v = pv[k];
        v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64);
        v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
        pv[k] = v;
    }
    pb2 += __shfl_xor(pb2, 16, 64); pb2 += __shfl_xor(pb2, 32, 64);
    __syncthreads();
    if (kq == 0 && (m & 3) == 0) {
#pragma unroll
        for (int k = 0; k < NV; ++k) s_part[wave][co][k] = pv[k];
    }
    if (kq == 0 && m < TT_C) s_part[wave][m][NV] = pb2;
    __syncthreads();
    for (int n = threadIdx.x; n < NP; n += DM_BLOCK) {
        int ch, k;
        if (n < NIN * TT_C) { ch = n % TT_C; k = n / TT_C; }                    // dW6[c][co]: k = c
        else if (n < NIN * TT_C + NIN) { ch = 0; k = NIN + (n - NIN * TT_C); }   // db6[c] from the co == 0 lanes
        else if (n < NIN * TT_C + NIN + TT_C) { ch = n - NIN * TT_C - NIN; k = 2 * NIN; }   // db4[co]
        else { ch = n - NIN * TT_C - NIN - TT_C; k = NV; }                        // db2[ci]
        const double s = s_part[0][ch][k] + s_part[1][ch][k] + s_part[2][ch][k] + s_part[3][ch][k];
        part[((long long)blockIdx.x * NP + n) * 2 + 0] = s;
        part[((long long)blockIdx.x * NP + n) * 2 + 1] = 0.0;
    }

    // ---- weight-gradient slab: combine the four waves in wave order (deterministic) --------------------------
    float *red = sG;
    for (int w = 0; w < 4; ++w) {
        __syncthreads();
        if (wave == w) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                f32x4 *pp = reinterpret_cast<f32x4 *>(red + (t * 64 + lane) * 4);
                if (w == 0) *pp = wacc[t];
                else *pp = *pp + wacc[t];
            }
        }
    }
    __syncthreads();
    float *slab = wslabs + (long long)blockIdx.x * (TT_C * TT_C * 16);
    for (int i = threadIdx.x; i < 4 * 256; i += DM_BLOCK) {
        const int j = i & 3, l = (i >> 2) & 63, t = i >> 8;
        const int cs = 4 * (l >> 4) + j, n = 16 * t + (l & 15);
        if (cs < TT_C) slab[cs * (TT_C * 16) + n] = red[i];
    }
}

int tail_grid(int ntiles) { return ntiles < TT_MAX_GRID ? ntiles : TT_MAX_GRID; }

int tail_checks(const char *who, int B, int C2, int NIN, int H2, int W2)
{
    DM_REQUIRE(C2 == TT_C, "%s: num_hiddens//4 = %d not built (4)", who, C2);
    DM_REQUIRE(NIN >= 1 && NIN <= 4, "%s: num_inputs %d not built (1..4)", who, NIN);
    DM_REQUIRE(W2 == TT_W && H2 % TT_TH == 0 && B > 0, "%s: d2 must be %d wide and a multiple of %d high (got %dx%d)",
               who, TT_W, TT_TH, H2, W2);
    DM_REQUIRE((long long)B * 4 * (2 * H2) * (2 * W2) < (1LL << 31), "%s: tensor too large for 32-bit offsets", who);
    return 0;
}

}  // namespace

extern "C" int dm_dec_tail_supported(int C2, int NIN, int H2, int W2)
{
    return C2 == TT_C && NIN >= 1 && NIN <= 4 && W2 == TT_W && H2 % TT_TH == 0;
}

extern "C" int dm_dec_tail_num_blocks(int B, int H2, int W2)
{
    (void)W2;
    return tail_grid(B * (H2 / TT_TH));
}

extern "C" int dm_dec_tail_forward(const float *d2, const float *w4, const float *b4, const float *w6, const float *b6,
                                   const float *x, const float *mask, int mask_channels, const float *channel_var,
                                   float *decoded, double *loss_slabs, int B, int C2, int NIN, int H2, int W2,
                                   void *stream)
{
    DM_REQUIRE(d2 && w4 && b4 && w6 && channel_var && decoded, "dm_dec_tail_forward: NULL pointer");
    DM_REQUIRE(!x || loss_slabs, "dm_dec_tail_forward: loss_slabs required when x is given");
    DM_REQUIRE(!mask || mask_channels == 1 || mask_channels == NIN, "dm_dec_tail_forward: mask channels %d", mask_channels);
    if (tail_checks("dm_dec_tail_forward", B, C2, NIN, H2, W2)) return -1;
    const int ntiles = B * (H2 / TT_TH), grid = tail_grid(ntiles);
    hipStream_t st = (hipStream_t)stream;
#define DM_TF(N_) hipLaunchKernelGGL((dec_tail_forward_kernel<N_>), dim3(grid), dim3(DM_BLOCK), 0, st, d2, w4, b4, w6, b6, \
                                     x, mask, mask_channels, channel_var, decoded, loss_slabs, H2, ntiles)
    switch (NIN) { case 1: DM_TF(1); break; case 2: DM_TF(2); break; case 3: DM_TF(3); break; default: DM_TF(4); }
#undef DM_TF
    return dm_launch_status("dm_dec_tail_forward");
}

extern "C" int dm_dec_tail_backward(const float *d2, const float *w4, const float *b4, const float *w6,
                                    const float *decoded, const float *x, const float *mask, int mask_channels,
                                    const float *channel_var, const float *gscale_dev, float *g2, double *part_slabs,
                                    float *w_slabs, int B, int C2, int NIN, int H2, int W2, void *stream)
{
    DM_REQUIRE(d2 && w4 && b4 && w6 && decoded && x && channel_var && gscale_dev && g2 && part_slabs && w_slabs,
               "dm_dec_tail_backward: NULL pointer");
    DM_REQUIRE(!mask || mask_channels == 1 || mask_channels == NIN, "dm_dec_tail_backward: mask channels %d", mask_channels);
    if (tail_checks("dm_dec_tail_backward", B, C2, NIN, H2, W2)) return -1;
    const int ntiles = B * (H2 / TT_TH), grid = tail_grid(ntiles);
    const double inv_count = 1.0 / ((double)B * NIN * (2.0 * H2) * (2.0 * W2));
    hipStream_t st = (hipStream_t)stream;
#define DM_TB(N_) hipLaunchKernelGGL((dec_tail_backward_kernel<N_>), dim3(grid), dim3(DM_BLOCK), 0, st, d2, w4, b4, w6, \
                                     decoded, x, mask, mask_channels, channel_var, gscale_dev, g2, part_slabs, w_slabs, \
                                     H2, ntiles, inv_count)
    switch (NIN) { case 1: DM_TB(1); break; case 2: DM_TB(2); break; case 3: DM_TB(3); break; default: DM_TB(4); }
#undef DM_TB
    return dm_launch_status("dm_dec_tail_backward");
}

// wgrad_mfma.hip -- convolution weight gradients on v_mfma_f32_16x16x4_f32.
//
// Replaces aten::convolution_backward(weight) for every Conv2d / ConvTranspose2d of the
// VQ-VAE (HiddenStateExtractor/vq_vae.py:203-209, 276-298) as driven by
// total_loss.backward() in run_training.py:406.
//
//   R[cs][ct][ky][kx] = sum_{b,y,x} S[b,cs,y,x] * T[b,ct, y*s+ky-p, x*s+kx-p]
//
// GEMM view: M = S channels (16 per tile), N = (ct,ky,kx) (16 per tile), K = positions, four
// consecutive x per MFMA step.  Conv2d: S = output gradient (BatchNorm backward folded into
// the operand load), T = layer input (BatchNorm + ReLU folded in).  ConvTranspose2d: S = layer
// input, T = output gradient.  In both cases R is already in the parameter's memory layout.
//
// Workgroups are persistent: each loops over many (sample, tile) pairs keeping its partial R in
// MFMA accumulators; the four waves split the rows of a tile (split-K), are combined through LDS
// in a fixed order, and each workgroup writes one slab.  A second kernel adds the slabs in slab
// order -> bitwise reproducible, no float atomics.
#include "dm_common.h"
#include "tile.h"
#include "mfma_util.h"

bool dm_backward_split_bf16();         // conv_mfma.hip: arithmetic of the gradient kernels (dm_backward_precision)

// fallback for shapes / channel families without an MFMA instantiation (conv_generic.hip)
int dm_generic_wgrad(const Operand &S, const Operand &T, float *slabs, int B, int CS, int CT, int CTphys, int Hs, int Ws,
                     int KK, int nslabs, hipStream_t st);
// arbitrary channel counts on the MFMA, 8 x 16 tiles (conv_wide.hip)
bool dm_wide_wgrad_ok(int Hs, int Ws);
bool dm_wide_wgrad_t_affine2_ok(int CS, int CT, int Hs, int Ws, int k);
int dm_wide_wgrad_slabs(int B, int CS, int CT, int Hs, int Ws, int k);
int dm_wide_wgrad(const Operand &S, const Operand &T, float *slabs, int B, int CS, int CT, int CTphys, int Hs, int Ws,
                  int k, int nslabs, hipStream_t st);

namespace {

constexpr int WG_MAX_BLOCKS = 512;     // 2 workgroups per CU

// BF: split-bf16 operands (tile.h: both tiles hold (hi, lo) bf16 pairs; four position steps per instruction pair).
template <int CS, int CT, int KK, int TH, int TW, bool STWO, bool BF = false>
__global__ __launch_bounds__(DM_BLOCK) void wgrad_kernel(Operand S, Operand T, float *__restrict__ slabs,
                                                         int CTphys, int Hs, int Ws, int ntiles)
{
    constexpr int STRIDE = KK == 4 ? 2 : 1, PAD = KK == 1 ? 0 : 1;
    constexpr int MT = (CS + 15) / 16, N = CT * KK * KK, NTT = (N + 15) / 16;
    // S tile [CS][TH][TW], plane stride == 4 (mod 32) dwords (2-way conflict on the A read at worst)
    constexpr int RSS = TW, PSS_RAW = TH * RSS, PSS = PSS_RAW + ((4 - (PSS_RAW % 32)) + 32) % 32;
    // T tile [CT][TH*s + KK - s][TW*s + 8*PAD]; col 0 <-> global x = x0*s - 4*PAD
    constexpr int TROWS = TH * STRIDE + KK - STRIDE, RST = TW * STRIDE + 8 * PAD, TCOLS4 = RST / 4;
    constexpr int PST = TROWS * RST;
    constexpr int RED = MT * NTT * 256;
    constexpr int LDS_TILES = CS * PSS + CT * PST;
    constexpr int LDS_FLOATS = LDS_TILES > RED ? LDS_TILES : RED;
    static_assert(TH % 4 == 0 && TW % 4 == 0, "tile shape");
    __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];
    __shared__ __attribute__((aligned(16))) float s_coefS[DM_COEF_MAX_C * 4];
    __shared__ __attribute__((aligned(16))) float s_coefT[DM_COEF_MAX_C * 4];
    float *sS = lds, *sT = lds + CS * PSS;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, m = lane & 15, kq = lane >> 4;
    const int Ht = Hs * STRIDE, Wt = Ws * STRIDE;
    const int tiles_x = Ws / TW, tiles_y = Hs / TH;

    // B-operand gather offsets: lane column n = 16*t + m <-> (ct, ky, kx)
    int boff[NTT];
#pragma unroll
    for (int t = 0; t < NTT; ++t) {
        int n = 16 * t + m;
        if (n >= N) n = N - 1;                       // garbage column, dropped at write time
        const int ct = n / (KK * KK), k2 = n % (KK * KK), ky = k2 / KK, kx = k2 % KK;
        boff[t] = ct * PST + ky * RST + kx + 3 * PAD + kq * STRIDE;
    }
    int aoff[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        int cs = 16 * t + m;
        if (cs >= CS) cs = CS - 1;                   // garbage row, dropped at write time
        aoff[t] = cs * PSS + kq;
    }

    f32x4 acc[MT][NTT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int t = 0; t < NTT; ++t) acc[i][t] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // software pipeline over this workgroup's tiles: the loads of tile i+1 are issued before the MFMA loop
    // of tile i and written to LDS after it (register staging, single LDS buffer).
    TileStage<CS, TH, TW / 4, RSS, PSS, STWO> stS;
    TileStage<CT, TROWS, TCOLS4, RST, PST, false> stT;
    stS.init(Hs, Ws);
    stT.init(Ht, Wt);
    int tile = blockIdx.x;
    int cb = 0, cy0 = 0, cx0 = 0;
    if (tile < ntiles) {
        int tid = tile;
        const int tx = tid % tiles_x; tid /= tiles_x;
        cy0 = (tid % tiles_y) * TH; cb = tid / tiles_y; cx0 = tx * TW;
        stS.issue(S, cb, CS, Hs, Ws, cy0, cx0);
        stT.issue(T, cb, CTphys, Ht, Wt, cy0 * STRIDE - PAD, cx0 * STRIDE - 4 * PAD);
        stage_coef(s_coefS, S, cb, CS);
        stage_coef(s_coefT, T, cb, CTphys);
    }
    while (tile < ntiles) {
        __syncthreads();                             // previous tile fully consumed; coefficient tables visible
        stS.template commit<BF>(sS, s_coefS, CS, Hs, Ws, cy0, cx0, S.mode);
        stT.template commit<BF>(sT, s_coefT, CTphys, Ht, Wt, cy0 * STRIDE - PAD, cx0 * STRIDE - 4 * PAD, T.mode);
        __syncthreads();
        const int next = tile + gridDim.x;
        const int pb = cb;                               // sample of the tile now in LDS
        {
            int tid = next < ntiles ? next : tile;       // (no next tile: empty descriptors, every load returns 0)
            const int tx = tid % tiles_x; tid /= tiles_x;
            cy0 = (tid % tiles_y) * TH; cb = tid / tiles_y; cx0 = tx * TW;
        }
        // the next tile's loads are requested element by element between the MFMA steps below (tile.h: issue_one)
        const auto cxS = stS.begin(S, next < ntiles, cb, CS, Hs, Ws, cy0, cx0);
        const auto cxT = stT.begin(T, next < ntiles, cb, CTphys, Ht, Wt, cy0 * STRIDE - PAD, cx0 * STRIDE - 4 * PAD);
        // per-sample coefficient tables: the next sample's rows are requested now and written at the end of the tile
        const bool recoefS = next < ntiles && coef_changes(S, pb, cb), recoefT = next < ntiles && coef_changes(T, pb, cb);
        f32x4 cfS = {1.f, 0.f, 0.f, 0.f}, cfT = {1.f, 0.f, 0.f, 0.f};
        if (recoefS) cfS = coef_fetch(S, cb, CS);
        if (recoefT) cfT = coef_fetch(T, cb, CTphys);
        constexpr int NES = decltype(stS)::N, NET = decltype(stT)::N, NE = NES + NET;
        constexpr int ROWS_W = TH / 4, NSTEPS = ROWS_W * (TW / 4);
        if constexpr (BF) {
            // quads of position steps: four 4-byte reads per operand tile and quad (the lane's element of steps 4q .. 4q+3)
            // make one operand of v_mfma_f32_16x16x32_bf16; operands of quad q+1 requested before the products of quad q
            static_assert((TW / 4) % 4 == 0, "position steps of a row in quads");
            constexpr int QR = TW / 16, NQ = ROWS_W * QR;         // quads per row, per wave and tile
            auto qoffA = [&](int q) { return (wave + 4 * (q / QR)) * RSS + 16 * (q % QR); };
            auto qoffB = [&](int q) { return (wave + 4 * (q / QR)) * STRIDE * RST + 16 * (q % QR) * STRIDE; };
            // units of (quad, group of GT N tiles): the B operands of the next unit (and, at a new quad, its A operands)
            // are requested before the products of this one -- 16 registers of B in flight whatever NTT is
            constexpr int GT = NTT < 4 ? NTT : 4, NGRP = (NTT + GT - 1) / GT, NU = NQ * NGRP;
            float a[2][MT][4], bv[2][GT][4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
#pragma unroll
                for (int i = 0; i < MT; ++i) a[0][i][j] = sS[aoff[i] + qoffA(0) + 4 * j];
#pragma unroll
                for (int t = 0; t < GT; ++t) bv[0][t][j] = sT[boff[t] + qoffB(0) + 4 * j * STRIDE];
            }
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                const int q = u / NGRP, grp = u % NGRP;
                if (u + 1 < NU) {
                    const int q1 = (u + 1) / NGRP, g1 = (u + 1) % NGRP;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if (g1 == 0) {
#pragma unroll
                            for (int i = 0; i < MT; ++i) a[q1 & 1][i][j] = sS[aoff[i] + qoffA(q1) + 4 * j];
                        }
#pragma unroll
                        for (int t = 0; t < GT; ++t)
                            if (g1 * GT + t < NTT) bv[(u + 1) & 1][t][j] = sT[boff[g1 * GT + t] + qoffB(q1) + 4 * j * STRIDE];
                    }
                }
#pragma unroll
                for (int e = 0; e < NE; ++e)
                    if (e >= u * NE / NU && e < (u + 1) * NE / NU) {
                        if (e < NES) stS.issue_one(e, cxS);
                        else stT.issue_one(e - NES, cxT);
                    }
                __builtin_amdgcn_sched_barrier(0);
                dm_u32x4_t a4[MT], ar[MT];
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    const float(&x)[4] = a[q & 1][i];
                    a4[i] = (dm_u32x4_t){__builtin_bit_cast(unsigned, x[0]), __builtin_bit_cast(unsigned, x[1]),
                                         __builtin_bit_cast(unsigned, x[2]), __builtin_bit_cast(unsigned, x[3])};
                    ar[i] = dm_rot16(a4[i]);
                }
#pragma unroll
                for (int t = 0; t < GT; ++t) {
                    if (grp * GT + t < NTT) {
                        const float(&y)[4] = bv[u & 1][t];
                        const dm_u32x4_t b4 = {__builtin_bit_cast(unsigned, y[0]), __builtin_bit_cast(unsigned, y[1]),
                                               __builtin_bit_cast(unsigned, y[2]), __builtin_bit_cast(unsigned, y[3])};
#pragma unroll
                        for (int i = 0; i < MT; ++i) acc[i][grp * GT + t] = dm_mfma_split(a4[i], ar[i], b4, acc[i][grp * GT + t]);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
#pragma unroll
        for (int ri = 0; ri < ROWS_W; ++ri) {
            const int r = wave + 4 * ri;
            // software pipeline over the position steps of a row: operands of step x4+1 are requested before
            // the MFMAs of step x4 (two register buffers, order pinned with sched_barrier)
            float a[2][MT], bv[2][NTT];
            const int rb = r * STRIDE * RST, ra = r * RSS;
#pragma unroll
            for (int i = 0; i < MT; ++i) a[0][i] = sS[aoff[i] + ra];
#pragma unroll
            for (int t = 0; t < NTT; ++t) bv[0][t] = sT[boff[t] + rb];
#pragma unroll
            for (int x4 = 0; x4 < TW / 4; ++x4) {
                if (x4 + 1 < TW / 4) {
#pragma unroll
                    for (int i = 0; i < MT; ++i) a[(x4 + 1) & 1][i] = sS[aoff[i] + ra + 4 * (x4 + 1)];
#pragma unroll
                    for (int t = 0; t < NTT; ++t) bv[(x4 + 1) & 1][t] = sT[boff[t] + rb + 4 * (x4 + 1) * STRIDE];
                }
                {
                    constexpr int dummy = 0; (void)dummy;
                    const int s = ri * (TW / 4) + x4;
#pragma unroll
                    for (int e = 0; e < NE; ++e)
                        if (e >= s * NE / NSTEPS && e < (s + 1) * NE / NSTEPS) {
                            if (e < NES) stS.issue_one(e, cxS);
                            else stT.issue_one(e - NES, cxT);
                        }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < NTT; ++t)
#pragma unroll
                    for (int i = 0; i < MT; ++i)
                        acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[x4 & 1][i], bv[x4 & 1][t], acc[i][t], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        }
        if (recoefS) coef_put(s_coefS, S, cfS, CS);
        if (recoefT) coef_put(s_coefT, T, cfT, CTphys);
        tile = next;
    }

    // combine the four waves in wave order (deterministic), then write this workgroup's slab
    float *red = lds;
    for (int w = 0; w < 4; ++w) {
        __syncthreads();
        if (wave == w) {
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int t = 0; t < NTT; ++t) {
                    f32x4 *p = reinterpret_cast<f32x4 *>(red + ((i * NTT + t) * 64 + lane) * 4);
                    if (w == 0) *p = acc[i][t];
                    else *p = *p + acc[i][t];
                }
        }
    }
    __syncthreads();
    float *slab = slabs + (long long)blockIdx.x * (CS * N);
    for (int i = threadIdx.x; i < RED; i += DM_BLOCK) {
        const int j = i & 3, l = (i >> 2) & 63, tt = i >> 8;
        const int t = tt % NTT, mi = tt / NTT;
        const int cs = 16 * mi + 4 * (l >> 4) + j, n = 16 * t + (l & 15);
        if (cs < CS && n < N) slab[cs * N + n] = red[i];
    }
}

// ---- 4x4 / stride 2 with 8 S channels (enc.0 o enc.1, dec.2): the direct mapping fills only 8 of the 16 M rows.
// Writing ky = 2*sy + py and summing over Y = y + sy instead of y,
//     R[cs][ct][2sy+py][kx] = sum_{Y,x} S[cs][Y-sy][x] * T[ct][2Y-1+py][2x-1+kx],
// M = (cs, sy) = 16 full rows and N = (ct, py, kx) = 8*CT columns: half the MFMAs per position step (a third with
// the extra position row: TH+1 rows Y per tile; S rows outside the tile's own TH rows are zeroed on the A side,
// they belong to the neighbouring tile's terms).  Each wave takes a quarter of the x steps of every row.
template <int CT, int TH, int TW, bool STWO>
__global__ __launch_bounds__(DM_BLOCK) void wgrad_ys_kernel(Operand S, Operand T, float *__restrict__ slabs,
                                                            int CTphys, int Hs, int Ws, int ntiles)
{
    constexpr int CS = 8, KK = 4, STRIDE = 2, PAD = 1;
    constexpr int N = CT * 8, NTT = (N + 15) / 16, NOUT = CT * 16;
    constexpr int RSS = TW, PSS_RAW = TH * RSS, PSS = PSS_RAW + ((4 - (PSS_RAW % 32)) + 32) % 32;
    constexpr int TROWS = TH * STRIDE + KK - STRIDE, RST = TW * STRIDE + 8 * PAD, TCOLS4 = RST / 4;
    constexpr int PST = TROWS * RST;
    constexpr int RED = NTT * 256;
    constexpr int LDS_TILES = CS * PSS + CT * PST;
    constexpr int LDS_FLOATS = LDS_TILES > RED ? LDS_TILES : RED;
    constexpr int XW = TW / 16;                          // x steps (of 4 positions) per wave and row
    static_assert(TW % 16 == 0 && TH % 4 == 0, "tile shape");
    __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];
    __shared__ __attribute__((aligned(16))) float s_coefS[DM_COEF_MAX_C * 4];
    __shared__ __attribute__((aligned(16))) float s_coefT[DM_COEF_MAX_C * 4];
    float *sS = lds, *sT = lds + CS * PSS;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, m = lane & 15, kq = lane >> 4;
    const int Ht = Hs * STRIDE, Wt = Ws * STRIDE;
    const int tiles_x = Ws / TW, tiles_y = Hs / TH;

    // A row m = (cs, sy): S[cs][Y - sy][x];  B column n = 16t + m = (ct, py, kx): T tile row 2Y + py, column 2x + 3 + kx
    const int sy = m & 1;
    const int aoff = (m >> 1) * PSS + kq + 4 * XW * wave;          // + (Y - sy) * RSS, never below row 0 (masked there)
    int boff[NTT];
#pragma unroll
    for (int t = 0; t < NTT; ++t) {
        int n = 16 * t + m;
        if (n >= N) n = N - 1;                           // garbage column, dropped at write time
        boff[t] = (n >> 3) * PST + ((n >> 2) & 1) * RST + (n & 3) + 3 + 2 * kq + 8 * XW * wave;
    }
    f32x4 acc[NTT];
#pragma unroll
    for (int t = 0; t < NTT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};

    TileStage<CS, TH, TW / 4, RSS, PSS, STWO> stS;
    TileStage<CT, TROWS, TCOLS4, RST, PST, false> stT;
    stS.init(Hs, Ws);
    stT.init(Ht, Wt);
    int tile = blockIdx.x;
    int cb = 0, cy0 = 0, cx0 = 0;
    if (tile < ntiles) {
        int tid = tile;
        const int tx = tid % tiles_x; tid /= tiles_x;
        cy0 = (tid % tiles_y) * TH; cb = tid / tiles_y; cx0 = tx * TW;
        stS.issue(S, cb, CS, Hs, Ws, cy0, cx0);
        stT.issue(T, cb, CTphys, Ht, Wt, cy0 * STRIDE - PAD, cx0 * STRIDE - 4 * PAD);
        stage_coef(s_coefS, S, cb, CS);
        stage_coef(s_coefT, T, cb, CTphys);
    }
    while (tile < ntiles) {
        __syncthreads();                             // previous tile fully consumed; coefficient tables visible
        stS.commit(sS, s_coefS, CS, Hs, Ws, cy0, cx0, S.mode);
        stT.commit(sT, s_coefT, CTphys, Ht, Wt, cy0 * STRIDE - PAD, cx0 * STRIDE - 4 * PAD, T.mode);
        __syncthreads();
        const int next = tile + gridDim.x;
        const int pb = cb;                               // sample of the tile now in LDS
        {
            int tid = next < ntiles ? next : tile;       // (no next tile: empty descriptors, every load returns 0)
            const int tx = tid % tiles_x; tid /= tiles_x;
            cy0 = (tid % tiles_y) * TH; cb = tid / tiles_y; cx0 = tx * TW;
        }
        const auto cxS = stS.begin(S, next < ntiles, cb, CS, Hs, Ws, cy0, cx0);
        const auto cxT = stT.begin(T, next < ntiles, cb, CTphys, Ht, Wt, cy0 * STRIDE - PAD, cx0 * STRIDE - 4 * PAD);
        // per-sample coefficient tables: the next sample's rows are requested now and written at the end of the tile
        const bool recoefS = next < ntiles && coef_changes(S, pb, cb), recoefT = next < ntiles && coef_changes(T, pb, cb);
        f32x4 cfS = {1.f, 0.f, 0.f, 0.f}, cfT = {1.f, 0.f, 0.f, 0.f};
        if (recoefS) cfS = coef_fetch(S, cb, CS);
        if (recoefT) cfT = coef_fetch(T, cb, CTphys);
        constexpr int NES = decltype(stS)::N, NET = decltype(stT)::N, NE = NES + NET;
        constexpr int NSTEPS = (TH + 1) * XW;
        // position rows Y = 0 .. TH; operands of step s+1 are requested before the MFMAs of step s
        float a[2], bv[2][NTT];
        a[0] = sS[aoff];
#pragma unroll
        for (int t = 0; t < NTT; ++t) bv[0][t] = sT[boff[t]];
#pragma unroll
        for (int s = 0; s < NSTEPS; ++s) {
            const int Y = s / XW, xs = s % XW;
            if (s + 1 < NSTEPS) {
                const int Yn = (s + 1) / XW, xn = (s + 1) % XW;
                a[(s + 1) & 1] = sS[aoff + (Yn == 0 ? 0 : Yn * RSS - sy * RSS) + 4 * xn];
#pragma unroll
                for (int t = 0; t < NTT; ++t) bv[(s + 1) & 1][t] = sT[boff[t] + 2 * Yn * RST + 8 * xn];
            }
#pragma unroll
            for (int e = 0; e < NE; ++e)
                if (e >= s * NE / NSTEPS && e < (s + 1) * NE / NSTEPS) {
                    if (e < NES) stS.issue_one(e, cxS);
                    else stT.issue_one(e - NES, cxT);
                }
            __builtin_amdgcn_sched_barrier(0);
            float av = a[s & 1];
            if (Y == 0) av = sy ? 0.f : av;              // S row -1: the neighbouring tile's term
            if (Y == TH) av = sy ? av : 0.f;             // S row TH likewise
            (void)xs;
#pragma unroll
            for (int t = 0; t < NTT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv[s & 1][t], acc[t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (recoefS) coef_put(s_coefS, S, cfS, CS);
        if (recoefT) coef_put(s_coefT, T, cfT, CTphys);
        tile = next;
    }

    // combine the four waves in wave order (deterministic), then write this workgroup's slab
    float *red = lds;
    for (int w = 0; w < 4; ++w) {
        __syncthreads();
        if (wave == w) {
#pragma unroll
            for (int t = 0; t < NTT; ++t) {
                f32x4 *p = reinterpret_cast<f32x4 *>(red + (t * 64 + lane) * 4);
                if (w == 0) *p = acc[t];
                else *p = *p + acc[t];
            }
        }
    }
    __syncthreads();
    float *slab = slabs + (long long)blockIdx.x * (CS * NOUT);
    for (int i = threadIdx.x; i < RED; i += DM_BLOCK) {
        const int j = i & 3, l = (i >> 2) & 63, t = i >> 8;
        const int mm = 4 * (l >> 4) + j, n = 16 * t + (l & 15);          // M row (cs, sy), N column (ct, py, kx)
        if (n < N) slab[(mm >> 1) * NOUT + (n >> 3) * 16 + (2 * (mm & 1) + ((n >> 2) & 1)) * 4 + (n & 3)] = red[i];
    }
}

// dst[e] = sum over slabs, in a fixed order: 16 slab groups x 16 elements per block; each thread adds its
// slabs (g, g+16, ...) sequentially with 4 loads in flight, the 16 group sums are then added in group order.
__global__ __launch_bounds__(256) void slab_reduce_kernel(const float *__restrict__ slabs, int nslabs, int E,
                                                          float *__restrict__ dst)
{
    __shared__ float part[16][17];
    const int el = threadIdx.x & 15, g = threadIdx.x >> 4;
    const int e = blockIdx.x * 16 + el;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (e < E) {
        int i = g;
        for (; i + 48 < nslabs; i += 64) {
            const float a = slabs[(long long)i * E + e], b = slabs[(long long)(i + 16) * E + e];
            const float c = slabs[(long long)(i + 32) * E + e], d = slabs[(long long)(i + 48) * E + e];
            s0 += a; s1 += b; s2 += c; s3 += d;
        }
        for (; i < nslabs; i += 16) s0 += slabs[(long long)i * E + e];
    }
    part[g][el] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (g == 0 && e < E) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) s += part[k][el];
        dst[e] = s;
    }
}

constexpr int SRM_MAX_SEGS = 32;       // (round 5: 32, was 16 -- decoder and encoder of a fused step in ONE launch)
struct ReduceSegs {
    const float *slabs[SRM_MAX_SEGS];
    float *dst[SRM_MAX_SEGS];
    int nslabs[SRM_MAX_SEGS];
    int E[SRM_MAX_SEGS];
    int stride[SRM_MAX_SEGS];           // elements from one slab to the next (float slabs: E)
    int dbl[SRM_MAX_SEGS];              // 1: the slabs are (value, second value) double pairs of a statistics epilogue; the first is summed
    int first_block[SRM_MAX_SEGS + 1];  // prefix sums of ceil(E / 64)
    int nseg;
};

// every (slabs, dst) pair of a backward pass in one launch; same arithmetic as slab_reduce_kernel (per element: sixteen
// groups g of slabs g, g + 16, ..., four accumulators each, the groups added in order).  1024 threads: a wave is one group
// and reads 64 consecutive elements of a slab -- 256 contiguous bytes per load (the 16-element rows of the 256-thread form
// fetched four 64-byte pieces per wave: 1.6 TB/s on the 25 MB of slabs of a backward pass).
constexpr int SRM_EL = 64;
__global__ __launch_bounds__(1024) void slab_reduce_multi_kernel(ReduceSegs rs)
{
    __shared__ float part[16][SRM_EL + 1];
    int k = 0;
    while (k + 1 < rs.nseg && (int)blockIdx.x >= rs.first_block[k + 1]) ++k;
    const float *__restrict__ slabs = rs.slabs[k];
    const int E = rs.E[k], nslabs = rs.nslabs[k];
    const int el = threadIdx.x & (SRM_EL - 1), g = threadIdx.x >> 6;
    const int e = ((int)blockIdx.x - rs.first_block[k]) * SRM_EL + el;
    if (rs.dbl[k]) {
        // bias gradients and the like: (sum, -) double pairs written by a statistics epilogue, summed in double as
        // dm_sum_slabs does (they ride in this launch instead of taking one of their own)
        __shared__ double dpart[16][SRM_EL + 1];
        const double *__restrict__ ds = reinterpret_cast<const double *>(slabs);
        const long long st = rs.stride[k];
        double a0 = 0.0, a1 = 0.0;
        if (e < E) {
            int i = g;
            for (; i + 16 < nslabs; i += 32) {
                const double a = ds[((long long)i * st + e) * 2], b = ds[((long long)(i + 16) * st + e) * 2];
                a0 += a; a1 += b;
            }
            for (; i < nslabs; i += 16) a0 += ds[((long long)i * st + e) * 2];
        }
        dpart[g][el] = a0 + a1;
        __syncthreads();
        if (g == 0 && e < E) {
            double t = 0.0;
#pragma unroll
            for (int j = 0; j < 16; ++j) t += dpart[j][el];
            rs.dst[k][e] = (float)t;
        }
        return;
    }
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (e < E) {
        int i = g;
        for (; i + 48 < nslabs; i += 64) {
            const float a = slabs[(long long)i * E + e], b = slabs[(long long)(i + 16) * E + e];
            const float c = slabs[(long long)(i + 32) * E + e], d = slabs[(long long)(i + 48) * E + e];
            s0 += a; s1 += b; s2 += c; s3 += d;
        }
        for (; i < nslabs; i += 16) s0 += slabs[(long long)i * E + e];
    }
    part[g][el] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (g == 0 && e < E) {
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) s += part[j][el];
        rs.dst[k][e] = s;
    }
}

int wgrad_tw(int Ws) { return Ws < 64 ? Ws : 64; }

// LDS floats of the S + T tiles for a TH-row tile; 8 rows unless that overflows the 64 KB static limit.
constexpr int wgrad_tile_floats(int CS, int CT, int KK, int TH, int TW)
{
    const int s = KK == 4 ? 2 : 1, pad = KK == 1 ? 0 : 1;
    const int pss_raw = TH * TW, pss = pss_raw + ((4 - (pss_raw % 32)) + 32) % 32;
    return CS * pss + CT * (TH * s + KK - s) * (TW * s + 8 * pad);
}
constexpr int wgrad_th(int CS, int CT, int KK, int TW)
{
    return wgrad_tile_floats(CS, CT, KK, 8, TW) <= 15800 ? 8 : 4;
}

template <int CS, int CT, int KK, int TW>
void launch_wgrad(const Operand &S, const Operand &T, float *slabs, int B, int CTphys, int Hs, int Ws, int grid,
                  hipStream_t st)
{
    constexpr int TH = wgrad_th(CS, CT, KK, TW);
    const int ntiles = B * (Hs / TH) * (Ws / TW);
    if constexpr (CS == 8 && KK == 4) {
        if (S.mode == DM_LOAD_AFFINE2)
            hipLaunchKernelGGL((wgrad_ys_kernel<CT, TH, TW, true>), dim3(grid), dim3(DM_BLOCK), 0, st, S, T, slabs, CTphys,
                               Hs, Ws, ntiles);
        else
            hipLaunchKernelGGL((wgrad_ys_kernel<CT, TH, TW, false>), dim3(grid), dim3(DM_BLOCK), 0, st, S, T, slabs, CTphys,
                               Hs, Ws, ntiles);
        return;
    }
    constexpr bool CAN_BF = DM_BUILD_SPLIT_BF16 && (TW / 4) % 4 == 0;             // position steps of a row in quads (retired: dm_common.h)
    if (CAN_BF && dm_backward_split_bf16()) {
        if (S.mode == DM_LOAD_AFFINE2)
            hipLaunchKernelGGL((wgrad_kernel<CS, CT, KK, TH, TW, true, CAN_BF>), dim3(grid), dim3(DM_BLOCK), 0, st, S, T, slabs,
                               CTphys, Hs, Ws, ntiles);
        else
            hipLaunchKernelGGL((wgrad_kernel<CS, CT, KK, TH, TW, false, CAN_BF>), dim3(grid), dim3(DM_BLOCK), 0, st, S, T, slabs,
                               CTphys, Hs, Ws, ntiles);
        return;
    }
    if (S.mode == DM_LOAD_AFFINE2)
        hipLaunchKernelGGL((wgrad_kernel<CS, CT, KK, TH, TW, true>), dim3(grid), dim3(DM_BLOCK), 0, st, S, T, slabs,
                           CTphys, Hs, Ws, ntiles);
    else
        hipLaunchKernelGGL((wgrad_kernel<CS, CT, KK, TH, TW, false>), dim3(grid), dim3(DM_BLOCK), 0, st, S, T, slabs,
                           CTphys, Hs, Ws, ntiles);
}

}  // namespace

extern "C" int dm_reduce_slabs_multi(const dm_reduce_seg *segs, int nseg, void *stream)
{
    DM_REQUIRE(segs && nseg >= 1 && nseg <= SRM_MAX_SEGS, "dm_reduce_slabs_multi: 1..%d segments", SRM_MAX_SEGS);
    ReduceSegs rs;
    rs.nseg = nseg;
    int blocks = 0;
    for (int k = 0; k < SRM_MAX_SEGS; ++k) {
        const bool on = k < nseg;
        if (on) DM_REQUIRE(segs[k].slabs && segs[k].dst && segs[k].nslabs > 0 && segs[k].E > 0, "dm_reduce_slabs_multi: bad segment %d", k);
        rs.slabs[k] = on ? segs[k].slabs : nullptr; rs.dst[k] = on ? segs[k].dst : nullptr;
        rs.nslabs[k] = on ? segs[k].nslabs : 0; rs.E[k] = on ? segs[k].E : 0;
        rs.dbl[k] = on ? (segs[k].pairs_of_doubles ? 1 : 0) : 0;
        rs.stride[k] = on ? (segs[k].stride > 0 ? segs[k].stride : segs[k].E) : 0;
        rs.first_block[k] = blocks;
        if (on) blocks += (segs[k].E + SRM_EL - 1) / SRM_EL;
    }
    rs.first_block[SRM_MAX_SEGS] = blocks;
    hipLaunchKernelGGL(slab_reduce_multi_kernel, dim3(blocks), dim3(1024), 0, (hipStream_t)stream, rs);
    return dm_launch_status("dm_reduce_slabs_multi");
}

extern "C" int dm_reduce_slabs(const float *slabs, int nslabs, int E, float *dst, void *stream)
{
    DM_REQUIRE(slabs && dst && nslabs > 0 && E > 0, "dm_reduce_slabs: bad argument");
    hipLaunchKernelGGL(slab_reduce_kernel, dim3((E + 15) / 16), dim3(256), 0, (hipStream_t)stream, slabs, nslabs, E, dst);
    return dm_launch_status("dm_reduce_slabs");
}

static bool wgrad_fast_tileable(int CS, int CT, int Hs, int Ws, int k)
{
    const int TW = wgrad_tw(Ws);
    if (TW != 16 && TW != 32 && TW != 64) return false;
    const int TH = wgrad_th(CS, CT, k, TW);
    return Hs % TH == 0 && Ws % TW == 0;
}

// the register-resident kernels instantiated below (DM_WG table)
static bool wgrad_has_kernel(int CS, int CT, int Hs, int Ws, int k)
{
    if (!wgrad_fast_tileable(CS, CT, Hs, Ws, k)) return false;
    const int TW = wgrad_tw(Ws);
    const bool t13 = TW == 16 || TW == 32;
    if (k == 4 && CS == 8) {
        if (CT == 3) return true;
        if (CT == 5 || CT == 2 || CT == 1) return TW == 64;
        if (CT == 4) return TW == 64 || TW == 32;
    }
    if (k == 4 && CS == 16) return CT == 8 || (CT == 16 && t13);
    if (k == 4 && CS == 4) return CT == 4 && TW == 64;
    if (k == 3) return (CS == 16 || CS == 32) && CT == 16 && t13;
    if (k == 1) return CS == 16 && CT == 32 && t13;
    return false;
}

extern "C" int dm_wgrad_t_affine2_supported(int CS, int CT, int Hs, int Ws, int k)
{
    return (!wgrad_has_kernel(CS, CT, Hs, Ws, k) && dm_wide_wgrad_t_affine2_ok(CS, CT, Hs, Ws, k)) ? 1 : 0;
}

extern "C" int dm_wgrad_num_blocks(int B, int CS, int CT, int Hs, int Ws, int k)
{
    if (B <= 0 || (k != 4 && k != 3 && k != 1)) return -1;
    if (!wgrad_has_kernel(CS, CT, Hs, Ws, k)) {
        if (dm_wide_wgrad_ok(Hs, Ws)) return dm_wide_wgrad_slabs(B, CS, CT, Hs, Ws, k);
        return B < WG_MAX_BLOCKS ? B : WG_MAX_BLOCKS;
    }
    const int TW = wgrad_tw(Ws), TH = wgrad_th(CS, CT, k, TW);
    const long long ntiles = (long long)B * (Hs / TH) * (Ws / TW);
    return (int)(ntiles < WG_MAX_BLOCKS ? ntiles : WG_MAX_BLOCKS);
}

extern "C" int dm_wgrad(const dm_operand *S, const dm_operand *T, float *slabs, float *dst,
                        int B, int CS, int CT, int Hs, int Ws, int k, void *stream)
{
    if (dm_check_operand(S, "dm_wgrad(S)") || dm_check_operand(T, "dm_wgrad(T)")) return -1;
    DM_REQUIRE(slabs, "dm_wgrad: NULL slabs");
    DM_REQUIRE((long long)B * (CS > 4 * CT ? CS : 4 * CT) * Hs * Ws < (1LL << 31), "dm_wgrad: tensor too large for 32-bit offsets");
    DM_REQUIRE(k == 4 || k == 3 || k == 1, "dm_wgrad: kernel size %d not built", k);
    DM_REQUIRE(!S->ones_channel, "dm_wgrad: S cannot carry a ones channel");
    const int TW = wgrad_tw(Ws);
    const bool fast = wgrad_has_kernel(CS, CT, Hs, Ws, k);
    const int grid = dm_wgrad_num_blocks(B, CS, CT, Hs, Ws, k);
    DM_REQUIRE(T->mode != DM_LOAD_AFFINE2 || (dm_wgrad_t_affine2_supported(CS, CT, Hs, Ws, k) && S->mode != DM_LOAD_AFFINE2 && !T->ones_channel),
               "dm_wgrad: T operand cannot be AFFINE2 for this shape (dm_wgrad_t_affine2_supported)");
    const int CTphys = CT - (T->ones_channel ? 1 : 0);
    DM_REQUIRE(CTphys > 0, "dm_wgrad: no physical T channel");
    hipStream_t st = (hipStream_t)stream;
    const Operand s = to_dev(S), t = to_dev(T);
    const int E = CS * CT * k * k;
    bool done = false;
#define DM_WG(CS_, CT_, K_, TW_)                                                         \
    if (fast && !done && CS == CS_ && CT == CT_ && k == K_ && TW == TW_) {               \
        launch_wgrad<CS_, CT_, K_, TW_>(s, t, slabs, B, CTphys, Hs, Ws, grid, st);       \
        done = true;                                                                     \
    }
    // enc.0 o enc.1 composite (x with ones channel), enc.4, enc.7
    DM_WG(8, 3, 4, 64) DM_WG(8, 3, 4, 32) DM_WG(8, 3, 4, 16) DM_WG(8, 5, 4, 64) DM_WG(8, 2, 4, 64) DM_WG(8, 4, 4, 64)
    DM_WG(8, 1, 4, 64)
    DM_WG(16, 8, 4, 32) DM_WG(16, 8, 4, 64) DM_WG(16, 8, 4, 16)
    DM_WG(16, 16, 4, 16) DM_WG(16, 16, 4, 32)
    // enc.10 and residual convs
    DM_WG(16, 16, 3, 16) DM_WG(16, 16, 3, 32)
    DM_WG(32, 16, 3, 16) DM_WG(32, 16, 3, 32)
    DM_WG(16, 32, 1, 16) DM_WG(16, 32, 1, 32)
    // decoder ConvTranspose2d layers (S = layer input, T = output gradient)
    DM_WG(8, 4, 4, 32) DM_WG(8, 4, 4, 64)
    DM_WG(4, 4, 4, 64)
#undef DM_WG
    if (!done && fast) {
        dm_set_error("dm_wgrad: kernel table and wgrad_has_kernel disagree (CS %d CT %d k %d %dx%d)", CS, CT, k, Hs, Ws);
        return -1;
    }
    if (!done) {     // no register-resident instantiation: implicit-GEMM kernel (conv_wide.hip), else the generic one
        if (dm_wide_wgrad_ok(Hs, Ws)) dm_wide_wgrad(s, t, slabs, B, CS, CT, CTphys, Hs, Ws, k, grid, st);
        else dm_generic_wgrad(s, t, slabs, B, CS, CT, CTphys, Hs, Ws, k, grid, st);
    }
    int rc = dm_launch_status("dm_wgrad");
    if (rc || !dst) return rc;          // dst == NULL: the caller reduces the slabs later (dm_reduce_slabs_multi)
    hipLaunchKernelGGL(slab_reduce_kernel, dim3((E + 15) / 16), dim3(256), 0, st, slabs, grid, E, dst);
    return dm_launch_status("dm_wgrad(reduce)");
}

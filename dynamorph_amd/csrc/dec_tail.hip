// dec_tail.hip -- the decoder tail fused around its two 128x128x4 tensors.
//
// Reference: HiddenStateExtractor/vq_vae.py:296-298 (dec.4 ConvTranspose2d(4,4,4,2,1), dec.5 ReLU,
// dec.6 Conv2d(4, num_inputs, 1)) and :320-323 (masked reconstruction loss), plus their backward as
// autograd derives it for total_loss.backward() (run_training.py:406).
//
// Unfused, d4 = relu(dec.4(d2)) and its gradient g4 are 262 144 B/patch each and are touched by five
// kernels (29 % of the training step).  Here:
//   forward   d2 -> dec.4 + ReLU into an LDS tile (VALU: one wave per channel, one lane per column) -> dec.6 + loss
//             partials from that tile.  d4 is never stored.
//   backward  d2 -> recompute d4 the same way -> g_dec from (decoded, x) -> g4 = (W6^T g_dec)*(d4>0) in place in the
//             LDS tile (with the halo rows the strided conv needs) -> from LDS: data gradient of dec.4 (VALU, masked
//             by d2>0) and weight gradient of dec.4 (MFMA); dW6/db6/db4/db2 partial sums on the way.
//             g4 never exists in HBM.
// Algorithmic bytes/patch: forward 65 536 (d2) + 131 072 (x) + 131 072 (decoded) = 327 680;
// backward 65 536 + 131 072 + 131 072 + 65 536 (g2) = 393 216  (unfused: 851 968 and 1 507 328).
//
// Built for the default decoder family: num_hiddens//4 = 4 channels.  d2 exactly 64 wide (128x128 patches): one tile spans
// the row.  Any other width (a multiple of 4; 256x256 patches: 128) takes the WIDE instantiation: a tile still is 64 lanes =
// 64 columns, x0 - 4 .. x0 + 59, of which the middle 56 are OWNED (sums, stores, weight-gradient terms); the four columns on
// either side are recomputed halo, so that everything an owned column needs from its neighbours (lane shifts, the g4 columns
// 2x-1 and 2x+2) exists in the tile exactly as in the full-row form.  Other channel counts use the unfused kernels
// (dynamorph_amd/engine.py decides).
#include "dm_common.h"
#include "tile.h"
#include "mfma_util.h"

namespace {

constexpr int TT_C = 4;        // channels of d2 / d4
constexpr int TT_TH = 8;       // d2 rows per tile
constexpr int TT_W = 64;       // d2 width (one tile spans the full row = one lane per column)
constexpr int TT_MAX_GRID = 768;            // 3 workgroups per CU (LDS-limited)
constexpr int TT_DRS = 2 * TT_W + 8;        // d4 / g4 LDS row: col j <-> ox = j - 4 (zero pad columns for the MFMA weight gradient)
constexpr int TT_HALO = 4;                  // WIDE: halo lanes on either side of a tile (a multiple of 4: 16-byte staging chunks)
constexpr int TT_OWN = TT_W - 2 * TT_HALO;  // WIDE: owned columns per tile

// Thin layers (4 channels) waste 75 % of an MFMA tile, so dec.4 forward/recompute and its data gradient run on the
// VALU: one wave per output channel (its 64 weights are wave-uniform), one lane per column; the left/right
// neighbours of a column come from the adjacent lanes with wave-shift DPP moves, whose out-of-range reads return 0
// = the zero padding at the image border (the tile spans the whole row).
// The wave index, hidden from the compiler's uniformity analysis: weights indexed with it are then loaded with
// vector loads into VGPRs.  Left uniform, hipcc keeps the 64-128 weights in SGPRs, runs out and spills them
// through v_writelane / v_readlane (650 such moves in the first build of the backward kernel).
__device__ __forceinline__ int wave_index_vgpr()
{
    int w;
    asm volatile("v_lshrrev_b32 %0, 6, %1" : "=v"(w) : "v"(threadIdx.x));
    return w;
}

__device__ __forceinline__ float lane_from_left(float v)    // value of lane - 1, 0 for lane 0
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xF, 0xF, true));
}
__device__ __forceinline__ float lane_from_right(float v)   // value of lane + 1, 0 for lane 63
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xF, 0xF, true));
}

// ConvTranspose2d(4,4,4,2,1) for ONE output channel, in packed fp32 (v_pk_fma_f32 does two FMAs per issue; the
// kernel is VALU-issue bound).  A lane owns column x and produces the output pair (ox = 2x, 2x+1):
//   out[oy][2x]   = sum_ci,d  in[d][x-1] * w[ky][3] + in[d][x]   * w[ky][1]
//   out[oy][2x+1] = sum_ci,d  in[d][x]   * w[ky][2] + in[d][x+1] * w[ky][0]
// The lane multiplies ITS value in[x] four ways: (w1, w2) into its own pair, (w3, w0) into a "side" pair whose halves
// belong to the right neighbour's out[2x'] and the left neighbour's out[2x'+1]; the side sums cross the lanes ONCE per
// output row (two wave-shift DPP moves whose out-of-range reads return 0 = the zero padding at the image border: the
// tile spans the whole row) instead of every input value crossing them first.
// Output rows 2y-1 and 2y both read exactly the input rows y-1 ("P") and y ("C"):
//   row 2y-1: P with ky = 2, C with ky = 0        row 2y: P with ky = 3, C with ky = 1
// (the four channels of a lane's column as two register PAIRS: a product needs the value in both halves of a packed
// operand, and v_pk_fma_f32 can take either half of a 64-bit operand for both of its lanes (op_sel) -- a splat of a lone
// 32-bit register costs a v_mov_b32 per use instead: 36 per tile in the first form of this kernel)
struct RowVals {
    f32x2 v[TT_C / 2];
};
__device__ __forceinline__ void load_row(const float *__restrict__ sA, int APS, int row, int lane, RowVals &r)
{
#pragma unroll
    for (int ci = 0; ci < TT_C; ci += 2)
        r.v[ci >> 1] = (f32x2){sA[ci * APS + row * TT_W + lane], sA[(ci + 1) * APS + row * TT_W + lane]};
}
// wp[ci][ky][0] = (w[ky][1], w[ky][2]) (own pair), wp[ci][ky][1] = (w[ky][3], w[ky][0]) (side pair) of W4[ci][co] for this wave's co
// (forward kernel: vector registers; the backward kernel keeps its weights in scalar registers, see there).
__device__ __forceinline__ void load_convT_weights(const float *__restrict__ w, int wv, f32x2 (&wp)[TT_C][4][2])
{
#pragma unroll
    for (int ci = 0; ci < TT_C; ++ci)
#pragma unroll
        for (int ky = 0; ky < 4; ++ky) {
            const f32x4 q = *reinterpret_cast<const f32x4 *>(w + (ci * TT_C + wv) * 16 + 4 * ky);
            wp[ci][ky][0] = (f32x2){q.y, q.z};
            wp[ci][ky][1] = (f32x2){q.w, q.x};
        }
}
__device__ __forceinline__ void convT_pair(const RowVals &P, const RowVals &C, const f32x2 (&wp)[TT_C][4][2], float bias,
                                           f32x2 &lo, f32x2 &hi)
{
    const f32x2 b2 = {bias, bias};
    f32x2 slo, shi;
#pragma unroll
    for (int ci = 0; ci < TT_C; ++ci) {
        const f32x2 pp = P.v[ci >> 1], cc = C.v[ci >> 1];
        const f32x2 p = (ci & 1) ? __builtin_shufflevector(pp, pp, 1, 1) : __builtin_shufflevector(pp, pp, 0, 0);
        const f32x2 c = (ci & 1) ? __builtin_shufflevector(cc, cc, 1, 1) : __builtin_shufflevector(cc, cc, 0, 0);
        // one statement per product: each contracts to a single v_pk_fma_f32 on its accumulator; the first channel
        // starts the four sums (no registers zeroed first)
        if (ci == 0) {
            lo = p * wp[ci][2][0] + b2; hi = p * wp[ci][3][0] + b2;
            slo = p * wp[ci][2][1]; shi = p * wp[ci][3][1];
        } else {
            lo += p * wp[ci][2][0]; hi += p * wp[ci][3][0];
            slo += p * wp[ci][2][1]; shi += p * wp[ci][3][1];
        }
        lo += c * wp[ci][0][0]; hi += c * wp[ci][1][0];
        slo += c * wp[ci][0][1]; shi += c * wp[ci][1][1];
    }
    lo.x += lane_from_left(slo.x); lo.y += lane_from_right(slo.y);
    hi.x += lane_from_left(shi.x); hi.y += lane_from_right(shi.y);
}
__device__ __forceinline__ f32x2 relu2(f32x2 v) { return (f32x2){dm_relu(v.x), dm_relu(v.y)}; }

// =================================================================================== forward
template <int NIN, bool WIDE>
__global__ __launch_bounds__(DM_BLOCK, 3)
void dec_tail_forward_kernel(const float *__restrict__ d2, const float *__restrict__ w4, const float *__restrict__ b4,
                             const float *__restrict__ w6, const float *__restrict__ b6, const float *__restrict__ x,
                             const float *__restrict__ mask, int MC, const float *__restrict__ cvar,
                             float *__restrict__ dec, double *__restrict__ loss_slabs, int H2, int ntiles, int W2arg, int tiles_x)
{
    constexpr int AROWS = TT_TH + 2, APS = AROWS * TT_W + 4;     // d2 rows y0-1 .. y0+TH
    constexpr int DROWS = 2 * TT_TH, DPS = DROWS * TT_DRS;       // d4 rows 2*y0 .. 2*y0+2*TH-1
    __shared__ __attribute__((aligned(16))) float sA[TT_C * APS];
    __shared__ __attribute__((aligned(16))) float sD[TT_C * DPS];
    __shared__ __attribute__((aligned(16))) float s_coef[DM_COEF_MAX_C * 4];
    __shared__ double s_red[4];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);    // = output channel in phase A
    const int W2 = WIDE ? W2arg : TT_W;
    const int tiles_y = H2 / TT_TH, OH = 2 * H2, OW = 2 * W2;
    Operand in;
    in.p0 = d2; in.p1 = nullptr; in.coef = nullptr; in.coef_bstride = 0; in.mode = DM_LOAD_IDENT; in.ones = 0;

    TileStage<TT_C, AROWS, TT_W / 4, TT_W, APS, false> stage;
    stage.init(H2, W2);
    // tile -> (sample, row band, column band); WIDE: lane <-> column x0 + lane, x0 = 56 * tx - 4
    auto tile_of = [&](int t, int &tb, int &ty0, int &tx0) {
        if (WIDE) { const int q = t / tiles_x; tx0 = (t - q * tiles_x) * TT_OWN - TT_HALO; t = q; } else tx0 = 0;
        ty0 = (t % tiles_y) * TT_TH; tb = t / tiles_y;
    };
    int tidx = blockIdx.x, b = 0, y0 = 0, x0 = 0;
    if (tidx < ntiles) {
        tile_of(tidx, b, y0, x0);
        stage.issue(in, b, TT_C, H2, W2, y0 - 1, x0);
    }
    if (threadIdx.x < TT_C) *reinterpret_cast<f32x4 *>(s_coef + threadIdx.x * 4) = (f32x4){1.f, 0.f, 0.f, -__builtin_inff()};

    const int wv = wave_index_vgpr();
    f32x2 wp[TT_C][4][2];
    load_convT_weights(w4, wv, wp);
    const float bias4 = b4[wv];
    float w6r[NIN][TT_C], b6r[NIN], ivar[NIN];
#pragma unroll
    for (int c = 0; c < NIN; ++c) {
        b6r[c] = b6 ? b6[c] : 0.f; ivar[c] = 1.f / cvar[c];
#pragma unroll
        for (int co = 0; co < TT_C; ++co) w6r[c][co] = w6[c * TT_C + co];
    }
    double loss = 0.0;

    while (tidx < ntiles) {
        __syncthreads();                                   // previous tile done with sA / sD
        stage.commit(sA, s_coef, TT_C, H2, W2, y0 - 1, x0, DM_LOAD_IDENT);
        __syncthreads();
        const int cb = b, cy0 = y0;
        const int colx = x0 + lane;                        // this lane's d2 column (WIDE: may lie outside the image / be halo)
        const bool colin = !WIDE || (unsigned)colx < (unsigned)W2;
        const bool ownl = !WIDE || (lane >= TT_HALO && lane < TT_W - TT_HALO && colx < W2);
        const int next = tidx + gridDim.x;
        if (next < ntiles) {
            tile_of(next, b, y0, x0);
            stage.issue(in, b, TT_C, H2, W2, y0 - 1, x0);
        }
        // x / mask rows this wave needs in phase B: requested now, they land while phase A computes
        f32x2 xr[DROWS / 4][NIN];       // (the optional batch_mask is read in phase B: rare path, no registers kept for it)
#pragma unroll
        for (int j = 0; j < DROWS / 4; ++j) {
            const int oy = 2 * cy0 + wave + 4 * j;
#pragma unroll
            for (int c = 0; c < NIN; ++c) {
                xr[j][c] = (f32x2){0.f, 0.f};
                if (x && colin) xr[j][c] = *reinterpret_cast<const f32x2 *>(x + ((cb * NIN + c) * OH + oy) * OW + 2 * colx);
            }
        }
        // ---- phase A: d4[co = wave] = relu(dec.4(d2)) for the 16 output rows of this tile -> LDS ---------------
        {
            RowVals P, C, N;                             // d2 rows y-1, y and (read one row ahead of its use) y+1; sA row 0 <-> d2 row y0-1
            load_row(sA, APS, 0, lane, P);
            load_row(sA, APS, 1, lane, N);
#pragma unroll
            for (int pr = 0; pr <= TT_TH; ++pr) {        // y = y0 + pr -> d4 rows 2y-1 (lo) and 2y (hi)
                C = N;
                if (pr < TT_TH) load_row(sA, APS, pr + 2, lane, N);
                __builtin_amdgcn_sched_barrier(0);
                f32x2 lo, hi;
                convT_pair(P, C, wp, bias4, lo, hi);
                float *row = sD + wave * DPS + 2 * pr * TT_DRS + 2 * lane + 4;
                if (pr > 0) *reinterpret_cast<f32x2 *>(row - TT_DRS) = relu2(lo);
                if (pr < TT_TH) *reinterpret_cast<f32x2 *>(row) = relu2(hi);
                P = C;
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();
        // ---- phase B: dec.6 (1x1) + loss; one wave per output row, one lane per column pair --------------------
        float tl = 0.f;
#pragma unroll
        for (int j = 0; j < DROWS / 4; ++j) {
            const int R = wave + 4 * j;
            const int oy = 2 * cy0 + R;
            f32x2 dv[TT_C];
#pragma unroll
            for (int co = 0; co < TT_C; ++co) dv[co] = *reinterpret_cast<const f32x2 *>(sD + co * DPS + R * TT_DRS + 2 * lane + 4);
#pragma unroll
            for (int c = 0; c < NIN; ++c) {
                f32x2 o = {b6r[c], b6r[c]};
#pragma unroll
                for (int co = 0; co < TT_C; ++co) o += w6r[c][co] * dv[co];
                const int off = ((cb * NIN + c) * OH + oy) * OW + 2 * colx;
                if (ownl) *reinterpret_cast<f32x2 *>(dec + off) = o;
                if (x && ownl) {
                    f32x2 t = o - xr[j][c];
                    if (mask) {
                        const f32x2 mv = *reinterpret_cast<const f32x2 *>(mask + ((cb * MC + (MC == 1 ? 0 : c)) * OH + oy) * OW + 2 * colx);
                        t = o * mv - xr[j][c] * mv;
                    }
                    tl += (t.x * t.x + t.y * t.y) * ivar[c];
                }
            }
        }
        loss += (double)tl;
        tidx = next;
    }
    const double tot = block_sum(loss, s_red);
    if (threadIdx.x == 0 && loss_slabs) loss_slabs[blockIdx.x] = tot;
}

// Weight gradient of dec.4 on the MFMA with a full 16x16 tile.  R[ci][co][ky][kx] = sum_{y,x} d2[ci][y][x] *
// g4[co][2y-1+ky][2x-1+kx]; writing ky = 2*sy + py, kx = 2*sx + px and summing over Y = y + sy, X = x + sx instead:
//     R[ci][co][2sy+py][2sx+px] = sum_{Y,X} d2[ci][Y-sy][X-sx] * g4[co][2Y-1+py][2X-1+px]
// so M = (ci, sy, sx) = 16 rows of shifted d2, N = (co, py, px) = 16 columns of g4 at even/odd offsets, K = the
// (TH+1) x (W+1) positions (Y, X) of the tile.  One MFMA per four positions (a [ci] x [co,ky,kx] mapping needs four:
// only 4 of its 16 M rows are real).  Terms whose d2 row/column falls outside the tile's own rows / the image are
// zeroed on the A side; the g4 tile has zero padding columns, and its halo rows belong to the neighbour's terms.
// Steps S0 .. S0+NS-1 of one position row (step s covers X = 4s .. 4s+3; s = 16 is X = 64, only sx = 1 is real).
template <int NS, bool FIRST, bool LAST>
__device__ __forceinline__ void wgrad_steps(const float *__restrict__ sA, const float *__restrict__ sG, int aoff, int boff,
                                            bool first_bad, bool last_ok, f32x4 (&acc)[2])
{
    // operands requested PD steps ahead (PD + 1 a power of two).  One step: three ahead measured SLOWER (0.2746 -> 0.2816 ms
    // per launch in the step) -- the other wave of the SIMD fills a matrix instruction's wait, more reads in flight only
    // queue ahead of its LDS traffic
    constexpr int PD = 1;
    float a[PD + 1], bq[PD + 1];
#pragma unroll
    for (int s = 0; s < PD && s < NS; ++s) { a[s] = sA[aoff + 4 * s]; bq[s] = sG[boff + 8 * s]; }
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        if (s + PD < NS) { a[(s + PD) & PD] = sA[aoff + 4 * (s + PD)]; bq[(s + PD) & PD] = sG[boff + 8 * (s + PD)]; }
        __builtin_amdgcn_sched_barrier(0);
        float av = a[s & PD];
        if (FIRST && s == 0) av = first_bad ? 0.f : av;
        if (LAST && s == NS - 1) av = last_ok ? av : 0.f;
        acc[s & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bq[s & PD], acc[s & 1], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// Resident workgroups per CU the backward kernel is built for (launch bound = register budget: 3 -> 168, 2 -> 256).  Three
// where the instantiation fits 168 registers without scratch (profiles/r05_kernel_resources.txt), two elsewhere.
template <int NIN, bool FUSED, bool WIDE, bool MASKED, bool EDGE = false>
constexpr int tail_bwd_occ()
{
    return (!WIDE && !EDGE && NIN <= 2) ? 3 : 2;
}

#ifdef DM_MEASURE
#define DT_DBG_PARAM , int dbg
#define DT_DBG_ARG , tail_dbg()
static int tail_dbg()
{
    static const int v = [] { const char *e = getenv("DM_DEC_TAIL_DBG"); return e ? atoi(e) : 0; }();
    return v;
}
#else
#define DT_DBG_PARAM
#define DT_DBG_ARG
#endif

// ================================================================================== backward
// FUSED (training pass, dm_dec_tail_train): decoded is not read but formed in phase B from the recomputed d4 tile
// (dec.6 is 1x1), the reconstruction-loss partials are taken there too, and `decoded` never exists in HBM.
// MASKED: batch_mask given (vq_vae.py:320-321).  A template flag, not a branch: with the uniform branch in the row loop of
// phase B the kernel needs 223 registers, without it 159 -- three resident workgroups per CU instead of two.
// EDGE (round 5; d2 a multiple of 64 wide, FUSED): tiles of 64 OWNED columns side by side instead of WIDE's 56 + halo (128
// columns: two tiles instead of three).  What a tile needs from across its vertical seams is one d2 column (for the d4 of its
// own first / last output column) and one g4 column (for its data and weight gradient): the d2 columns x0 - 1 and x0 + 64 come
// in beside the tile (sE), the seam terms of phase A and the d4 / g4 of the two PAD columns of the g4 tile (ox = 2 x0 - 1 and
// 2 x0 + 128, zero in the full-row form) are worked out by a few lanes per wave, and the lanes 0 / 63 pick their seam sums up
// from LDS where the full-row form's lane shifts return the zero padding.  At the image border the seam values are zeros.
template <int NIN, bool FUSED, bool WIDE, bool MASKED, bool EDGE = false>
__global__ __launch_bounds__(DM_BLOCK, (tail_bwd_occ<NIN, FUSED, WIDE, MASKED, EDGE>()))
void dec_tail_backward_kernel(const float *__restrict__ d2, const float *__restrict__ w4, const float *__restrict__ b4,
                              const float *__restrict__ w6, const float *__restrict__ b6, double *__restrict__ loss_slabs,
                              const float *__restrict__ decp, const float *__restrict__ x,
                              const float *__restrict__ mask, int MC, const float *__restrict__ cvar,
                              const float *__restrict__ gscale_dev, float *__restrict__ g2, double *__restrict__ part,
                              float *__restrict__ wslabs, int H2, int ntiles, double inv_count, int nslabs, int W2arg,
                              int tiles_x DT_DBG_PARAM)
{
#ifndef DM_MEASURE
    constexpr int dbg = 0;
#endif
    // dbg (DM_DEC_TAIL_DBG, measurement build only; results are then wrong, the time is what is read): 1 no phase A products,
    // 2 no phase B, 4 no phase 3 (data gradient), 8 no phase 4 (weight-gradient matrix instructions), 16 no loads / commits
    // after the first tile, 32 no workgroup barriers inside the tile loop
#define DT_SYNC() do { if (!(dbg & 32)) __syncthreads(); } while (0)
    constexpr int AROWS = TT_TH + 4, APS = AROWS * TT_W + 4;     // d2 rows y0-2 .. y0+TH+1
    constexpr int GROWS = 2 * TT_TH + 2, GPS = GROWS * TT_DRS;   // g4 rows 2*y0-1 .. 2*y0+2*TH ; col j <-> ox = j - 4
    constexpr int NP = NIN * TT_C + NIN + TT_C + TT_C;           // dW6 | db6 | db4 | db2
    constexpr int NV = NIN * TT_C + NIN + TT_C;                  // per-lane partials of phase B
    constexpr int ZOFF = TT_C * APS, ZLEN = 72;                  // zeros behind the tile: a weight-gradient row that reads as 0 (17 steps of 4)
    __shared__ __attribute__((aligned(16))) float sA[TT_C * APS + ZLEN];
    __shared__ __attribute__((aligned(16))) float sG[TT_C * GPS];
    __shared__ __attribute__((aligned(16))) float s_coef[DM_COEF_MAX_C * 4];
    __shared__ double s_part[4][NP];
    static_assert(!EDGE || (FUSED && !WIDE), "EDGE: the training kernel, tiles of 64 owned columns");
    constexpr int PADL = 3, PADR = TT_DRS - 4;                                    // g4-tile columns of ox = 2 x0 - 1 and 2 x0 + 128
    __shared__ __attribute__((aligned(16))) float s_w4e[EDGE ? TT_C * TT_C * 16 : 4];   // dec.4's weights for the seam lanes
    __shared__ float sE[EDGE ? 2 * TT_C * AROWS : 1];                             // d2 columns x0 - 1 | x0 + 64: [side][ci][row]
    // what lane 0 / lane 63 add to their output pair of phase A, as pairs (left seam: (e, 0); right seam: (0, e)), and a third
    // array of zeros for the lanes between them: [kind][co][g4-tile row]; likewise the seam sums of phase 3: [kind][ci][row]
    __shared__ __attribute__((aligned(8))) float sPA[EDGE ? 3 * TT_C * GROWS * 2 : 2];
    __shared__ float sP3[EDGE ? 3 * TT_C * TT_TH : 1];

    const int lane = threadIdx.x & 63, m = lane & 15, kq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int W2 = (WIDE || EDGE) ? W2arg : TT_W;
    const int ekind = lane == 0 ? 0 : (lane == TT_W - 1 ? 1 : 2);
    const int tiles_y = H2 / TT_TH, OH = 2 * H2, OW = 2 * W2;
    Operand in;
    in.p0 = d2; in.p1 = nullptr; in.coef = nullptr; in.coef_bstride = 0; in.mode = DM_LOAD_IDENT; in.ones = 0;

    TileStage<TT_C, AROWS, TT_W / 4, TT_W, APS, false> stage;
    stage.init(H2, W2);
    // tile -> (sample, row band, column band); WIDE: lane <-> column x0 + lane, x0 = 56 * tx - 4 (header comment)
    auto tile_of = [&](int t, int &tb, int &ty0, int &tx0) {
        if (WIDE) { const int q = t / tiles_x; tx0 = (t - q * tiles_x) * TT_OWN - TT_HALO; t = q; }
        else if (EDGE) { const int q = t / tiles_x; tx0 = (t - q * tiles_x) * TT_W; t = q; }
        else tx0 = 0;
        ty0 = (t % tiles_y) * TT_TH; tb = t / tiles_y;
    };
    // EDGE: thread (side, ci, row) fetches its element of the two d2 columns beside the tile one tile ahead (0 outside the image)
    const int e_side = threadIdx.x / (TT_C * AROWS), e_ci = (threadIdx.x / AROWS) % TT_C, e_row = threadIdx.x % AROWS;
    auto edge_load = [&](bool live, int tb, int ty0, int tx0) {
        const int y = ty0 - 2 + e_row, col = e_side ? tx0 + TT_W : tx0 - 1;
        const bool ok = live && threadIdx.x < 2 * TT_C * AROWS && (unsigned)y < (unsigned)H2 && (unsigned)col < (unsigned)W2;
        const float v = d2[ok ? ((tb * TT_C + e_ci) * H2 + y) * W2 + col : 0];
        return ok ? v : 0.f;
    };
    float edge_v = 0.f;
    int tidx = blockIdx.x, b = 0, y0 = 0, x0 = 0;
    if (tidx < ntiles) {
        tile_of(tidx, b, y0, x0);
        stage.issue(in, b, TT_C, H2, W2, y0 - 2, x0);
        if constexpr (EDGE) edge_v = edge_load(true, b, y0, x0);
    }
    if constexpr (EDGE) {
        s_w4e[threadIdx.x] = w4[threadIdx.x];
        for (int i = threadIdx.x; i < TT_C * GROWS * 2; i += DM_BLOCK) sPA[2 * TT_C * GROWS * 2 + i] = 0.f;
        if (threadIdx.x < TT_C * TT_TH) sP3[2 * TT_C * TT_TH + threadIdx.x] = 0.f;
    }
    if (threadIdx.x < TT_C) *reinterpret_cast<f32x4 *>(s_coef + threadIdx.x * 4) = (f32x4){1.f, 0.f, 0.f, -__builtin_inff()};
    if (threadIdx.x < ZLEN) sA[ZOFF + threadIdx.x] = 0.f;
    // the padding columns of the g4 tile (ox = -4..-1 and 128..131) are zero for every tile
    for (int i = threadIdx.x; i < TT_C * GROWS * 2; i += DM_BLOCK) {
        const int row = i >> 1, side = i & 1;
        *reinterpret_cast<f32x4 *>(sG + row * TT_DRS + (side ? TT_DRS - 4 : 0)) = (f32x4){0.f, 0.f, 0.f, 0.f};
    }

    // wave-uniform weights: phase A (recompute, co = wave): W4[ci][wave][ky][kx]; phase 3 (data gradient, ci = wave):
    // W4[wave][co][ky][kx] (64 contiguous floats)
    const int wv = wave_index_vgpr();
    const float bias4 = b4[wv];
    const float gs = (float)(2.0 * inv_count) * gscale_dev[0];
    float w6r[NIN][TT_C], gsv[NIN], b6r[NIN], ivar[NIN];
#pragma unroll
    for (int c = 0; c < NIN; ++c) {
        gsv[c] = gs / cvar[c];
        b6r[c] = (FUSED && b6) ? b6[c] : 0.f; ivar[c] = 1.f / cvar[c];
#pragma unroll
        for (int co = 0; co < TT_C; ++co) w6r[c][co] = w6[c * TT_C + co];
    }

    // weight-gradient (MFMA) operands (wgrad_steps): lane (m, kq) feeds A row m = (ci, sy, sx) and B column m = (co, py, px)
    // at position X = 4*step + kq of position row Yr = Y - y0:  sA row of d2 row Y-sy is Yr+2-sy, g4-tile row is 2*Yr+py,
    // g4-tile column of ox = 2X-1+px is 2X+3+px
    const int wsy = (m >> 1) & 1, wsx = m & 1;
    const int wa_base = (m >> 2) * APS + (2 - wsy) * TT_W + kq - wsx;
    const int wb_base = (m >> 2) * GPS + wsy * TT_DRS + 3 + wsx + 2 * kq;
    const bool w_first_bad = wsx == 1 && kq == 0;        // X = 0 with sx = 1: column -1
    const bool w_last_ok = wsx == 1 && kq == 0;          // X = 64: only column 63 of the sx = 1 rows
    f32x4 wacc[2];
    wacc[0] = (f32x4){0.f, 0.f, 0.f, 0.f}; wacc[1] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // per-lane partial sums over this workgroup's tiles (a few hundred fp32 terms each; the cross-lane / cross-
    // workgroup reductions are done in double): dW6[c][co] | db6[c] | db4[co] ; db2[ci = wave]
    f32x2 pv[NV];                                     // (even column, odd column) partials, added at the end
    float pb2 = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) pv[k] = (f32x2){0.f, 0.f};

    // decoded / x rows of the g4 rows this wave handles in phase B (gr = wave + 4j).  They are requested one tile
    // ahead and only CONSUMED in phase B (consuming them earlier would put the s_waitcnt right after the issue).
    // AHEAD = false: requested at the start of phase B instead, their latency exposed once per tile.  Where the rows in
    // flight through the phases that hold the 64 weights put the kernel at the 256-register limit with spills in the
    // loop: decoded AND x rows of three or four channels (60-80 registers), and the x rows alone in the full-row form
    // (two more matrix steps per row than the WIDE form).
    constexpr bool AHEAD = NIN <= 2 || (FUSED && (WIDE || EDGE));
    constexpr int BR = (GROWS + 3) / 4;
    f32x2 rdv[BR][NIN], rxv[BR][NIN];
    // one (row, channel) at a time through buffer descriptors rebased to the sample: rows outside the image (and
    // every row when there is no next tile: empty descriptor) get an out-of-range offset and read as 0.  The
    // requests are spread over the MFMA (weight-gradient) phase instead of going out in one burst (a CU keeps only so many bytes
    // in flight; a burst stalls the wave at the issue point).
    struct RowCtx { __amdgpu_buffer_rsrc_t rx, rd; int ty0, tx0; };
    auto rows_begin = [&](bool live, int tb, int ty0, int tx0) {
        RowCtx rc;
        const long long se = (long long)NIN * OH * OW;
        const int bytes = live ? (int)(se * 4) : 0;
        rc.rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(x + se * tb), 0, bytes, 0x00020000);
        rc.rd = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>((FUSED ? x : decp) + se * tb), 0, FUSED ? 0 : bytes,
                                                  0x00020000);
        rc.ty0 = ty0; rc.tx0 = tx0;
        return rc;
    };
    auto issue_row = [&](const RowCtx &rc, int j, int c) {
        const int gr = wave + 4 * j, oy = 2 * rc.ty0 - 1 + gr;
        const int cx = rc.tx0 + lane;
        const bool ok = gr < GROWS && (unsigned)oy < (unsigned)OH && (!WIDE || (unsigned)cx < (unsigned)W2);
        const int voff = ok ? ((c * OH + oy) * OW + 2 * cx) * 4 : 0x7ffffff0;
        if (!FUSED) rdv[j][c] = __builtin_amdgcn_raw_buffer_load_b64(rc.rd, voff, 0, 0);
        rxv[j][c] = __builtin_amdgcn_raw_buffer_load_b64(rc.rx, voff, 0, 0);
    };
    // EDGE: x at the two pad columns of the g4 tile, lane = (g4 row, side), for the wave that turns their d4 into g4 (phase B)
    float rxe[EDGE ? NIN : 1];
    const int p_gr = lane >> 1, p_side = lane & 1;
    auto issue_pad = [&](const RowCtx &rc) {
        if constexpr (EDGE) {
            const int oy = 2 * rc.ty0 - 1 + p_gr, px = p_side ? 2 * (rc.tx0 + TT_W) : 2 * rc.tx0 - 1;
            const bool ok = p_gr < GROWS && (unsigned)oy < (unsigned)OH && (unsigned)px < (unsigned)OW;
#pragma unroll
            for (int c = 0; c < NIN; ++c)
                rxe[c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rc.rx, ok ? ((c * OH + oy) * OW + px) * 4 : 0x7ffffff0, 0, 0));
        }
    };
    if constexpr (EDGE) {
        const RowCtx rc0 = rows_begin(tidx < ntiles, b, y0, x0);
        if (wave == 3) issue_pad(rc0);
    }
    if constexpr (AHEAD) {
        const RowCtx rc0 = rows_begin(tidx < ntiles, b, y0, x0);
#pragma unroll
        for (int j = 0; j < BR; ++j)
#pragma unroll
            for (int c = 0; c < NIN; ++c) issue_row(rc0, j, c);
    }
    double loss = 0.0;

    bool first_tile = true;
    while (tidx < ntiles) {
        DT_SYNC();                                         // previous tile done with sA and sG
        if (!(dbg & 16) || first_tile)
            stage.template commit<false, true>(sA, s_coef, TT_C, H2, W2, y0 - 2, x0, DM_LOAD_IDENT);   // 192 staging threads: waves 0..2
        if constexpr (EDGE) {
            if (threadIdx.x < 2 * TT_C * AROWS) sE[threadIdx.x] = edge_v;
        }
        first_tile = false;
        DT_SYNC();
        const int cb = b, cy0 = y0, cx0 = x0;
        const int colx = x0 + lane;                        // this lane's d2 column (WIDE: may be halo / outside the image)
        const bool colin = !WIDE || (unsigned)colx < (unsigned)W2;
        const bool ownl = !WIDE || (lane >= TT_HALO && lane < TT_W - TT_HALO && colx < W2);
        const int next = tidx + gridDim.x;
        if (next < ntiles) tile_of(next, b, y0, x0);
        const auto scx = stage.begin(in, next < ntiles && !(dbg & 16), b, TT_C, H2, W2, y0 - 2, x0);   // requested during phase A
        if constexpr (EDGE) edge_v = edge_load(next < ntiles, b, y0, x0);

        // ---- phase A: recompute d4[co = wave] on position rows y0-1 .. y0+TH -> sG rows 2*pr+py-1 -----------------
        // Two passes, one per output-row parity (d4 row 2y-1+par reads P = d2 row y-1 with ky = 2 + par and C = row y with
        // ky = par): a pass holds 32 weights, in SGPRs (wave-uniform; see phase 3), as the pairs (w0, w1), (w2, w3) they are in
        // memory.  A lane's value v = in[x] gives  T += v * (w0, w1) = (to the left neighbour's odd column, own even column),
        // U += v * (w2, w3) = (own odd column, to the right neighbour's even column).
        if constexpr (EDGE) {
            // lane = (g4-tile row gr, side): the seam term of the tile's own first / last output column (what the missing
            // neighbour lane's side sum would have carried) and d4 of the pad column beyond it, for co = wave.
            // d4 row 2y-1+par (y = cy0 + pr) reads d2 rows y-1 (ky = par + 2) and y (ky = par): sA / sE rows pr + 1, pr + 2.
            //   left  (side 0): seam = sum in[x0-1] w[ky][3];  pad ox = 2x0-1:   sum in[x0-1] w[ky][2] + in[x0]    w[ky][0]
            //   right (side 1): seam = sum in[x0+64] w[ky][0]; pad ox = 2x0+128: sum in[x0+64] w[ky][1] + in[x0+63] w[ky][3]
            if (lane < 2 * GROWS) {
                const int gr = lane >> 1, side = lane & 1, par = gr & 1, pr = gr >> 1;
                const int kxE = side ? 0 : 3, kxA = side ? 1 : 2, kxB = side ? 3 : 0, colB = side ? TT_W - 1 : 0;
                const bool outside = side ? cx0 + TT_W >= W2 : cx0 == 0;
                float e = 0.f, dp = bias4;
#pragma unroll
                for (int ci = 0; ci < TT_C; ++ci)
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        const int row = pr + 2 - t;
                        const float ve = sE[(side * TT_C + ci) * AROWS + row], vb = sA[ci * APS + row * TT_W + colB];
                        const float *__restrict__ wr = s_w4e + (ci * TT_C + wave) * 16 + (par + 2 * t) * 4;
                        e = __builtin_fmaf(ve, wr[kxE], e);
                        dp = __builtin_fmaf(ve, wr[kxA], dp);
                        dp = __builtin_fmaf(vb, wr[kxB], dp);
                    }
                e = outside ? 0.f : e;
                *reinterpret_cast<f32x2 *>(sPA + ((side * TT_C + wave) * GROWS + gr) * 2) = side ? (f32x2){0.f, e} : (f32x2){e, 0.f};
                sG[wave * GPS + gr * TT_DRS + (side ? PADR : PADL)] = outside ? 0.f : dm_relu(dp);
            }
        }
        {
        const f32x2 biasT = {0.f, bias4}, biasU = {bias4, 0.f};
#pragma unroll
        for (int par = 0; par < 2; ++par) {
            int woff = wave * 16 + 4 * par;
            asm volatile("" : "+s"(woff));
            f32x2 w01[TT_C][2], w23[TT_C][2];          // [ci][t]: ky = par + 2t  (t = 0: row C, t = 1: row P)
#pragma unroll
            for (int ci = 0; ci < TT_C; ++ci)
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const f32x4 q = *reinterpret_cast<const f32x4 *>(w4 + woff + ci * (TT_C * 16) + 8 * t);
                    w01[ci][t] = (f32x2){q.x, q.y};
                    w23[ci][t] = (f32x2){q.z, q.w};
                }
            RowVals P, C, N;                          // rows y-1, y and (read one row ahead of its use) y+1; sA row 0 <-> d2 row y0-2
            load_row(sA, APS, 1, lane, P);
            load_row(sA, APS, 2, lane, N);
#pragma unroll
            for (int pr = 0; pr <= TT_TH; ++pr) {     // y = y0 + pr -> g4-tile row 2pr + par (d4 row 2y-1+par)
                if (par == 0) {
                    constexpr int NE = decltype(stage)::N;
#pragma unroll
                    for (int e = 0; e < NE; ++e)
                        if (e >= pr * NE / (TT_TH + 1) && e < (pr + 1) * NE / (TT_TH + 1)) stage.issue_one(e, scx);
                }
                C = N;
                if (pr < TT_TH) load_row(sA, APS, pr + 3, lane, N);
                __builtin_amdgcn_sched_barrier(0);
                if (!(dbg & 1)) {
                    f32x2 T, U;
#pragma unroll
                    for (int ci = 0; ci < TT_C; ++ci) {
                        const f32x2 pp = P.v[ci >> 1], cc = C.v[ci >> 1];
                        const f32x2 pv2 = (ci & 1) ? __builtin_shufflevector(pp, pp, 1, 1) : __builtin_shufflevector(pp, pp, 0, 0);
                        const f32x2 cv2 = (ci & 1) ? __builtin_shufflevector(cc, cc, 1, 1) : __builtin_shufflevector(cc, cc, 0, 0);
                        if (ci == 0) { T = pv2 * w01[ci][1] + biasT; U = pv2 * w23[ci][1] + biasU; }
                        else { T += pv2 * w01[ci][1]; U += pv2 * w23[ci][1]; }
                        T += cv2 * w01[ci][0]; U += cv2 * w23[ci][0];
                    }
                    f32x2 o;
                    o.x = T.y + lane_from_left(U.y);
                    o.y = U.x + lane_from_right(T.x);
                    if constexpr (EDGE) o += *reinterpret_cast<const f32x2 *>(sPA + ((ekind * TT_C + wave) * GROWS + 2 * pr + par) * 2);
                    *reinterpret_cast<f32x2 *>(sG + wave * GPS + (2 * pr + par) * TT_DRS + 2 * lane + 4) = relu2(o);
                }
                P = C;
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        }
        DT_SYNC();

        // ---- phase B: g4 = (W6^T g_dec) * (d4 > 0) in place; dW6 / db6 / db4 partial sums ---------------------------
        if constexpr (!AHEAD) {
            const RowCtx rcc = rows_begin(!(dbg & 16), cb, cy0, cx0);
#pragma unroll
            for (int j = 0; j < BR; ++j)
#pragma unroll
                for (int c = 0; c < NIN; ++c) issue_row(rcc, j, c);
        }
        f32x2 tl = {0.f, 0.f};
        // (the d4 values of a row are read one row ahead of their use; only the last of a wave's rows can lie past the
        //  tile -- wave + 16 >= GROWS for waves 2, 3 -- so the others need no test and form one block with their neighbours)
        static_assert(4 * (BR - 1) + 3 >= GROWS && 4 * (BR - 2) + 3 < GROWS, "only the last row of a wave may be absent");
        // (one or two input channels; with three or four the longer live ranges spill)
        constexpr bool PF = NIN <= 2;
        f32x2 d4n[TT_C];
        if constexpr (PF) {
#pragma unroll
            for (int co = 0; co < TT_C; ++co) d4n[co] = *reinterpret_cast<const f32x2 *>(sG + co * GPS + wave * TT_DRS + 2 * lane + 4);
        }
#pragma unroll
        for (int j = 0; j < BR; ++j) {
            const int gr = wave + 4 * j;
            if (dbg & 2) break;
            if ((PF && j < BR - 1) || gr < GROWS) {
                const int oy = 2 * cy0 - 1 + gr;
                const bool live = oy >= 0 && oy < OH && colin;
                // rows 2*y0 .. 2*y0+2*TH-1 (WIDE: and the owned columns) belong to this tile: the halo gets its g4 but
                // adds nothing to the sums
                // (full-row form: only a wave's first and last row can be halo -- gr = 0 or 2 TH + 1 -- or lie outside the image;
                //  the rows between them are owned and live by construction and skip the 0 / 1 factors)
                const bool maybe_halo = WIDE || j == 0 || j == BR - 1;
                const float own = !maybe_halo ? 1.f : ((gr >= 1 && gr <= 2 * TT_TH && ownl) ? 1.f : 0.f);
                f32x2 gd[NIN], gdo[NIN], d4v[TT_C];
#pragma unroll
                for (int co = 0; co < TT_C; ++co)
                    d4v[co] = PF ? d4n[co] : *reinterpret_cast<const f32x2 *>(sG + co * GPS + gr * TT_DRS + 2 * lane + 4);
                if (PF && j + 1 < BR) {
                    const int gn = gr + 4 < GROWS ? gr + 4 : GROWS - 1;      // (an absent last row: any row, never used)
#pragma unroll
                    for (int co = 0; co < TT_C; ++co)
                        d4n[co] = *reinterpret_cast<const f32x2 *>(sG + co * GPS + gn * TT_DRS + 2 * lane + 4);
                }
#pragma unroll
                for (int c = 0; c < NIN; ++c) {
                    f32x2 dv = rdv[j][c];
                    if (FUSED) {                         // decoded = dec.6(d4); rows outside the image carry no gradient
                        dv = (f32x2){b6r[c], b6r[c]};
#pragma unroll
                        for (int co = 0; co < TT_C; ++co) dv += w6r[c][co] * d4v[co];
                    }
                    f32x2 t = dv - rxv[j][c];            // not FUSED: rows outside the image were loaded as 0
                    // (own and live are 0 / 1 factors: folded into the channel's constants -- the products are the same to the
                    //  bit and the kernel is bound by its vector instruction count; without WIDE both are wave-uniform)
                    const float oiv = maybe_halo ? own * ivar[c] : ivar[c];
                    if constexpr (MASKED) {
                        // (read unconditionally, from the nearest row / column inside the image where the position lies outside:
                        //  there dv and x are 0 (not FUSED) or the row's factors own and gsv are (FUSED), so any finite mask
                        //  value gives the same zeros)
                        const int oyc = oy < 0 ? 0 : (oy >= OH ? OH - 1 : oy);
                        const int cxc = !WIDE ? colx : (colx < 0 ? 0 : (colx >= W2 ? W2 - 1 : colx));
                        const f32x2 mv = *reinterpret_cast<const f32x2 *>(mask + ((cb * MC + (MC == 1 ? 0 : c)) * OH + oyc) * OW + 2 * cxc);
                        t = dv * mv - rxv[j][c] * mv;
                        if (FUSED) tl += (t * t) * oiv;
                        t = t * mv;
                    } else if (FUSED) {
                        tl += (t * t) * oiv;
                    }
                    gd[c] = t * ((!FUSED || !maybe_halo || live) ? gsv[c] : 0.f);   // FUSED: rows outside the image carry no gradient
                    gdo[c] = maybe_halo ? gd[c] * own : gd[c];
                    pv[NIN * TT_C + c] += gdo[c];
                }
#pragma unroll
                for (int co = 0; co < TT_C; ++co) {
                    float *pd = sG + co * GPS + gr * TT_DRS + 2 * lane + 4;
                    const f32x2 d4 = d4v[co];
                    f32x2 g4 = w6r[0][co] * gd[0];
#pragma unroll
                    for (int c = 1; c < NIN; ++c) g4 += w6r[c][co] * gd[c];
#pragma unroll
                    for (int c = 0; c < NIN; ++c) pv[c * TT_C + co] += gdo[c] * d4;
                    // (a packed multiply by clamp(d4 * inf) does this in two instructions instead of four -- measured: no gain, a
                    //  v_pk_* costs the port twice a plain instruction)
                    g4.x = d4.x > 0.f ? g4.x : 0.f;
                    g4.y = d4.y > 0.f ? g4.y : 0.f;
                    if (maybe_halo) pv[NIN * TT_C + NIN + co] += g4 * own;
                    else pv[NIN * TT_C + NIN + co] += g4;
                    *reinterpret_cast<f32x2 *>(pd) = g4;
                }
            }
        }
        if constexpr (EDGE) {
            // the pad columns' d4 -> g4 (a neighbouring tile owns these pixels: no sums), lane = (g4-tile row, side), by the wave
            // with the fewest rows of its own; a pad column outside the image holds d4 = 0 and stays 0
            if (wave == 3 && lane < 2 * GROWS && !(dbg & 2)) {
                const int oy = 2 * cy0 - 1 + p_gr, pj = p_side ? PADR : PADL;
                const bool live = oy >= 0 && oy < OH;
                float d4p[TT_C], gdp[NIN];
#pragma unroll
                for (int co = 0; co < TT_C; ++co) d4p[co] = sG[co * GPS + p_gr * TT_DRS + pj];
#pragma unroll
                for (int c = 0; c < NIN; ++c) {
                    float dv = b6r[c];
#pragma unroll
                    for (int co = 0; co < TT_C; ++co) dv += w6r[c][co] * d4p[co];
                    float t = dv - rxe[c];
                    if constexpr (MASKED) {
                        const int px = p_side ? 2 * (cx0 + TT_W) : 2 * cx0 - 1;
                        const int oyc = oy < 0 ? 0 : (oy >= OH ? OH - 1 : oy), pxc = px < 0 ? 0 : (px >= OW ? OW - 1 : px);
                        const float mv = mask[((cb * MC + (MC == 1 ? 0 : c)) * OH + oyc) * OW + pxc];
                        t = (dv * mv - rxe[c] * mv) * mv;
                    }
                    gdp[c] = t * (live ? gsv[c] : 0.f);
                }
#pragma unroll
                for (int co = 0; co < TT_C; ++co) {
                    float g4 = w6r[0][co] * gdp[0];
#pragma unroll
                    for (int c = 1; c < NIN; ++c) g4 += w6r[c][co] * gdp[c];
                    sG[co * GPS + p_gr * TT_DRS + pj] = d4p[co] > 0.f ? g4 : 0.f;
                }
            }
        }
        if (FUSED) loss += (double)tl.x + (double)tl.y;
        const RowCtx rcn = rows_begin(next < ntiles && !(dbg & 16), b, y0, x0);   // next tile's rows: requested during the wgrad phase
        if constexpr (EDGE) {
            if (wave == 3) issue_pad(rcn);
        }
        DT_SYNC();

        // ---- phase 3: data gradient of dec.4 for input channel ci = wave:  g2[ci][y][x] = sum_co,ky,kx
        //      g4[co][2y-1+ky][2x-1+kx] * W4[ci][co][ky][kx], masked by d2 > 0 ------------------------------------------
        if (!(dbg & 4)) {
        // per g4 value pair mid = (col 2x, 2x+1), left = col 2x-1, right = col 2x+2:
        //   g2[x] += left*w[ky][0] + mid.x*w[ky][1] + mid.y*w[ky][2] + right*w[ky][3]
        // left / right are the neighbours' mid.y / mid.x: instead of fetching them per value, every lane also accumulates
        // side = mid * (w3, w0) -- what ITS pair contributes to the left (x) and right (y) neighbour -- and the side sums
        // cross the lanes once per output row
        // Weights in SGPRs: W4[ci = wave][co][ky][kx] is wave-uniform, and a v_pk_fma_f32 takes one scalar register PAIR as an
        // operand.  Two passes over the g4 rows, one per row parity (a row of parity par feeds ky = par and ky = par + 2 only), so
        // that a pass holds 32 weights; the offset is hidden from the optimiser inside the tile loop so that the scalar loads
        // stay in the phase (hoisted out of the loop the 96 weights of phases A and 3 would not fit the scalar file).
        // Pairs as they lie in memory, (w0, w1) and (w2, w3): a lane's pair mid = (col 2x, col 2x+1) gives
        //   R += (mid.y * w0, mid.x * w1) = (to the right neighbour, own)     S += (mid.y * w2, mid.x * w3) = (own, to the left neighbour)
        if constexpr (EDGE) {
            // lane = (output row r, side): what the pad column of g4 adds to g2 of the tile's first / last column, ci = wave
            //   left: sum_co,ky g4[co][2r+ky][ox = 2x0-1] W[ci][co][ky][0];  right: ... [ox = 2x0+128] W[ci][co][ky][3]
            if (lane < 2 * TT_TH) {
                const int r = lane >> 1, side = lane & 1;
                float sm = 0.f;
#pragma unroll
                for (int co = 0; co < TT_C; ++co)
#pragma unroll
                    for (int ky = 0; ky < 4; ++ky)
                        sm = __builtin_fmaf(sG[co * GPS + (2 * r + ky) * TT_DRS + (side ? PADR : PADL)],
                                            s_w4e[(wave * TT_C + co) * 16 + ky * 4 + (side ? 3 : 0)], sm);
                sP3[(side * TT_C + wave) * TT_TH + r] = sm;
            }
        }
        f32x2 accR[TT_TH], accS[TT_TH];                // (started by their first product, not zeroed)
#pragma unroll
        for (int par = 0; par < 2; ++par) {
            int woff = wave * (TT_C * 16) + 4 * par;
            asm volatile("" : "+s"(woff));
            f32x2 w01[TT_C][2], w23[TT_C][2];          // [co][t]: ky = par + 2t
#pragma unroll
            for (int co = 0; co < TT_C; ++co)
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const f32x4 q = *reinterpret_cast<const f32x4 *>(w4 + woff + co * 16 + 8 * t);
                    w01[co][t] = (f32x2){q.x, q.y};
                    w23[co][t] = (f32x2){q.z, q.w};
                }
            f32x2 nm[TT_C];                            // the next g4 row of this parity: LDS reads one row ahead of their use
#pragma unroll
            for (int co = 0; co < TT_C; ++co) nm[co] = *reinterpret_cast<const f32x2 *>(sG + co * GPS + par * TT_DRS + 2 * lane + 4);
#pragma unroll
            for (int i = 0; i <= TT_TH; ++i) {         // g4 row gr = 2i + par: ky = par of output row i, ky = par + 2 of row i - 1
                f32x2 mid[TT_C];
#pragma unroll
                for (int co = 0; co < TT_C; ++co) mid[co] = nm[co];
                if (i < TT_TH) {
#pragma unroll
                    for (int co = 0; co < TT_C; ++co)
                        nm[co] = *reinterpret_cast<const f32x2 *>(sG + co * GPS + (2 * i + 2 + par) * TT_DRS + 2 * lane + 4);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int co = 0; co < TT_C; ++co) {
                    const f32x2 sw = __builtin_shufflevector(mid[co], mid[co], 1, 0);     // (mid.y, mid.x)
                    if (i < TT_TH) {
                        if (par == 0 && co == 0) { accR[i] = sw * w01[co][0]; accS[i] = sw * w23[co][0]; }
                        else { accR[i] += sw * w01[co][0]; accS[i] += sw * w23[co][0]; }
                    }
                    if (i >= 1) { accR[i - 1] += sw * w01[co][1]; accS[i - 1] += sw * w23[co][1]; }
                }
                if (par == 1 && i >= 1) {              // output row r = i - 1 is complete
                    const int r = i - 1;
                    const float dd = sA[wave * APS + (r + 2) * TT_W + lane];
                    float sum = (accR[r].y + accS[r].x) + (lane_from_left(accR[r].x) + lane_from_right(accS[r].y));
                    if constexpr (EDGE) sum += sP3[(ekind * TT_C + wave) * TT_TH + r];
                    const float v = (dd > 0.f && ownl) ? sum : 0.f;
                    if (ownl) g2[((cb * TT_C + wave) * H2 + cy0 + r) * W2 + colx] = v;
                    pb2 += v;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        }

        // ---- phase 4 (MFMA): weight gradient of dec.4 (wgrad_steps).  Position rows Yr = 0..TH: this wave takes rows
        //      wave and wave+4 whole and a quarter of row TH.  Row 0 has no own d2 row for sy = 1, row TH none for sy = 0.
        if (!(dbg & 8)) {
            constexpr int NR = BR * NIN;                   // next tile's (row, channel) requests, a third before each part
            auto issue_part = [&](int part) {
                if constexpr (AHEAD) {
#pragma unroll
                    for (int e = 0; e < NR; ++e)
                        if (e >= part * NR / 3 && e < (part + 1) * NR / 3) issue_row(rcn, e / NIN, e % NIN);
                }
            };
            // rows without an own d2 row for this lane's sy read zeros instead: the A operand's address goes to the zeroed
            // tail of sA (one select per row call, not a multiplication per step)
            const bool z0 = wave == 0 && wsy == 1, z8 = wsy != 1;
            if constexpr (!WIDE) {
                issue_part(0);
                wgrad_steps<17, true, true>(sA, sG, z0 ? ZOFF : wa_base + wave * TT_W, wb_base + 2 * wave * TT_DRS, w_first_bad,
                                            w_last_ok, wacc);
                issue_part(1);
                wgrad_steps<17, true, true>(sA, sG, wa_base + (wave + 4) * TT_W, wb_base + 2 * (wave + 4) * TT_DRS,
                                            w_first_bad, w_last_ok, wacc);
                issue_part(2);
                const int a8 = z8 ? ZOFF : wa_base + TT_TH * TT_W + 16 * wave, b8 = wb_base + 2 * TT_TH * TT_DRS + 32 * wave;
                if (wave < 3)
                    wgrad_steps<4, true, false>(sA, sG, a8, b8, w_first_bad && wave == 0, false, wacc);
                else
                    wgrad_steps<5, false, true>(sA, sG, a8, b8, false, w_last_ok, wacc);
            } else {
                // owned d2 columns 4 .. 59 of the tile: the term (X, sx) has its d2 column at X - sx, so steps 1 .. 15 with
                // X = 4 (sx = 1: column 3) struck from the first and only (X = 60, sx = 1: column 59) kept of the last --
                // the full-row form's edge handling, one step in.  Columns beyond the image hold zeros in sA.
                issue_part(0);
                wgrad_steps<15, true, true>(sA, sG, z0 ? ZOFF : wa_base + wave * TT_W + 4, wb_base + 2 * wave * TT_DRS + 8,
                                            w_first_bad, w_last_ok, wacc);
                issue_part(1);
                wgrad_steps<15, true, true>(sA, sG, wa_base + (wave + 4) * TT_W + 4, wb_base + 2 * (wave + 4) * TT_DRS + 8,
                                            w_first_bad, w_last_ok, wacc);
                issue_part(2);
                const int a8r = wa_base + TT_TH * TT_W + 16 * wave, b8 = wb_base + 2 * TT_TH * TT_DRS + 32 * wave;
                const int a8 = z8 ? ZOFF : a8r;
                if (wave == 0)
                    wgrad_steps<3, true, false>(sA, sG, z8 ? ZOFF : a8r + 4, b8 + 8, w_first_bad, false, wacc);
                else if (wave < 3)
                    wgrad_steps<4, false, false>(sA, sG, a8, b8, false, false, wacc);
                else
                    wgrad_steps<4, false, true>(sA, sG, a8, b8, false, w_last_ok, wacc);
            }
        }
        tidx = next;
    }

    // ---- partial sums: every lane holds partials of all NV phase-B sums; db2 is per wave (ci = wave) ---------------
    double pvd[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) pvd[k] = wave_sum((double)pv[k].x + (double)pv[k].y);
    const double pb2d = wave_sum((double)pb2);
    __syncthreads();
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < NV; ++k) s_part[wave][k] = pvd[k];
    }
    __syncthreads();
    for (int n = threadIdx.x; n < NV; n += DM_BLOCK) {
        const double sm = s_part[0][n] + s_part[1][n] + s_part[2][n] + s_part[3][n];
        part[((long long)blockIdx.x * NP + n) * 2 + 0] = sm;
        part[((long long)blockIdx.x * NP + n) * 2 + 1] = 0.0;
    }
    if (lane == 0) {                                        // db2[ci = wave]
        part[((long long)blockIdx.x * NP + NV + wave) * 2 + 0] = pb2d;
        part[((long long)blockIdx.x * NP + NV + wave) * 2 + 1] = 0.0;
    }

    if (FUSED) {                                            // reconstruction-loss partial of this workgroup
        const double wl = wave_sum(loss);
        __syncthreads();
        if (lane == 0) s_part[wave][0] = wl;
        __syncthreads();
        if (threadIdx.x == 0) loss_slabs[blockIdx.x] = s_part[0][0] + s_part[1][0] + s_part[2][0] + s_part[3][0];
    }

    // ---- weight-gradient slab: combine the four waves in wave order (deterministic).  Lane l, element j holds
    //      M row 4*(l>>4)+j = (ci = l>>4, sy = j>>1, sx = j&1) and N column l&15 = (co, py, px)
    float *red = sG;
    const f32x4 wsum = wacc[0] + wacc[1];
    for (int w = 0; w < 4; ++w) {
        __syncthreads();
        if (wave == w) {
            f32x4 *pp = reinterpret_cast<f32x4 *>(red + lane * 4);
            if (w == 0) *pp = wsum;
            else *pp = *pp + wsum;
        }
    }
    __syncthreads();
    float *slab = wslabs + (long long)blockIdx.x * (TT_C * TT_C * 16);
    {
        const int i = threadIdx.x, j = i & 3, l = i >> 2, n = l & 15;
        const int ky = 2 * (j >> 1) + ((n >> 1) & 1), kx = 2 * (j & 1) + (n & 1);
        slab[(l >> 4) * (TT_C * 16) + (n >> 2) * 16 + ky * 4 + kx] = red[i];
    }
    // slabs [gridDim, nslabs) exist in the caller's buffers (sized for the forward kernel's grid) but have no owner
    for (int t2 = blockIdx.x + gridDim.x; t2 < nslabs; t2 += gridDim.x) {
        for (int i = threadIdx.x; i < NP * 2; i += DM_BLOCK) part[(long long)t2 * NP * 2 + i] = 0.0;
        if (FUSED && threadIdx.x == 0) loss_slabs[t2] = 0.0;
        wslabs[(long long)t2 * (TT_C * TT_C * 16) + threadIdx.x] = 0.f;
    }
}

#undef DT_SYNC
int tail_tiles_x(int W2) { return W2 == TT_W ? 1 : (W2 + TT_OWN - 1) / TT_OWN; }               // (64 wide: one tile spans the row)
int tail_grid(int ntiles) { return ntiles < TT_MAX_GRID ? ntiles : TT_MAX_GRID; }               // forward: 3 per CU
int tail_grid_bwd(int ntiles, int occ)                                                          // backward: occ per CU
{
    int cap = 256 * occ;
#ifdef DM_MEASURE      // occupancy experiments: DM_DEC_TAIL_GRID=256 leaves one workgroup per CU
    static const int env_cap = [] { const char *e = getenv("DM_DEC_TAIL_GRID"); return e ? atoi(e) : 0; }();
    if (env_cap > 0) cap = env_cap;
#endif
    return ntiles < cap ? ntiles : cap;
}

int tail_checks(const char *who, int B, int C2, int NIN, int H2, int W2)
{
    DM_REQUIRE(C2 == TT_C, "%s: num_hiddens//4 = %d not built (4)", who, C2);
    DM_REQUIRE(NIN >= 1 && NIN <= 4, "%s: num_inputs %d not built (1..4)", who, NIN);
    DM_REQUIRE(W2 >= 4 && W2 % 4 == 0 && H2 > 0 && H2 % TT_TH == 0 && B > 0,
               "%s: d2 must be a multiple of 4 wide and a multiple of %d high (got %dx%d)", who, TT_TH, H2, W2);
    DM_REQUIRE((long long)B * 4 * (2 * H2) * (2 * W2) < (1LL << 31), "%s: tensor too large for 32-bit offsets", who);
    return 0;
}

}  // namespace

extern "C" int dm_dec_tail_supported(int C2, int NIN, int H2, int W2)
{
    return C2 == TT_C && NIN >= 1 && NIN <= 4 && W2 >= 4 && W2 % 4 == 0 && H2 > 0 && H2 % TT_TH == 0;
}

extern "C" int dm_dec_tail_num_blocks(int B, int H2, int W2)
{
    return tail_grid(B * (H2 / TT_TH) * tail_tiles_x(W2));
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
    const int tiles_x = tail_tiles_x(W2), ntiles = B * (H2 / TT_TH) * tiles_x, grid = tail_grid(ntiles);
    hipStream_t st = (hipStream_t)stream;
#define DM_TF(N_, W_) hipLaunchKernelGGL((dec_tail_forward_kernel<N_, W_>), dim3(grid), dim3(DM_BLOCK), 0, st, d2, w4, b4, w6, b6, \
                                         x, mask, mask_channels, channel_var, decoded, loss_slabs, H2, ntiles, W2, tiles_x)
    if (W2 == TT_W) {
        switch (NIN) { case 1: DM_TF(1, false); break; case 2: DM_TF(2, false); break; case 3: DM_TF(3, false); break; default: DM_TF(4, false); }
    } else {
        switch (NIN) { case 1: DM_TF(1, true); break; case 2: DM_TF(2, true); break; case 3: DM_TF(3, true); break; default: DM_TF(4, true); }
    }
#undef DM_TF
    return dm_launch_status("dm_dec_tail_forward");
}

namespace {
int tail_backward_launch(const char *who, bool fused, const float *d2, const float *w4, const float *b4, const float *w6,
                         const float *b6, double *loss_slabs, const float *decoded, const float *x, const float *mask,
                         int mask_channels, const float *channel_var, const float *gscale_dev, float *g2,
                         double *part_slabs, float *w_slabs, int B, int C2, int NIN, int H2, int W2, void *stream)
{
    DM_REQUIRE(d2 && w4 && b4 && w6 && x && channel_var && gscale_dev && g2 && part_slabs && w_slabs, "%s: NULL pointer", who);
    DM_REQUIRE(fused ? loss_slabs != nullptr : decoded != nullptr, "%s: NULL pointer", who);
    DM_REQUIRE(!mask || mask_channels == 1 || mask_channels == NIN, "%s: mask channels %d", who, mask_channels);
    if (tail_checks(who, B, C2, NIN, H2, W2)) return -1;
    const int tiles_x = tail_tiles_x(W2), ntiles = B * (H2 / TT_TH) * tiles_x, nslabs = tail_grid(ntiles);
    const double inv_count = 1.0 / ((double)B * NIN * (2.0 * H2) * (2.0 * W2));
    hipStream_t st = (hipStream_t)stream;
    // (the slab buffers are sized for the forward kernel's grid, 3 per CU: the backward grid never exceeds it)
#define DM_TB(N_, F_, W_, M_) { const int grid = tail_grid_bwd(ntiles, tail_bwd_occ<N_, F_, W_, M_>()); \
                                hipLaunchKernelGGL((dec_tail_backward_kernel<N_, F_, W_, M_>), dim3(grid), dim3(DM_BLOCK), 0, st, d2, w4, b4, \
                                                   w6, b6, loss_slabs, decoded, x, mask, mask_channels, channel_var, gscale_dev, g2, \
                                                   part_slabs, w_slabs, H2, ntiles, inv_count, nslabs, W2, tiles_x DT_DBG_ARG); }
    // EDGE: tiles of 64 owned columns (the slab buffers stay sized for the forward kernel's grid: nslabs)
#define DM_TBE(N_, M_) { const int tx_ = W2 / TT_W, nt_ = B * (H2 / TT_TH) * tx_;                                               \
                         const int grid = tail_grid_bwd(nt_, tail_bwd_occ<N_, true, false, M_, true>());                         \
                         hipLaunchKernelGGL((dec_tail_backward_kernel<N_, true, false, M_, true>), dim3(grid), dim3(DM_BLOCK), 0, st, d2, w4, \
                                            b4, w6, b6, loss_slabs, decoded, x, mask, mask_channels, channel_var, gscale_dev, g2, \
                                            part_slabs, w_slabs, H2, nt_, inv_count, nslabs, W2, tx_ DT_DBG_ARG); }
#define DM_TBEN(M_) switch (NIN) { case 1: DM_TBE(1, M_) break; case 2: DM_TBE(2, M_) break; case 3: DM_TBE(3, M_) break; default: DM_TBE(4, M_) }
#define DM_TBN(F_, W_, M_) switch (NIN) { case 1: DM_TB(1, F_, W_, M_) break; case 2: DM_TB(2, F_, W_, M_) break; case 3: DM_TB(3, F_, W_, M_) break; default: DM_TB(4, F_, W_, M_) }
#define DM_TBM(F_, W_) if (mask) { DM_TBN(F_, W_, true) } else { DM_TBN(F_, W_, false) }
    static const bool edge_off = [] { const char *e = getenv("DM_DEC_TAIL_EDGE"); return e && e[0] == '0'; }();
    if (W2 == TT_W) {
        if (fused) { DM_TBM(true, false) } else { DM_TBM(false, false) }
    } else if (fused && W2 % TT_W == 0 && !edge_off) {
        if (mask) { DM_TBEN(true) } else { DM_TBEN(false) }
    } else {
        if (fused) { DM_TBM(true, true) } else { DM_TBM(false, true) }
    }
#undef DM_TBEN
#undef DM_TBE
#undef DM_TBM
#undef DM_TBN
#undef DM_TB
    return dm_launch_status(who);
}
}  // namespace

extern "C" int dm_dec_tail_backward(const float *d2, const float *w4, const float *b4, const float *w6,
                                    const float *decoded, const float *x, const float *mask, int mask_channels,
                                    const float *channel_var, const float *gscale_dev, float *g2, double *part_slabs,
                                    float *w_slabs, int B, int C2, int NIN, int H2, int W2, void *stream)
{
    return tail_backward_launch("dm_dec_tail_backward", false, d2, w4, b4, w6, nullptr, nullptr, decoded, x, mask,
                                mask_channels, channel_var, gscale_dev, g2, part_slabs, w_slabs, B, C2, NIN, H2, W2, stream);
}

extern "C" int dm_dec_tail_train(const float *d2, const float *w4, const float *b4, const float *w6, const float *b6,
                                 const float *x, const float *mask, int mask_channels, const float *channel_var,
                                 const float *gscale_dev, float *g2, double *part_slabs, float *w_slabs,
                                 double *loss_slabs, int B, int C2, int NIN, int H2, int W2, void *stream)
{
    return tail_backward_launch("dm_dec_tail_train", true, d2, w4, b4, w6, b6, loss_slabs, nullptr, x, mask,
                                mask_channels, channel_var, gscale_dev, g2, part_slabs, w_slabs, B, C2, NIN, H2, W2, stream);
}

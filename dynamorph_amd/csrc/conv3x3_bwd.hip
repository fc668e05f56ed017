// conv3x3_bwd.hip -- backward of a 3x3 convolution on a 16 x 16 latent grid that feeds a train-mode BatchNorm: data AND
// weight gradient in one pass.
//
// Reference: enc.10 = Conv2d(num_hiddens, num_hiddens, 3, padding=1) and the first convolution of every ResidualBlock layer,
// Conv2d(num_hiddens, num_residual_hiddens, 3, padding=1) (HiddenStateExtractor/vq_vae.py:287, 205), as autograd
// differentiates them for total_loss.backward() (run_training.py:406): aten::convolution_backward (input and weight) with
// BatchNorm's backward folded into the output gradient's load, the ReLU mask (and residual join) of the layer below and the
// reductions its BatchNorm backward needs in the epilogue.
//
// As two kernels (conv3x3_kernel + wgrad_kernel) each layer staged its output gradient -- two tensors of CD channels -- and its
// input twice.  A 16 x 16 latent with its zero padding fits LDS whole, so here a workgroup keeps ONE patch at a time,
//     da  [CD][18][24]   = c0*dy + c1*y + c2 inside the image, 0 in the padding      (BatchNorm backward; dm_operand AFFINE2)
//     T   [16][18][24]   = relu(c0x*x + c2x) (or relu(x)) inside, 0 in the padding   (what the forward multiplied)
// and runs both products on v_mfma_f32_16x16x4_f32:
//     data gradient    dx[ci][y][x] = [T > 0] * sum_{co,ky,kx} da[co][y+1-ky][x+1-kx] * W[co][ci][ky][kx]  (+ resid)
//                      M = the 16 positions of a row, N = ci, K = (tap, co): 9 * CD / 4 steps
//     weight gradient  dW[co][ci][ky][kx] = sum_{y,x} da[co][y][x] * T[ci][y+ky-1][x+kx-1]
//                      M = co (CD / 16 tiles), N = (ci, ky, kx) = 9 tiles, K = positions, four consecutive x per step
// plus the (sum dx, sum dx*q) slabs of the BatchNorm below.  CD = 16: 256 threads, two workgroups per CU; CD = 32: 512 threads
// (the transposed weights and the weight-gradient accumulators are 72 registers each), one workgroup per CU.
#include "dm_common.h"

namespace {

constexpr int C3_HW = 16, C3_RS = 24, C3_ROWS = 18;
constexpr int C3_PS = C3_ROWS * C3_RS + 20;          // 452 == 4 (mod 32) dwords

// 16 bytes per lane from global memory straight into LDS (global_load_lds_dwordx4; LDS address = the first active lane's + 16 * lane)
__device__ __forceinline__ void c3_glds16(const float *g, float *l)
{
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g, (__attribute__((address_space(3))) void *)l, 16, 0, 0);
}
// BAND (round 5): 32-column latent grids (32 x 32: 256-pixel patches, default-width VQ_VAE_z32) as tiles of 8 rows x 32 columns --
// 256 positions like the whole 16 x 16 patch, the same 16 M tiles (two per row), the zero padding left and right still the
// image border.  Only the rows above and below are REAL halo: two full rows per plane = 16 aligned 16-byte pieces, which go
// global -> LDS directly (global_load_lds, no registers: the kernel is at its register limit) into a side buffer of raw values;
// the wave that requested them transforms them into the padded image at the commit (zeros outside the image).
// (A first form with 16 x 16 tiles fetched 36 + 32 single halo elements per plane, the columns one cache line each: slower than
// the two kernels it replaced.)
constexpr int C3B_ROWS = 8, C3B_W = 32, C3B_H = 32, C3B_RS = 40;
constexpr int C3B_PS = (C3B_ROWS + 2) * C3B_RS + 28;            // 428: the weight gradient's B reads (ci, ky, kx, kq) fall on 32 distinct banks
constexpr int C3B_HALO = 2 * C3B_W;                             // floats per plane in the side buffer

template <int CD, int NTH, bool BAND = false>
__global__ __launch_bounds__(NTH, NTH == 512 ? 1 : 2)
void conv3x3_bwd_kernel(Operand dy, const float *__restrict__ x, const float *__restrict__ xcoef, const float *__restrict__ w,
                        const float *__restrict__ resid, const float *__restrict__ q, float *__restrict__ dx,
                        double *__restrict__ stats, float *__restrict__ wslabs, int ntiles, int Harg)
{
    constexpr int CX = 16, NW = NTH / 64, MT = CD / 16, KS = 9 * CD / 4, NTT = 9;
    constexpr int ED = CD / NW, ET = CX / NW, RPW = C3_HW / NW;       // staged planes and M-tile rows per wave
    constexpr int PS = BAND ? C3B_PS : C3_PS, RS = BAND ? C3B_RS : C3_RS, W = BAND ? C3B_W : C3_HW;
    constexpr int H = BAND ? C3B_H : C3_HW, HW = H * W, bands = BAND ? H / C3B_ROWS : 1;     // tiles per patch
    (void)Harg;
    // M-tile row `y` (0..15) of the tile: image row / first column inside the tile (BAND: two M tiles per image row)
    auto ry = [](int y) { return BAND ? y >> 1 : y; };
    auto cx = [](int y) { return BAND ? 16 * (y & 1) : 0; };
    static_assert(CD % 16 == 0 && CD % NW == 0 && CX % NW == 0, "whole planes per wave");
    // WLDS (CD = 32): the transposed weights wait in LDS in operand order ([K step][lane]: one conflict-free read per step, shared
    // by the two rows in flight) -- as 72 more registers beside the 72 accumulators they spilled 47
    constexpr bool WLDS = CD > 16;
    extern __shared__ __attribute__((aligned(16))) float lds3[];
    float *sD = lds3, *sT = lds3 + CD * PS, *sW = lds3 + (CD + CX) * PS;
    // BAND: raw halo rows of the planes this wave stages: [dy planes | y planes | x planes][2 rows][32]
    float *sHalo = sW + (CD > 16 ? KS * 64 : 0);
    __shared__ double s_stat[NW][CX][2];

    const int lane = threadIdx.x & 63, m = lane & 15, kq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool two = dy.p1 != nullptr;

    // the padding never changes: zero both tile images once (the commits below write the interior only)
    for (int i = threadIdx.x; i < (CD + CX) * PS / 4; i += NTH) reinterpret_cast<f32x4 *>(lds3)[i] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // data gradient: K step s = (tap, channel group), B[k = kq][n = ci = m] = W[co = 4 cg + kq][ci][ky][kx]
    float wreg[WLDS ? 1 : KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const int tap = s / (CD / 4), cg = s - tap * (CD / 4);
        const float wv = w[((4 * cg + kq) * CX + m) * 9 + tap];
        if constexpr (WLDS) { if (wave == 0) sW[s * 64 + lane] = wv; } else wreg[s] = wv;
    }
    // BatchNorm (+ ReLU) of the layer input for the planes this wave stages, BatchNorm backward of the output gradient likewise
    float tc0[ET], tc2[ET], dc0[ED], dc1[ED], dc2[ED];
#pragma unroll
    for (int e = 0; e < ET; ++e) {
        const int c = e * NW + wave;
        tc0[e] = xcoef ? xcoef[c * 4] : 1.f;
        tc2[e] = xcoef ? xcoef[c * 4 + 2] : 0.f;
    }
#pragma unroll
    for (int e = 0; e < ED; ++e) {
        const int c = e * NW + wave;
        dc0[e] = dy.coef ? dy.coef[c * 4] : 1.f;
        dc1[e] = (dy.coef && two) ? dy.coef[c * 4 + 1] : 0.f;
        dc2[e] = dy.coef ? dy.coef[c * 4 + 2] : 0.f;
    }
    // weight gradient: B column n = 16 t + m = (ci, ky, kx): T[ci][y + ky - 1][x + kx - 1] <-> sT[ci*PS + (y + ky)*RS + x + kx + 3]
    int boff[NTT];
#pragma unroll
    for (int t = 0; t < NTT; ++t) {
        const int n = 16 * t + m, ci = n / 9, k2 = n - ci * 9, ky = k2 / 3, kx = k2 - ky * 3;
        boff[t] = ci * PS + ky * RS + kx + 3 + kq;
    }
    f32x4 wacc[MT][NTT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int t = 0; t < NTT; ++t) wacc[i][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    double s1 = 0.0, s2 = 0.0;

    // staging: float4 (lane) of plane e * NW + wave: row lane >> 2, columns 4 (lane & 3) ..
    // (BAND: row lane >> 3, columns 4 (lane & 7) .. of the 8 x 32 tile: 1 KB contiguous per plane)
    const int sq = BAND ? 4 * lane : (lane >> 2) * C3_HW + 4 * (lane & 3);     // offset inside the tile's plane
    const int sl = BAND ? ((lane >> 3) + 1) * RS + 4 + 4 * (lane & 7) : ((lane >> 2) + 1) * RS + 4 + 4 * (lane & 3);   // ... inside its padded LDS image
    f32x4 rv[ED], ru[ED], rx[ET];
    // tile k of this workgroup: BAND: the four bands of a patch one after the other in ONE workgroup (a band's halo rows are its
    // neighbours' interior rows: they are then re-read from this XCD's L2, not by another one from HBM), patches blockIdx.x,
    // blockIdx.x + gridDim.x, ...
    auto tile_origin = [&](int k, int &b, int &y0) {
        if constexpr (BAND) { b = blockIdx.x + (k / bands) * gridDim.x; y0 = (k % bands) * C3B_ROWS; }
        else { b = blockIdx.x + k * gridDim.x; y0 = 0; }
    };
    auto tile_live = [&](int k) { return (int)blockIdx.x + (k / bands) * (int)gridDim.x < ntiles; };     // ntiles: patches
    auto issue = [&](int t) {
        int b, y0;
        tile_origin(t, b, y0);
        if constexpr (BAND) {
            // (addresses as a uniform 64-bit base + a 32-bit lane offset: the 64-bit per-lane pointers of the whole-patch form
            //  would not fit beside the accumulators here -- their spills' reloads wait for vmcnt(0) while LDS-DMA is in flight
            //  and with it for the stores of dx just issued: 249 us per layer instead of 204)
            const float *__restrict__ b0 = dy.p0 + (long long)b * CD * HW, *__restrict__ b1 = dy.p1 + (long long)b * CD * HW;
            const float *__restrict__ bx = x + (long long)b * CX * HW;
            const unsigned lo = (unsigned)(y0 * W + sq);
#pragma unroll
            for (int e = 0; e < ED; ++e) {
                const unsigned off = (unsigned)((e * NW + wave) * HW) + lo;
                rv[e] = *reinterpret_cast<const f32x4 *>(b0 + off);
                if (two) ru[e] = *reinterpret_cast<const f32x4 *>(b1 + off);
            }
#pragma unroll
            for (int e = 0; e < ET; ++e) rx[e] = *reinterpret_cast<const f32x4 *>(bx + ((unsigned)((e * NW + wave) * HW) + lo));
            // lanes 0..7 the row above, 8..15 the row below (a row outside the image: the nearest one inside, replaced by 0 at the commit)
            if (lane < 16) {
                const int yy = lane < 8 ? y0 - 1 : y0 + C3B_ROWS, yc = yy < 0 ? 0 : (yy >= H ? H - 1 : yy);
                const unsigned ho = (unsigned)(yc * W + 4 * (lane & 7));
#pragma unroll
                for (int e = 0; e < ED; ++e) {
                    const unsigned pl = (unsigned)((e * NW + wave) * HW) + ho;
                    c3_glds16(b0 + pl, sHalo + (e * NW + wave) * C3B_HALO + 4 * lane);
                    if (two) c3_glds16(b1 + pl, sHalo + (CD + e * NW + wave) * C3B_HALO + 4 * lane);
                }
#pragma unroll
                for (int e = 0; e < ET; ++e)
                    c3_glds16(bx + ((unsigned)((e * NW + wave) * HW) + ho), sHalo + (2 * CD + e * NW + wave) * C3B_HALO + 4 * lane);
            }
        } else {
            const long long db = (long long)b * CD * HW, xb = (long long)b * CX * HW;
#pragma unroll
            for (int e = 0; e < ED; ++e) {
                const long long off = db + (long long)(e * NW + wave) * HW + sq;
                rv[e] = *reinterpret_cast<const f32x4 *>(dy.p0 + off);
                if (two) ru[e] = *reinterpret_cast<const f32x4 *>(dy.p1 + off);
            }
#pragma unroll
            for (int e = 0; e < ET; ++e) rx[e] = *reinterpret_cast<const f32x4 *>(x + xb + (long long)(e * NW + wave) * HW + sq);
        }
    };
    int tile = 0;
    if (tile_live(tile)) issue(tile);
    __syncthreads();                                             // the zero fill is complete

    // BAND: the two barriers of the tile loop as raw s_barrier behind a wait for the LDS counter only.  With LDS-DMA requests
    // in flight hipcc's __syncthreads() waits for vmcnt(0): every tile would drain the stores of dx it has just issued
    // (measured: 249 us per layer with __syncthreads(), 204 us without any halo work).
    auto tile_barrier = [&]() {
        if constexpr (BAND) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        } else __syncthreads();
    };
    while (tile_live(tile)) {
        if (tile != 0) tile_barrier();                           // the previous tile has been consumed
#pragma unroll
        for (int e = 0; e < ED; ++e) {
            f32x4 v = dc0[e] * rv[e] + dc2[e];                   // (the operand transform of tile.h: two fused multiply-adds)
            if (two) v += dc1[e] * ru[e];
            *reinterpret_cast<f32x4 *>(sD + (e * NW + wave) * PS + sl) = v;
        }
#pragma unroll
        for (int e = 0; e < ET; ++e) {
            f32x4 v = tc0[e] * rx[e] + tc2[e];
            if (!xcoef) v = rx[e];
            *reinterpret_cast<f32x4 *>(sT + (e * NW + wave) * PS + sl) = dm_relu4(v);
        }
        int b, ty0;
        tile_origin(tile, b, ty0);
        if constexpr (BAND) {
            // the halo rows of this wave's planes: raw values from the side buffer (its own requests),
            // the same transforms, zeros outside the image.  LDS rows 0 and 9, columns 4 + 4 (lane & 7) ..
            // (counted: the only vector-memory operations issued after the requests and possibly still pending are the previous
            //  tile's RPW stores of dx -- vmcnt retires in issue order on gfx9 -- and a wait for 0 would drain them every tile)
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(RPW) : "memory");
            if (lane < 16) {
                const int yy = lane < 8 ? ty0 - 1 : ty0 + C3B_ROWS;
                const bool in = (unsigned)yy < (unsigned)H;
                const int lo = (lane < 8 ? 0 : C3B_ROWS + 1) * RS + 4 + 4 * (lane & 7);
                const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
                // (plane by plane: reading every raw value first costs registers the kernel does not have -- 72 bytes of scratch, 309 us)
#pragma unroll
                for (int e = 0; e < ED; ++e) {
                    f32x4 v = dc0[e] * *reinterpret_cast<const f32x4 *>(sHalo + (e * NW + wave) * C3B_HALO + 4 * lane) + dc2[e];
                    if (two) v += dc1[e] * *reinterpret_cast<const f32x4 *>(sHalo + (CD + e * NW + wave) * C3B_HALO + 4 * lane);
                    *reinterpret_cast<f32x4 *>(sD + (e * NW + wave) * PS + lo) = in ? v : zero;
                }
#pragma unroll
                for (int e = 0; e < ET; ++e) {
                    const f32x4 r = *reinterpret_cast<const f32x4 *>(sHalo + (2 * CD + e * NW + wave) * C3B_HALO + 4 * lane);
                    f32x4 v = tc0[e] * r + tc2[e];
                    if (!xcoef) v = r;
                    *reinterpret_cast<f32x4 *>(sT + (e * NW + wave) * PS + lo) = in ? dm_relu4(v) : zero;
                }
            }
        }
        tile_barrier();
        ++tile;
        if (tile_live(tile)) issue(tile);                        // in flight during the products below

        // ---- weight gradient: this wave's rows, four positions per step
#pragma unroll
        for (int rr = 0; rr < RPW; ++rr) {
            const int y = wave + NW * rr;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                float a[MT];
#pragma unroll
                for (int i = 0; i < MT; ++i) a[i] = sD[(16 * i + m) * PS + (ry(y) + 1) * RS + cx(y) + 4 * s + kq + 4];
#pragma unroll
                for (int t = 0; t < NTT; ++t) {
                    const float bv = sT[boff[t] + ry(y) * RS + cx(y) + 4 * s];
#pragma unroll
                    for (int i = 0; i < MT; ++i) wacc[i][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], bv, wacc[i][t], 0, 0, 0);
                }
            }
        }
        // ---- data gradient: one M tile per row, two rows in flight
        const long long ob = (long long)b * CX * HW + ty0 * W;
#pragma unroll
        for (int rr = 0; rr < RPW; rr += 2) {
            f32x4 acc[2], rres[2], rq[2];
            const float *pa[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
                const int yj = wave + NW * (rr + j);
                pa[j] = sD + kq * PS + (ry(yj) + 2) * RS + cx(yj) + m + 5;
                // the epilogue's side inputs are requested now: their latency passes under the products
                const long long o = ob + (long long)m * HW + ry(yj) * W + cx(yj) + 4 * kq;
                if (resid) rres[j] = *reinterpret_cast<const f32x4 *>(resid + o);
                if (stats && q) rq[j] = *reinterpret_cast<const f32x4 *>(q + o);
            }
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const int tap = s / (CD / 4), cg = s - tap * (CD / 4), ky = tap / 3, kx = tap - 3 * ky;
                const int off = 4 * cg * PS - ky * RS - kx;
                const float wv = WLDS ? sW[s * 64 + lane] : wreg[WLDS ? 0 : s];
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[j][off], wv, acc[j], 0, 0, 0);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int y = wave + NW * (rr + j);
                // lane (m, kq): positions x = 4 kq .. 4 kq + 3 of row y, channel ci = m
                const f32x4 tv = *reinterpret_cast<const f32x4 *>(sT + m * PS + (ry(y) + 1) * RS + cx(y) + 4 + 4 * kq);
                f32x4 v = acc[j];
                v.x = tv.x > 0.f ? v.x : 0.f; v.y = tv.y > 0.f ? v.y : 0.f;
                v.z = tv.z > 0.f ? v.z : 0.f; v.w = tv.w > 0.f ? v.w : 0.f;
                const long long o = ob + (long long)m * HW + ry(y) * W + cx(y) + 4 * kq;
                if (resid) v += rres[j];
                *reinterpret_cast<f32x4 *>(dx + o) = v;
                if (stats) {
                    const f32x4 qv = q ? rq[j] : v;
                    s1 += (double)((v.x + v.y) + (v.z + v.w));
                    s2 += (double)((v.x * qv.x + v.y * qv.y) + (v.z * qv.z + v.w * qv.w));
                }
            }
        }
    }

    // ---- statistics slab: the four kq groups of a channel, then the waves in wave order
    __syncthreads();
    if (stats) {
        double a = s1, c = s2;
        a += __shfl_xor(a, 16, 64); c += __shfl_xor(c, 16, 64);
        a += __shfl_xor(a, 32, 64); c += __shfl_xor(c, 32, 64);
        if (lane < 16) { s_stat[wave][lane][0] = a; s_stat[wave][lane][1] = c; }
    }
    __syncthreads();
    // ---- weight-gradient slab: every wave's accumulators through LDS, summed in wave order, one M tile of 16 output-gradient
    //      channels at a time (the tile images are free now; all of CD = 32 at once would not fit them)
    float *red = lds3;                                           // [wave][NTT][64 lanes][4]
    static_assert(NW * NTT * 256 <= (CD + CX) * PS, "the slab combine reuses the tile images");
    if (stats && threadIdx.x < CX) {
        double ta = 0.0, tc = 0.0;
#pragma unroll
        for (int wv = 0; wv < NW; ++wv) { ta += s_stat[wv][threadIdx.x][0]; tc += s_stat[wv][threadIdx.x][1]; }
        stats[((long long)blockIdx.x * CX + threadIdx.x) * 2 + 0] = ta;
        stats[((long long)blockIdx.x * CX + threadIdx.x) * 2 + 1] = tc;
    }
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        if (i) __syncthreads();
#pragma unroll
        for (int t = 0; t < NTT; ++t) *reinterpret_cast<f32x4 *>(red + ((wave * NTT + t) * 64 + lane) * 4) = wacc[i][t];
        __syncthreads();
        // element e = dW[co = 16 i + c][n], n = (ci, ky, kx) = ci*9 + k2: accumulator row c = 4 kq + r, column n & 15 of N tile n >> 4
        for (int e = threadIdx.x; e < 16 * 144; e += NTH) {
            const int c = e / 144, n = e - c * 144;
            const int t = n >> 4, ln = (c >> 2) * 16 + (n & 15), r = c & 3;
            float sum = 0.f;
#pragma unroll
            for (int wv = 0; wv < NW; ++wv) sum += red[((wv * NTT + t) * 64 + ln) * 4 + r];
            wslabs[(long long)blockIdx.x * CD * 144 + (16 * i + c) * 144 + n] = sum;
        }
    }
}

bool conv3x3_bwd_band_on()
{
    static const bool off = [] { const char *e = getenv("DM_CONV3X3_BWD_BAND"); return e && e[0] == '0'; }();
    return !off;
}
bool conv3x3_bwd_shape(int CD, int CX, int H, int W)
{
    if (!((CD == 16 || CD == 32) && CX == 16)) return false;
    if (H == C3_HW && W == C3_HW) return true;
    // 32 x 32: bands of 8 rows x 32 columns -- for the residual layers' 32 output-gradient channels (238 us per layer against the
    // two kernels' 267 at C5's shape); with 16 (enc.10) the two kernels are as fast (120 against 125 us) and stay
    return conv3x3_bwd_band_on() && CD == 32 && W == C3B_W && H == C3B_H;
}
constexpr size_t conv3x3_bwd_lds(int CD, bool band = false)
{
    return ((size_t)(CD + 16) * (band ? C3B_PS : C3_PS) + (CD > 16 ? 9 * CD / 4 * 64 : 0) + (band ? (2 * CD + 16) * C3B_HALO : 0)) * sizeof(float);
}

}  // namespace

extern "C" int dm_conv3x3_bwd_fused_supported(int CD, int CX, int H, int W) { return conv3x3_bwd_shape(CD, CX, H, W) ? 1 : 0; }

extern "C" int dm_conv3x3_bwd_fused_num_blocks(int B, int CD, int CX, int H, int W)
{
    if (B <= 0 || !conv3x3_bwd_shape(CD, CX, H, W)) return -1;
    const int cap = CD == 32 ? 256 : 512;                        // resident workgroups: one / two per CU
    return B < cap ? B : cap;
}

extern "C" int dm_conv3x3_bwd_fused(const dm_operand *dy, const float *x, const float *xcoef, const float *w, const float *resid,
                                    const float *q, float *dx, double *stats, float *wslabs, int B, int CD, int CX, int H, int W,
                                    void *stream)
{
    DM_REQUIRE(dy && dy->p0 && x && w && dx && wslabs, "dm_conv3x3_bwd_fused: NULL pointer");
    DM_REQUIRE(B > 0 && conv3x3_bwd_shape(CD, CX, H, W), "dm_conv3x3_bwd_fused: shape %d -> %d channels on %dx%d not built", CX, CD, H, W);
    DM_REQUIRE(dy->mode == DM_LOAD_IDENT || dy->mode == DM_LOAD_AFFINE2, "dm_conv3x3_bwd_fused: dy operand must be IDENT or AFFINE2");
    DM_REQUIRE(dy->mode == DM_LOAD_IDENT || dy->coef, "dm_conv3x3_bwd_fused: AFFINE2 needs coefficients");
    DM_REQUIRE(dy->coef_bstride == 0 && !dy->ones_channel, "dm_conv3x3_bwd_fused: shared coefficients only");
    DM_REQUIRE(!q || stats, "dm_conv3x3_bwd_fused: q without a statistics destination");
    // (the band form reads the halo rows of a neighbouring band: dx written over an input would corrupt them)
    DM_REQUIRE(dx != dy->p0 && dx != dy->p1 && dx != x, "dm_conv3x3_bwd_fused: dx must not alias dy or x");
    Operand d = to_dev(dy);
    if (d.mode == DM_LOAD_IDENT) { d.coef = nullptr; d.p1 = nullptr; }
    const int grid = dm_conv3x3_bwd_fused_num_blocks(B, CD, CX, H, W);
    hipStream_t st = (hipStream_t)stream;
    static DmPerDeviceOnce attr_done;
    if (attr_done.need()) {
        hipError_t e = hipFuncSetAttribute((const void *)conv3x3_bwd_kernel<32, 512>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)conv3x3_bwd_lds(32));
        if (e == hipSuccess)
            e = hipFuncSetAttribute((const void *)conv3x3_bwd_kernel<16, 256>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)conv3x3_bwd_lds(16));
        if (e == hipSuccess)
            e = hipFuncSetAttribute((const void *)conv3x3_bwd_kernel<32, 512, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)conv3x3_bwd_lds(32, true));

        if (e != hipSuccess) {
            dm_set_error("dm_conv3x3_bwd_fused: hipFuncSetAttribute: %s", hipGetErrorString(e));
            return (int)e;
        }
        attr_done.mark();
    }
    const bool band = W != C3_HW;
    const int ntiles = B;                                        // (patches; BAND: four tiles each, taken by one workgroup)
    DM_REQUIRE((long long)B * (CD > CX ? CD : CX) * H * W < (1LL << 31), "dm_conv3x3_bwd_fused: tensor too large for 32-bit offsets");
    if (CD == 32 && band)
        hipLaunchKernelGGL((conv3x3_bwd_kernel<32, 512, true>), dim3(grid), dim3(512), conv3x3_bwd_lds(32, true), st, d, x, xcoef, w, resid,
                           q, dx, stats, wslabs, ntiles, H);
    else if (CD == 32)
        hipLaunchKernelGGL((conv3x3_bwd_kernel<32, 512>), dim3(grid), dim3(512), conv3x3_bwd_lds(32), st, d, x, xcoef, w, resid, q, dx,
                           stats, wslabs, ntiles, H);
    else
        hipLaunchKernelGGL((conv3x3_bwd_kernel<16, 256>), dim3(grid), dim3(256), conv3x3_bwd_lds(16), st, d, x, xcoef, w, resid, q, dx,
                           stats, wslabs, ntiles, H);
    return dm_launch_status("dm_conv3x3_bwd_fused");
}

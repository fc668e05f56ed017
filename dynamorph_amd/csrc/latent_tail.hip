// latent_tail.hip -- the 16 x 16 part of the encoder for the INFERENCE path (per-sample BatchNorm statistics), one patch
// per workgroup pass, activations never leaving the CU.
//
// Reference: HiddenStateExtractor/vq_vae.py:287-289 (enc.10 Conv2d(nh, nh, 3, padding=1), enc.11 BatchNorm2d, enc.12
// ResidualBlock) with :203-224 (ResidualBlock: x + [ReLU, Conv3x3(nh -> nrh), BN, ReLU, Conv1x1(nrh -> nh), BN](x), twice),
// called by pipeline/patch_VAE.py:445-452 one patch at a time in train mode: every BatchNorm normalises with THAT
// patch's statistics, so a patch's whole 16 x 16 stack depends on nothing but its own a3 = enc.7's output.
//
// As separate launches (rounds 1-2) these five convolutions, five per-sample BatchNorm finalisers and three apply / join
// kernels are 13 launches and 185 us of the 420 us a batch of 1024 patches takes, each round-tripping a 16-32 KB per-patch
// tensor through HBM and each limited by launch tails (1024 one-tile workgroups on 768 slots).  Here a workgroup keeps the
// patch in LDS / registers from a3 to z:
//   T  (16 ch, zero-padded 18 x 24 planes)  the 3x3 convolutions' input, ReLU already applied
//   U  (32 ch, 256-float planes)            the 1x1 convolution's input, BatchNorm + ReLU already applied
//   h  registers, in the accumulator layout of a 16-output-channel product: lane (n = lane & 15, q = lane >> 4) holds
//      h[n][4 wave + mt][4 q .. 4 q + 3] for mt = 0..3 -- the layout the 1x1 convolution's result arrives in, so the
//      residual join is 16 fused multiply-adds per lane and h is stored exactly once, as z.
// Convolutions are v_mfma_f32_16x16x4_f32 products (exact fp32 chains): M = 16 positions of one row, N = 16 output
// channels, K = 4 input channels per step; a wave owns four rows.  Per-sample statistics: every lane sums its 16 values per
// channel in double, two lane-pair steps and one LDS exchange combine the 256 positions; every thread then derives its own
// channel's coefficients (the arithmetic of bn_finalize_per_sample_kernel, bn.hip).  The sums are also written out, one
// slab per patch and layer, for dm_bn_running_replay (the running statistics are a side effect no kernel here reads).
// The next layer's weights are requested as soon as the current product is done and land under the statistics exchange.
#include "dm_common.h"
#include "mfma_util.h"

namespace {

constexpr int LT_C = 16, LT_CR = 32, LT_HW = 16;
constexpr int LT_RS = 24;                 // padded row of T: column x at j = x + 4 (16-byte aligned interior), j = 3 / 20 zero
constexpr int LT_PS = 18 * LT_RS;         // 432 floats per channel plane (rows y = -1 .. 16); 432 % 32 == 16: the four channel
                                          // groups of an A-operand read fall on the two halves of the banks
constexpr int LT_UPS = 256 + 16;          // plane of U, same bank argument
constexpr int LT_BLOCK = 256;
constexpr int LT_MAX_RES = 4;

struct LtParams {
    const float *a3;          // (B,16,16,16) raw output of enc.7
    const float *coef3;       // (B,16,4) per-sample BatchNorm coefficients of enc.8 (dm_bn_finalize, per-sample form)
    const float *w10, *b10, *g4, *be4;
    double *st4;              // (B,16,2) sums of enc.10's output (for the running statistics)
    // E7 form: the patch enters as a2 (raw output of enc.4, 16 x 32 x 32) and enc.7 / enc.8 run here too
    const float *a2, *coef2;  // (B,16,32,32); (B,16,4) per-sample coefficients of enc.5
    const float *w7, *b7, *g3, *be3;
    double *st3;              // (B,16,2) sums of enc.7's output
    float eps3;
    const float *wa[LT_MAX_RES], *ba[LT_MAX_RES], *ga[LT_MAX_RES], *bea[LT_MAX_RES];
    const float *wb[LT_MAX_RES], *bb[LT_MAX_RES], *gb[LT_MAX_RES], *beb[LT_MAX_RES];
    double *sta[LT_MAX_RES], *stb[LT_MAX_RES];     // (B,32,2), (B,16,2)
    float *z;                 // (B,16,16,16)
    float eps4, epsa[LT_MAX_RES], epsb[LT_MAX_RES];
    int B, nres, dbg;          // dbg: measurement only (DM_LT_DBG, tools/exp/lt_bench.py): 1 no products, 2 no statistics arithmetic, 4 no weight loads
};

// Per-sample statistics of a layer's raw output v (+ bias already added) and the BatchNorm coefficients of this lane's
// channels: scale[nt], shift[nt] for channel 16 nt + n.  buf: [4 waves][NT][16][2] doubles of LDS, a different one from
// the previous layer's.  One barrier.  Vector instructions here do not hide under the other workgroup's products (the
// f32-input matrix instruction and the vector unit exclude each other), so the arithmetic is kept short: sums in double
// (full rate), mean = sum / 256 as a multiplication, 1 / sqrt(var + eps) from v_rsq_f32 with two Newton steps in double
// (agrees with (float)(1.0 / sqrt(x)) to the last bit but for rounding ties) instead of a double division and square root.
template <int NT>
__device__ __forceinline__ void lt_batchnorm(const f32x4 (&v)[4][NT], double *__restrict__ buf, const float *__restrict__ gamma,
                                             const float *__restrict__ beta, float eps, double *__restrict__ slab, int wave,
                                             int lane, float (&scale)[NT], float (&shift)[NT], int dbg)
{
    const int n = lane & 15, q = lane >> 4;
    if (dbg & 2) {                                             // (measurement: barrier only)
        __syncthreads();
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) { scale[nt] = 1.f; shift[nt] = 0.f; }
        return;
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        double s1 = 0.0, s2 = 0.0;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double d = (double)v[mt][nt][r];
                s1 += d; s2 += d * d;
            }
        s1 = dm_row_sum_f64(s1);
        s2 = dm_row_sum_f64(s2);
        if (q == 0) {
            buf[((wave * NT + nt) * 16 + n) * 2 + 0] = s1;
            buf[((wave * NT + nt) * 16 + n) * 2 + 1] = s2;
        }
    }
    __syncthreads();
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        double t1 = 0.0, t2 = 0.0;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            t1 += buf[((w * NT + nt) * 16 + n) * 2 + 0];
            t2 += buf[((w * NT + nt) * 16 + n) * 2 + 1];
        }
        constexpr double inv_cnt = 1.0 / (double)(LT_HW * LT_HW);      // (a power of two: the product is the quotient)
        const double mean = t1 * inv_cnt;
        double var = t2 * inv_cnt - mean * mean;
        if (var < 0.0) var = 0.0;
        const float mean_f = (float)mean;
        const double xv = var + (double)eps;
        double y = (double)__builtin_amdgcn_rsqf((float)xv);           // ~1 ulp of float; two Newton steps: ~1e-15
        y = y * (1.5 - 0.5 * xv * y * y);
        y = y * (1.5 - 0.5 * xv * y * y);
        const float invstd = (float)y;
        const float g = gamma ? gamma[16 * nt + n] : 1.f, bt = beta ? beta[16 * nt + n] : 0.f;
        scale[nt] = g * invstd;
        shift[nt] = bt - mean_f * scale[nt];
        if (wave == 0 && q == 0) {
            slab[(16 * nt + n) * 2 + 0] = t1;
            slab[(16 * nt + n) * 2 + 1] = t2;
        }
    }
}

// weights of a 3x3 convolution with 16 input channels, as B operands: step s = 4 tap + cg <-> input channel 4 cg + q
template <int NT>
__device__ __forceinline__ void lt_load_w3(const float *__restrict__ w, int lane, float (&wreg)[NT][36])
{
    const int n = lane & 15, q = lane >> 4;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int s = 0; s < 36; ++s) wreg[nt][s] = w[((16 * nt + n) * LT_C + 4 * (s & 3) + q) * 9 + (s >> 2)];

}

// E7: the kernel starts one layer earlier, at a2 (enc.4's raw output): enc.7's 4x4 / stride-2 product runs on the two
// halves of the patch (8 output rows each: a 16 x 18 x 40 input tile, BatchNorm + ReLU of enc.5 / enc.6 applied on the way in,
// laid over T and U, which are dead then), its statistics and enc.8's coefficients are taken in registers like the others.
constexpr int LT_E_RS = 40, LT_E_PS = 18 * LT_E_RS;          // the 4x4 / stride-2 input tile: column x at j = x + 4
template <bool E7>
__global__ __launch_bounds__(LT_BLOCK, 2) void latent_tail_kernel(LtParams P)
{
    __shared__ __attribute__((aligned(16))) float s_all[LT_C * LT_PS + LT_CR * LT_UPS];
    float *const sT = s_all, *const sU = s_all + LT_C * LT_PS;
    static_assert(LT_C * LT_E_PS <= LT_C * LT_PS + LT_CR * LT_UPS, "the enc.7 input tile fits over T and U");
    __shared__ double s_stat[2][4 * 2 * 16 * 2];

    const int lane = threadIdx.x & 63, n = lane & 15, q = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);

    // the padding of T is zero for every patch; the interior is rewritten before every 3x3 product
    for (int i = threadIdx.x; i < LT_C * LT_PS; i += LT_BLOCK) sT[i] = 0.f;

    // A-operand bases: lane (m = n, q) reads channel 4 cg + q at column m; the wave's rows are 4 wave .. 4 wave + 3
    const float *apT[4], *apU[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        apT[mt] = sT + q * LT_PS + (4 * wave + mt) * LT_RS + n + 3;        // (+ dy rows, + dx columns per tap)
        apU[mt] = sU + q * LT_UPS + (4 * wave + mt) * LT_HW + n;
    }
    auto off3 = [](int s) { return (s >> 2) / 3 * LT_RS + (s >> 2) % 3 + 4 * (s & 3) * LT_PS; };
    auto off1 = [](int s) { return 4 * s * LT_UPS; };

    // this thread's part of a3: four 16-byte pieces (channel i >> 6, row (i & 63) >> 2, columns 4 (i & 3) ..)
    f32x4 pre[E7 ? 1 : 4];
    auto fetch = [&](int b) {
        if constexpr (!E7) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                pre[j] = *reinterpret_cast<const f32x4 *>(P.a3 + (long long)b * (LT_C * 256) + 4 * (threadIdx.x + LT_BLOCK * j));
        }
    };
    int b = blockIdx.x;
    if (b < P.B) fetch(b);
    float w3[2][36];
    // enc.7's weights as B operands: step s = 4 c + ky, k lane = kx (the operand layout of conv4x4s2_kernel, conv_mfma.hip)
    auto load_w7 = [&]() {
        float(&w7)[72] = reinterpret_cast<float(&)[72]>(w3);
#pragma unroll
        for (int s = 0; s < 64; ++s) w7[s] = P.w7[((n * LT_C + (s >> 2)) * 4 + (s & 3)) * 4 + q];
    };
    if constexpr (E7) load_w7();
    else lt_load_w3<1>(P.w10, lane, reinterpret_cast<float(&)[1][36]>(w3));
    __syncthreads();

    for (; b < P.B; b += gridDim.x) {
        const int nb = b + gridDim.x;
        if constexpr (E7) {
            // ---- enc.7 + enc.8 + enc.9: T <- relu(BatchNorm(conv4x4s2(relu(BatchNorm(a2))))) ------------------------------
            f32x4 a3r[4][1];
            float *const tile = s_all;
            const float *ap7[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) ap7[i] = tile + 2 * (2 * wave + i) * LT_E_RS + 2 * n + q + 3;
            auto off7 = [](int s) { return (s >> 2) * LT_E_PS + (s & 3) * LT_E_RS; };
#pragma unroll 1
            for (int half = 0; half < 2; ++half) {
                if (half) __syncthreads();                     // the first half's readers are done with the tile
                // (half 0: the previous patch's last readers of T / U are behind its last statistics barrier)
#pragma unroll
                for (int j = 0; j < 12; ++j) {
                    const int e = threadIdx.x + LT_BLOCK * j;  // (channel, tile row, 16-byte column): 16 x 18 x 10
                    const int c = e / 180, rem = e - c * 180, row = rem / 10, qq = rem - row * 10;
                    const int gy = 16 * half - 1 + row;
                    const bool in = e < 2880 && qq >= 1 && qq <= 8 && (unsigned)gy < 32u;
                    f32x4 v = {0.f, 0.f, 0.f, 0.f};
                    if (in) {
                        const f32x4 cf = *reinterpret_cast<const f32x4 *>(P.coef2 + ((long long)b * LT_C + c) * 4);
                        v = *reinterpret_cast<const f32x4 *>(P.a2 + (((long long)b * LT_C + c) * 32 + gy) * 32 + 4 * (qq - 1));
                        v = dm_relu4(cf.x * v + cf.z);
                    }
                    if (e < 2880) *reinterpret_cast<f32x4 *>(tile + c * LT_E_PS + row * LT_E_RS + 4 * qq) = v;
                }
                __syncthreads();
                f32x4 acc[2][1];
                acc[0][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc[1][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (!(P.dbg & 1)) mfma_tiles<2, 1, 64, 4>(ap7, reinterpret_cast<const float(&)[1][64]>(w3), acc, off7);
                a3r[2 * half][0] = acc[0][0]; a3r[2 * half + 1][0] = acc[1][0];
            }
            lt_load_w3<1>(P.w10, lane, reinterpret_cast<float(&)[1][36]>(w3));      // enc.10's weights: under the statistics
            const float bias7 = P.b7 ? P.b7[n] : 0.f;
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) a3r[mt][0] = a3r[mt][0] + bias7;
            float sc[1], sh[1];
            lt_batchnorm<1>(a3r, s_stat[1], P.g3, P.be3, P.eps3, P.st3 + (long long)b * LT_C * 2, wave, lane, sc, sh, P.dbg);
            // (the tile is dead: every wave is past its products)  T, with its zero padding (the tile lay over it)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const int y = 8 * (mt >> 1) + 2 * wave + (mt & 1);
                float *rowp = sT + n * LT_PS + (y + 1) * LT_RS;
                *reinterpret_cast<f32x4 *>(rowp + 4 + 4 * q) = dm_relu4(sc[0] * a3r[mt][0] + sh[0]);
                if (q == 0) rowp[3] = 0.f;
                if (q == 3) rowp[20] = 0.f;
            }
            if (wave == 0 || wave == 3)
                for (int i = lane; i < LT_C * LT_RS; i += 64) sT[(i / LT_RS) * LT_PS + (wave ? 17 : 0) * LT_RS + i % LT_RS] = 0.f;
            __syncthreads();
        } else {
        // ---- T <- relu(BatchNorm(a3)) (per-sample coefficients of enc.8) -------------------------------------------
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = threadIdx.x + LT_BLOCK * j, c = i >> 6, y = (i & 63) >> 2, x4 = i & 3;
            const f32x4 cf = *reinterpret_cast<const f32x4 *>(P.coef3 + ((long long)b * LT_C + c) * 4);
            *reinterpret_cast<f32x4 *>(sT + c * LT_PS + (y + 1) * LT_RS + 4 + 4 * x4) = dm_relu4(cf.x * pre[j] + cf.z);
        }
        __syncthreads();
        }

        // ---- enc.10 + enc.11: h = BatchNorm(conv3x3(T)) ------------------------------------------------------------
        f32x4 h[4];
        {
            f32x4 acc[4][1];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) acc[mt][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (!(P.dbg & 1)) mfma_tiles<4, 1, 36, 4>(apT, reinterpret_cast<const float(&)[1][36]>(w3), acc, off3);
            if (P.nres > 0 && !(P.dbg & 4)) lt_load_w3<2>(P.wa[0], lane, w3);
            const float bias = P.b10 ? P.b10[n] : 0.f;
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) acc[mt][0] = acc[mt][0] + bias;
            float sc[1], sh[1];
            lt_batchnorm<1>(acc, s_stat[0], P.g4, P.be4, P.eps4, P.st4 + (long long)b * LT_C * 2, wave, lane, sc, sh, P.dbg);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) h[mt] = sc[0] * acc[mt][0] + sh[0];
        }

        for (int l = 0; l < P.nres; ++l) {
            // (every wave is past its reads of T: the statistics barrier of the previous product)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
                *reinterpret_cast<f32x4 *>(sT + n * LT_PS + (4 * wave + mt + 1) * LT_RS + 4 + 4 * q) = dm_relu4(h[mt]);
            __syncthreads();
            // ---- 3x3, 16 -> 32 channels, BatchNorm, ReLU -> U -------------------------------------------------------
            float w1[1][8];
            {
                f32x4 acc[4][2];
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) { acc[mt][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc[mt][1] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
                if (!(P.dbg & 1)) mfma_tiles<4, 2, 36, 2>(apT, w3, acc, off3);
#pragma unroll
                for (int s = 0; s < 8; ++s) w1[0][s] = P.wb[l][n * LT_CR + 4 * s + q];
                const float b0 = P.ba[l] ? P.ba[l][n] : 0.f, b1 = P.ba[l] ? P.ba[l][16 + n] : 0.f;
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) { acc[mt][0] = acc[mt][0] + b0; acc[mt][1] = acc[mt][1] + b1; }
                float sc[2], sh[2];
                lt_batchnorm<2>(acc, s_stat[1], P.ga[l], P.bea[l], P.epsa[l], P.sta[l] + (long long)b * LT_CR * 2, wave, lane, sc, sh, P.dbg);
                // (U's previous readers are behind the barrier inside lt_batchnorm)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt)
                        *reinterpret_cast<f32x4 *>(sU + (16 * nt + n) * LT_UPS + (4 * wave + mt) * LT_HW + 4 * q) =
                            dm_relu4(sc[nt] * acc[mt][nt] + sh[nt]);
            }
            __syncthreads();
            // ---- 1x1, 32 -> 16 channels, BatchNorm, residual join ----------------------------------------------------
            {
                f32x4 acc[4][1];
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) acc[mt][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (!(P.dbg & 1)) mfma_tiles<4, 1, 8, 4>(apU, w1, acc, off1);
                if (l + 1 < P.nres && !(P.dbg & 4)) lt_load_w3<2>(P.wa[l + 1], lane, w3);
                const float bias = P.bb[l] ? P.bb[l][n] : 0.f;
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) acc[mt][0] = acc[mt][0] + bias;
                float sc[1], sh[1];
                lt_batchnorm<1>(acc, s_stat[0], P.gb[l], P.beb[l], P.epsb[l], P.stb[l] + (long long)b * LT_C * 2, wave, lane, sc, sh, P.dbg);
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) h[mt] = h[mt] + (sc[0] * acc[mt][0] + sh[0]);
            }
        }
        // ---- z ------------------------------------------------------------------------------------------------------
        if (nb < P.B) fetch(nb);                               // the next patch's a3 (requested here: no registers held across the products)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
            *reinterpret_cast<f32x4 *>(P.z + (long long)b * (LT_C * 256) + n * 256 + (4 * wave + mt) * LT_HW + 4 * q) = h[mt];
        if (nb < P.B && !(P.dbg & 4)) {
            if constexpr (E7) load_w7();
            else lt_load_w3<1>(P.w10, lane, reinterpret_cast<float(&)[1][36]>(w3));
        }
        // (T is rewritten at the top of the loop: its last readers are behind at least one barrier)
    }
}

}  // namespace

extern "C" int dm_latent_tail_supported(int C, int CR, int H, int W, int nres)
{
    return C == LT_C && CR == LT_CR && H == LT_HW && W == LT_HW && nres >= 0 && nres <= LT_MAX_RES;
}

extern "C" int dm_latent_tail_forward(const dm_latent_tail_args *a, void *stream)
{
    DM_REQUIRE(a, "dm_latent_tail_forward: NULL arguments");
    DM_REQUIRE(dm_latent_tail_supported(a->C, a->CR, a->H, a->W, a->nres), "dm_latent_tail_forward: built for 16 channels, 32 residual "
               "channels on a 16 x 16 latent grid, at most %d residual layers (got %d / %d on %d x %d, %d layers)", LT_MAX_RES,
               a->C, a->CR, a->H, a->W, a->nres);
    const bool e7 = a->a2 != nullptr;
    DM_REQUIRE(a->B > 0 && a->w10 && a->stats4 && a->z, "dm_latent_tail_forward: NULL pointer");
    DM_REQUIRE(e7 ? (a->coef2 && a->w7 && a->stats3) : (a->a3 && a->coef3),
               "dm_latent_tail_forward: give (a3, coef3) or (a2, coef2, w7, stats3)");
    {
        uintptr_t al = (uintptr_t)a->z | (uintptr_t)a->a3 | (uintptr_t)a->coef3 | (uintptr_t)a->a2 | (uintptr_t)a->coef2;
        DM_REQUIRE((al & 15) == 0, "dm_latent_tail_forward: a2 / a3 / coef / z must be 16-byte aligned");
    }
    LtParams P;
    P.a3 = a->a3; P.coef3 = a->coef3; P.w10 = a->w10; P.b10 = a->b10; P.g4 = a->gamma4; P.be4 = a->beta4; P.st4 = a->stats4;
    P.eps4 = a->eps4; P.z = a->z; P.B = a->B; P.nres = a->nres;
    P.a2 = a->a2; P.coef2 = a->coef2; P.w7 = a->w7; P.b7 = a->b7; P.g3 = a->gamma3; P.be3 = a->beta3; P.st3 = a->stats3; P.eps3 = a->eps3;
#ifdef DM_MEASURE      // ablation switches exist only in a measurement build (make measure): they make results wrong
    static const int dbg = [] { const char *e = getenv("DM_LT_DBG"); return e ? atoi(e) : 0; }();
#else
    static const int dbg = 0;
#endif
    P.dbg = dbg;
    for (int l = 0; l < LT_MAX_RES; ++l) {
        const bool on = l < a->nres;
        if (on)
            DM_REQUIRE(a->res[l].wa && a->res[l].wb && a->res[l].stats_a && a->res[l].stats_b, "dm_latent_tail_forward: residual layer %d: NULL pointer", l);
        P.wa[l] = on ? a->res[l].wa : nullptr; P.ba[l] = on ? a->res[l].ba : nullptr;
        P.ga[l] = on ? a->res[l].gamma_a : nullptr; P.bea[l] = on ? a->res[l].beta_a : nullptr;
        P.wb[l] = on ? a->res[l].wb : nullptr; P.bb[l] = on ? a->res[l].bb : nullptr;
        P.gb[l] = on ? a->res[l].gamma_b : nullptr; P.beb[l] = on ? a->res[l].beta_b : nullptr;
        P.sta[l] = on ? a->res[l].stats_a : nullptr; P.stb[l] = on ? a->res[l].stats_b : nullptr;
        P.epsa[l] = on ? a->res[l].eps_a : 0.f; P.epsb[l] = on ? a->res[l].eps_b : 0.f;
    }
    const int grid = a->B < 512 ? a->B : 512;                 // two workgroups per CU
    if (e7) hipLaunchKernelGGL(latent_tail_kernel<true>, dim3(grid), dim3(LT_BLOCK), 0, (hipStream_t)stream, P);
    else hipLaunchKernelGGL(latent_tail_kernel<false>, dim3(grid), dim3(LT_BLOCK), 0, (hipStream_t)stream, P);
    return dm_launch_status("dm_latent_tail_forward");
}

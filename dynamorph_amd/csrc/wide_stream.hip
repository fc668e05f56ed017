// wide_stream.hip -- the wide family's memory-bound layers as streaming kernels: operands straight from global memory
// into the matrix instruction's registers, no LDS tile, no workgroup barrier.
//
// conv_wide.hip's implicit GEMM stages every operand through LDS in chunks of 8 channels between two barriers.  That is the
// right shape for the 3x3 / 4x4 layers at 64 channels (MFMA bound), and the wrong one where a layer moves more bytes than
// it multiplies: the 1x1 convolutions of the residual blocks (64 -> 64: 6.4 GFLOP against 0.4-0.8 GB at B = 768) ran at
// 1.3-2 TB/s because each 16 KB chunk waited out its own global round trip.  For a 1x1 layer the matrix instruction's
// operand layout IS a memory layout: lane (i = l % 16, k = l / 16) of v_mfma_f32_16x16x4_f32 holds element (row i, k) of A
// and (k, column i) of B.  With pixels as the K dimension a lane loads FOUR consecutive pixels of ITS channel (one 16-byte
// load; the four lanes k = 0..3 of a channel cover 64 contiguous bytes) and the four values feed four K steps -- K step j
// pairs lane k with pixel 4 k + j on both operands, which is all a sum over pixels needs.  Every wave runs its own stream of
// 32-pixel groups (whole 128-byte lines per channel row), D groups in flight in registers, and nothing but the final slab
// write is shared.  The image-side 4x4 / stride 2 layers (1-4 channels) follow the same idea with the taps as a matrix
// dimension; the last transposed convolution (eight outputs per input pixel) is vector-unit work and sits here too.
#include "dm_common.h"
#include <stdlib.h>
#include <type_traits>

namespace {

__device__ __forceinline__ f32x4 sx_max(f32x4 v, float lo)
{
    return (f32x4){__builtin_elementwise_maximum(v.x, lo), __builtin_elementwise_maximum(v.y, lo),
                   __builtin_elementwise_maximum(v.z, lo), __builtin_elementwise_maximum(v.w, lo)};
}

// coefficient triple of an operand's channel c as the stream kernels apply it: y = max(c0 * x + (c1 * u + c2), lo) -- the
// identity for the modes without coefficients (1 * x + 0: a -0 becomes +0, which no sum can tell apart), lo = 0 for the
// ReLU modes and -inf otherwise (maximum(x, -inf) = x, a NaN stays a NaN: dm_relu's instruction either way)
struct StreamCoef {
    float c0, c1, c2;
};
__device__ __forceinline__ StreamCoef stream_coef(const Operand &op, int c)
{
    StreamCoef r{1.f, 0.f, 0.f};
    if (op.mode >= DM_LOAD_AFFINE) {
        const float *cf = op.coef + c * 4;
        r.c0 = cf[0];
        r.c1 = op.mode == DM_LOAD_AFFINE2 ? cf[1] : 0.f;
        r.c2 = cf[2];
    }
    return r;
}
__device__ __forceinline__ float stream_floor(const Operand &op)
{
    return (op.mode == DM_LOAD_RELU || op.mode == DM_LOAD_AFFINE_RELU) ? 0.f : -__builtin_inff();
}

// ------------------------------------------------------------------------------------ weight gradient of a 1x1 layer
// dW[cs][ct] = sum over samples and pixels of S'[cs][px] * T'[ct][px].  M = S channels (MT tiles of 16), N = T channels
// (NT tiles), K = pixels.  Wave w of a workgroup owns M tile w % MT and NT * MT / 4 of the N tiles; its lanes load the S
// rows of their own channel once per group and the T rows of NTW channels.  A workgroup takes a contiguous range of the
// B * HW / 32 pixel groups (coefficients are shared by all samples: batch-statistics BatchNorm, the host checks), so the
// stream runs across sample boundaries; its partial result is slab blockIdx.x.
// S2: the S operand is AFFINE2 (two tensors).  D: groups of 32 pixels in flight per wave.
template <int MT, int NT, bool S2, int D>
__global__ __launch_bounds__(256, D <= 1 ? 4 : 3) void wgrad1x1_stream_kernel(Operand S, Operand T, float *__restrict__ slabs,
                                                                            int B, int CS, int CT, int HW, int nslabs)
{
    constexpr int NG = 4 / MT, NTW = NT / NG;
    static_assert(MT * NG == 4 && NTW * NG == NT, "four waves share MT x NT tiles");
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, p = lane & 15, kq = lane >> 4;
    const int mt = wave % MT, ng = wave / MT;
    const int gpp = HW >> 5;                                     // groups of 32 pixels per plane
    const long long total = (long long)B * gpp;
    const int q0 = (int)(total * blockIdx.x / gridDim.x), q1 = (int)(total * (blockIdx.x + 1) / gridDim.x);
    const int n = q1 - q0;

    const int cs = mt * 16 + p, ct0 = ng * NTW * 16 + p;
    const StreamCoef sc = stream_coef(S, cs);
    const float slo = stream_floor(S), tlo = stream_floor(T);
    StreamCoef tc[NTW];
#pragma unroll
    for (int t = 0; t < NTW; ++t) tc[t] = stream_coef(T, ct0 + 16 * t);

    const unsigned bytesS = (unsigned)((long long)B * CS * HW * 4), bytesT = (unsigned)((long long)B * CT * HW * 4);
    const __amdgpu_buffer_rsrc_t rS = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(S.p0), 0, bytesS, 0x00020000);
    const __amdgpu_buffer_rsrc_t rU = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(S2 ? S.p1 : S.p0), 0, bytesS, 0x00020000);
    const __amdgpu_buffer_rsrc_t rT = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(T.p0), 0, bytesT, 0x00020000);
    const __amdgpu_buffer_rsrc_t dead = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(S.p0), 0, 0, 0x00020000);
    const int voS = (cs * HW + 4 * kq) * 4, voT = (ct0 * HW + 4 * kq) * 4;
    const int tstep = 16 * HW * 4;                              // bytes between the T channels of consecutive N tiles

    // cursor of the next group to request: byte offsets of (sample, group) in S and T
    int cb = q0 / gpp, cr = q0 - cb * gpp;
    unsigned offS = (unsigned)(((long long)cb * CS * HW + cr * 32) * 4), offT = (unsigned)(((long long)cb * CT * HW + cr * 32) * 4);
    const unsigned wrapS = (unsigned)(CS - 1) * HW * 4, wrapT = (unsigned)(CT - 1) * HW * 4;

    // a stage = one group of 32 pixels: per channel row the two 64-byte halves of a 128-byte line, requested back to back
    f32x4 rs[D][2], ru[D][2], rt[D][NTW][2];
    f32x4 acc[NTW];
#pragma unroll
    for (int t = 0; t < NTW; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // live == false: empty descriptors, the loads return zeros without touching memory (the requests themselves stay
    // unconditional: a load under a branch makes hipcc wait for everything in flight where the paths join)
    auto issue = [&](auto dc, bool live) {
        constexpr int d = decltype(dc)::value;
#pragma unroll
        for (int h = 0; h < 2; ++h) rs[d][h] = __builtin_amdgcn_raw_buffer_load_b128(live ? rS : dead, voS, offS + 64 * h, 0);
        if (S2) {
#pragma unroll
            for (int h = 0; h < 2; ++h) ru[d][h] = __builtin_amdgcn_raw_buffer_load_b128(live ? rU : dead, voS, offS + 64 * h, 0);
        }
#pragma unroll
        for (int t = 0; t < NTW; ++t)
#pragma unroll
            for (int h = 0; h < 2; ++h)
                rt[d][t][h] = __builtin_amdgcn_raw_buffer_load_b128(live ? rT : dead, voT, offT + t * tstep + 64 * h, 0);
        offS += 128; offT += 128;
        if (++cr == gpp) { cr = 0; offS += wrapS; offT += wrapT; }
        __builtin_amdgcn_sched_barrier(0);          // requests leave in stage order: the oldest stage is the one waited for
    };
    auto multiply = [&](auto dc) {
        constexpr int d = decltype(dc)::value;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            f32x4 a = sc.c0 * rs[d][h] + (S2 ? sc.c1 * ru[d][h] + sc.c2 : (f32x4){sc.c2, sc.c2, sc.c2, sc.c2});
            a = sx_max(a, slo);
            f32x4 bt[NTW];
#pragma unroll
            for (int t = 0; t < NTW; ++t) bt[t] = sx_max(tc[t].c0 * rt[d][t][h] + tc[t].c2, tlo);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int t = 0; t < NTW; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], bt[t][j], acc[t], 0, 0, 0);
        }
    };
    auto prologue = [&](auto self, auto dc) -> void {
        constexpr int d = decltype(dc)::value;
        if constexpr (d < D) { issue(dc, d < n); self(self, std::integral_constant<int, d + 1>{}); }
    };
    auto round = [&](auto self, auto dc, int i) -> void {
        constexpr int d = decltype(dc)::value;
        if constexpr (d < D) {
            if (i + d < n) multiply(dc);
            issue(dc, i + d + D < n);
            self(self, std::integral_constant<int, d + 1>{}, i);
        }
    };
    prologue(prologue, std::integral_constant<int, 0>{});
    for (int i = 0; i < n; i += D) round(round, std::integral_constant<int, 0>{}, i);

    // lane holds dW[cs = 16 mt + 4 kq + i][ct = 16 (ng NTW + t) + p]
    const long long E = (long long)CS * CT;
    float *row = slabs + (long long)blockIdx.x * E;
#pragma unroll
    for (int t = 0; t < NTW; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) row[(long long)(mt * 16 + kq * 4 + i) * CT + (ng * NTW + t) * 16 + p] = acc[t][i];
    for (int sl = blockIdx.x + gridDim.x; sl < nslabs; sl += gridDim.x)
        for (long long e = threadIdx.x; e < E; e += 256) slabs[(long long)sl * E + e] = 0.f;
}

// --------------------------------------------------- weight gradient of a 4x4 / stride 2 layer with few T channels
// dW[cs][ct][ky][kx] = sum of S'[cs][y][x] * T'[ct][2 y + ky - 1][2 x + kx - 1]: the first convolution of the wide encoder
// and the last transposed convolution of its decoder (T = the image side, 1-4 channels, twice the S grid).  6.4 GFLOP against
// 0.5-0.9 GB at B = 768: the tiled kernel spent 19 us per 32 KB unit on dependent round trips (0.6-1.0 TB/s).  M = S
// channels, N = (T channel, tap): one N tile of 16 taps per T channel, lane p = tap (ky, kx) = (p >> 2, p & 3); K = S pixels,
// the S rows loaded as in the 1x1 kernel.  For K step j lane (p, kq) needs T at column 2 (x0 + 4 kq + j) + kx - 1: the
// elements 0, 2, 4, 6 of the 8 floats from column 2 x0 + 8 kq + kx - 1 on (two unaligned 16-byte loads; the taps of a row
// read overlapping lines, which the vector L1 absorbs -- T is an eighth of the bytes).  Rows above / below the image and
// the columns -1 / Wt are zeroed after the transform.  Every wave owns ALL MT x NT tiles over its own stages of 32 pixels; the
// four waves of a workgroup add up through LDS into slab blockIdx.x.
template <int MT, int NT, bool S2, int D>
__global__ __launch_bounds__(256, 3) void wgrad_s2_thin_stream_kernel(Operand S, Operand T, float *__restrict__ slabs, int B,
                                                                      int CS, int Hs, int Ws, int nslabs)
{
    constexpr int CT = NT;
    __shared__ float s_acc[4 * MT * NT * 256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, p = lane & 15, kq = lane >> 4;
    const int ky = p >> 2, kx = p & 3;
    const int Ht = 2 * Hs, Wt = 2 * Ws, gx = Ws >> 5;
    const long long total = (long long)B * Hs * gx;              // stages of 32 pixels
    const int w4 = blockIdx.x * 4 + wave, nw = gridDim.x * 4;
    const int q0 = (int)(total * w4 / nw), q1 = (int)(total * (w4 + 1) / nw);
    const int n = q1 - q0;

    StreamCoef sc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) sc[mt] = stream_coef(S, mt * 16 + p);
    StreamCoef tc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) tc[t] = stream_coef(T, t);
    const float slo = stream_floor(S), tlo = stream_floor(T);
    const bool t_ident = T.mode == DM_LOAD_IDENT;

    const unsigned bytesS = (unsigned)((long long)B * CS * Hs * Ws * 4), bytesT = (unsigned)((long long)B * CT * Ht * Wt * 4);
    const __amdgpu_buffer_rsrc_t rS = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(S.p0), 0, bytesS, 0x00020000);
    const __amdgpu_buffer_rsrc_t rU = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(S2 ? S.p1 : S.p0), 0, bytesS, 0x00020000);
    const __amdgpu_buffer_rsrc_t rT = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(T.p0), 0, bytesT, 0x00020000);
    const __amdgpu_buffer_rsrc_t dead = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(S.p0), 0, 0, 0x00020000);
    const int voS = (p * Hs * Ws + 4 * kq) * 4, sstep = 16 * Hs * Ws * 4;
    // T: byte offset of the lane's first element relative to (sample, row 2 y, column 2 x0).  At the tensor's first row the
    // sum is negative for the lanes of row -1 (zeroed anyway) and -4 for tap (1, 0) of the first column: a negative offset
    // fails the descriptor's range check for ALL FOUR dwords, so those lanes load from offset 0 and take their elements one
    // position further left (multiply())
    const int voT = ((ky - 1) * Wt + 8 * kq + kx - 1) * 4, tstep = Ht * Wt * 4;

    int cb = q0 / (Hs * gx), cy = (q0 - cb * Hs * gx) / gx, cxg = q0 - (cb * Hs + cy) * gx;     // cursor of the next request
    f32x4 rs[D][MT][2], ru[D][MT][2], rt[D][NT][2][2];
    int sy[D], sx[D];                                            // row and first column of the stage in each slot
    bool s0[D];                                                  // the stage is the tensor's first
    f32x4 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[mt][t] = (f32x4){0.f, 0.f, 0.f, 0.f};

    auto issue = [&](auto dc, bool live) {
        constexpr int d = decltype(dc)::value;
        const unsigned offS = (unsigned)((((long long)cb * CS * Hs + cy) * Ws + cxg * 32) * 4);
        const int baseT = (int)((((long long)cb * CT * Ht + 2 * cy) * Wt + cxg * 64) * 4);
        sy[d] = cy; sx[d] = cxg * 32; s0[d] = (cb | cy | cxg) == 0;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                rs[d][mt][h] = __builtin_amdgcn_raw_buffer_load_b128(live ? rS : dead, voS, offS + mt * sstep + 64 * h, 0);
                if (S2) ru[d][mt][h] = __builtin_amdgcn_raw_buffer_load_b128(live ? rU : dead, voS, offS + mt * sstep + 64 * h, 0);
            }
        const int vt = baseT + voT, vt0 = vt < 0 ? 0 : vt;       // (negative for channel 0 only: a channel is more than a row)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int v = t == 0 && h == 0 ? vt0 : vt + t * tstep + 128 * h;
                rt[d][t][h][0] = __builtin_amdgcn_raw_buffer_load_b128(live ? rT : dead, v, 0, 0);
                rt[d][t][h][1] = __builtin_amdgcn_raw_buffer_load_b128(live ? rT : dead, v + 16, 0, 0);
            }
        if (++cxg == gx) { cxg = 0; if (++cy == Hs) { cy = 0; ++cb; } }
        __builtin_amdgcn_sched_barrier(0);
    };
    auto multiply = [&](auto dc) {
        constexpr int d = decltype(dc)::value;
        const bool row_out = (sy[d] == 0 && ky == 0) || (sy[d] == Hs - 1 && ky == 3);
        const bool first = row_out || (sx[d] == 0 && kq == 0 && kx == 0);               // column -1: K step 0 of half 0
        const bool last = row_out || (sx[d] + 32 == Ws && kq == 3 && kx == 3);           // column Wt: K step 3 of half 1
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            f32x4 a[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
                a[mt] = sx_max(sc[mt].c0 * rs[d][mt][h] + (S2 ? sc[mt].c1 * ru[d][mt][h] + sc[mt].c2
                                                             : (f32x4){sc[mt].c2, sc[mt].c2, sc[mt].c2, sc[mt].c2}), slo);
            f32x4 bt[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                f32x4 e = (f32x4){rt[d][t][h][0].x, rt[d][t][h][0].z, rt[d][t][h][1].x, rt[d][t][h][1].z};
                if (t == 0 && h == 0 && s0[d] && ky == 1 && kq == 0 && kx == 0)      // loaded from column 0 instead of -1
                    e = (f32x4){0.f, rt[d][t][h][0].y, rt[d][t][h][0].w, rt[d][t][h][1].y};
                if (!t_ident) e = sx_max(tc[t].c0 * e + tc[t].c2, tlo);
                e.x = (h == 0 ? first : row_out) ? 0.f : e.x;
                e.y = row_out ? 0.f : e.y;
                e.z = row_out ? 0.f : e.z;
                e.w = (h == 1 ? last : row_out) ? 0.f : e.w;
                bt[t] = e;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int t = 0; t < NT; ++t)
                        acc[mt][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt][j], bt[t][j], acc[mt][t], 0, 0, 0);
        }
    };
    auto prologue = [&](auto self, auto dc) -> void {
        constexpr int d = decltype(dc)::value;
        if constexpr (d < D) { issue(dc, d < n); self(self, std::integral_constant<int, d + 1>{}); }
    };
    auto round = [&](auto self, auto dc, int i) -> void {
        constexpr int d = decltype(dc)::value;
        if constexpr (d < D) {
            if (i + d < n) multiply(dc);
            issue(dc, i + d + D < n);
            self(self, std::integral_constant<int, d + 1>{}, i);
        }
    };
    prologue(prologue, std::integral_constant<int, 0>{});
    for (int i = 0; i < n; i += D) round(round, std::integral_constant<int, 0>{}, i);

    // the four waves' partial sums -> slab blockIdx.x; lane holds dW[cs = 16 mt + 4 kq + i][ct = t][tap = p]
    const int E = CS * CT * 16;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) s_acc[wave * (MT * NT * 256) + ((mt * 16 + kq * 4 + i) * CT + t) * 16 + p] = acc[mt][t][i];
    __syncthreads();
    float *row = slabs + (long long)blockIdx.x * E;
    for (int e = threadIdx.x; e < MT * NT * 256; e += 256)
        row[e] = (s_acc[e] + s_acc[MT * NT * 256 + e]) + (s_acc[2 * MT * NT * 256 + e] + s_acc[3 * MT * NT * 256 + e]);
    for (int sl = blockIdx.x + gridDim.x; sl < nslabs; sl += gridDim.x)
        for (int e = threadIdx.x; e < E; e += 256) slabs[(long long)sl * E + e] = 0.f;
}

// Epilogue of a unit of 64 output pixels shared by the streaming convolutions: acc[mt][j][i] is channel 16 mt + 4 kq + i of
// pixel 4 p + j, so the four N tiles of one (mt, i) are a float4 of consecutive pixels.  ob: offset of (sample, channel 0, the
// lane's first pixel); cstride: elements per channel plane; s_ep[co] = (bias, gate c0, gate c2, -).  Statistics: the sum over
// the 16 lanes of a row (the unit's 64 pixels) in float, then lane p == 4 mt + i of row kq adds it to ITS channel's double.
// PRE: the gate's values were requested ahead (gm[mt][i], same addresses as the output).
// obytes != 0: gate / residual / statistics operands are read through buffer descriptors (an empty one where the operand is
// absent), all four rows of an M tile requested before the first is used -- as plain loads under their `if (operand)` each was
// its own dependent round trip, up to 48 per unit in the data-gradient forms.
template <int MT, bool PRE = false>
__device__ __forceinline__ void stream_epilogue(const f32x4 (&acc)[MT][4], const Epilogue &ep, const float *s_ep, float *__restrict__ out,
                                                long long ob, long long cstride, int p, int kq, double &st1, double &st2,
                                                const f32x4 (*gm)[4] = nullptr, int co0 = 0, unsigned obytes = 0)
{
    float sel1 = 0.f, sel2 = 0.f;
    const bool q_is_gate = ep.stat_q && ep.stat_q == ep.mask.p0;
    const __amdgpu_buffer_rsrc_t rM = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(ep.mask.p0 ? ep.mask.p0 : out), 0,
                                                                        (ep.mask.p0 && !PRE) ? obytes : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rR = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(ep.resid ? ep.resid : out), 0,
                                                                        ep.resid ? obytes : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rQ = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(ep.stat_q ? ep.stat_q : out), 0,
                                                                        (ep.stats && ep.stat_q && !q_is_gate) ? obytes : 0u, 0x00020000);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        f32x4 bm[4], br[4], bq[4];
        if (obytes) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int off = (int)((ob + (long long)(co0 + 16 * mt + 4 * kq + i) * cstride) * 4);
                bm[i] = __builtin_amdgcn_raw_buffer_load_b128(rM, off, 0, 0);
                br[i] = __builtin_amdgcn_raw_buffer_load_b128(rR, off, 0, 0);
                bq[i] = __builtin_amdgcn_raw_buffer_load_b128(rQ, off, 0, 0);
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int co = co0 + 16 * mt + 4 * kq + i;
            const f32x4 ec = *reinterpret_cast<const f32x4 *>(&s_ep[co * 4]);
            const long long o = ob + (long long)co * cstride;
            f32x4 v = (f32x4){acc[mt][0][i], acc[mt][1][i], acc[mt][2][i], acc[mt][3][i]} + ec.x;
            if (ep.relu) v = dm_relu4(v);
            f32x4 m = v;
            if (ep.mask.p0) {
                m = PRE ? gm[mt][i] : (obytes ? bm[i] : *reinterpret_cast<const f32x4 *>(ep.mask.p0 + o));
                v.x = (ec.y * m.x + ec.z) > 0.f ? v.x : 0.f; v.y = (ec.y * m.y + ec.z) > 0.f ? v.y : 0.f;
                v.z = (ec.y * m.z + ec.z) > 0.f ? v.z : 0.f; v.w = (ec.y * m.w + ec.z) > 0.f ? v.w : 0.f;
            }
            if (ep.resid) v += obytes ? br[i] : *reinterpret_cast<const f32x4 *>(ep.resid + o);
            *reinterpret_cast<f32x4 *>(out + o) = v;
            if (ep.stats) {
                f32x4 q = v;
                if (ep.stat_q) q = q_is_gate ? m : (obytes ? bq[i] : *reinterpret_cast<const f32x4 *>(ep.stat_q + o));
                float a = (v.x + v.y) + (v.z + v.w), c = (v.x * q.x + v.y * q.y) + (v.z * q.z + v.w * q.w);
                a += dpp_mov<0xB1>(a); c += dpp_mov<0xB1>(c);
                a += dpp_mov<0x4E>(a); c += dpp_mov<0x4E>(c);
                a += dpp_mov<0x141>(a); c += dpp_mov<0x141>(c);
                a += dpp_mov<0x140>(a); c += dpp_mov<0x140>(c);
                sel1 = p == 4 * mt + i ? a : sel1;
                sel2 = p == 4 * mt + i ? c : sel2;
            }
        }
    }
    if (ep.stats && MT * 4 > p) { st1 += (double)sel1; st2 += (double)sel2; }
}

// the workgroup's statistics slab from its four waves' per-lane sums (lane (p, kq): channel 16 (p >> 2) + 4 kq + (p & 3))
template <int CO>
__device__ __forceinline__ void stream_stats_out(const Epilogue &ep, double *s_red, double st1, double st2, int nslabs)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, p = lane & 15, kq = lane >> 4;
    if (p < CO / 4) {
        const int co = 16 * (p >> 2) + 4 * kq + (p & 3);
        s_red[(wave * CO + co) * 2] = st1; s_red[(wave * CO + co) * 2 + 1] = st2;
    }
    __syncthreads();
    for (int i = tid; i < CO * 2; i += 256) {
        double s = 0.0;
        for (int w = 0; w < 4; ++w) s += s_red[(w * CO + (i >> 1)) * 2 + (i & 1)];
        ep.stats[((long long)blockIdx.x * CO + (i >> 1)) * 2 + (i & 1)] = s;
    }
    for (int sl = blockIdx.x + gridDim.x; sl < nslabs; sl += gridDim.x)
        for (int i = tid; i < CO * 2; i += 256) ep.stats[((long long)sl * CO + (i >> 1)) * 2 + (i & 1)] = 0.0;
}

// the epilogue's per-channel constants (bias, gate coefficients) -> s_ep[CO][4]; the caller synchronises
__device__ __forceinline__ void stream_stage_ep(const Epilogue &ep, float *s_ep, int CO)
{
    for (int i = threadIdx.x; i < CO; i += 256) {
        float mc0 = 1.f, mc2 = 0.f;
        if (ep.mask.p0 && ep.mask.mode >= DM_LOAD_AFFINE) { mc0 = ep.mask.coef[i * 4]; mc2 = ep.mask.coef[i * 4 + 2]; }
        s_ep[i * 4] = ep.bias ? ep.bias[i] : 0.f; s_ep[i * 4 + 1] = mc0; s_ep[i * 4 + 2] = mc2; s_ep[i * 4 + 3] = 0.f;
    }
}

// ------------------------------------------------------------------------------------------- 1x1 convolution
// out[co][px] = epilogue(sum over ci of W[co][ci] * in'[ci][px]).  M = output channels (the weights are the A operand and
// stay in registers: MT x KS values per lane), N = pixels, K = input channels.  Lane (p, kq) of a K step loads the four
// pixels px0 + 4 p .. + 3 of channel 4 ks + kq -- 16 lanes = 256 contiguous bytes of a channel row -- and the four values
// are the B operands of four N tiles: N tile j, column p is pixel px0 + 4 p + j.  The accumulator of (M tile mt, N tile j)
// then holds, per lane, channels 16 mt + 4 kq + i of that pixel, so the four N tiles of one (mt, i) are a float4 of
// consecutive pixels: 256-byte row segments again on the way out (gate, residual and statistics operands come in the same
// way).  A wave owns units of 64 pixels, R K steps of loads in flight across unit boundaries; the operand coefficients and
// the epilogue's per-channel constants sit in LDS tables (written once per workgroup).
template <int KS4, int MT, bool IN2, int R>
__global__ __launch_bounds__(256, 2) void conv1x1_stream_kernel(Operand in, WeightView wv, float *__restrict__ out, Epilogue ep,
                                                                int B, int HW, int nslabs)
{
    constexpr int KS = 4 * KS4, CIN = 4 * KS, CO = 16 * MT;
    static_assert(KS % R == 0, "the ring divides the K loop");
    __shared__ __attribute__((aligned(16))) float s_cf[CIN * 4];        // (c0, c1, c2, floor) of input channel ci
    __shared__ __attribute__((aligned(16))) float s_ep[CO * 4];         // (bias, gate c0, gate c2, -) of output channel co
    __shared__ double s_red[4 * CO * 2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, p = lane & 15, kq = lane >> 4;
    for (int i = tid; i < CIN; i += 256) {
        const StreamCoef c = stream_coef(in, i);
        s_cf[i * 4] = c.c0; s_cf[i * 4 + 1] = c.c1; s_cf[i * 4 + 2] = c.c2; s_cf[i * 4 + 3] = stream_floor(in);
    }
    stream_stage_ep(ep, s_ep, CO);
    float wA[MT][KS];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) wA[mt][ks] = wv.w[wv.off + (long long)(16 * mt + p) * wv.sn + (long long)(4 * ks + kq) * wv.sc];
    __syncthreads();

    const int upp = HW >> 6;                                     // units of 64 pixels per plane
    const long long total = (long long)B * upp;
    const int u0 = (int)(total * blockIdx.x / gridDim.x), u1 = (int)(total * (blockIdx.x + 1) / gridDim.x);
    const int n = (u1 - u0 - wave + 3) >> 2;                     // this wave's units: u0 + wave, + 4, ...
    const unsigned bytesI = (unsigned)((long long)B * CIN * HW * 4);
    const __amdgpu_buffer_rsrc_t rI = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(in.p0), 0, bytesI, 0x00020000);
    const __amdgpu_buffer_rsrc_t rU = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(IN2 ? in.p1 : in.p0), 0, bytesI, 0x00020000);
    const __amdgpu_buffer_rsrc_t dead = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(in.p0), 0, 0, 0x00020000);
    const int voI = (kq * HW + 4 * p) * 4;
    const int kstep = 4 * HW * 4;                                // bytes between the channel quads of consecutive K steps
    auto unit_in = [&](int u) { const int b = u / upp, r = u - b * upp; return (unsigned)(((long long)b * CIN * HW + r * 64) * 4); };
    auto unit_out = [&](int u) { const int b = u / upp, r = u - b * upp; return ((long long)b * CO * HW + r * 64); };

    f32x4 rx[R], ru[R];
    double st1 = 0.0, st2 = 0.0;                                 // lane p of row kq: channel 16 (p >> 2) + 4 kq + (p & 3)
    unsigned offI = unit_in(u0 + wave < u1 ? u0 + wave : u0);
    auto issue = [&](auto slot, int ks, bool live) {
        constexpr int sl = decltype(slot)::value;
        rx[sl] = __builtin_amdgcn_raw_buffer_load_b128(live ? rI : dead, voI, offI + ks * kstep, 0);
        if (IN2) ru[sl] = __builtin_amdgcn_raw_buffer_load_b128(live ? rU : dead, voI, offI + ks * kstep, 0);
        __builtin_amdgcn_sched_barrier(0);
    };
    auto prologue = [&](auto self, auto kc) -> void {
        constexpr int k = decltype(kc)::value;
        if constexpr (k < R) { issue(kc, k, n > 0); self(self, std::integral_constant<int, k + 1>{}); }
    };
    prologue(prologue, std::integral_constant<int, 0>{});

    for (int it = 0; it < n; ++it) {
        const int u = u0 + wave + 4 * it;
        f32x4 acc[MT][4];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[mt][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const bool more = it + 1 < n;
        const unsigned offN = unit_in(more ? u + 4 : u);
        auto kloop = [&](auto self, auto kc) -> void {
            constexpr int ks = decltype(kc)::value;
            if constexpr (ks < KS) {
                constexpr int sl = ks % R;
                const f32x4 cf = *reinterpret_cast<const f32x4 *>(&s_cf[(4 * ks + kq) * 4]);
                f32x4 x = cf.x * rx[sl] + (IN2 ? cf.y * ru[sl] + cf.z : (f32x4){cf.z, cf.z, cf.z, cf.z});
                x = sx_max(x, cf.w);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[mt][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wA[mt][ks], x[j], acc[mt][j], 0, 0, 0);
                // the slot is free: K step ks + R of this unit, or ks + R - KS of the wave's next one
                if constexpr (ks + R < KS) {
                    issue(std::integral_constant<int, sl>{}, ks + R, true);
                } else {
                    if (ks + R == KS) offI = offN;
                    issue(std::integral_constant<int, sl>{}, ks + R - KS, more);
                }
                self(self, std::integral_constant<int, ks + 1>{});
            }
        };
        kloop(kloop, std::integral_constant<int, 0>{});

        // (the batched descriptor form of the epilogue's operand loads spills here: 241 against 220 us in the data-gradient form)
        stream_epilogue<MT>(acc, ep, s_ep, out, unit_out(u) + 4 * p, HW, p, kq, st1, st2);
    }

    if (ep.stats) stream_stats_out<CO>(ep, s_red, st1, st2, nslabs);
}

// ------------------------------------------------------- 4x4 / stride 2 convolution with few input channels
// out[co][oy][ox] = sum over (ci, ky, kx) of W[co][ci][ky][kx] * in'[ci][2 oy + ky - 1][2 ox + kx - 1]: the wide encoder's first
// convolution (image -> 32 channels) and the data gradient of the decoder's last transposed convolution; 64 output columns
// (128-pixel patches).  K = (ci, ky, kx): K step (ci, kx), lane row kq = ky, so the four rows of lanes read four DIFFERENT
// input rows and nothing is requested twice.  A unit is one output row; for N tile j lane p holds pixel 4 p + j, whose input
// column is 8 p + kx - 1 + 2 j: lane p loads the 8 aligned floats 8 p .. 8 p + 7 of its row (16 lanes = 512 contiguous bytes),
// takes column 8 p - 1 from lane p - 1 and 8 p + 8 from lane p + 1 (DPP row shifts: the lanes beyond the row's ends read the 0
// of the padding) and has the operands of all four kx: (a-1, a1, a3, a5), (a0, a2, a4, a6), (a1, a3, a5, a7), (a2, a4, a6, a8).
// Two loads per input channel and unit instead of one pair per K step -- the form with kq = kx and unaligned 8-float windows
// asked the texture addresser for four times the bytes and ran at 1.8 TB/s.  Rows -1 / H (first and last output row) are
// zeroed after the transform.  The next unit's inputs are requested once this unit's are transformed, its gate values once
// this unit's are used: vector memory returns in order, a gate load issued inside the epilogue would wait for everything
// requested before it.
template <int CIN, int MT>
__global__ __launch_bounds__(256, (CIN * MT >= 6 || MT == 4) ? 2 : 3) void conv_s2_thin_stream_kernel(Operand in, WeightView wv, float *__restrict__ out, Epilogue ep,
                                                                     int B, int H, int nslabs)
{
    constexpr int KS = 4 * CIN, CO = 16 * MT, W = 128, OW = 64;
    __shared__ __attribute__((aligned(16))) float s_ep[CO * 4];
    __shared__ double s_red[4 * CO * 2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, p = lane & 15, kq = lane >> 4;
    stream_stage_ep(ep, s_ep, CO);
    float wA[MT][KS];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
            wA[mt][ks] = wv.w[wv.off + (long long)(16 * mt + p) * wv.sn + (long long)(ks >> 2) * wv.sc + (long long)kq * wv.sky + (long long)(ks & 3) * wv.skx];
    StreamCoef ic[CIN];
#pragma unroll
    for (int c = 0; c < CIN; ++c) ic[c] = stream_coef(in, c);
    const float ilo = stream_floor(in);
    const bool ident = in.mode == DM_LOAD_IDENT;
    __syncthreads();

    const int OH = H >> 1;
    const long long total = (long long)B * OH;
    const int u0 = (int)(total * blockIdx.x / gridDim.x), u1 = (int)(total * (blockIdx.x + 1) / gridDim.x);
    const int n = (u1 - u0 - wave + 3) >> 2;                     // this wave's output rows: u0 + wave, + 4, ...
    const unsigned bytesI = (unsigned)((long long)B * CIN * H * W * 4), bytesO = (unsigned)((long long)B * CO * OH * OW * 4);
    const __amdgpu_buffer_rsrc_t rI = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(in.p0), 0, bytesI, 0x00020000);
    const __amdgpu_buffer_rsrc_t rG = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(ep.mask.p0 ? ep.mask.p0 : in.p0), 0,
                                                                        ep.mask.p0 ? bytesO : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t dead = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(in.p0), 0, 0, 0x00020000);
    const int vl = ((kq - 1) * W + 8 * p) * 4;                   // the lane's row and columns relative to (row 2 oy, column 0)
    const int vg = (4 * kq * OH * OW + 4 * p) * 4;               // gate: channel 4 kq of an M tile, pixels 4 p ..

    f32x4 x0[CIN], x1[CIN], gm[MT][4];
    auto issue_in = [&](int u, bool live) {
        const int b = u / OH, oy = u - b * OH;
#pragma unroll
        for (int c = 0; c < CIN; ++c) {
            int v = (int)((((long long)b * CIN + c) * H + 2 * oy) * W * 4) + vl;
            v = v < 0 ? 0 : v;                                   // row -1 of the tensor's first plane (zeroed below)
            x0[c] = __builtin_amdgcn_raw_buffer_load_b128(live ? rI : dead, v, 0, 0);
            x1[c] = __builtin_amdgcn_raw_buffer_load_b128(live ? rI : dead, v + 16, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    auto issue_gate = [&](int u, bool live) {
        const int b = u / OH, oy = u - b * OH;
        const unsigned sb = (unsigned)((((long long)b * CO * OH + oy) * OW) * 4);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                gm[mt][i] = __builtin_amdgcn_raw_buffer_load_b128(live ? rG : dead, vg, sb + (unsigned)((16 * mt + i) * OH * OW * 4), 0);
        __builtin_amdgcn_sched_barrier(0);
    };
    issue_in(n > 0 ? u0 + wave : 0, n > 0);
    issue_gate(n > 0 ? u0 + wave : 0, n > 0);

    double st1 = 0.0, st2 = 0.0;
    for (int it = 0; it < n; ++it) {
        const int u = u0 + wave + 4 * it;
        const int b = u / OH, oy = u - b * OH;
        const bool more = it + 1 < n;
        f32x4 e[CIN][4];
#pragma unroll
        for (int c = 0; c < CIN; ++c) {
            f32x4 a0 = x0[c], a1 = x1[c];
            if (!ident) { a0 = sx_max(ic[c].c0 * a0 + ic[c].c2, ilo); a1 = sx_max(ic[c].c0 * a1 + ic[c].c2, ilo); }
            if ((oy == 0 && kq == 0) || (oy == OH - 1 && kq == 3)) { a0 = (f32x4){0.f, 0.f, 0.f, 0.f}; a1 = a0; }
            const float am = dpp_mov<0x111>(a1.w), ap = dpp_mov<0x101>(a0.x);      // row_shr:1 / row_shl:1, 0 beyond the row
            e[c][0] = (f32x4){am, a0.y, a0.w, a1.y};
            e[c][1] = (f32x4){a0.x, a0.z, a1.x, a1.z};
            e[c][2] = (f32x4){a0.y, a0.w, a1.y, a1.w};
            e[c][3] = (f32x4){a0.z, a1.x, a1.z, ap};
        }
        issue_in(more ? u + 4 : u, more);
        f32x4 acc[MT][4];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[mt][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[mt][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wA[mt][ks], e[ks >> 2][ks & 3][j], acc[mt][j], 0, 0, 0);
        stream_epilogue<MT, true>(acc, ep, s_ep, out, (((long long)b * CO * OH + oy) * OW) + 4 * p, (long long)OH * OW, p, kq, st1, st2, gm);
        issue_gate(more ? u + 4 : u, more);
    }
    if (ep.stats) stream_stats_out<CO>(ep, s_red, st1, st2, nslabs);
}

// -------------------------------------------- ConvTranspose2d(C -> 1 / 2 channels, 4, 2, 1): the wide decoder's last layer
// out[co][2 y + py][2 x + px] = bias + sum over (ci, a, b) of in'[ci][y - 1 + py + a][x - 1 + px + b] * W[ci][co][3 - py - 2 a][3 - px - 2 b].
// Eight outputs per input pixel: on the matrix cores M or N would be 8 of 16 and 5 of every 9 taps structural zeros (the tiled
// kernel ran it at 11 TFLOP/s of useful products, 550 us for 0.5 GB).  Here it is what the thin family's decoder tail does: vector
// units, one lane = 4 consecutive input pixels of a row (16 lanes = a 64-pixel row, the wave's four rows of lanes = four
// image rows), the weights of one input channel -- 16 taps x (co 0, co 1) -- in SCALAR registers as the pairs v_pk_fma_f32
// multiplies a broadcast input value with (packed [ci][tap][co] by convT_thin_pack_kernel).  Per input channel a lane loads its
// row and the rows above and below (three 16-byte loads), takes columns 4 c - 1 and 4 c + 4 from its neighbours by DPP
// (the row's ends read the padding's 0) and issues 64 packed multiply-adds.  Rows -1 / H are zeroed after the transform, in the
// units that touch them (a wave-uniform branch).
__global__ __launch_bounds__(256) void convT_thin_pack_kernel(WeightView wv, float *__restrict__ wp, int CIN, int COUT)
{
    for (int i = blockIdx.x * 256 + threadIdx.x; i < CIN * 32; i += gridDim.x * 256) {
        const int ci = i >> 5, tap = (i >> 1) & 15, co = i & 1;
        wp[i] = co < COUT ? wv.w[wv.off + (long long)co * wv.sn + (long long)ci * wv.sc + (tap >> 2) * wv.sky + (tap & 3) * wv.skx] : 0.f;
    }
}

template <int COUT>
__global__ __launch_bounds__(256, 4) void convT_thin_stream_kernel(Operand in, const float *__restrict__ wp, float *__restrict__ out,
                                                                   Epilogue ep, int B, int CIN, int H)
{
    constexpr int W = 64, OW = 128;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane >> 4, c = lane & 15;
    const int upb = H >> 2;                                      // units of 4 rows per sample
    const int units = B * upb;
    const unsigned bytesI = (unsigned)((long long)B * CIN * H * W * 4);
    const __amdgpu_buffer_rsrc_t rI = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(in.p0), 0, bytesI, 0x00020000);
    const bool aff = in.mode >= DM_LOAD_AFFINE;
    const float lo = stream_floor(in);
    const float b0 = ep.bias ? ep.bias[0] : 0.f, b1 = (ep.bias && COUT > 1) ? ep.bias[1] : 0.f;
    const int OH = 2 * H;

    for (int u = blockIdx.x * 4 + wave; u < units; u += gridDim.x * 4) {
        const int b = u / upb, y = ((u - b * upb) << 2) + r;
        const bool top = y == 0, bot = y == H - 1;               // the lane's row -1 / row H: zero padding
        const bool edge = (u - b * upb) == 0 || (u - b * upb) == upb - 1;
        // byte offsets of the lane's three rows in channel 0 (clamped into the tensor where the row is padding)
        const int plane = H * W * 4;
        const int o1 = ((b * CIN * H + y) * W + 4 * c) * 4;
        const int o0 = top ? o1 : o1 - W * 4, o2 = bot ? o1 : o1 + W * 4;
        f32x2 acc[4][2][2];                                      // [pixel][py][px] x (co 0, co 1)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[i][q >> 1][q & 1] = (f32x2){0.f, 0.f};
        f32x4 n0 = __builtin_amdgcn_raw_buffer_load_b128(rI, o0, 0, 0), n1 = __builtin_amdgcn_raw_buffer_load_b128(rI, o1, 0, 0),
              n2 = __builtin_amdgcn_raw_buffer_load_b128(rI, o2, 0, 0);
        for (int ci = 0; ci < CIN; ++ci) {
            f32x4 v0 = n0, v1 = n1, v2 = n2;
            {
                const int cn = ci + 1 < CIN ? ci + 1 : ci;       // (the last round re-reads its own rows: harmless)
                n0 = __builtin_amdgcn_raw_buffer_load_b128(rI, o0, cn * plane, 0);
                n1 = __builtin_amdgcn_raw_buffer_load_b128(rI, o1, cn * plane, 0);
                n2 = __builtin_amdgcn_raw_buffer_load_b128(rI, o2, cn * plane, 0);
            }
            float k0 = 1.f, k2 = 0.f;
            if (aff) { k0 = in.coef[ci * 4]; k2 = in.coef[ci * 4 + 2]; }
            v0 = sx_max(k0 * v0 + k2, lo); v1 = sx_max(k0 * v1 + k2, lo); v2 = sx_max(k0 * v2 + k2, lo);
            if (edge) {
                if (top) v0 = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (bot) v2 = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
            // rows as 6 columns: 4 c - 1 (from the left neighbour), 4 c .. 4 c + 3, 4 c + 4 (from the right neighbour)
            float t[3][6];
            t[0][0] = dpp_mov<0x111>(v0.w); t[0][1] = v0.x; t[0][2] = v0.y; t[0][3] = v0.z; t[0][4] = v0.w; t[0][5] = dpp_mov<0x101>(v0.x);
            t[1][0] = dpp_mov<0x111>(v1.w); t[1][1] = v1.x; t[1][2] = v1.y; t[1][3] = v1.z; t[1][4] = v1.w; t[1][5] = dpp_mov<0x101>(v1.x);
            t[2][0] = dpp_mov<0x111>(v2.w); t[2][1] = v2.x; t[2][2] = v2.y; t[2][3] = v2.z; t[2][4] = v2.w; t[2][5] = dpp_mov<0x101>(v2.x);
            const f32x2 *w2 = reinterpret_cast<const f32x2 *>(wp + ci * 32);      // [tap = 4 ky + kx] -> (co 0, co 1): scalar loads
#pragma unroll
            for (int py = 0; py < 2; ++py)
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int px = 0; px < 2; ++px)
#pragma unroll
                        for (int bb = 0; bb < 2; ++bb) {
                            const f32x2 wv2 = w2[(3 - py - 2 * a) * 4 + (3 - px - 2 * bb)];
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                const float x = t[py + a][i + px + bb];      // row y - 1 + py + a, column 4 c + i - 1 + px + bb
                                acc[i][py][px] += (f32x2){x, x} * wv2;
                            }
                        }
        }
        // lane holds, per (co, py): output row 2 y + py, columns 8 c .. 8 c + 7 in the order (pixel i, px)
#pragma unroll
        for (int co = 0; co < COUT; ++co)
#pragma unroll
            for (int py = 0; py < 2; ++py) {
                const float bias = co ? b1 : b0;
                f32x4 lo4 = (f32x4){acc[0][py][0][co], acc[0][py][1][co], acc[1][py][0][co], acc[1][py][1][co]} + bias;
                f32x4 hi4 = (f32x4){acc[2][py][0][co], acc[2][py][1][co], acc[3][py][0][co], acc[3][py][1][co]} + bias;
                if (ep.relu) { lo4 = dm_relu4(lo4); hi4 = dm_relu4(hi4); }
                float *o = out + (((long long)b * COUT + co) * OH + 2 * y + py) * OW + 8 * c;
                *reinterpret_cast<f32x4 *>(o) = lo4;
                *reinterpret_cast<f32x4 *>(o + 4) = hi4;
            }
    }
}

// ------------------------------------------------ 4x4 / stride 2 convolution 32 -> 64 channels on a 64 x 64 grid (-> 32 x 32)
// The wide encoder's second convolution and the data gradient of the decoder's first transposed convolution: 51.5 GFLOP per
// launch at B = 768, matrix bound -- the tiled kernel ran it at 70-80 TFLOP/s (two passes of 32 output channels, each staging
// the input again; a chunk of 8 channels between two barriers with 128 matrix instructions per wave to hide its loads behind).
// Here the WEIGHTS are what sits in LDS -- all of them, 128 KB, written once per workgroup in the order the A operand reads
// them ([K step = (ci, kx)][M tile][ky][co % 16]: 64 consecutive floats per read) -- and the activations go straight from
// global memory into the B operand as in conv_s2_thin_stream_kernel (K step (ci, kx), rows of lanes = ky, two aligned loads
// per input channel, neighbours by DPP).  Eight waves per workgroup (one workgroup per CU), each with its own stream of
// units and no barrier after the weights are in.  A unit is two output rows x 32 pixels: lanes p < 8 hold row oy, the others
// row oy + 1, pixel 4 (p & 7) + j in N tile j.
template <bool AFF, bool IN2 = false>
__global__ __launch_bounds__(512, 1) void conv_s2_wide_stream_kernel(Operand in, WeightView wv, float *__restrict__ out, Epilogue ep,
                                                                     int B, int nslabs)
{
    constexpr int CIN = 32, CO = 64, MT = 4, H = 64, W = 64, OH = 32, OW = 32, KS = CIN * 4, R = IN2 ? 2 : 4;   // (IN2: an AFFINE2 input, two tensors)
    extern __shared__ __attribute__((aligned(16))) float s_w[];          // [KS][MT][4 ky][16]: 128 KB
    __shared__ __attribute__((aligned(16))) float s_ep[CO * 4];
    __shared__ double s_red[8 * CO * 2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, p = lane & 15, kq = lane >> 4;
    for (int i = tid; i < KS * MT * 64; i += 512) {
        const int ks = i >> 8, mt = (i >> 6) & 3, ky = (i >> 4) & 3, m = i & 15;
        s_w[i] = wv.w[wv.off + (long long)(16 * mt + m) * wv.sn + (long long)(ks >> 2) * wv.sc + (long long)ky * wv.sky + (long long)(ks & 3) * wv.skx];
    }
    for (int i = tid; i < CO; i += 512) {
        float mc0 = 1.f, mc2 = 0.f;
        if (ep.mask.p0 && ep.mask.mode >= DM_LOAD_AFFINE) { mc0 = ep.mask.coef[i * 4]; mc2 = ep.mask.coef[i * 4 + 2]; }
        s_ep[i * 4] = ep.bias ? ep.bias[i] : 0.f; s_ep[i * 4 + 1] = mc0; s_ep[i * 4 + 2] = mc2; s_ep[i * 4 + 3] = 0.f;
    }
    __syncthreads();

    const int upb = OH >> 1;                                     // units of two output rows per sample
    const long long total = (long long)B * upb;
    const int u0 = (int)(total * blockIdx.x / gridDim.x), u1 = (int)(total * (blockIdx.x + 1) / gridDim.x);
    const int n = (u1 - u0 - wave + 7) >> 3;                     // this wave's units: u0 + wave, + 8, ...
    const unsigned bytesI = (unsigned)((long long)B * CIN * H * W * 4);
    const __amdgpu_buffer_rsrc_t rI = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(in.p0), 0, bytesI, 0x00020000);
    const __amdgpu_buffer_rsrc_t rU = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(IN2 ? in.p1 : in.p0), 0, bytesI, 0x00020000);
    const __amdgpu_buffer_rsrc_t dead = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(in.p0), 0, 0, 0x00020000);
    const int half = p >> 3, pc = p & 7;
    const float ilo = stream_floor(in);
    const float *sa = s_w + kq * 16 + p;                         // + (ks * MT + mt) * 64

    f32x4 x0[R], x1[R], y0[IN2 ? R : 1], y1[IN2 ? R : 1];
    int vbase = 0;                                               // byte offset of the lane's row / columns in channel 0 of the unit in flight
    bool rowbad = false;
    auto unit_begin = [&](int u) {
        const int b = u / upb, oy = ((u - b * upb) << 1) + half;
        const int r = 2 * oy + kq - 1;
        const int rc = r < 0 ? 0 : (r >= H ? H - 1 : r);
        vbase = ((b * CIN * H + rc) * W + 8 * pc) * 4;
        return r != rc;
    };
    auto issue = [&](auto slot, int ci, bool live) {
        constexpr int sl = decltype(slot)::value;
        x0[sl] = __builtin_amdgcn_raw_buffer_load_b128(live ? rI : dead, vbase, ci * (H * W * 4), 0);
        x1[sl] = __builtin_amdgcn_raw_buffer_load_b128(live ? rI : dead, vbase + 16, ci * (H * W * 4), 0);
        if constexpr (IN2) {
            y0[sl] = __builtin_amdgcn_raw_buffer_load_b128(live ? rU : dead, vbase, ci * (H * W * 4), 0);
            y1[sl] = __builtin_amdgcn_raw_buffer_load_b128(live ? rU : dead, vbase + 16, ci * (H * W * 4), 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    bool nextbad = unit_begin(n > 0 ? u0 + wave : 0);
    {
        auto pro = [&](auto self, auto kc) -> void {
            constexpr int k = decltype(kc)::value;
            if constexpr (k < R) { issue(kc, k, n > 0); self(self, std::integral_constant<int, k + 1>{}); }
        };
        pro(pro, std::integral_constant<int, 0>{});
    }

    double st1 = 0.0, st2 = 0.0;
    for (int it = 0; it < n; ++it) {
        const int u = u0 + wave + 8 * it;
        const int b = u / upb, oy = ((u - b * upb) << 1) + half;
        const bool more = it + 1 < n;
        rowbad = nextbad;
        f32x4 acc[MT][4];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[mt][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
        for (int cb = 0; cb < CIN; cb += R) {
            // R input channels per round: their loads were requested a round ago; the slots are refilled as they are consumed
            if (cb + R == CIN) nextbad = unit_begin(more ? u + 8 : u);          // the refills below belong to the next unit
            auto chan = [&](auto self, auto cc) -> void {
                constexpr int c = decltype(cc)::value;
                if constexpr (c < R) {
                    const int ci = cb + c;
                    f32x4 a0 = x0[c], a1 = x1[c];
                    if constexpr (IN2) {
                        const StreamCoef kc = stream_coef(in, ci);
                        a0 = kc.c0 * a0 + (kc.c1 * y0[c] + kc.c2); a1 = kc.c0 * a1 + (kc.c1 * y1[c] + kc.c2);
                    } else if (AFF) {
                        const StreamCoef kc = stream_coef(in, ci);
                        a0 = sx_max(kc.c0 * a0 + kc.c2, ilo); a1 = sx_max(kc.c0 * a1 + kc.c2, ilo);
                    }
                    if (rowbad) { a0 = (f32x4){0.f, 0.f, 0.f, 0.f}; a1 = a0; }
                    float am = dpp_mov<0x111>(a1.w), ap = dpp_mov<0x101>(a0.x);
                    am = pc == 0 ? 0.f : am;                                     // the lane to the left belongs to the other row
                    ap = pc == 7 ? 0.f : ap;
                    {
                        const int cn = cb + R + c;                               // this slot's next channel: this unit's, or the next unit's
                        issue(cc, cn < CIN ? cn : cn - CIN, cn < CIN ? true : more);
                    }
                    const f32x4 e0 = (f32x4){am, a0.y, a0.w, a1.y}, e1 = (f32x4){a0.x, a0.z, a1.x, a1.z};
                    const f32x4 e2 = (f32x4){a0.y, a0.w, a1.y, a1.w}, e3 = (f32x4){a0.z, a1.x, a1.z, ap};
                    const float *wp = sa + ci * (4 * MT * 64);
                    auto step = [&](const f32x4 &e, int kx) {
                        float wa[MT];
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) wa[mt] = wp[(kx * MT + mt) * 64];
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                            for (int j = 0; j < 4; ++j) acc[mt][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[mt], e[j], acc[mt][j], 0, 0, 0);
                    };
                    step(e0, 0); step(e1, 1); step(e2, 2); step(e3, 3);
                    self(self, std::integral_constant<int, c + 1>{});
                }
            };
            // (vbase switches to the next unit before the LAST round's refills; rows of THIS unit were all requested by then)
            chan(chan, std::integral_constant<int, 0>{});
        }
        stream_epilogue<MT>(acc, ep, s_ep, out, (((long long)b * CO * OH + oy) * OW) + 4 * pc, (long long)OH * OW, p, kq, st1, st2, nullptr, 0,
                            (unsigned)((long long)B * CO * OH * OW * 4));
    }
    if (ep.stats) {
        const int co = 16 * (p >> 2) + 4 * kq + (p & 3);
        s_red[(wave * CO + co) * 2] = st1; s_red[(wave * CO + co) * 2 + 1] = st2;
        __syncthreads();
        for (int i = tid; i < CO * 2; i += 512) {
            double sum = 0.0;
            for (int w = 0; w < 8; ++w) sum += s_red[(w * CO + (i >> 1)) * 2 + (i & 1)];
            ep.stats[((long long)blockIdx.x * CO + (i >> 1)) * 2 + (i & 1)] = sum;
        }
        for (int sl = blockIdx.x + gridDim.x; sl < nslabs; sl += gridDim.x)
            for (int i = tid; i < CO * 2; i += 512) ep.stats[((long long)sl * CO + (i >> 1)) * 2 + (i & 1)] = 0.0;
    }
}

// --------------------------------------- ConvTranspose2d(64 -> 32, 4, 2, 1) on a 32 x 32 grid (-> 64 x 64), weights in LDS
// The wide decoder's first transposed convolution and the data gradient of the encoder's second convolution: the mirror image
// of conv_s2_wide_stream_kernel.  out[co][2 y + py][2 x + px] = sum over (ci, a, b) of in'[ci][y - 1 + py + a][x - 1 + px + b] *
// W[ci][co][3 - py - 2 a][3 - px - 2 b]: per output parity a 2 x 2 convolution over 64 channels, K = 256.  M = output channels
// (2 tiles), N = input pixels, K step = 4 input channels (lane row kq = channel) x one (parity, tap): 16 of them per channel
// quad, 128 matrix instructions, no structural zero.  All weights (128 KB) sit in LDS as the A operand reads them; a lane loads
// its pixel quad of the rows y - 1, y, y + 1 of ITS channel (three 16-byte loads per K step, neighbours by DPP) and every
// (dy, dx) shift of them is a B operand.  A unit is two input rows x 32 pixels (lanes p < 8: row y, the others y + 1); the four
// parities of a lane's four pixels leave as 8 consecutive output columns per (channel, output row).
template <bool IN2>
__global__ __launch_bounds__(512, 1) void convT_wide_stream_kernel(Operand in, WeightView wv, float *__restrict__ out, Epilogue ep,
                                                                   int B, int nslabs)
{
    constexpr int CIN = 64, CO = 32, MT = 2, H = 32, W = 32, OH = 64, OW = 64, KS = CIN / 4, R = 2;
    extern __shared__ __attribute__((aligned(16))) float s_w[];          // [KS][16 taps][MT][4 ci][16 co]: 128 KB
    __shared__ __attribute__((aligned(16))) float s_cf[CIN * 4];
    __shared__ __attribute__((aligned(16))) float s_ep[CO * 4];
    __shared__ double s_red[8 * CO * 2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, p = lane & 15, kq = lane >> 4;
    for (int i = tid; i < KS * 16 * MT * 64; i += 512) {
        const int ks = i >> 11, tap = (i >> 7) & 15, mt = (i >> 6) & 1, c = (i >> 4) & 3, m = i & 15;
        s_w[i] = wv.w[wv.off + (long long)(16 * mt + m) * wv.sn + (long long)(4 * ks + c) * wv.sc + (long long)(tap >> 2) * wv.sky + (long long)(tap & 3) * wv.skx];
    }
    for (int i = tid; i < CIN; i += 512) {
        const StreamCoef c = stream_coef(in, i);
        s_cf[i * 4] = c.c0; s_cf[i * 4 + 1] = c.c1; s_cf[i * 4 + 2] = c.c2; s_cf[i * 4 + 3] = stream_floor(in);
    }
    for (int i = tid; i < CO; i += 512) {
        float mc0 = 1.f, mc2 = 0.f;
        if (ep.mask.p0 && ep.mask.mode >= DM_LOAD_AFFINE) { mc0 = ep.mask.coef[i * 4]; mc2 = ep.mask.coef[i * 4 + 2]; }
        s_ep[i * 4] = ep.bias ? ep.bias[i] : 0.f; s_ep[i * 4 + 1] = mc0; s_ep[i * 4 + 2] = mc2; s_ep[i * 4 + 3] = 0.f;
    }
    __syncthreads();

    const int upb = H >> 1;
    const long long total = (long long)B * upb;
    const int u0 = (int)(total * blockIdx.x / gridDim.x), u1 = (int)(total * (blockIdx.x + 1) / gridDim.x);
    const int n = (u1 - u0 - wave + 7) >> 3;                     // this wave's units: u0 + wave, + 8, ...
    const unsigned bytesI = (unsigned)((long long)B * CIN * H * W * 4);
    const __amdgpu_buffer_rsrc_t rI = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(in.p0), 0, bytesI, 0x00020000);
    const __amdgpu_buffer_rsrc_t rU = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(IN2 ? in.p1 : in.p0), 0, bytesI, 0x00020000);
    const __amdgpu_buffer_rsrc_t dead = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(in.p0), 0, 0, 0x00020000);
    const int half = p >> 3, pc = p & 7;
    const float *sa = s_w + kq * 16 + p;                         // + ((ks * 16 + tap) * MT + mt) * 64

    f32x4 r0[R], r1[R], r2[R], q0[IN2 ? R : 1], q1[IN2 ? R : 1], q2[IN2 ? R : 1];
    int v0 = 0, v1 = 0, v2 = 0;                                  // byte offsets of the lane's three rows in channel kq of the unit in flight
    bool top = false, bot = false;
    auto unit_begin = [&](int u, bool &t, bool &bo) {
        const int b = u / upb, y = ((u - b * upb) << 1) + half;
        t = y == 0; bo = y == H - 1;
        v1 = (((b * CIN + kq) * H + y) * W + 4 * pc) * 4;
        v0 = t ? v1 : v1 - W * 4;
        v2 = bo ? v1 : v1 + W * 4;
    };
    auto issue = [&](auto slot, int ks, bool live) {
        constexpr int sl = decltype(slot)::value;
        const int so = ks * (4 * H * W * 4);
        r0[sl] = __builtin_amdgcn_raw_buffer_load_b128(live ? rI : dead, v0, so, 0);
        r1[sl] = __builtin_amdgcn_raw_buffer_load_b128(live ? rI : dead, v1, so, 0);
        r2[sl] = __builtin_amdgcn_raw_buffer_load_b128(live ? rI : dead, v2, so, 0);
        if constexpr (IN2) {
            q0[sl] = __builtin_amdgcn_raw_buffer_load_b128(live ? rU : dead, v0, so, 0);
            q1[sl] = __builtin_amdgcn_raw_buffer_load_b128(live ? rU : dead, v1, so, 0);
            q2[sl] = __builtin_amdgcn_raw_buffer_load_b128(live ? rU : dead, v2, so, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    bool ntop = false, nbot = false;
    unit_begin(n > 0 ? u0 + wave : 0, ntop, nbot);
    issue(std::integral_constant<int, 0>{}, 0, n > 0);
    issue(std::integral_constant<int, 1>{}, 1, n > 0);

    double st1 = 0.0, st2 = 0.0;
    for (int it = 0; it < n; ++it) {
        const int u = u0 + wave + 8 * it;
        const int b = u / upb, y = ((u - b * upb) << 1) + half;
        const bool more = it + 1 < n;
        top = ntop; bot = nbot;
        f32x4 acc[2][2][MT][4];                                  // [py][px][mt][pixel j] x 4 channels
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[q >> 1][q & 1][mt][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
        for (int kb = 0; kb < KS; kb += R) {
            if (kb + R == KS) unit_begin(more ? u + 8 : u, ntop, nbot);     // the refills of the last round belong to the next unit
            auto kstep = [&](auto self, auto cc) -> void {
                constexpr int c = decltype(cc)::value;
                if constexpr (c < R) {
                    const int ks = kb + c;
                    const f32x4 cf = *reinterpret_cast<const f32x4 *>(&s_cf[(4 * ks + kq) * 4]);
                    f32x4 a0 = r0[c], a1 = r1[c], a2 = r2[c];
                    if constexpr (IN2) {
                        a0 = cf.x * a0 + (cf.y * q0[c] + cf.z); a1 = cf.x * a1 + (cf.y * q1[c] + cf.z); a2 = cf.x * a2 + (cf.y * q2[c] + cf.z);
                    } else {
                        a0 = cf.x * a0 + cf.z; a1 = cf.x * a1 + cf.z; a2 = cf.x * a2 + cf.z;
                    }
                    a0 = sx_max(a0, cf.w); a1 = sx_max(a1, cf.w); a2 = sx_max(a2, cf.w);
                    if (top) a0 = (f32x4){0.f, 0.f, 0.f, 0.f};
                    if (bot) a2 = (f32x4){0.f, 0.f, 0.f, 0.f};
                    {
                        const int kn = kb + R + c;
                        issue(cc, kn < KS ? kn : kn - KS, kn < KS ? true : more);
                    }
                    // rows as 6 columns: 4 pc - 1 (left neighbour; 0 at the row's start), 4 pc .. + 3, 4 pc + 4 (right neighbour)
                    float t[3][6];
                    const bool ls = pc == 0, rs = pc == 7;
#define DM_ROW6(T, A)                                                                                            \
                    { const float l = dpp_mov<0x111>(A.w), r = dpp_mov<0x101>(A.x);                               \
                      T[0] = ls ? 0.f : l; T[1] = A.x; T[2] = A.y; T[3] = A.z; T[4] = A.w; T[5] = rs ? 0.f : r; }
                    DM_ROW6(t[0], a0) DM_ROW6(t[1], a1) DM_ROW6(t[2], a2)
#undef DM_ROW6
                    const float *wp = sa + ks * (16 * MT * 64);
#pragma unroll
                    for (int py = 0; py < 2; ++py)
#pragma unroll
                        for (int a = 0; a < 2; ++a)
#pragma unroll
                            for (int px = 0; px < 2; ++px)
#pragma unroll
                                for (int bb = 0; bb < 2; ++bb) {
                                    const int tap = (3 - py - 2 * a) * 4 + (3 - px - 2 * bb);
                                    float wa[MT];
#pragma unroll
                                    for (int mt = 0; mt < MT; ++mt) wa[mt] = wp[(tap * MT + mt) * 64];
#pragma unroll
                                    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                                        for (int j = 0; j < 4; ++j)
                                            acc[py][px][mt][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[mt], t[py + a][j + px + bb], acc[py][px][mt][j], 0, 0, 0);
                                }
                    self(self, std::integral_constant<int, c + 1>{});
                }
            };
            kstep(kstep, std::integral_constant<int, 0>{});
        }
        // ---- epilogue: (mt, i) -> channel 16 mt + 4 kq + i; per output row 2 y + py the lane's 8 columns 8 pc .. 8 pc + 7 = (pixel j, px).
        // Gate / residual / statistics operands through descriptors (empty where absent), the four float4 of a channel's two
        // output rows requested together: as plain loads under their `if` each was its own dependent round trip.
        float sel1 = 0.f, sel2 = 0.f;
        const unsigned obytes = (unsigned)((long long)B * CO * OH * OW * 4);
        const bool q_is_gate = ep.stat_q && ep.stat_q == ep.mask.p0;
        const __amdgpu_buffer_rsrc_t rM = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(ep.mask.p0 ? ep.mask.p0 : out), 0, ep.mask.p0 ? obytes : 0u, 0x00020000);
        const __amdgpu_buffer_rsrc_t rR = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(ep.resid ? ep.resid : out), 0, ep.resid ? obytes : 0u, 0x00020000);
        const __amdgpu_buffer_rsrc_t rQ = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(ep.stat_q ? ep.stat_q : out), 0,
                                                                            (ep.stats && ep.stat_q && !q_is_gate) ? obytes : 0u, 0x00020000);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int co = 16 * mt + 4 * kq + i;
                const f32x4 ec = *reinterpret_cast<const f32x4 *>(&s_ep[co * 4]);
                float sa1 = 0.f, sa2 = 0.f;
                f32x4 bm[2][2], br[2][2], bq[2][2];
#pragma unroll
                for (int py = 0; py < 2; ++py) {
                    const int off = (int)(((((long long)b * CO + co) * OH + 2 * y + py) * OW + 8 * pc) * 4);
                    bm[py][0] = __builtin_amdgcn_raw_buffer_load_b128(rM, off, 0, 0); bm[py][1] = __builtin_amdgcn_raw_buffer_load_b128(rM, off + 16, 0, 0);
                    br[py][0] = __builtin_amdgcn_raw_buffer_load_b128(rR, off, 0, 0); br[py][1] = __builtin_amdgcn_raw_buffer_load_b128(rR, off + 16, 0, 0);
                    bq[py][0] = __builtin_amdgcn_raw_buffer_load_b128(rQ, off, 0, 0); bq[py][1] = __builtin_amdgcn_raw_buffer_load_b128(rQ, off + 16, 0, 0);
                }
#pragma unroll
                for (int py = 0; py < 2; ++py) {
                    const long long o = (((long long)b * CO + co) * OH + 2 * y + py) * OW + 8 * pc;
                    f32x4 lo = (f32x4){acc[py][0][mt][0][i], acc[py][1][mt][0][i], acc[py][0][mt][1][i], acc[py][1][mt][1][i]} + ec.x;
                    f32x4 hi = (f32x4){acc[py][0][mt][2][i], acc[py][1][mt][2][i], acc[py][0][mt][3][i], acc[py][1][mt][3][i]} + ec.x;
                    if (ep.relu) { lo = dm_relu4(lo); hi = dm_relu4(hi); }
                    const f32x4 ml = bm[py][0], mh = bm[py][1];
                    if (ep.mask.p0) {
                        lo.x = (ec.y * ml.x + ec.z) > 0.f ? lo.x : 0.f; lo.y = (ec.y * ml.y + ec.z) > 0.f ? lo.y : 0.f;
                        lo.z = (ec.y * ml.z + ec.z) > 0.f ? lo.z : 0.f; lo.w = (ec.y * ml.w + ec.z) > 0.f ? lo.w : 0.f;
                        hi.x = (ec.y * mh.x + ec.z) > 0.f ? hi.x : 0.f; hi.y = (ec.y * mh.y + ec.z) > 0.f ? hi.y : 0.f;
                        hi.z = (ec.y * mh.z + ec.z) > 0.f ? hi.z : 0.f; hi.w = (ec.y * mh.w + ec.z) > 0.f ? hi.w : 0.f;
                    }
                    if (ep.resid) { lo += br[py][0]; hi += br[py][1]; }
                    *reinterpret_cast<f32x4 *>(out + o) = lo;
                    *reinterpret_cast<f32x4 *>(out + o + 4) = hi;
                    if (ep.stats) {
                        f32x4 ql = lo, qh = hi;
                        if (ep.stat_q) { ql = q_is_gate ? ml : bq[py][0]; qh = q_is_gate ? mh : bq[py][1]; }
                        sa1 += ((lo.x + lo.y) + (lo.z + lo.w)) + ((hi.x + hi.y) + (hi.z + hi.w));
                        sa2 += ((lo.x * ql.x + lo.y * ql.y) + (lo.z * ql.z + lo.w * ql.w)) + ((hi.x * qh.x + hi.y * qh.y) + (hi.z * qh.z + hi.w * qh.w));
                    }
                }
                if (ep.stats) {
                    sa1 += dpp_mov<0xB1>(sa1); sa2 += dpp_mov<0xB1>(sa2);
                    sa1 += dpp_mov<0x4E>(sa1); sa2 += dpp_mov<0x4E>(sa2);
                    sa1 += dpp_mov<0x141>(sa1); sa2 += dpp_mov<0x141>(sa2);
                    sa1 += dpp_mov<0x140>(sa1); sa2 += dpp_mov<0x140>(sa2);
                    sel1 = p == 4 * mt + i ? sa1 : sel1;
                    sel2 = p == 4 * mt + i ? sa2 : sel2;
                }
            }
        if (ep.stats && p < 4 * MT) { st1 += (double)sel1; st2 += (double)sel2; }
    }
    if (ep.stats) {
        if (p < 4 * MT) {
            const int co = 16 * (p >> 2) + 4 * kq + (p & 3);
            s_red[(wave * CO + co) * 2] = st1; s_red[(wave * CO + co) * 2 + 1] = st2;
        }
        __syncthreads();
        for (int i = tid; i < CO * 2; i += 512) {
            double sum = 0.0;
            for (int w = 0; w < 8; ++w) sum += s_red[(w * CO + (i >> 1)) * 2 + (i & 1)];
            ep.stats[((long long)blockIdx.x * CO + (i >> 1)) * 2 + (i & 1)] = sum;
        }
        for (int sl = blockIdx.x + gridDim.x; sl < nslabs; sl += gridDim.x)
            for (int i = tid; i < CO * 2; i += 512) ep.stats[((long long)sl * CO + (i >> 1)) * 2 + (i & 1)] = 0.0;
    }
}

// ------------------------------------------------------- 3x3 convolution 64 -> 64 channels on a 32 x 32 grid, weights in LDS
// The residual blocks' 3x3 layers (forward and data gradient): the third member of the family above.  All 147 KB of weights in
// LDS as the A operand reads them ([K step = 4 input channels][tap][M tile][ci % 4][co % 16]); a lane loads its pixel quad of
// the rows y - 1, y, y + 1 of its channel and every (dy, dx) shift is a B operand: 9 taps x 4 M tiles x 4 pixel tiles = 144
// matrix instructions per three loads.  Two rows of 32 pixels per unit, eight independent waves per workgroup.
template <bool IN2>
__global__ __launch_bounds__(512, 1) void conv3x3_wide_stream_kernel(Operand in, WeightView wv, float *__restrict__ out, Epilogue ep,
                                                                     int B, int nslabs)
{
    constexpr int CIN = 64, CO = 64, MT = 4, H = 32, W = 32, KS = CIN / 4, R = 2;
    extern __shared__ __attribute__((aligned(16))) float s_w[];          // [KS][9 taps][MT][4 ci][16 co]: 144 KB
    __shared__ __attribute__((aligned(16))) float s_cf[CIN * 4];
    __shared__ __attribute__((aligned(16))) float s_ep[CO * 4];
    __shared__ double s_red[4 * CO * 2];                                 // (two waves share a row: see the end)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, p = lane & 15, kq = lane >> 4;
    for (int i = tid; i < KS * 9 * MT * 64; i += 512) {
        const int m = i & 15, c = (i >> 4) & 3, mt = (i >> 6) & 3, r = i >> 8, tap = r % 9, ks = r / 9;
        s_w[i] = wv.w[wv.off + (long long)(16 * mt + m) * wv.sn + (long long)(4 * ks + c) * wv.sc + (long long)(tap / 3) * wv.sky + (long long)(tap % 3) * wv.skx];
    }
    for (int i = tid; i < CIN; i += 512) {
        const StreamCoef c = stream_coef(in, i);
        s_cf[i * 4] = c.c0; s_cf[i * 4 + 1] = c.c1; s_cf[i * 4 + 2] = c.c2; s_cf[i * 4 + 3] = stream_floor(in);
    }
    for (int i = tid; i < CO; i += 512) {
        float mc0 = 1.f, mc2 = 0.f;
        if (ep.mask.p0 && ep.mask.mode >= DM_LOAD_AFFINE) { mc0 = ep.mask.coef[i * 4]; mc2 = ep.mask.coef[i * 4 + 2]; }
        s_ep[i * 4] = ep.bias ? ep.bias[i] : 0.f; s_ep[i * 4 + 1] = mc0; s_ep[i * 4 + 2] = mc2; s_ep[i * 4 + 3] = 0.f;
    }
    __syncthreads();

    const int upb = H >> 1;
    const long long total = (long long)B * upb;
    const int u0 = (int)(total * blockIdx.x / gridDim.x), u1 = (int)(total * (blockIdx.x + 1) / gridDim.x);
    const int n = (u1 - u0 - wave + 7) >> 3;
    const unsigned bytesI = (unsigned)((long long)B * CIN * H * W * 4);
    const __amdgpu_buffer_rsrc_t rI = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(in.p0), 0, bytesI, 0x00020000);
    const __amdgpu_buffer_rsrc_t rU = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(IN2 ? in.p1 : in.p0), 0, bytesI, 0x00020000);
    const __amdgpu_buffer_rsrc_t dead = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(in.p0), 0, 0, 0x00020000);
    const int half = p >> 3, pc = p & 7;
    const float *sa = s_w + kq * 16 + p;                         // + ((ks * 9 + tap) * MT + mt) * 64

    f32x4 r0[R], r1[R], r2[R], q0[IN2 ? R : 1], q1[IN2 ? R : 1], q2[IN2 ? R : 1];
    int v0 = 0, v1 = 0, v2 = 0;
    bool top = false, bot = false, ntop = false, nbot = false;
    auto unit_begin = [&](int u, bool &t, bool &bo) {
        const int b = u / upb, y = ((u - b * upb) << 1) + half;
        t = y == 0; bo = y == H - 1;
        v1 = (((b * CIN + kq) * H + y) * W + 4 * pc) * 4;
        v0 = t ? v1 : v1 - W * 4;
        v2 = bo ? v1 : v1 + W * 4;
    };
    auto issue = [&](auto slot, int ks, bool live) {
        constexpr int sl = decltype(slot)::value;
        const int so = ks * (4 * H * W * 4);
        r0[sl] = __builtin_amdgcn_raw_buffer_load_b128(live ? rI : dead, v0, so, 0);
        r1[sl] = __builtin_amdgcn_raw_buffer_load_b128(live ? rI : dead, v1, so, 0);
        r2[sl] = __builtin_amdgcn_raw_buffer_load_b128(live ? rI : dead, v2, so, 0);
        if constexpr (IN2) {
            q0[sl] = __builtin_amdgcn_raw_buffer_load_b128(live ? rU : dead, v0, so, 0);
            q1[sl] = __builtin_amdgcn_raw_buffer_load_b128(live ? rU : dead, v1, so, 0);
            q2[sl] = __builtin_amdgcn_raw_buffer_load_b128(live ? rU : dead, v2, so, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    unit_begin(n > 0 ? u0 + wave : 0, ntop, nbot);
    issue(std::integral_constant<int, 0>{}, 0, n > 0);
    issue(std::integral_constant<int, 1>{}, 1, n > 0);

    double st1 = 0.0, st2 = 0.0;
    for (int it = 0; it < n; ++it) {
        const int u = u0 + wave + 8 * it;
        const int b = u / upb, y = ((u - b * upb) << 1) + half;
        const bool more = it + 1 < n;
        top = ntop; bot = nbot;
        f32x4 acc[MT][4];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[mt][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
        for (int kb = 0; kb < KS; kb += R) {
            if (kb + R == KS) unit_begin(more ? u + 8 : u, ntop, nbot);
            auto kstep = [&](auto self, auto cc) -> void {
                constexpr int c = decltype(cc)::value;
                if constexpr (c < R) {
                    const int ks = kb + c;
                    const f32x4 cf = *reinterpret_cast<const f32x4 *>(&s_cf[(4 * ks + kq) * 4]);
                    f32x4 a0 = r0[c], a1 = r1[c], a2 = r2[c];
                    if constexpr (IN2) {
                        a0 = cf.x * a0 + (cf.y * q0[c] + cf.z); a1 = cf.x * a1 + (cf.y * q1[c] + cf.z); a2 = cf.x * a2 + (cf.y * q2[c] + cf.z);
                    } else {
                        a0 = cf.x * a0 + cf.z; a1 = cf.x * a1 + cf.z; a2 = cf.x * a2 + cf.z;
                    }
                    a0 = sx_max(a0, cf.w); a1 = sx_max(a1, cf.w); a2 = sx_max(a2, cf.w);
                    if (top) a0 = (f32x4){0.f, 0.f, 0.f, 0.f};
                    if (bot) a2 = (f32x4){0.f, 0.f, 0.f, 0.f};
                    {
                        const int kn = kb + R + c;
                        issue(cc, kn < KS ? kn : kn - KS, kn < KS ? true : more);
                    }
                    float t[3][6];
                    const bool ls = pc == 0, rs = pc == 7;
#define DM_ROW6(T, A)                                                                                            \
                    { const float l = dpp_mov<0x111>(A.w), r = dpp_mov<0x101>(A.x);                               \
                      T[0] = ls ? 0.f : l; T[1] = A.x; T[2] = A.y; T[3] = A.z; T[4] = A.w; T[5] = rs ? 0.f : r; }
                    DM_ROW6(t[0], a0) DM_ROW6(t[1], a1) DM_ROW6(t[2], a2)
#undef DM_ROW6
                    const float *wp = sa + ks * (9 * MT * 64);
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                        for (int kx = 0; kx < 3; ++kx) {
                            float wa[MT];
#pragma unroll
                            for (int mt = 0; mt < MT; ++mt) wa[mt] = wp[((ky * 3 + kx) * MT + mt) * 64];
#pragma unroll
                            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                                for (int j = 0; j < 4; ++j)
                                    acc[mt][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[mt], t[ky][j + kx], acc[mt][j], 0, 0, 0);
                        }
                    self(self, std::integral_constant<int, c + 1>{});
                }
            };
            kstep(kstep, std::integral_constant<int, 0>{});
        }
        stream_epilogue<MT>(acc, ep, s_ep, out, (((long long)b * CO * H + y) * W) + 4 * pc, (long long)H * W, p, kq, st1, st2, nullptr, 0,
                            (unsigned)((long long)B * CO * H * W * 4));
    }
    if (ep.stats) {
        // eight waves, four rows of s_red: waves w and w + 4 add up through two rounds
        const int co = 16 * (p >> 2) + 4 * kq + (p & 3);
        if (wave < 4) { s_red[(wave * CO + co) * 2] = st1; s_red[(wave * CO + co) * 2 + 1] = st2; }
        __syncthreads();
        if (wave >= 4) { s_red[((wave - 4) * CO + co) * 2] += st1; s_red[((wave - 4) * CO + co) * 2 + 1] += st2; }
        __syncthreads();
        for (int i = tid; i < CO * 2; i += 512) {
            double sum = 0.0;
            for (int w = 0; w < 4; ++w) sum += s_red[(w * CO + (i >> 1)) * 2 + (i & 1)];
            ep.stats[((long long)blockIdx.x * CO + (i >> 1)) * 2 + (i & 1)] = sum;
        }
        for (int sl = blockIdx.x + gridDim.x; sl < nslabs; sl += gridDim.x)
            for (int i = tid; i < CO * 2; i += 512) ep.stats[((long long)sl * CO + (i >> 1)) * 2 + (i & 1)] = 0.0;
    }
}

// ------------------------------------------- backward of the wide residual blocks' 1x1 convolution (64 -> 64), both gradients
// dm_conv1x1_bwd_fused at 64 channels.  As two launches (data gradient + weight gradient, above) the pair read dy, its
// BatchNorm-backward partner and the layer input twice from HBM (1.4 GB at B = 768, 209 + 146 us); here a wave takes 64 pixels
// through BOTH products before it moves on (237-275 us for both; by the counters the second reading still comes from HBM more
// often than from L2 -- 2 048 waves x 48 KB of unit in flight exceed it -- the gain is one launch and one set of prologues):
//     part 1   dx[ci][px] = (a0 x + a2 > 0) * sum_co W[co][ci] * da[co][px]       (M = ci from W^T in LDS, N = pixels, K = co)
//     part 2   dW[co][ci] += sum_px da[co][px] * relu(a0 x + a2)[ci][px]           (M = co, N = ci, K = pixels)
// with da = c0 dy + c1 y + c2 (BatchNorm backward folded into the load).  Part 1 reads da as the B operand (lane = pixel quad
// of channel 4 ks + kq), part 2 as the A operand (lane = four pixels of channel 16 mt + p): the same bytes in the other of the
// two layouts wide_stream.hip's header describes.  Every wave is on its own (no barrier after the tables are in) and writes its
// own statistics and weight slab.
template <bool IN2>
__global__ __launch_bounds__(256, 2) void conv1x1_bwd_wide_stream_kernel(Operand dy, const float *__restrict__ x, const float *__restrict__ xcoef,
                                                                         const float *__restrict__ w, float *__restrict__ dx,
                                                                         double *__restrict__ stats, float *__restrict__ wslabs,
                                                                         int B, int HW, int nslabs)
{
    constexpr int C = 64, MT = 4;
    __shared__ __attribute__((aligned(16))) float s_wt[16 * MT * 64];    // [ks][mt][co % 4][ci % 16] = w[co = 4 ks + ..][ci = 16 mt + ..]
    __shared__ __attribute__((aligned(16))) float s_cd[C * 4];           // (c0, c1, c2, -) of output-gradient channel co
    __shared__ __attribute__((aligned(16))) float s_cx[C * 4];           // (a0, -, a2, -) of layer-input channel ci
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, p = lane & 15, kq = lane >> 4;
    for (int i = tid; i < 16 * MT * 64; i += 256) {
        const int m = i & 15, c = (i >> 4) & 3, mt = (i >> 6) & 3, ks = i >> 8;
        s_wt[i] = w[(4 * ks + c) * C + 16 * mt + m];
    }
    for (int i = tid; i < C; i += 256) {
        const StreamCoef c = stream_coef(dy, i);
        s_cd[i * 4] = c.c0; s_cd[i * 4 + 1] = c.c1; s_cd[i * 4 + 2] = c.c2; s_cd[i * 4 + 3] = 0.f;
        s_cx[i * 4] = xcoef[i * 4]; s_cx[i * 4 + 1] = 0.f; s_cx[i * 4 + 2] = xcoef[i * 4 + 2]; s_cx[i * 4 + 3] = 0.f;
    }
    __syncthreads();

    const int upp = HW >> 6;
    const long long total = (long long)B * upp;
    const int gw = blockIdx.x * 4 + wave, nw = gridDim.x * 4;
    const int u0 = (int)(total * gw / nw), u1 = (int)(total * (gw + 1) / nw);
    const unsigned bytes = (unsigned)((long long)B * C * HW * 4);
    const __amdgpu_buffer_rsrc_t rD = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(dy.p0), 0, bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rY = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(IN2 ? dy.p1 : dy.p0), 0, bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(x), 0, bytes, 0x00020000);
    const int vo1 = (kq * HW + 4 * p) * 4;                       // part 1: channel 4 ks + kq, pixels 4 p ..
    const int voe = (4 * kq * HW + 4 * p) * 4;                   // its epilogue: channel 16 mt + 4 kq + i, pixels 4 p ..
    const int vo2 = (p * HW + 4 * kq) * 4;                       // part 2: channel 16 t + p, pixels 16 g + 4 kq ..

    f32x4 aw[MT][MT];                                            // dW[co = 16 mt + 4 kq + i][ci = 16 nt + p], over all of the wave's units
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b2 = 0; b2 < MT; ++b2) aw[a][b2] = (f32x4){0.f, 0.f, 0.f, 0.f};
    double st1 = 0.0, st2 = 0.0;                                 // lane p of row kq: channel 16 (p >> 2) + 4 kq + (p & 3)

    for (int u = u0; u < u1; ++u) {
        const int b = u / upp, r = u - b * upp;
        const unsigned ub = (unsigned)(((long long)b * C * HW + r * 64) * 4);
        // ---- part 1: the data gradient
        f32x4 ax[MT][4];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int j = 0; j < 4; ++j) ax[mt][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        {
            constexpr int R = 2;
            f32x4 rd[R], ry[R];
            auto issue = [&](auto slot, int ks) {
                constexpr int sl = decltype(slot)::value;
                rd[sl] = __builtin_amdgcn_raw_buffer_load_b128(rD, vo1, ub + ks * (4 * HW * 4), 0);
                if constexpr (IN2) ry[sl] = __builtin_amdgcn_raw_buffer_load_b128(rY, vo1, ub + ks * (4 * HW * 4), 0);
                __builtin_amdgcn_sched_barrier(0);
            };
            issue(std::integral_constant<int, 0>{}, 0); issue(std::integral_constant<int, 1>{}, 1);
            auto kloop = [&](auto self, auto kc) -> void {
                constexpr int ks = decltype(kc)::value;
                if constexpr (ks < 16) {
                    constexpr int sl = ks % R;
                    const f32x4 cf = *reinterpret_cast<const f32x4 *>(&s_cd[(4 * ks + kq) * 4]);
                    f32x4 v;
                    if constexpr (IN2) v = cf.x * rd[sl] + (cf.y * ry[sl] + cf.z);
                    else v = cf.x * rd[sl] + cf.z;
                    float wa[MT];
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) wa[mt] = s_wt[(ks * MT + mt) * 64 + kq * 16 + p];
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                        for (int j = 0; j < 4; ++j) ax[mt][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[mt], v[j], ax[mt][j], 0, 0, 0);
                    if constexpr (ks + R < 16) issue(std::integral_constant<int, sl>{}, ks + R);
                    self(self, std::integral_constant<int, ks + 1>{});
                }
            };
            kloop(kloop, std::integral_constant<int, 0>{});
        }
        // ---- part 2: the weight gradient, 16 pixels per step (the unit's bytes again); the loads of step g + 1
        //      are in flight while step g multiplies
        {
            f32x4 ld[2][MT], ly[IN2 ? 2 : 1][MT], lx[2][MT];
            auto req = [&](auto sc, int g) {
                constexpr int sl = decltype(sc)::value;
#pragma unroll
                for (int t = 0; t < MT; ++t) {
                    const unsigned so = ub + (unsigned)((16 * t) * HW * 4 + g * 64);
                    ld[sl][t] = __builtin_amdgcn_raw_buffer_load_b128(rD, vo2, so, 0);
                    if constexpr (IN2) ly[sl][t] = __builtin_amdgcn_raw_buffer_load_b128(rY, vo2, so, 0);
                    lx[sl][t] = __builtin_amdgcn_raw_buffer_load_b128(rX, vo2, so, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            };
            auto mul = [&](auto sc) {
                constexpr int sl = decltype(sc)::value;
                f32x4 da[MT], tx[MT];
#pragma unroll
                for (int t = 0; t < MT; ++t) {
                    const f32x4 cd = *reinterpret_cast<const f32x4 *>(&s_cd[(16 * t + p) * 4]);
                    const f32x4 cx = *reinterpret_cast<const f32x4 *>(&s_cx[(16 * t + p) * 4]);
                    if constexpr (IN2) da[t] = cd.x * ld[sl][t] + (cd.y * ly[sl][t] + cd.z);
                    else da[t] = cd.x * ld[sl][t] + cd.z;
                    tx[t] = dm_relu4(cx.x * lx[sl][t] + cx.z);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int a = 0; a < MT; ++a)
#pragma unroll
                        for (int b2 = 0; b2 < MT; ++b2) aw[a][b2] = __builtin_amdgcn_mfma_f32_16x16x4f32(da[a][j], tx[b2][j], aw[a][b2], 0, 0, 0);
            };
            using I0 = std::integral_constant<int, 0>;
            using I1 = std::integral_constant<int, 1>;
        // its epilogue: (mt, i) -> channel ci = 16 mt + 4 kq + i, pixels 4 p ..: gate by the layer input, store, statistics
            {
                float sel1 = 0.f, sel2 = 0.f;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    f32x4 xr[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) xr[i] = __builtin_amdgcn_raw_buffer_load_b128(rX, voe, ub + (unsigned)((16 * mt + i) * HW * 4), 0);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int ci = 16 * mt + 4 * kq + i;
                        const float a0 = s_cx[ci * 4], a2 = s_cx[ci * 4 + 2];
                        const f32x4 q = xr[i];
                        f32x4 v = (f32x4){ax[mt][0][i], ax[mt][1][i], ax[mt][2][i], ax[mt][3][i]};
                        v.x = (a0 * q.x + a2) > 0.f ? v.x : 0.f; v.y = (a0 * q.y + a2) > 0.f ? v.y : 0.f;
                        v.z = (a0 * q.z + a2) > 0.f ? v.z : 0.f; v.w = (a0 * q.w + a2) > 0.f ? v.w : 0.f;
                        *reinterpret_cast<f32x4 *>(dx + ((long long)b * C + ci) * HW + r * 64 + 4 * p) = v;
                        float sa = (v.x + v.y) + (v.z + v.w), sc = (v.x * q.x + v.y * q.y) + (v.z * q.z + v.w * q.w);
                        sa += dpp_mov<0xB1>(sa); sc += dpp_mov<0xB1>(sc);
                        sa += dpp_mov<0x4E>(sa); sc += dpp_mov<0x4E>(sc);
                        sa += dpp_mov<0x141>(sa); sc += dpp_mov<0x141>(sc);
                        sa += dpp_mov<0x140>(sa); sc += dpp_mov<0x140>(sc);
                        sel1 = p == 4 * mt + i ? sa : sel1;
                        sel2 = p == 4 * mt + i ? sc : sel2;
                    }
                }
                st1 += (double)sel1; st2 += (double)sel2;
            }
            req(I0{}, 0);
            req(I1{}, 1); mul(I0{});
            req(I0{}, 2); mul(I1{});
            req(I1{}, 3); mul(I0{});
            mul(I1{});
        }
    }
    // the wave's slabs: statistics (sum dx, sum dx * x) per channel and dW[co][ci]
    {
        const int ci = 16 * (p >> 2) + 4 * kq + (p & 3);
        stats[((long long)gw * C + ci) * 2] = st1; stats[((long long)gw * C + ci) * 2 + 1] = st2;
        float *row = wslabs + (long long)gw * C * C;
#pragma unroll
        for (int a = 0; a < MT; ++a)
#pragma unroll
            for (int b2 = 0; b2 < MT; ++b2)
#pragma unroll
                for (int i = 0; i < 4; ++i) row[(16 * a + 4 * kq + i) * C + 16 * b2 + p] = aw[a][b2][i];
    }
    for (int sl = nw + gw; sl < nslabs; sl += nw) {
        for (int i = lane; i < C * 2; i += 64) stats[(long long)sl * C * 2 + i] = 0.0;
        for (int i = lane; i < C * C; i += 64) wslabs[(long long)sl * C * C + i] = 0.f;
    }
}

int stream_switch()
{
    static const int v = getenv("DM_WIDE_STREAM") ? atoi(getenv("DM_WIDE_STREAM")) : 0x3ff;   // bit 0: 1x1 weight gradient, 1: 1x1 convolution, 2: thin 4x4/s2 weight gradient, 3: thin 4x4/s2 convolution, 4: thin transposed convolution, 5: 4x4/s2 convolution 32 -> 64 with the weights resident in LDS, 6: its transposed mirror 64 -> 32, 7 / 8: 3x3 64 -> 64 with the weights in LDS (forward form / forms with a gate), 9: both gradients of the 1x1 64 -> 64 in one launch
    return v;
}
int stream_depth()
{
    static const int v = getenv("DM_WIDE_STREAM_DEPTH") ? atoi(getenv("DM_WIDE_STREAM_DEPTH")) : 2;
    return v;
}

}  // namespace

// ---- entry points used by conv_wide.hip's dispatchers (not part of the public header) ----------------------------------
// shapes the 1x1 streaming weight gradient is built for (operands are checked at the launch: dm_stream_wgrad1x1 returns
// false and the caller falls back to the tiled kernel, which accepts any slab count)
bool dm_stream_wgrad1x1_shape(int B, int CS, int CT, int Hs, int Ws)
{
    if (!(stream_switch() & 1)) return false;
    const long long HW = (long long)Hs * Ws;
    return (CS == 64 || CS == 32) && (CT == 64 || CT == 32) && HW % 32 == 0 && (long long)B * 64 * HW * 4 < (1LL << 32);
}

int dm_stream_wgrad1x1_slabs(int B, int Hs, int Ws)
{
    const long long groups = (long long)B * Hs * Ws / 32;
    return (int)(groups < 768 ? groups : 768);                 // three workgroups per CU, one contiguous range each
}

bool dm_stream_wgrad1x1(const Operand &S, const Operand &T, float *slabs, int B, int CS, int CT, int Hs, int Ws, int nslabs,
                        hipStream_t st)
{
    if (!dm_stream_wgrad1x1_shape(B, CS, CT, Hs, Ws) || T.ones || S.ones || T.mode == DM_LOAD_AFFINE2) return false;
    if ((S.mode >= DM_LOAD_AFFINE && S.coef_bstride) || (T.mode >= DM_LOAD_AFFINE && T.coef_bstride)) return false;
    int grid = dm_stream_wgrad1x1_slabs(B, Hs, Ws);
    if (grid > nslabs) grid = nslabs;
    const int HW = Hs * Ws;
    const bool two = S.mode == DM_LOAD_AFFINE2, deep = stream_depth() >= 2;
#define DM_SW(MT_, NT_)                                                                                                  \
    if (CS == 16 * MT_ && CT == 16 * NT_) {                                                                              \
        if (two && deep) hipLaunchKernelGGL((wgrad1x1_stream_kernel<MT_, NT_, true, 2>), dim3(grid), dim3(256), 0, st, S, T, slabs, B, CS, CT, HW, nslabs); \
        else if (two) hipLaunchKernelGGL((wgrad1x1_stream_kernel<MT_, NT_, true, 1>), dim3(grid), dim3(256), 0, st, S, T, slabs, B, CS, CT, HW, nslabs); \
        else if (deep) hipLaunchKernelGGL((wgrad1x1_stream_kernel<MT_, NT_, false, 2>), dim3(grid), dim3(256), 0, st, S, T, slabs, B, CS, CT, HW, nslabs); \
        else hipLaunchKernelGGL((wgrad1x1_stream_kernel<MT_, NT_, false, 1>), dim3(grid), dim3(256), 0, st, S, T, slabs, B, CS, CT, HW, nslabs); \
    }
    DM_SW(4, 4) DM_SW(4, 2) DM_SW(2, 4) DM_SW(2, 2)
#undef DM_SW
    return true;
}

// 1x1 convolution, 64 -> 64 / 32 -> 32 ... channels (multiples of 16 up to 64 on both sides), shared coefficients, statistics
// per workgroup.  Returns false when the call is not one the streaming kernel takes (the caller runs the tiled kernel).
bool dm_stream_conv1x1(const Operand &in, const WeightView &wv, float *out, const Epilogue &ep, int B, int Cphys, int CIN,
                       int NOUT, int H, int W, int nslabs, int per_tile, hipStream_t st)
{
    if (!(stream_switch() & 2) || per_tile || Cphys != CIN || in.ones || ep.bias_border) return false;
    const long long HW = (long long)H * W;
    if (!((CIN == 64 || CIN == 32) && (NOUT == 64 || NOUT == 32)) || HW % 64 || (long long)B * 64 * HW * 4 >= (1LL << 32)) return false;
    if (in.mode >= DM_LOAD_AFFINE && in.coef_bstride) return false;
    if (ep.mask.p0 && (ep.mask.mode == DM_LOAD_RELU || ep.mask.mode > DM_LOAD_AFFINE || ep.mask.coef_bstride || ep.mask.ones)) return false;
    long long units = (long long)B * HW / 64;
    int grid = (int)(units / 4 < 512 ? (units + 3) / 4 : 512);
    if (ep.stats && grid > nslabs) grid = nslabs;
    if (grid < 1) grid = 1;
    const bool two = in.mode == DM_LOAD_AFFINE2;
#define DM_SC(K4_, MT_)                                                                                                      \
    if (CIN == 16 * K4_ && NOUT == 16 * MT_) {                                                                                \
        if (two) hipLaunchKernelGGL((conv1x1_stream_kernel<K4_, MT_, true, 4>), dim3(grid), dim3(256), 0, st, in, wv, out, ep, B, (int)HW, nslabs); \
        else hipLaunchKernelGGL((conv1x1_stream_kernel<K4_, MT_, false, 8>), dim3(grid), dim3(256), 0, st, in, wv, out, ep, B, (int)HW, nslabs); \
    }
    DM_SC(4, 4) DM_SC(4, 2) DM_SC(2, 4) DM_SC(2, 2)
#undef DM_SC
    return true;
}

// weight gradient of a 4x4 / stride 2 layer with 16 / 32 / 64 S channels and 1-4 T channels
bool dm_stream_wgrad_s2_thin_shape(int B, int CS, int CT, int Hs, int Ws)
{
    if (!(stream_switch() & 4)) return false;
    return (CS == 16 || CS == 32 || CS == 64) && CT >= 1 && CT <= (CS == 64 ? 1 : 4) && Ws % 32 == 0 &&
           (long long)B * CS * Hs * Ws * 4 < (1LL << 31) && (long long)B * CT * Hs * Ws * 16 < (1LL << 31);
}

int dm_stream_wgrad_s2_thin_slabs(int B, int Hs, int Ws)
{
    const long long stages = (long long)B * Hs * (Ws / 32);
    return (int)(stages / 4 < 768 ? (stages + 3) / 4 : 768);
}

bool dm_stream_wgrad_s2_thin(const Operand &S, const Operand &T, float *slabs, int B, int CS, int CT, int Hs, int Ws, int nslabs,
                             hipStream_t st)
{
    if (!dm_stream_wgrad_s2_thin_shape(B, CS, CT, Hs, Ws) || T.ones || S.ones || T.mode == DM_LOAD_AFFINE2) return false;
    if ((S.mode >= DM_LOAD_AFFINE && S.coef_bstride) || (T.mode >= DM_LOAD_AFFINE && T.coef_bstride)) return false;
    int grid = dm_stream_wgrad_s2_thin_slabs(B, Hs, Ws);
    if (grid > nslabs) grid = nslabs;
    const bool two = S.mode == DM_LOAD_AFFINE2;
#define DM_ST(MT_, NT_)                                                                                                      \
    if (CS == 16 * MT_ && CT == NT_) {                                                                                        \
        if (two) hipLaunchKernelGGL((wgrad_s2_thin_stream_kernel<MT_, NT_, true, 1>), dim3(grid), dim3(256), 0, st, S, T, slabs, B, CS, Hs, Ws, nslabs); \
        else hipLaunchKernelGGL((wgrad_s2_thin_stream_kernel<MT_, NT_, false, 1>), dim3(grid), dim3(256), 0, st, S, T, slabs, B, CS, Hs, Ws, nslabs); \
    }
    DM_ST(1, 1) DM_ST(1, 2) DM_ST(1, 3) DM_ST(1, 4) DM_ST(2, 1) DM_ST(2, 2) DM_ST(2, 3) DM_ST(2, 4) DM_ST(4, 1)
#undef DM_ST
    return true;
}

// 4x4 / stride 2 convolution, 1-4 input channels -> 16 / 32 / 64 output channels, 128-column input
bool dm_stream_conv_s2_thin(const Operand &in, const WeightView &wv, float *out, const Epilogue &ep, int B, int Cphys, int CIN,
                            int NOUT, int H, int W, int nslabs, int per_tile, hipStream_t st)
{
    if (!(stream_switch() & 8) || per_tile || Cphys != CIN || in.ones || ep.bias_border || in.mode == DM_LOAD_AFFINE2) return false;
    if (CIN < 1 || CIN > 4 || !(NOUT == 16 || NOUT == 32 || (NOUT == 64 && CIN <= 2)) || W != 128 || H % 2) return false;
    if ((long long)B * NOUT * (H / 2) * (W / 2) * 4 >= (1LL << 32) || (long long)B * CIN * H * W * 4 >= (1LL << 31)) return false;
    if (in.mode >= DM_LOAD_AFFINE && in.coef_bstride) return false;
    if (ep.mask.p0 && (ep.mask.mode == DM_LOAD_RELU || ep.mask.mode > DM_LOAD_AFFINE || ep.mask.coef_bstride || ep.mask.ones)) return false;
    const long long units = (long long)B * (H / 2);
    int grid = (int)(units / 4 < 768 ? (units + 3) / 4 : 768);
    if (ep.stats && grid > nslabs) grid = nslabs;
    if (grid < 1) grid = 1;
#define DM_CT(C_, MT_)                                                                                                       \
    if (CIN == C_ && NOUT == 16 * MT_)                                                                                        \
        hipLaunchKernelGGL((conv_s2_thin_stream_kernel<C_, MT_>), dim3(grid), dim3(256), 0, st, in, wv, out, ep, B, H, nslabs);
    DM_CT(1, 1) DM_CT(2, 1) DM_CT(3, 1) DM_CT(4, 1) DM_CT(1, 2) DM_CT(2, 2) DM_CT(3, 2) DM_CT(4, 2) DM_CT(1, 4) DM_CT(2, 4)
#undef DM_CT
    return true;
}

// ConvTranspose2d(4, 2, 1) from up to 64 channels on a 64-column grid to 1 or 2 channels, no gate / residual / statistics
bool dm_stream_convT_thin(const Operand &in, const WeightView &wv, float *scratch, float *out, const Epilogue &ep, int B, int Cphys,
                          int CIN, int NOUT, int H, int W, int per_tile, hipStream_t st)
{
    if (!(stream_switch() & 16) || per_tile || Cphys != CIN || in.ones || in.mode == DM_LOAD_AFFINE2) return false;
    if (!(NOUT == 4 || NOUT == 8) || W != 64 || H % 4 || CIN < 1 || CIN > 64 || !scratch) return false;
    if (ep.mask.p0 || ep.resid || ep.stats || ep.bias_border) return false;
    if ((in.mode >= DM_LOAD_AFFINE && in.coef_bstride) || (long long)B * CIN * H * W * 4 >= (1LL << 31)) return false;
    hipLaunchKernelGGL(convT_thin_pack_kernel, dim3((CIN * 32 + 255) / 256), dim3(256), 0, st, wv, scratch, CIN, NOUT / 4);
    const long long units = (long long)B * (H / 4);
    const int grid = (int)(units / 4 < 1024 ? (units + 3) / 4 : 1024);
    if (NOUT == 8) hipLaunchKernelGGL((convT_thin_stream_kernel<2>), dim3(grid), dim3(256), 0, st, in, (const float *)scratch, out, ep, B, CIN, H);
    else hipLaunchKernelGGL((convT_thin_stream_kernel<1>), dim3(grid), dim3(256), 0, st, in, (const float *)scratch, out, ep, B, CIN, H);
    return true;
}

// 4x4 / stride 2 convolution 32 -> 64 channels on a 64 x 64 input (the wide encoder's second convolution, the data gradient
// of the decoder's first transposed convolution)
bool dm_stream_conv_s2_wide(const Operand &in, const WeightView &wv, float *out, const Epilogue &ep, int B, int Cphys, int CIN,
                            int NOUT, int H, int W, int nslabs, int per_tile, hipStream_t st)
{
    if (!(stream_switch() & 32) || per_tile || Cphys != CIN || in.ones || ep.bias_border) return false;
    if (CIN != 32 || NOUT != 64 || H != 64 || W != 64 || (long long)B * 64 * 32 * 32 * 4 >= (1LL << 31)) return false;
    if (in.mode >= DM_LOAD_AFFINE && in.coef_bstride) return false;
    if (ep.mask.p0 && (ep.mask.mode == DM_LOAD_RELU || ep.mask.mode > DM_LOAD_AFFINE || ep.mask.coef_bstride || ep.mask.ones)) return false;
    const int lds = 128 * 1024;
    // (per launch, as dm_vq_backward does: the attribute belongs to the current device's copy of the kernel)
    if (hipFuncSetAttribute((const void *)conv_s2_wide_stream_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess ||
            hipFuncSetAttribute((const void *)conv_s2_wide_stream_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess ||
            hipFuncSetAttribute((const void *)conv_s2_wide_stream_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
        return false;
    const long long units = (long long)B * 16;
    int grid = (int)(units / 8 < 256 ? (units + 7) / 8 : 256);
    if (ep.stats && grid > nslabs) grid = nslabs;
    if (grid < 1) grid = 1;
    if (in.mode == DM_LOAD_AFFINE2)
        hipLaunchKernelGGL((conv_s2_wide_stream_kernel<true, true>), dim3(grid), dim3(512), lds, st, in, wv, out, ep, B, nslabs);
    else if (in.mode >= DM_LOAD_AFFINE || in.mode == DM_LOAD_RELU)
        hipLaunchKernelGGL((conv_s2_wide_stream_kernel<true>), dim3(grid), dim3(512), lds, st, in, wv, out, ep, B, nslabs);
    else
        hipLaunchKernelGGL((conv_s2_wide_stream_kernel<false>), dim3(grid), dim3(512), lds, st, in, wv, out, ep, B, nslabs);
    return true;
}

// ConvTranspose2d(64 -> 32, 4, 2, 1) on a 32 x 32 input (the wide decoder's first transposed convolution, the data gradient of
// the encoder's second convolution)
bool dm_stream_convT_wide(const Operand &in, const WeightView &wv, float *out, const Epilogue &ep, int B, int Cphys, int CIN,
                          int NOUT, int H, int W, int nslabs, int per_tile, hipStream_t st)
{
    if (!(stream_switch() & 64) || per_tile || Cphys != CIN || in.ones || ep.bias_border) return false;
    if (CIN != 64 || NOUT != 128 || H != 32 || W != 32 || (long long)B * 64 * 32 * 32 * 4 >= (1LL << 31)) return false;
    if (in.mode >= DM_LOAD_AFFINE && in.coef_bstride) return false;
    if (ep.mask.p0 && (ep.mask.mode == DM_LOAD_RELU || ep.mask.mode > DM_LOAD_AFFINE || ep.mask.coef_bstride || ep.mask.ones)) return false;
    const int lds = 128 * 1024;
    // (per launch, as dm_vq_backward does: the attribute belongs to the current device's copy of the kernel)
    if (hipFuncSetAttribute((const void *)convT_wide_stream_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess ||
            hipFuncSetAttribute((const void *)convT_wide_stream_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
        return false;
    const long long units = (long long)B * 16;
    int grid = (int)(units / 8 < 256 ? (units + 7) / 8 : 256);
    if (ep.stats && grid > nslabs) grid = nslabs;
    if (grid < 1) grid = 1;
    if (in.mode == DM_LOAD_AFFINE2)
        hipLaunchKernelGGL((convT_wide_stream_kernel<true>), dim3(grid), dim3(512), lds, st, in, wv, out, ep, B, nslabs);
    else
        hipLaunchKernelGGL((convT_wide_stream_kernel<false>), dim3(grid), dim3(512), lds, st, in, wv, out, ep, B, nslabs);
    return true;
}

// 3x3 convolution 64 -> 64 channels on a 32 x 32 grid (the wide residual blocks; forward and data-gradient form)
bool dm_stream_conv3x3_wide(const Operand &in, const WeightView &wv, float *out, const Epilogue &ep, int B, int Cphys, int CIN,
                            int NOUT, int H, int W, int nslabs, int per_tile, hipStream_t st)
{
    const int bit = ep.mask.p0 ? 256 : 128;
    if (!(stream_switch() & bit) || per_tile || Cphys != CIN || in.ones || ep.bias_border) return false;
    if (CIN != 64 || NOUT != 64 || H != 32 || W != 32 || (long long)B * 64 * 32 * 32 * 4 >= (1LL << 31)) return false;
    if (in.mode >= DM_LOAD_AFFINE && in.coef_bstride) return false;
    if (ep.mask.p0 && (ep.mask.mode == DM_LOAD_RELU || ep.mask.mode > DM_LOAD_AFFINE || ep.mask.coef_bstride || ep.mask.ones)) return false;
    const int lds = 9 * 64 * 64 * 4;
    // (per launch, as dm_vq_backward does: the attribute belongs to the current device's copy of the kernel)
    if (hipFuncSetAttribute((const void *)conv3x3_wide_stream_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess ||
            hipFuncSetAttribute((const void *)conv3x3_wide_stream_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
        return false;
    const long long units = (long long)B * 16;
    int grid = (int)(units / 8 < 256 ? (units + 7) / 8 : 256);
    if (ep.stats && grid > nslabs) grid = nslabs;
    if (grid < 1) grid = 1;
    if (in.mode == DM_LOAD_AFFINE2)
        hipLaunchKernelGGL((conv3x3_wide_stream_kernel<true>), dim3(grid), dim3(512), lds, st, in, wv, out, ep, B, nslabs);
    else
        hipLaunchKernelGGL((conv3x3_wide_stream_kernel<false>), dim3(grid), dim3(512), lds, st, in, wv, out, ep, B, nslabs);
    return true;
}

// dm_conv1x1_bwd_fused at 64 -> 64 channels: slabs = waves (every wave writes its own statistics and weight slab)
bool dm_stream_conv1x1_bwd_shape(int CD, int CX, int H, int W)
{
    return (stream_switch() & 512) && CD == 64 && CX == 64 && H > 0 && W > 0 && ((long long)H * W) % 64 == 0;
}

int dm_stream_conv1x1_bwd_slabs(int B, int H, int W)
{
    const long long units = (long long)B * H * W / 64;
    const long long wg = units / 4 < 512 ? (units + 3) / 4 : 512;
    return (int)(wg * 4);
}

int dm_stream_conv1x1_bwd(const Operand &dy, const float *x, const float *xcoef, const float *w, float *dx, double *stats,
                          float *wslabs, int B, int H, int W, hipStream_t st)
{
    const int nslabs = dm_stream_conv1x1_bwd_slabs(B, H, W);
    if (dy.mode == DM_LOAD_AFFINE2)
        hipLaunchKernelGGL((conv1x1_bwd_wide_stream_kernel<true>), dim3(nslabs / 4), dim3(256), 0, st, dy, x, xcoef, w, dx, stats, wslabs, B, H * W, nslabs);
    else
        hipLaunchKernelGGL((conv1x1_bwd_wide_stream_kernel<false>), dim3(nslabs / 4), dim3(256), 0, st, dy, x, xcoef, w, dx, stats, wslabs, B, H * W, nslabs);
    return 0;
}

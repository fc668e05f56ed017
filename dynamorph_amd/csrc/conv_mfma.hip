// conv_mfma.hip -- implicit-GEMM convolutions on v_mfma_f32_16x16x4_f32 (exact fp32 FMA chains).
//
// Replaces aten::convolution / aten::conv_transpose2d / aten::convolution_backward(data)
// for every layer of VQ_VAE.enc / VQ_VAE.dec (HiddenStateExtractor/vq_vae.py:276-298, the
// ResidualBlock convs at :203-209).
//
// GEMM view per workgroup: M = 16 consecutive output pixels of one row (one MFMA tile),
// N = output channels (16 per tile), K = input channels x taps.  The input tile (with its
// halo, after the on-load operand transform: BatchNorm apply / ReLU / BatchNorm backward)
// is staged in LDS; weights for the wave's N tiles live in registers for the whole
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
// Workgroups are persistent: weights, bias and mask coefficients are loaded once, then the workgroup
// walks tiles blockIdx, blockIdx + gridDim, ...; the global loads of tile i+1 are issued before the MFMAs
// and stores of tile i (register staging, tile.h), and inside a tile the LDS operands of K-chunk c+1 are
// requested before the MFMAs of chunk c, so HBM latency, LDS latency, the matrix pipe and the store
// stream overlap instead of adding up.
//
// Epilogue (dm_epilogue): bias, ReLU, ReLU-backward mask, residual add, store, and per-channel
// partial sums (sum v, sum v*q) in double.  One statistics slab per WORKGROUP: dm_conv*_num_blocks tells the
// caller how many (the persistent grid, or one per tile when stats_per_tile asks for per-sample grouping);
// everything is reduced deterministically in bn.hip.
#include "dm_common.h"
#include "tile.h"
#include "mfma_util.h"

// fallbacks for shapes / channel families without an MFMA instantiation (conv_generic.hip)
int dm_generic_conv_slabs(int B, int per_tile);
int dm_generic_conv(int form, const Operand &in, const WeightView &wv, float *out, const Epilogue &ep, int B, int Cphys,
                    int CIN, int NOUT, int H, int W, int taps, int nslabs, int per_tile, hipStream_t st);
// arbitrary channel counts on the MFMA, 8 x 16 tiles (conv_wide.hip); form 0: 4x4/s2, 1: 3x3 or 1x1, 2: transposed
bool dm_wide_conv_ok(int form, int H, int W);
int dm_wide_conv_slabs(int form, int B, int H, int W, int per_tile);
long long dm_wide_conv_scratch_floats(int form, int CIN, int NOUT, int taps);
int dm_wide_conv(int form, const Operand &in, const WeightView &wv, float *scratch, float *out, const Epilogue &ep, int B,
                 int Cphys, int CIN, int NOUT, int H, int W, int taps, int nslabs, int per_tile, hipStream_t st);

// arithmetic of the gradient kernels (defined with the C ABI at the end of this file; also used by wgrad_mfma.hip)
bool dm_backward_split_bf16();
bool dm_conv4x4s2_patch_forward(const Operand &in, const WeightView &wv, float *out, const Epilogue &ep, int B, int Cphys, int CIN,
                                int NOUT, int H, int W, int per_tile, int nslabs, hipStream_t stream, int *rc);
#ifdef DM_MEASURE
// Measurement builds only (make measure): DM_FORWARD_SPLIT=1 runs the FORWARD convolutions on the two-piece split-bf16
// products too.  Results are then not fp32 (latents pick codes): this exists to bound what ANY bf16-piece arithmetic could
// buy the convolution family and is compiled out of the shipped library.
static bool dm_forward_split() { static const bool v = getenv("DM_FORWARD_SPLIT") != nullptr; return v; }
#define DM_FWD_SPLIT(cond) ((cond) && dm_forward_split())
constexpr bool MEASURE_BF = true;
#else
#define DM_FWD_SPLIT(cond) false
constexpr bool MEASURE_BF = false;
#endif

namespace {

// ----------------------------------------------------------------------------- epilogue
// Side inputs of an output float4.  SIDE_NONE: none (forward convs).  SIDE_MASK: only the ReLU-backward
// mask tensor, which is also the second-moment partner when stat_q is given (every data gradient that
// feeds a BatchNorm backward).  SIDE_ALL: mask, residual and stat_q are three different tensors.
// They are loaded before the MFMA loop of their tile so the loads overlap it.
enum { SIDE_NONE = 0, SIDE_MASK = 1, SIDE_ALL = 2 };

template <int SIDE> struct EpiIn { f32x4 m; };
template <> struct EpiIn<SIDE_ALL> { f32x4 m, r, q; };

// Output and side tensors go through buffer descriptors rebased to the tile's sample, with NO lane predicate and no
// branch: a lane without an output channel (n >= NOUT) carries a byte offset beyond the descriptor, so its loads
// return 0 and its stores vanish; an absent tensor gets an empty descriptor.  With every memory operation issued
// unconditionally the compiler knows how many stores follow the next tile's loads and waits for the loads alone
// (s_waitcnt vmcnt(#stores)) instead of draining the stores of the finished tile before the next commit.
constexpr unsigned DM_RSRC_FLAGS = 0x00020000u;       // raw buffer, 32-bit data format (gfx9 family)
constexpr int DM_VOFF_NONE = 0x40000000;              // byte offset no sample reaches (checked on the host)

__device__ __forceinline__ __amdgpu_buffer_rsrc_t sample_rsrc(const float *base, long long sample_elems, int b)
{
    return base ? __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(base + sample_elems * b), 0,
                                                    (int)(sample_elems * 4), DM_RSRC_FLAGS)
                : __builtin_amdgcn_make_buffer_rsrc(nullptr, 0, 0, DM_RSRC_FLAGS);
}

template <int SIDE> struct EpiCtx {
    __amdgpu_buffer_rsrc_t out_r, m_r, r_r, q_r;
    __device__ __forceinline__ void rebase(const Epilogue &ep, float *out, long long sample_elems, int b)
    {
        out_r = sample_rsrc(out, sample_elems, b);
        if constexpr (SIDE != SIDE_NONE) m_r = sample_rsrc(ep.mask.p0, sample_elems, b);
        if constexpr (SIDE == SIDE_ALL) {
            r_r = sample_rsrc(ep.resid, sample_elems, b);
            q_r = sample_rsrc(ep.stat_q, sample_elems, b);
        }
    }
};

template <int SIDE>
__device__ __forceinline__ void epilogue_loads(EpiIn<SIDE> &e, const EpiCtx<SIDE> &cx, int voff)
{
    if constexpr (SIDE != SIDE_NONE) e.m = __builtin_amdgcn_raw_buffer_load_b128(cx.m_r, voff, 0, 0);
    if constexpr (SIDE == SIDE_ALL) {
        e.r = __builtin_amdgcn_raw_buffer_load_b128(cx.r_r, voff, 0, 0);
        e.q = __builtin_amdgcn_raw_buffer_load_b128(cx.q_r, voff, 0, 0);
    }
}

// mask coefficients (c0, c2) of the lane's output channel: keep v where c0*m + c2 > 0 (no mask tensor: 0*m + 1)
__device__ __forceinline__ void mask_coef(const Epilogue &ep, int b, int chan, float &c0, float &c2)
{
    c0 = ep.mask.p0 ? 1.f : 0.f; c2 = ep.mask.p0 ? 0.f : 1.f;
    if (ep.mask.p0 && ep.mask.mode >= DM_LOAD_AFFINE) {
        const float *cf = ep.mask.coef + (long long)b * ep.mask.coef_bstride + chan * 4;
        c0 = cf[0]; c2 = cf[2];
    }
}

// v: 4 consecutive output elements (along x) at element offset `off`.
template <int SIDE>
__device__ __forceinline__ void epilogue_tail(f32x4 v, const Epilogue &ep, const EpiCtx<SIDE> &cx, const EpiIn<SIDE> &e,
                                              float mc0, float mc2, int voff, double &s1, double &s2)
{
    f32x4 q = v;
    if constexpr (SIDE != SIDE_NONE) {
        const f32x4 mv = mc0 * e.m + mc2;
        v.x = mv.x > 0.f ? v.x : 0.f; v.y = mv.y > 0.f ? v.y : 0.f;
        v.z = mv.z > 0.f ? v.z : 0.f; v.w = mv.w > 0.f ? v.w : 0.f;
    }
    if constexpr (SIDE == SIDE_ALL) v += e.r;
    __builtin_amdgcn_raw_buffer_store_b128(v, cx.out_r, voff, 0, 0);
    if (ep.stats) {
        q = v;
        if constexpr (SIDE == SIDE_MASK) { if (ep.stat_q) q = e.m; }
        if constexpr (SIDE == SIDE_ALL) { if (ep.stat_q) q = e.q; }
        // four elements in fp32, then one promotion: fp64 converts/adds are half rate and this runs per tile
        s1 += (double)((v.x + v.y) + (v.z + v.w));
        s2 += (double)((v.x * q.x + v.y * q.y) + (v.z * q.z + v.w * q.w));
    }
}

__device__ __forceinline__ f32x4 bias_relu(f32x4 v, const Epilogue &ep, float bias)
{
    v += bias;
    if (ep.relu) v = dm_relu4(v);
    return v;
}

// Per-workgroup reduction of the per-lane channel partials -> stats[blockIdx][NCH][2].
// Lane layout: channel n = 16*t + (lane & 15) (PIX: channel = n >> 2); partials of the 4 lane
// quarters (lane >> 4) and, with PIX, of the 4 phase lanes are summed with shuffles.
template <int NTT, bool PIX>
__device__ __forceinline__ void stats_reduce(double (&s1)[NTT], double (&s2)[NTT], double (*s_stat)[2],
                                             const Epilogue &ep, int NCH, long long slab)
{
    // s_stat: [4 waves][NTT*16][2]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int t = 0; t < NTT; ++t) {
        double a = dm_row_sum_f64(s1[t]), c = dm_row_sum_f64(s2[t]);
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
        ep.stats[(slab * NCH + ch) * 2 + 0] = a;
        ep.stats[(slab * NCH + ch) * 2 + 1] = c;
    }
}

// slabs [gridDim, nslabs) exist in the caller's buffer but belong to no workgroup of this variant: zero them
__device__ __forceinline__ void zero_unowned_slabs(const Epilogue &ep, int NCH, int nslabs)
{
    for (int t2 = blockIdx.x + gridDim.x; t2 < nslabs; t2 += gridDim.x)
        for (int i = threadIdx.x; i < NCH * 2; i += DM_BLOCK) ep.stats[(long long)t2 * NCH * 2 + i] = 0.0;
}

// ============================================================================ kernel A
// BB: per-position bias from ep.bias_border[3][3][NOUT] (first / interior / last output row x column) instead of
// ep.bias -- the enc.0 bias seen through enc.1's zero padding, so the first conv needs no ones channel (K = 32, not 48).
// BF (gradients only: kernels with side inputs or the BatchNorm-backward operand): split-bf16 operands, tile.h
template <int CIN, int NT, int TH, int TW, int SIDE, int WPS, bool BB, bool BF = false>
__global__ __launch_bounds__(DM_BLOCK, WPS)
void conv4x4s2_kernel(Operand in, WeightView wv, float *__restrict__ out, Epilogue ep, int Cphys, int NOUT, int H,
                      int W, int ntiles, int nslabs, int per_tile)
{
    constexpr int IH = 2 * TH + 2, RS = 2 * TW + 8, COLS4 = RS / 4, PS = IH * RS;
    constexpr int KS = CIN * 4, CG = TW / 16, MT = TH * CG, MTW = MT / 4;
    constexpr int MP = NT == 1 ? 2 : 1;                    // M tiles in flight: two MFMA chains either way
    static_assert(MTW % MP == 0 && MTW >= MP, "M tiles per wave");
    __shared__ __attribute__((aligned(16))) float tile[CIN * PS];
    __shared__ __attribute__((aligned(16))) float s_coef[DM_COEF_MAX_C * 4];
    __shared__ double s_stat[4 * NT * 16][2];

    const int Ho = H >> 1, Wo = W >> 1;
    const int tiles_x = Wo / TW, tiles_y = Ho / TH;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, m = lane & 15, kq = lane >> 4;

    TileStage<CIN, IH, COLS4, RS, PS, false> stage;
    stage.init(H, W);
    int tidx = blockIdx.x, b = 0, oy0 = 0, ox0 = 0;
    if (tidx < ntiles) {
        int t = tidx;
        ox0 = (t % tiles_x) * TW; t /= tiles_x;
        oy0 = (t % tiles_y) * TH; b = t / tiles_y;
        stage.issue(in, b, Cphys, H, W, 2 * oy0 - 1, 2 * ox0 - 4);
        stage_coef(s_coef, in, b, Cphys);
    }

    static_assert(!BB || NT == 1, "bias_border: one N tile");
    float tb00 = 0.f, tb01 = 0.f, tb02 = 0.f, tb10 = 0.f, tb11 = 0.f, tb12 = 0.f, tb20 = 0.f, tb21 = 0.f, tb22 = 0.f;
    if constexpr (BB) {                                    // bias_border[row class][column class] of the lane's channel
        const float *tp = ep.bias_border + (m < NOUT ? m : 0);
        tb00 = tp[0 * NOUT]; tb01 = tp[1 * NOUT]; tb02 = tp[2 * NOUT];
        tb10 = tp[3 * NOUT]; tb11 = tp[4 * NOUT]; tb12 = tp[5 * NOUT];
        tb20 = tp[6 * NOUT]; tb21 = tp[7 * NOUT]; tb22 = tp[8 * NOUT];
    }
    float wreg[NT][KS], bias[NT], mc0[NT], mc2[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int n = 16 * t + m;
        const int nc = n < NOUT ? n : 0;
        bias[t] = ep.bias ? ep.bias[nc] : 0.f;
        mask_coef(ep, 0, nc, mc0[t], mc2[t]);
#pragma unroll
        for (int s = 0; s < KS; ++s)
            wreg[t][s] = n < NOUT ? wv.w[wv.off + n * wv.sn + (s >> 2) * wv.sc + (s & 3) * wv.sky + kq * wv.skx] : 0.f;
        if constexpr (BF) {
#pragma unroll
            for (int s = 0; s < KS; ++s) wreg[t][s] = split_pack1(wreg[t][s]);
        }
    }

    double s1[NT], s2[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) { s1[t] = 0.0; s2[t] = 0.0; }
    const int abase = 2 * m + kq + 3;
    auto off = [](int s) { return (s >> 2) * PS + (s & 3) * RS; };
    EpiCtx<SIDE> cx;
    const long long sample_elems = (long long)NOUT * Ho * Wo;
    int chan_off[NT];                                      // byte offset of the lane's channel plane (+ its x quad)
#pragma unroll
    for (int t = 0; t < NT; ++t) chan_off[t] = 16 * t + m < NOUT ? ((16 * t + m) * Ho * Wo + 4 * kq) * 4 : DM_VOFF_NONE;
    while (tidx < ntiles) {
        __syncthreads();                                   // previous tile consumed; coefficient table visible
        stage.template commit<BF>(tile, s_coef, Cphys, H, W, 2 * oy0 - 1, 2 * ox0 - 4, in.mode);
        __syncthreads();
        const int cb = b, cy0 = oy0, cx0 = ox0;            // the tile now in LDS
        const int next = tidx + gridDim.x;
        {                                                  // (no next tile: empty descriptor, every load returns 0)
            int t = next < ntiles ? next : tidx;
            ox0 = (t % tiles_x) * TW; t /= tiles_x;
            oy0 = (t % tiles_y) * TH; b = t / tiles_y;
        }
        // the next tile's loads are requested element by element between the M tiles below (tile.h: issue_one)
        const auto scx = stage.begin(in, next < ntiles, b, Cphys, H, W, 2 * oy0 - 1, 2 * ox0 - 4);
        const bool recoef = next < ntiles && coef_changes(in, cb, b);       // (uniform) the next tile is another sample's
        f32x4 cfn = {1.f, 0.f, 0.f, 0.f};
        if (recoef) cfn = coef_fetch(in, b, Cphys);
        if (SIDE != SIDE_NONE && ep.mask.p0 && ep.mask.coef_bstride) {
#pragma unroll
            for (int t = 0; t < NT; ++t) mask_coef(ep, cb, 16 * t + m < NOUT ? 16 * t + m : 0, mc0[t], mc2[t]);
        }
        cx.rebase(ep, out, sample_elems, cb);
        constexpr int NP = MTW / MP, NE = decltype(stage)::N;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
#pragma unroll
            for (int e = 0; e < NE; ++e)
                if (e >= p * NE / NP && e < (p + 1) * NE / NP) stage.issue_one(e, scx);
            const float *ap[MP];
            int o[MP][NT];
            EpiIn<SIDE> e[MP][NT];
#pragma unroll
            for (int i = 0; i < MP; ++i) {
                const int ti = wave + 4 * (MP * p + i);
                const int r = ti / CG, cg = ti % CG;
                ap[i] = tile + (2 * r) * RS + 32 * cg + abase;
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    o[i][t] = chan_off[t] + ((cy0 + r) * Wo + cx0 + 16 * cg) * 4;
                    epilogue_loads<SIDE>(e[i][t], cx, o[i][t]);
                }
            }
            f32x4 acc[MP][NT];
#pragma unroll
            for (int i = 0; i < MP; ++i)
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[i][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if constexpr (BF) mfma_tiles_split<MP, NT, KS>(ap, wreg, acc, off);
            else mfma_tiles<MP, NT, KS, 4>(ap, wreg, acc, off);
#pragma unroll
            for (int i = 0; i < MP; ++i)
#pragma unroll
                for (int t = 0; t < NT; ++t)
                {
                    f32x4 v = acc[i][t];
                    if constexpr (BB) {
                        // row class of this M tile (uniform), column class of the quad's first / last element
                        const int ti = wave + 4 * (MP * p + i);
                        const int y = cy0 + ti / CG, xq = cx0 + 16 * (ti % CG) + 4 * kq;
                        const bool r0 = y == 0, r2 = y == Ho - 1;
                        const float bl = r0 ? tb00 : (r2 ? tb20 : tb10);
                        const float bm = r0 ? tb01 : (r2 ? tb21 : tb11);
                        const float br = r0 ? tb02 : (r2 ? tb22 : tb12);
                        v += (f32x4){xq == 0 ? bl : bm, bm, bm, xq + 3 == Wo - 1 ? br : bm};
                        v = bias_relu(v, ep, 0.f);
                    } else {
                        v = bias_relu(v, ep, bias[t]);
                    }
                    epilogue_tail<SIDE>(v, ep, cx, e[i][t], mc0[t], mc2[t], o[i][t], s1[t], s2[t]);
                }
        }
        if (per_tile && ep.stats) {                         // per-sample BatchNorm statistics: one slab per TILE
            stats_reduce<NT, false>(s1, s2, s_stat, ep, NOUT, tidx);
#pragma unroll
            for (int t = 0; t < NT; ++t) { s1[t] = 0.0; s2[t] = 0.0; }
        }
        if (recoef) coef_put(s_coef, in, cfn, Cphys);   // visible after the barrier at the loop top
        tidx = next;
    }
    if (ep.stats && !per_tile) {
        stats_reduce<NT, false>(s1, s2, s_stat, ep, NOUT, blockIdx.x);
        zero_unowned_slabs(ep, NOUT, nslabs);
    }
}

// ============================================================================ kernel A2 (round 5)
// The first convolution (enc.0 o enc.1 composite: CIN -> 8 channels, 4x4 / stride 2, border-bias table) with the output positions
// of a row taken in PAIRS.  With 8 output channels kernel A leaves half of the 16-wide N dimension empty: 8 K-steps (channel, ky)
// x 4 taps kx per 16 positions, every second matrix column multiplying zeros.  Here an M row is the pair p = (X = 2p, 2p + 1) and
// N = (member j, channel).  The two members' 4-tap windows, input columns 4p-1 .. 4p+2 and 4p+1 .. 4p+4, share the columns 4p+1,
// 4p+2; the outer columns 4p-1, 4p (member 0) and 4p+3, 4p+4 (member 1) are the SAME pair of columns one M row apart.  So:
//     centre product   A = columns 4p+1, 4p+2 of two input rows (k lanes = (ky & 1, column)), B = W[kx = 2, 3 | 0, 1]
//     side product     A = columns 4p-1, 4p,                                             B = W[kx = 0, 1 | 2, 3]
//     out[p][member 0] = centre[p] + side[p]          out[p][member 1] = centre[p] + side[p + 1]
// CIN * 2 K-steps per product, every matrix column real.  A row of 64 positions = 32 pairs = two M tiles; the side product takes
// a third M tile for its row 32 (the last position's outer columns; its other 15 rows are never read): 5 CIN * 2 = 20 matrix
// instructions per row where kernel A issues 32.  The member-1 half takes its side value one M row up: a register rename inside a
// lane and one ds_bpermute per M tile (kernel D's recipe).  Output quads are formed with the partner lane as in kernel C.
template <int CIN, int TH, int WPS>
__global__ __launch_bounds__(DM_BLOCK, WPS)
void conv4x4s2_pair_kernel(Operand in, WeightView wv, float *__restrict__ out, Epilogue ep, int Cphys, int H, int W, int ntiles,
                           int nslabs, int per_tile)
{
    constexpr int NOUT = 8, TW = 64;
    constexpr int IH = 2 * TH + 2, RS = 2 * TW + 8, COLS4 = RS / 4, PS = IH * RS;
    constexpr int KS = CIN * 2, RPW = TH / 4;              // K-steps (channel, ky pair) per product; rows per wave
    static_assert(TH % 4 == 0, "rows over four waves");
    __shared__ __attribute__((aligned(16))) float tile[CIN * PS + 64];      // (+ the reach of the side product's third M tile)
    __shared__ __attribute__((aligned(16))) float s_coef[DM_COEF_MAX_C * 4];
    __shared__ double s_stat[4 * 16][2];

    const int Ho = H >> 1, Wo = W >> 1;
    const int tiles_x = Wo / TW, tiles_y = Ho / TH;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, m = lane & 15, kq = lane >> 4;
    const int co = m & 7, mj = m >> 3;                     // the lane's output channel and pair member
    const int kyl = kq >> 1, col = kq & 1;                 // k lane = (row of the ky pair, column of the column pair)

    TileStage<CIN, IH, COLS4, RS, PS, false> stage;
    stage.init(H, W);
    int tidx = blockIdx.x, b = 0, oy0 = 0, ox0 = 0;
    if (tidx < ntiles) {
        int t = tidx;
        ox0 = (t % tiles_x) * TW; t /= tiles_x;
        oy0 = (t % tiles_y) * TH; b = t / tiles_y;
        stage.issue(in, b, Cphys, H, W, 2 * oy0 - 1, 2 * ox0 - 4);
        stage_coef(s_coef, in, b, Cphys);
    }
    if (threadIdx.x < 64) tile[CIN * PS + threadIdx.x] = 0.f;

    // bias_border[row class][column class] of the lane's channel
    const float *tp = ep.bias_border + co;
    const float tb00 = tp[0 * NOUT], tb01 = tp[1 * NOUT], tb02 = tp[2 * NOUT];
    const float tb10 = tp[3 * NOUT], tb11 = tp[4 * NOUT], tb12 = tp[5 * NOUT];
    const float tb20 = tp[6 * NOUT], tb21 = tp[7 * NOUT], tb22 = tp[8 * NOUT];
    // B[k = kq][n = m]: K-step s = (channel s >> 1, ky pair s & 1): ky = 2 (s & 1) + kyl; centre kx = col + (member 0 ? 2 : 0),
    // side kx = col + (member 0 ? 0 : 2)
    float wc[1][KS], ws[1][KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const int c = s >> 1, ky = 2 * (s & 1) + kyl;
        const long long base = wv.off + co * wv.sn + c * wv.sc + ky * wv.sky;
        wc[0][s] = wv.w[base + (col + (mj ? 0 : 2)) * wv.skx];
        ws[0][s] = wv.w[base + (col + (mj ? 2 : 0)) * wv.skx];
    }
    float mc0, mc2;
    mask_coef(ep, 0, co, mc0, mc2);
    double s1 = 0.0, s2 = 0.0;
    // A[m = pair][k = kq]: LDS column of input column x is x - (2 ox0 - 4); centre columns 4p + 5 + col, side columns 4p + 3 + col
    const int abase = kyl * RS + 4 * m + 5 + col;
    auto off = [](int s) { return (s >> 1) * PS + 2 * (s & 1) * RS; };
    const int nb_addr = ((lane + 16) & 63) * 4;            // the lane that holds the next M row of this lane's last one
    EpiCtx<SIDE_NONE> cx;
    const long long sample_elems = (long long)NOUT * Ho * Wo;
    const int chan_off = (co * Ho * Wo + 8 * kq + 4 * mj) * 4;      // the lane's channel plane + its output quad inside 32 positions
    while (tidx < ntiles) {
        __syncthreads();                                   // previous tile consumed; coefficient table visible
        stage.commit(tile, s_coef, Cphys, H, W, 2 * oy0 - 1, 2 * ox0 - 4, in.mode);
        __syncthreads();
        const int cb = b, cy0 = oy0, cx0 = ox0;            // the tile now in LDS
        const int next = tidx + gridDim.x;
        {                                                  // (no next tile: empty descriptor, every load returns 0)
            int t = next < ntiles ? next : tidx;
            ox0 = (t % tiles_x) * TW; t /= tiles_x;
            oy0 = (t % tiles_y) * TH; b = t / tiles_y;
        }
        const auto scx = stage.begin(in, next < ntiles, b, Cphys, H, W, 2 * oy0 - 1, 2 * ox0 - 4);
        const bool recoef = next < ntiles && coef_changes(in, cb, b);       // (uniform) the next tile is another sample's
        f32x4 cfn = {1.f, 0.f, 0.f, 0.f};
        if (recoef) cfn = coef_fetch(in, b, Cphys);
        cx.rebase(ep, out, sample_elems, cb);
        constexpr int NE = decltype(stage)::N;
#pragma unroll
        for (int pp = 0; pp < RPW; ++pp) {
#pragma unroll
            for (int e = 0; e < NE; ++e)
                if (e >= pp * NE / RPW && e < (pp + 1) * NE / RPW) stage.issue_one(e, scx);
            const int r = wave + 4 * pp;
            const float *apc[2], *aps[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                aps[i] = tile + (2 * r) * RS + 64 * i + abase - 2;
                if (i < 2) apc[i] = tile + (2 * r) * RS + 64 * i + abase;
            }
            f32x4 accC[2][1], accS[3][1];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                accS[i][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (i < 2) accC[i][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
            mfma_tiles<2, 1, KS, KS < 4 ? KS : 4>(apc, wc, accC, off);
            mfma_tiles<3, 1, KS, KS < 4 ? KS : 4>(aps, ws, accS, off);
            // member 1 takes the side product one M row up: the row that crosses the lanes comes from 16 lanes up, wrapping into
            // the next M tile's first lane group
            float xs[3];
#pragma unroll
            for (int i = 0; i < 3; ++i)
                xs[i] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(nb_addr, __builtin_bit_cast(int, accS[i][0].x)));
            const int y = cy0 + r;
            const bool r0 = y == 0, r2 = y == Ho - 1;
            const float bl = r0 ? tb00 : (r2 ? tb20 : tb10);
            const float bm = r0 ? tb01 : (r2 ? tb21 : tb11);
            const float br = r0 ? tb02 : (r2 ? tb22 : tb12);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const f32x4 c = accC[i][0], sd = accS[i][0];
                const float up = kq == 3 ? xs[i + 1] : xs[i];
                const f32x4 v = mj ? (f32x4){c.x + sd.y, c.y + sd.z, c.z + sd.w, c.w + up}
                                   : (f32x4){c.x + sd.x, c.y + sd.y, c.z + sd.z, c.w + sd.w};
                // lane (member j, channel): pairs 16 i + 4 kq + (0..3); with the partner lane's values: four consecutive positions
                const f32x4 pv = lane_xor8(v);
                f32x4 qd = mj ? (f32x4){pv.z, v.z, pv.w, v.w} : (f32x4){v.x, pv.x, v.y, pv.y};
                const int xq = cx0 + 32 * i + 8 * kq + 4 * mj;
                qd += (f32x4){xq == 0 ? bl : bm, bm, bm, xq + 3 == Wo - 1 ? br : bm};
                qd = bias_relu(qd, ep, 0.f);
                EpiIn<SIDE_NONE> e0;
                epilogue_tail<SIDE_NONE>(qd, ep, cx, e0, mc0, mc2, chan_off + ((cy0 + r) * Wo + cx0 + 32 * i) * 4, s1, s2);
            }
        }
        if (per_tile && ep.stats) {                         // per-sample BatchNorm statistics: one slab per TILE
            double a1[1] = {s1 + __shfl_xor(s1, 8, 64)}, a2[1] = {s2 + __shfl_xor(s2, 8, 64)};      // the two members of a channel
            stats_reduce<1, false>(a1, a2, s_stat, ep, NOUT, tidx);
            s1 = 0.0; s2 = 0.0;
        }
        if (recoef) coef_put(s_coef, in, cfn, Cphys);   // visible after the barrier at the loop top
        tidx = next;
    }
    if (ep.stats && !per_tile) {
        double a1[1] = {s1 + __shfl_xor(s1, 8, 64)}, a2[1] = {s2 + __shfl_xor(s2, 8, 64)};
        stats_reduce<1, false>(a1, a2, s_stat, ep, NOUT, blockIdx.x);
        zero_unowned_slabs(ep, NOUT, nslabs);
    }
}

// ============================================================================ kernel B
template <int CIN, int NT, int NPASS, int TAPS, bool PIX, int TH, int TW, bool TWO, int SIDE, int WPS, bool BF = false>
__global__ __launch_bounds__(DM_BLOCK, WPS)
void conv3x3_kernel(Operand in, WeightView wv, float *__restrict__ out, Epilogue ep, int Cphys, int NOUT, int H,
                    int W, int ntiles, int nslabs, int per_tile)
{
    constexpr int PADR = TAPS == 9 ? 1 : 0;
    constexpr int IH = TH + 2 * PADR, RS = TAPS == 9 ? TW + 8 : TW, COLS4 = RS / 4;
    constexpr int PSRAW = IH * RS;
    constexpr int PS = PSRAW + ((16 - (PSRAW % 32)) + 32) % 32;      // PS == 16 (mod 32)
    constexpr int KS = (CIN / 4) * TAPS, CG = TW / 16, MT = TH * CG, MTW = MT / 4;
    constexpr int NTT = NT * NPASS;
    constexpr int MP = NT == 1 ? 2 : 1;
    static_assert(CIN % 4 == 0, "channel groups of 4");
    static_assert(MTW % MP == 0 && MTW >= MP, "M tiles per wave");
    static_assert(!PIX || TAPS == 9, "pixel shuffle is the 3x3 formulation");
    __shared__ __attribute__((aligned(16))) float tile[CIN * PS];
    __shared__ __attribute__((aligned(16))) float s_coef[DM_COEF_MAX_C * 4];
    __shared__ double s_stat[4 * NTT * 16][2];

    const int tiles_x = W / TW, tiles_y = H / TH;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, m = lane & 15, kq = lane >> 4;
    const int CO = PIX ? NOUT >> 2 : NOUT;           // physical output channels
    const int OH = PIX ? 2 * H : H, OW = PIX ? 2 * W : W;

    TileStage<CIN, IH, COLS4, RS, PS, TWO> stage;
    stage.init(H, W);

    double s1[NTT], s2[NTT];
#pragma unroll
    for (int t = 0; t < NTT; ++t) { s1[t] = 0.0; s2[t] = 0.0; }
    const int abase = kq * PS + m + 3 * PADR;
    auto off = [](int s) {
        const int cg4 = s / TAPS, tap = s % TAPS;
        return 4 * cg4 * PS + (TAPS == 9 ? (tap / 3) * RS + tap % 3 : 0);
    };

    EpiCtx<SIDE> cx;
    const long long sample_elems = (long long)CO * OH * OW;

    // NPASS > 1 (more output-channel tiles than fit in registers): the tile walk is repeated per pass
#pragma unroll
    for (int pass = 0; pass < NPASS; ++pass) {
        int tidx = blockIdx.x, b = 0, y0 = 0, x0 = 0;
        if (pass > 0) __syncthreads();                      // last tile of the previous pass fully consumed
        if (tidx < ntiles) {
            int t = tidx;
            x0 = (t % tiles_x) * TW; t /= tiles_x;
            y0 = (t % tiles_y) * TH; b = t / tiles_y;
            stage.issue(in, b, Cphys, H, W, y0 - PADR, x0 - 4 * PADR);
            stage_coef(s_coef, in, b, Cphys);
        }
        float wreg[NT][KS], bias[NT], mc0[NT], mc2[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int n = 16 * (pass * NT + t) + m;
            const int nb = n < NOUT ? (PIX ? n >> 2 : n) : 0;
            bias[t] = ep.bias ? ep.bias[nb] : 0.f;
            mask_coef(ep, 0, nb, mc0[t], mc2[t]);
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
                wreg[t][s] = BF ? split_pack1(wvl) : wvl;
            }
        }

        int chan_off[NT];                                  // byte offset of the lane's channel plane, phase row / x quad
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int n = 16 * (pass * NT + t) + m;
            if (PIX) {
                const int nb = n >> 2, py = (n >> 1) & 1, px = n & 1;
                chan_off[t] = n < NOUT ? ((nb * OH + py) * OW + 8 * kq + 4 * px) * 4 : DM_VOFF_NONE;
            } else {
                chan_off[t] = n < NOUT ? (n * OH * OW + 4 * kq) * 4 : DM_VOFF_NONE;
            }
        }

        while (tidx < ntiles) {
            __syncthreads();
            stage.template commit<BF>(tile, s_coef, Cphys, H, W, y0 - PADR, x0 - 4 * PADR, in.mode);
            __syncthreads();
            const int cb = b, cy0 = y0, cx0 = x0;
            const int next = tidx + gridDim.x;
            {                                              // (no next tile: empty descriptor, every load returns 0)
                int t = next < ntiles ? next : tidx;
                x0 = (t % tiles_x) * TW; t /= tiles_x;
                y0 = (t % tiles_y) * TH; b = t / tiles_y;
            }
            // the next tile's loads are requested element by element between the M tiles below (tile.h: issue_one)
            const auto scx = stage.begin(in, next < ntiles, b, Cphys, H, W, y0 - PADR, x0 - 4 * PADR);
            const bool recoef = next < ntiles && coef_changes(in, cb, b);   // (uniform) the next tile is another sample's
            f32x4 cfn = {1.f, 0.f, 0.f, 0.f};
            if (recoef) cfn = coef_fetch(in, b, Cphys);
            if (SIDE != SIDE_NONE && ep.mask.p0 && ep.mask.coef_bstride) {
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const int n = 16 * (pass * NT + t) + m;
                    mask_coef(ep, cb, n < NOUT ? (PIX ? n >> 2 : n) : 0, mc0[t], mc2[t]);
                }
            }
            cx.rebase(ep, out, sample_elems, cb);
            constexpr int NP = MTW / MP, NE = decltype(stage)::N;
#pragma unroll
            for (int p = 0; p < NP; ++p) {
#pragma unroll
                for (int e = 0; e < NE; ++e)
                    if (e >= p * NE / NP && e < (p + 1) * NE / NP) stage.issue_one(e, scx);
                const float *ap[MP];
                int o[MP][NT];
                EpiIn<SIDE> e[MP][NT];
#pragma unroll
                for (int i = 0; i < MP; ++i) {
                    const int ti = wave + 4 * (MP * p + i);
                    const int r = ti / CG, cg = ti % CG;
                    ap[i] = tile + r * RS + 16 * cg + abase;
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        if (PIX) o[i][t] = chan_off[t] + (2 * (cy0 + r) * OW + 2 * (cx0 + 16 * cg)) * 4;
                        else o[i][t] = chan_off[t] + ((cy0 + r) * OW + cx0 + 16 * cg) * 4;
                        epilogue_loads<SIDE>(e[i][t], cx, o[i][t]);
                    }
                }
                f32x4 acc[MP][NT];
#pragma unroll
                for (int i = 0; i < MP; ++i)
#pragma unroll
                    for (int t = 0; t < NT; ++t) acc[i][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if constexpr (BF) mfma_tiles_split<MP, NT, KS>(ap, wreg, acc, off);
                else mfma_tiles<MP, NT, KS, (TAPS == 9 ? 3 : 4)>(ap, wreg, acc, off);
#pragma unroll
                for (int i = 0; i < MP; ++i)
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        const int tt = pass * NT + t;
                        const int n = 16 * tt + m;
                        f32x4 v = bias_relu(acc[i][t], ep, bias[t]);
                        if (PIX) {
                            // partner lane (n ^ 1) holds the other x-phase of the same output row
                            const f32x4 pv = lane_xor1(v);
                            v = (n & 1) ? (f32x4){pv.z, v.z, pv.w, v.w} : (f32x4){v.x, pv.x, v.y, pv.y};
                        }
                        epilogue_tail<SIDE>(v, ep, cx, e[i][t], mc0[t], mc2[t], o[i][t], s1[tt], s2[tt]);
                    }
            }
            if (NPASS == 1 && per_tile && ep.stats) {       // per-sample BatchNorm statistics: one slab per TILE
                stats_reduce<NTT, PIX>(s1, s2, s_stat, ep, CO, tidx);
#pragma unroll
                for (int t = 0; t < NTT; ++t) { s1[t] = 0.0; s2[t] = 0.0; }
            }
            if (recoef) coef_put(s_coef, in, cfn, Cphys);   // visible after the barrier at the loop top
            tidx = next;
        }
    }
    if (ep.stats && !(NPASS == 1 && per_tile)) {
        // (NPASS > 1 with per-tile slabs: the launcher gives every tile its own workgroup, blockIdx.x == tile)
        stats_reduce<NTT, PIX>(s1, s2, s_stat, ep, CO, blockIdx.x);
        zero_unowned_slabs(ep, CO, nslabs);
    }
}

// ============================================================================ kernel C
// ConvTranspose2d(4,2,1) decomposed by output phase.  The 3x3-neighbourhood formulation of kernel B multiplies every
// phase (py,px) by all 9 taps although it only touches 2x2 of them (5/9 of its MFMA work is zero weights).  Here
//   COUT = 16: one N tile per phase (n = co), K = CIN x its 2x2 taps          -> 4 x CIN   MFMAs per 16 positions (9 x CIN there)
//   COUT =  8: one N tile per phase ROW py (n = px*8 + co), K = CIN x 2x3 taps -> 3 x CIN                      (4.5 x CIN there)
// The A operand is no longer shared between N tiles (each phase reads its own shifted window), which costs LDS reads,
// not matrix-pipe time -- and these kernels are matrix-pipe bound (60-70 % busy).
// Output: the two x phases of a row are interleaved in registers (COUT = 16: both live in the lane, two 16-byte
// stores per row; COUT = 8: lanes m and m^8 swap halves, one store), so stores stay 16 contiguous bytes per lane.
template <int CIN, int COUT, int TH, int TW, bool TWO, int SIDE, int WPS, bool BF = false>
__global__ __launch_bounds__(DM_BLOCK, WPS)
void convT_phase_kernel(Operand in, WeightView wv, float *__restrict__ out, Epilogue ep, int Cphys, int H, int W,
                        int ntiles, int nslabs)
{
    static_assert(COUT == 8 || COUT == 16, "built for 8 or 16 output channels");
    constexpr int IH = TH + 2, RS = TW + 8, COLS4 = RS / 4;
    constexpr int PSRAW = IH * RS;
    constexpr int PS = PSRAW + ((16 - (PSRAW % 32)) + 32) % 32;      // PS == 16 (mod 32)
    constexpr int CG = TW / 16, MT = TH * CG, MTW = MT / 4, MP = 2;
    constexpr int NPX = COUT == 16 ? 2 : 1;                // N tiles per phase row
    constexpr int TAPX = COUT == 16 ? 2 : 3;               // x taps per N tile
    constexpr int KS = (CIN / 4) * 2 * TAPX;               // K steps per N tile
    static_assert(CIN % 4 == 0 && MTW % MP == 0 && MTW >= MP, "shape");
    __shared__ __attribute__((aligned(16))) float tile[CIN * PS];
    __shared__ __attribute__((aligned(16))) float s_coef[DM_COEF_MAX_C * 4];
    __shared__ double s_stat[4 * 16][2];

    const int tiles_x = W / TW, tiles_y = H / TH;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, m = lane & 15, kq = lane >> 4;
    const int OH = 2 * H, OW = 2 * W;
    const int co = COUT == 16 ? m : (m & 7), pxl = COUT == 16 ? 0 : (m >> 3);   // lane's channel, x phase (COUT = 8)

    TileStage<CIN, IH, COLS4, RS, PS, TWO> stage;
    stage.init(H, W);
    int tidx = blockIdx.x, b = 0, y0 = 0, x0 = 0;
    if (tidx < ntiles) {
        int t = tidx;
        x0 = (t % tiles_x) * TW; t /= tiles_x;
        y0 = (t % tiles_y) * TH; b = t / tiles_y;
        stage.issue(in, b, Cphys, H, W, y0 - 1, x0 - 4);
        stage_coef(s_coef, in, b, Cphys);
    }

    // weights of N tile (py, pxt): step s = (cg4, a, bb) <-> input channel 4*cg4 + kq, tap row py + a, tap column
    // (COUT = 16: pxt + bb, COUT = 8: bb); ky = py + 3 - 2*row, kx = px + 3 - 2*col (zero when outside 0..3)
    float wreg[2][NPX][1][KS];
#pragma unroll
    for (int py = 0; py < 2; ++py)
#pragma unroll
        for (int pxt = 0; pxt < NPX; ++pxt)
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const int cg4 = s / (2 * TAPX), j = s % (2 * TAPX), a = j / TAPX, bb = j % TAPX;
                const int c = 4 * cg4 + kq;
                const int px = COUT == 16 ? pxt : pxl;
                const int row = py + a, col = COUT == 16 ? pxt + bb : bb;
                const int ky = py + 3 - 2 * row, kx = px + 3 - 2 * col;
                float wvl = 0.f;
                if (ky >= 0 && ky <= 3 && kx >= 0 && kx <= 3) wvl = wv.w[wv.off + co * wv.sn + c * wv.sc + ky * wv.sky + kx * wv.skx];
                wreg[py][pxt][0][s] = BF ? split_pack1(wvl) : wvl;
            }
    const float bias = ep.bias ? ep.bias[co] : 0.f;
    float mc0, mc2;
    mask_coef(ep, 0, co, mc0, mc2);

    double s1 = 0.0, s2 = 0.0;
    const int abase = kq * PS + m + 3;
    EpiCtx<SIDE> cx;
    const long long sample_elems = (long long)COUT * OH * OW;
    // byte offset of the lane's channel plane + its 8 (COUT = 16) or 4 (COUT = 8) output columns of a 16-position group
    const int chan_off = (co * OH * OW + 8 * kq + 4 * pxl) * 4;

    while (tidx < ntiles) {
        __syncthreads();
        stage.template commit<BF>(tile, s_coef, Cphys, H, W, y0 - 1, x0 - 4, in.mode);
        __syncthreads();
        const int cb = b, cy0 = y0, cx0 = x0;
        const int next = tidx + gridDim.x;
        {                                              // (no next tile: empty descriptor, every load returns 0)
            int t = next < ntiles ? next : tidx;
            x0 = (t % tiles_x) * TW; t /= tiles_x;
            y0 = (t % tiles_y) * TH; b = t / tiles_y;
        }
        const auto scx = stage.begin(in, next < ntiles, b, Cphys, H, W, y0 - 1, x0 - 4);
        const bool recoef = next < ntiles && coef_changes(in, cb, b);       // (uniform) the next tile is another sample's
        f32x4 cfn = {1.f, 0.f, 0.f, 0.f};
        if (recoef) cfn = coef_fetch(in, b, Cphys);
        if (SIDE != SIDE_NONE && ep.mask.p0 && ep.mask.coef_bstride) mask_coef(ep, cb, co, mc0, mc2);
        cx.rebase(ep, out, sample_elems, cb);
        constexpr int NP = MTW / MP, NE = decltype(stage)::N;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
#pragma unroll
            for (int e = 0; e < NE; ++e)
                if (e >= p * NE / NP && e < (p + 1) * NE / NP) stage.issue_one(e, scx);
            const float *ap[MP];
            int obase[MP];
#pragma unroll
            for (int i = 0; i < MP; ++i) {
                const int ti = wave + 4 * (MP * p + i);
                const int r = ti / CG, cg = ti % CG;
                ap[i] = tile + r * RS + 16 * cg + abase;
                obase[i] = chan_off + (2 * (cy0 + r) * OW + 2 * (cx0 + 16 * cg)) * 4;
            }
#pragma unroll
            for (int py = 0; py < 2; ++py) {
                EpiIn<SIDE> e[MP][NPX];
#pragma unroll
                for (int i = 0; i < MP; ++i)
#pragma unroll
                    for (int h = 0; h < NPX; ++h) epilogue_loads<SIDE>(e[i][h], cx, obase[i] + py * OW * 4 + 16 * h);
                f32x4 acc[NPX][MP][1];
#pragma unroll
                for (int pxt = 0; pxt < NPX; ++pxt) {
#pragma unroll
                    for (int i = 0; i < MP; ++i) acc[pxt][i][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
                    auto off = [py, pxt](int s) {
                        const int cg4 = s / (2 * TAPX), j = s % (2 * TAPX), a = j / TAPX, bb = j % TAPX;
                        return 4 * cg4 * PS + (py + a) * RS + (COUT == 16 ? pxt + bb : bb);
                    };
                    if constexpr (BF) mfma_tiles_split<MP, 1, KS>(ap, wreg[py][pxt], acc[pxt], off);
                    else mfma_tiles<MP, 1, KS, TAPX * 2>(ap, wreg[py][pxt], acc[pxt], off);
                }
#pragma unroll
                for (int i = 0; i < MP; ++i) {
                    if constexpr (COUT == 16) {
                        const f32x4 v0 = bias_relu(acc[0][i][0], ep, bias), v1 = bias_relu(acc[1][i][0], ep, bias);
                        epilogue_tail<SIDE>((f32x4){v0.x, v1.x, v0.y, v1.y}, ep, cx, e[i][0], mc0, mc2,
                                            obase[i] + py * OW * 4, s1, s2);
                        epilogue_tail<SIDE>((f32x4){v0.z, v1.z, v0.w, v1.w}, ep, cx, e[i][1], mc0, mc2,
                                            obase[i] + py * OW * 4 + 16, s1, s2);
                    } else {
                        const f32x4 v = bias_relu(acc[0][i][0], ep, bias);
                        const f32x4 pv = lane_xor8(v);     // partner lane holds the other x phase of the same channel
                        epilogue_tail<SIDE>(pxl ? (f32x4){pv.z, v.z, pv.w, v.w} : (f32x4){v.x, pv.x, v.y, pv.y}, ep, cx,
                                            e[i][0], mc0, mc2, obase[i] + py * OW * 4, s1, s2);
                    }
                }
            }
        }
        if (recoef) coef_put(s_coef, in, cfn, Cphys);   // visible after the barrier at the loop top
        tidx = next;
    }
    if (ep.stats) {
        // channel of a lane: m (COUT = 16) or m & 7 (COUT = 8: lanes m and m^8 hold the two x phases)
        double a = s1, c = s2;
        a += __shfl_xor(a, 16, 64); c += __shfl_xor(c, 16, 64);
        a += __shfl_xor(a, 32, 64); c += __shfl_xor(c, 32, 64);
        if (COUT == 8) { a += __shfl_xor(a, 8, 64); c += __shfl_xor(c, 8, 64); }
        if (lane < COUT) { s_stat[wave * 16 + lane][0] = a; s_stat[wave * 16 + lane][1] = c; }
        __syncthreads();
        if (threadIdx.x < COUT) {
            double ta = 0.0, tc = 0.0;
#pragma unroll
            for (int w = 0; w < 4; ++w) { ta += s_stat[w * 16 + threadIdx.x][0]; tc += s_stat[w * 16 + threadIdx.x][1]; }
            ep.stats[((long long)blockIdx.x * COUT + threadIdx.x) * 2 + 0] = ta;
            ep.stats[((long long)blockIdx.x * COUT + threadIdx.x) * 2 + 1] = tc;
        }
        zero_unowned_slabs(ep, COUT, nslabs);
    }
}

// ============================================================================ kernel D
// Backward of a Conv2d(CX -> CD, 4, stride 2, padding 1) in ONE pass over its operands: the data gradient (the
// phase-decomposed transposed convolution of kernel C) AND the weight gradient (wgrad_mfma.hip) from one staging of
//   da  = A*dy + B*a_out + C   (BatchNorm backward folded into the load; CD channels on the dy grid, with a 1-pixel halo)
//   T   = relu(scale*a_in + shift)   (the layer's input as the forward saw it; CX channels, the 2x grid, (2TH+2) x (2TW+8))
// As two kernels each of them staged da (two tensors) and re-read a_in: 1.34 GB moved for 0.54 GB of tensors on enc.4 at
// B = 2048.  One workgroup of 512 threads per CU (two waves per SIMD): the staging registers of both tiles are spread
// over 512 threads (64 VGPRs per thread), which is what lets the transposed convolution's weights (48), the weight
// gradient's accumulators (32) and the next tile's loads live together below 256 registers.
//   data gradient   wave w takes M tiles w and w + 8 of the tile's TH x TW/16 (16 positions each), both output phase rows
//   weight gradient wave w takes position row w: M = dy channels (one 16-row tile), N = (ct, ky, kx) in NTT tiles,
//                   K = the row's positions; the A operand is read 16 bytes per lane (positions 4 kq .. 4 kq + 3 of a
//                   16-position span = K-steps 0..3), one ds_read_b128 per four MFMA steps
// The weight-gradient accumulators persist over the workgroup's tiles; the eight waves are combined in wave order and the
// workgroup writes one slab (dm_reduce_slabs_multi adds the slabs in slab order): bit-reproducible.
// FB_BLOCK threads per workgroup: 512 (one workgroup per CU, two waves per SIMD; the staging registers of both tiles
// spread over 512 threads) or 256 (two independent workgroups per CU, whose load / commit phases can hide under each
// other's matrix phase instead of all eight waves of a CU committing at the same time).
template <int CD, int CX, int TH, int TW>
struct FusedBwdGeom {
    static_assert(CD == 16 && CX == 8 && (TH == 8 || TH == 4) && TW % 16 == 0, "built for enc.4: 8 -> 16 channels");
    static constexpr int IH = TH + 2, RS = TW + 8, COLS4 = RS / 4, PSRAW = IH * RS;
    static constexpr int PS = PSRAW + ((16 - (PSRAW % 32)) + 32) % 32;            // == 16 (mod 32): kernel C's A reads
    static constexpr int TROWS = 2 * TH + 2, RST = 2 * TW + 8, TCOLS4 = RST / 4, PST = TROWS * RST;
    static constexpr int N = CX * 16, NTT = N / 16;
    static constexpr int TILE_FLOATS = CD * PS + CX * PST, RED_FLOATS = NTT * 256;
    static constexpr int LDS_FLOATS = (TILE_FLOATS > RED_FLOATS ? TILE_FLOATS : RED_FLOATS) + 2 * DM_COEF_MAX_C * 4;
    static constexpr size_t LDS_BYTES = (size_t)LDS_FLOATS * 4 + 8 * 16 * 2 * sizeof(double);
    // role-split form: two tile buffers + the coefficient tables + the statistics scratch
    static constexpr size_t SPLIT_LDS_BYTES = (size_t)(2 * TILE_FLOATS + 2 * DM_COEF_MAX_C * 4) * 4 + 8 * 16 * 2 * sizeof(double);
    static_assert(RED_FLOATS <= TILE_FLOATS, "the slab combine reuses a tile buffer");
};

// TPRE: the next tile's T elements are prefetched into registers during the matrix phase like the da elements; false
// (the 256-thread build): they are loaded at the top of the tile, while the da tile is committed -- their registers are
// then dead during the matrix phase (64 fewer live there: no spills), and the load latency is left to the CU's other
// workgroup to fill.
template <int CD, int CX, int TH, int TW, int FB_BLOCK, bool TPRE>
__global__ __launch_bounds__(FB_BLOCK, 2)
void bwd_s2_fused_kernel(Operand dy, Operand tin, WeightView wv, float *__restrict__ dx, Epilogue ep,
                         float *__restrict__ wslabs, int H, int W, int ntiles)
{
    using G = FusedBwdGeom<CD, CX, TH, TW>;
    constexpr int FB_WAVES = FB_BLOCK / 64;
    constexpr int IH = G::IH, RS = G::RS, COLS4 = G::COLS4, PS = G::PS;
    constexpr int TROWS = G::TROWS, RST = G::RST, TCOLS4 = G::TCOLS4, PST = G::PST, NTT = G::NTT, N = G::N;
    constexpr int CGN = TW / 16, MP = 2;                   // 16-position groups per row; M tiles in flight per wave
    constexpr int NPASS = TH * CGN / (FB_WAVES * MP);      // data-gradient passes per wave: 1 (512 threads) or 2 (256)
    constexpr int WROWS = TH / FB_WAVES;                   // weight-gradient position rows per wave: 1 or 2
    static_assert(TH * CGN == FB_WAVES * MP * NPASS && TH == FB_WAVES * WROWS, "tile split over the waves");
    constexpr int TAPX = 3, KS = (CD / 4) * 2 * TAPX;      // kernel C with 8 output channels: n = (px, co), 2 x 3 taps
    extern __shared__ __attribute__((aligned(16))) float fb_lds[];
    float *tileD = fb_lds, *tileT = fb_lds + CD * PS;
    float *s_coefD = fb_lds + (G::LDS_FLOATS - 2 * DM_COEF_MAX_C * 4), *s_coefT = s_coefD + DM_COEF_MAX_C * 4;
    double (*s_stat)[2] = reinterpret_cast<double (*)[2]>(fb_lds + G::LDS_FLOATS);

    const int lane = threadIdx.x & 63, m = lane & 15, kq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int OH = 2 * H, OW = 2 * W;
    const int tiles_x = W / TW, tiles_y = H / TH;
    const int co = m & 7, pxl = m >> 3;                    // data gradient: the lane's a_in channel and x phase

    TileStage<CD, IH, COLS4, RS, PS, true, FB_BLOCK> stD;
    TileStage<CX, TROWS, TCOLS4, RST, PST, false, FB_BLOCK> stT;
    stD.init(H, W);
    stT.init(OH, OW);
    int tidx = blockIdx.x, b = 0, y0 = 0, x0 = 0;
    if (tidx < ntiles) {
        int t = tidx;
        x0 = (t % tiles_x) * TW; t /= tiles_x;
        y0 = (t % tiles_y) * TH; b = t / tiles_y;
        stD.issue(dy, b, CD, H, W, y0 - 1, x0 - 4);
        if constexpr (TPRE) stT.issue(tin, b, CX, OH, OW, 2 * y0 - 1, 2 * x0 - 4);
        stage_coef(s_coefD, dy, b, CD);
        stage_coef(s_coefT, tin, b, CX);
    }

    // data-gradient weights (kernel C, COUT = 8): N tile per phase row py, n = (px, co); step s = (cg4, a, bb)
    float wreg[2][1][KS];
#pragma unroll
    for (int py = 0; py < 2; ++py)
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int cg4 = s / (2 * TAPX), j = s % (2 * TAPX), a = j / TAPX, bb = j % TAPX;
            const int c = 4 * cg4 + kq;
            const int ky = py + 3 - 2 * (py + a), kx = pxl + 3 - 2 * bb;
            float wvl = 0.f;
            if (ky >= 0 && ky <= 3 && kx >= 0 && kx <= 3) wvl = wv.w[wv.off + co * wv.sn + c * wv.sc + ky * wv.sky + kx * wv.skx];
            wreg[py][0][s] = wvl;
        }
    float mc0, mc2;
    mask_coef(ep, 0, co, mc0, mc2);
    double s1 = 0.0, s2 = 0.0;
    const int abase = kq * PS + m + 3;
    EpiCtx<SIDE_MASK> cx;
    const long long sample_elems = (long long)CX * OH * OW;
    const int chan_off = (co * OH * OW + 8 * kq + 4 * pxl) * 4;

    // weight gradient: B column n = 16 t + m = (ct, ky, kx); this lane's positions of a 16-position span are 4 kq .. 4 kq + 3
    int bl[NTT];
#pragma unroll
    for (int t = 0; t < NTT; ++t) {
        const int n = 16 * t + m;
        bl[t] = (n >> 4) * PST + ((n >> 2) & 3) * RST + (n & 3) + 3 + 8 * kq + 2 * wave * RST;
    }
    const int al = m * PS + (wave + 1) * RS + 4 + 4 * kq;       // A row m = dy channel, position rows wave, wave + FB_WAVES
    f32x4 wacc[NTT];
#pragma unroll
    for (int t = 0; t < NTT; ++t) wacc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};

    while (tidx < ntiles) {
        if constexpr (!TPRE) stT.issue(tin, b, CX, OH, OW, 2 * y0 - 1, 2 * x0 - 4);      // (in flight across the barrier and the da commit)
        __syncthreads();                                   // previous tile consumed
        stD.commit(tileD, s_coefD, CD, H, W, y0 - 1, x0 - 4, dy.mode);
        stT.commit(tileT, s_coefT, CX, OH, OW, 2 * y0 - 1, 2 * x0 - 4, tin.mode);
        __syncthreads();
        const int cb = b, cy0 = y0, cx0 = x0;
        const int next = tidx + gridDim.x;
        {
            int t = next < ntiles ? next : tidx;
            x0 = (t % tiles_x) * TW; t /= tiles_x;
            y0 = (t % tiles_y) * TH; b = t / tiles_y;
        }
        const auto scD = stD.begin(dy, next < ntiles, b, CD, H, W, y0 - 1, x0 - 4);
        const auto scT = stT.begin(tin, TPRE && next < ntiles, b, CX, OH, OW, 2 * y0 - 1, 2 * x0 - 4);
        (void)scT;
        cx.rebase(ep, dx, sample_elems, cb);
        constexpr int NED = decltype(stD)::N, NET = decltype(stT)::N;

        // ---- data gradient: M tiles wave + FB_WAVES * (MP * pass + i), phase rows py = 0, 1 ------------------------------
#pragma unroll
        for (int pass = 0; pass < NPASS; ++pass) {
            const float *ap[MP];
            int obase[MP];
#pragma unroll
            for (int i = 0; i < MP; ++i) {
                const int ti = wave + FB_WAVES * (MP * pass + i);
                const int r = ti / CGN, cg = ti % CGN;
                ap[i] = tileD + r * RS + 16 * cg + abase;
                obase[i] = chan_off + (2 * (cy0 + r) * OW + 2 * (cx0 + 16 * cg)) * 4;
            }
#pragma unroll
            for (int py = 0; py < 2; ++py) {
                // the next tile's da elements are requested under the products of this one
                constexpr int NQ = 2 * NPASS;
                const int qd = 2 * pass + py;
#pragma unroll
                for (int e = 0; e < NED; ++e)
                    if (e >= qd * NED / NQ && e < (qd + 1) * NED / NQ) stD.issue_one(e, scD);
                EpiIn<SIDE_MASK> e[MP];
#pragma unroll
                for (int i = 0; i < MP; ++i) epilogue_loads<SIDE_MASK>(e[i], cx, obase[i] + py * OW * 4);
                f32x4 acc[MP][1];
#pragma unroll
                for (int i = 0; i < MP; ++i) acc[i][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
                auto off = [py](int s) {
                    const int cg4 = s / (2 * TAPX), j = s % (2 * TAPX), a = j / TAPX, bb = j % TAPX;
                    return 4 * cg4 * PS + (py + a) * RS + bb;
                };
                mfma_tiles<MP, 1, KS, TAPX * 2>(ap, wreg[py], acc, off);
#pragma unroll
                for (int i = 0; i < MP; ++i) {
                    const f32x4 v = acc[i][0];
                    const f32x4 pv = lane_xor8(v);         // partner lane holds the other x phase of the same channel
                    epilogue_tail<SIDE_MASK>(pxl ? (f32x4){pv.z, v.z, pv.w, v.w} : (f32x4){v.x, pv.x, v.y, pv.y}, ep, cx,
                                             e[i], mc0, mc2, obase[i] + py * OW * 4, s1, s2);
                }
            }
        }

        // ---- weight gradient: position rows wave (+ FB_WAVES), TW / 16 spans of 16 positions, 4 K-steps per span ---------
        {
            constexpr int NSPAN = TW / 16, NQ = WROWS * NSPAN * 4;      // q = (row, span, K-step)
            auto aoff = [](int q) { return (q / (NSPAN * 4)) * FB_WAVES * RS + 16 * ((q >> 2) % NSPAN); };
            auto boff = [](int q) { return (q / (NSPAN * 4)) * FB_WAVES * 2 * RST + 32 * ((q >> 2) % NSPAN) + 2 * (q & 3); };
            f32x4 av[2];
            float bv[2][NTT];
            av[0] = *reinterpret_cast<const f32x4 *>(tileD + al);
#pragma unroll
            for (int t = 0; t < NTT; ++t) bv[0][t] = tileT[bl[t]];
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                if (q + 1 < NQ) {
                    if (((q + 1) & 3) == 0) av[((q + 1) >> 2) & 1] = *reinterpret_cast<const f32x4 *>(tileD + al + aoff(q + 1));
#pragma unroll
                    for (int t = 0; t < NTT; ++t) bv[(q + 1) & 1][t] = tileT[bl[t] + boff(q + 1)];
                }
                // the next tile's T elements trickle out between the steps
                if constexpr (TPRE) {
#pragma unroll
                    for (int e = 0; e < NET; ++e)
                        if (e >= q * NET / NQ && e < (q + 1) * NET / NQ) stT.issue_one(e, scT);
                }
                __builtin_amdgcn_sched_barrier(0);
                const float a = av[(q >> 2) & 1][q & 3];
#pragma unroll
                for (int t = 0; t < NTT; ++t) wacc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bv[q & 1][t], wacc[t], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        tidx = next;
    }

    // ---- statistics of the data gradient: (sum v, sum v * a_in) per a_in channel -> stats[blockIdx][CX][2] --------------
    if (ep.stats) {
        double a = s1, c = s2;
        a += __shfl_xor(a, 16, 64); c += __shfl_xor(c, 16, 64);
        a += __shfl_xor(a, 32, 64); c += __shfl_xor(c, 32, 64);
        a += __shfl_xor(a, 8, 64); c += __shfl_xor(c, 8, 64);
        if (lane < CX) { s_stat[wave * 16 + lane][0] = a; s_stat[wave * 16 + lane][1] = c; }
        __syncthreads();
        if (threadIdx.x < CX) {
            double ta = 0.0, tc = 0.0;
#pragma unroll
            for (int w = 0; w < FB_WAVES; ++w) { ta += s_stat[w * 16 + threadIdx.x][0]; tc += s_stat[w * 16 + threadIdx.x][1]; }
            ep.stats[((long long)blockIdx.x * CX + threadIdx.x) * 2 + 0] = ta;
            ep.stats[((long long)blockIdx.x * CX + threadIdx.x) * 2 + 1] = tc;
        }
    }
    // ---- weight-gradient slab: the waves in wave order.  wacc[t][j] is dy channel 4 (lane >> 4) + j, column 16 t + m
    float *red = fb_lds;
    for (int w = 0; w < FB_WAVES; ++w) {
        __syncthreads();
        if (wave == w) {
#pragma unroll
            for (int t = 0; t < NTT; ++t) {
                f32x4 *p = reinterpret_cast<f32x4 *>(red + (t * 64 + lane) * 4);
                if (w == 0) *p = wacc[t];
                else *p = *p + wacc[t];
            }
        }
    }
    __syncthreads();
    float *slab = wslabs + (long long)blockIdx.x * (CD * N);
    for (int i = threadIdx.x; i < NTT * 256; i += FB_BLOCK) {
        const int j = i & 3, l = (i >> 2) & 63, t = i >> 8;
        slab[(4 * (l >> 4) + j) * N + 16 * t + (l & 15)] = red[i];
    }
}

// ---- kernel D, role-split form.  One 512-thread workgroup per CU, TWO LDS buffers (2 x 67 KB), and two kinds of waves:
//   waves 0..3  data gradient of tile i (kernel C's products and epilogue) -- they never touch global inputs;
//   waves 4..7  request tile i+1 (every load up front), run the weight-gradient products of tile i while the loads are
//               in flight, then transform + write tile i+1 into the other buffer.
// Each SIMD hosts one wave of either kind, one barrier per tile.  With all eight waves in the same phase (the form above)
// the matrix pipe idles while everybody commits and waits at the two barriers: 261 us for enc.4 at B = 2048 against a
// 137 us matrix floor; here the commit, the load latency and the epilogue of one kind overlap the other kind's products.
// The two kinds run separate loops (same trip count, one s_barrier per iteration each): their registers -- transposed-
// convolution weights on one side, weight-gradient accumulators and 32 staged float4 on the other -- are then never live
// together.
// BF: split-bf16 operands (tile.h: split_pack4) -- both LDS tiles and the transposed-convolution weights hold (hi, lo) bf16
// pairs, the products run on v_mfma_f32_16x16x32_bf16 at four K-steps per instruction pair: a quarter of the matrix time.
// ZF (round 5; a tile spans the row, TW == W): the data gradient without structural zeros.  Folding both x phases of an
// output pixel pair into the 16-wide N dimension makes them share a window of THREE dy columns of which each uses two: a third
// of the 24 K-steps multiplies zeros.  The centre column (kx = 1 + px) is used by BOTH phases; the outer ones by one each
// (px = 0: column X - 1 with kx = 3; px = 1: column X + 1 with kx = 0).  So two products share ONE A operand (dy column c = the
// M row's own column): "centre" with B = W[kx = 1 + px] and "side" with B = W[kx = 3 | 0], 8 K-steps each, and
//     dx[X][px = 0] = centre[X] + side[X - 1],     dx[X][px = 1] = centre[X] + side[X + 1]:
// the side accumulator shifted by one M row, a register rename inside a lane plus one cross-lane value per M tile
// (ds_bpermute); columns -1 and W are the zero padding.  16 matrix instructions per M tile and phase row instead of 24, a
// third of the A-operand LDS reads, 32 weight registers instead of 48.
template <int CD, int CX, int TH, int TW, bool BF, bool ZF>
__global__ __launch_bounds__(512, 2)
void bwd_s2_split_kernel(Operand dy, Operand tin, WeightView wv, float *__restrict__ dx, Epilogue ep,
                         float *__restrict__ wslabs, int H, int W, int ntiles, int dbg)
{
    // dbg (DM_FUSED_BWD_DBG, measurements only; results are then wrong): 1 skips the weight-gradient products, 2 the data
    // gradient, 4 the loads and commits of every tile but the first
    using G = FusedBwdGeom<CD, CX, TH, TW>;
    constexpr int RW = 4;                                   // waves per role
    constexpr int IH = G::IH, RS = G::RS, COLS4 = G::COLS4, PS = G::PS;
    constexpr int TROWS = G::TROWS, RST = G::RST, TCOLS4 = G::TCOLS4, PST = G::PST, NTT = G::NTT, N = G::N;
    constexpr int BUF = CD * PS + CX * PST;                 // floats of one (da, T) tile pair
    constexpr int CGN = TW / 16, MP = 2, NPASS = TH * CGN / (RW * MP), WROWS = TH / RW;
    static_assert(TH * CGN == RW * MP * NPASS && TH == RW * WROWS, "tile split over the waves of a role");
    constexpr int TAPX = 3, KS = (CD / 4) * 2 * TAPX;
    extern __shared__ __attribute__((aligned(16))) float fb_lds[];
    float *s_coefD = fb_lds + 2 * BUF, *s_coefT = s_coefD + DM_COEF_MAX_C * 4;
    double (*s_stat)[2] = reinterpret_cast<double (*)[2]>(s_coefT + DM_COEF_MAX_C * 4);

    const int lane = threadIdx.x & 63, m = lane & 15, kq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool loader = wave >= RW;                         // (wave-uniform)
    const int rw = wave & (RW - 1);
    const int OH = 2 * H, OW = 2 * W;
    const int tiles_x = W / TW, tiles_y = H / TH;
    auto coords = [&](int t, int &tb, int &ty0, int &tx0) {
        tx0 = (t % tiles_x) * TW; t /= tiles_x;
        ty0 = (t % tiles_y) * TH; tb = t / tiles_y;
    };
    stage_coef(s_coefD, dy, 0, CD);
    stage_coef(s_coefT, tin, 0, CX);
    int tidx = blockIdx.x;

    if (loader) {
        // ================================================================ waves 4..7: loads, commits, weight gradient
        TileStage<CD, IH, COLS4, RS, PS, true, 256> stD;
        TileStage<CX, TROWS, TCOLS4, RST, PST, false, 256> stT;
        const int tl = (int)threadIdx.x - 256;
        stD.init(H, W, tl);
        stT.init(OH, OW, tl);
        int b, y0, x0;
        if (tidx < ntiles) {
            coords(tidx, b, y0, x0);
            stD.issue(dy, b, CD, H, W, y0 - 1, x0 - 4);
            stT.issue(tin, b, CX, OH, OW, 2 * y0 - 1, 2 * x0 - 4);
        }
        int bl[NTT];
#pragma unroll
        for (int t = 0; t < NTT; ++t) {
            const int n = 16 * t + m;
            bl[t] = CD * PS + (n >> 4) * PST + ((n >> 2) & 3) * RST + (n & 3) + 3 + 8 * kq + 2 * rw * RST;
        }
        const int al = m * PS + (rw + 1) * RS + 4 + 4 * kq;
        f32x4 wacc[NTT];
#pragma unroll
        for (int t = 0; t < NTT; ++t) wacc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        __syncthreads();                                    // coefficient tables staged
        if (tidx < ntiles) {
            stD.template commit<BF>(fb_lds, s_coefD, CD, H, W, y0 - 1, x0 - 4, dy.mode);
            stT.template commit<BF>(fb_lds + CD * PS, s_coefT, CX, OH, OW, 2 * y0 - 1, 2 * x0 - 4, tin.mode);
        }
        __syncthreads();                                    // tile 0 in buffer 0
        int p = 0;
        while (tidx < ntiles) {
            const int next = tidx + gridDim.x;
            const float *cur = fb_lds + p * BUF;
            float *nxt = fb_lds + (1 - p) * BUF;
            int nb = 0, ny0 = 0, nx0 = 0;
            if (next < ntiles && !(dbg & 4)) {              // (uniform) every load of the next tile, now
                coords(next, nb, ny0, nx0);
                stD.issue(dy, nb, CD, H, W, ny0 - 1, nx0 - 4);
                stT.issue(tin, nb, CX, OH, OW, 2 * ny0 - 1, 2 * nx0 - 4);
            }
            // ---- weight gradient of the current tile: position rows rw, rw + 4; spans of 16 positions; 4 K-steps per span
            if (!(dbg & 1)) {
                constexpr int NSPAN = TW / 16, NQ = WROWS * NSPAN * 4;
                auto aoff = [](int q) { return (q / (NSPAN * 4)) * RW * RS + 16 * ((q >> 2) % NSPAN); };
                auto boff = [](int q) { return (q / (NSPAN * 4)) * RW * 2 * RST + 32 * ((q >> 2) % NSPAN) + 2 * (q & 3); };
                if constexpr (BF) {
                    // a span of 16 positions = four K-steps = ONE operand: A from one 16-byte read, B four 4-byte reads per
                    // N tile; units of (span, half of the N tiles), the next unit's operands requested before this one's products
                    constexpr int NU = WROWS * NSPAN * 2, HT = NTT / 2;
                    f32x4 av[2];
                    float bv[2][HT][4];
                    av[0] = *reinterpret_cast<const f32x4 *>(cur + al);
#pragma unroll
                    for (int t = 0; t < HT; ++t)
#pragma unroll
                        for (int j = 0; j < 4; ++j) bv[0][t][j] = cur[bl[t] + boff(j)];
#pragma unroll
                    for (int u = 0; u < NU; ++u) {
                        const int sp = u >> 1, half = u & 1;
                        if (u + 1 < NU) {
                            const int sp1 = (u + 1) >> 1, half1 = (u + 1) & 1;
                            if (half1 == 0) av[sp1 & 1] = *reinterpret_cast<const f32x4 *>(cur + al + aoff(4 * sp1));
#pragma unroll
                            for (int t = 0; t < HT; ++t)
#pragma unroll
                                for (int j = 0; j < 4; ++j) bv[(u + 1) & 1][t][j] = cur[bl[half1 * HT + t] + boff(4 * sp1 + j)];
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        const dm_u32x4_t a4 = __builtin_bit_cast(dm_u32x4_t, av[sp & 1]);
                        const dm_u32x4_t ar = dm_rot16(a4);
#pragma unroll
                        for (int t = 0; t < HT; ++t) {
                            const float(&b)[4] = bv[u & 1][t];
                            const dm_u32x4_t b4 = {__builtin_bit_cast(unsigned, b[0]), __builtin_bit_cast(unsigned, b[1]),
                                                   __builtin_bit_cast(unsigned, b[2]), __builtin_bit_cast(unsigned, b[3])};
                            wacc[half * HT + t] = dm_mfma_split(a4, ar, b4, wacc[half * HT + t]);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                } else {
                f32x4 av[2];
                float bv[2][NTT];
                av[0] = *reinterpret_cast<const f32x4 *>(cur + al);
#pragma unroll
                for (int t = 0; t < NTT; ++t) bv[0][t] = cur[bl[t]];
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    if (q + 1 < NQ) {
                        if (((q + 1) & 3) == 0) av[((q + 1) >> 2) & 1] = *reinterpret_cast<const f32x4 *>(cur + al + aoff(q + 1));
#pragma unroll
                        for (int t = 0; t < NTT; ++t) bv[(q + 1) & 1][t] = cur[bl[t] + boff(q + 1)];
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    const float a = av[(q >> 2) & 1][q & 3];
#pragma unroll
                    for (int t = 0; t < NTT; ++t) wacc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bv[q & 1][t], wacc[t], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
                }
            }
            if (next < ntiles && !(dbg & 4)) {              // the next tile into the other buffer (nobody reads it yet)
                stD.template commit<BF>(nxt, s_coefD, CD, H, W, ny0 - 1, nx0 - 4, dy.mode);
                stT.template commit<BF>(nxt + CD * PS, s_coefT, CX, OH, OW, 2 * ny0 - 1, 2 * nx0 - 4, tin.mode);
            }
            __syncthreads();                                // tile i consumed by everybody, tile i+1 complete
            p ^= 1;
            tidx = next;
        }
        // ---- weight-gradient slab: the four loader waves in wave order (buffer 0 is free: the last barrier is behind us)
        float *red = fb_lds;
        for (int w = 0; w < RW; ++w) {
            __syncthreads();
            if (rw == w) {
#pragma unroll
                for (int t = 0; t < NTT; ++t) {
                    f32x4 *pp = reinterpret_cast<f32x4 *>(red + (t * 64 + lane) * 4);
                    if (w == 0) *pp = wacc[t];
                    else *pp = *pp + wacc[t];
                }
            }
        }
        __syncthreads();
        __syncthreads();                                    // (the statistics write-out of the other role)
    } else {
        // ================================================================ waves 0..3: data gradient
        const int co = m & 7, pxl = m >> 3;
        constexpr int NTW = ZF ? 2 : 1, KSW = ZF ? (CD / 4) * 2 : KS;       // ZF: N tiles (centre, side) of 8 K-steps (cg4, a)
        float wreg[2][NTW][KSW];
#pragma unroll
        for (int py = 0; py < 2; ++py)
#pragma unroll
            for (int t = 0; t < NTW; ++t)
#pragma unroll
                for (int s = 0; s < KSW; ++s) {
                    int cg4, a, kx;
                    if constexpr (ZF) { cg4 = s >> 1; a = s & 1; kx = t == 0 ? 1 + pxl : (pxl ? 0 : 3); }
                    else { cg4 = s / (2 * TAPX); const int j = s % (2 * TAPX); a = j / TAPX; kx = pxl + 3 - 2 * (j % TAPX); }
                    const int c = 4 * cg4 + kq;
                    const int ky = py + 3 - 2 * (py + a);
                    float wvl = 0.f;
                    if (ky >= 0 && ky <= 3 && kx >= 0 && kx <= 3) wvl = wv.w[wv.off + co * wv.sn + c * wv.sc + ky * wv.sky + kx * wv.skx];
                    wreg[py][t][s] = BF ? split_pack1(wvl) : wvl;
                }
        float mc0, mc2;
        mask_coef(ep, 0, co, mc0, mc2);
        double s1 = 0.0, s2 = 0.0;
        const int abase = kq * PS + m + 3;
        EpiCtx<SIDE_MASK> cx;
        const long long sample_elems = (long long)CX * OH * OW;
        const int chan_off = (co * OH * OW + 8 * kq + 4 * pxl) * 4;
        // ZF: the lane that holds the neighbouring M row of this lane's first (px = 0) or last (px = 1) row: 16 lanes down or up
        const int nb_addr = ((pxl ? lane + 16 : lane - 16) & 63) * 4;
        const bool grp_first = kq == 0, grp_last = kq == 3;
        __syncthreads();                                    // (coefficient tables)
        __syncthreads();                                    // tile 0 in buffer 0
        int p = 0;
        while (tidx < ntiles) {
            int cb, cy0, cx0;
            coords(tidx, cb, cy0, cx0);
            const float *cur = fb_lds + p * BUF;
            cx.rebase(ep, dx, sample_elems, cb);
            if (!(dbg & 2)) {
#pragma unroll
            for (int pass = 0; pass < NPASS; ++pass) {
                const float *ap[MP];
                int obase[MP];
#pragma unroll
                for (int i = 0; i < MP; ++i) {
                    // ZF: a wave takes BOTH 16-position spans of a row (the side accumulators of neighbouring spans meet in
                    // its registers); else M tile rw + RW * (MP * pass + i)
                    const int ti = rw + RW * (MP * pass + i);
                    const int r = ZF ? rw + RW * pass : ti / CGN, cg = ZF ? i : ti % CGN;
                    ap[i] = cur + r * RS + 16 * cg + abase;
                    obase[i] = chan_off + (2 * (cy0 + r) * OW + 2 * (cx0 + 16 * cg)) * 4;
                }
#pragma unroll
                for (int py = 0; py < 2; ++py) {
                    EpiIn<SIDE_MASK> e[MP];
#pragma unroll
                    for (int i = 0; i < MP; ++i) epilogue_loads<SIDE_MASK>(e[i], cx, obase[i] + py * OW * 4);
                    f32x4 acc[MP][NTW];
#pragma unroll
                    for (int i = 0; i < MP; ++i)
#pragma unroll
                        for (int t = 0; t < NTW; ++t) acc[i][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
                    auto off = [py](int s) {
                        if constexpr (ZF) return 4 * (s >> 1) * PS + (py + (s & 1)) * RS + 1;
                        else {
                            const int cg4 = s / (2 * TAPX), j = s % (2 * TAPX), a = j / TAPX, bb = j % TAPX;
                            return 4 * cg4 * PS + (py + a) * RS + bb;
                        }
                    };
                    if constexpr (BF) mfma_tiles_split<MP, NTW, KSW>(ap, wreg[py], acc, off);
                    else mfma_tiles<MP, NTW, KSW, ZF ? 4 : TAPX * 2>(ap, wreg[py], acc, off);
                    f32x4 v[MP];
                    if constexpr (ZF) {
                        static_assert(MP == 2 && CGN == 2, "the two spans of a row in one wave");
                        // row P of the side product goes to row P + 1 (px = 0) or P - 1 (px = 1) of the result; M row =
                        // 16 i + 4 kq + register.  The one row per tile that crosses the lanes: the provider hands over its last
                        // (px = 0) or first (px = 1) register, the receiver reads 16 lanes down / up (wrapping into the other span)
                        float x[MP];
#pragma unroll
                        for (int i = 0; i < MP; ++i)
                            x[i] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(
                                       nb_addr, __builtin_bit_cast(int, pxl ? acc[i][1].x : acc[i][1].w)));
#pragma unroll
                        for (int i = 0; i < MP; ++i) {
                            const f32x4 c = acc[i][0], sd = acc[i][1];
                            // px = 0: first lane group of span 0 meets column -1 (zero), of span 1 the last group of span 0
                            // px = 1: last lane group of span 1 meets column W (zero), of span 0 the first group of span 1
                            const float e0 = grp_first ? (i == 0 ? 0.f : x[0]) : x[i];
                            const float e1 = grp_last ? (i == MP - 1 ? 0.f : x[MP - 1]) : x[i];
                            v[i] = pxl ? (f32x4){c.x + sd.y, c.y + sd.z, c.z + sd.w, c.w + e1}
                                       : (f32x4){c.x + e0, c.y + sd.x, c.z + sd.y, c.w + sd.z};
                        }
                    } else {
#pragma unroll
                        for (int i = 0; i < MP; ++i) v[i] = acc[i][0];
                    }
#pragma unroll
                    for (int i = 0; i < MP; ++i) {
                        const f32x4 pv = lane_xor8(v[i]);
                        epilogue_tail<SIDE_MASK>(pxl ? (f32x4){pv.z, v[i].z, pv.w, v[i].w} : (f32x4){v[i].x, pv.x, v[i].y, pv.y}, ep, cx,
                                                 e[i], mc0, mc2, obase[i] + py * OW * 4, s1, s2);
                    }
                }
            }
            }
            __syncthreads();
            p ^= 1;
            tidx += gridDim.x;
        }
        // (the loaders combine their accumulators through LDS: RW + 1 barriers)
        for (int w = 0; w < RW + 1; ++w) __syncthreads();
        if (ep.stats) {
            double a = s1, c = s2;
            a += __shfl_xor(a, 16, 64); c += __shfl_xor(c, 16, 64);
            a += __shfl_xor(a, 32, 64); c += __shfl_xor(c, 32, 64);
            a += __shfl_xor(a, 8, 64); c += __shfl_xor(c, 8, 64);
            if (lane < CX) { s_stat[rw * 16 + lane][0] = a; s_stat[rw * 16 + lane][1] = c; }
        }
        __syncthreads();
        if (ep.stats && threadIdx.x < CX) {
            double ta = 0.0, tc = 0.0;
#pragma unroll
            for (int w = 0; w < RW; ++w) { ta += s_stat[w * 16 + threadIdx.x][0]; tc += s_stat[w * 16 + threadIdx.x][1]; }
            ep.stats[((long long)blockIdx.x * CX + threadIdx.x) * 2 + 0] = ta;
            ep.stats[((long long)blockIdx.x * CX + threadIdx.x) * 2 + 1] = tc;
        }
    }
    // (barriers after the tile loop: RW + 2 on either side)
    // ---- all 512 threads: this workgroup's weight-gradient slab out of LDS.  red[(t * 64 + l) * 4 + j] is dy channel
    //      4 (l >> 4) + j, column 16 t + (l & 15)
    float *slab = wslabs + (long long)blockIdx.x * (CD * N);
    for (int i = threadIdx.x; i < NTT * 256; i += 512) {
        const int j = i & 3, l = (i >> 2) & 63, t = i >> 8;
        slab[(4 * (l >> 4) + j) * N + 16 * t + (l & 15)] = fb_lds[i];
    }
}

// ---- kernel D with three kinds of waves (round 5): 256 (2 + NDG) threads = 2 + NDG waves per SIMD, one barrier per tile as before.
//   NDG groups of 4 waves   data gradient of tile i (products + epilogue); two groups: four rows of the tile each
//   4 waves                 loads of tile i+1, BatchNorm-backward / ReLU transform, LDS writes into the other buffer -- vector work only
//   4 waves                 weight-gradient products of tile i
// In the two-role form a SIMD hosts two matrix-heavy waves and its port idles 37 % of the time (both waiting at once: LDS
// reads behind the barrier, the loader's global loads, the epilogue's stores); the third wave's transform and commit fill
// those gaps.  Every role stays below 168 registers: the staging registers (100) and the weight-gradient accumulators no
// longer meet in one wave.
template <int CD, int CX, int TH, int TW, bool BF, bool ZF, int NDG>
__global__ __launch_bounds__(256 * (2 + NDG), 1)
void bwd_s2_roles3_kernel(Operand dy, Operand tin, WeightView wv, float *__restrict__ dx, Epilogue ep,
                         float *__restrict__ wslabs, int H, int W, int ntiles, int dbg)
{
    // dbg (DM_FUSED_BWD_DBG, measurements only; results are then wrong): 1 skips the weight-gradient products, 2 the data
    // gradient, 4 the loads and commits of every tile but the first
    using G = FusedBwdGeom<CD, CX, TH, TW>;
    constexpr int RW = 4;                                   // waves per role
    constexpr int IH = G::IH, RS = G::RS, COLS4 = G::COLS4, PS = G::PS;
    constexpr int TROWS = G::TROWS, RST = G::RST, TCOLS4 = G::TCOLS4, PST = G::PST, NTT = G::NTT, N = G::N;
    constexpr int BUF = CD * PS + CX * PST;                 // floats of one (da, T) tile pair
    constexpr int CGN = TW / 16, MP = ZF ? CGN : 2, NPASS = TH * CGN / (RW * MP), WROWS = TH / RW;     // ZF: a wave takes a whole row
    static_assert(TH * CGN == RW * MP * NPASS && TH == RW * WROWS, "tile split over the waves of a role");
    constexpr int TAPX = 3, KS = (CD / 4) * 2 * TAPX;
    extern __shared__ __attribute__((aligned(16))) float fb_lds[];
    float *s_coefD = fb_lds + 2 * BUF, *s_coefT = s_coefD + DM_COEF_MAX_C * 4;
    double (*s_stat)[2] = reinterpret_cast<double (*)[2]>(s_coefT + DM_COEF_MAX_C * 4);

    const int lane = threadIdx.x & 63, m = lane & 15, kq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    static_assert(NDG == 1 || NDG == 2, "one or two groups of data-gradient waves");
    const int grp = wave / RW;                              // (wave-uniform) < NDG: data gradient, NDG: loads + commit, NDG + 1: weight gradient
    const int role = grp < NDG ? 0 : grp - NDG + 1;
    const int rw = wave & (RW - 1);
    const int OH = 2 * H, OW = 2 * W;
    const int tiles_x = W / TW, tiles_y = H / TH;
    auto coords = [&](int t, int &tb, int &ty0, int &tx0) {
        tx0 = (t % tiles_x) * TW; t /= tiles_x;
        ty0 = (t % tiles_y) * TH; tb = t / tiles_y;
    };
    stage_coef(s_coefD, dy, 0, CD);
    stage_coef(s_coefT, tin, 0, CX);
    int tidx = blockIdx.x;

    if (role == 1) {
        // ================================================================ waves 4..7: loads, transform, commits
        TileStage<CD, IH, COLS4, RS, PS, true, 256> stD;
        TileStage<CX, TROWS, TCOLS4, RST, PST, false, 256> stT;
        const int tl = (int)threadIdx.x - 256 * NDG;
        stD.init(H, W, tl);
        stT.init(OH, OW, tl);
        int b, y0, x0;
        if (tidx < ntiles) {
            coords(tidx, b, y0, x0);
            stD.issue(dy, b, CD, H, W, y0 - 1, x0 - 4);
        }
        __syncthreads();                                    // coefficient tables staged
        if (tidx < ntiles) {
            stD.template commit<BF>(fb_lds, s_coefD, CD, H, W, y0 - 1, x0 - 4, dy.mode);
            stT.issue(tin, b, CX, OH, OW, 2 * y0 - 1, 2 * x0 - 4);
            stT.template commit<BF>(fb_lds + CD * PS, s_coefT, CX, OH, OW, 2 * y0 - 1, 2 * x0 - 4, tin.mode);
        }
        __syncthreads();                                    // tile 0 in buffer 0
        int p = 0;
        while (tidx < ntiles) {
            const int next = tidx + gridDim.x;
            float *nxt = fb_lds + (1 - p) * BUF;
            int nb = 0, ny0 = 0, nx0 = 0;
            if (next < ntiles && !(dbg & 4)) {              // (uniform) the next tile into the other buffer (nobody reads it yet)
                // the two tensors one after the other: these waves have the tile's whole duration for two round trips, and the
                // staging registers of one tensor (56 / 44) fit where both (100) spilled
                coords(next, nb, ny0, nx0);
                stD.issue(dy, nb, CD, H, W, ny0 - 1, nx0 - 4);
                stD.template commit<BF>(nxt, s_coefD, CD, H, W, ny0 - 1, nx0 - 4, dy.mode);
                stT.issue(tin, nb, CX, OH, OW, 2 * ny0 - 1, 2 * nx0 - 4);
                stT.template commit<BF>(nxt + CD * PS, s_coefT, CX, OH, OW, 2 * ny0 - 1, 2 * nx0 - 4, tin.mode);
            }
            __syncthreads();                                // tile i consumed by everybody, tile i+1 complete
            p ^= 1;
            tidx = next;
        }
        for (int w = 0; w < RW + 2; ++w) __syncthreads();      // (the other roles' slab combine and statistics)
    } else if (role == 2) {
        // ================================================================ waves 8..11: weight gradient
        int bl[NTT];
#pragma unroll
        for (int t = 0; t < NTT; ++t) {
            const int n = 16 * t + m;
            bl[t] = CD * PS + (n >> 4) * PST + ((n >> 2) & 3) * RST + (n & 3) + 3 + 8 * kq + 2 * rw * RST;
        }
        const int al = m * PS + (rw + 1) * RS + 4 + 4 * kq;
        f32x4 wacc[NTT];
#pragma unroll
        for (int t = 0; t < NTT; ++t) wacc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        __syncthreads();                                    // coefficient tables staged
        __syncthreads();                                    // tile 0 in buffer 0
        int p = 0;
        while (tidx < ntiles) {
            const float *cur = fb_lds + p * BUF;
            // ---- weight gradient of the current tile: position rows rw, rw + 4; spans of 16 positions; 4 K-steps per span
            if (!(dbg & 1)) {
                constexpr int NSPAN = TW / 16, NQ = WROWS * NSPAN * 4;
                auto aoff = [](int q) { return (q / (NSPAN * 4)) * RW * RS + 16 * ((q >> 2) % NSPAN); };
                auto boff = [](int q) { return (q / (NSPAN * 4)) * RW * 2 * RST + 32 * ((q >> 2) % NSPAN) + 2 * (q & 3); };
                if constexpr (BF) {
                    // a span of 16 positions = four K-steps = ONE operand: A from one 16-byte read, B four 4-byte reads per
                    // N tile; units of (span, half of the N tiles), the next unit's operands requested before this one's products
                    constexpr int NU = WROWS * NSPAN * 2, HT = NTT / 2;
                    f32x4 av[2];
                    float bv[2][HT][4];
                    av[0] = *reinterpret_cast<const f32x4 *>(cur + al);
#pragma unroll
                    for (int t = 0; t < HT; ++t)
#pragma unroll
                        for (int j = 0; j < 4; ++j) bv[0][t][j] = cur[bl[t] + boff(j)];
#pragma unroll
                    for (int u = 0; u < NU; ++u) {
                        const int sp = u >> 1, half = u & 1;
                        if (u + 1 < NU) {
                            const int sp1 = (u + 1) >> 1, half1 = (u + 1) & 1;
                            if (half1 == 0) av[sp1 & 1] = *reinterpret_cast<const f32x4 *>(cur + al + aoff(4 * sp1));
#pragma unroll
                            for (int t = 0; t < HT; ++t)
#pragma unroll
                                for (int j = 0; j < 4; ++j) bv[(u + 1) & 1][t][j] = cur[bl[half1 * HT + t] + boff(4 * sp1 + j)];
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        const dm_u32x4_t a4 = __builtin_bit_cast(dm_u32x4_t, av[sp & 1]);
                        const dm_u32x4_t ar = dm_rot16(a4);
#pragma unroll
                        for (int t = 0; t < HT; ++t) {
                            const float(&b)[4] = bv[u & 1][t];
                            const dm_u32x4_t b4 = {__builtin_bit_cast(unsigned, b[0]), __builtin_bit_cast(unsigned, b[1]),
                                                   __builtin_bit_cast(unsigned, b[2]), __builtin_bit_cast(unsigned, b[3])};
                            wacc[half * HT + t] = dm_mfma_split(a4, ar, b4, wacc[half * HT + t]);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                } else {
                f32x4 av[2];
                float bv[2][NTT];
                av[0] = *reinterpret_cast<const f32x4 *>(cur + al);
#pragma unroll
                for (int t = 0; t < NTT; ++t) bv[0][t] = cur[bl[t]];
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    if (q + 1 < NQ) {
                        if (((q + 1) & 3) == 0) av[((q + 1) >> 2) & 1] = *reinterpret_cast<const f32x4 *>(cur + al + aoff(q + 1));
#pragma unroll
                        for (int t = 0; t < NTT; ++t) bv[(q + 1) & 1][t] = cur[bl[t] + boff(q + 1)];
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    const float a = av[(q >> 2) & 1][q & 3];
#pragma unroll
                    for (int t = 0; t < NTT; ++t) wacc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bv[q & 1][t], wacc[t], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
                }
            }
            __syncthreads();                                // tile i consumed by everybody, tile i+1 complete
            p ^= 1;
            tidx += gridDim.x;
        }
        // ---- weight-gradient slab: the four weight-gradient waves in wave order (buffer 0 is free: the last barrier is behind us)
        float *red = fb_lds;
        for (int w = 0; w < RW; ++w) {
            __syncthreads();
            if (rw == w) {
#pragma unroll
                for (int t = 0; t < NTT; ++t) {
                    f32x4 *pp = reinterpret_cast<f32x4 *>(red + (t * 64 + lane) * 4);
                    if (w == 0) *pp = wacc[t];
                    else *pp = *pp + wacc[t];
                }
            }
        }
        __syncthreads();
        __syncthreads();                                    // (the statistics write-out of the other role)
    } else {
        // ================================================================ waves 0..3: data gradient
        const int co = m & 7, pxl = m >> 3;
        constexpr int NTW = ZF ? 2 : 1, KSW = ZF ? (CD / 4) * 2 : KS;       // ZF: N tiles (centre, side) of 8 K-steps (cg4, a)
        float wreg[2][NTW][KSW];
#pragma unroll
        for (int py = 0; py < 2; ++py)
#pragma unroll
            for (int t = 0; t < NTW; ++t)
#pragma unroll
                for (int s = 0; s < KSW; ++s) {
                    int cg4, a, kx;
                    if constexpr (ZF) { cg4 = s >> 1; a = s & 1; kx = t == 0 ? 1 + pxl : (pxl ? 0 : 3); }
                    else { cg4 = s / (2 * TAPX); const int j = s % (2 * TAPX); a = j / TAPX; kx = pxl + 3 - 2 * (j % TAPX); }
                    const int c = 4 * cg4 + kq;
                    const int ky = py + 3 - 2 * (py + a);
                    float wvl = 0.f;
                    if (ky >= 0 && ky <= 3 && kx >= 0 && kx <= 3) wvl = wv.w[wv.off + co * wv.sn + c * wv.sc + ky * wv.sky + kx * wv.skx];
                    wreg[py][t][s] = BF ? split_pack1(wvl) : wvl;
                }
        float mc0, mc2;
        mask_coef(ep, 0, co, mc0, mc2);
        double s1 = 0.0, s2 = 0.0;
        const int abase = kq * PS + m + 3;
        EpiCtx<SIDE_MASK> cx;
        const long long sample_elems = (long long)CX * OH * OW;
        const int chan_off = (co * OH * OW + 8 * kq + 4 * pxl) * 4;
        // ZF: the lane that holds the neighbouring M row of this lane's first (px = 0) or last (px = 1) row: 16 lanes down or up
        const int nb_addr = ((pxl ? lane + 16 : lane - 16) & 63) * 4;
        const bool grp_first = kq == 0, grp_last = kq == 3;
        __syncthreads();                                    // (coefficient tables)
        __syncthreads();                                    // tile 0 in buffer 0
        int p = 0;
        while (tidx < ntiles) {
            int cb, cy0, cx0;
            coords(tidx, cb, cy0, cx0);
            const float *cur = fb_lds + p * BUF;
            cx.rebase(ep, dx, sample_elems, cb);
            if (!(dbg & 2)) {
#pragma unroll
            for (int pass0 = 0; pass0 < NPASS / NDG; ++pass0) {
                const int pass = NDG == 1 ? pass0 : __builtin_amdgcn_readfirstlane(grp) + NDG * pass0;      // (two groups: one pass each)
                const float *ap[MP];
                int obase[MP];
#pragma unroll
                for (int i = 0; i < MP; ++i) {
                    // ZF: a wave takes BOTH 16-position spans of a row (the side accumulators of neighbouring spans meet in
                    // its registers); else M tile rw + RW * (MP * pass + i)
                    const int ti = rw + RW * (MP * pass + i);
                    const int r = ZF ? rw + RW * pass : ti / CGN, cg = ZF ? i : ti % CGN;
                    ap[i] = cur + r * RS + 16 * cg + abase;
                    obase[i] = chan_off + (2 * (cy0 + r) * OW + 2 * (cx0 + 16 * cg)) * 4;
                }
#pragma unroll
                for (int py = 0; py < 2; ++py) {
                    EpiIn<SIDE_MASK> e[MP];
#pragma unroll
                    for (int i = 0; i < MP; ++i) epilogue_loads<SIDE_MASK>(e[i], cx, obase[i] + py * OW * 4);
                    f32x4 acc[MP][NTW];
#pragma unroll
                    for (int i = 0; i < MP; ++i)
#pragma unroll
                        for (int t = 0; t < NTW; ++t) acc[i][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
                    auto off = [py](int s) {
                        if constexpr (ZF) return 4 * (s >> 1) * PS + (py + (s & 1)) * RS + 1;
                        else {
                            const int cg4 = s / (2 * TAPX), j = s % (2 * TAPX), a = j / TAPX, bb = j % TAPX;
                            return 4 * cg4 * PS + (py + a) * RS + bb;
                        }
                    };
                    if constexpr (BF) mfma_tiles_split<MP, NTW, KSW>(ap, wreg[py], acc, off);
                    else mfma_tiles<MP, NTW, KSW, ZF ? 4 : TAPX * 2>(ap, wreg[py], acc, off);
                    f32x4 v[MP];
                    if constexpr (ZF) {
                        static_assert(MP == CGN, "every span of a row in one wave");
                        // row P of the side product goes to row P + 1 (px = 0) or P - 1 (px = 1) of the result; M row =
                        // 16 i + 4 kq + register.  The one row per tile that crosses the lanes: the provider hands over its last
                        // (px = 0) or first (px = 1) register, the receiver reads 16 lanes down / up (wrapping into the other span)
                        float x[MP];
#pragma unroll
                        for (int i = 0; i < MP; ++i)
                            x[i] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(
                                       nb_addr, __builtin_bit_cast(int, pxl ? acc[i][1].x : acc[i][1].w)));
#pragma unroll
                        for (int i = 0; i < MP; ++i) {
                            const f32x4 c = acc[i][0], sd = acc[i][1];
                            // px = 0: first lane group of span 0 meets column -1 (zero), of span i > 0 the last group of span i - 1
                            // px = 1: last lane group of the last span meets column W (zero), of span i the first group of span i + 1
                            const float e0 = grp_first ? (i == 0 ? 0.f : x[i > 0 ? i - 1 : 0]) : x[i];
                            const float e1 = grp_last ? (i == MP - 1 ? 0.f : x[i < MP - 1 ? i + 1 : MP - 1]) : x[i];
                            v[i] = pxl ? (f32x4){c.x + sd.y, c.y + sd.z, c.z + sd.w, c.w + e1}
                                       : (f32x4){c.x + e0, c.y + sd.x, c.z + sd.y, c.w + sd.z};
                        }
                    } else {
#pragma unroll
                        for (int i = 0; i < MP; ++i) v[i] = acc[i][0];
                    }
#pragma unroll
                    for (int i = 0; i < MP; ++i) {
                        const f32x4 pv = lane_xor8(v[i]);
                        epilogue_tail<SIDE_MASK>(pxl ? (f32x4){pv.z, v[i].z, pv.w, v[i].w} : (f32x4){v[i].x, pv.x, v[i].y, pv.y}, ep, cx,
                                                 e[i], mc0, mc2, obase[i] + py * OW * 4, s1, s2);
                    }
                }
            }
            }
            __syncthreads();
            p ^= 1;
            tidx += gridDim.x;
        }
        // (the loaders combine their accumulators through LDS: RW + 1 barriers)
        for (int w = 0; w < RW + 1; ++w) __syncthreads();
        if (ep.stats) {
            double a = s1, c = s2;
            a += __shfl_xor(a, 16, 64); c += __shfl_xor(c, 16, 64);
            a += __shfl_xor(a, 32, 64); c += __shfl_xor(c, 32, 64);
            a += __shfl_xor(a, 8, 64); c += __shfl_xor(c, 8, 64);
            if (lane < CX) { s_stat[(grp * RW + rw) * 16 + lane][0] = a; s_stat[(grp * RW + rw) * 16 + lane][1] = c; }
        }
        __syncthreads();
        if (ep.stats && threadIdx.x < CX) {
            double ta = 0.0, tc = 0.0;
#pragma unroll
            for (int w = 0; w < RW * NDG; ++w) { ta += s_stat[w * 16 + threadIdx.x][0]; tc += s_stat[w * 16 + threadIdx.x][1]; }
            ep.stats[((long long)blockIdx.x * CX + threadIdx.x) * 2 + 0] = ta;
            ep.stats[((long long)blockIdx.x * CX + threadIdx.x) * 2 + 1] = tc;
        }
    }
    // (barriers after the tile loop: RW + 2 on either side)
    // ---- all 512 threads: this workgroup's weight-gradient slab out of LDS.  red[(t * 64 + l) * 4 + j] is dy channel
    //      4 (l >> 4) + j, column 16 t + (l & 15)
    float *slab = wslabs + (long long)blockIdx.x * (CD * N);
    for (int i = threadIdx.x; i < NTT * 256; i += 256 * (2 + NDG)) {
        const int j = i & 3, l = (i >> 2) & 63, t = i >> 8;
        slab[(4 * (l >> 4) + j) * N + 16 * t + (l & 15)] = fb_lds[i];
    }
}

// ------------------------------------------------------------------------------ dispatch
struct ConvArgs {
    Operand in; WeightView wv; float *out; Epilogue ep;
    int B, Cphys, CIN, NOUT, H, W, per_tile;
    hipStream_t stream;
};

int side_mode(const Epilogue &ep)
{
    if (!ep.mask.p0 && !ep.resid && !ep.stat_q) return SIDE_NONE;
    if (ep.mask.p0 && !ep.resid && (!ep.stat_q || ep.stat_q == ep.mask.p0)) return SIDE_MASK;
    return SIDE_ALL;
}

// persistent grid: as many workgroups as are co-resident, unless the caller needs one statistics slab
// per tile with real contents (per-sample BatchNorm statistics)
int conv_grid(int ntiles, int wgs_per_cu, int per_tile, int npass = 1)
{
    if (per_tile && npass > 1) return ntiles;              // multi-pass kernels write their slab once, at the end
    const int cap = 256 * wgs_per_cu;
    return ntiles < cap ? ntiles : cap;
}

// statistics slabs the caller allocated (dm_conv*_num_blocks): variant independent
int conv_slabs(int ntiles, int per_tile) { return per_tile ? ntiles : (ntiles < 256 * 3 ? ntiles : 256 * 3); }

constexpr int lds_wgs(int lds_bytes) { return (160 * 1024) / (lds_bytes + 3072); }
constexpr int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// Waves per SIMD the register allocator must leave room for (launch bound): 3 (<= 168 VGPRs) when the
// weights, the staged next tile and the epilogue side inputs plausibly fit, else 2 (<= 256) -- a spilled
// MFMA loop is far worse than one resident workgroup fewer.  Never more than the LDS admits.
constexpr int conv_wps(int lds_bytes, int weights, int tile_f4, bool two, int mp, int nt, int side)
{
    const int stage = ((tile_f4 + DM_BLOCK - 1) / DM_BLOCK) * (two ? 9 : 5);
    const int epi = mp * nt * (6 + (side == SIDE_ALL ? 12 : (side == SIDE_MASK ? 4 : 0)));
    const int est = weights + stage + epi + 48;
    return clampi(lds_wgs(lds_bytes), 1, est > 145 ? 2 : 3);
}

int conv4_tw(int CIN, int Wo)
{
    const int cap = CIN <= 5 ? 64 : (CIN <= 8 ? 32 : 16);
    return Wo < cap ? Wo : cap;
}

// the first convolution with output positions in pairs (kernel A2); DM_CONV4_PAIR=0 keeps kernel A for A/B runs
bool conv4_pair_on()
{
    static const bool off = [] { const char *e = getenv("DM_CONV4_PAIR"); return e && e[0] == '0'; }();
    return !off;
}

template <int CIN, int TW>
void launch_conv4(const ConvArgs &a)
{
    constexpr int TH = 8;
    constexpr int F4 = CIN * (2 * TH + 2) * ((2 * TW + 8) / 4), LDS = 16 * F4;
    const int ntiles = a.B * ((a.H / 2) / TH) * ((a.W / 2) / TW);
#define DM_L4(SIDE_)                                                                                              \
    {                                                                                                             \
        constexpr int WPS = conv_wps(LDS, CIN * 4, F4, false, 2, 1, SIDE_);                                       \
        if (DM_FWD_SPLIT(SIDE_ == SIDE_NONE)) {                                                                   \
            if (CIN <= 5 && a.ep.bias_border)                                                                     \
                hipLaunchKernelGGL((conv4x4s2_kernel<CIN, 1, TH, TW, SIDE_NONE, WPS, true, MEASURE_BF>),              \
                                   dim3(conv_grid(ntiles, WPS, a.per_tile)), dim3(DM_BLOCK), 0, a.stream, a.in, a.wv, \
                                   a.out, a.ep, a.Cphys, a.NOUT, a.H, a.W, ntiles, conv_slabs(ntiles, a.per_tile),    \
                                   a.per_tile);                                                                       \
            else                                                                                                  \
                hipLaunchKernelGGL((conv4x4s2_kernel<CIN, 1, TH, TW, SIDE_NONE, WPS, false, MEASURE_BF>),             \
                                   dim3(conv_grid(ntiles, WPS, a.per_tile)), dim3(DM_BLOCK), 0, a.stream, a.in, a.wv, \
                                   a.out, a.ep, a.Cphys, a.NOUT, a.H, a.W, ntiles, conv_slabs(ntiles, a.per_tile),    \
                                   a.per_tile);                                                                       \
        } else if (SIDE_ == SIDE_NONE && CIN <= 4 && TW == 64 && a.NOUT == 8 && a.ep.bias_border && !a.ep.relu && conv4_pair_on()) \
            hipLaunchKernelGGL((conv4x4s2_pair_kernel<CIN <= 4 ? CIN : 1, TH, WPS>),                                   \
                               dim3(conv_grid(ntiles, WPS, a.per_tile)), dim3(DM_BLOCK), 0, a.stream, a.in, a.wv,     \
                               a.out, a.ep, a.Cphys, a.H, a.W, ntiles, conv_slabs(ntiles, a.per_tile), a.per_tile);    \
        else if (SIDE_ == SIDE_NONE && CIN <= 5 && a.ep.bias_border)                                              \
            hipLaunchKernelGGL((conv4x4s2_kernel<CIN, 1, TH, TW, SIDE_NONE, WPS, true>),                              \
                               dim3(conv_grid(ntiles, WPS, a.per_tile)), dim3(DM_BLOCK), 0, a.stream, a.in, a.wv,     \
                               a.out, a.ep, a.Cphys, a.NOUT, a.H, a.W, ntiles, conv_slabs(ntiles, a.per_tile),        \
                               a.per_tile);                                                                           \
        else if (SIDE_ != SIDE_NONE && dm_backward_split_bf16())        /* a data gradient: split-bf16 operands */ \
            hipLaunchKernelGGL((conv4x4s2_kernel<CIN, 1, TH, TW, SIDE_, WPS, false, DM_BUILD_SPLIT_BF16 && SIDE_ != SIDE_NONE>),             \
                               dim3(conv_grid(ntiles, WPS, a.per_tile)), dim3(DM_BLOCK), 0, a.stream, a.in, a.wv,     \
                               a.out, a.ep, a.Cphys, a.NOUT, a.H, a.W, ntiles, conv_slabs(ntiles, a.per_tile),        \
                               a.per_tile);                                                                           \
        else                                                                                                      \
            hipLaunchKernelGGL((conv4x4s2_kernel<CIN, 1, TH, TW, SIDE_, WPS, false>),                                 \
                               dim3(conv_grid(ntiles, WPS, a.per_tile)), dim3(DM_BLOCK), 0, a.stream, a.in, a.wv,     \
                               a.out, a.ep, a.Cphys, a.NOUT, a.H, a.W, ntiles, conv_slabs(ntiles, a.per_tile),        \
                               a.per_tile);                                                                           \
    }
    switch (side_mode(a.ep)) {
    case SIDE_NONE: DM_L4(SIDE_NONE) break;
    case SIDE_MASK: DM_L4(SIDE_MASK) break;
    default: DM_L4(SIDE_ALL)
    }
#undef DM_L4
}

// (32 input channels: 16-wide tiles whatever the width, 16 channels: at most 32 -- the wider instantiations with side
//  inputs spill ~300 bytes per lane, and scratch reloads in the MFMA loop cost more than the narrower tile; 32-wide
//  tiles also keep the 16-channel transposed convolutions on the phase-decomposed kernel)
int conv3_tw(int W, int CIN)
{
    const int cap = CIN >= 32 ? 16 : (CIN >= 16 ? 32 : 64);
    return W < cap ? W : cap;
}
// 16-wide tiles take 16 rows (a whole 16x16 latent), except with 32 input channels (staging registers)
constexpr int conv3_th(int TW, int CIN) { return (TW == 16 && CIN < 32) ? 16 : 8; }

template <int CIN, int NT, int NPASS, int TAPS, bool PIX, int TW>
void launch_conv3(const ConvArgs &a)
{
    constexpr int TH = conv3_th(TW, CIN);
    constexpr int PADR = TAPS == 9 ? 1 : 0;
    constexpr int F4 = CIN * (TH + 2 * PADR) * ((TW + 8 * PADR) / 4), LDS = 16 * F4 + 2048;
    const int ntiles = a.B * (a.H / TH) * (a.W / TW);
    const int side = side_mode(a.ep);
    const bool two = a.in.mode == DM_LOAD_AFFINE2;
#define DM_L3(TWO_, SIDE_)                                                                                        \
    {                                                                                                             \
        constexpr int WPS = conv_wps(LDS, NT * (CIN / 4) * TAPS, F4, TWO_, NT == 1 ? 2 : 1, NT, SIDE_);          \
        constexpr bool GRAD = (TWO_ || SIDE_ != SIDE_NONE) && ((CIN / 4) * TAPS) % 4 == 0;   /* a data gradient */      \
        constexpr bool FWDK = ((CIN / 4) * TAPS) % 4 == 0;                                                        \
        if ((GRAD && dm_backward_split_bf16()) || DM_FWD_SPLIT(!GRAD && FWDK))                                    \
            hipLaunchKernelGGL((conv3x3_kernel<CIN, NT, NPASS, TAPS, PIX, TH, TW, TWO_, SIDE_, WPS, (DM_BUILD_SPLIT_BF16 && GRAD) || (MEASURE_BF && FWDK)>),           \
                               dim3(conv_grid(ntiles, WPS, a.per_tile, NPASS)), dim3(DM_BLOCK), 0, a.stream, a.in, a.wv, \
                               a.out, a.ep, a.Cphys, a.NOUT, a.H, a.W, ntiles, conv_slabs(ntiles, a.per_tile), a.per_tile); \
        else                                                                                                      \
        hipLaunchKernelGGL((conv3x3_kernel<CIN, NT, NPASS, TAPS, PIX, TH, TW, TWO_, SIDE_, WPS>),                 \
                           dim3(conv_grid(ntiles, WPS, a.per_tile, NPASS)), dim3(DM_BLOCK), 0, a.stream, a.in, a.wv, \
                           a.out, a.ep, a.Cphys, a.NOUT, a.H, a.W, ntiles, conv_slabs(ntiles, a.per_tile), a.per_tile); \
    }
    // built variants: forward (no side inputs), data gradients with a mask-only side input (with or without
    // the BatchNorm-backward AFFINE2 operand), and the fully general one (residual join)
    if (!two && side == SIDE_NONE) DM_L3(false, SIDE_NONE)
    else if (!two && side == SIDE_MASK) DM_L3(false, SIDE_MASK)
    else if (two && side != SIDE_ALL) DM_L3(true, SIDE_MASK)
    else DM_L3(true, SIDE_ALL)        // TWO = true with a NULL p1 is safe: tile.h skips the second load
#undef DM_L3
}

template <int CIN, int COUT, int TW>
void launch_convT_phase(const ConvArgs &a)
{
    constexpr int TH = conv3_th(TW, CIN);
    constexpr int F4 = CIN * (TH + 2) * ((TW + 8) / 4), LDS = 16 * F4 + 2048;
    constexpr int KSW = (CIN / 4) * (COUT == 16 ? 4 : 6) * (COUT == 16 ? 4 : 2);      // weight registers
    const int ntiles = a.B * (a.H / TH) * (a.W / TW);
    const int side = side_mode(a.ep);
    const bool two = a.in.mode == DM_LOAD_AFFINE2;
#define DM_LP(TWO_, SIDE_)                                                                                        \
    {                                                                                                             \
        constexpr int WPS = clampi(conv_wps(LDS, KSW, F4, TWO_, 2, COUT == 16 ? 2 : 1, SIDE_), 1, 2);             \
        constexpr bool GRAD = TWO_ || SIDE_ != SIDE_NONE;                  /* a data gradient */                     \
        if ((GRAD && dm_backward_split_bf16()) || DM_FWD_SPLIT(!GRAD))                                            \
            hipLaunchKernelGGL((convT_phase_kernel<CIN, COUT, TH, TW, TWO_, SIDE_, WPS, (DM_BUILD_SPLIT_BF16 && GRAD) || MEASURE_BF>),         \
                               dim3(conv_grid(ntiles, WPS, a.per_tile)), dim3(DM_BLOCK), 0, a.stream, a.in, a.wv,     \
                               a.out, a.ep, a.Cphys, a.H, a.W, ntiles, conv_slabs(ntiles, a.per_tile));               \
        else                                                                                                      \
        hipLaunchKernelGGL((convT_phase_kernel<CIN, COUT, TH, TW, TWO_, SIDE_, WPS>),                             \
                           dim3(conv_grid(ntiles, WPS, a.per_tile)), dim3(DM_BLOCK), 0, a.stream, a.in, a.wv,     \
                           a.out, a.ep, a.Cphys, a.H, a.W, ntiles, conv_slabs(ntiles, a.per_tile));               \
    }
    if (!two && side == SIDE_NONE) DM_LP(false, SIDE_NONE)
    else if (!two && side == SIDE_MASK) DM_LP(false, SIDE_MASK)
    else if (two && side != SIDE_ALL) DM_LP(true, SIDE_MASK)
    else DM_LP(true, SIDE_ALL)
#undef DM_LP
}

}  // namespace

// =============================================================================== C ABI
static int conv_common_checks(const char *who, const dm_operand *in, const dm_weight_view *w, float *out,
                              const dm_epilogue *ep, int B, int CIN, int NOUT, int H, int W)
{
    if (dm_check_operand(in, who)) return -1;
    DM_REQUIRE(w && w->w && out, "%s: NULL weight or output", who);
    DM_REQUIRE(B > 0 && CIN > 0 && NOUT > 0 && H > 0 && W > 0, "%s: bad shape", who);
    // element offsets are 32-bit ints (byte offsets only appear relative to a sample, through the buffer descriptors)
    DM_REQUIRE((long long)B * (CIN > NOUT ? CIN : NOUT) * H * W < (1LL << 31),
               "%s: tensor too large for 32-bit element offsets", who);
    DM_REQUIRE(CIN - (in->ones_channel ? 1 : 0) > 0, "%s: no physical input channel", who);
    DM_REQUIRE((long long)NOUT * H * W * 4 < DM_VOFF_NONE, "%s: one output sample must stay below 1 GiB", who);
    if (ep && ep->mask.p0 && dm_check_operand(&ep->mask, who)) return -1;
    DM_REQUIRE(!(ep && ep->mask.p0 && ep->mask.ones_channel), "%s: mask operand cannot have a ones channel", who);
    DM_REQUIRE(!(ep && ep->mask.p0 && ep->mask.mode != DM_LOAD_IDENT && ep->mask.mode != DM_LOAD_AFFINE),
               "%s: mask operand must be IDENT or AFFINE", who);
    return 0;
}

// tile width of the MFMA path for this shape, 0 when the shape is not tileable by it
static int conv4_fast_tw(int CIN, int NOUT, int H, int W)
{
    const int Wo = W / 2, Ho = H / 2;
    const int TW = conv4_tw(CIN, Wo);
    if (H % 16 || W % 32 || NOUT > 16 || (TW != 16 && TW != 32 && TW != 64) || Ho % 8 || Wo % TW) return 0;
    return TW;
}

// the register-resident kernels instantiated below (DM_C4 table)
static bool conv4_has_kernel(int CIN, int TW)
{
    if (CIN >= 1 && CIN <= 5) return TW == 64 || TW == 32 || TW == 16;
    if (CIN == 8) return TW == 32 || TW == 16;
    return CIN == 16 && TW == 16;
}

extern "C" int dm_conv4x4s2_num_blocks(int B, int CIN, int NOUT, int H, int W, int per_tile)
{
    if (B <= 0 || H <= 0 || W <= 0 || (H & 1) || (W & 1)) return -1;
    const int TW = conv4_fast_tw(CIN, NOUT, H, W);
    if (TW && conv4_has_kernel(CIN, TW)) return conv_slabs(B * ((H / 2) / 8) * ((W / 2) / TW), per_tile);
    if (dm_wide_conv_ok(0, H, W)) return dm_wide_conv_slabs(0, B, H, W, per_tile);
    return dm_generic_conv_slabs(B, per_tile);
}

// scratch the weight view should carry (dm_weight_view.scratch): 0 when a register-resident kernel takes the shape.
// fallback != 0: the caller knows that kernel cannot be used (AFFINE2 input, border-bias table with side inputs).
extern "C" int64_t dm_conv4x4s2_scratch_floats(int CIN, int NOUT, int H, int W, int fallback)
{
    if (CIN <= 0 || NOUT <= 0 || !dm_wide_conv_ok(0, H, W)) return 0;
    const int TW = conv4_fast_tw(CIN, NOUT, H, W);
    if (TW && conv4_has_kernel(CIN, TW) && !fallback) return 0;
    return dm_wide_conv_scratch_floats(0, CIN, NOUT, 16);
}

extern "C" int dm_conv4x4s2(const dm_operand *in, const dm_weight_view *w, float *out, const dm_epilogue *ep,
                            int B, int CIN, int NOUT, int H, int W, void *stream)
{
    if (conv_common_checks("dm_conv4x4s2", in, w, out, ep, B, CIN, NOUT, H, W)) return -1;
    DM_REQUIRE(H % 2 == 0 && W % 2 == 0, "dm_conv4x4s2: H and W must be even (got %dx%d)", H, W);
    ConvArgs a{to_dev(in), to_dev(w), out, to_dev(ep), B, CIN - (in->ones_channel ? 1 : 0), CIN, NOUT, H, W,
               ep ? ep->stats_per_tile : 0, (hipStream_t)stream};
    {   // enc.7's shape: the whole-patch kernel (conv4x4s2_patch.hip)
        int rc = 0;
        if (dm_conv4x4s2_patch_forward(a.in, a.wv, out, a.ep, B, a.Cphys, CIN, NOUT, H, W, a.per_tile,
                                       dm_conv4x4s2_num_blocks(B, CIN, NOUT, H, W, a.per_tile), a.stream, &rc))
            return rc;
    }
    const int TW = conv4_fast_tw(CIN, NOUT, H, W);
    const bool border_ok = !(ep && ep->bias_border) || (CIN <= 5 && !ep->mask.p0 && !ep->resid && !ep->stat_q && W / 2 >= 8);
    if (TW && conv4_has_kernel(CIN, TW) && in->mode != DM_LOAD_AFFINE2 && border_ok) {
#define DM_C4(C, T) if (CIN == C && TW == T) { launch_conv4<C, T>(a); return dm_launch_status("dm_conv4x4s2"); }
        DM_C4(3, 64) DM_C4(3, 32) DM_C4(3, 16)
        DM_C4(4, 64) DM_C4(4, 32) DM_C4(4, 16)
        DM_C4(5, 64) DM_C4(5, 32) DM_C4(5, 16)
        DM_C4(2, 64) DM_C4(2, 32) DM_C4(2, 16)
        DM_C4(1, 64) DM_C4(1, 32) DM_C4(1, 16)
        DM_C4(8, 32) DM_C4(8, 16)
        DM_C4(16, 16)
#undef DM_C4
    }
    // no register-resident instantiation for this channel count / shape / operand mode: the implicit-GEMM kernel
    // (conv_wide.hip) when the output tiles by 8 x 16, else the generic kernel (conv_generic.hip)
    const int nslabs = dm_conv4x4s2_num_blocks(B, CIN, NOUT, H, W, a.per_tile);
    if (dm_wide_conv_ok(0, H, W) && w->scratch && w->scratch_floats >= dm_wide_conv_scratch_floats(0, CIN, NOUT, 16))
        dm_wide_conv(0, a.in, a.wv, w->scratch, out, a.ep, B, a.Cphys, CIN, NOUT, H, W, 16, nslabs, a.per_tile, a.stream);
    else
        dm_generic_conv(0, a.in, a.wv, out, a.ep, B, a.Cphys, CIN, NOUT, H, W, 16, nslabs, a.per_tile, a.stream);
    return dm_launch_status("dm_conv4x4s2");
}

static bool conv3_fast_tileable(int CIN, int H, int W)
{
    const int TW = conv3_tw(W, CIN), TH = conv3_th(TW, CIN);
    return (TW == 16 || TW == 32 || TW == 64) && W % TW == 0 && H % TH == 0 && CIN % 4 == 0 && CIN <= 32;
}

// the register-resident kernels instantiated below (DM_CP / DM_C3 tables)
static bool conv3_has_kernel(int CIN, int NOUT, int H, int W, int taps, bool pix, int per_tile)
{
    if (!conv3_fast_tileable(CIN, H, W)) return false;
    const int TW = conv3_tw(W, CIN), NTT = (NOUT + 15) / 16;
    const bool t13 = TW == 16 || TW == 32;
    if (pix && CIN == 16 && (NOUT == 32 || NOUT == 64) && t13 && !per_tile) return true;
    if (!pix && taps == 9) return t13 && ((CIN == 16 && (NTT == 1 || NTT == 2)) || (CIN == 32 && NTT == 1));
    if (!pix && taps == 1) return t13 && ((CIN == 32 && NTT == 1) || (CIN == 16 && NTT == 2));
    if (pix && taps == 9) {
        if (CIN == 16 && NTT == 2) return true;
        if (CIN == 16 && NTT == 4) return t13;
        if (CIN == 8 && NTT == 1) return TW == 32 || TW == 64;
        if (CIN == 4 && NTT == 1) return TW == 64;
    }
    return false;
}

extern "C" int dm_conv3x3_num_blocks(int B, int CIN, int NOUT, int H, int W, int taps, int pixel_shuffle, int per_tile)
{
    if (B <= 0 || H <= 0 || W <= 0) return -1;
    if (conv3_has_kernel(CIN, NOUT, H, W, taps, pixel_shuffle != 0, per_tile)) {
        const int TW = conv3_tw(W, CIN), TH = conv3_th(TW, CIN);
        return conv_slabs(B * (H / TH) * (W / TW), per_tile);
    }
    const int form = pixel_shuffle ? 2 : 1;
    if (dm_wide_conv_ok(form, H, W)) return dm_wide_conv_slabs(form, B, H, W, per_tile);
    return dm_generic_conv_slabs(B, per_tile);
}

extern "C" int64_t dm_conv3x3_scratch_floats(int CIN, int NOUT, int H, int W, int taps, int pixel_shuffle, int per_tile)
{
    const int form = pixel_shuffle ? 2 : 1;
    if (CIN <= 0 || NOUT <= 0 || (taps != 9 && taps != 1) || !dm_wide_conv_ok(form, H, W)) return 0;
    if (conv3_has_kernel(CIN, NOUT, H, W, taps, pixel_shuffle != 0, per_tile)) return 0;
    return dm_wide_conv_scratch_floats(form, CIN, NOUT, taps);
}

extern "C" int dm_conv3x3(const dm_operand *in, const dm_weight_view *w, float *out, const dm_epilogue *ep,
                          int B, int CIN, int NOUT, int H, int W, int taps, int pixel_shuffle, void *stream)
{
    if (conv_common_checks("dm_conv3x3", in, w, out, ep, B, CIN, NOUT, H, W)) return -1;
    DM_REQUIRE(taps == 9 || taps == 1, "dm_conv3x3: taps must be 9 or 1");
    DM_REQUIRE(!pixel_shuffle || (taps == 9 && NOUT % 4 == 0), "dm_conv3x3: pixel_shuffle needs taps=9, NOUT%%4==0");
    DM_REQUIRE(!in->ones_channel, "dm_conv3x3: ones_channel not supported");
    const int TW = conv3_tw(W, CIN);
    const bool fast = conv3_has_kernel(CIN, NOUT, H, W, taps, pixel_shuffle != 0, ep ? ep->stats_per_tile : 0);
    ConvArgs a{to_dev(in), to_dev(w), out, to_dev(ep), B, CIN, CIN, NOUT, H, W, ep ? ep->stats_per_tile : 0,
               (hipStream_t)stream};
    const int NTT = (NOUT + 15) / 16;
    const bool pix = pixel_shuffle != 0;
    if (fast) {
    // ConvTranspose2d with 16 input and 8 / 16 output channels (dec.0, data gradients of enc.4 / enc.7): phase-decomposed
    // kernel C; per-tile statistics slabs (per-sample BatchNorm) stay with the neighbourhood kernel B
#define DM_CP(CO, T)                                                                       \
    if (pix && CIN == 16 && NOUT == 4 * CO && TW == T && !a.per_tile) {                    \
        launch_convT_phase<16, CO, T>(a);                                                  \
        return dm_launch_status("dm_conv3x3");                                             \
    }
    DM_CP(8, 16) DM_CP(8, 32) DM_CP(16, 16) DM_CP(16, 32)
#undef DM_CP
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
    }
    // no register-resident instantiation for this channel count / shape: the implicit-GEMM kernel (conv_wide.hip)
    // when the base grid tiles by 8 x 16, else the generic kernel (conv_generic.hip)
    const int nslabs = dm_conv3x3_num_blocks(B, CIN, NOUT, H, W, taps, pixel_shuffle, a.per_tile);
    if (fast) {
        dm_set_error("dm_conv3x3: kernel table and conv3_has_kernel disagree (CIN %d NOUT %d %dx%d)", CIN, NOUT, H, W);
        return -1;
    }
    if (dm_wide_conv_ok(pix ? 2 : 1, H, W) && w->scratch &&
        w->scratch_floats >= dm_wide_conv_scratch_floats(pix ? 2 : 1, CIN, NOUT, taps))
        dm_wide_conv(pix ? 2 : 1, a.in, a.wv, w->scratch, out, a.ep, B, CIN, CIN, NOUT, H, W, taps, nslabs, a.per_tile, a.stream);
    else
        dm_generic_conv(pix ? 2 : 1, a.in, a.wv, out, a.ep, B, CIN, CIN, NOUT, H, W, taps, nslabs, a.per_tile, a.stream);
    return dm_launch_status("dm_conv3x3");
}


// ---- fused backward of the stride-2 encoder convolutions (kernel D) ------------------------------------------------------
static bool fused_bwd_shape(int CD, int CX, int H, int W)
{
    return CD == 16 && CX == 8 && H > 0 && W > 0 && H % 8 == 0 && W % 32 == 0;
}

extern "C" int dm_conv_bwd_s2_fused_supported(int CD, int CX, int H, int W) { return fused_bwd_shape(CD, CX, H, W) ? 1 : 0; }

// form of kernel D: 0 = role-split (default: one 512-thread workgroup per CU, two LDS buffers), 512 / 256 = the lockstep
// forms with that many threads per workgroup (DM_FUSED_BWD_BLOCK=512|256 in the environment: A/B measurements)
static int fused_bwd_block()
{
    static const int v = [] { const char *e = getenv("DM_FUSED_BWD_BLOCK"); const int n = e ? atoi(e) : 0; return (n == 512 || n == 256) ? n : 0; }();
    return v;
}

// Arithmetic of the BACKWARD matrix products (data and weight gradients): the f32-input instruction, bit for bit the fp32
// multiply-add chain -- every number the library produces is plain fp32 arithmetic.  The split-bf16 alternative of rounds
// 4-5 is retired (dm_common.h, DM_BUILD_SPLIT_BF16): asking for it is an error unless a measurement build instantiated it.
static int g_backward_split = -1;
bool dm_backward_split_bf16()
{
#if DM_BUILD_SPLIT_BF16
    if (g_backward_split < 0) {
        const char *e = getenv("DM_BACKWARD_PRECISION");
        g_backward_split = (e && e[0] == 's') ? 1 : 0;
        // an environment variable that changes results says so, once
        if (g_backward_split)
            fprintf(stderr, "libdynamorph_hip: DM_BACKWARD_PRECISION=%s -- gradient products run on split-bf16 operands "
                            "(~2^-17 relative per product, not the fp32 chain); forward pass and codes unchanged\n", e);
    }
    return g_backward_split != 0;
#else
    return false;
#endif
}
extern "C" int dm_backward_precision(int mode)
{
    const int cur = dm_backward_split_bf16() ? 1 : 0;
#if DM_BUILD_SPLIT_BF16
    if (mode == 0 || mode == 1) g_backward_split = mode;
#else
    (void)g_backward_split;
    DM_REQUIRE(mode != 1, "dm_backward_precision: the split-bf16 gradient kernels are not built (retired: slower than the "
                          "exact fp32 path on these layer widths; -DDM_BUILD_SPLIT_BF16=1 builds them for measurements)");
#endif
    return cur;
}

static int fused_bwd_dbg()
{
#ifdef DM_MEASURE      // ablation switches exist only in a measurement build (make measure): they make results wrong
    static const int v = [] { const char *e = getenv("DM_FUSED_BWD_DBG"); return e ? atoi(e) : 0; }();
#else
    static const int v = 0;
#endif
    return v;
}

extern "C" int dm_conv_bwd_s2_fused_num_blocks(int B, int CD, int CX, int H, int W)
{
    if (B <= 0 || !fused_bwd_shape(CD, CX, H, W)) return -1;
    const long long ntiles = (long long)B * (H / 8) * (W / 32);
    const long long cap = fused_bwd_block() == 256 ? 512 : 256;       // resident workgroups: one slab each
    return (int)(ntiles < cap ? ntiles : cap);
}

extern "C" int dm_conv_bwd_s2_fused(const dm_operand *dy, const dm_operand *tin, const dm_weight_view *w, float *dx,
                                    const dm_epilogue *ep, float *w_slabs, int B, int CD, int CX, int H, int W,
                                    void *stream)
{
    if (dm_check_operand(dy, "dm_conv_bwd_s2_fused(dy)") || dm_check_operand(tin, "dm_conv_bwd_s2_fused(in)")) return -1;
    DM_REQUIRE(w && w->w && dx && w_slabs && ep, "dm_conv_bwd_s2_fused: NULL pointer");
    DM_REQUIRE(fused_bwd_shape(CD, CX, H, W), "dm_conv_bwd_s2_fused: shape %d -> %d channels on %dx%d not built", CX, CD, H, W);
    DM_REQUIRE(B > 0 && (long long)B * CD * 4 * H * W < (1LL << 31), "dm_conv_bwd_s2_fused: tensor too large for 32-bit offsets");
    DM_REQUIRE(!dy->ones_channel && !tin->ones_channel && tin->mode != DM_LOAD_AFFINE2,
               "dm_conv_bwd_s2_fused: operand modes");
    DM_REQUIRE(dy->coef_bstride == 0 && tin->coef_bstride == 0, "dm_conv_bwd_s2_fused: batch-statistics coefficients only");
    // the ReLU mask of the data gradient and the second-moment partner are the layer input itself (SIDE_MASK of kernel C)
    DM_REQUIRE(ep->mask.p0 == tin->p0 && (!ep->stat_q || ep->stat_q == tin->p0) && !ep->resid && !ep->bias && !ep->relu &&
                   ep->mask.coef_bstride == 0,
               "dm_conv_bwd_s2_fused: the epilogue must mask by (and take its statistics against) the layer input");
    if (dm_check_operand(&ep->mask, "dm_conv_bwd_s2_fused(mask)")) return -1;
    using G = FusedBwdGeom<16, 8, 8, 32>;
    static DmPerDeviceOnce attr_set;
    if (attr_set.need()) {
        hipError_t e = hipFuncSetAttribute((const void *)bwd_s2_fused_kernel<16, 8, 8, 32, 512, true>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS_BYTES);
        if (e == hipSuccess)
            e = hipFuncSetAttribute((const void *)bwd_s2_fused_kernel<16, 8, 8, 32, 256, false>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS_BYTES);
#define DM_SPLIT_ATTR(BF_, ZF_)                                                                                       \
        if (e == hipSuccess)                                                                                         \
            e = hipFuncSetAttribute((const void *)bwd_s2_split_kernel<16, 8, 8, 32, DM_BUILD_SPLIT_BF16 && BF_, ZF_>,                       \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::SPLIT_LDS_BYTES);
        DM_SPLIT_ATTR(false, false) DM_SPLIT_ATTR(true, false) DM_SPLIT_ATTR(false, true) DM_SPLIT_ATTR(true, true)
#undef DM_SPLIT_ATTR
#define DM_ROLES3_ATTR(BF_, ZF_)                                                                                      \
        if (e == hipSuccess)                                                                                         \
            e = hipFuncSetAttribute((const void *)bwd_s2_roles3_kernel<16, 8, 8, 32, DM_BUILD_SPLIT_BF16 && BF_, ZF_, 1>,                   \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::SPLIT_LDS_BYTES);
        DM_ROLES3_ATTR(false, false) DM_ROLES3_ATTR(true, false) DM_ROLES3_ATTR(false, true) DM_ROLES3_ATTR(true, true)
#undef DM_ROLES3_ATTR
        if (e == hipSuccess)
            e = hipFuncSetAttribute((const void *)bwd_s2_roles3_kernel<16, 8, 4, 64, false, true, 1>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)FusedBwdGeom<16, 8, 4, 64>::SPLIT_LDS_BYTES);
        if (e != hipSuccess) { dm_set_error("dm_conv_bwd_s2_fused: cannot reserve %zu bytes of LDS: %s", G::LDS_BYTES, hipGetErrorString(e)); return (int)e; }
        attr_set.mark();
    }
    const int ntiles = B * (H / 8) * (W / 32);
    const int grid = dm_conv_bwd_s2_fused_num_blocks(B, CD, CX, H, W);
    // the data gradient without structural zeros where a tile spans the row (W == 32: enc.4 of 128-pixel patches);
    // DM_FUSED_BWD_ZF=0 keeps the three-column mapping for A/B runs
    static const bool zf_off = [] { const char *e = getenv("DM_FUSED_BWD_ZF"); return e && e[0] == '0'; }();
    const bool zf = W == 32 && !zf_off;
#define DM_SPLIT_LAUNCH(BF_, ZF_)                                                                                                  \
        hipLaunchKernelGGL((bwd_s2_split_kernel<16, 8, 8, 32, DM_BUILD_SPLIT_BF16 && BF_, ZF_>), dim3(grid), dim3(512), G::SPLIT_LDS_BYTES, (hipStream_t)stream, \
                           to_dev(dy), to_dev(tin), to_dev(w), dx, to_dev(ep), w_slabs, H, W, ntiles, fused_bwd_dbg())
    // three roles (768 threads) or two (512): DM_FUSED_BWD_ROLES=2 keeps the two-role form for A/B runs
    static const int roles = [] { const char *e = getenv("DM_FUSED_BWD_ROLES"); return e ? atoi(e) : 3; }();
    const bool roles3 = roles >= 3;
#define DM_ROLES3_LAUNCH(BF_, ZF_)                                                                                                 \
        /* (a second group of data-gradient waves, NDG = 2 / 1024 threads, measured slower: 240.6 against 227.4 us) */                 \
        { hipLaunchKernelGGL((bwd_s2_roles3_kernel<16, 8, 8, 32, DM_BUILD_SPLIT_BF16 && BF_, ZF_, 1>), dim3(grid), dim3(768), G::SPLIT_LDS_BYTES, (hipStream_t)stream, \
                               to_dev(dy), to_dev(tin), to_dev(w), dx, to_dev(ep), w_slabs, H, W, ntiles, fused_bwd_dbg()); }
    // 64-column grids (enc.4 of 256-pixel patches): tiles of 4 rows x 64 columns span the row, so the zero-free mapping applies
    // (the same number of tiles, hence of slabs, as 8 x 32)
    const bool zf64 = W == 64 && H % 4 == 0 && !zf_off && roles3 && fused_bwd_block() == 0 && !dm_backward_split_bf16();
    if (zf64) {
        using G64 = FusedBwdGeom<16, 8, 4, 64>;
        hipLaunchKernelGGL((bwd_s2_roles3_kernel<16, 8, 4, 64, false, true, 1>), dim3(grid), dim3(768), G64::SPLIT_LDS_BYTES,
                           (hipStream_t)stream, to_dev(dy), to_dev(tin), to_dev(w), dx, to_dev(ep), w_slabs, H, W, B * (H / 4),
                           fused_bwd_dbg());
    } else if (fused_bwd_block() == 0 && roles3) {
        const bool bf = dm_backward_split_bf16();
        if (bf && zf) DM_ROLES3_LAUNCH(true, true)
        else if (bf) DM_ROLES3_LAUNCH(true, false)
        else if (zf) DM_ROLES3_LAUNCH(false, true)
        else DM_ROLES3_LAUNCH(false, false)
    } else if (fused_bwd_block() == 0 && dm_backward_split_bf16()) {
        if (zf) DM_SPLIT_LAUNCH(true, true); else DM_SPLIT_LAUNCH(true, false);
    } else if (fused_bwd_block() == 0) {
        if (zf) DM_SPLIT_LAUNCH(false, true); else DM_SPLIT_LAUNCH(false, false);
    }
#undef DM_ROLES3_LAUNCH
#undef DM_SPLIT_LAUNCH
    else if (fused_bwd_block() == 512)
        hipLaunchKernelGGL((bwd_s2_fused_kernel<16, 8, 8, 32, 512, true>), dim3(grid), dim3(512), G::LDS_BYTES, (hipStream_t)stream,
                           to_dev(dy), to_dev(tin), to_dev(w), dx, to_dev(ep), w_slabs, H, W, ntiles);
    else
        hipLaunchKernelGGL((bwd_s2_fused_kernel<16, 8, 8, 32, 256, false>), dim3(grid), dim3(256), G::LDS_BYTES, (hipStream_t)stream,
                           to_dev(dy), to_dev(tin), to_dev(w), dx, to_dev(ep), w_slabs, H, W, ntiles);
    return dm_launch_status("dm_conv_bwd_s2_fused");
}

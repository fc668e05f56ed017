// conv_wide.hip -- MFMA convolution / weight-gradient kernels for arbitrary channel counts.
//
// conv_mfma.hip / wgrad_mfma.hip keep a layer's whole weight tensor in registers, which only works for the thin
// default family (num_hiddens 16).  The reference's example configuration (config_example.yml: VQ_VAE_z32 with
// num_hiddens 64, num_residual_hiddens 64, 512 codes) has 64 -> 64 channel 3x3 layers: 147 KB of weights, MFMA bound.
// These kernels are the classic implicit GEMM for that regime:
//   convolution     M = 8 x 16 pixels of one sample, N = up to 64 output channels per pass, K = (tap, channel) in chunks of
//                   8 channels; the input chunk (operand transform, zero padding, ones channel applied; rows of aligned
//                   float4 + halo scalars) and the weight chunk (re-laid once per call into caller scratch, copied as
//                   float4) are staged in LDS through registers, prefetched one chunk ahead; every wave owns 2 pixel
//                   rows x all N;
//   weight gradient M = 64 S channels (one 16-row tile per wave), N = (T channel, tap) flattened, K = the 128 pixels
//                   of a tile; both operand tiles in LDS, deterministic slabs as everywhere else.
// Same operands / epilogue / statistics-slab semantics as the thin kernels (dm_operand, dm_weight_view, dm_epilogue);
// the fp32 MFMA (v_mfma_f32_16x16x4_f32) keeps full precision.  ds_read_b32 / ds_write banks are dword address mod 32
// inside each 32-lane half: strides are chosen so that the two k lanes x 16 column lanes of a half hit 32 banks
// (HISTORY.md section 3a has the measurements behind each of these choices).
#include "dm_common.h"
#include <stdlib.h>
#include <type_traits>

namespace {

#ifndef DM_WIDE_OCC3
#define DM_WIDE_OCC3 0              // 1: every form compiled for three workgroups per CU (measurement builds)
#endif
enum { W_S2 = 0, W_S1 = 1, W_PIX = 2 };
constexpr int WKC = 8;                 // input channels per K chunk
constexpr int WIDE_MAX_BLOCKS = 768;   // persistent grid cap (x dimension)

// coefficient rows of channels [c_first, c_first + nch) of sample b -> s_cf[nch][4] (caller synchronises)
__device__ __forceinline__ void stage_coef(const Operand &op, float *s_cf, int nch, int c_first, int Cphys, int b, int tid, int nth = 256)
{
    if (op.mode < DM_LOAD_AFFINE) return;
    for (int i = tid; i < nch * 4; i += nth) {
        const int c = c_first + (i >> 2);
        s_cf[i] = c < Cphys ? op.coef[(long long)b * op.coef_bstride + c * 4 + (i & 3)] : 0.f;
    }
}

template <int FORM, int TAPS>
struct WideGeom {
    static constexpr int S = FORM == W_S2 ? 2 : 1;                       // stride
    static constexpr int R = FORM == W_S2 ? 1 : (TAPS == 9 ? 1 : 0);     // halo
    static constexpr int T = FORM == W_S1 ? TAPS : 16;                   // taps of the K loop (transposed: 4 parities x 2 x 2)
    static constexpr int ROWS = FORM == W_S2 ? 18 : 8 + 2 * R;           // input rows of a tile
    static constexpr int QW = FORM == W_S2 ? 8 : 4;                      // aligned float4 per input row (the interior)
    static constexpr bool HALO = R > 0;
    static constexpr bool PLANES = FORM == W_S2;
    // LDS image of one channel.  Plain forms: rows of RS floats, interior at column OFFC (16-byte aligned), halos at
    // OFFC-1 and OFFC+16.  Stride-2 form: two parity planes of 18 x 18 (odd window columns -> plane 1 at column wx>>1,
    // even ones -> plane 0 at column (wx>>1)+1), so that a stride-2 reader steps by 1 and the interior float4 of a row
    // lands as two aligned float2.
    static constexpr int RS = FORM == W_S2 ? 18 : (R ? 24 : 16);
    static constexpr int OFFC = (FORM != W_S2 && R) ? 4 : 0;
    static constexpr int PLS = ROWS * RS;
    static constexpr int RAW = FORM == W_S2 ? 2 * PLS : PLS;
    static constexpr int CHS = ((RAW - 16 + 31) / 32) * 32 + 16;         // channel stride = 16 (mod 32): see WideW
};

// Weight chunk in LDS: address = tap*W_TS + (cl >> 1)*W_PS + (cl & 1)*NS + n.  ds_read_b32 / ds_write_b32 bank = dword
// address mod 32 within each 32-lane half: the reader's two k lanes of a half (channels cl, cl+1) sit NS = 16 (mod 32)
// apart.  The same image is what dm_wide_pack writes to the caller's scratch, one block per (pass, chunk), so that the
// conv kernel stages a weight chunk as a straight float4 copy.
template <int FORM, int TAPS, int NPW>
struct WideW {
    static constexpr int T = FORM == W_S1 ? TAPS : 16;
    static constexpr int NS = NPW == 1 ? 16 : 16 * NPW + 16;
    static constexpr int W_PS = 2 * NS + 9;
    static constexpr int W_TS = ((WKC / 2) * W_PS + 30) / 32 * 32 + 1;
    static constexpr int BLK = (T * W_TS + 3) / 4 * 4;                   // floats per block
    static constexpr int NPASS = 16 * NPW;
};

// scratch[(pass*nchunks + chunk)*BLK + ...] <- the weight view's values in LDS layout (0 past CIN / the channel count).
// Transposed form: the N dimension is the output channel co (NOUT / 4 of them); the 16 "taps" are (output parity
// (py, px), a, b): output (2y+py, 2x+px) takes input (y-1+py+a, x-1+px+b) through kernel element (3-py-2a, 3-px-2b) --
// the 4 taps of the 4x4 kernel that reach that parity, instead of a 3x3 neighbourhood with 5 structural zeros.
// grid (nchunks, passes).
template <int FORM, int TAPS, int NPW>
__global__ __launch_bounds__(256) void wide_pack_kernel(WeightView wv, float *__restrict__ scratch, int CIN, int NOUT)
{
    using P = WideW<FORM, TAPS, NPW>;
    const int c0 = blockIdx.x * WKC, n0 = blockIdx.y * P::NPASS;
    float *dst = scratch + ((long long)blockIdx.y * gridDim.x + blockIdx.x) * P::BLK;
    for (int idx = threadIdx.x; idx < P::T * WKC * P::NPASS; idx += 256) {
        const int tap = idx % P::T, cl = (idx / P::T) % WKC, nl = idx / (P::T * WKC);
        const int n = n0 + nl, c = c0 + cl;
        float v = 0.f;
        if (n < (FORM == W_PIX ? NOUT >> 2 : NOUT) && c < CIN) {
            if (FORM == W_PIX) {
                const int py = tap >> 3, px = (tap >> 2) & 1, a = (tap >> 1) & 1, b = tap & 1;
                v = wv.w[wv.off + n * wv.sn + c * wv.sc + (3 - py - 2 * a) * wv.sky + (3 - px - 2 * b) * wv.skx];
            } else {
                constexpr int KW = FORM == W_S2 ? 4 : (TAPS == 9 ? 3 : 1);
                const int ky = tap / KW, kx = tap - ky * KW;
                v = wv.w[wv.off + n * wv.sn + c * wv.sc + ky * wv.sky + kx * wv.skx];
            }
        }
        dst[tap * P::W_TS + (cl >> 1) * P::W_PS + (cl & 1) * P::NS + nl] = v;
    }
}

// Input chunk (WKC channels of one tile's window) through registers: issue() starts the loads of chunk c+1 before the
// MFMAs of chunk c, commit() transforms and writes LDS after them.  A row is QW aligned float4 (the interior) plus two
// halo scalars, so a thread handles 2-5 float4 and 1-2 scalars per chunk instead of 6-20 scalars, each of which cost
// ~25 integer instructions of index arithmetic (measured: 10 VALU instructions per MFMA, no overlap with the MFMAs).
// P1: also prefetch the second tensor of an AFFINE2 operand (otherwise commit() reads it: exposed latency, rare path).
template <class G, int NCH, bool P1, int NTH = 256>
struct RowPrefetch {
    static constexpr int NU4 = NCH * G::ROWS * G::QW, J4 = (NU4 + NTH - 1) / NTH;
    static constexpr int NH = G::HALO ? NCH * G::ROWS * 2 : 0, JH = (NH + NTH - 1) / NTH;
    f32x4 v[J4], u[P1 ? J4 : 1];
    float hv[JH ? JH : 1], hu[(P1 && JH) ? JH : 1];
    int b, c0, gy0, gxi;
    unsigned phys_mask, ones_mask;       // bit j: float4 unit j, bit 16+j: halo unit j (of the chunk in flight)

    // live == false (no next chunk): empty descriptors, every load returns 0 without touching memory.  The call itself must
    // stay unconditional: loads under a branch make hipcc merge the two paths with register copies behind s_waitcnt vmcnt(0),
    // i.e. wait for the prefetch right where it was issued.
    // begin() fixes the chunk and its descriptors, slot(k) requests one float4 unit (k < J4) or one halo scalar: the
    // kernel spreads the slots over its MFMA steps.  A CU keeps only so many bytes in flight; a wave that issues a whole
    // chunk in one burst stalls at the issue point, in front of its MFMAs, until earlier requests return.
    __amdgpu_buffer_rsrc_t r0, r1;
    int CINl, Cph, Hh, Ww;
    static constexpr int NSLOT = J4 + JH;
    __device__ __forceinline__ void begin(const Operand &op, bool live, int b_, int c0_, int CIN, int Cphys, int gy0_, int gxi_,
                                          int H, int W)
    {
        b = b_; c0 = c0_; gy0 = gy0_; gxi = gxi_; CINl = CIN; Cph = Cphys; Hh = H; Ww = W;
        const long long se = (long long)Cphys * H * W;
        const int bytes = live ? (int)(se * 4) : 0;
        r0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(op.p0 + se * b), 0, bytes, 0x00020000);
        const bool two = P1 && op.mode == DM_LOAD_AFFINE2;
        r1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>((two ? op.p1 : op.p0) + se * b), 0, two ? bytes : 0, 0x00020000);
        phys_mask = 0; ones_mask = 0;
    }
    template <int K>
    __device__ __forceinline__ void slot(int tid)
    {
        if constexpr (K < J4) {
            constexpr int j = K;
            const int unit = j * NTH + tid, row = unit / G::QW, q = unit - row * G::QW;
            const int c = row / G::ROWS, iy = row - c * G::ROWS, chn = c0 + c, gy = gy0 + iy;
            const bool inimg = unit < NU4 && chn < CINl && (unsigned)gy < (unsigned)Hh;
            const bool phys = inimg && chn < Cph;
            phys_mask |= phys ? 1u << j : 0u;
            ones_mask |= (inimg && !phys) ? 1u << j : 0u;
            const int voff = phys ? ((chn * Hh + gy) * Ww + gxi + 4 * q) * 4 : 0x7ffffff0;
            v[j] = __builtin_amdgcn_raw_buffer_load_b128(r0, voff, 0, 0);
            if (P1) u[j] = __builtin_amdgcn_raw_buffer_load_b128(r1, voff, 0, 0);
        } else if constexpr (K < NSLOT) {
            constexpr int j = K - J4;
            const int unit = j * NTH + tid, row = unit >> 1, side = unit & 1;
            const int c = row / G::ROWS, iy = row - c * G::ROWS, chn = c0 + c, gy = gy0 + iy;
            const int gx = side ? gxi + 4 * G::QW : gxi - 1;
            const bool inimg = unit < NH && chn < CINl && (unsigned)gy < (unsigned)Hh && (unsigned)gx < (unsigned)Ww;
            const bool phys = inimg && chn < Cph;
            phys_mask |= phys ? 1u << (16 + j) : 0u;
            ones_mask |= (inimg && !phys) ? 1u << (16 + j) : 0u;
            const int voff = phys ? ((chn * Hh + gy) * Ww + gx) * 4 : 0x7ffffff0;
            hv[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r0, voff, 0, 0));
            if (P1) hu[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r1, voff, 0, 0));
        }
    }
    template <int K = 0>
    __device__ __forceinline__ void slots_all(int tid)
    {
        if constexpr (K < NSLOT) { slot<K>(tid); slots_all<K + 1>(tid); }
    }
    // live == false (no next chunk): empty descriptors, every load returns 0 without touching memory.  The calls themselves
    // must stay unconditional: loads under a branch make hipcc merge the two paths with register copies behind
    // s_waitcnt vmcnt(0), i.e. wait for the prefetch right where it was issued.
    __device__ __forceinline__ void issue(const Operand &op, bool live, int b_, int c0_, int CIN, int Cphys, int gy0_, int gxi_,
                                          int H, int W, int tid)
    {
        begin(op, live, b_, c0_, CIN, Cphys, gy0_, gxi_, H, W);
        slots_all<0>(tid);
    }

    // Operand transform without control flow: the mode is uniform, so every choice is a v_cndmask on a scalar condition.
    // (As nested ifs per element hipcc produced ~30 basic blocks per float4, each with its own s_waitcnt: the commit of
    //  3 units per thread cost as much as the chunk's 144 MFMAs.)
    struct Xf {
        bool aff, two, relu;
        __device__ __forceinline__ float one(float x, float uu, float c0, float c1, float c2) const
        {
            const float t = two ? c1 * uu + c2 : c2;
            const float y = c0 * x + t;
            x = aff ? y : x;
            const float lo = relu ? 0.f : -__builtin_inff();
            return x < lo ? lo : x;                       // a NaN stays NaN
        }
    };

    __device__ __forceinline__ void commit(const Operand &op, const float *s_cf, float *s_dst, int Cphys, int H, int W, int tid) const
    {
        const int mode = op.mode;
        const Xf xf{mode >= DM_LOAD_AFFINE, mode == DM_LOAD_AFFINE2, mode == DM_LOAD_RELU || mode == DM_LOAD_AFFINE_RELU};
        const bool slow_p1 = !P1 && mode == DM_LOAD_AFFINE2;
#pragma unroll
        for (int j = 0; j < J4; ++j) {
            const int unit = j * NTH + tid, row = unit / G::QW, q = unit - row * G::QW;
            if (unit >= NU4) continue;
            const int c = row / G::ROWS, iy = row - c * G::ROWS;
            f32x4 x = v[j];
            f32x4 uu = P1 ? u[j] : x;
            if (slow_p1 && ((phys_mask >> j) & 1))
                uu = *reinterpret_cast<const f32x4 *>(op.p1 + (((long long)b * Cphys + c0 + c) * H + gy0 + iy) * W + gxi + 4 * q);
            float k0 = 1.f, k1 = 0.f, k2 = 0.f;
            if (xf.aff) { k0 = s_cf[c * 4]; k1 = s_cf[c * 4 + 1]; k2 = s_cf[c * 4 + 2]; }
            x = (f32x4){xf.one(x.x, uu.x, k0, k1, k2), xf.one(x.y, uu.y, k0, k1, k2), xf.one(x.z, uu.z, k0, k1, k2),
                        xf.one(x.w, uu.w, k0, k1, k2)};
            const float f = (ones_mask >> j) & 1 ? 1.f : 0.f;
            if (!((phys_mask >> j) & 1)) x = (f32x4){f, f, f, f};
            if (G::PLANES) {
                float *base = s_dst + c * G::CHS + iy * G::RS + 2 * q;
                *reinterpret_cast<f32x2 *>(base + G::PLS) = (f32x2){x.x, x.z};      // odd window columns 1+4q, 3+4q
                *reinterpret_cast<f32x2 *>(base + 2) = (f32x2){x.y, x.w};           // even window columns 2+4q, 4+4q
            } else {
                *reinterpret_cast<f32x4 *>(s_dst + c * G::CHS + iy * G::RS + G::OFFC + 4 * q) = x;
            }
        }
#pragma unroll
        for (int j = 0; j < JH; ++j) {
            const int unit = j * NTH + tid, row = unit >> 1, side = unit & 1;
            if (unit >= NH) continue;
            const int c = row / G::ROWS, iy = row - c * G::ROWS;
            float x = hv[j];
            float uu = P1 ? hu[j] : x;
            if (slow_p1 && ((phys_mask >> (16 + j)) & 1))
                uu = op.p1[(((long long)b * Cphys + c0 + c) * H + gy0 + iy) * W + (side ? gxi + 4 * G::QW : gxi - 1)];
            float k0 = 1.f, k1 = 0.f, k2 = 0.f;
            if (xf.aff) { k0 = s_cf[c * 4]; k1 = s_cf[c * 4 + 1]; k2 = s_cf[c * 4 + 2]; }
            x = xf.one(x, uu, k0, k1, k2);
            if (!((phys_mask >> (16 + j)) & 1)) x = (ones_mask >> (16 + j)) & 1 ? 1.f : 0.f;
            int a;
            if (G::PLANES) a = c * G::CHS + iy * G::RS + (side ? G::PLS + 16 : 1);
            else a = c * G::CHS + iy * G::RS + (side ? G::OFFC + 16 : G::OFFC - 1);
            s_dst[a] = x;
        }
    }
};

// ---------------------------------------------------------------------------------------------- convolution
// grid (x: persistent over tiles or samples, y: passes of 16*NPW output channels).
//   per_tile == 0: workgroup x walks tiles x, x+gx, ...; statistics of all of them -> slab x; slabs >= gx are zeroed.
//   per_tile != 0: workgroup x walks samples x, x+gx, ...; statistics of a sample -> slab b*(nslabs/B), the sample's
//                  other slabs are zeroed (per-sample BatchNorm sums the slabs of a sample).
// wpk: the packed weights (wide_pack_kernel, same <FORM, TAPS, NPW>).
template <int FORM, int TAPS, int NPW>
__global__ __launch_bounds__(256, ((FORM == W_S1 && TAPS == 9 && NPW == 4) || DM_WIDE_OCC3) ? 3 : 2) void conv_wide_kernel(Operand in, const float *__restrict__ wpk, float *__restrict__ out,
                                                        Epilogue ep, int B, int Cphys, int CIN, int NOUT, int H, int W,
                                                        int nslabs, int per_tile)
{
    using G = WideGeom<FORM, TAPS>;
    using P = WideW<FORM, TAPS, NPW>;
    constexpr int NS = P::NS, W_PS = P::W_PS, W_TS = P::W_TS, NPASS = P::NPASS;
    __shared__ __attribute__((aligned(16))) float s_in[WKC * G::CHS];
    __shared__ __attribute__((aligned(16))) float s_w[P::BLK];
    __shared__ double s_red[4 * NPW * 16 * 2];
    __shared__ float s_cf[2][WKC * 4];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, p = lane & 15, kq = lane >> 4;
    const int CO = FORM == W_PIX ? NOUT >> 2 : NOUT;
    const int BH = FORM == W_S2 ? H >> 1 : H, BW = FORM == W_S2 ? W >> 1 : W;       // base grid the tiles cover
    const int OH = FORM == W_PIX ? 2 * H : BH, OW = FORM == W_PIX ? 2 * W : BW;
    const int tx_n = BW >> 4, tps = (BH >> 3) * tx_n;
    const int n0 = blockIdx.y * NPASS;
    const int nchunks = (CIN + WKC - 1) / WKC;
    const int spg = per_tile ? nslabs / B : 1;
    const int ngroups = per_tile ? B : 1;
    const __amdgpu_buffer_rsrc_t wr_live = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(wpk + (long long)blockIdx.y * nchunks * P::BLK), 0, nchunks * P::BLK * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t wr_dead = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(wpk), 0, 0, 0x00020000);

    for (int g = per_tile ? blockIdx.x : 0; g < ngroups; g += per_tile ? gridDim.x : 1) {
        double st1[NPW], st2[NPW];
#pragma unroll
        for (int t = 0; t < NPW; ++t) { st1[t] = 0.0; st2[t] = 0.0; }
        const int t_begin = per_tile ? g * tps : blockIdx.x, t_end = per_tile ? (g + 1) * tps : B * tps;
        const int t_step = per_tile ? 1 : gridDim.x;
        constexpr int JW = (P::BLK / 4 + 255) / 256;
        RowPrefetch<G, WKC, FORM != W_S2> pin;
        f32x4 pw[JW];
        int par = 0;

        // chunk (tile, ch) -> registers: input rows, the packed weight block, coefficient rows (double-buffered in LDS).
        // issue_begin fixes the chunk; issue_slot<K> requests one piece (input unit, halo scalar or weight float4).
        constexpr int NSLOT = decltype(pin)::NSLOT + JW;
        __amdgpu_buffer_rsrc_t wr;
        int wbase = 0;
        auto issue_begin = [&](int tile, int ch, bool live) {
            const int b = tile / tps, r = tile - b * tps;
            const int y0 = (r / tx_n) << 3, x0 = (r - (r / tx_n) * tx_n) << 4;
            const int c0 = ch * WKC;
            par ^= 1;
            stage_coef(in, s_cf[par], WKC, c0, Cphys, b, tid);
            pin.begin(in, live, b, c0, CIN, Cphys, G::S * y0 - G::R, G::S * x0, H, W);
            wr = live ? wr_live : wr_dead;
            wbase = ch * P::BLK;
        };
        auto issue_slot = [&](auto kc) {
            constexpr int K = decltype(kc)::value;
            if constexpr (K < decltype(pin)::NSLOT) {
                pin.template slot<K>(tid);
            } else if constexpr (K < NSLOT) {
                constexpr int j = K - decltype(pin)::NSLOT;
                const int idx = j * 256 + tid;
                pw[j] = __builtin_amdgcn_raw_buffer_load_b128(wr, idx < P::BLK / 4 ? (wbase + idx * 4) * 4 : 0x7ffffff0, 0, 0);
            }
        };
        auto issue_all = [&](auto self, auto kc) -> void {
            constexpr int K = decltype(kc)::value;
            if constexpr (K < NSLOT) { issue_slot(kc); self(self, std::integral_constant<int, K + 1>{}); }
        };
        auto issue = [&](int tile, int ch, bool live) {
            issue_begin(tile, ch, live);
            issue_all(issue_all, std::integral_constant<int, 0>{});
        };
        auto commit = [&]() {
            pin.commit(in, s_cf[par], s_in, Cphys, H, W, tid);
#pragma unroll
            for (int j = 0; j < JW; ++j) {
                const int idx = j * 256 + tid;
                if (idx < P::BLK / 4) *reinterpret_cast<f32x4 *>(&s_w[idx * 4]) = pw[j];
            }
        };

        issue(t_begin < t_end ? t_begin : 0, 0, t_begin < t_end);
        for (int tile = t_begin; tile < t_end; tile += t_step) {
            const int b = tile / tps, r = tile - b * tps;
            const int y0 = (r / tx_n) << 3, x0 = (r - (r / tx_n) * tx_n) << 4;
            constexpr int NPAR = FORM == W_PIX ? 4 : 1;              // transposed form: one accumulator set per output parity
            f32x4 acc[NPAR][2][NPW];
#pragma unroll
            for (int q = 0; q < NPAR; ++q)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int t = 0; t < NPW; ++t) acc[q][i][t] = (f32x4){0.f, 0.f, 0.f, 0.f};

            for (int ch = 0; ch < nchunks; ++ch) {
                __syncthreads();                 // the previous chunk's MFMAs are done with s_in / s_w
                commit();
                __syncthreads();
                // next chunk's loads fly during this chunk's MFMAs (and, across tiles, during the epilogue); they are
                // requested a few at a time between the MFMA steps
                {
                    const bool wrap = ch + 1 == nchunks;
                    const int ntile = wrap ? tile + t_step : tile;
                    const bool live = ntile < t_end;
                    issue_begin(live ? ntile : tile, wrap ? 0 : ch + 1, live);
                }
                // ---- MFMAs: wave owns base rows 2*wave, 2*wave+1.  K steps of 4 channels: step = 2*tap + channel quad.
                // The LDS operands of step s+1 are requested before the MFMAs of step s are issued (two register sets),
                // otherwise every group of MFMAs starts by waiting out an LDS round trip.
                auto lds_step = [&](int step, float (&av)[2], float (&bv)[NPW]) {
                    const int tap = step >> 1, cq = step & 1;
                    int toff;
                    if (FORM == W_S2) {
                        const int ky = tap >> 2, kx = tap & 3;
                        toff = (kx & 1) * G::PLS + ky * G::RS + (kx >> 1) + ((kx & 1) ? 0 : 1);
                    } else if (FORM == W_PIX) {
                        const int py = tap >> 3, px = (tap >> 2) & 1, a = (tap >> 1) & 1, b = tap & 1;
                        toff = (py + a) * G::RS + (px + b) + G::OFFC - 1;
                    } else {
                        toff = TAPS == 9 ? (tap / 3) * G::RS + (tap % 3) + G::OFFC - 1 : 0;
                    }
                    const float *ap = s_in + (cq * 4 + kq) * G::CHS + toff + (G::S * 2 * wave) * G::RS + p;
                    av[0] = ap[0];
                    av[1] = ap[G::S * G::RS];
                    const float *bp = s_w + tap * W_TS + (cq * 2 + (kq >> 1)) * W_PS + (kq & 1) * NS + p;
#pragma unroll
                    for (int t = 0; t < NPW; ++t) bv[t] = bp[t * 16];
                };
                auto mfma_step = [&](auto parc, const float (&av)[2], const float (&bv)[NPW]) {
                    constexpr int PAR = decltype(parc)::value;
#pragma unroll
                    for (int t = 0; t < NPW; ++t)
#pragma unroll
                        for (int i = 0; i < 2; ++i)
                            acc[PAR][i][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bv[t], acc[PAR][i][t], 0, 0, 0);
                };
                static_assert(WKC == 8, "two channel quads per tap");
                constexpr int NSTEP = 2 * G::T, NIT = G::T;
                // (all requests go out in the first half of the taps: the last one needs a round trip of lead before the
                //  commit that follows this loop)
                constexpr int NITL = NIT > 1 ? NIT / 2 : 1;
                constexpr int SPI = (NSLOT + NITL - 1) / NITL;        // load slots per iteration
                float a0[2], b0[NPW], a1[2], b1[NPW];
                lds_step(0, a0, b0);
                auto round = [&](auto self, auto itc) -> void {
                    constexpr int IT = decltype(itc)::value;
                    if constexpr (IT < NIT) {
                        constexpr int step = 2 * IT;
                        lds_step(step + 1, a1, b1);
                        auto slots = [&](auto self2, auto kc) -> void {
                            constexpr int K = decltype(kc)::value;
                            if constexpr (K < SPI) {
                                issue_slot(std::integral_constant<int, IT * SPI + K>{});
                                self2(self2, std::integral_constant<int, K + 1>{});
                            }
                        };
                        slots(slots, std::integral_constant<int, 0>{});
                        constexpr int PAR = FORM == W_PIX ? IT >> 2 : 0;            // both steps of a tap share its parity
                        __builtin_amdgcn_sched_barrier(0);
                        mfma_step(std::integral_constant<int, PAR>{}, a0, b0);
                        __builtin_amdgcn_sched_barrier(0);
                        lds_step(step + 2 < NSTEP ? step + 2 : step, a0, b0);       // last round: a harmless re-read
                        __builtin_amdgcn_sched_barrier(0);
                        mfma_step(std::integral_constant<int, PAR>{}, a1, b1);
                        __builtin_amdgcn_sched_barrier(0);
                        self(self, std::integral_constant<int, IT + 1>{});
                    }
                };
                round(round, std::integral_constant<int, 0>{});
            }

            // ---- epilogue: lane holds pixels (row 2*wave+i, columns 4*kq .. 4*kq+3) of channel n0 + 16*t + p
            //      (transposed form: of every output parity, i.e. 2 output rows x 8 consecutive output columns)
#pragma unroll
            for (int t = 0; t < NPW; ++t) {
                const int chn = n0 + t * 16 + p;
                const bool live = chn < CO;
                const float bias = (live && ep.bias) ? ep.bias[chn] : 0.f;
                float mc0 = 1.f, mc2 = 0.f;
                if (live && ep.mask.p0 && ep.mask.mode >= DM_LOAD_AFFINE) {
                    const float *cf = ep.mask.coef + (long long)b * ep.mask.coef_bstride + chn * 4;
                    mc0 = cf[0]; mc2 = cf[2];
                }
                auto emit = [&](f32x4 v, int oy, int ox) {
                    if (!live) return;
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
                        v = dm_relu4(v);
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
                };
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    if (FORM == W_PIX) {
#pragma unroll
                        for (int py = 0; py < 2; ++py) {
                            const f32x4 e = acc[py * 2][i][t], o = acc[py * 2 + 1][i][t];      // px = 0, 1
                            const int oy = 2 * (y0 + 2 * wave + i) + py, ox = 2 * (x0 + 4 * kq);
                            emit((f32x4){e.x, o.x, e.y, o.y}, oy, ox);
                            emit((f32x4){e.z, o.z, e.w, o.w}, oy, ox + 4);
                        }
                    } else {
                        emit(acc[0][i][t], y0 + 2 * wave + i, x0 + 4 * kq);
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
            constexpr int CPP = 16 * NPW;                                  // statistics channels of one pass
            for (int i = tid; i < CPP * 2; i += 256) {
                const int cl = i >> 1, k = i & 1;
                const int chn = n0 + cl;
                if (chn >= CO) continue;
                double s = 0.0;
                for (int w = 0; w < 4; ++w) s += s_red[((w * NPW + (cl >> 4)) * 16 + (cl & 15)) * 2 + k];
                ep.stats[(slab * CO + chn) * 2 + k] = s;
                if (per_tile)
                    for (int e = 1; e < spg; ++e) ep.stats[((slab + e) * CO + chn) * 2 + k] = 0.0;
            }
        }
    }
    if (ep.stats && !per_tile) {                                      // slabs no workgroup owns
        constexpr int CPP = 16 * NPW;
        const int cbase = n0;
        for (int sl = blockIdx.x + gridDim.x; sl < nslabs; sl += gridDim.x)
            for (int i = tid; i < CPP * 2; i += 256)
                if (cbase + (i >> 1) < CO) ep.stats[((long long)sl * CO + cbase + (i >> 1)) * 2 + (i & 1)] = 0.0;
    }
}

// ---------------------------------------------------------------------------------------------- weight gradient
// T tile geometry (same LDS images as the convolution input: RowPrefetch stages it)
template <int KK>
struct WgGeom {
    static constexpr int S = KK == 4 ? 2 : 1;
    static constexpr int R = KK == 1 ? 0 : 1;
    static constexpr int T2 = KK * KK;
    static constexpr int ROWS = KK == 4 ? 18 : 8 + 2 * R;
    static constexpr int QW = KK == 4 ? 8 : 4;
    static constexpr bool HALO = R > 0;
    static constexpr bool PLANES = KK == 4;
    static constexpr int RS = KK == 4 ? 18 : (R ? 24 : 16);
    static constexpr int OFFC = KK == 3 ? 4 : 0;
    // 4x4/s2: the 16 lanes of an N tile are the 16 taps of one channel = (plane, row ky, column) with 3 columns in use
    // per plane; rows step the bank by 18, so a plane stride of 8 (mod 32) puts all 24 addresses of a half on their own
    // bank (with the planes back to back, 324 = 4 mod 32, SQ_LDS_BANK_CONFLICT was 49 % of the LDS cycles)
    static constexpr int PLS = ROWS * RS + (KK == 4 ? 4 : 0);
    static constexpr int RAW = KK == 4 ? 2 * PLS : PLS;
    // channel stride: multiple of 4 (aligned float4 / float2 rows); = 4 (mod 32) so that the taps of neighbouring
    // channels inside one 16-lane N tile spread over the banks
    static constexpr int CHS = KK == 4 ? RAW : ((RAW - 4 + 31) / 32) * 32 + 4;
    static constexpr int NTW = KK == 1 ? 4 : 16;                         // N tiles (of 16) per pass
    static constexpr int NCTP = 16 * NTW / T2;                           // T channels per pass: 64 / 28 / 16
    static constexpr int ROWSTEP = S * RS;                               // LDS step of one S-grid row
};
constexpr int WG_CSS = 130;            // S channel stride: the 16 channel lanes x 2 k lanes of a half hit 32 distinct banks

// grid (x: persistent over (sample, tile) units -> slab x, y: passes of NCTP T channels, z: passes of 64 S channels)
template <int KK>
__global__ __launch_bounds__(256, 2) void wgrad_wide_kernel(Operand S, Operand T, float *__restrict__ slabs, int B, int CS,
                                                         int CT, int CTphys, int Hs, int Ws, int nslabs)
{
    using G = WgGeom<KK>;
    __shared__ __attribute__((aligned(16))) float s_S[64 * WG_CSS];
    __shared__ __attribute__((aligned(16))) float s_T[G::NCTP * G::CHS];
    __shared__ float s_cfS[64 * 4], s_cfT[G::NCTP * 4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, p = lane & 15, kq = lane >> 4;
    const int ct0 = blockIdx.y * G::NCTP, cs0 = blockIdx.z * 64;
    const int nct = CT - ct0 < G::NCTP ? CT - ct0 : G::NCTP;           // T channels of this pass
    const int ntw = (nct * G::T2 + 15) >> 4;                          // N tiles in use
    const int ncs = CS - cs0 < 64 ? CS - cs0 : 64;
    // waves -> (M tile of 16 S channels, group of N tiles): with fewer than 64 S channels the spare waves take a share
    // of the N tiles instead of multiplying zero rows (16 channels: 4 groups, 32: 2)
    const int MT = ncs <= 16 ? 1 : (ncs <= 32 ? 2 : 4);
    const int mt = wave & (MT - 1), ng = wave / MT, ngm = 4 / MT - 1;
    const int Ht = Hs * G::S, Wt = Ws * G::S;
    const int tx_n = Ws >> 4, tps = (Hs >> 3) * tx_n;
    const long long E = (long long)CS * CT * G::T2;

    // LDS offset of this lane's (T channel, tap) of N tile nt at pixel (0, kq); rows add ROWSTEP, k steps add 4
    int tb[G::NTW];
#pragma unroll
    for (int nt = 0; nt < G::NTW; ++nt) {
        const int n = nt * 16 + p, ctl = n / G::T2, tap = n - ctl * G::T2;
        const int ky = tap / KK, kx = tap - ky * KK;
        int o = ctl * G::CHS + kq;
        if (KK == 4) o += (kx & 1) * G::PLS + ky * G::RS + (kx >> 1) + ((kx & 1) ? 0 : 1);
        else if (KK == 3) o += ky * G::RS + kx + G::OFFC - 1;
        tb[nt] = ctl < nct ? o : kq;
    }
    f32x4 acc[G::NTW];
#pragma unroll
    for (int nt = 0; nt < G::NTW; ++nt) acc[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    RowPrefetch<G, G::NCTP, false> pt;
    const int units = B * tps;
    for (int u = blockIdx.x; u < units; u += gridDim.x) {
        const int b = u / tps, r = u - b * tps;
        const int y0 = (r / tx_n) << 3, x0 = (r - (r / tx_n) * tx_n) << 4;
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));      // staging index arithmetic is redone per unit, not hoisted into ~100 registers
        stage_coef(S, s_cfS, ncs, cs0, CS, b, tid);
        stage_coef(T, s_cfT, nct, ct0, CTphys, b, tid);
        __syncthreads();
        // T tile with halo (all of its loads in flight together), then the S tile in two halves
        pt.issue(T, true, b, ct0, CT, CTphys, G::S * y0 - G::R, G::S * x0, Ht, Wt, tid);
        pt.commit(T, s_cfT, s_T, CTphys, Ht, Wt, tid);
        // S tile: 64 channels x 8 rows x 16 columns as float4
        const long long sample = (long long)b * CS * Hs * Ws;
        const int smode = S.mode;
        constexpr int SB = KK == 3 ? 2 : 4;        // float4 per thread in flight (register budget: 2 workgroups per CU)
#pragma unroll 1
        for (int part = 0; part < 8 / SB; ++part) {
            f32x4 sv[SB], su[SB];
#pragma unroll
            for (int j = 0; j < SB; ++j) {
                const int idx = (part * SB + j) * 256 + tid, cl = idx >> 5, rr = (idx >> 2) & 7, c4 = idx & 3;
                const int off = cl < ncs ? ((cs0 + cl) * Hs + y0 + rr) * Ws + x0 + c4 * 4 : 0;
                sv[j] = *reinterpret_cast<const f32x4 *>(S.p0 + sample + off);
                if (smode == DM_LOAD_AFFINE2) su[j] = *reinterpret_cast<const f32x4 *>(S.p1 + sample + off);
            }
#pragma unroll
            for (int j = 0; j < SB; ++j) {
                const int idx = (part * SB + j) * 256 + tid, cl = idx >> 5, rr = (idx >> 2) & 7, c4 = idx & 3;
                f32x4 v = sv[j];
                if (smode == DM_LOAD_RELU) {
                    v = dm_relu4(v);
                } else if (smode == DM_LOAD_AFFINE2) {
                    v = s_cfS[cl * 4] * v + (s_cfS[cl * 4 + 1] * su[j] + s_cfS[cl * 4 + 2]);
                } else if (smode >= DM_LOAD_AFFINE) {
                    v = s_cfS[cl * 4] * v + s_cfS[cl * 4 + 2];
                    if (smode == DM_LOAD_AFFINE_RELU) {
                        v = dm_relu4(v);
                    }
                }
                if (cl >= ncs) v = (f32x4){0.f, 0.f, 0.f, 0.f};
                f32x2 *dst = reinterpret_cast<f32x2 *>(&s_S[cl * WG_CSS + rr * 16 + c4 * 4]);   // rows are only 8-byte aligned
                dst[0] = (f32x2){v.x, v.y};
                dst[1] = (f32x2){v.z, v.w};
            }
        }
        __syncthreads();
        // K = the 128 pixels: 8 rows x 4 steps of 4 columns.  Inside a row every LDS address is tb[nt] + constant.
        const float *sp = s_S + (mt * 16 + p) * WG_CSS + kq;
#pragma unroll 1
        for (int row = 0; row < 8; ++row) {
#pragma unroll
            for (int k4 = 0; k4 < 4; ++k4) {
                const float av = sp[row * 16 + k4 * 4];
                float bv[G::NTW];
#pragma unroll
                for (int nt = 0; nt < G::NTW; ++nt)
                    if (nt < ntw && (nt & ngm) == ng) bv[nt] = s_T[tb[nt] + k4 * 4];
#pragma unroll
                for (int nt = 0; nt < G::NTW; ++nt)
                    if (nt < ntw && (nt & ngm) == ng) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv[nt], acc[nt], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);          // keep the four steps' LDS reads from all moving to the top
            }
#pragma unroll
            for (int nt = 0; nt < G::NTW; ++nt) tb[nt] += G::ROWSTEP;
        }
#pragma unroll
        for (int nt = 0; nt < G::NTW; ++nt) tb[nt] -= 8 * G::ROWSTEP;
    }

    // lane holds R[cs = cs0 + 16*mt + 4*kq + j][n = 16*nt + p] for the N tiles of its group
    float *row = slabs + (long long)blockIdx.x * E;
#pragma unroll
    for (int nt = 0; nt < G::NTW; ++nt) {
        const int n = nt * 16 + p, ctl = n / G::T2, tap = n - ctl * G::T2;
        if (nt >= ntw || ctl >= nct || (nt & ngm) != ng) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int cs = cs0 + mt * 16 + kq * 4 + j;
            if (cs < CS) row[((long long)cs * CT + ct0 + ctl) * G::T2 + tap] = acc[nt][j];
        }
    }
    if (blockIdx.y == 0 && blockIdx.z == 0)
        for (int sl = blockIdx.x + gridDim.x; sl < nslabs; sl += gridDim.x)
            for (long long e = tid; e < E; e += 256) slabs[(long long)sl * E + e] = 0.f;
}


// ---- weight gradient of the 64-channel layers in ONE pass (3x3: 64 x 64, 4x4 / stride 2: 64 x 32) ---------------------------
// wgrad_wide_kernel above covers N = (T channel, tap) in passes of 16 N tiles -- three passes over 64 x 9 columns, each staging
// the S tile again -- and stages a unit between two barriers with the loads exposed: 40 % of its wave time was waiting
// (profiles/r06_z32ex_sq_counters.txt), 0.56 of what the clock allows.  Here a workgroup of EIGHT waves owns all N tiles
// (wave = M tile of 16 S channels x one half of the N tiles: 18 or 16 accumulators), both operand tiles of a unit (8 x 16
// S pixels) sit in LDS once (100-117 KB: one workgroup per CU, two waves per SIMD as before), and the NEXT unit's S and T rows
// are requested into registers a few at a time between the matrix steps of this one; after the K loop the workgroup
// transforms and writes them (commit) between two barriers.  The K loop is straight-line code: 32 K steps of one A read and
// NTW B reads whose addresses are a per-tile base register plus an immediate.
template <int KK, int NCT>
struct Wg1Geom : WgGeom<KK> {
    using G0 = WgGeom<KK>;
    static constexpr int NT = NCT * G0::T2 / 16, NTW = NT / 2;          // N tiles, N tiles per wave
    static_assert(NT * 16 == NCT * G0::T2 && NTW * 2 == NT, "the two halves split the N tiles evenly");
};

template <int KK, int NCT, bool S2, bool TTWO = false>
__global__ __launch_bounds__(512, 1) void wgrad_wide1_kernel(Operand S, Operand T, float *__restrict__ slabs, int B, int Hs, int Ws,
                                                            int nslabs)
{
    using G = Wg1Geom<KK, NCT>;
    constexpr int CS = 64, NTW = G::NTW, T2 = G::T2;
    __shared__ __attribute__((aligned(16))) float s_S[64 * WG_CSS];
    __shared__ __attribute__((aligned(16))) float s_T[NCT * G::CHS];
    __shared__ float s_cfS[64 * 4], s_cfT[NCT * 4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, p = lane & 15, kq = lane >> 4;
    const int mt = wave & 3, nh = wave >> 2;
    const int Ht = Hs * G::S, Wt = Ws * G::S;
    const int tx_n = Ws >> 4, tps = (Hs >> 3) * tx_n;
    const long long E = (long long)CS * NCT * T2;

    // LDS offset of this lane's (T channel, tap) of the wave's N tile t at pixel (0, kq); rows add ROWSTEP, K steps add 4
    int tb[NTW];
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
        const int n = (nh * NTW + t) * 16 + p, ctl = n / T2, tap = n - ctl * T2;
        const int ky = tap / KK, kx = tap - ky * KK;
        int o = ctl * G::CHS + kq;
        if (KK == 4) o += (kx & 1) * G::PLS + ky * G::RS + (kx >> 1) + ((kx & 1) ? 0 : 1);
        else if (KK == 3) o += ky * G::RS + kx + G::OFFC - 1;
        tb[t] = o;
    }
    f32x4 acc[NTW];
#pragma unroll
    for (int t = 0; t < NTW; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // shared coefficients (batch statistics: the host checks coef_bstride == 0)
    stage_coef(S, s_cfS, 64, 0, CS, 0, tid, 512);
    stage_coef(T, s_cfT, NCT, 0, NCT, 0, tid, 512);

    RowPrefetch<G, NCT, TTWO, 512> pt;                           // TTWO: the T operand is AFFINE2 (its second tensor is prefetched too)
    constexpr int JS = 64 * 8 * 4 / 512;                         // S tile: 64 channels x 8 rows x 4 float4 = 4 per thread
    f32x4 sv[JS], su[S2 ? JS : 1];
    __amdgpu_buffer_rsrc_t rS0, rS1;
    int sbase = 0;
    int tq = tid;                                                // the staging code's copy of the thread id (see the unit loop)
    auto s_begin = [&](int u, bool live) {
        const int b = u / tps, r = u - b * tps;
        const int y0 = (r / tx_n) << 3, x0 = (r - (r / tx_n) * tx_n) << 4;
        const long long se = (long long)CS * Hs * Ws;
        const int bytes = live ? (int)(se * 4) : 0;
        rS0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(S.p0 + se * b), 0, bytes, 0x00020000);
        rS1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>((S2 ? S.p1 : S.p0) + se * b), 0, S2 ? bytes : 0, 0x00020000);
        sbase = (y0 * Ws + x0) * 4;
        pt.begin(T, live, b, 0, NCT, NCT, G::S * y0 - G::R, G::S * x0, Ht, Wt);
    };
    constexpr int NSLOT = JS + decltype(pt)::NSLOT;
    auto slot = [&](auto kc) {
        constexpr int K = decltype(kc)::value;
        if constexpr (K < JS) {
            const int idx = K * 512 + tq, cl = idx >> 5, rr = (idx >> 2) & 7, c4 = idx & 3;
            const int off = sbase + ((cl * Hs + rr) * Ws + c4 * 4) * 4;
            sv[K] = __builtin_amdgcn_raw_buffer_load_b128(rS0, off, 0, 0);
            if constexpr (S2) su[K] = __builtin_amdgcn_raw_buffer_load_b128(rS1, off, 0, 0);
        } else if constexpr (K < NSLOT) {
            pt.template slot<K - JS>(tq);
        }
    };
    auto slots_from = [&](auto self, auto kc, auto endc) -> void {
        constexpr int K = decltype(kc)::value, END = decltype(endc)::value;
        if constexpr (K < END && K < NSLOT) { slot(kc); self(self, std::integral_constant<int, K + 1>{}, endc); }
    };
    auto commit = [&]() {
        const int smode = S.mode;
#pragma unroll
        for (int j = 0; j < JS; ++j) {
            const int idx = j * 512 + tq, cl = idx >> 5, rr = (idx >> 2) & 7, c4 = idx & 3;
            f32x4 v = sv[j];
            if constexpr (S2) {
                v = s_cfS[cl * 4] * v + (s_cfS[cl * 4 + 1] * su[j] + s_cfS[cl * 4 + 2]);
            } else if (smode == DM_LOAD_RELU) {
                v = dm_relu4(v);
            } else if (smode >= DM_LOAD_AFFINE) {
                v = s_cfS[cl * 4] * v + s_cfS[cl * 4 + 2];
                if (smode == DM_LOAD_AFFINE_RELU) v = dm_relu4(v);
            }
            f32x2 *dst = reinterpret_cast<f32x2 *>(&s_S[cl * WG_CSS + rr * 16 + c4 * 4]);     // rows are only 8-byte aligned
            dst[0] = (f32x2){v.x, v.y};
            dst[1] = (f32x2){v.z, v.w};
        }
        pt.commit(T, s_cfT, s_T, NCT, Ht, Wt, tq);
    };

    const int units = B * tps;
    int u = blockIdx.x;
    s_begin(u < units ? u : 0, u < units);
    slots_from(slots_from, std::integral_constant<int, 0>{}, std::integral_constant<int, NSLOT>{});
    __syncthreads();                                             // the coefficient tables
    for (; u < units; u += gridDim.x) {
        tq = threadIdx.x;
        asm volatile("" : "+v"(tq));       // staging index arithmetic is redone per unit, not hoisted into dozens of registers
        commit();
        __syncthreads();
        const int un = u + gridDim.x;
        s_begin(un < units ? un : u, un < units);
        // K = the 128 pixels: 8 rows x 4 steps of 4 columns; every LDS address is a base register + immediate.  The operands of
        // step s + 1 are requested before the products of step s are issued; the next unit's loads go out a few per step.
        const float *sp = s_S + (mt * 16 + p) * WG_CSS + kq;
        // ONE set of B registers: the read of step s + 1 into b[t] follows the product of step s that consumed b[t] (LDS returns
        // in order, so product t of the next step waits for read t only) -- a second set cost 18 registers and spilled
        float av = sp[0], bv[NTW];
#pragma unroll
        for (int t = 0; t < NTW; ++t) bv[t] = s_T[tb[t]];
        auto kloop = [&](auto self, auto sc) -> void {
            constexpr int st = decltype(sc)::value;
            if constexpr (st < 32) {
                constexpr int rn = (st + 1) >> 2, kn = (st + 1) & 3;
                float an = 0.f;
                if constexpr (st + 1 < 32) an = sp[rn * 16 + kn * 4];
#pragma unroll
                for (int t = 0; t < NTW; ++t) {
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv[t], acc[t], 0, 0, 0);
                    if constexpr (st + 1 < 32) bv[t] = s_T[tb[t] + rn * G::ROWSTEP + kn * 4];
                }
                constexpr int SPI = (NSLOT + 27) / 28;          // all requests leave in the first 28 steps: a round trip of lead
                slots_from(slots_from, std::integral_constant<int, st * SPI>{}, std::integral_constant<int, (st + 1) * SPI>{});
                av = an;
                self(self, std::integral_constant<int, st + 1>{});
            }
        };
        kloop(kloop, std::integral_constant<int, 0>{});
        __syncthreads();                                         // every wave is done with s_S / s_T
    }

    // lane holds R[cs = 16 mt + 4 kq + j][n = 16 (nh NTW + t) + p]
    float *row = slabs + (long long)blockIdx.x * E;
#pragma unroll
    for (int t = 0; t < NTW; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) row[(long long)(mt * 16 + kq * 4 + j) * (NCT * T2) + (nh * NTW + t) * 16 + p] = acc[t][j];
    for (int sl = blockIdx.x + gridDim.x; sl < nslabs; sl += gridDim.x)
        for (long long e = tid; e < E; e += 512) slabs[(long long)sl * E + e] = 0.f;
}

}  // namespace

bool dm_stream_conv1x1(const Operand &in, const WeightView &wv, float *out, const Epilogue &ep, int B, int Cphys, int CIN,
                       int NOUT, int H, int W, int nslabs, int per_tile, hipStream_t st);

bool dm_stream_conv_s2_thin(const Operand &in, const WeightView &wv, float *out, const Epilogue &ep, int B, int Cphys, int CIN,
                            int NOUT, int H, int W, int nslabs, int per_tile, hipStream_t st);

bool dm_stream_convT_thin(const Operand &in, const WeightView &wv, float *scratch, float *out, const Epilogue &ep, int B, int Cphys,
                          int CIN, int NOUT, int H, int W, int per_tile, hipStream_t st);

bool dm_stream_conv_s2_wide(const Operand &in, const WeightView &wv, float *out, const Epilogue &ep, int B, int Cphys, int CIN,
                            int NOUT, int H, int W, int nslabs, int per_tile, hipStream_t st);

bool dm_stream_convT_wide(const Operand &in, const WeightView &wv, float *out, const Epilogue &ep, int B, int Cphys, int CIN,
                          int NOUT, int H, int W, int nslabs, int per_tile, hipStream_t st);

bool dm_stream_conv3x3_wide(const Operand &in, const WeightView &wv, float *out, const Epilogue &ep, int B, int Cphys, int CIN,
                            int NOUT, int H, int W, int nslabs, int per_tile, hipStream_t st);

// ---- entry points used by the dispatchers in conv_mfma.hip / wgrad_mfma.hip (not part of the public header) ----------
// base grid (output pixels for the strided / plain forms, input pixels for the transposed form) must tile by 8 x 16
static int wide_disabled()
{
    static const int v = getenv("DM_NO_WIDE") ? atoi(getenv("DM_NO_WIDE")) : 0;       // debugging aid: 1 conv, 2 wgrad, 3 both
    return v;
}

bool dm_wide_conv_ok(int form, int H, int W)
{
    if ((wide_disabled() & 1) || (wide_disabled() & (4 << form))) return false;       // 4 / 8 / 16: one form only
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

// output channels per pass / 16: wider passes put the accumulators plus the prefetch registers past 256
// (the stride-2 form stages 2.5x as much input per chunk and the transposed form's unrolled loop is register-hungry:
//  32 channels per pass there)
static int wide_npw(int form, int NOUT)
{
    const int n = form == W_PIX ? NOUT / 4 : NOUT;          // transposed form: N = output channels, 4 parities each
    return n <= 16 ? 1 : ((n <= 32 || form != W_S1) ? 2 : 4);
}

template <int FORM, int TAPS, int NPW>
static long long wide_scratch_floats_t(int CIN, int NOUT)
{
    const int n = FORM == W_PIX ? NOUT / 4 : NOUT;
    const long long passes = (n + 16 * NPW - 1) / (16 * NPW), nchunks = (CIN + WKC - 1) / WKC;
    return passes * nchunks * WideW<FORM, TAPS, NPW>::BLK;
}

#define DM_WIDE_SWITCH(CALL)                                                                                    \
    if (form == W_S2) {                                                                                         \
        if (np == 1) CALL(W_S2, 16, 1) else if (np == 2) CALL(W_S2, 16, 2) else CALL(W_S2, 16, 4)              \
    } else if (form == W_PIX) {                                                                                 \
        if (np == 1) CALL(W_PIX, 9, 1) else CALL(W_PIX, 9, 2)                                                    \
    } else if (taps == 9) {                                                                                     \
        if (np == 1) CALL(W_S1, 9, 1) else if (np == 2) CALL(W_S1, 9, 2) else CALL(W_S1, 9, 4)                 \
    } else {                                                                                                    \
        if (np == 1) CALL(W_S1, 1, 1) else if (np == 2) CALL(W_S1, 1, 2) else CALL(W_S1, 1, 4)                 \
    }

// floats of caller scratch (dm_weight_view.scratch) the packed weights of this convolution take
long long dm_wide_conv_scratch_floats(int form, int CIN, int NOUT, int taps)
{
    const int np = wide_npw(form, NOUT);
    long long r = 0;
#define DM_WS(F, TP, NP_) { r = wide_scratch_floats_t<F, TP, NP_>(CIN, NOUT); }
    DM_WIDE_SWITCH(DM_WS)
#undef DM_WS
    return r;
}

int dm_wide_conv(int form, const Operand &in, const WeightView &wv, float *scratch, float *out, const Epilogue &ep, int B,
                 int Cphys, int CIN, int NOUT, int H, int W, int taps, int nslabs, int per_tile, hipStream_t st)
{
    if (form == W_PIX && CIN * 32 <= dm_wide_conv_scratch_floats(form, CIN, NOUT, taps) &&
        dm_stream_convT_thin(in, wv, scratch, out, ep, B, Cphys, CIN, NOUT, H, W, per_tile, st))
        return 0;
    if (form == W_S1 && taps == 9 && dm_stream_conv3x3_wide(in, wv, out, ep, B, Cphys, CIN, NOUT, H, W, nslabs, per_tile, st)) return 0;
    if (form == W_PIX && dm_stream_convT_wide(in, wv, out, ep, B, Cphys, CIN, NOUT, H, W, nslabs, per_tile, st)) return 0;
    if (form == W_S2 && dm_stream_conv_s2_wide(in, wv, out, ep, B, Cphys, CIN, NOUT, H, W, nslabs, per_tile, st)) return 0;
    if (form == W_S2 && dm_stream_conv_s2_thin(in, wv, out, ep, B, Cphys, CIN, NOUT, H, W, nslabs, per_tile, st)) return 0;
    if (form == W_S1 && taps == 1 && dm_stream_conv1x1(in, wv, out, ep, B, Cphys, CIN, NOUT, H, W, nslabs, per_tile, st)) return 0;
    const int BH = form == W_S2 ? H / 2 : H, BW = form == W_S2 ? W / 2 : W;
    const long long ntiles = (long long)B * (BH / 8) * (BW / 16);
    long long gx = per_tile ? B : ntiles;
    if (gx > WIDE_MAX_BLOCKS) gx = WIDE_MAX_BLOCKS;
    if (ep.stats && !per_tile && gx > nslabs) gx = nslabs;
    const int np = wide_npw(form, NOUT);
    const int nn = form == W_PIX ? NOUT / 4 : NOUT;
    const int passes = (nn + 16 * np - 1) / (16 * np), nchunks = (CIN + WKC - 1) / WKC;
    const dim3 grid((unsigned)gx, (unsigned)passes), pgrid((unsigned)nchunks, (unsigned)passes);
    // persistent grid = what is resident at once (workgroups per CU by registers / LDS x 256 CUs): with more, the late
    // starters run on a half-empty machine
#define DM_WL(F, TP, NP_)                                                                                           \
    {                                                                                                               \
        static int occ = 0;                                                                                         \
        if (!occ && hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, conv_wide_kernel<F, TP, NP_>, 256, 0) != hipSuccess) \
            occ = 2;                                                                                                \
        const unsigned cap = (unsigned)(occ > 0 ? occ : 1) * 256u;                                                  \
        dim3 grid2 = grid;                                                                                          \
        if (grid2.x * grid2.y > cap) grid2.x = cap / grid2.y > 0 ? cap / grid2.y : 1;                               \
        hipLaunchKernelGGL((wide_pack_kernel<F, TP, NP_>), pgrid, dim3(256), 0, st, wv, scratch, CIN, NOUT);        \
        hipLaunchKernelGGL((conv_wide_kernel<F, TP, NP_>), grid2, dim3(256), 0, st, in, (const float *)scratch, out, ep, B,  \
                           Cphys, CIN, NOUT, H, W, nslabs, per_tile);                                               \
    }
    DM_WIDE_SWITCH(DM_WL)
#undef DM_WL
    return 0;
}

// streaming forms of the memory-bound layers (wide_stream.hip)
bool dm_stream_wgrad1x1_shape(int B, int CS, int CT, int Hs, int Ws);
int dm_stream_wgrad1x1_slabs(int B, int Hs, int Ws);
bool dm_stream_wgrad1x1(const Operand &S, const Operand &T, float *slabs, int B, int CS, int CT, int Hs, int Ws, int nslabs,
                        hipStream_t st);

bool dm_stream_wgrad_s2_thin_shape(int B, int CS, int CT, int Hs, int Ws);
int dm_stream_wgrad_s2_thin_slabs(int B, int Hs, int Ws);
bool dm_stream_wgrad_s2_thin(const Operand &S, const Operand &T, float *slabs, int B, int CS, int CT, int Hs, int Ws, int nslabs,
                             hipStream_t st);

// T as an AFFINE2 operand (two tensors): only where the one-pass kernel prefetches both
bool dm_wide_wgrad_t_affine2_ok(int CS, int CT, int Hs, int Ws, int k);

bool dm_wide_wgrad_ok(int Hs, int Ws) { return !(wide_disabled() & 2) && Hs > 0 && Ws > 0 && Hs % 8 == 0 && Ws % 16 == 0; }

static void wide_wgrad_grid(int CS, int CT, int k, int &gy, int &gz, int &cap)
{
    const int nctp = k == 4 ? WgGeom<4>::NCTP : (k == 3 ? WgGeom<3>::NCTP : WgGeom<1>::NCTP);
    gy = (CT + nctp - 1) / nctp;
    gz = (CS + 63) / 64;
    cap = 1024 / (gy * gz);
    if (cap < 32) cap = 32;
    if (cap > 512) cap = 512;
}

static int wide_wgrad_one_pass()
{
    static const int v = getenv("DM_WIDE_WGRAD1") ? atoi(getenv("DM_WIDE_WGRAD1")) : 1;
    return v;
}
static bool wide_wgrad1_shape(int CS, int CT, int k) { return wide_wgrad_one_pass() && CS == 64 && ((k == 3 && CT == 64) || (k == 4 && CT == 32)); }

bool dm_wide_wgrad_t_affine2_ok(int CS, int CT, int Hs, int Ws, int k)
{
    return k == 4 && wide_wgrad1_shape(CS, CT, k) && dm_wide_wgrad_ok(Hs, Ws);
}

int dm_wide_wgrad_slabs(int B, int CS, int CT, int Hs, int Ws, int k)
{
    if (k == 1 && dm_stream_wgrad1x1_shape(B, CS, CT, Hs, Ws)) return dm_stream_wgrad1x1_slabs(B, Hs, Ws);
    if (k == 4 && dm_stream_wgrad_s2_thin_shape(B, CS, CT, Hs, Ws)) return dm_stream_wgrad_s2_thin_slabs(B, Hs, Ws);
    if (wide_wgrad1_shape(CS, CT, k)) {
        const long long units = (long long)B * (Hs / 8) * (Ws / 16);
        return (int)(units < 256 ? units : 256);                // one workgroup of eight waves per CU
    }
    int gy, gz, cap;
    wide_wgrad_grid(CS, CT, k, gy, gz, cap);
    const long long units = (long long)B * (Hs / 8) * (Ws / 16);
    return (int)(units < cap ? units : cap);
}

int dm_wide_wgrad(const Operand &S, const Operand &T, float *slabs, int B, int CS, int CT, int CTphys, int Hs, int Ws,
                  int k, int nslabs, hipStream_t st)
{
    if (k == 1 && dm_stream_wgrad1x1(S, T, slabs, B, CS, CT, Hs, Ws, nslabs, st)) return 0;
    if (k == 4 && CT == CTphys && dm_stream_wgrad_s2_thin(S, T, slabs, B, CS, CT, Hs, Ws, nslabs, st)) return 0;
    if (wide_wgrad1_shape(CS, CT, k) && CT == CTphys && !T.ones && (T.mode != DM_LOAD_AFFINE2 || (k == 4 && S.mode != DM_LOAD_AFFINE2)) &&
        !(S.mode >= DM_LOAD_AFFINE && S.coef_bstride) && !(T.mode >= DM_LOAD_AFFINE && T.coef_bstride)) {
        const long long units = (long long)B * (Hs / 8) * (Ws / 16);
        int g1 = (int)(units < 256 ? units : 256);
        if (g1 > nslabs) g1 = nslabs;
        const bool two = S.mode == DM_LOAD_AFFINE2;
        if (T.mode == DM_LOAD_AFFINE2) {       // the decoder's first transposed convolution: T = BatchNorm backward of its output gradient
            hipLaunchKernelGGL((wgrad_wide1_kernel<4, 32, false, true>), dim3(g1), dim3(512), 0, st, S, T, slabs, B, Hs, Ws, nslabs);
            return 0;
        }
        if (k == 3 && two) hipLaunchKernelGGL((wgrad_wide1_kernel<3, 64, true>), dim3(g1), dim3(512), 0, st, S, T, slabs, B, Hs, Ws, nslabs);
        else if (k == 3) hipLaunchKernelGGL((wgrad_wide1_kernel<3, 64, false>), dim3(g1), dim3(512), 0, st, S, T, slabs, B, Hs, Ws, nslabs);
        else if (two) hipLaunchKernelGGL((wgrad_wide1_kernel<4, 32, true>), dim3(g1), dim3(512), 0, st, S, T, slabs, B, Hs, Ws, nslabs);
        else hipLaunchKernelGGL((wgrad_wide1_kernel<4, 32, false>), dim3(g1), dim3(512), 0, st, S, T, slabs, B, Hs, Ws, nslabs);
        return 0;
    }
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

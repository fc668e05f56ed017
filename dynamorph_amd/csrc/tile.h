// tile.h -- staging of an input tile (with halo) from HBM into LDS through the operand transform.
//
// Register-staged, two-phase, so a persistent workgroup can keep the NEXT tile's loads in flight while
// it runs the MFMAs and stores of the current one (the LDS buffer is single, the registers are the
// second buffer):
//   init():   per-thread constants of its N tile elements (LDS offset, row, column, channel packed in one
//             register each) -- tile independent, computed once per workgroup.
//   issue():  all of a thread's 16-byte loads (both tensors for AFFINE2) go out back to back, predicated
//             only on "inside the image"; no branch depends on the operand mode.
//   commit(): v = c0*p0 + c1*p1 + c2, optional ReLU, zero in the padding, 1 in the synthetic ones channel;
//             one ds_write_b128 per element.  The per-channel coefficients come from a small LDS table
//             (stage_coef) so commit issues no global loads.
// IDENT/RELU use c0 = 1, c1 = c2 = 0, which leaves the value bit-identical.
#pragma once
#include "dm_common.h"

constexpr int DM_COEF_MAX_C = 64;

// LDS coefficient table [C][4] = (c0, c1, c2, relu_floor) for sample b.  relu_floor = 0 (ReLU) or -inf.
__device__ __forceinline__ void stage_coef(float *__restrict__ s_coef, const Operand &op, int b, int C)
{
    const int c = threadIdx.x;
    if (c < C) {
        float c0 = 1.f, c1 = 0.f, c2 = 0.f;
        if (op.mode >= DM_LOAD_AFFINE) {
            const float *cf = op.coef + (long long)b * op.coef_bstride + c * 4;
            c0 = cf[0]; c2 = cf[2];
            if (op.mode == DM_LOAD_AFFINE2) c1 = cf[1];
        }
        const bool relu = op.mode == DM_LOAD_RELU || op.mode == DM_LOAD_AFFINE_RELU;
        *reinterpret_cast<f32x4 *>(s_coef + c * 4) = (f32x4){c0, c1, c2, relu ? 0.f : -__builtin_inff()};
    }
}

template <int CIN, int ROWS, int COLS4, int RS, int PS, bool TWO>
struct TileStage {
    static constexpr int PER_C = ROWS * COLS4;
    static constexpr int TOTAL = CIN * PER_C;
    static constexpr int N = (TOTAL + DM_BLOCK - 1) / DM_BLOCK;
    static_assert(CIN * PS < (1 << 14) && ROWS < 64 && COLS4 < 64 && CIN <= 64, "metadata packing");
    f32x4 v[N];
    f32x4 u[TWO ? N : 1];
    int meta[N];     // lds offset | r << 14 | j4 << 20 | c << 26 ; -1 = no element

    __device__ __forceinline__ void init()
    {
#pragma unroll
        for (int k = 0; k < N; ++k) {
            const int i = threadIdx.x + k * DM_BLOCK;
            const int c = i / PER_C;
            const int rem = i - c * PER_C;
            const int r = rem / COLS4;
            const int j4 = rem - r * COLS4;
            meta[k] = i < TOTAL ? ((c * PS + r * RS + 4 * j4) | (r << 14) | (j4 << 20) | (c << 26)) : -1;
        }
    }

    // ---- one element at a time, through buffer descriptors -------------------------------------------------
    // A CU keeps only so many bytes in flight: a workgroup that issues a whole tile's loads in one burst stalls
    // at the issue point until earlier requests return (measured: 24-38 % of a wave's time in the conv and
    // weight-gradient kernels).  begin() prepares the tile's descriptors, issue_one(k) requests element k; the
    // kernels call it between the MFMA steps of the current tile so the requests trickle out under the matrix
    // work.  Elements outside the image (and every element when there is no next tile: empty descriptor) carry
    // an out-of-range offset -> the load returns 0 without a branch, which is exactly the zero padding.
    struct Ctx {
        __amdgpu_buffer_rsrc_t r0, r1;
        int H, W, Cphys, gy0, gx0;
    };
    __device__ __forceinline__ Ctx begin(const Operand &op, bool live, int b, int Cphys, int H, int W, int gy0, int gx0) const
    {
        Ctx cx;
        const long long se = (long long)Cphys * H * W;
        const int bytes = live ? (int)(se * 4) : 0;
        cx.r0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(op.p0 + se * b), 0, bytes, 0x00020000);
        cx.r1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>((TWO && op.p1 ? op.p1 : op.p0) + se * b), 0,
                                                  (TWO && op.p1) ? bytes : 0, 0x00020000);
        cx.H = H; cx.W = W; cx.Cphys = Cphys; cx.gy0 = gy0; cx.gx0 = gx0;
        return cx;
    }
    __device__ __forceinline__ void issue_one(int k, const Ctx &cx)
    {
        const int mt = meta[k];
        const int r = (mt >> 14) & 63, j4 = (mt >> 20) & 63, c = (mt >> 26) & 63;
        const int gy = cx.gy0 + r, gx = cx.gx0 + 4 * j4;
        const bool ok = mt >= 0 && c < cx.Cphys && (unsigned)gy < (unsigned)cx.H && (unsigned)gx < (unsigned)cx.W;
        const int voff = ok ? ((c * cx.H + gy) * cx.W + gx) * 4 : 0x7ffffff0;
        v[k] = __builtin_amdgcn_raw_buffer_load_b128(cx.r0, voff, 0, 0);
        if (TWO) u[k] = __builtin_amdgcn_raw_buffer_load_b128(cx.r1, voff, 0, 0);
    }

    __device__ __forceinline__ void issue(const Operand &op, int b, int Cphys, int H, int W, int gy0, int gx0)
    {
        // tensors stay below 2^31 elements (checked on the host): 32-bit element offsets
        const int base = ((b * Cphys) * H + gy0) * W + gx0;
#pragma unroll
        for (int k = 0; k < N; ++k) {
            const int mt = meta[k];
            const int r = (mt >> 14) & 63, j4 = (mt >> 20) & 63, c = (mt >> 26) & 63;
            const int gy = gy0 + r, gx = gx0 + 4 * j4;
            const bool ok = mt >= 0 && c < Cphys && gy >= 0 && gy < H && gx >= 0 && gx < W;
            const int off = base + (c * H + r) * W + 4 * j4;
            v[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (ok) v[k] = *reinterpret_cast<const f32x4 *>(op.p0 + off);
            if (TWO) {
                u[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (ok && op.p1) u[k] = *reinterpret_cast<const f32x4 *>(op.p1 + off);   // p1 == NULL: not AFFINE2
            }
        }
    }

    // mode: the operand's load mode (workgroup uniform).  IDENT tiles skip the transform altogether (their padding is
    // already zero: issue() leaves unloaded elements at 0); RELU tiles only clamp; the AFFINE family takes the
    // per-channel coefficients from the LDS table.
    __device__ __forceinline__ void commit(float *__restrict__ lds, const float *__restrict__ s_coef, int Cphys,
                                           int H, int W, int gy0, int gx0, int mode)
    {
        if (mode == DM_LOAD_IDENT || mode == DM_LOAD_RELU) {
            const float fl = mode == DM_LOAD_RELU ? 0.f : -__builtin_inff();
#pragma unroll
            for (int k = 0; k < N; ++k) {
                const int mt = meta[k];
                f32x4 val = v[k];
                if (mode == DM_LOAD_RELU) {
                    // select instead of fmaxf: a NaN activation stays NaN (torch.relu semantics)
                    val.x = val.x < fl ? fl : val.x; val.y = val.y < fl ? fl : val.y;
                    val.z = val.z < fl ? fl : val.z; val.w = val.w < fl ? fl : val.w;
                }
                const int c = (mt >> 26) & 63;
                if (c >= Cphys) {                          // synthetic ones channel: 1 inside the image, 0 in the padding
                    const int r = (mt >> 14) & 63, j4 = (mt >> 20) & 63;
                    const int gy = gy0 + r, gx = gx0 + 4 * j4;
                    const float one = (gy >= 0 && gy < H && gx >= 0 && gx < W) ? 1.f : 0.f;
                    val = (f32x4){one, one, one, one};
                }
                if (mt >= 0) *reinterpret_cast<f32x4 *>(lds + (mt & 0x3fff)) = val;
            }
            return;
        }
#pragma unroll
        for (int k = 0; k < N; ++k) {
            const int mt = meta[k];
            const int r = (mt >> 14) & 63, j4 = (mt >> 20) & 63, c = (mt >> 26) & 63;
            const int gy = gy0 + r, gx = gx0 + 4 * j4;
            const bool inside = gy >= 0 && gy < H && gx >= 0 && gx < W;
            f32x4 val;
            if (c < Cphys) {
                const f32x4 cf = *reinterpret_cast<const f32x4 *>(s_coef + c * 4);
                // padding: v (and u) are 0 there, so only the shift has to go; relu_floor <= 0 leaves the 0 alone
                const float cz = inside ? cf.z : 0.f;
                val = cf.x * v[k] + cz;
                if (TWO) val += cf.y * u[k];
                // select instead of fmaxf: a NaN activation stays NaN (torch.relu semantics)
                val.x = val.x < cf.w ? cf.w : val.x; val.y = val.y < cf.w ? cf.w : val.y;
                val.z = val.z < cf.w ? cf.w : val.z; val.w = val.w < cf.w ? cf.w : val.w;
            } else {
                const float one = inside ? 1.f : 0.f;
                val = (f32x4){one, one, one, one};
            }
            if (mt >= 0) *reinterpret_cast<f32x4 *>(lds + (mt & 0x3fff)) = val;
        }
    }
};

// tile.h -- staging of an input tile (with halo) from HBM into LDS through the operand transform.
//
// Register-staged, two-phase, so a persistent workgroup can keep the NEXT tile's loads in flight while
// it runs the MFMAs and stores of the current one (the LDS buffer is single, the registers are the
// second buffer).
//
// Element mapping (round 2): a thread owns the SAME position (row r, 16-byte column j4) of the tile in every channel
// step, so everything that depends on the position -- inside-the-image test, global offset, LDS offset -- is worked
// out once per tile (a handful of vector instructions), and an element costs its load, its transform and its
// ds_write_b128 only.  (Round 1 spread consecutive elements over the threads regardless of channel: ~30 vector
// instructions of decoding and bounds tests per element, in a dozen small basic blocks each with its own wait.)
// A channel plane of PER_C = ROWS * COLS4 elements is covered by P = ceil(PER_C / 256) passes of the workgroup; planes
// smaller than half a workgroup put G = 256 / PER_C channels side by side.  Element k = (channel step k / P, pass
// k % P) is channel G * (k / P) + cs of this thread's pass-p position.
//   init(H, W): per-thread constants of its P positions -- tile independent, computed once per workgroup.
//   begin():    the tile's buffer descriptors and, per position, the byte offset or an out-of-range marker.
//   issue_one(k): ONE 16-byte load per tensor; elements outside the image, channels past the physical ones and every
//               element when there is no next tile (empty descriptor) fail the range check and read as 0 = the padding.
//   commit():   v = c0*p0 + c1*p1 + c2, optional ReLU, zero in the padding, 1 in the synthetic ones channel;
//               one ds_write_b128 per element.  The per-channel coefficients come from a small LDS table
//               (stage_coef) so commit issues no global loads.
// IDENT/RELU use c0 = 1, c1 = c2 = 0, which leaves the value bit-identical.
#pragma once
#include "dm_common.h"

constexpr int DM_COEF_MAX_C = 64;

// LDS coefficient table [C][4] = (c0, c1, c2, relu_floor) for sample b.  relu_floor = 0 (ReLU) or -inf.
// Two halves, so a persistent kernel can request the next sample's row right after a tile is committed and write it
// into LDS at the end of the tile (the load's latency would otherwise sit between two barriers, once per tile):
//   coef_fetch: the global load of this thread's row (threads < C), nothing waits on it;
//   coef_put:   the row into the table.
// coef_changes: whether the table staged for sample cb is stale for sample nb (per-sample coefficients only: batch
// statistics, the training case, are staged once per workgroup).
__device__ __forceinline__ bool coef_changes(const Operand &op, int cb, int nb)
{
    return op.mode >= DM_LOAD_AFFINE && op.coef_bstride != 0 && nb != cb;
}
__device__ __forceinline__ f32x4 coef_fetch(const Operand &op, int b, int C)
{
    f32x4 raw = {1.f, 0.f, 0.f, 0.f};
    if ((int)threadIdx.x < C && op.mode >= DM_LOAD_AFFINE)
        raw = *reinterpret_cast<const f32x4 *>(op.coef + (long long)b * op.coef_bstride + threadIdx.x * 4);
    return raw;
}
__device__ __forceinline__ void coef_put(float *__restrict__ s_coef, const Operand &op, f32x4 raw, int C)
{
    const int c = threadIdx.x;
    if (c < C) {
        const float c1 = op.mode == DM_LOAD_AFFINE2 ? raw.y : 0.f;
        const bool relu = op.mode == DM_LOAD_RELU || op.mode == DM_LOAD_AFFINE_RELU;
        *reinterpret_cast<f32x4 *>(s_coef + c * 4) = (f32x4){raw.x, c1, raw.z, relu ? 0.f : -__builtin_inff()};
    }
}
__device__ __forceinline__ void stage_coef(float *__restrict__ s_coef, const Operand &op, int b, int C)
{
    coef_put(s_coef, op, coef_fetch(op, b, C), C);
}

typedef unsigned short dm_u16x2 __attribute__((ext_vector_type(2)));

// ReLU that keeps EVERY NaN, as torch.relu does: (v < 0) ? 0 : v -- the ordered compare is false for a NaN of either
// sign, so it passes through; -0 < 0 is false as well, so -0 stays -0 (as it does through ATen's clamp_min).
// v_max_f32 would return 0 for a NaN; the one-instruction integer form max(bits, 0) used until round 2 kept only the
// positive NaNs and turned a NaN with the sign bit set (x86's default 0/0 = 0xFFC00000 arriving in the data, and
// whatever FMAs propagate from it) into +0.  Costs a compare + a select per value.
__device__ __forceinline__ f32x4 relu_keep_nan(f32x4 v) { return dm_relu4(v); }

// ---- split-bf16 operands (backward kernels) ------------------------------------------------------------------------------
// A float v is stored in its own 4-byte slot as the pair (hi, lo) of bf16 values, hi = bf16(v) in the low half, lo =
// bf16(v - hi) in the high half: v = hi + lo + r with |r| <= 2^-18 |v|.  Eight such slots (four registers of a lane) are
// the A or B operand of one v_mfma_f32_16x16x32_bf16, whose 32 k-slots then are (hi, lo) of four K-steps of the f32
// kernels; with P_a, P_b the packed operands,
//     mfma(P_a, P_b) + mfma(rot16(P_a), P_b) = sum (a_hi + a_lo) (b_hi + b_lo)      (rot16 swaps hi and lo of every slot)
// i.e. the fp32 product to ~2^-17 relative, in 2 x 16 cycles for four K-steps instead of 4 x 32 on the f32-input
// instruction -- same LDS layouts, same lane mapping, same accumulator layout.  Used for GRADIENTS only (their consumers
// are Adam and tests gated on the float64 yardstick); the forward pass, whose latents pick the codes, stays exact.
__device__ __forceinline__ f32x4 split_pack4(f32x4 v)
{
    typedef __bf16 dm_bf16x2 __attribute__((ext_vector_type(2)));
    const unsigned h01 = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){v.x, v.y}, dm_bf16x2));
    const unsigned h23 = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){v.z, v.w}, dm_bf16x2));
    const float r0 = v.x - __builtin_bit_cast(float, h01 << 16), r1 = v.y - __builtin_bit_cast(float, h01 & 0xffff0000u);
    const float r2 = v.z - __builtin_bit_cast(float, h23 << 16), r3 = v.w - __builtin_bit_cast(float, h23 & 0xffff0000u);
    const unsigned l01 = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){r0, r1}, dm_bf16x2));
    const unsigned l23 = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){r2, r3}, dm_bf16x2));
    // (hi_i | lo_i << 16): bytes 0,1 of h, bytes 0,1 of l  /  bytes 2,3 of h, bytes 2,3 of l
    const unsigned p0 = __builtin_amdgcn_perm(l01, h01, 0x05040100u), p1 = __builtin_amdgcn_perm(l01, h01, 0x07060302u);
    const unsigned p2 = __builtin_amdgcn_perm(l23, h23, 0x05040100u), p3 = __builtin_amdgcn_perm(l23, h23, 0x07060302u);
    typedef unsigned dm_u32x4 __attribute__((ext_vector_type(4)));
    return __builtin_bit_cast(f32x4, (dm_u32x4){p0, p1, p2, p3});
}
__device__ __forceinline__ float split_pack1(float v)
{
    typedef __bf16 dm_bf16x2 __attribute__((ext_vector_type(2)));
    const unsigned h = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){v, 0.f}, dm_bf16x2)) & 0xffffu;
    const float r = v - __builtin_bit_cast(float, h << 16);
    const unsigned l = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){r, 0.f}, dm_bf16x2)) & 0xffffu;
    return __builtin_bit_cast(float, h | (l << 16));
}

// BLOCK: threads of the workgroup (256 everywhere except the fused backward kernel, conv_bwd_fused.hip: 512)
template <int CIN, int ROWS, int COLS4, int RS, int PS, bool TWO, int BLOCK = DM_BLOCK>
struct TileStage {
    static constexpr int PER_C = ROWS * COLS4;
    static constexpr int G = PER_C < BLOCK ? BLOCK / PER_C : 1;             // channels side by side across the workgroup
    static constexpr int P = (PER_C + BLOCK - 1) / BLOCK;                   // passes over one channel plane
    static constexpr int NC = (CIN + G - 1) / G;                            // channel steps
    static constexpr int N = NC * P;
    static constexpr unsigned OOB = 0x80000000u;   // + any channel offset (< 2^31) stays past every sample, never wraps
    static_assert(CIN * PS < (1 << 14) && ROWS < 0x4000 && COLS4 < 0x4000 && CIN <= 64, "LDS offsets fit ds immediates");
    f32x4 v[N];
    f32x4 u[TWO ? N : 1];
    int rj[P];       // r | j4 << 16 of this thread's pass-p position; r = 0x7fff: no such position (never inside)
    int loff[P];     // LDS float offset of (channel cs, r, j4)
    int goff[P];     // byte offset of (channel cs, row r, column 4*j4) from the sample's (channel 0, row 0, column 0)
    int cs;          // channel sub-index of this thread (< G)
    int HW4;         // bytes of one channel plane (uniform)

    __device__ __forceinline__ void init(int H, int W) { init(H, W, (int)threadIdx.x); }
    // t: this thread's index among the BLOCK threads that stage the tile (a role-split kernel: not threadIdx.x)
    __device__ __forceinline__ void init(int H, int W, int t)
    {
        cs = G > 1 ? t / PER_C : 0;
        const int e0 = G > 1 ? t - cs * PER_C : t;
        HW4 = H * W * 4;
#pragma unroll
        for (int p = 0; p < P; ++p) {
            const int e = e0 + p * BLOCK;
            const bool have = e < PER_C && cs < G;
            const int r = e / COLS4, j4 = e - r * COLS4;
            rj[p] = have ? (r | (j4 << 16)) : 0x7fff;
            loff[p] = cs * PS + r * RS + 4 * j4;
            goff[p] = ((cs * H + r) * W + 4 * j4) * 4;
        }
    }

    // inside the image?  (gy, gx / 4) = (r, j4) + (gy0, gx0 / 4) as two 16-bit lanes; inside iff both are below
    // (H, W / 4) as UNSIGNED numbers (a negative coordinate wraps to a large one).  Three vector instructions.
    __device__ __forceinline__ bool inside(int p, int H, int W, int gy0, int gx0) const
    {
        const dm_u16x2 org = {(unsigned short)gy0, (unsigned short)(gx0 >> 2)};
        const dm_u16x2 lim = {(unsigned short)(H - 1), (unsigned short)((W >> 2) - 1)};
        const dm_u16x2 pos = __builtin_bit_cast(dm_u16x2, rj[p]) + org;
        const dm_u16x2 cl = __builtin_elementwise_min(pos, lim);
        return __builtin_bit_cast(int, cl) == __builtin_bit_cast(int, pos);
    }

    // ---- one element at a time, through buffer descriptors -------------------------------------------------
    // A CU keeps only so many bytes in flight: a workgroup that issues a whole tile's loads in one burst stalls
    // at the issue point until earlier requests return (measured: 24-38 % of a wave's time in the conv and
    // weight-gradient kernels).  begin() prepares the tile's descriptors, issue_one(k) requests element k; the
    // kernels call it between the MFMA steps of the current tile so the requests trickle out under the matrix
    // work.
    struct Ctx {
        __amdgpu_buffer_rsrc_t r0, r1;
        unsigned voff[P];
    };
    __device__ __forceinline__ Ctx begin(const Operand &op, bool live, int b, int Cphys, int H, int W, int gy0, int gx0) const
    {
        Ctx cx;
        const long long se = (long long)Cphys * H * W;
        const int bytes = live ? (int)(se * 4) : 0;
        cx.r0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(op.p0 + se * b), 0, bytes, 0x00020000);
        cx.r1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>((TWO && op.p1 ? op.p1 : op.p0) + se * b), 0,
                                                  (TWO && op.p1) ? bytes : 0, 0x00020000);
        const int org = (gy0 * W + gx0) * 4;               // (uniform; may be negative: only used where inside)
#pragma unroll
        for (int p = 0; p < P; ++p) cx.voff[p] = inside(p, H, W, gy0, gx0) ? (unsigned)(goff[p] + org) : OOB;
        return cx;
    }
    __device__ __forceinline__ void issue_one(int k, const Ctx &cx)
    {
        const int p = k % P, step = k / P;
        // channels past Cphys (the synthetic ones channel, the unused tail of the last channel step) land past the
        // descriptor's range like everything else that must read as 0
        const unsigned vo = cx.voff[p] + (unsigned)(step * G) * (unsigned)HW4;
        v[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(cx.r0, (int)vo, 0, 0));
        if (TWO) u[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(cx.r1, (int)vo, 0, 0));
    }

    // all of a tile's loads back to back (the first tile of a workgroup: nothing to hide them under)
    __device__ __forceinline__ void issue(const Operand &op, int b, int Cphys, int H, int W, int gy0, int gx0)
    {
        const Ctx cx = begin(op, true, b, Cphys, H, W, gy0, gx0);
#pragma unroll
        for (int k = 0; k < N; ++k) issue_one(k, cx);
    }

    // mode: the operand's load mode (workgroup uniform).  IDENT tiles skip the transform altogether (their padding is
    // already zero: elements outside the image were read as 0); RELU tiles only clamp; the AFFINE family takes the
    // per-channel coefficients from the LDS table.
    // KIND 0: identity, 1: ReLU, 2: affine (AFFINE, AFFINE2), 3: affine + ReLU.  ONES: a synthetic ones channel follows
    // the physical channels.
    // One straight-line body per (KIND, ONES): the mode tests stay out of the element loop, so the coefficient reads
    // of a pass are requested together and the loop is a single basic block.
    // WT ("whole-wave tail"): the caller staged with t = threadIdx.x and PER_C is a multiple of 64, so the threads without a
    // position are whole waves and the test is a scalar branch -- no exec masking, no divergent join.  (Not only cheaper: in a
    // kernel at the 256-register limit hipcc placed live-range copies in that join block AHEAD of the instruction that
    // restores exec, and the wave that had skipped the block -- exec = 0 -- skipped the copies with it:
    // dec_tail_backward_kernel<4, false, true>.)
    template <int KIND, bool ONES, bool SPLIT = false, bool WT = false>
    __device__ __forceinline__ void commit_as(float *__restrict__ lds, const float *__restrict__ s_coef, int Cphys,
                                              int H, int W, int gy0, int gx0)
    {
        static_assert(!WT || (PER_C % 64 == 0 && BLOCK % 64 == 0), "whole-wave tail needs PER_C in whole waves");
#pragma unroll
        for (int p = 0; p < P; ++p) {
            // no position in this pass (at most the tail of the workgroup)
            if (WT) { if ((__builtin_amdgcn_readfirstlane(rj[p]) & 0xffff) == 0x7fff) continue; }
            else if ((rj[p] & 0xffff) == 0x7fff) continue;
            const bool in = inside(p, H, W, gy0, gx0);
            float *__restrict__ dst = lds + loff[p];
            const f32x4 *__restrict__ ctab = reinterpret_cast<const f32x4 *>(s_coef) + cs;
            f32x4 cf[KIND >= 2 ? NC : 1];
            if constexpr (KIND >= 2) {
#pragma unroll
                for (int step = 0; step < NC; ++step) cf[step] = ctab[step * G];
            }
#pragma unroll
            for (int step = 0; step < NC; ++step) {
                const int k = step * P + p;
                f32x4 val = v[k];
                if constexpr (KIND == 1) val = relu_keep_nan(val); else if constexpr (KIND >= 2) {
                    const f32x4 c = cf[step];
                    // padding: v (and u) are 0 there, so only the shift has to go; the ReLU leaves that 0 alone
                    const float cz = in ? c.z : 0.f;
                    val = c.x * val + cz;
                    if (TWO) val += c.y * u[k];
                    if constexpr (KIND == 3) val = relu_keep_nan(val);
                }
                if constexpr (ONES) {                      // 1 inside the image, 0 in the padding
                    const float one = in ? 1.f : 0.f;
                    if (step * G + cs >= Cphys) val = (f32x4){one, one, one, one};
                }
                if constexpr (SPLIT) val = split_pack4(val);       // (0 stays 0: the padding needs no special case)
                if (CIN % G != 0 && step == NC - 1) {      // tail of the last channel step
                    if (cs < CIN - step * G) *reinterpret_cast<f32x4 *>(dst + step * G * PS) = val;
                } else {
                    *reinterpret_cast<f32x4 *>(dst + step * G * PS) = val;
                }
            }
        }
    }

    // mode: the operand's load mode (workgroup uniform).  IDENT tiles skip the transform altogether (their padding is
    // already zero: elements outside the image were read as 0); RELU tiles only clamp; the AFFINE family takes the
    // per-channel coefficients from the LDS table.
    // SPLIT: every value is stored as its (hi, lo) bf16 pair (split_pack4) for the split-bf16 matrix products
    template <bool SPLIT = false, bool WT = false>
    __device__ __forceinline__ void commit(float *__restrict__ lds, const float *__restrict__ s_coef, int Cphys,
                                           int H, int W, int gy0, int gx0, int mode)
    {
        if (CIN > 1 && Cphys < CIN) {                      // (uniform) with a synthetic ones channel: rare, one body
            if (mode == DM_LOAD_IDENT) commit_as<0, true, SPLIT, WT>(lds, s_coef, Cphys, H, W, gy0, gx0);
            else if (mode == DM_LOAD_RELU) commit_as<1, true, SPLIT, WT>(lds, s_coef, Cphys, H, W, gy0, gx0);
            else if (mode == DM_LOAD_AFFINE_RELU) commit_as<3, true, SPLIT, WT>(lds, s_coef, Cphys, H, W, gy0, gx0);
            else commit_as<2, true, SPLIT, WT>(lds, s_coef, Cphys, H, W, gy0, gx0);
        } else if (mode == DM_LOAD_IDENT) {
            commit_as<0, false, SPLIT, WT>(lds, s_coef, Cphys, H, W, gy0, gx0);
        } else if (mode == DM_LOAD_RELU) {
            commit_as<1, false, SPLIT, WT>(lds, s_coef, Cphys, H, W, gy0, gx0);
        } else if (mode == DM_LOAD_AFFINE_RELU) {
            commit_as<3, false, SPLIT, WT>(lds, s_coef, Cphys, H, W, gy0, gx0);
        } else {
            commit_as<2, false, SPLIT, WT>(lds, s_coef, Cphys, H, W, gy0, gx0);
        }
    }
};

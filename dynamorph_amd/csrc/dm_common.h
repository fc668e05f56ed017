// dm_common.h -- shared host/device helpers for libdynamorph_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/dynamorph_hip.h"

// The opt-in split-bf16 operands of the gradient kernels (rounds 4-5: fp32 values as bf16 head + remainder pairs on the bf16
// matrix instruction) are no longer built: since the fp32 kernels were restructured the narrow layers (4-32 channels) are
// bound by their vector instructions, and the split -- 12 more of those per tile -- ran 2.7 % SLOWER than the exact fp32
// step (BENCH_r05: 1.941 against 1.8905 ms).  The templates keep their BF parameter; -DDM_BUILD_SPLIT_BF16=1 instantiates
// the split forms again (measurements only).
#ifndef DM_BUILD_SPLIT_BF16
#define DM_BUILD_SPLIT_BF16 0
#endif

#define DM_WAVE 64
#define DM_BLOCK 256          // 4 waves: one per SIMD of a CU

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// ------------------------------------------------------------------ host errors
void dm_set_error(const char *fmt, ...);
#define DM_REQUIRE(cond, ...)                         \
    do {                                              \
        if (!(cond)) {                                \
            dm_set_error(__VA_ARGS__);                \
            return -1;                                \
        }                                             \
    } while (0)

static inline int dm_launch_status(const char *what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        dm_set_error("%s: launch failed: %s", what, hipGetErrorString(e));
        return (int)e;
    }
    return 0;
}

// ------------------------------------------------------------- device operands
// Device-side copy of dm_operand (same fields; kept POD so it travels as a kernel argument).
struct Operand {
    const float *p0;
    const float *p1;
    const float *coef;
    long long coef_bstride;
    int mode;
    int ones;
};

static inline Operand to_dev(const dm_operand *o)
{
    Operand r;
    r.p0 = o->p0; r.p1 = o->p1; r.coef = o->coef; r.coef_bstride = o->coef_bstride;
    r.mode = o->mode; r.ones = o->ones_channel;
    return r;
}

static inline Operand null_operand()
{
    Operand r;
    r.p0 = nullptr; r.p1 = nullptr; r.coef = nullptr; r.coef_bstride = 0; r.mode = DM_LOAD_IDENT; r.ones = 0;
    return r;
}

struct WeightView {
    const float *w;
    long long off, sn, sc, sky, skx;
    float *scratch;
    long long scratch_floats;
};

static inline WeightView to_dev(const dm_weight_view *v)
{
    WeightView r;
    r.w = v->w; r.off = v->off; r.sn = v->sn; r.sc = v->sc; r.sky = v->sky; r.skx = v->skx;
    r.scratch = v->scratch; r.scratch_floats = v->scratch_floats;
    return r;
}

struct Epilogue {
    const float *bias;
    const float *bias_border;
    int relu;
    Operand mask;
    const float *resid;
    const float *stat_q;
    double *stats;
};

static inline Epilogue to_dev(const dm_epilogue *e)
{
    Epilogue r;
    if (!e) {
        r.bias = nullptr; r.bias_border = nullptr; r.relu = 0; r.mask = null_operand(); r.resid = nullptr; r.stat_q = nullptr; r.stats = nullptr;
        return r;
    }
    r.bias = e->bias; r.bias_border = e->bias_border; r.relu = e->relu; r.mask = to_dev(&e->mask); r.resid = e->resid;
    r.stat_q = e->stat_q; r.stats = e->stats;
    return r;
}

static inline int dm_check_operand(const dm_operand *o, const char *who)
{
    DM_REQUIRE(o && o->p0, "%s: operand p0 is NULL", who);
    DM_REQUIRE(o->mode >= DM_LOAD_IDENT && o->mode <= DM_LOAD_AFFINE2, "%s: bad operand mode %d", who, o->mode);
    DM_REQUIRE(o->mode < DM_LOAD_AFFINE || o->coef, "%s: operand mode %d needs coef", who, o->mode);
    DM_REQUIRE(o->mode != DM_LOAD_AFFINE2 || o->p1, "%s: AFFINE2 operand needs p1", who);
    return 0;
}

#ifdef __HIPCC__
// ReLU as torch.relu defines it: negative numbers to 0, every NaN (either sign) kept -- fmaxf / v_max_f32 would return 0
// for a NaN and hide it from everything downstream.  gfx950 has the IEEE 754-2019 maximum, which propagates NaNs, as ONE
// instruction (v_maximum3_f32 v, v, 0, 0); the compare + select form costs two, and the vector instructions of every load
// transform and epilogue share the issue port with the matrix instructions.
__device__ __forceinline__ float dm_relu(float v) { return __builtin_elementwise_maximum(v, 0.f); }
__device__ __forceinline__ f32x4 dm_relu4(f32x4 v) { return (f32x4){dm_relu(v.x), dm_relu(v.y), dm_relu(v.z), dm_relu(v.w)}; }

// Load 4 contiguous elements of channel c / sample b through the operand's transform.
__device__ __forceinline__ f32x4 operand_load4(const Operand &op, long long off, int b, int c)
{
    f32x4 v = *reinterpret_cast<const f32x4 *>(op.p0 + off);
    if (op.mode == DM_LOAD_IDENT) return v;
    if (op.mode == DM_LOAD_RELU) return dm_relu4(v);
    const float *cf = op.coef + (long long)b * op.coef_bstride + c * 4;
    const float c0 = cf[0], c2 = cf[2];
    if (op.mode == DM_LOAD_AFFINE2) {
        const float c1 = cf[1];
        f32x4 u = *reinterpret_cast<const f32x4 *>(op.p1 + off);
        v = c0 * v + (c1 * u + c2);
        return v;
    }
    v = c0 * v + c2;
    if (op.mode == DM_LOAD_AFFINE_RELU) v = dm_relu4(v);
    return v;
}

// Lane exchanges inside a 16-lane row as DPP moves (VALU, no LDS crossbar round trip; __shfl_xor compiles to
// ds_bpermute_b32).  xor 1/2: quad_perm; xor 4: row_half_mirror then quad reverse; xor 8: row_ror:8.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float lane_xor1(float v) { return dpp_mov<0xB1>(v); }
__device__ __forceinline__ float lane_xor2(float v) { return dpp_mov<0x4E>(v); }
__device__ __forceinline__ float lane_xor4(float v) { return dpp_mov<0x1B>(dpp_mov<0x141>(v)); }
__device__ __forceinline__ float lane_xor8(float v) { return dpp_mov<0x128>(v); }
__device__ __forceinline__ f32x4 lane_xor1(f32x4 v) { return (f32x4){lane_xor1(v.x), lane_xor1(v.y), lane_xor1(v.z), lane_xor1(v.w)}; }
__device__ __forceinline__ f32x4 lane_xor4(f32x4 v) { return (f32x4){lane_xor4(v.x), lane_xor4(v.y), lane_xor4(v.z), lane_xor4(v.w)}; }
__device__ __forceinline__ f32x4 lane_xor8(f32x4 v) { return (f32x4){lane_xor8(v.x), lane_xor8(v.y), lane_xor8(v.z), lane_xor8(v.w)}; }

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Sum of a double over the four 16-lane rows of the wave (lanes l, l ^ 16, l ^ 32, l ^ 48), every lane gets it:
// v_permlane16_swap / v_permlane32_swap exchange registers between partner rows without an LDS round trip (a __shfl_xor of
// a double is two ds_bpermute_b32).
__device__ __forceinline__ double dm_row_sum_f64(double v)
{
#pragma unroll
    for (int step = 0; step < 2; ++step) {
        const unsigned lo = (unsigned)(__builtin_bit_cast(unsigned long long, v) & 0xffffffffull);
        const unsigned hi = (unsigned)(__builtin_bit_cast(unsigned long long, v) >> 32);
        unsigned a0, a1, b0, b1;
        if (step == 0) {
            // (x, x): afterwards the even rows hold (own, partner's) and the odd rows (partner's, own)
            const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
            const auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
            a0 = a[0]; a1 = a[1]; b0 = b[0]; b1 = b[1];
        } else {
            const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
            const auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
            a0 = a[0]; a1 = a[1]; b0 = b[0]; b1 = b[1];
        }
        v = __builtin_bit_cast(double, ((unsigned long long)b0 << 32) | a0) + __builtin_bit_cast(double, ((unsigned long long)b1 << 32) | a1);
    }
    return v;
}

// Sum over the 256 threads of a block; result valid in thread 0.  scratch: >= 4 doubles of LDS.
__device__ __forceinline__ double block_sum(double v, double *scratch)
{
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) scratch[wave] = v;
    __syncthreads();
    double r = 0.0;
    if (threadIdx.x == 0) {
        const int nw = (blockDim.x + 63) >> 6;
        for (int i = 0; i < nw; ++i) r += scratch[i];
    }
    return r;
}

// One-off function attributes (hipFuncAttributeMaxDynamicSharedMemorySize) belong to a DEVICE's copy of the kernel: a process
// that drives a second GPU has to set them there too, so the "done" flag is per device ordinal.
struct DmPerDeviceOnce {
    bool done[64] = {};
    int dev() const { int d = 0; return (hipGetDevice(&d) == hipSuccess && d >= 0 && d < 64) ? d : -1; }
    bool need() const { const int d = dev(); return d < 0 || !done[d]; }
    void mark() { const int d = dev(); if (d >= 0) done[d] = true; }
};

#endif

#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
// variant 0: f32 16x16x4 (2 per pair of K-steps); 1: bf16 16x16x16 x2 + 2 rotates; 2: as 1 plus 8 filler VALU per pair
template <int V>
__global__ __launch_bounds__(256) void k(const unsigned* in, float* out, int iters, unsigned long long* cyc) {
    const int l = threadIdx.x;
    unsigned a0 = in[l], a1 = in[l + 256], b0 = in[l + 512], b1 = in[l + 768];
    f32x4 acc[4] = {{0,0,0,0},{0,0,0,0},{0,0,0,0},{0,0,0,0}};
    float f = 1.f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (V == 0) {
                acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(__builtin_bit_cast(float, a0), __builtin_bit_cast(float, b0), acc[c], 0, 0, 0);
                acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(__builtin_bit_cast(float, a1), __builtin_bit_cast(float, b1), acc[c], 0, 0, 0);
            } else {
                const unsigned r0 = __builtin_amdgcn_alignbit(a0, a0, 16), r1 = __builtin_amdgcn_alignbit(a1, a1, 16);
                acc[c] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, (u32x2){a0, a1}), __builtin_bit_cast(s16x4, (u32x2){b0, b1}), acc[c], 0, 0, 0);
                acc[c] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, (u32x2){r0, r1}), __builtin_bit_cast(s16x4, (u32x2){b0, b1}), acc[c], 0, 0, 0);
                if (V == 2) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) f = __builtin_fmaf(f, 1.0001f, 0.5f);
                }
            }
            a0 += 0x10001u * (c + 1); a1 ^= a0;     // keep operands changing (cheap VALU, both variants)
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    f32x4 s = acc[0] + acc[1] + acc[2] + acc[3];
    out[blockIdx.x * 256 + l] = s.x + s.y + s.z + s.w + f;
    if (l == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
    unsigned* in; float* out; unsigned long long* cyc;
    hipMalloc(&in, 4096); hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 8);
    hipMemset(in, 0x3f, 4096);
    const int iters = 2000;
    for (int wgs = 1; wgs <= 2; ++wgs)
        for (int v = 0; v < 3; ++v) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            auto launch = [&]() {
                if (v == 0) hipLaunchKernelGGL(k<0>, dim3(256 * wgs), dim3(256), 0, 0, in, out, iters, cyc);
                if (v == 1) hipLaunchKernelGGL(k<1>, dim3(256 * wgs), dim3(256), 0, 0, in, out, iters, cyc);
                if (v == 2) hipLaunchKernelGGL(k<2>, dim3(256 * wgs), dim3(256), 0, 0, in, out, iters, cyc);
            };
            launch(); hipDeviceSynchronize();
            hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
            // per iteration: 4 accumulators x 2 K-steps = 8 K-step-tiles of 16x16
            printf("waves/SIMD %d variant %d: %.3f ms, %.1f memtime-ticks per 16x16 tile K-step pair\n", wgs, v, ms, (double)c / iters / 4);
        }
    return 0;
}

// Experiment (not product): issue cost of v_fma_f32, v_pk_fma_f32 and v_mov_b32_dpp (wave shifts) on gfx950 at
// 1, 2, 3 and 4 waves per SIMD.  One workgroup per CU; cycles via s_memtime.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int IT = 256;     // IT iterations of CH independent chains
template <int MODE, int CH> __global__ void k(float *out, unsigned long long *cyc, float s)
{
    float a[CH]; f32x2 p[CH];
    for (int i = 0; i < CH; ++i) { a[i] = threadIdx.x * 0.5f + i; p[i] = (f32x2){a[i], a[i] + 1.f}; }
    const f32x2 s2 = {s, s * 0.5f};
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < IT; ++it) {
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            if (MODE == 0) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(s));
            if (MODE == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(p[i]) : "v"(s2));
            if (MODE == 2) asm volatile("v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a[i]));
            if (MODE == 3) asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a[i]));
            if (MODE == 4) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(s2));
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float r = 0.f;
    for (int i = 0; i < CH; ++i) r += a[i] + p[i].x + p[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int MODE, int CH> void run(const char *name, float *out, unsigned long long *cyc)
{
    for (int wps = 1; wps <= 4; ++wps) {
        const int threads = 256 * wps;
        hipLaunchKernelGGL((k<MODE, CH>), dim3(256), dim3(threads), 0, 0, out, cyc, 1.0001f);
        hipLaunchKernelGGL((k<MODE, CH>), dim3(256), dim3(threads), 0, 0, out, cyc, 1.0001f);
        unsigned long long h[256];
        hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
        double m = 0; for (int i = 0; i < 256; ++i) m += h[i];
        m /= 256;
        printf("%-22s chains %d waves/SIMD %d: %.2f cycles per instruction per wave, %.2f per SIMD-instruction\n", name, CH, wps,
               m / (IT * CH), m / (IT * CH) / wps);
    }
}
int main()
{
    float *out; unsigned long long *cyc;
    hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 256 * 8);
    run<0, 1>("v_fma_f32", out, cyc); run<0, 2>("v_fma_f32", out, cyc); run<0, 4>("v_fma_f32", out, cyc); run<0, 8>("v_fma_f32", out, cyc);
    run<1, 1>("v_pk_fma_f32", out, cyc); run<1, 2>("v_pk_fma_f32", out, cyc); run<1, 4>("v_pk_fma_f32", out, cyc); run<1, 8>("v_pk_fma_f32", out, cyc);
    run<2, 1>("dpp wave_shr", out, cyc); run<2, 8>("dpp wave_shr", out, cyc);
    return 0;
}

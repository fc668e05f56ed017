// Experiment (not product): inside ONE wave, do independent v_pk_fma_f32 issued after a matrix instruction execute in its
// shadow on gfx950?  One wave per SIMD; per iteration one matrix instruction (two alternating accumulators) followed by NV
// independent v_pk_fma_f32; cycles per iteration by s_memtime.
//   hipcc --offload-arch=gfx950 -O3 tools/exp/coexec_inwave_probe.hip -o tools/exp/coexec_inwave_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int IT = 1024;

template <int KIND, int NV, int WAVES, int VOP> __global__ __launch_bounds__(64 * WAVES) void k(float *out, unsigned long long *cyc, float s)
{
    f32x4 acc[2] = {(f32x4){s, s, s, s}, (f32x4){s, s, s, s}};
    f32x2 p[8];
    for (int i = 0; i < 8; ++i) p[i] = (f32x2){threadIdx.x * 0.5f + i, 1.f};
    const f32x2 s2 = {s, s * 0.5f};
    unsigned q[8];
    for (int i = 0; i < 8; ++i) q[i] = threadIdx.x + i;
    float a = threadIdx.x * 0.25f, b = s;
    bf16x8 ab, bb;
    for (int i = 0; i < 8; ++i) { ab[i] = (__bf16)(threadIdx.x * 0.01f); bb[i] = (__bf16)s; }
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < IT; ++it) {
        if (KIND == 0) acc[it & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[it & 1], 0, 0, 0);
        if (KIND == 1) acc[it & 1] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[it & 1], 0, 0, 0);
        if (KIND == 2) acc[it & 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, bb, acc[it & 1], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            if (VOP == 0) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(p[i & 7]) : "v"(s2));
            if (VOP == 1) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(p[i & 7].x) : "v"(s));
            if (VOP == 2) asm volatile("v_add_u32 %0, %0, %1" : "+v"(q[i & 7]) : "v"(threadIdx.x));
            if (VOP == 3) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i & 7]) : "v"(s2));
            if (VOP == 4) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i & 7]) : "v"(s2));
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_nop 0" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float r = acc[0].x + acc[0].y + acc[1].z + acc[1].w;
    for (int i = 0; i < 8; ++i) r += p[i].x + p[i].y + (float)q[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int KIND, int NV, int WAVES, int VOP> double one(float *out, unsigned long long *cyc)
{
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<KIND, NV, WAVES, VOP>), dim3(256), dim3(64 * WAVES), 0, 0, out, cyc, 1.0001f);
    unsigned long long h[256];
    (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double m = 0;
    for (int i = 0; i < 256; ++i) m += h[i];
    return m / 256 / IT;
}
template <int KIND, int WAVES, int VOP> void run(const char *name, float *out, unsigned long long *cyc)
{
    printf("%-26s %d wave(s)/SIMD, cycles per iteration with 0 / 2 / 4 / 6 / 8 / 12 / 16 vector instructions behind the matrix instruction: "
           "%.1f %.1f %.1f %.1f %.1f %.1f %.1f\n", name, WAVES / 4,
           one<KIND, 0, WAVES, VOP>(out, cyc), one<KIND, 2, WAVES, VOP>(out, cyc), one<KIND, 4, WAVES, VOP>(out, cyc), one<KIND, 6, WAVES, VOP>(out, cyc),
           one<KIND, 8, WAVES, VOP>(out, cyc), one<KIND, 12, WAVES, VOP>(out, cyc), one<KIND, 16, WAVES, VOP>(out, cyc));
}
int main()
{
    float *out; unsigned long long *cyc;
    (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&cyc, 256 * 8);
    const char *vn[5] = {"v_pk_fma_f32", "v_fma_f32", "v_add_u32", "v_pk_mul_f32", "v_pk_add_f32"};
#define ALL(V) printf("---- vector instruction: %s\n", vn[V]); \
    run<3, 4, V>("no matrix instruction", out, cyc); run<0, 4, V>("v_mfma_f32_16x16x4_f32", out, cyc); \
    run<1, 4, V>("v_mfma_f32_4x4x1_16B_f32", out, cyc); run<2, 4, V>("v_mfma_f32_16x16x32_bf16", out, cyc);
    ALL(0) ALL(1) ALL(2) ALL(3) ALL(4)
    return 0;
}

// Experiment (not product): do matrix instructions of one wave and vector instructions of ANOTHER wave of the same SIMD execute
// together on gfx950?  One 512-thread workgroup per CU (two waves per SIMD): waves 0-3 issue matrix instructions, waves 4-7
// v_pk_fma_f32; each group alone and both together, cycles by s_memtime (max over the waves of a workgroup, mean over CUs).
//   hipcc --offload-arch=gfx950 -O3 tools/exp/coexec_probe.hip -o tools/exp/coexec_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int IT = 512;

// KIND 0: v_mfma_f32_16x16x4_f32, 1: v_mfma_f32_4x4x1_16B_f32, 2: v_mfma_f32_16x16x32_bf16
template <int KIND> __global__ __launch_bounds__(512) void k(float *out, unsigned long long *cyc, int run_m, int run_v, float s, int swap, int prio)
{
    const int wave_raw = threadIdx.x >> 6;
    const int wave = swap ? (wave_raw ^ 4) : wave_raw;      // swap: the YOUNGER waves (4-7) issue the matrix instructions
    if (prio && wave >= 4) __builtin_amdgcn_s_setprio(3);   // prio: the vector waves win the arbitration
    f32x4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = (f32x4){s, s, s, s};
    f32x2 p[8];
    for (int i = 0; i < 8; ++i) p[i] = (f32x2){threadIdx.x * 0.5f + i, 1.f};
    const f32x2 s2 = {s, s * 0.5f};
    float a = threadIdx.x * 0.25f, b = s;
    bf16x8 ab, bb;
    for (int i = 0; i < 8; ++i) { ab[i] = (__bf16)(threadIdx.x * 0.01f); bb[i] = (__bf16)s; }
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (wave < 4) {
        if (run_m)
            for (int it = 0; it < IT; ++it) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (KIND == 0) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
                    if (KIND == 1) acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[i], 0, 0, 0);
                    if (KIND == 2) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, bb, acc[i], 0, 0, 0);
                }
                asm volatile("" : "+v"(a));
            }
    } else {
        if (run_v)
            for (int it = 0; it < IT; ++it) {
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(p[i]) : "v"(s2));
            }
    }
    asm volatile("s_nop 0" ::: "memory");
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float r = 0.f;
    for (int i = 0; i < 4; ++i) r += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
    for (int i = 0; i < 8; ++i) r += p[i].x + p[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;     // (slot = role: 0-3 matrix, 4-7 vector)
}

template <int KIND> void run(const char *name, int swap, int prio, float *out, unsigned long long *cyc)
{
    double res[3][2];
    for (int mode = 0; mode < 3; ++mode) {
        const int rm = mode != 1, rv = mode != 0;
        for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<KIND>), dim3(256), dim3(512), 0, 0, out, cyc, rm, rv, 1.0001f, swap, prio);
        unsigned long long h[256 * 8];
        hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
        double m = 0, v = 0;
        for (int i = 0; i < 256; ++i) {
            unsigned long long mm = 0, vv = 0;
            for (int w = 0; w < 4; ++w) { if (h[i * 8 + w] > mm) mm = h[i * 8 + w]; if (h[i * 8 + 4 + w] > vv) vv = h[i * 8 + 4 + w]; }
            m += mm; v += vv;
        }
        res[mode][0] = m / 256; res[mode][1] = v / 256;
    }
    // s_memtime ticks at a constant 100 MHz-class clock: ratios are what matters
    printf("%-26s swap %d prio %d: matrix waves alone %8.0f | vector waves alone %8.0f | together: matrix %8.0f vector %8.0f  -> together / (alone sum) = %.2f, / max = %.2f\n",
           name, swap, prio, res[0][0], res[1][1], res[2][0], res[2][1],
           (res[2][0] > res[2][1] ? res[2][0] : res[2][1]) / (res[0][0] + res[1][1]),
           (res[2][0] > res[2][1] ? res[2][0] : res[2][1]) / (res[0][0] > res[1][1] ? res[0][0] : res[1][1]));
}

int main()
{
    float *out; unsigned long long *cyc;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8 * 8);
    hipMemset(cyc, 0, 256 * 8 * 8);
    printf("%d iterations: 4 matrix instructions per iteration (waves 0-3) vs 8 v_pk_fma_f32 per iteration (waves 4-7), one pair per SIMD\n", IT);
    for (int cfg = 0; cfg < 4; ++cfg) {
        const int swap = cfg & 1, prio = cfg >> 1;
        run<0>("v_mfma_f32_16x16x4_f32", swap, prio, out, cyc);
        run<1>("v_mfma_f32_4x4x1_16B_f32", swap, prio, out, cyc);
        run<2>("v_mfma_f32_16x16x32_bf16", swap, prio, out, cyc);
    }
    return 0;
}

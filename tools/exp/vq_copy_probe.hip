// Memory-pattern probe for the VQ kernel: copy z -> out with exactly the kernel's access shape
// (a wave = 64 consecutive positions x 16 rows of one sample; lane (h, c) moves z[4 s + h][4 c .. 4 c + 3], 16 B),
// persistent workgroups, optional prefetch depth, optional int64 index store.  hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int PF, bool IDX>
__global__ __launch_bounds__(256) void probe(const float *__restrict__ z, float *__restrict__ out, long long *__restrict__ idx,
                                             unsigned NC, int HW)
{
    constexpr int S = 4, D = 16;
    const int lane = threadIdx.x & 63, h = lane >> 4, c = lane & 15;
    const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned cps = HW >> 6, qstep = 4 * gridDim.x;
    auto off = [&](unsigned chunk) {
        chunk = chunk < NC ? chunk : NC - 1;
        const unsigned b = chunk / cps, cw = chunk - b * cps;
        return ((long long)b * D + h) * HW + cw * 64 + 4 * c;
    };
    unsigned chunk = blockIdx.x * 4 + wave;
    f32x4 buf[PF + 1][S];
#pragma unroll
    for (int p = 0; p < PF; ++p) {
        const long long o = off(chunk + p * qstep);
#pragma unroll
        for (int s = 0; s < S; ++s) buf[p][s] = *reinterpret_cast<const f32x4 *>(z + o + (long long)(4 * s) * HW);
    }
    for (; chunk < NC; chunk += qstep) {
        {
            const long long o = off(chunk + PF * qstep);
#pragma unroll
            for (int s = 0; s < S; ++s) buf[PF][s] = *reinterpret_cast<const f32x4 *>(z + o + (long long)(4 * s) * HW);
        }
        const long long o = off(chunk);
#pragma unroll
        for (int s = 0; s < S; ++s) *reinterpret_cast<f32x4 *>(out + o + (long long)(4 * s) * HW) = buf[0][s] * 2.f;
        if (IDX && h < 2) {
            long long pair[2] = {lane, chunk};
            *reinterpret_cast<f32x4 *>(idx + (long long)chunk * 64 + 4 * c + 2 * h) = *reinterpret_cast<const f32x4 *>(pair);
        }
#pragma unroll
        for (int p = 0; p < PF; ++p)
#pragma unroll
            for (int s = 0; s < S; ++s) buf[p][s] = buf[p + 1][s];
    }
}

template <int PF, bool IDX>
void run(const char *name, const float *z, float *out, long long *idx, int B, int wgs)
{
    const int HW = 256;
    const unsigned NC = (unsigned)((long long)B * HW / 64);
    unsigned grid = NC / 4 < 256u * wgs ? NC / 4 : 256u * wgs;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((probe<PF, IDX>), dim3(grid), dim3(256), 0, 0, z, out, idx, NC, HW);
    CK(hipEventRecord(e0));
    const int it = 20;
    for (int i = 0; i < it; ++i) hipLaunchKernelGGL((probe<PF, IDX>), dim3(grid), dim3(256), 0, 0, z, out, idx, NC, HW);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= it;
    const double bytes = (double)B * HW * (128 + (IDX ? 8 : 0));
    printf("%-22s B=%5d wgs=%d grid=%5u: %7.1f us  %7.1f GB/s\n", name, B, wgs, grid, ms * 1e3, bytes / ms / 1e6);
}

int main()
{
    const int Bmax = 8192, HW = 256;
    float *z, *out; long long *idx;
    CK(hipMalloc(&z, (size_t)Bmax * 16 * HW * 4)); CK(hipMalloc(&out, (size_t)Bmax * 16 * HW * 4)); CK(hipMalloc(&idx, (size_t)Bmax * HW * 8));
    CK(hipMemset(z, 0x3c, (size_t)Bmax * 16 * HW * 4));
    for (int B : {2048, 8192})
        for (int wgs : {2, 4, 8}) {
            run<0, false>("pf0", z, out, idx, B, wgs);
            run<1, false>("pf1", z, out, idx, B, wgs);
            run<2, false>("pf2", z, out, idx, B, wgs);
            run<1, true>("pf1+idx", z, out, idx, B, wgs);
        }
    for (int B : {2048, 8192}) run<0, true>("one chunk per wave", z, out, idx, B, 1 << 20);
    return 0;
}

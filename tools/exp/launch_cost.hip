// Experiment (not product): cost of launching many short workgroups vs. their LDS footprint.
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int LDSF> __global__ __launch_bounds__(256) void k_empty(float *out, int n)
{
    __shared__ float s[LDSF];
    s[threadIdx.x] = (float)n;
    __syncthreads();
    if (n < 0) out[blockIdx.x] = s[(threadIdx.x + 1) & 255];
}
template <int LDSF> float run(int grid, float *d)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k_empty<LDSF>, dim3(grid), dim3(256), 0, 0, d, 1);
    hipEventRecord(a);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k_empty<LDSF>, dim3(grid), dim3(256), 0, 0, d, 1);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms / 20 * 1000;
}
int main()
{
    float *d; hipMalloc(&d, 1 << 20);
    int grids[] = {1024, 4096, 16384, 65536};
    for (int g : grids)
        printf("grid %6d: lds 1KB %7.1f us | 8KB %7.1f | 30KB %7.1f | 46KB %7.1f | 60KB %7.1f\n", g,
               run<256>(g, d), run<2048>(g, d), run<7680>(g, d), run<11776>(g, d), run<15360>(g, d));
    return 0;
}

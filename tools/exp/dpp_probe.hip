// Experiment (not product): semantics of wave-level DPP shifts on gfx950.
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int CTRL> __device__ int dpp(int v) { return __builtin_amdgcn_update_dpp(-1, v, CTRL, 0xF, 0xF, true); }
__global__ void k(int *out)
{
    const int l = threadIdx.x;
    out[l] = dpp<0x138>(l + 100);        // wave_shr:1
    out[64 + l] = dpp<0x130>(l + 100);   // wave_shl:1
    out[128 + l] = dpp<0x111>(l + 100);  // row_shr:1
    out[192 + l] = dpp<0x101>(l + 100);  // row_shl:1
}
int main()
{
    int *d, h[256];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char *names[] = {"wave_shr:1", "wave_shl:1", "row_shr:1", "row_shl:1"};
    for (int t = 0; t < 4; ++t) {
        printf("%s:", names[t]);
        for (int l : {0, 1, 2, 15, 16, 17, 31, 32, 33, 62, 63}) printf(" [%d]=%d", l, h[t * 64 + l]);
        printf("\n");
    }
    return 0;
}

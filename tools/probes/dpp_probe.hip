// Probe: lane movement of the DPP wave shifts on gfx950 (which lane each lane reads).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int *out)
{
    int v = threadIdx.x + 100;
    int a = __builtin_amdgcn_update_dpp(-1, v, 0x130, 0xF, 0xF, false); // wave_shl:1
    int b = __builtin_amdgcn_update_dpp(-1, v, 0x138, 0xF, 0xF, false); // wave_shr:1
    int c = __builtin_amdgcn_update_dpp(-1, v, 0x101, 0xF, 0xF, false); // row_shl:1
    int d = __builtin_amdgcn_update_dpp(-1, v, 0x111, 0xF, 0xF, false); // row_shr:1
    out[threadIdx.x] = a;
    out[64 + threadIdx.x] = b;
    out[128 + threadIdx.x] = c;
    out[192 + threadIdx.x] = d;
}
int main()
{
    int *d, h[256];
    hipMalloc(&d, sizeof h);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    const char *names[4] = {"wave_shl:1", "wave_shr:1", "row_shl:1", "row_shr:1"};
    for (int r = 0; r < 4; r++) {
        printf("%s:", names[r]);
        for (int i = 0; i < 64; i++) printf(" %d", h[r * 64 + i] < 0 ? -1 : h[r * 64 + i] - 100);
        printf("\n");
    }
    return 0;
}

// Does a process that made one cooperative launch exit cleanly under rocprofv3 --kernel-trace?
// build: hipcc --offload-arch=gfx950 -O2 tools/probes/coop_probe.hip -o tools/probes/coop_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(int *p) { if (threadIdx.x == 0) atomicAdd(p, 1); }
int main(int argc, char **argv)
{
    int *d = nullptr;
    if (hipMalloc((void **)&d, 4) != hipSuccess) return 1;
    (void)hipMemset(d, 0, 4);
    void *args[] = {&d};
    hipError_t rc;
    if (argc > 1) {
        hipLaunchKernelGGL(k, dim3(8), dim3(64), 0, 0, d);
        rc = hipGetLastError();
    } else {
        rc = hipLaunchCooperativeKernel((const void *)k, dim3(8), dim3(64), args, 0, 0);
    }
    int h = -1;
    (void)hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
    printf("%s launch rc=%d blocks=%d\n", argc > 1 ? "plain" : "cooperative", (int)rc, h);
    (void)hipFree(d);
    return 0;
}

// Rate probe for the f32-input matrix instruction of the network kernel on gfx950:
// cycles per v_mfma_f32_16x16x4_f32 per SIMD with W wavefronts per SIMD and 7
// independent accumulators, (a) operands in registers, (b) every A operand read from
// LDS by a ds_read_b32 13 instructions ahead (the network kernel's pattern), with a
// conflict-free and with the kernel's plane-strided address pattern.
// Build: hipcc --offload-arch=gfx950 -O3 mfma_probe.hip -o mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int NT = 7, KS = 36, REPS = 64;

template <int MODE> // 0: registers, 1: LDS conflict-free, 2: LDS, lanes of the four channel quarters 289 floats apart
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_probe(float *out, unsigned long long *cyc, int reps)
{
    extern __shared__ float lds[];
    const int lane = threadIdx.x;
    for (int i = lane; i < 4700; i += 64) lds[i] = (float)(i & 7);
    __syncthreads();
    f32x4 acc[NT];
    float a[NT], b[KS];
    int rd[NT];
    for (int t = 0; t < NT; t++) {
        acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        rd[t] = MODE == 2 ? (lane >> 4) * 289 + t * 16 + (lane & 15) : t * 64 + lane;
        a[t] = lds[rd[t]];
    }
    for (int k = 0; k < KS; k++) b[k] = 1.0f + lane * 0.001f + k;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < reps; r++) {
#pragma unroll
        for (int ks = 0; ks < KS; ks++) {
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < NT; t++) {
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], b[ks], acc[t], 0, 0, 0);
                if (MODE != 0) a[t] = lds[rd[t] + (ks & 3) * 1156 + (ks >> 2)];
            }
            if (MODE != 0) {
#pragma unroll
                for (int t = 0; t < NT; t++) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int t = 0; t < NT; t++) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
    out[blockIdx.x * 64 + lane] = s;
    if (lane == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE> void run(const char *name, int waves_per_simd)
{
    const int grid = 256 * 4 * waves_per_simd;
    float *out;
    unsigned long long *cyc;
    hipMalloc(&out, grid * 64 * sizeof(float));
    hipMalloc(&cyc, grid * sizeof(unsigned long long));
    hipFuncSetAttribute((const void *)k_probe<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 19000);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL(k_probe<MODE>, dim3(grid), dim3(64), 19000, 0, out, cyc, REPS);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_probe<MODE>, dim3(grid), dim3(64), 19000, 0, out, cyc, REPS);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(grid);
    hipMemcpy(h.data(), cyc, grid * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double mean = 0;
    for (auto v : h) mean += (double)v;
    mean /= grid;
    const double mfma_per_wave = (double)REPS * KS * NT;
    const double tflops = (double)grid * mfma_per_wave * 2048.0 / (ms * 1e-3) / 1e12;
    printf("%-34s waves/SIMD %d: %7.3f ms  %6.1f TFLOP/s  in-kernel ticks per MFMA per wave %6.1f (x %d waves = per SIMD)\n", name,
           waves_per_simd, ms, tflops, mean / mfma_per_wave, waves_per_simd);
    hipFree(out);
    hipFree(cyc);
}

int main()
{
    for (int w = 1; w <= 2; w++) {
        run<0>("registers", w);
        run<1>("LDS, conflict-free", w);
        run<2>("LDS, network pattern (2-way)", w);
    }
    return 0;
}

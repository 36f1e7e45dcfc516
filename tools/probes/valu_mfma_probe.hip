// Do vector-ALU instructions overlap with v_mfma_f32_16x16x4_f32 on gfx950?  16 independent
// accumulators; after each matrix instruction NV independent f32 additions (plain or packed)
// of the same wavefront, issued after every matrix instruction or after groups of G, W wavefronts per SIMD.  Prints cycles per matrix instruction per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 valu_mfma_probe.hip -o valu_mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int NA = 16, REPS = 256;

template <int NV, bool PK, int G, bool AG = false>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_probe(float *out, unsigned long long *cyc, int reps)
{
    const int lane = threadIdx.x;
    f32x4 acc[NA];
    float a[NA], b[NA];
    f32x2 v[8];
    for (int t = 0; t < NA; t++) {
        acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        a[t] = lane * 0.5f + t;
        b[t] = 1.0f + lane * 0.001f + t;
    }
    for (int i = 0; i < 8; i++) v[i] = (f32x2){(float)lane + i, (float)i};
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < reps; r++) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < NA; t++) {
            if constexpr (AG) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[t]) : "v"(a[t]), "v"(b[t]));
            else acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], b[t], acc[t], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < NV; i++) {
                if (PK) v[i & 7] = v[i & 7] + v[(i + 1) & 7];
                else v[i & 7].x = v[i & 7].x + v[(i + 3) & 7].y;
            }
        }
#pragma unroll
        for (int t = 0; t < NA / G; t++) {
            __builtin_amdgcn_sched_group_barrier(0x008, G, 0);
            if (NV > 0) __builtin_amdgcn_sched_group_barrier(0x002, NV * G, 0);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int t = 0; t < NA; t++) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
    for (int i = 0; i < 8; i++) s += v[i].x + v[i].y;
    out[blockIdx.x * 64 + lane] = s;
    if (lane == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NV, bool PK, int G = 1, bool AG = false> void run(int waves_per_simd)
{
    const int grid = 256 * 4 * waves_per_simd;
    float *out;
    unsigned long long *cyc;
    hipMalloc(&out, grid * 64 * sizeof(float));
    hipMalloc(&cyc, grid * sizeof(unsigned long long));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL((k_probe<NV, PK, G, AG>), dim3(grid), dim3(64), 0, 0, out, cyc, REPS);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k_probe<NV, PK, G, AG>), dim3(grid), dim3(64), 0, 0, out, cyc, REPS);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double mfma_per_wave = (double)REPS * NA;
    const double tflops = (double)grid * mfma_per_wave * 2048.0 / (ms * 1e-3) / 1e12;
    // SIMD time per matrix instruction in ns: ms / (mfma per SIMD)
    printf("%s%d %s adds per matrix instruction in groups of %d, %d waves/SIMD: %7.3f ms  %6.1f TFLOP/s  %6.2f ns per matrix instruction per SIMD\n", AG ? "[acc in AGPRs] " : "", NV,
           PK ? "packed" : "plain", G, waves_per_simd, ms, tflops, ms * 1e6 / (mfma_per_wave * waves_per_simd));
    hipFree(out);
    hipFree(cyc);
}

int main()
{
    for (int w = 1; w <= 2; w++) {
        run<0, false>(w);
        run<2, false>(w);
        run<4, false>(w);
        run<6, false>(w);
        run<8, false>(w);
        run<2, true>(w);
        run<4, true>(w);
        run<6, true>(w);
        run<2, false, 4>(w);
        run<4, false, 4>(w);
        run<4, false, 8>(w);
        run<4, false, 16>(w);
        run<2, false, 16>(w);
        run<2, true, 16>(w);
        run<0, false, 1, true>(w);
        run<4, false, 1, true>(w);
        run<4, false, 16, true>(w);
    }
    return 0;
}

// Latency probe for one wavefront per SIMD on gfx950: cycles from issue to
// completion (s_waitcnt vmcnt(0)) of a cold 16-byte-per-lane global load, of a
// store to a cold line, and of a load that has stores issued behind it.
// Build: hipcc --offload-arch=gfx950 -O2 mem_probe.hip -o mem_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ size_t next_off(unsigned long long &x, size_t lines)
{
    x ^= x >> 12; x ^= x << 25; x ^= x >> 27;
    return (size_t)((x * 0x2545F4914F6CDD1Dull) >> 20 & (lines - 1)) * 1024; // 1 KiB granules, `lines` a power of two
}

// mode 0: cold load only
//      1: two full-line stores (16 B per lane, contiguous) to a cold granule
//      2: four single-lane 4-byte stores to cold lines
//      3: load, then the stores of 1 and 2, wait for everything
//      4: load, then the same stores, wait for the load only (vmcnt(6))
//      5: as 3, time until the stores are ISSUED only (no wait)
__global__ void k_probe(char *buf, size_t granules, int iters, int mode, unsigned long long *out, int *sink)
{
    unsigned long long x = 88172645463325252ull + blockIdx.x * 7919ull;
    unsigned long long total = 0;
    int acc = 0;
    const int lane = threadIdx.x;
    for (int it = 0; it < iters; it++) {
        const size_t o = next_off(x, granules), o2 = next_off(x, granules), o3 = next_off(x, granules);
        int4 *p = reinterpret_cast<int4 *>(buf + o) + lane;
        int4 *q = reinterpret_cast<int4 *>(buf + o2) + lane;
        int *r = reinterpret_cast<int *>(buf + o3);
        asm volatile("" : "+v"(p), "+v"(q), "+s"(r)::"memory"); // addresses ready before the clock starts
        __builtin_amdgcn_s_waitcnt(0x0F70);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const unsigned long long t0 = __builtin_readcyclecounter();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        int4 v = make_int4(0, 0, 0, 0);
        if (mode == 0 || mode >= 3) v = *p;
        if (mode == 1 || mode >= 3) {
            q[0] = make_int4(it, lane, 1, 2);
            q[64] = make_int4(it, lane, 3, 4);
        }
        if (mode == 2 || mode >= 3) {
            if (lane == 0) { r[0] = it; r[300] = it; r[600] = it; r[900] = it; }
        }
        if (mode == 4) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else if (mode != 5) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long t1 = __builtin_readcyclecounter();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        total += t1 - t0;
        __builtin_amdgcn_s_waitcnt(0x0F70);
        acc += v.x;
    }
    if (lane == 0) out[blockIdx.x] = total / iters;
    if (acc == 0x7fffffff) sink[0] = acc;
}

int main()
{
    const size_t bytes = (16ull << 30) + (1 << 20); // 2^24 granules of 1 KiB: larger than the last-level cache by far
    char *buf;
    unsigned long long *out;
    int *sink;
    if (hipMalloc(&buf, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(buf, 0, bytes);
    hipMalloc(&out, 1024 * 8);
    hipMalloc(&sink, 4);
    const char *names[] = {"cold load", "2 full-line stores", "4 single-lane stores", "load + 6 stores, wait all",
                           "load + 6 stores, wait load only", "load + 6 stores, issue only"};
    for (int mode = 0; mode < 6; mode++) {
        hipLaunchKernelGGL(k_probe, dim3(1024), dim3(64), 0, 0, buf, (size_t)1 << 24, 200, mode, out, sink);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(1024);
        hipMemcpy(h.data(), out, 1024 * 8, hipMemcpyDeviceToHost);
        double s = 0;
        for (auto v : h) s += v;
        printf("%-34s %8.0f cycles\n", names[mode], s / 1024);
        fflush(stdout);
    }
    return 0;
}

// Instruction-cost probe for one wavefront per SIMD on gfx950: cycles per
// instruction for dependent / independent chains of the instruction kinds the
// search kernel is made of.  Build: hipcc --offload-arch=gfx950 -O1 issue_probe.hip -o issue_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)

#define PROBE(name, body)                                                                                  \
    __global__ void name(unsigned long long *out, double *sink)                                            \
    {                                                                                                      \
        double a = threadIdx.x * 1.0 + 1.0, b = 1.000001, c = 0.5, d = 2.0, e = 3.0;                       \
        float fa = threadIdx.x, fb = 1.0001f, fc = 0.5f, fd = 1.5f;                                        \
        int ia = threadIdx.x, ib = 3, ic = 5, id = 7, sa = 0;                                              \
        unsigned long long t0 = __builtin_readcyclecounter();                                              \
        for (int i = 0; i < 16; i++) { body }                                                              \
        unsigned long long t1 = __builtin_readcyclecounter();                                              \
        if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;                                                   \
        sink[threadIdx.x] = a + b + c + d + e + fa + fb + fc + fd + ia + ib + ic + id + sa;                \
    }

PROBE(k_empty, asm volatile("" ::: "memory");)
PROBE(k_f32_dep, REP64(asm volatile("v_add_f32 %0, %0, %1" : "+v"(fa) : "v"(fb));))
PROBE(k_f32_ind, REP16(asm volatile("v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4" : "+v"(fa), "+v"(fb), "+v"(fc), "+v"(fd) : "v"(1.0f));))
PROBE(k_f64_dep, REP64(asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));))
PROBE(k_f64_ind, REP16(asm volatile("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5" : "+v"(a), "+v"(c), "+v"(d), "+v"(e) : "v"(b), "v"(0.25));))
PROBE(k_i32_dep, REP64(asm volatile("v_add_u32 %0, %0, %1" : "+v"(ia) : "v"(ib));))
PROBE(k_cndmask_dep, REP64(asm volatile("v_cmp_lt_i32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(ia) : "v"(ib) : "vcc");))
PROBE(k_salu_dep, REP64(asm volatile("s_add_u32 %0, %0, 1" : "+s"(sa) :: "scc");))
PROBE(k_readlane_use, REP64(asm volatile("v_readlane_b32 %1, %0, 3\n s_nop 3\n v_add_u32 %0, %0, %1" : "+v"(ia), "+s"(sa));))
PROBE(k_rcp64_dep, REP64(asm volatile("v_rcp_f64 %0, %0" : "+v"(a));))
PROBE(k_dpp_dep, REP64(asm volatile("s_nop 1\n v_max_i32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(ia));))
PROBE(k_branch_taken, REP64(asm volatile("s_cmp_eq_u32 %0, 12345\n s_cbranch_scc1 1f\n v_add_u32 %1, %1, 1\n s_branch 2f\n 1: v_add_u32 %1, %1, 2\n 2:" : "+s"(sa), "+v"(ia) :: "scc");))
PROBE(k_branch_nottaken, REP64(asm volatile("s_cmp_eq_u32 %0, 12345\n s_cbranch_scc1 1f\n v_add_u32 %1, %1, 1\n 1:" : "+s"(sa), "+v"(ia) :: "scc");))
PROBE(k_execz_skip, REP64(asm volatile("v_cmp_eq_u32 vcc, 12345, %0\n s_and_saveexec_b64 s[20:21], vcc\n s_cbranch_execz 1f\n v_add_u32 %0, %0, 1\n 1: s_or_b64 exec, exec, s[20:21]" : "+v"(ia) :: "vcc", "scc", "s20", "s21");))
// taken branches whose target lies further ahead in the code (skipped filler = s_nop words)
PROBE(k_far64, REP64(asm volatile("s_branch 1f\n .fill 16, 4, 0xbf800000\n 1: v_add_u32 %0, %0, 1" : "+v"(ia));))
PROBE(k_far256, REP64(asm volatile("s_branch 1f\n .fill 64, 4, 0xbf800000\n 1: v_add_u32 %0, %0, 1" : "+v"(ia));))
PROBE(k_far1k, REP16(asm volatile("s_branch 1f\n .fill 256, 4, 0xbf800000\n 1: v_add_u32 %0, %0, 1" : "+v"(ia));))
PROBE(k_far4k, REP4(asm volatile("s_branch 1f\n .fill 1024, 4, 0xbf800000\n 1: v_add_u32 %0, %0, 1" : "+v"(ia));))
// backward and forward jumps between two places 2 KB apart (a loop with an out-of-line block)
PROBE(k_pingpong, REP16(asm volatile("s_branch 2f\n 1: v_add_u32 %0, %0, 1\n s_branch 3f\n .fill 512, 4, 0xbf800000\n 2: v_add_u32 %0, %0, 2\n s_branch 1b\n 3:" : "+v"(ia));))
PROBE(k_mixed, REP64(asm volatile("v_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 1" : "+v"(ia), "+s"(sa) :: "scc");))
__global__ void k_lds(unsigned long long *out, double *sink)
{
    __shared__ double buf[64];
    buf[threadIdx.x] = threadIdx.x;
    __syncthreads();
    int idx = threadIdx.x;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < 256; i++) idx = (int)buf[idx & 63]; // dependent LDS reads
    unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    sink[threadIdx.x] = idx;
}

template <class K> static void run(const char *name, K k, int n_inst, int waves_per_cu)
{
    unsigned long long *out;
    double *sink;
    const int blocks = 256 * waves_per_cu;
    hipMalloc(&out, blocks * 8);
    hipMalloc(&sink, 64 * 8);
    printf("%s ...\n", name);
    fflush(stdout);
    for (int it = 0; it < 2; it++) hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, out, sink);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks);
    hipMemcpy(h.data(), out, blocks * 8, hipMemcpyDeviceToHost);
    double s = 0;
    for (auto v : h) s += v;
    printf("%-20s waves/CU %2d  %8.2f cycles/inst  (total %.0f)\n", name, waves_per_cu, s / blocks / n_inst, s / blocks);
    fflush(stdout);
    hipFree(out);
    hipFree(sink);
}

int main()
{
    for (int w : {4}) {
        run("empty(16 iters)", k_empty, 16, w);
        run("v_add_f32 dep", k_f32_dep, 16 * 64, w);
        run("v_add_f32 indep", k_f32_ind, 16 * 64, w);
        run("v_fma_f64 dep", k_f64_dep, 16 * 64, w);
        run("v_fma_f64 indep", k_f64_ind, 16 * 64, w);
        run("v_add_u32 dep", k_i32_dep, 16 * 64, w);
        run("cmp+cndmask dep", k_cndmask_dep, 16 * 64 * 2, w);
        run("s_add_u32 dep", k_salu_dep, 16 * 64, w);
        run("readlane+nop+use", k_readlane_use, 16 * 64 * 3, w);
        run("v_rcp_f64 dep", k_rcp64_dep, 16 * 64, w);
        run("nop+max_dpp dep", k_dpp_dep, 16 * 64 * 2, w);
        run("branch taken x2", k_branch_taken, 16 * 64, w);
        run("branch not taken", k_branch_nottaken, 16 * 64, w);
        run("execz skip block", k_execz_skip, 16 * 64, w);
        run("taken, skip 64 B", k_far64, 16 * 64 * 2, w);
        run("taken, skip 256 B", k_far256, 16 * 64 * 2, w);
        run("taken, skip 1 KB", k_far1k, 16 * 16 * 2, w);
        run("taken, skip 4 KB", k_far4k, 16 * 4 * 2, w);
        run("out-of-line 2 KB x3", k_pingpong, 16 * 16 * 5, w);
        run("valu+salu mixed", k_mixed, 16 * 64 * 2, w);
        run("ds_read dep (+cvt)", k_lds, 256, w);
    }
    return 0;
}

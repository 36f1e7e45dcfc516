// az_net_parts.h -- device functions the batched network kernels of az_net.hip and az_net_w43.hip share: the list
// of positions a workgroup evaluates, compile-time loops, wave reductions, the policy head's partial planes.
#ifndef AZ_NET_PARTS_H
#define AZ_NET_PARTS_H

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <utility>

#include "../../include/azmcts.h"
#include "az_common.h"
#include "az_net_team.h"

namespace aznet {

// Packed parameter block, continued (after the direct-form blocks of az_net_team.h): the 16 -> 16 layers in the
// Winograd F(2x2,3x3) domain, then the input layer in that domain, then both again in the F(4x4,3x3) domain.
constexpr int WINO_FLOATS = 64 * 64 + CH; // U[16 positions][4 K steps][64 lanes], shift[16]
constexpr int WINO_FIRST_FLOATS = 16 * 64 + CH; // input layer: U[16 positions][64 lanes] (zero for the lanes of K indices 2..3), shift[16]
constexpr int W43_FLOATS = 4 * 36 * 64 + CH; // U[4 K steps][18 pairs of positions][64 lanes][2], shift[16]
constexpr int W43_FIRST_FLOATS = 36 * 64 + CH; // input layer: U[18 pairs of positions][64 lanes][2] (zero for the lanes of K indices 2..3), shift[16]

// Which positions a workgroup evaluates: those whose code is AZ_LEAF_EXPAND (a search that ended
// on a finished game, or an idle tree, needs no evaluation), split evenly over the workgroups in
// contiguous runs, so the workgroups stay balanced however the skipped ones are spread.  Every
// workgroup derives its run from the codes by itself (no atomics, no second kernel): each thread
// counts a slice of the codes (16-byte loads), a prefix sum places the slices, and only the few
// threads whose slice meets the workgroup's run walk their slice a second time.
// cnt: LDS scratch of NW ints (unused for a single wavefront).  Returns the length of the run,
// its positions are in todo[].  Ends with a barrier.
template <int NTHR> __device__ __forceinline__ int list_positions(const int *leaf_kind, int n, int *todo, int *cnt, int tid)
{
    constexpr int NW = NTHR / 64;
    const int lane = tid & 63, wave = tid >> 6;
    const int per = (((n + NTHR - 1) / NTHR) + 3) & ~3; // a multiple of four codes per thread
    const int lo = tid * per, hi = lo + per < n ? lo + per : n;
    const bool aligned = (reinterpret_cast<uintptr_t>(leaf_kind) & 15) == 0;
    int c = 0;
    if (aligned) {
        int i = lo;
#pragma unroll 8
        for (; i + 3 < hi; i += 4) {
            const int4 v = *reinterpret_cast<const int4 *>(leaf_kind + i);
            c += (v.x == AZ_LEAF_EXPAND) + (v.y == AZ_LEAF_EXPAND) + (v.z == AZ_LEAF_EXPAND) + (v.w == AZ_LEAF_EXPAND);
        }
        for (; i < hi; i++) c += leaf_kind[i] == AZ_LEAF_EXPAND;
    } else {
        for (int i = lo; i < hi; i++) c += leaf_kind[i] == AZ_LEAF_EXPAND;
    }
    int incl = c; // inclusive prefix over the wavefront, then over the wavefronts
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int up = __shfl_up(incl, o);
        if (lane >= o) incl += up;
    }
    int before = incl - c, total;
    if constexpr (NW > 1) {
        if (lane == 63) cnt[wave] = incl;
        __syncthreads();
        total = 0;
#pragma unroll
        for (int w = 0; w < NW; w++) {
            if (w < wave) before += cnt[w];
            total += cnt[w];
        }
    } else {
        total = __shfl(incl, 63);
    }
    const int first = (int)((long long)blockIdx.x * total / (int)gridDim.x);
    const int last = (int)((long long)(blockIdx.x + 1) * total / (int)gridDim.x); // (exclusive)
    if (c > 0 && before < last && before + c > first) {
        int i = lo;
        if (aligned) {
            // 32 codes per round, their loads issued together: a walk code by code would pay the full memory
            // latency for every code (the loop's exit depends on the codes before)
            for (; i + 31 < hi && before < last; i += 32) {
                int4 v[8];
#pragma unroll
                for (int q = 0; q < 8; q++) v[q] = *reinterpret_cast<const int4 *>(leaf_kind + i + 4 * q);
#pragma unroll
                for (int q = 0; q < 8; q++) {
                    const int e[4] = {v[q].x, v[q].y, v[q].z, v[q].w};
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const bool ex = e[r] == AZ_LEAF_EXPAND;
                        if (ex && before >= first && before < last) todo[before - first] = i + 4 * q + r;
                        before += ex ? 1 : 0;
                    }
                }
            }
        }
        for (; i < hi && before < last; i++)
            if (leaf_kind[i] == AZ_LEAF_EXPAND) {
                if (before >= first) todo[before - first] = i;
                before++;
            }
    }
    __syncthreads();
    return last - first;
}

// The same run, found from a bit mask (bit i set = position i needs an evaluation; bits past n are clear; word w
// at index w * AZ_MASK_STRIDE)
// instead of the codes: one wavefront reads the whole mask in one or a few loads per lane, so the list
// costs a workgroup about a microsecond where scanning the codes costs ten (2048 workgroups reading 32 KB
// of codes each is 64 MB of L2 traffic per launch).  For a single wavefront (NTHR = 64).
__device__ __forceinline__ int list_positions_mask(const unsigned *mask, int n, int *todo, int lane)
{
    const int words = (n + 31) >> 5;
    const int per = (words + 63) >> 6; // words per lane, consecutive
    const int lo = lane * per, hi = lo + per < words ? lo + per : words;
    int c = 0;
    for (int w = lo; w < hi; w++) c += __popc(mask[w * AZ_MASK_STRIDE]);
    int incl = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int up = __shfl_up(incl, o);
        if (lane >= o) incl += up;
    }
    int before = incl - c;
    const int total = __shfl(incl, 63);
    const int first = (int)((long long)blockIdx.x * total / (int)gridDim.x);
    const int last = (int)((long long)(blockIdx.x + 1) * total / (int)gridDim.x); // (exclusive)
    if (c > 0 && before < last && before + c > first) {
        for (int w = lo; w < hi && before < last; w++) {
            unsigned m = mask[w * AZ_MASK_STRIDE];
            const int pc = __popc(m);
            if (before + pc <= first) { // the whole word lies before the run
                before += pc;
                continue;
            }
            while (m != 0u && before < last) {
                const int b = __ffs(m) - 1;
                m &= m - 1u;
                if (before >= first) todo[before - first] = w * 32 + b;
                before++;
            }
        }
    }
    __syncthreads();
    return last - first;
}

// f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>): a loop whose index is a constant expression
// in the body (the scheduling builtins only take literals)
template <class F, int... I> __device__ __forceinline__ void static_for_impl(F &&f, std::integer_sequence<int, I...>)
{
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F> __device__ __forceinline__ void static_for(F &&f)
{
    static_for_impl(f, std::make_integer_sequence<int, N>{});
}

// Sum / maximum over the 64 lanes by DPP (quad swaps, mirrors inside rows of 16, then the row
// totals broadcast upwards); the result is read from lane 63 and is wave-uniform.
template <bool MAX> __device__ __forceinline__ float wave_reduce(float v)
{
    auto step = [&](auto ctrl, auto row_mask) {
        // lanes outside row_mask receive the identity: 0 for a sum, their own value for a maximum
        const int old = MAX ? __builtin_bit_cast(int, v) : 0;
        const float o = __builtin_bit_cast(
            float, __builtin_amdgcn_update_dpp(old, __builtin_bit_cast(int, v), decltype(ctrl)::value, decltype(row_mask)::value, 0xf, false));
        v = MAX ? fmaxf(v, o) : v + o;
    };
    step(std::integral_constant<int, 0xB1>{}, std::integral_constant<int, 0xf>{});  // quad_perm:[1,0,3,2]
    step(std::integral_constant<int, 0x4E>{}, std::integral_constant<int, 0xf>{});  // quad_perm:[2,3,0,1]
    step(std::integral_constant<int, 0x141>{}, std::integral_constant<int, 0xf>{}); // row_half_mirror
    step(std::integral_constant<int, 0x140>{}, std::integral_constant<int, 0xf>{}); // row_mirror
    step(std::integral_constant<int, 0x142>{}, std::integral_constant<int, 0xa>{}); // row_bcast:15 into rows 1 and 3
    step(std::integral_constant<int, 0x143>{}, std::integral_constant<int, 0xc>{}); // row_bcast:31 into rows 2 and 3
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// Geometry of the one-wavefront-per-SIMD Winograd F(2x2,3x3) kernel (k_net_wide in az_net.hip): the board's own tile
// grid (ceil(S/2) tiles per tile row), row-major, blocks of 16 tiles; one in-place LDS buffer per position.
template <int S> struct WideGeo {
    using G = Geo<S>;
    static constexpr int TW = (S + 1) / 2, TILES = TW * TW, NB = (TILES + 15) / 16;
    static constexpr unsigned DIV = 65536 / TW + 1; // T / TW == (T * DIV) >> 16 for T < 128
    static constexpr int LDS_BYTES = (G::BUF + 64) * 4; // + the zero tail and the dump cells
    static constexpr int WGS = 4;
    static constexpr int TPR = (S + 15) / 16; // matrix tiles per board row (policy head)
    static_assert(TW + 1 <= 16 && NB * 16 < 128 + 16, "tile grid");
    static_assert(WGS * (LDS_BYTES + 512) <= 160 * 1024, "four positions per CU");
};

// Policy head, first half, for any board: the nine per-tap partial planes written over planes 0..8.
// M = 16 cells of one board row (two tiles per row above 16 cells; lanes past the row end read its right
// halo cell, zero, and what they store is zero and lands on that cell again).  A cell's product reads and
// writes that cell only, so the rows go in groups: all reads of a group precede its writes.
// A team of `team` wavefronts shares the groups: wavefront `member` (wave-uniform) takes every team-th of them.
template <int S> __device__ __forceinline__ void policy_taps_wide(float *buf, const float (&wpb)[4], int lane, int member = 0, int team = 1)
{
    using G = Geo<S>;
    constexpr int TPR = WideGeo<S>::TPR, RG = 5, NG = (S + RG - 1) / RG; // RG board rows per group
    int l = lane;
    asm volatile("" : "+v"(l));
    const int co = l & 15, kq = l >> 4;
#pragma unroll
    for (int g = 0; g < NG; g++) {
        if (team > 1 && g % team != member) continue;
        const int rows = (g + 1) * RG <= S ? RG : S - g * RG;
        f32x4 acc[RG * TPR];
#pragma unroll
        for (int t = 0; t < RG * TPR; t++) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        __builtin_amdgcn_s_setprio(0);
#pragma unroll
        for (int ks = 0; ks < 4; ks++)
#pragma unroll
            for (int t = 0; t < RG * TPR; t++) {
                if (t / TPR >= rows) continue;
                const int cell = co + 16 * (t % TPR);
                const float av = buf[(ks * 4 + kq) * G::PS + (g * RG + t / TPR + 1) * G::ROW + (cell < S ? cell : S) + 1];
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, wpb[ks], acc[t], 0, 0, 0);
            }
        __builtin_amdgcn_s_setprio(3);
        if (co < 9) {
#pragma unroll
            for (int t = 0; t < RG * TPR; t++) {
                if (t / TPR >= rows) continue;
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int cell = 4 * kq + r + 16 * (t % TPR);
                    buf[co * G::PS + (g * RG + t / TPR + 1) * G::ROW + (cell < S ? cell : S) + 1] = acc[t][r];
                }
            }
        }
    }
}

} // namespace aznet

#endif

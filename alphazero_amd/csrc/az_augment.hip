// Trainer-side sample preparation on the GPU (gfx950): the eight board
// symmetries and the averaging of duplicate positions that the reference does
// with numpy on the host (src/tools.py:46-88).  Pure index / byte work plus
// float32 sums whose order is part of the result; HBM-bound.
//
// Sample layout = the reference's record fields as separate dense arrays:
//   board float32 [n][2][S][S], pi float32 [n][S*S], r float32 [n].
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/azmcts.h"

#pragma clang fp contract(off)

namespace {

constexpr int MAX_S = 20;

// source cell (si, sj) of output cell (i, j) for symmetry r in the order
// get_rotations emits them (tools.py:78-86): three quarter turns (np.rot90),
// the row flip of the third, three quarter turns of that.
__device__ __forceinline__ int sym_source(int r, int i, int j, int n)
{
    switch (r) {
    case 0: return j * n + (n - 1 - i);
    case 1: return (n - 1 - i) * n + (n - 1 - j);
    case 2: return (n - 1 - j) * n + i;
    case 3: return (n - 1 - j) * n + (n - 1 - i);
    case 4: return i * n + (n - 1 - j);
    case 5: return j * n + i;
    default: return (n - 1 - i) * n + j;
    }
}

// one workgroup per input record, 7 derived records out (record i*7 + r)
__global__ __launch_bounds__(256) void k_rotations(int S, int n, const float *board, const float *pi, const float *rv,
                                                   float *board_out, float *pi_out, float *r_out)
{
    const int rec = blockIdx.x;
    if (rec >= n) return;
    const int size = S * S;
    __shared__ float s_in[3 * MAX_S * MAX_S]; // planes 0, 1 and pi of this record
    for (int e = threadIdx.x; e < 2 * size; e += blockDim.x) s_in[e] = board[(size_t)rec * 2 * size + e];
    for (int e = threadIdx.x; e < size; e += blockDim.x) s_in[2 * size + e] = pi[(size_t)rec * size + e];
    __syncthreads();
    const float r = rv[rec];
    for (int t = threadIdx.x; t < 7 * size; t += blockDim.x) {
        const int sym = t / size, cell = t - sym * size;
        const int src = sym_source(sym, cell / S, cell % S, S);
        const size_t o = (size_t)rec * 7 + sym;
        board_out[o * 2 * size + cell] = s_in[src];
        board_out[o * 2 * size + size + cell] = s_in[size + src];
        pi_out[o * size + cell] = s_in[2 * size + src];
    }
    if (threadIdx.x < 7) r_out[(size_t)rec * 7 + threadIdx.x] = r;
}

// Sort key of a board: its 2*S*S cells as a bit string, first cell most
// significant, so that unsigned word-by-word order == lexicographic order of the
// flattened 0/1 board (the order np.unique(axis=0) returns, tools.py:49).
// keys[w * n + rec]; cells other than 0 and 1 raise *bad.
__global__ __launch_bounds__(64) void k_pack_keys(int S, int n, const float *board, unsigned long long *keys, int words,
                                                  int *bad)
{
    const int rec = blockIdx.x;
    if (rec >= n) return;
    const int cells = 2 * S * S;
    const int lane = threadIdx.x;
    for (int w = 0; w < words; w++) {
        const int e = w * 64 + lane;
        const float v = e < cells ? board[(size_t)rec * cells + e] : 0.f;
        if (v != 0.f && v != 1.f) *bad = 1;
        const unsigned long long bits = __ballot(v != 0.f); // lane l = bit l
        if (lane == 0) keys[(size_t)w * n + rec] = __brevll(bits); // cell 0 -> bit 63
    }
}

// float32 add.reduce of a contiguous array as numpy computes it (pairwise
// summation: sequential below 8 elements, eight interleaved accumulators up to
// 128, halves rounded down to a multiple of 8 above), over r[members[0..n)].
__device__ float numpy_sum_f32(const float *r, const int *members, int n)
{
    if (n < 8) {
        float res = 0.f;
        for (int i = 0; i < n; i++) res = res + r[members[i]];
        return res;
    }
    if (n <= 128) {
        float a[8];
        for (int k = 0; k < 8; k++) a[k] = r[members[k]];
        int i = 8;
        for (; i < n - (n % 8); i += 8)
            for (int k = 0; k < 8; k++) a[k] = a[k] + r[members[i + k]];
        float res = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
        for (; i < n; i++) res = res + r[members[i]];
        return res;
    }
    int n2 = n / 2;
    n2 -= n2 % 8;
    return numpy_sum_f32(r, members, n2) + numpy_sum_f32(r, members + n2, n - n2);
}

// One workgroup per distinct board.  members = record indices sorted by (key,
// original index); segment u covers members[seg_start[u] .. seg_start[u+1]).
// A board seen once keeps its record; otherwise pi = float32 running sum over
// the occurrences in original order / count, r = numpy's float32 mean
// (tools.py:57-62).
__global__ __launch_bounds__(256) void k_unique_average(int S, int n_unique, const int *members, const int *seg_start,
                                                        const float *board, const float *pi, const float *rv,
                                                        float *board_out, float *pi_out, float *r_out)
{
    const int u = blockIdx.x;
    if (u >= n_unique) return;
    const int size = S * S;
    const int b = seg_start[u], cnt = seg_start[u + 1] - b;
    const int first = members[b];
    for (int e = threadIdx.x; e < 2 * size; e += blockDim.x)
        board_out[(size_t)u * 2 * size + e] = board[(size_t)first * 2 * size + e];
    for (int c = threadIdx.x; c < size; c += blockDim.x) {
        float acc = pi[(size_t)first * size + c];
        if (cnt > 1) {
            for (int m = 1; m < cnt; m++) acc = acc + pi[(size_t)members[b + m] * size + c];
            acc = __fdiv_rn(acc, (float)cnt);
        }
        pi_out[(size_t)u * size + c] = acc;
    }
    if (threadIdx.x == 0)
        r_out[u] = cnt > 1 ? __fdiv_rn(numpy_sum_f32(rv, members + b, cnt), (float)cnt) : rv[first];
}

int launch_ok() { return hipGetLastError() == hipSuccess ? AZ_OK : AZ_ERR_HIP; }

} // namespace

extern "C" {

int az_aug_key_words(int shape) { return shape >= 1 && shape <= MAX_S ? (2 * shape * shape + 63) / 64 : 0; }

int az_aug_rotations(int shape, int n, const float *d_board, const float *d_pi, const float *d_r, float *d_board_out,
                     float *d_pi_out, float *d_r_out, void *stream)
{
    if (shape < 1 || shape > MAX_S || n < 0) return AZ_ERR_ARG;
    if (n == 0) return AZ_OK;
    if (!d_board || !d_pi || !d_r || !d_board_out || !d_pi_out || !d_r_out) return AZ_ERR_ARG;
    hipLaunchKernelGGL(k_rotations, dim3(n), dim3(256), 0, (hipStream_t)stream, shape, n, d_board, d_pi, d_r,
                       d_board_out, d_pi_out, d_r_out);
    return launch_ok();
}

int az_aug_pack_keys(int shape, int n, const float *d_board, unsigned long long *d_keys, int *d_bad, void *stream)
{
    if (shape < 1 || shape > MAX_S || n < 0) return AZ_ERR_ARG;
    if (n == 0) return AZ_OK;
    if (!d_board || !d_keys || !d_bad) return AZ_ERR_ARG;
    hipLaunchKernelGGL(k_pack_keys, dim3(n), dim3(64), 0, (hipStream_t)stream, shape, n, d_board, d_keys,
                       az_aug_key_words(shape), d_bad);
    return launch_ok();
}

int az_aug_unique_average(int shape, int n_unique, const int *d_members, const int *d_seg_start, const float *d_board,
                          const float *d_pi, const float *d_r, float *d_board_out, float *d_pi_out, float *d_r_out,
                          void *stream)
{
    if (shape < 1 || shape > MAX_S || n_unique < 0) return AZ_ERR_ARG;
    if (n_unique == 0) return AZ_OK;
    if (!d_members || !d_seg_start || !d_board || !d_pi || !d_r || !d_board_out || !d_pi_out || !d_r_out)
        return AZ_ERR_ARG;
    hipLaunchKernelGGL(k_unique_average, dim3(n_unique), dim3(256), 0, (hipStream_t)stream, shape, n_unique, d_members,
                       d_seg_start, d_board, d_pi, d_r, d_board_out, d_pi_out, d_r_out);
    return launch_ok();
}

} // extern "C"

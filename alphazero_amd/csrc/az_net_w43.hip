// Batched policy/value network evaluation (the reference's `simplerNN`, src/model.py:9-44, inference mode),
// Winograd F(4x4, 3x3) form of the 3x3 layers for boards of 15x15, 16x16 and (further down) 20x20: see az_net.hip for the other forms
// and for what is common to all of them (activations of a position resident in LDS, plane-major with a zero halo,
// every layer in place, f32 matrix instructions v_mfma_f32_16x16x4_f32).
//
// A 4x4 tile of outputs comes from the 6x6 input patch around it through 36 element-wise products in a
// transformed domain instead of 9 multiply-adds per output: 36 products per 16 outputs against 64 in the
// F(2x2,3x3) form and 144 in the direct form.  A board of 15x15 or 16x16 is a 4x4 grid of such tiles, i.e.
// exactly the 16 rows of the M side of one matrix instruction, so a 16 -> 16 channel layer is 36 (transformed
// positions) x 4 (K steps of four input channels) = 144 matrix instructions against 256 and 504.
//   One wavefront per position, two wavefronts per SIMD (eight positions per CU, no barrier anywhere): a
//   wavefront alone on a SIMD issues a vector instruction only every 5 - 8 cycles and its vector instructions
//   do not overlap with its own matrix instructions, two wavefronts fill each other's gaps.  256 registers per
//   wavefront then: the 36 accumulators of four registers each leave 112, so everything else streams.
//   Lane (m, kq) of K step ks loads the 6x6 patch of tile m (tile row m >> 2, tile column m & 3) of channel
//   4*ks + kq and forms V = Bt d B in two halves: the column transforms of the whole patch (t = Bt d, 72
//   operations, in the registers of the patch), then ROW BY ROW the row transform (12 operations) whose six
//   results are the A operands of the six matrix instructions issued right behind it; the registers of a row
//   of t then take the same row of the next K step's patch.  The B operands are the host-transformed weights
//   U = G g Gt with the batch-norm scale folded in, streamed from memory through a ring of 12 registers, two
//   rows ahead of their use and across layer ends (all wavefronts read the same 36 KB per layer: cache hits).
//   After the last K step lane (co, kq) holds, for the four tiles of tile row kq and its output channel co, the
//   36 transformed sums; Y = At M A gives the 4x4 outputs.  The batch-norm shift enters through the initial
//   value of the one accumulator (transformed position (1, 1)) that feeds every output with weight one.
//   The order of a row's instructions is the one written here: every row ends in a (wave-uniform) branch, which the
//   compiler does not schedule across; as one basic block per K step the same code ran 25 % slower.
// Patches of the last tile row / column reach past the halo: at 15x15 that is the next row's left halo cell, the
// next plane's top halo or the zero tail behind the last plane; at 16x16 they end on the halo exactly; at 13x13 and
// 14x14 they wrap into cells of the next row / the next plane, i.e. finite values that only the outputs outside the
// board depend on (the others see them through sums that cancel, at the level of the form's rounding).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <utility>

#include "../../include/azmcts.h"
#include "az_common.h"

#include "az_net_team.h"
#include "az_net_parts.h"

namespace {
using namespace aznet;

template <int S> struct W43Geo {
    using G = Geo<S>;
    static_assert(S >= 13 && S <= 16, "a 4x4 grid of tiles; what a patch reads past the board is zero or finite");
    static constexpr int LDS_BYTES = (G::BUF + 64) * 4; // + the zero tail behind the last plane
    static constexpr int WGS0 = (160 * 1024 - 2048) / (LDS_BYTES + 256); // positions per CU: eight at 15x15, seven at 16x16
#ifdef AZ_W43_WGS // (experiments: fewer positions per CU)
    static constexpr int WGS = AZ_W43_WGS;
#else
    static constexpr int WGS = WGS0 > 8 ? 8 : WGS0;
#endif
};

typedef float f32x2 __attribute__((ext_vector_type(2)));
template <class T> __device__ __forceinline__ T w43_splat(float v);
template <> __device__ __forceinline__ float w43_splat<float>(float v) { return v; }
template <> __device__ __forceinline__ f32x2 w43_splat<f32x2>(float v) { return (f32x2){v, v}; }
__device__ __forceinline__ float w43_fma(float a, float b, float c) { return fmaf(a, b, c); }
__device__ __forceinline__ f32x2 w43_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }

// one-dimensional input transform Bt d of F(4, 3); T = float, or two transforms side by side (packed instructions)
template <class T> __device__ __forceinline__ void w43_bt(T d0, T d1, T d2, T d3, T d4, T d5, T &t0, T &t1, T &t2, T &t3, T &t4, T &t5)
{
    const T k4 = w43_splat<T>(4.f), m4 = w43_splat<T>(-4.f), m5 = w43_splat<T>(-5.f), k2 = w43_splat<T>(2.f), m2 = w43_splat<T>(-2.f);
    const T p = w43_fma(m4, d2, d4), q = w43_fma(m4, d1, d3), c = d4 - d2, e = d3 - d1;
    t0 = w43_fma(k4, d0, w43_fma(m5, d2, d4));
    t1 = p + q;
    t2 = p - q;
    t3 = w43_fma(k2, e, c);
    t4 = w43_fma(m2, e, c);
    t5 = w43_fma(k4, d1, w43_fma(m5, d3, d5));
}
// one-dimensional output transform At m of F(4, 3)
template <class T> __device__ __forceinline__ void w43_at(T m0, T m1, T m2, T m3, T m4, T m5, T &y0, T &y1, T &y2, T &y3)
{
    const T k2 = w43_splat<T>(2.f), k4 = w43_splat<T>(4.f), k8 = w43_splat<T>(8.f);
    const T p = m1 + m2, q = m1 - m2, r = m3 + m4, s = m3 - m4;
    y0 = (m0 + p) + r;
    y1 = w43_fma(k2, s, q);
    y2 = w43_fma(k4, r, p);
    y3 = w43_fma(k8, s, q) + m5;
}

// The parked trunk output.  VEC: tile R, row a of the lane = four consecutive floats, one 16-byte private-memory
// access per tile row (fz is a multiple of four the compiler cannot see through); else one 4-byte access per cell.
// Measured per 8192 positions, vector against scalar: 13x13 0.303 / 0.310 ms, 20x20 0.612 / 0.634 ms, but
// 15x15 0.319 / 0.315 ms (twice, same box): the form is chosen per board size (w43_park_vec).
template <int S> constexpr bool w43_park_vec() { return S != 15 && S != 16; }
template <bool VEC> __device__ __forceinline__ void w43_park(float *fc, int fz, int R, int a, const float (&row)[4])
{
    if constexpr (VEC) {
        *reinterpret_cast<f32x4 *>(__builtin_assume_aligned(fc + fz + R * 16 + a * 4, 16)) = (f32x4){row[0], row[1], row[2], row[3]};
    } else {
#pragma unroll
        for (int b = 0; b < 4; b++) fc[fz + (a * 4 + b) * 4 + R] = row[b];
    }
}
template <bool VEC> __device__ __forceinline__ void w43_unpark(const float *fc, int fz, int R, int a, float (&row)[4])
{
    if constexpr (VEC) {
        const f32x4 v = *reinterpret_cast<const f32x4 *>(__builtin_assume_aligned(fc + fz + R * 16 + a * 4, 16));
        row[0] = v[0]; row[1] = v[1]; row[2] = v[2]; row[3] = v[3];
    } else {
#pragma unroll
        for (int b = 0; b < 4; b++) row[b] = fc[fz + (a * 4 + b) * 4 + R];
    }
}

// Stores (wr), adds up (sum) or parks (keep) the 4x4 outputs y[a][b] of the lane's R-th tile (tile row kq, tile
// column R), ReLU applied here.  Cells outside the board are replaced by zero: their columns are skipped, their rows
// (last tile row only) go to the halo row below the board, which stays zero that way.  cob = co*PS + (4*kq + 1)*ROW + 1.
// The flags are wave-uniform run-time values: one instance of the layer serves the whole network (the kernel's code
// has to stay well inside the instruction cache).
template <int S> struct W43Rows {
    static constexpr int LAST = S - 12; // board rows in the last tile row (1..4); row LAST of that tile is the halo row
};
// float offset of row a of the lane's tiles from row 0 (rows past the halo row fold onto it)
template <int S> __device__ __forceinline__ int w43_row_offset(int a, int kq)
{
    if (a <= W43Rows<S>::LAST) return a * Geo<S>::ROW;
    return (kq == 3 ? W43Rows<S>::LAST : a) * Geo<S>::ROW;
}
template <int S, int R>
__device__ __forceinline__ void w43_emit(float *buf, int cob, int kq, float (&y)[4][4], bool wr, bool keep, bool sum, float *fc, int fz,
                                         float &lsum)
{
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const bool col_ok = 4 * R + b < S; // (a constant)
            float v = __builtin_amdgcn_fmed3f(y[a][b], 0.f, INFINITY);
            if (!col_ok) v = 0.f;
            if (a >= W43Rows<S>::LAST && col_ok) v = kq == 3 ? 0.f : v;
            y[a][b] = v;
        }
    if (wr) {
#pragma unroll
        for (int a = 0; a < 4; a++) {
            const int ro = w43_row_offset<S>(a, kq);
#pragma unroll
            for (int b = 0; b < 4; b++)
                if (4 * R + b < S) buf[cob + ro + 4 * R + b] = y[a][b];
        }
    }
    if (sum) {
#pragma unroll
        for (int a = 0; a < 4; a++) lsum += (y[a][0] + y[a][1]) + (y[a][2] + y[a][3]);
    }
    if (keep) {
#pragma unroll
        for (int a = 0; a < 4; a++) w43_park<w43_park_vec<S>()>(fc, fz, R, a, y[a]);
    }
}

// the trunk output back from its private copy
template <int S> __device__ __forceinline__ void w43_restore(float *buf, int lane, const float *fc, int fz)
{
    using G = Geo<S>;
    int l = lane;
    asm volatile("" : "+v"(l));
    const int kq = l >> 4;
    const int cob = (l & 15) * G::PS + (4 * kq + 1) * G::ROW + 1;
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
        for (int a = 0; a < 4; a++) {
            const int ro = w43_row_offset<S>(a, kq);
            float row[4];
            w43_unpark<w43_park_vec<S>()>(fc, fz, r, a, row);
#pragma unroll
            for (int b = 0; b < 4; b++)
                if (4 * r + b < S) buf[cob + ro + 4 * r + b] = row[b];
        }
}

#ifndef AZ_W43_PK_IN
#define AZ_W43_PK_IN 1
#endif
#ifndef AZ_W43_PRIO_ROWS
#define AZ_W43_PRIO_ROWS 0
#endif
#ifndef AZ_W43_PRIO_OUT
#define AZ_W43_PRIO_OUT 3
#endif
constexpr int W43_RING = 12; // weight registers in flight: two rows of six matrix instructions
static_assert(36 % W43_RING == 0, "every K step starts at ring slot 0");

// t = Bt d in place: the six column transforms of a patch, two columns side by side (packed instructions)
__device__ __forceinline__ void w43_columns(float (&t)[36])
{
#if AZ_W43_PK_IN
#pragma unroll
    for (int j = 0; j < 6; j += 2) {
        f32x2 c[6], o[6];
#pragma unroll
        for (int i = 0; i < 6; i++) c[i] = (f32x2){t[i * 6 + j], t[i * 6 + j + 1]};
        w43_bt(c[0], c[1], c[2], c[3], c[4], c[5], o[0], o[1], o[2], o[3], o[4], o[5]);
#pragma unroll
        for (int i = 0; i < 6; i++) {
            t[i * 6 + j] = o[i][0];
            t[i * 6 + j + 1] = o[i][1];
        }
    }
#else
#pragma unroll
    for (int j = 0; j < 6; j++)
        w43_bt(t[j], t[6 + j], t[12 + j], t[18 + j], t[24 + j], t[30 + j], t[j], t[6 + j], t[12 + j], t[18 + j], t[24 + j], t[30 + j]);
#endif
}

// One K step: row by row the row transform of t (six A operands), six matrix instructions, the six weights of the
// row after the next into the ring slots just used, and (more) the same row of the next K step's patch into the
// registers of t's row; then that patch's column transforms.  FIRST: the accumulators start here (from the
// batch-norm shift).  wcur: this K step's weights [18 pairs][64 lanes][2]; wnxt: the block whose first W43_RING entries the last
// two rows fetch (the next K step, or the next layer); wo: the lane's byte offset inside a row of weights.
template <int S, bool FIRST>
__device__ __forceinline__ void w43_kstep(const float *pnext, bool more, float (&t)[36], f32x4 (&acc)[36], float (&w)[W43_RING],
                                          const char *wcur, const char *wnxt, unsigned wo, float sh)
{
    using G = Geo<S>;
    const f32x4 shv = (f32x4){sh, sh, sh, sh}, zero = (f32x4){0.f, 0.f, 0.f, 0.f};
    static_for<6>([&](auto ii) {
        constexpr int i = decltype(ii)::value;
        __builtin_amdgcn_sched_barrier(0);
        float a6[6];
        w43_bt(t[i * 6], t[i * 6 + 1], t[i * 6 + 2], t[i * 6 + 3], t[i * 6 + 4], t[i * 6 + 5], a6[0], a6[1], a6[2], a6[3], a6[4], a6[5]);
#pragma unroll
        for (int j = 0; j < 6; j++) {
            const int x = i * 6 + j;
            if (FIRST) acc[x] = __builtin_amdgcn_mfma_f32_16x16x4f32(a6[j], w[x % W43_RING], x == 7 ? shv : zero, 0, 0, 0);
            else acc[x] = __builtin_amdgcn_mfma_f32_16x16x4f32(a6[j], w[x % W43_RING], acc[x], 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < 6; j += 2) { // (weights lie in pairs per lane: one 8-byte request per two matrix instructions)
            const int g = i * 6 + j + W43_RING;
            const char *src = g < 36 ? wcur + (g / 2) * 512 : wnxt + ((g - 36) / 2) * 512;
            const f32x2 v = *reinterpret_cast<const f32x2 *>(src + wo);
            w[g % W43_RING] = v[0];
            w[(g + 1) % W43_RING] = v[1];
        }
        if (more) {
            const float *p = pnext + i * G::ROW;
#pragma unroll
            for (int j = 0; j < 6; j++) t[i * 6 + j] = p[j];
        }
    });
    __builtin_amdgcn_sched_barrier(0);
    if (more) w43_columns(t);
}

// One 3x3 layer, in place.  ksteps: 4 for 16 input channels, 1 for the two input planes (K indices 2..3 carry zero
// weights and read planes 0..1 again).  `layer`: the layer's weights [ksteps][36][64] in the order the matrix
// instructions take them; w[] holds the first W43_RING of them on entry and the first W43_RING of `nextp` (the
// layer that runs next) on return.  wr / keep / sum: what happens to the outputs (w43_emit).
#ifdef AZ_NET_STAMPS // diagnostic build: cycle sums of the layer's phases (prologue, K steps, output transform + stores)
#define W43_STAMP_PARAM , unsigned long long (&stl)[4]
#define W43_STAMP_ARG , stl
#define LST(i) do { unsigned long long t_ = __builtin_readcyclecounter(); stl[i] += t_ - stl[3]; stl[3] = t_; } while (0)
#else
#define W43_STAMP_PARAM
#define W43_STAMP_ARG
#define LST(i) do { } while (0)
#endif
template <int S>
__device__ __forceinline__ void w43_layer(float *buf, float (&w)[W43_RING], const float *layer, const float *nextp, float sh, int lane,
                                          int ksteps, bool wr, bool keep, bool sum, float *fc, int fz, float &lsum W43_STAMP_PARAM)
{
#ifdef AZ_NET_STAMPS
    stl[3] = __builtin_readcyclecounter();
#endif
    using G = Geo<S>;
    int l = lane;
    asm volatile("" : "+v"(l)); // (addresses are formed per layer instead of living in registers across the position loop)
    const int kq = l >> 4, co = l & 15;
    const int kqp = ksteps == 4 ? kq : (kq & 1); // plane of this lane's K index in K step 0
    const unsigned wo = (unsigned)l * 8u;
    const float *p0 = buf + kqp * G::PS + 4 * (co >> 2) * G::ROW + 4 * (co & 3); // channels and cells are constant offsets from here
    float t[36];
#pragma unroll
    for (int i = 0; i < 6; i++)
#pragma unroll
        for (int j = 0; j < 6; j++) t[i * 6 + j] = p0[i * G::ROW + j];
    w43_columns(t);
    LST(0);
    f32x4 acc[36];
    const char *wl = reinterpret_cast<const char *>(layer), *wn = reinterpret_cast<const char *>(nextp);
    __builtin_amdgcn_s_setprio(AZ_W43_PRIO_ROWS);
    w43_kstep<S, true>(p0 + 4 * G::PS, ksteps > 1, t, acc, w, wl, ksteps > 1 ? wl + 36 * 256 : wn, wo, sh);
#pragma nounroll
    for (int ks = 1; ks < ksteps; ks++) {
        const bool more = ks + 1 < ksteps;
        w43_kstep<S, false>(p0 + (ks + 1) * 4 * G::PS, more, t, acc, w, wl + ks * 36 * 256, more ? wl + (ks + 1) * 36 * 256 : wn, wo, sh);
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(AZ_W43_PRIO_OUT);
#ifdef AZ_NET_STAMPS
    asm volatile("s_nop 0" ::"v"(acc[35][3])); // (the last matrix instruction has delivered)
#endif
    LST(1);
    const int cob = co * G::PS + (4 * kq + 1) * G::ROW + 1;
    // Y = At M A, two tiles side by side (the components of an accumulator are consecutive registers)
    static_for<2>([&](auto hh) {
        constexpr int h = decltype(hh)::value;
        f32x2 m[36], s2[4][6], y2[4][4];
#pragma unroll
        for (int x = 0; x < 36; x++) m[x] = h == 0 ? __builtin_shufflevector(acc[x], acc[x], 0, 1) : __builtin_shufflevector(acc[x], acc[x], 2, 3);
#pragma unroll
        for (int j = 0; j < 6; j++) w43_at(m[j], m[6 + j], m[12 + j], m[18 + j], m[24 + j], m[30 + j], s2[0][j], s2[1][j], s2[2][j], s2[3][j]);
#pragma unroll
        for (int q = 0; q < 4; q++)
            w43_at(s2[q][0], s2[q][1], s2[q][2], s2[q][3], s2[q][4], s2[q][5], y2[q][0], y2[q][1], y2[q][2], y2[q][3]);
        float y[4][4];
#pragma unroll
        for (int q = 0; q < 4; q++)
#pragma unroll
            for (int b2 = 0; b2 < 4; b2++) y[q][b2] = y2[q][b2][0];
        w43_emit<S, 2 * h>(buf, cob, kq, y, wr, keep, sum, fc, fz, lsum);
#pragma unroll
        for (int q = 0; q < 4; q++)
#pragma unroll
            for (int b2 = 0; b2 < 4; b2++) y[q][b2] = y2[q][b2][1];
        w43_emit<S, 2 * h + 1>(buf, cob, kq, y, wr, keep, sum, fc, fz, lsum);
    });
    LST(2);
}

template <int S>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_net_w43(
    const float *boards, const int *leaf_kind, const unsigned *expand_mask, int n, const float *prm, int front, int pol, int val,
    float *value_out, float *logp_out)
{
    using G = Geo<S>;
    extern __shared__ float lds[];
    float *buf = lds;
    int *todo = reinterpret_cast<int *>(lds + G::BUF + 64); // [ceil(n / gridDim.x)] positions of this workgroup
    const int lane = threadIdx.x;
    const int co = lane & 15, kq = lane >> 4;
    for (int i = lane; i < G::BUF + 64; i += 64) buf[i] = 0.f; // halos (and the tail) stay zero from here on
    const int nl = front - 1 + pol + val;
    const float *p_policy = prm + FIRST_FLOATS + (size_t)nl * LAYER_FLOATS;
    const float *p_value = p_policy + POLICY_FLOATS;
    const float *p_w43 = p_value + VALUE_FLOATS + (size_t)nl * WINO_FLOATS + WINO_FIRST_FLOATS;
    // execution order: trunk, value branch, policy branch (parameter order: trunk, policy, value)
    auto layer_params = [&](int e) {
        const int idx = e < front - 1 ? e : (e < front - 1 + val ? e + pol : e - val);
        return p_w43 + (size_t)idx * W43_FLOATS;
    };
    const float *p_first = p_w43 + (size_t)nl * W43_FLOATS;
    float wpb[4]; // policy head as a B operand: B[k = ci][n = tap]
#pragma unroll
    for (int ks = 0; ks < 4; ks++) wpb[ks] = co < 9 ? p_policy[co * CH + ks * 4 + kq] : 0.f;
    const float p_bias = p_policy[9 * CH];
    const float wv = p_value[co], v_bias = p_value[CH];

    int n_mine;
    const bool listed = expand_mask != nullptr || leaf_kind != nullptr; // positions come from todo[]
    if (expand_mask != nullptr) {
        n_mine = list_positions_mask(expand_mask, n, todo, lane);
    } else if (leaf_kind != nullptr) {
        n_mine = list_positions<64>(leaf_kind, n, todo, nullptr, lane);
    } else {
        n_mine = n > (int)blockIdx.x ? (n - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x : 0;
    }
    __syncthreads(); // (a single wavefront: orders the LDS traffic)

#ifdef AZ_NET_STAMPS
    unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_t = 0, st_0 = __builtin_readcyclecounter(), stl[4] = {0, 0, 0, 0};
#define NST(i) do { unsigned long long t_ = __builtin_readcyclecounter(); st_acc[i] += t_ - st_t; st_t = t_; } while (0)
#else
#define NST(i) do { } while (0)
#endif
    // The trunk output, which both branches start from, is parked in private memory while the value branch owns the
    // buffer; indexed through a value the compiler cannot see through, so that the array really lives in memory.
    __attribute__((aligned(16))) float fc[64];
    int fz;
    asm volatile("s_mov_b32 %0, 0" : "=s"(fz));
    constexpr int PPL = (G::SIZE + 63) / 64; // pixels this lane loads, gathers and stores: lane, lane + 64, ...
    float in0[PPL], in1[PPL];                // the next position's input planes, requested one position ahead
    auto request_board = [&](int pos) {
        const float *bp = boards + (size_t)pos * CIN * G::SIZE;
#pragma unroll
        for (int i = 0; i < PPL; i++) {
            const int p = lane + 64 * i;
            in0[i] = p < G::SIZE ? bp[p] : 0.f;
            in1[i] = p < G::SIZE ? bp[G::SIZE + p] : 0.f;
        }
    };
    if (n_mine > 0) request_board(listed ? todo[0] : (int)blockIdx.x);
    float w[W43_RING]; // the weight ring: starts with the input layer's first rows, every layer refills it for the next
#pragma unroll
    for (int g = 0; g < W43_RING; g += 2) {
        const f32x2 v = *reinterpret_cast<const f32x2 *>(p_first + (g / 2) * 128 + lane * 2);
        w[g] = v[0];
        w[g + 1] = v[1];
    }
    const float u0sh = p_first[36 * 64 + co];
    float ush = 0.f;
    for (int it = 0; it < n_mine; it++) {
        const int pos = listed ? todo[it] : (int)blockIdx.x + it * (int)gridDim.x;
#ifdef AZ_NET_STAMPS
        st_t = __builtin_readcyclecounter();
#endif
#pragma unroll
        for (int i = 0; i < PPL; i++) { // the two input planes (requested while the previous position was in its heads)
            const int p = lane + 64 * i;
            if (p < G::SIZE) {
                const int ctr = pix_index<S>(p);
                buf[ctr] = in0[i];
                buf[G::PS + ctr] = in1[i];
            }
        }
        // 13x13 and 14x14: the patches of plane 1's last tile row reach rows 15..17, i.e. the first rows of plane 2,
        // which is no input plane and still holds what the previous position left there.  Zeroed, so that a position's
        // outputs never depend on what its workgroup evaluated before (sums that cancel only algebraically would
        // otherwise carry a rounding residue of those leftovers into the last board row).
        if constexpr (S < 15) {
            static_assert(3 * G::ROW <= 64, "one cell per lane");
            if (lane < 3 * G::ROW) buf[2 * G::PS + lane] = 0.f;
        }
        __syncthreads();
        NST(0);
        float vsum = 0.f;
        // e = -1: the input layer (also the trunk's last layer when front == 1), then the 16-channel layers in execution order
        for (int e = -1; e < nl; e++) {
            const bool last_trunk = e == front - 2;
            const bool last_value = val > 0 && e == front - 1 + val - 1;
            const bool first_policy = pol > 0 && e == front - 1 + val;
            const float *layer = e < 0 ? p_first : layer_params(e);
            const float *nextp = e + 1 < nl ? layer_params(e + 1) : p_first; // (the last layer fetches the next position's first)
            if (first_policy && val > 0) { // the trunk output comes back
                w43_restore<S>(buf, lane, fc, fz);
                __syncthreads();
            }
            const float sh = e < 0 ? u0sh : ush;
            if (e >= 0 || nl > 0) ush = layer_params(e + 1 < nl ? e + 1 : 0)[4 * 36 * 64 + co]; // (requested a layer ahead)
            NST(4);
            float lsum = 0.f;
            const bool sum = last_value || (last_trunk && val == 0);
            w43_layer<S>(buf, w, layer, nextp, sh, lane, e < 0 ? 1 : 4, !last_value, last_trunk && val > 0, sum, fc, fz, lsum W43_STAMP_ARG);
            if (sum) vsum = lsum;
            __syncthreads();
            if (e < 0) NST(6);
            else NST(1);
        }
        // policy head input: the policy branch's output, or the trunk output when there is no such branch
        if (pol == 0 && val > 0) {
            w43_restore<S>(buf, lane, fc, fz);
            __syncthreads();
        }
        // ---- value head: board average of the 1x1 convolution + bias, tanh
        const float vpart = wave_reduce<false>(wv * vsum);
        if (lane == 0) value_out[pos] = tanhf(vpart / (float)G::SIZE + v_bias);
        if (it + 1 < n_mine) request_board(listed ? todo[it + 1] : (int)blockIdx.x + (it + 1) * (int)gridDim.x);
        // ---- policy head: per-tap partial planes on the matrix cores, written over the input
        policy_taps_wide<S>(buf, wpb, lane);
        __syncthreads();
        // gather of the nine planes + bias, log-softmax over the board
        float logit[PPL];
        float lmax = -INFINITY;
#pragma unroll
        for (int i = 0; i < PPL; i++) {
            const int p = lane + 64 * i;
            const int ctr = pix_index<S>(p < G::SIZE ? p : 0);
            float s = p_bias;
#pragma unroll
            for (int tap = 0; tap < 9; tap++) s += buf[tap * G::PS + ctr + (tap / 3 - 1) * G::ROW + (tap % 3 - 1)];
            logit[i] = p < G::SIZE ? s : -INFINITY;
            lmax = fmaxf(lmax, logit[i]);
        }
        lmax = wave_reduce<true>(lmax);
        float esum = 0.f;
#pragma unroll
        for (int i = 0; i < PPL; i++) esum += lane + 64 * i < G::SIZE ? expf(logit[i] - lmax) : 0.f;
        esum = wave_reduce<false>(esum);
        const float lse = lmax + logf(esum);
#pragma unroll
        for (int i = 0; i < PPL; i++)
            if (lane + 64 * i < G::SIZE) logp_out[(size_t)pos * G::SIZE + lane + 64 * i] = logit[i] - lse;
        __syncthreads(); // the gather is over before the next position's input lands in the buffer
        NST(3);
        // planes 2..15 keep leftovers (partial planes, old activations): the input layer reads planes 0..1 (and, on boards
        // the tile grid overhangs, the first rows of plane 2, zeroed above)
    }
#ifdef AZ_NET_STAMPS
    if (lane == 0 && n_mine > 0 && leaf_kind == nullptr) { // diagnostic build: cycle sums over the log-priors of the first position
        st_acc[5] = __builtin_readcyclecounter() - st_0;
        st_acc[7] = (unsigned long long)n_mine;
        unsigned long long *o = reinterpret_cast<unsigned long long *>(logp_out + (size_t)blockIdx.x * G::SIZE);
        st_acc[2] = stl[0] * 1000000ull + stl[1] / 8; // (prologues, K steps / 8)
        st_acc[0] = stl[2]; // (output transforms and stores; replaces the input load)
        for (int q = 0; q < 8; q++) o[q] = st_acc[q];
    }
#endif
}

template <int S> int launch_w43(const float *boards, const int *leaf_kind, const unsigned *expand_mask, int n, const float *prm,
                                int front, int pol, int val, float *value_out, float *logp_out, hipStream_t st)
{
    using W = W43Geo<S>;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    int grid = cus * W::WGS;
    if (grid > n) grid = n;
    const size_t lds_bytes = (size_t)W::LDS_BYTES + (size_t)((n + grid - 1) / grid) * sizeof(int);
    static size_t configured[64] = {}; // largest size the kernel has been opened up to, per device
    if (dev < 0 || dev >= 64 || lds_bytes > configured[dev]) {
        if (hipFuncSetAttribute((const void *)k_net_w43<S>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess)
            return AZ_ERR_HIP;
        if (dev >= 0 && dev < 64) configured[dev] = lds_bytes;
    }
    hipLaunchKernelGGL(k_net_w43<S>, dim3(grid), dim3(64), lds_bytes, st, boards, leaf_kind, expand_mask, n, prm, front, pol, val,
                       value_out, logp_out);
    return hipGetLastError() == hipSuccess ? AZ_OK : AZ_ERR_HIP;
}

// ---------------------------------------------------------------------------------
// 17x17 - 20x20: a 5x5 grid of tiles = two blocks of 16 (seven dummies), a TEAM of two wavefronts per position, one block
// each (a board of that size fills 31 KB of LDS: four positions per CU, and two wavefronts per SIMD need eight
// wavefronts).  Both wavefronts read their patches from the one buffer and write their outputs back into it, so a
// barrier separates a layer's last patch read from its first store; the heads split the board rows.
template <int S_> struct W43TeamGeo {
    using G = Geo<S_>;
    static_assert(S_ >= 17 && S_ <= 20, "a 5x5 grid of tiles");
    static constexpr bool EXACT = S_ % 4 == 0; // the tiles end on the board edge: patches end on the halo, no partial tiles
    static constexpr int S = S_, SIZE = G::SIZE, ROW = G::ROW, PS = G::PS; // (what policy_softmax reads)
    static constexpr int NW = 2, NTHR = 128;
    static constexpr int TW = (S + 3) / 4, TILES = TW * TW;
    static constexpr unsigned DIV = 65536 / TW + 1; // T / TW == (T * DIV) >> 16 for T < 128
    static constexpr int TAIL = 64;
    static constexpr int LDS_BYTES = (G::BUF + TAIL + 16 + SIZE) * 4; // buffer, zero tail, reduction cells, logits
    static constexpr int WGS0 = (160 * 1024 - 2048) / (LDS_BYTES + 256);
    static constexpr int WGS = WGS0 > 4 ? 4 : WGS0;
    static_assert(TILES <= 32 && WGS >= 1, "two blocks");
};

// w43_emit for a tile whose place is a run-time value (T: tile index, tiles past the grid are dummies); R: the slot of
// the tile in the lane's private copy
template <int S, int R>
__device__ __forceinline__ void w43t_emit(float *buf, int co, unsigned T, float (&y)[4][4], bool wr, bool keep, bool sum, float *fc, int fz,
                                          float &lsum)
{
    using G = Geo<S>;
    using W = W43TeamGeo<S>;
    const bool valid = T < (unsigned)W::TILES;
    const unsigned ty = (T * W::DIV) >> 16, tx = T - ty * W::TW;
    // (boards that the tile grid overhangs: cells outside the board are replaced by zero and folded onto the halo
    // row below / the halo column right of the board, which stay zero that way)
    const int rows_in = valid ? S - 4 * (int)ty : 0, cols_in = S - 4 * (int)tx;
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const float v = __builtin_amdgcn_fmed3f(y[a][b], 0.f, INFINITY);
            if constexpr (W::EXACT) y[a][b] = valid ? v : 0.f;
            else y[a][b] = (a < rows_in && b < cols_in) ? v : 0.f;
        }
    if (wr && valid) {
        if constexpr (W::EXACT) {
            float *c = buf + co * G::PS + (int)((4 * ty + 1) * G::ROW + 4 * tx + 1);
#pragma unroll
            for (int a = 0; a < 4; a++)
#pragma unroll
                for (int b = 0; b < 4; b++) c[a * G::ROW + b] = y[a][b];
        } else {
            float *c = buf + co * G::PS + G::ROW + 1;
#pragma unroll
            for (int a = 0; a < 4; a++) {
                const int ra = (a < rows_in ? 4 * (int)ty + a : S) * G::ROW;
#pragma unroll
                for (int b = 0; b < 4; b++) c[ra + (b < cols_in ? 4 * (int)tx + b : S)] = y[a][b];
            }
        }
    }
    if (sum) {
#pragma unroll
        for (int a = 0; a < 4; a++) lsum += (y[a][0] + y[a][1]) + (y[a][2] + y[a][3]);
    }
    if (keep) {
#pragma unroll
        for (int a = 0; a < 4; a++) w43_park<w43_park_vec<S>()>(fc, fz, R, a, y[a]);
    }
}

template <int S> __device__ __forceinline__ void w43t_restore(float *buf, int wave, int lane, const float *fc, int fz)
{
    using G = Geo<S>;
    using W = W43TeamGeo<S>;
    int l = lane;
    asm volatile("" : "+v"(l));
    const int co = l & 15, kq = l >> 4;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const unsigned T = (unsigned)(16 * wave + 4 * kq + r);
        const unsigned ty = (T * W::DIV) >> 16, tx = T - ty * W::TW;
        if (T < (unsigned)W::TILES) {
            if constexpr (W::EXACT) {
                float *c = buf + co * G::PS + (int)((4 * ty + 1) * G::ROW + 4 * tx + 1);
#pragma unroll
                for (int a = 0; a < 4; a++) {
                    float row[4];
                    w43_unpark<w43_park_vec<S>()>(fc, fz, r, a, row);
#pragma unroll
                    for (int b = 0; b < 4; b++) c[a * G::ROW + b] = row[b];
                }
            } else { // (cells outside the board were parked as zeros: folded onto the halo as in w43t_emit)
                const int rows_in = S - 4 * (int)ty, cols_in = S - 4 * (int)tx;
                float *c = buf + co * G::PS + G::ROW + 1;
#pragma unroll
                for (int a = 0; a < 4; a++) {
                    const int ra = (a < rows_in ? 4 * (int)ty + a : S) * G::ROW;
                    float row[4];
                    w43_unpark<w43_park_vec<S>()>(fc, fz, r, a, row);
#pragma unroll
                    for (int b = 0; b < 4; b++) c[ra + (b < cols_in ? 4 * (int)tx + b : S)] = row[b];
                }
            }
        }
    }
}

// w43_layer for the block of tiles of wavefront `wave` of a team; contains one barrier (every wavefront of the
// workgroup has to call it)
template <int S>
__device__ __forceinline__ void w43t_layer(float *buf, float (&w)[W43_RING], const float *layer, const float *nextp, float sh, int wave,
                                           int lane, int ksteps, bool wr, bool keep, bool sum, float *fc, int fz, float &lsum)
{
    using G = Geo<S>;
    using W = W43TeamGeo<S>;
    int l = lane;
    asm volatile("" : "+v"(l));
    const int kq = l >> 4, co = l & 15;
    const int kqp = ksteps == 4 ? kq : (kq & 1);
    const unsigned wo = (unsigned)l * 8u;
    unsigned Tin = (unsigned)(16 * wave + co);
    Tin = Tin < (unsigned)W::TILES ? Tin : (unsigned)W::TILES - 1; // (a dummy tile reads the last tile again)
    const unsigned pty = (Tin * W::DIV) >> 16, ptx = Tin - pty * W::TW;
    const float *p0 = buf + kqp * G::PS + (int)(4 * pty * G::ROW + 4 * ptx);
    float t[36];
#pragma unroll
    for (int i = 0; i < 6; i++)
#pragma unroll
        for (int j = 0; j < 6; j++) t[i * 6 + j] = p0[i * G::ROW + j];
    w43_columns(t);
    f32x4 acc[36];
    const char *wl = reinterpret_cast<const char *>(layer), *wn = reinterpret_cast<const char *>(nextp);
    __builtin_amdgcn_s_setprio(AZ_W43_PRIO_ROWS);
    w43_kstep<S, true>(p0 + 4 * G::PS, ksteps > 1, t, acc, w, wl, ksteps > 1 ? wl + 36 * 256 : wn, wo, sh);
#pragma nounroll
    for (int ks = 1; ks < ksteps; ks++) {
        const bool more = ks + 1 < ksteps;
        w43_kstep<S, false>(p0 + (ks + 1) * 4 * G::PS, more, t, acc, w, wl + ks * 36 * 256, more ? wl + (ks + 1) * 36 * 256 : wn, wo, sh);
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(AZ_W43_PRIO_OUT);
    __syncthreads(); // both wavefronts have read their patches: the outputs may overwrite them
    static_for<2>([&](auto hh) {
        constexpr int h = decltype(hh)::value;
        f32x2 m[36], s2[4][6], y2[4][4];
#pragma unroll
        for (int x = 0; x < 36; x++) m[x] = h == 0 ? __builtin_shufflevector(acc[x], acc[x], 0, 1) : __builtin_shufflevector(acc[x], acc[x], 2, 3);
#pragma unroll
        for (int j = 0; j < 6; j++) w43_at(m[j], m[6 + j], m[12 + j], m[18 + j], m[24 + j], m[30 + j], s2[0][j], s2[1][j], s2[2][j], s2[3][j]);
#pragma unroll
        for (int q = 0; q < 4; q++)
            w43_at(s2[q][0], s2[q][1], s2[q][2], s2[q][3], s2[q][4], s2[q][5], y2[q][0], y2[q][1], y2[q][2], y2[q][3]);
        float y[4][4];
#pragma unroll
        for (int q = 0; q < 4; q++)
#pragma unroll
            for (int b2 = 0; b2 < 4; b2++) y[q][b2] = y2[q][b2][0];
        w43t_emit<S, 2 * h>(buf, co, (unsigned)(16 * wave + 4 * kq + 2 * h), y, wr, keep, sum, fc, fz, lsum);
#pragma unroll
        for (int q = 0; q < 4; q++)
#pragma unroll
            for (int b2 = 0; b2 < 4; b2++) y[q][b2] = y2[q][b2][1];
        w43t_emit<S, 2 * h + 1>(buf, co, (unsigned)(16 * wave + 4 * kq + 2 * h + 1), y, wr, keep, sum, fc, fz, lsum);
    });
}

template <int S>
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_net_w43t(
    const float *boards, const int *leaf_kind, const unsigned *expand_mask, int n, const float *prm, int front, int pol, int val,
    float *value_out, float *logp_out)
{
    using G = Geo<S>;
    using W = W43TeamGeo<S>;
    extern __shared__ float lds[];
    float *buf = lds;
    float *red = lds + G::BUF + W::TAIL; // [16]: two reduction rows of up to 8 wavefronts
    float *logit = red + 16;             // [SIZE]
    int *todo = reinterpret_cast<int *>(logit + G::SIZE); // [ceil(n / gridDim.x)] positions of this workgroup
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int co = lane & 15, kq = lane >> 4;
    for (int i = tid; i < G::BUF + W::TAIL; i += W::NTHR) buf[i] = 0.f; // halos (and the tail) stay zero from here on
    __syncthreads();
    const int nl = front - 1 + pol + val;
    const float *p_policy = prm + FIRST_FLOATS + (size_t)nl * LAYER_FLOATS;
    const float *p_value = p_policy + POLICY_FLOATS;
    const float *p_w43 = p_value + VALUE_FLOATS + (size_t)nl * WINO_FLOATS + WINO_FIRST_FLOATS;
    // execution order: trunk, value branch, policy branch (parameter order: trunk, policy, value)
    auto layer_params = [&](int e) {
        const int idx = e < front - 1 ? e : (e < front - 1 + val ? e + pol : e - val);
        return p_w43 + (size_t)idx * W43_FLOATS;
    };
    const float *p_first = p_w43 + (size_t)nl * W43_FLOATS;
    float wpb[4]; // policy head as a B operand: B[k = ci][n = tap]
#pragma unroll
    for (int ks = 0; ks < 4; ks++) wpb[ks] = co < 9 ? p_policy[co * CH + ks * 4 + kq] : 0.f;
    const float p_bias = p_policy[9 * CH];
    const float wv = p_value[co], v_bias = p_value[CH];

    int n_mine;
    const bool listed = expand_mask != nullptr || leaf_kind != nullptr; // positions come from todo[]
    if (expand_mask != nullptr) {
        n_mine = list_positions_mask(expand_mask, n, todo, lane); // (both wavefronts write the same list)
    } else if (leaf_kind != nullptr) {
        n_mine = list_positions<W::NTHR>(leaf_kind, n, todo, reinterpret_cast<int *>(red), tid);
    } else {
        n_mine = n > (int)blockIdx.x ? (n - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x : 0;
    }
    __syncthreads();

    __attribute__((aligned(16))) float fc[64]; // the trunk output of this wavefront's tiles, parked in private memory while the value branch owns the buffer
    int fz;
    asm volatile("s_mov_b32 %0, 0" : "=s"(fz));
    constexpr int PPL = (G::SIZE + W::NTHR - 1) / W::NTHR; // pixels this thread loads: tid, tid + 128, ...
    float in0[PPL], in1[PPL];                             // the next position's input planes, requested one position ahead
    auto request_board = [&](int pos) {
        const float *bp = boards + (size_t)pos * CIN * G::SIZE;
#pragma unroll
        for (int i = 0; i < PPL; i++) {
            const int p = tid + W::NTHR * i;
            in0[i] = p < G::SIZE ? bp[p] : 0.f;
            in1[i] = p < G::SIZE ? bp[G::SIZE + p] : 0.f;
        }
    };
    if (n_mine > 0) request_board(listed ? todo[0] : (int)blockIdx.x);
    float w[W43_RING];
#pragma unroll
    for (int g = 0; g < W43_RING; g += 2) {
        const f32x2 v = *reinterpret_cast<const f32x2 *>(p_first + (g / 2) * 128 + lane * 2);
        w[g] = v[0];
        w[g + 1] = v[1];
    }
    const float u0sh = p_first[36 * 64 + co];
    float ush = 0.f;
    for (int it = 0; it < n_mine; it++) {
        const int pos = listed ? todo[it] : (int)blockIdx.x + it * (int)gridDim.x;
#pragma unroll
        for (int i = 0; i < PPL; i++) { // the two input planes (requested while the previous position was in its heads)
            const int p = tid + W::NTHR * i;
            if (p < G::SIZE) {
                const int ctr = pix_index<S>(p);
                buf[ctr] = in0[i];
                buf[G::PS + ctr] = in1[i];
            }
        }
        // 17x17 and 18x18: the patches of plane 1's last tile row reach rows 19..21, the first rows of plane 2, which
        // still holds the previous position's leftovers: zeroed, so that the outputs depend on the position alone
        if constexpr (S < 19) {
            static_assert(3 * G::ROW <= W::NTHR, "one cell per thread");
            if (tid < 3 * G::ROW) buf[2 * G::PS + tid] = 0.f;
        }
        __syncthreads();
        float vsum = 0.f;
        // e = -1: the input layer (also the trunk's last layer when front == 1), then the 16-channel layers in execution order
        for (int e = -1; e < nl; e++) {
            const bool last_trunk = e == front - 2;
            const bool last_value = val > 0 && e == front - 1 + val - 1;
            const bool first_policy = pol > 0 && e == front - 1 + val;
            const float *layer = e < 0 ? p_first : layer_params(e);
            const float *nextp = e + 1 < nl ? layer_params(e + 1) : p_first; // (the last layer fetches the next position's first)
            if (first_policy && val > 0) { // the trunk output comes back
                w43t_restore<S>(buf, wave, lane, fc, fz);
                __syncthreads();
            }
            const float sh = e < 0 ? u0sh : ush;
            if (e >= 0 || nl > 0) ush = layer_params(e + 1 < nl ? e + 1 : 0)[4 * 36 * 64 + co]; // (requested a layer ahead)
            float lsum = 0.f;
            const bool sum = last_value || (last_trunk && val == 0);
            w43t_layer<S>(buf, w, layer, nextp, sh, wave, lane, e < 0 ? 1 : 4, !last_value, last_trunk && val > 0, sum, fc, fz, lsum);
            if (sum) vsum = lsum;
            __syncthreads();
        }
        // policy head input: the policy branch's output, or the trunk output when there is no such branch
        if (pol == 0 && val > 0) {
            w43t_restore<S>(buf, wave, lane, fc, fz);
            __syncthreads();
        }
        // ---- value head: board average of the 1x1 convolution + bias, tanh
        const float vpart = wave_reduce<false>(wv * vsum);
        if (it + 1 < n_mine) request_board(listed ? todo[it + 1] : (int)blockIdx.x + (it + 1) * (int)gridDim.x);
        // ---- policy head: per-tap partial planes on the matrix cores, written over the input; the wavefronts share the row groups
        policy_taps_wide<S>(buf, wpb, lane, wave, W::NW);
        if (lane == 0) red[wave] = vpart;
        __syncthreads();
        if (tid == 0) value_out[pos] = tanhf((red[0] + red[1]) / (float)G::SIZE + v_bias);
        __syncthreads(); // red is reused by the softmax
        policy_softmax<W>(buf, p_bias, red, logit, logp_out + (size_t)pos * G::SIZE, tid, wave, lane);
        __syncthreads(); // the gather is over before the next position's input lands in the buffer
    }
}

template <int S> int launch_w43t(const float *boards, const int *leaf_kind, const unsigned *expand_mask, int n, const float *prm,
                                 int front, int pol, int val, float *value_out, float *logp_out, hipStream_t st)
{
    using W = W43TeamGeo<S>;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    int grid = cus * W::WGS;
    if (grid > n) grid = n;
    const size_t lds_bytes = (size_t)W::LDS_BYTES + (size_t)((n + grid - 1) / grid) * sizeof(int);
    static size_t configured[64] = {}; // largest size the kernel has been opened up to, per device
    if (dev < 0 || dev >= 64 || lds_bytes > configured[dev]) {
        if (hipFuncSetAttribute((const void *)k_net_w43t<S>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess)
            return AZ_ERR_HIP;
        if (dev >= 0 && dev < 64) configured[dev] = lds_bytes;
    }
    hipLaunchKernelGGL(k_net_w43t<S>, dim3(grid), dim3(W::NTHR), lds_bytes, st, boards, leaf_kind, expand_mask, n, prm, front, pol, val,
                       value_out, logp_out);
    return hipGetLastError() == hipSuccess ? AZ_OK : AZ_ERR_HIP;
}

} // namespace

// Launches the F(4x4,3x3) kernel; AZ_ERR_ARG for a board size it is not built for.  Arguments as az_net_eval_masked.
int az_net_w43_launch(int shape, int n, const float *d_boards, const int *d_leaf_kind, const unsigned *d_expand_mask,
                      const float *d_params, int front_layers, int policy_layers, int value_layers, float *d_value, float *d_logp,
                      hipStream_t st)
{
    switch (shape) {
    case 13:
        return launch_w43<13>(d_boards, d_leaf_kind, d_expand_mask, n, d_params, front_layers, policy_layers, value_layers, d_value,
                              d_logp, st);
    case 14:
        return launch_w43<14>(d_boards, d_leaf_kind, d_expand_mask, n, d_params, front_layers, policy_layers, value_layers, d_value,
                              d_logp, st);
    case 15:
        return launch_w43<15>(d_boards, d_leaf_kind, d_expand_mask, n, d_params, front_layers, policy_layers, value_layers, d_value,
                              d_logp, st);
    case 16:
        return launch_w43<16>(d_boards, d_leaf_kind, d_expand_mask, n, d_params, front_layers, policy_layers, value_layers, d_value,
                              d_logp, st);
    case 17:
        return launch_w43t<17>(d_boards, d_leaf_kind, d_expand_mask, n, d_params, front_layers, policy_layers, value_layers, d_value,
                               d_logp, st);
    case 18:
        return launch_w43t<18>(d_boards, d_leaf_kind, d_expand_mask, n, d_params, front_layers, policy_layers, value_layers, d_value,
                               d_logp, st);
    case 19:
        return launch_w43t<19>(d_boards, d_leaf_kind, d_expand_mask, n, d_params, front_layers, policy_layers, value_layers, d_value,
                               d_logp, st);
    case 20:
        return launch_w43t<20>(d_boards, d_leaf_kind, d_expand_mask, n, d_params, front_layers, policy_layers, value_layers, d_value,
                               d_logp, st);
    default: return AZ_ERR_ARG;
    }
}

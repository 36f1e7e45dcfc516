// Batched policy/value network evaluation for the stepped search path: the
// reference's `simplerNN` (src/model.py:9-44; called once per expansion with
// batch 1 at mcts/src/cmcts.cpp:206-231) for ALL leaf positions of a search
// step in one launch, inference mode (batch-norm folded to scale/shift).
//
// gfx950 design.  The 16-channel activations of a position stay in LDS, plane-major with a
// zero halo, and every layer runs IN PLACE; HBM sees only the input planes and the outputs.
// Two kernels:
//  * k_net_solo (boards up to 15x15): ONE wavefront per position, eight positions per CU, no
//    barrier.  From 12x12 on the 3x3 layers run in the Winograd F(2x2,3x3) form (16
//    transformed products per 2x2 output tile instead of 36 multiply-adds per output, 256
//    instead of 504 matrix instructions per layer at 15x15; see `wino_layer`), below that in
//    the direct form (`solo_mma`).
//  * k_net_eval (16x16 - 20x20, where one wavefront cannot hold a board): a team of 2-4
//    wavefronts per position, two LDS buffers, direct form, one barrier pair per layer.
// Direct form: every 3x3 convolution is an implicit GEMM on the f32-input matrix cores
// (v_mfma_f32_16x16x4_f32: exact f32 products, f32 accumulate): M = 16 pixels that are
// consecutive in row-major order over the whole board (tiles cross row ends, so 15x15 = 14
// full tiles + one pixel, which a wavefront adds on the vector ALU with the weights it already
// holds), N = the 16 output channels, K = 4 input channels of one tap.  The policy head is a
// 1x1 product on the matrix cores too (16 channels -> 9 per-tap partial planes) followed by a
// 9-term gather; the value head is summed from the registers of the last value layer.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <utility>
#include <atomic>

#include "../../include/azmcts.h"
#include "az_common.h"

#include "az_net_team.h"
#include "az_net_parts.h"

namespace {
using namespace aznet;


template <class G>
__global__ __launch_bounds__(G::NTHR) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_net_eval(const float *boards, const int *leaf_kind, int n, const float *prm,
                                                           int front, int pol, int val, float *value_out, float *logp_out)
{
    extern __shared__ float lds[];
    float *F = lds, *W = lds + G::BUF;
    float *red = lds + 2 * G::BUF;     // [16]: two reduction rows of up to 8 wavefronts
    float *logit = red + 16;           // [SIZE]
    int *todo = reinterpret_cast<int *>(logit + G::SIZE); // [ceil(n / gridDim.x)] positions of this workgroup
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6); // wave-uniform: the per-wavefront variants branch on it
    for (int i = tid; i < 2 * G::BUF; i += G::NTHR) lds[i] = 0.f; // halos stay zero from here on
    __syncthreads();
    TeamCtx<G> ctx;
    team_setup<G>(ctx, prm, front, pol, val, wave, lane);

    // Positions to evaluate: this workgroup's share of those that need an evaluation, or every
    // gridDim.x-th position when no codes are given.
    int n_mine;
    if (leaf_kind != nullptr) {
        n_mine = list_positions<G::NTHR>(leaf_kind, n, todo, reinterpret_cast<int *>(red), tid);
    } else {
        n_mine = n > (int)blockIdx.x ? (n - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x : 0;
    }
    for (int it = 0; it < n_mine; it++) {
        const int pos = leaf_kind != nullptr ? todo[it] : (int)blockIdx.x + it * (int)gridDim.x;
        team_position<G>(ctx, F, W, red, logit, boards + (size_t)pos * CIN * G::SIZE, value_out + pos,
                         logp_out + (size_t)pos * G::SIZE, tid, wave, lane);
    }
}


// ---------------------------------------------------------------------------------
// Boards up to 15x15: ONE wavefront per position (a workgroup of 64 threads), eight
// positions per CU.  Nothing is shared between wavefronts, so there is no barrier
// anywhere: the two wavefronts of a SIMD drift freely against each other and one
// computes while the other is in an epilogue or a head.  A single activation buffer
// per position (every layer runs in place: all 14 tiles of a 15x15 board are held as
// accumulators, then overwrite the input); the trunk output, which both branches
// start from, waits in registers while the value branch uses the buffer.
// The weights of the next layer replace the current ones register by register as the
// K steps retire them, so one set of weight registers serves the whole network.
template <int S> struct SoloGeo {
    using G = Geo<S>;
    static constexpr int NT = G::NT;
    static constexpr int LDS_BYTES = (G::BUF + 64) * 4; // + the zero tail and the dump cell of the Winograd layers
    static constexpr int WGS0 = (160 * 1024 - 2048) / (LDS_BYTES + 256);
    static constexpr int WGS = WGS0 > 8 ? 8 : WGS0;
};

template <int S> struct SoloMap { // per-lane addressing of the tiles, fixed for the whole launch
    int rd[Geo<S>::NT];   // A-operand pixel of tile t (+ channel-quarter plane, biased by -(ROW+1))
    // byte offset of the lane's first output cell of tile t (co*PS + pixel; bits 0..14), and for cells
    // 1..3 whether a row end (two halo cells, 8 bytes) lies before them: 8 or 0 in byte 2 (cell 1) and
    // byte 3 (cell 2), bit 15 for cell 3
    int wrpk[Geo<S>::NT];
    int crd, cwr;         // the corner pixel of a board of 16k+1 pixels
    int last_valid;       // pixels of the lane's four that exist in a partial last tile
};
template <int S> __device__ __forceinline__ void make_solo_map(int lane, SoloMap<S> &m)
{
    using G = Geo<S>;
    const int co = lane & 15, kq = lane >> 4;
#pragma unroll
    for (int t = 0; t < G::NT; t++) {
        int p = t * 16 + (lane & 15);
        p = p < G::SIZE ? p : G::SIZE - 1;
        m.rd[t] = pix_index<S>(p) + kq * G::PS - (G::ROW + 1);
        int p0 = t * 16 + kq * 4;
        p0 = p0 < G::SIZE ? p0 : G::SIZE - 1;
        const int left = S - p0 % S;
        m.wrpk[t] = ((co * G::PS + pix_index<S>(p0)) * 4) | (left <= 3 ? 1 << 15 : 0) | (left <= 1 ? 8 << 16 : 0) |
                    (left <= 2 ? 8 << 24 : 0);
        static_assert(G::BUF * 4 < (1 << 15), "cell offsets need 15 bits");
    }
    const int rest = G::SIZE - ((G::NT - 1) * 16 + kq * 4);
    m.last_valid = rest < 0 ? 0 : (rest > 4 ? 4 : rest);
    m.crd = pix_index<S>(G::SIZE - 1) + kq * G::PS - (G::ROW + 1);
    m.cwr = co * G::PS + pix_index<S>(G::SIZE - 1);
}


// Stores the lane's four cells of one tile (see SoloMap::wrpk): one address instruction per cell.
template <int S, bool PARTIAL>
__device__ __forceinline__ void store_cells(float *buf, int pk, int last_valid, const float (&v)[4])
{
    asm volatile("" : "+v"(pk)); // (not hoisted: precomputed addresses of all tiles would not fit the register file)
    const unsigned u = (unsigned)pk;
    char *a0 = reinterpret_cast<char *>(buf) + (u & 0x7fffu);
    char *a1 = a0 + ((u >> 16) & 0xffu), *a2 = a0 + (u >> 24), *a3 = a0 + ((u >> 12) & 8u);
    if (!PARTIAL || 0 < last_valid) *reinterpret_cast<float *>(a0) = v[0];
    if (!PARTIAL || 1 < last_valid) *reinterpret_cast<float *>(a1 + 4) = v[1];
    if (!PARTIAL || 2 < last_valid) *reinterpret_cast<float *>(a2 + 8) = v[2];
    if (!PARTIAL || 3 < last_valid) *reinterpret_cast<float *>(a3 + 12) = v[3];
}

// The matrix part of one convolution over all tiles: acc[t] += A(tile t) x B for every K
// step.  The A operands of step ks+1 are requested into the registers step ks has just
// consumed, one request after each matrix instruction (scheduling groups), so a request
// is 13 matrix instructions ahead of its use.  ROLL: w.b[ks] is reloaded from `nextp`
// (the next layer's block) as soon as step ks is done with it.
template <int S, int KSTEPS, bool ROLL>
__device__ __forceinline__ void solo_mma(const float *buf, LayerRegs<KSTEPS> &w, const float *nextp, const SoloMap<S> &m,
                                         const int (&off0)[5], int lane, float init, f32x4 (&acc)[Geo<S>::NT], float &cacc)
{
    using G = Geo<S>;
    constexpr int NT = G::NT;
    const int kq = lane >> 4;
    const int kqp = KSTEPS == 36 ? 0 : kq * G::PS;
    auto koff = [&](int ks) {
        if constexpr (KSTEPS == 36) {
            const int tap = ks >> 2;
            return (ks & 3) * 4 * G::PS + (tap / 3) * G::ROW + (tap % 3);
        } else if constexpr (KSTEPS == 4) { // 1x1 product of the policy head: the centre tap
            return ks * 4 * G::PS + (G::ROW + 1);
        } else {
            return off0[ks] - kqp + (G::ROW + 1);
        }
    };
#pragma unroll
    for (int t = 0; t < NT; t++) acc[t] = (f32x4){init, init, init, init};
    cacc = 0.f;
    // Issue priority: low inside the matrix loop, high everywhere else.  The other wavefront
    // of the SIMD is usually in its own matrix loop; its long matrix instructions lose
    // nothing when this wavefront's short vector phases (epilogues, heads) go first, and
    // those phases take half the time.
    __builtin_amdgcn_s_setprio(0);
    // (the input layer's offsets vary per lane: its 5 x NT addresses are formed when needed,
    // from a base the compiler cannot trace, instead of living in registers all along)
    int rd[NT], crd = m.crd;
#pragma unroll
    for (int t = 0; t < NT; t++) {
        rd[t] = m.rd[t];
        if constexpr (KSTEPS == 5) asm volatile("" : "+v"(rd[t]));
    }
    if constexpr (KSTEPS == 5) asm volatile("" : "+v"(crd));
    float a[NT], ca = 0.f;
#pragma unroll
    for (int t = 0; t < NT; t++) a[t] = buf[rd[t] + koff(0)];
    if constexpr (G::CORNER) ca = buf[crd + koff(0)];
    const float *np = nextp + lane;
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ks++) {
        __builtin_amdgcn_sched_barrier(0);
        const float b = w.b[ks];
#pragma unroll
        for (int t = 0; t < NT; t++) {
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], b, acc[t], 0, 0, 0);
            if (ks + 1 < KSTEPS) a[t] = buf[rd[t] + koff(ks + 1)];
        }
        if constexpr (G::CORNER) {
            cacc = fmaf(ca, b, cacc);
            if (ks + 1 < KSTEPS) ca = buf[crd + koff(ks + 1)];
        }
        if constexpr (ROLL) w.b[ks] = np[ks * 64];
        if (ks + 1 < KSTEPS) {
#pragma unroll
            for (int t = 0; t < NT; t++) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); // one matrix instruction
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); // one LDS read
            }
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(3);
}

// ReLU of the accumulators (they started from the folded batch-norm shift, the scale is in
// the weights).  WRITE: back into the buffer; KEEP: also into the private copy of the trunk
// output; SUM: nothing is stored, the lane's outputs are added up for the value head.
enum { EP_WRITE = 1, EP_KEEP = 2, EP_SUM = 4 };
template <int S, int FLAGS>
__device__ __forceinline__ void solo_epilogue(float *buf, const f32x4 (&acc)[Geo<S>::NT], float cacc, float sh,
                                              const SoloMap<S> &m, int lane, float *fc, int fz, float &vsum)
{
    using G = Geo<S>;
    const int kq = lane >> 4;
#pragma unroll
    for (int t = 0; t < G::NT; t++) {
        constexpr bool dummy = false;
        (void)dummy;
        const bool partial = G::SIZE % 16 != 0 && !G::CORNER && t == G::NT - 1; // compile-time per t after unrolling
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; r++) v[r] = __builtin_amdgcn_fmed3f(acc[t][r], 0.f, INFINITY); // ReLU in one instruction
        if constexpr ((FLAGS & EP_WRITE) != 0) {
            if (partial) store_cells<S, true>(buf, m.wrpk[t], m.last_valid, v);
            else store_cells<S, false>(buf, m.wrpk[t], 4, v);
        }
        if constexpr ((FLAGS & EP_KEEP) != 0) {
#pragma unroll
            for (int r = 0; r < 4; r++) fc[fz + t * 4 + r] = v[r];
        }
        if constexpr ((FLAGS & EP_SUM) != 0) {
#pragma unroll
            for (int r = 0; r < 4; r++) vsum += (!partial || r < m.last_valid) ? v[r] : 0.f;
        }
    }
    if constexpr (G::CORNER) {
        const float v = __builtin_amdgcn_fmed3f(sum_over_kq(cacc) + sh, 0.f, INFINITY);
        if constexpr ((FLAGS & EP_KEEP) != 0) fc[fz + G::NT * 4] = v;
        if (kq == 0) {
            if constexpr ((FLAGS & EP_WRITE) != 0) buf[m.cwr] = v;
            if constexpr ((FLAGS & EP_SUM) != 0) vsum += v;
        }
    }
}
// the trunk output back from its private copy
template <int S> __device__ __forceinline__ void solo_restore(float *buf, const SoloMap<S> &m, int lane, const float *fc, int fz)
{
    using G = Geo<S>;
#pragma unroll
    for (int t = 0; t < G::NT; t++) {
        const bool partial = G::SIZE % 16 != 0 && !G::CORNER && t == G::NT - 1;
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; r++) v[r] = fc[fz + t * 4 + r];
        if (partial) store_cells<S, true>(buf, m.wrpk[t], m.last_valid, v);
        else store_cells<S, false>(buf, m.wrpk[t], 4, v);
    }
    if (G::CORNER && (lane >> 4) == 0) buf[m.cwr] = fc[fz + G::NT * 4];
}


// ---------------------------------------------------------------------------------
// Winograd F(2x2, 3x3) form of the 3x3 layers for the one-wavefront kernel: a 2x2 tile of
// outputs comes from the 4x4 input patch around it through 16 element-wise products in a
// transformed domain instead of 36 multiply-adds per output, so a 16 -> 16 channel layer
// needs 16 (transformed positions) x 4 (K steps) matrix instructions per 16 tiles: 256 for
// a 15x15 board against 504 in the direct form.
//   Tile grid: 8 tiles per tile row for every board (tiles past the board's last column are
//   dummies), two tile rows per block of 16 = the M side of a matrix instruction, lane
//   m = 2*tx + (ty & 1): horizontally adjacent tiles sit two lanes apart.
//   K = 4 input channels, N = the 16 output channels.
//   Lane (m, kq) loads the two left columns of the 4x4 patch of ITS tile and channel
//   4*ks + kq and forms their column transforms; the two right columns are the left columns
//   of the tile to the right, whose transforms it takes from two lanes up (DPP, zero behind
//   the eighth tile: halo).  24 additions give V = Bt d B, the 16 results ARE the lane's A
//   operands of the 16 matrix instructions of this K step; the B operands are the
//   host-transformed weights (U = G g Gt, batch-norm scale folded in), 64 registers per layer.
//   After the last K step a lane holds, for each of four tiles, the 16 transformed sums of
//   its output channel; the inverse transform (24 additions per tile) gives the 2x2 outputs.
//   The batch-norm shift enters through the initial value of the four accumulators that
//   each feed exactly one output.
// The layer runs in place: the outputs of a block are written one block late, after every
// patch of the following block has been requested (a block's patches reach one row into its
// neighbours).  Patches of the last tile row / column read past the halo: that is the next
// row's halo, the next plane's top halo or the zero tail behind the last plane.
// Matrix instructions are issued in bursts of 16 with the vector work of a K step behind
// them at raised priority: vector instructions placed between matrix instructions cost about
// 2 cycles of matrix time each, after a burst the other wavefront's burst hides them.


template <int S> struct WinoGeo {
    static constexpr int TH = (S + 1) / 2;  // tile rows = real tiles per tile row
    static constexpr int NB = (TH + 1) / 2; // blocks of two tile rows
    static_assert(TH <= 8, "eight tiles per tile row");
};
constexpr int WINO_TAIL = 32;             // zero floats behind the last plane (a patch may reach ROW + 1 floats past it)
constexpr int WINO_DUMP = WINO_TAIL;      // + ROW + 2 cells that take the stores of dummy tiles

template <int S> __device__ __forceinline__ void load_wino(const float *prm, int lane, float (&u)[64])
{
#pragma unroll
    for (int c = 0; c < 4; c++) {
        const float *p = prm + c * 16 * 64 + lane;
        asm volatile("" : "+v"(p));
#pragma unroll
        for (int i = 0; i < 16; i++) u[c * 16 + i] = p[i * 64];
    }
}

// The 2x2 outputs o[q][r] (q = 2*row + column inside the tile, r = the lane's r-th tile of block
// mb: tile column 2*kq + (r >> 1), tile row 2*mb + (r & 1)) of this lane's output channel.
// Cells outside the board (last row / column of an odd board) are replaced by zero and stored all
// the same, they land on the halo; dummy tiles store into a dump area.  `cob` = float offset of the
// channel's plane + one halo row and column.  wr: store into the buffer; keep: also into the
// private copy; sum: the lane's sum of the stored values is added to lsum.
template <int S, bool wr, bool keep, bool sum>
__device__ __forceinline__ void wino_store(float *buf, int mb, int cob, int kq, float (&o)[4][4], float *fc, int fz, float &lsum)
{
    using G = Geo<S>;
    using W = WinoGeo<S>;
    constexpr bool ODD = (S & 1) != 0;
    const int base = cob + 4 * kq; // tile column 2*kq
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int ty = 2 * mb + (r & 1); // (a constant)
        if (ty >= W::TH) {              // a dummy tile row
#pragma unroll
            for (int q = 0; q < 4; q++) o[q][r] = 0.f;
            continue;
        }
        const int tx = 2 * kq + (r >> 1);
        const bool below = !ODD || ty != W::TH - 1; // (a constant)
        int at = base + ty * 2 * G::ROW + (r >> 1) * 2;
        if (W::TH < 8 && 6 + (r >> 1) >= W::TH) { // this tile can be a dummy column (kq <= 3)
            const bool dummy = tx >= W::TH;
            at = dummy ? G::BUF + WINO_DUMP : at;
#pragma unroll
            for (int q = 0; q < 4; q++) o[q][r] = dummy ? 0.f : o[q][r];
        }
        if (ODD && 6 + (r >> 1) >= W::TH - 1) { // this tile can be the last of its row
            const bool last = tx == W::TH - 1;
            o[1][r] = last ? 0.f : o[1][r];
            o[3][r] = last ? 0.f : o[3][r];
        }
        if (!below) o[2][r] = o[3][r] = 0.f;
        if constexpr (wr) {
            float *c = buf + at;
            c[0] = o[0][r];
            c[1] = o[1][r];
            if (below) { // (the row below the board is halo and stays zero)
                c[G::ROW] = o[2][r];
                c[G::ROW + 1] = o[3][r];
            }
        }
        if constexpr (sum) lsum += (o[0][r] + o[1][r]) + (o[2][r] + o[3][r]);
    }
    if constexpr (keep) {
#pragma unroll
        for (int q = 0; q < 4; q++)
#pragma unroll
            for (int r = 0; r < 4; r++) fc[fz + mb * 16 + q * 4 + r] = o[q][r];
    }
}

// value of `v` two lanes up inside the row of 16 lanes, zero for the last two lanes
__device__ __forceinline__ float from_lane_plus_2(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x102 /* row_shl:2 */, 0xf, 0xf, true));
}

#ifndef AZ_WINO_PRIO
#define AZ_WINO_PRIO 3
#endif
// KS = K steps: 4 for 16 input channels, 1 for the two input planes (K indices 2..3 carry zero weights and
// read planes 0..1 again).  ROLL: the next layer's weights replace u[] as the last block retires them.
template <int S, int KS, bool ROLL, bool wr, bool keep, bool sum>
__device__ __forceinline__ void wino_layer(float *buf, float (&u)[16 * KS], const float *nextp, float sh, int lane, float *fc,
                                           int fz, float &lsum)
{
    using G = Geo<S>;
    using W = WinoGeo<S>;
    constexpr int NB = W::NB, NC = NB * KS;
    int l = lane;
    asm volatile("" : "+v"(l)); // (addresses are formed per layer instead of living in registers across the position loop)
    const int kq = l >> 4, co = l & 15;
    const int kqp = KS == 4 ? kq : (kq & 1); // plane of this lane's K index in K step 0
    const int cob = co * G::PS + G::ROW + 1;
    // top-left cell of the patch of this lane's tile in block 0 (block mb: + 4*mb rows; a dummy tile row reads the last real one again)
    const int pb0 = kqp * G::PS + 2 * (co & 1) * G::ROW + 2 * (co >> 1);
    auto load_patch = [&](int c, float (&d)[8]) {
        const int mb = c / KS;
        const int rows = (2 * mb + 1 < W::TH) ? 4 * mb * G::ROW : -1; // (only the lanes of the second tile row can be dummies)
        int po = pb0 + (c % KS) * 4 * G::PS + (rows >= 0 ? rows : (4 * mb - 2 * (co & 1)) * G::ROW);
        asm("" : "+v"(po)); // one address per patch; the cells are immediate offsets (not volatile: a volatile
                            // statement orders every memory access around it and the patches could not be requested early)
        const float *p = buf + po;
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 2; j++) d[i * 2 + j] = p[i * G::ROW + j];
    };
    auto transform = [&](const float (&d)[8], float (&a)[16]) {
        float t[8];
#pragma unroll
        for (int j = 0; j < 2; j++) {
            t[j] = d[j] - d[4 + j];
            t[2 + j] = d[2 + j] + d[4 + j];
            t[4 + j] = d[4 + j] - d[2 + j];
            t[6 + j] = d[2 + j] - d[6 + j];
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const float t2 = from_lane_plus_2(t[i * 2]), t3 = from_lane_plus_2(t[i * 2 + 1]);
            a[i * 4] = t[i * 2] - t2;
            a[i * 4 + 1] = t[i * 2 + 1] + t2;
            a[i * 4 + 2] = t2 - t[i * 2 + 1];
            a[i * 4 + 3] = t[i * 2 + 1] - t3;
        }
    };
    float a[16], d[8];
    load_patch(0, d);
    transform(d, a);
    if (NC > 1) load_patch(1, d);
    f32x4 acc[16];
    float pend[4][4];
    static_for<NC>([&](auto cc) {
        constexpr int c = decltype(cc)::value;
        constexpr int mb = c / KS, ks = c % KS;
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(0);
        if (ks == 0) {
#pragma unroll
            for (int x = 0; x < 16; x++) {
                const float v = (x == 0 || x == 15) ? sh : ((x == 3 || x == 12) ? -sh : 0.f);
                acc[x] = (f32x4){v, v, v, v};
            }
        }
        // (column by column: the inverse transform's column sums can start while the last products run)
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int x = i * 4 + j;
                acc[x] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[x], u[x * KS + ks], acc[x], 0, 0, 0);
            }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(AZ_WINO_PRIO);
        float an[16];
        if (c + 1 < NC) transform(d, an);
        if (c + 2 < NC) load_patch(c + 2, d);
        if (ROLL && mb == NB - 1) { // the next layer's weights replace this layer's as their last use retires
            // (uniform base + 32-bit byte offset: one offset register for all 16 loads)
            unsigned o = (unsigned)(l + ks * 64) * 4u;
            asm("" : "+v"(o));
#pragma unroll
            for (int x = 0; x < 16; x++)
                u[x * KS + ks] = *reinterpret_cast<const float *>(reinterpret_cast<const char *>(nextp + x * KS * 64) + o);
        }
        // (the previous block's outputs: every patch of this block has been requested by now)
        if (ks == (KS == 4 ? 2 : 0) && mb > 0) wino_store<S, wr, keep, sum>(buf, mb - 1, cob, kq, pend, fc, fz, lsum);
        if (ks == KS - 1) {
            // (element by element: packed additions cost more issue time next to matrix instructions than two plain ones)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                float s[8];
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    s[j] = acc[j][r] + acc[4 + j][r] + acc[8 + j][r];
                    s[4 + j] = acc[4 + j][r] - acc[8 + j][r] - acc[12 + j][r];
                }
                pend[0][r] = __builtin_amdgcn_fmed3f(s[0] + s[1] + s[2], 0.f, INFINITY);
                pend[1][r] = __builtin_amdgcn_fmed3f(s[1] - s[2] - s[3], 0.f, INFINITY);
                pend[2][r] = __builtin_amdgcn_fmed3f(s[4] + s[5] + s[6], 0.f, INFINITY);
                pend[3][r] = __builtin_amdgcn_fmed3f(s[5] - s[6] - s[7], 0.f, INFINITY);
            }
            if (mb == NB - 1) wino_store<S, wr, keep, sum>(buf, mb, cob, kq, pend, fc, fz, lsum);
        }
#pragma unroll
        for (int x = 0; x < 16; x++) a[x] = an[x];
    });
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(3);
}
// the trunk output back from its private copy
template <int S> __device__ __forceinline__ void wino_restore(float *buf, int lane, float *fc, int fz)
{
    using G = Geo<S>;
    using W = WinoGeo<S>;
    int l = lane;
    asm volatile("" : "+v"(l));
    float unused = 0.f;
#pragma unroll
    for (int mb = 0; mb < W::NB; mb++) {
        float o[4][4];
#pragma unroll
        for (int q = 0; q < 4; q++)
#pragma unroll
            for (int r = 0; r < 4; r++) o[q][r] = fc[fz + mb * 16 + q * 4 + r];
        wino_store<S, true, false, false>(buf, mb, (l & 15) * G::PS + G::ROW + 1, l >> 4, o, fc, fz, unused);
    }
}


// Policy head, first half: the nine per-tap partial planes (a 1x1 product of the 16 channels
// with the head's 3x3 weights, tap = output column of the matrix instruction) written over
// planes 0..8.  M = the cells of one board row (lanes past the row read its right halo cell:
// zero, so what they store is zero and lands on that halo cell again), so every address is
// one lane value plus a constant.  All reads precede all writes.
template <int S> __device__ __forceinline__ void policy_taps_rows(float *buf, const float (&wpb)[4], int lane)
{
    using G = Geo<S>;
    static_assert(S <= 15, "one matrix tile per board row");
    int l = lane;
    asm volatile("" : "+v"(l));
    const int co = l & 15, kq = l >> 4;
    const float *rd = buf + kq * G::PS + G::ROW + (co < S ? co : S) + 1;
    f32x4 acc[S];
    float a[S];
#pragma unroll
    for (int t = 0; t < S; t++) {
        acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        a[t] = rd[t * G::ROW];
    }
    __builtin_amdgcn_s_setprio(0);
#pragma unroll
    for (int ks = 0; ks < 4; ks++) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < S; t++) {
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], wpb[ks], acc[t], 0, 0, 0);
            if (ks + 1 < 4) a[t] = rd[(ks + 1) * 4 * G::PS + t * G::ROW];
        }
        if (ks + 1 < 4) {
#pragma unroll
            for (int t = 0; t < S; t++) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(3);
    if (co < 9) {
        float *wr[4];
#pragma unroll
        for (int r = 0; r < 4; r++) wr[r] = buf + co * G::PS + G::ROW + (4 * kq + r < S ? 4 * kq + r : S) + 1;
#pragma unroll
        for (int t = 0; t < S; t++)
#pragma unroll
            for (int r = 0; r < 4; r++) wr[r][t * G::ROW] = acc[t][r];
    }
}

template <int S, bool WINO>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_net_solo(
    const float *boards, const int *leaf_kind, const unsigned *expand_mask, int n, const float *prm, int front, int pol, int val,
    float *value_out, float *logp_out)
{
    using G = Geo<S>;
    constexpr int NT = G::NT;
    extern __shared__ float lds[];
    float *buf = lds;
    int *todo = reinterpret_cast<int *>(lds + G::BUF + 64); // [ceil(n / gridDim.x)] positions of this workgroup
    const int lane = threadIdx.x;
    const int co = lane & 15, kq = lane >> 4;
    for (int i = lane; i < G::BUF + 64; i += 64) buf[i] = 0.f; // halos (and the tail) stay zero from here on
    const int nl = front - 1 + pol + val;
    const float *p_layers = prm + FIRST_FLOATS;
    const float *p_policy = p_layers + (size_t)nl * LAYER_FLOATS;
    const float *p_value = p_policy + POLICY_FLOATS;
    const float *p_wino = p_value + VALUE_FLOATS;
    // execution order: trunk, value branch, policy branch (parameter order: trunk, policy, value)
    auto layer_params = [&](int e) {
        const int idx = e < front - 1 ? e : (e < front - 1 + val ? e + pol : e - val);
        return WINO ? p_wino + (size_t)idx * WINO_FLOATS : p_layers + (size_t)idx * LAYER_FLOATS;
    };
    LayerRegs<5> w0;
    if constexpr (!WINO) load_layer<5>(prm, lane, w0);
    int off0[5];
#pragma unroll
    for (int ks = 0; ks < 5; ks++) {
        const int kk = ks * 4 + kq;
        const int tap = kk < 18 ? kk >> 1 : 0;
        off0[ks] = (kk & 1) * G::PS + (tap / 3 - 1) * G::ROW + (tap % 3 - 1);
    }
    SoloMap<S> m;
    if constexpr (!WINO) make_solo_map<S>(lane, m);
    LayerRegs<4> wp; // policy head as a B operand: B[k = ci][n = tap]
#pragma unroll
    for (int ks = 0; ks < 4; ks++) wp.b[ks] = co < 9 ? p_policy[co * CH + ks * 4 + kq] : 0.f;
    wp.sc = wp.sh = 0.f;
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
    __syncthreads(); // (a single wavefront: orders the LDS traffic, costs nothing)

#ifdef AZ_NET_STAMPS
    unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_t = 0, st_0 = __builtin_readcyclecounter();
#define NST(i) do { unsigned long long t_ = __builtin_readcyclecounter(); st_acc[i] += t_ - st_t; st_t = t_; } while (0)
#else
#define NST(i) do { } while (0)
#endif
    f32x4 acc[NT];
    float cacc = 0.f;
    // The trunk output, which both branches start from, is parked in private memory while the
    // value branch owns the buffer (one store and one load per value and position; the
    // register file has no room for a second set of NT tiles).  Indexed through a value the
    // compiler cannot see through, so that the array really lives in memory.
    float fc[WINO ? WinoGeo<S>::NB * 16 : NT * 4 + 1];
    int fz;
    asm volatile("s_mov_b32 %0, 0" : "=s"(fz));
    // pixels this lane loads, gathers and stores: lane, lane + 64, ...
    constexpr int PPL = (G::SIZE + 63) / 64;
    float in0[PPL], in1[PPL]; // the next position's input planes, requested one position ahead
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
    // the weights of the first 16-channel layer: loaded here once, later rolled in by the last layer
    LayerRegs<WINO ? 1 : 36> wq;
    float u[WINO ? 64 : 1], ush = 0.f;
    float u0[WINO ? 16 : 1], u0sh = 0.f; // the input layer in the Winograd domain
    if constexpr (WINO) {
        const float *p0 = p_wino + (size_t)nl * WINO_FLOATS;
#pragma unroll
        for (int x = 0; x < 16; x++) u0[x] = p0[x * 64 + lane];
        u0sh = p0[16 * 64 + co];
        if (nl > 0) {
            load_wino<S>(layer_params(0), lane, u);
            ush = layer_params(0)[64 * 64 + co];
        }
    } else {
        if (nl > 0) load_layer<36>(layer_params(0), lane, wq);
    }
    for (int it = 0; it < n_mine; it++) {
        const int pos = listed ? todo[it] : (int)blockIdx.x + it * (int)gridDim.x;
#ifdef AZ_NET_STAMPS
        st_t = __builtin_readcyclecounter();
#endif
        {   // the two input planes (requested while the previous position was in its heads) into planes 0..1
            int l0 = lane;
            asm volatile("" : "+v"(l0));
#pragma unroll
            for (int i = 0; i < PPL; i++) {
                const int p = l0 + 64 * i;
                if (p < G::SIZE) {
                    const int ctr = pix_index<S>(p);
                    buf[ctr] = in0[i];
                    buf[G::PS + ctr] = in1[i];
                }
            }
        }
        __syncthreads();
        float vsum = 0.f;
        NST(0);
        // input layer; it is also the trunk's last layer when front == 1
        if constexpr (WINO) {
            float lsum = 0.f;
            if (front > 1) wino_layer<S, 1, false, true, false, false>(buf, u0, nullptr, u0sh, lane, fc, fz, lsum);
            else if (val > 0) wino_layer<S, 1, false, true, true, false>(buf, u0, nullptr, u0sh, lane, fc, fz, lsum);
            else wino_layer<S, 1, false, true, false, true>(buf, u0, nullptr, u0sh, lane, fc, fz, lsum);
            if (front == 1 && val == 0) vsum = lsum;
            __syncthreads();
            NST(6);
        } else {
            solo_mma<S, 5, false>(buf, w0, prm, m, off0, lane, w0.sh, acc, cacc);
            NST(6);
            if (front > 1) solo_epilogue<S, EP_WRITE>(buf, acc, cacc, w0.sh, m, lane, fc, fz, vsum);
            else if (val > 0) solo_epilogue<S, EP_WRITE | EP_KEEP>(buf, acc, cacc, w0.sh, m, lane, fc, fz, vsum);
            else solo_epilogue<S, EP_WRITE | EP_SUM>(buf, acc, cacc, w0.sh, m, lane, fc, fz, vsum);
            __syncthreads();
            NST(2);
        }
        for (int e = 0; e < nl; e++) {
            const bool last_trunk = e == front - 2;
            const bool last_value = val > 0 && e == front - 1 + val - 1;
            const bool first_policy = pol > 0 && e == front - 1 + val;
            const float *nextp = layer_params(e + 1 < nl ? e + 1 : 0); // (the last layer fetches the next position's first)
            if constexpr (WINO) {
                if (first_policy && val > 0) { // the trunk output comes back from its private copy
                    wino_restore<S>(buf, lane, fc, fz);
                    __syncthreads();
                }
                const float sh = ush;
                ush = nextp[64 * 64 + co]; // (requested a layer ahead: its latency runs under this layer)
                NST(4);
                float lsum = 0.f;
                if (last_value) wino_layer<S, 4, true, false, false, true>(buf, u, nextp, sh, lane, fc, fz, lsum);
                else if (last_trunk && val > 0) wino_layer<S, 4, true, true, true, false>(buf, u, nextp, sh, lane, fc, fz, lsum);
                else if (last_trunk) wino_layer<S, 4, true, true, false, true>(buf, u, nextp, sh, lane, fc, fz, lsum);
                else wino_layer<S, 4, true, true, false, false>(buf, u, nextp, sh, lane, fc, fz, lsum);
                if (last_value || (last_trunk && val == 0)) vsum = lsum;
                __syncthreads();
                NST(1);
            } else {
                if (first_policy && val > 0) { // the trunk output comes back from its private copy
                    solo_restore<S>(buf, m, lane, fc, fz);
                    __syncthreads();
                }
                const float sh = wq.sh;
                NST(4);
                solo_mma<S, 36, true>(buf, wq, nextp, m, off0, lane, sh, acc, cacc);
                NST(1);
                wq.sh = nextp[36 * 64 + 16 + co];
                if (last_value) solo_epilogue<S, EP_SUM>(buf, acc, cacc, sh, m, lane, fc, fz, vsum);
                else if (last_trunk && val > 0) solo_epilogue<S, EP_WRITE | EP_KEEP>(buf, acc, cacc, sh, m, lane, fc, fz, vsum);
                else if (last_trunk) solo_epilogue<S, EP_WRITE | EP_SUM>(buf, acc, cacc, sh, m, lane, fc, fz, vsum);
                else solo_epilogue<S, EP_WRITE>(buf, acc, cacc, sh, m, lane, fc, fz, vsum);
                __syncthreads();
                NST(2);
            }
        }
        // policy head input: the policy branch's output, or the trunk output when there is no such branch
        if (pol == 0 && val > 0) {
            if constexpr (WINO) wino_restore<S>(buf, lane, fc, fz);
            else solo_restore<S>(buf, m, lane, fc, fz);
            __syncthreads();
        }
        if constexpr (WINO) {
            // The input layer's weights are fetched again for every position (from an offset the compiler
            // cannot see through), so that they do not occupy registers during the 16-channel layers.
            unsigned o = (unsigned)lane * 4u;
            asm volatile("" : "+v"(o));
            const char *p0 = reinterpret_cast<const char *>(p_wino + (size_t)nl * WINO_FLOATS);
#pragma unroll
            for (int x = 0; x < 16; x++) u0[x] = *reinterpret_cast<const float *>(p0 + x * 256 + o);
        }
        // ---- value head: board average of the 1x1 convolution + bias, tanh
        float vpart = wv * vsum;
        if constexpr (WINO) {
            vpart = wave_reduce<false>(vpart);
        } else {
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) vpart += __shfl_xor(vpart, o);
        }
        if (lane == 0) value_out[pos] = tanhf(vpart / (float)G::SIZE + v_bias);
        if (it + 1 < n_mine) request_board(listed ? todo[it + 1] : (int)blockIdx.x + (it + 1) * (int)gridDim.x);
        // ---- policy head: per-tap partial planes on the matrix cores, written over the input
        if constexpr (WINO) policy_taps_rows<S>(buf, wp.b, lane);
        else solo_mma<S, 4, false>(buf, wp, prm, m, off0, lane, 0.f, acc, cacc);
        if (!WINO && co < 9) {
#pragma unroll
            for (int t = 0; t < NT; t++) {
                const bool partial = G::SIZE % 16 != 0 && !G::CORNER && t == NT - 1;
                const float v[4] = {acc[t][0], acc[t][1], acc[t][2], acc[t][3]};
                if (partial) store_cells<S, true>(buf, m.wrpk[t], m.last_valid, v);
                else store_cells<S, false>(buf, m.wrpk[t], 4, v);
            }
        }
        if constexpr (G::CORNER && !WINO) {
            const float v = sum_over_kq(cacc);
            if (kq == 0 && co < 9) buf[m.cwr] = v;
        }
        __syncthreads();
        // gather of the nine planes + bias, log-softmax over the board
        float logit[PPL];
        float lmax = -INFINITY;
        int l1 = lane;
        asm volatile("" : "+v"(l1));
#pragma unroll
        for (int i = 0; i < PPL; i++) {
            const int p = l1 + 64 * i;
            const int ctr = pix_index<S>(p < G::SIZE ? p : 0);
            float s = p_bias;
#pragma unroll
            for (int tap = 0; tap < 9; tap++) s += buf[tap * G::PS + ctr + (tap / 3 - 1) * G::ROW + (tap % 3 - 1)];
            logit[i] = p < G::SIZE ? s : -INFINITY;
            lmax = fmaxf(lmax, logit[i]);
        }
        if constexpr (WINO) {
            lmax = wave_reduce<true>(lmax);
        } else {
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) lmax = fmaxf(lmax, __shfl_xor(lmax, o));
        }
        float lsum = 0.f;
#pragma unroll
        for (int i = 0; i < PPL; i++) lsum += lane + 64 * i < G::SIZE ? expf(logit[i] - lmax) : 0.f;
        if constexpr (WINO) {
            lsum = wave_reduce<false>(lsum);
        } else {
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) lsum += __shfl_xor(lsum, o);
        }
        const float lse = lmax + logf(lsum);
#pragma unroll
        for (int i = 0; i < PPL; i++)
            if (lane + 64 * i < G::SIZE) logp_out[(size_t)pos * G::SIZE + lane + 64 * i] = logit[i] - lse;
        __syncthreads(); // the gather is over before the next position's input lands in the buffer
        NST(3);
        // planes 2..15 keep leftovers (partial planes, old activations): the input layer reads planes 0..1 only
    }
#ifdef AZ_NET_STAMPS
    if (lane == 0 && n_mine > 0 && leaf_kind == nullptr) { // diagnostic build: cycle sums over the log-priors of the first position
        st_acc[5] = __builtin_readcyclecounter() - st_0;
        st_acc[7] = (unsigned long long)n_mine;
        unsigned long long *o = reinterpret_cast<unsigned long long *>(logp_out + (size_t)blockIdx.x * G::SIZE);
        for (int q = 0; q < 8; q++) o[q] = st_acc[q];
    }
#endif
}

// ---------------------------------------------------------------------------------
// Boards of 16x16 - 20x20 in the Winograd form: a team of four wavefronts per position (the
// activations of such a board fill 21 - 31 KB, so a CU holds two positions with their two
// buffers and the eight wavefronts come from teams).  The tile grid is the board's own
// (ceil(S/2) tiles per row, row-major, blocks of 16), wavefront w takes blocks w and w + 4;
// every lane loads the whole 4x4 patch of its tile and channel.  A layer that runs in place
// keeps the outputs of its blocks in registers until every wavefront of the team has read its
// patches (one barrier), then writes them.
template <int S> struct WinoTeamGeo {
    using G = Geo<S>;
    static constexpr int TW = (S + 1) / 2, TILES = TW * TW, NB = (TILES + 15) / 16;
    static constexpr int BPW = (NB + G::NW - 1) / G::NW; // blocks a wavefront takes at most
    static constexpr unsigned DIV = 65536 / TW + 1;       // T / TW == (T * DIV) >> 16 for T < 128
    static_assert(G::NW == 4 && BPW <= 2 && NB >= G::NW, "four wavefronts, one or two blocks each");
    static constexpr int TAIL = 32; // zero floats behind the second buffer (a patch may reach ROW + 1 floats past a plane)
};

// returns the lane's sum of its outputs (MODE_VALUE_SUM; nothing is written then)
template <int S, int KS, bool ROLL, int MODE, bool INPLACE = false>
__device__ __forceinline__ float wino_team_layer(const float *src, float *dst, float *dump, float (&u)[16 * KS], const float *nextp,
                                                 float sh, int wave, int lane)
{
    using G = Geo<S>;
    using W = WinoTeamGeo<S>;
    constexpr bool ODD = (S & 1) != 0;
    int l = lane;
    asm volatile("" : "+v"(l)); // (addresses are formed per layer instead of living in registers across the position loop)
    const int kq = l >> 4, co = l & 15;
    const int kqp = KS == 4 ? kq : (kq & 1);
    const int cob = co * G::PS + G::ROW + 1;
    const bool two = W::BPW == 2 && wave + G::NW < W::NB; // (wave-uniform) this wavefront has a second block
    constexpr bool inplace = MODE == MODE_WRITE && INPLACE; // (src == dst)
    float pend[W::BPW][4][4];
    float lsum = 0.f;
    // the outputs of block K of this wavefront: into dst, or added up
    auto emit = [&](auto kk) {
        constexpr int K = decltype(kk)::value;
        const int mb = wave + K * G::NW;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const unsigned T = (unsigned)(mb * 16 + r) + 4u * (unsigned)kq;
            const unsigned ty = (T * W::DIV) >> 16, tx = T - ty * W::TW;
            const bool valid = W::TILES % 16 == 0 || T < (unsigned)W::TILES;
            const bool right = !ODD || tx != W::TW - 1, below = !ODD || ty != W::TW - 1;
            const float o0 = valid ? pend[K][0][r] : 0.f, o1 = valid && right ? pend[K][1][r] : 0.f;
            const float o2 = valid && below ? pend[K][2][r] : 0.f, o3 = valid && right && below ? pend[K][3][r] : 0.f;
            if constexpr (MODE == MODE_WRITE) {
                // (cells outside an odd board are halo cells: they receive zero; a dummy tile stores into the dump area)
                float *c = valid ? dst + cob + (int)(ty * (2 * G::ROW) + tx * 2) : dump;
                c[0] = o0;
                c[1] = o1;
                c[G::ROW] = o2;
                c[G::ROW + 1] = o3;
            } else {
                lsum += (o0 + o1) + (o2 + o3);
            }
        }
    };
    static_for<W::BPW>([&](auto kk) {
        constexpr int K = decltype(kk)::value;
        if (K == 1 && !two) return;
        const bool last = K == 1 || !two; // this wavefront's last block of the layer
        const int mb = wave + K * G::NW;
        unsigned T = (unsigned)(mb * 16 + co);
        T = T < (unsigned)W::TILES ? T : (unsigned)W::TILES - 1; // (a dummy tile reads the last tile again)
        const unsigned pty = (T * W::DIV) >> 16, ptx = T - pty * W::TW;
        const int pb = (int)(kqp * G::PS + 2 * pty * G::ROW + 2 * ptx);
        auto load_patch = [&](int ks, float (&d)[16]) {
            int po = pb + ks * 4 * G::PS;
            asm("" : "+v"(po));
            const float *p = src + po;
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 4; j++) d[i * 4 + j] = p[i * G::ROW + j];
        };
        auto transform = [&](const float (&d)[16], float (&a)[16]) {
            float t[16];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                t[j] = d[j] - d[8 + j];
                t[4 + j] = d[4 + j] + d[8 + j];
                t[8 + j] = d[8 + j] - d[4 + j];
                t[12 + j] = d[4 + j] - d[12 + j];
            }
#pragma unroll
            for (int i = 0; i < 4; i++) {
                a[i * 4] = t[i * 4] - t[i * 4 + 2];
                a[i * 4 + 1] = t[i * 4 + 1] + t[i * 4 + 2];
                a[i * 4 + 2] = t[i * 4 + 2] - t[i * 4 + 1];
                a[i * 4 + 3] = t[i * 4 + 1] - t[i * 4 + 3];
            }
        };
        float a[16], d[16];
        load_patch(0, d);
        transform(d, a);
        if (KS > 1) load_patch(1, d);
        f32x4 acc[16];
        static_for<KS>([&](auto kss) {
            constexpr int ks = decltype(kss)::value;
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(0);
            if (ks == 0) {
#pragma unroll
                for (int x = 0; x < 16; x++) {
                    const float v = (x == 0 || x == 15) ? sh : ((x == 3 || x == 12) ? -sh : 0.f);
                    acc[x] = (f32x4){v, v, v, v};
                }
            }
#pragma unroll
            for (int j = 0; j < 4; j++)
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const int x = i * 4 + j;
                    acc[x] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[x], u[x * KS + ks], acc[x], 0, 0, 0);
                }
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(AZ_WINO_PRIO);
            float an[16];
            if (ks + 1 < KS) transform(d, an);
            if (ks + 2 < KS) load_patch(ks + 2, d);
            if (ROLL && last) { // the next layer's weights replace this layer's as their last use retires
                unsigned o = (unsigned)(l + ks * 64) * 4u;
                asm("" : "+v"(o));
#pragma unroll
                for (int x = 0; x < 16; x++)
                    u[x * KS + ks] = *reinterpret_cast<const float *>(reinterpret_cast<const char *>(nextp + x * KS * 64) + o);
            }
            if (ks == KS - 1) {
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    float s[8];
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        s[j] = acc[j][r] + acc[4 + j][r] + acc[8 + j][r];
                        s[4 + j] = acc[4 + j][r] - acc[8 + j][r] - acc[12 + j][r];
                    }
                    pend[K][0][r] = __builtin_amdgcn_fmed3f(s[0] + s[1] + s[2], 0.f, INFINITY);
                    pend[K][1][r] = __builtin_amdgcn_fmed3f(s[1] - s[2] - s[3], 0.f, INFINITY);
                    pend[K][2][r] = __builtin_amdgcn_fmed3f(s[4] + s[5] + s[6], 0.f, INFINITY);
                    pend[K][3][r] = __builtin_amdgcn_fmed3f(s[5] - s[6] - s[7], 0.f, INFINITY);
                }
            }
#pragma unroll
            for (int x = 0; x < 16; x++) a[x] = an[x];
        });
        if (!inplace) emit(kk); // (another buffer, or sums only: nothing to wait for)
    });
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(3);
    if (inplace) {
        __syncthreads(); // every wavefront of the team has read its patches
        static_for<W::BPW>([&](auto kk) {
            if (decltype(kk)::value == 1 && !two) return;
            emit(kk);
        });
    }
    return lsum;
}

template <int S>
__global__ __launch_bounds__(Geo<S>::NTHR) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_net_team_wino(
    const float *boards, const int *leaf_kind, int n, const float *prm, int front, int pol, int val, float *value_out,
    float *logp_out)
{
    using G = Geo<S>;
    using WG = WinoTeamGeo<S>;
    extern __shared__ float lds[];
    float *F = lds, *W = lds + G::BUF;
    float *red = lds + 2 * G::BUF + WG::TAIL; // [16]: two reduction rows of up to 8 wavefronts
    float *logit = red + 16;                  // [SIZE]
    float *dump = red;                        // (stores of dummy tiles: ROW + 2 floats, free while the layers run)
    int *todo = reinterpret_cast<int *>(logit + G::SIZE); // [ceil(n / gridDim.x)] positions of this workgroup
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int co = lane & 15, kq = lane >> 4;
    for (int i = tid; i < 2 * G::BUF + WG::TAIL; i += G::NTHR) lds[i] = 0.f; // halos and the tail stay zero from here on
    __syncthreads();
    const int nl = front - 1 + pol + val;
    const float *p_policy = prm + FIRST_FLOATS + (size_t)nl * LAYER_FLOATS;
    const float *p_value = p_policy + POLICY_FLOATS;
    const float *p_wino = p_value + VALUE_FLOATS;
    // execution order: trunk, value branch, policy branch (parameter order: trunk, policy, value)
    auto layer_params = [&](int e) {
        const int idx = e < front - 1 ? e : (e < front - 1 + val ? e + pol : e - val);
        return p_wino + (size_t)idx * WINO_FLOATS;
    };
    const float *p_first = p_wino + (size_t)nl * WINO_FLOATS;
    float bp[4]; // policy head as a B operand: B[k = ci][n = tap]
#pragma unroll
    for (int ks = 0; ks < 4; ks++) bp[ks] = co < 9 ? p_policy[co * CH + ks * 4 + kq] : 0.f;
    const float p_bias = p_policy[9 * CH];
    const float wv = p_value[co], v_bias = p_value[CH];

    int n_mine;
    if (leaf_kind != nullptr) {
        n_mine = list_positions<G::NTHR>(leaf_kind, n, todo, reinterpret_cast<int *>(red), tid);
    } else {
        n_mine = n > (int)blockIdx.x ? (n - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x : 0;
    }
    float u[64], ush = 0.f;
    if (nl > 0) {
        load_wino<S>(layer_params(0), lane, u);
        ush = layer_params(0)[64 * 64 + co];
    }
    const float u0sh = p_first[16 * 64 + co];
    for (int it = 0; it < n_mine; it++) {
        const int pos = leaf_kind != nullptr ? todo[it] : (int)blockIdx.x + it * (int)gridDim.x;
        // raw planes into channels 0..1 of W (its other channels may hold interior values of the
        // previous position: the input layer never reads them)
        for (int i = tid; i < CIN * G::SIZE; i += G::NTHR) {
            const int c = i / G::SIZE, p = i - c * G::SIZE;
            W[c * G::PS + pix_index<S>(p)] = boards[(size_t)pos * CIN * G::SIZE + i];
        }
        float u0[16]; // (fetched per position from an offset the compiler cannot see through: no registers held across the layers)
        {
            unsigned o = (unsigned)lane * 4u;
            asm volatile("" : "+v"(o));
#pragma unroll
            for (int x = 0; x < 16; x++) u0[x] = *reinterpret_cast<const float *>(reinterpret_cast<const char *>(p_first) + x * 256 + o);
        }
        __syncthreads();
        wino_team_layer<S, 1, false, MODE_WRITE>(W, F, dump, u0, nullptr, u0sh, wave, lane);
        __syncthreads();
        float vpart = 0.f; // this lane's share of the value head's sum
        // Buffers: a layer writes the buffer it does not read wherever that one is free (no wait between
        // a wavefront's matrix work and its stores); only the later layers of the value branch run in
        // place, while the other buffer still holds the trunk output for the policy branch.
        float *cur = F, *trunk_out = F; // where the latest activations / the trunk output are
        // without a value branch the head reads the trunk output (before the policy branch reuses its buffer)
        auto value_from_trunk = [&]() {
            for (int p = tid; p < G::SIZE; p += G::NTHR) {
                const int ctr = pix_index<S>(p);
#pragma unroll
                for (int ci = 0; ci < CH; ci++) vpart += p_value[ci] * trunk_out[ci * G::PS + ctr];
            }
            __syncthreads();
        };
        if (val == 0 && front == 1) value_from_trunk();
        for (int e = 0; e < nl; e++) {
            const bool trunk = e < front - 1;
            const bool in_value = !trunk && e < front - 1 + val;
            const int j = trunk ? 0 : (in_value ? e - (front - 1) : e - (front - 1) - val); // index inside a branch
            const float *src = (!trunk && j == 0) ? trunk_out : cur;
            float *other = src == F ? W : F;
            float *dst = (in_value && j > 0) ? cur : other;
            const float *nextp = layer_params(e + 1 < nl ? e + 1 : 0); // (the last layer fetches the next position's first)
            const float sh = ush;
            if (val > 0 && e == front - 1 + val - 1) {
                // last layer of the value branch: only the per-channel board sums are needed
                vpart = wv * wino_team_layer<S, 4, true, MODE_VALUE_SUM>(src, dst, dump, u, nextp, sh, wave, lane);
            } else {
                if (src == dst) wino_team_layer<S, 4, true, MODE_WRITE, true>(src, dst, dump, u, nextp, sh, wave, lane);
                else wino_team_layer<S, 4, true, MODE_WRITE, false>(src, dst, dump, u, nextp, sh, wave, lane);
            }
            ush = nextp[64 * 64 + co];
            cur = dst;
            if (trunk) trunk_out = dst;
            __syncthreads();
            if (val == 0 && e == front - 2) value_from_trunk();
        }
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) vpart += __shfl_xor(vpart, o);
        // ---- policy head: partial planes on the matrix cores, gather + log-softmax
        const float *hsrc = pol > 0 ? cur : trunk_out;
        float *tb = hsrc == F ? W : F;
        {
            TileMap<G::TQ + 1> tm; // (formed here: the layers need the registers)
            int lo = lane;
            asm volatile("" : "+v"(lo));
            make_tile_map<G>(wave, lo, tm);
            policy_taps_layer<G>(hsrc, tb, bp, tm, wave, lane);
        }
        if (lane == 0) red[wave] = vpart;
        __syncthreads();
        if (tid == 0) { // ---- value head: board average of the 1x1 convolution + bias, tanh
            float vt = 0.f;
#pragma unroll
            for (int w = 0; w < G::NW; w++) vt += red[w];
            value_out[pos] = tanhf(vt / (float)G::SIZE + v_bias);
        }
        __syncthreads(); // red is reused by the softmax
        policy_softmax<G>(tb, p_bias, red, logit, logp_out + (size_t)pos * G::SIZE, tid, wave, lane);
        __syncthreads(); // (the dump area overlaps red / logit: the next position's layers wait for the softmax)
    }
}

// ---------------------------------------------------------------------------------
// Winograd form with ONE wavefront per SIMD (four positions per CU): a wavefront alone on a
// SIMD owns the whole register file (512 registers), so
//  * the 16 transformed accumulators exist twice: the matrix instructions of block b+1 are
//    issued before the inverse transform of block b, which then runs from the other set while
//    the matrix cores work (the matrix instructions of a wavefront and its own vector
//    instructions overlap as long as they do not depend on each other);
//  * the trunk output waits in registers, not in private memory, while the value branch owns
//    the buffer;
//  * a single in-place LDS buffer per position is enough for every board up to 20x20.
// Tile grid: the board's own (ceil(S/2) tiles per tile row), row-major, blocks of 16 tiles =
// the M side of a matrix instruction; a lane loads the whole 4x4 patch of its tile and channel.
// In place: the outputs of block b are stored once every patch of block b+1 has been requested
// (a patch reaches one row and one column into the neighbouring tiles, i.e. at most
// ceil(S/2) + 1 <= 16 tiles back).

// Stores (wr), adds up (sum) or parks (keep) the outputs o[q][r] of block mb: q = 2*row + column inside
// the 2x2 tile, r = the lane's r-th tile (tile 16*mb + 4*kq + r), channel co.  Cells outside an odd board
// are halo cells and receive zero; tiles past the grid store into the dump cells.
template <int S, bool wr, bool keep, bool sum>
__device__ __forceinline__ void wide_emit(float *buf, int mb, int cob, int kq, float (&o)[4][4], float (&fc)[WideGeo<S>::NB * 16],
                                          float &lsum)
{
    using G = Geo<S>;
    using W = WideGeo<S>;
    constexpr bool ODD = (S & 1) != 0;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const unsigned T = (unsigned)(mb * 16 + r) + 4u * (unsigned)kq;
        const unsigned ty = (T * W::DIV) >> 16, tx = T - ty * W::TW;
        const bool valid = (mb + 1) * 16 <= W::TILES || T < (unsigned)W::TILES;
        const bool right = !ODD || tx != W::TW - 1, below = !ODD || ty != W::TW - 1;
        o[0][r] = valid ? o[0][r] : 0.f;
        o[1][r] = valid && right ? o[1][r] : 0.f;
        o[2][r] = valid && below ? o[2][r] : 0.f;
        o[3][r] = valid && right && below ? o[3][r] : 0.f;
        if constexpr (wr) {
            float *c = buf + (valid ? cob + (int)(ty * (2 * G::ROW) + tx * 2) : G::BUF + WINO_DUMP);
            c[0] = o[0][r];
            c[1] = o[1][r];
            c[G::ROW] = o[2][r];
            c[G::ROW + 1] = o[3][r];
        }
        if constexpr (sum) lsum += (o[0][r] + o[1][r]) + (o[2][r] + o[3][r]);
        if constexpr (keep) {
#pragma unroll
            for (int q = 0; q < 4; q++) fc[mb * 16 + q * 4 + r] = o[q][r];
        }
    }
}

// KS = K steps: 4 for 16 input channels, 1 for the two input planes (K indices 2..3 carry zero weights and
// read planes 0..1 again).  ROLL: the next layer's weights replace u[] as the last block retires them.
template <int S, int KS, bool ROLL, bool wr, bool keep, bool sum>
__device__ __forceinline__ void wide_layer(float *buf, float (&u)[16 * KS], const float *nextp, float sh, int lane,
                                           float (&fc)[WideGeo<S>::NB * 16], float &lsum)
{
    using G = Geo<S>;
    using W = WideGeo<S>;
    constexpr int NB = W::NB, NC = NB * KS;
    int l = lane;
    asm volatile("" : "+v"(l));
    const int kq = l >> 4, co = l & 15;
    const int kqp = KS == 4 ? kq : (kq & 1); // plane of this lane's K index in K step 0
    const int cob = co * G::PS + G::ROW + 1;
    auto load_patch = [&](int c, float (&d)[16]) {
        const int mb = c / KS;
        unsigned T = (unsigned)(mb * 16) + (unsigned)co;
        if ((mb + 1) * 16 > W::TILES) T = T < (unsigned)W::TILES ? T : (unsigned)W::TILES - 1; // (a dummy tile reads the last tile again)
        const unsigned ty = (T * W::DIV) >> 16, tx = T - ty * W::TW;
        int po = (int)(kqp * G::PS + 2 * ty * G::ROW + 2 * tx) + (c % KS) * 4 * G::PS;
        asm("" : "+v"(po)); // one address per patch; the cells are immediate offsets
        const float *p = buf + po;
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) d[i * 4 + j] = p[i * G::ROW + j];
    };
    auto transform = [&](const float (&d)[16], float (&a)[16]) { // Bt d B
        float t[16];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            t[j] = d[j] - d[8 + j];
            t[4 + j] = d[4 + j] + d[8 + j];
            t[8 + j] = d[8 + j] - d[4 + j];
            t[12 + j] = d[4 + j] - d[12 + j];
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {
            a[i * 4] = t[i * 4] - t[i * 4 + 2];
            a[i * 4 + 1] = t[i * 4 + 1] + t[i * 4 + 2];
            a[i * 4 + 2] = t[i * 4 + 2] - t[i * 4 + 1];
            a[i * 4 + 3] = t[i * 4 + 1] - t[i * 4 + 3];
        }
    };
    // At Y A of the accumulators' components r0..r1, ReLU
    auto inverse = [&](const f32x4 (&acc)[16], float (&o)[4][4], int r0, int r1) {
#pragma unroll
        for (int r = r0; r < r1; r++) {
            float s[8];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                s[j] = acc[j][r] + acc[4 + j][r] + acc[8 + j][r];
                s[4 + j] = acc[4 + j][r] - acc[8 + j][r] - acc[12 + j][r];
            }
            o[0][r] = __builtin_amdgcn_fmed3f(s[0] + s[1] + s[2], 0.f, INFINITY);
            o[1][r] = __builtin_amdgcn_fmed3f(s[1] - s[2] - s[3], 0.f, INFINITY);
            o[2][r] = __builtin_amdgcn_fmed3f(s[4] + s[5] + s[6], 0.f, INFINITY);
            o[3][r] = __builtin_amdgcn_fmed3f(s[5] - s[6] - s[7], 0.f, INFINITY);
        }
    };
    float a[16], d[16];
    load_patch(0, d);
    transform(d, a);
    if (NC > 1) load_patch(1, d);
    // The batch-norm shift enters through the C operand of a block's first matrix instructions: the four
    // accumulators that each feed exactly one of the 2x2 outputs start from +-shift, the others from zero.
    const f32x4 shp = (f32x4){sh, sh, sh, sh}, shn = (f32x4){-sh, -sh, -sh, -sh}, zero = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 acc[2][16]; // the accumulators of the block in flight and of the block before it
    float pend[4][4];
    static_for<NC>([&](auto cc) {
        constexpr int c = decltype(cc)::value;
        constexpr int mb = c / KS, ks = c % KS, cur = mb & 1;
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(0);
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int x = i * 4 + j;
                if (ks == 0)
                    acc[cur][x] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[x], u[x * KS + ks],
                                                                       (x == 0 || x == 15) ? shp : ((x == 3 || x == 12) ? shn : zero), 0, 0, 0);
                else acc[cur][x] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[x], u[x * KS + ks], acc[cur][x], 0, 0, 0);
            }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(AZ_WINO_PRIO);
        // ---- vector work in the shadow of the matrix instructions just issued (none of it reads acc[cur])
        float an[16];
        if (c + 1 < NC) transform(d, an);
        if (c + 2 < NC) load_patch(c + 2, d);
        if (ROLL && mb == NB - 1) { // the next layer's weights replace this layer's as their last use retires
            unsigned o = (unsigned)(l + ks * 64) * 4u;
            asm("" : "+v"(o));
#pragma unroll
            for (int x = 0; x < 16; x++)
                u[x * KS + ks] = *reinterpret_cast<const float *>(reinterpret_cast<const char *>(nextp + x * KS * 64) + o);
        }
        if (mb > 0) { // the block before: inverse transform over the first K steps, then its stores
            if constexpr (KS == 4) {
                if (ks == 0) inverse(acc[cur ^ 1], pend, 0, 2);
                if (ks == 1) inverse(acc[cur ^ 1], pend, 2, 4);
                // (every patch of this block has been requested by the end of K step 1)
                if (ks == 2) wide_emit<S, wr, keep, sum>(buf, mb - 1, cob, kq, pend, fc, lsum);
            } else {
                inverse(acc[cur ^ 1], pend, 0, 4);
                wide_emit<S, wr, keep, sum>(buf, mb - 1, cob, kq, pend, fc, lsum);
            }
        }
#pragma unroll
        for (int x = 0; x < 16; x++) a[x] = an[x];
    });
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(3);
    inverse(acc[(NB - 1) & 1], pend, 0, 4);
    wide_emit<S, wr, keep, sum>(buf, NB - 1, cob, kq, pend, fc, lsum);
}

// the trunk output back from the registers it was parked in
template <int S> __device__ __forceinline__ void wide_restore(float *buf, int lane, float (&fc)[WideGeo<S>::NB * 16])
{
    using G = Geo<S>;
    using W = WideGeo<S>;
    int l = lane;
    asm volatile("" : "+v"(l));
    float unused = 0.f;
#pragma unroll
    for (int mb = 0; mb < W::NB; mb++) {
        float o[4][4];
#pragma unroll
        for (int q = 0; q < 4; q++)
#pragma unroll
            for (int r = 0; r < 4; r++) o[q][r] = fc[mb * 16 + q * 4 + r];
        wide_emit<S, true, false, false>(buf, mb, (l & 15) * G::PS + G::ROW + 1, l >> 4, o, fc, unused);
    }
}


template <int S>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1))) void k_net_wide(
    const float *boards, const int *leaf_kind, const unsigned *expand_mask, int n, const float *prm, int front, int pol, int val,
    float *value_out, float *logp_out)
{
    using G = Geo<S>;
    using W = WideGeo<S>;
    extern __shared__ float lds[];
    float *buf = lds;
    int *todo = reinterpret_cast<int *>(lds + G::BUF + 64); // [ceil(n / gridDim.x)] positions of this workgroup
    const int lane = threadIdx.x;
    const int co = lane & 15, kq = lane >> 4;
    for (int i = lane; i < G::BUF + 64; i += 64) buf[i] = 0.f; // halos (and the tail) stay zero from here on
    const int nl = front - 1 + pol + val;
    const float *p_policy = prm + FIRST_FLOATS + (size_t)nl * LAYER_FLOATS;
    const float *p_value = p_policy + POLICY_FLOATS;
    const float *p_wino = p_value + VALUE_FLOATS;
    // execution order: trunk, value branch, policy branch (parameter order: trunk, policy, value)
    auto layer_params = [&](int e) {
        const int idx = e < front - 1 ? e : (e < front - 1 + val ? e + pol : e - val);
        return p_wino + (size_t)idx * WINO_FLOATS;
    };
    const float *p_first = p_wino + (size_t)nl * WINO_FLOATS;
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

    float fc[W::NB * 16]; // the trunk output while the value branch owns the buffer
#pragma unroll
    for (int i = 0; i < W::NB * 16; i++) fc[i] = 0.f;
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
    float u[64], ush = 0.f; // the weights of the first 16-channel layer: loaded here once, later rolled in by the last layer
    float u0[16];           // the input layer
#pragma unroll
    for (int x = 0; x < 16; x++) u0[x] = p_first[x * 64 + lane];
    const float u0sh = p_first[16 * 64 + co];
    if (nl > 0) {
        load_wino<S>(layer_params(0), lane, u);
        ush = layer_params(0)[64 * 64 + co];
    }
    for (int it = 0; it < n_mine; it++) {
        const int pos = listed ? todo[it] : (int)blockIdx.x + it * (int)gridDim.x;
#pragma unroll
        for (int i = 0; i < PPL; i++) { // the two input planes (requested while the previous position was in its heads)
            const int p = lane + 64 * i;
            if (p < G::SIZE) {
                const int ctr = pix_index<S>(p);
                buf[ctr] = in0[i];
                buf[G::PS + ctr] = in1[i];
            }
        }
        __syncthreads();
        float vsum = 0.f, lsum = 0.f;
        // input layer; it is also the trunk's last layer when front == 1
        if (front > 1) wide_layer<S, 1, false, true, false, false>(buf, u0, nullptr, u0sh, lane, fc, lsum);
        else if (val > 0) wide_layer<S, 1, false, true, true, false>(buf, u0, nullptr, u0sh, lane, fc, lsum);
        else wide_layer<S, 1, false, true, false, true>(buf, u0, nullptr, u0sh, lane, fc, lsum);
        if (front == 1 && val == 0) vsum = lsum;
        __syncthreads();
        for (int e = 0; e < nl; e++) {
            const bool last_trunk = e == front - 2;
            const bool last_value = val > 0 && e == front - 1 + val - 1;
            const bool first_policy = pol > 0 && e == front - 1 + val;
            const float *nextp = layer_params(e + 1 < nl ? e + 1 : 0); // (the last layer fetches the next position's first)
            if (first_policy && val > 0) { // the trunk output comes back
                wide_restore<S>(buf, lane, fc);
                __syncthreads();
            }
            const float sh = ush;
            ush = nextp[64 * 64 + co]; // (requested a layer ahead: its latency runs under this layer)
            lsum = 0.f;
            if (last_value) wide_layer<S, 4, true, false, false, true>(buf, u, nextp, sh, lane, fc, lsum);
            else if (last_trunk && val > 0) wide_layer<S, 4, true, true, true, false>(buf, u, nextp, sh, lane, fc, lsum);
            else if (last_trunk) wide_layer<S, 4, true, true, false, true>(buf, u, nextp, sh, lane, fc, lsum);
            else wide_layer<S, 4, true, true, false, false>(buf, u, nextp, sh, lane, fc, lsum);
            if (last_value || (last_trunk && val == 0)) vsum = lsum;
            __syncthreads();
        }
        // policy head input: the policy branch's output, or the trunk output when there is no such branch
        if (pol == 0 && val > 0) {
            wide_restore<S>(buf, lane, fc);
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
        // planes 2..15 keep leftovers (partial planes, old activations): the input layer reads planes 0..1 only
    }
}

template <int S> int launch_wide(const float *boards, const int *leaf_kind, const unsigned *expand_mask, int n, const float *prm,
                                 int front, int pol, int val, float *value_out, float *logp_out, hipStream_t st)
{
    using W = WideGeo<S>;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    int grid = cus * W::WGS;
    if (grid > n) grid = n;
    const size_t lds_bytes = (size_t)W::LDS_BYTES + (size_t)((n + grid - 1) / grid) * sizeof(int);
    static size_t configured[64] = {};
    if (dev < 0 || dev >= 64 || lds_bytes > configured[dev]) {
        if (hipFuncSetAttribute((const void *)k_net_wide<S>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess)
            return AZ_ERR_HIP;
        if (dev >= 0 && dev < 64) configured[dev] = lds_bytes;
    }
    hipLaunchKernelGGL(k_net_wide<S>, dim3(grid), dim3(64), lds_bytes, st, boards, leaf_kind, expand_mask, n, prm, front, pol, val,
                       value_out, logp_out);
    return hipGetLastError() == hipSuccess ? AZ_OK : AZ_ERR_HIP;
}

template <int S> int launch_team_wino(const float *boards, const int *leaf_kind, int n, const float *prm, int front, int pol, int val,
                                      float *value_out, float *logp_out, hipStream_t st)
{
    using G = Geo<S>;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    int grid = cus * G::WGS;
    if (grid > n) grid = n;
    const size_t lds_bytes = (size_t)G::FIXED_BYTES + WinoTeamGeo<S>::TAIL * 4 + (size_t)((n + grid - 1) / grid) * sizeof(int);
    static size_t configured[64] = {};
    if (dev < 0 || dev >= 64 || lds_bytes > configured[dev]) {
        if (hipFuncSetAttribute((const void *)k_net_team_wino<S>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) !=
            hipSuccess)
            return AZ_ERR_HIP;
        if (dev >= 0 && dev < 64) configured[dev] = lds_bytes;
    }
    hipLaunchKernelGGL(k_net_team_wino<S>, dim3(grid), dim3(G::NTHR), lds_bytes, st, boards, leaf_kind, n, prm, front, pol, val,
                       value_out, logp_out);
    return hipGetLastError() == hipSuccess ? AZ_OK : AZ_ERR_HIP;
}

template <int S, bool WINO> int launch_solo(const float *boards, const int *leaf_kind, const unsigned *expand_mask, int n,
                                            const float *prm, int front, int pol, int val, float *value_out, float *logp_out,
                                            hipStream_t st)
{
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    static const int wgs_override = getenv("AZ_NET_SOLO_WGS") != nullptr ? atoi(getenv("AZ_NET_SOLO_WGS")) : 0; // (experiments)
    int grid = cus * (wgs_override > 0 ? wgs_override : SoloGeo<S>::WGS);
    if (grid > n) grid = n;
    const size_t lds_bytes = (size_t)SoloGeo<S>::LDS_BYTES + (size_t)((n + grid - 1) / grid) * sizeof(int);
    static size_t configured[64] = {};
    if (dev < 0 || dev >= 64 || lds_bytes > configured[dev]) {
        if (hipFuncSetAttribute((const void *)k_net_solo<S, WINO>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) !=
            hipSuccess)
            return AZ_ERR_HIP;
        if (dev >= 0 && dev < 64) configured[dev] = lds_bytes;
    }
    hipLaunchKernelGGL((k_net_solo<S, WINO>), dim3(grid), dim3(64), lds_bytes, st, boards, leaf_kind, expand_mask, n, prm, front, pol,
                       val, value_out, logp_out);
    return hipGetLastError() == hipSuccess ? AZ_OK : AZ_ERR_HIP;
}

template <class G> int launch(const float *boards, const int *leaf_kind, int n, const float *prm, int front, int pol, int val,
                              float *value_out, float *logp_out, hipStream_t st)
{
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    int grid = cus * G::WGS;
    if (grid > n) grid = n;
    // + the workgroup's own list of positions (at most ceil(n / grid) entries)
    const size_t lds_bytes = (size_t)G::FIXED_BYTES + (size_t)((n + grid - 1) / grid) * sizeof(int);
    // largest size the kernel has been opened up to, per device (the attribute is per device)
    static size_t configured[64] = {};
    if (dev < 0 || dev >= 64 || lds_bytes > configured[dev]) {
        if (hipFuncSetAttribute((const void *)k_net_eval<G>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds_bytes) != hipSuccess)
            return AZ_ERR_HIP;
        if (dev >= 0 && dev < 64) configured[dev] = lds_bytes;
    }
    hipLaunchKernelGGL(k_net_eval<G>, dim3(grid), dim3(G::NTHR), lds_bytes, st, boards, leaf_kind, n, prm, front, pol, val,
                       value_out, logp_out);
    return hipGetLastError() == hipSuccess ? AZ_OK : AZ_ERR_HIP;
}

} // namespace

// the F(4x4,3x3) kernel (az_net_w43.hip, a translation unit of its own: it is compiled with other register options)
int az_net_w43_launch(int shape, int n, const float *d_boards, const int *d_leaf_kind, const unsigned *d_expand_mask,
                      const float *d_params, int front_layers, int policy_layers, int value_layers, float *d_value, float *d_logp,
                      hipStream_t st);

// kernel choice of az_net_eval: AZ_NET_FORM_AUTO unless the environment (AZ_NET_FORM=<number>) or
// az_net_set_form says otherwise
static std::atomic<int> g_net_form{getenv("AZ_NET_FORM") != nullptr ? atoi(getenv("AZ_NET_FORM")) : AZ_NET_FORM_AUTO};

int az_net_get_form() { return g_net_form.load(std::memory_order_relaxed); }

extern "C" {

int az_net_set_form(int form)
{
    if (form < AZ_NET_FORM_AUTO || form > AZ_NET_FORM_W43) return AZ_ERR_ARG;
    g_net_form.store(form, std::memory_order_relaxed);
    return AZ_OK;
}

int az_net_param_floats(int front_layers, int policy_layers, int value_layers)
{
    if (front_layers < 1 || policy_layers < 0 || value_layers < 0) return 0;
    return FIRST_FLOATS + (front_layers - 1 + policy_layers + value_layers) * (LAYER_FLOATS + WINO_FLOATS + W43_FLOATS) +
           POLICY_FLOATS + VALUE_FLOATS + WINO_FIRST_FLOATS + W43_FIRST_FLOATS;
}

int az_net_eval(int shape, int n, const float *d_boards, const int *d_leaf_kind, const float *d_params,
                int front_layers, int policy_layers, int value_layers, float *d_value, float *d_logp, void *stream)
{
    return az_net_eval_masked(shape, n, d_boards, d_leaf_kind, nullptr, d_params, front_layers, policy_layers, value_layers, d_value,
                              d_logp, stream);
}

} // extern "C"

// d_expand_mask: see az_common.h.  The kernels that are nobody's default (the teams) go by the codes.
int az_net_eval_masked(int shape, int n, const float *d_boards, const int *d_leaf_kind, const unsigned *d_expand_mask,
                       const float *d_params, int front_layers, int policy_layers, int value_layers, float *d_value,
                       float *d_logp, void *stream)
{
    if (n < 0 || front_layers < 1 || policy_layers < 0 || value_layers < 0) return AZ_ERR_ARG;
    if (n == 0) return AZ_OK;
    if (!d_boards || !d_params || !d_value || !d_logp) return AZ_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    // Boards up to 15x15 take the one-wavefront-per-position kernel with two wavefronts per SIMD, in the
    // Winograd form of the 3x3 layers from 12x12 on (fewer matrix instructions than the direct form there);
    // larger boards the Winograd kernel with one wavefront per SIMD (k_net_wide: measured faster than both
    // team kernels at every size from 16x16 to 20x20); 13x13 - 16x16, which are a 4x4 grid of 4x4 output
    // tiles, and 19x19 / 20x20 (5x5 tiles, two wavefronts per position) the Winograd F(4x4,3x3) kernels
    // (az_net_w43.hip: per 8192 positions 0.29 ms at 13x13 - 15x15 against 0.36 - 0.38, 0.35 against 0.46 at
    // 16x16, 0.72 against 0.87 at 19x19, 0.64 against 0.85 at 20x20; at 17x17 and 18x18, where the masks of
    // the overhanging tiles cost what the form saves, 0.73 / 0.71 against 0.73 / 0.70: k_net_wide stays).
    // az_net_set_form overrides the choice.
    const int form = g_net_form.load(std::memory_order_relaxed);
    // A batch that leaves CUs idle (one tree behind the cmcts facade, a small arena) is bound by the latency of
    // one position, not by throughput: a team of wavefronts per position finishes it sooner than one wavefront
    // (measured per launch at n <= 256, eight wavefronts per position: 25 us against 52 us at 15x15, 37 us
    // against 106 us at 20x20; the one-wavefront kernels win from about 2 positions per CU on, above 15x15
    // from about 3).
    static const int cus = [] {
        int dev = 0, c = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev);
        return c;
    }();
    const bool few = form == AZ_NET_FORM_AUTO && (n <= cus || (shape >= 16 && n <= 2 * cus));
    const bool team = form == AZ_NET_FORM_TEAM || few;
    const bool wino = form == AZ_NET_FORM_WINOGRAD || (form == AZ_NET_FORM_AUTO && shape >= 12);
    if ((shape >= 13 && form == AZ_NET_FORM_W43) || (((shape >= 13 && shape <= 16) || shape >= 19) && form == AZ_NET_FORM_AUTO && !few))
        return az_net_w43_launch(shape, n, d_boards, d_leaf_kind, d_expand_mask, d_params, front_layers, policy_layers, value_layers,
                                 d_value, d_logp, st);
    switch (shape) {
#define AZ_NET_CASE(s)                                                                                                   \
    case s:                                                                                                              \
        if (s >= 12 && (form == AZ_NET_FORM_WIDE || (form == AZ_NET_FORM_AUTO && s >= 16 && !few)))                      \
            return launch_wide<(s >= 12 ? s : 12)>(d_boards, d_leaf_kind, d_expand_mask, n, d_params, front_layers,      \
                                                   policy_layers, value_layers, d_value, d_logp, st);                    \
        if (s <= 15 && !team && wino)                                                                                    \
            return launch_solo<(s <= 15 ? s : 15), true>(d_boards, d_leaf_kind, d_expand_mask, n, d_params, front_layers, \
                                                         policy_layers, value_layers, d_value, d_logp, st);              \
        if (s <= 15 && !team)                                                                                            \
            return launch_solo<(s <= 15 ? s : 15), false>(d_boards, d_leaf_kind, d_expand_mask, n, d_params, front_layers, \
                                                          policy_layers, value_layers, d_value, d_logp, st);             \
        if (s >= 16 && form == AZ_NET_FORM_WINOGRAD)                                                                     \
            return launch_team_wino<(s >= 16 ? s : 16)>(d_boards, d_leaf_kind, n, d_params, front_layers, policy_layers, \
                                                       value_layers, d_value, d_logp, st);                              \
        if (few && n <= cus) /* one position per CU at most: eight wavefronts per position */                            \
            return launch<Geo<s, 8>>(d_boards, d_leaf_kind, n, d_params, front_layers, policy_layers, value_layers,      \
                                     d_value, d_logp, st);                                                               \
        return launch<Geo<s>>(d_boards, d_leaf_kind, n, d_params, front_layers, policy_layers, value_layers, d_value, d_logp, st);
#ifdef AZ_NET_DEV_SHAPE // (development builds: one board size only)
        AZ_NET_CASE(AZ_NET_DEV_SHAPE)
#else
        AZ_NET_CASE(5) AZ_NET_CASE(6) AZ_NET_CASE(7) AZ_NET_CASE(8) AZ_NET_CASE(9) AZ_NET_CASE(10) AZ_NET_CASE(11)
        AZ_NET_CASE(12) AZ_NET_CASE(13) AZ_NET_CASE(14) AZ_NET_CASE(15) AZ_NET_CASE(16) AZ_NET_CASE(17)
        AZ_NET_CASE(18) AZ_NET_CASE(19) AZ_NET_CASE(20)
#endif
#undef AZ_NET_CASE
    default: return AZ_ERR_ARG;
    }
}



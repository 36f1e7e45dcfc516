// az_net_team.h -- device functions of the direct-form network evaluation by a team of wavefronts per
// position (geometry, packed parameter layout, implicit-GEMM convolution, policy head): shared by
// az_net.hip (k_net_eval, the batched evaluator) and az_kernels.hip (the one-launch network search of
// batches that leave CUs idle).  See the head of az_net.hip for the design.
#ifndef AZ_NET_TEAM_H
#define AZ_NET_TEAM_H

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace aznet {

constexpr int CH = 16;  // conv channels of the reference configuration (config.py:15)
constexpr int CIN = 2;  // input planes (config.py:12)
typedef float f32x4 __attribute__((ext_vector_type(4)));

// NWF = 0: the team size chosen for throughput (below); NWF > 0: that many wavefronts per position (as far
// as the board has tiles), for launches that are bound by the latency of one position
template <int S_, int NWF = 0> struct Geo {
    static constexpr int S = S_;
    static constexpr int SIZE = S * S;
    static constexpr int ROW = S + 2;               // one halo column on each side
    static constexpr int PLANE = ROW * ROW;
    // plane stride: odd, so the epilogue's writes (lanes step by one plane per output
    // channel) spread over all LDS banks
    static constexpr int PS = PLANE | 1;
    // a board of 16k + 1 pixels: the last pixel is computed on the vector ALU
    static constexpr bool CORNER = SIZE % 16 == 1;
    static constexpr int NT = CORNER ? SIZE / 16 : (SIZE + 15) / 16; // 16-pixel tiles
    static constexpr int BUF = CH * PS;             // floats of one activation buffer
    static constexpr int FIXED_BYTES = (2 * BUF + SIZE + 16) * 4;
    // Workgroups per CU and wavefronts per workgroup: the kernel is built for two wavefronts
    // per SIMD (256 registers each), i.e. eight per CU: four workgroups of two where the LDS
    // holds four positions, else two workgroups of four.
    static constexpr int WGS_FIT = (158 * 1024) / FIXED_BYTES;
    static constexpr int WGS = NWF > 0 ? (8 / NWF < 1 ? 1 : (8 / NWF > WGS_FIT ? WGS_FIT : 8 / NWF)) : (WGS_FIT >= 4 ? 4 : 2);
    static constexpr int NW0 = NWF > 0 ? NWF : (WGS == 4 ? 2 : 4);
    static constexpr int NW = NW0 > NT ? NT : NW0;
    static constexpr int TQ = NT / NW, TR = NT % NW; // wavefronts below TR take TQ + 1 tiles
    static constexpr int NTHR = 64 * NW;
    static_assert(WGS >= 1 && TQ >= 1, "geometry");
};

// index of pixel p inside a plane (with halo)
template <int S> __device__ __forceinline__ int pix_index(int p)
{
    return (p / S + 1) * Geo<S>::ROW + p % S + 1;
}

// packed parameter block (floats), built by the host:
//   first layer : B[5][64], scale[16], shift[16]
//   each 16->16 : B[36][64], scale[16], shift[16]      (front-1, policy, value layers in that order)
//   policy head : w[9][16], bias
//   value head  : w[16], bias
constexpr int FIRST_FLOATS = 5 * 64 + 32;
constexpr int LAYER_FLOATS = 36 * 64 + 32;
constexpr int POLICY_FLOATS = 9 * CH + 1;
constexpr int VALUE_FLOATS = CH + 1;

// parameters of one convolution as each lane needs them: its B operands (one per
// K step) and the folded batch-norm scale/shift of its output channel
template <int KSTEPS> struct LayerRegs {
    float b[KSTEPS];
    float sc, sh;
};
template <int KSTEPS> __device__ __forceinline__ void load_layer(const float *prm, int lane, LayerRegs<KSTEPS> &w)
{
    // One address register pair per 16 loads (the immediate offset of a load reaches 4 KB); the
    // empty asm keeps the compiler from deriving, and keeping, a separate 64-bit address per load.
#pragma unroll
    for (int c = 0; c < (KSTEPS + 15) / 16; c++) {
        const float *p = prm + c * 16 * 64 + lane;
        asm volatile("" : "+v"(p));
#pragma unroll
        for (int i = c * 16; i < KSTEPS && i < c * 16 + 16; i++) w.b[i] = p[(i - c * 16) * 64];
    }
    const float *q = prm + KSTEPS * 64 + (lane & 15);
    asm volatile("" : "+v"(q));
    w.sc = q[0];
    w.sh = q[16];
}

// sum over the four lanes that hold the same output channel (lane ^ 16, lane ^ 32, lane ^ 48)
__device__ __forceinline__ float sum_over_kq(float v)
{
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    return v;
}

enum { MODE_WRITE = 0, MODE_VALUE_SUM = 1 };

// What a lane needs to know about the tiles of its wavefront, computed once per
// launch (kept few on purpose: everything here stays in registers for the whole kernel).
//   rd[t]   : plane index of the pixel whose A operand this lane reads for tile t
//             (+ this lane's channel-quarter plane, so a K step adds only a constant)
//   wr[t]   : co*PS + plane index of the first of the lane's four output pixels of tile t
//   wrap[t] : how many of those four pixels lie before a row end (4 = no row end inside)
template <int NTMAX> struct TileMap {
    int rd[NTMAX], wr[NTMAX], wrap[NTMAX];
    int tile0;      // first tile of this wavefront (wave-uniform)
    int last_valid; // pixels of the lane's four that exist in the board's last, partial tile
    int crd, cwr;   // the same for the corner pixel of a board of 16k+1 pixels
};
template <class G> __device__ __forceinline__ void make_tile_map(int wave, int lane, TileMap<G::TQ + 1> &m)
{
    constexpr int S = G::S;
    const int co = lane & 15, kq = lane >> 4;
    m.tile0 = wave < G::TR ? wave * (G::TQ + 1) : G::TR * (G::TQ + 1) + (wave - G::TR) * G::TQ;
#pragma unroll
    for (int t = 0; t < G::TQ + 1; t++) {
        int p = (m.tile0 + t) * 16 + (lane & 15);
        p = p < G::SIZE ? p : G::SIZE - 1; // lanes past the board read a pixel nobody stores
        m.rd[t] = pix_index<S>(p) + kq * G::PS - (G::ROW + 1); // biased by the most negative tap offset: every K step adds a constant >= 0
        int p0 = (m.tile0 + t) * 16 + kq * 4;
        p0 = p0 < G::SIZE ? p0 : G::SIZE - 1;
        m.wr[t] = co * G::PS + pix_index<S>(p0);
        const int left = S - p0 % S;
        m.wrap[t] = left < 4 ? left : 4;
    }
    const int rest = G::SIZE - ((G::NT - 1) * 16 + kq * 4);
    m.last_valid = rest < 0 ? 0 : (rest > 4 ? 4 : rest);
    m.crd = pix_index<S>(G::SIZE - 1) + kq * G::PS - (G::ROW + 1);
    m.cwr = co * G::PS + pix_index<S>(G::SIZE - 1);
}

// One 3x3 convolution + folded batch norm + ReLU over the NTW tiles of this wavefront,
// `src` -> `dst` (LDS planes; dst may be src: the results are written after a barrier
// that follows the last read).
// KSTEPS = 36 for 16 input channels (tap-major, 4 channels per step), 5 for the
// 2-plane input layer (K index = tap*2 + plane, padded with zero weights; off0 =
// this lane's operand offset for each of its five K indices, relative to plane 0).
// CORNERW: this wavefront also computes the board's last pixel (boards of 16k+1 pixels).
// MODE_VALUE_SUM: nothing is written; returns this lane's sum of its outputs (the
// value head only needs the board sums per channel).
template <class G, int KSTEPS, int NTW, bool CORNERW, int MODE>
__device__ __forceinline__ float conv3x3(const float *src, float *dst, const LayerRegs<KSTEPS> &w, const int (&off0)[5],
                                         const TileMap<G::TQ + 1> &m, int lane)
{
    const int kq = lane >> 4;
    // (the batch-norm scale is folded into the weights by the host: the accumulators start
    // from the shift and the epilogue is the ReLU alone)
    f32x4 acc[NTW];
#pragma unroll
    for (int t = 0; t < NTW; t++) acc[t] = (f32x4){w.sh, w.sh, w.sh, w.sh};
    float cacc = 0.f;
    const int kqp = KSTEPS == 36 ? 0 : kq * G::PS; // the input layer's K index does not follow the channel quarters
    // operand offset of K step ks (a constant after unrolling for the 16-channel layers)
    auto koff = [&](int ks) {
        if constexpr (KSTEPS == 36) {
            const int tap = ks >> 2;
            return (ks & 3) * 4 * G::PS + (tap / 3) * G::ROW + (tap % 3); // (relative to the biased rd[])
        } else {
            return off0[ks] - kqp + (G::ROW + 1);
        }
    };
    // (the read bases pass through an empty asm per layer: otherwise every address of every
    // layer is precomputed outside the position loop and the kernel runs out of registers)
    int rd[NTW], crd = m.crd;
#pragma unroll
    for (int t = 0; t < NTW; t++) {
        rd[t] = m.rd[t];
        asm volatile("" : "+v"(rd[t]));
    }
    asm volatile("" : "+v"(crd));
    // The A operands of step ks+1 are requested into the registers step ks has just consumed,
    // one request after each matrix instruction (scheduling groups): a request is NTW - 1
    // matrix instructions ahead of its use.  Issue priority is low inside this loop and high
    // outside it: the vector phases of a wavefront go first while the other wavefront of the
    // SIMD issues long matrix instructions.
    float a[NTW], ca = 0.f;
#pragma unroll
    for (int t = 0; t < NTW; t++) a[t] = src[rd[t] + koff(0)];
    if constexpr (CORNERW) ca = src[crd + koff(0)];
    __builtin_amdgcn_s_setprio(0);
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ks++) {
        __builtin_amdgcn_sched_barrier(0);
        const float b = w.b[ks];
#pragma unroll
        for (int t = 0; t < NTW; t++) {
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], b, acc[t], 0, 0, 0);
            if (ks + 1 < KSTEPS) a[t] = src[rd[t] + koff(ks + 1)];
        }
        if constexpr (CORNERW) {
            cacc = fmaf(ca, b, cacc);
            if (ks + 1 < KSTEPS) ca = src[crd + koff(ks + 1)];
        }
        if (ks + 1 < KSTEPS) {
#pragma unroll
            for (int t = 0; t < NTW; t++) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); // one matrix instruction
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); // one LDS read
            }
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(3);
    if (MODE == MODE_WRITE && src == dst) __syncthreads(); // every wavefront has read its operands
    float vsum = 0.f;
#pragma unroll
    for (int t = 0; t < NTW; t++) {
        // accumulator register r of this lane: pixel tile*16 + kq*4 + r, channel co
        const bool partial = G::SIZE % 16 != 0 && !G::CORNER && m.tile0 + t == G::NT - 1; // wave-uniform
        int wr = m.wr[t], wrap = m.wrap[t];
        asm volatile("" : "+v"(wr), "+v"(wrap));
        float *d0 = dst + wr, *d1 = d0 + 2; // a row end skips two halo cells
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const float v = __builtin_amdgcn_fmed3f(acc[t][r], 0.f, INFINITY); // ReLU
            if (!partial || r < m.last_valid) {
                if constexpr (MODE == MODE_WRITE) (r >= wrap ? d1 : d0)[r] = v;
                else vsum += v;
            }
        }
    }
    if constexpr (CORNERW) {
        const float v = __builtin_amdgcn_fmed3f(sum_over_kq(cacc) + w.sh, 0.f, INFINITY);
        if (kq == 0) {
            if constexpr (MODE == MODE_WRITE) dst[m.cwr] = v;
            else vsum += v;
        }
    }
    return vsum;
}

// the wavefront's share of a layer: wavefronts below TR take TQ + 1 tiles, the last
// wavefront takes the corner pixel
template <class G, int KSTEPS, int MODE>
__device__ __forceinline__ float conv_layer(const float *src, float *dst, const LayerRegs<KSTEPS> &w, const int (&off0)[5],
                                            const TileMap<G::TQ + 1> &m, int wave, int lane)
{
    if constexpr (G::TR != 0) {
        if (wave < G::TR) return conv3x3<G, KSTEPS, G::TQ + 1, false, MODE>(src, dst, w, off0, m, lane);
    }
    if constexpr (G::CORNER) {
        if (wave == G::NW - 1) return conv3x3<G, KSTEPS, G::TQ, true, MODE>(src, dst, w, off0, m, lane);
    }
    return conv3x3<G, KSTEPS, G::TQ, false, MODE>(src, dst, w, off0, m, lane);
}

// Policy head, first half: the 3x3 convolution to one channel is split into a 1x1
// product on the matrix cores, T[tap][q] = sum_ci w[tap][ci] * src[ci][q] (N = the 9 taps),
// written as 9 planes of `tb`; the logit of pixel p is then bias + sum_tap T[tap][p + offset(tap)].
template <class G, int NTW, bool CORNERW>
__device__ __forceinline__ void policy_taps(const float *src, float *tb, const float (&bp)[4],
                                            const TileMap<G::TQ + 1> &m, int lane)
{
    const int n = lane & 15, kq = lane >> 4;
    f32x4 acc[NTW];
#pragma unroll
    for (int t = 0; t < NTW; t++) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float cacc = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ks++) {
#pragma unroll
        for (int t = 0; t < NTW; t++)
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(src[m.rd[t] + ks * 4 * G::PS + (G::ROW + 1)], bp[ks], acc[t], 0, 0, 0);
        if constexpr (CORNERW) cacc = fmaf(src[m.crd + ks * 4 * G::PS + (G::ROW + 1)], bp[ks], cacc);
    }
#pragma unroll
    for (int t = 0; t < NTW; t++) {
        const bool partial = G::SIZE % 16 != 0 && !G::CORNER && m.tile0 + t == G::NT - 1;
        int wr = m.wr[t], wrap = m.wrap[t];
        asm volatile("" : "+v"(wr), "+v"(wrap));
#pragma unroll
        for (int r = 0; r < 4; r++)
            if (n < 9 && (!partial || r < m.last_valid)) tb[wr + r + (r >= wrap ? 2 : 0)] = acc[t][r];
    }
    if constexpr (CORNERW) {
        const float v = sum_over_kq(cacc);
        if (kq == 0 && n < 9) tb[m.cwr] = v;
    }
}
template <class G>
__device__ __forceinline__ void policy_taps_layer(const float *src, float *tb, const float (&bp)[4],
                                                  const TileMap<G::TQ + 1> &m, int wave, int lane)
{
    if constexpr (G::TR != 0) {
        if (wave < G::TR) return policy_taps<G, G::TQ + 1, false>(src, tb, bp, m, lane);
    }
    if constexpr (G::CORNER) {
        if (wave == G::NW - 1) return policy_taps<G, G::TQ, true>(src, tb, bp, m, lane);
    }
    policy_taps<G, G::TQ, false>(src, tb, bp, m, lane);
}

// Policy head, second half: gather of the 9 partial planes + bias, log-softmax over the board
template <class G>
__device__ __forceinline__ void policy_softmax(const float *tb, float bias, float *red, float *logit, float *logp_row,
                                               int tid, int wave, int lane)
{
    float lmax = -INFINITY;
    for (int p = tid; p < G::SIZE; p += G::NTHR) {
        const int ctr = pix_index<G::S>(p);
        float acc = bias;
#pragma unroll
        for (int tap = 0; tap < 9; tap++) acc += tb[tap * G::PS + ctr + (tap / 3 - 1) * G::ROW + (tap % 3 - 1)];
        logit[p] = acc;
        lmax = fmaxf(lmax, acc);
    }
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) lmax = fmaxf(lmax, __shfl_xor(lmax, o));
    if (lane == 0) red[wave] = lmax;
    __syncthreads();
    lmax = red[0];
#pragma unroll
    for (int w = 1; w < G::NW; w++) lmax = fmaxf(lmax, red[w]);
    float lsum = 0.f;
    for (int p = tid; p < G::SIZE; p += G::NTHR) lsum += expf(logit[p] - lmax);
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) lsum += __shfl_xor(lsum, o);
    if (lane == 0) red[8 + wave] = lsum;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int w = 0; w < G::NW; w++) tot += red[8 + w];
    const float lse = lmax + logf(tot);
    for (int p = tid; p < G::SIZE; p += G::NTHR) logp_row[p] = logit[p] - lse;
}


// What a thread keeps for a whole launch: the parameters that do not change from position to position
template <class G> struct TeamCtx {
    LayerRegs<5> w0;        // input layer
    int off0[5];            // input layer: operand offset of this lane's K index per K step
    TileMap<G::TQ + 1> tm;
    float bp[4];            // policy head as a B operand: B[k = ci][n = tap]
    float p_bias, wv, v_bias;
    const float *p_layers, *p_value;
    int front, pol, val, nl; // nl = 16->16 layers: trunk, policy branch, value branch (parameter order)
};
template <class G>
__device__ __forceinline__ void team_setup(TeamCtx<G> &c, const float *prm, int front, int pol, int val, int wave, int lane)
{
    const int co = lane & 15, kq = lane >> 4;
    c.front = front;
    c.pol = pol;
    c.val = val;
    c.nl = front - 1 + pol + val;
    c.p_layers = prm + FIRST_FLOATS;
    const float *p_policy = c.p_layers + (size_t)c.nl * LAYER_FLOATS;
    c.p_value = p_policy + POLICY_FLOATS;
    load_layer<5>(prm, lane, c.w0);
#pragma unroll
    for (int ks = 0; ks < 5; ks++) { // K index kk = ks*4 + kq -> tap kk>>1 of plane kk&1
        const int kk = ks * 4 + kq;
        const int tap = kk < 18 ? kk >> 1 : 0;
        c.off0[ks] = (kk & 1) * G::PS + (tap / 3 - 1) * G::ROW + (tap % 3 - 1);
    }
    make_tile_map<G>(wave, lane, c.tm);
#pragma unroll
    for (int ks = 0; ks < 4; ks++) c.bp[ks] = co < 9 ? p_policy[co * CH + ks * 4 + kq] : 0.f;
    c.p_bias = p_policy[9 * CH];
    c.wv = c.p_value[co];
    c.v_bias = c.p_value[CH];
}

// One position by the whole workgroup.  in: its raw planes float [2][SIZE] (any address space);
// *value_out and logp_row[SIZE] receive the outputs.  F, W: the two activation buffers (halos zero),
// red [16], logit [SIZE]: LDS scratch.  Starts without a barrier (W's input planes are free at that point)
// and ends without one: the caller synchronises before it reads logp_row through another wavefront.
template <class G>
__device__ __forceinline__ void team_position(const TeamCtx<G> &c, float *F, float *W, float *red, float *logit, const float *in,
                                              float *value_out, float *logp_row, int tid, int wave, int lane)
{
    const int front = c.front, pol = c.pol, val = c.val, nl = c.nl;
    // Execution order: trunk, VALUE branch, policy branch (the policy head then finds the
    // trunk buffer free for its partial planes); parameters of execution step e:
    auto layer_params = [&](int e) {
        const int idx = e < front - 1 ? e : (e < front - 1 + val ? e + pol : e - val);
        return c.p_layers + (size_t)idx * LAYER_FLOATS;
    };
    // raw planes into channels 0..1 of W (its other channels may hold interior values of
    // the previous position: the 2-plane layer never reads them)
    for (int i = tid; i < CIN * G::SIZE; i += G::NTHR) {
        const int ch = i / G::SIZE, p = i - ch * G::SIZE;
        W[ch * G::PS + pix_index<G::S>(p)] = in[i];
    }
    // The parameters of the next convolution are requested one layer ahead, so
    // their L2 latency runs under the current layer's matrix instructions.
    LayerRegs<36> wq, wn;
    if (nl > 0) load_layer<36>(layer_params(0), lane, wq);
    __syncthreads();
    conv_layer<G, 5, MODE_WRITE>(W, F, c.w0, c.off0, c.tm, wave, lane);
    __syncthreads();
    float vpart = 0.f; // this lane's share of the value head's sum
    bool v_done = false;
    for (int e = 0; e < nl; e++) {
        if (e + 1 < nl) load_layer<36>(layer_params(e + 1), lane, wn);
        // trunk: F in place.  A branch: F -> W, then W in place.
        const bool trunk = e < front - 1;
        const int j = trunk ? 0 : (e < front - 1 + val ? e - (front - 1) : e - (front - 1) - val);
        const float *src = (trunk || j == 0) ? F : W;
        float *dst = trunk ? F : W;
        if (val > 0 && e == front - 1 + val - 1) {
            // last layer of the value branch: only the per-channel board sums are needed
            vpart = c.wv * conv_layer<G, 36, MODE_VALUE_SUM>(src, dst, wq, c.off0, c.tm, wave, lane);
            v_done = true;
        } else {
            conv_layer<G, 36, MODE_WRITE>(src, dst, wq, c.off0, c.tm, wave, lane);
        }
        __syncthreads();
        if (e + 1 < nl) wq = wn;
    }
    if (!v_done) { // no value branch: the head reads the trunk output
        for (int p = tid; p < G::SIZE; p += G::NTHR) {
            const int ctr = pix_index<G::S>(p);
#pragma unroll
            for (int ci = 0; ci < CH; ci++) vpart += c.p_value[ci] * F[ci * G::PS + ctr];
        }
        __syncthreads(); // the policy head's partial planes may go into F
    }
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) vpart += __shfl_xor(vpart, o);
    // ---- policy head: partial planes on the matrix cores, gather + log-softmax
    const float *hsrc = pol > 0 ? W : F;
    float *tb = pol > 0 ? F : W;
    policy_taps_layer<G>(hsrc, tb, c.bp, c.tm, wave, lane);
    if (lane == 0) red[wave] = vpart;
    __syncthreads();
    if (tid == 0) { // ---- value head: board average of the 1x1 convolution + bias, tanh
        float vt = 0.f;
#pragma unroll
        for (int w = 0; w < G::NW; w++) vt += red[w];
        *value_out = tanhf(vt / (float)G::SIZE + c.v_bias);
    }
    __syncthreads(); // red is reused by the softmax
    policy_softmax<G>(tb, c.p_bias, red, logit, logp_row, tid, wave, lane);
    // No barrier here: every read of tb's planes precedes the two barriers inside the
    // softmax, and the next position first writes W's input planes, which nobody reads
    // at this point (tb == W only when there is no policy branch, and then the gather
    // is over for every wavefront that has left the softmax).
}

} // namespace aznet

#endif

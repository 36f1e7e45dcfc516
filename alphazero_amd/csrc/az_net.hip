// Batched policy/value network evaluation for the stepped search path: the
// reference's `simplerNN` (src/model.py:9-44; called once per expansion with
// batch 1 at mcts/src/cmcts.cpp:206-231) for ALL leaf positions of a search
// step in one launch, inference mode (batch-norm folded to scale/shift).
//
// gfx950 design: a workgroup of four wavefronts owns one position at a time.
// The 16-channel activations of the whole net stay in LDS (three planes-major
// buffers with a zero halo), so HBM sees only the input planes and the outputs.
// Every 3x3 convolution is an implicit GEMM on the f32-input matrix cores
// (v_mfma_f32_16x16x4_f32: exact f32 products, f32 accumulate): M = 16 pixels of
// one board row (boards wider than 16: 16 consecutive pixels in row-major order,
// see Geo::FLAT), N = the 16 output channels, K = 4 input channels of one tap;
// each wavefront carries four independent accumulators (four rows), which is
// what the 40-cycle dependent latency of that instruction needs.  The 1-channel
// heads run on the VALU.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/azmcts.h"

namespace {

#ifndef AZ_NET_FORCE_NW
#define AZ_NET_FORCE_NW 0
#endif
constexpr int CH = 16;  // conv channels of the reference configuration (config.py:15)
constexpr int CIN = 2;  // input planes (config.py:12)
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int S> struct Geo {
    static constexpr int SIZE = S * S;
    static constexpr int ROW = S + 2;               // one halo column on each side
    static constexpr int PLANE = ROW * ROW;
    static constexpr int TPR = (S + 15) / 16;       // 16-pixel tiles per board row
    // Tiles of one row each keep every operand read inside a row (fewer LDS bank
    // conflicts: 16 % faster at 15x15), but a row of 17..20 pixels needs two tiles,
    // the second nearly empty.  There the tiles run over the board in row-major
    // order instead and cross row ends (25 instead of 40 tiles at 20x20, two rounds
    // of the four wavefronts instead of three: 19 % faster).
    static constexpr int NT_ROW = S * TPR, NT_FLAT = (SIZE + 15) / 16;
    static constexpr bool FLAT = (NT_FLAT + 15) / 16 < (NT_ROW + 15) / 16;
    static constexpr int NT = FLAT ? NT_FLAT : NT_ROW; // tiles per plane
    // plane stride: room for the over-reads of the padded tile lanes, and odd (= 1
    // mod 32) so the epilogue's writes, whose lanes step by one PLANE per output
    // channel, spread over all LDS banks (measured 5 % faster than = 16 mod 32,
    // which favours the operand reads)
    static constexpr int PS = ((PLANE + 16 * TPR + 31) / 32) * 32 + 1;
    // Wavefronts per workgroup (= per position).  Two workgroups of four share a CU
    // while three activation buffers fit twice into its 160 KB of LDS; on the
    // widest boards only one workgroup fits, which then runs eight wavefronts so
    // that every SIMD still holds two.
    static constexpr int NW = (AZ_NET_FORCE_NW != 0) ? AZ_NET_FORCE_NW : (2 * 3 * CH * PS * 4 > 150 * 1024 ? 8 : 4);
    static constexpr int TPW = 16 / NW;  // tiles (accumulators) per wavefront and round
    static constexpr int NTHR = 64 * NW;
};

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
#pragma unroll
    for (int i = 0; i < KSTEPS; i++) w.b[i] = prm[i * 64 + lane];
    w.sc = prm[KSTEPS * 64 + (lane & 15)];
    w.sh = prm[KSTEPS * 64 + 16 + (lane & 15)];
}

// one 3x3 convolution + folded batch norm + ReLU, `in` -> `out` (LDS planes).
// KSTEPS = 36 for 16 input channels (tap-major, 4 channels per step), 5 for the
// 2-plane input layer (K index = tap*2 + plane, padded with zero weights).
template <int S, int KSTEPS>
__device__ __forceinline__ void conv3x3(const float *in, float *out, const LayerRegs<KSTEPS> &w, int wave, int lane)
{
    using G = Geo<S>;
    const int co = lane & 15, kq = lane >> 4;
    constexpr int TPW = G::TPW;
    for (int tile0 = wave * TPW; tile0 < G::NT; tile0 += 16) {
        f32x4 acc[TPW];
        int base[TPW];
#pragma unroll
        for (int t = 0; t < TPW; t++) {
            acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if constexpr (G::FLAT) {
                int p = (tile0 + t) * 16 + (lane & 15);
                p = p < G::SIZE ? p : G::SIZE - 1; // lanes past the board compute a pixel nobody stores
                base[t] = (p / S + 1) * G::ROW + p % S + 1;
            } else {
                const int tile = tile0 + t < G::NT ? tile0 + t : G::NT - 1;
                const int y = tile / G::TPR, x = (tile % G::TPR) * 16 + (lane & 15);
                base[t] = (y + 1) * G::ROW + x + 1;
            }
        }
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ks++) {
            int tap, ci;
            if (KSTEPS == 36) { tap = ks >> 2; ci = (ks & 3) * 4 + kq; }
            else { const int kk = ks * 4 + kq; tap = kk < 18 ? kk >> 1 : 0; ci = kk & 1; }
            const int off = ci * G::PS + (tap / 3 - 1) * G::ROW + (tap % 3 - 1);
#pragma unroll
            for (int t = 0; t < TPW; t++)
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(in[base[t] + off], w.b[ks], acc[t], 0, 0, 0);
        }
#pragma unroll
        for (int t = 0; t < TPW; t++) {
            const int tile = tile0 + t;
            if constexpr (G::FLAT) {
                // accumulator register r of this lane: pixel tile*16 + kq*4 + r, channel co
                const int p0 = tile * 16 + kq * 4;
                int y = p0 / S, x = p0 % S;
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    if (p0 + r < G::SIZE)
                        out[co * G::PS + (y + 1) * G::ROW + x + 1] = fmaxf(acc[t][r] * w.sc + w.sh, 0.f);
                    if (++x == S) { x = 0; y++; }
                }
            } else if (tile < G::NT) {
                const int y = tile / G::TPR, x0 = (tile % G::TPR) * 16 + kq * 4;
#pragma unroll
                for (int r = 0; r < 4; r++)
                    if (x0 + r < S)
                        out[co * G::PS + (y + 1) * G::ROW + x0 + r + 1] = fmaxf(acc[t][r] * w.sc + w.sh, 0.f);
            }
        }
    }
}

// policy head: 3x3 conv to one channel + bias, log-softmax over the board
template <int S>
__device__ __forceinline__ void policy_head(const float *src, const float *p_policy, float *red, float *logit,
                                            float *logp_row, int tid, int wave, int lane)
{
    using G = Geo<S>;
    float lmax = -INFINITY;
    for (int p = tid; p < G::SIZE; p += G::NTHR) {
        const int ctr = (p / S + 1) * G::ROW + p % S + 1;
        float acc = p_policy[9 * CH];
        // (taps not unrolled: with all 144 operand reads of a pixel hoisted at once the
        // kernel runs out of registers and spills; 4 % slower at 15x15, 18 % at 20x20)
#pragma unroll 1
        for (int tap = 0; tap < 9; tap++) {
            const int off = ctr + (tap / 3 - 1) * G::ROW + (tap % 3 - 1);
#pragma unroll
            for (int ci = 0; ci < CH; ci++) acc += p_policy[tap * CH + ci] * src[ci * G::PS + off];
        }
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
    if (lane == 0) red[G::NW + wave] = lsum;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int w = 0; w < G::NW; w += 4) tot += (red[G::NW + w] + red[G::NW + w + 1]) + (red[G::NW + w + 2] + red[G::NW + w + 3]);
    const float lse = lmax + logf(tot);
    for (int p = tid; p < G::SIZE; p += G::NTHR) logp_row[p] = logit[p] - lse;
    __syncthreads(); // scratch and the branch buffers are free again
}

template <int S>
__global__ __launch_bounds__(Geo<S>::NTHR) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_net_eval(const float *boards, const int *leaf_kind, int n, const float *prm, int front, int pol,
                                                  int val, float *value_out, float *logp_out)
{
    using G = Geo<S>;
    extern __shared__ float lds[];
    float *A = lds, *B = lds + CH * G::PS, *C = lds + 2 * CH * G::PS;
    float *red = lds + 3 * CH * G::PS; // [2 * NW] reduction scratch + logits [SIZE]
    float *logit = red + 2 * G::NW;
    int *todo = reinterpret_cast<int *>(logit + G::SIZE); // [ceil(n / gridDim.x)] positions of this workgroup
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    for (int i = tid; i < 3 * CH * G::PS; i += G::NTHR) lds[i] = 0.f; // halos stay zero from here on
    __syncthreads();
    const int nl = front - 1 + pol + val; // 16->16 layers: trunk, policy branch, value branch
    const float *p_layers = prm + FIRST_FLOATS;
    const float *p_policy = p_layers + (size_t)nl * LAYER_FLOATS;
    const float *p_value = p_policy + POLICY_FLOATS;
    // Buffers: the input layer writes B, the trunk ping-pongs B <-> A and leaves
    // `front` in F; X is the other one of {A, B}.  Each branch runs F -> X -> C -> X ...
    float *F = ((front - 1) & 1) ? A : B;
    float *X = ((front - 1) & 1) ? B : A;
    const int head_at = front - 1 + pol; // 16->16 layers done when the policy head runs
    LayerRegs<5> w0;
    load_layer<5>(prm, lane, w0);

    // Positions to evaluate: every gridDim.x-th one, counted among those that need an
    // evaluation (a search that ended on a finished game, or an idle tree, needs
    // none), so the workgroups stay balanced however the skipped ones are spread.
    // Every workgroup derives its list from the codes by itself: no atomics, no
    // second kernel.
    int n_mine;
    if (leaf_kind != nullptr) {
        int *cnt = reinterpret_cast<int *>(red); // [NW] per-wavefront totals (scratch is free before the first head)
        const int per = (n + G::NTHR - 1) / G::NTHR, lo = tid * per, hi = lo + per < n ? lo + per : n;
        int c = 0;
        for (int i = lo; i < hi; i++) c += leaf_kind[i] == AZ_LEAF_EXPAND;
        int incl = c; // inclusive prefix over the wavefront, then over the four wavefronts
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int up = __shfl_up(incl, o);
            if (lane >= o) incl += up;
        }
        if (lane == 63) cnt[wave] = incl;
        __syncthreads();
        int before = incl - c, total = 0;
#pragma unroll
        for (int w = 0; w < G::NW; w++) {
            if (w < wave) before += cnt[w];
            total += cnt[w];
        }
        for (int i = lo; i < hi; i++)
            if (leaf_kind[i] == AZ_LEAF_EXPAND) {
                if (before % (int)gridDim.x == (int)blockIdx.x) todo[before / (int)gridDim.x] = i;
                before++;
            }
        n_mine = total > (int)blockIdx.x ? (total - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x : 0;
        __syncthreads();
    } else {
        n_mine = n > (int)blockIdx.x ? (n - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x : 0;
    }
    for (int it = 0; it < n_mine; it++) {
        const int pos = leaf_kind != nullptr ? todo[it] : (int)blockIdx.x + it * (int)gridDim.x;
        // raw planes into channels 0..1 of A (channels 2..15 of A may hold stale
        // interior values of the previous position: the 2-plane layer never reads them)
        for (int i = tid; i < CIN * G::SIZE; i += G::NTHR) {
            const int c = i / G::SIZE, p = i - c * G::SIZE;
            A[c * G::PS + (p / S + 1) * G::ROW + p % S + 1] = boards[(size_t)pos * CIN * G::SIZE + i];
        }
        // The parameters of the next convolution are requested one layer ahead, so
        // their L2 latency runs under the current layer's matrix instructions.
        LayerRegs<36> wq, wn;
        if (nl > 0) load_layer<36>(p_layers, lane, wq);
        __syncthreads();
        conv3x3<S, 5>(A, B, w0, wave, lane);
        __syncthreads();
        const float *head_src = F; // what the policy / value heads read when their branch is empty
        if (head_at == 0) policy_head<S>(F, p_policy, red, logit, logp_out + (size_t)pos * G::SIZE, tid, wave, lane);
        for (int l = 0; l < nl; l++) {
            if (l + 1 < nl) load_layer<36>(p_layers + (size_t)(l + 1) * LAYER_FLOATS, lane, wn);
            const float *src;
            float *dst;
            if (l < front - 1) { // trunk
                src = (l & 1) ? A : B;
                dst = (l & 1) ? B : A;
            } else {             // branch layer j
                const int j = l - (front - 1) - (l >= front - 1 + pol ? pol : 0);
                src = j == 0 ? F : ((j & 1) ? X : C);
                dst = (j & 1) ? C : X;
            }
            conv3x3<S, 36>(src, dst, wq, wave, lane);
            __syncthreads();
            head_src = dst;
            if (l + 1 == head_at) {
                policy_head<S>(head_src, p_policy, red, logit, logp_out + (size_t)pos * G::SIZE, tid, wave, lane);
                head_src = F; // the value branch starts from the trunk output again
            }
            if (l + 1 < nl) wq = wn;
        }
        // ---- value head: 1x1 conv + bias, board average, tanh
        float vs = 0.f;
        for (int p = tid; p < G::SIZE; p += G::NTHR) {
            const int ctr = (p / S + 1) * G::ROW + p % S + 1;
#pragma unroll
            for (int ci = 0; ci < CH; ci++) vs += p_value[ci] * head_src[ci * G::PS + ctr];
        }
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) vs += __shfl_xor(vs, o);
        if (lane == 0) red[wave] = vs;
        __syncthreads();
        if (tid == 0) {
            float vt = 0.f;
#pragma unroll
            for (int w = 0; w < G::NW; w += 4) vt += (red[w] + red[w + 1]) + (red[w + 2] + red[w + 3]);
            value_out[pos] = tanhf(vt / (float)G::SIZE + p_value[CH]);
        }
        __syncthreads();
    }
}

template <int S> int launch(const float *boards, const int *leaf_kind, int n, const float *prm, int front, int pol, int val, float *value_out,
                            float *logp_out, hipStream_t st)
{
    using G = Geo<S>;
    const size_t fixed_bytes = (size_t)(3 * CH * G::PS + 2 * G::NW + G::SIZE) * sizeof(float);
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const int per_cu = (int)(160 * 1024 / fixed_bytes) > 0 ? (int)(160 * 1024 / fixed_bytes) : 1;
    int grid = cus * per_cu;
    if (grid > n) grid = n;
    // + the workgroup's own list of positions (at most ceil(n / grid) entries)
    const size_t lds_bytes = fixed_bytes + (size_t)((n + grid - 1) / grid) * sizeof(int);
    // largest size the kernel has been opened up to, per device (the attribute is per device)
    static size_t configured[64] = {};
    if (dev < 0 || dev >= 64 || lds_bytes > configured[dev]) {
        if (hipFuncSetAttribute((const void *)k_net_eval<S>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds_bytes) != hipSuccess)
            return AZ_ERR_HIP;
        if (dev >= 0 && dev < 64) configured[dev] = lds_bytes;
    }
    hipLaunchKernelGGL(k_net_eval<S>, dim3(grid), dim3(G::NTHR), lds_bytes, st, boards, leaf_kind, n, prm, front, pol, val, value_out,
                       logp_out);
    return hipGetLastError() == hipSuccess ? AZ_OK : AZ_ERR_HIP;
}

} // namespace

extern "C" {

int az_net_param_floats(int front_layers, int policy_layers, int value_layers)
{
    if (front_layers < 1 || policy_layers < 0 || value_layers < 0) return 0;
    return FIRST_FLOATS + (front_layers - 1 + policy_layers + value_layers) * LAYER_FLOATS + POLICY_FLOATS + VALUE_FLOATS;
}

int az_net_eval(int shape, int n, const float *d_boards, const int *d_leaf_kind, const float *d_params,
                int front_layers, int policy_layers, int value_layers, float *d_value, float *d_logp, void *stream)
{
    if (n < 0 || front_layers < 1 || policy_layers < 0 || value_layers < 0) return AZ_ERR_ARG;
    if (n == 0) return AZ_OK;
    if (!d_boards || !d_params || !d_value || !d_logp) return AZ_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    switch (shape) {
#define AZ_NET_CASE(s) \
    case s: return launch<s>(d_boards, d_leaf_kind, n, d_params, front_layers, policy_layers, value_layers, d_value, d_logp, st);
        AZ_NET_CASE(5) AZ_NET_CASE(6) AZ_NET_CASE(7) AZ_NET_CASE(8) AZ_NET_CASE(9) AZ_NET_CASE(10) AZ_NET_CASE(11)
        AZ_NET_CASE(12) AZ_NET_CASE(13) AZ_NET_CASE(14) AZ_NET_CASE(15) AZ_NET_CASE(16) AZ_NET_CASE(17)
        AZ_NET_CASE(18) AZ_NET_CASE(19) AZ_NET_CASE(20)
#undef AZ_NET_CASE
    default: return AZ_ERR_ARG;
    }
}

} // extern "C"

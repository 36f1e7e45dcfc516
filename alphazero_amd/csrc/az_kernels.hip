// az_kernels.hip -- gfx950 (MI355X, CDNA4) kernels of the batched MCTS engine.
//
// Compiled once per board shape with -DAZ_S=<shape>; every size below is a
// compile-time constant of that translation unit.
//
// Execution model: ONE 64-lane wavefront owns ONE tree (workgroup = 1 wave,
// grid = n_trees).  Lane L holds actions [L*EPL, L*EPL+EPL) of whatever node
// the search stands on, so a node's statistics arrive with 16-byte vector
// loads that are contiguous across the wave, PUCT is computed lane-parallel
// and the arg-max is a DPP max-reduction plus one ballot.  Searches inside a
// tree run strictly one after another (the reference's only deterministic
// mode, threads <= 1, mcts/src/cmcts.cpp:120-122), parallelism comes from the
// thousands of trees in flight.
//
// Arithmetic contract (bit-exact against the reference, mcts/src/node.cpp):
// fp64 throughout, IEEE division and square root, NO fused multiply-add
// (-ffp-contract=off plus the pragma below), operation order as written in
// node.cpp:105-108, strict '>' arg-max so the lowest action index wins ties.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/azmcts.h"
#include "az_common.h"

#pragma clang fp contract(off)

#ifndef AZ_S
#error "build with -DAZ_S=<board edge>"
#endif

namespace {

constexpr int S = AZ_S;
constexpr int SIZE = S * S;
constexpr int EPL = (SIZE + 63) / 64; // actions per lane
constexpr int SP = 64 * EPL;          // padded actions per node
constexpr int REC_BYTES = SP * 24;
constexpr int OFF_N = 0;
constexpr int OFF_LINK = SP * 4;
constexpr int OFF_W = SP * 8;
constexpr int OFF_P = SP * 16;
constexpr int PATH_CAP = SIZE + 1;

static_assert(S >= 5 && S <= AZ_MAX_SHAPE, "shape out of range");
static_assert(EPL >= 1 && EPL <= 8, "EPL");

struct NodeRegs {
    int n[EPL];
    int link[EPL];
    double w[EPL];
    double p[EPL];
};

struct WState {
    unsigned rb;   // lane L: row (L&31) of plane (L>>5)
    unsigned occ;  // lane L: bit k = action L*EPL+k occupied
    int player;    // wave-uniform
    int move_cnt;  // wave-uniform
    float winner;  // wave-uniform
};

// ---------------------------------------------------------------- utilities
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ float unif(float v)
{
    return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v)));
}
__device__ __forceinline__ int lane_bcast_i32(int v, int src_lane) // src_lane wave-uniform
{
    return __builtin_amdgcn_readlane(v, src_lane);
}
__device__ __forceinline__ double lane_bcast_f64(double v, int src_lane)
{
    int lo = __builtin_amdgcn_readlane(__double2loint(v), src_lane);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), src_lane);
    return __hiloint2double(hi, lo);
}

template <int CTRL> __device__ __forceinline__ double dpp_f64(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xF, 0xF, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
template <int CTRL> __device__ __forceinline__ int dpp_i32(int v)
{
    return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xF, 0xF, false);
}

// max over the 64 lanes, result in every lane.  DPP butterflies inside each row
// of 16 lanes (quad_perm xor1, xor2, row_half_mirror, row_mirror), then the four
// row results are read to scalars.  No NaN ever enters (callers keep them out).
__device__ __forceinline__ double wave_max_f64(double v)
{
    v = fmax(v, dpp_f64<0xB1>(v));  // quad_perm [1,0,3,2]
    v = fmax(v, dpp_f64<0x4E>(v));  // quad_perm [2,3,0,1]
    v = fmax(v, dpp_f64<0x141>(v)); // row_half_mirror
    v = fmax(v, dpp_f64<0x140>(v)); // row_mirror
    double r0 = lane_bcast_f64(v, 0), r1 = lane_bcast_f64(v, 16);
    double r2 = lane_bcast_f64(v, 32), r3 = lane_bcast_f64(v, 48);
    return fmax(fmax(r0, r1), fmax(r2, r3));
}
__device__ __forceinline__ int wave_max_i32(int v)
{
    v = max(v, dpp_i32<0xB1>(v));
    v = max(v, dpp_i32<0x4E>(v));
    v = max(v, dpp_i32<0x141>(v));
    v = max(v, dpp_i32<0x140>(v));
    int r0 = lane_bcast_i32(v, 0), r1 = lane_bcast_i32(v, 16);
    int r2 = lane_bcast_i32(v, 32), r3 = lane_bcast_i32(v, 48);
    return max(max(r0, r1), max(r2, r3));
}

__device__ __forceinline__ char *rec_ptr(const AzParams &P, int tree, int slot)
{
    return P.nodes + ((size_t)tree * P.cap + slot) * REC_BYTES;
}

// lane-blocked vector loads / stores of one per-action array of a record
__device__ __forceinline__ void ld_i32(const char *base, int lane, int (&out)[EPL])
{
    const int *p = reinterpret_cast<const int *>(base) + lane * EPL;
    if constexpr (EPL % 4 == 0) {
#pragma unroll
        for (int j = 0; j < EPL / 4; j++) {
            int4 v = reinterpret_cast<const int4 *>(p)[j];
            out[4 * j] = v.x; out[4 * j + 1] = v.y; out[4 * j + 2] = v.z; out[4 * j + 3] = v.w;
        }
    } else if constexpr (EPL % 2 == 0) {
#pragma unroll
        for (int j = 0; j < EPL / 2; j++) {
            int2 v = reinterpret_cast<const int2 *>(p)[j];
            out[2 * j] = v.x; out[2 * j + 1] = v.y;
        }
    } else {
#pragma unroll
        for (int j = 0; j < EPL; j++) out[j] = p[j];
    }
}
__device__ __forceinline__ void st_i32(char *base, int lane, const int (&in)[EPL])
{
    int *p = reinterpret_cast<int *>(base) + lane * EPL;
    if constexpr (EPL % 4 == 0) {
#pragma unroll
        for (int j = 0; j < EPL / 4; j++)
            reinterpret_cast<int4 *>(p)[j] = make_int4(in[4 * j], in[4 * j + 1], in[4 * j + 2], in[4 * j + 3]);
    } else if constexpr (EPL % 2 == 0) {
#pragma unroll
        for (int j = 0; j < EPL / 2; j++)
            reinterpret_cast<int2 *>(p)[j] = make_int2(in[2 * j], in[2 * j + 1]);
    } else {
#pragma unroll
        for (int j = 0; j < EPL; j++) p[j] = in[j];
    }
}
__device__ __forceinline__ void ld_f64(const char *base, int lane, double (&out)[EPL])
{
    const double *p = reinterpret_cast<const double *>(base) + lane * EPL;
    if constexpr (EPL % 2 == 0) {
#pragma unroll
        for (int j = 0; j < EPL / 2; j++) {
            double2 v = reinterpret_cast<const double2 *>(p)[j];
            out[2 * j] = v.x; out[2 * j + 1] = v.y;
        }
    } else {
#pragma unroll
        for (int j = 0; j < EPL; j++) out[j] = p[j];
    }
}
__device__ __forceinline__ void st_f64(char *base, int lane, const double (&in)[EPL])
{
    double *p = reinterpret_cast<double *>(base) + lane * EPL;
    if constexpr (EPL % 2 == 0) {
#pragma unroll
        for (int j = 0; j < EPL / 2; j++)
            reinterpret_cast<double2 *>(p)[j] = make_double2(in[2 * j], in[2 * j + 1]);
    } else {
#pragma unroll
        for (int j = 0; j < EPL; j++) p[j] = in[j];
    }
}

__device__ __forceinline__ void load_node(const AzParams &P, int tree, int slot, int lane, NodeRegs &nd)
{
    const char *rec = rec_ptr(P, tree, slot);
    ld_i32(rec + OFF_N, lane, nd.n);
    ld_i32(rec + OFF_LINK, lane, nd.link);
    ld_f64(rec + OFF_W, lane, nd.w);
    ld_f64(rec + OFF_P, lane, nd.p);
}
__device__ __forceinline__ void store_node(const AzParams &P, int tree, int slot, int lane, const NodeRegs &nd)
{
    char *rec = rec_ptr(P, tree, slot);
    st_i32(rec + OFF_N, lane, nd.n);
    st_i32(rec + OFF_LINK, lane, nd.link);
    st_f64(rec + OFF_W, lane, nd.w);
    st_f64(rec + OFF_P, lane, nd.p);
}

// ------------------------------------------------------------------- rules
// Order in which State::is_end (mcts/src/state.cpp:162-267) meets a window,
// used only when BOTH planes hold a visible five (possible only through
// set_player or moves made after the game ended).  Windows are scanned in
// groups, plane 0 before plane 1 inside a group; the plane whose earliest
// group comes first is reported, plane 0 on a tie.
__device__ __noinline__ int scan_order_winner(unsigned h, unsigned v, unsigned d, unsigned an, int lane)
{
    const int r = lane & 31;
    constexpr int WN = S - 4;      // windows per line of full length
    constexpr int D0 = S * WN;     // groups in the rows/columns phase
    unsigned key = 0xFFFFFFFFu;
    auto diag_off = [](int j) { return j == 0 ? 0 : WN + 2 * ((j - 1) * WN) - (j - 1) * j; };
    for (unsigned m = h; m; m &= m - 1) {
        int c = __ffs(m) - 1;
        key = min(key, (unsigned)(r * WN + c));
    }
    for (unsigned m = v; m; m &= m - 1) {
        int c = __ffs(m) - 1;
        key = min(key, (unsigned)(c * WN + r));
    }
    for (unsigned m = d; m; m &= m - 1) {
        int c = __ffs(m) - 1;
        unsigned k = c >= r ? (unsigned)(D0 + diag_off(c - r) + r)
                            : (unsigned)(D0 + diag_off(r - c) + (WN - (r - c)) + c);
        key = min(key, k);
    }
    for (unsigned m = an; m; m &= m - 1) {
        int c = __ffs(m) - 1;
        int j = S - 1 - r - c; // >= 0 for every visible "/" window
        unsigned k = (unsigned)(D0 + diag_off(j) + r);
        if (j >= 1)
            k = min(k, (unsigned)(D0 + diag_off(j) + (WN - j) + (c - 4)));
        key = min(key, k);
    }
    // min inside each half of 32 lanes
#pragma unroll
    for (int o = 1; o < 32; o <<= 1)
        key = min(key, (unsigned)__shfl_xor((int)key, o));
    unsigned k0 = (unsigned)lane_bcast_i32((int)key, 0);
    unsigned k1 = (unsigned)lane_bcast_i32((int)key, 32);
    return k1 < k0 ? 1 : 0;
}

// State::make_move + State::is_end (mcts/src/state.cpp:81-110,152-272) on the
// wave-resident bitboards.  `a` must be a legal, wave-uniform action.
__device__ __forceinline__ void state_move(WState &s, int a, int lane)
{
    const int r = a / S, c = a - r * S;
    if (lane == s.player * 32 + r) s.rb |= 1u << c;
    if (lane == a / EPL) s.occ |= 1u << (a % EPL);
    s.player = s.player ? 0 : 1; // state.cpp:89
    s.move_cnt += 1;
    if (s.move_cnt < 9) return;  // state.cpp:158-159
    // five-in-a-row windows anchored at this lane's row: bit c of
    //   h : (r,c..c+4)   v : (r..r+4,c)   d : (r+i,c+i)   an : (r+i,c-i)
    const unsigned r0 = s.rb;
    const unsigned r1 = (unsigned)__shfl_down((int)r0, 1);
    const unsigned r2 = (unsigned)__shfl_down((int)r0, 2);
    const unsigned r3 = (unsigned)__shfl_down((int)r0, 3);
    const unsigned r4 = (unsigned)__shfl_down((int)r0, 4);
    const unsigned h = r0 & (r0 >> 1) & (r0 >> 2) & (r0 >> 3) & (r0 >> 4);
    const unsigned v = r0 & r1 & r2 & r3 & r4;
    const unsigned d = r0 & (r1 >> 1) & (r2 >> 2) & (r3 >> 3) & (r4 >> 4);
    unsigned an = r0 & (r1 << 1) & (r2 << 2) & (r3 << 3) & (r4 << 4);
    // the reference only walks "/" lines with row+col <= S-1 (state.cpp:209-210,
    // 222-223, 243-244, 257-258): keep windows whose top cell has c <= S-1-r
    const int row = lane & 31;
    an &= row < S ? ((1u << (S - row)) - 1u) : 0u;
    const unsigned long long hit = __ballot((h | v | d | an) != 0u);
    const bool f0 = (unsigned)hit != 0u, f1 = (unsigned)(hit >> 32) != 0u;
    if (f0 || f1) {
        int wp = f1 ? 1 : 0;
        if (f0 && f1) wp = scan_order_winner(h, v, d, an, lane);
        s.winner = (float)wp; // plane index of the winner (state.cpp:95-99)
    } else if (s.move_cnt == SIZE) {
        s.winner = 0.5f; // state.cpp:269-270, 100-102
    }
}

// ------------------------------------------------------------------ select
// Node::select (mcts/src/node.cpp:85-124).  Returns the action (wave-uniform)
// or -1 when no legal action has u > -inf; `wl`/`wk` = owning lane and slot.
__device__ __forceinline__ int select_action(const NodeRegs &nd, int node_n, unsigned occ, double cpuct,
                                             int lane, int &wl, int &wk)
{
    const double sq = __dsqrt_rn((double)node_n + 1e-8);
    double best = -INFINITY;
    int bk = -1;
#pragma unroll
    for (int k = 0; k < EPL; k++) {
        const int a = lane * EPL + k;
        const bool valid = (a < SIZE) && !((occ >> k) & 1u);
        const double t = (cpuct * nd.p[k]) * sq;
        double u = t;
        if (nd.n[k] != 0)
            u = nd.w[k] / (double)nd.n[k] + t / (double)(nd.n[k] + 1);
        if (valid && u > best) {
            best = u;
            bk = k;
        }
    }
    const double m = wave_max_f64(best);
    const unsigned long long mask = __ballot(bk >= 0 && best == m);
    if (mask == 0ull) return -1;
    wl = __ffsll((long long)mask) - 1;
    wk = lane_bcast_i32(bk, wl);
    return wl * EPL + wk;
}

// ------------------------------------------------------ in-kernel evaluators
__device__ __forceinline__ unsigned long long mix64(unsigned long long x)
{
    x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ull;
    x ^= x >> 27; x *= 0x94d049bb133111ebull;
    x ^= x >> 31;
    return x;
}

// value (wave-uniform) and this lane's linear priors for the position `s`
__device__ __forceinline__ float evaluate(const AzFusedArgs &A, const WState &s, int lane, float (&pr)[EPL])
{
    if (A.eval_kind == AZ_EVAL_UNIFORM) {
#pragma unroll
        for (int k = 0; k < EPL; k++) pr[k] = (lane * EPL + k < SIZE) ? A.const_prior : 0.f;
        return A.const_value;
    }
    // hash evaluator: key = seed ^ XOR_rows mix64((plane*32+row+1) << 32 | rowbits)
    unsigned long long hk = ((lane & 31) < S) ? mix64(((unsigned long long)(lane + 1) << 32) | s.rb) : 0ull;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) hk ^= (unsigned long long)__shfl_xor((long long)hk, o);
    const unsigned long long key = A.hash_seed ^ hk;
    const unsigned long long hv = mix64(key ^ 0x9e3779b97f4a7c15ull);
    const float value = (float)((int)(hv & 0xFFFF) - 32768) / 32768.0f;
#pragma unroll
    for (int k = 0; k < EPL; k++) {
        const int a = lane * EPL + k;
        const unsigned long long hi = mix64(key + (unsigned long long)(a + 1) * 0x9e3779b97f4a7c15ull);
        pr[k] = a < SIZE ? (float)((hi & 0xFFFF) + 1) * (1.0f / (65536.0f * 128.0f)) : 0.f;
    }
    return value;
}

// Node::set_prior (mcts/src/node.cpp:38-45): P = (1-eps)*prior + eps*noise.
__device__ __forceinline__ void mix_prior(const AzParams &P, const float (&pr)[EPL], bool is_log,
                                          int noise_row, int lane, double (&out)[EPL])
{
    const double prior_eps = 1. - P.eps;
    double nz[EPL];
#pragma unroll
    for (int k = 0; k < EPL; k++) nz[k] = 0.;
    if (P.noise != nullptr)
        ld_f64(reinterpret_cast<const char *>(P.noise + (size_t)noise_row * SP), lane, nz);
#pragma unroll
    for (int k = 0; k < EPL; k++) {
        const double p = is_log ? (double)expf(pr[k]) : (double)pr[k];
        out[k] = (lane * EPL + k < SIZE) ? prior_eps * p + P.eps * nz[k] : 0.;
    }
}

// writes a freshly expanded node: priors, zeroed statistics, nodeN = 0
__device__ __forceinline__ void write_new_node(const AzParams &P, int tree, int slot, int parent_slot,
                                               const double (&pp)[EPL], int lane)
{
    char *rec = rec_ptr(P, tree, slot);
    int zi[EPL];
    double zd[EPL];
#pragma unroll
    for (int k = 0; k < EPL; k++) { zi[k] = 0; zd[k] = 0.; }
    st_i32(rec + OFF_N, lane, zi);
    st_i32(rec + OFF_LINK, lane, zi);
    st_f64(rec + OFF_W, lane, zd);
    st_f64(rec + OFF_P, lane, pp);
    if (lane == 0) {
        P.node_n[(size_t)tree * P.cap + slot] = 0;
        P.parent[(size_t)tree * P.cap + slot] = parent_slot;
    }
}

__device__ __forceinline__ void load_state(const AzParams &P, int tree, int lane, WState &s)
{
    const AzTree *T = &P.trees[tree];
    s.rb = P.rows[tree * 64 + lane];
    s.occ = P.occ[tree * 64 + lane];
    s.player = uni(T->player);
    s.move_cnt = uni(T->move_cnt);
    s.winner = unif(T->winner);
}

// the raw, un-swapped planes as float [2][S][S] (the tensor of cmcts.cpp:206-210)
__device__ __forceinline__ void write_board(const WState &s, float *dst, int lane)
{
    const int row = lane & 31, plane = lane >> 5;
    if (row < S) {
        float *p = dst + plane * SIZE + row * S;
#pragma unroll
        for (int c = 0; c < S; c++) p[c] = (float)((s.rb >> c) & 1u);
    }
}

// ------------------------------------------------------------ fused simulate
// Cmcts::simulate -> worker -> search (mcts/src/cmcts.cpp:83-282) for an
// in-kernel evaluator: all n searches of one tree in one launch.  The root's
// statistics stay in registers for the whole call (they are read by every
// search) and are written back once at the end; deeper nodes stream from the
// node pool.  Virtual loss (node.cpp:118) and its undo (node.cpp:80) are two
// separately rounded fp64 operations, kept as such: w1 = W - 1.0 at select,
// W = w1 + (value + 1.0) at backup; only the final W is stored because no other
// search of this tree can observe the intermediate.
__global__ __launch_bounds__(64) void k_simulate_fused(AzParams P, AzFusedArgs A)
{
    const int tree = blockIdx.x;
    const int lane = threadIdx.x;
    AzTree *T = &P.trees[tree];
    if (uni(T->active) == 0 || uni(T->err) != 0) return;

    __shared__ int s_node[PATH_CAP];
    __shared__ int s_act[PATH_CAP];
    __shared__ double s_w1[PATH_CAP];

    WState root_state;
    load_state(P, tree, lane, root_state);
    int root = uni(T->root);
    int free_top = uni(T->free_top);
    int n_order = uni(T->n_order);
    int noise_row = 0;
    int err = 0;
    const size_t tbase = (size_t)tree * P.cap;
    unsigned long long n_sims = 0, n_exp = 0, n_term = 0, n_sel = 0;

    NodeRegs R;
    int root_n = 0;
    if (root >= 0) {
        load_node(P, tree, root, lane, R);
        root_n = uni(P.node_n[tbase + root]);
    }
    // next free slot is fetched one expansion ahead so its latency is hidden
    // (kept in a vector register: the wait lands at its first use, not here)
    int next_free = free_top > 0 ? P.free_stack[tbase + free_top - 1] : -1;

    for (int i = 0; i < A.n_total && err == 0; ++i) {
        if (A.per_worker > 0 && i % A.per_worker == 0) noise_row = 0; // new worker, new generator (cmcts.cpp:132-133)
        WState s = root_state;

        if (root < 0) {
            // first search on a fresh root only expands it (cmcts.cpp:190-243)
            if (free_top == 0) { err = AZ_ERR_POOL_FULL; break; }
            if (P.noise != nullptr && noise_row >= P.noise_rows) { err = AZ_ERR_NOISE_ROWS; break; }
            const int slot = uni(next_free);
            free_top--;
            if (lane == 0) P.order[tbase + n_order] = slot;
            n_order++;
            next_free = free_top > 0 ? P.free_stack[tbase + free_top - 1] : -1;
            float pr[EPL];
            (void)evaluate(A, s, lane, pr);
            mix_prior(P, pr, false, noise_row, lane, R.p);
            if (P.noise != nullptr) noise_row++;
#pragma unroll
            for (int k = 0; k < EPL; k++) { R.n[k] = 0; R.link[k] = 0; R.w[k] = 0.; }
            root_n = 0;
            root = slot;
            if (lane == 0) P.parent[tbase + slot] = -1;
            n_exp++; n_sims++;
            continue;
        }

        // level 0: the root, in registers
        int wl, wk;
        int a = select_action(R, root_n, s.occ, P.cpuct, lane, wl, wk);
        if (a < 0) { err = AZ_ERR_NO_ACTION; break; }
        n_sel++;
        int my_link = 0;
#pragma unroll
        for (int k = 0; k < EPL; k++)
            if (lane == wl && k == wk) {
                R.n[k] += 1;
                R.w[k] -= 1.; // virtual loss (node.cpp:118)
                my_link = R.link[k];
            }
        root_n += 1;
        int link = lane_bcast_i32(my_link, wl);
        const int root_l = wl, root_k = wk;
        state_move(s, a, lane);

        int depth = 0;           // path entries below the root
        int parent_slot = root;  // node that will own the link to a new child
        int parent_a = a;
        float value = 0.f;
        for (;;) {
            if (s.winner != -1.f) { // cmcts.cpp:253-267: no node for a finished game
                value = (s.winner == 0.f || s.winner == 1.f) ? 1.f : 0.f;
                n_term++;
                break;
            }
            if (link == 0) { // Node::next_node creates the child (node.cpp:126-134), the loop expands it
                if (free_top == 0) { err = AZ_ERR_POOL_FULL; break; }
                if (P.noise != nullptr && noise_row >= P.noise_rows) { err = AZ_ERR_NOISE_ROWS; break; }
                const int slot = uni(next_free);
                free_top--;
                if (lane == 0) P.order[tbase + n_order] = slot;
                n_order++;
                next_free = free_top > 0 ? P.free_stack[tbase + free_top - 1] : -1;
                float pr[EPL];
                const float v = evaluate(A, s, lane, pr);
                double pp[EPL];
                mix_prior(P, pr, false, noise_row, lane, pp);
                if (P.noise != nullptr) noise_row++;
                write_new_node(P, tree, slot, parent_slot, pp, lane);
                if (depth == 0) {
#pragma unroll
                    for (int k = 0; k < EPL; k++)
                        if (lane == root_l && k == root_k) R.link[k] = slot + 1;
                } else if (lane == 0) {
                    reinterpret_cast<int *>(rec_ptr(P, tree, parent_slot) + OFF_LINK)[parent_a] = slot + 1;
                }
                value = -v; // cmcts.cpp:228
                n_exp++;
                break;
            }
            const int cur = link - 1;
            NodeRegs nd;
            load_node(P, tree, cur, lane, nd);
            const int nn = uni(P.node_n[tbase + cur]);
            a = select_action(nd, nn, s.occ, P.cpuct, lane, wl, wk);
            if (a < 0) { err = AZ_ERR_NO_ACTION; break; }
            n_sel++;
            int my_n = 0;
            double my_w = 0.;
            my_link = 0;
#pragma unroll
            for (int k = 0; k < EPL; k++)
                if (k == wk) { my_n = nd.n[k]; my_w = nd.w[k]; my_link = nd.link[k]; }
            if (lane == wl) reinterpret_cast<int *>(rec_ptr(P, tree, cur) + OFF_N)[a] = my_n + 1;
            if (lane == 0) P.node_n[tbase + cur] = nn + 1;
            const double w1 = lane_bcast_f64(my_w, wl) - 1.; // virtual loss
            link = lane_bcast_i32(my_link, wl);
            s_node[depth] = cur;
            s_act[depth] = a;
            s_w1[depth] = w1;
            depth++;
            parent_slot = cur;
            parent_a = a;
            state_move(s, a, lane);
        }
        if (err != 0) break;
        // backup (cmcts.cpp:271-279, node.cpp:75-83): deepest first, sign flips
        for (int d = depth - 1; d >= 0; --d) {
            if (lane == 0)
                reinterpret_cast<double *>(rec_ptr(P, tree, s_node[d]) + OFF_W)[s_act[d]] =
                    s_w1[d] + ((double)value + 1.);
            value = -value;
        }
#pragma unroll
        for (int k = 0; k < EPL; k++)
            if (lane == root_l && k == root_k) R.w[k] += (double)value + 1.;
        n_sims++;
    }

    if (root >= 0) {
        store_node(P, tree, root, lane, R);
        if (lane == 0) P.node_n[tbase + root] = root_n;
    }
    if (lane == 0) {
        T->root = root;
        T->free_top = free_top;
        T->n_order = n_order;
        T->noise_row = noise_row;
        if (err) T->err = err;
        T->stats[0] += n_sims;
        T->stats[1] += n_exp;
        T->stats[2] += n_term;
        T->stats[3] += n_sel;
    }
}

// --------------------------------------------------------------- step: select
// Cmcts::search down to the leaf (mcts/src/cmcts.cpp:189-268) for an external
// evaluator.  Virtual loss is stored (node.cpp:115-118) because the backup
// runs in a later launch.
__global__ __launch_bounds__(64) void k_select(AzParams P, float *d_boards, int *d_leaf_kind)
{
    const int tree = blockIdx.x;
    const int lane = threadIdx.x;
    AzTree *T = &P.trees[tree];
    if (uni(T->active) == 0 || uni(T->err) != 0) {
        if (lane == 0) { T->leaf_kind = AZ_LEAF_NONE; d_leaf_kind[tree] = AZ_LEAF_NONE; }
        return;
    }
    WState s;
    load_state(P, tree, lane, s);
    const size_t tbase = (size_t)tree * P.cap;
    int *pn = P.path_node + (size_t)tree * P.path_cap;
    int *pa = P.path_act + (size_t)tree * P.path_cap;
    int cur = uni(T->root);
    int depth = 0, kind = AZ_LEAF_EXPAND, err = 0;
    float leaf_value = 0.f;
    unsigned long long n_sel = 0;
    while (cur >= 0) {
        NodeRegs nd;
        load_node(P, tree, cur, lane, nd);
        const int nn = uni(P.node_n[tbase + cur]);
        int wl, wk;
        const int a = select_action(nd, nn, s.occ, P.cpuct, lane, wl, wk);
        if (a < 0) { err = AZ_ERR_NO_ACTION; break; }
        n_sel++;
        int my_link = 0;
#pragma unroll
        for (int k = 0; k < EPL; k++)
            if (lane == wl && k == wk) {
                char *rec = rec_ptr(P, tree, cur);
                reinterpret_cast<int *>(rec + OFF_N)[a] = nd.n[k] + 1;
                reinterpret_cast<double *>(rec + OFF_W)[a] = nd.w[k] - 1.;
                my_link = nd.link[k];
            }
        if (lane == 0) {
            P.node_n[tbase + cur] = nn + 1;
            pn[depth] = cur;
            pa[depth] = a;
        }
        depth++;
        const int link = lane_bcast_i32(my_link, wl);
        state_move(s, a, lane);
        if (s.winner != -1.f) {
            kind = AZ_LEAF_TERMINAL;
            leaf_value = (s.winner == 0.f || s.winner == 1.f) ? 1.f : 0.f;
            break;
        }
        cur = link - 1; // -1 ends the walk: the child does not exist yet
    }
    if (err) kind = AZ_LEAF_NONE;
    if (kind == AZ_LEAF_EXPAND) write_board(s, d_boards + (size_t)tree * 2 * SIZE, lane);
    if (lane == 0) {
        T->leaf_kind = kind;
        T->path_len = depth;
        T->leaf_value = leaf_value;
        T->stats[3] += n_sel;
        if (err) T->err = err;
        d_leaf_kind[tree] = kind;
    }
}

// ------------------------------------------------------ step: expand + backup
// Node::set_prior + the backup loop (mcts/src/node.cpp:26-47,75-83;
// cmcts.cpp:228-231,271-279) for the step prepared by k_select.
__global__ __launch_bounds__(64) void k_expand_backup(AzParams P, const float *d_value, const float *d_prior,
                                                       int prior_is_log)
{
    const int tree = blockIdx.x;
    const int lane = threadIdx.x;
    AzTree *T = &P.trees[tree];
    const int kind = uni(T->leaf_kind);
    if (kind == AZ_LEAF_NONE) return;
    const size_t tbase = (size_t)tree * P.cap;
    const int *pn = P.path_node + (size_t)tree * P.path_cap;
    const int *pa = P.path_act + (size_t)tree * P.path_cap;
    const int depth = uni(T->path_len);
    int err = 0;
    float value;
    if (kind == AZ_LEAF_EXPAND) {
        int free_top = uni(T->free_top);
        int n_order = uni(T->n_order);
        int noise_row = uni(T->noise_row);
        if (free_top == 0) err = AZ_ERR_POOL_FULL;
        else if (P.noise != nullptr && noise_row >= P.noise_rows) err = AZ_ERR_NOISE_ROWS;
        if (err) {
            if (lane == 0) { T->err = err; T->leaf_kind = AZ_LEAF_NONE; }
            return;
        }
        const int slot = uni(P.free_stack[tbase + free_top - 1]);
        float pr[EPL];
#pragma unroll
        for (int k = 0; k < EPL; k++) {
            const int a = lane * EPL + k;
            pr[k] = a < SIZE ? d_prior[(size_t)tree * SIZE + a] : 0.f;
        }
        double pp[EPL];
        mix_prior(P, pr, prior_is_log != 0, noise_row, lane, pp);
        const int parent_slot = depth > 0 ? uni(pn[depth - 1]) : -1;
        write_new_node(P, tree, slot, parent_slot, pp, lane);
        if (lane == 0) {
            if (depth > 0)
                reinterpret_cast<int *>(rec_ptr(P, tree, parent_slot) + OFF_LINK)[pa[depth - 1]] = slot + 1;
            else
                T->root = slot;
            P.order[tbase + n_order] = slot;
            T->free_top = free_top - 1;
            T->n_order = n_order + 1;
            if (P.noise != nullptr) T->noise_row = noise_row + 1;
            T->stats[1] += 1;
        }
        value = -d_value[tree]; // cmcts.cpp:228
    } else {
        value = unif(T->leaf_value);
        if (lane == 0) T->stats[2] += 1;
    }
    // backup: lane d handles path level d (independent addresses)
    for (int d0 = 0; d0 < depth; d0 += 64) {
        const int d = d0 + lane;
        if (d < depth) {
            // sign at level d: value flips once per level walking up from the leaf
            const float vd = ((depth - 1 - d) & 1) ? -value : value;
            double *w = reinterpret_cast<double *>(rec_ptr(P, tree, pn[d]) + OFF_W) + pa[d];
            *w += (double)vd + 1.; // node.cpp:80
        }
    }
    if (lane == 0) {
        T->leaf_kind = AZ_LEAF_NONE;
        T->stats[0] += 1;
    }
}

// ------------------------------------------------------------------ make_move
// Cmcts::make_move (mcts/src/cmcts.cpp:284-294): apply the move to the state,
// keep the chosen child's subtree (Node::make_move, node.cpp:147-158) and free
// every other slot.  Liveness is found in ONE pass over the allocation-ordered
// slot list: a parent always precedes its children, so a slot is kept iff it is
// the new root or its parent is already marked (marks live in LDS, one bit per
// slot); parents inside the current 64-slot chunk are resolved by iterating the
// chunk to a fixed point.  The list is compacted in place and dead slots go
// back on the free stack.
__global__ __launch_bounds__(64) void k_make_move(AzParams P, const int *d_actions)
{
    extern __shared__ unsigned s_marks[];
    const int tree = blockIdx.x;
    const int lane = threadIdx.x;
    AzTree *T = &P.trees[tree];
    const int a = uni(d_actions[tree]);
    if (a < 0) return;
    WState s;
    load_state(P, tree, lane, s);
    // State::is_valid (state.cpp:29-37)
    bool ok = a < SIZE;
    if (ok) {
        const unsigned long long occupied = __ballot(lane == a / EPL && ((s.occ >> (a % EPL)) & 1u));
        ok = occupied == 0ull;
    }
    if (!ok) {
        if (lane == 0 && T->err == 0) T->err = AZ_ERR_INVALID_MOVE;
        return;
    }
    state_move(s, a, lane);
    P.rows[tree * 64 + lane] = s.rb;
    P.occ[tree * 64 + lane] = s.occ;

    const size_t tbase = (size_t)tree * P.cap;
    const int root = uni(T->root);
    int new_root = -1;
    if (root >= 0) new_root = uni(reinterpret_cast<const int *>(rec_ptr(P, tree, root) + OFF_LINK)[a]) - 1;
    const int n = uni(T->n_order);
    int free_top = uni(T->free_top);
    int kept = 0;
    const int words = (P.cap + 31) / 32;
    for (int i = lane; i < words; i += 64) s_marks[i] = 0u;
    __builtin_amdgcn_s_waitcnt(0xC07F); // lgkmcnt(0): zeroing lands before the marks are read
    const unsigned long long lt = (1ull << lane) - 1ull;
    for (int c0 = 0; c0 < n; c0 += 64) {
        const int i = c0 + lane;
        const int idx = i < n ? P.order[tbase + i] : -1;
        const int par = idx >= 0 ? P.parent[tbase + idx] : -1;
        bool m = false;
        for (;;) {
            bool nm = m;
            if (!m && idx >= 0) {
                if (idx == new_root) nm = true;
                else if (par >= 0 && new_root >= 0) nm = (s_marks[par >> 5] >> (par & 31)) & 1u;
            }
            if (nm && !m) atomicOr(&s_marks[idx >> 5], 1u << (idx & 31));
            const unsigned long long changed = __ballot(nm != m);
            m = nm;
            if (changed == 0ull) break;
        }
        const unsigned long long keep = __ballot(m);
        const unsigned long long dead = __ballot(idx >= 0 && !m);
        if (m) P.order[tbase + kept + __popcll(keep & lt)] = idx;
        if (idx >= 0 && !m) {
            P.free_stack[tbase + free_top + __popcll(dead & lt)] = idx;
            P.parent[tbase + idx] = -2;
        }
        kept += __popcll(keep);
        free_top += __popcll(dead);
    }
    if (lane == 0) {
        if (new_root >= 0) P.parent[tbase + new_root] = -1;
        T->root = new_root;
        T->n_order = kept;
        T->free_top = free_top;
        T->player = s.player;
        T->move_cnt = s.move_cnt;
        T->winner = s.winner;
    }
}

// Cmcts::clear (cmcts.cpp:30-37) for the trees whose mask byte is set
__global__ __launch_bounds__(64) void k_clear(AzParams P, const unsigned char *d_mask)
{
    const int tree = blockIdx.x;
    const int lane = threadIdx.x;
    if (d_mask != nullptr && d_mask[tree] == 0) return;
    AzTree *T = &P.trees[tree];
    const size_t tbase = (size_t)tree * P.cap;
    const int n = uni(T->n_order);
    const int free_top = uni(T->free_top);
    for (int i = lane; i < n; i += 64) {
        const int idx = P.order[tbase + i];
        P.free_stack[tbase + free_top + i] = idx;
        P.parent[tbase + idx] = -2;
    }
    P.rows[tree * 64 + lane] = 0u;
    P.occ[tree * 64 + lane] = 0u;
    if (lane == 0) {
        T->root = -1;
        T->n_order = 0;
        T->free_top = free_top + n;
        T->player = 0;
        T->move_cnt = 0;
        T->winner = -1.f;
        T->err = 0;
        T->noise_row = 0;
        T->leaf_kind = AZ_LEAF_NONE;
        T->path_len = 0;
    }
}

// first-maximum arg-max of the root visit counts (pi.argmax(), src/eval.py:192)
__global__ __launch_bounds__(64) void k_select_greedy(AzParams P, int *d_actions)
{
    const int tree = blockIdx.x;
    const int lane = threadIdx.x;
    const AzTree *T = &P.trees[tree];
    const int root = uni(T->root);
    if (uni(T->active) == 0 || root < 0) {
        if (lane == 0) d_actions[tree] = -1;
        return;
    }
    int n[EPL];
    ld_i32(rec_ptr(P, tree, root) + OFF_N, lane, n);
    int best = -1, bk = 0;
#pragma unroll
    for (int k = 0; k < EPL; k++)
        if (lane * EPL + k < SIZE && n[k] > best) { best = n[k]; bk = k; }
    const int m = wave_max_i32(best);
    const unsigned long long mask = __ballot(best == m);
    const int wl = __ffsll((long long)mask) - 1;
    const int wk = lane_bcast_i32(bk, wl);
    if (lane == 0) d_actions[tree] = wl * EPL + wk;
}

// Node::counts / nodeN of every root (node.cpp:160-164) into dense arrays
__global__ __launch_bounds__(64) void k_get_counts(AzParams P, int *d_counts, int *d_root_n)
{
    const int tree = blockIdx.x;
    const int lane = threadIdx.x;
    const AzTree *T = &P.trees[tree];
    const int root = uni(T->root);
    int n[EPL];
#pragma unroll
    for (int k = 0; k < EPL; k++) n[k] = 0;
    if (root >= 0) ld_i32(rec_ptr(P, tree, root) + OFF_N, lane, n);
    if (d_counts != nullptr) {
#pragma unroll
        for (int k = 0; k < EPL; k++) {
            const int a = lane * EPL + k;
            if (a < SIZE) d_counts[(size_t)tree * SIZE + a] = n[k];
        }
    }
    if (d_root_n != nullptr && lane == 0)
        d_root_n[tree] = root >= 0 ? P.node_n[(size_t)tree * P.cap + root] : -1;
}

int launch_status()
{
    return hipGetLastError() == hipSuccess ? AZ_OK : AZ_ERR_HIP;
}

int l_simulate_fused(const AzParams &P, const AzFusedArgs &A, void *stream)
{
    hipLaunchKernelGGL(k_simulate_fused, dim3(P.n_trees), dim3(64), 0, (hipStream_t)stream, P, A);
    return launch_status();
}
int l_select(const AzParams &P, float *d_boards, int *d_leaf_kind, void *stream)
{
    hipLaunchKernelGGL(k_select, dim3(P.n_trees), dim3(64), 0, (hipStream_t)stream, P, d_boards, d_leaf_kind);
    return launch_status();
}
int l_expand_backup(const AzParams &P, const float *d_value, const float *d_prior, int is_log, void *stream)
{
    hipLaunchKernelGGL(k_expand_backup, dim3(P.n_trees), dim3(64), 0, (hipStream_t)stream, P, d_value, d_prior,
                       is_log);
    return launch_status();
}
int l_make_move(const AzParams &P, const int *d_actions, void *stream)
{
    const size_t lds = (size_t)((P.cap + 31) / 32) * 4;
    hipLaunchKernelGGL(k_make_move, dim3(P.n_trees), dim3(64), lds, (hipStream_t)stream, P, d_actions);
    return launch_status();
}
int l_clear(const AzParams &P, const unsigned char *d_mask, void *stream)
{
    hipLaunchKernelGGL(k_clear, dim3(P.n_trees), dim3(64), 0, (hipStream_t)stream, P, d_mask);
    return launch_status();
}
int l_select_greedy(const AzParams &P, int *d_actions, void *stream)
{
    hipLaunchKernelGGL(k_select_greedy, dim3(P.n_trees), dim3(64), 0, (hipStream_t)stream, P, d_actions);
    return launch_status();
}
int l_get_counts(const AzParams &P, int *d_counts, int *d_root_n, void *stream)
{
    hipLaunchKernelGGL(k_get_counts, dim3(P.n_trees), dim3(64), 0, (hipStream_t)stream, P, d_counts, d_root_n);
    return launch_status();
}

} // namespace

#define AZ_CAT2(a, b) a##b
#define AZ_CAT(a, b) AZ_CAT2(a, b)

// host-side dispatch entry of this shape (a function, so that the table of host
// launchers is never emitted into the device image)
extern "C" const AzShapeOps *AZ_CAT(az_shape_ops_, AZ_S)(void)
{
    static const AzShapeOps ops = {
        AZ_S, l_simulate_fused, l_select, l_expand_backup, l_make_move, l_clear, l_select_greedy, l_get_counts,
    };
    return &ops;
}

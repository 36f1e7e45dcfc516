// az_kernels.hip -- gfx950 (MI355X, CDNA4) kernels of the batched MCTS engine.
//
// Compiled once per board shape with -DAZ_S=<shape>; every size below is a
// compile-time constant of that translation unit.
//
// Execution model: one workgroup owns one tree (grid = n_trees).  A group of
// T = 64*WPT threads (WPT wavefronts) gives thread t the actions
// [t*E, t*E+E) of whatever node the search stands on, E = SP/T, so a node's
// statistics arrive with contiguous vector loads, PUCT is computed
// thread-parallel and the arg-max is a DPP max-reduction per wavefront, one
// ballot, and (WPT > 1) one LDS exchange + barrier across the wavefronts.
// Searches inside a tree run strictly one after another (the reference's only
// deterministic mode, threads <= 1, mcts/src/cmcts.cpp:120-122); parallelism
// comes from the trees in flight, and splitting a tree over several wavefronts
// lets each SIMD interleave several trees so node-pool latency is hidden.
//
// Arithmetic contract (bit-exact against the reference, mcts/src/node.cpp):
// fp64 throughout, IEEE division and square root, NO fused multiply-add
// (-ffp-contract=off plus the pragma below), operation order as written in
// node.cpp:105-108, strict '>' arg-max so the lowest action index wins ties.
//
// Memory discipline inside a launch: every global word is only ever written
// and later re-read by the SAME thread (the owner of that action), so no
// fence or cross-wavefront visibility is needed; wavefronts exchange data
// through LDS and s_barrier only.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/azmcts.h"
#include "az_common.h"
#include "az_net_team.h"

#pragma clang fp contract(off)

#ifndef AZ_S
#error "build with -DAZ_S=<board edge>"
#endif

namespace {

constexpr int S = AZ_S;
constexpr int SIZE = S * S;
constexpr int E4 = (SIZE + 255) / 256;                 // actions per thread at the widest group
constexpr int WPT_MAX = (SIZE + 64 * E4 - 1) / (64 * E4); // wavefronts of the widest group (<= 4)
constexpr int SP = WPT_MAX * 64 * E4;                  // padded actions per node
constexpr int EPL = SP / 64;                           // actions per lane for one-wavefront kernels
constexpr int REC_BYTES = SP * 24;
constexpr int OFF_N = 0;
constexpr int OFF_LINK = SP * 4;
constexpr int OFF_W = SP * 8;
constexpr int OFF_P = SP * 16;
constexpr int PATH_CAP = SIZE + 1;

static_assert(S >= 5 && S <= AZ_MAX_SHAPE, "shape out of range");
static_assert(WPT_MAX >= 1 && WPT_MAX <= 4 && EPL <= 8, "geometry");

template <int E> struct NodeRegs {
    int n[E];
    int link[E];
    double w[E];
    double p[E];
};

struct WState {
    unsigned rb;   // lane L of EVERY wavefront: row (L&31) of plane (L>>5)
    unsigned occ;  // thread t: bit k = action t*E+k occupied
    int player;    // uniform
    int move_cnt;  // uniform
    float winner;  // uniform
};

// outcome of one select: the chosen action and that action's statistics
struct Pick {
    int a;      // -1: no legal action with u > -inf
    int n;      // N[a] before the increment
    int link;   // LINK[a]
    double w;   // W[a] before the virtual loss
};

struct Slot { // LDS exchange record of one wavefront
    double m;
    double w;
    int a;
    int n;
    int link;
    int pad;
};

// ---------------------------------------------------------------- utilities
// Branch hints.  With one wavefront per SIMD a taken branch costs about 33
// cycles and a fall-through one about 13 (measured, tools/probes/issue_probe.hip),
// so rare blocks should be laid out off the straight path.
#define AZ_LIKELY(x) __builtin_expect(!!(x), 1)
#define AZ_UNLIKELY(x) __builtin_expect(!!(x), 0)
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
// Thread and position inside the thread that hold action a (thread-blocked, E
// actions per thread); unsigned arithmetic: a shift and a mask for E a power of
// two, without the sign fix-up of an int division (an action is never negative
// where these are used; a negative one matches no thread).
template <int E> __device__ __forceinline__ int owner_of(int a) { return (int)((unsigned)a / (unsigned)E); }
template <int E> __device__ __forceinline__ int slot_of(int a) { return (int)((unsigned)a % (unsigned)E); }
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

// Lane permutation inside a row of 16 lanes.  Every lane has a source, so the
// `old` operand is never used; written this way (old = 0, bound_ctrl) the
// compiler folds the move into the consuming v_max_*_dpp.
template <int CTRL> __device__ __forceinline__ int dpp_i32(int v)
{
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true);
}

// Row-to-row steps of a wavefront max reduction: row_bcast:15 hands lane 15 of
// each row to the next row (applied to rows 1 and 3), row_bcast:31 hands lane 31
// to rows 2 and 3; rows outside the row mask keep their value, so lane 63 ends up
// with the maximum of the four rows.  Written as the two instructions themselves:
// through the dpp intrinsic each step compiles to copy + move + max.  (s_nop 1:
// a DPP operand needs two wait states after the VALU write of its source.)
__device__ __forceinline__ int rows_max_i32(int v)
{
    asm("s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf"
        : "+v"(v));
    return v;
}
__device__ __forceinline__ unsigned rows_max_u32(unsigned v)
{
    asm("s_nop 1\n\tv_max_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "s_nop 1\n\tv_max_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf"
        : "+v"(v));
    return v;
}

// max over the 64 lanes, result wave-uniform.  DPP butterflies inside each row
// of 16 lanes (quad_perm xor1, xor2, row_half_mirror, row_mirror), two
// row broadcasts that gather the four row results in lane 63, one v_readlane.
__device__ __forceinline__ int wave_max_i32(int v)
{
    v = max(v, dpp_i32<0xB1>(v));  // quad_perm [1,0,3,2]
    v = max(v, dpp_i32<0x4E>(v));  // quad_perm [2,3,0,1]
    v = max(v, dpp_i32<0x141>(v)); // row_half_mirror
    v = max(v, dpp_i32<0x140>(v)); // row_mirror
    return lane_bcast_i32(rows_max_i32(v), 63);
}
__device__ __forceinline__ unsigned wave_max_u32(unsigned v)
{
    v = max(v, (unsigned)dpp_i32<0xB1>((int)v));
    v = max(v, (unsigned)dpp_i32<0x4E>((int)v));
    v = max(v, (unsigned)dpp_i32<0x141>((int)v));
    v = max(v, (unsigned)dpp_i32<0x140>((int)v));
    return (unsigned)lane_bcast_i32((int)rows_max_u32(v), 63);
}

// Lanes holding the maximum of v over the wavefront (those with `ok`), and the
// maximum itself.  The fp64 order is mapped to an integer order (high word
// signed, low word unsigned, both flipped for negatives) so each butterfly step
// is one 32-bit DPP max instead of moves plus a 64-bit max: first the high
// words, then (if need be) the low words among the lanes that tie on the high word.
// v + 0.0 folds -0.0 onto +0.0, which compare equal as doubles; no NaN enters.
__device__ __forceinline__ unsigned long long wave_argmax_f64(double v, bool ok, double &m)
{
    v = v + 0.0;
    const int hi = __double2hiint(v), lo = __double2loint(v);
    const int sg = hi >> 31;
    const int khi = hi ^ (sg & 0x7fffffff);
    const unsigned klo = (unsigned)(lo ^ sg);
    const int mhi = wave_max_i32(khi);
    // The lanes that tie on the high word nearly always share the low word too (a
    // single lane, or scores that are equal bit for bit): then they all hold the
    // maximum and the second reduction is skipped.
    const unsigned long long thi = __ballot(khi == mhi); // never empty
    unsigned mlo = (unsigned)lane_bcast_i32((int)klo, __ffsll((long long)thi) - 1);
    unsigned long long top = thi;
    if (AZ_UNLIKELY((thi & ~__ballot(klo == mlo)) != 0ull)) {
        mlo = wave_max_u32(khi == mhi ? klo : 0u);
        top = __ballot(khi == mhi && klo == mlo);
    }
    const int msg = mhi >> 31;
    m = __hiloint2double(mhi ^ (msg & 0x7fffffff), (int)mlo ^ msg);
    return top & __ballot(ok);
}

// node arena of one tree (wave-uniform) and a record's byte offset inside it
// (az_create keeps cap * REC_BYTES below 4 GiB)
__device__ __forceinline__ char *arena_ptr(const AzParams &P, int tree)
{
    return P.nodes + (size_t)tree * P.cap * REC_BYTES;
}
__device__ __forceinline__ unsigned rec_off(int slot) { return (unsigned)slot * (unsigned)REC_BYTES; }
__device__ __forceinline__ char *rec_ptr(const AzParams &P, int tree, int slot)
{
    // 64-bit arena base (loop-invariant) + 32-bit offset inside the tree's arena
    // (az_create keeps cap * REC_BYTES below 4 GiB): one scalar multiply per record
    return P.nodes + (size_t)tree * P.cap * REC_BYTES + (unsigned)slot * (unsigned)REC_BYTES;
}

// thread-blocked vector loads / stores of one per-action array of a record
// (base + one zero-extended 32-bit byte offset: with a wave-uniform base this
// selects the scalar-base + vector-offset addressing form, one VALU add per access)
template <int E> __device__ __forceinline__ void ld_i32(const char *base, unsigned off, unsigned tid, int (&out)[E])
{
    const int *p = reinterpret_cast<const int *>(base + (off + tid * (unsigned)(E * 4)));
    if constexpr (E % 4 == 0) {
#pragma unroll
        for (int j = 0; j < E / 4; j++) {
            int4 v = reinterpret_cast<const int4 *>(p)[j];
            out[4 * j] = v.x; out[4 * j + 1] = v.y; out[4 * j + 2] = v.z; out[4 * j + 3] = v.w;
        }
    } else if constexpr (E % 2 == 0) {
#pragma unroll
        for (int j = 0; j < E / 2; j++) {
            int2 v = reinterpret_cast<const int2 *>(p)[j];
            out[2 * j] = v.x; out[2 * j + 1] = v.y;
        }
    } else {
#pragma unroll
        for (int j = 0; j < E; j++) out[j] = p[j];
    }
}
template <int E> __device__ __forceinline__ void st_i32(char *base, unsigned off, unsigned tid, const int (&in)[E])
{
    int *p = reinterpret_cast<int *>(base + (off + tid * (unsigned)(E * 4)));
    if constexpr (E % 4 == 0) {
#pragma unroll
        for (int j = 0; j < E / 4; j++)
            reinterpret_cast<int4 *>(p)[j] = make_int4(in[4 * j], in[4 * j + 1], in[4 * j + 2], in[4 * j + 3]);
    } else if constexpr (E % 2 == 0) {
#pragma unroll
        for (int j = 0; j < E / 2; j++)
            reinterpret_cast<int2 *>(p)[j] = make_int2(in[2 * j], in[2 * j + 1]);
    } else {
#pragma unroll
        for (int j = 0; j < E; j++) p[j] = in[j];
    }
}
template <int E> __device__ __forceinline__ void ld_f64(const char *base, unsigned off, unsigned tid, double (&out)[E])
{
    const double *p = reinterpret_cast<const double *>(base + (off + tid * (unsigned)(E * 8)));
    if constexpr (E % 2 == 0) {
#pragma unroll
        for (int j = 0; j < E / 2; j++) {
            double2 v = reinterpret_cast<const double2 *>(p)[j];
            out[2 * j] = v.x; out[2 * j + 1] = v.y;
        }
    } else {
#pragma unroll
        for (int j = 0; j < E; j++) out[j] = p[j];
    }
}
template <int E> __device__ __forceinline__ void st_f64(char *base, unsigned off, unsigned tid, const double (&in)[E])
{
    double *p = reinterpret_cast<double *>(base + (off + tid * (unsigned)(E * 8)));
    if constexpr (E % 2 == 0) {
#pragma unroll
        for (int j = 0; j < E / 2; j++)
            reinterpret_cast<double2 *>(p)[j] = make_double2(in[2 * j], in[2 * j + 1]);
    } else {
#pragma unroll
        for (int j = 0; j < E; j++) p[j] = in[j];
    }
}

// node with nodeN > 0: all four arrays; fresh node (nodeN == 0): only P exists.
// ZERO_FRESH = false leaves n/link/w of a fresh node unset for callers that never
// read them then: writing the zeros would first have to wait (s_waitcnt vmcnt(0))
// for every load still in flight into those registers, the P load just issued
// included, and the latency overlap with the rules code would be gone.
template <int E, bool ZERO_FRESH = true>
__device__ __forceinline__ void load_node(const AzParams &P, int tree, int slot, int tid, bool fresh, NodeRegs<E> &nd)
{
    const char *arena = arena_ptr(P, tree);
    const unsigned ro = rec_off(slot);
    ld_f64<E>(arena, ro + OFF_P, tid, nd.p);
    if (fresh) {
        if constexpr (ZERO_FRESH) {
#pragma unroll
            for (int k = 0; k < E; k++) { nd.n[k] = 0; nd.link[k] = 0; nd.w[k] = 0.; }
        }
    } else {
        ld_i32<E>(arena, ro + OFF_N, tid, nd.n);
        ld_i32<E>(arena, ro + OFF_LINK, tid, nd.link);
        ld_f64<E>(arena, ro + OFF_W, tid, nd.w);
    }
}
template <int E>
__device__ __forceinline__ void store_stats(const AzParams &P, int tree, int slot, int tid, const NodeRegs<E> &nd)
{
    char *arena = arena_ptr(P, tree);
    const unsigned ro = rec_off(slot);
    st_i32<E>(arena, ro + OFF_N, tid, nd.n);
    st_i32<E>(arena, ro + OFF_LINK, tid, nd.link);
    st_f64<E>(arena, ro + OFF_W, tid, nd.w);
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
// wavefront-resident bitboards.  `a` must be a legal, uniform action.  Every
// wavefront of a group keeps its own copy of the rows and runs this
// identically, so no exchange is needed.
template <int E> __device__ __forceinline__ void state_move(WState &s, int a, int tid)
{
    const int lane = tid & 63;
    const int r = (int)((unsigned)a / (unsigned)S), c = a - r * S;
    if (lane == s.player * 32 + r) s.rb |= 1u << c;
    if (tid == owner_of<E>(a)) s.occ |= 1u << (slot_of<E>(a));
    s.player = s.player ? 0 : 1; // state.cpp:89
    s.move_cnt += 1;
    if (AZ_UNLIKELY(s.move_cnt < 9)) return;  // state.cpp:158-159
    // five-in-a-row windows anchored at this lane's row: bit c of
    //   h : (r,c..c+4)   v : (r..r+4,c)   d : (r+i,c+i)   an : (r+i,c-i)
    // rows r+1..r+4 come from the next lanes: DPP wave_shl:1 (lane i reads lane
    // i+1, lane 63 reads 0), four times; rows >= S are zero, so windows never
    // leak from plane 0 into plane 1
    const unsigned r0 = s.rb;
    const unsigned r1 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)r0, 0x130, 0xF, 0xF, false);
    const unsigned r2 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)r1, 0x130, 0xF, 0xF, false);
    const unsigned r3 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)r2, 0x130, 0xF, 0xF, false);
    const unsigned r4 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)r3, 0x130, 0xF, 0xF, false);
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
    if (AZ_UNLIKELY(f0 || f1)) {
        int wp = f1 ? 1 : 0;
        // (readfirstlane: a call's result lives in a vector register, and a value
        // the compiler takes for lane-varying would turn every later search-loop
        // branch into exec-mask code)
        if (AZ_UNLIKELY(f0 && f1)) wp = uni(scan_order_winner(h, v, d, an, lane));
        s.winner = (float)wp; // plane index of the winner (state.cpp:95-99)
    } else if (AZ_UNLIKELY(s.move_cnt == SIZE)) {
        s.winner = 0.5f; // state.cpp:269-270, 100-102
    }
}

template <int E> __device__ __forceinline__ void load_state(const AzParams &P, int tree, int tid, WState &s)
{
    const AzTree *T = &P.trees[tree];
    const int lane = tid & 63;
    s.rb = P.rows[tree * 64 + lane];
    s.player = uni(T->player);
    s.move_cnt = uni(T->move_cnt);
    s.winner = unif(T->winner);
    // State::is_valid (state.cpp:29-37) for this thread's actions, from the rows
    s.occ = 0u;
#pragma unroll
    for (int k = 0; k < E; k++) {
        const int a = tid * E + k;
        const int r = a < SIZE ? a / S : 0, c = a < SIZE ? a % S : 0;
        const unsigned both = (unsigned)__shfl((int)s.rb, r) | (unsigned)__shfl((int)s.rb, 32 + r);
        if (a < SIZE && ((both >> c) & 1u)) s.occ |= 1u << k;
    }
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

// ------------------------------------------------------------------ select
// Per-thread part of Node::select (mcts/src/node.cpp:95-114): PUCT of this
// thread's actions in fp64, operation order as the reference writes it, strict
// '>' so the first maximum is kept.  `sq` = sqrt(nodeN + 1e-8).
// A thread's best action so far together with the statistics the pick hands
// on, carried through the scan by selects so that nothing has to be looked up
// by index once the winning lane is known.
struct Cand {
    double u, w;
    int k, n, link;
};
__device__ __forceinline__ void cand_init(Cand &c)
{
    c.u = -INFINITY; c.w = 0.; c.k = -1; c.n = 0; c.link = 0;
}
__device__ __forceinline__ void cand_take(Cand &c, bool gt, double u, int k, int n, int link, double w)
{
    c.u = gt ? u : c.u;
    c.k = gt ? k : c.k;
    c.n = gt ? n : c.n;
    c.link = gt ? link : c.link;
    c.w = gt ? w : c.w;
}
template <int E>
__device__ __forceinline__ void scan_exact(const NodeRegs<E> &nd, double sq, unsigned occ, double cpuct, int tid,
                                           Cand &c)
{
    cand_init(c);
#pragma unroll
    for (int k = 0; k < E; k++) {
        const int a = tid * E + k;
        const bool valid = (a < SIZE) && !((occ >> k) & 1u);
        const double t = (cpuct * nd.p[k]) * sq;
        double u = t;
        if (nd.n[k] != 0)
            u = nd.w[k] / (double)nd.n[k] + t / (double)(nd.n[k] + 1);
        cand_take(c, valid && u > c.u, u, k, nd.n[k], nd.link[k], nd.w[k]);
    }
}

// The same for the root held in registers across a whole simulate() call, with
// the loop-invariant pieces cached per action: cp = cpuct*P, np1 = (double)(N+1)
// and q = W/N.  Each cached value is the very fp64 operation the reference
// evaluates at this point (q is refreshed whenever W or N of the action
// changes), so u is bit-identical; per search only one W/N division remains.
template <int E> struct RootCache {
    double cp[E], np1[E], q[E], rinv[E];
};
// t / b correctly rounded, given rinv = RN(1/b): product, then two
// residual corrections (Markstein).  The first correction makes the quotient
// faithful, the second one rounds it correctly when b's significand is not all
// ones; b is a small integer here and t = cpuct*P*sqrt(..) is far from the
// fp64 under/overflow thresholds.  The residuals need a fused multiply-add,
// which is why they are spelled as __fma_rn under -ffp-contract=off.
__device__ __forceinline__ double div_by_cached(double t, double b, double rinv)
{
    double q = t * rinv;
    q = __fma_rn(__fma_rn(-q, b, t), rinv, q);
    q = __fma_rn(__fma_rn(-q, b, t), rinv, q);
    return q;
}
// `u` keeps every action's score (-inf for an illegal one) for the next-pick hint.
// NP1_CACHED = false: (double)(N + 1) is formed from the visit count each time instead of being carried in
// registers (the same value: the conversion of a small integer is exact).
template <int E, bool NP1_CACHED = true>
__device__ __forceinline__ void scan_root(const NodeRegs<E> &R, const RootCache<E> &C, double sq, unsigned occ,
                                          int tid, Cand &c, double (&uk)[E])
{
    cand_init(c);
#pragma unroll
    for (int k = 0; k < E; k++) {
        const int a = tid * E + k;
        const bool valid = (a < SIZE) && !((occ >> k) & 1u);
        const double np1 = NP1_CACHED ? C.np1[k] : (double)(R.n[k] + 1);
        // No N == 0 branch: there q = 0 and np1 = 1, and 0 + t/1 is t itself
        // (t >= 0), so the one expression serves both cases of node.cpp:105-108
        // and the four divisions of a thread overlap instead of running as
        // four separate masked blocks.
        const double u = C.q[k] + div_by_cached(C.cp[k] * sq, np1, C.rinv[k]);
        uk[k] = valid ? u : -INFINITY;
        cand_take(c, valid && u > c.u, u, k, R.n[k], R.link[k], R.w[k]);
    }
}

// Hint for the next search: the runner-up of this root scan (the winner's own
// visit lowers its score, the common sqrt factor moves all others alike).
// Ranked on the high word of the score only, with the low bits of that word
// replaced by the action's position in its thread so that one integer maximum
// yields both the value and whose it is; a wrong hint costs nothing but the
// early load.  Every thread ranks its actions again without the winner (only the
// winner's thread gets a new answer): branch-free.  Returns the runner-up's
// child link (0 = none) and visit count.
template <int E>
__device__ __forceinline__ void runner_up(const NodeRegs<E> &R, const double (&uk)[E], int win_a, int tid, int &link2,
                                          int &n2)
{
    constexpr int KM = E <= 2 ? 1 : E <= 4 ? 3 : 7; // bits that hold the position
    const int wk = tid == owner_of<E>(win_a) ? slot_of<E>(win_a) : -1;
    int best = INT_MIN;
#pragma unroll
    for (int k = 0; k < E; k++) {
        const int hi = __double2hiint(uk[k]);
        int key = hi ^ ((hi >> 31) & 0x7fffffff);  // integer order of the fp64 order
        key = (key & ~KM) | (KM - k);              // earlier action wins a tie
        key = k == wk ? INT_MIN : key;
        best = key > best ? key : best;
    }
    const int j = KM - (best & KM);
    int l2 = R.link[0], m2 = R.n[0];
#pragma unroll
    for (int k = 1; k < E; k++) {
        l2 = j == k ? R.link[k] : l2;
        m2 = j == k ? R.n[k] : m2;
    }
    // between threads the position bits must not count: equal scores go to the
    // lowest lane, i.e. the lowest action
    const int top = best & ~KM;
    const int mx = wave_max_i32(top);
    const unsigned long long mask = __ballot(top == mx);
    const int wl = __ffsll((long long)mask) - 1; // never empty: some lane holds the maximum
    link2 = lane_bcast_i32(l2, wl);
    n2 = lane_bcast_i32(m2, wl);
}

// Arg-max over the group of WPT wavefronts from the per-thread results:
// DPP max reduction + ballot inside a wavefront, then (WPT > 1) one
// double-buffered LDS exchange; wavefront order = action order, so the first
// strict maximum wins exactly as in the reference's scan.
template <int WPT, int E>
__device__ __forceinline__ Pick finish_pick(const Cand &c, int tid, Slot *slots, int &phase)
{
    // (a literal 0 for one wavefront per tree makes the pick provably wave-uniform,
    // so the search loop compiles to scalar branches instead of exec-mask code)
    const int lane = tid & 63, wave = WPT == 1 ? 0 : tid >> 6;
    double m;
    const unsigned long long mask = wave_argmax_f64(c.u, c.k >= 0, m);
    Pick pk;
    pk.a = -1; pk.n = 0; pk.link = 0; pk.w = 0.;
    if (AZ_LIKELY(mask != 0ull)) {
        const int wl = __ffsll((long long)mask) - 1;
        pk.a = (wave * 64 + wl) * E + lane_bcast_i32(c.k, wl);
        pk.n = lane_bcast_i32(c.n, wl);
        pk.link = lane_bcast_i32(c.link, wl);
        pk.w = lane_bcast_f64(c.w, wl);
    }
    if constexpr (WPT == 1) {
        return pk;
    } else {
        Slot *sl = slots + phase * WPT;
        if (lane == 0) {
            sl[wave].m = m; sl[wave].w = pk.w; sl[wave].a = pk.a; sl[wave].n = pk.n; sl[wave].link = pk.link;
        }
        __syncthreads();
        Pick out;
        out.a = -1; out.n = 0; out.link = 0; out.w = 0.;
        double bm = -INFINITY;
#pragma unroll
        for (int w = 0; w < WPT; w++) {
            const int aw = sl[w].a;
            const double mw = sl[w].m;
            if (aw >= 0 && (out.a < 0 || mw > bm)) {
                bm = mw; out.a = aw; out.n = sl[w].n; out.link = sl[w].link; out.w = sl[w].w;
            }
        }
        phase ^= 1;
        return out;
    }
}

// Node::select (mcts/src/node.cpp:85-124) on a node loaded from the pool
template <int WPT, int E>
__device__ __forceinline__ Pick select_pick(const NodeRegs<E> &nd, bool fresh, double sq, unsigned occ, double cpuct,
                                            int tid, Slot *slots, int &phase)
{
    Cand c;
    if (fresh) {
        // never selected from: every N and W is zero, so u = cpuct*P*sqrt(nodeN + 1e-8)
        // (the N == 0 branch of node.cpp:107-108) and the pick hands on zeros
        cand_init(c);
#pragma unroll
        for (int k = 0; k < E; k++) {
            const int a = tid * E + k;
            const bool valid = (a < SIZE) && !((occ >> k) & 1u);
            const double u = (cpuct * nd.p[k]) * sq;
            const bool gt = valid && u > c.u;
            c.u = gt ? u : c.u;
            c.k = gt ? k : c.k;
        }
    } else {
        scan_exact<E>(nd, sq, occ, cpuct, tid, c);
    }
    return finish_pick<WPT, E>(c, tid, slots, phase);
}

// sqrt(nodeN + 1e-8) (node.cpp:106,108): a table of the same fp64 operation for
// the first SQ_TAB values (nodeN is an integer), the operation itself beyond
constexpr int SQ_TAB = 2048;     // entries of the table when one wavefront has a SIMD to itself
constexpr int SQ_TAB_LEAN = 512; // ... when the launch wants three wavefronts per SIMD (LDS bounds the residency)
template <int TAB = SQ_TAB>
__device__ __forceinline__ double sqrt_from_table(int node_n, const double *tab) // tab: an LDS table of TAB entries, never null
{
    if (AZ_LIKELY(node_n < TAB)) return tab[node_n];
    return __dsqrt_rn((double)node_n + 1e-8);
}
__device__ __forceinline__ double sqrt_node_n(int node_n, const double *tab)
{
    if (AZ_LIKELY(tab != nullptr && node_n < SQ_TAB)) return tab[node_n];
    return __dsqrt_rn((double)node_n + 1e-8);
}

// ------------------------------------------------------ in-kernel evaluators
__device__ __forceinline__ unsigned long long mix64(unsigned long long x)
{
    x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ull;
    x ^= x >> 27; x *= 0x94d049bb133111ebull;
    x ^= x >> 31;
    return x;
}

// value (uniform) and this thread's linear priors for the position `s`
// EVAL: AZ_EVAL_UNIFORM / AZ_EVAL_HASH compiled in, or -1 = A.eval_kind decides.
template <int E, int EVAL = -1>
__device__ __forceinline__ float evaluate(const AzFusedArgs &A, const WState &s, int tid, float (&pr)[E])
{
    if (EVAL == AZ_EVAL_UNIFORM || (EVAL < 0 && A.eval_kind == AZ_EVAL_UNIFORM)) {
#pragma unroll
        for (int k = 0; k < E; k++) pr[k] = (tid * E + k < SIZE) ? A.const_prior : 0.f;
        return A.const_value;
    }
    // hash evaluator: key = seed ^ XOR_rows mix64((plane*32+row+1) << 32 | rowbits)
    const int lane = tid & 63;
    unsigned long long hk = ((lane & 31) < S) ? mix64(((unsigned long long)(lane + 1) << 32) | s.rb) : 0ull;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) hk ^= (unsigned long long)__shfl_xor((long long)hk, o);
    const unsigned long long key = A.hash_seed ^ hk;
    const unsigned long long hv = mix64(key ^ 0x9e3779b97f4a7c15ull);
    const float value = unif((float)((int)(hv & 0xFFFF) - 32768) / 32768.0f); // same in every lane
#pragma unroll
    for (int k = 0; k < E; k++) {
        const int a = tid * E + k;
        const unsigned long long hi = mix64(key + (unsigned long long)(a + 1) * 0x9e3779b97f4a7c15ull);
        pr[k] = a < SIZE ? (float)((hi & 0xFFFF) + 1) * (1.0f / (65536.0f * 128.0f)) : 0.f;
    }
    return value;
}

// Node::set_prior (mcts/src/node.cpp:38-45): P = (1-eps)*prior + eps*noise.
// MIX: 0 = the mix is known to be the identity (eps == 0, no table, linear
// priors), 1 = general, -1 = decided at run time.
template <int E, int MIX = -1>
__device__ __forceinline__ void mix_prior(const AzParams &P, const float (&pr)[E], bool is_log, int noise_row,
                                          int tid, double (&out)[E])
{
    if (MIX == 0 || (MIX < 0 && P.eps == 0. && P.noise == nullptr && !is_log)) {
        // (1-0)*p + 0*0 = p: the mix is the identity, skip its arithmetic
#pragma unroll
        for (int k = 0; k < E; k++) out[k] = (tid * E + k < SIZE) ? (double)pr[k] : 0.;
        return;
    }
    const double prior_eps = 1. - P.eps;
    double nz[E];
#pragma unroll
    for (int k = 0; k < E; k++) nz[k] = 0.;
    if (P.noise != nullptr) {
        ld_f64<E>(reinterpret_cast<const char *>(P.noise), (unsigned)noise_row * (unsigned)(SP * 8), tid, nz);
        // complete the load inside the branch: a wait placed after the merge would
        // also run (as vmcnt(0), i.e. behind every earlier store) when there is no noise
        __builtin_amdgcn_s_waitcnt(0x0F70);
    }
#pragma unroll
    for (int k = 0; k < E; k++) {
        const double p = is_log ? (double)expf(pr[k]) : (double)pr[k];
        out[k] = (tid * E + k < SIZE) ? prior_eps * p + P.eps * nz[k] : 0.;
    }
}

// Path of the running search: entry d lives in lane d of four vector registers
// (lane-select write, v_readlane with a scalar index), entries beyond 64 spill to LDS.
struct PathRegs {
    int node, act, w_lo, w_hi;
};
// d < 64 only; deeper entries go to LDS in code kept out of the common path
__device__ __forceinline__ void path_put(PathRegs &pr, int d, int node, int act, double w1, int lane)
{
    const bool mine = lane == d;
    pr.node = mine ? node : pr.node;
    pr.act = mine ? act : pr.act;
    pr.w_lo = mine ? __double2loint(w1) : pr.w_lo;
    pr.w_hi = mine ? __double2hiint(w1) : pr.w_hi;
}
__device__ __forceinline__ void path_get(const PathRegs &pr, int d, int &node, int &act, double &w1)
{
    node = __builtin_amdgcn_readlane(pr.node, d);
    act = __builtin_amdgcn_readlane(pr.act, d);
    w1 = __hiloint2double(__builtin_amdgcn_readlane(pr.w_hi, d), __builtin_amdgcn_readlane(pr.w_lo, d));
}

// pops the top of the tree's free stack (caller has checked free_top > 0)
__device__ __forceinline__ int pop_free(const AzParams &P, size_t tbase, int &free_top, int &batch, int &pos, int tid)
{
    if (AZ_UNLIKELY(pos == 64)) { // refill: lane j takes the j-th entry below the top
        const int lane = tid & 63;
        const int idx = free_top - 1 - lane;
        batch = idx >= 0 ? P.free_stack[tbase + idx] : -1;
        __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0), once per 64 expansions
        pos = 0;
    }
    const int slot = __builtin_amdgcn_readlane(batch, pos);
    pos++;
    free_top--;
    return slot;
}

// Diagnostic build (-DAZ_STAMPS): per-phase cycle sums of wavefront 0 of each
// tree, written to P.stamps; never compiled into the shipped library.
#ifdef AZ_STAMPS
#define STAMP_DECL unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long st_t = __builtin_readcyclecounter();
#define STAMP(i) do { unsigned long long t_ = __builtin_readcyclecounter(); st_acc[i] += t_ - st_t; st_t = t_; } while (0)
// (bits 40.. of entries 6 and 7 carry the XCC_ID and HW_ID registers: where the wavefront ran)
#define STAMP_FLUSH(tree) do { if (tid == 0 && P.stamps) { \
        st_acc[6] |= (unsigned long long)(__builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xf) << 40; \
        st_acc[7] |= (unsigned long long)(__builtin_amdgcn_s_getreg((31 << 11) | 4) & 0xffff) << 40; \
        for (int q = 0; q < 8; q++) P.stamps[(size_t)(tree) * 8 + q] = st_acc[q]; } } while (0)
// (the worker kernels run once per search step: their sums ADD to what is there, the reader takes differences)
#define STAMP_ADD(tree) do { if (lane == 0 && P.stamps) { \
        for (int q = 0; q < 8; q++) P.stamps[(size_t)(tree) * 8 + q] += st_acc[q]; } } while (0)
#define STAMP_PARAM , unsigned long long (&st_acc)[8], unsigned long long &st_t
#define STAMP_ARG , st_acc, st_t
#else
#define STAMP_DECL
#define STAMP(i) do { } while (0)
#define STAMP_FLUSH(tree) do { } while (0)
#define STAMP_ADD(tree) do { } while (0)
#define STAMP_PARAM
#define STAMP_ARG
#endif

// ------------------------------------------------------------ fused simulate
// Cmcts::simulate -> worker -> search (mcts/src/cmcts.cpp:83-282) for an
// in-kernel evaluator: all n searches of one tree in one launch.  The root's
// statistics stay in registers for the whole call (every search reads them)
// and are written back once at the end; deeper nodes stream from the node
// pool.  Virtual loss (node.cpp:118) and its undo (node.cpp:80) are two
// separately rounded fp64 operations, kept as such: w1 = W - 1.0 at select,
// W = w1 + (value + 1.0) at backup; only the final W is stored because no other
// search of this tree can observe the intermediate.
// LEAN = false: the form for launches of about one wavefront per SIMD.  The child record of the likely next
//   root pick is requested one search ahead into registers of its own (EARLY_LOAD) and the sqrt table has
//   2048 entries: 166-182 registers and 19 KB of LDS, i.e. two wavefronts per SIMD at most.
// LEAN = true: the form for launches of many trees.  No early load (24 registers less) and a table of 512
//   entries (sqrt computed beyond it): three wavefronts per SIMD.  Measured on MI355X at 15x15, uniform
//   evaluator: 1024 trees 9 % slower, 2048 trees 3 % faster, 8192 trees 5 % faster; hash evaluator (whose
//   next pick is harder to predict): 1024 trees 4 % faster, 8192 trees 21 % faster.  (A form forced down to
//   128 registers for four wavefronts per SIMD spills and is no faster than two.)
// EVAL / MIX: see evaluate() and mix_prior(); the launcher picks the instance,
// so the evaluator and noise tests leave the search loop.
#ifndef AZ_FUSED_ATTR
#define AZ_FUSED_ATTR
#endif
// FORM 2 (compact): the lean form without the cached N + 1 of the root's actions, compiled for four wavefronts per
//   SIMD (128 registers): for launches of many more trees than SIMDs, where the wavefronts in flight hide each
//   other's latencies.
template <int WPT, int EVAL, int MIX, int FORM>
__device__ __forceinline__ void simulate_fused_body(const AzParams &P, const AzFusedArgs &A)
{
    constexpr bool LEAN = FORM >= 1, COMPACT = FORM == 2;
    constexpr bool EARLY_LOAD = !LEAN;
    constexpr int TAB = LEAN ? SQ_TAB_LEAN : SQ_TAB;
    constexpr int T = 64 * WPT;
    static_assert(SP % T == 0, "group does not tile the padded action range");
    constexpr int E = SP / T;
    const int tree = blockIdx.x;
    const int tid = threadIdx.x;
    AzTree *Tr = &P.trees[tree];
    if (uni(Tr->active) == 0 || uni(Tr->err) != 0) return;
    // A tree whose nodes carry visit deficits (left by concurrent workers, see AzParams::aux) is not searched here:
    // this kernel derives every node's nodeN from its parent's visit count alone.
    if (AZ_UNLIKELY((uni(Tr->flags) & AZ_TREE_DEFICIT) != 0)) {
        if (tid == 0) Tr->err = AZ_ERR_STATE;
        return;
    }

    // overflow of the register-resident path (depth > 64); every wavefront
    // writes the same values
    constexpr int PATH_SPILL = PATH_CAP > 64 ? PATH_CAP - 64 : 1;
    __shared__ int s_node[PATH_SPILL];
    __shared__ int s_act[PATH_SPILL];
    __shared__ double s_w1[PATH_SPILL];
    PathRegs path = {0, 0, 0, 0};
    __shared__ Slot s_slots[2 * WPT];
    __shared__ double s_sqrt[TAB];
    int phase = 0;
    for (int i = tid; i < TAB; i += T) s_sqrt[i] = __dsqrt_rn((double)i + 1e-8);
    __syncthreads();

    WState root_state;
    load_state<E>(P, tree, tid, root_state);
    int root = uni(Tr->root);
    int free_top = uni(Tr->free_top);
    int n_order = uni(Tr->n_order);
    int noise_row = 0;
    int err = 0;
    const size_t tbase = (size_t)tree * P.cap;
    // (the first three are bounded by the int n_total; select calls are added once per search)
    unsigned n_sims = 0, n_exp = 0, n_term = 0;
    unsigned long long n_sel = 0;

    NodeRegs<E> R;
    RootCache<E> C;
    int root_n = 0;
    if (root >= 0) {
        load_node<E>(P, tree, root, tid, false, R); // a root is always materialised
        root_n = uni(Tr->root_n);
#pragma unroll
        for (int k = 0; k < E; k++) {
            C.cp[k] = P.cpuct * R.p[k];
            if constexpr (!COMPACT) C.np1[k] = (double)(R.n[k] + 1);
            C.rinv[k] = 1. / (double)(R.n[k] + 1);
            C.q[k] = R.n[k] != 0 ? R.w[k] / (double)R.n[k] : 0.;
        }
    }
    // The root registers must be complete BEFORE the search loop: otherwise the
    // compiler guards their first use inside the loop with s_waitcnt vmcnt(0),
    // which on every iteration also waits for the previous search's stores.
    __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0)
    // Free slots are popped from a 64-entry batch held in one vector register
    // (lane j = j-th next pop), refilled every 64 expansions: a pop is a
    // v_readlane, not a dependent memory load on the critical path.
    int fs_batch = 0, fs_pos = 64;
    STAMP_DECL
#ifdef AZ_PF_STATS
    int pf_last = 0, pf_hit = 0, pf_total = 0;
#endif

    double sq_root = sqrt_from_table<TAB>(root_n, s_sqrt); // sqrt(nodeN + 1e-8) of the root, kept one search ahead

    // A fresh root is expanded by the first search of the call, which does nothing
    // else (cmcts.cpp:190-243).  That search is taken out of the loop: inside it the
    // rewrite of every root register would be carried around each iteration.
    int worker_left = 0, first = 0;
    if (root < 0 && A.n_total > 0) {
        noise_row = 0;
        worker_left = (A.per_worker > 0 ? A.per_worker : A.n_total) - 1;
        first = 1;
        if (free_top == 0) {
            err = AZ_ERR_POOL_FULL;
        } else if (MIX != 0 && P.noise != nullptr && noise_row >= P.noise_rows) {
            err = AZ_ERR_NOISE_ROWS;
        } else {
            const int slot = pop_free(P, tbase, free_top, fs_batch, fs_pos, tid);
            if (tid == 0) { P.order[tbase + n_order] = slot; P.parent[tbase + slot] = -1; }
            n_order++;
            float pr[E];
            (void)evaluate<E, EVAL>(A, root_state, tid, pr);
            mix_prior<E, MIX>(P, pr, false, noise_row, tid, R.p);
            if (MIX != 0 && P.noise != nullptr) noise_row++;
            st_f64<E>(arena_ptr(P, tree), rec_off(slot) + OFF_P, tid, R.p);
#pragma unroll
            for (int k = 0; k < E; k++) {
                R.n[k] = 0; R.link[k] = 0; R.w[k] = 0.;
                C.cp[k] = P.cpuct * R.p[k]; C.rinv[k] = 1.; C.q[k] = 0.;
                if constexpr (!COMPACT) C.np1[k] = 1.;
            }
            root_n = 0;
            sq_root = sqrt_from_table<TAB>(0, s_sqrt);
            root = slot;
            n_exp++; n_sims++;
            __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0): noise loads into the root registers are complete
        }
    }
    // Child record of the likely next root pick, requested into registers of its own
    // while the running search expands and backs up, so that the next search's first
    // step below the root finds its node already loaded.
    NodeRegs<E> nx;
    int nx_link = 0; // LINK value of the record in flight into nx, 0 = none
    for (int i = first; i < A.n_total && err == 0; ++i) {
        if (AZ_UNLIKELY(worker_left == 0)) { // new worker, new generator (cmcts.cpp:132-133)
            noise_row = 0;
            worker_left = A.per_worker > 0 ? A.per_worker : A.n_total;
        }
        worker_left--;
        WState s = root_state;

        // level 0: the root, in registers
        STAMP(7);
        Pick pk;
        int pf_link = 0, pf_n = 0; // child of the likely next root pick
        {
            Cand c;
            double uk[E];
            scan_root<E, !COMPACT>(R, C, sq_root, s.occ, tid, c, uk);
            pk = finish_pick<WPT, E>(c, tid, s_slots, phase);
            if constexpr (WPT == 1 && EARLY_LOAD)
                if (AZ_LIKELY(pk.a >= 0)) runner_up<E>(R, uk, pk.a, tid, pf_link, pf_n);
        }
        STAMP(0);
#ifdef AZ_PF_STATS
        if (pk.link > 0) { pf_total++; if (pk.link == pf_last) pf_hit++; }
        pf_last = pf_link;
#endif
        if (AZ_UNLIKELY(pk.a < 0)) { err = AZ_ERR_NO_ACTION; break; }
        const int root_a = pk.a;
        const bool root_owner = tid == owner_of<E>(root_a);
        // the visit and the virtual loss (node.cpp:116-118) of the picked action, as
        // unconditional additions of 0 or 1 to all of a thread's actions (x + 0.0 and
        // x - 0.0 are x for these values): two selects and three additions per
        // action, no branch and no select of the results
        {
            const int root_k = slot_of<E>(root_a);
#pragma unroll
            for (int k = 0; k < E; k++) {
                const bool hit = root_owner && root_k == k;
                const int onei = hit ? 1 : 0;
                const double one = hit ? 1. : 0.;
                R.n[k] += onei;
                if constexpr (!COMPACT) C.np1[k] += one;
                R.w[k] -= one;
            }
        }
        root_n += 1;
        sq_root = sqrt_from_table<TAB>(root_n, s_sqrt); // for the next search: the table read overlaps this one

        int depth = 0;           // path entries below the root
        int parent_slot = root;  // node that will own the link to a new child
        int parent_a = pk.a;
        // The descent only walks; how it ended is acted upon after the loop, so the
        // expansion's bookkeeping is not carried around the loop.
        enum { GO = 0, TERMINAL = 1, EXPAND = 2, NO_ACTION = 3 };
        int leaf = GO;
        do {
            // pk = the pick made at the node above; its move is not applied yet.
            // The child's nodeN is the parent's visit count of the action minus
            // the visit that created it.  Its record is requested BEFORE the
            // move is applied so the rules code overlaps the memory latency.
            const int cur = pk.link - 1;
            const int nn = pk.n - 1;
            const bool fresh = nn == 0;
            NodeRegs<E> nd;
            // (links are unique per tree: an equal link is that very node, and what was
            // loaded is current because no search has run since)
            const bool piped = WPT == 1 && EARLY_LOAD && cur >= 0 && pk.link == nx_link;
            if (cur >= 0 && !piped) load_node<E, false>(P, tree, cur, tid, fresh, nd);
            state_move<E>(s, pk.a, tid);
            STAMP(1);
            if (AZ_UNLIKELY(s.winner != -1.f)) { // cmcts.cpp:253-267: no node for a finished game
                leaf = TERMINAL;
            } else if (cur < 0) { // Node::next_node creates the child (node.cpp:126-134)
                leaf = EXPAND;
            } else {
#ifdef AZ_STAMPS
                __builtin_amdgcn_s_waitcnt(0x0F70);
                STAMP(2);
#endif
                if (piped) {
                    // keeps the arithmetic on nx (and the wait for its loads) inside this branch
#pragma unroll
                    for (int k = 0; k < E; k++)
                        asm volatile("" : "+v"(nx.n[k]), "+v"(nx.link[k]), "+v"(nx.w[k]), "+v"(nx.p[k]));
                    pk = select_pick<WPT, E>(nx, fresh, sqrt_from_table<TAB>(nn, s_sqrt), s.occ, P.cpuct, tid, s_slots, phase);
                } else {
                    pk = select_pick<WPT, E>(nd, fresh, sqrt_from_table<TAB>(nn, s_sqrt), s.occ, P.cpuct, tid, s_slots, phase);
                }
                STAMP(3);
                if (AZ_UNLIKELY(pk.a < 0)) {
                    leaf = NO_ACTION;
                } else {
                    if (fresh) { // first select on this node: materialise its statistics
                        // (from registers of their own: reusing the load targets would
                        // make later loads wait for these stores to complete)
                        NodeRegs<E> z;
#pragma unroll
                        for (int k = 0; k < E; k++) {
                            z.n[k] = (tid == owner_of<E>(pk.a) && k == slot_of<E>(pk.a)) ? 1 : 0;
                            z.link[k] = 0;
                            z.w[k] = 0.;
                        }
                        store_stats<E>(P, tree, cur, tid, z);
                    } else if (tid == owner_of<E>(pk.a)) {
                        *reinterpret_cast<int *>(arena_ptr(P, tree) + (rec_off(cur) + OFF_N + (unsigned)pk.a * 4u)) = pk.n + 1;
                    }
                    // (lane == depth matches no lane once depth >= 64: the register write is
                    // then a no-op and needs no branch of its own)
                    path_put(path, depth, cur, pk.a, pk.w - 1. /* virtual loss */, tid & 63);
                    if (AZ_UNLIKELY(depth >= 64)) {
                        s_node[depth - 64] = cur; s_act[depth - 64] = pk.a; s_w1[depth - 64] = pk.w - 1.;
                    }
                    depth++;
                    parent_slot = cur;
                    parent_a = pk.a;
                    STAMP(5);
                }
            }
        } while (leaf == GO);

        if constexpr (WPT == 1 && EARLY_LOAD) {
            // requested now, after this search's last load (loads return in order: an
            // earlier request would hold up the descent's own), used by the next search
            if (pf_link > 0) load_node<E, false>(P, tree, pf_link - 1, tid, pf_n <= 1, nx);
            nx_link = pf_link; // 0 = nothing requested: an older copy must never be reused
        }
        float value = 0.f;
        if (AZ_UNLIKELY(leaf == TERMINAL)) {
            value = (s.winner == 0.f || s.winner == 1.f) ? 1.f : 0.f;
            n_term++;
        } else if (AZ_LIKELY(leaf == EXPAND)) { // the new child is expanded at once (cmcts.cpp:190-243)
            if (AZ_UNLIKELY(free_top == 0)) { err = AZ_ERR_POOL_FULL; break; }
            if (MIX != 0 && AZ_UNLIKELY(P.noise != nullptr && noise_row >= P.noise_rows)) { err = AZ_ERR_NOISE_ROWS; break; }
            const int slot = pop_free(P, tbase, free_top, fs_batch, fs_pos, tid);
            if (tid == 0) { P.order[tbase + n_order] = slot; P.parent[tbase + slot] = parent_slot; }
            n_order++;
            float pr[E];
            const float v = evaluate<E, EVAL>(A, s, tid, pr);
            double pp[E];
            mix_prior<E, MIX>(P, pr, false, noise_row, tid, pp);
            if (MIX != 0 && P.noise != nullptr) noise_row++;
            st_f64<E>(arena_ptr(P, tree), rec_off(slot) + OFF_P, tid, pp); // statistics stay implicit zeros
            if (tid == owner_of<E>(parent_a)) {
                if (depth == 0) {
#pragma unroll
                    for (int k = 0; k < E; k++)
                        if (k == slot_of<E>(parent_a)) R.link[k] = slot + 1;
                } else {
                    *reinterpret_cast<int *>(arena_ptr(P, tree) + (rec_off(parent_slot) + OFF_LINK + (unsigned)parent_a * 4u)) =
                        slot + 1;
                }
            }
            value = -v; // cmcts.cpp:228
            n_exp++;
            STAMP(4);
        } else {
            err = AZ_ERR_NO_ACTION;
        }
        if (AZ_UNLIKELY(err != 0)) break;
        // backup (cmcts.cpp:271-279, node.cpp:75-83): deepest first, sign flips;
        // each W word is written by the thread that owns (and later re-reads) it
        int d = depth - 1;
        for (; __builtin_expect(d >= 64, 0); --d) { // entries spilled to LDS (rare)
            const int act = s_act[d - 64];
            if (tid == owner_of<E>(act))
                *reinterpret_cast<double *>(arena_ptr(P, tree) + (rec_off(s_node[d - 64]) + OFF_W + (unsigned)act * 8u)) =
                    s_w1[d - 64] + ((double)value + 1.);
            value = -value;
        }
        for (; d >= 0; --d) {
            int node, act;
            double w1;
            path_get(path, d, node, act, w1);
            if (tid == owner_of<E>(act))
                *reinterpret_cast<double *>(arena_ptr(P, tree) + (rec_off(node) + OFF_W + (unsigned)act * 8u)) =
                    w1 + ((double)value + 1.);
            value = -value;
        }
        if (root_owner) { // the two divisions of this search: W/N and 1/(N+1)
#pragma unroll
            for (int k = 0; k < E; k++)
                if (k == slot_of<E>(root_a)) {
                    R.w[k] += (double)value + 1.;
                    C.q[k] = R.w[k] / (double)R.n[k];
                    C.rinv[k] = 1. / (COMPACT ? (double)(R.n[k] + 1) : C.np1[k]);
                }
        }
        n_sims++;
        n_sel += (unsigned)(depth + 1); // Node::select calls of this search: the root's and one per level below
        STAMP(6);
    }
#ifdef AZ_PF_STATS
    st_acc[5] = ((unsigned long long)pf_hit << 20) | pf_total;
#endif
    STAMP_FLUSH(tree);
    __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0): the last search's early load is not left in flight

    if (root >= 0) store_stats<E>(P, tree, root, tid, R);
    if (tid == 0) {
        Tr->root = root;
        Tr->root_n = root_n;
        Tr->free_top = free_top;
        Tr->n_order = n_order;
        Tr->noise_row = noise_row;
        if (err) Tr->err = err;
        Tr->stats[0] += n_sims;
        Tr->stats[1] += n_exp;
        Tr->stats[2] += n_term;
        Tr->stats[3] += n_sel;
    }
}
template <int WPT, int EVAL, int MIX, bool LEAN = false>
__global__ __launch_bounds__(64 * WPT) AZ_FUSED_ATTR void k_simulate_fused(AzParams P, AzFusedArgs A)
{
    simulate_fused_body<WPT, EVAL, MIX, LEAN ? 1 : 0>(P, A);
}
// the compact form: register budget of four wavefronts per SIMD
template <int EVAL, int MIX>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_simulate_fused_c(AzParams P, AzFusedArgs A)
{
    simulate_fused_body<1, EVAL, MIX, 2>(P, A);
}

// --------------------------------------------------------------- step: select
// Cmcts::search down to the leaf (mcts/src/cmcts.cpp:189-268) for an external
// evaluator, one wavefront per tree.  Virtual loss is stored
// (node.cpp:115-118) because the backup runs in a later launch.
// d_expand_mask (may be null): one bit per tree, set here when the leaf is to be expanded, for an evaluator
// that wants the list of those trees without scanning the codes; d_clear_mask: the mask of the step
// before, which nobody reads any more, zeroed here for the step after (the caller alternates two masks).
// out: the tree's index in the output arrays (d_boards, d_leaf_kind, the masks): the tree itself for the
// batched kernels, 0 for a caller that hands in buffers of its own
__device__ __forceinline__ void select_body(const AzParams &P, int tree, int out, int lane, float *d_boards, int *d_leaf_kind,
                                            unsigned *d_expand_mask, unsigned *d_clear_mask)
{
    constexpr int E = EPL;
    AzTree *T = &P.trees[tree];
    if (d_clear_mask != nullptr && lane == 0 && (out & 31) == 0) d_clear_mask[(out >> 5) * AZ_MASK_STRIDE] = 0u;
    if (uni(T->active) == 0 || uni(T->err) != 0) {
        if (lane == 0) { T->leaf_kind = AZ_LEAF_NONE; d_leaf_kind[out] = AZ_LEAF_NONE; }
        return;
    }
    WState s;
    load_state<E>(P, tree, lane, s);
    int *pn = P.path_node + (size_t)tree * P.path_cap;
    int *pa = P.path_act + (size_t)tree * P.path_cap;
    const int root = uni(T->root);
    int cur = root;
    int nn = root >= 0 ? uni(T->root_n) : 0;
    int depth = 0, kind = AZ_LEAF_EXPAND, err = 0, phase = 0;
    float leaf_value = 0.f;
    unsigned long long n_sel = 0;
    const bool deficits = (uni(T->flags) & AZ_TREE_DEFICIT) != 0; // (rare: left by concurrent workers)
    while (cur >= 0) {
        const bool fresh = nn == 0 && cur != root; // a root is always materialised
        NodeRegs<E> nd;
        load_node<E>(P, tree, cur, lane, fresh, nd);
        const Pick pk = select_pick<1, E>(nd, fresh, sqrt_node_n(nn, nullptr), s.occ, P.cpuct, lane, nullptr, phase);
        if (pk.a < 0) { err = AZ_ERR_NO_ACTION; break; }
        n_sel++;
        char *rec = rec_ptr(P, tree, cur);
        if (fresh) {
#pragma unroll
            for (int k = 0; k < E; k++)
                if (lane == owner_of<E>(pk.a) && k == slot_of<E>(pk.a)) { nd.n[k] = 1; nd.w[k] = -1.; }
            store_stats<E>(P, tree, cur, lane, nd);
        } else if (lane == owner_of<E>(pk.a)) {
            reinterpret_cast<int *>(rec + OFF_N)[pk.a] = pk.n + 1;
            reinterpret_cast<double *>(rec + OFF_W)[pk.a] = pk.w - 1.;
        }
        if (lane == 0) {
            if (cur == root) T->root_n = nn + 1;
            pn[depth] = cur;
            pa[depth] = pk.a;
        }
        depth++;
        state_move<E>(s, pk.a, lane);
        if (s.winner != -1.f) {
            kind = AZ_LEAF_TERMINAL;
            leaf_value = (s.winner == 0.f || s.winner == 1.f) ? 1.f : 0.f;
            break;
        }
        cur = pk.link - 1; // -1 ends the walk: the child does not exist yet
        nn = pk.n - 1;
        if (AZ_UNLIKELY(deficits) && cur >= 0) nn -= uni(P.aux[(size_t)tree * P.cap + cur]);
    }
    if (err) kind = AZ_LEAF_NONE;
    if (kind == AZ_LEAF_EXPAND) write_board(s, d_boards + (size_t)out * 2 * SIZE, lane);
    if (lane == 0) {
        T->leaf_kind = kind;
        T->path_len = depth;
        T->leaf_value = leaf_value;
        T->stats[3] += n_sel;
        if (err) T->err = err;
        d_leaf_kind[out] = kind;
        if (d_expand_mask != nullptr && kind == AZ_LEAF_EXPAND) atomicOr(&d_expand_mask[(out >> 5) * AZ_MASK_STRIDE], 1u << (out & 31));
    }
}
// Register budget of the step kernels.  Boards up to 16x16 (four actions per lane): eight wavefronts per SIMD,
// i.e. 64 registers, so that 8192 trees are resident at once instead of in two rounds of seven per SIMD (no
// spill; 8 us less per search step at 8192 trees).  With eight actions per lane 64 registers would spill.
// (-DAZ_STEP_NO_OCC builds them without the bound, for A/B timing.)
#ifdef AZ_STEP_NO_OCC
#define AZ_STEP_OCC
#else
#ifndef AZ_STEP_OCC_WIDE
#define AZ_STEP_OCC_WIDE 1 /* (experiments: the bound for boards with eight actions per lane) */
#endif
#define AZ_STEP_OCC __attribute__((amdgpu_waves_per_eu(EPL <= 4 ? 8 : AZ_STEP_OCC_WIDE, 8)))
#endif
__global__ __launch_bounds__(64) AZ_STEP_OCC void k_select(AzParams P, float *d_boards, int *d_leaf_kind, unsigned *d_expand_mask,
                                               unsigned *d_clear_mask)
{
    const int tree = blockIdx.x + P.tree_base;
    select_body(P, tree, tree, threadIdx.x, d_boards, d_leaf_kind, d_expand_mask, d_clear_mask);
}

// ------------------------------------------------------ step: expand + backup
// Node::set_prior + the backup loop (mcts/src/node.cpp:26-47,75-83;
// cmcts.cpp:228-231,271-279) for the step prepared by k_select.
// Every word a later select of the same wavefront reads back (W, LINK, P, the tree header) is written by
// the lane that will load it (the action's owner, lane 0 for the header), so k_backup_select needs no
// fence between its two halves: a lane's load follows its own store in program order.
__device__ __forceinline__ void expand_backup_body(const AzParams &P, int tree, int out, int lane, const float *d_value,
                                                   const float *d_prior, int prior_is_log)
{
    constexpr int E = EPL;
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
        const int free_top = uni(T->free_top);
        const int n_order = uni(T->n_order);
        const int noise_row = uni(T->noise_row);
        if (free_top == 0) err = AZ_ERR_POOL_FULL;
        else if (P.noise != nullptr && noise_row >= P.noise_rows) err = AZ_ERR_NOISE_ROWS;
        if (err) {
            if (lane == 0) { T->err = err; T->leaf_kind = AZ_LEAF_NONE; }
            return;
        }
        const int slot = uni(P.free_stack[tbase + free_top - 1]);
        float pr[E];
#pragma unroll
        for (int k = 0; k < E; k++) {
            const int a = lane * E + k;
            pr[k] = a < SIZE ? d_prior[(size_t)out * SIZE + a] : 0.f;
        }
        NodeRegs<E> nd;
        mix_prior<E>(P, pr, prior_is_log != 0, noise_row, lane, nd.p);
        st_f64<E>(rec_ptr(P, tree, slot), OFF_P, lane, nd.p);
        const int parent_slot = depth > 0 ? uni(pn[depth - 1]) : -1;
        if (depth == 0) { // the new node is the root: materialise it
#pragma unroll
            for (int k = 0; k < E; k++) { nd.n[k] = 0; nd.link[k] = 0; nd.w[k] = 0.; }
            store_stats<E>(P, tree, slot, lane, nd);
        }
        if (depth > 0) {
            const int parent_a = uni(pa[depth - 1]);
            if (lane == owner_of<E>(parent_a)) reinterpret_cast<int *>(rec_ptr(P, tree, parent_slot) + OFF_LINK)[parent_a] = slot + 1;
        }
        if (lane == 0) {
            if (depth == 0) {
                T->root = slot;
                T->root_n = 0;
            }
            P.parent[tbase + slot] = parent_slot;
            P.order[tbase + n_order] = slot;
            T->free_top = free_top - 1;
            T->n_order = n_order + 1;
            if (P.noise != nullptr) T->noise_row = noise_row + 1;
            T->stats[1] += 1;
        }
        value = -d_value[out]; // cmcts.cpp:228
    } else {
        value = unif(T->leaf_value);
        if (lane == 0) T->stats[2] += 1;
    }
    // backup: the owner of each level's action updates its W (independent addresses)
    for (int d = 0; d < depth; d++) {
        const int node = uni(pn[d]), act = uni(pa[d]);
        if (lane == owner_of<E>(act)) {
            // sign at level d: value flips once per level walking up from the leaf
            const float vd = ((depth - 1 - d) & 1) ? -value : value;
            double *w = reinterpret_cast<double *>(rec_ptr(P, tree, node) + OFF_W) + act;
            *w += (double)vd + 1.; // node.cpp:80
        }
    }
    if (lane == 0) {
        T->leaf_kind = AZ_LEAF_NONE;
        T->stats[0] += 1;
    }
}
__global__ __launch_bounds__(64) void k_expand_backup(AzParams P, const float *d_value, const float *d_prior,
                                                       int prior_is_log)
{
    const int tree = blockIdx.x + P.tree_base;
    expand_backup_body(P, tree, tree, threadIdx.x, d_value, d_prior, prior_is_log);
}
// The backup of one search step and the select of the next in one launch (the network search's inner
// loop: one launch and one pass over the tree header less per step).
__global__ __launch_bounds__(64) AZ_STEP_OCC void k_backup_select(AzParams P, const float *d_value, const float *d_prior,
                                                       int prior_is_log, float *d_boards, int *d_leaf_kind,
                                                       unsigned *d_expand_mask, unsigned *d_clear_mask)
{
    const int tree = blockIdx.x + P.tree_base;
    expand_backup_body(P, tree, tree, threadIdx.x, d_value, d_prior, prior_is_log);
    select_body(P, tree, tree, threadIdx.x, d_boards, d_leaf_kind, d_expand_mask, d_clear_mask);
}

// ------------------------------------------------ concurrent workers, lock step
// threads > 1 in the reference (cmcts.cpp:98-119): T std::threads search one tree at once, kept apart by the
// virtual loss (node.cpp:115-118); how they interleave is up to the OS.  Here they advance in lock step, which
// is one legal interleaving: per step the workers 0..T-1, in that order, descend to their leaf (k_workers_select:
// every visit and virtual loss is stored at once, so a later worker of the step sees it), the T leaves of every
// tree are evaluated in ONE batch, and the workers finish in the same order (k_workers_backup: set_prior, which
// does nothing when an earlier worker of the step expanded the same node, node.cpp:30-33; backup).
// Node::next_node (node.cpp:126-134) creates the child during the DESCENT, so a later worker of the same step can
// arrive at a node that exists but is not expanded: such a node is marked in AzParams::aux (see az_common.h); the
// workers that meet there all count a visit of the parent's action, none of them selects from the child, and the
// surplus is kept as the child's visit deficit once it is expanded.
// The tree header and the workers' records live in registers for the duration of a worker kernel: lane i holds
// 32-bit word i of the header (one global round trip instead of one per field and worker; with one tree per GPU a
// step's duration is the sum of its dependent round trips), the fields that change are carried in scalars and
// written back once at the end.
struct WHdr {
    int root, root_n, free_top, n_order, err, flags, active;
    unsigned n_sims, n_exp, n_term;
    unsigned long long n_sel;
    int words; // lane i: word i of the AzTree as loaded
    int fbatch; // lane j: the j-th entry below the top of the free stack as it was at load time
    int fpos;   // entries of fbatch handed out
};
static_assert(sizeof(AzTree) % 4 == 0 && sizeof(AzTree) <= 256, "the header fits one word per lane");
static_assert(sizeof(AzWorker) == 32, "eight words per worker: eight workers per 64-lane chunk");
#define AZ_HDR_WORD(words, field) __builtin_amdgcn_readlane(words, (int)(offsetof(AzTree, field) / 4))
__device__ __forceinline__ void whdr_free_batch(const AzParams &P, int tree, int lane, WHdr &h)
{
    const int idx = h.free_top - 1 - lane;
    h.fbatch = idx >= 0 ? P.free_stack[(size_t)tree * P.cap + idx] : -1;
    h.fpos = 0;
}
// the next free slot (the caller has checked free_top > 0).  A pop per created node would be a memory round trip
// each; the 64 entries on top of the stack are requested once, right after the header.
__device__ __forceinline__ int whdr_pop_free(const AzParams &P, int tree, int lane, WHdr &h)
{
    if (AZ_UNLIKELY(h.fpos == 64)) whdr_free_batch(P, tree, lane, h);
    const int slot = __builtin_amdgcn_readlane(h.fbatch, h.fpos);
    h.fpos++;
    h.free_top--;
    return slot;
}
__device__ __forceinline__ void whdr_load(const AzParams &P, int tree, int lane, WHdr &h)
{
    const AzTree *T = &P.trees[tree];
    h.words = lane < (int)(sizeof(AzTree) / 4) ? reinterpret_cast<const int *>(T)[lane] : 0;
    h.root = AZ_HDR_WORD(h.words, root);
    h.root_n = AZ_HDR_WORD(h.words, root_n);
    h.free_top = AZ_HDR_WORD(h.words, free_top);
    h.n_order = AZ_HDR_WORD(h.words, n_order);
    h.err = AZ_HDR_WORD(h.words, err);
    h.flags = AZ_HDR_WORD(h.words, flags);
    h.active = AZ_HDR_WORD(h.words, active);
    h.n_sims = h.n_exp = h.n_term = 0u;
    h.n_sel = 0ull;
    whdr_free_batch(P, tree, lane, h);
}
__device__ __forceinline__ void whdr_store(AzTree *T, int lane, const WHdr &h)
{
    if (lane != 0) return;
    T->root = h.root;
    T->root_n = h.root_n;
    T->free_top = h.free_top;
    T->n_order = h.n_order;
    T->err = h.err;
    T->flags = h.flags;
    const int s0 = (int)(offsetof(AzTree, stats) / 4);
    const unsigned long long add[4] = {h.n_sims, h.n_exp, h.n_term, h.n_sel};
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const unsigned lo = (unsigned)__builtin_amdgcn_readlane(h.words, s0 + 2 * k), hi = (unsigned)__builtin_amdgcn_readlane(h.words, s0 + 2 * k + 1);
        T->stats[k] = (((unsigned long long)hi << 32) | lo) + add[k];
    }
}
// the board and whose move it is, from the header words and the bit rows
template <int E> __device__ __forceinline__ void load_state_hdr(const AzParams &P, int tree, int lane, int words, WState &s)
{
    s.rb = P.rows[tree * 64 + lane];
    s.player = AZ_HDR_WORD(words, player);
    s.move_cnt = AZ_HDR_WORD(words, move_cnt);
    s.winner = __int_as_float(AZ_HDR_WORD(words, winner));
    s.occ = 0u;
#pragma unroll
    for (int k = 0; k < E; k++) {
        const int a = lane * E + k;
        const int r = a < SIZE ? a / S : 0, c = a < SIZE ? a % S : 0;
        const unsigned both = (unsigned)__shfl((int)s.rb, r) | (unsigned)__shfl((int)s.rb, 32 + r);
        if (a < SIZE && ((both >> c) & 1u)) s.occ |= 1u << k;
    }
}
// fields of a worker's record inside a chunk of eight records held one word per lane
enum { WF_KIND = 0, WF_PATH_LEN = 1, WF_VALUE = 2, WF_NOISE_ROW = 3, WF_LEAF_SLOT = 4 };
__device__ __forceinline__ int wget(int wr, int wi, int field) { return __builtin_amdgcn_readlane(wr, wi * 8 + field); }
__device__ __forceinline__ void wput(int &wr, int lane, int wi, int field, int v) { wr = lane == wi * 8 + field ? v : wr; }

// Accesses that are coherent between the XCDs of the device without a cache write-back or invalidate: relaxed atomics
// at agent scope go through to memory (COH = true below: the exchange of the one-launch cluster search, where the
// master keeps its tree in its L2 for the whole call and an acquire would drop it at every step).
__device__ __forceinline__ void coh_store(int *p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int coh_load(const int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float coh_load_f(const float *p) { return __int_as_float(coh_load(reinterpret_cast<const int *>(p))); }
// a 32-bit payload with a 32-bit tag in ONE 64-bit access: whoever reads the tag it waits for has the payload too,
// with no ordering between separate stores to arrange
__device__ __forceinline__ void coh_store_tagged(unsigned long long *p, int tag, unsigned payload)
{
    __hip_atomic_store(p, ((unsigned long long)(unsigned)tag << 32) | payload, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long coh_load_tagged(const unsigned long long *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// worker (chunk position wi, index w) descends to its leaf
// COH: the leaf is not written to d_boards: its 64 bit rows (this lane's one) are handed back in leaf_rb
template <bool COH = false>
__device__ __forceinline__ void wselect_body(const AzParams &P, int tree, int w, int wi, int nw, int lane, const WState &root_state,
                                             WHdr &h, int &wr, float *d_boards, int *d_leaf_kind, unsigned *d_expand_mask,
                                             unsigned *d_clear_mask, unsigned &leaf_rb STAMP_PARAM)
{
    constexpr int E = EPL;
    const int out = tree * nw + w;
    leaf_rb = 0u;
    if (d_clear_mask != nullptr && lane == 0 && (out & 31) == 0) d_clear_mask[(out >> 5) * AZ_MASK_STRIDE] = 0u;
    if (h.active == 0 || h.err != 0) { // (an error raised by an earlier worker of this step stops the others too)
        wput(wr, lane, wi, WF_KIND, AZ_LEAF_NONE);
        if (lane == 0) d_leaf_kind[out] = AZ_LEAF_NONE;
        return;
    }
    const size_t tbase = (size_t)tree * P.cap;
    int *pn = P.wpath_node + ((size_t)tree * P.max_workers + w) * P.path_cap;
    int *pa = P.wpath_act + ((size_t)tree * P.max_workers + w) * P.path_cap;
    WState s = root_state;
    const int root = h.root;
    int cur = root;
    int nn = root >= 0 ? h.root_n : 0;
    int depth = 0, kind = AZ_LEAF_EXPAND, err = 0, phase = 0, leaf_slot = -1;
    float leaf_value = 0.f;
    NodeRegs<E> nd;
    if (cur >= 0) load_node<E>(P, tree, cur, lane, false, nd); // a root is always materialised
    while (cur >= 0) {
        const bool fresh = nn == 0 && cur != root;
        if (fresh) { // (the record was requested whole before its nodeN was known: only its priors exist yet)
#pragma unroll
            for (int k = 0; k < E; k++) { nd.n[k] = 0; nd.link[k] = 0; nd.w[k] = 0.; }
        }
        STAMP(3);
        const Pick pk = select_pick<1, E>(nd, fresh, sqrt_node_n(nn, nullptr), s.occ, P.cpuct, lane, nullptr, phase);
        STAMP(4);
        if (pk.a < 0) { err = AZ_ERR_NO_ACTION; break; }
        h.n_sel++;
        char *rec = rec_ptr(P, tree, cur);
        if (fresh) {
#pragma unroll
            for (int k = 0; k < E; k++)
                if (lane == owner_of<E>(pk.a) && k == slot_of<E>(pk.a)) { nd.n[k] = 1; nd.w[k] = -1.; }
            store_stats<E>(P, tree, cur, lane, nd);
        } else if (lane == owner_of<E>(pk.a)) {
            reinterpret_cast<int *>(rec + OFF_N)[pk.a] = pk.n + 1;
            reinterpret_cast<double *>(rec + OFF_W)[pk.a] = pk.w - 1.;
        }
        if (cur == root) h.root_n = nn + 1;
        if (lane == 0) {
            pn[depth] = cur;
            pa[depth] = pk.a;
        }
        depth++;
        // the child's record and its mark are requested together, before the rules code: one round trip per level
        const int child = pk.link - 1;
        int ax = 0;
        if (child >= 0) {
            ax = P.aux[tbase + child];
            load_node<E>(P, tree, child, lane, false, nd);
        }
        STAMP(5);
        state_move<E>(s, pk.a, lane);
        STAMP(6);
        if (s.winner != -1.f) {
            kind = AZ_LEAF_TERMINAL;
            leaf_value = (s.winner == 0.f || s.winner == 1.f) ? 1.f : 0.f;
            break;
        }
        if (child < 0) { // Node::next_node creates the child now (node.cpp:126-134); this worker will expand it
            if (h.free_top == 0) { err = AZ_ERR_POOL_FULL; break; }
            const int slot = whdr_pop_free(P, tree, lane, h);
            if (lane == owner_of<E>(pk.a)) reinterpret_cast<int *>(rec + OFF_LINK)[pk.a] = slot + 1;
            if (lane == 0) {
                P.parent[tbase + slot] = cur;
                P.order[tbase + h.n_order] = slot;
                P.aux[tbase + slot] = -1;
            }
            h.n_order++;
            leaf_slot = slot;
            break;
        }
        ax = uni(ax); // (written by lane 0, read back through lane 0)
        if (ax < 0) { // created by an earlier worker of this step and not expanded yet: the leaf of this worker too
            if (lane == 0) P.aux[tbase + child] = ax - 1;
            leaf_slot = child;
            break;
        }
        cur = child;
        nn = pk.n - 1 - ax;
    }
    if (err) { kind = AZ_LEAF_NONE; h.err = err; }
    if (kind == AZ_LEAF_EXPAND) {
        if (COH) leaf_rb = s.rb;
        else write_board(s, d_boards + (size_t)out * 2 * SIZE, lane);
    }
    wput(wr, lane, wi, WF_KIND, kind);
    wput(wr, lane, wi, WF_PATH_LEN, depth);
    wput(wr, lane, wi, WF_VALUE, __float_as_int(leaf_value));
    wput(wr, lane, wi, WF_LEAF_SLOT, leaf_slot);
    if (lane == 0) {
        d_leaf_kind[out] = kind;
        if (d_expand_mask != nullptr && kind == AZ_LEAF_EXPAND) atomicOr(&d_expand_mask[(out >> 5) * AZ_MASK_STRIDE], 1u << (out & 31));
    }
}

// worker (chunk position wi, index w) finishes its search: set_prior + backup
// COH: the value and the priors were read by the caller (v_pre, pr_pre); d_value / d_prior are not touched
template <bool COH = false>
__device__ __forceinline__ void wbackup_body(const AzParams &P, int tree, int w, int wi, int nw, int lane, WHdr &h, int &wr,
                                             const float *d_value, const float *d_prior, int prior_is_log, float v_pre = 0.f,
                                             const float *pr_pre = nullptr)
{
    constexpr int E = EPL;
    const int kind = wget(wr, wi, WF_KIND);
    if (kind == AZ_LEAF_NONE) return;
    wput(wr, lane, wi, WF_KIND, AZ_LEAF_NONE);
    if (h.err != 0) return; // an earlier worker of this step failed: the tree is left as it is (no backup either)
    const int out = tree * nw + w;
    const size_t tbase = (size_t)tree * P.cap;
    const int *pn = P.wpath_node + ((size_t)tree * P.max_workers + w) * P.path_cap;
    const int *pa = P.wpath_act + ((size_t)tree * P.max_workers + w) * P.path_cap;
    const int depth = wget(wr, wi, WF_PATH_LEN);
    // the path, one level per lane (levels past the 64th are read one by one below)
    const int pnv = lane < depth ? pn[lane] : 0, pav = lane < depth ? pa[lane] : 0;
    // The W words of the first PRE levels are requested now, together with everything else this worker reads, so
    // that the whole backup costs two memory round trips (a path visits a word once, and the expansion below
    // touches none of them).  Each word is read and written by the lane that owns its action.
    constexpr int PRE = 8;
    double oldw[PRE];
#pragma unroll
    for (int d = 0; d < PRE; d++) {
        oldw[d] = 0.;
        if (d < depth) {
            const int node = __builtin_amdgcn_readlane(pnv, d), act = __builtin_amdgcn_readlane(pav, d);
            if (lane == owner_of<E>(act)) oldw[d] = *(reinterpret_cast<const double *>(rec_ptr(P, tree, node) + OFF_W) + act);
        }
    }
    float value;
    if (kind == AZ_LEAF_EXPAND) {
        const int leaf_slot = wget(wr, wi, WF_LEAF_SLOT);
        const bool is_root = leaf_slot < 0;
        const int noise_row = wget(wr, wi, WF_NOISE_ROW);
        // requested before it is known whether this worker expands: the node's mark, the value, the priors, the noise
        int ax = is_root ? 0 : P.aux[tbase + leaf_slot];
        const float v_in = COH ? v_pre : d_value[out];
        float pr[E];
#pragma unroll
        for (int k = 0; k < E; k++) {
            const int a = lane * E + k;
            pr[k] = a < SIZE ? (COH ? pr_pre[k] : d_prior[(size_t)out * SIZE + a]) : 0.f;
        }
        double nz[E];
#pragma unroll
        for (int k = 0; k < E; k++) nz[k] = 0.;
        if (P.noise != nullptr && noise_row < P.noise_rows)
            ld_f64<E>(reinterpret_cast<const char *>(P.noise), (unsigned)noise_row * (unsigned)(SP * 8), lane, nz);
        ax = uni(ax);
        int slot = leaf_slot, err = 0;
        bool expand;
        if (is_root) {
            expand = h.root < 0; // else an earlier worker of this step has expanded the root
            if (expand) {
                if (h.free_top == 0) err = AZ_ERR_POOL_FULL;
                else slot = whdr_pop_free(P, tree, lane, h);
            }
        } else {
            expand = ax < 0; // else expanded by the worker that created it (node.cpp:30-33: nothing to do)
        }
        if (expand && !err && P.noise != nullptr && noise_row >= P.noise_rows) err = AZ_ERR_NOISE_ROWS;
        if (err) { h.err = err; return; }
        if (expand) {
            // Node::set_prior (node.cpp:38-45), the arithmetic of mix_prior on the rows loaded above
            NodeRegs<E> nd;
            const double prior_eps = 1. - P.eps;
#pragma unroll
            for (int k = 0; k < E; k++) {
                const double p = prior_is_log ? (double)expf(pr[k]) : (double)pr[k];
                nd.p[k] = (lane * E + k < SIZE) ? prior_eps * p + P.eps * nz[k] : 0.;
            }
            st_f64<E>(rec_ptr(P, tree, slot), OFF_P, lane, nd.p);
            if (is_root) { // the root is always materialised
#pragma unroll
                for (int k = 0; k < E; k++) { nd.n[k] = 0; nd.link[k] = 0; nd.w[k] = 0.; }
                store_stats<E>(P, tree, slot, lane, nd);
                if (lane == 0) {
                    P.parent[tbase + slot] = -1;
                    P.order[tbase + h.n_order] = slot;
                    P.aux[tbase + slot] = 0;
                }
                h.root = slot;
                h.root_n = 0;
                h.n_order++;
            } else {
                const int extra = -ax - 1; // workers that arrived after the one that created the node
                if (lane == 0) P.aux[tbase + slot] = extra;
                if (extra > 0) h.flags |= AZ_TREE_DEFICIT;
            }
            if (P.noise != nullptr) wput(wr, lane, wi, WF_NOISE_ROW, noise_row + 1);
            h.n_exp++;
        }
        value = -unif(v_in); // cmcts.cpp:228
    } else {
        value = __int_as_float(wget(wr, wi, WF_VALUE));
        h.n_term++;
    }
    // backup (node.cpp:75-83): the owner of each level's action updates its W
#pragma unroll
    for (int d = 0; d < PRE; d++) {
        if (d < depth) {
            const int node = __builtin_amdgcn_readlane(pnv, d), act = __builtin_amdgcn_readlane(pav, d);
            if (lane == owner_of<E>(act)) {
                const float vd = ((depth - 1 - d) & 1) ? -value : value;
                *(reinterpret_cast<double *>(rec_ptr(P, tree, node) + OFF_W) + act) = oldw[d] + ((double)vd + 1.);
            }
        }
    }
    for (int d = PRE; d < depth; d++) {
        const int node = d < 64 ? __builtin_amdgcn_readlane(pnv, d) : uni(pn[d]);
        const int act = d < 64 ? __builtin_amdgcn_readlane(pav, d) : uni(pa[d]);
        if (lane == owner_of<E>(act)) {
            const float vd = ((depth - 1 - d) & 1) ? -value : value;
            double *wp = reinterpret_cast<double *>(rec_ptr(P, tree, node) + OFF_W) + act;
            *wp += (double)vd + 1.;
        }
    }
    h.n_sims++;
}

// the workers' records of chunk c (eight records, one word per lane)
__device__ __forceinline__ int wchunk_load(const AzParams &P, int tree, int c, int nw, int lane)
{
    const int *base = reinterpret_cast<const int *>(P.workers + (size_t)tree * P.max_workers + c);
    return lane < 8 * min(8, nw - c) ? base[lane] : 0;
}
__device__ __forceinline__ void wchunk_store(const AzParams &P, int tree, int c, int nw, int lane, int wr)
{
    int *base = reinterpret_cast<int *>(P.workers + (size_t)tree * P.max_workers + c);
    if (lane < 8 * min(8, nw - c)) base[lane] = wr;
}
__device__ __forceinline__ void workers_select_phase(const AzParams &P, int tree, int nw, int lane, WHdr &h, float *d_boards,
                                                     int *d_leaf_kind, unsigned *d_expand_mask, unsigned *d_clear_mask STAMP_PARAM)
{
    WState rs;
    load_state_hdr<EPL>(P, tree, lane, h.words, rs);
    for (int c = 0; c < nw; c += 8) {
        int wr = wchunk_load(P, tree, c, nw, lane);
        for (int wi = 0; wi < min(8, nw - c); wi++) {
            unsigned unused_rb;
            wselect_body(P, tree, c + wi, wi, nw, lane, rs, h, wr, d_boards, d_leaf_kind, d_expand_mask, d_clear_mask,
                         unused_rb STAMP_ARG);
            STAMP(7);
        }
        wchunk_store(P, tree, c, nw, lane, wr);
    }
}
__device__ __forceinline__ void workers_backup_phase(const AzParams &P, int tree, int nw, int lane, WHdr &h, const float *d_value,
                                                     const float *d_prior, int prior_is_log)
{
    for (int c = 0; c < nw; c += 8) {
        int wr = wchunk_load(P, tree, c, nw, lane);
        for (int wi = 0; wi < min(8, nw - c); wi++) wbackup_body(P, tree, c + wi, wi, nw, lane, h, wr, d_value, d_prior, prior_is_log);
        wchunk_store(P, tree, c, nw, lane, wr);
    }
}

__global__ __launch_bounds__(64) void k_workers_select(AzParams P, int nw, float *d_boards, int *d_leaf_kind, unsigned *d_expand_mask,
                                                       unsigned *d_clear_mask)
{
    const int tree = blockIdx.x + P.tree_base, lane = threadIdx.x;
    WHdr h;
    STAMP_DECL;
    whdr_load(P, tree, lane, h);
    workers_select_phase(P, tree, nw, lane, h, d_boards, d_leaf_kind, d_expand_mask, d_clear_mask STAMP_ARG);
    whdr_store(&P.trees[tree], lane, h);
}
__global__ __launch_bounds__(64) void k_workers_backup(AzParams P, int nw, const float *d_value, const float *d_prior, int prior_is_log)
{
    const int tree = blockIdx.x + P.tree_base, lane = threadIdx.x;
    WHdr h;
    whdr_load(P, tree, lane, h);
    workers_backup_phase(P, tree, nw, lane, h, d_value, d_prior, prior_is_log);
    whdr_store(&P.trees[tree], lane, h);
}
__global__ __launch_bounds__(64) void k_workers_backup_select(AzParams P, int nw, const float *d_value, const float *d_prior,
                                                              int prior_is_log, float *d_boards, int *d_leaf_kind,
                                                              unsigned *d_expand_mask, unsigned *d_clear_mask)
{
    const int tree = blockIdx.x + P.tree_base, lane = threadIdx.x;
    WHdr h;
    STAMP_DECL;
    whdr_load(P, tree, lane, h);
    STAMP(0);
    workers_backup_phase(P, tree, nw, lane, h, d_value, d_prior, prior_is_log);
    STAMP(1);
    workers_select_phase(P, tree, nw, lane, h, d_boards, d_leaf_kind, d_expand_mask, d_clear_mask STAMP_ARG);
    whdr_store(&P.trees[tree], lane, h);
    STAMP(2);
    STAMP_ADD(tree);
}

// ------------------------------------------- network search in one launch (few trees)
// Cmcts::simulate with the reference network (cmcts.cpp:83-282) for batches of at most one tree per CU:
// ONE launch for all searches of a call instead of two launches per search, whose dispatch gaps are a
// third of such a step.  A workgroup of eight wavefronts per tree: wavefront 0 runs select and expand +
// backup (select_body / expand_backup_body, the code of the stepped kernels), all eight evaluate the leaf
// (aznet::team_position, the code of k_net_eval); the leaf's planes, its code, the value and the log-priors
// pass through LDS.  Same arithmetic as the launches it replaces, hence the same results.
using NetG = aznet::Geo<S, 8>;
constexpr int NET_SEARCH_LDS_FLOATS = 2 * NetG::BUF + 16 + NetG::SIZE + 2 * NetG::SIZE + NetG::SIZE + 16 + 64; // (+ 64: the cluster search's bit rows)
__global__ __launch_bounds__(NetG::NTHR) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_net_search(
    AzParams P, int th_num, int per, const float *prm, int front, int pol, int val)
{
    using G = NetG;
    extern __shared__ float lds[];
    float *F = lds, *W = lds + G::BUF;
    float *red = lds + 2 * G::BUF;      // [16] reductions of the heads
    float *logit = red + 16;            // [SIZE]
    float *planes = logit + G::SIZE;    // [2][SIZE] the leaf as select writes it
    float *logp = planes + 2 * G::SIZE; // [SIZE]
    float *value = logp + G::SIZE;      // [1]
    int *kind = reinterpret_cast<int *>(value + 1); // [2] the step's AZ_LEAF_* code, steps alternate between the cells
    const int tree = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = uni(tid >> 6);
    AzTree *T = &P.trees[tree];
    if (uni(T->active) == 0 || uni(T->err) != 0) return; // (the whole workgroup: nothing to search)
    for (int i = tid; i < 2 * G::BUF; i += G::NTHR) lds[i] = 0.f; // halos stay zero from here on
    aznet::TeamCtx<G> ctx;
    aznet::team_setup<G>(ctx, prm, front, pol, val, wave, lane);
    __syncthreads();
    int step = 0;
    for (int w = 0; w < th_num; w++) {
        // a new worker builds a new generator (cmcts.cpp:132-133): the noise stream restarts
        if (tid == 0) T->noise_row = 0;
        for (int i = 0; i < per; i++, step++) {
            // (the code goes to the cell of this step's parity: a wavefront that has not read the step's code
            // yet when wavefront 0 is already selecting for the next step finds it untouched; by the time the
            // cell is written again every wavefront has passed the barrier of the step in between)
            int *kd = kind + (step & 1);
            if (wave == 0) select_body(P, tree, 0, lane, planes, kd, nullptr, nullptr);
            __syncthreads();
            if (kd[0] == AZ_LEAF_EXPAND) { // (workgroup-uniform)
                aznet::team_position<G>(ctx, F, W, red, logit, planes, value, logp, tid, wave, lane);
                __syncthreads();
            }
            if (wave == 0) expand_backup_body(P, tree, 0, lane, value, logp, 1);
            // (wavefront 0 goes on to the next select; the others wait for it at the barrier above)
        }
    }
}

// ----------------------- network search in one launch, concurrent workers (few trees)
// The lock-step schedule of k_workers_* / az_net_eval for batches small enough that every tree gets nw + 1 CUs of
// its own: ONE launch for all steps of a call.  Workgroup 0 of a tree's cluster is the master (its wavefront 0 runs
// the workers' descents and backups, the code of the stepped kernels); workgroup 1 + w is an eight-wavefront
// evaluator team for worker w's leaf (aznet::team_position, the code of k_net_eval).  Hence the same results as the
// stepped launches, which the tests require bit for bit.  What the launch saves: two dispatches per step, and the
// workers' leaves no longer wait for each other -- worker w's position is evaluated while the master still descends
// for worker w + 1, and its backup runs while later positions are still being evaluated.
// The workgroups of a cluster run on different CUs (and XCDs, whose L2 caches are not coherent with each other) and
// talk through device memory with relaxed atomics at agent scope, which go through to memory: no cache is written
// back or invalidated, so the master's tree stays in its L2 for the whole call.  Every word exchanged travels in a
// 64-bit access together with a tag that names the step, so the reader that finds the tag it waits for has the word
// as well and nobody waits for a store to arrive before raising a flag:
//   leaf    [position][64]       master -> team: tag 4 * (step + 1) + the leaf's AZ_LEAF_* code, payload one bit row;
//                                written at every step whatever the code (the team counts its steps by the tags)
//   result  [position][SIZE + 1] team -> master: tag step + 1, payload a log-prior (SIZE of them) or the value
// The master waits for a team only when the leaf is to be evaluated, so a team can find its tags several steps
// ahead: those steps had nothing for it, and it skips them.
// Every workgroup of the grid must be resident for the waits to end: the launcher asks for a cooperative launch, which
// guarantees it or fails.  Every wait is bounded all the same: a wait that runs out raises `stop`, which ends every
// other wait of the launch at once, and the tree reports AZ_ERR_STATE.
constexpr int CLUSTER_SPINS = 1 << 22;
constexpr int CLUSTER_RESULT = SIZE + 1; // tagged words of a position's result
// true while waiting may go on
__device__ __forceinline__ bool cluster_keep_waiting(int &it, int *stop)
{
    it++;
    if ((it & 31) == 0 && coh_load(stop) != 0) return false;
    if (it >= CLUSTER_SPINS) {
        coh_store(stop, 1);
        return false;
    }
    __builtin_amdgcn_s_sleep(1);
    return true;
}
__global__ __launch_bounds__(NetG::NTHR) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_net_search_workers(
    AzParams P, int nw, int per, const float *prm, int front, int pol, int val, int *d_leaf_kind, int *d_flags)
{
    using G = NetG;
    extern __shared__ float lds[];
    const int cluster = nw + 1;
    const int tree = (int)(blockIdx.x / cluster), role = (int)(blockIdx.x % cluster);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = uni(tid >> 6);
    const int n_pos = (int)(gridDim.x / cluster) * nw;
    int *stop = d_flags;
    unsigned long long *leaf_x = reinterpret_cast<unsigned long long *>(d_flags + 64);
    unsigned long long *result_x = leaf_x + (size_t)n_pos * 64;
    if (role == 0) { // ---- the master: the worker kernels' code, all steps
        if (tid >= 64) return;
        constexpr int E = EPL;
        WHdr h;
        STAMP_DECL;
        whdr_load(P, tree, lane, h);
        WState rs;
        load_state_hdr<EPL>(P, tree, lane, h.words, rs);
        int wr = wchunk_load(P, tree, 0, nw, lane);
        for (int step = 0; step < per; step++) {
            for (int wi = 0; wi < nw; wi++) {
                unsigned leaf_rb;
                wselect_body<true>(P, tree, wi, wi, nw, lane, rs, h, wr, nullptr, d_leaf_kind, nullptr, nullptr, leaf_rb STAMP_ARG);
                coh_store_tagged(leaf_x + (size_t)(tree * nw + wi) * 64 + lane, 4 * (step + 1) + wget(wr, wi, WF_KIND), leaf_rb);
            }
            for (int wi = 0; wi < nw; wi++) {
                float v_pre = 0.f, pr_pre[E];
#pragma unroll
                for (int k = 0; k < E; k++) pr_pre[k] = 0.f;
                if (wget(wr, wi, WF_KIND) == AZ_LEAF_EXPAND) {
                    // this lane's log-priors and the value, until every word carries this step's tag
                    const unsigned long long *res = result_x + (size_t)(tree * nw + wi) * CLUSTER_RESULT;
                    int it = 0;
                    bool got = false;
                    for (;;) {
                        bool mine = true;
#pragma unroll
                        for (int k = 0; k < E; k++) {
                            const int a = lane * E + k;
                            if (a < SIZE) {
                                const unsigned long long v = coh_load_tagged(res + a);
                                mine = mine && (int)(v >> 32) == step + 1;
                                pr_pre[k] = __int_as_float((int)(unsigned)v);
                            }
                        }
                        const unsigned long long vv = coh_load_tagged(res + SIZE);
                        mine = mine && (int)(vv >> 32) == step + 1;
                        v_pre = __int_as_float((int)(unsigned)vv);
                        if (__all(mine)) { got = true; break; }
                        if (!cluster_keep_waiting(it, stop)) break;
                    }
                    if (!got) {
                        h.err = AZ_ERR_STATE;
                        wput(wr, lane, wi, WF_KIND, AZ_LEAF_NONE);
                    }
                }
                wbackup_body<true>(P, tree, wi, wi, nw, lane, h, wr, nullptr, nullptr, 1, v_pre, pr_pre);
            }
        }
        wchunk_store(P, tree, 0, nw, lane, wr);
        whdr_store(&P.trees[tree], lane, h);
        return;
    }
    // ---- an evaluator team: worker `role - 1` of this tree
    float *F = lds, *W = lds + G::BUF;
    float *red = lds + 2 * G::BUF;                       // [16] reductions of the heads
    float *logit = red + 16;                             // [SIZE]
    float *planes = logit + G::SIZE;                     // [2][SIZE] the leaf
    float *logp = planes + 2 * G::SIZE;                  // [SIZE]
    float *value = logp + G::SIZE;                       // [1]
    int *code = reinterpret_cast<int *>(value + 1);      // [2] 1 evaluate / 0 nothing to do / -1 stop, steps alternate
    int *rows = code + 2;                                // [64] the leaf's bit rows
    const int out = tree * nw + role - 1;
    for (int i = tid; i < 2 * G::BUF; i += G::NTHR) lds[i] = 0.f; // halos stay zero from here on
    aznet::TeamCtx<G> ctx;
    aznet::team_setup<G>(ctx, prm, front, pol, val, wave, lane);
    __syncthreads();
    for (int step = 0; step < per; step++) {
        // (a wavefront that has not read the step's code yet when wavefront 0 is already waiting for the next step
        // finds it untouched in the cell of its parity; by the time the cell is written again every wavefront has
        // passed the barrier of the step in between)
        int *cd = code + (step & 1);
        if (wave == 0) { // every lane waits for its own row
            const unsigned long long *src = leaf_x + (size_t)out * 64 + lane;
            unsigned long long v = 0ull;
            int it = 0, verdict = -1;
            for (;;) {
                v = coh_load_tagged(src);
                if (__all((int)(v >> 34) >= step + 1)) { // (tag >> 2 = the step the row belongs to, counted from 1)
                    const int tag = (int)(v >> 32);
                    verdict = __all((tag >> 2) == step + 1 && (tag & 3) == AZ_LEAF_EXPAND) ? 1 : 0;
                    break;
                }
                if (!cluster_keep_waiting(it, stop)) break;
            }
            rows[lane] = (int)(unsigned)v;
            if (lane == 0) cd[0] = verdict;
        }
        __syncthreads();
        const int verdict = cd[0];
        if (verdict < 0) break; // (workgroup-uniform)
        if (verdict == 1) {
            for (int i = tid; i < 2 * G::SIZE; i += G::NTHR) { // the planes as select writes them (write_board)
                const int plane = i / G::SIZE, p = i - plane * G::SIZE;
                planes[i] = (float)(((unsigned)rows[plane * 32 + p / S] >> (p % S)) & 1u);
            }
            __syncthreads();
#ifdef AZ_CLUSTER_NOEVAL // (timing experiment: the protocol without the network)
            for (int i = tid; i < G::SIZE; i += G::NTHR) logp[i] = -5.4f;
            if (tid == 0) value[0] = 0.f;
#else
            aznet::team_position<G>(ctx, F, W, red, logit, planes, value, logp, tid, wave, lane);
#endif
            __syncthreads();
            unsigned long long *res = result_x + (size_t)out * CLUSTER_RESULT;
            for (int i = tid; i < G::SIZE; i += G::NTHR) coh_store_tagged(res + i, step + 1, (unsigned)__float_as_int(logp[i]));
            if (tid == 0) coh_store_tagged(res + G::SIZE, step + 1, (unsigned)__float_as_int(value[0]));
            // (logp and value are written again only after the next step's barriers)
        }
    }
}

// first maximum of the root's visit counts, -1 for an idle tree or an unexpanded root
template <int E> __device__ __forceinline__ int greedy_action(const AzParams &P, const AzTree *T, int tree, int lane)
{
    const int root = uni(T->root);
    if (uni(T->active) == 0 || root < 0) return -1;
    int n[E];
    ld_i32<E>(rec_ptr(P, tree, root), OFF_N, lane, n);
    int best = -1, bk = 0;
#pragma unroll
    for (int k = 0; k < E; k++)
        if (lane * E + k < SIZE && n[k] > best) { best = n[k]; bk = k; }
    const int m = wave_max_i32(best);
    const unsigned long long mask = __ballot(best == m);
    const int wl = __ffsll((long long)mask) - 1;
    return wl * E + lane_bcast_i32(bk, wl);
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
// d_actions == nullptr: every active tree plays the first maximum of its root's
// visit counts (pi.argmax() followed by make_move, src/eval.py:192-193)
__global__ __launch_bounds__(64) void k_make_move(AzParams P, const int *d_actions)
{
    constexpr int E = EPL;
    extern __shared__ unsigned s_marks[];
    const int tree = blockIdx.x;
    const int lane = threadIdx.x;
    AzTree *T = &P.trees[tree];
    const int a = d_actions != nullptr ? uni(d_actions[tree]) : greedy_action<E>(P, T, tree, lane);
    if (a < 0) return;
    WState s;
    load_state<E>(P, tree, lane, s);
    // State::is_valid (state.cpp:29-37)
    bool ok = a < SIZE;
    if (ok) {
        const unsigned long long occupied = __ballot(lane == owner_of<E>(a) && ((s.occ >> (slot_of<E>(a))) & 1u));
        ok = occupied == 0ull;
    }
    if (!ok) {
        if (lane == 0 && T->err == 0) T->err = AZ_ERR_INVALID_MOVE;
        return;
    }
    state_move<E>(s, a, lane);
    P.rows[tree * 64 + lane] = s.rb;

    const size_t tbase = (size_t)tree * P.cap;
    const int root = uni(T->root);
    int new_root = -1, new_root_n = -1;
    if (root >= 0) {
        const char *rec = rec_ptr(P, tree, root);
        new_root = uni(reinterpret_cast<const int *>(rec + OFF_LINK)[a]) - 1;
        new_root_n = uni(reinterpret_cast<const int *>(rec + OFF_N)[a]) - 1;
        if (new_root >= 0) new_root_n -= uni(P.aux[tbase + new_root]); // (0 unless concurrent workers met in it)
    }
    if (new_root >= 0 && new_root_n == 0) { // a never-selected node becomes the root: materialise it
        NodeRegs<E> z;
#pragma unroll
        for (int k = 0; k < E; k++) { z.n[k] = 0; z.link[k] = 0; z.w[k] = 0.; }
        store_stats<E>(P, tree, new_root, lane, z);
    }
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
            P.aux[tbase + idx] = 0;
        }
        kept += __popcll(keep);
        free_top += __popcll(dead);
    }
    if (lane == 0) {
        if (new_root >= 0) P.parent[tbase + new_root] = -1;
        T->root = new_root;
        T->root_n = new_root >= 0 ? new_root_n : -1;
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
        P.aux[tbase + idx] = 0;
    }
    P.rows[tree * 64 + lane] = 0u;
    if (lane == 0) {
        T->root = -1;
        T->root_n = -1;
        T->n_order = 0;
        T->free_top = free_top + n;
        T->player = 0;
        T->move_cnt = 0;
        T->winner = -1.f;
        T->err = 0;
        T->flags = 0;
        T->noise_row = 0;
        T->leaf_kind = AZ_LEAF_NONE;
        T->path_len = 0;
    }
}

// What a game loop does once a game is over (src/self_play.py:86-91: mcts.clear(),
// then tools.make_init_moves, src/tools.py:41-44), for every tree whose game is
// over, in one launch: Cmcts::clear as above, then the opening moves.  On the
// empty tree Cmcts::make_move (cmcts.cpp:284-294) only changes the state.
__global__ __launch_bounds__(64) void k_restart(AzParams P, const int *d_openings, int n_plies, int *d_count)
{
    constexpr int E = EPL;
    const int tree = blockIdx.x;
    const int lane = threadIdx.x;
    AzTree *T = &P.trees[tree];
    if (unif(T->winner) == -1.f) return;
    // a tree in error keeps its error (and its finished game) until the caller
    // has seen it and cleared the tree: a restart must not hide a failed search
    if (uni(T->err) != 0) return;
    const size_t tbase = (size_t)tree * P.cap;
    const int n = uni(T->n_order);
    const int free_top = uni(T->free_top);
    for (int i = lane; i < n; i += 64) {
        const int idx = P.order[tbase + i];
        P.free_stack[tbase + free_top + i] = idx;
        P.parent[tbase + idx] = -2;
        P.aux[tbase + idx] = 0;
    }
    WState s;
    s.rb = 0u; s.occ = 0u; s.player = 0; s.move_cnt = 0; s.winner = -1.f;
    int err = 0;
    for (int ply = 0; ply < n_plies; ply++) {
        const int a = uni(d_openings[(size_t)ply * P.n_trees + tree]);
        if (a < 0) continue;
        bool ok = a < SIZE; // State::is_valid (state.cpp:29-37)
        if (ok) ok = __ballot(lane == owner_of<E>(a) && ((s.occ >> (slot_of<E>(a))) & 1u)) == 0ull;
        if (!ok) { err = AZ_ERR_INVALID_MOVE; break; }
        state_move<E>(s, a, lane);
    }
    P.rows[tree * 64 + lane] = s.rb;
    if (lane == 0) {
        T->root = -1;
        T->root_n = -1;
        T->n_order = 0;
        T->free_top = free_top + n;
        T->player = s.player;
        T->move_cnt = s.move_cnt;
        T->winner = s.winner;
        T->err = err;
        T->flags = 0;
        T->noise_row = 0;
        T->leaf_kind = AZ_LEAF_NONE;
        T->path_len = 0;
        if (d_count != nullptr) atomicAdd(d_count, 1);
    }
}

// first-maximum arg-max of the root visit counts (pi.argmax(), src/eval.py:192)
__global__ __launch_bounds__(64) void k_select_greedy(AzParams P, int *d_actions)
{
    constexpr int E = EPL;
    const int tree = blockIdx.x;
    const int lane = threadIdx.x;
    const int a = greedy_action<E>(P, &P.trees[tree], tree, lane);
    if (lane == 0) d_actions[tree] = a;
}

// Node::counts / nodeN of every root (node.cpp:160-164) into dense arrays
__global__ __launch_bounds__(64) void k_get_counts(AzParams P, int *d_counts, int *d_root_n)
{
    constexpr int E = EPL;
    const int tree = blockIdx.x;
    const int lane = threadIdx.x;
    const AzTree *T = &P.trees[tree];
    const int root = uni(T->root);
    int n[E];
#pragma unroll
    for (int k = 0; k < E; k++) n[k] = 0;
    if (root >= 0) ld_i32<E>(rec_ptr(P, tree, root), OFF_N, lane, n);
    if (d_counts != nullptr) {
#pragma unroll
        for (int k = 0; k < E; k++) {
            const int a = lane * E + k;
            if (a < SIZE) d_counts[(size_t)tree * SIZE + a] = n[k];
        }
    }
    if (d_root_n != nullptr && lane == 0) d_root_n[tree] = root >= 0 ? T->root_n : -1;
}

// ------------------------------------------------- move choice + sample record
// One move decision of self_play_game / eval_game for every active tree
// (src/self_play.py:157-165, src/eval.py:191-193) without leaving the GPU:
//   pi    = Cmcts::get_prob (mcts/src/cmcts.cpp:323-339): (float)((double)N/nodeN),
//           prob_sum accumulated in index order, "Not a distribution" unless
//           round(prob_sum*1000) == 1000;
//   move  = pi.argmax() (first maximum) and pi := one-hot when `greedy` (or when
//           the tree's u is negative),
//           else numpy.random.choice(SIZE, p=pi) for the uniform number u it is
//           given: cdf = cumsum(p as fp64) in index order, cdf /= cdf[-1],
//           move = number of cdf entries <= u (searchsorted, side='right');
//   board = State::get_board (mcts/src/state.cpp:39-48): plane 0 = side to move.
// The two running sums are sequential on purpose (one lane): their rounding is
// part of the result.
// d_ply (may be null): the record fields go to row d_ply[tree] of the tree's staging rows instead of dense
// arrays: d_board = the staging base, float [n_trees][max_plies][3 * SIZE + 1] (board, pi, r), d_pi unused.
__global__ __launch_bounds__(64) void k_choose_moves(AzParams P, const double *d_u, int greedy, int *d_actions,
                                                     float *d_pi, float *d_board, const int *d_ply, int max_plies)
{
    constexpr int E = EPL;
    const int tree = blockIdx.x;
    const int lane = threadIdx.x;
    AzTree *T = &P.trees[tree];
    if (uni(T->active) == 0 || uni(T->err) != 0) {
        if (lane == 0) d_actions[tree] = -1;
        return;
    }
    __shared__ float s_pi[SP];
    __shared__ double s_cdf[SP];
    __shared__ int s_bad;
    const int root = uni(T->root);
    const double node_n = root >= 0 ? (double)uni(T->root_n) : -1.; // an unexpanded root has nodeN = -1
    int n[E];
#pragma unroll
    for (int k = 0; k < E; k++) n[k] = 0;
    if (root >= 0) ld_i32<E>(rec_ptr(P, tree, root), OFF_N, lane, n);
    float pi[E];
#pragma unroll
    for (int k = 0; k < E; k++) {
        pi[k] = (float)((double)n[k] / node_n);
        s_pi[lane * E + k] = pi[k];
    }
    __syncthreads();
    if (lane == 0) {
        double prob_sum = 0., run = 0.;
        for (int i = 0; i < SIZE; i++) {
            const double p = (double)s_pi[i];
            prob_sum += p;
            run += p;
            s_cdf[i] = run;
        }
        s_bad = round(prob_sum * 1000.) != 1000. ? 1 : 0;
    }
    __syncthreads();
    if (s_bad) {
        if (lane == 0) { T->err = AZ_ERR_NOT_DIST; d_actions[tree] = -1; }
        return;
    }
    // a negative uniform number asks for the greedy rule for this tree alone
    // (games of one batch cross the sampling limit at different times)
    if (!greedy && d_u[tree] < 0.) greedy = 1;
    int move;
    if (greedy) {
        float best = -INFINITY;
        int bk = -1;
#pragma unroll
        for (int k = 0; k < E; k++)
            if (lane * E + k < SIZE && pi[k] > best) { best = pi[k]; bk = k; }
        double m;
        const unsigned long long mask = wave_argmax_f64((double)best, bk >= 0, m);
        const int wl = __ffsll((long long)mask) - 1;
        move = wl * E + lane_bcast_i32(bk, wl);
    } else {
        const double u = d_u[tree];
        const double total = s_cdf[SIZE - 1];
        int cnt = 0;
#pragma unroll
        for (int k = 0; k < E; k++) {
            const int a = lane * E + k;
            cnt += __popcll(__ballot(a < SIZE && s_cdf[a < SIZE ? a : 0] / total <= u));
        }
        move = cnt < SIZE ? cnt : SIZE - 1;
    }
    if (lane == 0) d_actions[tree] = move;
    float *pi_out = d_pi != nullptr ? d_pi + (size_t)tree * SIZE : nullptr;
    float *board_out = d_board != nullptr ? d_board + (size_t)tree * 2 * SIZE : nullptr;
    if (d_ply != nullptr) {
        constexpr int REC = 3 * SIZE + 1;
        const int row = uni(d_ply[tree]);
        if (row < 0 || row >= max_plies) { // (cannot happen for a game of at most SIZE moves: refuse rather than write outside)
            if (lane == 0) { T->err = AZ_ERR_STATE; d_actions[tree] = -1; }
            return;
        }
        board_out = d_board + ((size_t)tree * max_plies + row) * REC;
        pi_out = board_out + 2 * SIZE;
    }
    if (pi_out != nullptr) {
#pragma unroll
        for (int k = 0; k < E; k++) {
            const int a = lane * E + k;
            if (a < SIZE) pi_out[a] = greedy ? (a == move ? 1.f : 0.f) : pi[k];
        }
    }
    if (board_out != nullptr) {
        const unsigned rb = P.rows[tree * 64 + lane];
        const int row = lane & 31, plane = lane >> 5;
        const int out_plane = uni(T->player) == 1 ? 1 - plane : plane;
        if (row < S) {
            float *dst = board_out + out_plane * SIZE + row * S;
#pragma unroll
            for (int c = 0; c < S; c++) dst[c] = (float)((rb >> c) & 1u);
        }
    }
}

// --------------------------------------------------------- Dirichlet noise
// Production replacement of gsl_ran_dirichlet (mcts/src/node.cpp:38): row r of
// the table = SIZE independent Gamma(alpha, 1) draws divided by their sum.
// Counter-based generator: draw j of (seed, row, action) is a hash of those
// four numbers, so rows can be produced in any order and on any device.
// Gamma: Marsaglia-Tsang squeeze with the alpha < 1 boost G(a) = G(a+1) U^(1/a);
// normals by Box-Muller.  Not GSL's stream: validated statistically.
__device__ __forceinline__ double u01(unsigned long long seed, unsigned row, unsigned a, unsigned j)
{
    // (the seed is hashed first: seeds that differ in a few low bits must not give tables that are permutations of each other)
    const unsigned long long x = mix64(mix64(mix64(seed) ^ ((unsigned long long)row << 32 | a)) + 0x9e3779b97f4a7c15ull * ((unsigned long long)j + 1ull));
    return ((double)(x >> 11) + 0.5) * (1.0 / 9007199254740992.0); // in (0, 1)
}
__device__ double gamma_draw(double alpha, unsigned long long seed, unsigned row, unsigned a)
{
    const double a1 = alpha < 1. ? alpha + 1. : alpha;
    const double d = a1 - 1. / 3.;
    const double c = 1. / sqrt(9. * d);
    unsigned j = 0;
    double g = d; // value after 64 rejected rounds (probability below 1e-80): the mode-like d
    for (int it = 0; it < 64; it++) {
        const double u1 = u01(seed, row, a, j), u2 = u01(seed, row, a, j + 1), u3 = u01(seed, row, a, j + 2);
        j += 3;
        const double x = sqrt(-2. * log(u1)) * cos(6.283185307179586 * u2);
        double v = 1. + c * x;
        if (v <= 0.) continue;
        v = v * v * v;
        if (log(u3) < 0.5 * x * x + d - d * v + d * log(v)) { g = d * v; break; }
    }
    if (alpha < 1.) g *= pow(u01(seed, row, a, 0xFFFFFFFFu), 1. / alpha);
    return g;
}
__global__ __launch_bounds__(64) void k_dirichlet(double *table, int rows, double alpha, unsigned long long seed)
{
    constexpr int E = EPL;
    const int row = blockIdx.x;
    const int lane = threadIdx.x;
    if (row >= rows) return;
    double g[E];
    double part = 0.;
#pragma unroll
    for (int k = 0; k < E; k++) {
        const int a = lane * E + k;
        g[k] = a < SIZE ? gamma_draw(alpha, seed, (unsigned)row, (unsigned)a) : 0.;
        part += g[k];
    }
    // butterfly: every lane ends with the same total (fixed association)
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) part += __shfl_xor(part, o);
    double out[E];
#pragma unroll
    for (int k = 0; k < E; k++) out[k] = g[k] / part;
    st_f64<E>(reinterpret_cast<char *>(table + (size_t)row * SP), 0u, lane, out);
}

int launch_status()
{
    return hipGetLastError() == hipSuccess ? AZ_OK : AZ_ERR_HIP;
}

// Wavefronts per tree for the fused kernel.  One wavefront per tree is the
// default: the search is bound by the instruction stream of a single
// wavefront, and splitting a tree over WPT_MAX wavefronts replicates the
// scalar part of that stream in each of them (measured slower on MI355X at
// 1024 trees).  The wide variant stays selectable (AzFusedArgs::waves_per_tree).
int pick_wpt(const AzParams &P, const AzFusedArgs &A)
{
    int wpt = A.waves_per_tree;
    if (wpt <= 0) wpt = 1;
    if (wpt != 1 && wpt != WPT_MAX) wpt = wpt > 1 ? WPT_MAX : 1;
    return wpt;
}

// trees per SIMD from which the compact form is launched (it needs more wavefronts in flight than the lean form's three)
#ifndef AZ_COMPACT_MIN_WAVES
#define AZ_COMPACT_MIN_WAVES 4
#endif
int l_simulate_fused(const AzParams &P, const AzFusedArgs &A, void *stream)
{
    const int wpt = pick_wpt(P, A);
    const dim3 grid(P.n_trees);
    hipStream_t st = (hipStream_t)stream;
    if (wpt != 1) {
        hipLaunchKernelGGL((k_simulate_fused<WPT_MAX, -1, -1>), grid, dim3(64 * WPT_MAX), 0, st, P, A);
        return launch_status();
    }
    const bool identity_mix = P.eps == 0. && P.noise == nullptr;
    const bool uniform = A.eval_kind == AZ_EVAL_UNIFORM;
    // the lean form (see k_simulate_fused) where it measured faster: more trees than the chip has SIMDs,
    // and the hash evaluator at any width; AzFusedArgs::lean overrides the choice
    static const int simds = [] {
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        return 4 * cus;
    }();
    // form: 0 = early load, 1 = lean, 2 = compact (four wavefronts per SIMD): AzFusedArgs::lean overrides the choice
    // Measured on MI355X at 8192 trees (eight per SIMD), compact against lean: uniform evaluator 792 against 771 M sims/s;
    // hash evaluator (whose 64-bit integer arithmetic spills 12 - 24 registers at 128) 543 against 580 M at 15x15 and 170
    // against 276 M at 20x20: the compact form is the launcher's choice for the uniform evaluator only.
    const int form = A.lean >= 0 ? A.lean
                     : !uniform ? 1
                     : P.n_trees >= AZ_COMPACT_MIN_WAVES * simds ? 2
                     : P.n_trees > simds ? 1 : 0;
#define AZ_FUSED_LAUNCH(EV, MX)                                                                                   \
    do {                                                                                                          \
        if (form == 2) hipLaunchKernelGGL((k_simulate_fused_c<EV, MX>), grid, dim3(64), 0, st, P, A);             \
        else if (form == 1) hipLaunchKernelGGL((k_simulate_fused<1, EV, MX, true>), grid, dim3(64), 0, st, P, A); \
        else hipLaunchKernelGGL((k_simulate_fused<1, EV, MX, false>), grid, dim3(64), 0, st, P, A);               \
    } while (0)
    if (uniform && identity_mix) AZ_FUSED_LAUNCH(AZ_EVAL_UNIFORM, 0);
    else if (uniform) AZ_FUSED_LAUNCH(AZ_EVAL_UNIFORM, 1);
    else if (identity_mix) AZ_FUSED_LAUNCH(AZ_EVAL_HASH, 0);
    else AZ_FUSED_LAUNCH(AZ_EVAL_HASH, 1);
#undef AZ_FUSED_LAUNCH
    return launch_status();
}
int l_select(const AzParams &P, float *d_boards, int *d_leaf_kind, unsigned *d_expand_mask, unsigned *d_clear_mask, void *stream)
{
    hipLaunchKernelGGL(k_select, dim3(P.n_trees), dim3(64), 0, (hipStream_t)stream, P, d_boards, d_leaf_kind, d_expand_mask,
                       d_clear_mask);
    return launch_status();
}
int l_backup_select(const AzParams &P, const float *d_value, const float *d_prior, int is_log, float *d_boards,
                    int *d_leaf_kind, unsigned *d_expand_mask, unsigned *d_clear_mask, void *stream)
{
    hipLaunchKernelGGL(k_backup_select, dim3(P.n_trees), dim3(64), 0, (hipStream_t)stream, P, d_value, d_prior, is_log,
                       d_boards, d_leaf_kind, d_expand_mask, d_clear_mask);
    return launch_status();
}
int l_workers_select(const AzParams &P, int nw, float *d_boards, int *d_leaf_kind, unsigned *d_expand_mask, unsigned *d_clear_mask,
                     void *stream)
{
    hipLaunchKernelGGL(k_workers_select, dim3(P.n_trees), dim3(64), 0, (hipStream_t)stream, P, nw, d_boards, d_leaf_kind,
                       d_expand_mask, d_clear_mask);
    return launch_status();
}
int l_workers_backup(const AzParams &P, int nw, const float *d_value, const float *d_prior, int is_log, void *stream)
{
    hipLaunchKernelGGL(k_workers_backup, dim3(P.n_trees), dim3(64), 0, (hipStream_t)stream, P, nw, d_value, d_prior, is_log);
    return launch_status();
}
int l_workers_backup_select(const AzParams &P, int nw, const float *d_value, const float *d_prior, int is_log, float *d_boards,
                            int *d_leaf_kind, unsigned *d_expand_mask, unsigned *d_clear_mask, void *stream)
{
    hipLaunchKernelGGL(k_workers_backup_select, dim3(P.n_trees), dim3(64), 0, (hipStream_t)stream, P, nw, d_value, d_prior, is_log,
                       d_boards, d_leaf_kind, d_expand_mask, d_clear_mask);
    return launch_status();
}
int l_net_search(const AzParams &P, int th_num, int per, const float *d_params, int front, int pol, int val, void *stream)
{
    const size_t lds_bytes = (size_t)NET_SEARCH_LDS_FLOATS * sizeof(float);
    static bool configured[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64 || !configured[dev]) {
        if (hipFuncSetAttribute((const void *)k_net_search, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess)
            return 1;
        if (dev >= 0 && dev < 64) configured[dev] = true;
    }
    hipLaunchKernelGGL(k_net_search, dim3(P.n_trees), dim3(NetG::NTHR), lds_bytes, (hipStream_t)stream, P, th_num, per, d_params,
                       front, pol, val);
    return launch_status();
}
// returns 0 when launched, 1 when a launch was tried and failed, 2 when this device cannot hold the grid at once
// (the caller then takes the stepped launches)
int l_net_search_workers(const AzParams &P, int nw, int per, const float *d_params, int front, int pol, int val, int *d_leaf_kind,
                         int *d_flags, void *stream)
{
    const size_t lds_bytes = (size_t)NET_SEARCH_LDS_FLOATS * sizeof(float);
    static int blocks_per_cu[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 2;
    if (blocks_per_cu[dev] == 0) {
        if (hipFuncSetAttribute((const void *)k_net_search_workers, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) !=
            hipSuccess)
            return 2;
        int coop = 0, per_cu = 0;
        if (hipDeviceGetAttribute(&coop, hipDeviceAttributeCooperativeLaunch, dev) != hipSuccess || !coop ||
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_net_search_workers, NetG::NTHR, lds_bytes) != hipSuccess)
            per_cu = -1;
        blocks_per_cu[dev] = per_cu > 0 ? per_cu : -1;
    }
    int cus = 0;
    if (blocks_per_cu[dev] < 0 || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 2;
    const int grid = P.n_trees * (nw + 1);
    if (nw > 8 || grid > cus) return 2; // (one workgroup per CU: a team then has its CU's matrix cores to itself)
    AzParams Pv = P;
    void *args[] = {&Pv, &nw, &per, &d_params, &front, &pol, &val, &d_leaf_kind, &d_flags};
    if (hipLaunchCooperativeKernel((const void *)k_net_search_workers, dim3(grid), dim3(NetG::NTHR), args, (unsigned)lds_bytes,
                                   (hipStream_t)stream) != hipSuccess) {
        (void)hipGetLastError();
        return 2;
    }
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
int l_restart(const AzParams &P, const int *d_openings, int n_plies, int *d_count, void *stream)
{
    hipLaunchKernelGGL(k_restart, dim3(P.n_trees), dim3(64), 0, (hipStream_t)stream, P, d_openings, n_plies, d_count);
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
int l_choose_moves(const AzParams &P, const double *d_u, int greedy, int *d_actions, float *d_pi, float *d_board,
                   void *stream)
{
    hipLaunchKernelGGL(k_choose_moves, dim3(P.n_trees), dim3(64), 0, (hipStream_t)stream, P, d_u, greedy, d_actions,
                       d_pi, d_board, nullptr, 0);
    return launch_status();
}
int l_choose_moves_staged(const AzParams &P, const double *d_u, int *d_actions, float *d_stage, int max_plies, const int *d_ply,
                          void *stream)
{
    hipLaunchKernelGGL(k_choose_moves, dim3(P.n_trees), dim3(64), 0, (hipStream_t)stream, P, d_u, 0, d_actions, nullptr, d_stage,
                       d_stage != nullptr ? d_ply : nullptr, max_plies);
    return launch_status();
}
int l_dirichlet(double *d_table, int rows, double alpha, unsigned long long seed, void *stream)
{
    hipLaunchKernelGGL(k_dirichlet, dim3(rows), dim3(64), 0, (hipStream_t)stream, d_table, rows, alpha, seed);
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
        AZ_S, SP, l_simulate_fused, l_select, l_expand_backup, l_backup_select, l_net_search, l_make_move, l_clear, l_select_greedy, l_get_counts,
        l_choose_moves, l_dirichlet, l_restart, l_workers_select, l_workers_backup, l_workers_backup_select, l_choose_moves_staged,
        l_net_search_workers,
    };
    return &ops;
}

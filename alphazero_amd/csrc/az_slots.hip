// az_slots.hip -- device-resident bookkeeping of the callers' game loops.
//
// The reference plays one game after another on two cmcts.mcts objects
// (src/self_play.py:86-117 around self_play_game, src/eval.py:111-149 around eval_game).
// Here the G tree slots of two engines hold G games in flight; what the host loop decides
// between two moves -- "is the game over", "who moves", "start the next game in the freed
// slot", "label the finished game's records" -- is decided by the kernels below from device
// arrays, so a half-move is a fixed sequence of launches and the host never waits for one:
//   game_of[G]   game played in each slot, -1 = idle          ply[G]  half-moves that game has played
//   counters[8]  AZ_SLOT_* (next game index, games finished, records emitted, trees in error, ...)
// Board-shape independent: the kernels only touch the tree headers (AzTree) and plain arrays.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/azmcts.h"
#include "az_common.h"

namespace {

constexpr int SCAN_THREADS = 1024;

__device__ __forceinline__ unsigned long long mix64(unsigned long long x)
{
    x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ull;
    x ^= x >> 27; x *= 0x94d049bb133111ebull;
    x ^= x >> 31;
    return x;
}

// exclusive prefix of `v` over the block's threads in thread order, and the block total
__device__ __forceinline__ int block_exclusive_scan(int v, int *s_wave, int &total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int up = __shfl_up(incl, o);
        if (lane >= o) incl += up;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    int base = 0, all = 0;
    for (int w = 0; w < SCAN_THREADS / 64; w++) {
        const int c = s_wave[w];
        if (w < wave) base += c;
        all += c;
    }
    __syncthreads();
    total = all;
    return base + incl - v;
}

// Idle slots, in slot order, take the next games while games remain (the loop head of
// src/self_play.py:86-91 / src/eval.py:111-116: one more game, on cleared trees).  mask[t] = 1 for the
// slots that start a game (the caller clears those trees and plays the opening); both engines' trees
// of such a slot become active.
__global__ __launch_bounds__(SCAN_THREADS) void k_slots_admit(AzTree *ta, AzTree *tb, int n, int *game_of, int *ply, int *counters,
                                                               int n_games, unsigned char *mask)
{
    __shared__ int s_wave[SCAN_THREADS / 64];
    const int per = (n + SCAN_THREADS - 1) / SCAN_THREADS;
    const int lo = min(n, (int)threadIdx.x * per), hi = min(n, lo + per);
    int idle = 0;
    for (int t = lo; t < hi; t++) idle += game_of[t] < 0;
    int total;
    int rank = block_exclusive_scan(idle, s_wave, total);
    const int next = counters[AZ_SLOT_NEXT_GAME];
    const int room = n_games - next;
    for (int t = lo; t < hi; t++) {
        const bool is_idle = game_of[t] < 0;
        const bool start = is_idle && rank < room;
        mask[t] = start ? 1 : 0;
        if (start) {
            game_of[t] = next + rank;
            ply[t] = 0;
            ta[t].active = 1;
            if (tb) tb[t].active = 1;
        }
        rank += is_idle;
    }
    __syncthreads(); // every thread has read the counter
    if (threadIdx.x == 0) counters[AZ_SLOT_NEXT_GAME] = next + min(total, room > 0 ? room : 0);
}

// actions[t] = the opening move for the slots in `mask`, -1 (leave the tree alone) elsewhere; the move is `action`
// for every game, or per_game[game_of[t]] when per_game is given
__global__ void k_slots_fill_actions(const unsigned char *mask, int n, int action, const int *per_game, const int *game_of, int *actions)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) actions[t] = mask[t] ? (per_game != nullptr ? per_game[game_of[t]] : action) : -1;
}

// Which trees of an engine search this half-move: the live slots whose game has (ply + game * by_game) of the
// given parity (src/eval.py:117,134: the two networks take turns opening, each searches when it is to move;
// src/self_play.py:156,172: mcts0 searches the even half-moves).  parity < 0: every live slot.
__global__ void k_slots_turn(AzTree *trees, int n, const int *game_of, const int *ply, int by_game, int parity)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const int g = game_of[t];
    int on = g >= 0;
    if (on && parity >= 0) on = ((ply[t] + (by_game ? g : 0)) & 1) == parity;
    trees[t].active = on;
}

// The uniform number behind np.random.choice (src/self_play.py:164) for the game in each slot: a hash of
// (seed, game, half-move), so a game's moves do not depend on which slot or batch it is played in.
// Negative = take the first maximum instead (greedy: the game's pair index ply/2 is past tau,
// src/self_play.py:159-162, or the slot is idle).
__global__ void k_slots_uniform(const int *game_of, const int *ply, int n, int tau, unsigned long long seed, double *u)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const int g = game_of[t], p = ply[t];
    double v = -1.;
    if (g >= 0 && (p >> 1) <= tau) {
        // (the seed is hashed first, so that neighbouring seeds -- one per rank, say -- give unrelated streams)
        const unsigned long long x = mix64(mix64(mix64(seed) ^ ((unsigned long long)(unsigned)g << 32 | (unsigned)p)) + 0x9e3779b97f4a7c15ull);
        v = ((double)(x >> 11) + 0.5) * (1.0 / 9007199254740992.0); // in (0, 1)
    }
    u[t] = v;
}

// After a move: a live slot whose game is over (State::winner != -1 in engine a's tree; src/self_play.py:169,185,
// src/eval.py:194,204) reports its outcome and goes idle, the others count one more half-move.  With a record
// arena the finished games also get their place in it, in slot order: emit_off[t] = first record, emit_n[t] =
// number of records (one per half-move played), -1 / 0 for the other slots; k_slots_emit moves the records.
__global__ __launch_bounds__(SCAN_THREADS) void k_slots_finish(AzTree *ta, AzTree *tb, int n, int *game_of, int *ply, int *counters,
                                                                float *outcome, int with_records, long long arena_records,
                                                                int *game_off, int *game_len, int *emit_off, int *emit_n,
                                                                int *emit_game)
{
    __shared__ int s_wave[SCAN_THREADS / 64];
    __shared__ int s_err;
    if (threadIdx.x == 0) s_err = 0;
    __syncthreads();
    const int per = (n + SCAN_THREADS - 1) / SCAN_THREADS;
    const int lo = min(n, (int)threadIdx.x * per), hi = min(n, lo + per);
    int recs = 0, done = 0, bad = 0, live = 0;
    for (int t = lo; t < hi; t++) {
        if (ta[t].err != 0 || (tb && tb[t].err != 0)) bad++;
        if (game_of[t] >= 0) live++;
        if (game_of[t] >= 0 && ta[t].winner != -1.f) { recs += ply[t] + 1; done++; }
    }
    if (bad) atomicAdd(&s_err, bad);
    int total_recs, total_done;
    int at = block_exclusive_scan(recs, s_wave, total_recs);
    (void)block_exclusive_scan(done, s_wave, total_done);
    int total_live;
    (void)block_exclusive_scan(live, s_wave, total_live);
    const int cursor = counters[AZ_SLOT_RECORDS];
    const bool fits = !with_records || (long long)cursor + total_recs <= arena_records;
    for (int t = lo; t < hi; t++) {
        const int g = game_of[t];
        int eo = -1, en = 0;
        if (g >= 0) {
            const float w = ta[t].winner;
            if (w != -1.f) {
                const int nrec = ply[t] + 1;
                outcome[g] = w;
                if (with_records && fits) {
                    eo = cursor + at;
                    en = nrec;
                    game_off[g] = eo;
                    game_len[g] = nrec;
                    emit_game[t] = g;
                }
                at += nrec;
                game_of[t] = -1;
                ta[t].active = 0;
                if (tb) tb[t].active = 0;
            } else {
                ply[t] += 1;
            }
        }
        if (with_records) { emit_off[t] = eo; emit_n[t] = en; }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        counters[AZ_SLOT_FINISHED] += total_done;
        if (with_records && fits) counters[AZ_SLOT_RECORDS] = cursor + total_recs;
        if (with_records && !fits) counters[AZ_SLOT_ARENA_FULL] = 1;
        counters[AZ_SLOT_ERRORS] = s_err; // trees in error right now (a tree in error never finishes its game)
        counters[AZ_SLOT_HALF_MOVES] += 1;
        counters[AZ_SLOT_TREE_MOVES] += total_live;
    }
}

// The finished games' records move from their slot's staging rows to the arena, labelled
// (src/self_play.py:188-200): the records of the even half-moves (mcts0's) first, then the odd ones; r = 1 for
// the half whose index equals State::winner, -1 for the other, 1e-3 everywhere for a draw.
// stage: [n][max_plies][rec] floats, rec = board (2 planes) + pi + r; one workgroup per slot.
__global__ __launch_bounds__(256) void k_slots_emit(const float *stage, long long slot_stride, int rec, const int *emit_off,
                                                     const int *emit_n, const int *emit_game, const float *outcome, float *arena)
{
    const int t = blockIdx.x;
    const int n = emit_n[t];
    if (n <= 0) return;
    const float w = outcome[emit_game[t]];
    const bool decided = w == 0.f || w == 1.f;
    const int n0 = (n + 1) >> 1;
    const float *src = stage + (size_t)t * slot_stride;
    float *dst = arena + (size_t)emit_off[t] * rec;
    for (int p = 0; p < n; p++) {
        const int side = p & 1;
        const float *s = src + (size_t)p * rec;
        float *d = dst + (size_t)((side ? n0 : 0) + (p >> 1)) * rec;
        for (int i = threadIdx.x; i < rec - 1; i += 256) d[i] = s[i];
        if (threadIdx.x == 0) d[rec - 1] = decided ? (w == (float)side ? 1.f : -1.f) : 1e-3f;
    }
}

int status() { return hipGetLastError() == hipSuccess ? 0 : 1; }

} // namespace

int az_slots_launch_admit(AzTree *ta, AzTree *tb, int n, int *game_of, int *ply, int *counters, int n_games, unsigned char *mask,
                          hipStream_t st)
{
    hipLaunchKernelGGL(k_slots_admit, dim3(1), dim3(SCAN_THREADS), 0, st, ta, tb, n, game_of, ply, counters, n_games, mask);
    return status();
}
int az_slots_launch_fill_actions(const unsigned char *mask, int n, int action, const int *per_game, const int *game_of, int *actions,
                                 hipStream_t st)
{
    hipLaunchKernelGGL(k_slots_fill_actions, dim3((n + 255) / 256), dim3(256), 0, st, mask, n, action, per_game, game_of, actions);
    return status();
}
int az_slots_launch_turn(AzTree *trees, int n, const int *game_of, const int *ply, int by_game, int parity, hipStream_t st)
{
    hipLaunchKernelGGL(k_slots_turn, dim3((n + 255) / 256), dim3(256), 0, st, trees, n, game_of, ply, by_game, parity);
    return status();
}
int az_slots_launch_uniform(const int *game_of, const int *ply, int n, int tau, unsigned long long seed, double *u, hipStream_t st)
{
    hipLaunchKernelGGL(k_slots_uniform, dim3((n + 255) / 256), dim3(256), 0, st, game_of, ply, n, tau, seed, u);
    return status();
}
int az_slots_launch_finish(AzTree *ta, AzTree *tb, int n, int *game_of, int *ply, int *counters, float *outcome, int with_records,
                           long long arena_records, int *game_off, int *game_len, int *emit_off, int *emit_n, int *emit_game,
                           hipStream_t st)
{
    hipLaunchKernelGGL(k_slots_finish, dim3(1), dim3(SCAN_THREADS), 0, st, ta, tb, n, game_of, ply, counters, outcome, with_records,
                       arena_records, game_off, game_len, emit_off, emit_n, emit_game);
    return status();
}
int az_slots_launch_emit(const float *stage, long long slot_stride, int rec, int n, const int *emit_off, const int *emit_n,
                         const int *emit_game, const float *outcome, float *arena, hipStream_t st)
{
    hipLaunchKernelGGL(k_slots_emit, dim3(n), dim3(256), 0, st, stage, slot_stride, rec, emit_off, emit_n, emit_game, outcome, arena);
    return status();
}

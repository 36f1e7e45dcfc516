/*
 * host_search.c -- a host in plain C on the C ABI of libazmcts.so (include/azmcts.h), no Python and
 * no torch in the process: what a C / C++ caller of the reference's Cmcts (mcts/src/cmcts.h:25-81,
 * the way mcts/src/main.cpp:7-21 drives it) looks like on this library.
 *
 *   host_search <board edge> <trees> <sims per move> <plies> <hash seed>
 *
 * Every tree opens with its own stone, then per ply: simulate(sims) with the in-kernel hash
 * evaluator, the first maximum of the root's visit counts as the move (pi.argmax(),
 * src/eval.py:192), make_move.  Prints one line per ply and tree:
 *   ply tree root_n move checksum      (checksum = sum of (action + 1) * count over the root)
 * and the engine's counters at the end.  tests/test_gpu_host_c.py replays the same games on the
 * CPU oracle and compares every number.
 *
 * build:  gcc -O2 -Iinclude examples/host_search.c -Lalphazero_amd -lazmcts -L/opt/rocm/lib -lamdhip64
 *         (the library has no DT_NEEDED for the HIP runtime: the host links the one it wants)
 */
#include <stdio.h>
#include <stdlib.h>

#include "azmcts.h"

#define CHECK(call)                                                        \
    do {                                                                   \
        int rc_ = (call);                                                  \
        if (rc_ != AZ_OK) {                                                \
            fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, az_last_error()); \
            return 1;                                                      \
        }                                                                  \
    } while (0)

int main(int argc, char **argv)
{
    if (argc != 6) {
        fprintf(stderr, "usage: %s <board edge> <trees> <sims> <plies> <hash seed>\n", argv[0]);
        return 2;
    }
    const int shape = atoi(argv[1]), trees = atoi(argv[2]), sims = atoi(argv[3]), plies = atoi(argv[4]);
    const unsigned long long seed = strtoull(argv[5], NULL, 10);
    const int size = shape * shape;
    az_engine *e = NULL;
    CHECK(az_create(&e, 0, trees, shape, sims * (plies + 1) + 64, 1.0, 4.0));
    CHECK(az_set_eps(e, 0.0)); /* no noise table: the prior mix is the identity */
    int *counts = malloc(sizeof(int) * (size_t)trees * size), *root_n = malloc(sizeof(int) * trees);
    int *moves = malloc(sizeof(int) * trees);
    for (int t = 0; t < trees; t++) moves[t] = (7 * t + 3) % size; /* distinct first stones */
    CHECK(az_make_move(e, moves, 0, NULL));
    for (int ply = 0; ply < plies; ply++) {
        CHECK(az_simulate_fused(e, sims, AZ_EVAL_HASH, 0.f, 0.f, seed, NULL));
        CHECK(az_sync_check(e, NULL));
        CHECK(az_get_counts(e, counts, root_n));
        for (int t = 0; t < trees; t++) {
            const int *c = counts + (size_t)t * size;
            int best = 0;
            long long checksum = 0;
            for (int a = 0; a < size; a++) {
                if (c[a] > c[best]) best = a;
                checksum += (long long)(a + 1) * c[a];
            }
            moves[t] = best;
            printf("%d %d %d %d %lld\n", ply, t, root_n[t], best, checksum);
        }
        CHECK(az_make_move(e, moves, 0, NULL));
    }
    unsigned long long s[4];
    CHECK(az_get_stats(e, &s[0], &s[1], &s[2], &s[3]));
    printf("stats %llu %llu %llu %llu\n", s[0], s[1], s[2], s[3]);
    printf("version %s\n", az_version());
    free(counts); free(root_n); free(moves);
    az_destroy(e);
    return 0;
}

/*
 * az_oracle.h -- CPU restatement of the reference cmcts search engine.
 *
 * TEST INFRASTRUCTURE ONLY.  This library is the checker the HIP engine is
 * compared against (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline
 * leg).  Nothing in the product package may import, link or call it.
 *
 * It restates, in plain C with a run-time board shape, the algorithm of
 *   /root/reference/mcts/src/state.cpp   (gomoku rules)
 *   /root/reference/mcts/src/node.cpp    (PUCT select, virtual loss, backup, prior mix)
 *   /root/reference/mcts/src/cmcts.cpp   (search, simulate fan-out, re-rooting, get_prob)
 * Every function in az_oracle.c cites the reference lines it follows.
 *
 * Pinning status:
 *   - rules (State): PINNED.  Checked move-for-move against the reference's own
 *     state.cpp compiled unmodified (oracle/_ref, see oracle/Makefile) through
 *     the fixtures in tests/golden/rules_*.json.
 *   - Node / Cmcts arithmetic (select, backup, prior mix, search sequencing):
 *     PARITY UNPINNED by execution.  node.cpp and cmcts.cpp include GSL headers
 *     that this image does not have, so they cannot be built here; the
 *     restatement follows the source text line by line instead.
 *   - Dirichlet noise: the reference draws it from GSL (absent).  Here noise is
 *     an input table: row k is the k-th draw of a worker's generator, which the
 *     reference re-creates (default seed) at every simulate() call
 *     (cmcts.cpp:132-133), so "row index = expansion ordinal within the call"
 *     reproduces its stream structure.
 */
#ifndef AZ_ORACLE_H
#define AZ_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct azo_tree azo_tree;

/* evaluator kinds */
enum {
    AZO_EVAL_UNIFORM  = 0, /* prior = const_prior for every action, value = const_value */
    AZO_EVAL_HASH     = 1, /* integer hash of the two bit planes, exact in fp32        */
    AZO_EVAL_CALLBACK = 2  /* user function, linear or log priors                      */
};

/* callback: board is the raw 2*SIZE bytes (plane 0 = first player, never swapped),
 * writes *value and prior[SIZE]; returns 0 for linear priors, 1 for log priors
 * (log priors go through expf like node.cpp:42). */
typedef int (*azo_eval_fn)(void *ctx, const signed char *board, int player,
                           float *value, float *prior);

typedef struct azo_evaluator {
    int         kind;
    float       const_prior;
    float       const_value;
    unsigned long long hash_seed;
    azo_eval_fn fn;
    void       *ctx;
} azo_evaluator;

/* error codes (0 = ok).  azo_last_error() returns the message of the last
 * failure on the calling thread, worded like the reference's exceptions. */
enum {
    AZO_OK = 0,
    AZO_ERR_INVALID_MOVE   = 1,  /* state.cpp:86   */
    AZO_ERR_BAD_SIMS       = 2,  /* cmcts.cpp:87   */
    AZO_ERR_NOT_DIST       = 3,  /* cmcts.cpp:338  */
    AZO_ERR_EPS_RANGE      = 4,  /* cmcts.cpp:53   */
    AZO_ERR_BAD_XY         = 5,  /* cmcts.cpp:300  */
    AZO_ERR_NO_ACTION      = 6,  /* node.cpp:115 (.at(-1) out of range)        */
    AZO_ERR_NOISE_ROWS     = 7,  /* oracle only: noise table shorter than needed */
    AZO_ERR_INVALID_VALUE  = 8,  /* cmcts.cpp:70   */
    AZO_ERR_NO_PREDICTOR   = 9,  /* cmcts.cpp:90   */
    AZO_ERR_SEARCH         = 10  /* cmcts.cpp:160 "Deleting threads"           */
};
const char *azo_last_error(void);

azo_tree *azo_create(int shape, double alpha, double cpuct);   /* cmcts.cpp:5-21  */
void      azo_destroy(azo_tree *t);                            /* cmcts.cpp:23-28 */
void      azo_clear(azo_tree *t);                              /* cmcts.cpp:30-37 */

int    azo_set_eps(azo_tree *t, double eps);                   /* cmcts.cpp:51-55 */
double azo_get_eps(const azo_tree *t);
void   azo_set_cpuct(azo_tree *t, float cpuct);                /* cmcts.cpp:58    */
float  azo_get_cpuct(const azo_tree *t);                       /* cmcts.cpp:57    */
void   azo_set_threads(azo_tree *t, int threads);              /* cmcts.cpp:61    */
int    azo_get_threads(const azo_tree *t);
/* how the workers of threads > 1 interleave: real threads as in the reference (the default; not
 * reproducible), or the fixed lock-step interleaving described at simulate_lockstep in az_oracle.c
 * (per step: every worker descends to its leaf in worker order, the leaves are evaluated, every
 * worker finishes in worker order) -- one legal execution of cmcts.cpp:98-119, the one the GPU
 * engine's concurrent schedule is compared with */
enum { AZO_SCHEDULE_THREADS = 0, AZO_SCHEDULE_LOCKSTEP = 1 };
void   azo_set_schedule(azo_tree *t, int schedule);
int    azo_get_schedule(const azo_tree *t);
void   azo_set_alpha(azo_tree *t, double alpha);               /* cmcts.cpp:64    */
double azo_get_alpha(const azo_tree *t);                       /* cmcts.cpp:63    */
void   azo_set_alpha_default(azo_tree *t);                     /* cmcts.cpp:75-81 */
int    azo_set_cuda(azo_tree *t, int cuda);                    /* cmcts.cpp:70    */
void   azo_set_player(azo_tree *t, int player);                /* cmcts.cpp:48    */
int    azo_get_player(const azo_tree *t);
int    azo_get_move_cnt(const azo_tree *t);                    /* cmcts.cpp:72    */
float  azo_get_winner(const azo_tree *t);                      /* cmcts.cpp:73    */
int    azo_shape(const azo_tree *t);

/* noise: rows x SIZE doubles, row-major; may be NULL when eps == 0. */
int azo_simulate(azo_tree *t, int n, const azo_evaluator *ev,
                 const double *noise, int noise_rows);          /* cmcts.cpp:83-125 */
int azo_make_move(azo_tree *t, int action);                     /* cmcts.cpp:284-294 */
int azo_make_movexy(azo_tree *t, int x, int y);                 /* cmcts.cpp:296-309 */
int azo_get_prob(const azo_tree *t, float *out);                /* cmcts.cpp:317-341 */
void azo_get_board(const azo_tree *t, float *out);              /* state.cpp:39-48   */

/* introspection used by the parity tests */
long long azo_root_n(const azo_tree *t);
void azo_root_counts(const azo_tree *t, long *out);             /* node.cpp:160-164 */
void azo_root_w(const azo_tree *t, double *out);
void azo_root_p(const azo_tree *t, double *out);
void azo_raw_board(const azo_tree *t, signed char *out);        /* 2*SIZE bytes */
/* walk from the root along path[0..len); returns 0 and fills the arrays of the
 * node reached, or 1 if a link on the way is null (node.cpp:136-145). */
int azo_node_at(const azo_tree *t, const int *path, int len, long long *node_n,
                long *counts, double *w, double *p);
/* counters since creation / clear */
void azo_stats(const azo_tree *t, long long *sims, long long *expansions,
               long long *terminals, long long *select_calls);
long long azo_live_nodes(const azo_tree *t);

/* stand-alone pieces exposed for unit parity tests */
/* rules: plays `moves` from the empty board; per ply writes player, move_cnt,
 * winner; stops at the first invalid move and returns its ply index, else n. */
int azo_rules_play(int shape, const int *moves, int n, int *player, int *move_cnt,
                   float *winner);
int azo_rules_play_forced(int shape, const int *moves, const int *force, int n, int *player,
                          int *move_cnt, float *winner);
/* select on explicit arrays (node.cpp:85-124): returns the action or -1. */
int azo_select_arrays(int size, const long *n, const double *w, const double *p,
                      long long node_n, const signed char *board, double cpuct);
/* hash evaluator on a raw board (2*SIZE bytes) */
void azo_hash_eval(int shape, unsigned long long seed, const signed char *board,
                   float *value, float *prior);

/* distinct two-ply opening of game g (shared by the baseline and the GPU bench) */
void azo_opening(int size, int g, int *a0, int *a1);

/* CPU baseline legs for bench.py: `workers` host threads play `games` games
 * (two-ply opening, then per ply simulate(sims) + greedy move, finished games
 * restarted), each tree searched with `tree_threads` threads (<= 1: sequential;
 * > 1: the shared-tree fan-out of cmcts.cpp:98-119).  The first `skip` plies of a
 * game are untimed.  Returns the timed searches; *rate = sum over workers of
 * searches per second, *secs = longest worker's timed span. */
long long azo_baseline_run(int shape, int games, int workers, int tree_threads, int skip, int plies,
                           int sims, double cpuct, const azo_evaluator *ev, double *rate, double *secs);

#ifdef __cplusplus
}
#endif
#endif

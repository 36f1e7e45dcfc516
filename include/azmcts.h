/*
 * azmcts.h -- C ABI of the MI355X batched MCTS engine (libazmcts.so).
 *
 * One engine holds `n_trees` independent gomoku search trees (one reference
 * `cmcts.mcts` object each) as structure-of-arrays node pools in HBM of ONE
 * GPU.  Every entry point takes plain pointers and sizes; pointers named
 * `d_*` are device addresses on the engine's GPU, all others are host memory.
 * `stream` arguments are a hipStream_t passed as void* (NULL = default stream).
 * Entry points WITHOUT a stream argument that touch device state (setters of
 * per-tree state, az_clear*, the az_get_* read-backs) first wait for all work
 * THIS ENGINE has queued, on whatever streams (an event per stream, recorded
 * behind every asynchronous call), so they are ordered after its searches and
 * moves; work of other engines, of the caller or of a collective library on
 * the same GPU is not waited for.
 *
 * Each function cites the reference interface it replaces
 * (paths relative to the reference repository root).
 *
 * Return value: 0 on success, an AZ_ERR_* code otherwise; az_last_error()
 * returns the message for the calling thread, worded like the reference's
 * std::runtime_error texts so a Python facade can raise RuntimeError with it.
 */
#ifndef AZMCTS_H
#define AZMCTS_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct az_engine az_engine;

enum {
    AZ_OK = 0,
    AZ_ERR_INVALID_MOVE  = 1,  /* mcts/src/state.cpp:85-86   "Invalid move"                  */
    AZ_ERR_BAD_SIMS      = 2,  /* mcts/src/cmcts.cpp:86-87   "Invalid input n must be >= 1"  */
    AZ_ERR_NOT_DIST      = 3,  /* mcts/src/cmcts.cpp:336-339 "Not a distribution"            */
    AZ_ERR_EPS_RANGE     = 4,  /* mcts/src/cmcts.cpp:52-53   "Epsilon must be in range <0,1>"*/
    AZ_ERR_BAD_XY        = 5,  /* mcts/src/cmcts.cpp:299-300 "Invalid index y.. or x.."      */
    AZ_ERR_NO_ACTION     = 6,  /* mcts/src/node.cpp:115      childN.at(-1) / "No action chosen" */
    AZ_ERR_NOISE_ROWS    = 7,  /* engine: noise table shorter than the expansions of a call  */
    AZ_ERR_INVALID_VALUE = 8,  /* mcts/src/cmcts.cpp:70      "Invalid value"                 */
    AZ_ERR_NO_PREDICTOR  = 9,  /* mcts/src/cmcts.cpp:89-91   "Predictor missing!"            */
    AZ_ERR_POOL_FULL     = 10, /* engine: a tree ran out of node-pool slots                  */
    AZ_ERR_ARG           = 11, /* engine: bad argument (shape, tree index, null pointer)     */
    AZ_ERR_HIP           = 12, /* engine: a HIP runtime call failed                          */
    AZ_ERR_STATE         = 13  /* engine: call sequence violated (expand without select ...) */
};

/* evaluator kinds of az_simulate_fused (test / isolation evaluators computed
 * inside the kernel; a neural net goes through az_select + az_expand_backup) */
enum {
    AZ_EVAL_UNIFORM = 0, /* prior = const_prior everywhere, value = const_value        */
    AZ_EVAL_HASH    = 1  /* integer hash of the two bit planes, exact in fp32          */
};

/* leaf kinds written by az_select */
enum {
    AZ_LEAF_NONE     = 0, /* tree inactive or in error: no work this step              */
    AZ_LEAF_EXPAND   = 1, /* needs (value, priors) for the board written to d_boards   */
    AZ_LEAF_TERMINAL = 2  /* search ended on a finished game: no evaluation needed     */
};

const char *az_last_error(void);

/* replaces the module-level functions of mcts/src/module.cpp:6-16 */
const char *az_version(void);          /* module.cpp:15  version()          */
const char *az_build_timestamp(void);  /* module.cpp:16  build_timestamp()  */

/* Cmcts::Cmcts (mcts/src/cmcts.cpp:5-21), for n_trees objects at once.
 * shape: board edge (5..20; the reference's -DSHAPE macro, mcts/CMakeLists.txt:36).
 * nodes_per_tree: node-pool slots per tree (each slot holds the dense child
 * statistics of one position).  alpha/cpuct as in the reference constructor;
 * eps starts at 0.25 and threads at 0 like cmcts.cpp:8-9. */
int az_create(az_engine **out, int device, int n_trees, int shape, int nodes_per_tree,
              double alpha, double cpuct);
/* Cmcts::~Cmcts (cmcts.cpp:23-28) */
void az_destroy(az_engine *e);

int az_n_trees(const az_engine *e);
int az_shape(const az_engine *e);
int az_nodes_per_tree(const az_engine *e);
/* bytes of HBM the engine allocated */
unsigned long long az_device_bytes(const az_engine *e);

/* setters / getters: mcts/src/cmcts.cpp:46-81 (module.cpp:19-58).  They apply
 * to every tree of the engine. */
int    az_set_eps(az_engine *e, double eps);          /* cmcts.cpp:51-55, range-checked */
double az_get_eps(const az_engine *e);
int    az_set_cpuct(az_engine *e, float cpuct);       /* cmcts.cpp:58 (float in, double stored) */
float  az_get_cpuct(const az_engine *e);              /* cmcts.cpp:57 */
int    az_set_alpha(az_engine *e, double alpha);      /* cmcts.cpp:64 */
double az_get_alpha(const az_engine *e);              /* cmcts.cpp:63 */
int    az_set_alpha_default(az_engine *e);            /* cmcts.cpp:75-81 */
int    az_set_threads(az_engine *e, int threads);     /* cmcts.cpp:61 */
int    az_get_threads(const az_engine *e);
/* How az_simulate_net and the az_*_workers entry points run the `threads` workers of a tree
 * (cmcts.cpp:98-119; in the reference they are std::threads and their interleaving is not
 * reproducible):
 *   AZ_WORKERS_CONCURRENT (default)  in lock step: per step every worker descends to its leaf in
 *       worker order under virtual loss, the leaves of all workers and trees are evaluated in one
 *       batch, every worker finishes (set_prior + backup) in worker order.  simulate(n) then
 *       takes ceil(n / threads) evaluator launches instead of n.
 *   AZ_WORKERS_SEQUENTIAL            one worker after another, each running its searches alone.
 * Both are legal executions of the reference's thread fan-out, with its rounding of the
 * per-worker count and its per-worker noise stream.  az_simulate_fused always runs the
 * sequential schedule.  (No counterpart in the reference: it has one, unnamed, schedule.) */
enum { AZ_WORKERS_SEQUENTIAL = 0, AZ_WORKERS_CONCURRENT = 1 };
#define AZ_MAX_WORKERS 64
int    az_set_schedule(az_engine *e, int schedule);
int    az_get_schedule(const az_engine *e);
int    az_set_cuda(az_engine *e, int cuda);           /* cmcts.cpp:70: always fails, like the reference */
int    az_get_cuda(const az_engine *e);               /* cmcts.cpp:69 */

/* per-tree state: cmcts.cpp:47-48,72-73 */
int az_set_player(az_engine *e, int tree, int player);
int az_get_state(az_engine *e, int tree, int *player, int *move_cnt, float *winner);
/* all trees at once (host arrays of n_trees; any may be NULL) */
int az_get_states(az_engine *e, int *player, int *move_cnt, float *winner);

/* Cmcts::clear (cmcts.cpp:30-37) for one tree (tree >= 0) or all (tree < 0) */
int az_clear(az_engine *e, int tree);
/* clear the trees whose mask byte is non-zero (host array of n_trees) */
int az_clear_masked(az_engine *e, const unsigned char *mask);
/* pi.argmax() followed by make_move (src/eval.py:192-193, the greedy branch of
 * src/self_play.py:159-165) for every active tree with an expanded root, in one
 * launch: the move is the first maximum of the root's visit counts.
 * Asynchronous on `stream`. */
int az_make_move_greedy(az_engine *e, void *stream);
/* What a game loop does with a finished game (src/self_play.py:86-91: mcts.clear()
 * then tools.make_init_moves, src/tools.py:41-44) for every tree whose game is
 * over, without a host round trip: Cmcts::clear, then the moves
 * d_openings[ply][tree] (device int32 [n_plies][n_trees]) for ply = 0..n_plies-1.
 * *d_count (device int, may be NULL) is increased by the number of such trees.
 * Asynchronous on `stream`; errors surface at az_sync_check. */
int az_restart_finished(az_engine *e, const int *d_openings, int n_plies, int *d_count, void *stream);

/* which trees take part in simulate / select (host array of n_trees bytes;
 * NULL = all).  Trees left out keep their state untouched. */
int az_set_active(az_engine *e, const unsigned char *mask);

/* Dirichlet noise as an input table: rows x SIZE doubles, row-major, host
 * memory.  Row k is mixed into the k-th expansion of a simulate() call of a
 * tree -- the reference re-creates a default-seeded generator per call
 * (cmcts.cpp:132-133), so every call replays the same stream.  rows = 0 removes
 * the table (then eps must be 0).  Replaces gsl_ran_dirichlet at node.cpp:38. */
int az_set_noise_table(az_engine *e, const double *noise, int rows);
/* Production replacement of the gsl_ran_dirichlet call of Node::set_prior
 * (mcts/src/node.cpp:38): fills the engine's noise table on the GPU with `rows`
 * rows of Dirichlet(alpha, ..., alpha) over the SIZE actions (alpha = the
 * engine's, cmcts.cpp:56-81), from a counter-based generator keyed by `seed`
 * (Marsaglia-Tsang gamma draws normalised by their sum).  Same seed, same
 * table; it is not GSL's stream.  Row k feeds the k-th expansion of a simulate
 * call, like a table given to az_set_noise_table. */
int az_fill_noise_dirichlet(az_engine *e, int rows, unsigned long long seed, void *stream);
/* copies the first `rows` rows of the current noise table to host memory
 * [rows][SIZE] (tests and diagnostics) */
int az_get_noise_table(az_engine *e, double *noise, int rows);

/* Cmcts::simulate (cmcts.cpp:83-125) with the evaluator computed in-kernel:
 * n searches on every active tree, whole call in one launch.  Thread fan-out
 * rounding of cmcts.cpp:98-107 is applied when threads > 1 (workers run one
 * after another, each restarting the noise stream).  A tree in which concurrent
 * workers have left visit deficits (see az_set_schedule; only az_simulate_net and
 * az_*_workers do) is refused with AZ_ERR_STATE until it is cleared. */
int az_simulate_fused(az_engine *e, int n, int eval_kind, float const_prior, float const_value,
                      unsigned long long hash_seed, void *stream);

/* One search step for an external evaluator (the batched policy/value net):
 *   az_begin_simulate        once per simulate() call: restarts the noise stream
 *   az_select                Cmcts::search up to the leaf (cmcts.cpp:189-268):
 *                            writes, per tree, the leaf kind and the raw
 *                            un-swapped board as float [n_trees][2][S][S]
 *                            (the tensor of cmcts.cpp:206-210) to device memory
 *   az_expand_backup         set_prior + backup (node.cpp:26-47, cmcts.cpp:228-279):
 *                            d_value  float [n_trees]  (net output, not negated)
 *                            d_prior  float [n_trees][SIZE]; log-probabilities
 *                            if prior_is_log (expf applied, node.cpp:42), else
 *                            linear probabilities */
int az_begin_simulate(az_engine *e, void *stream);
int az_select(az_engine *e, float *d_boards, int *d_leaf_kind, void *stream);
int az_expand_backup(az_engine *e, const float *d_value, const float *d_prior, int prior_is_log,
                     void *stream);
/* The same step for n_workers concurrent workers per tree (threads > 1 with the concurrent
 * schedule; n_workers <= the engine's threads): all arrays hold n_workers entries per tree,
 * worker w of tree t at index t * n_workers + w.  az_select_workers runs the workers' descents in
 * worker order (every visit and virtual loss, node.cpp:115-118, is visible to the next worker;
 * Node::next_node, node.cpp:126-134, creates the child during the descent, so later workers of the
 * step may stop at the same unexpanded node); az_expand_backup_workers finishes them in worker
 * order (set_prior is skipped for a node an earlier worker of the step has expanded,
 * node.cpp:30-33).  az_begin_simulate restarts every worker's noise stream. */
int az_select_workers(az_engine *e, int n_workers, float *d_boards, int *d_leaf_kind, void *stream);
int az_expand_backup_workers(az_engine *e, int n_workers, const float *d_value, const float *d_prior, int prior_is_log,
                             void *stream);

/* Cmcts::simulate (cmcts.cpp:83-125) with the reference network as the predictor
 * (the torch::jit module of cmcts.cpp:137,217): n searches on every active tree,
 * each search step = az_select, az_net_eval on the step's leaves, az_expand_backup,
 * all queued on `stream` by this one call (the loop of cmcts.cpp:151-154 without a
 * return to the caller per search; threads > 1: the workers in lock step or one after
 * another, see az_set_schedule).  An
 * engine of at most one tree per CU runs the whole call as ONE launch (a workgroup
 * per tree selects, evaluates and backs up; same arithmetic, same results), and so
 * does an engine with concurrent workers (at most 8) whose every tree can have
 * threads + 1 CUs to itself (a master workgroup and one evaluator team per worker).
 * AZ_NET_SEARCH=0 in the environment keeps the stepped launches.
 * d_params / *_layers: as az_net_eval takes them.  The step's leaf boards, codes,
 * values and log-priors live in a buffer the engine allocates on first use.
 * Asynchronous; errors surface at az_sync_check. */
int az_simulate_net(az_engine *e, int n, const float *d_params, int front_layers, int policy_layers,
                    int value_layers, void *stream);
/* Which launch sequence the engine's last az_simulate_net / az_simulate_net_noalloc call took
 * (diagnostics and tests; the results do not depend on it; no counterpart in the reference). */
enum { AZ_NET_PATH_NONE = 0, AZ_NET_PATH_STEPPED = 1, AZ_NET_PATH_ONE_LAUNCH = 2, AZ_NET_PATH_CLUSTER = 3 };
int az_get_net_path(const az_engine *e);
/* az_simulate_net for callers that cannot allocate device memory (code under HIP
 * graph capture, tight serving loops, harnesses that forbid hipMalloc): same
 * arguments, same kernels, same results, but the step staging lives in a fixed
 * 5 MiB buffer inside the library's device image instead of a buffer allocated
 * on first use, and the call makes no hipMalloc / hipFree.  No counterpart in the
 * reference.  Fails with AZ_ERR_ARG when n_trees * (3 * SIZE + 2) * 4 bytes (plus
 * alignment) exceed the buffer (1935 trees at 15x15, 1090 at 20x20 fit).  All
 * engines on one GPU share the buffer: two such calls whose work overlaps in time
 * on one GPU (different streams) race; calls on one stream are safe. */
int az_simulate_net_noalloc(az_engine *e, int n, const float *d_params, int front_layers, int policy_layers,
                            int value_layers, void *stream);

/* Cmcts::make_move (cmcts.cpp:284-294) on every tree with action >= 0
 * (action < 0 = leave the tree alone).  `actions` is a host array unless
 * on_device.  Illegal moves change nothing; with a host array the call fails
 * with AZ_ERR_INVALID_MOVE naming the first offending tree. */
int az_make_move(az_engine *e, const int *actions, int on_device, void *stream);
/* Cmcts::make_movexy (cmcts.cpp:296-309) on one tree */
int az_make_movexy(az_engine *e, int tree, int x, int y);

/* argmax of the root visit counts of every tree, first maximum wins (what
 * `pi.argmax()` picks at src/eval.py:192, src/self_play.py:160); -1 for
 * inactive trees.  Result stays on the device for az_make_move(on_device=1). */
int az_select_greedy(az_engine *e, int *d_actions, void *stream);

/* One move decision of self_play_game / eval_game for every active tree, on the
 * GPU (src/self_play.py:157-165,172-181; src/eval.py:191-193,201-203):
 *   pi   = Cmcts::get_prob (mcts/src/cmcts.cpp:317-341), incl. its "Not a
 *          distribution" check (reported by az_sync_check with the tree index);
 *   move = greedy ? pi.argmax() (first maximum), pi := one-hot
 *                 : numpy.random.choice(SIZE, p=pi) evaluated for the uniform
 *                   number d_u[tree] in [0,1): cdf = cumsum(p) in fp64 and index
 *                   order, cdf /= cdf[-1], move = #{cdf <= u};
 *   board= State::get_board (mcts/src/state.cpp:39-48), plane 0 = side to move,
 *          taken before the move like the callers do.
 * d_actions [n_trees] receives the move (-1 for inactive trees), d_pi
 * [n_trees][SIZE] and d_board [n_trees][2][S][S] (either may be NULL) receive the
 * training record fields; rows of inactive trees are left untouched.  d_u may
 * be NULL when greedy; with greedy == 0 a negative d_u[tree] selects the greedy
 * rule for that tree alone (games of one batch pass the callers' sampling limit
 * TAU at different times once finished games are replaced by new ones).  The
 * moves are NOT applied: pass d_actions to
 * az_make_move(..., actions_on_device = 1, ...). */
int az_choose_moves(az_engine *e, const double *d_u, int greedy, int *d_actions, float *d_pi, float *d_board,
                    void *stream);

/* root statistics of every tree: counts int32 [n_trees][SIZE] (Node::counts,
 * node.cpp:160-164), root_n int32 [n_trees] (Node::nodeN; -1 = unexpanded).
 * Host arrays; any may be NULL. */
int az_get_counts(az_engine *e, int *counts, int *root_n);
/* Cmcts::get_prob (cmcts.cpp:317-341) for one tree: float32[SIZE]; fails with
 * AZ_ERR_NOT_DIST exactly when the reference throws. */
int az_get_prob(az_engine *e, int tree, float *prob);
/* State::get_board (state.cpp:39-48): float32 [2][S][S], planes swapped when
 * player == 1 so plane 0 is the side to move. */
int az_get_board(az_engine *e, int tree, float *board);
/* raw planes as the evaluator sees them (cmcts.cpp:206-210): int8 [2][S][S] */
int az_get_raw_board(az_engine *e, int tree, signed char *board);

/* the same for every tree at once: int8 [n_trees][2][S][S] (host) */
int az_get_raw_boards(az_engine *e, signed char *boards);

/* introspection for parity tests and print_node (cmcts.cpp:404-417): walks from
 * the root of `tree` along path[0..len).  Returns AZ_OK and fills node_n,
 * counts[SIZE], w[SIZE], p[SIZE] (any may be NULL), or sets *found = 0 when a
 * link on the way is null (Node::is_null, node.cpp:136-145). */
int az_get_node(az_engine *e, int tree, const int *path, int len, int *found, int *node_n,
                int *counts, double *w, double *p);

/* counters since creation / last az_reset_stats, summed over trees:
 * simulations, expansions, terminal hits, select calls (the quantities the
 * reference's unused Stats header names, mcts/src/stats.h:6-25).  Any NULL. */
int az_get_stats(az_engine *e, unsigned long long *sims, unsigned long long *expansions,
                 unsigned long long *terminals, unsigned long long *select_calls);
int az_reset_stats(az_engine *e);
/* live node-pool slots of one tree, and the maximum over all trees */
int az_get_pool_usage(az_engine *e, int tree, int *used, int *max_used);

/* waits for the engine's outstanding work on `stream` and reports the first
 * per-tree error raised inside a kernel, with the number of trees in error.
 * AZ_ERR_INVALID_MOVE and AZ_ERR_NOT_DIST leave their tree untouched and are
 * cleared by the report.  AZ_ERR_POOL_FULL, AZ_ERR_NOISE_ROWS and
 * AZ_ERR_NO_ACTION stop a search half way (the reference throws out of
 * Cmcts::search at the same points, cmcts.cpp:216-240, leaving its tree in the
 * same state): the tree stays in error, is skipped by every later search and is
 * reported by every later check until az_clear / az_clear_masked resets it
 * (az_restart_finished leaves such a tree alone). */
int az_sync_check(az_engine *e, void *stream);
/* the pending error code of every tree after the work on `stream` (host int
 * [n_trees], 0 = none); clears nothing */
int az_get_errors(az_engine *e, void *stream, int *codes);
/* az_sync_check followed by az_get_states from one device-to-host copy: what a
 * caller does after every move (`mcts.winner != -1`, src/self_play.py:169,185;
 * src/eval.py:194,204).  Any of the three output arrays may be NULL. */
int az_sync_states(az_engine *e, void *stream, int *player, int *move_cnt, float *winner);

/* ---- game loops without a host round trip per move.
 * The reference's callers decide between two moves, on the host, whether the game is
 * over, who moves next and when the next game starts (src/self_play.py:86-117 around
 * self_play_game :155-186; src/eval.py:111-149 around eval_game :187-209).  The entry
 * points below keep those decisions in device arrays the caller owns, so a half-move is
 * a fixed sequence of asynchronous calls on one stream and the host only polls the
 * counters, as late as it likes:
 *   d_game_of  int32 [n_trees]  game played in each tree slot, -1 = idle (start: all -1)
 *   d_ply      int32 [n_trees]  half-moves that game has played
 *   d_counters int32 [AZ_SLOT_COUNTERS], start: all 0 (indices AZ_SLOT_*)
 * Engines `a` and `b` hold the two trees of every game (mcts0 / mcts1, or the two
 * networks' trees), slot t of both = the same game; b may be NULL (one tree per game).
 * All of them are asynchronous on `stream`. */
enum {
    AZ_SLOT_NEXT_GAME = 0, /* index the next admitted game gets                             */
    AZ_SLOT_FINISHED = 1,  /* games finished so far                                         */
    AZ_SLOT_RECORDS = 2,   /* records emitted to the arena so far                           */
    AZ_SLOT_ERRORS = 3,    /* trees in error at the last az_slots_finish (their games stall) */
    AZ_SLOT_ARENA_FULL = 4,/* set when a finished game's records did not fit the arena      */
    AZ_SLOT_HALF_MOVES = 5,/* az_slots_finish calls so far                                  */
    AZ_SLOT_TREE_MOVES = 6,/* moves played so far, summed over the slots (live slots per half-move) */
    AZ_SLOT_COUNTERS = 8
};
/* Loop head (src/self_play.py:86-91, src/eval.py:111-116): every idle slot, in slot order,
 * takes the next game while fewer than n_games have been started: Cmcts::clear on its
 * tree(s), then the opening moves `openings[0..n_open)` (host array, the same for every
 * game: tools.make_init_moves with INIT_MOVES, src/tools.py:41-44), then the game's own
 * opening moves d_game_openings[k][game], k < n_game_open (device int32 [n_game_open][n_games],
 * may be NULL; a negative entry plays nothing), ply = 0, trees active. */
int az_slots_admit(az_engine *a, az_engine *b, int *d_game_of, int *d_ply, int *d_counters, int n_games,
                   const int *openings, int n_open, const int *d_game_openings, int n_game_open, void *stream);
/* Which trees of `e` search next: the live slots whose (ply + game * by_game) has the given
 * parity (the arena's networks take turns opening, src/eval.py:117,134; mcts0 searches the
 * even half-moves, src/self_play.py:156,172); parity < 0: every live slot. */
int az_slots_turn(az_engine *e, const int *d_game_of, const int *d_ply, int by_game, int parity, void *stream);
/* az_choose_moves for the games in the slots: the sampling limit is the game's own pair
 * index (ply / 2 <= tau samples, src/self_play.py:159-164), the uniform number of a
 * sampled move is a hash of (seed, game, ply) -- mix64(mix64(mix64(seed) ^ (game << 32 | ply)) +
 * 0x9e3779b97f4a7c15), top 53 bits + 0.5, over 2^53, mix64 = the splitmix64 finaliser -- so
 * a game's moves do not depend on the slot it is played in; tau < 0: always the first
 * maximum (the arena, src/eval.py:192).  d_stage (may be NULL): float [n_trees][max_plies][rec],
 * rec = 3 * SIZE + 1: the record of the move (board, pi; r is filled at the end of the
 * game) is written to row `ply` of the slot.  The moves go to d_actions, not applied. */
int az_slots_choose(az_engine *e, const int *d_game_of, const int *d_ply, int tau, unsigned long long seed,
                    int *d_actions, float *d_stage, int max_plies, void *stream);
/* After the move has been applied to both engines (az_make_move): slots whose game is over
 * (State::winner of a's tree != -1; src/self_play.py:169,185) write d_outcome[game] (float
 * [n_games]) and go idle, the others count the half-move.  With d_stage / d_arena (float
 * [arena_records][rec]) the finished games' records move to the arena as one block per game,
 * mcts0's records first, labelled as src/self_play.py:188-200 (1 / -1 by winner, 1e-3 for a
 * draw); d_game_off / d_game_len (int32 [n_games]) receive the block's first record and
 * length.  Blocks are laid out in the order games finish (slot order inside a half-move). */
int az_slots_finish(az_engine *a, az_engine *b, int *d_game_of, int *d_ply, int *d_counters, float *d_outcome,
                    const float *d_stage, int max_plies, float *d_arena, long long arena_records, int *d_game_off,
                    int *d_game_len, void *stream);

/* HIP-event timing of the engine's own kernels on the stream they were
 * launched on: az_timing_begin records, az_timing_end records + waits and
 * returns elapsed milliseconds. */
int az_timing_begin(az_engine *e, void *stream);
int az_timing_end(az_engine *e, void *stream, float *ms);
/* The same without a wait inside a loop: az_timing_mark records the start
 * (end = 0) or end (end = 1) event of `slot`, az_timing_elapsed waits for the
 * slot's end event and returns the time between the two. */
int az_timing_mark(az_engine *e, int slot, int end, void *stream);
int az_timing_elapsed(az_engine *e, int slot, float *ms);

/* ---- trainer-side sample preparation (src/tools.py:46-88), no engine needed.
 * Samples are the record fields of src/self_play.py:153 as dense device arrays:
 * board float32 [n][2][S][S], pi float32 [n][S*S], r float32 [n]. */

/* get_rotations (src/tools.py:67-88): writes, for every record i in order, its 7
 * derived symmetries as records i*7 + 0..6 (three np.rot90 turns, the row flip
 * of the third, three turns of that); the caller keeps the originals in front,
 * as the reference's np.concatenate([data, rotations]) does. */
int az_aug_rotations(int shape, int n, const float *d_board, const float *d_pi, const float *d_r, float *d_board_out,
                     float *d_pi_out, float *d_r_out, void *stream);
/* get_unique, step 1 (src/tools.py:49): sort keys whose unsigned word-by-word
 * order is np.unique's lexicographic board order.  d_keys [words][n], words =
 * az_aug_key_words(shape); *d_bad is set when a board cell is neither 0 nor 1. */
int az_aug_key_words(int shape);
int az_aug_pack_keys(int shape, int n, const float *d_board, unsigned long long *d_keys, int *d_bad, void *stream);
/* get_unique, step 2 (src/tools.py:56-63): d_members = record indices sorted by
 * (key, index); distinct board u owns d_members[d_seg_start[u] .. d_seg_start[u+1]).
 * Output record u: the board, and pi / r averaged over its occurrences in
 * float32 with numpy's summation orders (a single occurrence is copied). */
int az_aug_unique_average(int shape, int n_unique, const int *d_members, const int *d_seg_start, const float *d_board,
                          const float *d_pi, const float *d_r, float *d_board_out, float *d_pi_out, float *d_r_out,
                          void *stream);

/* ---- batched network evaluation for az_select / az_expand_backup.
 * The reference evaluates its network (`simplerNN`, src/model.py:9-44, traced in
 * eval mode: src/self_play.py:60-63) with batch 1 at every expansion
 * (mcts/src/cmcts.cpp:206-231); this evaluates all n leaf boards of a search
 * step in one launch.  d_boards: float32 [n][2][S][S] raw planes (what az_select
 * writes); outputs d_value [n] and d_logp [n][S*S] (log-softmax, so
 * az_expand_backup takes them with prior_is_log = 1).  d_leaf_kind (may be
 * NULL): the step's AZ_LEAF_* codes as az_select wrote them; boards whose code
 * is not AZ_LEAF_EXPAND are skipped and their outputs left untouched
 * (az_expand_backup does not read them: the reference, too, calls its network
 * only when it expands, cmcts.cpp:190-231).  d_params: the network's
 * inference parameters packed by the host (alphazero_amd/evaluator.py,
 * FusedNetEvaluator.pack): az_net_param_floats(...) floats; 2 input planes and
 * 16 conv channels (config.py:12-15), layer counts as given. */
int az_net_param_floats(int front_layers, int policy_layers, int value_layers);
/* Which kernel az_net_eval runs (no counterpart in the reference; for tuning and for
 * tests that compare the forms): AUTO = the direct-form team kernel for batches of at most one
 * position per CU (two above 15x15), otherwise by board size (one wavefront per position and two
 * per SIMD up to 12x12, the 3x3 layers in the Winograd F(2x2,3x3) form at 12x12; the
 * Winograd F(4x4,3x3) kernels at 13x13 - 16x16, 19x19 and 20x20; the F(2x2,3x3) kernel with one
 * wavefront per SIMD at 17x17 and 18x18); DIRECT / WINOGRAD = that form
 * of the 3x3 layers for every size (one wavefront per position up to 15x15, a team of
 * four above; DIRECT above 15x15 is the same as TEAM); TEAM = the direct-form team kernel
 * for every size; WIDE = the one-wavefront-per-SIMD Winograd kernel from 12x12 on (AUTO's
 * choice below that); W43 = the Winograd F(4x4,3x3) kernels at 13x13 - 16x16 (one wavefront per
 * position) and 17x17 - 20x20 (two), AUTO's choice below 13x13.  Process-wide; initial value from the environment
 * (AZ_NET_FORM). */
enum { AZ_NET_FORM_AUTO = 0, AZ_NET_FORM_DIRECT = 1, AZ_NET_FORM_WINOGRAD = 2, AZ_NET_FORM_TEAM = 3, AZ_NET_FORM_WIDE = 4,
       AZ_NET_FORM_W43 = 5 };
int az_net_set_form(int form);
int az_net_eval(int shape, int n, const float *d_boards, const int *d_leaf_kind, const float *d_params,
                int front_layers, int policy_layers, int value_layers, float *d_value, float *d_logp, void *stream);

#ifdef __cplusplus
}
#endif
#endif

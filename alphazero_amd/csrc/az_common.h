// az_common.h -- data layout shared by the host engine and the gfx950 kernels.
//
// Node pool (per GPU):  n_trees arenas of `cap` fixed-size records.
//   record(tree t, slot i) at nodes + ((size_t)t * cap + i) * REC_BYTES, holding
//   the dense child statistics of ONE position over SP padded actions (SP is a
//   multiple of 64 fixed per board shape, see az_kernels.hip):
//       N    int32  [SP]   visit counts          (Node::childN, node.h:46)
//       LINK int32  [SP]   child slot + 1, 0=null (Node::child,  node.h:44)
//       W    double [SP]   value sums            (Node::childW, node.h:47)
//       P    double [SP]   priors                (Node::childP, node.h:37)
//   Actions are thread-blocked: a group of T threads working on a node gives
//   thread t the actions [t*E, t*E+E), E = SP/T, so every array is read and
//   written with contiguous, vector-width accesses.
//   Node::nodeN (node.h:35) is not stored for inner nodes: a child reached
//   through action a of its parent has nodeN = N_parent[a] - 1 (the first
//   visit of the action expands the child, every later one selects from it);
//   the root's nodeN lives in the tree header.
//   N, LINK and W of a node are materialised lazily: while nodeN == 0 they are
//   implicitly zero and only P is in memory; the first select on the node (or
//   its promotion to root) writes them.
//   parent[t*cap+i]  int32: parent slot, -1 for a root, -2 for a free slot
//   free_stack[t*cap + k], k < free_top: free slots, popped from the top
//   order[t*cap + k],  k < n_order: live slots in allocation order (a parent
//                      always precedes its children), used by re-rooting
// Per tree: AzTree header, plus the board as bitboards in wave layout:
//   rows[t*64 + L]: bit c = stone at (row L&31, col c) of plane L>>5
#ifndef AZ_COMMON_H
#define AZ_COMMON_H

#include <stdint.h>

#define AZ_MAX_SHAPE 20
// board edges compiled into the library (keep in step with SHAPES in the Makefile)
#ifndef AZ_SHAPE_LIST
#define AZ_SHAPE_LIST(X) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15) X(16) X(17) X(18) X(19) X(20)
#endif

struct AzTree {
    int root;        // slot of the root node, -1 while the root is unexpanded (nodeN == -1)
    int root_n;      // Node::nodeN of the root (valid when root >= 0)
    int free_top;    // entries in this tree's free stack
    int n_order;     // live slots
    int player;      // 0/1 to move               (State::player,  state.h:42)
    int move_cnt;    //                           (State::move_cnt, state.h:43)
    float winner;    // -1, 0, 1, 0.5             (State::winner,  state.h:44)
    int err;         // first AZ_ERR_* raised inside a kernel for this tree
    int active;      // takes part in simulate / select
    int noise_row;   // next row of the noise table (restarts per simulate call)
    int leaf_kind;   // AZ_LEAF_* of the pending step (select -> expand_backup)
    int path_len;    // depth of the pending step's path
    float leaf_value;// value of a terminal leaf of the pending step
    int flags;       // AZ_TREE_*
    unsigned long long stats[4]; // sims, expansions, terminals, select calls
};
// A node of this tree carries a visit deficit (aux[slot] > 0): several concurrent workers met in it while it
// was unexpanded, so the visits of its parent's action exceed its own nodeN by more than the one that created it.
#define AZ_TREE_DEFICIT 1

// One concurrent worker of a tree (the reference's std::thread of cmcts.cpp:98-119) between the select and the
// backup of its search: what AzTree's leaf_kind / path_len / leaf_value / noise_row are for the single worker.
struct AzWorker {
    int leaf_kind;    // AZ_LEAF_* of the pending search
    int path_len;
    float leaf_value; // value of a finished game
    int noise_row;    // next row of the worker's own noise stream (restarts per simulate call)
    int leaf_slot;    // node to expand: its slot, -1 = the root
    int pad[3];
};

struct AzParams {
    char *nodes;
    int *parent;
    int *free_stack;
    int *order;
    AzTree *trees;
    unsigned *rows;
    int *path_node;   // [n_trees][path_cap] slots along the pending path
    int *path_act;    // [n_trees][path_cap]
    // aux[t*cap+i]: >= 0: the node's visit deficit (nodeN = N_parent[a] - 1 - deficit; 0 unless concurrent workers met
    // in the node before it was expanded); < 0: the node was created by a worker of the running step and is not
    // expanded yet, -aux - 1 further workers have arrived at it since.  0 for free slots.
    int *aux;
    AzWorker *workers; // [n_trees][max_workers], null until az_set_threads asks for concurrent workers
    int *wpath_node;   // [n_trees][max_workers][path_cap]
    int *wpath_act;
    int max_workers;
    unsigned long long *stamps; // diagnostic builds only (AZ_STAMPS): [n_trees][8] cycle sums, else null
    const double *noise; // [noise_rows][SP] or null
    int noise_rows;
    int n_trees;      // trees of the engine; for a launch over a group of trees: the trees of the group (the grid)
    int tree_base;    // first tree of the group a step kernel is launched for (its blocks handle tree_base + blockIdx.x)
    int cap;
    int path_cap;
    double cpuct;
    double eps;
};

struct AzFusedArgs {
    int waves_per_tree;  // 0 = let the launcher choose from the number of trees
    int lean;            // kernel form (see simulate_fused_body): 0 early load / 1 lean / 2 compact, -1 = let the launcher choose
    int n_total;         // searches per tree in this launch
    int per_worker;      // noise stream restarts every per_worker searches
    int eval_kind;
    float const_prior;
    float const_value;
    unsigned long long hash_seed;
};

// ints of the exchange area of net_search_workers for n positions of `size` actions: the stop word (padded to 64
// ints), then per position 64 tagged bit rows and size + 1 tagged result words, two ints each
#define AZ_CLUSTER_FLAG_INTS(n, size) (64 + 2 * (size_t)(n) * (64 + (size_t)(size) + 1))

// launchers exported by each per-shape translation unit (az_kernels.hip built
// with -DAZ_S=<shape>); the engine picks the table entry of its shape.
struct AzShapeOps {
    int shape;
    int sp;              // padded actions per node record of this shape
    int (*simulate_fused)(const AzParams &, const AzFusedArgs &, void *stream);
    // d_expand_mask / d_clear_mask (both may be null): bit t of the first is set when tree t's leaf is to be
    // expanded; the second, the mask of the step before, is zeroed for the step after
    int (*select)(const AzParams &, float *d_boards, int *d_leaf_kind, unsigned *d_expand_mask, unsigned *d_clear_mask,
                  void *stream);
    int (*expand_backup)(const AzParams &, const float *d_value, const float *d_prior,
                         int prior_is_log, void *stream);
    // expand_backup of one search step and select of the next in one launch
    int (*backup_select)(const AzParams &, const float *d_value, const float *d_prior, int prior_is_log, float *d_boards,
                         int *d_leaf_kind, unsigned *d_expand_mask, unsigned *d_clear_mask, void *stream);
    // the whole network search of a simulate call in one launch (at most one tree per CU): th_num workers of
    // `per` searches each, d_params as az_net_eval takes them
    int (*net_search)(const AzParams &, int th_num, int per, const float *d_params, int front_layers, int policy_layers,
                      int value_layers, void *stream);
    int (*make_move)(const AzParams &, const int *d_actions, void *stream);
    int (*clear)(const AzParams &, const unsigned char *d_mask, void *stream);
    int (*select_greedy)(const AzParams &, int *d_actions, void *stream);
    int (*get_counts)(const AzParams &, int *d_counts, int *d_root_n, void *stream);
    int (*choose_moves)(const AzParams &, const double *d_u, int greedy, int *d_actions, float *d_pi,
                        float *d_board, void *stream);
    int (*dirichlet)(double *d_table, int rows, double alpha, unsigned long long seed, void *stream);
    int (*restart)(const AzParams &, const int *d_openings, int n_plies, int *d_count, void *stream);
    // One step of `nw` concurrent workers per tree (lock-step schedule of the thread fan-out, cmcts.cpp:98-119):
    // the workers' selects in worker order (position tree * nw + w of d_boards / d_leaf_kind / the masks), and the
    // workers' expand + backup in worker order; the third is the backups of a step and the selects of the next.
    int (*workers_select)(const AzParams &, int nw, float *d_boards, int *d_leaf_kind, unsigned *d_expand_mask,
                          unsigned *d_clear_mask, void *stream);
    int (*workers_backup)(const AzParams &, int nw, const float *d_value, const float *d_prior, int prior_is_log, void *stream);
    int (*workers_backup_select)(const AzParams &, int nw, const float *d_value, const float *d_prior, int prior_is_log,
                                 float *d_boards, int *d_leaf_kind, unsigned *d_expand_mask, unsigned *d_clear_mask, void *stream);
    // choose_moves with the record fields written to row d_ply[tree] of the tree's staging rows:
    // d_stage + tree * max_plies * rec + d_ply[tree] * rec, rec = 3 * SIZE + 1 (board, pi, r)
    int (*choose_moves_staged)(const AzParams &, const double *d_u, int *d_actions, float *d_stage, int max_plies,
                               const int *d_ply, void *stream);
    // the whole network search of a simulate call with nw concurrent workers per tree in one launch (a cluster of
    // nw + 1 workgroups per tree, at most one workgroup per CU); d_leaf_kind as the stepped launches use it,
    // d_flags: AZ_CLUSTER_FLAG_INTS(n_trees * nw, SIZE) zeroed ints, 256-byte aligned (the exchange area between the
    // workgroups of a cluster).  0 = launched, 1 = the launch failed, 2 = not possible on this device or for this
    // batch (nothing was launched: take the stepped launches)
    int (*net_search_workers)(const AzParams &, int nw, int per, const float *d_params, int front_layers, int policy_layers,
                              int value_layers, int *d_leaf_kind, int *d_flags, void *stream);
};

// Words of the expand masks are AZ_MASK_STRIDE words apart: the trees' atomic ORs then spread over four times
// as many cache lines (8192 atomics on eight lines cost k_select 5 us), while a reader still fetches only 4 KB.
#define AZ_MASK_STRIDE 4

// the kernel form az_net_eval is set to (AZ_NET_FORM_*; az_net.hip)
int az_net_get_form();

// az_net_eval (az_net.hip) with the bit mask k_select fills (one bit per position, word w at index
// w * AZ_MASK_STRIDE, set = evaluate it; null: the codes are scanned as az_net_eval does).  Internal to the
// library: az_simulate_net's steps.
int az_net_eval_masked(int shape, int n, const float *d_boards, const int *d_leaf_kind, const unsigned *d_expand_mask,
                       const float *d_params, int front_layers, int policy_layers, int value_layers, float *d_value,
                       float *d_logp, void *stream);

// launchers of az_slots.hip (device-resident bookkeeping of the callers' game loops; board-shape independent)
int az_slots_launch_admit(AzTree *ta, AzTree *tb, int n, int *game_of, int *ply, int *counters, int n_games, unsigned char *mask,
                          hipStream_t st);
int az_slots_launch_fill_actions(const unsigned char *mask, int n, int action, const int *per_game, const int *game_of, int *actions,
                                 hipStream_t st);
int az_slots_launch_turn(AzTree *trees, int n, const int *game_of, const int *ply, int by_game, int parity, hipStream_t st);
int az_slots_launch_uniform(const int *game_of, const int *ply, int n, int tau, unsigned long long seed, double *u, hipStream_t st);
int az_slots_launch_finish(AzTree *ta, AzTree *tb, int n, int *game_of, int *ply, int *counters, float *outcome, int with_records,
                           long long arena_records, int *game_off, int *game_len, int *emit_off, int *emit_n, int *emit_game,
                           hipStream_t st);
int az_slots_launch_emit(const float *stage, long long slot_stride, int rec, int n, const int *emit_off, const int *emit_n,
                         const int *emit_game, const float *outcome, float *arena, hipStream_t st);

#endif

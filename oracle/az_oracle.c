/*
 * az_oracle.c -- CPU restatement of the reference cmcts engine (see az_oracle.h).
 * TEST INFRASTRUCTURE ONLY: never linked or imported by the product package.
 *
 * What pins it:
 *   - the rules (state_*: state.cpp) are PINNED: tests/golden/rules_S*.json are answers of the
 *     reference's own state.cpp, compiled unmodified (oracle/Makefile `ref`), and this file must
 *     reproduce them (tests/test_oracle_rules.py);
 *   - the caller loops built on it are pinned by records the reference's own self_play_game /
 *     eval_game produced (tests/golden/caller_games.npz);
 *   - the tree arithmetic (node_*: node.cpp, search / azo_simulate: cmcts.cpp) is PARITY UNPINNED by
 *     execution: node.cpp and cmcts.cpp include GSL headers the build image lacks and cannot be
 *     compiled there.  That part follows the source text line by line (citations at every function)
 *     and reproduces the facts recorded from the real reference (tests/test_oracle_search.py).
 *
 * All reference citations are relative to /root/reference/mcts/src/.
 * Build:  gcc -O2 -ffp-contract=off -fPIC -shared -pthread az_oracle.c -o libaz_oracle.so -lm
 * (-ffp-contract=off: the reference is built without optimisation for baseline
 *  x86-64, so no fused multiply-add may appear in the PUCT expression.)
 */
#define _GNU_SOURCE
#include "az_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#define AZO_MAX_SHAPE 32

static __thread char g_err[512];
const char *azo_last_error(void) { return g_err; }

static int fail(int code, const char *fmt, long a, long b, long c)
{
    snprintf(g_err, sizeof g_err, fmt, a, b, c);
    return code;
}

/* ------------------------------------------------------------------ */
/* State: state.h:20-46                                                */
/* ------------------------------------------------------------------ */
typedef struct {
    int shape, size;
    signed char *board; /* 2*size bytes, plane 0 = first player (state.h:20) */
    int player;         /* 0/1 to move */
    int move_cnt;
    float winner;       /* -1 none, 0/1 plane index of the winner, 0.5 draw */
} az_state;

static void state_init(az_state *s, int shape)
{
    s->shape = shape;
    s->size = shape * shape;
    s->board = (signed char *)calloc((size_t)2 * s->size, 1);
    s->player = 0;       /* state.cpp:3-8 */
    s->move_cnt = 0;
    s->winner = -1.f;
}

static void state_free(az_state *s) { free(s->board); s->board = NULL; }

/* state.cpp:64-79 */
static void state_reset(az_state *s)
{
    s->player = 0;
    s->move_cnt = 0;
    s->winner = -1.f;
    memset(s->board, 0, (size_t)2 * s->size);
}

/* state.cpp:50-62 (clear(const State*)) */
static void state_copy(az_state *dst, const az_state *src)
{
    dst->player = src->player;
    dst->move_cnt = src->move_cnt;
    dst->winner = src->winner;
    memcpy(dst->board, src->board, (size_t)2 * src->size);
}

/* state.cpp:29-37 */
static int state_is_valid(const az_state *s, int a)
{
    if (a >= s->size || a < 0)
        return 0;
    if (s->board[a] == 1 || s->board[s->size + a] == 1)
        return 0;
    return 1;
}

/* all five cells of the window starting at (r,c) with step (dr,dc) hold a
 * stone of `plane`.  A reference sliding sum equals 5 exactly when this is
 * true because cells are 0/1 (state.cpp:166-171, 181-184). */
static int five(const az_state *s, int plane, int r, int c, int dr, int dc)
{
    const signed char *b = s->board + plane * s->size;
    for (int t = 0; t < 5; t++)
        if (b[(r + t * dr) * s->shape + (c + t * dc)] != 1)
            return 0;
    return 1;
}

/* The reference tests, for a pair of windows scanned together, plane 0 on
 * either window first, then plane 1 on either (state.cpp:172-177 etc.). */
#define CHECK_PAIR(r0, c0, dr0, dc0, r1, c1, dr1, dc1)                         \
    do {                                                                       \
        if (five(s, 0, r0, c0, dr0, dc0) || five(s, 0, r1, c1, dr1, dc1))      \
            return s->player == 0 ? 1.f : -1.f;                                \
        if (five(s, 1, r0, c0, dr0, dc0) || five(s, 1, r1, c1, dr1, dc1))      \
            return s->player == 1 ? 1.f : -1.f;                                \
    } while (0)

/* state.cpp:152-272.  Returns +1 / -1 relative to the side to move, 1e-8 for
 * a full board, 0 to continue.  Scan order is the reference's; the "/" scans
 * of both halves walk the SAME anti-diagonals row+col = S-1-j
 * (state.cpp:209-210,222-223 and 243-244,257-258), so fives on anti-diagonals
 * with row+col >= S are never seen -- restated as is. */
static float state_is_end(const az_state *s)
{
    const int S = s->shape;
    if (s->move_cnt < 9) /* state.cpp:158-159 */
        return 0.f;

    /* rows and columns: state.cpp:162-193.  k indexes the window start. */
    for (int j = 0; j < S; j++)
        for (int k = 0; k <= S - 5; k++)
            CHECK_PAIR(j, k, 0, 1, /* row j, cols k..k+4 */
                       k, j, 1, 0  /* col j, rows k..k+4 */);

    /* diagonals: state.cpp:197-267 */
    for (int j = 0; j < S - 4; j++) {
        /* first half, offset j along x: state.cpp:205-231
         *   "\" window rows k..k+4, cols k+j..k+j+4
         *   "/" window rows k..k+4, cols S-1-k-j down to S-5-k-j */
        for (int k = 0; k <= S - 5 - j; k++)
            CHECK_PAIR(k, k + j, 1, 1,
                       k, S - 1 - k - j, 1, -1);
        if (j == 0) /* state.cpp:232-233 */
            continue;
        /* second half, offset j along y: state.cpp:239-266
         *   "\" window rows k+j..k+j+4, cols k..k+4
         *   "/" window rows S-1-k-j up to S-5-k-j, cols k..k+4 */
        for (int k = 0; k <= S - 5 - j; k++)
            CHECK_PAIR(k + j, k, 1, 1,
                       S - 1 - k - j, k, -1, 1);
    }

    if (s->move_cnt == s->size) /* state.cpp:269-270 */
        return 1e-8f;
    return 0.f;
}

/* state.cpp:81-110.  Returns 0, or 1 when the move is invalid. */
static int state_make_move(az_state *s, int a)
{
    if (!state_is_valid(s, a))
        return 1;
    s->board[s->player * s->size + a] = 1;
    s->player = s->player ? 0 : 1;
    s->move_cnt += 1;
    float end = state_is_end(s);
    if (end != 0.f) {
        if (end == -1.f)
            s->winner = s->player == 0 ? 1.f : 0.f;
        else if (end == 1.f)
            s->winner = s->player == 1 ? 1.f : 0.f;
        else
            s->winner = 0.5f; /* draw */
    }
    return 0;
}

/* ------------------------------------------------------------------ */
/* Node: node.h:23-48                                                  */
/* ------------------------------------------------------------------ */
typedef struct az_node {
    long long nodeN; /* -1 = unexpanded (node.cpp:6) */
    pthread_mutex_t mu;
    struct az_node **child;
    long *childN;
    double *childW;
    double *childP;
} az_node;

static long long g_live_nodes; /* debugging aid, relaxed */

static az_node *node_new(int size)
{
    size_t bytes = sizeof(az_node) + (size_t)size * (sizeof(az_node *) + sizeof(long) + 2 * sizeof(double));
    az_node *n = (az_node *)calloc(1, bytes);
    char *p = (char *)(n + 1);
    n->child = (az_node **)p;  p += (size_t)size * sizeof(az_node *);
    n->childN = (long *)p;     p += (size_t)size * sizeof(long);
    n->childW = (double *)p;   p += (size_t)size * sizeof(double);
    n->childP = (double *)p;
    n->nodeN = -1;
    pthread_mutex_init(&n->mu, NULL);
    __atomic_add_fetch(&g_live_nodes, 1, __ATOMIC_RELAXED);
    return n;
}

/* recursive delete, as unique_ptr members do (node.h:44) */
static void node_free(az_node *n, int size)
{
    if (!n)
        return;
    for (int i = 0; i < size; i++)
        node_free(n->child[i], size);
    pthread_mutex_destroy(&n->mu);
    free(n);
    __atomic_sub_fetch(&g_live_nodes, 1, __ATOMIC_RELAXED);
}

/* node.cpp:85-124 */
static int node_select(az_node *n, const az_state *s, double cpuct)
{
    if (__atomic_load_n(&n->nodeN, __ATOMIC_RELAXED) == -1)
        return -2; /* "Node has not been visited yet" (node.cpp:88-89) */
    pthread_mutex_lock(&n->mu);
    int best_a = -1;
    double best_u = -INFINITY;
    const int size = s->size;
    /* std::sqrt(nodeN + 1e-8) is loop-invariant (node.cpp:106,108) */
    const double root = sqrt((double)n->nodeN + 1e-8);
    for (int a = 0; a < size; a++) {
        if (!state_is_valid(s, a))
            continue;
        double u;
        if (n->childN[a] != 0)
            u = n->childW[a] / (double)n->childN[a]
              + cpuct * n->childP[a] * root / (double)(n->childN[a] + 1);
        else
            u = cpuct * n->childP[a] * root;
        if (u > best_u) {
            best_u = u;
            best_a = a;
        }
    }
    if (best_a < 0) { /* childN.at(-1) throws (node.cpp:115) */
        pthread_mutex_unlock(&n->mu);
        return -1;
    }
    n->childN[best_a] += 1;
    n->nodeN += 1;
    n->childW[best_a] -= 1.; /* virtual loss (node.cpp:118) */
    pthread_mutex_unlock(&n->mu);
    return best_a;
}

/* node.cpp:75-83: value is a float, promoted, plus 1., then accumulated */
static void node_backpropagate(az_node *n, int action, float value)
{
    pthread_mutex_lock(&n->mu);
    n->childW[action] += value + 1.;
    pthread_mutex_unlock(&n->mu);
}

/* node.cpp:26-47.  `noise` is the Dirichlet draw for this expansion (NULL
 * stands for "not available": only legal when dir_eps == 0, where it is
 * multiplied by zero anyway).  Returns 1 if this call did the expansion. */
static int node_set_prior(az_node *n, int size, const float *prior, int is_log,
                          const double *noise, double dir_eps)
{
    pthread_mutex_lock(&n->mu);
    if (n->nodeN != -1) { /* some other thread did the job (node.cpp:30-33) */
        pthread_mutex_unlock(&n->mu);
        return 0;
    }
    double prior_eps = 1. - dir_eps;
    for (int i = 0; i < size; i++) {
        double d = noise ? noise[i] : 0.;
        double p = is_log ? (double)expf(prior[i]) : (double)prior[i];
        n->childP[i] = prior_eps * p + dir_eps * d;
    }
    __atomic_store_n(&n->nodeN, 0, __ATOMIC_RELAXED);
    pthread_mutex_unlock(&n->mu);
    return 1;
}

/* node.cpp:126-134 */
static az_node *node_next(az_node *n, int action, int size)
{
    if (n->child[action] == NULL)
        n->child[action] = node_new(size);
    return n->child[action];
}

/* ------------------------------------------------------------------ */
/* hash evaluator (test instrument, exact in fp32)                      */
/* ------------------------------------------------------------------ */
static uint64_t mix64(uint64_t x)
{
    x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ULL;
    x ^= x >> 27; x *= 0x94d049bb133111ebULL;
    x ^= x >> 31;
    return x;
}

void azo_hash_eval(int shape, unsigned long long seed, const signed char *board,
                   float *value, float *prior)
{
    const int size = shape * shape;
    uint64_t key = seed;
    for (int p = 0; p < 2; p++)
        for (int r = 0; r < shape; r++) {
            uint32_t bits = 0;
            for (int c = 0; c < shape; c++)
                if (board[p * size + r * shape + c] == 1)
                    bits |= 1u << c;
            key ^= mix64(((uint64_t)(p * 32 + r + 1) << 32) | bits);
        }
    uint64_t hv = mix64(key ^ 0x9e3779b97f4a7c15ULL);
    *value = (float)((int)(hv & 0xFFFF) - 32768) / 32768.0f;
    for (int i = 0; i < size; i++) {
        uint64_t hi = mix64(key + (uint64_t)(i + 1) * 0x9e3779b97f4a7c15ULL);
        prior[i] = (float)((hi & 0xFFFF) + 1) * (1.0f / (65536.0f * 128.0f));
    }
}

/* ------------------------------------------------------------------ */
/* Cmcts: cmcts.h:25-81                                                */
/* ------------------------------------------------------------------ */
struct azo_tree {
    int shape, size;
    az_node *root;
    az_state state;
    double *alpha;
    double cpuct;   /* stored as double (cmcts.h:76) */
    double dir_eps;
    int threads;
    int schedule; /* AZO_SCHEDULE_*: how the workers of threads > 1 interleave */
    int cuda;
    /* counters */
    long long sims, expansions, terminals, select_calls;
    pthread_mutex_t stat_mu;
};

azo_tree *azo_create(int shape, double alpha, double cpuct)
{
    if (shape < 5 || shape > AZO_MAX_SHAPE)
        return NULL;
    azo_tree *t = (azo_tree *)calloc(1, sizeof *t);
    t->shape = shape;
    t->size = shape * shape;
    t->cpuct = cpuct;
    t->cuda = 0;
    t->dir_eps = 0.25;
    t->threads = 0;
    t->root = node_new(t->size);
    state_init(&t->state, shape);
    t->alpha = (double *)malloc(sizeof(double) * t->size);
    for (int i = 0; i < t->size; i++)
        t->alpha[i] = alpha;
    pthread_mutex_init(&t->stat_mu, NULL);
    return t;
}

void azo_destroy(azo_tree *t)
{
    if (!t)
        return;
    node_free(t->root, t->size);
    state_free(&t->state);
    free(t->alpha);
    pthread_mutex_destroy(&t->stat_mu);
    free(t);
}

void azo_clear(azo_tree *t)
{
    node_free(t->root, t->size);
    t->root = node_new(t->size);
    state_reset(&t->state);
}

int azo_set_eps(azo_tree *t, double eps)
{
    if (eps > 1. || eps < 0.)
        return fail(AZO_ERR_EPS_RANGE, "Epsilon must be in range <0,1>", 0, 0, 0);
    t->dir_eps = eps;
    return 0;
}
double azo_get_eps(const azo_tree *t) { return t->dir_eps; }
void azo_set_cpuct(azo_tree *t, float cpuct) { t->cpuct = cpuct; }
float azo_get_cpuct(const azo_tree *t) { return (float)t->cpuct; }
void azo_set_threads(azo_tree *t, int threads) { t->threads = threads; }
void azo_set_schedule(azo_tree *t, int schedule) { t->schedule = schedule; }
int azo_get_schedule(const azo_tree *t) { return t->schedule; }
int azo_get_threads(const azo_tree *t) { return t->threads; }
void azo_set_alpha(azo_tree *t, double alpha)
{
    for (int i = 0; i < t->size; i++)
        t->alpha[i] = alpha;
}
double azo_get_alpha(const azo_tree *t) { return t->alpha[0]; }

void azo_set_alpha_default(azo_tree *t)
{
    double len = t->shape * 2; /* average game length estimate */
    double num = (t->size * len - (len * len + len) * 0.5) / len;
    for (int i = 0; i < t->size; i++)
        t->alpha[i] = 10 / num;
}

int azo_set_cuda(azo_tree *t, int cuda)
{
    /* `cuda != 1 || cuda != 0` holds for every int (cmcts.cpp:70) */
    if (cuda != 1 || cuda != 0)
        return fail(AZO_ERR_INVALID_VALUE, "Invalid value", 0, 0, 0);
    t->cuda = cuda;
    return 0;
}
void azo_set_player(azo_tree *t, int player) { t->state.player = player; }
int azo_get_player(const azo_tree *t) { return t->state.player; }
int azo_get_move_cnt(const azo_tree *t) { return t->state.move_cnt; }
float azo_get_winner(const azo_tree *t) { return t->state.winner; }
int azo_shape(const azo_tree *t) { return t->shape; }

/* one worker: cmcts.cpp:127-166 */
typedef struct {
    azo_tree *t;
    int n;
    const azo_evaluator *ev;
    const double *noise;
    int noise_rows;
    int rc;
    char err[256];
} az_worker;

static int evaluate(const azo_tree *t, const azo_evaluator *ev, const az_state *s,
                    float *value, float *prior)
{
    switch (ev->kind) {
    case AZO_EVAL_UNIFORM:
        for (int i = 0; i < t->size; i++)
            prior[i] = ev->const_prior;
        *value = ev->const_value;
        return 0;
    case AZO_EVAL_HASH:
        azo_hash_eval(t->shape, ev->hash_seed, s->board, value, prior);
        return 0;
    default:
        return ev->fn(ev->ctx, s->board, s->player, value, prior);
    }
}

/* one simulation: cmcts.cpp:168-282 */
static int search(az_worker *w, az_state *ss, az_node **nodes, int *actions,
                  float *prior, int *noise_row, long long *cnt)
{
    azo_tree *t = w->t;
    az_node *cur = t->root;
    int depth = 0;
    float value = 0.f;

    for (;;) {
        if (__atomic_load_n(&cur->nodeN, __ATOMIC_RELAXED) == -1) {
            /* expansion: cmcts.cpp:190-243.  The evaluator sees the raw,
             * un-swapped board (cmcts.cpp:206-210). */
            float v;
            int is_log = evaluate(t, w->ev, ss, &v, prior);
            value = -v; /* cmcts.cpp:228 */
            const double *row = NULL;
            if (w->noise) {
                if (*noise_row >= w->noise_rows) {
                    snprintf(w->err, sizeof w->err, "noise table exhausted (%d rows)", w->noise_rows);
                    return AZO_ERR_NOISE_ROWS;
                }
                row = w->noise + (size_t)(*noise_row) * t->size;
            } else if (t->dir_eps != 0.) {
                snprintf(w->err, sizeof w->err, "eps != 0 needs a noise table");
                return AZO_ERR_NOISE_ROWS;
            }
            if (node_set_prior(cur, t->size, prior, is_log, row, t->dir_eps)) {
                if (w->noise)
                    (*noise_row)++;
                cnt[1]++;
            }
            break;
        }
        int a = node_select(cur, ss, t->cpuct); /* cmcts.cpp:247 */
        cnt[3]++;
        if (a < 0) {
            snprintf(w->err, sizeof w->err, a == -1 ? "No action chosen. Incorrect behaviour"
                                                     : "Node has not been visited yet. Can't select next_node");
            return AZO_ERR_NO_ACTION;
        }
        state_make_move(ss, a);
        nodes[depth] = cur;
        actions[depth] = a;
        depth++;
        if (ss->winner != -1.f) { /* cmcts.cpp:253-267: no node for a terminal */
            if (ss->winner == 0.f || ss->winner == 1.f)
                value = 1.f;
            else
                value = 0.f;
            cnt[2]++;
            break;
        }
        cur = node_next(cur, a, t->size); /* cmcts.cpp:268 */
    }
    /* backup: cmcts.cpp:271-279 */
    while (depth > 0) {
        depth--;
        node_backpropagate(nodes[depth], actions[depth], value);
        value = -value;
    }
    cnt[0]++;
    return 0;
}

static void *worker_main(void *arg)
{
    az_worker *w = (az_worker *)arg;
    azo_tree *t = w->t;
    az_state ss;
    state_init(&ss, t->shape);
    state_copy(&ss, &t->state); /* cmcts.cpp:145 */
    az_node **nodes = (az_node **)malloc(sizeof(az_node *) * (t->size + 1));
    int *actions = (int *)malloc(sizeof(int) * (t->size + 1));
    float *prior = (float *)malloc(sizeof(float) * t->size);
    int noise_row = 0; /* fresh default-seeded generator per call (cmcts.cpp:132-133) */
    long long cnt[4] = {0, 0, 0, 0};
    w->rc = 0;
    for (int i = 0; i < w->n; i++) { /* cmcts.cpp:151-154 */
        int rc = search(w, &ss, nodes, actions, prior, &noise_row, cnt);
        if (rc) {
            w->rc = rc;
            break;
        }
        state_copy(&ss, &t->state);
    }
    pthread_mutex_lock(&t->stat_mu);
    t->sims += cnt[0];
    t->expansions += cnt[1];
    t->terminals += cnt[2];
    t->select_calls += cnt[3];
    pthread_mutex_unlock(&t->stat_mu);
    free(nodes);
    free(actions);
    free(prior);
    state_free(&ss);
    return NULL;
}

/* ------------------------------------------------------------------ */
/* Lock-step schedule of the thread fan-out (cmcts.cpp:98-119).          */
/* The reference's workers are std::threads whose interleaving the OS    */
/* decides; every interleaving in which each node operation is atomic    */
/* (they run under the node's mutex) is a legal execution.  This is ONE  */
/* of them, fixed: in step i the workers 0..T-1, in that order, run      */
/* their i-th search down to its leaf (select with virtual loss,         */
/* make_move, next_node: cmcts.cpp:189-268), then the leaves are         */
/* evaluated, then the workers 0..T-1, in that order, finish their       */
/* search (set_prior, which does nothing when an earlier worker expanded */
/* the same node, node.cpp:30-33; backup, cmcts.cpp:271-279).            */
/* ------------------------------------------------------------------ */
typedef struct {
    az_state ss;
    az_node **nodes;
    int *actions;
    int depth;
    az_node *leaf;   /* node to expand, NULL when the search ended on a finished game */
    float value;     /* value of a finished game */
    int noise_row;   /* the worker's own generator (cmcts.cpp:132-133) */
} az_pending;

static int lockstep_descend(azo_tree *t, az_pending *p, long long *cnt, char *err, size_t err_len)
{
    az_node *cur = t->root;
    state_copy(&p->ss, &t->state);
    p->depth = 0;
    p->leaf = NULL;
    p->value = 0.f;
    for (;;) {
        if (cur->nodeN == -1) { /* cmcts.cpp:190 */
            p->leaf = cur;
            return 0;
        }
        int a = node_select(cur, &p->ss, t->cpuct);
        cnt[3]++;
        if (a < 0) {
            snprintf(err, err_len, "No action chosen. Incorrect behaviour");
            return AZO_ERR_NO_ACTION;
        }
        state_make_move(&p->ss, a);
        p->nodes[p->depth] = cur;
        p->actions[p->depth] = a;
        p->depth++;
        if (p->ss.winner != -1.f) { /* cmcts.cpp:253-267 */
            p->value = (p->ss.winner == 0.f || p->ss.winner == 1.f) ? 1.f : 0.f;
            cnt[2]++;
            return 0;
        }
        cur = node_next(cur, a, t->size);
    }
}

static int lockstep_finish(azo_tree *t, az_pending *p, const azo_evaluator *ev, const double *noise, int noise_rows,
                           float *prior, long long *cnt, char *err, size_t err_len)
{
    float value = p->value;
    if (p->leaf) {
        float v;
        int is_log = evaluate(t, ev, &p->ss, &v, prior);
        value = -v; /* cmcts.cpp:228 */
        const double *row = NULL;
        if (noise) {
            /* (only a worker that really expands needs a row; checked like the sequential path) */
            if (p->leaf->nodeN == -1 && p->noise_row >= noise_rows) {
                snprintf(err, err_len, "noise table exhausted (%d rows)", noise_rows);
                return AZO_ERR_NOISE_ROWS;
            }
            row = noise + (size_t)p->noise_row * t->size;
        } else if (t->dir_eps != 0.) {
            snprintf(err, err_len, "eps != 0 needs a noise table");
            return AZO_ERR_NOISE_ROWS;
        }
        if (node_set_prior(p->leaf, t->size, prior, is_log, row, t->dir_eps)) {
            if (noise)
                p->noise_row++;
            cnt[1]++;
        }
    }
    for (int d = p->depth - 1; d >= 0; d--) {
        node_backpropagate(p->nodes[d], p->actions[d], value);
        value = -value;
    }
    cnt[0]++;
    return 0;
}

static int simulate_lockstep(azo_tree *t, int th_num, int per, const azo_evaluator *ev, const double *noise, int noise_rows)
{
    az_pending *ps = (az_pending *)calloc(th_num, sizeof(az_pending));
    float *prior = (float *)malloc(sizeof(float) * t->size);
    for (int w = 0; w < th_num; w++) {
        state_init(&ps[w].ss, t->shape);
        ps[w].nodes = (az_node **)malloc(sizeof(az_node *) * (t->size + 1));
        ps[w].actions = (int *)malloc(sizeof(int) * (t->size + 1));
    }
    long long cnt[4] = {0, 0, 0, 0};
    int rc = 0;
    for (int i = 0; i < per && !rc; i++) {
        for (int w = 0; w < th_num && !rc; w++)
            rc = lockstep_descend(t, &ps[w], cnt, g_err, sizeof g_err);
        for (int w = 0; w < th_num && !rc; w++)
            rc = lockstep_finish(t, &ps[w], ev, noise, noise_rows, prior, cnt, g_err, sizeof g_err);
    }
    t->sims += cnt[0];
    t->expansions += cnt[1];
    t->terminals += cnt[2];
    t->select_calls += cnt[3];
    for (int w = 0; w < th_num; w++) {
        state_free(&ps[w].ss);
        free(ps[w].nodes);
        free(ps[w].actions);
    }
    free(ps);
    free(prior);
    return rc;
}

/* cmcts.cpp:83-125 */
int azo_simulate(azo_tree *t, int n, const azo_evaluator *ev, const double *noise, int noise_rows)
{
    if (n < 1)
        return fail(AZO_ERR_BAD_SIMS, "Invalid input %ld must be at least 1!", n, 0, 0);
    if (!ev)
        return fail(AZO_ERR_NO_PREDICTOR, "Predictor missing!", 0, 0, 0);

    if (t->threads > 1) {
        int th_num = t->threads;
        if (n < th_num) {
            th_num = n;
            n = 1;
        } else if (n % th_num == 0)
            n = n / th_num;
        else
            n = n / th_num + 1; /* overshoots the request (cmcts.cpp:106-107) */
        if (t->schedule == AZO_SCHEDULE_LOCKSTEP)
            return simulate_lockstep(t, th_num, n, ev, noise, noise_rows);
        pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * th_num);
        az_worker *ws = (az_worker *)calloc(th_num, sizeof(az_worker));
        for (int i = 0; i < th_num; i++) {
            ws[i].t = t; ws[i].n = n; ws[i].ev = ev;
            ws[i].noise = noise; ws[i].noise_rows = noise_rows;
            pthread_create(&th[i], NULL, worker_main, &ws[i]);
        }
        int rc = 0;
        for (int i = 0; i < th_num; i++) {
            pthread_join(th[i], NULL);
            if (ws[i].rc && !rc) {
                rc = ws[i].rc;
                snprintf(g_err, sizeof g_err, "%s", ws[i].err);
            }
        }
        free(th);
        free(ws);
        return rc;
    }
    az_worker w;
    memset(&w, 0, sizeof w);
    w.t = t; w.n = n; w.ev = ev; w.noise = noise; w.noise_rows = noise_rows;
    worker_main(&w);
    if (w.rc)
        snprintf(g_err, sizeof g_err, "%s", w.err);
    return w.rc;
}

/* cmcts.cpp:284-294 with node.cpp:147-158 */
int azo_make_move(azo_tree *t, int action)
{
    if (state_make_move(&t->state, action))
        return fail(AZO_ERR_INVALID_MOVE, "Invalid move %ld", action, 0, 0);
    az_node *old = t->root;
    az_node *keep = old->child[action];
    old->child[action] = NULL; /* release() */
    if (keep == NULL)
        keep = node_new(t->size);
    t->root = keep;
    node_free(old, t->size);
    return 0;
}

/* cmcts.cpp:296-309 */
int azo_make_movexy(azo_tree *t, int x, int y)
{
    if (y >= t->shape || x >= t->shape)
        return fail(AZO_ERR_BAD_XY, "Invalid index y:%ld or x: %ld (range is 0-%ld)!", y, x, t->size);
    return azo_make_move(t, y * t->shape + x);
}

/* cmcts.cpp:317-341 */
int azo_get_prob(const azo_tree *t, float *out)
{
    double sum = (double)t->root->nodeN;
    double prob_sum = 0;
    for (int i = 0; i < t->size; i++) {
        out[i] = (float)((double)t->root->childN[i] / sum);
        prob_sum += out[i];
    }
    if (round(prob_sum * 1000.) != 1000)
        return fail(AZO_ERR_NOT_DIST, "Not a distribution", 0, 0, 0);
    return 0;
}

/* state.cpp:39-48: planes swapped when player == 1 */
void azo_get_board(const azo_tree *t, float *out)
{
    const int size = t->size;
    const az_state *s = &t->state;
    for (int i = 0; i < 2 * size; i++)
        out[i] = (float)s->board[i];
    if (s->player == 1)
        for (int i = 0; i < size; i++) {
            float tmp = out[i];
            out[i] = out[size + i];
            out[size + i] = tmp;
        }
}

long long azo_root_n(const azo_tree *t) { return t->root->nodeN; }
void azo_root_counts(const azo_tree *t, long *out) { memcpy(out, t->root->childN, sizeof(long) * t->size); }
void azo_root_w(const azo_tree *t, double *out) { memcpy(out, t->root->childW, sizeof(double) * t->size); }
void azo_root_p(const azo_tree *t, double *out) { memcpy(out, t->root->childP, sizeof(double) * t->size); }
void azo_raw_board(const azo_tree *t, signed char *out) { memcpy(out, t->state.board, (size_t)2 * t->size); }

int azo_node_at(const azo_tree *t, const int *path, int len, long long *node_n,
                long *counts, double *w, double *p)
{
    const az_node *n = t->root;
    for (int i = 0; i < len; i++) {
        if (path[i] < 0 || path[i] >= t->size || n->child[path[i]] == NULL)
            return 1;
        n = n->child[path[i]];
    }
    *node_n = n->nodeN;
    if (counts) memcpy(counts, n->childN, sizeof(long) * t->size);
    if (w) memcpy(w, n->childW, sizeof(double) * t->size);
    if (p) memcpy(p, n->childP, sizeof(double) * t->size);
    return 0;
}

void azo_stats(const azo_tree *t, long long *sims, long long *expansions,
               long long *terminals, long long *select_calls)
{
    *sims = t->sims;
    *expansions = t->expansions;
    *terminals = t->terminals;
    *select_calls = t->select_calls;
}

long long azo_live_nodes(const azo_tree *t)
{
    (void)t;
    return __atomic_load_n(&g_live_nodes, __ATOMIC_RELAXED);
}

/* ------------------------------------------------------------------ */
/* stand-alone pieces for unit parity tests                             */
/* ------------------------------------------------------------------ */
int azo_rules_play(int shape, const int *moves, int n, int *player, int *move_cnt, float *winner)
{
    az_state s;
    state_init(&s, shape);
    int i;
    for (i = 0; i < n; i++) {
        if (state_make_move(&s, moves[i]))
            break;
        player[i] = s.player;
        move_cnt[i] = s.move_cnt;
        winner[i] = s.winner;
    }
    state_free(&s);
    return i;
}

/* same with the side to move forced before a ply (force[i] >= 0), as
 * Cmcts::set_player allows (cmcts.cpp:48) */
int azo_rules_play_forced(int shape, const int *moves, const int *force, int n, int *player,
                          int *move_cnt, float *winner)
{
    az_state s;
    state_init(&s, shape);
    int i;
    for (i = 0; i < n; i++) {
        if (force[i] >= 0)
            s.player = force[i];
        if (state_make_move(&s, moves[i]))
            break;
        player[i] = s.player;
        move_cnt[i] = s.move_cnt;
        winner[i] = s.winner;
    }
    state_free(&s);
    return i;
}

int azo_select_arrays(int size, const long *n, const double *w, const double *p,
                      long long node_n, const signed char *board, double cpuct)
{
    int shape = (int)lround(sqrt((double)size));
    az_state s;
    state_init(&s, shape);
    memcpy(s.board, board, (size_t)2 * size);
    az_node *nd = node_new(size);
    memcpy(nd->childN, n, sizeof(long) * size);
    memcpy(nd->childW, w, sizeof(double) * size);
    memcpy(nd->childP, p, sizeof(double) * size);
    nd->nodeN = node_n;
    int a = node_select(nd, &s, cpuct);
    node_free(nd, size);
    state_free(&s);
    return a;
}

/* ------------------------------------------------------------------ */
/* CPU baseline legs (bench.py): the same game loop the GPU bench steps  */
/* through -- two-ply opening, then per ply simulate(sims) + greedy move,*/
/* finished games restarted from their opening.  The first `skip` plies  */
/* of every game are played but not timed, so the timed plies are the    */
/* ones the GPU leg times after its warm-up steps.                       */
/* ------------------------------------------------------------------ */
typedef struct {
    int shape, games, skip, plies, sims, tree_threads;
    double cpuct;
    const azo_evaluator *ev;
    int next_game;        /* atomic */
    long long total_sims; /* atomic: searches inside timed plies (as counted by the trees) */
    double rate_sum;      /* sum over workers of timed searches / timed seconds (under mu) */
    double secs_max;      /* longest timed span of a worker (under mu) */
    pthread_mutex_t mu;
} az_bl_job;

void azo_opening(int size, int g, int *a0, int *a1)
{
    *a0 = g % size;
    *a1 = (*a0 + 1 + (g / size) % (size - 1)) % size;
}

static void bl_open(azo_tree *t, int g)
{
    int a0, a1;
    azo_opening(t->size, g, &a0, &a1);
    azo_make_move(t, a0);
    azo_make_move(t, a1);
}

static double bl_now(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static void *bl_main(void *arg)
{
    az_bl_job *job = (az_bl_job *)arg;
    double secs = 0.;
    long long sims = 0;
    for (;;) {
        int g = __atomic_fetch_add(&job->next_game, 1, __ATOMIC_RELAXED);
        if (g >= job->games)
            break;
        azo_tree *t = azo_create(job->shape, 1.0, job->cpuct);
        azo_set_alpha_default(t);
        azo_set_eps(t, 0.0);
        azo_set_threads(t, job->tree_threads);
        bl_open(t, g);
        long *counts = (long *)malloc(sizeof(long) * t->size);
        for (int ply = 0; ply < job->skip + job->plies; ply++) {
            const int timed = ply >= job->skip;
            const long long before = t->sims;
            const double t0 = timed ? bl_now() : 0.;
            if (azo_simulate(t, job->sims, job->ev, NULL, 0))
                break;
            azo_root_counts(t, counts);
            int best = 0;
            for (int i = 1; i < t->size; i++)
                if (counts[i] > counts[best])
                    best = i;
            azo_make_move(t, best);
            if (t->state.winner != -1.f) { /* finished: restart from the opening */
                azo_clear(t);
                bl_open(t, g);
            }
            if (timed) {
                secs += bl_now() - t0;
                sims += t->sims - before; /* the fan-out may overshoot the request (cmcts.cpp:106-107) */
            }
        }
        free(counts);
        azo_destroy(t);
    }
    __atomic_add_fetch(&job->total_sims, sims, __ATOMIC_RELAXED);
    pthread_mutex_lock(&job->mu);
    if (secs > 0.)
        job->rate_sum += (double)sims / secs;
    if (secs > job->secs_max)
        job->secs_max = secs;
    pthread_mutex_unlock(&job->mu);
    return NULL;
}

/* `workers` host threads play `games` games between them, each game on a tree of
 * its own searched with `tree_threads` threads (0/1 = the sequential worker of
 * cmcts.cpp:120-122; > 1 = the shared-tree fan-out of cmcts.cpp:98-119).
 * Returns the searches done inside timed plies; *rate = sum over the workers of
 * their own searches per second of timed plies (the workers run side by side for
 * the whole call), *secs = the longest worker's timed span. */
long long azo_baseline_run(int shape, int games, int workers, int tree_threads, int skip, int plies,
                           int sims, double cpuct, const azo_evaluator *ev, double *rate, double *secs)
{
    az_bl_job job;
    memset(&job, 0, sizeof job);
    job.shape = shape; job.games = games; job.skip = skip; job.plies = plies; job.sims = sims;
    job.tree_threads = tree_threads; job.cpuct = cpuct; job.ev = ev;
    pthread_mutex_init(&job.mu, NULL);
    if (workers < 1)
        workers = 1;
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * workers);
    int started = 0;
    for (int i = 0; i < workers; i++)
        if (pthread_create(&th[i], NULL, bl_main, &job) == 0)
            th[started++] = th[i];
    for (int i = 0; i < started; i++)
        pthread_join(th[i], NULL);
    free(th);
    pthread_mutex_destroy(&job.mu);
    if (rate) *rate = job.rate_sum;
    if (secs) *secs = job.secs_max;
    return job.total_sims;
}

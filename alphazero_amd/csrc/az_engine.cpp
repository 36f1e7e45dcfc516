// az_engine.cpp -- host side of libazmcts.so: node-pool allocation, shape
// dispatch and the C ABI declared in include/azmcts.h.  No torch types, no
// CPU search path: every search entry point launches gfx950 kernels.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/azmcts.h"
#include "az_common.h"

#define AZ_DECL(s) extern "C" const AzShapeOps *az_shape_ops_##s(void);
AZ_SHAPE_LIST(AZ_DECL)
#undef AZ_DECL

namespace {

thread_local char g_err[512];

int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return fail(AZ_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_));            \
    } while (0)

const AzShapeOps *find_ops(int shape)
{
#define AZ_PICK(s) \
    if (shape == s) return az_shape_ops_##s();
    AZ_SHAPE_LIST(AZ_PICK)
#undef AZ_PICK
    return nullptr;
}

} // namespace

// Step staging of az_simulate_net_noalloc: one fixed buffer per GPU, part of the
// library's device image (every GPU the library runs on holds it until the
// process ends).  Host code never uses the array's name as an address: the
// per-device address is looked up once per device (resolve_scratch, from
// az_create) and passed to the kernels like any other buffer.
#define AZ_NOALLOC_SCRATCH_BYTES (5u << 20)
#define AZ_MAX_DEVICES 64
static __device__ __align__(256) unsigned char scratch[AZ_NOALLOC_SCRATCH_BYTES];
static char *g_scratch_ptr[AZ_MAX_DEVICES] = {};
static std::mutex g_scratch_mu;

struct az_engine {
    int device = 0;
    int n_trees = 0, shape = 0, size = 0, epl = 0, sp = 0, cap = 0;
    size_t rec_bytes = 0;
    const AzShapeOps *ops = nullptr;
    AzParams P{};
    double alpha = 1.0;
    int threads = 0;
    int schedule = AZ_WORKERS_CONCURRENT; // how az_simulate_net runs the workers of threads > 1
    int net_path = AZ_NET_PATH_NONE;      // the launch sequence of the last network search
    int cuda = 0;
    unsigned long long device_bytes = 0;
    // device scratch
    int *d_actions = nullptr;      // [n_trees]
    unsigned char *d_mask = nullptr; // [n_trees]
    int *d_counts = nullptr;       // [n_trees][size]
    int *d_root_n = nullptr;       // [n_trees]
    double *d_noise = nullptr;     // [rows][sp]
    int noise_cap = 0;             // rows allocated at d_noise
    char *d_stage = nullptr;       // az_simulate_net: leaf boards, codes, values, log-priors of one step
    size_t stage_bytes = 0;
    double *d_u = nullptr;         // [n_trees] az_slots_choose: the uniform number of each slot's move
    int *d_emit = nullptr;         // [3][n_trees] az_slots_finish: first record, record count and game of the slots that finish
    std::vector<AzTree> h_trees;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // streams of the tree groups of az_simulate_net beyond the first (which runs on the caller's stream), and the
    // events that tie them to it
    hipStream_t aux_streams[3] = {nullptr, nullptr, nullptr}; // (the device's shared ones, see device_aux_streams)
    int n_aux_streams = 0;
    hipEvent_t ev_fork = nullptr, ev_join[3] = {nullptr, nullptr, nullptr};
    std::vector<hipEvent_t> ev_slots; // [2 * slot + end] for az_timing_mark, created on first use
    // one event per stream this engine has queued work on, re-recorded behind every asynchronous call: what the
    // synchronous entry points wait for (instead of the whole device: other engines, torch and RCCL keep running)
    struct StreamMark { hipStream_t stream; hipEvent_t done; };
    std::vector<StreamMark> marks;
};

namespace {

struct DeviceGuard {
    int prev = -1;
    bool ok = true;
    explicit DeviceGuard(int dev)
    {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) ok = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceGuard()
    {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

template <typename T> int dev_alloc(az_engine *e, T **p, size_t count)
{
    size_t bytes = count * sizeof(T);
    if (bytes == 0) bytes = sizeof(T);
    HIP_TRY(hipMalloc((void **)p, bytes));
    e->device_bytes += bytes;
    return AZ_OK;
}

int fetch_trees(az_engine *e)
{
    HIP_TRY(hipMemcpy(e->h_trees.data(), e->P.trees, sizeof(AzTree) * e->n_trees, hipMemcpyDeviceToHost));
    return AZ_OK;
}

// address of `scratch` on `device`, which must be the current device
int resolve_scratch(int device)
{
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    if (g_scratch_ptr[device]) return AZ_OK;
    void *ptr = nullptr;
    HIP_TRY(hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)));
    if (!ptr) return fail(AZ_ERR_HIP, "hipGetSymbolAddress returned a null address");
    g_scratch_ptr[device] = static_cast<char *>(ptr);
    return AZ_OK;
}

// The streams the tree groups of az_simulate_net run on beside the caller's: ONE set per device for all engines (a
// process has few hardware queues -- four by default -- and streams beyond them share a queue, which would put a
// group's work behind the very stream it is meant to overlap with), created with the device's first engine and kept
// until the process ends.  As many as the AZ_NET_GROUPS override asks for, one by default.  Returns their number.
int device_aux_streams(int device, hipStream_t (&out)[3])
{
    static hipStream_t pool[AZ_MAX_DEVICES][3] = {};
    static int count[AZ_MAX_DEVICES] = {};
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    int want = 1;
    if (const char *w = getenv("AZ_NET_GROUPS")) want = atoi(w) - 1;
    want = want < 0 ? 0 : (want > 3 ? 3 : want);
    while (count[device] < want) {
        if (hipStreamCreateWithFlags(&pool[device][count[device]], hipStreamNonBlocking) != hipSuccess) {
            (void)hipGetLastError();
            break;
        }
        count[device]++;
    }
    for (int k = 0; k < 3; k++) out[k] = k < count[device] ? pool[device][k] : nullptr;
    return count[device];
}

int check_tree(const az_engine *e, int tree)
{
    if (!e) return fail(AZ_ERR_ARG, "null engine");
    if (tree < 0 || tree >= e->n_trees) return fail(AZ_ERR_ARG, "tree index %d out of range (0-%d)", tree, e->n_trees - 1);
    return AZ_OK;
}

// Called behind every asynchronous entry point: remembers that `stream` carries work of this engine.  A stream under
// graph capture is left alone (an event recorded there belongs to the graph and cannot be waited for).
void note_queued(az_engine *e, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cap) != hipSuccess) { (void)hipGetLastError(); return; }
    if (cap != hipStreamCaptureStatusNone) return;
    for (auto &m : e->marks)
        if (m.stream == st) { (void)hipEventRecord(m.done, st); return; }
    hipEvent_t ev = nullptr;
    if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); return; }
    (void)hipEventRecord(ev, st);
    e->marks.push_back({st, ev});
}

// waits for everything this engine has queued, on whatever stream
int wait_queued(az_engine *e)
{
    for (auto &m : e->marks) HIP_TRY(hipEventSynchronize(m.done));
    return AZ_OK;
}

const char *kernel_err_text(int code)
{
    switch (code) {
    case AZ_ERR_POOL_FULL: return "node pool exhausted";
    case AZ_ERR_NO_ACTION: return "No action chosen. Incorrect behaviour";
    case AZ_ERR_NOISE_ROWS: return "noise table exhausted";
    case AZ_ERR_INVALID_MOVE: return "Invalid move";
    case AZ_ERR_NOT_DIST: return "Not a distribution";
    case AZ_ERR_STATE: return "the tree's state does not allow this call (a record row outside the staging rows, az_simulate_fused "
                              "on a tree whose nodes carry visit deficits from concurrent workers, or a one-launch search with "
                              "concurrent workers that was abandoned because a workgroup of its cluster never answered)";
    default: return "kernel error";
    }
}

} // namespace

extern "C" {

const char *az_last_error(void) { return g_err; }

#ifndef AZ_GIT
#define AZ_GIT "unknown"
#endif
const char *az_version(void) { return AZ_GIT; }
const char *az_build_timestamp(void) { return __DATE__ " " __TIME__; }

int az_create(az_engine **out, int device, int n_trees, int shape, int nodes_per_tree, double alpha, double cpuct)
{
    if (!out) return fail(AZ_ERR_ARG, "null output pointer");
    *out = nullptr;
    const AzShapeOps *ops = find_ops(shape);
    if (!ops) return fail(AZ_ERR_ARG, "board shape %d is not compiled into this library", shape);
    if (n_trees < 1) return fail(AZ_ERR_ARG, "n_trees must be at least 1");
    if (nodes_per_tree < 2 || nodes_per_tree > (1 << 19))
        return fail(AZ_ERR_ARG, "nodes_per_tree must be in 2..524288");
    // the kernels address a record as <tree arena> + 32-bit byte offset
    if ((size_t)nodes_per_tree * ops->sp * 24 > 0xFFFFFFFFull)
        return fail(AZ_ERR_ARG, "nodes_per_tree * record size exceeds 4 GiB per tree (at most %zu slots for this shape)",
                    (size_t)(0xFFFFFFFFull / ((size_t)ops->sp * 24)));
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev || device >= AZ_MAX_DEVICES)
        return fail(AZ_ERR_ARG, "device %d not present (%d visible)", device, ndev);

    az_engine *e = new az_engine();
    e->device = device;
    DeviceGuard g(device);
    if (!g.ok) { delete e; return fail(AZ_ERR_HIP, "hipSetDevice(%d) failed", device); }
    if (int rc = resolve_scratch(device)) { delete e; return rc; }
    e->n_trees = n_trees;
    e->shape = shape;
    e->size = shape * shape;
    e->sp = ops->sp;
    e->epl = e->sp / 64;
    e->cap = nodes_per_tree;
    e->rec_bytes = (size_t)e->sp * 24;
    e->ops = ops;
    e->alpha = alpha;
    AzParams &P = e->P;
    P.n_trees = n_trees;
    P.cap = nodes_per_tree;
    P.path_cap = e->size + 1;
    P.cpuct = cpuct;
    P.eps = 0.25; // cmcts.cpp:8
    P.noise = nullptr;
    P.noise_rows = 0;
    P.stamps = nullptr;
    const size_t slots = (size_t)n_trees * nodes_per_tree;
    int rc = AZ_OK;
    if (!rc) rc = dev_alloc(e, &P.nodes, slots * e->rec_bytes);
    if (!rc) rc = dev_alloc(e, &P.parent, slots);
    if (!rc) rc = dev_alloc(e, &P.free_stack, slots);
    if (!rc) rc = dev_alloc(e, &P.order, slots);
    if (!rc) rc = dev_alloc(e, &P.trees, (size_t)n_trees);
    if (!rc) rc = dev_alloc(e, &P.rows, (size_t)n_trees * 64);
    if (!rc) rc = dev_alloc(e, &P.path_node, (size_t)n_trees * P.path_cap);
    if (!rc) rc = dev_alloc(e, &P.path_act, (size_t)n_trees * P.path_cap);
    if (!rc) rc = dev_alloc(e, &P.aux, slots);
    P.workers = nullptr;
    P.wpath_node = P.wpath_act = nullptr;
    P.max_workers = 0;
    if (!rc) rc = dev_alloc(e, &e->d_actions, (size_t)n_trees);
    if (!rc) rc = dev_alloc(e, &e->d_mask, (size_t)n_trees);
    if (!rc) rc = dev_alloc(e, &e->d_counts, (size_t)n_trees * e->size);
    if (!rc) rc = dev_alloc(e, &e->d_root_n, (size_t)n_trees);
    if (!rc) rc = dev_alloc(e, &e->d_u, (size_t)n_trees);
    if (!rc) rc = dev_alloc(e, &e->d_emit, (size_t)3 * n_trees);
#ifdef AZ_STAMPS
    if (!rc) rc = dev_alloc(e, &P.stamps, (size_t)n_trees * 8);
#endif
    if (rc) { az_destroy(e); return rc; }

    // initial pool state: every slot free, stack pops 0,1,2,...
    {
        std::vector<int> stack(slots), par(slots, -2);
        for (int t = 0; t < n_trees; t++)
            for (int k = 0; k < nodes_per_tree; k++) stack[(size_t)t * nodes_per_tree + k] = nodes_per_tree - 1 - k;
        hipError_t he = hipMemcpy(P.free_stack, stack.data(), slots * sizeof(int), hipMemcpyHostToDevice);
        if (he == hipSuccess) he = hipMemcpy(P.parent, par.data(), slots * sizeof(int), hipMemcpyHostToDevice);
        if (he == hipSuccess) he = hipMemset(P.rows, 0, (size_t)n_trees * 64 * sizeof(unsigned));
        if (he == hipSuccess) he = hipMemset(P.aux, 0, slots * sizeof(int));
        e->h_trees.assign(n_trees, AzTree{});
        for (auto &t : e->h_trees) {
            t.root = -1;
            t.root_n = -1;
            t.free_top = nodes_per_tree;
            t.winner = -1.f;
            t.active = 1;
        }
        if (he == hipSuccess)
            he = hipMemcpy(P.trees, e->h_trees.data(), sizeof(AzTree) * n_trees, hipMemcpyHostToDevice);
        if (he == hipSuccess) he = hipEventCreate(&e->ev0);
        if (he == hipSuccess) he = hipEventCreate(&e->ev1);
        if (he == hipSuccess) he = hipEventCreateWithFlags(&e->ev_fork, hipEventDisableTiming);
        for (int k = 0; k < 3 && he == hipSuccess; k++) he = hipEventCreateWithFlags(&e->ev_join[k], hipEventDisableTiming);
        if (he == hipSuccess) e->n_aux_streams = device_aux_streams(device, e->aux_streams);
        if (he != hipSuccess) {
            az_destroy(e);
            return fail(AZ_ERR_HIP, "engine initialisation failed: %s", hipGetErrorString(he));
        }
    }
    *out = e;
    return AZ_OK;
}

void az_destroy(az_engine *e)
{
    if (!e) return;
    DeviceGuard g(e->device);
    (void)wait_queued(e);
    (void)hipStreamSynchronize(nullptr);
    AzParams &P = e->P;
    void *ptrs[] = {P.nodes, P.parent, P.free_stack, P.order, P.trees, P.rows,
                    P.path_node, P.path_act, P.aux, P.workers, P.wpath_node, P.wpath_act, P.stamps, e->d_actions, e->d_mask, e->d_counts, e->d_root_n, e->d_noise, e->d_stage,
                    e->d_u, e->d_emit};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    if (e->ev0) (void)hipEventDestroy(e->ev0);
    if (e->ev1) (void)hipEventDestroy(e->ev1);
    if (e->ev_fork) (void)hipEventDestroy(e->ev_fork);
    for (int k = 0; k < 3; k++)
        if (e->ev_join[k]) (void)hipEventDestroy(e->ev_join[k]);
    for (hipEvent_t ev : e->ev_slots)
        if (ev) (void)hipEventDestroy(ev);
    for (auto &m : e->marks) (void)hipEventDestroy(m.done);
    delete e;
}

int az_n_trees(const az_engine *e) { return e ? e->n_trees : 0; }
int az_shape(const az_engine *e) { return e ? e->shape : 0; }
int az_nodes_per_tree(const az_engine *e) { return e ? e->cap : 0; }
unsigned long long az_device_bytes(const az_engine *e) { return e ? e->device_bytes : 0; }

int az_set_eps(az_engine *e, double eps)
{
    if (!e) return fail(AZ_ERR_ARG, "null engine");
    if (eps > 1. || eps < 0.) return fail(AZ_ERR_EPS_RANGE, "Epsilon must be in range <0,1>");
    e->P.eps = eps;
    return AZ_OK;
}
double az_get_eps(const az_engine *e) { return e->P.eps; }
int az_set_cpuct(az_engine *e, float cpuct)
{
    if (!e) return fail(AZ_ERR_ARG, "null engine");
    e->P.cpuct = cpuct; // float argument widened into the double member (cmcts.cpp:58, cmcts.h:76)
    return AZ_OK;
}
float az_get_cpuct(const az_engine *e) { return (float)e->P.cpuct; }
int az_set_alpha(az_engine *e, double alpha)
{
    if (!e) return fail(AZ_ERR_ARG, "null engine");
    e->alpha = alpha;
    return AZ_OK;
}
double az_get_alpha(const az_engine *e) { return e->alpha; }
int az_set_alpha_default(az_engine *e)
{
    if (!e) return fail(AZ_ERR_ARG, "null engine");
    double len = e->shape * 2; // average game length estimate (cmcts.cpp:78)
    double num = (e->size * len - (len * len + len) * 0.5) / len;
    e->alpha = 10 / num;
    return AZ_OK;
}
// per-worker state for `workers` concurrent workers per tree (grows only; the search entry points never allocate it)
static int ensure_workers(az_engine *e, int workers)
{
    if (workers <= e->P.max_workers) return AZ_OK;
    if (workers > AZ_MAX_WORKERS) return fail(AZ_ERR_ARG, "at most %d concurrent workers per tree", AZ_MAX_WORKERS);
    DeviceGuard g(e->device);
    if (int rc = wait_queued(e)) return rc;
    void *old[] = {e->P.workers, e->P.wpath_node, e->P.wpath_act};
    for (void *p : old)
        if (p) (void)hipFree(p);
    e->P.workers = nullptr;
    e->P.wpath_node = e->P.wpath_act = nullptr;
    e->P.max_workers = 0;
    const size_t n = (size_t)e->n_trees * workers;
    if (int rc = dev_alloc(e, &e->P.workers, n)) return rc;
    if (int rc = dev_alloc(e, &e->P.wpath_node, n * e->P.path_cap)) return rc;
    if (int rc = dev_alloc(e, &e->P.wpath_act, n * e->P.path_cap)) return rc;
    HIP_TRY(hipMemset(e->P.workers, 0, n * sizeof(AzWorker)));
    e->P.max_workers = workers;
    return AZ_OK;
}

int az_set_threads(az_engine *e, int threads)
{
    if (!e) return fail(AZ_ERR_ARG, "null engine");
    if (threads > 1 && e->schedule == AZ_WORKERS_CONCURRENT)
        if (int rc = ensure_workers(e, threads)) return rc;
    e->threads = threads;
    return AZ_OK;
}
int az_set_schedule(az_engine *e, int schedule)
{
    if (!e) return fail(AZ_ERR_ARG, "null engine");
    if (schedule != AZ_WORKERS_SEQUENTIAL && schedule != AZ_WORKERS_CONCURRENT) return fail(AZ_ERR_ARG, "unknown schedule %d", schedule);
    if (schedule == AZ_WORKERS_CONCURRENT && e->threads > 1)
        if (int rc = ensure_workers(e, e->threads)) return rc;
    e->schedule = schedule;
    return AZ_OK;
}
int az_get_schedule(const az_engine *e) { return e ? e->schedule : 0; }
int az_get_net_path(const az_engine *e) { return e ? e->net_path : AZ_NET_PATH_NONE; }
int az_get_threads(const az_engine *e) { return e->threads; }
int az_set_cuda(az_engine *e, int cuda)
{
    if (!e) return fail(AZ_ERR_ARG, "null engine");
    // the reference's guard `cuda != 1 || cuda != 0` holds for every int, so its
    // setter always throws (cmcts.cpp:70); kept: the evaluator device is not chosen here
    (void)cuda;
    return fail(AZ_ERR_INVALID_VALUE, "Invalid value");
}
int az_get_cuda(const az_engine *e) { return e->cuda; }

int az_set_player(az_engine *e, int tree, int player)
{
    if (int rc = check_tree(e, tree)) return rc;
    if (player != 0 && player != 1) return fail(AZ_ERR_ARG, "player must be 0 or 1");
    DeviceGuard g(e->device);
    if (int rc_ = wait_queued(e)) return rc_;
    HIP_TRY(hipMemcpy((char *)(e->P.trees + tree) + offsetof(AzTree, player), &player, sizeof(int),
                      hipMemcpyHostToDevice));
    return AZ_OK;
}

int az_get_states(az_engine *e, int *player, int *move_cnt, float *winner)
{
    if (!e) return fail(AZ_ERR_ARG, "null engine");
    DeviceGuard g(e->device);
    if (int rc_ = wait_queued(e)) return rc_; // work queued on any stream lands first
    if (int rc = fetch_trees(e)) return rc;
    for (int t = 0; t < e->n_trees; t++) {
        if (player) player[t] = e->h_trees[t].player;
        if (move_cnt) move_cnt[t] = e->h_trees[t].move_cnt;
        if (winner) winner[t] = e->h_trees[t].winner;
    }
    return AZ_OK;
}

int az_get_state(az_engine *e, int tree, int *player, int *move_cnt, float *winner)
{
    if (int rc = check_tree(e, tree)) return rc;
    DeviceGuard g(e->device);
    if (int rc_ = wait_queued(e)) return rc_;
    AzTree t;
    HIP_TRY(hipMemcpy(&t, e->P.trees + tree, sizeof t, hipMemcpyDeviceToHost));
    if (player) *player = t.player;
    if (move_cnt) *move_cnt = t.move_cnt;
    if (winner) *winner = t.winner;
    return AZ_OK;
}

int az_clear_masked(az_engine *e, const unsigned char *mask)
{
    if (!e) return fail(AZ_ERR_ARG, "null engine");
    DeviceGuard g(e->device);
    if (int rc_ = wait_queued(e)) return rc_; // searches queued on any stream finish before their trees are reset
    if (mask) {
        HIP_TRY(hipMemcpy(e->d_mask, mask, e->n_trees, hipMemcpyHostToDevice));
        if (e->ops->clear(e->P, e->d_mask, nullptr)) return fail(AZ_ERR_HIP, "clear launch failed");
    } else {
        if (e->ops->clear(e->P, nullptr, nullptr)) return fail(AZ_ERR_HIP, "clear launch failed");
    }
    HIP_TRY(hipStreamSynchronize(nullptr));
    return AZ_OK;
}

int az_make_move_greedy(az_engine *e, void *stream)
{
    if (!e) return fail(AZ_ERR_ARG, "null engine");
    DeviceGuard g(e->device);
    if (e->ops->make_move(e->P, nullptr, stream)) return fail(AZ_ERR_HIP, "make_move launch failed");
    note_queued(e, stream);
    return AZ_OK;
}

int az_restart_finished(az_engine *e, const int *d_openings, int n_plies, int *d_count, void *stream)
{
    if (!e || n_plies < 0 || (n_plies > 0 && !d_openings)) return fail(AZ_ERR_ARG, "bad argument");
    DeviceGuard g(e->device);
    if (e->ops->restart(e->P, d_openings, n_plies, d_count, stream)) return fail(AZ_ERR_HIP, "restart launch failed");
    note_queued(e, stream);
    return AZ_OK;
}

int az_clear(az_engine *e, int tree)
{
    if (!e) return fail(AZ_ERR_ARG, "null engine");
    if (tree < 0) return az_clear_masked(e, nullptr);
    if (int rc = check_tree(e, tree)) return rc;
    std::vector<unsigned char> m(e->n_trees, 0);
    m[tree] = 1;
    return az_clear_masked(e, m.data());
}

int az_set_active(az_engine *e, const unsigned char *mask)
{
    if (!e) return fail(AZ_ERR_ARG, "null engine");
    DeviceGuard g(e->device);
    if (int rc_ = wait_queued(e)) return rc_;
    if (int rc = fetch_trees(e)) return rc;
    for (int t = 0; t < e->n_trees; t++) e->h_trees[t].active = mask ? (mask[t] != 0) : 1;
    HIP_TRY(hipMemcpy(e->P.trees, e->h_trees.data(), sizeof(AzTree) * e->n_trees, hipMemcpyHostToDevice));
    return AZ_OK;
}

int az_set_noise_table(az_engine *e, const double *noise, int rows)
{
    if (!e) return fail(AZ_ERR_ARG, "null engine");
    DeviceGuard g(e->device);
    if (int rc_ = wait_queued(e)) return rc_;
    if (e->d_noise) {
        (void)hipFree(e->d_noise);
        e->d_noise = nullptr;
    }
    e->noise_cap = 0;
    e->P.noise = nullptr;
    e->P.noise_rows = 0;
    if (rows <= 0 || !noise) return AZ_OK;
    std::vector<double> padded((size_t)rows * e->sp, 0.0);
    for (int r = 0; r < rows; r++)
        memcpy(&padded[(size_t)r * e->sp], noise + (size_t)r * e->size, sizeof(double) * e->size);
    HIP_TRY(hipMalloc((void **)&e->d_noise, padded.size() * sizeof(double)));
    HIP_TRY(hipMemcpy(e->d_noise, padded.data(), padded.size() * sizeof(double), hipMemcpyHostToDevice));
    e->noise_cap = rows;
    e->P.noise = e->d_noise;
    e->P.noise_rows = rows;
    return AZ_OK;
}

static int check_noise(const az_engine *e)
{
    if (e->P.eps != 0. && e->P.noise == nullptr)
        return fail(AZ_ERR_NOISE_ROWS, "eps != 0 needs a noise table (az_set_noise_table)");
    return AZ_OK;
}

int az_simulate_fused(az_engine *e, int n, int eval_kind, float const_prior, float const_value,
                      unsigned long long hash_seed, void *stream)
{
    if (!e) return fail(AZ_ERR_ARG, "null engine");
    if (n < 1) return fail(AZ_ERR_BAD_SIMS, "Invalid input %d must be at least 1!", n);
    if (eval_kind != AZ_EVAL_UNIFORM && eval_kind != AZ_EVAL_HASH) return fail(AZ_ERR_NO_PREDICTOR, "Predictor missing!");
    if (int rc = check_noise(e)) return rc;
    AzFusedArgs A{};
    // thread fan-out of cmcts.cpp:98-119: th_num workers of `per` searches each
    int th_num = 1, per = n;
    if (e->threads > 1) {
        th_num = e->threads;
        if (n < th_num) { th_num = n; per = 1; }
        else if (n % th_num == 0) per = n / th_num;
        else per = n / th_num + 1;
    }
    if (const char *w = getenv("AZ_WAVES_PER_TREE")) A.waves_per_tree = atoi(w); // tuning override
    A.lean = -1;
    if (const char *w = getenv("AZ_FUSED_LEAN")) A.lean = atoi(w) < 0 ? -1 : (atoi(w) > 2 ? 2 : atoi(w)); // tuning / test override of the kernel form
    A.n_total = th_num * per;
    A.per_worker = per;
    A.eval_kind = eval_kind;
    A.const_prior = const_prior;
    A.const_value = const_value;
    A.hash_seed = hash_seed;
    DeviceGuard g(e->device);
    if (e->ops->simulate_fused(e->P, A, stream)) return fail(AZ_ERR_HIP, "simulate launch failed");
    note_queued(e, stream);
    return AZ_OK;
}

int az_begin_simulate(az_engine *e, void *stream)
{
    if (!e) return fail(AZ_ERR_ARG, "null engine");
    if (int rc = check_noise(e)) return rc;
    DeviceGuard g(e->device);
    // noise_row := 0 for every tree: a strided 4-byte memset
    HIP_TRY(hipMemset2DAsync((char *)e->P.trees + offsetof(AzTree, noise_row), sizeof(AzTree), 0, sizeof(int),
                             e->n_trees, (hipStream_t)stream));
    if (e->P.workers) // every concurrent worker builds a generator of its own (cmcts.cpp:132-133)
        HIP_TRY(hipMemset2DAsync((char *)e->P.workers + offsetof(AzWorker, noise_row), sizeof(AzWorker), 0, sizeof(int),
                                 (size_t)e->n_trees * e->P.max_workers, (hipStream_t)stream));
    note_queued(e, stream);
    return AZ_OK;
}

static int check_workers(const az_engine *e, int n_workers)
{
    if (n_workers < 1 || n_workers > e->P.max_workers)
        return fail(AZ_ERR_ARG, "n_workers must be in 1..%d (az_set_threads with the concurrent schedule sets the limit)",
                    e->P.max_workers);
    return AZ_OK;
}

int az_select_workers(az_engine *e, int n_workers, float *d_boards, int *d_leaf_kind, void *stream)
{
    if (!e || !d_boards || !d_leaf_kind) return fail(AZ_ERR_ARG, "null argument");
    if (int rc = check_workers(e, n_workers)) return rc;
    DeviceGuard g(e->device);
    if (e->ops->workers_select(e->P, n_workers, d_boards, d_leaf_kind, nullptr, nullptr, stream))
        return fail(AZ_ERR_HIP, "workers_select launch failed");
    note_queued(e, stream);
    return AZ_OK;
}

int az_expand_backup_workers(az_engine *e, int n_workers, const float *d_value, const float *d_prior, int prior_is_log, void *stream)
{
    if (!e || !d_value || !d_prior) return fail(AZ_ERR_ARG, "null argument");
    if (int rc = check_workers(e, n_workers)) return rc;
    DeviceGuard g(e->device);
    if (e->ops->workers_backup(e->P, n_workers, d_value, d_prior, prior_is_log, stream))
        return fail(AZ_ERR_HIP, "workers_backup launch failed");
    note_queued(e, stream);
    return AZ_OK;
}

int az_select(az_engine *e, float *d_boards, int *d_leaf_kind, void *stream)
{
    if (!e || !d_boards || !d_leaf_kind) return fail(AZ_ERR_ARG, "null argument");
    DeviceGuard g(e->device);
    if (e->ops->select(e->P, d_boards, d_leaf_kind, nullptr, nullptr, stream)) return fail(AZ_ERR_HIP, "select launch failed");
    note_queued(e, stream);
    return AZ_OK;
}

int az_expand_backup(az_engine *e, const float *d_value, const float *d_prior, int prior_is_log, void *stream)
{
    if (!e || !d_value || !d_prior) return fail(AZ_ERR_ARG, "null argument");
    DeviceGuard g(e->device);
    if (e->ops->expand_backup(e->P, d_value, d_prior, prior_is_log, stream))
        return fail(AZ_ERR_HIP, "expand_backup launch failed");
    note_queued(e, stream);
    return AZ_OK;
}

// Staging of one search step between az_select, az_net_eval and az_expand_backup
// for n trees of `size` actions: leaf boards float [n][2][size], log-priors float
// [n][size], values float [n], leaf codes int [n], two bit masks of the trees whose
// leaf is to be expanded (steps alternate between them); every section 256-byte aligned.
struct StageLayout {
    size_t boards, logp, value, kinds, mask[2], mask_bytes, flags, flag_bytes, bytes;
};
static StageLayout stage_layout(int n, int size)
{
    auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
    StageLayout L;
    L.boards = 0;
    L.logp = up((size_t)n * 2 * size * sizeof(float));
    L.value = L.logp + up((size_t)n * size * sizeof(float));
    L.kinds = L.value + up((size_t)n * sizeof(float));
    L.mask_bytes = up((size_t)((n + 31) / 32) * AZ_MASK_STRIDE * sizeof(unsigned));
    L.mask[0] = L.kinds + up((size_t)n * sizeof(int));
    L.mask[1] = L.mask[0] + L.mask_bytes;
    // the exchange area of the one-launch search with concurrent workers (fewer positions than the device has CUs)
    L.flag_bytes = n <= 512 ? up(AZ_CLUSTER_FLAG_INTS(n, size) * sizeof(int)) : 0;
    L.flags = L.mask[1] + L.mask_bytes;
    L.bytes = L.flags + L.flag_bytes;
    return L;
}

// Cmcts::simulate with the network evaluated by az_net_eval: every search step is
// three launches on `stream` (select, network, expand + backup), the host never waits.
// Tree groups of the network search.  With all trees in one batch the tree kernels of a step wait for the last
// workgroups of the evaluator launch before them and the evaluator waits for the tree kernels: the GPU is partly idle
// at every such seam.  Split into groups that run the same launch sequence on streams of their own, one group's tree
// kernels and the tails of its evaluator launches overlap with the other group's evaluation (measured on MI355X, 8192
// trees: 15x15 25.8 -> 27.4 M searches/s, 20x20 12.1 -> 12.6).  Trees are independent, so the results do not change.
#define AZ_MAX_GROUPS 4
struct NetGroup {
    AzParams P;      // the engine's parameters with tree_base / n_trees set to the group
    int base, count; // trees [base, base + count)
    hipStream_t st;
};

static int net_groups(az_engine *e, int width, void *stream, NetGroup (&g)[AZ_MAX_GROUPS])
{
    // two groups once each still fills the evaluator's grid (and stays above the batch sizes for which az_net_eval
    // picks its low-latency kernel: a group must get the kernel the whole batch would get).  Measured at 15x15, one
    // group against two, M searches/s by leaves per step: 1024: 15.5 / 13.2 and 23.1 / 11.2; 2048: 21.9 / 24.2 and
    // 19.0 / 20.9; 2560: 14.6 / 20.4; 4096: 23.9 / 25.3; at 20x20 (two wavefronts per position in the evaluator) 2048
    // leaves: 10.6 / 9.9, 4096 and more: two groups ahead
    int want = (long long)e->n_trees * width >= (e->shape <= 16 ? 2048 : 4096) ? 2 : 1;
    if (const char *w = getenv("AZ_NET_GROUPS")) want = atoi(w); // tuning / test override (the caller then fixes the kernel form)
    if (want > 1 + e->n_aux_streams) want = 1 + e->n_aux_streams;
    if (want < 1) want = 1;
    // group boundaries on multiples of 32 trees: a group's positions then own whole words of the position masks
    const int blocks = (e->n_trees + 31) / 32;
    if (want > blocks) want = blocks;
    int n = 0, at = 0;
    for (int k = 0; k < want; k++) {
        const int hi = k + 1 == want ? e->n_trees : (int)((long long)blocks * (k + 1) / want) * 32;
        if (hi <= at) continue;
        g[n].P = e->P;
        g[n].P.tree_base = at;
        g[n].P.n_trees = hi - at;
        g[n].base = at;
        g[n].count = hi - at;
        g[n].st = n == 0 ? (hipStream_t)stream : e->aux_streams[n - 1];
        at = hi;
        n++;
    }
    return n;
}

// Cmcts::simulate with the network evaluated by az_net_eval: every search step is two launches per tree group
// (network; backup + next select), the host never waits.
static int simulate_net_steps(az_engine *e, int n, const float *d_params, int front, int pol, int val, char *stage,
                              void *stream, bool one_launch_workers)
{
    // thread fan-out of cmcts.cpp:98-119: th_num workers of `per` searches
    int th_num = 1, per = n;
    if (e->threads > 1) {
        th_num = e->threads;
        if (n < th_num) { th_num = n; per = 1; }
        else if (n % th_num == 0) per = n / th_num;
        else per = n / th_num + 1;
    }
    const bool concurrent = th_num > 1 && e->schedule == AZ_WORKERS_CONCURRENT;
    const int width = concurrent ? th_num : 1; // leaves per tree and step
    const StageLayout L = stage_layout(e->n_trees * width, e->size);
    float *boards = reinterpret_cast<float *>(stage + L.boards);
    float *logp = reinterpret_cast<float *>(stage + L.logp);
    float *value = reinterpret_cast<float *>(stage + L.value);
    int *kinds = reinterpret_cast<int *>(stage + L.kinds);
    unsigned *mask[2] = {reinterpret_cast<unsigned *>(stage + L.mask[0]), reinterpret_cast<unsigned *>(stage + L.mask[1])};
    // At most one tree per CU and one worker at a time: the launches of a search step would cost more than their
    // work, so the whole call is one launch (k_net_search; same arithmetic).  AZ_NET_SEARCH=0 in the environment, or a
    // kernel form forced through az_net_set_form, keeps the stepped launches.
    if (!concurrent) {
        int cus = 0;
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, e->device);
        const char *sw = getenv("AZ_NET_SEARCH");
        if (e->n_trees <= cus && az_net_get_form() == AZ_NET_FORM_AUTO && !(sw && atoi(sw) == 0)) {
            if (e->ops->net_search(e->P, th_num, per, d_params, front, pol, val, stream))
                return fail(AZ_ERR_HIP, "net_search launch failed");
            e->net_path = AZ_NET_PATH_ONE_LAUNCH;
            return AZ_OK;
        }
    }
    if (concurrent)
        if (int rc = check_workers(e, th_num)) return rc;
    HIP_TRY(hipMemsetAsync(stage + L.mask[0], 0, 2 * L.mask_bytes, (hipStream_t)stream));
    if (concurrent) // every worker builds a generator of its own (cmcts.cpp:132-133)
        HIP_TRY(hipMemset2DAsync((char *)e->P.workers + offsetof(AzWorker, noise_row), sizeof(AzWorker), 0, sizeof(int),
                                 (size_t)e->n_trees * e->P.max_workers, (hipStream_t)stream));
    // Concurrent workers, and every tree can have th_num + 1 CUs to itself: the whole call is one launch here too
    // (k_net_search_workers; same arithmetic, same switches as above).  Not from the allocation-free entry point, whose
    // callers may be capturing the stream (a cooperative launch cannot be captured).
    if (concurrent && one_launch_workers && L.flag_bytes != 0 && th_num <= 8) {
        int cus = 0;
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, e->device);
        const char *sw = getenv("AZ_NET_SEARCH");
        if ((long long)e->n_trees * (th_num + 1) <= cus && az_net_get_form() == AZ_NET_FORM_AUTO && !(sw && atoi(sw) == 0)) {
            HIP_TRY(hipMemsetAsync(stage + L.flags, 0, L.flag_bytes, (hipStream_t)stream)); // every tag: no step yet
            const int rc = e->ops->net_search_workers(e->P, th_num, per, d_params, front, pol, val, kinds,
                                                      reinterpret_cast<int *>(stage + L.flags), stream);
            if (rc == 0) { e->net_path = AZ_NET_PATH_CLUSTER; return AZ_OK; }
            if (rc == 1) return fail(AZ_ERR_HIP, "net_search_workers launch failed");
        }
    }

    e->net_path = AZ_NET_PATH_STEPPED;
    NetGroup groups[AZ_MAX_GROUPS];
    const int ng = net_groups(e, width, stream, groups);
    if (ng > 1) { // the groups' streams start behind everything queued on the caller's stream so far
        HIP_TRY(hipEventRecord(e->ev_fork, (hipStream_t)stream));
        for (int k = 1; k < ng; k++) HIP_TRY(hipStreamWaitEvent(groups[k].st, e->ev_fork, 0));
    }
    // the network on a group's leaves: position p of the group is position base * width + p of the staging arrays
    auto evaluate = [&](const NetGroup &g, const unsigned *m) -> int {
        const size_t p0 = (size_t)g.base * width;
        return az_net_eval_masked(e->shape, g.count * width, boards + p0 * 2 * e->size, kinds + p0, m + (p0 / 32) * AZ_MASK_STRIDE,
                                  d_params, front, pol, val, value + p0, logp + p0 * e->size, g.st);
    };
    int step = 0;
    if (concurrent) {
        // the workers in lock step: per step their th_num selects per tree, ONE evaluation of all th_num leaves of every
        // tree of a group, their th_num backups; the backups of a step and the selects of the next share a launch
        for (int k = 0; k < ng; k++)
            if (e->ops->workers_select(groups[k].P, th_num, boards, kinds, mask[0], mask[1], groups[k].st))
                return fail(AZ_ERR_HIP, "workers_select launch failed");
        for (int i = 0; i < per; i++) {
            unsigned *cur = mask[step & 1], *other = mask[(step + 1) & 1];
            step++;
            for (int k = 0; k < ng; k++) {
                const NetGroup &g = groups[k];
                if (int rc = evaluate(g, cur)) return fail(rc, "network evaluation launch failed");
                if (i + 1 < per) {
                    if (e->ops->workers_backup_select(g.P, th_num, value, logp, 1, boards, kinds, other, cur, g.st))
                        return fail(AZ_ERR_HIP, "workers_backup_select launch failed");
                } else if (e->ops->workers_backup(g.P, th_num, value, logp, 1, g.st)) {
                    return fail(AZ_ERR_HIP, "workers_backup launch failed");
                }
            }
        }
    } else {
        // sequential schedule: the workers one after another
        for (int w = 0; w < th_num; w++) {
            // a new worker builds a new generator (cmcts.cpp:132-133): the noise stream restarts (a strided memset per
            // group, on the group's stream: the group's previous worker has finished there)
            for (int k = 0; k < ng; k++) {
                const NetGroup &g = groups[k];
                HIP_TRY(hipMemset2DAsync((char *)(e->P.trees + g.base) + offsetof(AzTree, noise_row), sizeof(AzTree), 0, sizeof(int),
                                         g.count, g.st));
                // per search: select, network, expand + backup; the backup of a search and the select of the
                // worker's next search share a launch
                if (e->ops->select(g.P, boards, kinds, mask[step & 1], mask[(step + 1) & 1], g.st))
                    return fail(AZ_ERR_HIP, "select launch failed");
            }
            for (int i = 0; i < per; i++) {
                unsigned *cur = mask[step & 1], *other = mask[(step + 1) & 1];
                step++;
                for (int k = 0; k < ng; k++) {
                    const NetGroup &g = groups[k];
                    if (int rc = evaluate(g, cur)) return fail(rc, "network evaluation launch failed");
                    if (i + 1 < per) { // (the next search fills `other` and clears `cur` for the one after it)
                        if (e->ops->backup_select(g.P, value, logp, 1, boards, kinds, other, cur, g.st))
                            return fail(AZ_ERR_HIP, "backup_select launch failed");
                    } else if (e->ops->expand_backup(g.P, value, logp, 1, g.st)) {
                        return fail(AZ_ERR_HIP, "expand_backup launch failed");
                    }
                }
            }
        }
    }
    for (int k = 1; k < ng; k++) { // the caller's stream continues behind every group
        HIP_TRY(hipEventRecord(e->ev_join[k - 1], groups[k].st));
        HIP_TRY(hipStreamWaitEvent((hipStream_t)stream, e->ev_join[k - 1], 0));
    }
    return AZ_OK;
}

int az_simulate_net(az_engine *e, int n, const float *d_params, int front_layers, int policy_layers, int value_layers,
                    void *stream)
{
    if (!e) return fail(AZ_ERR_ARG, "null engine");
    if (n < 1) return fail(AZ_ERR_BAD_SIMS, "Invalid input %d must be at least 1!", n);
    if (!d_params || az_net_param_floats(front_layers, policy_layers, value_layers) == 0)
        return fail(AZ_ERR_NO_PREDICTOR, "Predictor missing!");
    if (int rc = check_noise(e)) return rc;
    DeviceGuard g(e->device);
    const int width = e->threads > 1 && e->schedule == AZ_WORKERS_CONCURRENT ? e->threads : 1; // leaves per tree and step
    const StageLayout L = stage_layout(e->n_trees * width, e->size);
    if (!e->d_stage || L.bytes > e->stage_bytes) {
        if (e->d_stage) {
            if (int rc = wait_queued(e)) return rc;
            (void)hipFree(e->d_stage);
            e->device_bytes -= e->stage_bytes;
            e->d_stage = nullptr;
            e->stage_bytes = 0;
        }
        HIP_TRY(hipMalloc((void **)&e->d_stage, L.bytes));
        e->stage_bytes = L.bytes;
        e->device_bytes += L.bytes;
    }
    const int rc = simulate_net_steps(e, n, d_params, front_layers, policy_layers, value_layers, e->d_stage, stream, true);
    note_queued(e, stream);
    return rc;
}

int az_simulate_net_noalloc(az_engine *e, int n, const float *d_params, int front_layers, int policy_layers,
                            int value_layers, void *stream)
{
    if (!e) return fail(AZ_ERR_ARG, "null engine");
    if (n < 1) return fail(AZ_ERR_BAD_SIMS, "Invalid input %d must be at least 1!", n);
    if (!d_params || az_net_param_floats(front_layers, policy_layers, value_layers) == 0)
        return fail(AZ_ERR_NO_PREDICTOR, "Predictor missing!");
    if (int rc = check_noise(e)) return rc;
    const int width = e->threads > 1 && e->schedule == AZ_WORKERS_CONCURRENT ? e->threads : 1; // leaves per tree and step
    const StageLayout L = stage_layout(e->n_trees * width, e->size);
    if (L.bytes > AZ_NOALLOC_SCRATCH_BYTES) {
        const size_t per_tree = (size_t)(3 * e->size + 2) * 4 * width;
        return fail(AZ_ERR_ARG,
                    "az_simulate_net_noalloc: %d trees of %dx%d need %zu bytes of step staging, the fixed buffer "
                    "holds %u (about %zu trees of this shape); use az_simulate_net",
                    e->n_trees, e->shape, e->shape, L.bytes, AZ_NOALLOC_SCRATCH_BYTES,
                    (size_t)(AZ_NOALLOC_SCRATCH_BYTES - 4096) / (per_tree + 1));
    }
    DeviceGuard g(e->device);
    char *ptr = g_scratch_ptr[e->device]; // resolved by az_create on this device
    if (!ptr) return fail(AZ_ERR_STATE, "the fixed staging buffer was not resolved on device %d", e->device);
    const int rc = simulate_net_steps(e, n, d_params, front_layers, policy_layers, value_layers, ptr, stream, false);
    note_queued(e, stream);
    return rc;
}

// Errors a kernel raises for a tree.  INVALID_MOVE and NOT_DIST leave the tree
// untouched, so reporting them clears them.  POOL_FULL, NOISE_ROWS and NO_ACTION
// stop a search half way (visits counted along the path, no child linked, no
// backup): such a tree stays in error -- and is skipped by every search -- until
// az_clear / az_clear_masked resets it.
static bool err_is_sticky(int code)
{
    return code == AZ_ERR_POOL_FULL || code == AZ_ERR_NOISE_ROWS || code == AZ_ERR_NO_ACTION;
}

static int sync_check_impl(az_engine *e, void *stream, int *tree_out)
{
    DeviceGuard g(e->device);
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    if (int rc = fetch_trees(e)) return rc;
    int first = -1, n_err = 0, n_transient = 0;
    for (int t = 0; t < e->n_trees; t++)
        if (e->h_trees[t].err) {
            if (first < 0) first = t;
            n_err++;
            if (!err_is_sticky(e->h_trees[t].err)) n_transient++;
        }
    if (tree_out) *tree_out = first;
    if (first < 0) return AZ_OK;
    const int code = e->h_trees[first].err;
    if (n_transient) { // write back the error words with the transient codes cleared
        std::vector<int> words(e->n_trees);
        for (int t = 0; t < e->n_trees; t++) words[t] = err_is_sticky(e->h_trees[t].err) ? e->h_trees[t].err : 0;
        HIP_TRY(hipMemcpy2D((char *)e->P.trees + offsetof(AzTree, err), sizeof(AzTree), words.data(), sizeof(int),
                            sizeof(int), e->n_trees, hipMemcpyHostToDevice));
    }
    if (n_err > 1) return fail(code, "%s (tree %d; %d trees in error)", kernel_err_text(code), first, n_err);
    return fail(code, "%s (tree %d)", kernel_err_text(code), first);
}

int az_sync_check(az_engine *e, void *stream)
{
    if (!e) return fail(AZ_ERR_ARG, "null engine");
    return sync_check_impl(e, stream, nullptr);
}

int az_sync_states(az_engine *e, void *stream, int *player, int *move_cnt, float *winner)
{
    if (!e) return fail(AZ_ERR_ARG, "null engine");
    if (int rc = sync_check_impl(e, stream, nullptr)) return rc;
    for (int t = 0; t < e->n_trees; t++) { // from the headers the check has just fetched
        if (player) player[t] = e->h_trees[t].player;
        if (move_cnt) move_cnt[t] = e->h_trees[t].move_cnt;
        if (winner) winner[t] = e->h_trees[t].winner;
    }
    return AZ_OK;
}

int az_get_errors(az_engine *e, void *stream, int *codes)
{
    if (!e || !codes) return fail(AZ_ERR_ARG, "null argument");
    DeviceGuard g(e->device);
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    if (int rc = fetch_trees(e)) return rc;
    for (int t = 0; t < e->n_trees; t++) codes[t] = e->h_trees[t].err;
    return AZ_OK;
}

int az_make_move(az_engine *e, const int *actions, int on_device, void *stream)
{
    if (!e || !actions) return fail(AZ_ERR_ARG, "null argument");
    DeviceGuard g(e->device);
    const int *d_act = actions;
    if (!on_device) {
        HIP_TRY(hipMemcpyAsync(e->d_actions, actions, sizeof(int) * e->n_trees, hipMemcpyHostToDevice,
                               (hipStream_t)stream));
        d_act = e->d_actions;
    }
    if (e->ops->make_move(e->P, d_act, stream)) return fail(AZ_ERR_HIP, "make_move launch failed");
    note_queued(e, stream);
    if (!on_device) {
        int tree = -1;
        int rc = sync_check_impl(e, stream, &tree);
        // an illegal move of this call comes first (worded like State::make_move,
        // state.cpp:85-86); h_trees still holds the error words as fetched
        for (int t = 0; t < e->n_trees; t++)
            if (e->h_trees[t].err == AZ_ERR_INVALID_MOVE)
                return fail(AZ_ERR_INVALID_MOVE, "Invalid move %d (tree %d)", actions[t], t);
        (void)tree;
        return rc;
    }
    return AZ_OK;
}

int az_make_movexy(az_engine *e, int tree, int x, int y)
{
    if (int rc = check_tree(e, tree)) return rc;
    if (y >= e->shape || x >= e->shape) // cmcts.cpp:299-300
        return fail(AZ_ERR_BAD_XY, "Invalid index y:%d or x: %d (range is 0-%d)!", y, x, e->size);
    std::vector<int> act(e->n_trees, -1);
    act[tree] = y * e->shape + x;
    if (act[tree] < 0) return fail(AZ_ERR_INVALID_MOVE, "Invalid move %d (tree %d)", act[tree], tree);
    return az_make_move(e, act.data(), 0, nullptr);
}

int az_select_greedy(az_engine *e, int *d_actions, void *stream)
{
    if (!e || !d_actions) return fail(AZ_ERR_ARG, "null argument");
    DeviceGuard g(e->device);
    if (e->ops->select_greedy(e->P, d_actions, stream)) return fail(AZ_ERR_HIP, "select_greedy launch failed");
    note_queued(e, stream);
    return AZ_OK;
}

int az_choose_moves(az_engine *e, const double *d_u, int greedy, int *d_actions, float *d_pi, float *d_board,
                    void *stream)
{
    if (!e || !d_actions) return fail(AZ_ERR_ARG, "null argument");
    if (!greedy && !d_u) return fail(AZ_ERR_ARG, "sampling needs one uniform number per tree (d_u)");
    DeviceGuard g(e->device);
    if (e->ops->choose_moves(e->P, d_u, greedy ? 1 : 0, d_actions, d_pi, d_board, stream))
        return fail(AZ_ERR_HIP, "choose_moves launch failed");
    note_queued(e, stream);
    return AZ_OK;
}

int az_fill_noise_dirichlet(az_engine *e, int rows, unsigned long long seed, void *stream)
{
    if (!e) return fail(AZ_ERR_ARG, "null engine");
    if (rows < 1) return fail(AZ_ERR_ARG, "rows must be at least 1");
    if (!(e->alpha > 0.)) return fail(AZ_ERR_ARG, "alpha must be positive");
    DeviceGuard g(e->device);
    if (rows > e->noise_cap) {
        if (int rc_ = wait_queued(e)) return rc_; // searches in flight may still read the old table
        if (e->d_noise) (void)hipFree(e->d_noise);
        e->d_noise = nullptr;
        e->noise_cap = 0;
        e->P.noise = nullptr;
        e->P.noise_rows = 0;
        HIP_TRY(hipMalloc((void **)&e->d_noise, (size_t)rows * e->sp * sizeof(double)));
        e->noise_cap = rows;
    }
    if (e->ops->dirichlet(e->d_noise, rows, e->alpha, seed, stream)) return fail(AZ_ERR_HIP, "dirichlet launch failed");
    e->P.noise = e->d_noise;
    e->P.noise_rows = rows;
    note_queued(e, stream);
    return AZ_OK;
}

int az_get_noise_table(az_engine *e, double *noise, int rows)
{
    if (!e || !noise) return fail(AZ_ERR_ARG, "null argument");
    if (rows < 0 || rows > e->P.noise_rows) return fail(AZ_ERR_ARG, "the table holds %d rows", e->P.noise_rows);
    DeviceGuard g(e->device);
    if (int rc_ = wait_queued(e)) return rc_;
    std::vector<double> padded((size_t)rows * e->sp);
    if (rows) HIP_TRY(hipMemcpy(padded.data(), e->d_noise, padded.size() * sizeof(double), hipMemcpyDeviceToHost));
    for (int r = 0; r < rows; r++)
        memcpy(noise + (size_t)r * e->size, &padded[(size_t)r * e->sp], sizeof(double) * e->size);
    return AZ_OK;
}

int az_get_counts(az_engine *e, int *counts, int *root_n)
{
    if (!e) return fail(AZ_ERR_ARG, "null engine");
    DeviceGuard g(e->device);
    if (int rc_ = wait_queued(e)) return rc_;
    if (e->ops->get_counts(e->P, e->d_counts, e->d_root_n, nullptr)) return fail(AZ_ERR_HIP, "get_counts launch failed");
    if (counts)
        HIP_TRY(hipMemcpy(counts, e->d_counts, sizeof(int) * (size_t)e->n_trees * e->size, hipMemcpyDeviceToHost));
    if (root_n) HIP_TRY(hipMemcpy(root_n, e->d_root_n, sizeof(int) * e->n_trees, hipMemcpyDeviceToHost));
    return AZ_OK;
}

// reads one record's arrays of `tree`/`slot` into host vectors (the node must
// be materialised, i.e. its nodeN > 0 or it is a root)
static int read_slot(az_engine *e, int tree, int slot, std::vector<int> *n, std::vector<int> *link,
                     std::vector<double> *w, std::vector<double> *p)
{
    const char *rec = e->P.nodes + ((size_t)tree * e->cap + slot) * e->rec_bytes;
    const int sp = e->sp;
    if (n) { n->resize(sp); HIP_TRY(hipMemcpy(n->data(), rec, sp * 4, hipMemcpyDeviceToHost)); }
    if (link) { link->resize(sp); HIP_TRY(hipMemcpy(link->data(), rec + (size_t)sp * 4, sp * 4, hipMemcpyDeviceToHost)); }
    if (w) { w->resize(sp); HIP_TRY(hipMemcpy(w->data(), rec + (size_t)sp * 8, sp * 8, hipMemcpyDeviceToHost)); }
    if (p) { p->resize(sp); HIP_TRY(hipMemcpy(p->data(), rec + (size_t)sp * 16, sp * 8, hipMemcpyDeviceToHost)); }
    return AZ_OK;
}

int az_get_prob(az_engine *e, int tree, float *prob)
{
    if (int rc = check_tree(e, tree)) return rc;
    if (!prob) return fail(AZ_ERR_ARG, "null argument");
    DeviceGuard g(e->device);
    if (int rc_ = wait_queued(e)) return rc_;
    AzTree t;
    HIP_TRY(hipMemcpy(&t, e->P.trees + tree, sizeof t, hipMemcpyDeviceToHost));
    std::vector<int> n(e->sp, 0);
    int node_n = -1; // unexpanded root (node.cpp:6)
    if (t.root >= 0) {
        node_n = t.root_n;
        if (int rc = read_slot(e, tree, t.root, &n, nullptr, nullptr, nullptr)) return rc;
    }
    // cmcts.cpp:323-339
    const double sum = (double)node_n;
    double prob_sum = 0;
    for (int i = 0; i < e->size; i++) {
        prob[i] = (float)((double)n[i] / sum);
        prob_sum += prob[i];
    }
    if (std::round(prob_sum * 1000.) != 1000) return fail(AZ_ERR_NOT_DIST, "Not a distribution");
    return AZ_OK;
}

int az_get_raw_board(az_engine *e, int tree, signed char *board)
{
    if (int rc = check_tree(e, tree)) return rc;
    if (!board) return fail(AZ_ERR_ARG, "null argument");
    DeviceGuard g(e->device);
    if (int rc_ = wait_queued(e)) return rc_;
    unsigned rows[64];
    HIP_TRY(hipMemcpy(rows, e->P.rows + (size_t)tree * 64, sizeof rows, hipMemcpyDeviceToHost));
    const int S = e->shape;
    for (int pl = 0; pl < 2; pl++)
        for (int r = 0; r < S; r++)
            for (int c = 0; c < S; c++) board[pl * e->size + r * S + c] = (signed char)((rows[pl * 32 + r] >> c) & 1u);
    return AZ_OK;
}

int az_get_raw_boards(az_engine *e, signed char *boards)
{
    if (!e || !boards) return fail(AZ_ERR_ARG, "null argument");
    DeviceGuard g(e->device);
    if (int rc_ = wait_queued(e)) return rc_;
    std::vector<unsigned> rows((size_t)e->n_trees * 64);
    HIP_TRY(hipMemcpy(rows.data(), e->P.rows, rows.size() * sizeof(unsigned), hipMemcpyDeviceToHost));
    const int S = e->shape;
    for (int t = 0; t < e->n_trees; t++) {
        signed char *b = boards + (size_t)t * 2 * e->size;
        for (int pl = 0; pl < 2; pl++)
            for (int r = 0; r < S; r++) {
                const unsigned bits = rows[(size_t)t * 64 + pl * 32 + r];
                for (int c = 0; c < S; c++) b[pl * e->size + r * S + c] = (signed char)((bits >> c) & 1u);
            }
    }
    return AZ_OK;
}

int az_get_board(az_engine *e, int tree, float *board)
{
    if (int rc = check_tree(e, tree)) return rc;
    if (!board) return fail(AZ_ERR_ARG, "null argument");
    std::vector<signed char> raw((size_t)2 * e->size);
    if (int rc = az_get_raw_board(e, tree, raw.data())) return rc;
    int player = 0;
    if (int rc = az_get_state(e, tree, &player, nullptr, nullptr)) return rc;
    // state.cpp:43-44: planes swapped when player == 1
    for (int pl = 0; pl < 2; pl++) {
        const int src = player == 1 ? 1 - pl : pl;
        for (int i = 0; i < e->size; i++) board[pl * e->size + i] = (float)raw[src * e->size + i];
    }
    return AZ_OK;
}

int az_get_node(az_engine *e, int tree, const int *path, int len, int *found, int *node_n, int *counts, double *w,
                double *p)
{
    if (int rc = check_tree(e, tree)) return rc;
    if (!found || (len > 0 && !path)) return fail(AZ_ERR_ARG, "null argument");
    DeviceGuard g(e->device);
    if (int rc_ = wait_queued(e)) return rc_;
    *found = 0;
    AzTree t;
    HIP_TRY(hipMemcpy(&t, e->P.trees + tree, sizeof t, hipMemcpyDeviceToHost));
    int slot = t.root;
    if (slot < 0) return AZ_OK;
    int nn = t.root_n; // Node::nodeN of the node the walk stands on
    for (int i = 0; i < len; i++) {
        if (path[i] < 0 || path[i] >= e->size) return fail(AZ_ERR_ARG, "Index is out of bounds"); // node.cpp:139-140
        if (nn == 0 && i > 0) return AZ_OK; // never selected from: no children yet
        int link = 0, cnt = 0;
        const char *rec = e->P.nodes + ((size_t)tree * e->cap + slot) * e->rec_bytes;
        HIP_TRY(hipMemcpy(&link, rec + (size_t)e->sp * 4 + (size_t)path[i] * 4, sizeof(int), hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(&cnt, rec + (size_t)path[i] * 4, sizeof(int), hipMemcpyDeviceToHost));
        if (link == 0) return AZ_OK;
        slot = link - 1;
        int deficit = 0; // visits of the action by concurrent workers that met in the unexpanded child (AzParams::aux)
        HIP_TRY(hipMemcpy(&deficit, e->P.aux + (size_t)tree * e->cap + slot, sizeof(int), hipMemcpyDeviceToHost));
        nn = cnt - 1 - (deficit > 0 ? deficit : 0); // the first visit of the action created the child
    }
    const bool fresh = nn == 0 && len > 0; // statistics not materialised yet: implicit zeros
    std::vector<int> n;
    std::vector<double> vw, vp;
    if (int rc = read_slot(e, tree, slot, (counts && !fresh) ? &n : nullptr, nullptr, (w && !fresh) ? &vw : nullptr,
                           p ? &vp : nullptr))
        return rc;
    if (node_n) *node_n = nn;
    if (counts) { if (fresh) memset(counts, 0, sizeof(int) * e->size); else memcpy(counts, n.data(), sizeof(int) * e->size); }
    if (w) { if (fresh) memset(w, 0, sizeof(double) * e->size); else memcpy(w, vw.data(), sizeof(double) * e->size); }
    if (p) memcpy(p, vp.data(), sizeof(double) * e->size);
    *found = 1;
    return AZ_OK;
}

int az_get_stats(az_engine *e, unsigned long long *sims, unsigned long long *expansions, unsigned long long *terminals,
                 unsigned long long *select_calls)
{
    if (!e) return fail(AZ_ERR_ARG, "null engine");
    DeviceGuard g(e->device);
    if (int rc_ = wait_queued(e)) return rc_;
    if (int rc = fetch_trees(e)) return rc;
    unsigned long long s[4] = {0, 0, 0, 0};
    for (const auto &t : e->h_trees)
        for (int k = 0; k < 4; k++) s[k] += t.stats[k];
    if (sims) *sims = s[0];
    if (expansions) *expansions = s[1];
    if (terminals) *terminals = s[2];
    if (select_calls) *select_calls = s[3];
    return AZ_OK;
}

int az_reset_stats(az_engine *e)
{
    if (!e) return fail(AZ_ERR_ARG, "null engine");
    DeviceGuard g(e->device);
    if (int rc_ = wait_queued(e)) return rc_;
    HIP_TRY(hipMemset2D((char *)e->P.trees + offsetof(AzTree, stats), sizeof(AzTree), 0, sizeof(unsigned long long) * 4,
                        e->n_trees));
    return AZ_OK;
}

int az_get_pool_usage(az_engine *e, int tree, int *used, int *max_used)
{
    if (!e) return fail(AZ_ERR_ARG, "null engine");
    DeviceGuard g(e->device);
    if (int rc_ = wait_queued(e)) return rc_;
    if (int rc = fetch_trees(e)) return rc;
    if (used) {
        if (int rc = check_tree(e, tree)) return rc;
        *used = e->h_trees[tree].n_order;
    }
    if (max_used) {
        int m = 0;
        for (const auto &t : e->h_trees) m = t.n_order > m ? t.n_order : m;
        *max_used = m;
    }
    return AZ_OK;
}

// ---- device-resident game loops (az_slots.hip)
static int check_pair(const az_engine *a, const az_engine *b)
{
    if (!a) return fail(AZ_ERR_ARG, "null engine");
    if (b && (b->n_trees != a->n_trees || b->shape != a->shape || b->device != a->device))
        return fail(AZ_ERR_ARG, "the two engines of a game loop must agree in trees, board and device");
    return AZ_OK;
}

int az_slots_admit(az_engine *a, az_engine *b, int *d_game_of, int *d_ply, int *d_counters, int n_games, const int *openings,
                   int n_open, const int *d_game_openings, int n_game_open, void *stream)
{
    if (int rc = check_pair(a, b)) return rc;
    if (!d_game_of || !d_ply || !d_counters || n_games < 0 || n_open < 0 || (n_open > 0 && !openings) || n_game_open < 0 ||
        (n_game_open > 0 && !d_game_openings))
        return fail(AZ_ERR_ARG, "bad argument");
    DeviceGuard g(a->device);
    hipStream_t st = (hipStream_t)stream;
    if (az_slots_launch_admit(a->P.trees, b ? b->P.trees : nullptr, a->n_trees, d_game_of, d_ply, d_counters, n_games, a->d_mask, st))
        return fail(AZ_ERR_HIP, "admit launch failed");
    if (a->ops->clear(a->P, a->d_mask, stream)) return fail(AZ_ERR_HIP, "clear launch failed");
    if (b && b->ops->clear(b->P, a->d_mask, stream)) return fail(AZ_ERR_HIP, "clear launch failed");
    for (int k = 0; k < n_open + n_game_open; k++) {
        // first the opening every game shares (tools.make_init_moves), then the game's own
        const int *per_game = k < n_open ? nullptr : d_game_openings + (size_t)(k - n_open) * n_games;
        const int action = k < n_open ? openings[k] : -1;
        if (k < n_open && (action < 0 || action >= a->size)) return fail(AZ_ERR_INVALID_MOVE, "Invalid move %d (opening)", action);
        if (az_slots_launch_fill_actions(a->d_mask, a->n_trees, action, per_game, d_game_of, a->d_actions, st))
            return fail(AZ_ERR_HIP, "opening launch failed");
        if (a->ops->make_move(a->P, a->d_actions, stream)) return fail(AZ_ERR_HIP, "make_move launch failed");
        if (b && b->ops->make_move(b->P, a->d_actions, stream)) return fail(AZ_ERR_HIP, "make_move launch failed");
    }
    note_queued(a, stream);
    if (b) note_queued(b, stream);
    return AZ_OK;
}

int az_slots_turn(az_engine *e, const int *d_game_of, const int *d_ply, int by_game, int parity, void *stream)
{
    if (!e || !d_game_of || !d_ply) return fail(AZ_ERR_ARG, "null argument");
    DeviceGuard g(e->device);
    if (az_slots_launch_turn(e->P.trees, e->n_trees, d_game_of, d_ply, by_game ? 1 : 0, parity < 0 ? -1 : (parity & 1),
                             (hipStream_t)stream))
        return fail(AZ_ERR_HIP, "turn launch failed");
    note_queued(e, stream);
    return AZ_OK;
}

int az_slots_choose(az_engine *e, const int *d_game_of, const int *d_ply, int tau, unsigned long long seed, int *d_actions,
                    float *d_stage, int max_plies, void *stream)
{
    if (!e || !d_game_of || !d_ply || !d_actions) return fail(AZ_ERR_ARG, "null argument");
    if (d_stage && max_plies < 1) return fail(AZ_ERR_ARG, "max_plies must be at least 1");
    DeviceGuard g(e->device);
    if (az_slots_launch_uniform(d_game_of, d_ply, e->n_trees, tau, seed, e->d_u, (hipStream_t)stream))
        return fail(AZ_ERR_HIP, "uniform launch failed");
    if (e->ops->choose_moves_staged(e->P, e->d_u, d_actions, d_stage, max_plies, d_ply, stream))
        return fail(AZ_ERR_HIP, "choose_moves launch failed");
    note_queued(e, stream);
    return AZ_OK;
}

int az_slots_finish(az_engine *a, az_engine *b, int *d_game_of, int *d_ply, int *d_counters, float *d_outcome, const float *d_stage,
                    int max_plies, float *d_arena, long long arena_records, int *d_game_off, int *d_game_len, void *stream)
{
    if (int rc = check_pair(a, b)) return rc;
    if (!d_game_of || !d_ply || !d_counters || !d_outcome) return fail(AZ_ERR_ARG, "null argument");
    const bool rec = d_stage != nullptr;
    if (rec && (!d_arena || !d_game_off || !d_game_len || max_plies < 1 || arena_records < 0 || arena_records > 0x7fffffffll))
        return fail(AZ_ERR_ARG, "records need an arena (at most 2^31-1 records), the two directory arrays and max_plies");
    DeviceGuard g(a->device);
    hipStream_t st = (hipStream_t)stream;
    int *eo = a->d_emit, *en = a->d_emit + a->n_trees, *eg = a->d_emit + 2 * (size_t)a->n_trees;
    if (az_slots_launch_finish(a->P.trees, b ? b->P.trees : nullptr, a->n_trees, d_game_of, d_ply, d_counters, d_outcome, rec ? 1 : 0,
                               arena_records, d_game_off, d_game_len, eo, en, eg, st))
        return fail(AZ_ERR_HIP, "finish launch failed");
    if (rec) {
        const int recf = 3 * a->size + 1;
        if (az_slots_launch_emit(d_stage, (long long)max_plies * recf, recf, a->n_trees, eo, en, eg, d_outcome, d_arena, st))
            return fail(AZ_ERR_HIP, "emit launch failed");
    }
    note_queued(a, stream);
    if (b) note_queued(b, stream);
    return AZ_OK;
}

#ifdef AZ_STAMPS
// diagnostic library only: per-tree phase cycle sums of the last fused launch
int az_debug_stamps(az_engine *e, unsigned long long *out)
{
    if (!e || !out) return fail(AZ_ERR_ARG, "null argument");
    DeviceGuard g(e->device);
    if (int rc_ = wait_queued(e)) return rc_;
    HIP_TRY(hipMemcpy(out, e->P.stamps, sizeof(unsigned long long) * 8 * e->n_trees, hipMemcpyDeviceToHost));
    return AZ_OK;
}
#endif

int az_timing_begin(az_engine *e, void *stream)
{
    if (!e) return fail(AZ_ERR_ARG, "null engine");
    DeviceGuard g(e->device);
    HIP_TRY(hipEventRecord(e->ev0, (hipStream_t)stream));
    return AZ_OK;
}

int az_timing_end(az_engine *e, void *stream, float *ms)
{
    if (!e || !ms) return fail(AZ_ERR_ARG, "null argument");
    DeviceGuard g(e->device);
    HIP_TRY(hipEventRecord(e->ev1, (hipStream_t)stream));
    HIP_TRY(hipEventSynchronize(e->ev1));
    HIP_TRY(hipEventElapsedTime(ms, e->ev0, e->ev1));
    return AZ_OK;
}

int az_timing_mark(az_engine *e, int slot, int end, void *stream)
{
    if (!e || slot < 0 || slot >= 65536) return fail(AZ_ERR_ARG, "bad timing slot");
    DeviceGuard g(e->device);
    const size_t idx = 2 * (size_t)slot + (end ? 1 : 0);
    if (e->ev_slots.size() <= idx) e->ev_slots.resize(2 * (size_t)slot + 2, nullptr);
    if (!e->ev_slots[idx]) HIP_TRY(hipEventCreate(&e->ev_slots[idx]));
    HIP_TRY(hipEventRecord(e->ev_slots[idx], (hipStream_t)stream));
    return AZ_OK;
}

int az_timing_elapsed(az_engine *e, int slot, float *ms)
{
    if (!e || !ms || slot < 0) return fail(AZ_ERR_ARG, "bad argument");
    const size_t idx = 2 * (size_t)slot;
    if (e->ev_slots.size() < idx + 2 || !e->ev_slots[idx] || !e->ev_slots[idx + 1])
        return fail(AZ_ERR_ARG, "timing slot was not marked at both ends");
    DeviceGuard g(e->device);
    HIP_TRY(hipEventSynchronize(e->ev_slots[idx + 1]));
    HIP_TRY(hipEventElapsedTime(ms, e->ev_slots[idx], e->ev_slots[idx + 1]));
    return AZ_OK;
}

} // extern "C"

"""ctypes binding of libazmcts.so (C ABI: include/azmcts.h).

`Engine` owns `n_trees` search trees on one GPU.  Device pointers are passed
as plain integers (e.g. `tensor.data_ptr()`), streams as the integer handle of
a hipStream_t (`torch.cuda.current_stream().cuda_stream`).
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# (AZ_LIBAZMCTS: another build of the library, e.g. an experiment variant, for A/B runs)
LIB_PATH = os.environ.get("AZ_LIBAZMCTS") or os.path.join(_HERE, "libazmcts.so")

EVAL_UNIFORM, EVAL_HASH = 0, 1
WORKERS_SEQUENTIAL, WORKERS_CONCURRENT = 0, 1
MAX_WORKERS = 64  # AZ_MAX_WORKERS: concurrent workers per tree
NET_PATH_NONE, NET_PATH_STEPPED, NET_PATH_ONE_LAUNCH, NET_PATH_CLUSTER = 0, 1, 2, 3
LEAF_NONE, LEAF_EXPAND, LEAF_TERMINAL = 0, 1, 2

ERR_INVALID_MOVE, ERR_BAD_SIMS, ERR_NOT_DIST, ERR_EPS_RANGE, ERR_BAD_XY = 1, 2, 3, 4, 5
ERR_NO_ACTION, ERR_NOISE_ROWS, ERR_INVALID_VALUE, ERR_NO_PREDICTOR = 6, 7, 8, 9
ERR_POOL_FULL, ERR_ARG, ERR_HIP, ERR_STATE = 10, 11, 12, 13

# every symbol include/azmcts.h declares: (name, restype, argtypes)
_vp, _i, _d, _f, _ull = C.c_void_p, C.c_int, C.c_double, C.c_float, C.c_ulonglong
SYMBOLS = [
    ("az_last_error", C.c_char_p, []),
    ("az_version", C.c_char_p, []),
    ("az_build_timestamp", C.c_char_p, []),
    ("az_create", _i, [C.POINTER(_vp), _i, _i, _i, _i, _d, _d]),
    ("az_destroy", None, [_vp]),
    ("az_n_trees", _i, [_vp]),
    ("az_shape", _i, [_vp]),
    ("az_nodes_per_tree", _i, [_vp]),
    ("az_device_bytes", _ull, [_vp]),
    ("az_set_eps", _i, [_vp, _d]),
    ("az_get_eps", _d, [_vp]),
    ("az_set_cpuct", _i, [_vp, _f]),
    ("az_get_cpuct", _f, [_vp]),
    ("az_set_alpha", _i, [_vp, _d]),
    ("az_get_alpha", _d, [_vp]),
    ("az_set_alpha_default", _i, [_vp]),
    ("az_set_threads", _i, [_vp, _i]),
    ("az_get_threads", _i, [_vp]),
    ("az_set_schedule", _i, [_vp, _i]),
    ("az_get_schedule", _i, [_vp]),
    ("az_get_net_path", _i, [_vp]),
    ("az_select_workers", _i, [_vp, _i, _vp, _vp, _vp]),
    ("az_expand_backup_workers", _i, [_vp, _i, _vp, _vp, _i, _vp]),
    ("az_set_cuda", _i, [_vp, _i]),
    ("az_get_cuda", _i, [_vp]),
    ("az_set_player", _i, [_vp, _i, _i]),
    ("az_get_state", _i, [_vp, _i, C.POINTER(_i), C.POINTER(_i), C.POINTER(_f)]),
    ("az_get_states", _i, [_vp, _vp, _vp, _vp]),
    ("az_clear", _i, [_vp, _i]),
    ("az_clear_masked", _i, [_vp, _vp]),
    ("az_set_active", _i, [_vp, _vp]),
    ("az_set_noise_table", _i, [_vp, _vp, _i]),
    ("az_simulate_fused", _i, [_vp, _i, _i, _f, _f, _ull, _vp]),
    ("az_begin_simulate", _i, [_vp, _vp]),
    ("az_select", _i, [_vp, _vp, _vp, _vp]),
    ("az_expand_backup", _i, [_vp, _vp, _vp, _i, _vp]),
    ("az_simulate_net", _i, [_vp, _i, _vp, _i, _i, _i, _vp]),
    ("az_simulate_net_noalloc", _i, [_vp, _i, _vp, _i, _i, _i, _vp]),
    ("az_get_errors", _i, [_vp, _vp, _vp]),
    ("az_make_move", _i, [_vp, _vp, _i, _vp]),
    ("az_make_movexy", _i, [_vp, _i, _i, _i]),
    ("az_select_greedy", _i, [_vp, _vp, _vp]),
    ("az_get_counts", _i, [_vp, _vp, _vp]),
    ("az_choose_moves", _i, [_vp, _vp, _i, _vp, _vp, _vp, _vp]),
    ("az_fill_noise_dirichlet", _i, [_vp, _i, C.c_ulonglong, _vp]),
    ("az_get_noise_table", _i, [_vp, _vp, _i]),
    ("az_net_param_floats", _i, [_i, _i, _i]),
    ("az_net_set_form", _i, [_i]),
    ("az_net_eval", _i, [_i, _i, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp]),
    ("az_aug_key_words", _i, [_i]),
    ("az_aug_rotations", _i, [_i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    ("az_aug_pack_keys", _i, [_i, _i, _vp, _vp, _vp, _vp]),
    ("az_aug_unique_average", _i, [_i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    ("az_get_prob", _i, [_vp, _i, _vp]),
    ("az_get_board", _i, [_vp, _i, _vp]),
    ("az_get_raw_board", _i, [_vp, _i, _vp]),
    ("az_get_raw_boards", _i, [_vp, _vp]),
    ("az_get_node", _i, [_vp, _i, _vp, _i, C.POINTER(_i), C.POINTER(_i), _vp, _vp, _vp]),
    ("az_get_stats", _i, [_vp] + [C.POINTER(_ull)] * 4),
    ("az_reset_stats", _i, [_vp]),
    ("az_get_pool_usage", _i, [_vp, _i, C.POINTER(_i), C.POINTER(_i)]),
    ("az_sync_check", _i, [_vp, _vp]),
    ("az_sync_states", _i, [_vp, _vp, _vp, _vp, _vp]),
    ("az_make_move_greedy", _i, [_vp, _vp]),
    ("az_restart_finished", _i, [_vp, _vp, _i, _vp, _vp]),
    ("az_slots_admit", _i, [_vp, _vp, _vp, _vp, _vp, _i, _vp, _i, _vp, _i, _vp]),
    ("az_slots_turn", _i, [_vp, _vp, _vp, _i, _i, _vp]),
    ("az_slots_choose", _i, [_vp, _vp, _vp, _i, _ull, _vp, _vp, _i, _vp]),
    ("az_slots_finish", _i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, C.c_longlong, _vp, _vp, _vp]),
    ("az_timing_mark", _i, [_vp, _i, _i, _vp]),
    ("az_timing_elapsed", _i, [_vp, _i, C.POINTER(_f)]),
    ("az_timing_begin", _i, [_vp, _vp]),
    ("az_timing_end", _i, [_vp, _vp, C.POINTER(_f)]),
]

_lib = None


def _load_hip_runtime():
    """libazmcts.so carries no DT_NEEDED for the HIP runtime, so that it shares
    the ONE runtime of the process.  PyTorch wheels bundle their own
    libamdhip64.so; if torch is installed that copy is loaded (whether or not
    torch has been imported yet), else the system ROCm one."""
    import importlib.util

    candidates = []
    spec = importlib.util.find_spec("torch")
    if spec is not None and spec.origin:
        candidates.append(os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so"))
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    candidates += [os.path.join(rocm, "lib", "libamdhip64.so"), "libamdhip64.so"]
    last = None
    for path in candidates:
        if os.path.isabs(path) and not os.path.exists(path):
            continue
        try:
            return C.CDLL(path, mode=C.RTLD_GLOBAL)
        except OSError as e:  # try the next candidate
            last = e
    raise ImportError(f"no HIP runtime (libamdhip64.so) could be loaded: {last}")


def load_library():
    """Loads libazmcts.so; raises if it has not been built (no CPU fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `make -C alphazero_amd/csrc` "
                "(or __graft_entry__.build()); the engine has no CPU fallback")
        _load_hip_runtime()
        L = C.CDLL(LIB_PATH)
        for name, res, args in SYMBOLS:
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


class EngineError(RuntimeError):
    """Failure reported by the C ABI; `.code` is the AZ_ERR_* value."""

    def __init__(self, code, msg):
        super().__init__(msg)
        self.code = code


def _ptr(x):
    if x is None:
        return None
    if isinstance(x, np.ndarray):
        return x.ctypes.data
    return int(x)


class Engine:
    def __init__(self, n_trees, shape, nodes_per_tree=2048, alpha=1.0, cpuct=1.0, device=0):
        self.L = load_library()
        h = C.c_void_p()
        rc = self.L.az_create(C.byref(h), device, n_trees, shape, nodes_per_tree, alpha, cpuct)
        self.h = h if rc == 0 else None
        self._check(rc)
        self.n_trees, self.shape, self.size = n_trees, shape, shape * shape
        self.device = device

    def _check(self, rc):
        if rc != 0:
            raise EngineError(rc, self.L.az_last_error().decode())

    def close(self):
        if getattr(self, "h", None):
            self.L.az_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- parameters (cmcts.cpp:46-81)
    def set_eps(self, v): self._check(self.L.az_set_eps(self.h, v))
    def get_eps(self): return self.L.az_get_eps(self.h)
    def set_cpuct(self, v): self._check(self.L.az_set_cpuct(self.h, v))
    def get_cpuct(self): return self.L.az_get_cpuct(self.h)
    def set_alpha(self, v): self._check(self.L.az_set_alpha(self.h, v))
    def get_alpha(self): return self.L.az_get_alpha(self.h)
    def set_alpha_default(self): self._check(self.L.az_set_alpha_default(self.h))
    def set_threads(self, v): self._check(self.L.az_set_threads(self.h, v))
    def get_threads(self): return self.L.az_get_threads(self.h)
    def set_schedule(self, v): self._check(self.L.az_set_schedule(self.h, int(v)))
    def get_schedule(self): return self.L.az_get_schedule(self.h)
    def get_net_path(self): return self.L.az_get_net_path(self.h)
    def set_cuda(self, v): self._check(self.L.az_set_cuda(self.h, v))
    def get_cuda(self): return self.L.az_get_cuda(self.h)
    def device_bytes(self): return self.L.az_device_bytes(self.h)

    # ---- per-tree state
    def set_player(self, tree, player): self._check(self.L.az_set_player(self.h, tree, player))

    def get_state(self, tree):
        p, m, w = C.c_int(), C.c_int(), C.c_float()
        self._check(self.L.az_get_state(self.h, tree, C.byref(p), C.byref(m), C.byref(w)))
        return p.value, m.value, w.value

    def get_states(self):
        p = np.empty(self.n_trees, np.int32)
        m = np.empty(self.n_trees, np.int32)
        w = np.empty(self.n_trees, np.float32)
        self._check(self.L.az_get_states(self.h, _ptr(p), _ptr(m), _ptr(w)))
        return p, m, w

    def clear(self, tree=-1): self._check(self.L.az_clear(self.h, tree))

    def clear_masked(self, mask):
        mask = np.ascontiguousarray(mask, np.uint8)
        assert mask.shape == (self.n_trees,)
        self._check(self.L.az_clear_masked(self.h, _ptr(mask)))

    def set_active(self, mask=None):
        if mask is not None:
            mask = np.ascontiguousarray(mask, np.uint8)
            assert mask.shape == (self.n_trees,)
        self._check(self.L.az_set_active(self.h, _ptr(mask)))

    def set_noise_table(self, noise):
        if noise is None:
            self._check(self.L.az_set_noise_table(self.h, None, 0))
            return
        noise = np.ascontiguousarray(noise, np.float64)
        assert noise.ndim == 2 and noise.shape[1] == self.size
        self._check(self.L.az_set_noise_table(self.h, _ptr(noise), noise.shape[0]))

    # ---- search
    def simulate_fused(self, n, eval_kind=EVAL_UNIFORM, const_prior=0.0, const_value=0.0, hash_seed=0,
                       stream=None, sync=True):
        self._check(self.L.az_simulate_fused(self.h, n, eval_kind, const_prior, const_value, hash_seed, stream))
        if sync:
            self.sync_check(stream)

    def begin_simulate(self, stream=None): self._check(self.L.az_begin_simulate(self.h, stream))

    def select(self, d_boards, d_leaf_kind, stream=None):
        self._check(self.L.az_select(self.h, _ptr(d_boards), _ptr(d_leaf_kind), stream))

    def expand_backup(self, d_value, d_prior, prior_is_log, stream=None):
        self._check(self.L.az_expand_backup(self.h, _ptr(d_value), _ptr(d_prior), int(prior_is_log), stream))

    def select_workers(self, n_workers, d_boards, d_leaf_kind, stream=None):
        self._check(self.L.az_select_workers(self.h, int(n_workers), _ptr(d_boards), _ptr(d_leaf_kind), stream))

    def expand_backup_workers(self, n_workers, d_value, d_prior, prior_is_log, stream=None):
        self._check(self.L.az_expand_backup_workers(self.h, int(n_workers), _ptr(d_value), _ptr(d_prior), int(prior_is_log), stream))

    def simulate_net(self, n, d_params, layers, stream=None, sync=True):
        """n searches per active tree with the fused network kernel as the predictor,
        all steps queued by one call (d_params: device pointer of the packed
        parameters, layers = (front, policy, value) counts)."""
        self._check(self.L.az_simulate_net(self.h, int(n), _ptr(d_params), int(layers[0]), int(layers[1]),
                                           int(layers[2]), stream))
        if sync:
            self.sync_check(stream)

    def simulate_net_noalloc(self, n, d_params, layers, stream=None, sync=True):
        """simulate_net with the step staging in the library's fixed per-GPU buffer (no allocation)."""
        self._check(self.L.az_simulate_net_noalloc(self.h, int(n), _ptr(d_params), int(layers[0]), int(layers[1]),
                                                   int(layers[2]), stream))
        if sync:
            self.sync_check(stream)

    def get_errors(self, stream=None):
        codes = np.zeros(self.n_trees, np.int32)
        self._check(self.L.az_get_errors(self.h, stream, _ptr(codes)))
        return codes

    def make_move(self, actions, stream=None):
        actions = np.ascontiguousarray(actions, np.int32)
        assert actions.shape == (self.n_trees,)
        self._check(self.L.az_make_move(self.h, _ptr(actions), 0, stream))

    def make_move_device(self, d_actions, stream=None):
        self._check(self.L.az_make_move(self.h, _ptr(d_actions), 1, stream))

    def make_movexy(self, tree, x, y): self._check(self.L.az_make_movexy(self.h, tree, x, y))

    def select_greedy(self, d_actions, stream=None):
        self._check(self.L.az_select_greedy(self.h, _ptr(d_actions), stream))

    def sync_check(self, stream=None): self._check(self.L.az_sync_check(self.h, stream))

    def sync_states(self, stream=None):
        """sync_check + get_states with one copy of the tree headers."""
        player = np.zeros(self.n_trees, np.int32)
        move_cnt = np.zeros(self.n_trees, np.int32)
        winner = np.zeros(self.n_trees, np.float32)
        self._check(self.L.az_sync_states(self.h, stream, _ptr(player), _ptr(move_cnt), _ptr(winner)))
        return player, move_cnt, winner

    # ---- results
    def choose_moves(self, d_u, greedy, d_actions, d_pi=None, d_board=None, stream=None):
        """Move decision + training record fields on the GPU (device pointers)."""
        self._check(self.L.az_choose_moves(self.h, _ptr(d_u) if d_u is not None else None, int(bool(greedy)),
                                           _ptr(d_actions), _ptr(d_pi) if d_pi is not None else None,
                                           _ptr(d_board) if d_board is not None else None, stream))

    def fill_noise_dirichlet(self, rows, seed, stream=None):
        self._check(self.L.az_fill_noise_dirichlet(self.h, int(rows), int(seed) & 0xFFFFFFFFFFFFFFFF, stream))

    def get_noise_table(self, rows):
        out = np.zeros((rows, self.size), np.float64)
        self._check(self.L.az_get_noise_table(self.h, _ptr(out), rows))
        return out

    def get_counts(self):
        counts = np.empty((self.n_trees, self.size), np.int32)
        root_n = np.empty(self.n_trees, np.int32)
        self._check(self.L.az_get_counts(self.h, _ptr(counts), _ptr(root_n)))
        return counts, root_n

    def get_prob(self, tree):
        out = np.empty(self.size, np.float32)
        self._check(self.L.az_get_prob(self.h, tree, _ptr(out)))
        return out

    def get_board(self, tree):
        out = np.empty((2, self.shape, self.shape), np.float32)
        self._check(self.L.az_get_board(self.h, tree, _ptr(out)))
        return out

    def get_raw_board(self, tree):
        out = np.empty((2, self.shape, self.shape), np.int8)
        self._check(self.L.az_get_raw_board(self.h, tree, _ptr(out)))
        return out

    def get_raw_boards(self):
        out = np.empty((self.n_trees, 2, self.shape, self.shape), np.int8)
        self._check(self.L.az_get_raw_boards(self.h, _ptr(out)))
        return out

    def get_node(self, tree, path):
        path = np.ascontiguousarray(path, np.int32)
        found, node_n = C.c_int(), C.c_int()
        counts = np.empty(self.size, np.int32)
        w = np.empty(self.size, np.float64)
        p = np.empty(self.size, np.float64)
        self._check(self.L.az_get_node(self.h, tree, _ptr(path) if len(path) else None, len(path),
                                       C.byref(found), C.byref(node_n), _ptr(counts), _ptr(w), _ptr(p)))
        if not found.value:
            return None
        return node_n.value, counts, w, p

    def get_stats(self):
        v = [C.c_ulonglong() for _ in range(4)]
        self._check(self.L.az_get_stats(self.h, *[C.byref(x) for x in v]))
        return dict(sims=v[0].value, expansions=v[1].value, terminals=v[2].value, select_calls=v[3].value)

    def reset_stats(self): self._check(self.L.az_reset_stats(self.h))

    def pool_usage(self, tree=0):
        u, m = C.c_int(), C.c_int()
        self._check(self.L.az_get_pool_usage(self.h, tree, C.byref(u), C.byref(m)))
        return u.value, m.value

    def make_move_greedy(self, stream=None):
        """argmax of the root visit counts + make_move for every active tree, on the GPU."""
        self._check(self.L.az_make_move_greedy(self.h, stream))

    def restart_finished(self, d_openings, n_plies, d_count=None, stream=None):
        """clear + opening moves for every finished game, on the GPU (device pointers)."""
        self._check(self.L.az_restart_finished(self.h, _ptr(d_openings) if n_plies else None, int(n_plies),
                                               _ptr(d_count) if d_count is not None else None, stream))

    # ---- device-resident game loops (slot bookkeeping in device arrays; everything asynchronous)
    def slots_admit(self, other, d_game_of, d_ply, d_counters, n_games, openings=(), stream=None, d_game_openings=None,
                    n_game_open=0):
        op = np.ascontiguousarray(openings, np.int32)
        self._check(self.L.az_slots_admit(self.h, other.h if other is not None else None, _ptr(d_game_of), _ptr(d_ply),
                                          _ptr(d_counters), int(n_games), _ptr(op) if op.size else None, int(op.size),
                                          _ptr(d_game_openings) if n_game_open else None, int(n_game_open), stream))

    def slots_turn(self, d_game_of, d_ply, by_game, parity, stream=None):
        self._check(self.L.az_slots_turn(self.h, _ptr(d_game_of), _ptr(d_ply), int(bool(by_game)), int(parity), stream))

    def slots_choose(self, d_game_of, d_ply, tau, seed, d_actions, d_stage=None, max_plies=0, stream=None):
        self._check(self.L.az_slots_choose(self.h, _ptr(d_game_of), _ptr(d_ply), int(tau), int(seed) & 0xFFFFFFFFFFFFFFFF,
                                           _ptr(d_actions), _ptr(d_stage) if d_stage is not None else None, int(max_plies),
                                           stream))

    def slots_finish(self, other, d_game_of, d_ply, d_counters, d_outcome, d_stage=None, max_plies=0, d_arena=None,
                     arena_records=0, d_game_off=None, d_game_len=None, stream=None):
        opt = lambda x: _ptr(x) if x is not None else None
        self._check(self.L.az_slots_finish(self.h, other.h if other is not None else None, _ptr(d_game_of), _ptr(d_ply),
                                           _ptr(d_counters), _ptr(d_outcome), opt(d_stage), int(max_plies), opt(d_arena),
                                           int(arena_records), opt(d_game_off), opt(d_game_len), stream))

    def timing_mark(self, slot, end, stream=None):
        self._check(self.L.az_timing_mark(self.h, int(slot), int(bool(end)), stream))

    def timing_elapsed(self, slot):
        ms = _f()
        self._check(self.L.az_timing_elapsed(self.h, int(slot), C.byref(ms)))
        return ms.value

    def timing_begin(self, stream=None): self._check(self.L.az_timing_begin(self.h, stream))

    def timing_end(self, stream=None):
        ms = C.c_float()
        self._check(self.L.az_timing_end(self.h, stream, C.byref(ms)))
        return ms.value

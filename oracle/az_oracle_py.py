"""ctypes binding of oracle/libaz_oracle.so -- TEST INFRASTRUCTURE ONLY.

The oracle is the CPU checker the HIP engine is compared against.  Only
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module; the product package (alphazero_amd/) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libaz_oracle.so")

EVAL_UNIFORM, EVAL_HASH, EVAL_CALLBACK = 0, 1, 2

EVAL_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_byte), C.c_int,
                      C.POINTER(C.c_float), C.POINTER(C.c_float))


class Evaluator(C.Structure):
    _fields_ = [("kind", C.c_int), ("const_prior", C.c_float), ("const_value", C.c_float),
                ("hash_seed", C.c_ulonglong), ("fn", EVAL_FN), ("ctx", C.c_void_p)]


def build(force=False):
    """Compile libaz_oracle.so with gcc if missing or stale."""
    src = os.path.join(_HERE, "az_oracle.c")
    hdr = os.path.join(_HERE, "az_oracle.h")
    stale = (not os.path.exists(_LIB_PATH)
             or os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(src), os.path.getmtime(hdr)))
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-B", "libaz_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.azo_last_error.restype = C.c_char_p
        L.azo_create.restype = C.c_void_p
        L.azo_create.argtypes = [C.c_int, C.c_double, C.c_double]
        for name in ("azo_destroy", "azo_clear", "azo_set_alpha_default"):
            getattr(L, name).argtypes = [C.c_void_p]
            getattr(L, name).restype = None
        L.azo_set_eps.argtypes = [C.c_void_p, C.c_double]
        L.azo_get_eps.argtypes = [C.c_void_p]
        L.azo_get_eps.restype = C.c_double
        L.azo_set_cpuct.argtypes = [C.c_void_p, C.c_float]
        L.azo_set_cpuct.restype = None
        L.azo_get_cpuct.argtypes = [C.c_void_p]
        L.azo_get_cpuct.restype = C.c_float
        L.azo_set_threads.argtypes = [C.c_void_p, C.c_int]
        L.azo_set_threads.restype = None
        L.azo_set_schedule.argtypes = [C.c_void_p, C.c_int]
        L.azo_set_schedule.restype = None
        L.azo_get_threads.argtypes = [C.c_void_p]
        L.azo_set_alpha.argtypes = [C.c_void_p, C.c_double]
        L.azo_set_alpha.restype = None
        L.azo_get_alpha.argtypes = [C.c_void_p]
        L.azo_get_alpha.restype = C.c_double
        L.azo_set_cuda.argtypes = [C.c_void_p, C.c_int]
        L.azo_set_player.argtypes = [C.c_void_p, C.c_int]
        L.azo_set_player.restype = None
        L.azo_get_player.argtypes = [C.c_void_p]
        L.azo_get_move_cnt.argtypes = [C.c_void_p]
        L.azo_get_winner.argtypes = [C.c_void_p]
        L.azo_get_winner.restype = C.c_float
        L.azo_simulate.argtypes = [C.c_void_p, C.c_int, C.POINTER(Evaluator), C.c_void_p, C.c_int]
        L.azo_make_move.argtypes = [C.c_void_p, C.c_int]
        L.azo_make_movexy.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.azo_get_prob.argtypes = [C.c_void_p, C.c_void_p]
        L.azo_get_board.argtypes = [C.c_void_p, C.c_void_p]
        L.azo_get_board.restype = None
        L.azo_root_n.argtypes = [C.c_void_p]
        L.azo_root_n.restype = C.c_longlong
        for name in ("azo_root_counts", "azo_root_w", "azo_root_p", "azo_raw_board"):
            getattr(L, name).argtypes = [C.c_void_p, C.c_void_p]
            getattr(L, name).restype = None
        L.azo_node_at.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_longlong),
                                  C.c_void_p, C.c_void_p, C.c_void_p]
        L.azo_stats.argtypes = [C.c_void_p] + [C.POINTER(C.c_longlong)] * 4
        L.azo_stats.restype = None
        L.azo_live_nodes.argtypes = [C.c_void_p]
        L.azo_live_nodes.restype = C.c_longlong
        L.azo_rules_play.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.azo_rules_play_forced.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                            C.c_void_p, C.c_void_p]
        L.azo_select_arrays.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_longlong,
                                        C.c_void_p, C.c_double]
        L.azo_hash_eval.argtypes = [C.c_int, C.c_ulonglong, C.c_void_p, C.c_void_p, C.c_void_p]
        L.azo_hash_eval.restype = None
        L.azo_opening.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.azo_opening.restype = None
        L.azo_baseline_run.argtypes = [C.c_int] * 7 + [C.c_double, C.POINTER(Evaluator), C.POINTER(C.c_double),
                                                      C.POINTER(C.c_double)]
        L.azo_baseline_run.restype = C.c_longlong
        _lib = L
    return _lib


class OracleError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(msg)
        self.code = code


def _check(rc):
    if rc != 0:
        raise OracleError(rc, lib().azo_last_error().decode())


def uniform_evaluator(prior, value=0.0):
    return Evaluator(EVAL_UNIFORM, float(prior), float(value), 0, EVAL_FN(), None)


def hash_evaluator(seed):
    return Evaluator(EVAL_HASH, 0.0, 0.0, int(seed), EVAL_FN(), None)


def callback_evaluator(pyfn, shape, log_priors=False):
    """pyfn(raw_board int8[2,S,S], player) -> (value, prior[SIZE])."""
    size = shape * shape

    def tramp(ctx, board, player, value, prior):
        b = np.ctypeslib.as_array(board, shape=(2 * size,)).astype(np.int8).reshape(2, shape, shape)
        v, p = pyfn(b, player)
        value[0] = float(v)
        pa = np.ctypeslib.as_array(prior, shape=(size,))
        pa[:] = np.asarray(p, dtype=np.float32).reshape(-1)
        return 1 if log_priors else 0

    fn = EVAL_FN(tramp)
    ev = Evaluator(EVAL_CALLBACK, 0.0, 0.0, 0, fn, None)
    ev._keep = (fn, tramp)
    return ev


class OracleTree:
    """One reference-equivalent `cmcts.mcts` object on the CPU."""

    def __init__(self, shape, alpha=1.0, cpuct=1.0):
        self.L = lib()
        self.shape = shape
        self.size = shape * shape
        self.h = self.L.azo_create(shape, alpha, cpuct)
        if not self.h:
            raise ValueError("bad shape")

    def __del__(self):
        if getattr(self, "h", None):
            self.L.azo_destroy(self.h)
            self.h = None

    # setters / getters mirroring module.cpp:19-58
    def set_eps(self, e): _check(self.L.azo_set_eps(self.h, e))
    def get_eps(self): return self.L.azo_get_eps(self.h)
    def set_cpuct(self, c): self.L.azo_set_cpuct(self.h, c)
    def get_cpuct(self): return self.L.azo_get_cpuct(self.h)
    def set_threads(self, t): self.L.azo_set_threads(self.h, t)
    def set_schedule(self, s): self.L.azo_set_schedule(self.h, s)  # 0 = real threads, 1 = lock step (az_oracle.h)
    def set_alpha(self, a): self.L.azo_set_alpha(self.h, a)
    def get_alpha(self): return self.L.azo_get_alpha(self.h)
    def set_alpha_default(self): self.L.azo_set_alpha_default(self.h)
    def set_cuda(self, c): _check(self.L.azo_set_cuda(self.h, c))
    def set_player(self, p): self.L.azo_set_player(self.h, p)
    def clear(self): self.L.azo_clear(self.h)

    @property
    def player(self): return self.L.azo_get_player(self.h)
    @property
    def move_cnt(self): return self.L.azo_get_move_cnt(self.h)
    @property
    def winner(self): return self.L.azo_get_winner(self.h)

    def simulate(self, n, ev, noise=None):
        if noise is not None:
            noise = np.ascontiguousarray(noise, dtype=np.float64)
            assert noise.ndim == 2 and noise.shape[1] == self.size
            _check(self.L.azo_simulate(self.h, n, C.byref(ev), noise.ctypes.data, noise.shape[0]))
        else:
            _check(self.L.azo_simulate(self.h, n, C.byref(ev), None, 0))

    def make_move(self, a): _check(self.L.azo_make_move(self.h, int(a)))
    def make_movexy(self, x, y): _check(self.L.azo_make_movexy(self.h, x, y))

    def get_prob(self):
        out = np.empty(self.size, np.float32)
        _check(self.L.azo_get_prob(self.h, out.ctypes.data))
        return out

    def get_board(self):
        out = np.empty((2, self.shape, self.shape), np.float32)
        self.L.azo_get_board(self.h, out.ctypes.data)
        return out

    def raw_board(self):
        out = np.empty((2, self.shape, self.shape), np.int8)
        self.L.azo_raw_board(self.h, out.ctypes.data)
        return out

    def root_n(self): return self.L.azo_root_n(self.h)

    def root_counts(self):
        out = np.empty(self.size, np.int64)
        self.L.azo_root_counts(self.h, out.ctypes.data)
        return out

    def root_w(self):
        out = np.empty(self.size, np.float64)
        self.L.azo_root_w(self.h, out.ctypes.data)
        return out

    def root_p(self):
        out = np.empty(self.size, np.float64)
        self.L.azo_root_p(self.h, out.ctypes.data)
        return out

    def node_at(self, path):
        path = np.asarray(path, np.int32)
        n = C.c_longlong()
        cnt = np.empty(self.size, np.int64)
        w = np.empty(self.size, np.float64)
        p = np.empty(self.size, np.float64)
        rc = self.L.azo_node_at(self.h, path.ctypes.data, len(path), C.byref(n),
                                cnt.ctypes.data, w.ctypes.data, p.ctypes.data)
        if rc:
            return None
        return n.value, cnt, w, p

    def stats(self):
        v = [C.c_longlong() for _ in range(4)]
        self.L.azo_stats(self.h, *[C.byref(x) for x in v])
        return dict(sims=v[0].value, expansions=v[1].value, terminals=v[2].value,
                    select_calls=v[3].value)


def rules_play(shape, moves, force=None):
    """Plays moves from the empty board; returns (n_ok, player[], move_cnt[], winner[])."""
    moves = np.asarray(moves, np.int32)
    n = len(moves)
    pl = np.zeros(n, np.int32)
    mc = np.zeros(n, np.int32)
    wi = np.zeros(n, np.float32)
    if force is None:
        k = lib().azo_rules_play(shape, moves.ctypes.data, n, pl.ctypes.data, mc.ctypes.data,
                                 wi.ctypes.data)
    else:
        force = np.asarray(force, np.int32)
        k = lib().azo_rules_play_forced(shape, moves.ctypes.data, force.ctypes.data, n,
                                        pl.ctypes.data, mc.ctypes.data, wi.ctypes.data)
    return k, pl[:k], mc[:k], wi[:k]


def select_arrays(n, w, p, node_n, board, cpuct):
    n = np.ascontiguousarray(n, np.int64)
    w = np.ascontiguousarray(w, np.float64)
    p = np.ascontiguousarray(p, np.float64)
    board = np.ascontiguousarray(board, np.int8)
    return lib().azo_select_arrays(len(n), n.ctypes.data, w.ctypes.data, p.ctypes.data,
                                   int(node_n), board.ctypes.data, float(cpuct))


def hash_eval(shape, seed, raw_board):
    raw_board = np.ascontiguousarray(raw_board, np.int8)
    v = C.c_float()
    p = np.empty(shape * shape, np.float32)
    lib().azo_hash_eval(shape, int(seed), raw_board.ctypes.data, C.byref(v), p.ctypes.data)
    return v.value, p


def opening(size, g):
    a0, a1 = C.c_int(), C.c_int()
    lib().azo_opening(size, g, C.byref(a0), C.byref(a1))
    return a0.value, a1.value


def baseline_run(shape, games, workers, tree_threads, skip, plies, sims, cpuct, ev):
    """-> (timed searches, searches per second summed over the workers, longest worker's timed seconds)"""
    rate, secs = C.c_double(), C.c_double()
    total = lib().azo_baseline_run(shape, games, workers, tree_threads, skip, plies, sims, cpuct, C.byref(ev),
                                   C.byref(rate), C.byref(secs))
    return total, rate.value, secs.value

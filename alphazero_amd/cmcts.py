"""Drop-in for the reference's `cmcts` extension module (mcts/src/module.cpp:12-68).

    import alphazero_amd.cmcts as cmcts
    m = cmcts.mcts(cpuct=4.)         # module.cpp:18
    m.set_alpha_default(); m.set_params(path_or_evaluator); m.eps = 0.25
    m.simulate(800); pi = m.get_prob(); board = m.get_board(); m.make_move(pi.argmax())

Same names, argument meaning and error behaviour (RuntimeError with the
reference's messages).  The search runs on the GPU: one `mcts` object is a
one-tree engine (alphazero_amd.engine.Engine); the batched drivers in
alphazero_amd.selfplay run thousands of such trees in one engine.

Differences that cannot be avoided:
  * board size is a run-time value here (`shape=` keyword or AZ_SHAPE env,
    default 13 like general_config.py:111); the reference bakes it in at
    compile time (mcts/CMakeLists.txt:36).
  * `params` accepts the reference's TorchScript file path (loaded with
    torch.jit.load onto the GPU) or an evaluator object (see
    alphazero_amd.evaluator); the net is evaluated once per search step.
  * Dirichlet noise comes from a table generated on the GPU by the engine's
    counter-based sampler (az_fill_noise_dirichlet, seed `noise_seed`, default
    0 -- the reference also re-seeds its generator with the default seed at
    every simulate() call, cmcts.cpp:132-133), not from GSL.
"""
import os

import numpy as np

from . import engine as _E

MAJOR, MINOR = 0, 6  # mcts/setup.py version numbers printed by test()
DEFAULT_SHAPE = int(os.environ.get("AZ_SHAPE", "13"))
DEFAULT_NODES = int(os.environ.get("AZ_NODES_PER_TREE", "8192"))


def test(i):
    """module.cpp:6-10: prints MAJOR.MINOR and returns i+1."""
    print(f"{MAJOR}.{MINOR}")
    return i + 1


def version():
    """module.cpp:15"""
    return _E.load_library().az_version().decode()


def build_timestamp():
    """module.cpp:16"""
    return _E.load_library().az_build_timestamp().decode()


def _raise(exc):
    raise RuntimeError(str(exc)) from None


class mcts:
    """One search tree + game state: the reference's `Cmcts` (mcts/src/cmcts.h:25-81)."""

    def __init__(self, alpha=1, cpuct=1, shape=None, nodes_per_tree=None, device=0):
        self._shape = int(shape or DEFAULT_SHAPE)
        self._size = self._shape * self._shape
        try:
            self._eng = _E.Engine(1, self._shape, nodes_per_tree or DEFAULT_NODES,
                                  float(alpha), float(cpuct), device)
        except _E.EngineError as e:
            _raise(e)
        self._device = device
        self._params = ""
        self._moves = []  # State::moves (state.cpp:91), shown by __repr__
        self._evaluator = None
        self.noise_seed = 0
        self._noise_key = None
        self._bufs = None

    # ------------------------------------------------------------ properties
    # module.cpp:19-46,57-58
    @property
    def params(self):
        return self._params

    @params.setter
    def params(self, file_name):
        self.set_params(file_name)

    def set_params(self, file_name):
        """cmcts.cpp:67.  A path to a TorchScript file, or an evaluator object
        with `__call__(boards[B,2,S,S] float32 cuda) -> (value[B...], logp[B,SIZE])`."""
        if isinstance(file_name, str):
            self._params = file_name
            self._evaluator = None  # loaded at the next simulate(), like the reference's worker
        else:
            self._params = getattr(file_name, "name", repr(file_name))
            self._evaluator = file_name

    def clear_params(self):
        """cmcts.cpp:39-44"""
        self._params = ""
        self._evaluator = None

    @property
    def eps(self):
        return self._eng.get_eps()

    @eps.setter
    def eps(self, v):
        try:
            self._eng.set_eps(float(v))
        except _E.EngineError as e:
            _raise(e)

    @property
    def cuda(self):
        return self._eng.get_cuda()

    @cuda.setter
    def cuda(self, v):
        try:
            self._eng.set_cuda(int(v))  # always raises "Invalid value", as cmcts.cpp:70 does
        except _E.EngineError as e:
            _raise(e)

    @property
    def alpha(self):
        return self._eng.get_alpha()

    @alpha.setter
    def alpha(self, v):
        self.set_alpha(v)

    def set_alpha(self, v):
        self._eng.set_alpha(float(v))

    def set_alpha_default(self):
        self._eng.set_alpha_default()

    @property
    def threads(self):
        return self._eng.get_threads()

    @threads.setter
    def threads(self, v):
        self.set_threads(v)

    def set_threads(self, v):
        self._eng.set_threads(int(v))

    @property
    def schedule(self):
        """Not in the reference: how the workers of threads > 1 interleave (engine.WORKERS_CONCURRENT, the
        default: in lock step, their leaves evaluated together; engine.WORKERS_SEQUENTIAL: one after another)."""
        return self._eng.get_schedule()

    @schedule.setter
    def schedule(self, v):
        try:
            self._eng.set_schedule(int(v))
        except _E.EngineError as e:
            _raise(e)

    @property
    def player(self):
        return self._eng.get_state(0)[0]

    @player.setter
    def player(self, v):
        try:
            self._eng.set_player(0, int(v))
        except _E.EngineError as e:
            _raise(e)

    @property
    def cpuct(self):
        return self._eng.get_cpuct()

    @cpuct.setter
    def cpuct(self, v):
        self.set_cpuct(v)

    def set_cpuct(self, v):
        self._eng.set_cpuct(float(v))

    @property
    def winner(self):
        return self._eng.get_state(0)[2]

    @property
    def move_cnt(self):
        return self._eng.get_state(0)[1]

    # --------------------------------------------------------------- methods
    def clear(self):
        """cmcts.cpp:30-37: new root, empty board; parameters are kept."""
        self._eng.clear(0)
        self._moves = []

    def _ensure_evaluator(self):
        if self._evaluator is None:
            if not self._params:
                raise RuntimeError("Predictor missing!")  # cmcts.cpp:89-91
            from .evaluator import load_evaluator
            self._evaluator = load_evaluator(self._params, self._shape, self._device)
        return self._evaluator

    def _ensure_noise(self, rows):
        if self._eng.get_eps() == 0.0:
            return
        key = (self._eng.get_alpha(), self.noise_seed)
        if self._noise_key is None or self._noise_key[:2] != key or self._noise_key[2] < rows:
            rows = max(rows, 1024)
            self._eng.fill_noise_dirichlet(rows, self.noise_seed)
            self._noise_key = key + (rows,)

    def set_noise_table(self, table):
        """Test hook: fixed noise rows (row k = k-th expansion of a simulate() call)."""
        self._eng.set_noise_table(table)
        self._noise_key = ("fixed", None, 1 << 30) if table is not None else None

    def simulate(self, n):
        """cmcts.cpp:83-125: n searches (thread fan-out rounding included)."""
        n = int(n)
        if n < 1:
            raise RuntimeError(f"Invalid input {n} must be at least 1!")
        ev = self._ensure_evaluator()
        import torch

        th_num, per = 1, n
        threads = self._eng.get_threads()
        if threads > 1:  # cmcts.cpp:98-107
            th_num = threads
            if n < th_num:
                th_num, per = n, 1
            elif n % th_num == 0:
                per = n // th_num
            else:
                per = n // th_num + 1
        if self._noise_key is None or self._noise_key[0] != "fixed":
            self._ensure_noise(per)
        dev = torch.device("cuda", self._device)
        if self._bufs is None:
            self._bufs = (torch.zeros((1, 2, self._shape, self._shape), dtype=torch.float32, device=dev),
                          torch.zeros(1, dtype=torch.int32, device=dev))
        boards, kinds = self._bufs
        stream = torch.cuda.current_stream(dev).cuda_stream
        if getattr(ev, "fused_layers", None) is not None:
            try:  # the library's own network kernel: one call queues every step of the search
                self._eng.simulate_net(n, ev.params.data_ptr(), ev.fused_layers, stream)
            except _E.EngineError as e:
                _raise(e)
            return
        if th_num > 1 and self._eng.get_schedule() == _E.WORKERS_CONCURRENT:
            wb = torch.zeros((th_num, 2, self._shape, self._shape), dtype=torch.float32, device=dev)
            wk = torch.zeros(th_num, dtype=torch.int32, device=dev)
            try:  # the workers in lock step: their th_num leaves in one evaluator call
                with torch.no_grad():
                    self._eng.begin_simulate(stream)
                    for _ in range(per):
                        self._eng.select_workers(th_num, wb.data_ptr(), wk.data_ptr(), stream)
                        value, logp = ev(wb)
                        self._eng.expand_backup_workers(th_num, value.data_ptr(), logp.data_ptr(), ev.log_priors, stream)
                self._eng.sync_check(stream)
            except _E.EngineError as e:
                _raise(e)
            return
        try:
            with torch.no_grad():
                for _ in range(th_num):
                    self._eng.begin_simulate(stream)
                    for _ in range(per):
                        self._eng.select(boards.data_ptr(), kinds.data_ptr(), stream)
                        value, logp = ev(boards)
                        self._eng.expand_backup(value.data_ptr(), logp.data_ptr(), ev.log_priors, stream)
            self._eng.sync_check(stream)
        except _E.EngineError as e:
            _raise(e)

    def make_move(self, action):
        """cmcts.cpp:284-294"""
        try:
            self._eng.make_move(np.array([int(action)], np.int32))
        except _E.EngineError as e:
            if e.code == _E.ERR_INVALID_MOVE:
                raise RuntimeError(f"Invalid move {int(action)}\n {self!r}!") from None  # state.cpp:86
            _raise(e)
        self._moves.append(int(action))  # state.cpp:91

    def make_movexy(self, x, y):
        """cmcts.cpp:296-309"""
        try:
            self._eng.make_movexy(0, int(x), int(y))
        except _E.EngineError as e:
            _raise(e)
        self._moves.append(int(y) * self._shape + int(x))

    def get_board(self):
        """state.cpp:39-48: float32 [2,S,S], plane 0 = side to move."""
        return self._eng.get_board(0)

    def get_prob(self):
        """cmcts.cpp:317-341"""
        try:
            return self._eng.get_prob(0)
        except _E.EngineError as e:
            if e.code == _E.ERR_NOT_DIST:
                print(self._node_repr([]))  # the reference prints the node before throwing
            _raise(e)

    # ------------------------------------------------------------- debugging
    def _node_repr(self, path):
        node = self._eng.get_node(0, list(path))
        if node is None:
            return "child is nullptr"
        node_n, counts, w, p = node
        S = self._shape
        lines = [f"Visits: {node_n}", "", "Counts:"]
        for name, arr, fmt in (("Counts", counts, "{:3d}"), ("Probs", p, "{:.3g}"), ("Total child values", w, "{:.3g}")):
            if name != "Counts":
                lines += ["", f"{name}:"]
            for r in range(S):
                lines.append(" ".join(fmt.format(x) for x in arr[r * S:(r + 1) * S]) + " ")
            lines += ["", f"{name} total: {arr.sum()}"]
        return "\n".join(lines)

    def print_node(self, v):
        """cmcts.cpp:404-417"""
        print(self._node_repr(v))

    def print_u(self, v):
        """cmcts.cpp:419-432: PUCT value of every legal action at the node reached by v."""
        node = self._eng.get_node(0, list(v))
        if node is None:
            print("child is nullptr")
            return
        node_n, counts, w, p = node
        if node_n == -1:
            raise RuntimeError("Node has not been visited yet. Can't select next_node")
        raw = self._eng.get_raw_board(0).reshape(2, -1)
        occupied = (raw[0] == 1) | (raw[1] == 1)
        c = float(self._eng.get_cpuct())
        root = np.sqrt(node_n + 1e-8)
        with np.errstate(divide="ignore", invalid="ignore"):
            u = np.where(counts != 0, w / counts + c * p * root / (counts + 1), c * p * root)
        S = self._shape
        lines = ["", "Total child values:"]
        for r in range(S):
            lines.append(" ".join(" ___ " if occupied[r * S + k] else f"{u[r * S + k]:.3f}" for k in range(S)))
        u = np.where(occupied, -np.inf, u)
        lines += ["", f"best action: {int(u.argmax())}"]
        print("\n".join(lines))

    def __repr__(self):
        """cmcts.cpp:386-402 + state.cpp:112-150"""
        player, move_cnt, winner = self._eng.get_state(0)
        raw = self._eng.get_raw_board(0)
        S = self._shape
        s = [f"Alpha: {self.alpha:f}", f"CPUCT: {float(self.cpuct):f}"]
        if not self._params:
            s.append("Heuristic: yes")
        else:
            s += ["Heuristic: no", f"Parameters from: {self._params}"]
        s.append("First player: x")
        s.append("Player: x" if player == 0 else "Player: o")
        s.append({0.0: "Winner: x", 1.0: "Winner: o", -1.0: "Winner: _"}.get(winner, "Winner: draw"))
        s.append(f"Move count: {move_cnt}")
        s.append("Moves: " + "".join(f"{m}, " for m in self._moves))
        s.append(f"Board size: {self._size}")
        s.append("Board:")
        for r in range(S):
            s.append(" ".join("x" if raw[0, r, c] else ("o" if raw[1, r, c] else "_") for c in range(S)) + " ")
        return "\n".join(s) + "\n"

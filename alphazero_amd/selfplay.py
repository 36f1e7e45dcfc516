"""Batched self-play and arena drivers.

The reference plays ONE game at a time with two `cmcts.mcts` objects
(`self_play_game`, src/self_play.py:134-202; `eval_game`, src/eval.py:187-209).
Here the same per-move protocol runs for thousands of games in lock step: each
side owns one engine with one tree per game, a search step evaluates every
tree's leaf in a single batched network call, and games are independent, so
they shard over GPUs with no collective inside the search (gather_samples /
reduce_counters are the only exchanges, over RCCL when launched with one
process per GPU).
"""
import time

import numpy as np

from . import engine as _E


def sample_dtype(shape):
    """Record layout of the reference's training samples (src/self_play.py:153)."""
    return np.dtype([("board", "f4", (2, shape, shape)), ("pi", "f4", (shape * shape,)), ("r", "f4")])


def shard_range(n_items, rank, world):
    """Contiguous block of game indices owned by `rank` (independent games shard naturally)."""
    base, extra = divmod(n_items, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


class TreeBatch:
    """`n_trees` reference `cmcts.mcts` objects driven together by one evaluator."""

    def __init__(self, n_trees, shape, evaluator, cpuct=1.0, eps=0.25, alpha=None, threads=0,
                 nodes_per_tree=4096, device=0, noise_seed=0):
        import torch

        self.torch = torch
        self.eng = _E.Engine(n_trees, shape, nodes_per_tree, 1.0 if alpha is None else alpha, cpuct, device)
        if alpha is None:
            self.eng.set_alpha_default()  # what the callers do (src/self_play.py:70, src/eval.py:83)
        self.eng.set_eps(eps)
        self.eng.set_threads(threads)
        self.n_trees, self.shape, self.size = n_trees, shape, shape * shape
        self.evaluator = evaluator
        self.noise_seed = noise_seed
        self._noise_rows = 0
        self.dev = torch.device("cuda", device)
        self.boards = torch.zeros((n_trees, 2, shape, shape), dtype=torch.float32, device=self.dev)
        self.kinds = torch.zeros(n_trees, dtype=torch.int32, device=self.dev)
        self._wboards = self._wkinds = None  # leaf staging of concurrent workers, made on first use

    def close(self):
        self.eng.close()

    def _ensure_noise(self, rows):
        if self.eng.get_eps() == 0.0 or rows <= self._noise_rows:
            return
        rows = max(rows, 1024)
        self.eng.fill_noise_dirichlet(rows, self.noise_seed)  # on the GPU, replaces gsl_ran_dirichlet (node.cpp:38)
        self._noise_rows = rows

    def set_noise_table(self, table):
        self.eng.set_noise_table(table)
        self._noise_rows = 1 << 30 if table is not None else 0

    def simulate(self, n, active=None):
        """Cmcts::simulate(n) (cmcts.cpp:83-125) on the active trees; one batched
        evaluator call per search step."""
        if n < 1:
            raise RuntimeError(f"Invalid input {n} must be at least 1!")
        self.eng.set_active(None if active is None else np.asarray(active, np.uint8))
        self.queue_simulate(n)
        try:
            self.eng.sync_check(self.torch.cuda.current_stream(self.dev).cuda_stream)
        except _E.EngineError as e:
            raise RuntimeError(str(e)) from None

    def queue_simulate(self, n):
        """simulate(n) on the trees whose active flag is set, queued on the current stream:
        no host wait, errors surface at the next sync_check."""
        eng, torch = self.eng, self.torch
        th_num, per = 1, n
        threads = eng.get_threads()
        if threads > 1:  # cmcts.cpp:98-107
            th_num = threads
            if n < th_num:
                th_num, per = n, 1
            elif n % th_num == 0:
                per = n // th_num
            else:
                per = n // th_num + 1
        self._ensure_noise(per)
        stream = torch.cuda.current_stream(self.dev).cuda_stream
        if getattr(self.evaluator, "fused_layers", None) is not None:
            # the library's own network kernel: one call queues every step of the search
            try:
                eng.simulate_net(n, self.evaluator.params.data_ptr(), self.evaluator.fused_layers, stream, sync=False)
            except _E.EngineError as e:
                raise RuntimeError(str(e)) from None
            return
        if th_num > 1 and eng.get_schedule() == _E.WORKERS_CONCURRENT:
            # the workers in lock step (az_set_schedule): th_num leaves per tree in every evaluator call
            if self._wboards is None or self._wboards.shape[0] != self.n_trees * th_num:
                self._wboards = torch.zeros((self.n_trees * th_num, 2, self.shape, self.shape), dtype=torch.float32, device=self.dev)
                self._wkinds = torch.zeros(self.n_trees * th_num, dtype=torch.int32, device=self.dev)
            with torch.no_grad():
                eng.begin_simulate(stream)
                for _ in range(per):
                    eng.select_workers(th_num, self._wboards.data_ptr(), self._wkinds.data_ptr(), stream)
                    if getattr(self.evaluator, "takes_leaf_kinds", False):
                        value, prior = self.evaluator(self._wboards, self._wkinds)
                    else:
                        value, prior = self.evaluator(self._wboards)
                    eng.expand_backup_workers(th_num, value.data_ptr(), prior.data_ptr(), self.evaluator.log_priors, stream)
            return
        with torch.no_grad():
            for _ in range(th_num):
                eng.begin_simulate(stream)
                for _ in range(per):
                    eng.select(self.boards.data_ptr(), self.kinds.data_ptr(), stream)
                    if getattr(self.evaluator, "takes_leaf_kinds", False):
                        value, prior = self.evaluator(self.boards, self.kinds)  # idle / terminal leaves skipped
                    else:
                        value, prior = self.evaluator(self.boards)
                    eng.expand_backup(value.data_ptr(), prior.data_ptr(), self.evaluator.log_priors, stream)

    def probs(self, trees=None):
        """Cmcts::get_prob (cmcts.cpp:317-341) for many trees: float32 [n, SIZE];
        raises "Not a distribution" like the reference."""
        counts, root_n = self.eng.get_counts()
        if trees is not None:
            counts, root_n = counts[trees], root_n[trees]
        with np.errstate(divide="ignore", invalid="ignore"):
            pi = (counts.astype(np.float64) / root_n.astype(np.float64)[:, None]).astype(np.float32)
        bad = np.round(pi.astype(np.float64).sum(axis=1) * 1000.0) != 1000
        if bad.any():
            raise RuntimeError("Not a distribution")
        return pi

    def boards_to_move(self, trees=None):
        """State::get_board (state.cpp:39-48) for many trees: plane 0 = side to move."""
        raw = self.eng.get_raw_boards().astype(np.float32)
        player, _, _ = self.eng.get_states()
        swap = player == 1
        raw[swap] = raw[swap][:, ::-1]
        return raw if trees is None else raw[trees]

    def make_move(self, actions):
        try:
            self.eng.make_move(np.asarray(actions, np.int32))
        except _E.EngineError as e:
            raise RuntimeError(str(e)) from None

    def winners(self):
        return self.eng.get_states()[2]


def choice_from_uniform(pi, u):
    """numpy.random.choice(len(pi), p=pi) (src/self_play.py:164) for a given
    uniform number u in [0,1): cdf = cumsum(p) in fp64, cdf /= cdf[-1],
    searchsorted(cdf, u, side='right').  The random stream is the one input the
    GPU path cannot share with numpy, so it is made explicit; numpy's check
    that p sums to 1 within sqrt(eps) is not replicated (pi is renormalised by
    the division, a float32 pi that misses the tolerance does not abort the game)."""
    cdf = np.cumsum(np.asarray(pi, np.float64))
    cdf /= cdf[-1]
    return int(min(np.searchsorted(cdf, u, side="right"), len(cdf) - 1))


_M64 = (1 << 64) - 1


def _mix64(x):
    """splitmix64 finaliser on uint64 arrays (wrap-around arithmetic)."""
    x = np.asarray(x, np.uint64)
    with np.errstate(over="ignore"):
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xbf58476d1ce4e5b9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94d049bb133111eb)
        return x ^ (x >> np.uint64(31))


class KeyedUniform:
    """The uniform number behind a sampled move as a function of (seed, game, half-move):
    the host mirror of az_slots_choose's generator (include/azmcts.h), so a game's moves
    depend on neither the tree slot nor the batch it is played in.  self_play_stream takes
    one as its `rng`; self_play_resident computes the same numbers on the GPU."""

    def __init__(self, seed):
        self.seed = _mix64(np.uint64(int(seed) & _M64))  # hashed: neighbouring seeds give unrelated streams

    def for_games(self, games, plies):
        key = (np.asarray(games, np.uint64) << np.uint64(32)) | np.asarray(plies, np.uint64)
        with np.errstate(over="ignore"):
            x = _mix64(_mix64(self.seed ^ key) + np.uint64(0x9e3779b97f4a7c15))
        return ((x >> np.uint64(11)).astype(np.float64) + 0.5) * (1.0 / 9007199254740992.0)


def _choose_moves(pi, greedy, rng):
    """src/self_play.py:159-164: argmax + one-hot pi when greedy, else sample from pi
    (one uniform number per game, drawn in game order)."""
    moves = np.empty(pi.shape[0], np.int64)
    u = None if greedy else rng.random(pi.shape[0])
    for j in range(pi.shape[0]):
        if greedy:
            m = int(pi[j].argmax())
            pi[j, :] = 0
            pi[j, m] = 1
        else:
            m = choice_from_uniform(pi[j], u[j])
        moves[j] = m
    return moves


def self_play_games(batch0, batch1, sims, tau, init_moves=(), rng=None):
    """`self_play_game` (src/self_play.py:134-202) for every game of the two batches.

    batch0 / batch1: TreeBatch of equal size, tree g of each = mcts0 / mcts1 of game g.
    Returns a list with one structured array per game (dtype sample_dtype), first
    player's records first, labelled r = 1 / -1, or 1e-3 everywhere for a draw.
    """
    rng = np.random.default_rng() if rng is None else rng
    G, S, size = batch0.n_trees, batch0.shape, batch0.size
    assert batch1.n_trees == G and batch1.shape == S
    dt = sample_dtype(S)
    for m in init_moves:  # tools.make_init_moves (src/tools.py:41-44)
        batch0.make_move(np.full(G, m, np.int32))
        batch1.make_move(np.full(G, m, np.int32))
    data = [([], []) for _ in range(G)]
    alive = np.ones(G, bool)
    for i in range(size):
        for side, searcher in ((0, batch0), (1, batch1)):
            if not alive.any():
                break
            idx = np.flatnonzero(alive)
            searcher.simulate(sims, alive)
            pi = searcher.probs(idx)
            boards = searcher.boards_to_move(idx)
            moves = _choose_moves(pi, i > tau, rng)
            for j, g in enumerate(idx):
                data[g][side].append((boards[j], pi[j].copy(), -1.0))
            act = np.full(G, -1, np.int32)
            act[idx] = moves
            batch0.make_move(act)
            batch1.make_move(act)
            alive &= batch0.winners() == -1.0  # src/self_play.py:169,185
        if not alive.any():
            break
    winner = batch0.winners()
    out = []
    for g in range(G):
        rec = np.array([(b, p, r) for b, p, r in data[g][0] + data[g][1]], dtype=dt)
        n0 = len(data[g][0])
        if winner[g] == 0.0:
            rec["r"][:n0] = 1.0
        elif winner[g] == 1.0:
            rec["r"][n0:] = 1.0
        else:
            rec["r"] = 1e-3  # draw (or the move limit was reached)
        out.append(rec)
    return out


class GameRecords(list):
    """Per-game record arrays (what the batched self-play drivers return) that are
    slices of one array, or of a few arrays (`pieces`, each holding whole games):
    `flat` is all records as one array, so that writing or gathering them needs no
    per-game concatenation."""

    def __init__(self, flat, bounds):
        super().__init__(flat[bounds[g]:bounds[g + 1]] for g in range(len(bounds) - 1))
        self._flat = flat
        self.pieces = [flat]

    @classmethod
    def from_pieces(cls, pieces, game_off, game_len):
        """pieces: record arrays that follow one another in one record numbering;
        game g = records [game_off[g], game_off[g] + game_len[g]), never split over two pieces."""
        self = cls.__new__(cls)
        starts = np.cumsum([0] + [len(p) for p in pieces])
        which = np.searchsorted(starts, game_off, side="right") - 1
        list.__init__(self, (pieces[w][o - starts[w]:o - starts[w] + n] for w, o, n in zip(which, game_off, game_len)))
        self._flat = pieces[0] if len(pieces) == 1 else None
        self.pieces = list(pieces)
        return self

    @property
    def flat(self):
        if self._flat is None:
            self._flat = np.concatenate(self.pieces)
        return self._flat


def all_records(games):
    """All records of a list of per-game arrays as one array (no copy when the list
    came from a batched driver)."""
    if isinstance(games, GameRecords):
        return games.flat
    if isinstance(games, (list, tuple)):
        return np.concatenate(games) if len(games) else None
    return games


def _assemble_records(torch, dev, slabs, outcome, shape, pinned_bytes=256 << 20):
    """Training records of many games from per-half-move device slabs, assembled ON THE GPU.
    slabs: list of (step, side, pi [G,SIZE] device, board [G,2,S,S] device, played bool[G],
    game_of int64[G]); outcome[game] = the game's winner (0 / 1, anything else = draw).
    Returns (records, bounds): one structured array (sample_dtype) ordered by game, first
    player's records first, ply order inside a side, and the per-game slice bounds.  The
    ordering keys are small host arrays.  The list is CONSUMED: every slab is dropped as soon as
    its records have been scattered to their final rows, so the peak device use is the slabs plus
    one copy of the records (no stacked copy of all slabs), and the records reach the host
    through a page-locked staging buffer of bounded size."""
    dt = sample_dtype(shape)
    size = shape * shape
    n_games = len(outcome)
    if not slabs:
        return np.zeros(0, dt), np.zeros(n_games + 1, np.int64)
    played = np.stack([sl[4] for sl in slabs])                       # [H, G]
    games = np.stack([sl[5] for sl in slabs])                        # [H, G]
    h_idx, g_idx = np.nonzero(played)                                # row-major: slab by slab
    game = games[h_idx, g_idx]
    side = np.array([sl[1] for sl in slabs], np.int64)[h_idx]
    step = np.array([sl[0] for sl in slabs], np.int64)[h_idx]
    order = np.lexsort((step, side, game))
    n = order.size
    dest = np.empty(n, np.int64)                                     # final row of every record, in slab order
    dest[order] = np.arange(n)
    won = outcome[game[order]]
    decided = (won == 0.0) | (won == 1.0)
    r = np.where(decided, np.where(won == side[order], 1.0, -1.0), 1e-3).astype(np.float32)  # src/self_play.py:188-200
    out = torch.empty((n, 3 * size + 1), dtype=torch.float32, device=dev)
    per_slab = played.sum(axis=1)
    at = 0
    for h in range(len(slabs)):
        k = int(per_slab[h])
        _, _, pi, board, _, _ = slabs[h]
        slabs[h] = None                                              # this slab's memory goes back once scattered
        if k:
            src = torch.from_numpy(g_idx[at:at + k]).to(dev)
            dst = torch.from_numpy(dest[at:at + k]).to(dev)
            out[:, :2 * size].index_copy_(0, dst, board.reshape(board.shape[0], 2 * size).index_select(0, src))
            out[:, 2 * size:3 * size].index_copy_(0, dst, pi.index_select(0, src))
        del pi, board
        at += k
    slabs.clear()
    out[:, 3 * size].copy_(torch.from_numpy(r))
    rec = np.empty(n, dt)
    flat = rec.view(np.float32).reshape(n, 3 * size + 1)
    rows = max(1, int(pinned_bytes) // ((3 * size + 1) * 4))
    stage = torch.empty((min(rows, max(n, 1)), 3 * size + 1), dtype=torch.float32, pin_memory=True)
    for a in range(0, n, rows):
        b = min(n, a + rows)
        stage[:b - a].copy_(out[a:b])
        flat[a:b] = stage[:b - a].numpy()
    bounds = np.searchsorted(game[order], np.arange(n_games + 1))
    return rec, bounds


def self_play_games_device(batch0, batch1, sims, tau, init_moves=(), rng=None):
    """self_play_games with the per-move work kept on the GPU: visit-count
    distribution, move choice, the record's board and pi (az_choose_moves) and
    the move itself never visit the host; per half-move the host uploads one
    uniform number per game (while sampling) and reads the winners.  The record
    slabs come back once, at the end.  Same results as self_play_games for the
    same `rng` (it is consumed identically: one number per live game, in game
    order, only while i <= tau)."""
    torch = batch0.torch
    rng = np.random.default_rng() if rng is None else rng
    G, S, size = batch0.n_trees, batch0.shape, batch0.size
    assert batch1.n_trees == G and batch1.shape == S
    dev = batch0.dev
    stream = torch.cuda.current_stream(dev).cuda_stream
    for m in init_moves:
        batch0.make_move(np.full(G, m, np.int32))
        batch1.make_move(np.full(G, m, np.int32))
    d_act = torch.empty(G, dtype=torch.int32, device=dev)
    d_u = torch.zeros(G, dtype=torch.float64, device=dev)
    slabs = ([], [])          # per side: list of (pi [G,SIZE], board [G,2,S,S]) device tensors, one per ply
    played = ([], [])         # per side: list of bool[G] = game was alive at that ply
    alive = np.ones(G, bool)
    for i in range(size):
        for side, searcher in ((0, batch0), (1, batch1)):
            if not alive.any():
                break
            searcher.simulate(sims, alive)  # leaves exactly the live trees active
            greedy = i > tau
            if not greedy:
                u = np.zeros(G)
                u[alive] = rng.random(int(alive.sum()))
                d_u.copy_(torch.from_numpy(u))
            pi = torch.empty((G, size), dtype=torch.float32, device=dev)
            board = torch.empty((G, 2, S, S), dtype=torch.float32, device=dev)
            searcher.eng.choose_moves(None if greedy else d_u.data_ptr(), greedy, d_act.data_ptr(), pi.data_ptr(),
                                      board.data_ptr(), stream)
            slabs[side].append((pi, board))
            played[side].append(alive.copy())
            try:
                batch0.eng.make_move_device(d_act.data_ptr(), stream)
                batch1.eng.make_move_device(d_act.data_ptr(), stream)
                searcher.eng.sync_check(stream)
                (batch1 if searcher is batch0 else batch0).eng.sync_check(stream)
            except _E.EngineError as e:
                raise RuntimeError(str(e)) from None
            alive &= batch0.winners() == -1.0
        if not alive.any():
            break
    winner = batch0.winners()
    slab_list = []
    for side in (0, 1):
        for step, ((pi, board), alive_then) in enumerate(zip(slabs[side], played[side])):
            slab_list.append((step, side, pi, board, alive_then, np.arange(G, dtype=np.int64)))
    slabs[0].clear()  # (the assembly drops each slab as it goes)
    slabs[1].clear()
    pi = board = None
    rec, bounds = _assemble_records(torch, dev, slab_list, np.asarray(winner, np.float32), S)
    return GameRecords(rec, bounds)


def self_play_stream(batch0, batch1, sims, tau, n_games, init_moves=(), rng=None, flush_plies=None,
                     device_record_bytes=None):
    """`n_games` games of `self_play_game` (src/self_play.py:134-202) on the G tree
    slots of the two batches, a finished game's slot being cleared and handed to
    the next game instead of idling until the longest game of the batch is over
    (game lengths spread over a factor of four, so a fixed batch leaves about
    half of the evaluator's positions empty).

    Every game follows the reference protocol on its own clock: pair index i of
    the game decides sampling (i <= tau) or greedy, the first player's records
    come first, labels as in src/self_play.py:194-200.  Slots stay in phase (the
    first player of every live game moves on even half-steps), so a game that
    ends on an even half-step leaves its slot empty for one half-step.  `rng`
    supplies one uniform number per sampling game and half-step, in slot order.
    The per-half-step pi / board slabs stay on the GPU and are turned into records there at
    the end (`_assemble_records`); `flush_plies` (or slabs beyond `device_record_bytes`, by
    default a quarter of the device memory that is free when the call starts)
    selects the bounded-memory path that moves them to the host every so many half-steps.
    Returns the list of per-game record arrays in game order."""
    torch = batch0.torch
    rng = np.random.default_rng() if rng is None else rng
    G, S, size = batch0.n_trees, batch0.shape, batch0.size
    assert batch1.n_trees == G and batch1.shape == S
    dev = batch0.dev
    stream = torch.cuda.current_stream(dev).cuda_stream
    dt = sample_dtype(S)
    d_act = torch.empty(G, dtype=torch.int32, device=dev)
    d_u = torch.zeros(G, dtype=torch.float64, device=dev)
    if device_record_bytes is None:  # the assembly at the end needs one more copy of the records beside the slabs
        device_record_bytes = torch.cuda.mem_get_info(dev)[0] // 4

    game_of = np.full(G, -1, np.int64)   # game played in each slot, -1 = none
    pair = np.zeros(G, np.int64)         # the game's pair index i (src/self_play.py:155)
    alive = np.zeros(G, bool)
    next_game = 0
    rec_parts, key_parts = [], []        # flushed records and their (game, side, half-step) keys
    outcome = np.full(n_games, -1.0, np.float32)
    slabs = []  # (half-step, pi [G,SIZE] device, board [G,2,S,S] device, played bool[G], game_of copy)

    def flush():
        for step, pi, board, played, games in slabs:
            idx = np.flatnonzero(played)
            if idx.size == 0:
                continue
            sel = torch.from_numpy(idx).to(dev)
            rec = np.zeros(idx.size, dt)
            rec["pi"] = pi.index_select(0, sel).cpu().numpy()
            rec["board"] = board.index_select(0, sel).cpu().numpy()
            key = np.empty((idx.size, 3), np.int64)
            key[:, 0], key[:, 1], key[:, 2] = games[idx], step & 1, step
            rec_parts.append(rec)
            key_parts.append(key)
        slabs.clear()

    def start_games():
        nonlocal next_game
        free = np.flatnonzero(~alive)
        n_new = min(free.size, n_games - next_game)
        if n_new == 0:
            return
        slots = free[:n_new]
        mask = np.zeros(G, np.uint8)
        mask[slots] = 1
        batch0.eng.clear_masked(mask)  # Cmcts::clear (cmcts.cpp:23-28) for these slots
        batch1.eng.clear_masked(mask)
        for m in init_moves:
            act = np.full(G, -1, np.int32)
            act[slots] = m
            batch0.make_move(act)
            batch1.make_move(act)
        game_of[slots] = np.arange(next_game, next_game + n_new)
        pair[slots] = 0
        alive[slots] = True
        next_game += n_new

    half = 0
    while True:
        side = half & 1
        if side == 0:
            start_games()
        if not alive.any():
            if next_game >= n_games:
                break
            half += 1  # every slot is waiting for the next even half-step
            continue
        searcher = batch0 if side == 0 else batch1
        searcher.simulate(sims, alive)  # leaves exactly the live trees active
        u = np.full(G, -1.0)            # negative = greedy for that tree
        sampling = alive & (pair <= tau)
        if hasattr(rng, "for_games"):  # keyed by (game, half-move of the game) instead of drawn in slot order
            u[sampling] = rng.for_games(game_of[sampling], 2 * pair[sampling] + side)
        else:
            u[sampling] = rng.random(int(sampling.sum()))
        d_u.copy_(torch.from_numpy(u))
        pi = torch.empty((G, size), dtype=torch.float32, device=dev)
        board = torch.empty((G, 2, S, S), dtype=torch.float32, device=dev)
        searcher.eng.choose_moves(d_u.data_ptr(), False, d_act.data_ptr(), pi.data_ptr(), board.data_ptr(), stream)
        slabs.append((half, pi, board, alive.copy(), game_of.copy()))
        try:
            batch0.eng.make_move_device(d_act.data_ptr(), stream)
            batch1.eng.make_move_device(d_act.data_ptr(), stream)
            searcher.eng.sync_check(stream)
            winner = (batch1 if searcher is batch0 else batch0).eng.sync_states(stream)[2]
        except _E.EngineError as e:
            raise RuntimeError(str(e)) from None
        done = alive & (winner != -1.0)
        outcome[game_of[done]] = winner[done]
        alive &= ~done
        if side == 1:
            pair[alive] += 1
        slab_bytes = len(slabs) * G * (3 * size + 1) * 4
        if (flush_plies is not None and len(slabs) >= flush_plies) or (rec_parts and len(slabs) >= 32) or \
                slab_bytes > device_record_bytes:
            flush()
        half += 1
    if not rec_parts:  # everything is still on the GPU: order and label the records there
        slab_list = [(step, step & 1, pi, board, played, games) for step, pi, board, played, games in slabs]
        slabs.clear()  # (the assembly drops each slab as it goes)
        pi = board = None
        rec, bounds = _assemble_records(torch, dev, slab_list, outcome, S)
        return GameRecords(rec, bounds)
    flush()
    rec = np.concatenate(rec_parts)
    key = np.concatenate(key_parts)
    order = np.lexsort((key[:, 2], key[:, 1], key[:, 0]))  # by game, first player's records first, ply order
    rec, key = rec[order], key[order]
    won = outcome[key[:, 0]]
    decided = (won == 0.0) | (won == 1.0)
    rec["r"] = np.where(decided, np.where(won == key[:, 1], 1.0, -1.0), 1e-3).astype(np.float32)
    bounds = np.searchsorted(key[:, 0], np.arange(n_games + 1))
    return GameRecords(rec, bounds)


class _SlotLoop:
    """Device arrays and the lagging host poll shared by the resident drivers."""

    NEXT_GAME, FINISHED, RECORDS, ERRORS, ARENA_FULL, HALF_MOVES, TREE_MOVES = range(7)  # AZ_SLOT_* of include/azmcts.h

    def __init__(self, a, b, n_games, lag, init_moves=(), game_openings=None):
        torch = a.torch
        self.torch, self.a, self.b, self.n_games, self.lag = torch, a, b, n_games, max(1, int(lag))
        self.dev = a.dev
        G = a.n_trees
        self.game_of = torch.full((G,), -1, dtype=torch.int32, device=self.dev)
        self.ply = torch.zeros(G, dtype=torch.int32, device=self.dev)
        self.counters = torch.zeros(8, dtype=torch.int32, device=self.dev)
        self.outcome = torch.full((max(n_games, 1),), -1.0, dtype=torch.float32, device=self.dev)
        self.ring = self.lag + 2
        self.host = torch.zeros((self.ring, 8), dtype=torch.int32, pin_memory=True)
        self.events = [None] * self.ring
        self.half = 0
        self.timeline = []  # (host time, half-move, games started, games finished, moves played) per poll
        self.openings = [int(m) for m in init_moves]
        self.d_game_openings, self.n_game_open = None, 0
        if game_openings is not None:
            op = np.ascontiguousarray(game_openings, np.int32).reshape(-1, n_games)
            self.d_game_openings, self.n_game_open = torch.from_numpy(op).to(self.dev), op.shape[0]

    def admit(self, stream):
        self.a.eng.slots_admit(self.b.eng, self.game_of.data_ptr(), self.ply.data_ptr(), self.counters.data_ptr(), self.n_games,
                               self.openings, stream,
                               self.d_game_openings.data_ptr() if self.n_game_open else None, self.n_game_open)

    def mark(self):
        """Behind the last call of a half-move: its counters start their way to the host."""
        k = self.half % self.ring
        self.host[k].copy_(self.counters, non_blocking=True)
        ev = self.torch.cuda.Event()
        ev.record()
        self.events[k] = ev
        self.half += 1

    def poll(self):
        """Counters of the half-move `lag` behind the one just queued (None while fewer have been
        queued): the host stays that far ahead of the GPU and never drains it."""
        i = self.half - 1 - self.lag
        if i < 0:
            return None
        return self._read(i)

    def _read(self, i):
        k = i % self.ring
        self.events[k].synchronize()
        c = self.host[k].numpy().copy()
        if c[self.ERRORS]:
            stream = self.torch.cuda.current_stream(self.dev).cuda_stream
            for batch in (self.a, self.b):
                try:
                    batch.eng.sync_check(stream)
                except _E.EngineError as e:
                    raise RuntimeError(str(e)) from None
            raise RuntimeError(f"{int(c[self.ERRORS])} trees in error")
        if c[self.ARENA_FULL]:
            raise RuntimeError("record arena full: give self_play_resident a larger arena_records")
        self.timeline.append((time.perf_counter(), i, int(c[self.NEXT_GAME]), int(c[self.FINISHED]), int(c[self.TREE_MOVES])))
        return c

    def steady_moves_per_s(self):
        """Moves per second (summed over the slots) between the first poll and the last poll at which games were
        still waiting for a slot, i.e. while every slot was busy; None when that window is empty."""
        busy = [p for p in self.timeline if p[2] < self.n_games]
        if len(busy) < 2 or busy[-1][0] <= busy[0][0]:
            return None
        return (busy[-1][4] - busy[0][4]) / (busy[-1][0] - busy[0][0])

    def last(self):
        return self._read(self.half - 1)


def self_play_resident(batch0, batch1, sims, tau, n_games, init_moves=(), seed=0, lag=3, arena_records=None,
                       game_openings=None):
    """`n_games` games of `self_play_game` (src/self_play.py:134-202) on the G tree slots of the
    two batches with NO host decision between two moves: which slot is idle, which game it gets,
    sampling or greedy, the uniform number of a sampled move (KeyedUniform(seed), computed on the
    GPU), the end of a game, the labels of its records and their place in the output all live in
    device arrays (az_slots_* of the C ABI).  A half-move is a fixed sequence of launches; the
    host queues them `lag` half-moves ahead of the GPU, reads the counters of the half-move that
    far back, and copies the finished games' records (whole games, labelled, mcts0's records
    first) from the device arena while the search goes on.
    Same games as self_play_stream(..., rng=KeyedUniform(seed)), record for record.
    game_openings (int [k][n_games] or [n_games], optional): moves of its own that game g plays after
    init_moves, before the first search.
    Returns GameRecords in game order; `.pieces` are the record arrays in the order games finished,
    `.loop` the bookkeeping object (poll timeline, steady_moves_per_s)."""
    torch = batch0.torch
    G, S, size = batch0.n_trees, batch0.shape, batch0.size
    assert batch1.n_trees == G and batch1.shape == S
    dev = batch0.dev
    stream = torch.cuda.current_stream(dev).cuda_stream
    dt = sample_dtype(S)
    rec = 3 * size + 1
    assert dt.itemsize == rec * 4
    if n_games <= 0:
        return GameRecords(np.zeros(0, dt), np.zeros(1, np.int64))
    max_plies = size  # a game ends with the board full at the latest
    loop = _SlotLoop(batch0, batch1, n_games, lag, init_moves, game_openings)
    e0, e1 = batch0.eng, batch1.eng
    game_off = torch.zeros(n_games, dtype=torch.int32, device=dev)
    game_len = torch.zeros(n_games, dtype=torch.int32, device=dev)
    stage = torch.empty((G, max_plies, rec), dtype=torch.float32, device=dev)
    if arena_records is None:
        free = torch.cuda.mem_get_info(dev)[0]
        arena_records = max(1, min(n_games * max_plies, int(free * 0.5) // (rec * 4), (1 << 31) - 1))
    arena = torch.empty((arena_records, rec), dtype=torch.float32, device=dev)
    d_act = torch.empty(G, dtype=torch.int32, device=dev)
    copier = torch.cuda.Stream(dev)
    pieces, drained = [], 0

    def drain(upto):  # records [drained, upto) are complete on the device: whole games
        nonlocal drained
        if upto > drained:
            with torch.cuda.stream(copier):
                pieces.append(arena[drained:upto].to("cpu").numpy().view(dt).reshape(-1))
            drained = int(upto)

    while True:
        side = loop.half & 1
        searcher = batch0 if side == 0 else batch1
        try:
            if side == 0:  # slots stay in phase: mcts0 of every live game searches the even half-moves
                loop.admit(stream)
                if loop.half == 0:  # slots that got no game must not be searched (a new engine's trees are all active)
                    e0.slots_turn(loop.game_of.data_ptr(), loop.ply.data_ptr(), False, -1, stream)
                    e1.slots_turn(loop.game_of.data_ptr(), loop.ply.data_ptr(), False, -1, stream)
            searcher.queue_simulate(sims)
            searcher.eng.slots_choose(loop.game_of.data_ptr(), loop.ply.data_ptr(), tau, seed, d_act.data_ptr(),
                                      stage.data_ptr(), max_plies, stream)
            e0.make_move_device(d_act.data_ptr(), stream)
            e1.make_move_device(d_act.data_ptr(), stream)
            e0.slots_finish(e1, loop.game_of.data_ptr(), loop.ply.data_ptr(), loop.counters.data_ptr(),
                            loop.outcome.data_ptr(), stage.data_ptr(), max_plies, arena.data_ptr(), arena_records,
                            game_off.data_ptr(), game_len.data_ptr(), stream)
        except _E.EngineError as e:
            raise RuntimeError(str(e)) from None
        loop.mark()
        c = loop.poll()
        if c is not None:
            drain(c[loop.RECORDS])
            if c[loop.FINISHED] >= n_games:
                break
    c = loop.last()  # the half-moves queued past the end found every slot idle
    drain(c[loop.RECORDS])
    try:
        e0.sync_check(stream)
        e1.sync_check(stream)
    except _E.EngineError as e:
        raise RuntimeError(str(e)) from None
    out = GameRecords.from_pieces(pieces, game_off.cpu().numpy().astype(np.int64), game_len.cpu().numpy().astype(np.int64))
    out.outcome = loop.outcome.cpu().numpy()[:n_games]
    out.half_moves = loop.half
    out.loop = loop
    return out


def eval_games_resident(best, latest, sims_best, sims_latest, n_games=None, init_moves=(), lag=3, game_openings=None,
                        info=None):
    """The arena (src/eval.py:111-149 around eval_game :187-209) without a host decision between
    two moves: game g is opened by `best` when g is even, each network searches the slots where it
    is to move, always the first maximum; finished games free their slot for the next game
    (n_games may exceed the number of slots; default: one game per slot).  Same tallies as
    eval_games.  game_openings: as in self_play_resident.  info (dict, optional) receives the bookkeeping object
    under "loop".  Returns (wins_best, wins_latest, draws)."""
    torch = best.torch
    G = best.n_trees
    assert latest.n_trees == G and latest.shape == best.shape
    n_games = G if n_games is None else int(n_games)
    if n_games <= 0:
        return 0, 0, 0
    dev = best.dev
    stream = torch.cuda.current_stream(dev).cuda_stream
    loop = _SlotLoop(best, latest, n_games, lag, init_moves, game_openings)
    if info is not None:
        info["loop"] = loop
    eb, el = best.eng, latest.eng
    act_b = torch.empty(G, dtype=torch.int32, device=dev)
    act_l = torch.empty(G, dtype=torch.int32, device=dev)
    gp, pp, cp = loop.game_of.data_ptr(), loop.ply.data_ptr(), loop.counters.data_ptr()
    while True:
        try:
            loop.admit(stream)
            # `best` moves when game + half-move is even (it opens the even games, src/eval.py:117,134)
            eb.slots_turn(gp, pp, True, 0, stream)
            el.slots_turn(gp, pp, True, 1, stream)
            best.queue_simulate(sims_best)
            latest.queue_simulate(sims_latest)
            eb.slots_choose(gp, pp, -1, 0, act_b.data_ptr(), None, 0, stream)  # always greedy (src/eval.py:192,202)
            el.slots_choose(gp, pp, -1, 0, act_l.data_ptr(), None, 0, stream)
            act = torch.maximum(act_b, act_l)  # every live game was searched by exactly one side
            eb.make_move_device(act.data_ptr(), stream)
            el.make_move_device(act.data_ptr(), stream)
            eb.slots_finish(el, gp, pp, cp, loop.outcome.data_ptr(), stream=stream)
        except _E.EngineError as e:
            raise RuntimeError(str(e)) from None
        loop.mark()
        c = loop.poll()
        if c is not None and c[loop.FINISHED] >= n_games:
            break
    loop.last()
    try:
        eb.sync_check(stream)
        el.sync_check(stream)
    except _E.EngineError as e:
        raise RuntimeError(str(e)) from None
    winner = loop.outcome.cpu().numpy()[:n_games]
    # the player to move once the opening has been played (src/eval.py:105)
    first_player = (len(loop.openings) + loop.n_game_open) % 2
    second_player = 1 - first_player
    best_opens = (np.arange(n_games) % 2) == 0
    opener_won = winner == first_player
    other_won = winner == second_player
    draws = int((~(opener_won | other_won)).sum())
    wins_best = int((opener_won & best_opens).sum() + (other_won & ~best_opens).sum())
    wins_latest = int((opener_won & ~best_opens).sum() + (other_won & best_opens).sum())
    return wins_best, wins_latest, draws


def eval_games(best, latest, sims_best, sims_latest, init_moves=()):
    """`eval_models`' game loop (src/eval.py:111-149,187-209) for every game at once.

    best / latest: TreeBatch of equal size holding the two networks' trees.  Game g
    is opened by `best` when g is even, by `latest` when odd (src/eval.py:117,134).
    Returns (wins_best, wins_latest, draws).
    """
    G, size = best.n_trees, best.size
    assert latest.n_trees == G
    for m in init_moves:
        best.make_move(np.full(G, m, np.int32))
        latest.make_move(np.full(G, m, np.int32))
    first_player = int(best.eng.get_state(0)[0]) if G else 0  # src/eval.py:105
    second_player = 1 - first_player
    best_opens = (np.arange(G) % 2) == 0
    alive = np.ones(G, bool)
    for _ in range(size):
        for half in (0, 1):
            if not alive.any():
                break
            best_to_move = best_opens if half == 0 else ~best_opens
            act = np.full(G, -1, np.int32)
            for batch, mine, sims in ((best, best_to_move, sims_best), (latest, ~best_to_move, sims_latest)):
                mask = alive & mine
                if not mask.any():
                    continue
                batch.simulate(sims, mask)
                idx = np.flatnonzero(mask)
                act[idx] = batch.probs(idx).argmax(axis=1)  # always greedy (src/eval.py:192,202)
            best.make_move(act)
            latest.make_move(act)
            alive &= best.winners() == -1.0
        if not alive.any():
            break
    winner = best.winners()
    wins_best = wins_latest = draws = 0
    for g in range(G):
        opener_won = winner[g] == first_player
        other_won = winner[g] == second_player
        if not (opener_won or other_won):
            draws += 1
        elif opener_won == bool(best_opens[g]):
            wins_best += 1
        else:
            wins_latest += 1
    return wins_best, wins_latest, draws


def eval_games_device(best, latest, sims_best, sims_latest, init_moves=()):
    """eval_games with the move decision on the GPU (az_choose_moves, greedy): per
    half-move the host only reads the winners.  Same tallies as eval_games."""
    torch = best.torch
    G, size = best.n_trees, best.size
    assert latest.n_trees == G
    dev = best.dev
    stream = torch.cuda.current_stream(dev).cuda_stream
    for m in init_moves:
        best.make_move(np.full(G, m, np.int32))
        latest.make_move(np.full(G, m, np.int32))
    first_player = int(best.eng.get_state(0)[0]) if G else 0  # src/eval.py:105
    second_player = 1 - first_player
    best_opens = (np.arange(G) % 2) == 0
    alive = np.ones(G, bool)
    act_b = torch.empty(G, dtype=torch.int32, device=dev)
    act_l = torch.empty(G, dtype=torch.int32, device=dev)
    for _ in range(size):
        for half in (0, 1):
            if not alive.any():
                break
            best_to_move = best_opens if half == 0 else ~best_opens
            act_b.fill_(-1)
            act_l.fill_(-1)
            for batch, mine, sims, d_act in ((best, best_to_move, sims_best, act_b),
                                             (latest, ~best_to_move, sims_latest, act_l)):
                mask = alive & mine
                if not mask.any():
                    continue
                batch.simulate(sims, mask)  # leaves exactly these trees active
                batch.eng.choose_moves(None, True, d_act.data_ptr(), None, None, stream)
            act = torch.maximum(act_b, act_l)  # every live game was searched by exactly one side
            try:
                best.eng.make_move_device(act.data_ptr(), stream)
                latest.eng.make_move_device(act.data_ptr(), stream)
                latest.eng.sync_check(stream)
                winner = best.eng.sync_states(stream)[2]
            except _E.EngineError as e:
                raise RuntimeError(str(e)) from None
            alive &= winner == -1.0
        if not alive.any():
            break
    winner = best.winners()
    opener_won = winner == first_player
    other_won = winner == second_player
    draws = int((~(opener_won | other_won)).sum())
    wins_best = int((opener_won & best_opens).sum() + (other_won & ~best_opens).sum())
    wins_latest = int((opener_won & ~best_opens).sum() + (other_won & best_opens).sum())
    return wins_best, wins_latest, draws


def promote(wins_best, wins_latest, threshold=0.6):
    """src/eval.py:176-180 with EVAL_TRESHOLD (general_config.py:24)."""
    return wins_latest / (wins_latest + wins_best) >= threshold


# ---------------------------------------------------------------- multi-GPU
def gather_samples(samples, dst=0, group=None, all_ranks=False, chunk_bytes=64 << 20):
    """Every rank's sample records in rank order, on rank `dst` (the other ranks get an
    empty array, or the whole set too with all_ranks=True).  Point-to-point transfers
    in pieces of at most `chunk_bytes` through ONE staging buffer of that size per
    rank (device memory with the nccl backend = RCCL over xGMI, host memory with
    gloo), so the exchange needs no world x payload buffers: a 65,536-game batch of
    20x20 records is about 19 GB, which only the host of `dst` ever holds whole.
    `samples`: structured array of this rank (any length, also 0)."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return samples
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    nccl = dist.get_backend(group) == "nccl"
    dev = torch.device("cuda", torch.cuda.current_device()) if nccl else torch.device("cpu")
    raw = np.frombuffer(samples.tobytes(), np.uint8)
    n = torch.tensor([raw.size], dtype=torch.int64, device=dev)
    sizes = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(sizes, n, group=group)  # 8 bytes per rank
    sizes = [int(s.item()) for s in sizes]
    chunk = max(1, int(chunk_bytes))
    stage = torch.empty(min(chunk, max(sum(sizes) if all_ranks else max(sizes), 1)), dtype=torch.uint8, device=dev)
    # dist.send/recv address ranks of the default group
    to_global = (lambda r: r) if group is None else (lambda r: dist.get_global_rank(group, r))
    if rank == dst:
        total = np.empty(sum(sizes), np.uint8)
        at = 0
        for src in range(world):
            for off in range(0, sizes[src], chunk):
                k = min(chunk, sizes[src] - off)
                if src == dst:
                    total[at:at + k] = raw[off:off + k]
                else:
                    dist.recv(stage[:k], src=to_global(src), group=group)
                    total[at:at + k] = stage[:k].cpu().numpy()
                at += k
    else:
        total = None
        for off in range(0, sizes[rank], chunk):
            k = min(chunk, sizes[rank] - off)
            stage[:k].copy_(torch.from_numpy(raw[off:off + k].copy()))
            dist.send(stage[:k], dst=to_global(dst), group=group)
    if all_ranks:
        n_total = sum(sizes)
        if total is None:
            total = np.empty(n_total, np.uint8)
        for off in range(0, n_total, chunk):
            k = min(chunk, n_total - off)
            if rank == dst:
                stage[:k].copy_(torch.from_numpy(total[off:off + k].copy()))
            dist.broadcast(stage[:k], src=to_global(dst), group=group)
            if rank != dst:
                total[off:off + k] = stage[:k].cpu().numpy()
    if total is None:
        return np.zeros(0, samples.dtype)
    return np.frombuffer(total.tobytes(), dtype=samples.dtype)


def broadcast_parameters(tensors, src=0, group=None):
    """The evaluator replicas of all ranks take rank `src`'s weights (one
    broadcast per tensor, in place; RCCL with the nccl backend).  `tensors`: an
    evaluator with a packed `params` tensor (FusedNetEvaluator), a torch module,
    or an iterable of tensors."""
    import torch
    import torch.distributed as dist

    if hasattr(tensors, "params") and isinstance(tensors.params, torch.Tensor):
        tensors = [tensors.params]
    elif isinstance(tensors, torch.nn.Module):
        tensors = list(tensors.parameters()) + list(tensors.buffers())
    tensors = list(tensors)
    if dist.is_available() and dist.is_initialized():
        for t in tensors:
            dist.broadcast(t.data, src=src, group=group)
    return tensors


def reduce_counters(values, group=None):
    """Sum of per-rank counters (sims, expansions, wins ...) over all ranks."""
    import torch
    import torch.distributed as dist

    arr = np.asarray(values, np.float64)
    if not (dist.is_available() and dist.is_initialized()):
        return arr
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    t = torch.from_numpy(arr.copy()).to(dev)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t.cpu().numpy()


def save_samples(path, games):
    """One `.npy` structured array, the file format src/train.py loads
    (np.save at src/self_play.py:111,128)."""
    data = all_records(games)
    np.save(path, data)
    return data

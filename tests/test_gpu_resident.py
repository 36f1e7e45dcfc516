"""The device-resident game loops (self_play_resident, eval_games_resident: slot bookkeeping,
move sampling, game ends, record labels and record layout decided on the GPU through the
az_slots_* entry points) against the host-stepped drivers, which are themselves pinned to the
reference's self_play_game / eval_game (tests/test_gpu_drivers.py, tests/test_gpu_stream.py).
Bit-exact: every record field of every game, every tally."""
import numpy as np
import pytest

from helpers import ExactTorchEvaluator

pytestmark = pytest.mark.gpu


def _pair(G, S, sims, eps, ev=None, ev1=None, noise_seeds=(0, 0), threads=0):
    from alphazero_amd import selfplay as sp

    ev = ExactTorchEvaluator() if ev is None else ev
    ev1 = ev if ev1 is None else ev1
    return (sp.TreeBatch(G, S, ev, cpuct=4.0, eps=eps, nodes_per_tree=sims * 6 + 64, noise_seed=noise_seeds[0], threads=threads),
            sp.TreeBatch(G, S, ev1, cpuct=4.0, eps=eps, nodes_per_tree=sims * 6 + 64, noise_seed=noise_seeds[1], threads=threads))


def _same_games(a, b):
    assert len(a) == len(b)
    for g, (x, y) in enumerate(zip(a, b)):
        assert x.dtype == y.dtype and len(x) == len(y), g
        for f in ("board", "pi", "r"):
            assert np.array_equal(x[f], y[f]), (g, f)


def test_keyed_uniform_is_a_function_of_game_and_ply():
    from alphazero_amd import selfplay as sp

    k = sp.KeyedUniform(77)
    u = k.for_games(np.array([3, 3, 4, 3]), np.array([0, 1, 0, 0]))
    assert u[0] == u[3] and len({u[0], u[1], u[2]}) == 3 and np.all((u > 0) & (u < 1))
    big = k.for_games(np.arange(20000) % 5000, np.arange(20000) // 5000)
    assert abs(big.mean() - 0.5) < 0.01 and abs(big.var() - 1 / 12) < 0.005


@pytest.mark.parametrize("eps,tau,lag,init", [(0.0, 2, 3, ()), (0.25, 2, 1, ()), (0.25, 50, 5, (24,)), (0.0, -1, 2, (3, 10))])
def test_resident_self_play_equals_stream_driver(eps, tau, lag, init):
    """Slot reuse (11 games on 4 slots), sampled and greedy phases of games of different ages in
    one batch, openings: the same games, record for record, as the host-stepped driver fed with
    the host mirror of the device's uniform numbers."""
    from alphazero_amd import selfplay as sp

    G, S, sims, n_games, seed = 4, 7, 30, 11, 5
    b0, b1 = _pair(G, S, sims, eps)
    want = sp.self_play_stream(b0, b1, sims, tau, n_games, init_moves=init, rng=sp.KeyedUniform(seed))
    b0.close(); b1.close()
    b0, b1 = _pair(G, S, sims, eps)
    got = sp.self_play_resident(b0, b1, sims, tau, n_games, init_moves=init, seed=seed, lag=lag)
    b0.close(); b1.close()
    _same_games(want, got)
    # the pieces hold whole games in the order they finished; together they are all records
    assert sum(len(p) for p in got.pieces) == sum(len(g) for g in got) == len(got.flat)
    if tau >= 0:
        assert len({g["pi"].tobytes() for g in got}) > 1  # sampling made the games differ


def test_resident_self_play_with_the_fused_network_on_the_stepped_path():
    """More slots than CUs (the stepped launches of az_simulate_net), the library's own network
    kernel, 15x15: 700 games on 300 slots, both drivers bit-identical."""
    import torch
    from alphazero_amd import selfplay as sp
    from alphazero_amd.evaluator import FusedNetEvaluator
    from alphazero_amd.model import PolicyValueNet

    G, S, sims, tau, n_games, seed = 300, 15, 12, 3, 700, 9
    torch.manual_seed(2)
    ev = FusedNetEvaluator(PolicyValueNet(S).eval(), S, 0)
    b0, b1 = _pair(G, S, sims, 0.25, ev, noise_seeds=(0, 1000))
    want = sp.self_play_stream(b0, b1, sims, tau, n_games, rng=sp.KeyedUniform(seed))
    b0.close(); b1.close()
    b0, b1 = _pair(G, S, sims, 0.25, ev, noise_seeds=(0, 1000))
    got = sp.self_play_resident(b0, b1, sims, tau, n_games, seed=seed)
    stats = b0.eng.get_stats()
    b0.close(); b1.close()
    _same_games(want, got)
    assert stats["sims"] > 0
    decided = sum(1 for g in got if g["r"][0] != np.float32(1e-3))
    assert decided > 0


@pytest.mark.parametrize("fused,G,S,threads", [(False, 4, 7, 4), (True, 20, 9, 4), (True, 300, 9, 2)])
def test_resident_self_play_with_concurrent_workers(fused, G, S, threads):
    """threads > 1 (the reference's callers play with four, general_config.py:72) inside the resident loop: the workers of
    every tree in lock step, with an external evaluator (az_select_workers / az_expand_backup_workers), with the fused
    network on few slots (the one-launch cluster search) and on many (stepped launches); the same games, record for
    record, as the host-stepped driver with the same worker count."""
    import torch
    from alphazero_amd import engine as E
    from alphazero_amd import selfplay as sp
    from alphazero_amd.evaluator import FusedNetEvaluator
    from alphazero_amd.model import PolicyValueNet

    sims, tau, n_games, seed = 26, 2, 2 * G + 3, 13
    ev = None
    if fused:
        torch.manual_seed(4)
        ev = FusedNetEvaluator(PolicyValueNet(S).eval(), S, 0)
    b0, b1 = _pair(G, S, sims, 0.25, ev, noise_seeds=(0, 1000), threads=threads)
    want = sp.self_play_stream(b0, b1, sims, tau, n_games, rng=sp.KeyedUniform(seed))
    b0.close(); b1.close()
    b0, b1 = _pair(G, S, sims, 0.25, ev, noise_seeds=(0, 1000), threads=threads)
    got = sp.self_play_resident(b0, b1, sims, tau, n_games, seed=seed)
    path = b0.eng.get_net_path()
    b0.close(); b1.close()
    _same_games(want, got)
    if fused:
        assert path == (E.NET_PATH_CLUSTER if G * (threads + 1) <= 256 else E.NET_PATH_STEPPED)


def test_resident_record_arena_overflow_is_reported():
    from alphazero_amd import selfplay as sp

    b0, b1 = _pair(4, 7, 20, 0.0)
    with pytest.raises(RuntimeError, match="arena full"):
        sp.self_play_resident(b0, b1, 20, 2, 8, seed=1, arena_records=5)
    b0.close(); b1.close()


@pytest.mark.parametrize("eps,init", [(0.0, ()), (0.25, (24,))])
def test_resident_arena_equals_host_arena(eps, init):
    """Two different evaluators, alternating openers: the tallies of the host-stepped arena, with one
    game per slot and with 13 games on 4 slots (per game the same trees, so the same outcomes)."""
    from alphazero_amd import selfplay as sp

    class Other(ExactTorchEvaluator):
        def __call__(self, boards):
            v, p = super().__call__(boards)
            return (-v).contiguous(), p.flip(1).contiguous()

    S, sb, sl, n_games = 7, 30, 24, 13
    best, latest = _pair(n_games, S, 40, eps, ExactTorchEvaluator(), Other(), noise_seeds=(0, 1000))
    want = sp.eval_games(best, latest, sb, sl, init_moves=init)
    best.close(); latest.close()
    best, latest = _pair(n_games, S, 40, eps, ExactTorchEvaluator(), Other(), noise_seeds=(0, 1000))
    assert sp.eval_games_resident(best, latest, sb, sl, init_moves=init) == want
    best.close(); latest.close()
    best, latest = _pair(4, S, 40, eps, ExactTorchEvaluator(), Other(), noise_seeds=(0, 1000))
    assert sp.eval_games_resident(best, latest, sb, sl, n_games=n_games, init_moves=init, lag=2) == want
    best.close(); latest.close()
    assert sum(want) == n_games


def test_resident_loops_with_an_opening_of_its_own_per_game():
    """game_openings: game g starts from its own first stone (what a host loop does with make_move before
    the first search): arena tallies and self-play records equal the host-stepped drivers'."""
    from alphazero_amd import selfplay as sp

    S, sims, n_games = 7, 24, 10
    first = (np.arange(n_games) * 5 + 3) % (S * S)
    best, latest = _pair(n_games, S, 40, 0.25, noise_seeds=(0, 1000))
    for b in (best, latest):
        b.make_move(first.astype(np.int32))
    want = sp.eval_games_device(best, latest, sims, sims)
    best.close(); latest.close()
    best, latest = _pair(3, S, 40, 0.25, noise_seeds=(0, 1000))
    info = {}
    got = sp.eval_games_resident(best, latest, sims, sims, n_games=n_games, game_openings=first, info=info)
    best.close(); latest.close()
    assert got == want
    tl = info["loop"].timeline
    assert tl[-1][3] == n_games and tl[-1][4] > 0  # every game finished; moves were counted
    # self-play: the same opening stone per game through the host-stepped stream driver is not expressible
    # (its openings are shared), so compare game by game against one-game batches
    b0, b1 = _pair(3, S, 40, 0.0)
    got = sp.self_play_resident(b0, b1, sims, 1, n_games, seed=4, game_openings=first)
    b0.close(); b1.close()
    for g in (0, 4, 9):
        b0, b1 = _pair(1, S, 40, 0.0)

        class OneGame(sp.KeyedUniform):  # game 0 of the one-game batch draws game g's numbers
            def for_games(self, games, plies):
                return super().for_games(np.full_like(np.asarray(games), g), plies)

        one = sp.self_play_stream(b0, b1, sims, 1, 1, init_moves=(int(first[g]),), rng=OneGame(4))
        b0.close(); b1.close()
        _same_games([got[g]], one)

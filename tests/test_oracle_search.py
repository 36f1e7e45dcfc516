"""Known answers for the oracle's search restatement.

node.cpp / cmcts.cpp cannot be built in this image (GSL headers absent), so
these pins are the facts recorded from the real reference at survey time plus
properties that follow from its source text (cited per test)."""
import numpy as np
import pytest

import az_oracle_py as O


def _uniform(S):
    return O.uniform_evaluator(np.float32(1.0 / (S * S)))


def test_default_alpha_values():
    # cmcts.cpp:75-81; 15x15 value recorded from the reference: 0.0477326968973747
    for S, want in ((15, 0.0477326968973747), (20, None), (13, None)):
        t = O.OracleTree(S)
        t.set_alpha_default()
        L = 2.0 * S
        assert t.get_alpha() == 10 / ((S * S * L - (L * L + L) * 0.5) / L)
        if want is not None:
            assert repr(t.get_alpha()) == repr(want)
    t = O.OracleTree(15, alpha=0.3)
    assert t.get_alpha() == 0.3  # constructor fills the array (cmcts.cpp:20)


def test_fresh_root_counts_and_rerooting():
    # expanding the root is not a visit: 800 searches leave nodeN == 799 (cmcts.cpp:190-243);
    # a re-rooted child keeps its statistics and the next call adds to them (cmcts.cpp:284-294)
    t = O.OracleTree(15, cpuct=4.0)
    t.set_eps(0.0)
    ev = _uniform(15)
    t.simulate(800, ev)
    assert t.root_n() == 799 and t.root_counts().sum() == 799
    a = int(t.root_counts().argmax())
    kept = t.node_at([a])[0]
    t.make_move(a)
    assert t.root_n() == kept
    t.simulate(800, ev)
    assert t.root_n() == kept + 800


def test_simulate_one_then_get_prob_throws():
    t = O.OracleTree(15, cpuct=4.0)
    t.set_eps(0.0)
    t.simulate(1, _uniform(15))
    with pytest.raises(O.OracleError, match="Not a distribution"):
        t.get_prob()
    t2 = O.OracleTree(15)
    with pytest.raises(O.OracleError, match="Not a distribution"):
        t2.get_prob()  # unexpanded root: 0 / -1


def test_uniform_prior_selection_depth():
    # recorded from the reference at survey time: mean selection depth 1.75 for 800
    # searches x 40 moves with a constant prior and zero value
    t = O.OracleTree(15, cpuct=4.0)
    t.set_eps(0.0)
    ev = _uniform(15)
    for _ in range(40):
        t.simulate(800, ev)
        t.make_move(int(t.root_counts().argmax()))
        if t.winner != -1:
            break
    st = t.stats()
    assert 1.70 < st["select_calls"] / st["sims"] < 1.80


def test_error_paths():
    t = O.OracleTree(15)
    with pytest.raises(O.OracleError) as ei:
        t.simulate(0, _uniform(15))
    assert ei.value.code == 2 and "must be at least 1" in str(ei.value)
    with pytest.raises(O.OracleError, match="Epsilon must be in range"):
        t.set_eps(1.01)
    with pytest.raises(O.OracleError, match="Invalid value"):
        t.set_cuda(0)  # the guard is true for every int (cmcts.cpp:70)
    with pytest.raises(O.OracleError, match="Invalid index"):
        t.make_movexy(15, 0)
    t.make_movexy(3, 2)
    assert t.raw_board()[0, 2, 3] == 1 and t.player == 1 and t.move_cnt == 1
    with pytest.raises(O.OracleError, match="Invalid move"):
        t.make_move(2 * 15 + 3)


def test_virtual_loss_is_two_roundings():
    # W -= 1.0 at select (node.cpp:118), W += (double)v + 1.0 at backup (node.cpp:80):
    # with v = 1e-20f the value is absorbed: W ends at 0, not at -v
    S = 5
    v = np.float32(1e-20)
    ev = O.callback_evaluator(lambda b, p: (float(v), np.full(25, 1 / 25, np.float32)), S)
    t = O.OracleTree(S, cpuct=1.0)
    t.set_eps(0.0)
    t.simulate(3, ev)  # expand root, then two searches each expanding a child
    w = t.root_w()
    a = int(np.flatnonzero(t.root_counts())[0])
    want = (0.0 - 1.0) + (float(-v) + 1.0)
    assert w[a] == want == 0.0 and want != float(-v)


def test_noise_rows_restart_every_call():
    # every simulate() call re-creates a default-seeded generator (cmcts.cpp:132-133):
    # the k-th expansion of each call mixes in the same noise row
    S, size = 7, 49
    noise = np.random.default_rng(1).dirichlet(np.full(size, 0.3), size=8)
    ev = O.hash_evaluator(3)
    t = O.OracleTree(S, cpuct=2.0)
    t.set_eps(0.5)
    t.simulate(1, ev, noise)  # expands the root with row 0
    v, p = O.hash_eval(S, 3, t.raw_board())
    assert np.array_equal(t.root_p(), 0.5 * p.astype(np.float64) + 0.5 * noise[0])
    t.simulate(2, ev, noise)  # first expansion of THIS call uses row 0 again
    a = int(np.flatnonzero(t.root_counts())[0])
    child = t.node_at([a])
    t2 = O.OracleTree(S, cpuct=2.0)
    t2.make_move(a)
    v2, p2 = O.hash_eval(S, 3, t2.raw_board())
    assert np.array_equal(child[3], 0.5 * p2.astype(np.float64) + 0.5 * noise[0])
    with pytest.raises(O.OracleError):
        t.simulate(20, ev, noise)  # more expansions than rows


def test_thread_fanout_overshoots():
    # n % threads != 0 -> every worker runs n/threads + 1 searches (cmcts.cpp:104-107)
    t = O.OracleTree(9, cpuct=2.0)
    t.set_eps(0.0)
    t.set_threads(4)
    t.simulate(50, O.hash_evaluator(1))
    assert t.stats()["sims"] == 52
    t2 = O.OracleTree(9, cpuct=2.0)
    t2.set_eps(0.0)
    t2.set_threads(8)
    t2.simulate(3, O.hash_evaluator(1))  # n < threads: n workers of one search
    assert t2.stats()["sims"] == 3


def test_select_arrays_branches_and_ties():
    # node.cpp:95-114: N == 0 takes the prior-only branch, strict '>' keeps the first maximum,
    # occupied squares are skipped
    size = 25
    board = np.zeros((2, size), np.int8)
    n = np.zeros(size, np.int64)
    w = np.zeros(size)
    p = np.full(size, 0.04)
    assert O.select_arrays(n, w, p, 0, board, 4.0) == 0
    board[0, 0] = 1
    board[1, 1] = 1
    assert O.select_arrays(n, w, p, 0, board, 4.0) == 2
    n[2], w[2] = 1, 0.9
    node_n = 1
    u_visited = 0.9 / 1 + 4.0 * 0.04 * np.sqrt(node_n + 1e-8) / 2
    u_fresh = 4.0 * 0.04 * np.sqrt(node_n + 1e-8)
    assert u_visited > u_fresh
    assert O.select_arrays(n, w, p, node_n, board, 4.0) == 2
    w[2] = -0.5
    assert O.select_arrays(n, w, p, node_n, board, 4.0) == 3
    # one ulp decides
    p2 = p.copy()
    p2[7] = np.nextafter(p2[7], 1.0)
    assert O.select_arrays(n, w, p2, node_n, board, 4.0) == 7
    board[:] = 1
    assert O.select_arrays(n, w, p, node_n, board, 4.0) == -1  # no legal action


def test_lockstep_schedule_of_the_thread_fanout_is_deterministic_and_conserves_visits():
    """threads > 1 with the lock-step interleaving (az_oracle.h): in the first step of a call on a fresh
    root all T workers meet the unexpanded root, one expands it and none counts a visit
    (cmcts.cpp:190-243, node.cpp:30-33), so root nodeN = T * (per - 1); the visit counts add up;
    two runs agree bit for bit; workers that meet in one unexpanded child both count a visit of the
    action while the child itself has been selected from by neither."""
    S, size = 9, 81
    for T, n in ((4, 50), (2, 7), (8, 3), (3, 30)):
        per = 1 if n < T else (n // T if n % T == 0 else n // T + 1)
        th = min(T, n)
        runs = []
        for _ in range(2):
            t = O.OracleTree(S, cpuct=4.0)
            t.set_eps(0.25)
            t.set_alpha_default()
            t.set_threads(T)
            t.set_schedule(1)
            noise = np.random.default_rng(5).dirichlet(np.full(size, t.get_alpha()), size=per + 2)
            t.make_move(40)
            t.simulate(n, O.hash_evaluator(3), noise)
            assert t.root_n() == th * (per - 1)
            assert t.root_counts().sum() == t.root_n()
            st = t.stats()
            assert st["sims"] == th * per and st["expansions"] + st["terminals"] <= st["sims"]
            runs.append((t.root_counts().copy(), t.root_w().copy(), t.root_p().copy(), st))
        assert np.array_equal(runs[0][0], runs[1][0]) and np.array_equal(runs[0][1], runs[1][1])
        assert np.array_equal(runs[0][2], runs[1][2]) and runs[0][3] == runs[1][3]
    # a sharp prior drives several workers of a step into the same new child
    t = O.OracleTree(S, cpuct=4.0)
    t.set_eps(0.0)
    t.set_threads(4)
    t.set_schedule(1)

    def sharp(board, player):
        p = np.full(size, 1e-4, np.float32)
        empty = np.flatnonzero(board.reshape(2, size).sum(axis=0) == 0)
        p[empty[0]] = 0.9
        return 0.0, p

    t.simulate(12, O.callback_evaluator(sharp, S, log_priors=False))
    counts = t.root_counts()
    a = int(counts.argmax())
    assert counts[a] >= 2
    child = t.node_at([a])
    assert child is not None and child[0] < counts[a] - 1  # fewer selects from the child than visits of the action minus one

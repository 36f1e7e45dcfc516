#!/usr/bin/env python3
"""Generates tests/golden/caller_games.npz: games played by the reference's OWN
caller functions -- `self_play_game` (/root/reference/src/self_play.py:134-202)
and `eval_game` (/root/reference/src/eval.py:187-209) -- driving this build's
CPU oracle trees through the `cmcts.mcts` method names.  The records pin the
caller rules (sampling while i <= tau, one-hot pi afterwards, first-argmax,
labels 1 / -1 / 1e-3, the board view stored with each record) for the GPU
drivers and for the loop restatements in tests/helpers.py.

Run in the build container only (the reference does not travel):
    cd /tmp && python /root/repo/tests/golden/make_caller_golden.py
The reference modules do `import cmcts`; this build's drop-in module of that
name is what they get.  Board edge = the reference's general_config.SHAPE (13).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
sys.dont_write_bytecode = True
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import alphazero_amd.cmcts as _cmcts  # noqa: E402

sys.modules["cmcts"] = _cmcts
sys.path.insert(0, "/root/reference")
sys.path.insert(0, "/root/reference/src")
import self_play as ref_self_play  # noqa: E402  (reference modules)
import eval as ref_eval  # noqa: E402

import az_oracle_py as O  # noqa: E402
from helpers import oracle_exact_evaluator  # noqa: E402

S = ref_self_play.SHAPE
SIZE = S * S


class Mcts:
    """An oracle tree behind the names the reference callers use."""

    def __init__(self, eps, noise, opening):
        self.t = O.OracleTree(S, cpuct=4.0)
        self.t.set_alpha_default()
        self.t.set_eps(eps)
        self.noise = noise
        self.ev = oracle_exact_evaluator(S)
        self.moves = []
        for m in opening:
            self.t.make_move(m)

    def simulate(self, n):
        self.t.simulate(n, self.ev, self.noise)

    def get_prob(self):
        return self.t.get_prob()

    def get_board(self):
        return self.t.get_board()

    def make_move(self, m):
        self.moves.append(int(m))
        self.t.make_move(int(m))

    @property
    def winner(self):
        return self.t.winner

    def __repr__(self):
        return "oracle tree"


def main():
    out = {"shape": np.int32(S)}
    cases = [  # (name, sims, tau, eps, seed, opening)
        ("sp0", 30, 4, 0.0, 11, (84, 85)),
        ("sp1", 40, -1, 0.25, 5, (70, 98)),
        ("sp2", 25, 200, 0.0, 3, ()),
    ]
    for name, sims, tau, eps, seed, opening in cases:
        noise = None
        if eps > 0:
            alpha = O.OracleTree(S).get_alpha()
            t = O.OracleTree(S); t.set_alpha_default(); alpha = t.get_alpha()
            noise = np.random.default_rng(seed).dirichlet(np.full(SIZE, alpha), size=sims + 2)
        m0, m1 = Mcts(eps, noise, opening), Mcts(eps, noise, opening)
        np.random.seed(seed)
        data = ref_self_play.self_play_game(m0, m1, sims, tau)
        out[f"{name}_params"] = np.array([sims, tau, seed], np.int64)
        out[f"{name}_eps"] = np.float64(eps)
        out[f"{name}_opening"] = np.array(opening, np.int64)
        out[f"{name}_data"] = data
        out[f"{name}_moves"] = np.array(m0.moves, np.int64)
        out[f"{name}_winner"] = np.float64(m0.winner)
        if noise is not None:
            out[f"{name}_noise"] = noise
        print(name, len(data), m0.winner)
    for name, s_first, s_second, opening in (("ev0", 30, 45, (84, 85)), ("ev1", 45, 30, (20, 21))):
        first, second = Mcts(0.0, None, opening), Mcts(0.0, None, opening)
        ref_eval.eval_game(first, s_first, second, s_second)
        out[f"{name}_params"] = np.array([s_first, s_second], np.int64)
        out[f"{name}_opening"] = np.array(opening, np.int64)
        out[f"{name}_moves"] = np.array(first.moves, np.int64)
        out[f"{name}_winner"] = np.float64(first.winner)
        print(name, len(first.moves), first.winner)
    np.savez_compressed(os.path.join(HERE, "caller_games.npz"), **out)


if __name__ == "__main__":
    main()

"""GPU rules parity: the engine's bitboard make_move / five-in-a-row kernels
against the fixtures produced by the reference's own State (tests/golden/rules_S*.json).
Bit-exact: player, move_cnt and winner after every ply, get_board, illegal moves.
"""
import glob
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

FILES = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "rules_S*.json")))


@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(p) for p in FILES])
def test_rules_on_gpu(path):
    from alphazero_amd import engine as E

    g = json.load(open(path))
    S = g["shape"]
    cases = g["cases"]
    B = len(cases)
    eng = E.Engine(B, S, nodes_per_tree=4)
    max_len = max(len(c["moves"]) for c in cases)
    done = [False] * B
    for ply in range(max_len):
        act = np.full(B, -1, np.int32)
        for t, c in enumerate(cases):
            if done[t] or ply >= len(c["moves"]):
                continue
            if ply >= c["n_ok"]:
                continue  # the illegal move is tried separately below
            if "force" in c and c["force"][ply] >= 0:
                eng.set_player(t, c["force"][ply])
            act[t] = c["moves"][ply]
        eng.make_move(act)
        pl, mc, wi = eng.get_states()
        for t, c in enumerate(cases):
            if act[t] < 0:
                continue
            assert pl[t] == c["player"][ply], (c["name"], ply)
            assert mc[t] == c["move_cnt"][ply], (c["name"], ply)
            assert float(wi[t]) == c["winner"][ply], (c["name"], ply)
    # illegal moves: the call fails, names the move, and changes nothing
    for t, c in enumerate(cases):
        if c["n_ok"] < len(c["moves"]):
            bad = c["moves"][c["n_ok"]]
            before = eng.get_state(t)
            act = np.full(B, -1, np.int32)
            act[t] = bad
            if bad < 0:
                continue  # negative actions mean "leave the tree alone" in the batched call
            with pytest.raises(E.EngineError) as ei:
                eng.make_move(act)
            assert ei.value.code == E.ERR_INVALID_MOVE
            assert "Invalid move" in str(ei.value)
            assert eng.get_state(t) == before
    # get_board (planes swapped for player 1) and validity probes
    for t, c in enumerate(cases):
        if "get_board" in c:
            got = eng.get_board(t).astype(np.int64).reshape(-1).tolist()
            assert got == c["get_board"], c["name"]
        if "valid" in c:
            for a, v in zip(c["valid_probe"], c["valid"]):
                if a < 0:
                    continue
                if not v:
                    act = np.full(B, -1, np.int32)
                    act[t] = a
                    with pytest.raises(E.EngineError):
                        eng.make_move(act)
    eng.close()

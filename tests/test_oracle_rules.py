"""The oracle's gomoku rules against the fixtures the reference's own State produced.

tests/golden/rules_S*.json were written by tests/golden/make_rules_golden.py
from /root/reference/mcts/src/state.cpp compiled unmodified; this test pins
oracle/az_oracle.c's restatement of state.cpp:29-110,152-272 to them.
"""
import glob
import json
import os

import numpy as np
import pytest

import az_oracle_py as O

FILES = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "rules_S*.json")))


def test_fixture_files_present():
    assert len(FILES) >= 8


@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(p) for p in FILES])
def test_rules_match_reference(path):
    g = json.load(open(path))
    S = g["shape"]
    for c in g["cases"]:
        k, pl, mc, wi = O.rules_play(S, c["moves"], c.get("force"))
        assert k == c["n_ok"], c["name"]
        assert pl.tolist() == c["player"], c["name"]
        assert mc.tolist() == c["move_cnt"], c["name"]
        assert [float(x) for x in wi] == c["winner"], c["name"]


@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(p) for p in FILES])
def test_get_board_and_is_valid_match_reference(path):
    g = json.load(open(path))
    S = g["shape"]
    n_board = 0
    for c in g["cases"]:
        if "get_board" not in c and "valid" not in c:
            continue
        t = O.OracleTree(S)
        force = c.get("force")
        for i, m in enumerate(c["moves"]):
            if force is not None and force[i] >= 0:
                t.set_player(force[i])
            t.make_move(m)
        if "get_board" in c:
            assert t.get_board().astype(np.int64).reshape(-1).tolist() == c["get_board"], c["name"]
            n_board += 1
        if "valid" in c:
            for a, v in zip(c["valid_probe"], c["valid"]):
                if v:
                    t2 = O.OracleTree(S)
                    for m in c["moves"]:
                        t2.make_move(m)
                    t2.make_move(a)  # must not raise
                else:
                    with pytest.raises(O.OracleError):
                        t.make_move(a)
    assert n_board >= 2

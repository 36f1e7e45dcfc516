#!/usr/bin/env python3
"""Generates tests/golden/rules_S<shape>.json from the REFERENCE's own State.

Runs only in the build container: it loads oracle/_ref/libref_state_<S>.so,
which oracle/Makefile (`make ref`) compiles from the reference's unmodified
mcts/src/state.cpp.  The JSON files it writes are data -- move lists and the
(player, move_cnt, winner) / get_board / is_valid answers the reference gave --
and are what travels; the reference itself never does.

    python3 tests/golden/make_rules_golden.py
"""
import ctypes as C
import json
import os
import random

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.path.join(HERE, "..", "..", "oracle", "_ref")
SHAPES = [5, 6, 7, 9, 13, 15, 19, 20]


def load(shape):
    L = C.PyDLL(os.path.join(REF, f"libref_state_{shape}.so"))
    assert L.ref_shape() == shape
    return L


def ref_play(L, moves, force=None):
    n = len(moves)
    mv = np.asarray(moves, np.int32)
    pl = np.zeros(n, np.int32)
    mc = np.zeros(n, np.int32)
    wi = np.zeros(n, np.float32)
    if force is None:
        k = L.ref_play(mv.ctypes.data_as(C.c_void_p), n, pl.ctypes.data_as(C.c_void_p),
                       mc.ctypes.data_as(C.c_void_p), wi.ctypes.data_as(C.c_void_p))
    else:
        fo = np.asarray(force, np.int32)
        k = L.ref_play_forced(mv.ctypes.data_as(C.c_void_p), fo.ctypes.data_as(C.c_void_p), n,
                              pl.ctypes.data_as(C.c_void_p), mc.ctypes.data_as(C.c_void_p),
                              wi.ctypes.data_as(C.c_void_p))
    return k, pl[:k].tolist(), mc[:k].tolist(), [float(x) for x in wi[:k]]


def case(L, name, moves, force=None, board=False, valid_probe=None):
    k, pl, mc, wi = ref_play(L, moves, force)
    c = dict(name=name, moves=[int(m) for m in moves], n_ok=k, player=pl, move_cnt=mc, winner=wi)
    if force is not None:
        c["force"] = [int(f) for f in force]
    if board and k == len(moves):
        S = L.ref_shape()
        out = np.zeros(2 * S * S, np.float32)
        mv = np.asarray(moves, np.int32)
        L.ref_get_board(mv.ctypes.data_as(C.c_void_p), len(moves), out.ctypes.data_as(C.c_void_p))
        c["get_board"] = [int(x) for x in out]
    if valid_probe is not None and k == len(moves):
        mv = np.asarray(moves, np.int32)
        c["valid_probe"] = [int(a) for a in valid_probe]
        c["valid"] = [int(L.ref_is_valid(mv.ctypes.data_as(C.c_void_p), len(moves), int(a)))
                      for a in valid_probe]
    return c


def interleave(first, second):
    """first player's stones and second player's stones -> alternating move list."""
    out = []
    for i in range(max(len(first), len(second))):
        if i < len(first):
            out.append(first[i])
        if i < len(second):
            out.append(second[i])
    return out


def scatter(S, avoid, n, rng, step=None):
    """n harmless cells for the other side: no two adjacent in any line."""
    cells = []
    cand = [(r, c) for r in range(0, S, 2) for c in range((r // 2) % 2 * 1, S, 3)]
    rng.shuffle(cand)
    for rc in cand:
        a = rc[0] * S + rc[1]
        if a in avoid:
            continue
        cells.append(a)
        if len(cells) == n:
            break
    assert len(cells) == n, "board too small for scatter"
    return cells


def clustered_game(S, rng, continue_after_end=0):
    """random legal moves biased next to existing stones; plays to the end."""
    size = S * S
    occ = set()
    moves = []
    for _ in range(size):
        if occ and rng.random() < 0.85:
            for _try in range(50):
                b = rng.choice(moves)
                r, c = divmod(b, S)
                r2, c2 = r + rng.randint(-1, 1), c + rng.randint(-1, 1)
                if 0 <= r2 < S and 0 <= c2 < S and r2 * S + c2 not in occ:
                    a = r2 * S + c2
                    break
            else:
                a = rng.choice([x for x in range(size) if x not in occ])
        else:
            a = rng.choice([x for x in range(size) if x not in occ])
        occ.add(a)
        moves.append(a)
    return moves


def gen(shape):
    rng = random.Random(1000 + shape)
    L = load(shape)
    S = shape
    size = S * S
    cases = []

    def line(r, c, dr, dc, n=5):
        return [(r + i * dr) * S + (c + i * dc) for i in range(n)]

    def add_line(name, cells, **kw):
        other = scatter(S, set(cells), len(cells), rng) if S >= 7 else None
        if other is None:
            return
        cases.append(case(L, name, interleave(cells, other), **kw))
        # same line for the second player: first player scatters one extra stone
        other2 = scatter(S, set(cells), len(cells), rng)
        cases.append(case(L, name + "_second", interleave(other2, cells), **kw))

    if S >= 7:
        add_line("row_top", line(0, 1, 0, 1), board=True)
        add_line("row_bottom_right", line(S - 1, S - 5, 0, 1))
        add_line("col_left", line(1, 0, 1, 0))
        add_line("col_right_bottom", line(S - 5, S - 1, 1, 0))
        add_line("diag_main", line(1, 1, 1, 1))
        add_line("diag_upper", line(0, S - 5, 1, 1))
        add_line("diag_lower", line(S - 5, 0, 1, 1))
        # "/" lines: row+col = k.  k <= S-1 is scanned, k >= S is the blind spot.
        add_line("anti_seen_k_eq_Sm1", line(0, S - 1, 1, -1))
        add_line("anti_seen_k_eq_4", line(0, 4, 1, -1))
        add_line("anti_seen_mid", line(1, S - 3, 1, -1))
        add_line("anti_blind_k_eq_S", line(1, S - 1, 1, -1))
        add_line("anti_blind_k_eq_Sp1", line(2, S - 1, 1, -1))
        add_line("anti_blind_corner", line(S - 5, S - 1, 1, -1))
        # overline: six in a row built so the sixth stone closes the gap
        six = line(2, 1, 0, 1, 6)
        order = [six[0], six[1], six[2], six[4], six[5], six[3]]
        add_line("overline_row", order)
        six = line(0, 0, 1, 1, 6)
        order = [six[0], six[1], six[3], six[4], six[5], six[2]]
        add_line("overline_diag", order)
    # move_cnt < 9 gate: force the same side to move five times in a row
    # (Cmcts::set_player allows it); the five stands at move_cnt 5..8 unseen and
    # is reported at the first move with move_cnt >= 9.
    if S >= 7:
        cells = line(0, 0, 0, 1)
        far = [(S - 1) * S + c for c in range(0, S, 2)][:5]
        moves = cells + far
        force = [0] * 5 + [1] * 5
        cases.append(case(L, "gate_move_cnt_lt_9_forced", moves, force=force))
        # both sides hold a five when the gate opens: scan order decides
        cells0 = line(0, 0, 0, 1)
        cells1 = line(S - 1, 0, 0, 1)
        moves = cells0[:4] + cells1[:4] + [cells0[4], cells1[4]]
        force = [0] * 4 + [1] * 4 + [0, 1]
        cases.append(case(L, "both_fives_forced_a", moves, force=force))
        cells0 = line(S - 1, 0, 0, 1)
        cells1 = line(0, 0, 0, 1)
        moves = cells0[:4] + cells1[:4] + [cells0[4], cells1[4]]
        cases.append(case(L, "both_fives_forced_b", moves, force=force))
        cells0 = line(0, S - 1, 1, 0)          # column for plane 0
        cells1 = line(1, 0, 1, 1)              # diagonal for plane 1
        moves = cells0[:4] + cells1[:4] + [cells0[4], cells1[4]]
        cases.append(case(L, "both_fives_forced_c", moves, force=force))
    # illegal moves
    cases.append(case(L, "illegal_occupied", [0, 1, 0, 2]))
    cases.append(case(L, "illegal_negative", [3, -1]))
    cases.append(case(L, "illegal_too_big", [3, size]))
    cases.append(case(L, "valid_probe", [0, size - 1, S], board=True,
                      valid_probe=[-1, 0, 1, S, size - 1, size, size + 5]))
    # random clustered games, played over the whole board (moves continue after
    # the first winner is set, as the reference allows)
    n_rand = 10 if S >= 13 else 16
    for i in range(n_rand):
        cases.append(case(L, f"clustered_{i}", clustered_game(S, rng), board=(i < 2)))
    # full-board draws on small boards: search random fills the reference calls a draw
    if S <= 7:
        found = 0
        tries = 0
        while found < 3 and tries < 200000:
            tries += 1
            perm = list(range(size))
            rng.shuffle(perm)
            k, pl, mc, wi = ref_play(L, perm)
            if k == size and wi[-1] == 0.5 and all(w == -1.0 for w in wi[:-1]):
                cases.append(case(L, f"full_board_draw_{found}", perm, board=True))
                found += 1
    return dict(shape=S, source="reference mcts/src/state.cpp compiled unmodified (oracle/_ref)",
                cases=cases)


def main():
    for s in SHAPES:
        g = gen(s)
        path = os.path.join(HERE, f"rules_S{s}.json")
        with open(path, "w") as f:
            json.dump(g, f, separators=(",", ":"))
        ends = sum(1 for c in g["cases"] if any(w != -1.0 for w in c["winner"]))
        draws = sum(1 for c in g["cases"] if any(w == 0.5 for w in c["winner"]))
        print(f"S={s}: {len(g['cases'])} cases, {ends} with a result, {draws} with a draw, "
              f"{os.path.getsize(path)} bytes")


if __name__ == "__main__":
    main()

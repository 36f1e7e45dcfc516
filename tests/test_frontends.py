"""The two small callers beside the arena: the "best parameters" pointer file
(src/tools.py:245-313, driven from src/eval.py:176-182) and the piskvork byte
protocol (src/server.py:37-57, src/pbrain-server.py:57-99)."""
import json
import os

import numpy as np
import pytest

from alphazero_amd import best_params as bp
from alphazero_amd import pbrain


def test_best_pointer_round_trip(tmp_path, monkeypatch):
    monkeypatch.delenv("DRY_RUN", raising=False)
    params = tmp_path / "params"
    params.mkdir()
    book = bp.BestParams(str(tmp_path / ".param_best"), str(params), "simplerNN13_*.pyt")
    book.init_file()
    assert book.get_latest() is None and book.get_best() is None  # nothing trained yet
    for name in ("simplerNN13_0001.pyt", "simplerNN13_0002.pyt", "other13_0009.pyt"):
        (params / name).write_bytes(b"x")
    assert book.get_latest() == str(params / "simplerNN13_0002.pyt")
    # an empty pointer adopts the latest file and records it in the reference's format
    assert book.get_best() == str(params / "simplerNN13_0002.pyt")
    assert json.load(open(book.pointer_file)) == {"path": str(params / "simplerNN13_0002.pyt")}
    book.set_best("simplerNN13_0001.pyt")  # a bare name is looked up inside the parameter directory
    assert book.get_best() == str(params / "simplerNN13_0001.pyt")
    with pytest.raises(RuntimeError, match="does not exist"):
        book.set_best("simplerNN13_0404.pyt")
    # a recorded file that moved: same name inside the parameter directory
    with open(book.pointer_file, "w") as f:
        json.dump({"path": "/somewhere/else/simplerNN13_0002.pyt"}, f)
    assert book.get_best() == os.path.join(str(params), "simplerNN13_0002.pyt")
    # the match verdict of src/eval.py:176-182
    started = []
    assert bp.conclude_match(book, "simplerNN13_0002.pyt", 4, 6, 0.6, lambda: started.append(1)) is True
    assert started == [1] and book.get_best() == str(params / "simplerNN13_0002.pyt")
    assert bp.conclude_match(book, "simplerNN13_0001.pyt", 5, 5, 0.6) is False
    assert book.get_best() == str(params / "simplerNN13_0002.pyt")
    with pytest.raises(ZeroDivisionError):
        bp.conclude_match(book, "simplerNN13_0001.pyt", 0, 0)
    monkeypatch.setenv("DRY_RUN", "1")
    book.set_best("simplerNN13_0001.pyt")  # dry run: the pointer stays
    assert book.get_best() == str(params / "simplerNN13_0002.pyt")


def test_piskvork_wire_format():
    assert pbrain.decode_request(bytes([pbrain.OP_MOVE, 7, 3])) == (pbrain.OP_MOVE, (7, 3))
    assert pbrain.decode_request(bytes([pbrain.MAKE_MOVE])) == (pbrain.MAKE_MOVE, None)
    assert pbrain.decode_request(bytes([pbrain.LOAD_MOVE, 1, 2, 1])) == (pbrain.LOAD_MOVE, (1, 2, 1))
    assert pbrain.decode_request(bytes([pbrain.INITIALIZE, 20])) == (pbrain.INITIALIZE, 20)
    assert pbrain.decode_request(b"") == (pbrain.END, None)
    assert pbrain.encode_move(3 * 15 + 7, 15) == bytes([7, 3])


class _Script:
    """a fake cmcts.mcts recording the calls of a session"""

    def __init__(self, size):
        self.size, self.calls = size, []

    def make_movexy(self, x, y):
        self.calls.append(("xy", x, y))

    def simulate(self, n):
        self.calls.append(("sim", n))

    def get_prob(self):
        pi = np.zeros(self.size, np.float32)
        pi[[31, 40]] = 0.5  # first maximum wins
        return pi

    def make_move(self, a):
        self.calls.append(("move", a))

    def clear(self):
        self.calls.append(("clear",))


def test_piskvork_session_dialogue():
    m = _Script(225)
    requests = iter([bytes([pbrain.INITIALIZE, 15]), bytes([pbrain.OP_MOVE, 7, 7]), bytes([pbrain.MAKE_MOVE]), b""])
    sent = []
    played = pbrain.play_session(m, lambda: next(requests), sent.append, sims=50, shape=15)
    assert m.calls == [("clear",), ("xy", 7, 7), ("sim", 50), ("move", 31)]
    assert played == [31] and sent == [bytes([31 % 15, 31 // 15])]
    with pytest.raises(NotImplementedError):
        pbrain.play_session(m, lambda: bytes([pbrain.LOAD_MOVE, 0, 0, 1]), sent.append, 1, 15)
    with pytest.raises(NotImplementedError):
        pbrain.play_session(m, lambda: bytes([9]), sent.append, 1, 15)


@pytest.mark.gpu
def test_piskvork_session_on_the_engine_over_tcp(tmp_path):
    """scripted opcodes through a real socket into the batch-of-one facade: make_movexy,
    simulate, argmax, make_move, the (x, y) reply; an occupied cell is refused."""
    import socket
    import threading

    import torch

    import alphazero_amd.cmcts as cmcts
    from alphazero_amd.model import PolicyValueNet

    S, sims = 15, 60
    torch.manual_seed(2)
    path = str(tmp_path / "net.pt")
    torch.jit.trace(PolicyValueNet(S).eval(), torch.rand(1, 2, S, S)).save(path)
    m = cmcts.mcts(cpuct=4.0, shape=S, nodes_per_tree=4096)
    m.set_alpha_default()
    m.set_params(path)
    m.eps = 0.25
    srv = pbrain.TcpServer("127.0.0.1", 0)
    replies = []

    def manager():
        c = socket.create_connection(("127.0.0.1", srv.port()))
        pause = threading.Event()
        c.sendall(bytes([pbrain.INITIALIZE, S]))
        for x, y in ((7, 7), (8, 8), (6, 8)):
            pause.wait(0.3)  # one request per read: the protocol has no framing (src/server.py:39)
            c.sendall(bytes([pbrain.OP_MOVE, x, y]))
            pause.wait(0.3)
            c.sendall(bytes([pbrain.MAKE_MOVE]))
            replies.append(tuple(c.recv(2)))
        c.close()

    th = threading.Thread(target=manager)
    th.start()
    srv.connect()
    played = pbrain.play_session(m, srv.recv, srv.send, sims, S)
    th.join()
    srv.close()
    assert len(played) == 3 and replies == [(a % S, a // S) for a in played]
    assert m.move_cnt == 6 and m.player == 0
    raw = m._eng.get_raw_board(0)
    assert [int(raw[0, y, x]) for x, y in ((7, 7), (8, 8), (6, 8))] == [1, 1, 1]  # the opponent opened: plane 0
    assert [int(raw[1, a // S, a % S]) for a in played] == [1, 1, 1]
    with pytest.raises(RuntimeError, match="Invalid move"):
        m.make_movexy(x=7, y=7)

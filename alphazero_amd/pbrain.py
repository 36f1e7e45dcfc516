"""Single-game front end in the shape of the reference's piskvork bridge
(src/pbrain-server.py:57-99 over src/server.py): a byte-opcode dialogue with the
Gomocup manager's proxy, played on one `cmcts.mcts` object (a one-tree engine).

Wire format (src/server.py:37-57): a request is up to 32 bytes, byte 0 the opcode,
then its arguments as single bytes; the reply to MAKE_MOVE is two bytes (x, y).
"""
import socket

import numpy as np

OP_MOVE, MAKE_MOVE, LOAD_MOVE, INITIALIZE, END = 1, 2, 3, 4, 5  # src/server.py:7-11


def decode_request(data):
    """bytes -> (opcode, argument) as Server.get_cmd does: an empty read ends the game."""
    if len(data) == 0:
        return END, None
    cmd = data[0]
    if cmd == LOAD_MOVE:
        return cmd, (data[1], data[2], data[3])
    if cmd == OP_MOVE:
        return cmd, (data[1], data[2])
    if cmd == INITIALIZE:
        return cmd, data[1]
    return cmd, None


def encode_move(action, shape):
    """reply to MAKE_MOVE: (x, y) = (action % S, action // S) (pbrain-server.py:78)"""
    return bytes(bytearray((int(action) % shape, int(action) // shape)))


def play_session(mcts, recv, send, sims, shape, log=None):
    """One connection's dialogue (pbrain-server.py:61-99).  recv() -> request bytes,
    send(bytes) delivers a reply.  The opponent's stones arrive as OP_MOVE (x, y) and
    go through make_movexy; MAKE_MOVE searches `sims` simulations, plays the most
    visited move and reports it; INITIALIZE clears the tree; END (or a closed
    connection) returns.  Returns the list of actions this side played."""
    played = []
    while True:
        cmd, arg = decode_request(recv())
        if cmd == OP_MOVE:
            mcts.make_movexy(x=arg[0], y=arg[1])
        elif cmd == MAKE_MOVE:
            mcts.simulate(sims)
            pi = mcts.get_prob()
            if log is not None:
                log(mcts._node_repr([]) if hasattr(mcts, "_node_repr") else "")
            move = int(np.argmax(pi))
            mcts.make_move(move)
            played.append(move)
            send(encode_move(move, shape))
        elif cmd == LOAD_MOVE:
            raise NotImplementedError  # as the reference (pbrain-server.py:82)
        elif cmd == INITIALIZE:
            mcts.clear()
        elif cmd == END:
            return played
        else:
            raise NotImplementedError("Error: unknow command: {}".format(cmd))


class TcpServer:
    """Listening socket of the bridge (src/server.py:13-35): 127.0.0.1:27015 by default."""

    def __init__(self, ip="127.0.0.1", port=27015):
        self.sock = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
        self.sock.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
        self.sock.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
        self.sock.bind((ip, port))
        self.sock.listen()
        self.connection = None

    def port(self):
        return self.sock.getsockname()[1]

    def connect(self):
        self.connection, _ = self.sock.accept()

    def recv(self):
        return self.connection.recv(32)

    def send(self, data):
        self.connection.sendall(data)

    def close(self):
        if self.connection is not None:
            self.connection.close()
            self.connection = None
        self.sock.close()


def serve(mcts, sims, shape, ip="127.0.0.1", port=27015, sessions=None):
    """The outer loop of pbrain-server.py:57-99: one dialogue per connection."""
    srv = TcpServer(ip, port)
    try:
        n = 0
        while sessions is None or n < sessions:
            srv.connect()
            play_session(mcts, srv.recv, srv.send, sims, shape)
            n += 1
    finally:
        srv.close()

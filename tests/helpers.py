"""Shared test helpers: an evaluator that is exact in integer arithmetic on both
sides (numpy for the CPU oracle, torch for the GPU engine), and reference-shaped
game loops driven by the oracle (the checker for the batched drivers)."""
import numpy as np

import az_oracle_py as O


def exact_eval_numpy(raw_board):
    """raw_board int8 [2,S,S] -> (value, prior[SIZE]); every step is integer
    arithmetic followed by a division by a power of two, hence identical in
    numpy and torch."""
    S = raw_board.shape[-1]
    size = S * S
    idx = np.arange(size, dtype=np.int64)
    b = raw_board.reshape(2, size).astype(np.int64)
    key = int((b[0] * (idx + 3)).sum() * 5 + (b[1] * (idx + 11)).sum() * 7 + b.sum() * 13)
    prior = (((idx * 31 + key) % 257) + 1).astype(np.float32) / np.float32(32768.0)
    value = np.float32(((key * 3 + 1) % 129) - 64) / np.float32(64.0)
    return float(value), prior


class ExactTorchEvaluator:
    """Same function on the GPU (linear priors)."""

    log_priors = False
    name = "exact"

    def __call__(self, boards):
        import torch

        B, _, S, _ = boards.shape
        size = S * S
        idx = torch.arange(size, dtype=torch.int64, device=boards.device)
        b = boards.reshape(B, 2, size).to(torch.int64)
        key = (b[:, 0] * (idx + 3)).sum(1) * 5 + (b[:, 1] * (idx + 11)).sum(1) * 7 + b.sum((1, 2)) * 13
        prior = (((idx[None, :] * 31 + key[:, None]) % 257) + 1).to(torch.float32) / 32768.0
        value = (((key * 3 + 1) % 129) - 64).to(torch.float32) / 64.0
        return value.contiguous(), prior.contiguous()


def oracle_exact_evaluator(shape):
    return O.callback_evaluator(lambda b, player: exact_eval_numpy(b), shape, log_priors=False)


def oracle_self_play_game(t0, t1, sims, tau, ev, rng, noise=None):
    """The per-move protocol of the reference's self_play_game
    (src/self_play.py:155-200) on two oracle trees; returns (records, n_first)
    with records = list of (board, pi, r)."""
    size = t0.size
    d0, d1 = [], []
    for i in range(size):
        for searcher, mine in ((t0, d0), (t1, d1)):
            searcher.simulate(sims, ev, noise)
            pi = searcher.get_prob()
            board = searcher.get_board()
            if i > tau:
                move = int(pi.argmax())
                pi[:] = 0
                pi[move] = 1
            else:
                # numpy.random.choice(p=pi) for the stream's next uniform number
                cdf = np.cumsum(pi.astype(np.float64))
                cdf /= cdf[-1]
                move = int(np.searchsorted(cdf, rng.random(), side="right"))
            mine.append([board, pi.copy(), -1.0])
            t0.make_move(move)
            t1.make_move(move)
            if t0.winner != -1:
                break
        if t0.winner != -1:
            break
    if t0.winner == 0.0:
        for r in d0:
            r[2] = 1.0
    elif t0.winner == 1.0:
        for r in d1:
            r[2] = 1.0
    else:
        for r in d0 + d1:
            r[2] = 1e-3
    return d0 + d1, len(d0)


def oracle_eval_game(t_first, sims_first, ev_first, t_second, sims_second, ev_second, noise=None):
    """eval_game (src/eval.py:187-209) on two oracle trees."""
    for _ in range(t_first.size):
        for searcher, sims, ev in ((t_first, sims_first, ev_first), (t_second, sims_second, ev_second)):
            searcher.simulate(sims, ev, noise)
            move = int(searcher.get_prob().argmax())
            t_first.make_move(move)
            t_second.make_move(move)
            if t_first.winner != -1:
                return


class LegacyStream:
    """The uniform numbers numpy's legacy global generator hands to
    np.random.choice after np.random.seed(seed) (one per sampled move), behind
    the `.random()` name the drivers call."""

    def __init__(self, seed):
        self.rs = np.random.RandomState(seed)

    def random(self, n=None):
        return self.rs.random_sample(n)


def replay_board(shape, moves):
    """raw planes [2,S,S] after the alternating move list (first mover on plane 0)."""
    b = np.zeros((2, shape * shape), np.float32)
    for k, m in enumerate(moves):
        b[k % 2, m] = 1
    return b.reshape(2, shape, shape)

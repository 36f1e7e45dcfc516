"""Diagnostic: arena tallies of two random-init networks under different noise seeds and batch widths."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from alphazero_amd import selfplay as sp
from alphazero_amd.evaluator import FusedNetEvaluator
from alphazero_amd.model import PolicyValueNet

S, sims = 15, 100
nets = []
for seed in (0, 1):
    torch.manual_seed(seed)
    nets.append(FusedNetEvaluator(PolicyValueNet(S).eval(), S, 0, name=f"seed{seed}"))
for G, seeds, eps in ((2048, (0, 1000), 0.25), (2048, (0, 1), 0.25), (2048, (2, 3), 0.25), (2048, (7, 1234567), 0.25), (2048, (0, 1000), 0.0), (512, (0, 1000), 0.25)):
    b = sp.TreeBatch(G, S, nets[0], cpuct=4.0, eps=eps, nodes_per_tree=364, noise_seed=seeds[0])
    l = sp.TreeBatch(G, S, nets[1], cpuct=4.0, eps=eps, nodes_per_tree=364, noise_seed=seeds[1])
    first = (np.arange(G) % (S * S)).astype(np.int32)
    print(G, seeds, eps, sp.eval_games_resident(b, l, sims, sims, n_games=G, game_openings=first), flush=True)
    b.close(); l.close()

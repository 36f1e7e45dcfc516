"""Cycle stamps of the network kernel (diagnostic build libazmcts_nst.so, -DAZ_NET_STAMPS):
where a wavefront's time goes per position."""
import os, sys
ROOT = '/root/repo'
sys.path.insert(0, ROOT)
import numpy as np, torch
from alphazero_amd import engine as E
E.LIB_PATH = os.path.join(ROOT, 'alphazero_amd', sys.argv[1] if len(sys.argv) > 1 else 'libazmcts_nst.so')
from alphazero_amd.evaluator import FusedNetEvaluator
from alphazero_amd.model import PolicyValueNet
S = int(sys.argv[2]) if len(sys.argv) > 2 else 15
NW = int(sys.argv[3]) if len(sys.argv) > 3 else 2
torch.manual_seed(0)
ev = FusedNetEvaluator(PolicyValueNet(S).eval(), S)
NPOS = int(sys.argv[5]) if len(sys.argv) > 5 else 8192
b = (torch.rand(NPOS, 2, S, S, device='cuda') < 0.2).float()
for _ in range(3):
    v, lp = ev(b)
torch.cuda.synchronize()
grid = int(sys.argv[4]) if len(sys.argv) > 4 else 1024
raw = np.ascontiguousarray(lp[:grid].cpu().numpy().view(np.uint8).reshape(grid, -1)[:, :NW * 128]).view(np.uint64).reshape(grid, NW, 16).astype(np.float64)
n = raw[:, :, 7]
names = ["input load", "conv16 matrix loops", "epilogues", "heads (value, policy)", "restore + layer setup", "TOTAL", "first layer matrix loop", "positions", "map", "value head", "policy mma", "policy store", "gather", "softmax", "x", "y"]
per = raw / n[:, :, None]
for k, nm in enumerate(names):
    print(f"{nm:22s} mean {per[:, :, k].mean():10.0f}  per wave {[int(x) for x in per[:, :, k].mean(axis=0)]}")
print("sum of parts", per[:, :, [0, 1, 2, 3, 4, 6]].sum(axis=2).mean())

import sys, os, statistics, subprocess
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT)
if len(sys.argv) > 2 and sys.argv[1] == 'child':
    import torch, numpy as np
    from alphazero_amd import engine as E
    E.LIB_PATH = os.path.join(ROOT, 'alphazero_amd', sys.argv[2])
    from alphazero_amd.evaluator import FusedNetEvaluator
    from alphazero_amd.model import PolicyValueNet
    S = int(sys.argv[3])
    torch.manual_seed(0)
    ev = FusedNetEvaluator(PolicyValueNet(S).eval(), S)
    b = (torch.rand(8192, 2, S, S, device='cuda') < 0.2).float()
    for _ in range(3): ev(b)
    torch.cuda.synchronize()
    ts = []
    for _ in range(20):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ev(b); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1))
    print(statistics.median(ts))
else:
    for S in (tuple(int(x) for x in os.environ["AB_SHAPES"].split(",")) if "AB_SHAPES" in os.environ else (15, 20)):
        for lib in sys.argv[1:]:
            out = subprocess.run([sys.executable, __file__, 'child', lib, str(S)], capture_output=True, text=True)
            print(S, lib, out.stdout.strip().splitlines()[-1] if out.returncode == 0 else out.stderr[-300:])

"""Batched policy/value evaluators for the stepped search path.

An evaluator is a callable  boards[B,2,S,S] float32 (cuda, raw un-swapped
planes, the tensor of mcts/src/cmcts.cpp:206-210)  ->  (value[B] float32,
prior[B,SIZE] float32), both contiguous on the same device, plus the attribute
`log_priors` (True when `prior` holds log-probabilities, as the reference's
network returns, src/model.py:41).  The reference calls its network with batch
1 once per expansion (cmcts.cpp:217); here one call serves every tree of the
engine for one search step.
"""
import torch


class TorchEvaluator:
    """Wraps a torch.nn.Module / ScriptModule returning (value, log_softmax policy)."""

    log_priors = True

    def __init__(self, module, device=0, name="module"):
        self.device = torch.device("cuda", device) if not isinstance(device, torch.device) else device
        self.module = module.to(self.device).eval()
        self.name = name

    @torch.no_grad()
    def __call__(self, boards):
        value, logp = self.module(boards)
        return (value.reshape(-1).to(torch.float32).contiguous(),
                logp.reshape(boards.shape[0], -1).to(torch.float32).contiguous())


class TorchScriptEvaluator(TorchEvaluator):
    """The reference hands `cmcts` a traced TorchScript file (src/self_play.py:62-72,
    loaded at cmcts.cpp:137); this loads the same file once, onto the GPU."""

    def __init__(self, path, device=0):
        dev = torch.device("cuda", device)
        super().__init__(torch.jit.load(path, map_location=dev), dev, name=path)


class ConstantEvaluator:
    """Uniform prior / fixed value (linear priors): the tree-kernel isolation evaluator."""

    log_priors = False

    def __init__(self, size, prior=None, value=0.0, name="constant"):
        self.size, self.prior, self.value, self.name = size, (1.0 / size if prior is None else prior), value, name
        self._cache = {}

    def __call__(self, boards):
        key = (boards.shape[0], boards.device)
        if key not in self._cache:
            self._cache[key] = (torch.full((boards.shape[0],), self.value, dtype=torch.float32, device=boards.device),
                                torch.full((boards.shape[0], self.size), self.prior, dtype=torch.float32,
                                           device=boards.device))
        return self._cache[key]

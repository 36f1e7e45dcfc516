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


class FusedNetEvaluator:
    """The reference network (src/model.py:9-44, inference mode) evaluated by the
    engine library's own kernel (az_net_eval: f32 matrix-core convolutions with
    the activations resident in LDS) instead of PyTorch.  Takes a module with the
    reference's sub-module names (`front`, `policy_conv`, `policy`, `value_conv`,
    `value`; 2 input planes, 16 conv channels) or its state_dict; PyTorch only
    provides the tensors."""

    log_priors = True

    def __init__(self, module_or_state, shape, device=0, name="fused-net"):
        from . import engine as _E

        self.L = _E.load_library()
        self.device = torch.device("cuda", device) if not isinstance(device, torch.device) else device
        self.shape, self.name = shape, name
        state = module_or_state.state_dict() if hasattr(module_or_state, "state_dict") else module_or_state
        packed, self.layers = self.pack(state)
        assert packed.numel() == self.L.az_net_param_floats(*self.layers)
        self.params = packed.to(self.device)
        # (front, policy, value) layer counts: marks the evaluator as one the engine can
        # drive by itself (az_simulate_net), without a Python round trip per search step
        self.fused_layers = self.layers

    @staticmethod
    def supports(state):
        """True when `state` (a state_dict) is the reference architecture this kernel is built for."""
        try:
            w0 = state["front.0.weight"]
            return (tuple(w0.shape[:2]) == (16, 2) and tuple(w0.shape[2:]) == (3, 3)
                    and tuple(state["policy.0.weight"].shape) == (1, 16, 3, 3)
                    and tuple(state["value.0.weight"].shape) == (1, 16, 1, 1)
                    and "front.1.running_mean" in state)
        except KeyError:
            return False

    @staticmethod
    def _fold(state, prefix, i):
        """conv weight + inference batch norm of block i of a Sequential (conv at 3i, bn at 3i+1)."""
        w = state[f"{prefix}.{3 * i}.weight"].detach().double().cpu()
        g, b = state[f"{prefix}.{3 * i + 1}.weight"].double().cpu(), state[f"{prefix}.{3 * i + 1}.bias"].double().cpu()
        mean, var = state[f"{prefix}.{3 * i + 1}.running_mean"].double().cpu(), state[f"{prefix}.{3 * i + 1}.running_var"].double().cpu()
        scale = g / torch.sqrt(var + 1e-5)  # nn.BatchNorm2d default eps
        return w, scale, b - mean * scale

    @classmethod
    def pack(cls, state):
        def count(prefix):
            n = 0
            while f"{prefix}.{3 * n}.weight" in state:
                n += 1
            return n

        front, pol, val = count("front"), count("policy_conv"), count("value_conv")
        lane = torch.arange(64)
        co, kq = lane & 15, lane >> 4
        out = []
        w, sc, sh = cls._fold(state, "front", 0)
        if w.shape[:2] != (16, 2):
            raise ValueError("the fused evaluator is built for 2 input planes and 16 conv channels")
        b = torch.zeros(5, 64, dtype=torch.float64)
        for ks in range(5):
            kk = ks * 4 + kq
            ok = kk < 18
            tap, ci = torch.where(ok, kk >> 1, 0), kk & 1
            b[ks] = torch.where(ok, w[co, ci, tap // 3, tap % 3] * sc[co], torch.zeros(64, dtype=torch.float64))
        # (the inference batch-norm scale is folded into the weights; the kernels add the shift only,
        # the scale slot of the block stays for layout compatibility and holds ones)
        out += [b.reshape(-1), torch.ones_like(sc), sh]
        blocks = [("front", i) for i in range(1, front)] + [("policy_conv", i) for i in range(pol)] + \
                 [("value_conv", i) for i in range(val)]
        for prefix, i in blocks:
            w, sc, sh = cls._fold(state, prefix, i)
            if w.shape[:2] != (16, 16):
                raise ValueError("the fused evaluator is built for 16 conv channels")
            b = torch.zeros(36, 64, dtype=torch.float64)
            for tap in range(9):
                for g in range(4):
                    b[tap * 4 + g] = w[co, 4 * g + kq, tap // 3, tap % 3] * sc[co]
            out += [b.reshape(-1), torch.ones_like(sc), sh]
        wp = state["policy.0.weight"].detach().double().cpu()  # [1,16,3,3]
        out += [wp[0].permute(1, 2, 0).reshape(-1), state["policy.0.bias"].detach().double().cpu().reshape(-1)]
        wv = state["value.0.weight"].detach().double().cpu()   # [1,16,1,1]
        out += [wv.reshape(-1), state["value.0.bias"].detach().double().cpu().reshape(-1)]
        # The same 16 -> 16 layers in the Winograd F(2x2, 3x3) domain, for the one-wavefront kernel:
        # U = G g Gt per (output, input) channel pair, stored as the B operand of the matrix
        # instruction for transformed position x = 4*xi + nu and K step ks: lane (co, kq) holds
        # U[co, 4*ks + kq, xi, nu]; then the batch-norm shift.
        gm = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float64)
        for prefix, i in blocks:
            w, sc, sh = cls._fold(state, prefix, i)
            uu = gm @ (w * sc.view(-1, 1, 1, 1)) @ gm.T    # [co, ci, 4, 4]
            b = torch.zeros(16, 4, 64, dtype=torch.float64)
            for x in range(16):
                for ks in range(4):
                    b[x, ks] = uu[co, 4 * ks + kq, x // 4, x % 4]
            out += [b.reshape(-1), sh]
        # the input layer in the same form: one K step, K index = input plane, zero for K indices 2..3
        w, sc, sh = cls._fold(state, "front", 0)
        uu = gm @ (w * sc.view(-1, 1, 1, 1)) @ gm.T        # [co, 2, 4, 4]
        b = torch.zeros(16, 64, dtype=torch.float64)
        for x in range(16):
            b[x] = torch.where(kq < 2, uu[co, kq & 1, x // 4, x % 4], torch.zeros(64, dtype=torch.float64))
        out += [b.reshape(-1), sh]
        # Both again in the Winograd F(4x4, 3x3) domain (6x6 transformed positions x = 6*xi + nu), K step
        # outermost: lane (co, kq) holds U[co, 4*ks + kq, xi, nu]; positions in pairs per lane
        # ([ks][x // 2][lane][x % 2]: one 8-byte load brings two matrix instructions' operands); then the shift.
        gm4 = torch.tensor([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6],
                            [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]], dtype=torch.float64)
        xi, nu = torch.arange(36) // 6, torch.arange(36) % 6
        for prefix, i in blocks:
            w, sc, sh = cls._fold(state, prefix, i)
            uu = gm4 @ (w * sc.view(-1, 1, 1, 1)) @ gm4.T  # [co, ci, 6, 6]
            b = torch.zeros(4, 36, 64, dtype=torch.float64)
            for ks in range(4):
                b[ks] = uu[co.view(1, 64), (4 * ks + kq).view(1, 64), xi.view(36, 1), nu.view(36, 1)]
            out += [b.view(4, 18, 2, 64).permute(0, 1, 3, 2).reshape(-1), sh]
        w, sc, sh = cls._fold(state, "front", 0)
        uu = gm4 @ (w * sc.view(-1, 1, 1, 1)) @ gm4.T      # [co, 2, 6, 6]
        b = uu[co.view(1, 64), (kq & 1).view(1, 64), xi.view(36, 1), nu.view(36, 1)] * (kq < 2).view(1, 64)
        out += [b.view(18, 2, 64).permute(0, 2, 1).reshape(-1), sh]
        return torch.cat(out).to(torch.float32).contiguous(), (front, pol, val)

    takes_leaf_kinds = True

    def __call__(self, boards, leaf_kinds=None):
        """leaf_kinds: the int32 codes az_select wrote for this step (device tensor);
        boards that need no evaluation are skipped, their outputs are undefined."""
        n, size = boards.shape[0], self.shape * self.shape
        boards = boards.contiguous()
        value = torch.empty(n, dtype=torch.float32, device=boards.device)
        logp = torch.empty((n, size), dtype=torch.float32, device=boards.device)
        rc = self.L.az_net_eval(self.shape, n, boards.data_ptr(),
                                None if leaf_kinds is None else leaf_kinds.data_ptr(), self.params.data_ptr(), *self.layers,
                                value.data_ptr(), logp.data_ptr(), torch.cuda.current_stream(boards.device).cuda_stream)
        if rc != 0:
            raise RuntimeError(f"az_net_eval failed (code {rc})")
        return value, logp


def load_evaluator(path, shape, device=0):
    """Evaluator for the TorchScript file the reference hands to cmcts
    (src/self_play.py:62-72): the fused kernel when the file holds the reference
    architecture, the PyTorch module otherwise."""
    dev = torch.device("cuda", device) if not isinstance(device, torch.device) else device
    module = torch.jit.load(path, map_location=dev)
    state = module.state_dict()
    if FusedNetEvaluator.supports(state):
        return FusedNetEvaluator(state, shape, dev, name=path)
    return TorchEvaluator(module, dev, name=path)

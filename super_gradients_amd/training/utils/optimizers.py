"""Arena optimizers: AdamW / SGD over the network's flat parameter arena - one kernel launch per step.

Reference behaviour mirrored: torch.optim.AdamW / SGD as built by training/utils/optimizer_utils.py:88-143 with the
zero-weight-decay grouping of :32-59 (BatchNorm affine parameters and every bias get weight_decay 0 when
`zero_weight_decay_on_bias_and_bn`).  They subclass torch.optim.Optimizer, so LR callbacks that write
`param_group["lr"]` (callbacks.py:374-392, 489-514) and `state_dict()` checkpointing keep working.
Dead parameters (QARepVGGBlock.rbr_reparam) are not in the arena and are never touched - torch.optim skips them too,
because they never receive a gradient.
"""
import torch

from ... import kernels as K
from ...common.registry import register_optimizer
from ...modules.engine import SgxNetwork


def _segments(net: SgxNetwork, weight_decay: float, zero_wd_on_bias_bn: bool):
    ends, wds = [], []
    for i, s in enumerate(net.slots):
        end = net.slots[i + 1].start if i + 1 < len(net.slots) else net.p_arena.size
        wd = 0.0 if (zero_wd_on_bias_bn and s.no_wd) else float(weight_decay)
        if wds and wds[-1] == wd:
            ends[-1] = end
        else:
            ends.append(end)
            wds.append(wd)
    dev = net.p_arena.buf.device
    return torch.tensor(ends, dtype=torch.int64, device=dev), torch.tensor(wds, dtype=torch.float32, device=dev)


class _ArenaOptimizer(torch.optim.Optimizer):
    def __init__(self, net: SgxNetwork, defaults: dict, zero_weight_decay_on_bias_and_bn: bool):
        if not isinstance(net, SgxNetwork):
            raise TypeError("arena optimizers take the network itself (an SgxNetwork), not parameter lists: the step is one kernel over its arena")
        net.materialize()
        self.net = net
        decay = [s.param for s in net.slots if not (zero_weight_decay_on_bias_and_bn and s.no_wd)]
        no_decay = [s.param for s in net.slots if zero_weight_decay_on_bias_and_bn and s.no_wd]
        groups = [{"named_params": None, "params": decay, "name": "decay"}]
        if no_decay:
            groups.insert(0, {"params": no_decay, "weight_decay": 0.0, "name": "no_decay"})
        for g in groups:
            g.pop("named_params", None)
        super().__init__(groups, defaults)
        self._zero_wd = zero_weight_decay_on_bias_and_bn
        self._seg = None
        self._seg_wd_value = None
        self._steps = 0

    def _lr_wd(self):
        lrs = {float(g["lr"]) for g in self.param_groups}
        if len(lrs) != 1:
            raise NotImplementedError("per-group learning rates are not supported by the arena optimizers (one launch over the whole arena)")
        wd = float([g for g in self.param_groups if g.get("name") == "decay"][0]["weight_decay"])
        if self._seg is None or self._seg_wd_value != wd:
            self._seg = _segments(self.net, wd, self._zero_wd)
            self._seg_wd_value = wd
        return lrs.pop(), self._seg

    def zero_grad(self, set_to_none: bool = False):
        self.net.zero_grad()


@register_optimizer("AdamW")
class ArenaAdamW(_ArenaOptimizer):
    def __init__(self, net, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, zero_weight_decay_on_bias_and_bn=False):
        super().__init__(net, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay), zero_weight_decay_on_bias_and_bn)
        n = net.p_arena.buf.numel()
        self.exp_avg = torch.zeros(n, device=net.p_arena.buf.device)
        self.exp_avg_sq = torch.zeros(n, device=net.p_arena.buf.device)

    @torch.no_grad()
    def step(self, closure=None, grad_scale=None):
        lr, (seg_end, seg_wd) = self._lr_wd()
        b1, b2 = self.param_groups[0]["betas"]
        self._steps += 1
        K.adamw_step(self.net.p_arena.buf, self.net.g_arena.buf, self.exp_avg, self.exp_avg_sq, lr, b1, b2, self.param_groups[0]["eps"], self._steps,
                     seg_end, seg_wd, grad_scale)


@register_optimizer("SGD")
class ArenaSGD(_ArenaOptimizer):
    def __init__(self, net, lr=1e-3, momentum=0.0, dampening=0.0, weight_decay=0.0, nesterov=False, zero_weight_decay_on_bias_and_bn=False):
        super().__init__(net, dict(lr=lr, momentum=momentum, dampening=dampening, weight_decay=weight_decay, nesterov=nesterov),
                         zero_weight_decay_on_bias_and_bn)
        self.momentum_buffer = torch.zeros(net.p_arena.buf.numel(), device=net.p_arena.buf.device)

    @torch.no_grad()
    def step(self, closure=None):
        lr, (seg_end, seg_wd) = self._lr_wd()
        g = self.param_groups[0]
        self._steps += 1
        K.sgd_step(self.net.p_arena.buf, self.net.g_arena.buf, self.momentum_buffer, lr, g["momentum"], g["dampening"], g["nesterov"],
                   self._steps == 1, seg_end, seg_wd)


def build_optimizer(net, lr: float, training_params) -> torch.optim.Optimizer:
    """optimizer_utils.py:88-143: `optimizer` is a name ("AdamW", "SGD") with `optimizer_params`, `zero_weight_decay_on_bias_and_bn`."""
    from .utils import get_param

    name = get_param(training_params, "optimizer", "SGD")
    if not isinstance(name, str):
        return name  # an already-built optimizer
    zero = bool(get_param(training_params, "zero_weight_decay_on_bias_and_bn", False))
    cls = {"adamw": ArenaAdamW, "sgd": ArenaSGD}.get(name.lower())
    if cls is None:
        raise NotImplementedError(f"optimizer '{name}' is not available on the HIP path (AdamW, SGD)")
    # optimizer_utils.py:23-28,104-106: the recipe's optimizer_params are laid over per-optimizer defaults (SGD: weight decay 1e-4 and momentum
    # 0.9 - not torch's zeros; AdamW has no entry there and keeps torch's own defaults), and the merged dictionary is written back
    params = dict(OPTIMIZERS_DEFAULT_PARAMS.get(cls, {}))
    params.update(get_param(training_params, "optimizer_params", {}) or {})
    if hasattr(training_params, "override"):
        training_params.override(optimizer_params=dict(params))
    elif isinstance(training_params, dict):
        training_params["optimizer_params"] = dict(params)
    return cls(net, lr=lr, zero_weight_decay_on_bias_and_bn=zero, **params)


OPTIMIZERS_DEFAULT_PARAMS = {ArenaSGD: {"weight_decay": 1e-4, "momentum": 0.9}}  # training/params.py:88

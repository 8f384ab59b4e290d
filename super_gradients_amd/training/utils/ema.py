"""ModelEMA over the arenas (reference: training/utils/ema.py:20-141, ema_decay_schedules.py:21-55).

The reference deep-copies the model and lerps all 921 floating state_dict tensors one by one (3 ATen kernels each);
here the EMA state is a copy of the parameter arena and of the BN-buffer arena, updated by two launches:
    ema = ema * d + (1 - d) * model,   d = decay_function(decay, step, total_steps).
`with ema.averaged():` evaluates the same network on the averaged weights (arena swap, no model copy).
"""
import contextlib
import math

import torch

from ... import kernels as K
from ...modules.engine import SgxNetwork


class ConstantDecay:
    def __init__(self, **kwargs):
        pass

    def __call__(self, decay, step, total_steps):
        return decay


class ThresholdDecay:
    def __init__(self, **kwargs):
        pass

    def __call__(self, decay, step, total_steps):
        return min(decay, (1 + step) / (10 + step))


class ExpDecay:
    def __init__(self, beta: float, **kwargs):
        self.beta = beta

    def __call__(self, decay, step, total_steps):
        return decay * (1 - math.exp(-(step / total_steps) * self.beta))


EMA_DECAY_FUNCTIONS = {"constant": ConstantDecay, "threshold": ThresholdDecay, "exp": ExpDecay}


class ModelEMA:
    def __init__(self, model: SgxNetwork, decay: float, decay_function):
        if not isinstance(model, SgxNetwork):
            raise TypeError("ModelEMA on the HIP path averages an SgxNetwork's arenas")
        model.materialize()
        self.model = model
        self.decay, self.decay_function = decay, decay_function
        self.p_ema = model.p_arena.buf.clone()
        self.b_ema = model.b_arena.buf.clone()

    @classmethod
    def from_params(cls, model, decay_type: str = None, decay: float = None, **kwargs):
        decay = 0.9999 if decay is None else decay
        if decay_type is None:
            decay_type = "exp"
            kwargs.setdefault("beta", 15)
        if decay_type not in EMA_DECAY_FUNCTIONS:
            from ...common.factories import UnknownTypeException

            raise UnknownTypeException(decay_type, list(EMA_DECAY_FUNCTIONS.keys()))
        return cls(model, decay, EMA_DECAY_FUNCTIONS[decay_type](**kwargs))

    @torch.no_grad()
    def update(self, model, step: int, total_steps: int):
        d = float(self.decay_function(self.decay, step, total_steps))
        K.ema_update(self.p_ema, self.model.p_arena.buf, d)
        K.ema_update(self.b_ema, self.model.b_arena.buf, d)

    @contextlib.contextmanager
    def averaged(self):
        """`with ema.averaged(): validate(model)` - runs the SAME network on the averaged weights (arena contents are
        swapped in and out; no second copy of the model).  This is what Trainer uses where the reference evaluates
        `ema_model.ema`."""
        m = self.model
        keep_p, keep_b = m.p_arena.buf.clone(), m.b_arena.buf.clone()
        m.p_arena.buf.copy_(self.p_ema)
        m.b_arena.buf.copy_(self.b_ema)
        m.weights_changed()  # folded eval filters (prep_model_for_conversion) must not outlive the swap
        try:
            yield m
        finally:
            m.p_arena.buf.copy_(keep_p)
            m.b_arena.buf.copy_(keep_b)
            m.weights_changed()

    def state_dict(self):
        """state_dict of the averaged network (same keys as the model's), e.g. for the checkpoint's `ema_net` entry."""
        with self.averaged() as m:
            return {k: v.detach().clone() for k, v in m.state_dict().items()}

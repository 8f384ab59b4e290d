"""Phase callbacks and the hard-coded learning-rate schedules of the train-step driver.

Reference: training/utils/callbacks/base_callbacks.py (Phase :13-23, PhaseContext :28-135, Callback :138-470,
CallbackHandler :473-880) and training/utils/callbacks/callbacks.py (LRCallbackBase :230-268, LinearEpochLRWarmup :271-314,
LinearBatchLRWarmup :317-392, StepLRScheduler :395-430, ExponentialLRScheduler :433-452, PolyLRScheduler :455-477,
CosineLRScheduler :480-514, FunctionLRScheduler :517-543).  Host-side scalar arithmetic only (fp64 numpy like the
reference); the schedules write `param_group["lr"]`, which the arena optimizers read at every step.
"""
import math
import numbers
from collections.abc import Mapping
from enum import Enum
from typing import List

import numpy as np

from ...common.registry import register_callback, register_lr_scheduler, register_lr_warmup


class Phase(Enum):
    PRE_TRAINING = "PRE_TRAINING"
    TRAIN_EPOCH_START = "TRAIN_EPOCH_START"
    TRAIN_BATCH_END = "TRAIN_BATCH_END"
    TRAIN_BATCH_STEP = "TRAIN_BATCH_STEP"
    TRAIN_EPOCH_END = "TRAIN_EPOCH_END"
    VALIDATION_BATCH_END = "VALIDATION_BATCH_END"
    VALIDATION_EPOCH_END = "VALIDATION_EPOCH_END"
    VALIDATION_END_BEST_EPOCH = "VALIDATION_END_BEST_EPOCH"
    POST_TRAINING = "POST_TRAINING"


class PhaseContext:
    """Attribute bag handed to the callbacks (base_callbacks.py:28-135)."""

    def __init__(self, **kwargs):
        self.epoch = self.batch_idx = None
        self.optimizer = self.net = self.criterion = self.inputs = self.preds = self.target = None
        self.metrics_dict = self.metrics_compute_fn = self.loss_avg_meter = self.loss_log_items = None
        self.experiment_name = self.ckpt_dir = self.lr_warmup_epochs = self.sg_logger = None
        self.train_loader = self.valid_loader = self.training_params = self.ddp_silent_mode = self.checkpoint_params = None
        self.architecture = self.arch_params = self.metric_to_watch = self.valid_metrics = self.ema_model = None
        self.loss_logging_items_names = self.additional_batch_items = self.stop_training = None
        self.update_context(**kwargs)

    def update_context(self, **kwargs):
        for k, v in kwargs.items():
            setattr(self, k, v)


class Callback:
    """Event interface (base_callbacks.py:138-470); every hook is optional."""

    def on_training_start(self, context): pass  # noqa: E704
    def on_train_loader_start(self, context): pass  # noqa: E704
    def on_train_batch_start(self, context): pass  # noqa: E704
    def on_train_batch_loss_end(self, context): pass  # noqa: E704
    def on_train_batch_backward_end(self, context): pass  # noqa: E704
    def on_train_batch_gradient_step_start(self, context): pass  # noqa: E704
    def on_train_batch_gradient_step_end(self, context): pass  # noqa: E704
    def on_train_batch_end(self, context): pass  # noqa: E704
    def on_train_loader_end(self, context): pass  # noqa: E704
    def on_validation_loader_start(self, context): pass  # noqa: E704
    def on_validation_batch_start(self, context): pass  # noqa: E704
    def on_validation_batch_end(self, context): pass  # noqa: E704
    def on_validation_loader_end(self, context): pass  # noqa: E704
    def on_validation_end_best_epoch(self, context): pass  # noqa: E704
    def on_training_end(self, context): pass  # noqa: E704


class PhaseCallback(Callback):
    """A callback bound to ONE phase through `__call__` (base_callbacks.py:884-960)."""

    def __init__(self, phase: Phase):
        self.phase = phase

    def __call__(self, context, **kwargs):
        raise NotImplementedError

    def _fire(self, phase, context):
        if self.phase == phase:
            self(context)

    def on_training_start(self, context): self._fire(Phase.PRE_TRAINING, context)  # noqa: E704
    def on_train_loader_start(self, context): self._fire(Phase.TRAIN_EPOCH_START, context)  # noqa: E704
    def on_train_batch_loss_end(self, context): self._fire(Phase.TRAIN_BATCH_END, context)  # noqa: E704
    def on_train_batch_gradient_step_end(self, context): self._fire(Phase.TRAIN_BATCH_STEP, context)  # noqa: E704
    def on_train_loader_end(self, context): self._fire(Phase.TRAIN_EPOCH_END, context)  # noqa: E704
    def on_validation_batch_end(self, context): self._fire(Phase.VALIDATION_BATCH_END, context)  # noqa: E704
    def on_validation_loader_end(self, context): self._fire(Phase.VALIDATION_EPOCH_END, context)  # noqa: E704
    def on_validation_end_best_epoch(self, context): self._fire(Phase.VALIDATION_END_BEST_EPOCH, context)  # noqa: E704
    def on_training_end(self, context): self._fire(Phase.POST_TRAINING, context)  # noqa: E704


class CallbackHandler(Callback):
    def __init__(self, callbacks: List[Callback]):
        self.callbacks = list(callbacks)

    def _all(self, name, context):
        for cb in self.callbacks:
            getattr(cb, name)(context)


for _n in [n for n in dir(Callback) if n.startswith("on_")]:
    setattr(CallbackHandler, _n, (lambda name: lambda self, context: self._all(name, context))(_n))


# ------------------------------------------------------------------------------------------------ LR schedules
@register_callback("LRCallbackBase")
class LRCallbackBase(PhaseCallback):
    def __init__(self, phase, initial_lr, update_param_groups, train_loader_len, net, training_params, **kwargs):
        super().__init__(phase)
        if not isinstance(initial_lr, dict):
            initial_lr = {"default": float(initial_lr)}
        self.initial_lr = initial_lr
        self.lr = dict(initial_lr)
        self.update_param_groups = update_param_groups
        self.train_loader_len = train_loader_len
        self.net = net
        self.training_params = training_params

    def __call__(self, context, **kwargs):
        if self.is_lr_scheduling_enabled(context):
            self.perform_scheduling(context)

    def is_lr_scheduling_enabled(self, context):
        raise NotImplementedError

    def perform_scheduling(self, context):
        raise NotImplementedError

    def update_lr(self, optimizer, epoch, batch_idx=None):
        for g in optimizer.param_groups:
            g["lr"] = self.lr.get(g.get("name", "default"), self.lr["default"])


@register_lr_warmup("LinearEpochLRWarmup", deprecated_name="linear_epoch_step")
class LinearEpochLRWarmup(LRCallbackBase):
    def __init__(self, **kwargs):
        super().__init__(Phase.TRAIN_EPOCH_START, **kwargs)
        tp = self.training_params
        if tp.warmup_initial_lr is not None:
            if isinstance(tp.warmup_initial_lr, numbers.Number):
                wl = {k: float(tp.warmup_initial_lr) for k in self.initial_lr}
            elif isinstance(tp.warmup_initial_lr, Mapping):
                wl = dict(tp.warmup_initial_lr)
            else:
                raise TypeError("Warmup initial lr expected to be of type float or Mapping.")
        else:
            wl = {k: v / (tp.lr_warmup_epochs + 1) for k, v in self.initial_lr.items()}
        self.warmup_initial_lr = wl
        self.warmup_step_size = {k: (self.initial_lr[k] - wl[k]) / tp.lr_warmup_epochs if tp.lr_warmup_epochs > 0 else 0 for k in self.initial_lr}

    def perform_scheduling(self, context):
        for k in self.initial_lr:
            self.lr[k] = self.warmup_initial_lr[k] + context.epoch * self.warmup_step_size[k]
        self.update_lr(context.optimizer, context.epoch, None)

    def is_lr_scheduling_enabled(self, context):
        return self.training_params.lr_warmup_epochs > 0 and self.training_params.lr_warmup_epochs >= context.epoch


@register_lr_warmup("LinearBatchLRWarmup", deprecated_name="linear_batch_step")
class LinearBatchLRWarmup(Callback):
    def __init__(self, warmup_initial_lr, initial_lr, train_loader_len, lr_warmup_steps, training_params, net, **kwargs):
        if isinstance(initial_lr, numbers.Number):
            initial_lr = {"default": initial_lr}
        self.initial_lr = initial_lr
        self.lr = dict(initial_lr)
        if isinstance(warmup_initial_lr, numbers.Number):
            warmup_initial_lr = {k: warmup_initial_lr for k in self.lr}
        elif not isinstance(warmup_initial_lr, Mapping):
            raise TypeError("Warmup initial lr expected to be of type float or Mapping.")
        lr_warmup_steps = min(lr_warmup_steps, train_loader_len)
        self.learning_rates = {k: np.linspace(start=warmup_initial_lr[k], stop=initial_lr[k], num=lr_warmup_steps, endpoint=True) for k in initial_lr}
        self.training_params, self.net = training_params, net
        self.train_loader_len, self.lr_warmup_steps = train_loader_len, lr_warmup_steps

    def on_train_batch_start(self, context):
        step = context.batch_idx + context.epoch * self.train_loader_len
        if step < self.lr_warmup_steps:
            for k in self.initial_lr:
                self.lr[k] = float(self.learning_rates[k][step])
            for g in context.optimizer.param_groups:
                g["lr"] = self.lr.get(g.get("name", "default"), self.lr["default"])


@register_lr_scheduler("StepLRScheduler", deprecated_name="step")
class StepLRScheduler(LRCallbackBase):
    def __init__(self, lr_updates, lr_decay_factor, step_lr_update_freq=None, **kwargs):
        super().__init__(Phase.TRAIN_EPOCH_END, **kwargs)
        lr_updates = list(lr_updates or [])
        if step_lr_update_freq and len(lr_updates):
            raise ValueError("Parameters lr_updates and step_lr_update_freq are mutually exclusive and cannot be passed to StepLRScheduler constructor simultaneously")
        if step_lr_update_freq is None and len(lr_updates) == 0:
            raise ValueError("At least one of [lr_updates, step_lr_update_freq] parameters should be passed to StepLRScheduler constructor")
        if step_lr_update_freq:
            max_epochs = self.training_params.max_epochs - self.training_params.lr_cooldown_epochs
            warm = self.training_params.lr_warmup_epochs
            lr_updates = [int(np.ceil(step_lr_update_freq * x)) for x in range(1, max_epochs) if warm <= int(np.ceil(step_lr_update_freq * x)) < max_epochs]
        self.lr_updates, self.lr_decay_factor = lr_updates, lr_decay_factor

    def perform_scheduling(self, context):
        passed = [x for x in self.lr_updates if x <= context.epoch]
        for k in self.lr:
            self.lr[k] = self.initial_lr[k] * self.lr_decay_factor ** len(passed)
        self.update_lr(context.optimizer, context.epoch, None)

    def is_lr_scheduling_enabled(self, context):
        return self.training_params.lr_warmup_epochs <= context.epoch


class _BatchStepScheduler(LRCallbackBase):
    def __init__(self, **kwargs):
        super().__init__(Phase.TRAIN_BATCH_STEP, **kwargs)

    def is_lr_scheduling_enabled(self, context):
        post = self.training_params.max_epochs - self.training_params.lr_cooldown_epochs
        return self.training_params.lr_warmup_epochs <= context.epoch < post


@register_lr_scheduler("ExponentialLRScheduler", deprecated_name="exp")
class ExponentialLRScheduler(_BatchStepScheduler):
    def __init__(self, lr_decay_factor: float, **kwargs):
        super().__init__(**kwargs)
        self.lr_decay_factor = lr_decay_factor

    def perform_scheduling(self, context):
        it = self.train_loader_len * (context.epoch - self.training_params.lr_warmup_epochs) + context.batch_idx
        for k in self.lr:
            self.lr[k] = self.initial_lr[k] * self.lr_decay_factor ** (it / self.train_loader_len)
        self.update_lr(context.optimizer, context.epoch, context.batch_idx)


@register_lr_scheduler("PolyLRScheduler", deprecated_name="poly")
class PolyLRScheduler(_BatchStepScheduler):
    def __init__(self, max_epochs, **kwargs):
        super().__init__(**kwargs)
        self.max_epochs = max_epochs

    def perform_scheduling(self, context):
        tp = self.training_params
        eff_epoch = context.epoch - tp.lr_warmup_epochs
        eff_max = self.max_epochs - tp.lr_warmup_epochs - tp.lr_cooldown_epochs
        it = (self.train_loader_len * eff_epoch + context.batch_idx) / tp.batch_accumulate
        max_it = self.train_loader_len * eff_max / tp.batch_accumulate
        for k in self.lr:
            self.lr[k] = self.initial_lr[k] * pow((1.0 - (it / max_it)), 0.9)
        self.update_lr(context.optimizer, context.epoch, context.batch_idx)


@register_lr_scheduler("CosineLRScheduler", deprecated_name="cosine")
class CosineLRScheduler(_BatchStepScheduler):
    def __init__(self, max_epochs, cosine_final_lr_ratio, **kwargs):
        super().__init__(**kwargs)
        self.max_epochs, self.cosine_final_lr_ratio = max_epochs, cosine_final_lr_ratio

    def perform_scheduling(self, context):
        tp = self.training_params
        eff_epoch = context.epoch - tp.lr_warmup_epochs
        eff_max = self.max_epochs - tp.lr_warmup_epochs - tp.lr_cooldown_epochs
        it = max(0, self.train_loader_len * eff_epoch + context.batch_idx - tp.lr_warmup_steps)
        max_it = self.train_loader_len * eff_max - tp.lr_warmup_steps
        for k in self.lr:
            self.lr[k] = float(self.compute_learning_rate(it, max_it, self.initial_lr[k], self.cosine_final_lr_ratio))
        self.update_lr(context.optimizer, context.epoch, context.batch_idx)

    def is_lr_scheduling_enabled(self, context):
        if self.training_params.lr_warmup_steps > 0:  # per-step warmup (callbacks.py:500-504)
            return self.train_loader_len * context.epoch + context.batch_idx >= self.training_params.lr_warmup_steps
        return super().is_lr_scheduling_enabled(context)

    @classmethod
    def compute_learning_rate(cls, step, total_steps, initial_lr, final_lr_ratio):
        lr = 0.5 * initial_lr * (1.0 + np.cos(step / (total_steps + 1) * math.pi))  # note the (total_steps + 1) of the reference
        return lr * (1 - final_lr_ratio) + (initial_lr * final_lr_ratio)


@register_lr_scheduler("FunctionLRScheduler", deprecated_name="function")
class FunctionLRScheduler(_BatchStepScheduler):
    def __init__(self, max_epochs, lr_schedule_function, **kwargs):
        super().__init__(**kwargs)
        assert callable(lr_schedule_function), "self.lr_function must be callable"
        self.lr_schedule_function, self.max_epochs = lr_schedule_function, max_epochs

    def perform_scheduling(self, context):
        tp = self.training_params
        eff_epoch = context.epoch - tp.lr_warmup_epochs
        eff_max = self.max_epochs - tp.lr_warmup_epochs - tp.lr_cooldown_epochs
        for k in self.lr:
            self.lr[k] = self.lr_schedule_function(initial_lr=self.initial_lr[k], epoch=eff_epoch, iter=context.batch_idx, max_epoch=eff_max,
                                                   iters_per_epoch=self.train_loader_len)
        self.update_lr(context.optimizer, context.epoch, context.batch_idx)

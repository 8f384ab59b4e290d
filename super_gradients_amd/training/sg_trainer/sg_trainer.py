"""Trainer - the train-step driver behind `Trainer(...).train(model, training_params, train_loader, valid_loader)`.

Reference: training/sg_trainer/sg_trainer.py - constructor :135-158, train() :792-1560 (parameter handling, optimizer /
LR-callback / EMA construction), the hot loop _train_epoch :461-534, _get_losses :536-560, _backward_step :611-647,
validation :1900-2010, checkpoint layout :649-720; required parameters training/params.py:113
(max_epochs, lr_mode, initial_lr, loss); batch formats training/utils/sg_trainer_utils.py:373-395.

What runs per batch is identical in order to the reference (callbacks -> forward -> loss -> backward -> [clip] -> optimizer ->
zero_grad -> EMA -> callbacks); what differs is what the calls cost: the network is one autograd node over libsgx_hip kernels,
optimizer / zero_grad / EMA are one launch each over the arenas, the gradient all-reduce is a handful of RCCL calls overlapped
with backward, and nothing in the loop synchronises with the host except the once-per-epoch read-back of the averaged loss.
`sync_bn: True` synchronises BatchNorm statistics across ranks (SgxNetwork.set_sync_bn).
Out of the hot path and not provided: sg_loggers / tensorboard, dataset statistics, torch.compile, QAT, remote
checkpoints, model averaging; DetectionMetrics matching is SURVEY 8(f)-2 (next).
"""
import os
import warnings
from typing import Dict, Mapping, Optional

import torch
from torch import nn

from ...common.registry import LOSSES, LR_SCHEDULERS_CLS_DICT, LR_WARMUP_CLS_DICT, METRICS, warn_if_deprecated
from ...modules.engine import SgxNetwork
from ..utils import distributed_training_utils as dtu
from ..utils.callbacks import Callback, CallbackHandler, LRCallbackBase, PhaseContext
from ..utils.checkpoint_utils import plain_number, read_checkpoint
from ..utils.ema import ModelEMA
from ..utils.optimizers import build_optimizer
from ..utils.utils import HpmStruct

DEFAULT_TRAINING_PARAMS = dict(
    lr_warmup_epochs=0, lr_warmup_steps=0, lr_cooldown_epochs=0, warmup_initial_lr=None, step_lr_update_freq=None, cosine_final_lr_ratio=0.01,
    warmup_mode="LinearEpochLRWarmup", lr_updates=[], lr_decay_factor=0.1, lr_schedule_function=None, optimizer="SGD", optimizer_params={},
    zero_weight_decay_on_bias_and_bn=False, criterion_params={}, ema=False, ema_params=dict(decay=0.9999, decay_type="exp", beta=15),
    train_metrics_list=[], valid_metrics_list=[], metric_to_watch="Accuracy", greater_metric_to_watch_is_better=True, mixed_precision=False,
    batch_accumulate=1, run_validation_freq=1, save_model=True, seed=42, phase_callbacks=[], clip_grad_norm=None, ckpt_name="ckpt_latest.pth",
    ckpt_best_name="ckpt_best.pth", max_train_batches=None, max_valid_batches=None, silent_mode=False, sync_bn=False, resume=False, resume_path=None,
    load_opt_params=True, save_ckpt_epoch_list=[], pre_prediction_callback=None,
)
REQUIRED = ("max_epochs", "lr_mode", "initial_lr", "loss")  # training/params.py:113


class DDPNotSetupException(Exception):
    def __init__(self):
        super().__init__("Your environment was not setup correctly for DDP: call setup_device(multi_gpu=..., num_gpus=...) before instantiating Trainer, "
                         "or launch with `python -m torch.distributed.run`.")


class AverageMeter:
    """Running mean of the loss items kept ON DEVICE (the reference's AverageMeter calls .item() semantics per batch)."""

    def __init__(self):
        self.sum, self.n = None, 0

    def update(self, value: torch.Tensor, batch_size: int):
        v = value.detach().float().reshape(-1) * batch_size
        self.sum = v.clone() if self.sum is None else self.sum + v
        self.n += batch_size

    def all_reduce(self, device=None):
        """Sum the meter over the data-parallel ranks, so that every rank reports - and selects its best checkpoint by - the same numbers.
        EVERY rank takes part, also one that saw no batch (an empty shard, max_valid_batches): it learns the item count from the others
        (one MAX reduction of the length) and contributes zeros - a rank that skipped the collective would leave the others waiting."""
        if dtu.get_world_size() <= 1:
            return
        dev = self.sum.device if self.sum is not None else torch.device(device if device is not None else "cpu")
        k = torch.tensor([0 if self.sum is None else self.sum.numel()], device=dev, dtype=torch.int64)
        torch.distributed.all_reduce(k, op=torch.distributed.ReduceOp.MAX)
        k = int(k.item())
        if k == 0:
            return
        mine = self.sum.double() if self.sum is not None else torch.zeros(k, device=dev, dtype=torch.float64)
        t = torch.cat([mine, torch.tensor([float(self.n)], device=dev, dtype=torch.float64)])
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.SUM)
        self.sum, self.n = t[:-1].float(), int(round(float(t[-1])))

    @property
    def average(self):
        if self.sum is None:
            return ()
        return tuple((self.sum / max(self.n, 1)).tolist())


def _update_metric(m, outputs, targets, inputs, extras):
    """Classification metrics take (preds, target); detection metrics also need the image batch (for H, W) and optional crowd targets
    (metrics/detection_metrics.py:166-200)."""
    from ..metrics.detection_metrics import DetectionMetrics

    if isinstance(m, DetectionMetrics):
        m.update(outputs, targets, device=str(inputs.device), inputs=inputs, crowd_targets=(extras or {}).get("crowd_targets"))
    else:
        m.update(outputs.detach() if torch.is_tensor(outputs) else outputs, targets)


def _metric_results(m):
    r = m.compute()
    return dict(r) if isinstance(r, dict) else {type(m).__name__: r}


def unpack_batch_items(batch_items):
    if len(batch_items) == 2:
        return batch_items[0], batch_items[1], {}
    if len(batch_items) == 3:
        return batch_items
    raise ValueError(f"Batch items aren't in the supported formats: (inputs, targets) or (inputs, targets, additional_batch_items); got {len(batch_items)} items")


class Accuracy:
    """Top-1 accuracy with the update/compute protocol of the torchmetrics object the reference uses (metrics/classification_metrics.py:37-52)."""

    greater_is_better = True

    def __init__(self, top_k: int = 1):
        self.top_k = top_k
        self.reset()

    def reset(self):
        self.correct = None
        self.total = 0

    def update(self, preds: torch.Tensor, target: torch.Tensor):
        if target.dim() == 2:
            target = target.argmax(1)
        top = preds.topk(self.top_k, dim=1).indices
        hit = (top == target.to(preds.device).reshape(-1, 1)).any(1).sum()
        self.correct = hit if self.correct is None else self.correct + hit
        self.total += int(target.shape[0])

    def compute(self):
        return float(self.correct) / max(self.total, 1) if self.correct is not None else 0.0


class Top5(Accuracy):
    def __init__(self):
        super().__init__(top_k=5)


METRICS.register("Accuracy")(Accuracy)
METRICS.register("Top5")(Top5)


class Trainer:
    def __init__(self, experiment_name: str, device: Optional[str] = None, multi_gpu=None, ckpt_root_dir: Optional[str] = None):
        if device is not None or multi_gpu is not None:
            raise KeyError("Trainer does not accept anymore 'device' and 'multi_gpu' as argument. Both should instead be passed to "
                           "super_gradients_amd.training.utils.distributed_training_utils.setup_device(device=..., multi_gpu=..., num_gpus=...)")
        if int(os.environ.get("WORLD_SIZE", "1")) > 1 and not dtu.is_distributed():
            raise DDPNotSetupException()
        self.experiment_name = experiment_name
        self.ckpt_root_dir = ckpt_root_dir or os.path.join(os.getcwd(), "checkpoints")
        self.checkpoints_dir_path = os.path.join(self.ckpt_root_dir, experiment_name)
        self.net = self.optimizer = self.criterion = self.ema_model = self.reducer = None
        self.training_params = None
        self.best_metric = None
        self.results = []  # one dict per epoch: train loss items, validation loss items / metrics, lr

    # ------------------------------------------------------------------------------------------------ set-up
    @staticmethod
    def _device():
        from ... import _lib

        if torch.cuda.is_available():
            return torch.device("cuda", torch.cuda.current_device())
        if _lib._TEST_HOST_MODE:  # tests/emu only
            return torch.device("cpu")
        raise RuntimeError("Trainer needs a HIP GPU: the train step has no CPU path")

    def _params(self, training_params) -> HpmStruct:
        given = training_params.to_dict() if hasattr(training_params, "to_dict") else dict(training_params or {})
        missing = [k for k in REQUIRED if given.get(k) is None and not (k == "initial_lr" and isinstance(given.get("optimizer"), torch.optim.Optimizer))]
        if missing:
            raise ValueError(f"training_params is missing required entries {missing} (required: {list(REQUIRED)})")
        merged = dict(DEFAULT_TRAINING_PARAMS)
        merged.update(given)
        # a recipe file's own containers (yaml.safe_load of training_hyperparams/*.yaml): `_target_` entries and exponent-form numbers
        from ...common.factories import resolve_recipe_values

        keep = {k: merged.pop(k) for k in ("loss", "optimizer", "phase_callbacks", "warmup_mode", "lr_mode") if k in merged and not isinstance(merged[k], (str, dict, list))}
        merged = resolve_recipe_values(merged)
        merged.update(keep)
        if merged["mixed_precision"]:
            warnings.warn("mixed_precision=True is accepted for recipe compatibility; the MI355X path computes in fp32 (parity mode)")
        # recipe switches that change what is trained must not be ignored silently (logging / tensorboard / torch.compile entries may be)
        for key, what in (("finetune", "freezing all but the fine-tune layers (get_finetune_lr_dict)"), ("precise_bn", "the precise-BN pass after each epoch")):
            if merged.get(key):
                raise NotImplementedError(f"training_params.{key}=True ({what}) is outside the MI355X train-step path")
        return HpmStruct(**merged)

    def _build_loss(self, tp):
        loss = tp.loss
        if isinstance(loss, Mapping):  # {LossName: {kwargs}} (LossesFactory)
            from ...common.factories import LossesFactory

            loss = LossesFactory().get(loss)
        if isinstance(loss, str):
            warn_if_deprecated(loss, LOSSES)
            if loss not in LOSSES:
                raise ValueError(f"Unknown loss '{loss}'; registered: {[k for k in LOSSES if not k.startswith('_')]}")
            loss = LOSSES[loss](**(tp.criterion_params or {}))
        if not isinstance(loss, nn.Module) and not callable(loss):
            raise TypeError("loss must be a registered name or an nn.Module")
        return loss

    def _build_lr_callbacks(self, tp, train_loader_len):
        cbs = []
        common = dict(train_loader_len=train_loader_len, net=self.net, training_params=tp, update_param_groups=False)
        warm = tp.warmup_mode
        if isinstance(warm, str):
            warn_if_deprecated(warm, LR_WARMUP_CLS_DICT)
            warm = LR_WARMUP_CLS_DICT[warm]
        elif warm is not None and not (isinstance(warm, type) and issubclass(warm, Callback)):
            raise RuntimeError("warmup_mode has to be either a name of a mode (str) or a subclass of PhaseCallback")
        if warm is not None and (tp.lr_warmup_epochs > 0 or tp.lr_warmup_steps > 0):
            wl = tp.warmup_initial_lr
            if wl is None and getattr(warm, "__name__", "") == "LinearBatchLRWarmup":
                wl = tp.initial_lr / (tp.lr_warmup_steps + 1)
            cbs.append(warm(warmup_initial_lr=wl, initial_lr=tp.initial_lr, lr_warmup_steps=tp.lr_warmup_steps, **common))
        mode = tp.lr_mode
        if isinstance(mode, str):
            warn_if_deprecated(mode, LR_SCHEDULERS_CLS_DICT)
            if mode not in LR_SCHEDULERS_CLS_DICT:
                raise ValueError(f"Unknown lr_mode '{mode}'; registered: {[k for k in LR_SCHEDULERS_CLS_DICT if not k.startswith('_')]}")
            cbs.append(LR_SCHEDULERS_CLS_DICT[mode](initial_lr=tp.initial_lr, lr_updates=tp.lr_updates, lr_decay_factor=tp.lr_decay_factor,
                                                    step_lr_update_freq=tp.step_lr_update_freq, max_epochs=tp.max_epochs,
                                                    cosine_final_lr_ratio=tp.cosine_final_lr_ratio, lr_schedule_function=tp.lr_schedule_function, **common))
        elif isinstance(mode, type) and issubclass(mode, LRCallbackBase):
            cbs.append(mode(initial_lr=tp.initial_lr, max_epochs=tp.max_epochs, **common))
        elif mode is not None:
            raise NotImplementedError("lr_mode must be a registered scheduler name or an LRCallbackBase subclass on the HIP path "
                                      "(torch.optim.lr_scheduler mappings are outside the hot path)")
        return cbs

    # ------------------------------------------------------------------------------------------------ train
    def train(self, model: nn.Module, training_params=None, train_loader=None, valid_loader=None, test_loaders=None, additional_configs_to_log: Dict = None):
        if train_loader is None:
            raise ValueError("No `train_loader` found. Please provide a value for `train_loader`")
        if test_loaders:
            raise NotImplementedError("test_loaders (extra evaluation sets per epoch) are outside the MI355X train-step path; validate them with "
                                      "separate calls")
        if not isinstance(model, SgxNetwork):
            raise TypeError("Trainer on the MI355X path trains models obtained from super_gradients_amd.training.models.get() (SgxNetwork)")
        tp = self.training_params = self._params(training_params)
        if tp.seed is not None:
            torch.manual_seed(int(tp.seed) + dtu.get_rank())  # per-rank seed as training/utils/utils.py:376-388
        device = self._device()
        self.net = model.materialize(device)
        world = dtu.get_world_size()
        self.reducer = None
        if world > 1:
            self.reducer = dtu.GradientAllReducer(self.net, self.net.gradient_buckets())
            self.reducer.broadcast_parameters(0)
            if tp.sync_bn:
                self.net.set_sync_bn(True)
        self.criterion = self._build_loss(tp)
        if isinstance(self.criterion, nn.Module):
            self.criterion.to(device)
        self.optimizer = build_optimizer(self.net, tp.initial_lr, tp)
        if tp.initial_lr is None:
            tp.initial_lr = float(self.optimizer.param_groups[0]["lr"])
        n_train = len(train_loader) if tp.max_train_batches is None else min(len(train_loader), tp.max_train_batches)
        lr_callbacks = self._build_lr_callbacks(tp, len(train_loader))
        self.ema_model = ModelEMA.from_params(self.net, **dict(tp.ema_params or {})) if tp.ema else None
        handler = CallbackHandler(lr_callbacks + list(tp.phase_callbacks or []))
        from ...common.factories import MetricsFactory

        valid_metrics = [MetricsFactory().get(m) for m in (tp.valid_metrics_list or [])]  # name / {Name: kwargs} / instance
        train_metrics = [MetricsFactory().get(m) for m in (tp.train_metrics_list or [])]
        context = PhaseContext(optimizer=self.optimizer, net=self.net, criterion=self.criterion, experiment_name=self.experiment_name,
                               ckpt_dir=self.checkpoints_dir_path, train_loader=train_loader, valid_loader=valid_loader, training_params=tp,
                               ema_model=self.ema_model, metric_to_watch=tp.metric_to_watch, valid_metrics=valid_metrics, lr_warmup_epochs=tp.lr_warmup_epochs,
                               stop_training=False)
        self._processing_params = self._get_preprocessing_from_valid_loader(valid_loader)
        if self._processing_params is not None:  # sg_trainer.py:1704-1707: the trained model can predict() without further set-up
            self.net.set_dataset_processing_params(**self._processing_params)
        start_epoch = 0
        if tp.resume or tp.resume_path:
            start_epoch = self._load_checkpoint(tp.resume_path or os.path.join(self.checkpoints_dir_path, tp.ckpt_name), tp.load_opt_params)
        self.loss_logging_items_names = None
        self._global_steps_done = 0
        handler.on_training_start(context)
        for epoch in range(start_epoch, tp.max_epochs):
            if context.stop_training:
                break
            context.update_context(epoch=epoch)
            train_items = self._train_epoch(context, handler, train_loader, n_train, train_metrics, world)
            row = {"epoch": epoch, "lr": float(self.optimizer.param_groups[0]["lr"]), "train": dict(zip(self.loss_logging_items_names, train_items))}
            for m in train_metrics:
                row["train"].update(_metric_results(m))
            if valid_loader is not None and (epoch + 1) % tp.run_validation_freq == 0:
                row["valid"] = self._validate_epoch(context, handler, valid_loader, valid_metrics)
                self._track_best(context, handler, row, epoch)
            self.results.append(row)
            if tp.save_model and dtu.get_rank() == 0:
                self._save_checkpoint(epoch, tp.ckpt_name, row)
                if epoch in (tp.save_ckpt_epoch_list or []):
                    self._save_checkpoint(epoch, f"ckpt_epoch_{epoch}.pth", row)
            if not tp.silent_mode and dtu.get_rank() == 0:
                print(f"[{self.experiment_name}] epoch {epoch}: " + ", ".join(f"{k}={v}" for k, v in row.items() if k != "epoch"), flush=True)
        handler.on_training_end(context)
        return self.results

    def _train_epoch(self, context, handler, train_loader, expected_iterations, train_metrics, world):
        tp = self.training_params
        self.net.train()
        device = self.net._device
        meter = AverageMeter()
        for m in train_metrics:
            m.reset()
        context.update_context(loss_avg_meter=meter, metrics_compute_fn=train_metrics)
        handler.on_train_loader_start(context)
        total_steps = len(train_loader) * tp.max_epochs
        for batch_idx, batch_items in enumerate(train_loader):
            if expected_iterations <= batch_idx:
                break
            inputs, targets, extras = unpack_batch_items(batch_items)
            inputs = inputs.to(device, non_blocking=True)
            targets = targets.to(device, non_blocking=True) if torch.is_tensor(targets) else targets
            if tp.pre_prediction_callback is not None:
                inputs, targets = tp.pre_prediction_callback(inputs, targets, batch_idx)
            context.update_context(batch_idx=batch_idx, inputs=inputs, target=targets, additional_batch_items=extras, **extras)
            handler.on_train_batch_start(context)
            if self.reducer is not None and world > 1:
                self.reducer.broadcast_buffers(0)  # DDP(broadcast_buffers=True) semantics
            outputs = self.net(inputs)
            loss, items = self._get_losses(outputs, targets)
            context.update_context(preds=outputs, loss_log_items=items, loss_logging_items_names=self.loss_logging_items_names)
            handler.on_train_batch_loss_end(context)
            meter.update(items, int(inputs.shape[0]))
            for m in train_metrics:
                _update_metric(m, outputs, targets, inputs, extras)
            global_step = batch_idx + 1 + len(train_loader) * context.epoch
            if self.reducer is not None:
                # gradient accumulation under data parallelism: the arena sums the micro-batches locally and is exchanged ONCE, by the
                # backward that precedes the optimizer step (all-reducing an already reduced sum again would count it `world` times)
                self.reducer.sync = global_step % tp.batch_accumulate == 0
            loss.backward()
            handler.on_train_batch_backward_end(context)
            if global_step % tp.batch_accumulate == 0:
                handler.on_train_batch_gradient_step_start(context)
                if tp.clip_grad_norm:
                    self._clip_grad_norm(float(tp.clip_grad_norm), world)
                if self.reducer is not None and world > 1 and hasattr(self.optimizer, "exp_avg"):
                    self.optimizer.step(grad_scale=self.reducer.grad_scale)
                elif self.reducer is not None and world > 1:
                    self.net.g_arena.buf.mul_(1.0 / world)
                    self.optimizer.step()
                else:
                    self.optimizer.step()
                self.optimizer.zero_grad()
                if self.ema_model is not None:
                    self.ema_model.update(self.net, step=global_step, total_steps=total_steps)
                handler.on_train_batch_gradient_step_end(context)
            handler.on_train_batch_end(context)
        handler.on_train_loader_end(context)
        return meter.average

    def _get_losses(self, outputs, targets):
        loss = self.criterion(outputs, targets)
        if isinstance(loss, tuple):
            loss, items = loss
        else:
            items = loss.unsqueeze(0).detach()
        if self.loss_logging_items_names is None:  # sg_trainer.py:2407-2420: "<Criterion>/<component>" titles, or the bare class name
            crit = type(self.criterion).__name__
            names = getattr(self.criterion, "component_names", None)
            if names is None and len(items) > 1:
                names = [f"loss_{i}" for i in range(len(items))]
            self.loss_logging_items_names = [f"{crit}/{n}" for n in names] if names is not None else [crit]
            if names is not None and self.training_params.metric_to_watch in names:  # a bare component name is what recipes write
                self.training_params.metric_to_watch = f"{crit}/{self.training_params.metric_to_watch}"
        if len(items) != len(self.loss_logging_items_names):
            raise ValueError(f"Loss output length must match loss_logging_items_names. Got {len(items)}, and {len(self.loss_logging_items_names)}")
        return loss, items

    def _clip_grad_norm(self, max_norm, world):
        g = self.net.g_arena.buf
        norm = torch.linalg.vector_norm(g) / world  # the arena holds the SUM over ranks until the optimizer folds in 1/world
        g.mul_(torch.clamp(max_norm / (norm + 1e-6), max=1.0))

    @torch.no_grad()
    def _validate_epoch(self, context, handler, valid_loader, metrics):
        tp = self.training_params
        device = self.net._device
        meter = AverageMeter()
        for m in metrics:
            m.reset()
        handler.on_validation_loader_start(context)

        def run(net):
            net.eval()
            for batch_idx, batch_items in enumerate(valid_loader):
                if tp.max_valid_batches is not None and tp.max_valid_batches <= batch_idx:
                    break
                inputs, targets, extras = unpack_batch_items(batch_items)
                inputs = inputs.to(device, non_blocking=True)
                targets = targets.to(device, non_blocking=True) if torch.is_tensor(targets) else targets
                context.update_context(batch_idx=batch_idx, inputs=inputs, target=targets, additional_batch_items=extras)
                handler.on_validation_batch_start(context)
                outputs = net(inputs)
                loss = self.criterion(outputs, targets)
                items = loss[1] if isinstance(loss, tuple) else loss.reshape(1)
                meter.update(items, int(inputs.shape[0]))
                for m in metrics:
                    _update_metric(m, outputs, targets, inputs, extras)
                context.update_context(preds=outputs, loss_log_items=items)
                handler.on_validation_batch_end(context)

        if self.ema_model is not None:  # the reference validates the EMA weights (sg_trainer.py:1418-1424)
            with self.ema_model.averaged() as net:
                run(net)
        else:
            run(self.net)
        self.net.train()
        meter.all_reduce(device=device)  # ranks validate different shards: _track_best must see one value everywhere
        out = dict(zip(self.loss_logging_items_names or [], meter.average))
        for m in metrics:
            out.update(_metric_results(m))
        context.update_context(metrics_dict=out)
        handler.on_validation_loader_end(context)
        return out

    def _track_best(self, context, handler, row, epoch):
        tp = self.training_params
        watch = tp.metric_to_watch
        vals = row.get("valid", {})
        if watch not in vals:
            cand = [k for k in vals if k.lower() == str(watch).lower()]
            if not cand:
                return
            watch = cand[0]
        v = vals[watch]
        better = self.best_metric is None or (v > self.best_metric if tp.greater_metric_to_watch_is_better else v < self.best_metric)
        if better:
            self.best_metric = v
            if tp.save_model and dtu.get_rank() == 0:
                self._save_checkpoint(epoch, tp.ckpt_best_name, row)
            handler.on_validation_end_best_epoch(context)

    def _get_preprocessing_from_valid_loader(self, valid_loader) -> Optional[dict]:
        """sg_trainer.py:1709-1720: a validation dataset that knows its pre-processing (`get_dataset_preprocessing_params()` -> class_names,
        image_processor, iou, conf) hands it to a model that can predict(); failures only warn."""
        ds = getattr(valid_loader, "dataset", None)
        if not (hasattr(self.net, "set_dataset_processing_params") and hasattr(ds, "get_dataset_preprocessing_params")):
            return None
        try:
            return dict(ds.get_dataset_preprocessing_params())
        except Exception as e:  # noqa: BLE001
            warnings.warn(f"Could not set preprocessing pipeline from the validation dataset:\n {e}.\n Before calling predict make sure to call "
                          "set_dataset_processing_params.")
            return None

    # ------------------------------------------------------------------------------------------------ checkpoints
    def _save_checkpoint(self, epoch, name, row):
        """Same dictionary layout as the reference's checkpoints (sg_trainer.py:649-720): net / ema_net / optimizer_state_dict / epoch / metrics."""
        os.makedirs(self.checkpoints_dir_path, exist_ok=True)
        state = {"net": {k: v.detach().cpu().clone() for k, v in self.net.state_dict().items()}, "epoch": epoch, "metrics": row,
                 "optimizer_state_dict": self._optimizer_state(), "acc": None if self.best_metric is None else float(self.best_metric)}
        state["metrics"] = {split: {k: float(v) for k, v in vals.items()} if isinstance(vals, dict) else vals for split, vals in row.items()} if isinstance(row, dict) else row
        if self.ema_model is not None:
            state["ema_net"] = {k: v.cpu() for k, v in self.ema_model.state_dict().items()}
        if getattr(self, "_processing_params", None) is not None:  # sg_trainer.py:710-712, with the image processor as a plain config
            pp = dict(self._processing_params)
            ip = pp.get("image_processor")
            if hasattr(ip, "to_config"):
                pp["image_processor"] = ip.to_config()
            elif isinstance(ip, (list, tuple)) and all(hasattr(q, "to_config") for q in ip):  # a list of Processing objects = their composition
                pp["image_processor"] = {"ComposeProcessing": {"processings": [q.to_config() for q in ip]}}
            if pp.get("class_names") is not None:
                pp["class_names"] = list(pp["class_names"])
            state["processing_params"] = pp
        torch.save(state, os.path.join(self.checkpoints_dir_path, name))

    def _optimizer_state(self):
        o = self.optimizer
        st = {"param_groups": [{k: v for k, v in g.items() if k != "params"} for g in o.param_groups], "steps": getattr(o, "_steps", 0)}
        for name in ("exp_avg", "exp_avg_sq", "momentum_buffer"):
            if hasattr(o, name):
                st[name] = getattr(o, name).detach().cpu().clone()
        return st

    def _load_checkpoint(self, path, load_opt):
        ckpt = read_checkpoint(path)  # tensors, numbers, strings and plain containers only (same rule as models.get)
        self.net.load_state_dict(ckpt["net"], strict=True)
        if self.ema_model is not None and "ema_net" in ckpt:
            with self.ema_model.averaged() as net:
                net.load_state_dict(ckpt["ema_net"], strict=True)
                self.ema_model.p_ema.copy_(net.p_arena.buf)
                self.ema_model.b_ema.copy_(net.b_arena.buf)
        elif self.ema_model is not None:
            # a checkpoint without EMA weights: the average restarts from the loaded weights (the reference builds its EMA from the
            # already-loaded net), not from the random initialisation the EMA arenas were cloned from
            self.ema_model.p_ema.copy_(self.net.p_arena.buf)
            self.ema_model.b_ema.copy_(self.net.b_arena.buf)
        if load_opt and "optimizer_state_dict" in ckpt:
            st = ckpt["optimizer_state_dict"]
            for name in ("exp_avg", "exp_avg_sq", "momentum_buffer"):
                if name in st and hasattr(self.optimizer, name):
                    getattr(self.optimizer, name).copy_(st[name])
            self.optimizer._steps = plain_number(st, "steps", int, 0)
            for g, s in zip(self.optimizer.param_groups, st["param_groups"]):
                g.update({k: v for k, v in s.items() if k in ("lr",)})
        self.best_metric = plain_number(ckpt, "acc", float)
        return plain_number(ckpt, "epoch", int, -1) + 1

    # ------------------------------------------------------------------------------------------------ recipe entry
    @classmethod
    def train_from_config(cls, cfg: Mapping):
        """Minimal mirror of Trainer.train_from_config (sg_trainer.py:203-297) for already-resolved configs:
        cfg = {experiment_name, ckpt_root_dir, architecture, arch_params, num_classes, training_hyperparams, train_loader, valid_loader}."""
        from .. import models

        trainer = cls(experiment_name=cfg["experiment_name"], ckpt_root_dir=cfg.get("ckpt_root_dir"))
        model = models.get(cfg["architecture"], arch_params=cfg.get("arch_params"), num_classes=cfg.get("num_classes"),
                           checkpoint_path=(cfg.get("checkpoint_params") or {}).get("checkpoint_path"))
        res = trainer.train(model, cfg["training_hyperparams"], cfg["train_loader"], cfg.get("valid_loader"))
        return model, res

"""Leaf layers of the MI355X train-step: the kernel call sequences (forward and hand-written backward) that replace the
reference's Conv2d / BatchNorm2d / ReLU / ConvTranspose2d / MaxPool2d / Linear modules.

Every class keeps the reference's parameter names and logical shapes (state_dict compatible):
  ConvLayer        nn.Conv2d                         weight [K,C,R,S] (+ bias)      modules/conv_bn_act_block.py:88
  BatchNorm        nn.BatchNorm2d                    weight, bias, running_mean, running_var, num_batches_tracked
  ConvTranspose2x2 nn.ConvTranspose2d(k=2,s=2)       weight [C,K,2,2], bias          modules/sampling.py:72-73
  LinearLayer      nn.Linear                         weight [K,C], bias              classification_models/resnet.py:186
The fused sequences (conv -> BN statistics in the conv epilogue -> one affine+activation sweep) are assembled by the
blocks in conv_bn_act_block.py / qarepvgg_block.py.
"""
import math
import os

import torch
from torch import nn

from .. import kernels as K
from .engine import SgxBlock

ACT_NAMES = {None: None, "none": None, "identity": None, "relu": "relu", "silu": "silu", "swish": "silu"}


def act_name(activation_type) -> str:
    """Accepts what the reference's ActivationsTypeFactory accepts for this path: a string, None, or an nn.Module type."""
    if activation_type is None:
        return None
    if isinstance(activation_type, str):
        key = activation_type.lower()
        if key not in ACT_NAMES:
            raise ValueError(f"activation '{activation_type}' is not available on the HIP path (relu, silu, none)")
        return ACT_NAMES[key]
    if isinstance(activation_type, type):
        if issubclass(activation_type, nn.ReLU):
            return "relu"
        if issubclass(activation_type, nn.SiLU):
            return "silu"
        if issubclass(activation_type, nn.Identity):
            return None
    raise ValueError(f"activation {activation_type!r} is not available on the HIP path (relu, silu, none)")


class ConvLayer(SgxBlock):
    """Convolution parameters + the three conv kernels.  Not a block by itself: owners call conv()/wgrad()/dgrad()."""

    _param_kinds = {"weight": "conv"}

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, bias=False):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.padding = kernel_size, stride, padding
        w = torch.empty(out_channels, in_channels, kernel_size, kernel_size)
        nn.init.kaiming_uniform_(w, a=math.sqrt(5))  # nn.Conv2d default
        self.weight = nn.Parameter(w)
        if bias:
            bound = 1.0 / math.sqrt(in_channels * kernel_size * kernel_size)
            self.bias = nn.Parameter(torch.empty(out_channels).uniform_(-bound, bound))
        else:
            self.register_parameter("bias", None)
        self._w = self._gw = None

    def on_materialize(self):
        slots = {s.param: s for s in self._net.slots}
        s = slots[self.weight]
        self._w, self._gw = s.kernel_view, s.grad_kernel_view
        # persistent buffer of the data gradient's transposed weights (filled once per step by the network's batched transpose launch);
        # filter counts that are not a multiple of 4 (class-prediction convs) run their zero-padded backward with per-call transposes
        pre = (self._net.aux_stream is not None or self._net.wt_batch) and self._w.shape[0] % 4 == 0
        self._wt = K.conv2d_wt_buffer(self._w, self._w.device) if pre else None

    def transpose_weights(self):
        K.conv2d_transpose_weights(self._w, self._wt, stride=self.stride, pad=self.padding)

    def conv(self, x, out=None, act=None, addend=None, stats=False):
        return K.conv2d_fwd(x, self._w, bias=self.bias, addend=addend, out=out, act=act, stride=self.stride, pad=self.padding, stat_partials=stats)

    def wgrad(self, x, dy, bias_grad=True):
        """bias_grad=False: the caller knows sum(dy) is zero per channel (dy is a training-mode BatchNorm's input gradient)"""
        net = self._net
        if net.wg_group_flops > 0:  # queued: the network launches the weight gradients of a stretch of backward together
            if self.bias is not None and bias_grad:
                gb = self.bias.grad
                net.fork_side(lambda: K.colsum(dy, gb), dy)
            net.queue_wgrad(x, dy, self._gw, self.stride, self.padding)
            return
        net.fork_side(lambda: K.conv2d_bwd_weight(x, dy, self._gw, self.bias.grad if (self.bias is not None and bias_grad) else None,
                                                  stride=self.stride, pad=self.padding), x, dy)

    def dgrad(self, dy, x_shape, out=None, accumulate=False, addend=None, reqs=None):
        """reqs: BatchNorm-backward reduce requests (kernels.BnReduceRequest) of the layer(s) whose output gradient this launch finalises -
        carried by the launch's epilogue where it can (their .parts is set), left alone otherwise (the layer reduces on its own)."""
        if self._wt is not None and self._net._wt_valid:  # transposed under the forward pass (engine.prefetch_dgrad_weights)
            return K.conv2d_bwd_data_wt(dy, self._w, self._wt, x_shape, stride=self.stride, pad=self.padding, addend=addend, out=out, accumulate=accumulate,
                                        reqs=reqs if self._net.fuse_bn_reduce else None)
        return K.conv2d_bwd_data(dy, self._w, x_shape, stride=self.stride, pad=self.padding, addend=addend, out=out, accumulate=accumulate)


class BatchNorm(SgxBlock):
    """BatchNorm2d parameters/buffers + statistics finalisation.  Statistics arrive as per-workgroup partial sums from
    the producing kernel's epilogue (conv or affine sweep), so the activation tensor is never re-read for them."""

    def __init__(self, num_features, eps=1e-5, momentum=0.1):
        super().__init__()
        self.num_features, self.eps, self.momentum = num_features, eps, momentum
        self.weight = nn.Parameter(torch.ones(num_features))
        self.bias = nn.Parameter(torch.zeros(num_features))
        self.register_buffer("running_mean", torch.zeros(num_features))
        self.register_buffer("running_var", torch.ones(num_features))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))
        self.sync = False  # SgxNetwork.set_sync_bn: statistics over all data-parallel ranks (nn.SyncBatchNorm semantics)

    def on_materialize(self):
        pass

    def _synced(self):
        from ..training.utils.distributed_training_utils import collectives_active

        return self.sync and collectives_active()

    def scale_shift(self, parts, M, training):
        """-> (scale, shift, save_mean, save_invstd); eval mode folds the running statistics."""
        if training:
            if M <= 1:  # F.batch_norm's own check (torch/nn/functional.py: _verify_batch_size): the unbiased variance divides by M - 1
                raise ValueError(f"Expected more than 1 value per channel when training, got {M} value(s) per channel ({self.num_features} channels)")
            if self._synced():
                return K.bn_finalize_sync(parts, M, self.weight, self.bias, self.eps, self.momentum, self.running_mean, self.running_var)
            return K.bn_finalize(parts, M, self.weight, self.bias, self.eps, self.momentum, self.running_mean, self.running_var)
        sc, sh = K.bn_eval_scale_shift(self.weight, self.bias, self.running_mean, self.running_var, self.eps)
        return sc, sh, None, None

    def backward(self, dy, t, scale, shift, mean, invstd, act, dx_out=None, want_g=False, parts=None):
        return K.bn_bwd(dy, t, scale, shift, self.weight, mean, invstd, self.weight.grad, self.bias.grad, act=act, dx_out=dx_out, want_g=want_g,
                        sync=self._synced(), parts=parts)


class ConvTranspose2x2(SgxBlock):
    _param_kinds = {"weight": "convT"}

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        w = torch.empty(in_channels, out_channels, 2, 2)
        nn.init.kaiming_uniform_(w, a=math.sqrt(5))
        self.weight = nn.Parameter(w)
        bound = 1.0 / math.sqrt(out_channels * 4)  # nn.ConvTranspose2d: fan_in computed on dim 1
        self.bias = nn.Parameter(torch.empty(out_channels).uniform_(-bound, bound))

    def on_materialize(self):
        slots = {s.param: s for s in self._net.slots}
        s = slots[self.weight]
        self._w, self._gw = s.kernel_view, s.grad_kernel_view
        # the forward IS the data gradient of the adjoint 2x2 stride-2 convolution: its transposed filter rides in the network's per-step
        # transpose batch like every data gradient's (engine.prefetch_dgrad_weights) - four transpose launches per call off the forward chain
        self.stride, self.padding = 2, 0
        self._wt = K.conv2d_wt_buffer(self._w, self._w.device) if (self._net.wt_batch and os.environ.get("SGX_CONVT_PRETRANSPOSED", "1") != "0") else None  # (0: measurement switch)

    def transpose_weights(self):
        K.conv2d_transpose_weights(self._w, self._wt, stride=2, pad=0)

    def fwd(self, x, out=None):
        self._x = x if self.training else None
        pre = self._wt is not None and self._net._wt_valid and self._net._step_forward and x.dtype == torch.float32
        return K.convT2x2_fwd(x, self._w, self.bias, out=out, wtt=self._wt if pre else None)

    def bwd(self, dy, dx_out=None, accumulate=False, addend=None, need_dx=True):
        x, self._x = self._x, None
        self._net.fork_side(lambda: K.convT2x2_bwd_weight(x, dy, self._gw, self.bias.grad), x, dy)
        if not need_dx:
            return None
        if accumulate or addend is not None:
            dx = K.convT2x2_bwd_data(dy, self._w)
            if addend is not None:
                K.axpy(addend, out=dx, accumulate=True)
            if accumulate:
                K.axpy(dx, out=dx_out, accumulate=True)
                return dx_out
            return dx
        return K.convT2x2_bwd_data(dy, self._w, out=dx_out)


class MaxPool(SgxBlock):
    def __init__(self, kernel_size, stride, padding):
        super().__init__()
        self.k, self.s, self.p = kernel_size, stride, padding

    def on_materialize(self):
        pass

    def fwd(self, x, out=None):
        y, am = K.maxpool_fwd(x, self.k, self.s, self.p, out=out, want_argmax=self.training)
        self._ctx = (am, tuple(x.shape))
        return y

    def bwd(self, dy, dx_out=None, accumulate=False, addend=None, need_dx=True):
        (am, shape), self._ctx = self._ctx, None
        return K.maxpool_bwd(dy, am, shape, self.k, self.s, self.p, out=dx_out, accumulate=accumulate)


class LinearLayer(SgxBlock):
    """nn.Linear on the conv kernels: a 1x1 convolution over an [N,1,1,C] tensor.  Output features are padded to a
    multiple of 4 inside (the backward kernels read dy in 16-byte groups); callers see exactly out_features."""

    def __init__(self, in_features, out_features):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self._kp = (out_features + 3) // 4 * 4
        w = torch.empty(out_features, in_features)
        nn.init.kaiming_uniform_(w, a=math.sqrt(5))
        self.weight = nn.Parameter(w)
        bound = 1.0 / math.sqrt(in_features)
        self.bias = nn.Parameter(torch.empty(out_features).uniform_(-bound, bound))

    def on_materialize(self):
        pass

    def _padded(self):
        if self._kp == self.out_features:
            return self.weight.view(self.out_features, self.in_features, 1, 1), self.bias
        w = torch.zeros(self._kp, self.in_features, device=self.weight.device)
        w[: self.out_features].copy_(self.weight)
        b = torch.zeros(self._kp, device=self.weight.device)
        b[: self.out_features].copy_(self.bias)
        return w.view(self._kp, self.in_features, 1, 1), b

    def fwd(self, x2d, out=None):
        n = x2d.shape[0]
        x = x2d.view(n, 1, 1, self.in_features)
        w, b = self._padded()
        y = K.conv2d_fwd(x, w, bias=b)
        self._ctx = (x, w) if self.training else None
        return y.view(n, self._kp)[:, : self.out_features]

    def bwd(self, dy2d, dx_out=None, accumulate=False, addend=None, need_dx=True):
        (x, w), self._ctx = self._ctx, None
        n = x.shape[0]
        if self._kp == self.out_features:
            dy = dy2d.contiguous().view(n, 1, 1, self._kp)
            gw, gb = self.weight.grad.view(self._kp, self.in_features, 1, 1), self.bias.grad
        else:
            dy = torch.zeros(n, 1, 1, self._kp, device=x.device)
            dy.view(n, self._kp)[:, : self.out_features].copy_(dy2d)
            gw = torch.zeros(self._kp, self.in_features, 1, 1, device=x.device)
            gb = torch.zeros(self._kp, device=x.device)
        K.conv2d_bwd_weight(x, dy, gw, gb)
        if self._kp != self.out_features:
            K.axpy(gw.view(1, 1, self._kp, self.in_features)[:, :, : self.out_features], out=self.weight.grad.view(1, 1, self.out_features, self.in_features),
                   accumulate=True)
            self.bias.grad.add_(gb[: self.out_features])
        return K.conv2d_bwd_data(dy, w, tuple(x.shape)).view(n, self.in_features)

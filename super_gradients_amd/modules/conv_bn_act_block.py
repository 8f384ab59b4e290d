"""conv -> BatchNorm -> activation blocks on the HIP kernels.

Reference classes mirrored (same constructor arguments for the supported subset, same state_dict keys):
  Conv        modules/conv_bn_act_block.py:80-93    keys conv.weight, bn.*       (autopad, no conv bias)
  ConvBNAct   modules/conv_bn_act_block.py:9-69     keys seq.conv.*, seq.bn.*
  ConvBNReLU  modules/conv_bn_relu_block.py:8-60    keys seq.conv.*, seq.bn.*
Kernel sequence (training):  conv (BN partial statistics emitted by the conv epilogue) -> bn_finalize (tiny)
-> affine+activation sweep.  Backward: BN(+act) backward in three sweeps (activation mask recomputed from the saved
conv output, nothing else stored) -> weight gradient -> data gradient.
"""
import torch
from torch import nn

from .. import kernels as K
from .engine import SgxBlock
from .layers import BatchNorm, ConvLayer, act_name


class _ConvBN(SgxBlock):
    """Shared implementation; subclasses only choose where `conv`/`bn` are registered (key names)."""

    _folded = None  # (filter with the BatchNorm scale folded in, shift as bias): eval form set by prep_model_for_conversion
    _folded_half = None  # the same for a channel-padded filter (C % 4 != 0: an RGB stem) - read by the half-precision path only, which re-lays filters out itself

    def _parts(self):
        raise NotImplementedError

    def on_materialize(self):
        pass

    def prep_model_for_conversion(self, input_size=None, **kwargs):
        """Eval form as ONE launch: act(conv(x, w * s[k]) + t[k]) with s = gamma / sqrt(running_var + eps), t = beta - running_mean * s -
        bias and activation live in the conv epilogue, the BatchNorm apply sweep and the scale/shift kernel of the eval path disappear.
        What torch.nn.utils.fusion.fuse_conv_bn_eval does for the reference's export paths; here the parameters stay as they are (the block
        still trains: train() drops the folded copy), only eval-mode forward reads the folded filter.  One-off preparation in torch ops."""
        conv, bn = self._parts()
        w = conv._w
        if w is None:
            raise RuntimeError("prep_model_for_conversion needs a materialised model (the fold reads the arena views)")
        K_, C_, R_, S_ = w.shape
        if w.stride() != (R_ * S_ * C_, 1, S_ * C_, C_):
            # channel-padded filters (C % 4 != 0): the fp32 path keeps the general eval sequence; the half-precision path converts filters to
            # its own bf16 layout anyway (kernels.half_filter takes any strides) - fold for it alone (prep used to return silently
            # and the bf16 forward then raised "call prep_model_for_conversion() first")
            with torch.no_grad():
                s = bn.weight.detach() / torch.sqrt(bn.running_var + bn.eps)
                self._folded_half = ((w.detach() * s.view(-1, 1, 1, 1)).contiguous(), (bn.bias.detach() - bn.running_mean * s).contiguous())
            return
        with torch.no_grad():
            s = bn.weight.detach() / torch.sqrt(bn.running_var + bn.eps)
            wf = K.ohwi_empty(K_, C_, R_, S_, w.device)
            wf.copy_(w.detach() * s.view(-1, 1, 1, 1))
            self._folded = (wf, (bn.bias.detach() - bn.running_mean * s).contiguous())

    def train(self, mode: bool = True):
        if mode:
            self._folded = self._folded_half = None  # the weights are about to change
        return super().train(mode)

    def fwd(self, x, out=None, post_add=None, post_scale=None):
        """post_add: tensor added AFTER the activation (pp_yolo_head.py:205 `stem_cls(feat, avg_feat) + feat`); its gradient is the
        caller's (dy reaches it unchanged).  post_scale (half-precision inference only): a multiplier of post_add."""
        conv, bn = self._parts()
        if x.dtype == K.HALF:  # half-precision inference: the folded deployment form, everything in ONE bf16 launch
            folded = self._folded if self._folded is not None else self._folded_half
            if self.training or folded is None:
                raise RuntimeError("half-precision inference runs the folded deployment form: call prep_model_for_conversion() in eval mode first")
            return K.conv2d_fwd(x, folded[0], bias=folded[1], out=out, act=self.act, stride=conv.stride, pad=conv.padding,
                                post_add=post_add, post_scale=post_scale)
        if post_scale is not None:
            raise RuntimeError("post_scale: half-precision inference only")
        if self.training:
            self._folded = self._folded_half = None  # a training step follows: a folded eval filter would be stale afterwards
            t, parts = conv.conv(x, stats=True)
            M = t.shape[0] * t.shape[1] * t.shape[2]
            scale, shift, mean, invstd = bn.scale_shift(parts, M, True)
            if post_add is None:
                y = K.affine_act(t, scale, shift, act=self.act, out=out)
            else:
                y = K.dual_affine_act(t, scale, shift, post_add=post_add, act=self.act, out=out)
            self._ctx = (x, t, scale, shift, mean, invstd)
            self._req = None
            return y
        if self._folded is not None and post_add is None:
            return K.conv2d_fwd(x, self._folded[0], bias=self._folded[1], out=out, act=self.act, stride=conv.stride, pad=conv.padding)
        t = conv.conv(x)
        scale, shift, _, _ = bn.scale_shift(None, 0, False)
        if post_add is not None:
            return K.dual_affine_act(t, scale, shift, post_add=post_add, act=self.act, out=out if out is not None else t)
        return K.affine_act(t, scale, shift, act=self.act, out=out if out is not None else t)

    def bn_reduce_request(self):
        """This layer's BatchNorm-backward reduce as a request for the data-gradient launch that finalises its output gradient (the caller
        hands it to that launch; bwd() then finds the partial sums ready and skips its own reduce sweep).  None when it cannot be handed over
        (synchronised BatchNorm reduces across ranks on its own path)."""
        conv, bn = self._parts()
        if self._ctx is None or bn._synced() or not self._net.fuse_bn_reduce:
            return None
        _, t, scale, shift, mean, _ = self._ctx
        self._req = K.BnReduceRequest(t, scale, shift, mean, self.act)
        return self._req

    def bwd(self, dy, dx_out=None, accumulate=False, addend=None, need_dx=True, dx_req=None):
        """dx_req: reduce requests of the layer(s) whose output gradient is this call's dx (ConvLayer.dgrad)"""
        conv, bn = self._parts()
        (x, t, scale, shift, mean, invstd), self._ctx = self._ctx, None
        req, self._req = getattr(self, "_req", None), None
        parts = req.parts if req is not None else None
        dt = bn.backward(dy, t, scale, shift, mean, invstd, self.act, dx_out=t, parts=parts)  # in place over the saved conv output
        conv.wgrad(x, dt)
        if not need_dx:
            return None
        return conv.dgrad(dt, tuple(x.shape), out=dx_out, accumulate=accumulate, addend=addend, reqs=dx_req)


class Conv(_ConvBN):
    """Reference `Conv` (conv_bn_act_block.py:80): Conv2d(k, stride, autopad, bias=False) + BatchNorm2d + activation."""

    def __init__(self, input_channels, output_channels, kernel, stride, activation_type, padding: int = None, groups: int = None):
        super().__init__()
        if groups not in (None, 1):
            raise NotImplementedError("grouped convolution is outside the MI355X hot path (SURVEY.md 8: groups=1 only)")
        pad = kernel // 2 if padding is None else padding
        self.conv = ConvLayer(input_channels, output_channels, kernel, stride, pad, bias=False)
        self.bn = BatchNorm(output_channels)
        self.act = act_name(activation_type)

    def _parts(self):
        return self.conv, self.bn


class _Seq(nn.Module):
    """Namespace module so that keys read seq.conv.weight / seq.bn.* like the reference's nn.Sequential."""


class ConvBNAct(_ConvBN):
    """Reference `ConvBNAct` (conv_bn_act_block.py:9-69), supported subset: groups=1, dilation=1, zero padding, use_normalization=True."""

    def __init__(self, in_channels, out_channels, kernel_size, padding=0, activation_type=None, stride=1, dilation=1, groups=1, bias=True,
                 padding_mode="zeros", use_normalization=True, eps=1e-5, momentum=0.1, affine=True, track_running_stats=True,
                 device=None, dtype=None, activation_kwargs=None):
        super().__init__()
        if groups != 1 or dilation != 1 or padding_mode != "zeros" or not use_normalization or not affine or not track_running_stats:
            raise NotImplementedError("ConvBNAct on the HIP path supports groups=1, dilation=1, zero padding, affine BN with running stats")
        if bias:
            raise NotImplementedError("conv bias followed by BatchNorm is redundant; the HIP path implements bias=False (as all hot-path call sites use)")
        self.seq = _Seq()
        self.seq.add_module("conv", ConvLayer(in_channels, out_channels, kernel_size, stride, padding, bias=False))
        self.seq.add_module("bn", BatchNorm(out_channels, eps=eps, momentum=momentum))
        self.act = act_name(activation_type)

    def _parts(self):
        return self.seq.conv, self.seq.bn


class ConvBNReLU(ConvBNAct):
    """Reference `ConvBNReLU` (conv_bn_relu_block.py:8-60)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True, padding_mode="zeros",
                 use_normalization=True, eps=1e-5, momentum=0.1, affine=True, track_running_stats=True, device=None, dtype=None,
                 use_activation=True, inplace=False):
        super().__init__(in_channels, out_channels, kernel_size, padding=padding, activation_type="relu" if use_activation else None, stride=stride,
                         dilation=dilation, groups=groups, bias=bias, padding_mode=padding_mode, use_normalization=use_normalization, eps=eps,
                         momentum=momentum, affine=affine, track_running_stats=track_running_stats)

"""RepVGGBlock (training form and fused deployment form) on the HIP kernels.

Reference: modules/repvgg_block.py:8-232 -
    y = act(se( bn3(conv3x3(x)) + alpha * bn1(conv1x1(x)) + [bn_id(x)] ))
with state_dict keys branch_3x3.{conv.weight,bn.*}, branch_1x1.{conv.weight,bn.*} (+ alpha when use_alpha).  Supported subset = what
PP-YOLOE / PP-YOLOE+ build (csp_resnet.py:38-40): no SE, no identity-BN branch (use_residual_connection=False or in != out), groups =
dilation = 1; `use_alpha=True` (PP-YOLOE+, repvgg_block.py:31,77-87): the learnable [1] multiplier of the 1x1 branch.

Kernel sequence (training): the two convolutions run side by side (1x1 on the side stream), each emitting its BatchNorm partial
statistics from the conv epilogue; two tiny finalizes; ONE sweep computes act(s3*t3 + b3 + s1*t1 + b1) [+ residual] - the reference
runs 2 conv + 2 BN + add + activation (+ add).  Backward: one sweep for the gradient through the activation (pre-activation
recomputed from the saved conv outputs) that also leaves the reduce rows of both BatchNorm backward passes (round 5), then the two
BatchNorm backward applies in place over t3 / t1, weight gradients on the side stream, and the 1x1 data gradient accumulated into the
3x3 one.
"""
import torch
from torch import nn

from .. import kernels as K
from .engine import SgxBlock
from .layers import BatchNorm, ConvLayer, act_name


_FUSED_REDUCE = __import__("os").environ.get("SGX_REPVGG_FUSED_REDUCE", "1") != "0"


class _ConvBNBranch(nn.Module):
    """Namespace so that keys read branch_*.conv.weight / branch_*.bn.* like the reference's nn.Sequential (repvgg_block.py:211-232)."""


class RepVGGBlock(SgxBlock):
    def __init__(self, in_channels, out_channels, stride=1, dilation=1, groups=1, activation_type=nn.ReLU, activation_kwargs=None, se_type=nn.Identity,
                 se_kwargs=None, build_residual_branches=True, use_residual_connection=True, use_alpha=False):
        super().__init__()
        if dilation != 1 or groups != 1:
            raise NotImplementedError("RepVGGBlock on the HIP path: dilation=1, groups=1")
        if se_type not in (None, nn.Identity):
            raise NotImplementedError("RepVGGBlock on the HIP path: no SE block inside (PP-YOLOE passes nn.Identity)")
        if use_residual_connection and in_channels == out_channels and stride == 1:
            raise NotImplementedError("RepVGGBlock with the identity-BatchNorm branch (RepVGG classifiers) is not on the HIP path; PP-YOLOE builds "
                                      "its blocks with use_residual_connection=False")
        if not build_residual_branches:
            raise NotImplementedError("build a training-form block and call fuse_block_residual_branches() for the deployment form")
        self.in_channels, self.out_channels, self.stride, self.groups = in_channels, out_channels, stride, groups
        self.act = act_name(activation_type)
        # reference :77-87: a learnable [1] multiplier of the 1x1 branch, initialised at 1 + N(0, 0.01^2); else the constant 1
        self.alpha = nn.Parameter(torch.tensor([1.0]) + torch.randn((1,)) * 0.01, requires_grad=True) if use_alpha else 1
        self.no_conv_branch = None
        self.branch_3x3 = _ConvBNBranch()
        self.branch_3x3.add_module("conv", ConvLayer(in_channels, out_channels, 3, stride, 1, bias=False))
        self.branch_3x3.add_module("bn", BatchNorm(out_channels))
        self.branch_1x1 = _ConvBNBranch()
        self.branch_1x1.add_module("conv", ConvLayer(in_channels, out_channels, 1, stride, 0, bias=False))
        self.branch_1x1.add_module("bn", BatchNorm(out_channels))
        self.build_residual_branches = True
        self._fused_w = self._fused_b = None

    def on_materialize(self):
        pass

    def fwd(self, x, out=None, post_add=None):
        """post_add: added after the activation (CSPResNetBasicBlock's `x + y`, csp_resnet.py:43-49)."""
        c3, bn3, c1, bn1 = self.branch_3x3.conv, self.branch_3x3.bn, self.branch_1x1.conv, self.branch_1x1.bn
        if not self.build_residual_branches:  # deployment form: one 3x3 convolution with fused bias + activation
            if self.training:
                raise RuntimeError("a fused RepVGGBlock is inference-only on the HIP path (re-parameterised training is outside the hot path)")
            if x.dtype == K.HALF:  # half-precision inference: bias, activation and the block's `x + y` in the bf16 convolution's epilogue
                return K.conv2d_fwd(x, self._fused_w, bias=self._fused_b, out=out, act=self.act, stride=self.stride, pad=1, post_add=post_add)
            y = K.conv2d_fwd(x, self._fused_w, bias=self._fused_b, out=out if post_add is None else None, act=self.act, stride=self.stride, pad=1)
            return y if post_add is None else K.affine_act(y, r1=post_add, out=out if out is not None else y)
        if self.training:
            t1 = torch.empty(K.conv_out_shape(x, self.out_channels, 1, 1, self.stride, 0), device=x.device, dtype=torch.float32)
            _, parts1 = self._net.fork_side(lambda: c1.conv(x, out=t1, stats=True), x, t1)
            t3, parts3 = c3.conv(x, stats=True)
            M = t3.shape[0] * t3.shape[1] * t3.shape[2]
            s3, b3, m3, i3 = bn3.scale_shift(parts3, M, True)
            self._net.join_side()
            s1, b1, m1, i1 = bn1.scale_shift(parts1, M, True)
            s1a, b1a = self._scaled(s1, b1)  # alpha * bn1(.) = (alpha s1) t1 + alpha b1: alpha rides in the sweep's per-channel constants
            y = K.dual_affine_act(t3, s3, b3, t1, s1a, b1a, post_add=post_add, act=self.act, out=out)
            self._ctx = (x, t3, t1, s3, b3, m3, i3, s1, b1, m1, i1)
            return y
        t3, t1 = c3.conv(x), c1.conv(x)
        s3, b3, _, _ = bn3.scale_shift(None, 0, False)
        s1, b1, _, _ = bn1.scale_shift(None, 0, False)
        s1, b1 = self._scaled(s1, b1)
        return K.dual_affine_act(t3, s3, b3, t1, s1, b1, post_add=post_add, act=self.act, out=out if out is not None else t3)

    def _scaled(self, s1, b1):
        if not isinstance(self.alpha, torch.Tensor):
            return s1, b1
        a = self.alpha.detach()
        return s1 * a, b1 * a  # ([C] vectors on the device)

    def bwd(self, dy, dx_out=None, accumulate=False, addend=None, need_dx=True):
        c3, bn3, c1, bn1 = self.branch_3x3.conv, self.branch_3x3.bn, self.branch_1x1.conv, self.branch_1x1.bn
        (x, t3, t1, s3, b3, m3, i3, s1, b1, m1, i1), self._ctx = self._ctx, None
        s1a, b1a = self._scaled(s1, b1)
        # one sweep: the gradient through the activation AND the reduce rows of both BatchNorm backward passes (round 5: two passes over
        # g and the saved conv outputs less per block)
        if _FUSED_REDUCE:
            g, parts3, parts1 = K.dual_affine_act_bwd_reduce(dy, t3, s3, b3, m3, t1, s1a, b1a, m1, act=self.act)
        else:  # (measurement: SGX_REPVGG_FUSED_REDUCE=0 - each BatchNorm backward runs its own reduce sweep)
            g, parts3, parts1 = K.dual_affine_act_bwd(dy, t3, s3, b3, t1, s1a, b1a, act=self.act), None, None
        if isinstance(self.alpha, torch.Tensor):
            # The BatchNorm backward is linear in its upstream gradient (alpha g here): run it on g with scratch parameter gradients, then
            #   d gamma1 = alpha dg', d beta1 = alpha db', d t1 = alpha dt1'   and   d alpha = <g, bn1(t1)> = sum_c (gamma1 dg' + beta1 db')
            # - exact for every alpha (zero included); one extra in-place pass over the 1x1 branch's gradient.
            dg, db = torch.zeros_like(bn1.weight), torch.zeros_like(bn1.bias)
            dt1 = K.bn_bwd(g, t1, s1, b1, bn1.weight, m1, i1, dg, db, act=None, dx_out=t1, sync=bn1._synced(), parts=parts1)
            a = self.alpha.detach()
            self.alpha.grad.add_((bn1.weight.detach() * dg + bn1.bias.detach() * db).sum())
            bn1.weight.grad.add_(dg * a)
            bn1.bias.grad.add_(db * a)
            dt1 = K.axpy(dt1, a_dev=self.alpha, out=dt1)
        else:
            dt1 = bn1.backward(g, t1, s1, b1, m1, i1, None, dx_out=t1, parts=parts1)   # in place over the saved conv outputs
        c1.wgrad(x, dt1)
        dt3 = bn3.backward(g, t3, s3, b3, m3, i3, None, dx_out=t3, parts=parts3)
        c3.wgrad(x, dt3)
        if not need_dx:
            return None
        shape = tuple(x.shape)
        dx = c3.dgrad(dt3, shape, out=dx_out, accumulate=accumulate, addend=addend)
        return c1.dgrad(dt1, shape, out=dx, accumulate=True)

    # ---- re-parameterisation (reference: repvgg_block.py:109-209) ---------------------------------------------------------
    @staticmethod
    def _fuse_bn_tensor(branch):
        bn = branch.bn
        std = (bn.running_var + bn.eps).sqrt()
        t = (bn.weight.detach() / std).reshape(-1, 1, 1, 1)
        return branch.conv.weight.detach() * t, bn.bias.detach() - bn.running_mean * bn.weight.detach() / std

    def _get_equivalent_kernel_bias(self):
        k3, b3 = self._fuse_bn_tensor(self.branch_3x3)
        k1, b1 = self._fuse_bn_tensor(self.branch_1x1)
        alpha = self.alpha.detach() if isinstance(self.alpha, torch.Tensor) else self.alpha
        return k3 + alpha * torch.nn.functional.pad(k1, [1, 1, 1, 1]), b3 + alpha * b1

    def fuse_block_residual_branches(self):
        """Training form -> one 3x3 conv + bias (`rbr_reparam`, as in the reference).  Unlike the reference the branch modules stay
        (the arenas own their storage); forward switches to the fused kernel and the block becomes inference-only."""
        if not self.build_residual_branches:
            return
        kernel, bias = self._get_equivalent_kernel_bias()
        self.rbr_reparam = nn.Conv2d(self.in_channels, self.out_channels, 3, self.stride, 1, bias=True).to(kernel.device)
        self.rbr_reparam.weight.data = kernel.contiguous()
        self.rbr_reparam.bias.data = bias.contiguous()
        for p in self.rbr_reparam.parameters():
            p.requires_grad_(False)
        k, c = kernel.shape[:2]
        cp = (c + 3) // 4 * 4   # physical form for the kernels: OHWI, channel axis padded to 4 floats
        w = torch.zeros(k, 3, 3, cp, device=kernel.device, dtype=torch.float32)
        w[..., :c] = kernel.permute(0, 2, 3, 1)
        self._fused_w, self._fused_b = w.permute(0, 3, 1, 2), bias.contiguous().float()
        self.build_residual_branches = False

    def prep_model_for_conversion(self, input_size=None, **kwargs):
        self.fuse_block_residual_branches()

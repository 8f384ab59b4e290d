"""QARepVGGBlock (training form) on the HIP kernels.

Reference: modules/qarepvgg_block.py:10-204 -
    y = act(post_bn( bn3(conv3x3(x)) + alpha * (conv1x1(x) + b) + [x] ))
with state_dict keys branch_3x3.{conv.weight,bn.*}, branch_1x1.{weight,bias}, post_bn.*, rbr_reparam.{weight,bias}
(the last one is the reference's unused deployment placeholder, qarepvgg_block.py:166-176: kept so checkpoints load,
never touched by any kernel, excluded from optimizer / all-reduce / EMA).

Kernel sequence (training) - 3 launches forward, 7 backward, against the reference's 2 conv + 2 BN + 2 add + ReLU ops and their autograd:
  y3, u = conv3x3(x), conv1x1(x; W1 + I) + b   ONE launch (the 1x1 filter reads the 3x3 filter's centre tap; the identity branch is folded
                                                into the 1x1 filter once per step); its epilogue leaves the five moments of (y3, u)
  finalize                                      BOTH BatchNorms' statistics from those moments: s = bn3(y3) + u is affine in (y3, u)
  out = act(a*y3 + scale_p*u + c)               ONE sweep; s itself is never written
Backward: one reduce sweep over (dout, y3, u) (four sums), finalize, one apply sweep that writes the gradients of u and y3 in place; the two
weight gradients; dx = dgrad3x3(dy3) + dgrad1x1(ds) as ONE launch (two K-axis sources, no accumulate pass over dx).
Blocks this form does not cover (learnable alpha, fewer than 16 channels, synchronised BatchNorm) run the general sequence:
  t3 = conv3x3(x) (+ BN3 partial statistics), t1 = conv1x1(x) + b, s = scale3*t3 + shift3 + alpha*t1 + x (one sweep, post_bn partial
  statistics), y = act(scale_p*s + shift_p) (one sweep); backward: post_bn and BN3 backward in place, dx = dgrad3x3 + dgrad1x1 + ds.
"""
from torch import nn
import os

import torch

from .. import kernels as K
from .engine import SgxBlock
from .layers import BatchNorm, ConvLayer, act_name


class _Branch(nn.Module):
    pass


_TAIL_CHUNKS = int(os.environ.get("SGX_STEM_BWD_CHUNKS", "1"))  # measurement switch (r6ad: 2 runs 794.0 against 793.4 images/s, 4 and 8 slower - the backlog behind the last sweep is stage 1's, not the stem's)


class QARepVGGBlock(SgxBlock):
    def __init__(self, in_channels, out_channels, stride=1, dilation=1, groups=1, activation_type="relu", activation_kwargs=None,
                 se_type=None, se_kwargs=None, build_residual_branches=True, use_residual_connection=True, use_alpha=False,
                 use_1x1_bias=True, use_post_bn=True):
        super().__init__()
        if dilation != 1 or groups != 1:
            raise NotImplementedError("QARepVGGBlock on the HIP path: dilation=1, groups=1")
        if se_type not in (None, nn.Identity):
            raise NotImplementedError("QARepVGGBlock on the HIP path: no SE block (YOLO-NAS uses none)")
        if not (build_residual_branches and use_1x1_bias and use_post_bn):
            raise NotImplementedError("QARepVGGBlock on the HIP path implements the training form (3 branches, 1x1 bias, post-BN)")
        self.in_channels, self.out_channels, self.stride = in_channels, out_channels, stride
        self.act = act_name(activation_type)
        # reference :130-136: a learnable [1] multiplier of the 1x1 branch when use_alpha, else the float 1.0 (YOLO-NAS recipes: False)
        self.alpha = nn.Parameter(torch.tensor([1.0]), requires_grad=True) if use_alpha else 1.0
        self.branch_3x3 = _Branch()
        self.branch_3x3.add_module("conv", ConvLayer(in_channels, out_channels, 3, stride, 1, bias=False))
        self.branch_3x3.add_module("bn", BatchNorm(out_channels))
        self.branch_1x1 = ConvLayer(in_channels, out_channels, 1, stride, 0, bias=True)
        self.use_residual_connection = bool(use_residual_connection and in_channels == out_channels and stride == 1)
        self.post_bn = BatchNorm(out_channels)
        # reference placeholder (requires_grad=True there, never used in training mode): state_dict compatibility; it receives
        # the fused kernel / bias in partial_fusion() / full_fusion() exactly as in the reference
        self.rbr_reparam = nn.Conv2d(in_channels, out_channels, 3, stride, 1, bias=True)
        self.partially_fused = self.fully_fused = False
        self._fused_w = self._fused_b = None

    def on_materialize(self):
        self._w1p = self._w1pt = None

    def qarep_prep_job(self):
        """-> the block's sgx_qarep_prep_job record (called once by SgxNetwork.materialize, after every layer owns its arena views), or None
        when the block runs the general sequence.  Allocates the persistent buffers of the two-branch-per-launch form: W1 + I and its
        transpose, refreshed once per step by the network's sgx_qarep_prep_batch launch (engine.prefetch_dgrad_weights)."""
        c1 = self.branch_1x1
        if isinstance(self.alpha, torch.Tensor) or float(self.alpha) != 1.0 or self.out_channels < 16 or not self._net.wt_batch:
            return None
        # (round 6: blocks with fewer than 16 input channels - the RGB stem - take the two-output launch too, on the flattened K axis, while
        # their filters fit its 64-wide tiles; they produce no input gradient, so the two-source data gradient is never asked of them)
        if self.in_channels < 16 and (self.out_channels > 64 or self.use_residual_connection or os.environ.get("SGX_QAREP_STEM_DUAL", "1") == "0"):
            return None  # (SGX_QAREP_STEM_DUAL=0: measurement switch, the general sequence of rounds 1 - 5)
        K_, C_ = c1._w.shape[0], c1._w.shape[1]
        self._w1p = K.ohwi_empty(K_, C_, 1, 1, c1._w.device)
        self._w1pt = torch.empty(C_, K_, device=c1._w.device, dtype=torch.float32)
        return K.qarep_prep_job(c1._w, self._w1p, self._w1pt, self.use_residual_connection)

    def _two_branch_launch(self):
        return (self._w1p is not None and self._net._wt_valid and not (self.branch_3x3.bn._synced() or self.post_bn._synced()))

    def _alpha(self):
        """-> (host float, device scalar or None) as the sweeps take it"""
        return (1.0, self.alpha) if isinstance(self.alpha, torch.Tensor) else (float(self.alpha), None)

    def fwd(self, x, out=None, post_add=None, post_scale=None):
        """post_add / post_scale: out = block(x) + post_scale * post_add (the YOLO-NAS bottleneck's shortcut, yolo_stages.py:61-63),
        written by the block's last sweep on the two-branch path."""
        if x.dtype == K.HALF:  # half-precision inference: the fully fused deployment form, shortcut included, as ONE bf16 launch
            if self.training or not self.fully_fused:
                raise RuntimeError("half-precision inference runs the fully fused deployment form: call prep_model_for_conversion(full_fusion=True) "
                                   "in eval mode first")
            return K.conv2d_fwd(x, self._fused_w, bias=self._fused_b, out=out, act=self.act, stride=self.stride, pad=1, post_add=post_add,
                                post_scale=post_scale)
        if post_add is not None and not (self.training and self._two_branch_launch()):
            y = self.fwd(x)
            a, a_dev = (1.0, post_scale) if torch.is_tensor(post_scale) else (1.0 if post_scale is None else float(post_scale), None)
            return K.affine_act(y, r1=post_add, a1=a, a1_dev=a_dev, out=out if out is not None else y)
        c3, bn3, c1, pbn = self.branch_3x3.conv, self.branch_3x3.bn, self.branch_1x1, self.post_bn
        res = x if self.use_residual_connection else None
        a, a_dev = self._alpha()
        if self.training:
            if self.partially_fused or self.fully_fused:
                raise RuntimeError("a fused QARepVGGBlock is inference-only on the HIP path (the reference's fused block trains a single conv; "
                                   "re-parameterised training is outside the hot path)")
            if self._two_branch_launch():
                y3, u, stat5 = K.conv2d_fwd_dual(x, c3._w, self._w1p, c1.bias, stride=self.stride)
                cf, sv = K.qarep_fwd_finalize(stat5, y3.shape[0] * y3.shape[1] * y3.shape[2], c1.bias, bn3, pbn)
                a3, c3_, ap, cp = cf.unbind(0)  # the four operand rows of the forward sweep
                y = K.dual_affine_act(y3, a3, c3_, u, ap, cp, act=self.act, out=out, post_add=post_add, post_scale=post_scale)
                self._ctx = ("dual", x, y3, u, cf, sv)
                return y
            # the two branches read the same x and are independent: the 1x1 branch runs on the side stream beside the 3x3 one
            t1 = torch.empty(K.conv_out_shape(x, self.out_channels, 1, 1, self.stride, 0), device=x.device, dtype=torch.float32)
            self._net.fork_side(lambda: c1.conv(x, out=t1), x, t1)
            t3, parts = c3.conv(x, stats=True)
            M = t3.shape[0] * t3.shape[1] * t3.shape[2]
            sc3, sh3, m3, i3 = bn3.scale_shift(parts, M, True)
            self._net.join_side()
            # s overwrites t1 - unless alpha is learnable: its gradient is <ds, t1>, so the 1x1 branch output is kept
            s, parts_s = K.affine_act(t3, sc3, sh3, r1=t1, a1=a, a1_dev=a_dev, r2=res, a2=1.0, out=t1 if a_dev is None else None, want_stats=True)
            scp, shp, mp, ip = pbn.scale_shift(parts_s, M, True)
            y = K.affine_act(s, scp, shp, act=self.act, out=out)
            self._ctx = (x, t3, s, sc3, sh3, m3, i3, scp, shp, mp, ip, t1 if a_dev is not None else None)
            return y
        if self.fully_fused:      # deployment form: ONE 3x3 convolution with fused bias + activation
            return K.conv2d_fwd(x, self._fused_w, bias=self._fused_b, out=out, act=self.act, stride=self.stride, pad=1)
        if self.partially_fused:  # branches fused, post_bn kept
            t = K.conv2d_fwd(x, self._fused_w, bias=self._fused_b, stride=self.stride, pad=1)
            scp, shp, _, _ = pbn.scale_shift(None, 0, False)
            return K.affine_act(t, scp, shp, act=self.act, out=out if out is not None else t)
        t3 = c3.conv(x)
        sc3, sh3, _, _ = bn3.scale_shift(None, 0, False)
        t1 = c1.conv(x)
        s = K.affine_act(t3, sc3, sh3, r1=t1, a1=a, a1_dev=a_dev, r2=res, a2=1.0, out=t1)
        scp, shp, _, _ = pbn.scale_shift(None, 0, False)
        return K.affine_act(s, scp, shp, act=self.act, out=out if out is not None else s)

    # ---- re-parameterisation (reference: qarepvgg_block.py:206-321) -------------------------------------------------------
    @staticmethod
    def _fuse_bn_tensor(kernel, bias, running_mean, running_var, gamma, beta, eps):
        std = torch.sqrt(running_var + eps)
        a = gamma / std
        return kernel * a.reshape(-1, 1, 1, 1), bias * a + (beta - gamma * running_mean / std)

    def _get_equivalent_kernel_bias_for_branches(self):
        """3x3 (with its BN folded) + alpha * zero-padded 1x1 (+ bias) + identity, as one 3x3 kernel / bias."""
        bn3 = self.branch_3x3.bn
        w3 = self.branch_3x3.conv.weight.detach()
        k3, b3 = self._fuse_bn_tensor(w3, 0.0, bn3.running_mean, bn3.running_var, bn3.weight.detach(), bn3.bias.detach(), bn3.eps)
        k1 = torch.nn.functional.pad(self.branch_1x1.weight.detach(), [1, 1, 1, 1])
        alpha = self.alpha.detach().to(k1.device) if isinstance(self.alpha, torch.Tensor) else self.alpha
        k = k3 + alpha * k1
        if self.use_residual_connection:
            cin = self.in_channels
            ident = torch.zeros(cin, cin, 3, 3, device=k.device, dtype=k.dtype)
            ident[torch.arange(cin), torch.arange(cin), 1, 1] = 1.0
            k = k + ident
        return k, b3 + alpha * self.branch_1x1.bias.detach()

    def _install_fused(self, kernel, bias):
        self.rbr_reparam.weight.data = kernel.to(self.rbr_reparam.weight.device if not getattr(self, "_net", None) else kernel.device).contiguous()
        self.rbr_reparam.bias.data = bias.contiguous()
        # physical form for the kernels: OHWI with the channel axis padded to 4 floats like every other conv weight
        k, c = kernel.shape[:2]
        cp = (c + 3) // 4 * 4
        w = torch.zeros(k, 3, 3, cp, device=kernel.device, dtype=torch.float32)
        w[..., :c] = kernel.permute(0, 2, 3, 1)
        self._fused_w = w.permute(0, 3, 1, 2)
        self._fused_b = bias.contiguous().float()

    def partial_fusion(self):
        """Fuse the branches into one kernel, keep post_bn (reference :281-307).  Unlike the reference the branch modules are kept
        (the arenas own their storage); eval-mode forward switches to the fused kernel."""
        if self.partially_fused:
            return
        if self.fully_fused:
            raise NotImplementedError("QARepVGGBlock can't be converted to partially fused from fully fused")
        self._install_fused(*self._get_equivalent_kernel_bias_for_branches())
        self.partially_fused, self.fully_fused = True, False

    def full_fusion(self):
        """Fuse everything into conv + bias + activation (reference :253-279); the block becomes inference-only."""
        if self.fully_fused:
            return
        if not self.partially_fused:
            self.partial_fusion()
        pbn = self.post_bn
        k, b = self._fuse_bn_tensor(self.rbr_reparam.weight.detach(), self.rbr_reparam.bias.detach(), pbn.running_mean, pbn.running_var, pbn.weight.detach(),
                                    pbn.bias.detach(), pbn.eps)
        self._install_fused(k, b)
        self.partially_fused, self.fully_fused = False, True

    def fuse_block_residual_branches(self):
        self.partial_fusion()

    def prep_model_for_conversion(self, input_size=None, full_fusion: bool = False, **kwargs):
        if full_fusion:
            self.full_fusion()
        else:
            self.partial_fusion()

    def bwd(self, dy, dx_out=None, accumulate=False, addend=None, need_dx=True, addend2=None, addend2_scale=None, dx_req=None):
        """addend (dx's strides) and addend2_scale * addend2 (its own strides) are added to dx; on the two-branch path both ride in
        the data-gradient launch's epilogue - and so do the BatchNorm-backward reduce requests (dx_req) of the layer(s) whose output
        gradient dx is."""
        c3, bn3, c1, pbn = self.branch_3x3.conv, self.branch_3x3.bn, self.branch_1x1, self.post_bn
        if addend2 is not None and need_dx and not (self._ctx[0] == "dual" and self.stride == 1 and K.nhwc_strides(addend2)[0] % 4 == 0):
            a, a_dev = (1.0, addend2_scale) if torch.is_tensor(addend2_scale) else (1.0 if addend2_scale is None else float(addend2_scale), None)
            dx_out = K.axpy(addend2, a=a, a_dev=a_dev, out=dx_out, accumulate=accumulate and dx_out is not None)
            accumulate, addend2 = True, None
        if self._ctx[0] == "dual":
            (_, x, y3, u, cf, sv), self._ctx = self._ctx, None
            if not need_dx and _TAIL_CHUNKS > 1 and x.shape[0] % _TAIL_CHUNKS == 0:
                # The first block of a network (no data gradient): its two weight gradients are the last work of the step, and the main chain has
                # nothing left to run beside them - the apply sweep goes out in runs of images, each run's weight gradients behind it, so that
                # only the last run's are left when the sweep ends (r6fin2's trace: 0.43 ms of stem weight gradients behind the last sweep).
                def run_done(i, n0, n1):
                    c1.wgrad(x[n0:n1], u[n0:n1], bias_grad=False)
                    c3.wgrad(x[n0:n1], y3[n0:n1])
                    self._net.flush_wgrads()

                K.qarep_bwd(dy, y3, u, cf, sv, bn3, pbn, self.act, chunks=_TAIL_CHUNKS, after_chunk=run_done)
                return None
            ds, dy3 = K.qarep_bwd(dy, y3, u, cf, sv, bn3, pbn, self.act)   # in place over u / y3
            c1.wgrad(x, ds, bias_grad=False)  # d b1 = sum ds = 0: post_bn's input gradient sums to zero per channel
            c3.wgrad(x, dy3)
            if not need_dx:
                return None
            return K.conv2d_bwd_data_dual(dy3, c3._w, c3._wt, ds, self._w1pt, tuple(x.shape), stride=self.stride, addend=addend, out=dx_out,
                                          accumulate=accumulate, addend2=addend2, addend2_scale=addend2_scale,
                                          reqs=dx_req if self._net.fuse_bn_reduce else None)
        (x, t3, s, sc3, sh3, m3, i3, scp, shp, mp, ip, t1), self._ctx = self._ctx, None
        ds = pbn.backward(dy, s, scp, shp, mp, ip, self.act, dx_out=s)          # in place over s
        ds1 = ds                                                                # gradient of the 1x1 branch output: alpha * ds
        if t1 is not None:                                                      # learnable alpha: d alpha = <ds, conv1x1(x) + b>
            K.dot_sum(t1, ds, self.alpha.grad, accumulate=True)
            ds1 = K.axpy(ds, a_dev=self.alpha, out=t1)                          # in place over t1
        # d b1 = sum ds1 = alpha * sum ds = 0: ds is post_bn's input gradient, which sums to zero per channel (as on the two-branch path above).
        # (Round 6: this path - the RGB stem's - still took the sum: a 629 MB column sum at 320 x 320 in the step's tail, where the main stream
        # has nothing left to run beside it, for a value that is zero up to round-off.  r6p, three interleaved pairs on one box: 749.0 / 753.5 / 753.2
        # -> 754.2 / 754.8 / 754.2 images/s; flushing the 1x1 weight gradient out ahead of the second BatchNorm backward as well: 755.4 on average, not kept.)
        c1.wgrad(x, ds1, bias_grad=False)
        dt3 = bn3.backward(ds, t3, sc3, sh3, m3, i3, None, dx_out=t3)   # in place over t3
        c3.wgrad(x, dt3)
        if not need_dx:
            return None
        shape = tuple(x.shape)
        if self.use_residual_connection:
            if dx_out is not None and K.nhwc_strides(dx_out) != K.nhwc_strides(ds):  # a strided destination: the epilogue addend can't be ds
                K.axpy(ds, out=dx_out, accumulate=accumulate)
                dx = c3.dgrad(dt3, shape, out=dx_out, accumulate=True)
            else:
                dx = c3.dgrad(dt3, shape, out=dx_out, accumulate=accumulate, addend=ds)
            return c1.dgrad(ds1, shape, out=dx, accumulate=True, addend=addend)
        dx = c3.dgrad(dt3, shape, out=dx_out, accumulate=accumulate, addend=addend)
        return c1.dgrad(ds1, shape, out=dx, accumulate=True)

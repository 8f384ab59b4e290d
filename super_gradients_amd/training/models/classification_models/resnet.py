"""ResNet / CifarResNet on the HIP kernels.

Reference: training/models/classification_models/resnet.py - BasicResNetBlock (:26-50), Bottleneck (:53-84), CifarResNet
(:87-137), ResNet (:140-260), the registered variants (:263-375).  Same constructor arguments for the supported subset,
same state_dict keys (conv1.weight, bn1.*, layer{i}.{j}.{conv1,bn1,conv2,bn2[,conv3,bn3]}, layer{i}.{j}.shortcut.{0,1}, linear.*).

Kernel sequence of a block (training):
    t1 = conv1(x)  [BN statistics from the conv epilogue]   a1 = relu(bn1(t1))                 one sweep
    t2 = conv2(a1) ...                                      (bottleneck: one more conv/BN/ReLU)
    shortcut: identity, or ts = conv_s(x) -> u = bn_s(ts)                                       one sweep
    out = relu(bn_last(t_last) + shortcut)                                                      one sweep (affine + residual + act)
Backward: g = dy * [out > 0] (sgx_relu_bwd: the ReLU follows the residual add, so its mask comes from `out`), then the BN
backward sweeps / weight gradients / data gradients of the branch, the shortcut gradient folded into the data-gradient
epilogue (identity) or produced by the shortcut conv's own backward.
"""
import os
from typing import Dict

import torch
from torch import nn

from .... import kernels as K
from ....common.registry import register_model
from ....modules.engine import SgxBlock, SgxNetwork
from ....modules.layers import BatchNorm, ConvLayer, LinearLayer, MaxPool
from ...utils.utils import get_param


def width_multiplier(original, factor, to_int=True):
    return int(original * factor) if to_int else original * factor


class _Seq(nn.Module):
    """Namespace so that the shortcut's keys read shortcut.0.weight / shortcut.1.* like the reference's nn.Sequential."""

    def __len__(self):
        return len(self._modules)


class _ConvBN:
    """conv -> training-mode BN bookkeeping shared by the blocks (not a module: the owners register conv/bn themselves)."""

    @staticmethod
    def fwd(conv: ConvLayer, bn: BatchNorm, x, training):
        if training:
            t, parts = conv.conv(x, stats=True)
            M = t.shape[0] * t.shape[1] * t.shape[2]
            return (t,) + tuple(bn.scale_shift(parts, M, True))
        t = conv.conv(x)
        return (t,) + tuple(bn.scale_shift(None, 0, False))


_RELU_REDUCE = os.environ.get("SGX_RESNET_RELU_REDUCE", "1") != "0"  # measurement switch (r6aa): 0 = mask sweep + reduce sweep


class _ResBlock(SgxBlock):
    """out = [relu]( bn_n(conv_n(... relu(bn_1(conv_1(x))) ...)) + shortcut(x) )"""

    def _branch(self):
        raise NotImplementedError

    def on_materialize(self):
        pass

    def _init_shortcut(self, in_planes, out_planes, stride):
        self.shortcut = _Seq()
        if stride != 1 or in_planes != out_planes:
            self.shortcut.add_module("0", ConvLayer(in_planes, out_planes, 1, stride, 0, bias=False))
            self.shortcut.add_module("1", BatchNorm(out_planes))

    def _shortcut_fwd(self, x):
        ts, scs, shs, ms, ivs = _ConvBN.fwd(getattr(self.shortcut, "0"), getattr(self.shortcut, "1"), x, self.training)
        return K.affine_act(ts, scs, shs), (ts, scs, shs, ms, ivs)

    def fwd(self, x, out=None):
        branch = self._branch()
        saved = []
        a = x
        # the projection shortcut (conv -> BatchNorm, three short dependent kernels) meets the main branch in its last sweep: branch stream
        # (engine.fork_branch, site 16), joined there
        net = getattr(self, "_net", None)
        forked = None
        self._branched = bool(len(self.shortcut)) and net is not None and self.training and net.branches(16, x.shape[0] * x.shape[1] * x.shape[2], 64)
        if self._branched:
            forked = net.fork_branch(lambda: self._shortcut_fwd(x))
        for i, (conv, bn) in enumerate(branch):
            t, sc, sh, mean, invstd = _ConvBN.fwd(conv, bn, a, self.training)
            last = i == len(branch) - 1
            if not last:
                nxt = K.affine_act(t, sc, sh, act="relu", out=None if self.training else t)
                saved.append((a, t, sc, sh, mean, invstd))
                a = nxt
            else:
                saved.append((a, t, sc, sh, mean, invstd))
        short = x
        sc_ctx = None
        if forked is not None:
            (short, sc_ctx), joined = forked
            joined()
        elif len(self.shortcut):
            short, sc_ctx = self._shortcut_fwd(x)
        _, t, sc, sh, _, _ = saved[-1]
        y = K.affine_act(t, sc, sh, r1=short, a1=1.0, act="relu" if self.final_relu else None, out=out)
        self._ctx = (x, saved, sc_ctx, y) if self.training else None
        return y

    def bwd(self, dy, dx_out=None, accumulate=False, addend=None, need_dx=True):
        (x, saved, sc_ctx, y), self._ctx = self._ctx, None
        branch = self._branch()
        # the final ReLU's mask and the reduce of the last BatchNorm's backward in one sweep (sgx_relu_bwd_bn_reduce, round 6: 16 reduce sweeps
        # of a ResNet-50 step gone); SGX_FUSE_BN_REDUCE=0 or a synchronised BatchNorm: the two passes
        parts_last = None
        if self.final_relu and self._net.fuse_bn_reduce and not branch[-1][1]._synced() and _RELU_REDUCE:
            g, parts_last = K.relu_bwd_bn_reduce(dy, y, saved[-1][1], saved[-1][4])
        else:
            g = K.relu_bwd(dy, y) if self.final_relu else dy
        # the projection shortcut's backward needs g only: forked at the start onto the branch stream (when its forward ran there: its saved
        # tensors are that stream's pool's), its data gradient into a tensor of its own that the main branch's last data gradient adds in its
        # epilogue - the same two-term sum as the accumulate pass
        net = getattr(self, "_net", None)
        fork_sc = (sc_ctx is not None and getattr(self, "_branched", False) and net is not None and net.branches(16, 0, 0, True)
                   and need_dx and addend is None and not accumulate)
        if fork_sc:
            def shortcut_bwd():
                ts, scs, shs, ms, ivs = sc_ctx
                cs, bs = getattr(self.shortcut, "0"), getattr(self.shortcut, "1")
                dts = bs.backward(g, ts, scs, shs, ms, ivs, None, dx_out=ts)
                cs.wgrad(x, dts)
                return cs.dgrad(dts, tuple(x.shape))

            dxs, joined = net.fork_branch(shortcut_bwd, backward=True)
        d = g
        parts = parts_last
        for i in range(len(branch) - 1, 0, -1):
            conv, bn = branch[i]
            a, t, sc, sh, mean, invstd = saved[i]
            dt = bn.backward(d, t, sc, sh, mean, invstd, None if i == len(branch) - 1 else "relu", dx_out=t, parts=parts)
            conv.wgrad(a, dt)
            # (round 6) the BatchNorm-backward reduce of the layer below rides in this data gradient's epilogue (sgx_bn_reduce_req: the launch
            # that writes that layer's output gradient also leaves its two per-channel sums) - as the YOLO-NAS blocks have done since round 4;
            # the ResNet blocks still ran a reduce sweep over (gradient, saved conv output) per layer: 32 of a ResNet-50 step's 53
            _, pt, psc, psh, pmean, _ = saved[i - 1]
            pbn = branch[i - 1][1]
            req = K.BnReduceRequest(pt, psc, psh, pmean, "relu") if (self._net.fuse_bn_reduce and not pbn._synced()) else None
            d = conv.dgrad(dt, tuple(a.shape), reqs=[req] if req is not None else None)
            parts = req.parts if req is not None else None
        conv, bn = branch[0]
        a, t, sc, sh, mean, invstd = saved[0]
        dt = bn.backward(d, t, sc, sh, mean, invstd, None if len(branch) == 1 else "relu", dx_out=t, parts=parts)
        conv.wgrad(x, dt)
        if fork_sc:
            joined()
            return conv.dgrad(dt, tuple(x.shape), out=dx_out, addend=dxs)
        if sc_ctx is not None:
            ts, scs, shs, ms, ivs = sc_ctx
            cs, bs = getattr(self.shortcut, "0"), getattr(self.shortcut, "1")
            dts = bs.backward(g, ts, scs, shs, ms, ivs, None, dx_out=ts)
            cs.wgrad(x, dts)
        if not need_dx:
            return None
        shape = tuple(x.shape)
        if sc_ctx is not None:
            dx = conv.dgrad(dt, shape, out=dx_out, accumulate=accumulate, addend=addend)
            return getattr(self.shortcut, "0").dgrad(dts, shape, out=dx, accumulate=True)
        dx = conv.dgrad(dt, shape, out=dx_out, accumulate=accumulate, addend=g)  # identity shortcut: + g in the epilogue
        if addend is not None:
            K.axpy(addend, out=dx, accumulate=True)
        return dx


class BasicResNetBlock(_ResBlock):
    def __init__(self, in_planes, planes, stride=1, expansion=1, final_relu=True, droppath_prob=0.0):
        super().__init__()
        if droppath_prob:
            raise NotImplementedError("DropPath (droppath_prob > 0) is not on the HIP path")
        self.expansion, self.final_relu = expansion, final_relu
        self.conv1 = ConvLayer(in_planes, planes, 3, stride, 1, bias=False)
        self.bn1 = BatchNorm(planes)
        self.conv2 = ConvLayer(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = BatchNorm(planes)
        self._init_shortcut(in_planes, expansion * planes, stride)

    def _branch(self):
        return [(self.conv1, self.bn1), (self.conv2, self.bn2)]


class Bottleneck(_ResBlock):
    def __init__(self, in_planes, planes, stride=1, expansion=4, final_relu=True, droppath_prob=0.0):
        super().__init__()
        if droppath_prob:
            raise NotImplementedError("DropPath (droppath_prob > 0) is not on the HIP path")
        self.expansion, self.final_relu = expansion, final_relu
        self.conv1 = ConvLayer(in_planes, planes, 1, 1, 0, bias=False)
        self.bn1 = BatchNorm(planes)
        self.conv2 = ConvLayer(planes, planes, 3, stride, 1, bias=False)
        self.bn2 = BatchNorm(planes)
        self.conv3 = ConvLayer(planes, expansion * planes, 1, 1, 0, bias=False)
        self.bn3 = BatchNorm(expansion * planes)
        self._init_shortcut(in_planes, expansion * planes, stride)

    def _branch(self):
        return [(self.conv1, self.bn1), (self.conv2, self.bn2), (self.conv3, self.bn3)]


class _Layer(nn.Module):
    """nn.Sequential of blocks (keys layer{i}.{j}.*)."""

    def __init__(self, blocks):
        super().__init__()
        for i, b in enumerate(blocks):
            self.add_module(str(i), b)

    def blocks(self):
        return list(self._modules.values())


class _ResNetBase(SgxNetwork):
    """Shared driver: stem -> layer1..4 -> global average pool -> linear, and its backward."""

    def _make_layer(self, block, planes, num_blocks, stride):
        if num_blocks == 0:
            raise NotImplementedError("ResNet layers with num_blocks == 0 (conv-only layers of the customised variants) are not on the HIP path")
        blocks = []
        for s in [stride] + [1] * (num_blocks - 1):
            blocks.append(block(self.in_planes, planes, s, self.expansion) if block is Bottleneck else block(self.in_planes, planes, s, self.expansion))
            self.in_planes = planes * self.expansion
        return _Layer(blocks)

    def _stem_fwd(self, xh):
        raise NotImplementedError

    def _stem_bwd(self, d):
        raise NotImplementedError

    def _fwd(self, x):
        if x.dim() != 4 or x.shape[1] != self.conv1.in_channels:
            raise ValueError(f"expected an NCHW batch with {self.conv1.in_channels} channels, got {tuple(x.shape)}")
        a = self._stem_fwd(K.input_to_nhwc(x))
        for layer in (self.layer1, self.layer2, self.layer3, self.layer4):
            for blk in layer.blocks():
                a = blk.fwd(a)
        self._feat_shape = tuple(a.shape)
        pooled = K.avgpool_fwd(a)
        logits = self.linear.fwd(pooled)
        return (logits.contiguous(),)

    def _bwd(self, d_logits):
        d = self.linear.bwd(d_logits.contiguous())
        d = K.avgpool_bwd(d.contiguous(), self._feat_shape)
        ready = self._bucket_ready
        ready("linear.")
        for name in ("layer4", "layer3", "layer2", "layer1"):
            for blk in reversed(getattr(self, name).blocks()):
                d = blk.bwd(d)
            ready(f"{name}.")
        self._stem_bwd(d)
        ready("bn1.")
        ready("conv1.")

    def gradient_buckets(self):
        """Arena ranges in parameter order; `stem` covers conv1/bn1 (GradientAllReducer matches by name prefix)."""
        return ["conv1.", "bn1.", "layer1.", "layer2.", "layer3.", "layer4.", "linear."]

    # ---- SgModule-style helpers the reference exposes -------------------------------------------------------------
    def get_input_channels(self) -> int:
        return self.conv1.in_channels

    def replace_input_channels(self, in_channels: int, compute_new_weights_fn=None):
        """Reference resnet.py:130-136, 249-258 with modules/weight_replacement_utils.py:24-65: conv1 keeps the weights of the channels it
        already has; extra channels are drawn from a normal distribution with the old weights' mean / std.  Before materialisation only."""
        if self._materialized:
            raise RuntimeError("replace_input_channels must be called before the model is materialized in HBM (before the first forward)")
        old = self.conv1
        if compute_new_weights_fn is not None:
            self.conv1 = compute_new_weights_fn(old, in_channels)
            return
        new = ConvLayer(in_channels, old.out_channels, old.kernel_size, old.stride, old.padding, bias=old.bias is not None)
        w = old.weight.data
        if in_channels <= old.in_channels:
            new.weight.data = w[:, :in_channels].clone()
        else:
            new.weight.data[:, : old.in_channels] = w
            torch.nn.init.normal_(new.weight.data[:, old.in_channels:], mean=float(w.mean()), std=float(w.std()))
        self.conv1 = new

    def get_finetune_lr_dict(self, lr: float) -> Dict[str, float]:
        return {"linear": lr, "default": 0}


class CifarResNet(_ResNetBase):
    def __init__(self, block, num_blocks, num_classes=10, width_mult=1, expansion=1, in_channels: int = 3):
        super().__init__()
        self.expansion = expansion
        self.structure = [num_blocks, width_mult]
        self.in_planes = width_multiplier(64, width_mult)
        self.conv1 = ConvLayer(in_channels, width_multiplier(64, width_mult), 3, 1, 1, bias=False)
        self.bn1 = BatchNorm(width_multiplier(64, width_mult))
        self.layer1 = self._make_layer(block, width_multiplier(64, width_mult), num_blocks[0], 1)
        self.layer2 = self._make_layer(block, width_multiplier(128, width_mult), num_blocks[1], 2)
        self.layer3 = self._make_layer(block, width_multiplier(256, width_mult), num_blocks[2], 2)
        self.layer4 = self._make_layer(block, width_multiplier(512, width_mult), num_blocks[3], 2)
        self.linear = LinearLayer(width_multiplier(512, width_mult) * expansion, num_classes)

    def _stem_fwd(self, xh):
        t, sc, sh, mean, invstd = _ConvBN.fwd(self.conv1, self.bn1, xh, self.training)
        a = K.affine_act(t, sc, sh, act="relu", out=None if self.training else t)
        self._stem = (xh, t, sc, sh, mean, invstd) if self.training else None
        return a

    def _stem_bwd(self, d):
        (xh, t, sc, sh, mean, invstd), self._stem = self._stem, None
        dt = self.bn1.backward(d, t, sc, sh, mean, invstd, "relu", dx_out=t)
        self.conv1.wgrad(xh, dt)


class ResNet(_ResNetBase):
    def __init__(self, block, num_blocks: list, num_classes: int = 10, width_mult: float = 1, expansion: int = 1, droppath_prob=0.0,
                 input_batchnorm: bool = False, backbone_mode: bool = False, in_channels: int = 3):
        super().__init__()
        if droppath_prob or input_batchnorm or backbone_mode:
            raise NotImplementedError("ResNet on the HIP path: droppath_prob=0, input_batchnorm=False, backbone_mode=False")
        self.expansion, self.width_mult = expansion, width_mult
        self.structure = [num_blocks, width_mult]
        self.in_planes = width_multiplier(64, width_mult)
        self.conv1 = ConvLayer(in_channels, width_multiplier(64, width_mult), 7, 2, 3, bias=False)
        self.bn1 = BatchNorm(width_multiplier(64, width_mult))
        self.maxpool = MaxPool(3, 2, 1)
        self.layer1 = self._make_layer(block, width_multiplier(64, width_mult), num_blocks[0], 1)
        self.layer2 = self._make_layer(block, width_multiplier(128, width_mult), num_blocks[1], 2)
        self.layer3 = self._make_layer(block, width_multiplier(256, width_mult), num_blocks[2], 2)
        self.layer4 = self._make_layer(block, width_multiplier(512, width_mult), num_blocks[3], 2)
        self.linear = LinearLayer(width_multiplier(512, width_mult) * expansion, num_classes)

    def _stem_fwd(self, xh):
        t, sc, sh, mean, invstd = _ConvBN.fwd(self.conv1, self.bn1, xh, self.training)
        a = K.affine_act(t, sc, sh, act="relu", out=None if self.training else t)
        self._stem = (xh, t, sc, sh, mean, invstd) if self.training else None
        return self.maxpool.fwd(a)

    def _stem_bwd(self, d):
        (xh, t, sc, sh, mean, invstd), self._stem = self._stem, None
        d = self.maxpool.bwd(d)
        dt = self.bn1.backward(d, t, sc, sh, mean, invstd, "relu", dx_out=t)
        self.conv1.wgrad(xh, dt)

    def replace_head(self, new_num_classes=None, new_head=None):
        if new_num_classes is None and new_head is None:
            raise ValueError("At least one of new_num_classes, new_head must be given to replace output layer.")
        if new_head is not None:
            raise NotImplementedError("replace_head(new_head=...) is not on the HIP path; pass new_num_classes")
        if self._materialized:
            raise RuntimeError("replace_head must be called before the model is materialized in HBM")
        self.linear = LinearLayer(width_multiplier(512, self.width_mult) * self.expansion, new_num_classes)


def _nc(arch_params, num_classes):
    return num_classes or get_param(arch_params, "num_classes")


def _resnet(name, block, layers, expansion=1):
    def init(self, arch_params, num_classes=None):
        ResNet.__init__(self, block, layers, num_classes=_nc(arch_params, num_classes), expansion=expansion,
                        droppath_prob=get_param(arch_params, "droppath_prob", 0), backbone_mode=get_param(arch_params, "backbone_mode", False))

    return register_model(name)(type(name.title().replace("_", ""), (ResNet,), {"__init__": init}))


ResNet18 = _resnet("resnet18", BasicResNetBlock, [2, 2, 2, 2])
ResNet34 = _resnet("resnet34", BasicResNetBlock, [3, 4, 6, 3])
ResNet50 = _resnet("resnet50", Bottleneck, [3, 4, 6, 3], expansion=4)
ResNet101 = _resnet("resnet101", Bottleneck, [3, 4, 23, 3], expansion=4)
ResNet152 = _resnet("resnet152", Bottleneck, [3, 8, 36, 3], expansion=4)


@register_model("resnet18_cifar")
class ResNet18Cifar(CifarResNet):
    def __init__(self, arch_params, num_classes=None):
        super().__init__(BasicResNetBlock, [2, 2, 2, 2], num_classes=_nc(arch_params, num_classes))


@register_model("resnet50_3343")
class ResNet50_3343(ResNet):
    def __init__(self, arch_params, num_classes=None):
        super().__init__(Bottleneck, [3, 3, 4, 3], num_classes=_nc(arch_params, num_classes), expansion=4,
                         droppath_prob=get_param(arch_params, "droppath_prob", 0), backbone_mode=get_param(arch_params, "backbone_mode", False))

"""CSPResNet backbone of PP-YOLOE on the HIP kernels.

Reference (training/models/detection_models/csp_resnet.py): CSPResNetBasicBlock :21-49, CSPResStage :52-113,
CSPResNetBackbone :116-254.  Same constructor arguments, same child names -> same state_dict keys
(stem.conv{1,2,3}.seq.*, stages.{i}.{conv_down,conv1,conv2,conv3}.seq.*, stages.{i}.blocks.{j}.{conv1.seq.*,conv2.branch_*},
stages.{i}.attn.project.*).

MI355X structure: no torch.cat - conv1 and the last basic block write straight into the two channel halves of one NHWC buffer
that the squeeze-excitation gate and conv3 then read; the `x + y` of a basic block is folded into the RepVGG activation sweep.
"""
from typing import Tuple

import torch
from torch import nn

from .... import kernels as K
from ....common.registry import register_detection_module
from ....modules.base_modules import BaseDetectionModule
from ....modules.conv_bn_act_block import ConvBNAct
from ....modules.engine import SgxBlock
from ....modules.layers import act_name
from ....modules.repvgg_block import RepVGGBlock
from ....modules.se_blocks import EffectiveSEBlock

__all__ = ["CSPResNetBackbone", "CSPResNetBasicBlock", "CSPResStage"]


class CSPResNetBasicBlock(SgxBlock):
    def __init__(self, in_channels: int, out_channels: int, activation_type, use_residual_connection: bool = True, use_alpha=False):
        super().__init__()
        if use_residual_connection and in_channels != out_channels:
            raise RuntimeError(f"Number of input channels (got {in_channels}) must be equal to the number of output channels (got {out_channels}) "
                               f"when use_residual_connection=True")
        self.conv1 = ConvBNAct(in_channels, out_channels, kernel_size=3, stride=1, padding=1, activation_type=activation_type, bias=False)
        self.conv2 = RepVGGBlock(out_channels, out_channels, activation_type=activation_type, se_type=nn.Identity, use_residual_connection=False,
                                 use_alpha=use_alpha)
        self.use_residual_connection = use_residual_connection

    def on_materialize(self):
        pass

    def fwd(self, x, out=None):
        return self.conv2.fwd(self.conv1.fwd(x), out=out, post_add=x if self.use_residual_connection else None)

    def bwd(self, dy, dx_out=None, accumulate=False, addend=None, need_dx=True):
        g = self.conv2.bwd(dy)
        if not self.use_residual_connection:
            return self.conv1.bwd(g, dx_out=dx_out, accumulate=accumulate, addend=addend, need_dx=need_dx)
        if addend is not None:
            raise NotImplementedError("CSPResNetBasicBlock.bwd: residual form takes no extra addend")
        if not need_dx:
            return self.conv1.bwd(g, need_dx=False)
        # + dy (the skip connection): folded into the data-gradient epilogue when dy and dx share their strides
        like = dx_out if dx_out is not None else None
        same = (like is None and dy.is_contiguous()) or (like is not None and K.nhwc_strides(like) == K.nhwc_strides(dy))
        if same:
            return self.conv1.bwd(g, dx_out=dx_out, accumulate=accumulate, addend=dy)
        dx = self.conv1.bwd(g, dx_out=dx_out, accumulate=accumulate)
        return K.axpy(dy, out=dx, accumulate=True)


class _BlockList(nn.Module):
    """Children under integer names (state_dict keys blocks.{i}.*) like the reference's nn.Sequential."""

    def __init__(self, mods):
        super().__init__()
        for i, m in enumerate(mods):
            self.add_module(str(i), m)

    def __iter__(self):
        return iter(self._modules.values())

    def __len__(self):
        return len(self._modules)


class CSPResStage(SgxBlock):
    def __init__(self, in_channels: int, out_channels: int, num_blocks, stride: int, activation_type, use_attention: bool = True,
                 use_alpha: bool = False):
        super().__init__()
        mid_channels = (in_channels + out_channels) // 2
        half_mid_channels = mid_channels // 2
        mid_channels = 2 * half_mid_channels
        if half_mid_channels % 4:
            raise NotImplementedError(f"CSPResStage on the HIP path: half of the mid channels ({half_mid_channels}) must be a multiple of 4 "
                                      "(16-byte channel groups); all PP-YOLOE S/M/L/X widths are")
        self.half = half_mid_channels
        self.conv_down = ConvBNAct(in_channels, mid_channels, 3, stride=stride, padding=1, activation_type=activation_type, bias=False) \
            if stride != 1 else None
        self.conv1 = ConvBNAct(mid_channels, half_mid_channels, kernel_size=1, stride=1, padding=0, activation_type=activation_type, bias=False)
        self.conv2 = ConvBNAct(mid_channels, half_mid_channels, kernel_size=1, stride=1, padding=0, activation_type=activation_type, bias=False)
        self.blocks = _BlockList([CSPResNetBasicBlock(half_mid_channels, half_mid_channels, activation_type=activation_type, use_alpha=use_alpha)
                                  for _ in range(num_blocks)])
        self.attn = EffectiveSEBlock(mid_channels) if use_attention else nn.Identity()
        self.conv3 = ConvBNAct(mid_channels, out_channels, kernel_size=1, stride=1, padding=0, activation_type=activation_type, bias=False)

    def on_materialize(self):
        pass

    def fwd(self, x, out=None):
        if self.conv_down is not None:
            x = self.conv_down.fwd(x)
        n, h, w, _ = x.shape
        half = self.half
        cat = torch.empty(n, h, w, 2 * half, device=x.device, dtype=x.dtype)  # (bf16 on the half-precision inference path)
        self.conv1.fwd(x, out=cat[..., :half])
        blocks = list(self.blocks)
        cur = self.conv2.fwd(x, out=cat[..., half:] if not blocks else None)
        for i, b in enumerate(blocks):
            cur = b.fwd(cur, out=cat[..., half:] if i == len(blocks) - 1 else None)
        y = self.attn.fwd(cat) if isinstance(self.attn, EffectiveSEBlock) else cat
        return self.conv3.fwd(y, out=out)

    def bwd(self, dy, dx_out=None, accumulate=False, addend=None, need_dx=True):
        half = self.half
        dcat = self.conv3.bwd(dy)
        if isinstance(self.attn, EffectiveSEBlock):
            dcat = self.attn.bwd(dcat)
        g = dcat[..., half:]
        for b in reversed(list(self.blocks)):
            g = b.bwd(g)
        if self.conv_down is None:
            dx = self.conv2.bwd(g, dx_out=dx_out, accumulate=accumulate, addend=addend, need_dx=need_dx)
            return self.conv1.bwd(dcat[..., :half], dx_out=dx, accumulate=True, need_dx=need_dx)
        dx = self.conv2.bwd(g)
        dx = self.conv1.bwd(dcat[..., :half], dx_out=dx, accumulate=True)
        return self.conv_down.bwd(dx, dx_out=dx_out, accumulate=accumulate, addend=addend, need_dx=need_dx)


class _Stem(nn.Module):
    """conv1 / conv2 [/ conv3] under their reference names (an OrderedDict nn.Sequential there, csp_resnet.py:150-191)."""

    def __iter__(self):
        return iter(self._modules.values())


@register_detection_module()
class CSPResNetBackbone(BaseDetectionModule):
    def __init__(self, layers: Tuple[int, ...], channels: Tuple[int, ...], activation, return_idx: Tuple[int, int, int], use_large_stem: bool,
                 width_mult: float, depth_mult: float, use_alpha: bool, pretrained_weights=None, in_channels: int = 3):
        super().__init__(in_channels)
        if pretrained_weights:
            raise NotImplementedError("pretrained_weights: checkpoint download / loading is outside the MI355X hot path (no network); "
                                      "load a state_dict explicitly")
        act = act_name(activation)
        channels = [max(round(num_channels * width_mult), 1) for num_channels in channels]
        layers = [max(round(num_layers * depth_mult), 1) for num_layers in layers]
        self.stem = _Stem()
        if use_large_stem:
            self.stem.add_module("conv1", ConvBNAct(in_channels, channels[0] // 2, 3, stride=2, padding=1, activation_type=act, bias=False))
            self.stem.add_module("conv2", ConvBNAct(channels[0] // 2, channels[0] // 2, 3, stride=1, padding=1, activation_type=act, bias=False))
            self.stem.add_module("conv3", ConvBNAct(channels[0] // 2, channels[0], 3, stride=1, padding=1, activation_type=act, bias=False))
        else:
            self.stem.add_module("conv1", ConvBNAct(3, channels[0] // 2, 3, stride=2, padding=1, activation_type=act, bias=False))
            self.stem.add_module("conv2", ConvBNAct(channels[0] // 2, channels[0], 3, stride=1, padding=1, activation_type=act, bias=False))
        n = len(channels) - 1
        self.stages = nn.ModuleList([CSPResStage(channels[i], channels[i + 1], layers[i], stride=2, activation_type=act, use_alpha=use_alpha)
                                     for i in range(n)])
        self._out_channels = channels[1:]
        self._out_strides = [4 * 2 ** i for i in range(n)]
        self.return_idx = tuple(return_idx)

    @property
    def out_channels(self):
        return tuple(self._out_channels)

    def get_input_channels(self) -> int:
        return next(iter(self.stem)).seq.conv.in_channels

    def fwd(self, x, out=None):
        for m in self.stem:
            x = m.fwd(x)
        outs = []
        for idx, stage in enumerate(self.stages):
            x = stage.fwd(x)
            if idx in self.return_idx:
                outs.append(x)
        return outs

    def bwd(self, grads, on_layer_done=None):
        """grads: gradients of the returned feature maps (same order as fwd's list); walks the stages backwards, adding each external
        gradient where its tensor was produced."""
        ext = dict(zip([i for i in range(len(self.stages)) if i in self.return_idx], grads))
        g = None
        for idx in range(len(self.stages) - 1, -1, -1):
            e = ext.get(idx)
            if g is None:
                g = e
            elif e is not None:
                K.axpy(e, out=g, accumulate=True)
            if g is None:
                continue
            g = self.stages[idx].bwd(g)
            if on_layer_done is not None:
                on_layer_done(f"stages.{idx}")
        stem = list(self.stem)
        for i in range(len(stem) - 1, -1, -1):
            g = stem[i].bwd(g, need_dx=i != 0)
        if on_layer_done is not None:
            on_layer_done("stem")
        return g

    def prep_model_for_conversion(self, input_size=None, **kwargs):
        for module in self.modules():
            if isinstance(module, RepVGGBlock):
                module.fuse_block_residual_branches()

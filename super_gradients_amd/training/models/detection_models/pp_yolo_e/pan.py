"""PPYoloECSPPAN neck on the HIP kernels.

Reference (training/models/detection_models/pp_yolo_e/pan.py): PPYoloESPP :14-39, CSPStage :42-70, PPYoloECSPPAN :73-195 - a
top-down FPN (CSP stage -> 1x1 route -> nearest x2 up-sampling -> concat with the next backbone level) followed by a bottom-up
PAN (3x3 stride-2 route -> concat with the FPN feature -> CSP stage); returns the maps coarse-to-fine (strides 32, 16, 8).
Same constructor arguments and child names -> same state_dict keys (fpn_stages.{i}.{j}.*, fpn_routes.{i}.seq.*, pan_stages, pan_routes).

MI355X structure: every torch.cat of the reference is one preallocated NHWC buffer whose producers (up-sampling kernel, route convs,
CSP-stage conv3, SPP pooling kernels) write their channel slice in place; in backward the consumers read their slice of the concat
gradient in place and tensors with several consumers get their gradients accumulated inside the data-gradient epilogues.
"""
from typing import List, Tuple

import torch
from torch import nn

from ..... import kernels as K
from .....common.registry import register_detection_module
from .....modules.base_modules import BaseDetectionModule
from .....modules.conv_bn_act_block import ConvBNAct
from .....modules.engine import SgxBlock
from .....modules.layers import MaxPool, act_name
from ..csp_resnet import CSPResNetBasicBlock

__all__ = ["PPYoloECSPPAN"]


class PPYoloESPP(SgxBlock):
    """cat([x, maxpool5(x), maxpool9(x), maxpool13(x)]) -> ConvBNAct.  fwd takes the producer of x as a callable so that x is written
    straight into channel slice 0 of the concat buffer."""

    def __init__(self, in_channels: int, out_channels: int, kernel_size: int, pool_size: Tuple[int, ...], activation_type):
        super().__init__()
        self.cin = in_channels
        mid_channels = in_channels * (1 + len(pool_size))
        self.pool = nn.ModuleList([MaxPool(kernel_size=size, stride=1, padding=size // 2) for size in pool_size])
        self.conv = ConvBNAct(mid_channels, out_channels, kernel_size, padding=kernel_size // 2, activation_type=activation_type, stride=1, bias=False)

    def on_materialize(self):
        pass

    def alloc(self, n, h, w, device, dtype=torch.float32):
        return torch.empty(n, h, w, self.cin * (1 + len(self.pool)), device=device, dtype=dtype)

    def fwd(self, cat, out=None):
        """cat: buffer from alloc() whose slice 0 already holds x."""
        c = self.cin
        x = cat[..., :c]
        for i, pool in enumerate(self.pool):
            pool.fwd(x, out=cat[..., (i + 1) * c:(i + 2) * c])
        return self.conv.fwd(cat, out=out)

    def bwd(self, dy):
        """-> gradient of x (slice 0 of the concat gradient, the pooling backward passes accumulated into it)."""
        c = self.cin
        dcat = self.conv.bwd(dy)
        g = dcat[..., :c]
        for i, pool in enumerate(self.pool):
            pool.bwd(dcat[..., (i + 1) * c:(i + 2) * c], dx_out=g, accumulate=True)
        return g


class _NamedSeq(nn.Module):
    def __init__(self, named):
        super().__init__()
        for name, m in named:
            self.add_module(name, m)

    def __iter__(self):
        return iter(self._modules.values())

    def __len__(self):
        return len(self._modules)


class CSPStage(SgxBlock):
    def __init__(self, in_channels: int, out_channels: int, n, activation_type, spp: bool):
        super().__init__()
        ch_mid = int(out_channels // 2)
        if ch_mid % 4:
            raise NotImplementedError(f"CSPStage on the HIP path: out_channels/2 ({ch_mid}) must be a multiple of 4")
        self.ch_mid = ch_mid
        self.conv1 = ConvBNAct(in_channels, ch_mid, kernel_size=1, padding=0, activation_type=activation_type, stride=1, bias=False)
        self.conv2 = ConvBNAct(in_channels, ch_mid, kernel_size=1, padding=0, activation_type=activation_type, stride=1, bias=False)
        convs = []
        next_ch_in = ch_mid
        for i in range(n):
            convs.append((str(i), CSPResNetBasicBlock(next_ch_in, ch_mid, activation_type=activation_type, use_residual_connection=False)))
            if i == (n - 1) // 2 and spp:
                convs.append(("spp", PPYoloESPP(ch_mid, ch_mid, 1, (5, 9, 13), activation_type=activation_type)))
            next_ch_in = ch_mid
        self.convs = _NamedSeq(convs)
        self.conv3 = ConvBNAct(ch_mid * 2, out_channels, kernel_size=1, padding=0, activation_type=activation_type, stride=1, bias=False)

    def on_materialize(self):
        pass

    def fwd(self, x, out=None):
        n, h, w, _ = x.shape
        mid = self.ch_mid
        cat = torch.empty(n, h, w, 2 * mid, device=x.device, dtype=x.dtype)
        self.conv1.fwd(x, out=cat[..., :mid])
        seq = list(self.convs)
        cur = self.conv2.fwd(x, out=cat[..., mid:] if not seq else None)
        for i, m in enumerate(seq):
            dst = cat[..., mid:] if i == len(seq) - 1 else None
            if isinstance(m, PPYoloESPP):
                cur = m.fwd(cur, out=dst)   # cur is the SPP concat buffer (its slice 0 was written by the previous block)
            elif i + 1 < len(seq) and isinstance(seq[i + 1], PPYoloESPP):
                buf = seq[i + 1].alloc(n, h, w, x.device, x.dtype)
                m.fwd(cur, out=buf[..., :mid])
                cur = buf
            else:
                cur = m.fwd(cur, out=dst)
        return self.conv3.fwd(cat, out=out)

    def bwd(self, dy, dx_out=None, accumulate=False, addend=None, need_dx=True):
        mid = self.ch_mid
        dcat = self.conv3.bwd(dy)
        g = dcat[..., mid:]
        for m in reversed(list(self.convs)):
            g = m.bwd(g)
        dx = self.conv2.bwd(g, dx_out=dx_out, accumulate=accumulate, addend=addend, need_dx=need_dx)
        return self.conv1.bwd(dcat[..., :mid], dx_out=dx, accumulate=True, need_dx=need_dx)


class _StageSeq(_NamedSeq):
    """fpn_stages[i] / pan_stages[i]: `stage_num` CSP stages in sequence (child names "0", "1", ...)."""

    def fwd(self, x, out=None):
        mods = list(self)
        for j, m in enumerate(mods):
            x = m.fwd(x, out=out if j == len(mods) - 1 else None)
        return x

    def bwd(self, dy, **kw):
        mods = list(self)
        for j in range(len(mods) - 1, 0, -1):
            dy = mods[j].bwd(dy)
        return mods[0].bwd(dy, **kw)


@register_detection_module()
class PPYoloECSPPAN(BaseDetectionModule):
    def __init__(self, in_channels: Tuple[int, ...], out_channels: Tuple[int, ...], activation, stage_num: int, block_num: int, spp: bool,
                 width_mult: float, depth_mult: float):
        super().__init__(in_channels)
        act = act_name(activation)
        in_channels = [max(round(c * width_mult), 1) for c in in_channels]
        out_channels = [max(round(c * width_mult), 1) for c in out_channels]
        if len(in_channels) != len(out_channels):
            raise ValueError("in_channels and out_channels must have the same length")
        block_num = max(round(block_num * depth_mult), 1)
        self.num_blocks = len(in_channels)
        self._out_channels = out_channels
        in_channels = in_channels[::-1]
        self._cin = list(in_channels)
        fpn_stages, fpn_routes = [], []
        ch_pre = None
        for i, (ch_in, ch_out) in enumerate(zip(in_channels, out_channels)):
            if i > 0:
                ch_in += ch_pre // 2
            fpn_stages.append(_StageSeq([(str(j), CSPStage(ch_in if j == 0 else ch_out, ch_out, block_num, activation_type=act, spp=(spp and i == 0)))
                                         for j in range(stage_num)]))
            if i < self.num_blocks - 1:
                fpn_routes.append(ConvBNAct(in_channels=ch_out, out_channels=ch_out // 2, kernel_size=1, stride=1, padding=0, activation_type=act,
                                            bias=False))
            ch_pre = ch_out
        self.fpn_stages = nn.ModuleList(fpn_stages)
        self.fpn_routes = nn.ModuleList(fpn_routes)
        pan_stages, pan_routes = [], []
        for i in reversed(range(self.num_blocks - 1)):
            pan_routes.append(ConvBNAct(in_channels=out_channels[i + 1], out_channels=out_channels[i + 1], kernel_size=3, stride=2, padding=1,
                                        activation_type=act, bias=False))
            ch_in = out_channels[i] + out_channels[i + 1]
            ch_out = out_channels[i]
            pan_stages.append(_StageSeq([(str(j), CSPStage(ch_in if j == 0 else ch_out, ch_out, block_num, activation_type=act, spp=False))
                                         for j in range(stage_num)]))
        self.pan_stages = nn.ModuleList(pan_stages[::-1])
        self.pan_routes = nn.ModuleList(pan_routes[::-1])

    @property
    def out_channels(self) -> Tuple[int, ...]:
        return tuple(self._out_channels)

    def fwd(self, blocks: List[torch.Tensor], out=None):
        blocks = list(blocks)[::-1]   # coarse -> fine
        nb, oc = self.num_blocks, self._out_channels
        n = blocks[0].shape[0]
        dev = blocks[0].device
        # PAN concat buffers [route(out[i+1]) | fpn_feat i (out[i])] at the resolution of level i: the FPN stage of level i writes its
        # output straight into its slice
        pan_cat = [torch.empty(n, blocks[i].shape[1], blocks[i].shape[2], oc[i + 1] + oc[i], device=dev, dtype=blocks[0].dtype) for i in range(nb - 1)]
        fpn_feats = []
        src = blocks[0]
        for i in range(nb):
            dst = pan_cat[i][..., oc[i + 1]:] if i < nb - 1 else None
            feat = self.fpn_stages[i].fwd(src, out=dst)
            fpn_feats.append(feat)
            if i < nb - 1:
                route = self.fpn_routes[i].fwd(feat)
                nxt = blocks[i + 1]
                r = route.shape[3]
                cat = torch.empty(n, nxt.shape[1], nxt.shape[2], r + nxt.shape[3], device=dev, dtype=nxt.dtype)
                K.upsample2x_fwd(route, out=cat[..., :r])
                K.axpy(nxt, out=cat[..., r:])
                src = cat
        self._route_ch = [fpn_feats[i].shape[3] // 2 for i in range(nb - 1)]
        pan_feats = [fpn_feats[-1]]
        route = fpn_feats[-1]
        for i in reversed(range(nb - 1)):
            self.pan_routes[i].fwd(route, out=pan_cat[i][..., :oc[i + 1]])
            route = self.pan_stages[i].fwd(pan_cat[i])
            pan_feats.append(route)
        return pan_feats[::-1]

    def bwd(self, *dfeats):
        """dfeats: gradients of the returned maps (coarse -> fine, owned buffers).  -> gradients of the backbone maps in the order the
        backbone returned them (fine -> coarse)."""
        nb, oc = self.num_blocks, self._out_channels
        dfeats = list(dfeats)
        # bottom-up PAN, walked top-down: level 0 (coarsest) first
        d_fpn = [None] * nb      # gradient of fpn_feats[i] from the PAN concat (a slice view)
        g_route_in = None        # gradient flowing into pan_feats of the next finer level from pan_routes
        for i in range(nb - 1):
            g = dfeats[i]
            if g_route_in is not None:
                K.axpy(g_route_in, out=g, accumulate=True)
            dcat = self.pan_stages[i].bwd(g)
            d_fpn[i] = dcat[..., oc[i + 1]:]
            g_route_in = self.pan_routes[i].bwd(dcat[..., :oc[i + 1]])
        # finest level: pan_feats[-1] IS fpn_feats[-1] (returned to the head and fed to pan_routes[nb-2])
        g_last = dfeats[nb - 1]
        if g_route_in is not None:
            K.axpy(g_route_in, out=g_last, accumulate=True)
        d_fpn[nb - 1] = g_last
        # top-down FPN, walked bottom-up: finest level first
        dblocks = [None] * nb
        g_up = None              # gradient of fpn_routes[i]'s output (through the up-sampling)
        for i in range(nb - 1, -1, -1):
            g = d_fpn[i]
            if i < nb - 1:
                g = self.fpn_routes[i].bwd(g_up, addend=None)
                K.axpy(d_fpn[i], out=g, accumulate=True)
            dsrc = self.fpn_stages[i].bwd(g)
            if i > 0:
                r = self._route_ch[i - 1]
                g_up = K.upsample2x_bwd(dsrc[..., :r])
                dblocks[i] = dsrc[..., r:]
            else:
                dblocks[i] = dsrc
        return dblocks[::-1]

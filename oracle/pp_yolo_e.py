"""TEST INFRASTRUCTURE ONLY (the parity oracle) - never imported by the product package.

CPU restatement, in plain PyTorch fp32 (NCHW, ATen kernels, autograd for the backward), of the reference's PP-YOLOE forward path.
Module and parameter names reproduce the reference's state_dict, so `load_state_dict(reference.state_dict())` works both ways;
that is how this file is pinned against the real reference (tests/test_oracle_vs_reference.py, oracle/make_golden.py ->
tests/golden/ppyoloe_s.pt).

Reference behaviour restated here (paths under /root/reference/src/super_gradients):
  ConvBNAct                         modules/conv_bn_act_block.py:9-69           keys seq.conv.weight, seq.bn.*
  RepVGGBlock.forward               modules/repvgg_block.py:98-107              act(bn(conv3x3) + alpha*bn(conv1x1) [+ bn(x)]), alpha = 1
  EffectiveSEBlock.forward          modules/se_blocks.py:39-42
  CSPResNetBasicBlock / CSPResStage / CSPResNetBackbone     training/models/detection_models/csp_resnet.py:21-225
  PPYoloESPP / CSPStage / PPYoloECSPPAN                     .../pp_yolo_e/pan.py:14-195
  ESEAttn / PPYOLOEHead (train + eval forward, init)        .../pp_yolo_e/pp_yolo_head.py:79-301
  generate_anchors_for_grid_cell                            .../pp_yolo_e/pp_yolo_head.py:21-76
  batch_distance2bbox                                       training/utils/bbox_utils.py:9-29
  arch tables                                               recipes/arch_params/ppyoloe_{,s,m,l,x}_arch_params.yaml
"""
import collections
import math

import torch
import torch.nn.functional as F
from torch import nn

MULTS = {"s": (0.33, 0.50), "m": (0.67, 0.75), "l": (1.0, 1.0), "x": (1.33, 1.25)}   # (depth_mult, width_mult)
ACTS = {"silu": nn.SiLU, "relu": nn.ReLU}


class ConvBNAct(nn.Module):
    def __init__(self, cin, cout, k, stride, pad, act):
        super().__init__()
        self.seq = nn.Sequential(collections.OrderedDict([("conv", nn.Conv2d(cin, cout, k, stride, pad, bias=False)), ("bn", nn.BatchNorm2d(cout))]))
        self.act = act()

    def forward(self, x):
        return self.act(self.seq(x))


def _conv_bn(cin, cout, k, stride, pad):
    return nn.Sequential(collections.OrderedDict([("conv", nn.Conv2d(cin, cout, k, stride, pad, bias=False)), ("bn", nn.BatchNorm2d(cout))]))


class RepVGGBlock(nn.Module):   # PP-YOLOE configuration: no identity branch, no SE; alpha = 1, or (PP-YOLOE+) a learnable [1] multiplier
    def __init__(self, cin, cout, act, use_alpha=False):
        super().__init__()
        self.branch_3x3 = _conv_bn(cin, cout, 3, 1, 1)
        self.branch_1x1 = _conv_bn(cin, cout, 1, 1, 0)
        self.nonlinearity = act()
        # modules/repvgg_block.py:77-87: alpha = 1 + N(0, 0.01^2) as a parameter when use_alpha, else the constant 1
        self.alpha = nn.Parameter(torch.tensor([1.0]) + torch.randn((1,)) * 0.01, requires_grad=True) if use_alpha else 1

    def forward(self, x):  # modules/repvgg_block.py:94-104
        return self.nonlinearity(self.branch_3x3(x) + self.alpha * self.branch_1x1(x))

    def fused(self):
        """(kernel, bias) of the equivalent single 3x3 conv (repvgg_block.py:109-165)."""
        def fuse(branch):
            std = (branch.bn.running_var + branch.bn.eps).sqrt()
            t = (branch.bn.weight / std).reshape(-1, 1, 1, 1)
            return branch.conv.weight * t, branch.bn.bias - branch.bn.running_mean * branch.bn.weight / std
        k3, b3 = fuse(self.branch_3x3)
        k1, b1 = fuse(self.branch_1x1)
        return k3 + self.alpha * F.pad(k1, [1, 1, 1, 1]), b3 + self.alpha * b1


class EffectiveSEBlock(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.project = nn.Conv2d(c, c, 1)

    def forward(self, x):
        return x * F.hardsigmoid(self.project(x.mean((2, 3), keepdim=True)))


class BasicBlock(nn.Module):
    def __init__(self, cin, cout, act, residual=True):
        super().__init__()
        self.conv1 = ConvBNAct(cin, cout, 3, 1, 1, act)
        self.conv2 = RepVGGBlock(cout, cout, act)
        self.residual = residual

    def forward(self, x):
        y = self.conv2(self.conv1(x))
        return x + y if self.residual else y


class CSPResStage(nn.Module):
    def __init__(self, cin, cout, n, act):
        super().__init__()
        mid = (cin + cout) // 2
        half = mid // 2
        mid = 2 * half
        self.conv_down = ConvBNAct(cin, mid, 3, 2, 1, act)
        self.conv1 = ConvBNAct(mid, half, 1, 1, 0, act)
        self.conv2 = ConvBNAct(mid, half, 1, 1, 0, act)
        self.blocks = nn.Sequential(*[BasicBlock(half, half, act) for _ in range(n)])
        self.attn = EffectiveSEBlock(mid)
        self.conv3 = ConvBNAct(mid, cout, 1, 1, 0, act)

    def forward(self, x):
        x = self.conv_down(x)
        return self.conv3(self.attn(torch.cat([self.conv1(x), self.blocks(self.conv2(x))], dim=1)))


class Backbone(nn.Module):
    def __init__(self, depth, width, act):
        super().__init__()
        ch = [max(round(c * width), 1) for c in (64, 128, 256, 512, 1024)]
        layers = [max(round(n * depth), 1) for n in (3, 6, 6, 3)]
        self.stem = nn.Sequential(collections.OrderedDict([("conv1", ConvBNAct(3, ch[0] // 2, 3, 2, 1, act)), ("conv2", ConvBNAct(ch[0] // 2, ch[0] // 2, 3, 1, 1, act)),
                                                           ("conv3", ConvBNAct(ch[0] // 2, ch[0], 3, 1, 1, act))]))
        self.stages = nn.ModuleList([CSPResStage(ch[i], ch[i + 1], layers[i], act) for i in range(4)])
        self.out_channels = ch[2:]

    def forward(self, x):
        x = self.stem(x)
        outs = []
        for i, s in enumerate(self.stages):
            x = s(x)
            if i in (1, 2, 3):
                outs.append(x)
        return outs


class SPP(nn.Module):
    def __init__(self, c, act):
        super().__init__()
        self.pool = nn.ModuleList([nn.MaxPool2d(k, 1, k // 2) for k in (5, 9, 13)])
        self.conv = ConvBNAct(4 * c, c, 1, 1, 0, act)

    def forward(self, x):
        return self.conv(torch.cat([x] + [p(x) for p in self.pool], dim=1))


class CSPStage(nn.Module):
    def __init__(self, cin, cout, n, act, spp):
        super().__init__()
        mid = cout // 2
        self.conv1 = ConvBNAct(cin, mid, 1, 1, 0, act)
        self.conv2 = ConvBNAct(cin, mid, 1, 1, 0, act)
        convs = []
        for i in range(n):
            convs.append((str(i), BasicBlock(mid, mid, act, residual=False)))
            if i == (n - 1) // 2 and spp:
                convs.append(("spp", SPP(mid, act)))
        self.convs = nn.Sequential(collections.OrderedDict(convs))
        self.conv3 = ConvBNAct(2 * mid, cout, 1, 1, 0, act)

    def forward(self, x):
        return self.conv3(torch.cat([self.conv1(x), self.convs(self.conv2(x))], dim=1))


class Neck(nn.Module):
    def __init__(self, depth, width, act):
        super().__init__()
        cin = [max(round(c * width), 1) for c in (256, 512, 1024)][::-1]
        cout = [max(round(c * width), 1) for c in (768, 384, 192)]
        n = max(round(3 * depth), 1)
        fpn_stages, fpn_routes, pre = [], [], None
        for i, (ci, co) in enumerate(zip(cin, cout)):
            if i > 0:
                ci += pre // 2
            fpn_stages.append(nn.Sequential(collections.OrderedDict([("0", CSPStage(ci, co, n, act, spp=(i == 0)))])))
            if i < 2:
                fpn_routes.append(ConvBNAct(co, co // 2, 1, 1, 0, act))
            pre = co
        self.fpn_stages, self.fpn_routes = nn.ModuleList(fpn_stages), nn.ModuleList(fpn_routes)
        pan_stages, pan_routes = [], []
        for i in (1, 0):
            pan_routes.append(ConvBNAct(cout[i + 1], cout[i + 1], 3, 2, 1, act))
            pan_stages.append(nn.Sequential(collections.OrderedDict([("0", CSPStage(cout[i] + cout[i + 1], cout[i], n, act, spp=False))])))
        self.pan_stages, self.pan_routes = nn.ModuleList(pan_stages[::-1]), nn.ModuleList(pan_routes[::-1])
        self.out_channels = cout

    def forward(self, blocks):
        blocks = blocks[::-1]
        fpn, route = [], None
        for i, b in enumerate(blocks):
            if i > 0:
                b = torch.cat([route, b], dim=1)
            route = self.fpn_stages[i](b)
            fpn.append(route)
            if i < 2:
                route = F.interpolate(self.fpn_routes[i](route), scale_factor=2, mode="nearest")
        pan, route = [fpn[-1]], fpn[-1]
        for i in (1, 0):
            route = self.pan_stages[i](torch.cat([self.pan_routes[i](route), fpn[i]], dim=1))
            pan.append(route)
        return pan[::-1]


class ESEAttn(nn.Module):
    def __init__(self, c, act):
        super().__init__()
        self.fc = nn.Conv2d(c, c, 1)
        self.conv = ConvBNAct(c, c, 1, 1, 0, act)
        nn.init.normal_(self.fc.weight, std=0.001)

    def forward(self, feat, avg):
        return self.conv(feat * torch.sigmoid(self.fc(avg)))


def anchors_for_grid_cell(sizes, strides, scale=5.0, offset=0.5):
    anchors, pts, counts, st = [], [], [], []
    for (h, w), s in zip(sizes, strides):
        half = scale * s * 0.5
        sx = (torch.arange(end=w) + offset) * s
        sy = (torch.arange(end=h) + offset) * s
        yy, xx = torch.meshgrid(sy, sx, indexing="ij")
        anchors.append(torch.stack([xx - half, yy - half, xx + half, yy + half], dim=-1).to(torch.float32).reshape(-1, 4))
        pts.append(torch.stack([xx, yy], dim=-1).to(torch.float32).reshape(-1, 2))
        counts.append(h * w)
        st.append(torch.full([h * w, 1], s, dtype=torch.float32))
    return torch.cat(anchors), torch.cat(pts), counts, torch.cat(st)


class Head(nn.Module):
    def __init__(self, num_classes, width, act, reg_max=16, strides=(32, 16, 8)):
        super().__init__()
        ch = [max(round(c * width), 1) for c in (768, 384, 192)]
        self.num_classes, self.reg_max, self.fpn_strides = num_classes, reg_max, strides
        self.stem_cls = nn.ModuleList([ESEAttn(c, act) for c in ch])
        self.stem_reg = nn.ModuleList([ESEAttn(c, act) for c in ch])
        self.pred_cls = nn.ModuleList([nn.Conv2d(c, num_classes, 3, padding=1) for c in ch])
        self.pred_reg = nn.ModuleList([nn.Conv2d(c, 4 * (reg_max + 1), 3, padding=1) for c in ch])
        for c_, r_ in zip(self.pred_cls, self.pred_reg):
            nn.init.constant_(c_.weight, 0.0)
            nn.init.constant_(c_.bias, -math.log((1 - 0.01) / 0.01))
            nn.init.constant_(r_.weight, 0.0)
            nn.init.constant_(r_.bias, 1.0)

    def forward(self, feats):
        cls, reg = [], []
        for i, f in enumerate(feats):
            avg = F.adaptive_avg_pool2d(f, (1, 1))
            cls.append(self.pred_cls[i](self.stem_cls[i](f, avg) + f).flatten(2).permute(0, 2, 1))
            reg.append(self.pred_reg[i](self.stem_reg[i](f, avg)).flatten(2).permute(0, 2, 1))
        logits, distri = torch.cat(cls, dim=1), torch.cat(reg, dim=1)
        sizes = [(f.shape[2], f.shape[3]) for f in feats]
        anchors, pts, counts, strides = anchors_for_grid_cell(sizes, self.fpn_strides)
        raw = (logits, distri, anchors.to(logits.dtype), pts.to(logits.dtype), counts, strides.to(logits.dtype))
        if self.training:
            return raw
        B, L, _ = distri.shape
        proj = torch.linspace(0, self.reg_max, self.reg_max + 1, dtype=distri.dtype)
        dist = (F.softmax(distri.reshape(B, L, 4, self.reg_max + 1), dim=-1) * proj).sum(-1)
        grid = torch.cat([torch.stack(torch.meshgrid(torch.arange(h, dtype=torch.float32) + 0.5, torch.arange(w, dtype=torch.float32) + 0.5, indexing="ij")[::-1],
                                      dim=-1).reshape(-1, 2) for h, w in sizes]).to(distri.dtype)
        boxes = torch.cat([grid - dist[..., :2], grid + dist[..., 2:]], dim=-1) * strides.to(distri.dtype)
        return (boxes, logits.sigmoid()), raw


class PPYoloE(nn.Module):
    def __init__(self, variant="s", num_classes=80, activation="silu"):
        super().__init__()
        depth, width = MULTS[variant]
        act = ACTS[activation]
        self.backbone = Backbone(depth, width, act)
        self.neck = Neck(depth, width, act)
        self.head = Head(num_classes, width, act)

    def forward(self, x):
        return self.head(self.neck(self.backbone(x)))

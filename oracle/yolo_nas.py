"""TEST INFRASTRUCTURE ONLY (the parity oracle) - never imported by the product package.

CPU restatement, in plain PyTorch fp32 (NCHW, ATen/oneDNN kernels), of the reference's YOLO-NAS
forward path.  Module and parameter names reproduce the reference's state_dict,
so `load_state_dict(reference.state_dict())` works both ways; that is how this file is pinned
against the real reference (tests/test_oracle_vs_reference.py, oracle/make_golden.py).

Reference behaviour restated here (paths under /root/reference/src/super_gradients):
  QARepVGGBlock.forward            modules/qarepvgg_block.py:184-204
  Conv (conv-bn-act)               modules/conv_bn_act_block.py:80-93
  ConvBNReLU / ConvBNAct           modules/conv_bn_relu_block.py:8-60, conv_bn_act_block.py:9-69
  YoloNASBottleneck / CSPLayer     training/models/detection_models/yolo_nas/yolo_stages.py:23-150
  Stem/Stage/UpStage/DownStage     .../yolo_nas/yolo_stages.py:153-395
  SPP                              training/models/detection_models/csp_darknet53.py:136-157
  NStageBackbone                   modules/detection_modules.py:34-102
  YoloNASPANNeckWithC2             .../yolo_nas/panneck.py:12-64
  YoloNASDFLHead / NDFLHeads       .../yolo_nas/dfl_heads.py:21-282
  anchors                          .../pp_yolo_e/pp_yolo_head.py:21-76
  bn eps/momentum override         .../customizable_detector.py:97-104
"""
import math
from typing import List

import torch
import torch.nn.functional as F
from torch import nn

# (stage hidden, stage blocks, stage concat), neck (blocks, hidden) x4, head width_mult, neck concat
ARCH = {
    "s": dict(stage_hidden=[32, 64, 96, 192], stage_blocks=[2, 3, 5, 2], stage_concat=[False] * 4,
              neck_blocks=[2, 2, 2, 2], neck_hidden=[64, 48, 64, 64], neck_concat=False, head_mult=0.5),
    "m": dict(stage_hidden=[64, 128, 256, 384], stage_blocks=[2, 3, 5, 2], stage_concat=[True, True, True, False],
              neck_blocks=[2, 3, 2, 3], neck_hidden=[192, 64, 192, 256], neck_concat=False, head_mult=0.75),
    "l": dict(stage_hidden=[96, 128, 256, 512], stage_blocks=[2, 3, 5, 2], stage_concat=[True] * 4,
              neck_blocks=[4, 4, 4, 4], neck_hidden=[128, 128, 128, 256], neck_concat=False, head_mult=1.0),
}
STAGE_OUT = [96, 192, 384, 768]
NECK_OUT = [192, 96, 192, 384]
HEAD_INTER = [128, 256, 512]
STRIDES = [8, 16, 32]


def _ceil_mult(v, mult, div):
    return math.ceil(int(v * mult) / div) * div


class ConvBnAct(nn.Module):  # reference `Conv`: keys conv.weight, bn.*
    def __init__(self, cin, cout, k, s):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, k, s, k // 2, bias=False)
        self.bn = nn.BatchNorm2d(cout)

    def forward(self, x):
        return F.relu(self.bn(self.conv(x)))


class SeqConvBnRelu(nn.Module):  # reference `ConvBNReLU`: keys seq.conv.weight, seq.bn.*
    def __init__(self, cin, cout, k):
        super().__init__()
        self.seq = nn.Sequential()
        self.seq.add_module("conv", nn.Conv2d(cin, cout, k, 1, k // 2, bias=False))
        self.seq.add_module("bn", nn.BatchNorm2d(cout))

    def forward(self, x):
        return F.relu(self.seq(x))


class QARep(nn.Module):
    def __init__(self, cin, cout, stride=1, residual=True, use_alpha=False):
        super().__init__()
        # qarepvgg_block.py:130-136: learnable [1] multiplier of the 1x1 branch when use_alpha, else the float 1.0
        self.alpha = nn.Parameter(torch.tensor([1.0]), requires_grad=True) if use_alpha else 1.0
        self.branch_3x3 = nn.Sequential()
        self.branch_3x3.add_module("conv", nn.Conv2d(cin, cout, 3, stride, 1, bias=False))
        self.branch_3x3.add_module("bn", nn.BatchNorm2d(cout))
        self.branch_1x1 = nn.Conv2d(cin, cout, 1, stride, 0, bias=True)
        self.residual = residual and cin == cout and stride == 1
        self.post_bn = nn.BatchNorm2d(cout)
        self.rbr_reparam = nn.Conv2d(cin, cout, 3, stride, 1, bias=True)  # unused placeholder (qarepvgg_block.py:166)

    def forward(self, x):
        s = self.branch_3x3(x) + self.alpha * self.branch_1x1(x)
        if self.residual:
            s = s + x
        return F.relu(self.post_bn(s))


class Bottleneck(nn.Module):
    def __init__(self, c, block):
        super().__init__()
        self.cv1 = block(c, c)
        self.cv2 = block(c, c)
        self.alpha = nn.Parameter(torch.tensor([1.0]))

    def forward(self, x):
        return self.alpha * x + self.cv2(self.cv1(x))


class CSP(nn.Module):
    def __init__(self, cin, cout, n, hidden, concat, block):
        super().__init__()
        self.conv1 = ConvBnAct(cin, hidden, 1, 1)
        self.conv2 = ConvBnAct(cin, hidden, 1, 1)
        self.conv3 = ConvBnAct(hidden * (2 + concat * n), cout, 1, 1)
        self.bottlenecks = nn.Sequential(*[Bottleneck(hidden, block) for _ in range(n)])
        self.concat = concat

    def forward(self, x):
        y = self.conv1(x)
        outs = [y]
        for b in self.bottlenecks:
            outs.append(b(outs[-1]))
        if not self.concat:
            outs = outs[-1:]
        return self.conv3(torch.cat(outs + [self.conv2(x)], 1))


def _qa(cin, cout):
    return QARep(cin, cout)


def _plain3(cin, cout):
    return ConvBnAct(cin, cout, 3, 1)


class Stem(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.conv = QARep(cin, cout, 2, residual=False)

    def forward(self, x):
        return self.conv(x)


class Stage(nn.Module):
    def __init__(self, cin, cout, n, hidden, concat):
        super().__init__()
        self.downsample = QARep(cin, cout, 2, residual=False)
        self.blocks = CSP(cout, cout, n, hidden, concat, _qa)

    def forward(self, x):
        return self.blocks(self.downsample(x))


class SPP(nn.Module):
    def __init__(self, cin, cout, ks=(5, 9, 13)):
        super().__init__()
        self.cv1 = ConvBnAct(cin, cin // 2, 1, 1)
        self.cv2 = ConvBnAct(cin // 2 * (len(ks) + 1), cout, 1, 1)
        self.ks = ks

    def forward(self, x):
        x = self.cv1(x)
        return self.cv2(torch.cat([x] + [F.max_pool2d(x, k, 1, k // 2) for k in self.ks], 1))


class Backbone(nn.Module):
    def __init__(self, a, in_channels):
        super().__init__()
        self.stem = Stem(in_channels, 48)
        c = 48
        for i in range(4):
            setattr(self, f"stage{i + 1}", Stage(c, STAGE_OUT[i], a["stage_blocks"][i], a["stage_hidden"][i], a["stage_concat"][i]))
            c = STAGE_OUT[i]
        self.context_module = SPP(768, 768)

    def forward(self, x):
        x = self.stem(x)
        c2 = self.stage1(x)
        c3 = self.stage2(c2)
        c4 = self.stage3(c3)
        c5 = self.context_module(self.stage4(c4))
        return c2, c3, c4, c5


class UpStage(nn.Module):
    def __init__(self, cins, cout, n, hidden, concat):
        super().__init__()
        cin, cs1, cs2 = cins
        self.reduce_skip1 = ConvBnAct(cs1, cout, 1, 1)
        self.reduce_skip2 = ConvBnAct(cs2, cout, 1, 1)
        self.conv = ConvBnAct(cin, cout, 1, 1)
        self.upsample = nn.ConvTranspose2d(cout, cout, 2, 2)
        self.downsample = ConvBnAct(cout, cout, 3, 2)
        self.reduce_after_concat = ConvBnAct(3 * cout, cout, 1, 1)
        self.blocks = CSP(cout, cout, n, hidden, concat, _qa)

    def forward(self, x, s1, s2):
        xi = self.conv(x)
        y = torch.cat([self.upsample(xi), self.reduce_skip1(s1), self.downsample(self.reduce_skip2(s2))], 1)
        return xi, self.blocks(self.reduce_after_concat(y))


class DownStage(nn.Module):
    def __init__(self, cins, cout, n, hidden, concat):
        super().__init__()
        cin, cskip = cins
        self.conv = ConvBnAct(cin, cout // 2, 3, 2)
        self.blocks = CSP(cout // 2 + cskip, cout, n, hidden, concat, _plain3)

    def forward(self, x, skip):
        return self.blocks(torch.cat([self.conv(x), skip], 1))


class Neck(nn.Module):
    def __init__(self, a):
        super().__init__()
        nb, nh, cc = a["neck_blocks"], a["neck_hidden"], a["neck_concat"]
        self.neck1 = UpStage([768, 384, 192], NECK_OUT[0], nb[0], nh[0], cc)
        self.neck2 = UpStage([NECK_OUT[0], 192, 96], NECK_OUT[1], nb[1], nh[1], cc)
        self.neck3 = DownStage([NECK_OUT[1], NECK_OUT[1]], NECK_OUT[2], nb[2], nh[2], cc)
        self.neck4 = DownStage([NECK_OUT[2], NECK_OUT[0]], NECK_OUT[3], nb[3], nh[3], cc)

    def forward(self, c2, c3, c4, c5):
        i1, x = self.neck1(c5, c4, c3)
        i2, p3 = self.neck2(x, c3, c2)
        p4 = self.neck3(p3, i2)
        p5 = self.neck4(p4, i1)
        return p3, p4, p5


class DFLHead(nn.Module):
    def __init__(self, cin, inter, num_classes, reg_max):
        super().__init__()
        self.stem = SeqConvBnRelu(cin, inter, 1)
        self.cls_convs = nn.Sequential(SeqConvBnRelu(inter, inter, 3))
        self.reg_convs = nn.Sequential(SeqConvBnRelu(inter, inter, 3))
        self.cls_pred = nn.Conv2d(inter, num_classes, 1)
        self.reg_pred = nn.Conv2d(inter, 4 * (reg_max + 1), 1)
        nn.init.constant_(self.cls_pred.bias, -math.log((1 - 1e-2) / 1e-2))  # dfl_heads.py:98-100

    def forward(self, x):
        x = self.stem(x)
        return self.reg_pred(self.reg_convs(x)), self.cls_pred(self.cls_convs(x))


def make_anchors(hw: List, strides, cell=5.0, offset=0.5):
    """pp_yolo_head.py:21-76 (anchors/anchor_points in pixels) and dfl_heads.py:251-282 (points in grid units)."""
    anchors, pts, pts_grid, strd, counts = [], [], [], [], []
    for (h, w), s in zip(hw, strides):
        half = cell * s * 0.5
        sx = (torch.arange(w) + offset) * s
        sy = (torch.arange(h) + offset) * s
        yy, xx = torch.meshgrid(sy, sx, indexing="ij")
        anchors.append(torch.stack([xx - half, yy - half, xx + half, yy + half], -1).float().reshape(-1, 4))
        pts.append(torch.stack([xx, yy], -1).float().reshape(-1, 2))
        gx = torch.arange(w, dtype=torch.float32) + offset
        gy = torch.arange(h, dtype=torch.float32) + offset
        gyy, gxx = torch.meshgrid(gy, gx, indexing="ij")
        pts_grid.append(torch.stack([gxx, gyy], -1).reshape(-1, 2))
        counts.append(h * w)
        strd.append(torch.full([h * w, 1], float(s)))
    return torch.cat(anchors), torch.cat(pts), torch.cat(pts_grid), counts, torch.cat(strd)


class Heads(nn.Module):
    def __init__(self, a, cins, num_classes, reg_max=16):
        super().__init__()
        self.reg_max, self.num_classes = reg_max, num_classes
        for i in range(3):
            setattr(self, f"head{i + 1}", DFLHead(cins[i], _ceil_mult(HEAD_INTER[i], a["head_mult"], 8), num_classes, reg_max))

    def forward(self, feats):
        cls, dist, red, hw = [], [], [], []
        for i, f in enumerate(feats):
            b, _, h, w = f.shape
            r, c = getattr(self, f"head{i + 1}")(f)
            hw.append((h, w))
            dist.append(r.flatten(2).permute(0, 2, 1))
            d = r.reshape(b, 4, self.reg_max + 1, h * w).permute(0, 2, 3, 1)  # [B,17,HW,4]
            proj = torch.linspace(0, self.reg_max, self.reg_max + 1).reshape(1, -1, 1, 1)
            red.append((F.softmax(d, dim=1) * proj).sum(1))
            cls.append(c.reshape(b, self.num_classes, h * w))
        cls = torch.cat(cls, -1).permute(0, 2, 1)
        dist = torch.cat(dist, 1)
        red = torch.cat(red, 1)
        anchors, pts, pts_grid, counts, strides = make_anchors(hw, STRIDES)
        lt, rb = red[..., :2], red[..., 2:]
        boxes = torch.cat([pts_grid - lt, pts_grid + rb], -1) * strides
        return (boxes, cls.sigmoid()), (cls, dist, anchors, pts, counts, strides)


class YoloNAS(nn.Module):
    def __init__(self, variant="s", num_classes=80, in_channels=3, bn_eps=1e-3, bn_momentum=0.03):
        super().__init__()
        a = ARCH[variant]
        self.backbone = Backbone(a, in_channels)
        self.neck = Neck(a)
        self.heads = Heads(a, [NECK_OUT[1], NECK_OUT[2], NECK_OUT[3]], num_classes)
        for m in self.modules():
            if type(m) is nn.BatchNorm2d:
                m.eps, m.momentum = bn_eps, bn_momentum

    def forward(self, x):
        return self.heads(self.neck(*self.backbone(x)))

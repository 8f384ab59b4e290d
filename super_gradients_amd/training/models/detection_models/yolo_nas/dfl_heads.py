"""YOLO-NAS detection heads on the HIP kernels.

Reference: YoloNASDFLHead yolo_nas/dfl_heads.py:21-117 (1x1 stem; 3x3 -> 1x1 class branch; 3x3 -> 1x1 regression branch,
class bias initialised to -log(99)), NDFLHeads :120-282 (per-level heads, softmax(17)·[0..16] decode, anchors from
pp_yolo_e/pp_yolo_head.py:21-76).  Output structure is the reference's:
  ((pred_bboxes [B,L,4] xyxy px, pred_scores [B,L,C]),
   (cls_logits [B,L,C], reg_distri [B,L,4*(reg_max+1)], anchors [L,4], anchor_points [L,2], num_anchors_list, stride [L,1]))
MI355X structure: the prediction convs write NHWC rows straight into their level's row range of the [B,L,C] /
[B,L,68] buffers (no permute / flatten / cat), decode + sigmoid is one kernel, anchors are cached per feature size.
"""
import math
from typing import List, Tuple

import torch
from torch import nn

from ..... import kernels as K
from .....common.factories import DetectionModulesFactory
from .....common.registry import register_detection_module
from .....modules.base_modules import BaseDetectionModule, width_multiplier
from .....modules.conv_bn_act_block import ConvBNReLU
from .....modules.engine import SgxBlock
from .....modules.layers import ConvLayer


class _Seq1(nn.Module):
    """cls_convs / reg_convs: a one-element sequence (child name "0"), as in the reference for first_conv_group_size=0."""

    def __init__(self, block):
        super().__init__()
        self.add_module("0", block)

    @property
    def block(self):
        return self._modules["0"]


class _PredConv(ConvLayer):
    """nn.Conv2d(inter, n_out, k) with bias; writes into a row range of the level-concatenated prediction buffer.

    n_out is the class count for the classification branch - any number, not only multiples of 4 (fine-tuning on custom datasets:
    the reference's own unit test builds YOLO-NAS with 17 classes, tests/unit_tests/yolo_nas_tests.py:13-18).  The forward conv kernel
    has a scalar epilogue for that; the backward kernels read dy in 16-byte groups, so for n_out % 4 != 0 the gradient is computed
    against zero-padded copies (dy, weights) and the first n_out rows of the padded weight gradient are added into the arena."""

    def fwd(self, x, out=None):
        self._x = x if self.training else None
        return self.conv(x, out=out)

    def bwd(self, dy, need_dx=True, dx_req=None, **kw):
        """dx_req: BatchNorm reduce requests of the layer whose output gradient this conv's dx is (ConvLayer.dgrad)"""
        x, self._x = self._x, None
        if self.out_channels % 4 == 0:
            self.wgrad(x, dy)
            return self.dgrad(dy, tuple(x.shape), reqs=dx_req, **kw) if need_dx else None
        k, kp, r, dev = self.out_channels, (self.out_channels + 3) // 4 * 4, self.kernel_size, dy.device
        cp = self._w.shape[1]
        dyp = torch.zeros(*dy.shape[:3], kp, device=dev, dtype=torch.float32)
        dyp[..., :k].copy_(dy)

        def rows(t, n):   # [K,C,R,S] logical / OHWI memory -> [1,1,n,R*S*C] view of its first n filters
            return t.permute(0, 2, 3, 1).reshape(1, 1, t.shape[0], -1)[:, :, :n]

        def weight_gradient():
            gw = K.ohwi_empty(kp, cp, r, r, dev)
            gw.zero_()
            gb = torch.zeros(kp, device=dev, dtype=torch.float32)
            K.conv2d_bwd_weight(x, dyp, gw, gb, stride=self.stride, pad=self.padding)
            K.axpy(rows(gw, k), out=rows(self._gw, k), accumulate=True)
            self.bias.grad.add_(gb[:k])

        self._net.fork_side(weight_gradient, x, dyp)
        if not need_dx:
            return None
        wp = K.ohwi_empty(kp, cp, r, r, dev)
        wp.zero_()
        wp[:k].copy_(self._w)
        return K.conv2d_bwd_data(dyp, wp, tuple(x.shape), stride=self.stride, pad=self.padding, **kw)


@register_detection_module()
class YoloNASDFLHead(BaseDetectionModule):
    def __init__(self, in_channels, inter_channels, width_mult, first_conv_group_size, num_classes, stride, reg_max, cls_dropout_rate=0.0,
                 reg_dropout_rate=0.0):
        super().__init__(in_channels)
        if first_conv_group_size != 0:
            raise NotImplementedError("YoloNASDFLHead on the HIP path: first_conv_group_size=0 (no grouped first conv), as in the S/M/L arch YAMLs")
        if cls_dropout_rate or reg_dropout_rate:
            raise NotImplementedError("head dropout is not used by the S/M/L recipes")
        inter = width_multiplier(inter_channels, width_mult, 8)
        self.num_classes, self.reg_max, self.stride = num_classes, reg_max, stride
        self.stem = ConvBNReLU(in_channels, inter, kernel_size=1, stride=1, padding=0, bias=False)
        self.cls_convs = _Seq1(ConvBNReLU(inter, inter, kernel_size=3, stride=1, padding=1, bias=False))
        self.reg_convs = _Seq1(ConvBNReLU(inter, inter, kernel_size=3, stride=1, padding=1, bias=False))
        self.cls_pred = _PredConv(inter, num_classes, 1, 1, 0, bias=True)
        self.reg_pred = _PredConv(inter, 4 * (reg_max + 1), 1, 1, 0, bias=True)
        self.prior_prob = 1e-2
        nn.init.constant_(self.cls_pred.bias, -math.log((1 - self.prior_prob) / self.prior_prob))

    def replace_num_classes(self, num_classes: int, compute_new_weights_fn):
        """dfl_heads.py:77-79: only the class-prediction conv is replaced."""
        self.cls_pred = compute_new_weights_fn(self.cls_pred, num_classes)
        self.num_classes = num_classes

    @property
    def out_channels(self):
        return None

    def fwd(self, x, out=None):
        """out = (reg_view [B,H,W,68], cls_view [B,H,W,C]) slices of the level-concatenated buffers."""
        reg_out, cls_out = out
        f = self.stem.fwd(x)
        self.cls_pred.fwd(self.cls_convs.block.fwd(f), out=cls_out)
        self.reg_pred.fwd(self.reg_convs.block.fwd(f), out=reg_out)

    def bwd(self, d_reg, d_cls):
        # BatchNorm reduces riding in data gradients: the two 3x3 blocks' in their prediction convs', the stem's in the regression
        # block's (the second, accumulating writer of the stem's output gradient)
        def req(block):
            r = block.bn_reduce_request()
            return [r] if r is not None else None

        df = self.cls_convs.block.bwd(self.cls_pred.bwd(d_cls, dx_req=req(self.cls_convs.block)))
        self.reg_convs.block.bwd(self.reg_pred.bwd(d_reg, dx_req=req(self.reg_convs.block)), dx_out=df, accumulate=True, dx_req=req(self.stem))
        return self.stem.bwd(df)


@register_detection_module()
class NDFLHeads(BaseDetectionModule):
    def __init__(self, num_classes, in_channels: Tuple[int, int, int], heads_list, grid_cell_scale=5.0, grid_cell_offset=0.5, reg_max=16,
                 eval_size=None, width_mult=1.0):
        super().__init__(in_channels)
        in_channels = [max(round(c * width_mult), 1) for c in in_channels]
        self.in_channels = tuple(in_channels)
        self.num_classes, self.reg_max = num_classes, reg_max
        self.grid_cell_scale, self.grid_cell_offset = grid_cell_scale, grid_cell_offset
        self.eval_size = eval_size
        f = DetectionModulesFactory()
        heads_list = list(heads_list)
        strides: List[int] = []
        self.num_heads = len(heads_list)
        for i in range(self.num_heads):
            conf = f.insert_module_param(heads_list[i], "num_classes", num_classes)
            conf = f.insert_module_param(conf, "reg_max", reg_max)
            head = f.get(f.insert_module_param(conf, "in_channels", in_channels[i]))
            strides.append(head.stride)
            setattr(self, f"head{i + 1}", head)
        self.fpn_strides = tuple(strides)
        self._anchor_cache = {}

    @property
    def out_channels(self):
        return None

    def replace_num_classes(self, num_classes: int, compute_new_weights_fn):
        """dfl_heads.py:165-170"""
        for i in range(self.num_heads):
            getattr(self, f"head{i + 1}").replace_num_classes(num_classes, compute_new_weights_fn)
        self.num_classes = num_classes

    def anchors_for(self, sizes, device):
        """generate_anchors_for_grid_cell (pp_yolo_head.py:21-76) + the grid-unit points of dfl_heads.py:251-282; cached."""
        key = (tuple(sizes), str(device))
        hit = self._anchor_cache.get(key)
        if hit is not None:
            return hit
        anchors, pts, pts_grid, counts, strides = [], [], [], [], []
        for (h, w), s in zip(sizes, self.fpn_strides):
            half = self.grid_cell_scale * s * 0.5
            sx = (torch.arange(end=w) + self.grid_cell_offset) * s
            sy = (torch.arange(end=h) + self.grid_cell_offset) * s
            yy, xx = torch.meshgrid(sy, sx, indexing="ij")
            anchors.append(torch.stack([xx - half, yy - half, xx + half, yy + half], dim=-1).to(torch.float32).reshape(-1, 4))
            pts.append(torch.stack([xx, yy], dim=-1).to(torch.float32).reshape(-1, 2))
            gx = torch.arange(end=w, dtype=torch.float32) + self.grid_cell_offset
            gy = torch.arange(end=h, dtype=torch.float32) + self.grid_cell_offset
            gyy, gxx = torch.meshgrid(gy, gx, indexing="ij")
            pts_grid.append(torch.stack([gxx, gyy], dim=-1).reshape(-1, 2))
            counts.append(h * w)
            strides.append(torch.full([h * w, 1], s, dtype=torch.float32))
        hit = (torch.cat(anchors).to(device), torch.cat(pts).to(device), torch.cat(pts_grid).to(device).contiguous(), counts,
               torch.cat(strides).to(device))
        self._anchor_cache[key] = hit
        return hit

    def fwd(self, feats, out=None):
        feats = feats[: self.num_heads]
        B = feats[0].shape[0]
        sizes = [(f.shape[1], f.shape[2]) for f in feats]
        anchors, pts, pts_grid, counts, strides = self.anchors_for(sizes, feats[0].device)
        L, C, R4 = sum(counts), self.num_classes, 4 * (self.reg_max + 1)
        logits = torch.empty(B, L, C, device=feats[0].device, dtype=torch.float32)
        distri = torch.empty(B, L, R4, device=feats[0].device, dtype=torch.float32)
        off, calls = 0, []
        for i, f in enumerate(feats):
            h, w = sizes[i]
            calls.append((getattr(self, f"head{i + 1}"), f, (distri[:, off:off + h * w].view(B, h, w, R4), logits[:, off:off + h * w].view(B, h, w, C))))
            off += h * w
        # the levels are independent of one another: the coarse ones (a quarter and a sixteenth of the first level's pixels - launches that do
        # not fill the chip) run on the branch stream beside the first (engine.fork_branch), joined before the decode
        net = getattr(self, "_net", None)
        self._branched = net is not None and len(calls) > 1 and self.training and net.branches(2, B * counts[1], 64)
        joins = [net.fork_branch(lambda c=c: c[0].fwd(c[1], out=c[2]), lane=i)[1] for i, c in enumerate(calls[1:])] if self._branched else []
        for h_, f, o in (calls[:1] if self._branched else calls):
            h_.fwd(f, out=o)
        for j in joins:
            j()
        boxes, scores = K.dfl_decode(logits, distri, pts_grid, strides, self.reg_max)
        self._sizes = sizes
        return boxes, scores, logits, distri, anchors, pts, counts, strides

    def bwd(self, d_logits, d_distri):
        sizes = self._sizes
        B, _, C = d_logits.shape
        R4 = d_distri.shape[2]
        calls, off = [], 0
        for i, (h, w) in enumerate(sizes):
            calls.append((getattr(self, f"head{i + 1}"), d_distri[:, off:off + h * w].view(B, h, w, R4), d_logits[:, off:off + h * w].view(B, h, w, C)))
            off += h * w
        # (branch stream in backward only for a forward that ran there: the coarse levels' saved tensors are that stream's pool's then)
        net = getattr(self, "_net", None)
        if getattr(self, "_branched", False) and net is not None and net.branches(2, 0, 0, True):
            forks = [net.fork_branch(lambda c=c: c[0].bwd(c[1], c[2]), backward=True, lane=i) for i, c in enumerate(calls[1:])]
            first = calls[0][0].bwd(calls[0][1], calls[0][2])
            for _, j in forks:
                j()
            return [first] + [g for g, _ in forks]
        return [h_.bwd(dr, dc) for h_, dr, dc in calls]

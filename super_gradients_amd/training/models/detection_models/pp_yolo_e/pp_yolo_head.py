"""PPYOLOEHead on the HIP kernels.

Reference (training/models/detection_models/pp_yolo_e/pp_yolo_head.py): ESEAttn :79-93 (conv(feat * sigmoid(fc(avg_feat)))),
PPYOLOEHead :96-301 - per level: cls_logit = pred_cls(stem_cls(feat, avg) + feat), reg_distri = pred_reg(stem_reg(feat, avg)),
3x3 prediction convs with bias (class bias -log(99), regression bias 1.0, zero weights at init), anchors from
generate_anchors_for_grid_cell :21-76; training returns the raw 6-tuple, eval ((boxes, scores), raw).
State_dict keys: stem_cls.{i}.{fc.weight,fc.bias,conv.seq.*}, stem_reg.{i}.*, pred_cls.{i}.{weight,bias}, pred_reg.{i}.{weight,bias}.

MI355X structure: the per-image channel means are computed once per level and shared by both attention stems; the gate multiply is
one sweep; `+ feat` is folded into the stem's BatchNorm/activation sweep; the prediction convs write NHWC rows straight into
their level's row range of the [B,L,C] / [B,L,4*(reg_max+1)] buffers (no permute / flatten / cat).
"""
import math
from typing import Tuple

import torch
from torch import nn

from ..... import kernels as K
from .....common.registry import register_detection_module
from .....modules.base_modules import BaseDetectionModule
from .....modules.conv_bn_act_block import ConvBNAct
from .....modules.engine import SgxBlock
from .....modules.layers import ConvLayer, act_name
from ..yolo_nas.dfl_heads import NDFLHeads, _PredConv


class ESEAttn(SgxBlock):
    GATE = "sigmoid"

    def __init__(self, feat_channels: int, activation_type):
        super().__init__()
        self.fc = ConvLayer(feat_channels, feat_channels, 1, 1, 0, bias=True)
        self.conv = ConvBNAct(feat_channels, feat_channels, kernel_size=1, padding=0, stride=1, activation_type=activation_type, bias=False)
        torch.nn.init.normal_(self.fc.weight, std=0.001)

    def on_materialize(self):
        pass

    def fwd(self, feat, avg_feat, out=None, post_add=None):
        """avg_feat: [N,1,1,C] per-image channel means of feat."""
        n, c = feat.shape[0], feat.shape[3]
        pre = self.fc.conv(avg_feat).view(n, c)
        gated = K.channel_gate(feat, pre, self.GATE)
        self._ctx = (feat, avg_feat, pre) if self.training else None
        return self.conv.fwd(gated, out=out, post_add=post_add)

    def bwd(self, dy, davg=None):
        """-> (d_gated, pre, davg): the caller folds d_gated * sigmoid(pre) into its accumulated feature gradient; davg [N,1,1,C] is
        accumulated across the level's two stems."""
        (feat, avg_feat, pre), self._ctx = self._ctx, None
        n, c = feat.shape[0], feat.shape[3]
        dg = self.conv.bwd(dy)
        dpre = K.image_colsum(dg, v=feat, pre=pre, gate=self.GATE).view(n, 1, 1, c)
        self.fc.wgrad(avg_feat, dpre)
        davg = self.fc.dgrad(dpre, (n, 1, 1, c), out=davg, accumulate=davg is not None)
        return dg, pre, davg


@register_detection_module()
class PPYOLOEHead(BaseDetectionModule):
    anchors_for = NDFLHeads.anchors_for  # generate_anchors_for_grid_cell + grid-unit points, cached per feature-map size

    def __init__(self, num_classes: int, in_channels: Tuple[int, int, int], activation="silu", fpn_strides: Tuple[int, int, int] = (32, 16, 8),
                 grid_cell_scale=5.0, grid_cell_offset=0.5, reg_max=16, eval_size: Tuple[int, int] = None, width_mult: float = 1.0):
        super().__init__(in_channels)
        act = act_name(activation)
        in_channels = [max(round(c * width_mult), 1) for c in in_channels]
        self.in_channels = tuple(in_channels)
        self.num_classes = num_classes
        self.fpn_strides = tuple(fpn_strides)
        self.grid_cell_scale, self.grid_cell_offset = grid_cell_scale, grid_cell_offset
        self.reg_max = reg_max
        self.eval_size = eval_size
        self.stem_cls = nn.ModuleList([ESEAttn(c, activation_type=act) for c in self.in_channels])
        self.stem_reg = nn.ModuleList([ESEAttn(c, activation_type=act) for c in self.in_channels])
        self.pred_cls = nn.ModuleList([_PredConv(c, num_classes, 3, 1, 1, bias=True) for c in self.in_channels])
        self.pred_reg = nn.ModuleList([_PredConv(c, 4 * (reg_max + 1), 3, 1, 1, bias=True) for c in self.in_channels])
        self._anchor_cache = {}
        self._init_weights()

    def _init_weights(self):
        bias_cls = -math.log((1 - 0.01) / 0.01)   # bias_init_with_prob(0.01)
        for cls_, reg_ in zip(self.pred_cls, self.pred_reg):
            torch.nn.init.constant_(cls_.weight, 0.0)
            torch.nn.init.constant_(cls_.bias, bias_cls)
            torch.nn.init.constant_(reg_.weight, 0.0)
            torch.nn.init.constant_(reg_.bias, 1.0)

    def replace_num_classes(self, num_classes: int):
        """pp_yolo_head.py:167-177: new class-prediction convs, zero weights and the -log(99) prior bias."""
        self.num_classes = num_classes
        self.pred_cls = nn.ModuleList([_PredConv(c, num_classes, 3, 1, 1, bias=True) for c in self.in_channels])
        for cls_ in self.pred_cls:
            torch.nn.init.constant_(cls_.weight, 0.0)
            torch.nn.init.constant_(cls_.bias, -math.log((1 - 0.01) / 0.01))

    def cache_anchors(self, input_size):
        self.eval_size = list(input_size)[-2:]

    @property
    def out_channels(self):
        return None

    def fwd(self, feats, out=None):
        B = feats[0].shape[0]
        dev = feats[0].device
        sizes = [(f.shape[1], f.shape[2]) for f in feats]
        anchors, pts, pts_grid, counts, strides = self.anchors_for(sizes, dev)
        L, C, R4 = sum(counts), self.num_classes, 4 * (self.reg_max + 1)
        logits = torch.empty(B, L, C, device=dev, dtype=torch.float32)
        distri = torch.empty(B, L, R4, device=dev, dtype=torch.float32)
        off = 0
        for i, f in enumerate(feats):
            h, w = sizes[i]
            c = f.shape[3]
            avg = K.image_colsum(f, scale=1.0 / (h * w)).view(B, 1, 1, c)
            self.pred_cls[i].fwd(self.stem_cls[i].fwd(f, avg, post_add=f), out=logits[:, off:off + h * w].view(B, h, w, C))
            self.pred_reg[i].fwd(self.stem_reg[i].fwd(f, avg), out=distri[:, off:off + h * w].view(B, h, w, R4))
            off += h * w
        self._sizes = sizes
        if self.training:
            return None, None, logits, distri, anchors, pts, counts, strides
        boxes, scores = K.dfl_decode(logits, distri, pts_grid, strides, self.reg_max)
        return boxes, scores, logits, distri, anchors, pts, counts, strides

    def bwd(self, d_logits, d_distri):
        """-> gradients of the input feature maps (same order as fwd's feats)."""
        sizes = self._sizes
        B, _, C = d_logits.shape
        R4 = d_distri.shape[2]
        grads, off = [], 0
        for i, (h, w) in enumerate(sizes):
            dfeat = self.pred_cls[i].bwd(d_logits[:, off:off + h * w].view(B, h, w, C))   # d(stem_cls out + feat): its `+ feat` term
            dg, pre, davg = self.stem_cls[i].bwd(dfeat)
            K.channel_gate(dg, pre, ESEAttn.GATE, out=dfeat, accumulate=True)
            dg, pre, davg = self.stem_reg[i].bwd(self.pred_reg[i].bwd(d_distri[:, off:off + h * w].view(B, h, w, R4)), davg)
            c = dfeat.shape[3]
            K.channel_gate(dg, pre, ESEAttn.GATE, bias=davg.view(B, c), bias_scale=1.0 / (h * w), out=dfeat, accumulate=True)
            grads.append(dfeat)
            off += h * w
        return grads

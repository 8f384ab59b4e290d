"""YOLO-NAS stages on the HIP kernels: stem, backbone stage, CSP layer, bottleneck, up/down neck stages.

Reference (training/models/detection_models/yolo_nas/yolo_stages.py): YoloNASBottleneck :23-63, YoloNASCSPLayer :85-150,
YoloNASStem :153-184, YoloNASStage :187-235, YoloNASUpStage :238-332, YoloNASDownStage :335-395.  Same constructor
arguments (the ones the arch YAMLs use), same child names -> same state_dict keys.

MI355X-specific structure: there is no torch.cat.  Every producer of a concatenated tensor writes its NHWC output
straight into its channel slice of one preallocated buffer (the conv / sweep kernels take explicit pixel strides), and
in backward the consumers read their slice of the concat gradient in place.
"""
import os
from functools import partial
from typing import List

import torch
from torch import nn

from ..... import kernels as K
from .....common.registry import register_detection_module
from .....modules.base_modules import BaseDetectionModule, width_multiplier
from .....modules.conv_bn_act_block import Conv
from .....modules.engine import SgxBlock
from .....modules.layers import ConvTranspose2x2, act_name
from .....modules.qarepvgg_block import QARepVGGBlock

_DALPHA_SYNC = os.environ.get("SGX_DALPHA_SYNC", "0") == "1"  # bit-reproducible d alpha at -8 % of the step (YoloNASBottleneck.bwd)


def _empty(n, h, w, c, like):
    return torch.empty(n, h, w, c, device=like.device, dtype=like.dtype)  # (bf16 on the half-precision inference path, fp32 otherwise)


class YoloNASBottleneck(SgxBlock):
    """z = alpha * x + cv2(cv1(x))   (yolo_stages.py:61-63); alpha is a learnable [1] parameter when use_alpha."""

    def __init__(self, input_channels, output_channels, block_type, activation_type, shortcut: bool, use_alpha: bool, drop_path_rate: float = 0.0):
        super().__init__()
        if drop_path_rate > 0.0:
            raise NotImplementedError("drop_path inside YOLO-NAS bottlenecks is not used by the S/M/L recipes")
        self.cv1 = block_type(input_channels, output_channels, activation_type=activation_type)
        self.cv2 = block_type(output_channels, output_channels, activation_type=activation_type)
        self.add = shortcut and input_channels == output_channels
        if use_alpha:
            self.alpha = nn.Parameter(torch.tensor([1.0]), requires_grad=True)
        else:
            self.alpha = 1.0

    def on_materialize(self):
        pass

    def _alpha(self):
        return (1.0, self.alpha) if isinstance(self.alpha, torch.Tensor) else (float(self.alpha), None)

    def fwd(self, x, out=None):
        if not self.add:
            return self.cv2.fwd(self.cv1.fwd(x), out=out)
        a, a_dev = self._alpha()
        self._x = x if self.training else None
        if x.dtype == K.HALF:  # half-precision inference: the shortcut rides in the epilogue of cv2's (fused) convolution
            return self.cv2.fwd(self.cv1.fwd(x), out=out, post_add=x, post_scale=a_dev if a_dev is not None else a)
        if isinstance(self.cv2, QARepVGGBlock):  # the shortcut rides in cv2's last sweep
            return self.cv2.fwd(self.cv1.fwd(x), out=out, post_add=x, post_scale=a_dev if a_dev is not None else a)
        y = self.cv2.fwd(self.cv1.fwd(x))
        return K.affine_act(y, r1=x, a1=a, a1_dev=a_dev, out=out if out is not None else y)

    def _mid_req(self):
        """cv1's BatchNorm reduce rides in cv2's data gradient (the only writer of cv1's output gradient) when cv1 is a plain conv block"""
        r = self.cv1.bn_reduce_request() if hasattr(self.cv1, "bn_reduce_request") else None
        return [r] if r is not None else None

    def bwd(self, dz, dx_out=None, accumulate=False, addend=None, need_dx=True, dx_req=None):
        """dx_req: reduce requests of the layer(s) whose output gradient this block's dx is - cv1's data gradient writes it last"""
        if not self.add:
            return self.cv1.bwd(self.cv2.bwd(dz, dx_req=self._mid_req()), dx_out=dx_out, accumulate=accumulate, addend=addend, need_dx=need_dx, dx_req=dx_req)
        x, self._x = self._x, None
        a, a_dev = self._alpha()
        joined = None
        if a_dev is not None:
            # d alpha = <x, dz>: two launches that read both tensors and that nothing waits for before the optimizer - second lane of the
            # branch stream (the first carries the CSP layer's conv2 chain at this point), joined when this block's backward is enqueued
            net = getattr(self, "_net", None)
            d_alpha = lambda: K.dot_sum(x, dz, self.alpha.grad, accumulate=True)  # noqa: E731
            # (Round 6, DESIGN.md 11.12: this dot's last bit did not repeat while a weight-gradient kernel of the side stream was resident beside
            # it - the compiler had packed its TwoSum lanes into v_pk_*_f32 instructions, whose results move beside those kernels; the library is
            # built without such instructions now.  SGX_DALPHA_SYNC=1 - the main chain waits for the side stream first, -8 % of the step - is
            # the switch the cause was bracketed with; off.)
            if _DALPHA_SYNC and net is not None and net.side_stream is not None:
                torch.cuda.current_stream().wait_stream(net.side_stream)
            if net is not None and net.branches(32, 0, 0, True):
                joined = net.fork_branch(d_alpha, backward=True, lane=1, queues_wgrads=False)[1]
            else:
                d_alpha()
        dmid = self.cv2.bwd(dz, dx_req=self._mid_req())
        if isinstance(self.cv1, QARepVGGBlock):  # d(alpha * x) = alpha * dz rides in cv1's data-gradient launch
            dx = self.cv1.bwd(dmid, dx_out=dx_out, accumulate=accumulate, addend=addend, addend2=dz, addend2_scale=a_dev if a_dev is not None else a,
                              dx_req=dx_req)
        else:
            if dx_out is not None:
                K.axpy(dz, a=a, a_dev=a_dev, out=dx_out, accumulate=accumulate)
                pre = dx_out
            else:
                pre = K.axpy(dz, a=a, a_dev=a_dev)
            dx = self.cv1.bwd(dmid, dx_out=pre, accumulate=True, addend=addend, dx_req=dx_req)
        if joined is not None:
            joined()
        return dx


class _BottleneckList(nn.Module):
    """Holds bottlenecks under integer child names (state_dict keys bottlenecks.{i}.*), like the reference's
    SequentialWithIntermediates (yolo_stages.py:66-82)."""

    def __init__(self, output_intermediates, *mods):
        super().__init__()
        self.output_intermediates = output_intermediates
        for i, m in enumerate(mods):
            self.add_module(str(i), m)

    def __iter__(self):
        return iter(self._modules.values())

    def __len__(self):
        return len(self._modules)


class YoloNASCSPLayer(SgxBlock):
    def __init__(self, in_channels, out_channels, num_bottlenecks, block_type, activation_type, shortcut=True, use_alpha=True, expansion=0.5,
                 hidden_channels=None, concat_intermediates=False, drop_path_rates=None, dropout_rate=0.0):
        super().__init__()
        if dropout_rate > 0.0:
            raise NotImplementedError("dropout inside YoloNASCSPLayer is not used by the S/M/L recipes")
        drop_path_rates = [0.0] * num_bottlenecks if drop_path_rates is None else tuple(drop_path_rates)
        if len(drop_path_rates) != num_bottlenecks:
            raise ValueError(f"Argument drop_path_rates ({drop_path_rates}, len {len(drop_path_rates)} must have the length equal to the "
                             f"num_bottlenecks ({num_bottlenecks}).")
        if hidden_channels is None:
            hidden_channels = int(out_channels * expansion)
        self.hidden = hidden_channels
        self.conv1 = Conv(in_channels, hidden_channels, 1, stride=1, activation_type=activation_type)
        self.conv2 = Conv(in_channels, hidden_channels, 1, stride=1, activation_type=activation_type)
        self.n_cat = 2 + int(bool(concat_intermediates)) * num_bottlenecks
        self.conv3 = Conv(hidden_channels * self.n_cat, out_channels, 1, stride=1, activation_type=activation_type)
        self.bottlenecks = _BottleneckList(
            concat_intermediates,
            *[YoloNASBottleneck(hidden_channels, hidden_channels, block_type, activation_type, shortcut, use_alpha, drop_path_rate=drop_path_rates[i])
              for i in range(num_bottlenecks)])
        self.concat_intermediates = bool(concat_intermediates)

    def on_materialize(self):
        pass

    def fwd(self, x, out=None):
        n, h, w, _ = x.shape
        hid = self.hidden
        cat = _empty(n, h, w, hid * self.n_cat, x)
        sl = lambda i: cat[..., i * hid:(i + 1) * hid]  # noqa: E731
        blocks = list(self.bottlenecks)
        # conv2's chain (GEMM, finalize, sweep) is needed by conv3 only: branch stream (engine.fork_branch), joined before conv3
        net = getattr(self, "_net", None)
        self._branched = net is not None and bool(blocks) and self.training and net.branches(1, n * h * w, hid)
        joined = net.fork_branch(lambda: self.conv2.fwd(x, out=sl(self.n_cat - 1)))[1] if self._branched else None
        if self.concat_intermediates:
            cur = self.conv1.fwd(x, out=sl(0))
            for i, b in enumerate(blocks):
                cur = b.fwd(cur, out=sl(i + 1))
        else:
            cur = self.conv1.fwd(x, out=sl(0) if not blocks else None)
            for i, b in enumerate(blocks):
                cur = b.fwd(cur, out=sl(0) if i == len(blocks) - 1 else None)
        if joined is None:
            self.conv2.fwd(x, out=sl(self.n_cat - 1))
        else:
            joined()
        return self.conv3.fwd(cat, out=out)

    def bwd(self, dy, dx_out=None, accumulate=False, addend=None, need_dx=True, dx_req=None):
        """dx_req: reduce requests of the layer(s) whose output gradient this layer's dx is - conv1's data gradient writes it last.
        Inside: conv2's output gradient is a slice of conv3's data gradient (its only writer); conv1's is written last by the first
        bottleneck's data gradient (or is a slice of conv3's when there are no bottlenecks)."""
        hid = self.hidden
        blocks = list(self.bottlenecks)
        r2, r1 = self.conv2.bn_reduce_request(), self.conv1.bn_reduce_request()
        cat_req = [r2.at((self.n_cat - 1) * hid)] if r2 is not None else []
        if not blocks and r1 is not None:
            cat_req.append(r1.at(0))
        dcat = self.conv3.bwd(dy, dx_req=cat_req or None)
        sl = lambda i: dcat[..., i * hid:(i + 1) * hid]  # noqa: E731
        # conv2's backward (finalize, apply sweep, data gradient into dx) meets the main chain at conv1's data gradient: branch stream
        net = getattr(self, "_net", None)
        keep = (self.conv2._ctx, getattr(self.conv2, "_req", None))  # conv2's saved tensors stay referenced until the join (they may be the main stream's pool's)
        conv2_bwd = lambda: self.conv2.bwd(sl(self.n_cat - 1), dx_out=dx_out, accumulate=accumulate, addend=addend)  # noqa: E731
        rows = dy.shape[0] * dy.shape[1] * dy.shape[2]
        dx, joined = net.fork_branch(conv2_bwd, backward=True) if (net is not None and blocks and net.branches(1, rows, hid, True)) else (conv2_bwd(), None)
        first_req = [r1] if (blocks and r1 is not None) else None
        if self.concat_intermediates:
            g = sl(len(blocks))
            for i in range(len(blocks) - 1, -1, -1):  # the block's input gradient accumulates onto the concat slice of the same tensor
                g = blocks[i].bwd(g, dx_out=sl(i), accumulate=True, dx_req=first_req if i == 0 else None)
        else:
            g = sl(0)
            for i in range(len(blocks) - 1, -1, -1):
                g = blocks[i].bwd(g, dx_req=first_req if i == 0 else None)
        if joined is not None:
            joined()
        del keep
        return self.conv1.bwd(g, dx_out=dx, accumulate=True, dx_req=dx_req)


@register_detection_module()
class YoloNASStem(BaseDetectionModule):
    def __init__(self, in_channels: int, out_channels: int, stride: int = 2):
        super().__init__(in_channels)
        self._out_channels = out_channels
        self.conv = QARepVGGBlock(in_channels, out_channels, stride=stride, use_residual_connection=False)

    @property
    def out_channels(self):
        return self._out_channels

    def fwd(self, x, out=None):
        return self.conv.fwd(x, out=out)

    def bwd(self, dy, **kw):
        return self.conv.bwd(dy, **kw)

    def get_input_channels(self) -> int:
        return self.conv.in_channels

    def replace_input_channels(self, in_channels: int, compute_new_weights_fn=None):
        """Reference yolo_stages.py:176-177: the stem block is rebuilt with fresh weights for the new channel count (the optional weight
        function is ignored there too).  Before materialisation only: the arenas own the parameters afterwards."""
        stride = self.conv.stride
        self.conv = QARepVGGBlock(in_channels, self._out_channels, stride=stride, use_residual_connection=False)
        self.in_channels = in_channels


@register_detection_module()
class YoloNASStage(BaseDetectionModule):
    def __init__(self, in_channels, out_channels, num_blocks, activation_type, hidden_channels=None, concat_intermediates=False,
                 drop_path_rates=None, dropout_rate=0.0, stride=2):
        super().__init__(in_channels)
        self._out_channels = out_channels
        act = act_name(activation_type)
        self.downsample = QARepVGGBlock(in_channels, out_channels, stride=stride, activation_type=act, use_residual_connection=False)
        self.blocks = YoloNASCSPLayer(out_channels, out_channels, num_blocks, QARepVGGBlock, act, True, hidden_channels=hidden_channels,
                                      concat_intermediates=concat_intermediates, drop_path_rates=drop_path_rates, dropout_rate=dropout_rate)

    @property
    def out_channels(self):
        return self._out_channels

    def fwd(self, x, out=None):
        return self.blocks.fwd(self.downsample.fwd(x), out=out)

    def bwd(self, dy, dx_out=None, accumulate=False, addend=None, need_dx=True):
        return self.downsample.bwd(self.blocks.bwd(dy), dx_out=dx_out, accumulate=accumulate, addend=addend, need_dx=need_dx)


@register_detection_module()
class YoloNASUpStage(BaseDetectionModule):
    """inputs (x, skip1, skip2) -> (x_inter, out)   (yolo_stages.py:319-332).  Three-input form with reduce_channels
    and ConvTranspose2d upsampling - the configuration every YOLO-NAS arch YAML uses."""

    def __init__(self, in_channels: List[int], out_channels, width_mult, num_blocks, depth_mult, activation_type, hidden_channels=None,
                 concat_intermediates=False, reduce_channels=False, drop_path_rates=None, dropout_rate=0.0, upsample_mode="conv_transpose"):
        super().__init__(in_channels)
        if len(in_channels) != 3 or not reduce_channels or str(upsample_mode).lower() not in ("conv_transpose", "upsamplemode.conv_transpose"):
            raise NotImplementedError("YoloNASUpStage on the HIP path: 3 inputs, reduce_channels=True, conv_transpose upsampling")
        cin, cs1, cs2 = in_channels
        out_channels = width_multiplier(out_channels, width_mult, 8)
        num_blocks = max(round(num_blocks * depth_mult), 1) if num_blocks > 1 else num_blocks
        act = act_name(activation_type)
        self.reduce_skip1 = Conv(cs1, out_channels, 1, 1, act)
        self.reduce_skip2 = Conv(cs2, out_channels, 1, 1, act)
        self.conv = Conv(cin, out_channels, 1, 1, act)
        self.upsample = ConvTranspose2x2(out_channels, out_channels)
        self.downsample = Conv(out_channels, out_channels, kernel=3, stride=2, activation_type=act)
        self.reduce_after_concat = Conv(3 * out_channels, out_channels, 1, 1, act)
        self.blocks = YoloNASCSPLayer(out_channels, out_channels, num_blocks, QARepVGGBlock, act, hidden_channels=hidden_channels,
                                      concat_intermediates=concat_intermediates, drop_path_rates=drop_path_rates, dropout_rate=dropout_rate)
        self._oc = out_channels
        self._out_channels = [out_channels, out_channels]

    @property
    def out_channels(self):
        return self._out_channels

    def pre_skip(self, which: int, s):
        """Start skip branch `which` (1: reduce_skip1(s), 2: downsample(reduce_skip2(s))) on the branch stream as soon as its input exists -
        the backbone calls this while its deeper stages, whose launches do not fill the chip, are still to run; fwd() joins.  No-op (fwd()
        runs the branch) in eval mode or with the branch stream off."""
        net = getattr(self, "_net", None)
        n, h, w, _ = s.shape
        oc = self._oc
        if net is None or not self.training or not net.branches(4, n * h * w, oc) or os.environ.get("SGX_BRANCH_EARLY_FORK", "0") == "0":
            return
        pre = self.__dict__.setdefault("_pre", {})

        def run():
            if "cat" not in pre:
                pre["cat"] = _empty(n, h, w, 3 * oc, s) if which == 1 else _empty(n, (h - 1) // 2 + 1, (w - 1) // 2 + 1, 3 * oc, s)
            cat = pre["cat"]
            if which == 1:
                self.reduce_skip1.fwd(s, out=cat[..., oc:2 * oc])
            else:
                self.downsample.fwd(self.reduce_skip2.fwd(s), out=cat[..., 2 * oc:])

        pre[which] = net.fork_branch(run)[1]

    def fwd(self, inputs, out=None):
        x, s1, s2 = inputs
        n, h, w, _ = s1.shape
        oc = self._oc
        pre, self._pre = getattr(self, "_pre", None) or {}, {}
        cat = pre["cat"] if "cat" in pre else _empty(n, h, w, 3 * oc, x)
        if tuple(cat.shape) != (n, h, w, 3 * oc):
            raise RuntimeError(f"YoloNASUpStage: skip branches were started for a {tuple(cat.shape)} concat, the inputs make {(n, h, w, 3 * oc)}")

        def skips():
            if 1 not in pre:
                self.reduce_skip1.fwd(s1, out=cat[..., oc:2 * oc])
            if 2 not in pre:
                self.downsample.fwd(self.reduce_skip2.fwd(s2), out=cat[..., 2 * oc:])

        # the two skip branches meet the main one in the concat: branch stream (engine.fork_branch; sized by reduce_skip2's GEMM) - started
        # by pre_skip() when the backbone produced their inputs, else here
        net = getattr(self, "_net", None)
        self._branched = net is not None and self.training and net.branches(4, s2.shape[0] * s2.shape[1] * s2.shape[2], oc)
        joins = [pre[k] for k in (1, 2) if k in pre]
        if len(joins) < 2:
            joins.append(net.fork_branch(skips)[1] if self._branched else skips())
        self._branched = self._branched or bool(pre)
        x_inter = self.conv.fwd(x)
        self.upsample.fwd(x_inter, out=cat[..., :oc])
        for j in joins:
            if j is not None:
                j()
        return x_inter, self.blocks.fwd(self.reduce_after_concat.fwd(cat), out=out)

    def bwd(self, d_inter, d_out, dx=None, ds1=None, ds2=None, late_join=False):
        """d_inter: gradient arriving at x_inter from its other consumer (a down stage), or None.
        dx/ds1/ds2: (buffer, accumulate) destinations for the three input gradients.
        late_join: when the skip branches' backward was forked onto the branch stream, leave the join to the caller (join_bwd) - the two skip
        gradients are not read before the backbone's backward reaches their layers."""
        oc = self._oc
        # BatchNorm reduces that ride in data gradients: reduce_after_concat's in the CSP layer's last launch; reduce_skip1's and
        # downsample's (two slices of the concat gradient) in reduce_after_concat's; reduce_skip2's in downsample's (stride 2)
        r_rac = self.reduce_after_concat.bn_reduce_request()
        g_rac = self.blocks.bwd(d_out, dx_req=[r_rac] if r_rac is not None else None)
        r_s1, r_ds = self.reduce_skip1.bn_reduce_request(), self.downsample.bn_reduce_request()
        cat_req = [r.at(c) for r, c in ((r_s1, oc), (r_ds, 2 * oc)) if r is not None]
        dcat = self.reduce_after_concat.bwd(g_rac, dx_req=cat_req or None)

        def skips():
            g1 = self.reduce_skip1.bwd(dcat[..., oc:2 * oc], dx_out=ds1[0], accumulate=ds1[1])
            r_s2 = self.reduce_skip2.bn_reduce_request()
            g2 = self.reduce_skip2.bwd(self.downsample.bwd(dcat[..., 2 * oc:], dx_req=[r_s2] if r_s2 is not None else None), dx_out=ds2[0], accumulate=ds2[1])
            return g1, g2

        # (branch stream in backward only for a forward that ran there: the skip branches' saved tensors are that stream's pool's then)
        net = getattr(self, "_net", None)
        joined = None
        if getattr(self, "_branched", False) and net is not None and net.branches(4, 0, 0, True):
            (g1, g2), joined = net.fork_branch(skips, backward=True)
        g_inter = self.upsample.bwd(dcat[..., :oc])
        if d_inter is not None:
            K.axpy(d_inter, out=g_inter, accumulate=True)
        gx = self.conv.bwd(g_inter, dx_out=dx[0], accumulate=dx[1])
        if joined is None:
            g1, g2 = skips()
        elif late_join:
            self._late = (joined, dcat)  # the caller joins (join_bwd) before g1 / g2 are read; the concat gradient stays referenced until then
        else:
            joined()
        return gx, g1, g2

    def join_bwd(self):
        late, self._late = getattr(self, "_late", None), None
        if late is not None:
            late[0]()


@register_detection_module()
class YoloNASDownStage(BaseDetectionModule):
    """inputs (x, skip) -> out   (yolo_stages.py:390-395); bottlenecks are plain 3x3 Conv blocks."""

    def __init__(self, in_channels: List[int], out_channels, width_mult, num_blocks, depth_mult, activation_type, hidden_channels=None,
                 concat_intermediates=False, drop_path_rates=None, dropout_rate=0.0):
        super().__init__(in_channels)
        cin, cskip = in_channels
        out_channels = width_multiplier(out_channels, width_mult, 8)
        num_blocks = max(round(num_blocks * depth_mult), 1) if num_blocks > 1 else num_blocks
        act = act_name(activation_type)
        self.conv = Conv(cin, out_channels // 2, 3, 2, act)
        self._half, self._cskip = out_channels // 2, cskip
        self.blocks = YoloNASCSPLayer(in_channels=out_channels // 2 + cskip, out_channels=out_channels, num_bottlenecks=num_blocks,
                                      block_type=partial(Conv, kernel=3, stride=1), activation_type=act, hidden_channels=hidden_channels,
                                      concat_intermediates=concat_intermediates, drop_path_rates=drop_path_rates, dropout_rate=dropout_rate)
        self._out_channels = out_channels

    @property
    def out_channels(self):
        return self._out_channels

    def fwd(self, inputs, out=None):
        x, skip = inputs
        n, h, w, _ = skip.shape
        cat = _empty(n, h, w, self._half + self._cskip, x)
        self.conv.fwd(x, out=cat[..., : self._half])
        K.axpy(skip, out=cat[..., self._half:])
        return self.blocks.fwd(cat, out=out)

    def bwd(self, d_out, dx=None):
        """-> (gx, d_skip view); d_skip is a channel slice of the concat gradient (read in place by the up stage)."""
        r = self.conv.bn_reduce_request()  # its output gradient = the first half of the CSP layer's input gradient
        dcat = self.blocks.bwd(d_out, dx_req=[r.at(0)] if r is not None else None)
        gx = self.conv.bwd(dcat[..., : self._half], dx_out=dx[0], accumulate=dx[1])
        return gx, dcat[..., self._half:]

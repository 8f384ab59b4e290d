"""NStageBackbone and SPP on the HIP kernels.

Reference: NStageBackbone modules/detection_modules.py:34-102 (stem -> N stages -> context module, returns the layers
named in out_layers); SPP training/models/detection_models/csp_darknet53.py:136-157 (1x1 -> max-pool 5/9/13 stride 1 ->
concat -> 1x1).  The SPP concat is one NHWC buffer: cv1 and the three pooling kernels write their channel slices.
"""
from typing import List

import os

import torch

from .. import kernels as K
from ..common.factories import DetectionModulesFactory
from ..common.registry import register_detection_module
from .base_modules import BaseDetectionModule
from .conv_bn_act_block import Conv
from .layers import MaxPool, act_name


_FOLD_EXT = os.environ.get("SGX_BACKBONE_ADDEND", "1") != "0"  # measurement switch (r6s): 0 = the accumulate passes of rounds 1 - 5


@register_detection_module()
class SPP(BaseDetectionModule):
    def __init__(self, in_channels, output_channels, k, activation_type):
        super().__init__(in_channels)
        self._output_channels = output_channels
        hidden = in_channels // 2
        act = act_name(activation_type)
        self.hidden = hidden
        self.cv1 = Conv(in_channels, hidden, 1, 1, act)
        self.cv2 = Conv(hidden * (len(k) + 1), output_channels, 1, 1, act)
        self.m = torch.nn.ModuleList([MaxPool(kernel_size=x, stride=1, padding=x // 2) for x in k])

    @property
    def out_channels(self):
        return self._output_channels

    def fwd(self, x, out=None):
        n, h, w, _ = x.shape
        hid = self.hidden
        cat = torch.empty(n, h, w, hid * (len(self.m) + 1), device=x.device, dtype=x.dtype)
        y = self.cv1.fwd(x, out=cat[..., :hid])
        ks = [m.k for m in self.m]
        # inference: stride-1 max pools with -inf padding compose exactly (pool_a o pool_b = pool_{a+b-1}: every point between an output
        # and a source inside the image is inside the image), so 5 / 9 / 13 are three 5-wide passes - 75 window reads instead of 275
        chain = not self.training and all(m.s == 1 and 2 * m.p + 1 == m.k for m in self.m) and all(b == a + ks[0] - 1 for a, b in zip(ks, ks[1:]))
        for i, m in enumerate(self.m):
            dst = cat[..., (i + 1) * hid:(i + 2) * hid]
            if chain and i > 0:
                self.m[0].fwd(cat[..., i * hid:(i + 1) * hid], out=dst)
            else:
                m.fwd(y, out=dst)
        return self.cv2.fwd(cat, out=out)

    def bwd(self, dy, dx_out=None, accumulate=False, addend=None, need_dx=True):
        hid = self.hidden
        dcat = self.cv2.bwd(dy)
        g = dcat[..., :hid]  # gradient of cv1's output: its own slice + the three pooling backward passes, accumulated in place
        for i, m in enumerate(self.m):
            m.bwd(dcat[..., (i + 1) * hid:(i + 2) * hid], dx_out=g, accumulate=True)
        return self.cv1.bwd(g, dx_out=dx_out, accumulate=accumulate, addend=addend, need_dx=need_dx)


@register_detection_module()
class NStageBackbone(BaseDetectionModule):
    def __init__(self, in_channels: int, out_layers: List[str], stem, stages, context_module):
        super().__init__(in_channels)
        factory = DetectionModulesFactory()
        self.num_stages = len(stages)
        self.stem = factory.get(factory.insert_module_param(stem, "in_channels", in_channels))
        prev = self.stem.out_channels
        for i in range(self.num_stages):
            stage = factory.get(factory.insert_module_param(stages[i], "in_channels", prev))
            setattr(self, f"stage{i + 1}", stage)
            prev = stage.out_channels
        self.context_module = factory.get(factory.insert_module_param(context_module, "in_channels", prev)) if context_module is not None else None
        self.out_layers = list(out_layers)
        self._all_layers = ["stem"] + [f"stage{i}" for i in range(1, self.num_stages + 1)] + (["context_module"] if self.context_module is not None else [])
        self._out_channels = [getattr(self, layer).out_channels for layer in self.out_layers]

    @property
    def out_channels(self):
        return self._out_channels

    def get_input_channels(self) -> int:
        return self.stem.get_input_channels()

    def replace_input_channels(self, in_channels: int, compute_new_weights_fn=None):
        """Reference csp_darknet53.py:235-241 / detection_modules: delegate to the first block."""
        if not hasattr(self.stem, "replace_input_channels"):
            raise NotImplementedError(f"`{type(self.stem).__name__}` does not support `replace_input_channels`")
        self.stem.replace_input_channels(in_channels=in_channels, compute_new_weights_fn=compute_new_weights_fn)
        self.in_channels = in_channels

    def fwd(self, x, out=None, on_output=None):
        """on_output(i, tensor): called as soon as output i exists (a neck that starts work on it beside the deeper stages)"""
        outs = []
        for layer in self._all_layers:
            x = getattr(self, layer).fwd(x)
            if layer in self.out_layers:
                outs.append(x)
                if on_output is not None:
                    on_output(len(outs) - 1, x)
        return outs

    def bwd(self, grads: dict, on_layer_done=None, ext_ready=None):
        """grads: {layer_name: gradient of that layer's output coming from outside the backbone (the neck)} for the
        layers in out_layers.  Walks the chain backwards, adding each external gradient where its tensor was produced."""
        # (Round 6: the neck's gradient of a layer's output is ADDED IN THE EPILOGUE of the data-gradient launch that produces the backbone's
        # own gradient of that output - the next layer's `addend` - instead of an accumulate pass behind it: three passes over the 160 x 160,
        # 80 x 80 and 40 x 40 feature-map gradients per step became one read.  An external gradient that is not a dense tensor - a slice of
        # a concat gradient - keeps the pass.)
        # ext_ready: called once, before the first external gradient other than the deepest layer's is read (the neck may still be producing
        # those on its branch stream while this walk starts: the deepest one is the main chain's)
        order = list(reversed(self._all_layers))
        g, folded = None, set()
        for i, layer in enumerate(order):
            ext = grads.get(layer)
            if g is None:
                g = ext
            elif ext is not None and layer not in folded:
                if ext_ready is not None:
                    ext_ready, _ = None, ext_ready()
                K.axpy(ext, out=g, accumulate=True)
            if g is None:
                continue
            kw = {}
            nxt = order[i + 1] if i + 1 < len(order) else None
            add = grads.get(nxt) if nxt is not None else None
            if add is not None and ext_ready is not None:
                ext_ready, _ = None, ext_ready()
            if add is not None and add.is_contiguous() and add.dtype == torch.float32 and _FOLD_EXT:
                kw["addend"] = add
                folded.add(nxt)
            g = getattr(self, layer).bwd(g, need_dx=layer != self._all_layers[0], **kw)
            if on_layer_done is not None:
                on_layer_done(layer)
        if ext_ready is not None:
            ext_ready()
        return g

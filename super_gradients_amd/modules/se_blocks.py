"""EffectiveSEBlock on the HIP kernels (reference: modules/se_blocks.py:29-42):  y = x * hardsigmoid(project(mean_hw(x))).

Kernels: one deterministic per-image column reduction for the mean ([N,C]), the 1x1 `project` convolution on the [N,1,1,C] means
through the ordinary conv kernels, one gate sweep.  Backward: d(pre) = hardsigmoid'(pre) * sum_hw(dy * x) (one reduction), the
project convolution's weight / data gradients, then dx = dy * gate + d(mean)/HW in one sweep.
"""
from .. import kernels as K
from .engine import SgxBlock
from .layers import ConvLayer


class EffectiveSEBlock(SgxBlock):
    GATE = "hardsigmoid"

    def __init__(self, in_channels: int):
        super().__init__()
        self.project = ConvLayer(in_channels, in_channels, 1, 1, 0, bias=True)

    def on_materialize(self):
        pass

    def fwd(self, x, out=None):
        n, h, w, c = x.shape
        mean = K.image_colsum(x, scale=1.0 / (h * w)).view(n, 1, 1, c)
        pre = self.project.conv(mean).view(n, c)
        self._ctx = (x, mean, pre) if self.training else None
        return K.channel_gate(x, pre, self.GATE, out=out)

    def bwd(self, dy, dx_out=None, accumulate=False, addend=None, need_dx=True):
        (x, mean, pre), self._ctx = self._ctx, None
        n, h, w, c = x.shape
        dpre = K.image_colsum(dy, v=x, pre=pre, gate=self.GATE).view(n, 1, 1, c)
        self.project.wgrad(mean, dpre)
        dmean = self.project.dgrad(dpre, (n, 1, 1, c)).view(n, c)
        if addend is not None:
            raise NotImplementedError("EffectiveSEBlock.bwd: no addend")
        if dx_out is None:
            dx_out, accumulate = dy, False  # in place over the incoming gradient (element-wise: each value is read before it is written)
        return K.channel_gate(dy, pre, self.GATE, bias=dmean, bias_scale=1.0 / (h * w), out=dx_out, accumulate=accumulate)

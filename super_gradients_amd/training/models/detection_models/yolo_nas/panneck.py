"""YoloNASPANNeckWithC2 on the HIP kernels (reference: yolo_nas/panneck.py:12-64): two up stages that also take a
higher-resolution skip, two down stages; returns (p3, p4, p5).  The backward routes the gradients of the four backbone
tensors back out, summing where a tensor has several consumers (c3: neck1 skip2 + neck2 skip1)."""
import os
from typing import List

from .....common.factories import DetectionModulesFactory
from .....common.registry import register_detection_module
from .....modules.base_modules import BaseDetectionModule


@register_detection_module("YoloNASPANNeckWithC2")
class YoloNASPANNeckWithC2(BaseDetectionModule):
    def __init__(self, in_channels: List[int], neck1, neck2, neck3, neck4):
        super().__init__(in_channels)
        c2, c3, c4, c5 = in_channels
        f = DetectionModulesFactory()
        self.neck1 = f.get(f.insert_module_param(neck1, "in_channels", [c5, c4, c3]))
        self.neck2 = f.get(f.insert_module_param(neck2, "in_channels", [self.neck1.out_channels[1], c3, c2]))
        self.neck3 = f.get(f.insert_module_param(neck3, "in_channels", [self.neck2.out_channels[1], self.neck2.out_channels[0]]))
        self.neck4 = f.get(f.insert_module_param(neck4, "in_channels", [self.neck3.out_channels, self.neck1.out_channels[0]]))
        self._out_channels = [self.neck2.out_channels[1], self.neck3.out_channels, self.neck4.out_channels]

    @property
    def out_channels(self):
        return self._out_channels

    def pre(self, i: int, t):
        """Backbone output i (0: c2 ... 3: c5) exists: start the up stages' skip branches that read it (YoloNASUpStage.pre_skip)."""
        if i == 0:  # a new forward: whatever an aborted one left started is not this one's
            for stage in (self.neck1, self.neck2):
                if hasattr(stage, "pre_skip"):
                    stage._pre = {}
        for stage, which in {0: ((self.neck2, 2),), 1: ((self.neck1, 2), (self.neck2, 1)), 2: ((self.neck1, 1),)}.get(i, ()):
            if hasattr(stage, "pre_skip"):
                stage.pre_skip(which, t)

    def fwd(self, inputs, out=None):
        c2, c3, c4, c5 = inputs
        i1, x = self.neck1.fwd([c5, c4, c3])
        i2, p3 = self.neck2.fwd([x, c3, c2])
        p4 = self.neck3.fwd([p3, i2])
        p5 = self.neck4.fwd([p4, i1])
        return p3, p4, p5

    def bwd(self, dp3, dp4, dp5, late_join=False):
        """dp3/dp4/dp5: gradients from the heads (owned buffers: accumulated into).  -> (dc2, dc3, dc4, dc5)"""
        g_p4, d_i1 = self.neck4.bwd(dp5, dx=(dp4, True))          # p4 feeds head2 and neck4
        g_p3, d_i2 = self.neck3.bwd(g_p4, dx=(dp3, True))          # p3 feeds head1 and neck3
        # (the up stages' skip-branch gradients dc2 / dc3 / dc4 may still be in flight on the branch stream: join_bwd() before they are read -
        # both stages fork onto the same in-order stream, so neck1's accumulation into dc3 is ordered behind neck2's write of it)
        late = late_join and os.environ.get("SGX_BRANCH_LATE_JOIN", "1") != "0"
        g_x, dc3, dc2 = self.neck2.bwd(d_i2, g_p3, dx=(None, False), ds1=(None, False), ds2=(None, False), late_join=late)
        dc5, dc4, dc3 = self.neck1.bwd(d_i1, g_x, dx=(None, False), ds1=(None, False), ds2=(dc3, True), late_join=late)
        return dc2, dc3, dc4, dc5

    def join_bwd(self):
        self.neck2.join_bwd()
        self.neck1.join_bwd()

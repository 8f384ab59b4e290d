from . import detection_modules  # noqa: F401  (registers NStageBackbone, SPP)
from .conv_bn_act_block import Conv, ConvBNAct, ConvBNReLU  # noqa: F401
from .qarepvgg_block import QARepVGGBlock  # noqa: F401

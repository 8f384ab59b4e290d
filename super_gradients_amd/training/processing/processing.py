"""Image processing for predict(): the reference's Processing classes (training/processing/processing.py) with the same names, constructor
arguments, metadata and box post-processing - but the image side runs on the device.

The reference applies each Processing to each image on the host, one numpy / cv2 pass per stage (ComposeProcessing.preprocess_image,
processing.py:143-149).  Here every Processing only *describes* its stage: a ComposeProcessing folds the stages of a whole batch into one
geometry table + one set of photometric parameters, uploads the raw uint8 images, and ONE kernel launch (sgx_preprocess_u8_hwc,
csrc/image.hip) writes the standardized fp32 NHWC batch the first convolution reads.  Stage order the fused launch covers (each optional):

    ReverseImageChannels -> Detection[LongestMaxSize]Rescale -> Detection{Center,BottomRight,Auto}Padding -> StandardizeImage
                         -> NormalizeImage -> ImagePermute((2, 0, 1))

which contains every detection pipeline the reference defines (processing.py:913-980: default_yolox / default_ppyoloe / default_yolo_nas_coco_processing_params).
Any other order raises NotImplementedError - there is no host fallback.  postprocess_predictions works on the few [N, 4] boxes a
prediction holds, on the host, with the reference's arithmetic (transforms/utils.py:44-57,161-172).
"""
from abc import ABC, abstractmethod
from dataclasses import dataclass
from typing import List, Optional, Tuple, Union

import numpy as np
import torch

from ... import kernels as K
from ...common.registry import register_processing
from ..utils.predict import DetectionPrediction, Prediction


@dataclass
class PaddingCoordinates:
    top: int
    bottom: int
    left: int
    right: int


@dataclass
class ProcessingMetadata(ABC):
    pass


@dataclass
class ComposeProcessingMetadata(ProcessingMetadata):
    metadata_lst: List[Union[None, ProcessingMetadata]]


@dataclass
class DetectionPadToSizeMetadata(ProcessingMetadata):
    padding_coordinates: PaddingCoordinates


@dataclass
class RescaleMetadata(ProcessingMetadata):
    original_shape: Tuple[int, int]
    scale_factor_h: float
    scale_factor_w: float


_STAGES = ("reverse", "rescale", "pad", "standardize", "normalize", "permute")


class _ImagePlan:
    """What the fused launch needs for one image, accumulated stage by stage."""

    def __init__(self, shape):
        if len(shape) != 3 or not 1 <= shape[2] <= 4:
            raise ValueError(f"predict() images are [H, W, C] with 1..4 channels, got shape {tuple(shape)}")
        self.channels = int(shape[2])
        self.h, self.w = int(shape[0]), int(shape[1])  # size after the rescale stage
        self.top = self.left = 0
        self.H = self.W = None                          # slot size once a padding stage ran
        self.pad_value = 0
        self.reverse = False
        self.max_value = None
        self.mean = self.std = None
        self.permutation = None
        self._stage = -1

    def enter(self, stage: str, who) -> bool:
        """-> True when the stage repeats the previous one (only a rescale may: dataset-derived lists hold e.g. DetectionLongestMaxSize
        followed by DetectionPaddedRescale's own rescale, preprocessing_unit_test.py:119-123 - the second one is then a no-op)."""
        i = _STAGES.index(stage)
        if stage == "reverse" and not self.reverse and self._stage == _STAGES.index("pad"):
            # Channel reversal commutes with padding (the reference's skip_image_resizing compose puts its auto-padding FIRST, processing.py:
            # 186-202, and dataset-derived / PP-YOLOE pipelines start with ReverseImageChannels): reverse-then-pad with the per-channel pad
            # value reversed is the same image.  The stage pointer stays at 'pad'.
            pv = np.asarray(self.pad_value)
            if pv.ndim:
                self.pad_value = tuple(pv.reshape(-1)[::-1].tolist())
            return False
        if i < self._stage or (i == self._stage and stage != "rescale"):
            raise NotImplementedError(f"{type(who).__name__} after '{_STAGES[self._stage]}': the fused device pre-processing covers the stage order "
                                      f"{' -> '.join(_STAGES)} (each at most once; a repeated rescale only when it leaves the size unchanged)")
        repeat = i == self._stage
        self._stage = i
        return repeat

    def resize_to(self, h: int, w: int, repeat: bool, who):
        if repeat and (h, w) != (self.h, self.w):
            raise NotImplementedError(f"{type(who).__name__}: a second rescale that changes the size again ({self.h}x{self.w} -> {h}x{w}) would be "
                                      "two successive resamplings; the fused device pre-processing resamples once")
        self.h, self.w = int(h), int(w)

    @property
    def out_hw(self):
        return (self.H, self.W) if self.H is not None else (self.h, self.w)

    def photometric_key(self):
        pv = tuple(np.broadcast_to(np.asarray(self.pad_value), (self.channels,)).tolist())
        return (self.channels, self.out_hw, pv, self.reverse, self.max_value, None if self.mean is None else tuple(self.mean),
                None if self.std is None else tuple(self.std), self.permutation)


class Processing(ABC):
    """Reference interface (processing.py:68-111): preprocess_image(image) -> (image, metadata); postprocess_predictions(pred, metadata)."""

    @abstractmethod
    def _describe(self, plan: _ImagePlan) -> Union[None, ProcessingMetadata]:
        """add this stage to the image's plan, return the stage's metadata"""

    def preprocess_image(self, image: np.ndarray):
        return ComposeProcessing([self]).preprocess_image(image)

    @abstractmethod
    def postprocess_predictions(self, predictions: Prediction, metadata: Union[None, ProcessingMetadata]) -> Prediction:
        pass

    def inverse_box_steps(self, metadata: Union[None, ProcessingMetadata]):
        """What `postprocess_predictions` does to detection boxes, as data: a list of (kind, a_x, a_y) steps - kind 0: x += a_x, y += a_y;
        kind 1: x *= a_x, y *= a_y - applied in order, one float32 rounding each (exactly the numpy float32 passes of the stage), so that
        DetectionPipeline can run the whole batch's inverse maps in one device launch (kernels.detection_unmap).  None: this stage has no
        such description (a user-defined Processing) - the pipeline then maps boxes on the host through postprocess_predictions."""
        return None

    def infer_image_input_shape(self) -> Optional[Tuple[int, int]]:
        return None

    @property
    def resizes_image(self) -> bool:
        return False

    def to_config(self):
        """`{TypeName: {constructor kwargs}}` of plain values: what ProcessingFactory.get() rebuilds this stage from, and the form the Trainer
        stores in checkpoints (loadable with torch.load(weights_only=True); the reference pickles the objects, sg_trainer.py:710-712)."""
        return {type(self).__name__: {k: (list(v) if isinstance(v, tuple) else v) for k, v in self._config_kwargs().items()}}

    def _config_kwargs(self) -> dict:
        return {}


class AutoPadding(Processing, ABC):
    def __init__(self, shape_multiple: Tuple[int, int], pad_value: int):
        if isinstance(shape_multiple, int):
            shape_multiple = (shape_multiple, shape_multiple)
        self.shape_multiple = tuple(shape_multiple)
        self.pad_value = pad_value

    def _config_kwargs(self):
        return dict(shape_multiple=self.shape_multiple, pad_value=self.pad_value)

    def _get_padding_params(self, input_shape: Tuple[int, int]) -> PaddingCoordinates:
        h, w = input_shape
        mh, mw = self.shape_multiple
        return PaddingCoordinates(top=0, left=0, bottom=(h + mh - 1) // mh * mh - h, right=(w + mw - 1) // mw * mw - w)


def _to_device_u8(image, device):
    if isinstance(image, torch.Tensor):
        t = image
    else:
        t = torch.from_numpy(np.ascontiguousarray(image))
    if t.dtype != torch.uint8:
        raise ValueError(f"predict() takes uint8 images (the reference's load_images contract), got {t.dtype}")
    return t.to(device)


@register_processing("ComposeProcessing")
class ComposeProcessing(Processing):
    def __init__(self, processings: List[Processing]):
        self.processings = list(processings)

    def to_config(self):
        return {"ComposeProcessing": {"processings": [p.to_config() for p in self.processings]}}

    def _flat(self):
        for p in self.processings:
            if isinstance(p, ComposeProcessing):
                yield from p._flat()
            else:
                yield p

    def _describe(self, plan):
        return ComposeProcessingMetadata([p._describe(plan) for p in self.processings])

    def plan_image(self, shape):
        plan = _ImagePlan(shape)
        return plan, self._describe(plan)

    def preprocess_batch(self, images, device=None):
        """images: uint8 [h, w, C] arrays / tensors -> (fp32 batch as a logical NCHW view of the NHWC buffer - or NHWC when the compose holds
        no ImagePermute -, [metadata per image]).  One launch for the whole batch."""
        if len(images) == 0:
            raise ValueError("preprocess_batch needs at least one image")
        device = torch.device(device) if device is not None else next((i.device for i in images if isinstance(i, torch.Tensor)), None)
        if device is None:
            raise ValueError("preprocess_batch: pass the device the model lives on")
        plans, metas = zip(*(self.plan_image(tuple(i.shape)) for i in images))
        key = plans[0].photometric_key()
        for img, p in zip(images, plans):
            if p.photometric_key() != key:  # pipelines.py:203-209
                raise ValueError(f"Images have different shapes ({p.out_hw} != {plans[0].out_hw})!\nEither resize the images to the same size, "
                                 "set `skip_image_resizing=False` or pass one image at a time.")
        p0 = plans[0]
        if p0.permutation not in (None, (2, 0, 1)):
            raise NotImplementedError("ImagePermute on the device path: (2, 0, 1) (HWC -> CHW), which is a view of the NHWC batch")
        H, W = p0.out_hw
        dev_imgs = [_to_device_u8(i, device) for i in images]
        c = p0.channels
        pv = np.broadcast_to(np.asarray(p0.pad_value), (c,))
        if np.any(pv != np.round(pv)) or pv.min() < 0 or pv.max() > 255:
            raise ValueError(f"pad_value {p0.pad_value!r}: the padding is applied to the uint8 image (integers 0..255)")
        pad = torch.tensor(pv.astype(np.uint8), dtype=torch.uint8).to(device)
        mean = None if p0.mean is None else torch.tensor(p0.mean, dtype=torch.float32).to(device)
        std = None if p0.std is None else torch.tensor(p0.std, dtype=torch.float32).to(device)
        y = K.preprocess_u8(dev_imgs, [(p.h, p.w, p.top, p.left) for p in plans], H, W, pad, reverse_channels=p0.reverse, max_value=p0.max_value,
                            mean=mean, std=std)
        out = K.nhwc_as_nchw_view(y, c) if p0.permutation is not None else y[..., :c]
        return out, list(metas)

    def preprocess_image(self, image):
        """Reference signature: one image in, (numpy image, metadata) out - computed by the same device launch as a batch of one."""
        device = image.device if isinstance(image, torch.Tensor) and image.is_cuda else torch.device("cuda" if torch.cuda.is_available() else "cpu")
        out, metas = self.preprocess_batch([image], device=device)  # (a CPU device only works under the tests' host emulation of the kernels)
        plan, _ = self.plan_image(tuple(image.shape))
        arr = out[0].contiguous().cpu().numpy()
        if plan.max_value is None and plan.mean is None:
            arr = arr.astype(np.uint8)  # no photometric stage: the reference's image is still uint8
        return arr, metas[0]

    def postprocess_predictions(self, predictions, metadata: ComposeProcessingMetadata):
        for p, m in zip(self.processings[::-1], metadata.metadata_lst[::-1]):
            predictions = p.postprocess_predictions(predictions, m)
        return predictions

    def inverse_box_steps(self, metadata: ComposeProcessingMetadata):
        steps = []
        for p, m in zip(self.processings[::-1], metadata.metadata_lst[::-1]):
            st = p.inverse_box_steps(m)
            if st is None:
                return None
            steps.extend(st)
        return steps

    def infer_image_input_shape(self):
        shape = None
        for p in self.processings:
            s = p.infer_image_input_shape()
            shape = s if s is not None else shape
        return shape

    @property
    def resizes_image(self) -> bool:
        return any(p.resizes_image for p in self.processings)

    def get_equivalent_compose_without_resizing(self, auto_padding: AutoPadding) -> "ComposeProcessing":
        """processing.py:186-202: drop every stage that resizes, pad to the model's shape multiple first instead."""
        out = [auto_padding]
        for p in self.processings:
            if isinstance(p, ComposeProcessing):
                out.append(p.get_equivalent_compose_without_resizing(auto_padding))
            elif not p.resizes_image:
                out.append(p)
        return ComposeProcessing(out)


@register_processing("ImagePermute")
class ImagePermute(Processing):
    def __init__(self, permutation: Tuple[int, int, int] = (2, 0, 1)):
        self.permutation = tuple(permutation)

    def _config_kwargs(self):
        return dict(permutation=self.permutation)

    def _describe(self, plan):
        plan.enter("permute", self)
        plan.permutation = self.permutation
        return None

    def postprocess_predictions(self, predictions, metadata):
        return predictions

    def inverse_box_steps(self, metadata):
        return []


@register_processing("ReverseImageChannels")
class ReverseImageChannels(Processing):
    def _describe(self, plan):
        plan.enter("reverse", self)
        plan.reverse = True
        return None

    def postprocess_predictions(self, predictions, metadata):
        return predictions

    def inverse_box_steps(self, metadata):
        return []


@register_processing("StandardizeImage")
class StandardizeImage(Processing):
    def __init__(self, max_value: float = 255.0):
        self.max_value = float(max_value)

    def _config_kwargs(self):
        return dict(max_value=self.max_value)

    def _describe(self, plan):
        plan.enter("standardize", self)
        plan.max_value = self.max_value
        return None

    def postprocess_predictions(self, predictions, metadata):
        return predictions

    def inverse_box_steps(self, metadata):
        return []


@register_processing("NormalizeImage")
class NormalizeImage(Processing):
    def __init__(self, mean: List[float], std: List[float]):
        self.mean = np.array(mean).reshape((1, 1, -1)).astype(np.float32)
        self.std = np.array(std).reshape((1, 1, -1)).astype(np.float32)
        self._mean_arg, self._std_arg = [float(v) for v in np.ravel(mean)], [float(v) for v in np.ravel(std)]

    def _config_kwargs(self):
        return dict(mean=self._mean_arg, std=self._std_arg)

    def _describe(self, plan):
        plan.enter("normalize", self)
        if self.mean.size != plan.channels or self.std.size != plan.channels:
            raise ValueError(f"NormalizeImage has {self.mean.size} mean / {self.std.size} std values for a {plan.channels}-channel image")
        plan.mean, plan.std = self.mean.reshape(-1).tolist(), self.std.reshape(-1).tolist()
        return None

    def postprocess_predictions(self, predictions, metadata):
        return predictions

    def inverse_box_steps(self, metadata):
        return []


def _shift_bboxes_xyxy(boxes: np.ndarray, shift_w: float, shift_h: float) -> np.ndarray:
    out = boxes.copy()
    out[:, [0, 2]] += shift_w
    out[:, [1, 3]] += shift_h
    return out


def _rescale_bboxes(boxes: np.ndarray, scale_factors: Tuple[float, float]) -> np.ndarray:
    out = boxes.astype(np.float32, copy=True)
    sy, sx = scale_factors
    out[:, :4] *= np.array([[sx, sy, sx, sy]], dtype=out.dtype)
    return out


class _DetectionPadding(Processing, ABC):
    """Pads to output_shape; the image must not be larger (processing.py:326-370)."""

    def __init__(self, output_shape: Tuple[int, int], pad_value: int):
        self.output_shape = tuple(output_shape)
        self.pad_value = pad_value

    def _config_kwargs(self):
        return dict(output_shape=self.output_shape, pad_value=self.pad_value)

    @abstractmethod
    def _get_padding_params(self, input_shape) -> PaddingCoordinates:
        pass

    def _describe(self, plan):
        plan.enter("pad", self)
        c = self._get_padding_params((plan.h, plan.w))
        if min(c.top, c.bottom, c.left, c.right) < 0:
            raise ValueError(f"{type(self).__name__}: a {plan.h}x{plan.w} image does not fit output_shape {self.output_shape}")
        plan.top, plan.left, plan.H, plan.W, plan.pad_value = c.top, c.left, plan.h + c.top + c.bottom, plan.w + c.left + c.right, self.pad_value
        return DetectionPadToSizeMetadata(padding_coordinates=c)

    def postprocess_predictions(self, predictions: DetectionPrediction, metadata: DetectionPadToSizeMetadata):
        c = metadata.padding_coordinates
        predictions.bboxes_xyxy = _shift_bboxes_xyxy(predictions.bboxes_xyxy, shift_w=-c.left, shift_h=-c.top)
        return predictions

    def inverse_box_steps(self, metadata: DetectionPadToSizeMetadata):
        c = metadata.padding_coordinates
        return [(0.0, float(-c.left), float(-c.top))]

    def infer_image_input_shape(self):
        return self.output_shape

    @property
    def resizes_image(self) -> bool:
        return True


@register_processing("DetectionCenterPadding")
class DetectionCenterPadding(_DetectionPadding):
    def _get_padding_params(self, input_shape):
        ph, pw = self.output_shape[0] - input_shape[0], self.output_shape[1] - input_shape[1]
        return PaddingCoordinates(top=ph // 2, bottom=ph - ph // 2, left=pw // 2, right=pw - pw // 2)


@register_processing("DetectionBottomRightPadding")
class DetectionBottomRightPadding(_DetectionPadding):
    def _get_padding_params(self, input_shape):
        return PaddingCoordinates(top=0, bottom=self.output_shape[0] - input_shape[0], left=0, right=self.output_shape[1] - input_shape[1])


@register_processing("DetectionAutoPadding")
class DetectionAutoPadding(AutoPadding):
    def _describe(self, plan):
        plan.enter("pad", self)
        c = self._get_padding_params((plan.h, plan.w))
        plan.top, plan.left, plan.H, plan.W, plan.pad_value = 0, 0, plan.h + c.bottom, plan.w + c.right, self.pad_value
        return DetectionPadToSizeMetadata(padding_coordinates=c)

    def postprocess_predictions(self, predictions: DetectionPrediction, metadata: DetectionPadToSizeMetadata):
        c = metadata.padding_coordinates
        predictions.bboxes_xyxy = _shift_bboxes_xyxy(predictions.bboxes_xyxy, shift_w=-c.left, shift_h=-c.top)
        return predictions

    def inverse_box_steps(self, metadata: DetectionPadToSizeMetadata):
        c = metadata.padding_coordinates
        return [(0.0, float(-c.left), float(-c.top))]


class _DetectionRescaleBase(Processing, ABC):
    def __init__(self, output_shape: Tuple[int, int]):
        self.output_shape = tuple(output_shape)

    def _config_kwargs(self):
        return dict(output_shape=self.output_shape)

    def postprocess_predictions(self, predictions: DetectionPrediction, metadata: RescaleMetadata):
        predictions.bboxes_xyxy = _rescale_bboxes(predictions.bboxes_xyxy, (1 / metadata.scale_factor_h, 1 / metadata.scale_factor_w))
        return predictions

    def inverse_box_steps(self, metadata: RescaleMetadata):
        return [(1.0, 1 / metadata.scale_factor_w, 1 / metadata.scale_factor_h)]

    @property
    def resizes_image(self) -> bool:
        return True


@register_processing("DetectionRescale")
class DetectionRescale(_DetectionRescaleBase):
    """To output_shape without keeping the aspect ratio (processing.py:510-538)."""

    def _describe(self, plan):
        repeat = plan.enter("rescale", self)
        h0, w0 = plan.h, plan.w
        plan.resize_to(self.output_shape[0], self.output_shape[1], repeat, self)
        return RescaleMetadata(original_shape=(h0, w0), scale_factor_h=self.output_shape[0] / h0, scale_factor_w=self.output_shape[1] / w0)

    def infer_image_input_shape(self):
        return self.output_shape


@register_processing("DetectionLongestMaxSizeRescale")
class DetectionLongestMaxSizeRescale(_DetectionRescaleBase):
    """Longest side to output_shape, aspect ratio kept (processing.py:541-575): scale = min(H / h, W / w), new size = round(h * scale), ..."""

    def _describe(self, plan):
        repeat = plan.enter("rescale", self)
        h0, w0 = plan.h, plan.w
        s = min(self.output_shape[0] / h0, self.output_shape[1] / w0)
        if s != 1.0:
            plan.resize_to(round(h0 * s), round(w0 * s), repeat, self)
        return RescaleMetadata(original_shape=(h0, w0), scale_factor_h=s, scale_factor_w=s)


COCO_DETECTION_CLASSES_LIST = [
    "person", "bicycle", "car", "motorcycle", "airplane", "bus", "train", "truck", "boat", "traffic light", "fire hydrant", "stop sign",
    "parking meter", "bench", "bird", "cat", "dog", "horse", "sheep", "cow", "elephant", "bear", "zebra", "giraffe", "backpack", "umbrella",
    "handbag", "tie", "suitcase", "frisbee", "skis", "snowboard", "sports ball", "kite", "baseball bat", "baseball glove", "skateboard",
    "surfboard", "tennis racket", "bottle", "wine glass", "cup", "fork", "knife", "spoon", "bowl", "banana", "apple", "sandwich", "orange",
    "broccoli", "carrot", "hot dog", "pizza", "donut", "cake", "chair", "couch", "potted plant", "bed", "dining table", "toilet", "tv",
    "laptop", "mouse", "remote", "keyboard", "cell phone", "microwave", "oven", "toaster", "sink", "refrigerator", "book", "clock", "vase",
    "scissors", "teddy bear", "hair drier", "toothbrush",
]


def default_yolo_nas_coco_processing_params() -> dict:
    """processing.py:960-980"""
    image_processor = ComposeProcessing([
        DetectionLongestMaxSizeRescale(output_shape=(636, 636)),
        DetectionCenterPadding(output_shape=(640, 640), pad_value=114),
        StandardizeImage(max_value=255.0),
        ImagePermute(permutation=(2, 0, 1)),
    ])
    return dict(class_names=COCO_DETECTION_CLASSES_LIST, image_processor=image_processor, iou=0.7, conf=0.25)


def default_ppyoloe_coco_processing_params() -> dict:
    """processing.py:935-957"""
    image_processor = ComposeProcessing([
        ReverseImageChannels(),
        DetectionRescale(output_shape=(640, 640)),
        NormalizeImage(mean=[123.675, 116.28, 103.53], std=[58.395, 57.12, 57.375]),
        ImagePermute(permutation=(2, 0, 1)),
    ])
    return dict(class_names=COCO_DETECTION_CLASSES_LIST, image_processor=image_processor, iou=0.65, conf=0.5)

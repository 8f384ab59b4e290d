"""predict(): images -> device pre-processing -> fused eval forward -> NMS -> boxes in the original image frames.

Reference: training/pipelines/pipelines.py:47-283 (Pipeline) and :285-370 (DetectionPipeline).  Same constructor arguments, call
protocol (`pipeline(images, batch_size)`) and result containers; what differs is where the work runs:

  reference (per batch)                                    here
  image_processor.preprocess_image per image, numpy/cv2    ONE launch for the batch (ComposeProcessing.preprocess_batch, csrc/image.hip)
  np.array(list) -> torch.from_numpy -> .to(device)        raw uint8 images are uploaded, the fp32 batch is born in HBM
  deepcopy + prep_model_for_conversion on first batch      same, full fusion (QARepVGG / RepVGG blocks and every conv+BN pair become ONE conv launch
                                                           with bias + activation in its epilogue)
  autocast(fp16) forward                                   fp16=True (the reference's default): the fused copy runs on the half-precision kernels
                                                           (csrc/half.hip: bf16 activations / filters, fp32 accumulation on the bf16 matrix
                                                           pipe, fp32 prediction outputs) for the architectures that have them (YOLO-NAS);
                                                           fp16=False, or an architecture without them: the fp32 path
  post_prediction_callback (python loop + torchvision)     the batched NMS kernels (csrc/nms.hip)
  postprocess_predictions per image (numpy)                same arithmetic, on the <= max_predictions boxes of each image
Video / webcam sources are cv2 I/O and outside the path.
"""
import copy
import os
from abc import ABC, abstractmethod
from typing import Iterable, List, Optional, Union

import numpy as np
import torch

from ..processing.processing import ComposeProcessing, ImagePermute, Processing
from ..utils.predict import DetectionPrediction, ImageDetectionPrediction, ImagesDetectionPrediction

IMG_EXTENSIONS = ("bmp", "dng", "jpeg", "jpg", "mpo", "pfm", "pgm", "png", "ppm", "tif", "tiff", "webp")
VIDEO_EXTENSIONS = (".mp4", ".avi", ".mov", ".wmv", ".flv", ".gif")


def _load_image(image) -> Union[np.ndarray, torch.Tensor]:
    """utils/media/image.py:107-141: arrays / tensors as they are, PIL images and image files as RGB uint8 arrays."""
    if isinstance(image, (np.ndarray, torch.Tensor)):
        return image
    try:
        from PIL import Image
    except ImportError:  # pragma: no cover
        Image = None
    if Image is not None and isinstance(image, Image.Image):
        return np.asarray(image.convert("RGB"))
    if isinstance(image, str):
        if image.startswith(("http://", "https://")):
            raise ValueError("predict(): URL sources need network access; pass a local file, an array or a tensor")
        if Image is None:  # pragma: no cover
            raise ValueError("predict(): reading image files needs PIL")
        return np.asarray(Image.open(image).convert("RGB"))
    raise ValueError(f"Input {type(image)} not supported for prediction.")


def load_images(images) -> List[Union[np.ndarray, torch.Tensor]]:
    """utils/media/image.py:19-68: one image, a list / iterator of images, a 4-D stack, or a folder of image files."""
    if isinstance(images, str) and os.path.isdir(images):
        names = sorted(n for n in os.listdir(images) if n.lower().rsplit(".", 1)[-1] in IMG_EXTENSIONS)
        return [_load_image(os.path.join(images, n)) for n in names]
    if isinstance(images, (np.ndarray, torch.Tensor)) and images.ndim == 4:
        return [images[i] for i in range(images.shape[0])]
    if isinstance(images, (list, tuple)) or hasattr(images, "__next__"):
        return [_load_image(i) for i in images]
    return [_load_image(images)]


_FP16_NOTED = set()


def _note_fp16_once(model, why="has no half-precision kernels on this build"):
    """fp16=True is the reference's default (pipelines.py:76: torch.autocast around the forward).  Architectures without half-precision
    kernels here (everything but YOLO-NAS) run the forward in fp32 - results are at least as precise, throughput is the fp32 path's.
    Said LOUDLY (a warnings.warn UserWarning and a log record) once per model class: the fallback must not pass for the requested mode."""
    name = type(model).__name__
    if name not in _FP16_NOTED:
        _FP16_NOTED.add(name)
        import logging
        import warnings

        msg = f"predict(fp16=True): {name} {why} - the forward runs in fp32 (pass fp16=False to silence this note)"
        logging.getLogger(__name__).warning(msg)
        warnings.warn(msg, UserWarning, stacklevel=3)


class Pipeline(ABC):
    def __init__(self, model, image_processor: Union[Processing, List[Processing]], class_names: List[str], device: Optional[str] = None,
                 fuse_model: bool = True, dtype: Optional[torch.dtype] = None, fp16: bool = True):
        self.model = model
        if getattr(model, "_materialized", False):
            self.device = model._device  # a materialised model lives in its HBM arenas and cannot move (the reference moves the model here)
            if device is not None and torch.device(device).type != self.device.type:
                raise RuntimeError(f"the model is materialised on {self.device}; predict(device={device!r}) cannot move it")
        else:
            from ... import _lib

            self.device = torch.device(device) if device is not None else torch.device("cpu" if _lib._TEST_HOST_MODE else "cuda")
            if hasattr(model, "materialize"):
                model.materialize(self.device)  # before the fused copy is taken: fusion reads the arena views
        self.dtype = dtype or torch.float32
        self.class_names = class_names
        if isinstance(image_processor, list):
            image_processor = ComposeProcessing(image_processor)
        self.image_processor = image_processor
        self.fuse_model = fuse_model  # fused on the first batch, like the reference (pipelines.py:91,95-100)
        self.fp16 = fp16
        # the forward's compute type: bf16 on the fused copy of an architecture that has the kernels, fp32 otherwise (an unfused model has
        # BatchNorm / two-branch blocks the half path does not carry)
        self.half = bool(fp16 and fuse_model and getattr(model, "supports_half_inference", lambda: False)())
        if fp16 and not self.half:
            _note_fp16_once(model)

    def _fuse_model(self, input_size):
        cache, self.model._pipeline_cache = getattr(self.model, "_pipeline_cache", None), None  # (it holds this pipeline: not part of the copy)
        try:
            fused = copy.deepcopy(self.model)
        finally:
            self.model._pipeline_cache = cache
        self.model = fused
        self.model.eval()
        # the copy is private to this pipeline and inference-only, so it takes the deepest form every block offers (the reference's call
        # leaves QARepVGG blocks partially fused - post-BN as a separate op - because its copy stays trainable): same function, fewer passes
        self.model.prep_model_for_conversion(input_size=input_size, full_fusion=True)
        if self.half:
            #  a conv + BatchNorm pair that could not be folded - a channel-padded filter, C % 4 != 0 - has no half-precision form:
            # fall back to the fp32 path for the whole model, loudly, instead of raising at the first forward
            unfolded = [n for n, m in self.model.named_modules() if hasattr(m, "_folded") and hasattr(m, "_parts") and m._folded is None and getattr(m, "_folded_half", None) is None]
            if unfolded:
                self.half = False
                _note_fp16_once(self.model, f"has conv + BatchNorm pairs without a folded half-precision form ({unfolded[0]}, {len(unfolded)} in all)")
        if self.half:
            self.model.half_inference(True)  # the private fused copy only: the caller's model keeps training in fp32
        self.fuse_model = False

    def __call__(self, inputs, batch_size: Optional[int] = 32):
        if isinstance(inputs, str) and inputs.lower().endswith(VIDEO_EXTENSIONS):
            raise NotImplementedError("predict() on video files is cv2 I/O, outside the MI355X hot path; pass frames as images")
        return self.predict_images(inputs, batch_size)

    def predict_images(self, images, batch_size: Optional[int] = 32):
        images = load_images(images)
        return self._combine_image_prediction_to_images(self._generate_prediction_result(images, batch_size), n_images=len(images))

    def _generate_prediction_result(self, images, batch_size: Optional[int] = None):
        batch_size = batch_size or len(images)
        for start in range(0, len(images), batch_size):
            yield from self._generate_prediction_result_single_batch(images[start:start + batch_size])

    def _generate_prediction_result_single_batch(self, images):
        batch, metadatas = self.image_processor.preprocess_batch(images, device=self.device)
        predictions = self.pass_images_through_model(batch)
        for image, prediction, metadata in zip(images, predictions, metadatas):
            prediction = self.image_processor.postprocess_predictions(predictions=prediction, metadata=metadata)
            yield self._instantiate_image_prediction(image=image, prediction=prediction)  # the caller's own object (a device tensor stays one)

    def pass_images_through_model(self, batch: torch.Tensor):
        return self._forward_raw(batch, self._decode_model_output)

    def _forward_raw(self, batch: torch.Tensor, decode):
        """Input-shape check, fuse-on-first-batch, eval forward; `decode(model_output, model_input=batch)` runs inside the no-grad / eval scope."""
        if hasattr(self.model, "get_input_shape_steps"):  # SupportsInputShapeCheck.validate_input_shape
            sh, sw = self.model.get_input_shape_steps()
            mh, mw = self.model.get_minimum_input_shape_size()
            h, w = batch.shape[-2:]
            if h % sh or w % sw or h < mh or w < mw:
                raise ValueError(f"Invalid input size ({h}, {w}): the model takes sizes that are multiples of ({sh}, {sw}) and at least ({mh}, {mw})")
        if self.fuse_model:
            self._fuse_model(tuple(batch.shape[-2:]))
        # (round 6: the mode is switched only when it has to be - eval() / train() walk the whole module tree, ~500 modules with an
        # attribute write each, and the fused copy a pipeline owns is in eval mode for good: two walks per batch were a quarter of a bf16
        # predict() batch's host time, tools/predict_profile.py)
        was_training = self.model.training
        if was_training:
            self.model.eval()
        try:
            with torch.no_grad():
                out = self.model(batch)
                return decode(out, model_input=batch)
        finally:
            if was_training:
                self.model.train(True)

    @abstractmethod
    def _decode_model_output(self, model_output, model_input):
        pass

    @abstractmethod
    def _instantiate_image_prediction(self, image, prediction):
        pass

    @abstractmethod
    def _combine_image_prediction_to_images(self, images_prediction_lst: Iterable, n_images: Optional[int] = None):
        pass


class DetectionPipeline(Pipeline):
    def __init__(self, model, class_names: List[str], post_prediction_callback, device: Optional[str] = None,
                 image_processor: Union[Processing, List[Processing]] = None, fuse_model: bool = True, fp16: bool = True):
        if isinstance(image_processor, list):
            image_processor = ComposeProcessing(image_processor)
        if not isinstance(image_processor, ComposeProcessing):
            image_processor = ComposeProcessing([image_processor])
        if not any(isinstance(p, ImagePermute) for p in image_processor._flat()):  # pipelines.py:311-313
            image_processor = ComposeProcessing(list(image_processor.processings) + [ImagePermute()])
        super().__init__(model=model, device=device, image_processor=image_processor, class_names=class_names, fuse_model=fuse_model, fp16=fp16)
        self.post_prediction_callback = post_prediction_callback

    def _decode_model_output(self, model_output, model_input):
        post_nms = self.post_prediction_callback(model_output, device=self.device)
        counts = [0 if r is None else int(r.shape[0]) for r in post_nms]
        kept = [r.detach().reshape(-1, 6) for r in post_nms if r is not None and r.shape[0]]
        flat = torch.cat(kept).cpu().numpy() if kept else np.zeros((0, 6), dtype=np.float32)  # ONE device-to-host copy for the batch
        preds, start = [], 0
        for n, image in zip(counts, model_input):
            rows, start = flat[start:start + n], start + n
            preds.append(DetectionPrediction(bboxes=rows[:, :4], confidence=rows[:, 4], labels=rows[:, 5].astype(int), bbox_format="xyxy",
                                             image_shape=tuple(image.shape)))
        return preds

    def _generate_prediction_result_single_batch(self, images):
        """Round 6: everything behind the forward stays on the device until ONE copy.  The reference (pipelines.py:222-247) and rounds 2-5 here
        copied the counts, concatenated and copied the kept rows, then mapped every image's boxes back through its processing stages in numpy
        (two fancy-indexed passes per image for the default YOLO-NAS processing) - 40 % of a bf16 predict() batch was spent behind the
        forward.  Now: NMS rows + counts stay device tensors, kernels.detection_unmap applies every image's inverse maps (the stages'
        `inverse_box_steps`: the same float32 operations in the same order, bit-identical boxes) and packs rows and counts into one
        buffer, one device-to-host copy, and the per-image objects are views of that array.  A processing stage without a step description
        (user-defined) or a callback without `forward_batched` keeps the host path."""
        from ... import kernels as K

        batch, metadatas = self.image_processor.preprocess_batch(images, device=self.device)
        steps = [self.image_processor.inverse_box_steps(m) for m in metadatas]
        if any(s is None for s in steps) or not hasattr(self.post_prediction_callback, "forward_batched") or os.environ.get("SGX_PREDICT_HOST_POST") == "1":
            yield from self._host_postprocess(images, self.pass_images_through_model(batch), metadatas)
            return
        B, nst = len(steps), max(len(s) for s in steps)
        st = None
        if nst:
            arr = np.full((B, nst, 3), 2.0, dtype=np.float32)
            for b, s in enumerate(steps):
                if s:
                    arr[b, :len(s)] = np.asarray(s, dtype=np.float64)  # (float64 -> float32 once, as numpy rounds the python floats of the host path)
            st = torch.from_numpy(arr).to(self.device)

        def decode(model_output, model_input):
            rows, cnt, _ = self.post_prediction_callback.forward_batched(model_output)
            return K.detection_unmap(rows, cnt, st).cpu().numpy(), int(rows.shape[1])

        flat, P = self._forward_raw(batch, decode)
        counts = flat[B * P * 6:].view(np.int32)
        rows = flat[:B * P * 6].reshape(B, P, 6)
        shape = tuple(batch.shape[1:])
        for b, image in enumerate(images):
            r = rows[b, :int(counts[b])]
            pred = DetectionPrediction(bboxes=r[:, :4], confidence=r[:, 4], labels=r[:, 5].astype(int), bbox_format="xyxy", image_shape=shape)
            yield self._instantiate_image_prediction(image=image, prediction=pred)

    def _host_postprocess(self, images, predictions, metadatas):
        for image, prediction, metadata in zip(images, predictions, metadatas):
            prediction = self.image_processor.postprocess_predictions(predictions=prediction, metadata=metadata)
            yield self._instantiate_image_prediction(image=image, prediction=prediction)

    def _instantiate_image_prediction(self, image, prediction):
        return ImageDetectionPrediction(image=image, prediction=prediction, class_names=self.class_names)

    def _combine_image_prediction_to_images(self, images_predictions, n_images: Optional[int] = None):
        if n_images == 1:
            return next(iter(images_predictions))
        return ImagesDetectionPrediction(_images_prediction_lst=list(images_predictions))

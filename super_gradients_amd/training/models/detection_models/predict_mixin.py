"""predict() surface of the detectors (reference: customizable_detector.py:155-338, the same block in pp_yolo_e.py): processing
parameters set on the model, a cached pipeline per argument set, `predict(images, ...)`."""
from typing import List, Optional


class DetectionPredictMixin:
    def _init_processing_params(self):
        self._class_names: Optional[List[str]] = None
        self._image_processor = None
        self._default_nms_iou, self._default_nms_conf, self._default_nms_top_k = 0.7, 0.5, 1024
        self._default_max_predictions, self._default_multi_label_per_box, self._default_class_agnostic_nms = 300, True, False
        self._pipeline_cache = None

    def set_dataset_processing_params(self, class_names: Optional[List[str]] = None, image_processor=None, iou: Optional[float] = None,
                                      conf: Optional[float] = None, nms_top_k: Optional[int] = None, max_predictions: Optional[int] = None,
                                      multi_label_per_box: Optional[bool] = None, class_agnostic_nms: Optional[bool] = None) -> None:
        from ....common.factories import ProcessingFactory

        if class_names is not None:
            self._class_names = tuple(class_names)
        if image_processor is not None:
            self._image_processor = ProcessingFactory().get(image_processor)  # @resolve_param("image_processor", ProcessingFactory())
        if iou is not None:
            self._default_nms_iou = float(iou)
        if conf is not None:
            self._default_nms_conf = float(conf)
        if nms_top_k is not None:
            self._default_nms_top_k = int(nms_top_k)
        if max_predictions is not None:
            self._default_max_predictions = int(max_predictions)
        if multi_label_per_box is not None:
            self._default_multi_label_per_box = bool(multi_label_per_box)
        if class_agnostic_nms is not None:
            self._default_class_agnostic_nms = bool(class_agnostic_nms)
        self._pipeline_cache = None

    def get_dataset_processing_params(self):
        """customizable_detector.py:222-232 - including its quirk: `conf` reports the IoU default."""
        return dict(class_names=self._class_names, image_processor=self._image_processor, iou=self._default_nms_iou, conf=self._default_nms_iou,
                    nms_top_k=self._default_nms_top_k, max_predictions=self._default_max_predictions,
                    multi_label_per_box=self._default_multi_label_per_box, class_agnostic_nms=self._default_class_agnostic_nms)

    def get_processing_params(self):
        return self._image_processor

    def get_class_names(self):
        return self._class_names

    def _get_pipeline(self, *, iou=None, conf=None, fuse_model: bool = True, skip_image_resizing: bool = False, nms_top_k=None,
                      max_predictions=None, multi_label_per_box=None, class_agnostic_nms=None, fp16: bool = True):
        from ...pipelines.pipelines import DetectionPipeline
        from ...processing.processing import ComposeProcessing, DetectionAutoPadding

        if None in (self._class_names, self._image_processor, self._default_nms_iou, self._default_nms_conf):
            raise RuntimeError("You must set the dataset processing parameters before calling predict.\n"
                               "Please call `model.set_dataset_processing_params(...)` first.")
        key = (iou, conf, fuse_model, skip_image_resizing, nms_top_k, max_predictions, multi_label_per_box, class_agnostic_nms, fp16)
        if self._pipeline_cache is not None and self._pipeline_cache[0] == key:  # @lru_cache(maxsize=1)
            return self._pipeline_cache[1]
        iou = self._default_nms_iou if iou is None else iou
        conf = self._default_nms_conf if conf is None else conf
        nms_top_k = self._default_nms_top_k if nms_top_k is None else nms_top_k
        max_predictions = self._default_max_predictions if max_predictions is None else max_predictions
        multi_label_per_box = self._default_multi_label_per_box if multi_label_per_box is None else multi_label_per_box
        class_agnostic_nms = self._default_class_agnostic_nms if class_agnostic_nms is None else class_agnostic_nms
        image_processor = self._image_processor
        if isinstance(image_processor, ComposeProcessing) and skip_image_resizing:  # the input must stay a multiple of 32
            image_processor = image_processor.get_equivalent_compose_without_resizing(DetectionAutoPadding(shape_multiple=(32, 32), pad_value=0))
        pipeline = DetectionPipeline(
            model=self, image_processor=image_processor, class_names=self._class_names, fuse_model=fuse_model, fp16=fp16,
            post_prediction_callback=self.get_post_prediction_callback(iou=iou, conf=conf, nms_top_k=nms_top_k, max_predictions=max_predictions,
                                                                       multi_label_per_box=multi_label_per_box, class_agnostic_nms=class_agnostic_nms))
        self._pipeline_cache = (key, pipeline)
        return pipeline

    def predict(self, images, iou: Optional[float] = None, conf: Optional[float] = None, batch_size: int = 32, fuse_model: bool = True,
                skip_image_resizing: bool = False, nms_top_k: Optional[int] = None, max_predictions: Optional[int] = None,
                multi_label_per_box: Optional[bool] = None, class_agnostic_nms: Optional[bool] = None, fp16: bool = True):
        """customizable_detector.py:286-330.  -> ImageDetectionPrediction (one image) / ImagesDetectionPrediction."""
        pipeline = self._get_pipeline(iou=iou, conf=conf, fuse_model=fuse_model, skip_image_resizing=skip_image_resizing, nms_top_k=nms_top_k,
                                      max_predictions=max_predictions, multi_label_per_box=multi_label_per_box,
                                      class_agnostic_nms=class_agnostic_nms, fp16=fp16)
        return pipeline(images, batch_size=batch_size)

    def predict_webcam(self, *a, **k):
        raise NotImplementedError("predict_webcam is cv2 camera I/O, outside the MI355X hot path")

    def train(self, mode: bool = True):
        self._pipeline_cache = None  # customizable_detector.py:366-369: a cached pipeline holds a fused copy of stale weights
        return super().train(mode)

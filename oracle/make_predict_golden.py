"""TEST INFRASTRUCTURE.  Golden vectors for predict()'s image processing, produced by the REFERENCE's own classes.

    python oracle/make_predict_golden.py        # writes tests/golden/predict_processing.pt  (needs /root/reference)

The reference's training/processing/processing.py runs unmodified through oracle/ref_shim.py for every stage except cv2.resize (cv2 is
not installed; the shim's stub would return garbage), so each case is built so that the reference never reaches cv2: images whose longest
side already equals the rescale target (scale factor 1.0: processing.py:554 skips the resize), and the `skip_image_resizing=True` form
(`get_equivalent_compose_without_resizing`, processing.py:186-202).  Stored per case: the compose as a config list, the uint8 image, the
reference's pre-processed image and metadata, random boxes and the reference's postprocess_predictions of them.  The box post-processing
of the *rescale* stages does not touch cv2 either, so those cases carry scale factors != 1.
tests/test_predict.py holds oracle/image.py and the HIP path (bit-exact) to these vectors.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tests", "golden", "predict_processing.pt")

IMAGENET_MEAN, IMAGENET_STD = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]

# (name, compose config, image shape, skip_image_resizing)
CASES = [
    ("yolo_nas_like_scale1", [{"DetectionLongestMaxSizeRescale": {"output_shape": (60, 60)}}, {"DetectionCenterPadding": {"output_shape": (64, 64), "pad_value": 114}},
                              {"StandardizeImage": {"max_value": 255.0}}, {"ImagePermute": {"permutation": (2, 0, 1)}}], (60, 45, 3), False),
    ("yolo_nas_like_scale1_wide", [{"DetectionLongestMaxSizeRescale": {"output_shape": (60, 60)}}, {"DetectionCenterPadding": {"output_shape": (64, 64), "pad_value": 114}},
                                   {"StandardizeImage": {"max_value": 255.0}}, {"ImagePermute": {"permutation": (2, 0, 1)}}], (31, 60, 3), False),
    ("yolo_nas_skip_resizing", [{"DetectionLongestMaxSizeRescale": {"output_shape": (636, 636)}}, {"DetectionCenterPadding": {"output_shape": (640, 640), "pad_value": 114}},
                                {"StandardizeImage": {"max_value": 255.0}}, {"ImagePermute": {"permutation": (2, 0, 1)}}], (50, 70, 3), True),
    ("yolo_nas_skip_resizing_aligned", [{"DetectionLongestMaxSizeRescale": {"output_shape": (636, 636)}}, {"DetectionCenterPadding": {"output_shape": (640, 640), "pad_value": 114}},
                                        {"StandardizeImage": {"max_value": 255.0}}, {"ImagePermute": {"permutation": (2, 0, 1)}}], (64, 96, 3), True),
    ("ppyoloe_like_no_rescale", [{"ReverseImageChannels": {}}, {"DetectionBottomRightPadding": {"output_shape": (96, 96), "pad_value": 114}},
                                 {"NormalizeImage": {"mean": [123.675, 116.28, 103.53], "std": [58.395, 57.12, 57.375]}},
                                 {"ImagePermute": {"permutation": (2, 0, 1)}}], (80, 90, 3), False),
    ("yolox_like_uint8_out", [{"ReverseImageChannels": {}}, {"DetectionLongestMaxSizeRescale": {"output_shape": (64, 64)}},
                              {"DetectionBottomRightPadding": {"output_shape": (64, 64), "pad_value": 114}}, {"ImagePermute": {"permutation": (2, 0, 1)}}],
     (64, 40, 3), False),
    ("standardize_and_normalize_tuple_pad", [{"DetectionCenterPadding": {"output_shape": (64, 64), "pad_value": (10, 20, 30)}}, {"StandardizeImage": {"max_value": 255.0}},
                                             {"NormalizeImage": {"mean": IMAGENET_MEAN, "std": IMAGENET_STD}}, {"ImagePermute": {"permutation": (2, 0, 1)}}],
     (33, 57, 3), False),
    ("standardize_other_max_value", [{"StandardizeImage": {"max_value": 127.5}}, {"ImagePermute": {"permutation": (2, 0, 1)}}], (32, 32, 3), False),
]

# box post-processing of single stages, with metadata that never came from an actual resize: (class name, ctor kwargs, metadata kind, metadata values)
POST_CASES = [
    ("DetectionLongestMaxSizeRescale", {"output_shape": (636, 636)}, "rescale", dict(original_shape=(480, 640), scale_factor_h=0.99375, scale_factor_w=0.99375)),
    ("DetectionRescale", {"output_shape": (640, 640)}, "rescale", dict(original_shape=(427, 640), scale_factor_h=640 / 427, scale_factor_w=1.0)),
    ("DetectionCenterPadding", {"output_shape": (640, 640), "pad_value": 114}, "pad", dict(top=80, bottom=80, left=2, right=2)),
    ("DetectionAutoPadding", {"shape_multiple": (32, 32), "pad_value": 0}, "pad", dict(top=0, bottom=14, left=0, right=26)),
]


def main():
    from oracle import ref_shim

    if not ref_shim.available():
        print("/root/reference is not present: the fixture cannot be generated", file=sys.stderr)
        return 1
    ref_shim.install()
    from super_gradients.training.processing import processing as P
    from super_gradients.training.transforms.utils import PaddingCoordinates
    from super_gradients.training.utils.predict import DetectionPrediction

    def build(cfg):
        return P.ComposeProcessing([getattr(P, name)(**kw) for c in cfg for name, kw in c.items()])

    def boxes_for(rng, n=7):
        xy = rng.uniform(0, 40, (n, 2))
        return np.concatenate([xy, xy + rng.uniform(2, 30, (n, 2))], 1).astype(np.float32)

    def meta_plain(m):
        if m is None:
            return None
        if hasattr(m, "metadata_lst"):
            return [meta_plain(x) for x in m.metadata_lst]
        if hasattr(m, "padding_coordinates"):
            c = m.padding_coordinates
            return dict(kind="pad", top=int(c.top), bottom=int(c.bottom), left=int(c.left), right=int(c.right))
        return dict(kind="rescale", original_shape=tuple(int(v) for v in m.original_shape), scale_factor_h=float(m.scale_factor_h), scale_factor_w=float(m.scale_factor_w))

    fx = dict(cases=[], post_cases=[])
    for i, (name, cfg, shape, skip) in enumerate(CASES):
        rng = np.random.default_rng(500 + i)
        image = rng.integers(0, 256, shape, dtype=np.uint8)
        cp = build(cfg)
        if skip:
            cp = cp.get_equivalent_compose_without_resizing(auto_padding=P.DetectionAutoPadding(shape_multiple=(32, 32), pad_value=0))
        out, md = cp.preprocess_image(image)
        boxes = boxes_for(rng)
        pred = DetectionPrediction(bboxes=boxes.copy(), bbox_format="xyxy", confidence=np.ones(len(boxes), np.float32), labels=np.zeros(len(boxes), int), image_shape=out.shape)
        post = cp.postprocess_predictions(pred, md).bboxes_xyxy
        fx["cases"].append(dict(name=name, config=cfg, skip_image_resizing=skip, image=torch.from_numpy(image), output=torch.from_numpy(np.ascontiguousarray(out)),
                                metadata=meta_plain(md), boxes=torch.from_numpy(boxes), post_boxes=torch.from_numpy(np.asarray(post))))
    for i, (cls, kw, kind, vals) in enumerate(POST_CASES):
        rng = np.random.default_rng(900 + i)
        proc = getattr(P, cls)(**kw)
        md = P.RescaleMetadata(**vals) if kind == "rescale" else P.DetectionPadToSizeMetadata(padding_coordinates=PaddingCoordinates(**vals))
        boxes = boxes_for(rng, 9) * 10
        pred = DetectionPrediction(bboxes=boxes.copy(), bbox_format="xyxy", confidence=np.ones(len(boxes), np.float32), labels=np.zeros(len(boxes), int), image_shape=(640, 640))
        post = proc.postprocess_predictions(pred, md).bboxes_xyxy
        fx["post_cases"].append(dict(cls=cls, kwargs=kw, kind=kind, values=vals, boxes=torch.from_numpy(boxes), post_boxes=torch.from_numpy(np.asarray(post))))
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    torch.save(fx, OUT)
    print(f"wrote {OUT}: {len(fx['cases'])} pre-processing cases, {len(fx['post_cases'])} box post-processing cases")
    return 0


if __name__ == "__main__":
    sys.exit(main())

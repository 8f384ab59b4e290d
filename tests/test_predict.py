"""predict(): device image pre-processing (csrc/image.hip), the Processing classes, the pipeline and the box post-processing.

Oracle chain: the reference's own processing classes (live through oracle/ref_shim.py when /root/reference is present; otherwise the vectors
they produced, tests/golden/predict_processing.pt) pin oracle/image.py AND the HIP path, bit-exact, for every stage except cv2.resize.  The
rescale stage is checked against oracle/image.py's restatement of OpenCV's INTER_LINEAR only (cv2 is not installed: parity unpinned, said in
DESIGN.md), plus a sanity bound against ideal bilinear interpolation."""
import os

import numpy as np
import pytest
import torch

from oracle import image as O
from oracle import ref_shim

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "predict_processing.pt")


def _fixture():
    return torch.load(GOLDEN, weights_only=False)


def _product_compose(cfg, skip):
    from super_gradients_amd.common.factories import ProcessingFactory
    from super_gradients_amd.training.processing import DetectionAutoPadding

    cp = ProcessingFactory().get([{k: dict(v) for k, v in c.items()} for c in cfg])
    if skip:
        cp = cp.get_equivalent_compose_without_resizing(DetectionAutoPadding(shape_multiple=(32, 32), pad_value=0))
    return cp


def _meta_plain(m):
    if m is None:
        return None
    if hasattr(m, "metadata_lst"):
        return [_meta_plain(x) for x in m.metadata_lst]
    if hasattr(m, "padding_coordinates"):
        c = m.padding_coordinates
        return dict(kind="pad", top=int(c.top), bottom=int(c.bottom), left=int(c.left), right=int(c.right))
    return dict(kind="rescale", original_shape=tuple(int(v) for v in m.original_shape), scale_factor_h=float(m.scale_factor_h),
                scale_factor_w=float(m.scale_factor_w))


def _oracle_preprocess(cfg, skip, image):
    """oracle/image.py driven by the same config list -> (image as the reference returns it, [(kind, values) per stage])"""
    stages = [(k, dict(v)) for c in cfg for k, v in c.items()]
    if skip:
        stages = [("DetectionAutoPadding", dict(shape_multiple=(32, 32), pad_value=0))] + [s for s in stages if "Rescale" not in s[0] and "Padding" not in s[0]]
    x = image
    for name, kw in stages:
        if name == "ReverseImageChannels":
            x = x[..., ::-1]
        elif name == "DetectionLongestMaxSizeRescale":
            h, w, _ = O.longest_max_size(x.shape[:2], kw["output_shape"])
            x = O.resize_linear_u8(x, (h, w))
        elif name == "DetectionRescale":
            x = O.resize_linear_u8(x, kw["output_shape"])
        elif name == "DetectionCenterPadding":
            x = O.pad(x, O.center_padding(x.shape[:2], kw["output_shape"]), kw["pad_value"])
        elif name == "DetectionBottomRightPadding":
            x = O.pad(x, O.bottom_right_padding(x.shape[:2], kw["output_shape"]), kw["pad_value"])
        elif name == "DetectionAutoPadding":
            x = O.pad(x, O.auto_padding(x.shape[:2], kw["shape_multiple"]), kw["pad_value"])
        elif name == "StandardizeImage":
            x = O.standardize(x, kw["max_value"])
        elif name == "NormalizeImage":
            x = O.normalize(x, kw["mean"], kw["std"])
        elif name == "ImagePermute":
            x = np.ascontiguousarray(x.transpose(kw["permutation"]))
    return x


def test_golden_fixture_matches_live_reference():
    """The committed vectors are what the reference's classes produce today (skipped where /root/reference does not exist)."""
    if not ref_shim.available():
        pytest.skip("/root/reference not present")
    ref_shim.install()
    from super_gradients.training.processing import processing as P

    for case in _fixture()["cases"]:
        cp = P.ComposeProcessing([getattr(P, name)(**kw) for c in case["config"] for name, kw in c.items()])
        if case["skip_image_resizing"]:
            cp = cp.get_equivalent_compose_without_resizing(auto_padding=P.DetectionAutoPadding(shape_multiple=(32, 32), pad_value=0))
        out, md = cp.preprocess_image(case["image"].numpy())
        assert out.dtype == case["output"].numpy().dtype and np.array_equal(out, case["output"].numpy()), case["name"]


def test_oracle_image_matches_reference_vectors():
    for case in _fixture()["cases"]:
        out = _oracle_preprocess(case["config"], case["skip_image_resizing"], case["image"].numpy())
        ref = case["output"].numpy()
        assert out.dtype == ref.dtype and np.array_equal(out, ref), case["name"]


def test_oracle_resize_is_bilinear_within_one_level():
    """Sanity bound for the unpinned restatement: within one grey level of exact half-pixel-centre bilinear interpolation, up and down."""
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (37, 53, 3), dtype=np.uint8)
    for h, w in [(60, 80), (20, 30), (37, 106), (74, 53), (100, 100)]:
        ys, xs = (np.arange(h) + 0.5) * (37 / h) - 0.5, (np.arange(w) + 0.5) * (53 / w) - 0.5
        y0, x0 = np.floor(ys).astype(int), np.floor(xs).astype(int)
        fy, fx = (ys - y0)[:, None, None], (xs - x0)[None, :, None]
        yc, y1, xc, x1 = np.clip(y0, 0, 36), np.clip(y0 + 1, 0, 36), np.clip(x0, 0, 52), np.clip(x0 + 1, 0, 52)
        f = img.astype(np.float64)
        ideal = (f[yc][:, xc] * (1 - fx) + f[yc][:, x1] * fx) * (1 - fy) + (f[y1][:, xc] * (1 - fx) + f[y1][:, x1] * fx) * fy
        assert np.abs(O.resize_linear_u8(img, (h, w)) - ideal).max() <= 1.0
    # the textbook two-pixel case: cv2.resize(np.array([[0, 255]], np.uint8), (4, 1)) is [[0, 64, 191, 255]] (quarter / three-quarter weights
    # 512 and 1536 of 2048, rounded by the fixed-point cast) - a known answer, hand-checked through the arithmetic, not a cv2 run
    assert O.resize_linear_u8(np.array([[[0], [255]]], np.uint8), (1, 4))[0, :, 0].tolist() == [0, 64, 191, 255]
    even = rng.integers(0, 256, (40, 64, 3), dtype=np.uint8)  # exact 2x: the 2x2 mean shortcut
    area = O.resize_linear_u8(even, (20, 32)).astype(int)
    mean = even.reshape(20, 2, 32, 2, 3).astype(int).sum((1, 3))
    assert np.array_equal(area, (mean + 2) >> 2)


def test_processing_matches_reference_vectors(backend):
    """Every reference-produced vector: the one-launch device pre-processing is bit-exact, the metadata equal, the boxes mapped back equal."""
    from super_gradients_amd.training.utils.predict import DetectionPrediction

    for case in _fixture()["cases"]:
        cp = _product_compose(case["config"], case["skip_image_resizing"])
        image = case["image"].numpy()
        batch, metas = cp.preprocess_batch([image], device=backend)
        ref = case["output"].numpy()
        got = batch[0].cpu().numpy()
        assert got.shape == ref.shape, case["name"]
        assert np.array_equal(got, ref.astype(np.float32)), f"{case['name']}: max diff {np.abs(got - ref).max()}"
        arr, md = cp.preprocess_image(torch.from_numpy(image).to(backend) if backend.type == "cuda" else image)
        assert arr.dtype == ref.dtype and np.array_equal(arr, ref), case["name"]
        assert _meta_plain(md) == case["metadata"] == _meta_plain(metas[0]), case["name"]
        boxes = case["boxes"].numpy()
        pred = DetectionPrediction(bboxes=boxes.copy(), bbox_format="xyxy", confidence=np.ones(len(boxes), np.float32), labels=np.zeros(len(boxes), int),
                                   image_shape=ref.shape)
        post = cp.postprocess_predictions(pred, md).bboxes_xyxy
        assert post.dtype == case["post_boxes"].numpy().dtype and np.array_equal(post, case["post_boxes"].numpy()), case["name"]


def test_box_postprocessing_matches_reference_vectors():
    from super_gradients_amd.training import processing as P
    from super_gradients_amd.training.utils.predict import DetectionPrediction

    for case in _fixture()["post_cases"]:
        proc = getattr(P, case["cls"])(**case["kwargs"])
        md = P.RescaleMetadata(**case["values"]) if case["kind"] == "rescale" else P.DetectionPadToSizeMetadata(P.PaddingCoordinates(**case["values"]))
        boxes = case["boxes"].numpy()
        pred = DetectionPrediction(bboxes=boxes.copy(), bbox_format="xyxy", confidence=np.ones(len(boxes), np.float32), labels=np.zeros(len(boxes), int),
                                   image_shape=(640, 640))
        post = proc.postprocess_predictions(pred, md).bboxes_xyxy
        ref = case["post_boxes"].numpy()
        assert post.dtype == ref.dtype and np.array_equal(post, ref), case["cls"]


@pytest.mark.parametrize("shape,target", [((50, 70, 3), (60, 60)), ((97, 61, 3), (64, 64)), ((30, 45, 3), (64, 64)), ((128, 96, 3), (64, 64)),
                                          ((64, 64, 1), (48, 48)), ((41, 83, 4), (32, 32))])
def test_rescale_pad_standardize_vs_oracle(backend, shape, target):
    """Ragged batches through the rescaling stages (down, up, exact 2x, 1- and 4-channel images): HIP launch == oracle/image.py, bit-exact.
    (The restated cv2 arithmetic itself is unpinned - module docstring.)"""
    from super_gradients_amd.training.processing import (ComposeProcessing, DetectionCenterPadding, DetectionLongestMaxSizeRescale, DetectionRescale,
                                                         ImagePermute, NormalizeImage, ReverseImageChannels, StandardizeImage)

    rng = np.random.default_rng(hash(shape) % 1000)
    c = shape[2]
    images = [rng.integers(0, 256, shape, dtype=np.uint8), rng.integers(0, 256, (shape[1], shape[0], c), dtype=np.uint8),
              rng.integers(0, 256, (target[0], target[1] - 3, c), dtype=np.uint8)]
    H, W = target[0] + 4, target[1] + 4
    cfg = [{"DetectionLongestMaxSizeRescale": {"output_shape": target}}, {"DetectionCenterPadding": {"output_shape": (H, W), "pad_value": 114}},
           {"StandardizeImage": {"max_value": 255.0}}, {"ImagePermute": {"permutation": (2, 0, 1)}}]
    cp = ComposeProcessing([DetectionLongestMaxSizeRescale(target), DetectionCenterPadding((H, W), 114), StandardizeImage(255.0), ImagePermute()])
    batch, metas = cp.preprocess_batch([torch.from_numpy(i) for i in images], device=backend)
    assert tuple(batch.shape) == (3, c, H, W)
    for i, img in enumerate(images):
        assert np.array_equal(batch[i].cpu().numpy(), _oracle_preprocess(cfg, False, img)), f"image {i}"
    if c == 3:  # the PP-YOLOE form: reversed channels, aspect-changing rescale, ImageNet statistics on the 0..255 scale
        mean, std = [123.675, 116.28, 103.53], [58.395, 57.12, 57.375]
        cfg2 = [{"ReverseImageChannels": {}}, {"DetectionRescale": {"output_shape": target}}, {"NormalizeImage": {"mean": mean, "std": std}},
                {"ImagePermute": {"permutation": (2, 0, 1)}}]
        cp2 = ComposeProcessing([ReverseImageChannels(), DetectionRescale(target), NormalizeImage(mean, std), ImagePermute()])
        batch2, metas2 = cp2.preprocess_batch(images, device=backend)
        for i, img in enumerate(images):
            assert np.array_equal(batch2[i].cpu().numpy(), _oracle_preprocess(cfg2, False, img)), f"ppyoloe form, image {i}"
        assert metas2[0].metadata_lst[1].scale_factor_h == target[0] / shape[0] and metas2[0].metadata_lst[1].scale_factor_w == target[1] / shape[1]


def test_processing_errors():
    from super_gradients_amd.training.processing import ComposeProcessing, DetectionCenterPadding, ImagePermute, StandardizeImage

    with pytest.raises(NotImplementedError):  # a stage order the fused launch does not cover: no silent host fallback
        ComposeProcessing([StandardizeImage(), DetectionCenterPadding((64, 64), 0)]).plan_image((32, 32, 3))
    with pytest.raises(ValueError):
        ComposeProcessing([DetectionCenterPadding((16, 16), 0)]).plan_image((32, 32, 3))
    with pytest.raises(ValueError):
        ComposeProcessing([ImagePermute()]).plan_image((32, 32))
    # dataset-derived lists repeat the rescale (preprocessing_unit_test.py:119-123): fine while the second one changes nothing
    from super_gradients_amd.training.processing import DetectionBottomRightPadding, DetectionLongestMaxSizeRescale, DetectionRescale, ReverseImageChannels

    twice = ComposeProcessing([ReverseImageChannels(), DetectionLongestMaxSizeRescale((64, 64)), DetectionLongestMaxSizeRescale((64, 64)),
                               DetectionBottomRightPadding((64, 64), 114), ImagePermute()])
    plan, md = twice.plan_image((100, 50, 3))
    assert (plan.h, plan.w, plan.out_hw) == (64, 32, (64, 64)) and md.metadata_lst[1].scale_factor_h == 0.64 and md.metadata_lst[2].scale_factor_h == 1.0
    with pytest.raises(NotImplementedError, match="successive resamplings"):
        ComposeProcessing([DetectionLongestMaxSizeRescale((64, 64)), DetectionRescale((32, 32))]).plan_image((100, 50, 3))
    with pytest.raises(ValueError, match="0..255"):  # checked before anything is uploaded
        ComposeProcessing([DetectionCenterPadding((64, 64), 300)]).preprocess_batch([np.zeros((32, 32, 3), np.uint8)], device="cpu")


def _shrunk_arch():
    """YOLO-NAS-S's wiring with 1/6 of the channels and one bottleneck per stage: the host emulation of the kernels runs it in seconds."""
    st = lambda c, h: {"YoloNASStage": {"out_channels": c, "num_blocks": 1, "activation_type": "relu", "hidden_channels": h, "concat_intermediates": False}}  # noqa: E731
    up = lambda c, h: {"YoloNASUpStage": {"out_channels": c, "num_blocks": 1, "hidden_channels": h, "width_mult": 1, "depth_mult": 1,  # noqa: E731
                                          "activation_type": "relu", "reduce_channels": True}}
    down = lambda c, h: {"YoloNASDownStage": {"out_channels": c, "num_blocks": 1, "hidden_channels": h, "activation_type": "relu", "width_mult": 1,  # noqa: E731
                                              "depth_mult": 1}}
    head = lambda c, s: {"YoloNASDFLHead": {"inter_channels": c, "width_mult": 0.5, "first_conv_group_size": 0, "stride": s}}  # noqa: E731
    return dict(
        backbone={"NStageBackbone": {"stem": {"YoloNASStem": {"out_channels": 8}}, "stages": [st(16, 8), st(32, 8), st(64, 16), st(128, 32)],
                                     "context_module": {"SPP": {"output_channels": 128, "activation_type": "relu", "k": [5, 9, 13]}},
                                     "out_layers": ["stage1", "stage2", "stage3", "context_module"]}},
        neck={"YoloNASPANNeckWithC2": {"neck1": up(32, 8), "neck2": up(16, 8), "neck3": down(32, 8), "neck4": down(64, 8)}},
        heads={"NDFLHeads": {"num_classes": 3, "reg_max": 16, "heads_list": [head(32, 8), head(64, 16), head(128, 32)]}})


def _small_detector(device, num_classes=3):
    from super_gradients_amd.training import models

    net = models.get("yolo_nas_s", num_classes=num_classes, arch_params=None if device.type == "cuda" else _shrunk_arch())
    g = torch.Generator().manual_seed(11)
    for m in net.modules():  # non-trivial running statistics, so the fused and branch forms differ in arithmetic
        if hasattr(m, "running_var"):
            m.running_mean.normal_(0, 0.1, generator=g)
            m.running_var.uniform_(0.8, 1.2, generator=g)
    net.materialize(device)
    return net


def test_predict_pipeline(backend):
    """model.predict(): ragged images -> one pre-processing launch per batch -> fused copy of the model -> NMS -> boxes in each image's own
    frame.  Checked stage by stage against the same stages run by hand (device pre-processing and NMS have their own bit-exact tests): the
    pipeline's batching, metadata bookkeeping, the reference's fuse-on-first-batch copy, and the inverse box maps (oracle/image.py)."""
    from super_gradients_amd.modules.qarepvgg_block import QARepVGGBlock
    from super_gradients_amd.training.processing import (ComposeProcessing, DetectionCenterPadding, DetectionLongestMaxSizeRescale, ImagePermute,
                                                         StandardizeImage)
    from super_gradients_amd.training.utils.predict import ImageDetectionPrediction, ImagesDetectionPrediction

    net = _small_detector(backend)
    with pytest.raises(RuntimeError):
        net.predict(np.zeros((32, 32, 3), np.uint8))
    proc = [{"DetectionLongestMaxSizeRescale": {"output_shape": (30, 30)}}, {"DetectionCenterPadding": {"output_shape": (32, 32), "pad_value": 114}},
            {"StandardizeImage": {"max_value": 255.0}}, {"ImagePermute": {"permutation": (2, 0, 1)}}]
    net.set_dataset_processing_params(class_names=["a", "b", "c"], image_processor=proc, iou=0.6, conf=0.0)
    assert isinstance(net.get_processing_params(), ComposeProcessing) and net.get_class_names() == ("a", "b", "c")
    assert net.get_dataset_processing_params()["conf"] == 0.6  # the reference reports the IoU default there (customizable_detector.py:227)
    rng = np.random.default_rng(5)
    images = [rng.integers(0, 256, s, dtype=np.uint8) for s in [(25, 35, 3), (45, 20, 3)]]
    net.train()
    res = net.predict(images, batch_size=2, max_predictions=10, nms_top_k=40)
    assert net.training and not any(m.partially_fused or m.fully_fused for m in net.modules() if isinstance(m, QARepVGGBlock))  # the caller's model is untouched
    assert isinstance(res, ImagesDetectionPrediction) and len(res) == 2

    # the same stages by hand
    cp = ComposeProcessing([DetectionLongestMaxSizeRescale((30, 30)), DetectionCenterPadding((32, 32), 114), StandardizeImage(255.0), ImagePermute()])
    pipe = net._get_pipeline(max_predictions=10, nms_top_k=40)
    assert pipe is net._get_pipeline(max_predictions=10, nms_top_k=40)  # cached per argument set
    fused = pipe.model
    assert fused is not net and all(m.fully_fused for m in fused.modules() if isinstance(m, QARepVGGBlock))
    cb = net.get_post_prediction_callback(conf=0.0, iou=0.6, nms_top_k=40, max_predictions=10, multi_label_per_box=True, class_agnostic_nms=False)
    for start in (0,):
        chunk = images[start:start + 2]
        batch, metas = cp.preprocess_batch(chunk, device=backend)
        with torch.no_grad():
            rows = cb(fused(batch), device=backend)
        for j, (img, r, md) in enumerate(zip(chunk, rows, metas)):
            r = r.cpu().numpy()
            pad, scale = md.metadata_lst[1].padding_coordinates, md.metadata_lst[0].scale_factor_h
            boxes = O.rescale_boxes(O.shift_boxes(r[:, :4], -pad.left, -pad.top), (1 / scale, 1 / scale))
            got = res[start + j]
            assert isinstance(got, ImageDetectionPrediction) and got.class_names == ("a", "b", "c") and got.image is img
            assert len(got.prediction) == len(r) > 0
            assert np.array_equal(got.prediction.bboxes_xyxy, boxes) and np.array_equal(got.prediction.confidence, r[:, 4])
            assert np.array_equal(got.prediction.labels, r[:, 5].astype(int))
    seen = []  # batch_size splits the image list (pipelines.py:182-190); no forward needed to see it
    single, pipe._generate_prediction_result_single_batch = pipe._generate_prediction_result_single_batch, lambda imgs: seen.append(len(imgs)) or iter(())
    list(pipe._generate_prediction_result(images * 3 + images[:1], batch_size=3))
    pipe._generate_prediction_result_single_batch = single
    assert seen == [3, 3, 1]
    # one image: the bare ImageDetectionPrediction; skip_image_resizing pads to a multiple of 32 instead of rescaling
    one = net.predict(images[0], skip_image_resizing=True, max_predictions=5)
    assert isinstance(one, ImageDetectionPrediction) and len(one.prediction) == 5
    with pytest.raises(ValueError):  # ragged sizes cannot share a batch without the resize
        net.predict(images, skip_image_resizing=True)
    net.eval()
    net.train()
    assert net._pipeline_cache is None  # train() drops the cached pipelines (their fused copies hold stale weights)


@pytest.mark.gpu
def test_predict_eval_forward_matches_oracle(gpu_device):
    """The forward predict() runs (fused copy, eval) against the oracle network on the oracle's pre-processed batch: same detections up to
    fp32 round-off of the fused arithmetic (scores 1e-4 relative, boxes 1e-3 px)."""
    from oracle.yolo_nas import YoloNAS as OracleYoloNAS
    from super_gradients_amd.training.processing import ComposeProcessing, DetectionCenterPadding, ImagePermute, StandardizeImage

    backend = gpu_device
    net = _small_detector(backend)
    ref = OracleYoloNAS("s", num_classes=3)
    ref.load_state_dict({k: v.cpu() for k, v in net.state_dict().items()}, strict=True)
    ref.eval()
    rng = np.random.default_rng(8)
    images = [rng.integers(0, 256, (64, 50, 3), dtype=np.uint8), rng.integers(0, 256, (40, 64, 3), dtype=np.uint8)]
    cfg = [{"DetectionCenterPadding": {"output_shape": (64, 64), "pad_value": 114}}, {"StandardizeImage": {"max_value": 255.0}},
           {"ImagePermute": {"permutation": (2, 0, 1)}}]
    x_ref = torch.from_numpy(np.stack([_oracle_preprocess(cfg, False, i) for i in images]))
    with torch.no_grad():
        (bx_r, sc_r), _ = ref(x_ref)
    net.set_dataset_processing_params(class_names=["a", "b", "c"], image_processor=ComposeProcessing(
        [DetectionCenterPadding((64, 64), 114), StandardizeImage(255.0), ImagePermute()]), conf=0.0)
    pipe = net._get_pipeline(fp16=False)  # the fp32 path: parity with the fp32 oracle (the half path: tests/test_half.py)
    batch, _ = pipe.image_processor.preprocess_batch(images, device=backend)
    assert np.array_equal(batch.cpu().numpy(), x_ref.numpy())
    pipe.pass_images_through_model(batch)  # fuses on the first batch
    with torch.no_grad():
        (bx, sc), _ = pipe.model(batch)
    from util import assert_close

    assert_close(sc.cpu(), sc_r, 1e-4, "scores")
    assert float((bx.cpu() - bx_r).abs().max()) < 1e-3


def test_processing_params_round_trip_through_checkpoint(tmp_path):
    """Processing.to_config() -> checkpoint["processing_params"] (plain containers: loads with weights_only=True) -> models.get(checkpoint_path)
    -> set_dataset_processing_params (checkpoint_utils.py:1625-1651); the Trainer's hand-over from a validation dataset that knows its
    pre-processing (sg_trainer.py:1704-1720)."""
    from super_gradients_amd.common.factories import ProcessingFactory
    from super_gradients_amd.training import Trainer, models
    from super_gradients_amd.training.processing import ComposeProcessing, default_ppyoloe_coco_processing_params, default_yolo_nas_coco_processing_params

    for params in (default_yolo_nas_coco_processing_params(), default_ppyoloe_coco_processing_params()):
        cfg = params["image_processor"].to_config()
        again = ProcessingFactory().get(cfg)
        assert isinstance(again, ComposeProcessing) and again.to_config() == cfg
        p0, m0 = params["image_processor"].plan_image((480, 640, 3))
        p1, m1 = again.plan_image((480, 640, 3))
        assert p0.photometric_key() == p1.photometric_key() and (p0.h, p0.w, p0.top, p0.left) == (p1.h, p1.w, p1.top, p1.left) and m0 == m1

    net = models.get("yolo_nas_s", num_classes=80)
    params = default_yolo_nas_coco_processing_params()
    path = str(tmp_path / "ckpt.pth")
    torch.save({"net": net.state_dict(), "processing_params": dict(class_names=list(params["class_names"]), iou=0.65, conf=0.3,
                                                                  image_processor=params["image_processor"].to_config())}, path)
    loaded = models.get("yolo_nas_s", num_classes=80, checkpoint_path=path)
    assert loaded.get_class_names() == tuple(params["class_names"]) and loaded._default_nms_iou == 0.65 and loaded._default_nms_conf == 0.3
    assert loaded.get_processing_params().to_config() == params["image_processor"].to_config()

    class _DS:
        def get_dataset_preprocessing_params(self):
            return default_yolo_nas_coco_processing_params()

    class _Loader:
        dataset = _DS()

    tr = Trainer("pp", ckpt_root_dir=str(tmp_path))
    tr.net = net
    got = tr._get_preprocessing_from_valid_loader(_Loader())
    assert got["conf"] == 0.25 and isinstance(got["image_processor"], ComposeProcessing)
    assert tr._get_preprocessing_from_valid_loader(object()) is None


def test_predict_on_a_fresh_model(backend):
    """models.get(...) -> set_dataset_processing_params -> predict() with nothing in between (the reference's usual call sequence): the
    pipeline materialises the model in HBM before it takes the fused copy."""
    from super_gradients_amd.training import models
    from super_gradients_amd.training.utils.predict import ImageDetectionPrediction

    net = models.get("yolo_nas_s", num_classes=3, arch_params=None if backend.type == "cuda" else _shrunk_arch())
    net.set_dataset_processing_params(class_names=["a", "b", "c"], conf=0.0, image_processor=[
        {"DetectionCenterPadding": {"output_shape": (32, 32), "pad_value": 114}}, {"StandardizeImage": {"max_value": 255.0}}])
    img = np.random.default_rng(0).integers(0, 256, (30, 28, 3), dtype=np.uint8)
    res = net.predict(img, max_predictions=3)
    assert isinstance(res, ImageDetectionPrediction) and len(res.prediction) == 3 and net._materialized
    assert res.prediction.bboxes_xyxy.dtype == np.float32 and res.prediction.labels.dtype.kind == "i"


def test_reference_unit_test_vectors_for_processing_helpers():
    """The reference's own known answers for this path (tests/unit_tests/transforms_test.py: test_rescale_bboxes :231-244, test_shift_bboxes
    :301-308, test_rescale_xyxy_bboxes :310-317, test_padding :319-349, test_get_padding_coordinates :351-405), transplanted: the numbers are
    the reference's, the functions under test are oracle/image.py and the product's processing helpers."""
    from super_gradients_amd.training.processing import DetectionBottomRightPadding, DetectionCenterPadding, PaddingCoordinates
    from super_gradients_amd.training.processing.processing import _rescale_bboxes, _shift_bboxes_xyxy

    # box rescale by (sy, sx) = (2.0, 0.5): empty and non-empty (extra columns travel unchanged)
    boxes = np.array([[10, 20, 50, 60, 1], [30, 40, 80, 90, 2]], dtype=np.float32)
    want = np.array([[5.0, 40.0, 25.0, 120.0, 1.0], [15.0, 80.0, 40.0, 180.0, 2.0]], dtype=np.float32)
    for fn in (_rescale_bboxes, O.rescale_boxes):
        assert np.array_equal(fn(np.zeros((0, 4)), (2.0, 0.5)), np.zeros((0, 4)))
        assert np.array_equal(fn(boxes, (2.0, 0.5)), want)
        assert np.array_equal(fn(boxes, (0.5, 0.5)), np.array([[5.0, 10.0, 25.0, 30.0, 1.0], [15.0, 20.0, 40.0, 45.0, 2.0]], dtype=np.float32))
    # box shift by (w, h) = (60, 80)
    shifted = np.array([[70, 100, 110, 140, 1], [90, 120, 140, 170, 2]], dtype=np.float32)
    assert np.array_equal(_shift_bboxes_xyxy(boxes, 60, 80), shifted) and np.array_equal(O.shift_boxes(boxes, 60, 80), shifted)
    # padding coordinates: width only, height only, both, image larger than the output (negative coordinates), 3-channel shape
    cases = [((640, 480), (640, 640), (0, 0, 80, 80), (0, 0, 0, 160)), ((480, 640), (640, 640), (80, 80, 0, 0), (0, 160, 0, 0)),
             ((480, 640), (800, 800), (160, 160, 80, 80), (0, 320, 0, 160)), ((800, 800), (640, 640), (-80, -80, -80, -80), (0, -160, 0, -160)),
             ((480, 640, 3), (800, 800), (160, 160, 80, 80), (0, 320, 0, 160))]
    for shape, out, center, br in cases:
        assert O.center_padding(shape, out) == center and O.bottom_right_padding(shape, out) == br
        c = DetectionCenterPadding(out, 114)._get_padding_params(shape)
        b = DetectionBottomRightPadding(out, 114)._get_padding_params(shape)
        assert c == PaddingCoordinates(top=center[0], bottom=center[1], left=center[2], right=center[3])
        assert b == PaddingCoordinates(top=br[0], bottom=br[1], left=br[2], right=br[3])
    # padding itself: bottom 1 / right 2 with 0 on a 2x2x3 image
    img = np.array([[[1, 2, 3], [4, 5, 6]], [[7, 8, 9], [10, 11, 12]]], dtype=np.uint8)
    want = np.array([[[1, 2, 3], [4, 5, 6], [0, 0, 0], [0, 0, 0]], [[7, 8, 9], [10, 11, 12], [0, 0, 0], [0, 0, 0]], [[0, 0, 0]] * 4], dtype=np.uint8)
    assert np.array_equal(O.pad(img, (0, 1, 0, 2), 0), want)
    assert np.array_equal(O.pad(img, (0, 0, 0, 0), 114), img)


def test_skip_image_resizing_with_reversed_channels(backend):
    """the reference's skip_image_resizing compose puts the auto-padding FIRST (processing.py:186-202), and the PP-YOLOE /
    dataset-derived pipelines start with ReverseImageChannels - so the reversal arrives after a padding stage.  It commutes with the padding
    (per-channel pad value reversed): the fused launch must take it, bit-exact against the stage-by-stage oracle, for a scalar and for a
    per-channel pad value."""
    from super_gradients_amd.training.processing import (ComposeProcessing, DetectionAutoPadding, DetectionRescale, ImagePermute, NormalizeImage,
                                                         ReverseImageChannels, StandardizeImage, default_ppyoloe_coco_processing_params)

    rng = np.random.default_rng(11)
    img = rng.integers(0, 256, (45, 70, 3), dtype=np.uint8)
    mean, std = [123.675, 116.28, 103.53], [58.395, 57.12, 57.375]
    for pad_value in (0, (10, 20, 30)):
        cp = ComposeProcessing([ReverseImageChannels(), DetectionRescale((64, 64)), NormalizeImage(mean, std), ImagePermute()])
        skip = cp.get_equivalent_compose_without_resizing(DetectionAutoPadding(shape_multiple=(32, 32), pad_value=pad_value))
        batch, metas = skip.preprocess_batch([img], device=backend)
        x = O.pad(img, O.auto_padding(img.shape[:2], (32, 32)), pad_value)[..., ::-1]
        want = np.ascontiguousarray(O.normalize(x, mean, std).transpose(2, 0, 1))
        assert tuple(batch.shape) == (1, 3, 64, 96)
        assert np.array_equal(batch[0].cpu().numpy(), want), f"pad_value {pad_value}"
    # the reference's default PP-YOLOE processing, as predict(skip_image_resizing=True) builds it
    params = default_ppyoloe_coco_processing_params()
    proc = params["image_processor"]
    skip = proc.get_equivalent_compose_without_resizing(DetectionAutoPadding(shape_multiple=(32, 32), pad_value=0))
    batch, _ = skip.preprocess_batch([img], device=backend)
    assert tuple(batch.shape)[2:] == (64, 96)


def test_predict_device_postprocess_equals_host_postprocess(backend, monkeypatch):
    """Round 6: predict() maps every image's boxes back through its processing stages on the DEVICE (kernels.detection_unmap, the stages'
    `inverse_box_steps`) and copies rows + counts to the host once.  The result must equal - bit for bit - the host path the reference
    describes (processing.py:361-364, 401-403: numpy float32 passes per stage per image), which stays the fallback for stages without a
    step description.  Ragged images: a different scale factor and padding per image."""
    from super_gradients_amd.training.processing import ComposeProcessing

    net = _small_detector(backend)
    proc = [{"DetectionLongestMaxSizeRescale": {"output_shape": (60, 60)}}, {"DetectionCenterPadding": {"output_shape": (64, 64), "pad_value": 114}},
            {"StandardizeImage": {"max_value": 255.0}}, {"ImagePermute": {"permutation": (2, 0, 1)}}]
    net.set_dataset_processing_params(class_names=["a", "b", "c"], image_processor=proc, iou=0.6, conf=0.0)
    rng = np.random.default_rng(5)
    images = [rng.integers(0, 256, s, dtype=np.uint8) for s in ((64, 50, 3), (40, 64, 3), (97, 31, 3), (23, 23, 3))]
    dev = net.predict(images, max_predictions=20, nms_top_k=100, fp16=False)
    pipe = net._get_pipeline(max_predictions=20, nms_top_k=100, fp16=False)
    steps = [pipe.image_processor.inverse_box_steps(pipe.image_processor.preprocess_batch([im], device=pipe.device)[1][0]) for im in images]
    assert all(len(s) == 2 and s[0][0] == 0.0 and s[1][0] == 1.0 for s in steps)  # un-pad, then un-scale
    assert len({s[1][1] for s in steps}) > 1                                       # (the images really differ in scale)
    monkeypatch.setattr(ComposeProcessing, "inverse_box_steps", lambda self, m: None)
    host = net.predict(images, max_predictions=20, nms_top_k=100, fp16=False)
    n = 0
    for a, b in zip(dev, host):
        pa, pb = a.prediction, b.prediction
        assert len(pa) == len(pb) and pa.image_shape == pb.image_shape
        assert np.array_equal(pa.bboxes_xyxy, pb.bboxes_xyxy) and pa.bboxes_xyxy.dtype == pb.bboxes_xyxy.dtype
        assert np.array_equal(pa.confidence, pb.confidence) and np.array_equal(pa.labels, pb.labels)
        n += len(pa)
    assert n > 0

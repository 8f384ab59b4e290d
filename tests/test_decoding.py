"""Pre-NMS decoding modules: oracle vs the reference's own YoloNASDecodingModule / PPYoloEDecodingModule (live, CPU),
product (post-prediction kernel in its no-suppression mode) vs the oracle - bit-exact rows and indices."""
import pytest
import torch

from oracle import decoding as odec
from oracle import ref_shim


def _case(B, L, C, seed, ties=False):
    g = torch.Generator().manual_seed(seed)
    xy = torch.rand(B, L, 2, generator=g) * 600
    wh = torch.rand(B, L, 2, generator=g) * 80 + 1
    boxes = torch.cat([xy, xy + wh], -1)
    # distinct class confidences by construction (torch.topk's order among EQUAL values is unspecified - random fp32 scores collide
    # more often than one would think: ~2 expected collisions among 8400 uniform values): background scores < 0.4, one class per anchor
    # gets 0.5 + a permutation of L distinct levels
    scores = torch.rand(B, L, C, generator=g) * 0.4
    cls = torch.randint(0, C, (B, L, 1), generator=g)
    lvl = torch.stack([torch.randperm(L, generator=g) for _ in range(B)]).float() / (2 * L) + 0.5
    scores.scatter_(2, cls, lvl[:, :, None])
    if ties:  # blocks of anchors with identical score rows: the tie rule (lower anchor index first) decides
        scores[:, 5:25] = scores[:, 5:6]
        scores[:, L // 2:L // 2 + 7] = scores[:, 3:4]
    return boxes, scores


@pytest.mark.skipif(not ref_shim.available(), reason="/root/reference not present (GPU box)")
@pytest.mark.parametrize("family", ["yolo_nas", "ppyoloe"])
def test_oracle_decoding_live(family):
    ref_shim.install()
    if family == "yolo_nas":
        from super_gradients.training.models.detection_models.yolo_nas.yolo_nas_variants import YoloNASDecodingModule as Ref
    else:
        from super_gradients.training.models.detection_models.pp_yolo_e.pp_yolo_e import PPYoloEDecodingModule as Ref
    for B, L, C, k in ((2, 300, 7, 50), (1, 8400, 80, 1000), (3, 64, 1, 64)):
        boxes, scores = _case(B, L, C, seed=L)
        rb, rs = Ref(k)(((boxes, scores), None))
        ob, os_, _ = odec.decode_topk(boxes, scores, k)
        assert torch.equal(rb, ob) and torch.equal(rs, os_)
        assert Ref(k).get_num_pre_nms_predictions() == k


@pytest.mark.parametrize("shape", [(2, 300, 7, 50), (2, 2100, 80, 1000), (1, 64, 1, 64), (2, 500, 5, 300)])
def test_decode_topk_vs_oracle(backend, shape):
    from super_gradients_amd import kernels as K

    B, L, C, k = shape
    boxes, scores = _case(B, L, C, seed=L + C, ties=(k == 300))
    ob, os_, oi = odec.decode_topk(boxes, scores, k)
    pb, ps, pi = K.decode_topk(boxes.to(backend), scores.to(backend), k)
    assert torch.equal(pi.cpu(), oi), "anchor order differs from the oracle"
    assert torch.equal(pb.cpu(), ob) and torch.equal(ps.cpu(), os_)


def test_decoding_modules_api(backend):
    from super_gradients_amd.training.models.detection_models.pp_yolo_e.pp_yolo_e import PPYoloEDecodingModule
    from super_gradients_amd.training.models.detection_models.yolo_nas.yolo_nas_variants import YoloNASDecodingModule

    boxes, scores = _case(2, 400, 6, seed=1)
    outputs = ((boxes.to(backend), scores.to(backend)), None)
    for cls in (YoloNASDecodingModule, PPYoloEDecodingModule):
        m = cls(num_pre_nms_predictions=100)
        assert m.get_num_pre_nms_predictions() == 100 and m.infer_total_number_of_predictions(outputs) == 400
        b, s = m(outputs)
        ob, os_, _ = odec.decode_topk(boxes, scores, 100)
        assert torch.equal(b.cpu(), ob) and torch.equal(s.cpu(), os_)
    with pytest.raises(ValueError):
        YoloNASDecodingModule(401)(outputs)       # torch.topk raises for k > number of anchors, too

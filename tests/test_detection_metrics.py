"""Validation metrics: prediction-to-target matching and AP.

  reference (live, through the shim)            -> oracle/detection_metrics.py and the product's host-side AP       CPU
  golden fixture tests/golden/detection_metrics.pt (from the reference) -> oracle, product matching kernel          CPU (emu) + GPU
Flags are booleans: bit-exact.  AP / precision / recall: 1e-6 (same float32 arithmetic, different association in a mean).
"""
import os

import numpy as np
import pytest
import torch

from oracle import detection_metrics as OD
from oracle import golden_util as G
from oracle import ref_shim

THR = torch.linspace(0.5, 0.95, 10)


def _case(seed, B=4, nmax_pred=40, ncls=5, size=200, normalized=False, crowd=True):
    """Clustered predictions around ground-truth boxes (so that IoUs straddle the thresholds), score ties, an image without
    predictions, an image without targets, crowd boxes."""
    g = np.random.RandomState(seed)
    targets, crowds, preds = [], [], []
    for b in range(B):
        k = 0 if b == 1 else g.randint(1, 7)
        gts = []
        for _ in range(k):
            cx, cy = g.uniform(30, size - 30, 2)
            w, h = g.uniform(15, 70, 2)
            c = g.randint(0, ncls)
            gts.append((c, cx, cy, w, h))
            targets.append([b, c, cx, cy, w, h])
        if crowd and b % 2 == 0:
            cx, cy, w, h = g.uniform(40, size - 40), g.uniform(40, size - 40), g.uniform(60, 120), g.uniform(60, 120)
            crowds.append([b, g.randint(0, ncls), cx, cy, w, h])
        n = 0 if b == 2 else g.randint(5, nmax_pred)
        rows = []
        for _ in range(n):
            if gts and g.rand() < 0.7:
                c, cx, cy, w, h = gts[g.randint(len(gts))]
                cx, cy = cx + g.normal(0, 4), cy + g.normal(0, 4)
                w, h = w * g.uniform(0.8, 1.25), h * g.uniform(0.8, 1.25)
                c = c if g.rand() < 0.85 else g.randint(0, ncls)
            else:
                cx, cy, w, h, c = g.uniform(0, size), g.uniform(0, size), g.uniform(10, 80), g.uniform(10, 80), g.randint(0, ncls)
            score = np.round(g.uniform(0.02, 1.0), 2)  # 2 decimals: score ties
            rows.append([cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2, score, c])
        rows.sort(key=lambda r: -r[4])
        preds.append(torch.tensor(rows, dtype=torch.float32).reshape(-1, 6) if rows else None)
    t = torch.tensor(targets, dtype=torch.float32).reshape(-1, 6)
    c = torch.tensor(crowds, dtype=torch.float32).reshape(-1, 6)
    if normalized:
        t[:, 2:] /= size
        c[:, 2:] /= size
    return preds, t, c, size


def _reference(preds, targets, crowd, size, top_k, denorm):
    ref_shim.install()
    import super_gradients.training.utils.detection_utils as D

    out = D.compute_detection_matching([None if p is None else p.clone() for p in preds], targets.clone(), size, size, iou_thresholds=THR,
                                       matching_strategy=D.IoUMatching(THR), crowd_targets=crowd.clone(), denormalize_targets=denorm, device="cpu", top_k=top_k)
    flat = [torch.cat(x, 0) for x in zip(*out)]
    met = D.compute_detection_metrics(*flat, device="cpu", score_threshold=0.1)
    return out, flat, met


CASES = [dict(seed=1, top_k=100, normalized=False), dict(seed=2, top_k=3, normalized=False), dict(seed=3, top_k=100, normalized=True),
         dict(seed=4, top_k=100, normalized=False, crowd=False)]


@pytest.mark.skipif(not ref_shim.available(), reason="/root/reference not present")
@pytest.mark.parametrize("case", CASES)
def test_oracle_and_host_ap_live(case):
    from super_gradients_amd.training.metrics import compute_detection_metrics

    preds, t, c, size = _case(case["seed"], normalized=case["normalized"], crowd=case.get("crowd", True))
    out, flat, met = _reference(preds, t, c, size, case["top_k"], case["normalized"])
    mine = OD.match_batch([None if p is None else p.numpy() for p in preds], t.numpy(), c.numpy(), size, size, THR.numpy(), case["top_k"], case["normalized"])
    for b, ((m, ig), ref) in enumerate(zip(mine, out)):
        assert np.array_equal(m, ref[0].numpy()), f"image {b}: matched flags differ"
        assert np.array_equal(ig, ref[1].numpy()), f"image {b}: ignore flags differ"
    o = OD.average_precision(*[x.numpy() for x in flat], score_threshold=0.1)
    p = compute_detection_metrics(*flat, score_threshold=0.1)
    for name, a, b_, r in (("ap", o["ap"], p[0], met[0]), ("precision", o["precision"], p[1], met[1]), ("recall", o["recall"], p[2], met[2]), ("f1", o["f1"], p[3], met[3])):
        assert np.allclose(a, r.numpy(), atol=1e-6), f"oracle {name}"
        assert torch.allclose(b_, r, atol=1e-6), f"product host {name}"
    assert abs(float(o["best_score_threshold"]) - float(met[5])) < 1e-6 and abs(float(p[5]) - float(met[5])) < 1e-6
    assert np.allclose(o["best_per_class"], met[6].numpy(), atol=1e-6) and torch.allclose(p[6], met[6], atol=1e-6)


def _fixture():
    return torch.load(os.path.join(G.GOLDEN_DIR, "detection_metrics.pt"), weights_only=False)


def test_oracle_golden():
    for rec in _fixture():
        case = rec["case"]
        preds, t, c, size = _case(case["seed"], normalized=case["normalized"], crowd=case.get("crowd", True))
        mine = OD.match_batch([None if p is None else p.numpy() for p in preds], t.numpy(), c.numpy(), size, size, THR.numpy(), case["top_k"], case["normalized"])
        for (m, ig), rm, ri in zip(mine, rec["matched"], rec["ignore"]):
            assert np.array_equal(m, rm.numpy()) and np.array_equal(ig, ri.numpy())
        flat = [np.concatenate([x[0] for x in mine]), np.concatenate([x[1] for x in mine]), rec["scores"].numpy(), rec["pred_cls"].numpy(), rec["target_cls"].numpy()]
        o = OD.average_precision(*flat, score_threshold=0.1)
        assert np.allclose(o["ap"], rec["ap"].numpy(), atol=1e-6) and np.allclose(o["recall"], rec["recall"].numpy(), atol=1e-6)


def test_product_matching_and_metrics_golden(backend):
    """The matching kernel (emu / MI355X) bit-exact against the reference's flags; DetectionMetrics.update/compute end to end."""
    from super_gradients_amd import kernels as K
    from super_gradients_amd.training.metrics import DetectionMetrics

    for rec in _fixture():
        case = rec["case"]
        preds, t, c, size = _case(case["seed"], normalized=case["normalized"], crowd=case.get("crowd", True))
        B = len(preds)
        P = max([len(p) for p in preds if p is not None] + [1])
        rows = torch.zeros(B, P, 6)
        counts = torch.zeros(B, dtype=torch.int32)
        for b, p in enumerate(preds):
            if p is not None:
                rows[b, : len(p)] = p
                counts[b] = len(p)
        m, ig = K.detection_match(rows.to(backend), counts.to(backend), t.to(backend), c.to(backend), THR, size, size, case["top_k"], case["normalized"])
        for b in range(B):
            n = int(counts[b])
            assert torch.equal(m[b, :n].cpu().bool(), rec["matched"][b]), f"seed {case['seed']} image {b}: matched"
            assert torch.equal(ig[b, :n].cpu().bool(), rec["ignore"][b]), f"seed {case['seed']} image {b}: ignore"
        metric = DetectionMetrics(num_cls=5, post_prediction_callback=None, normalize_targets=not case["normalized"], top_k_predictions=case["top_k"])  # reference semantics: normalize_targets=False means "targets come normalised, de-normalise them"
        half = B // 2  # two update() calls: batch-local image indices, accumulation
        for lo, hi in ((0, half), (half, B)):
            tt = t[(t[:, 0] >= lo) & (t[:, 0] < hi)].clone()
            tt[:, 0] -= lo
            cc = c[(c[:, 0] >= lo) & (c[:, 0] < hi)].clone()
            cc[:, 0] -= lo
            lst = [(p.to(backend) if p is not None else torch.zeros(0, 6, device=backend)) for p in preds[lo:hi]]
            metric.update(lst, tt.to(backend), inputs=torch.zeros(hi - lo, 3, size, size), crowd_targets=cc.to(backend))
        res = metric.compute()
        assert abs(res["mAP@0.50:0.95"] - float(rec["ap"].mean())) < 1e-6
        assert abs(res["Recall@0.50:0.95"] - float(rec["recall"].mean())) < 1e-6
        assert abs(res["Precision@0.50:0.95"] - float(rec["precision"].mean())) < 1e-6
        assert abs(res["Best_score_threshold"] - float(rec["best_score_threshold"])) < 1e-6

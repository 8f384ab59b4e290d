"""TEST INFRASTRUCTURE ONLY - shared by oracle/make_golden.py (run in the build container, against the real reference)
and tests/test_oracle_vs_reference.py / tests/test_golden_gpu.py (run anywhere, against the committed fixtures).

Weights are never stored in the fixtures (YOLO-NAS-S alone is 76 MB): both sides regenerate them with
`deterministic_fill`, which depends only on the state_dict key order, the tensor shapes and a seed - and the key order /
shapes are themselves part of what is being pinned (the reference's state_dict layout).
"""
import math
import os

import torch

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def deterministic_fill(module: torch.nn.Module, seed: int = 0) -> None:
    """Overwrite every state_dict entry with a seeded, well-conditioned value (in place)."""
    sd = module.state_dict()
    for i, (name, t) in enumerate(sd.items()):
        g = torch.Generator().manual_seed(seed * 100003 + i)
        if not t.dtype.is_floating_point:
            t.zero_()
            continue
        leaf = name.rsplit(".", 1)[-1]
        if leaf == "running_var":
            v = torch.empty(t.shape).uniform_(0.5, 1.5, generator=g)
        elif leaf == "running_mean":
            v = torch.randn(t.shape, generator=g) * 0.1
        elif t.dim() >= 2:  # conv / linear / conv-transpose weights: unit-gain fan-in scaling
            fan_in = t[0].numel() if t.dim() > 1 else 1
            v = torch.randn(t.shape, generator=g) / math.sqrt(max(fan_in, 1))
        elif leaf == "weight":  # BatchNorm gamma
            v = torch.empty(t.shape).uniform_(0.5, 1.5, generator=g)
        elif leaf == "alpha":
            v = 1.0 + torch.randn(t.shape, generator=g) * 0.1
        elif name.endswith("cls_pred.bias") or (".pred_cls." in name and leaf == "bias"):
            # the detection prior the reference initialises with (dfl_heads.py:98-100, -log((1-0.01)/0.01)): keeps the
            # classification loss - and with it the conditioning of the loss gradient - in its realistic regime
            v = -math.log(99.0) + torch.randn(t.shape, generator=g) * 0.1
        else:  # biases
            v = torch.randn(t.shape, generator=g) * 0.1
        t.copy_(v.to(t.dtype))


def seeded_input(batch, channels, size, seed):
    return torch.rand(batch, channels, size, size, generator=torch.Generator().manual_seed(seed))


def detection_targets(batch, size, seed, kmax=4, num_classes=80, empty_last=True):
    """[T,6] (img, class, cx, cy, w, h) pixels; the last image carries no box when empty_last (the reference's
    zero-padding + pad_gt_mask branch, ppyolo_loss.py:726-775)."""
    g = torch.Generator().manual_seed(seed)
    rows = []
    for b in range(batch):
        if empty_last and b == batch - 1 and batch > 1:
            continue
        k = int(torch.randint(1, kmax + 1, (1,), generator=g))
        for _ in range(k):
            cx, cy = (torch.rand(2, generator=g) * 0.8 + 0.1) * size
            w, h = torch.rand(2, generator=g) * (0.4 * size - 8) + 8
            cls = int(torch.randint(0, num_classes, (1,), generator=g))
            rows.append([b, cls, float(cx), float(cy), float(w), float(h)])
    return torch.tensor(rows, dtype=torch.float32).reshape(-1, 6)


# the fixed target tensor of the reference's own loss unit test (tests/unit_tests/ppyoloe_unit_test.py:59-72)
REFERENCE_UNIT_TEST_TARGETS = torch.tensor(
    [
        [0, 2, 40, 60, 100, 200], [0, 3, 100, 200, 100, 200], [0, 4, 200, 300, 100, 200], [0, 5, 300, 400, 100, 200],
        [0, 6, 400, 500, 100, 200], [1, 2, 40, 60, 100, 200], [1, 3, 100, 200, 100, 200], [1, 4, 200, 300, 100, 200],
        [2, 2, 40, 60, 100, 200], [2, 3, 100, 200, 100, 200],
    ]
).float()


def synthetic_predictions(batch, sizes, num_classes, reg_max, seed, make_anchors):
    """Random head outputs in the layout NDFLHeads returns (dfl_heads.py:237-245): (logits, distri, anchors, points,
    counts, strides).  `make_anchors(hw_list, strides)` is the anchor generator of the side being exercised."""
    g = torch.Generator().manual_seed(seed)
    strides = [8, 16, 32]
    anchors, points, counts, stride_t = make_anchors([(s, s) for s in sizes], strides)
    L = sum(counts)
    logits = torch.randn(batch, L, num_classes, generator=g) * 1.5 - 2.0
    distri = torch.randn(batch, L, 4 * (reg_max + 1), generator=g) * 1.2
    return logits, distri, anchors, points, counts, stride_t


def nms_cases():
    """Seeded candidate sets with the hazards SURVEY 8(c) lists: exact score ties, IoU exactly at the threshold,
    more than nms_top_k candidates, an image with no candidate, a >4000-coordinate class-aware case."""
    cases = []
    g = torch.Generator().manual_seed(7)

    def boxes(n, centres, spread, size):
        c = torch.rand(centres, 2, generator=g) * 500 + 50
        idx = torch.randint(0, centres, (n,), generator=g)
        ctr = c[idx] + torch.randn(n, 2, generator=g) * spread
        wh = torch.rand(n, 2, generator=g) * size + 8
        return torch.cat([ctr - wh / 2, ctr + wh / 2], 1)

    # 1: clustered boxes, scores quantised to 1/64 -> many exact ties
    B, L, C = 2, 600, 6
    bx = torch.stack([boxes(L, 12, 6.0, 80.0) for _ in range(B)])
    sc = (torch.rand(B, L, C, generator=g) * 64).floor() / 64 * (torch.rand(B, L, C, generator=g) < 0.25)
    cases.append(dict(name="ties", boxes=bx, scores=sc, score_threshold=0.1, nms_threshold=0.6, nms_top_k=1000, max_predictions=300))
    # 2: boxes on an integer lattice: IoU values hit simple fractions; threshold 0.5 is reached exactly (strict > keeps them)
    L = 256
    xy = torch.randint(0, 12, (1, L, 2), generator=g).float() * 8
    bx = torch.cat([xy, xy + 16], -1)
    sc = torch.rand(1, L, 3, generator=g) * (torch.rand(1, L, 3, generator=g) < 0.5)
    cases.append(dict(name="iou_at_threshold", boxes=bx, scores=sc, score_threshold=0.05, nms_threshold=0.5, nms_top_k=1000, max_predictions=300))
    # 3: more candidates than nms_top_k + second image without any candidate
    B, L, C = 2, 1500, 4
    bx = torch.stack([boxes(L, 30, 10.0, 120.0) for _ in range(B)])
    sc = torch.rand(B, L, C, generator=g)
    sc[1] = 0.0
    cases.append(dict(name="topk_and_empty", boxes=bx, scores=sc, score_threshold=0.3, nms_threshold=0.7, nms_top_k=400, max_predictions=100))
    # 4: class-aware, 1024 candidates after top-k (4*K = 4096 > 4000 -> torchvision takes its per-class loop on CPU)
    B, L, C = 1, 900, 5
    bx = torch.stack([boxes(L, 20, 8.0, 100.0) for _ in range(B)])
    sc = torch.rand(B, L, C, generator=g) * (torch.rand(B, L, C, generator=g) < 0.5)
    cases.append(dict(name="class_aware_large", boxes=bx, scores=sc, score_threshold=0.2, nms_threshold=0.65, nms_top_k=1024, max_predictions=300))
    return cases

"""Known-answer vectors for the two restatements whose third-party originals are absent here (torchvision's NMS, OpenCV's 8-bit
INTER_LINEAR resize): every expected value below is WORKED BY HAND from the published arithmetic - power-of-two coordinates, exact
fractions, the documented fixed-point formula - and written as a literal, so neither oracle/nms.c nor oracle/image.py is its own judge.
Both the CPU restatement and the HIP kernel (C ABI; `emu` = same kernel sources on the host emulation) are held to the literals.
These do not replace a pin against real torchvision / cv2 output (a20 and the rescale stay "parity unpinned" until
oracle/make_nms_golden.py runs somewhere those packages exist); they fix the corner semantics an implementation is most likely to get
wrong: strict `>` at the threshold, stable order among equal scores, 0/0 IoU of empty boxes, the 4000-element dispatch of batched_nms
with the rounding of the coordinate-offset trick, OpenCV's 11-bit weights / two-stage shifts / border clamps / exact-2x shortcut.
"""
import numpy as np
import pytest
import torch

from super_gradients_amd import kernels as K

from oracle import image as OI
from oracle import nms as onms


def _run_hip(boxes, scores, thr, backend, class_agnostic, top_k=2048, max_pred=2048):
    """boxes [L,4], scores [L,C] -> kept rows [n,6] of the post-prediction kernel (multi-label, score > 0.01)."""
    out, cnt, _, _ = K.nms(torch.tensor(boxes, dtype=torch.float32, device=backend)[None], torch.tensor(scores, dtype=torch.float32, device=backend)[None],
                           0.01, thr, top_k, max_pred, multi_label=True, class_mode=0 if class_agnostic else 3)
    return out[0, : int(cnt[0])].cpu().numpy()


def _run_oracle(boxes, scores, thr, class_agnostic, top_k=2048, max_pred=2048):
    res = onms.post_prediction(torch.tensor(boxes, dtype=torch.float32)[None], torch.tensor(scores, dtype=torch.float32)[None], score_threshold=0.01,
                               nms_threshold=thr, nms_top_k=top_k, max_predictions=max_pred, multi_label_per_box=True, class_agnostic_nms=class_agnostic)
    return res[0].numpy()


def _both(boxes, scores, thr, backend, class_agnostic=True):
    a, b = _run_oracle(boxes, scores, thr, class_agnostic), _run_hip(boxes, scores, thr, backend, class_agnostic)
    return a, b


def test_nms_known_answers(backend):
    f32 = np.float32
    # A. IoU exactly AT the threshold is kept (torchvision: suppress iff ovr > thr).  [0,0,4,4] vs [0,0,4,2]: inter 8, union 16 -> 0.5.
    boxes = [[0, 0, 4, 4], [0, 0, 4, 2]]
    scores = [[0.9], [0.8]]
    want_keep = np.array([[0, 0, 4, 4, f32(0.9), 0], [0, 0, 4, 2, f32(0.8), 0]], f32)
    for got in _both(boxes, scores, 0.5, backend):
        assert np.array_equal(got, want_keep), "IoU == threshold must NOT suppress"
    below = float(np.nextafter(f32(0.5), f32(0)))  # 0.5 - 2^-25: now 0.5 > thr
    for got in _both(boxes, scores, below, backend):
        assert np.array_equal(got, want_keep[:1]), "IoU just above the threshold must suppress"
    # B. equal scores: stable descending sort = original order.  b0-b1 and b1-b2 overlap by 1/3 (8 / (16 + 16 - 8)), b0-b2 are disjoint:
    # walking 0, 1, 2 keeps 0, drops 1, keeps 2 -> [0, 2]; walking the other way would give [2, 0].
    boxes = [[0, 0, 4, 4], [2, 0, 6, 4], [4, 0, 8, 4]]
    scores = [[0.5], [0.5], [0.5]]
    want = np.array([[0, 0, 4, 4, 0.5, 0], [4, 0, 8, 4, 0.5, 0]], f32)
    for got in _both(boxes, scores, 0.25, backend):
        assert np.array_equal(got, want), "equal scores: lower index first"
    # C. empty boxes: inter = 0, union = 0 -> 0/0 = NaN, and NaN > thr is false: two identical zero-area boxes are BOTH kept; a zero-area
    # box inside a big one has inter 0 -> IoU 0 -> kept
    boxes = [[1, 1, 1, 1], [1, 1, 1, 1], [0, 0, 4, 4]]
    scores = [[0.9], [0.8], [0.7]]
    want = np.array([[1, 1, 1, 1, f32(0.9), 0], [1, 1, 1, 1, f32(0.8), 0], [0, 0, 4, 4, f32(0.7), 0]], f32)
    for got in _both(boxes, scores, 0.1, backend):
        assert np.array_equal(got, want), "zero-area boxes: 0/0 IoU never suppresses"
    # D. a contained box: [0,0,8,8] vs [0,0,4,4]: inter 16, union 64 -> 0.25; thr 0.25 keeps, thr just below drops
    boxes = [[0, 0, 8, 8], [0, 0, 4, 4]]
    scores = [[0.6], [0.3]]
    for got in _both(boxes, scores, 0.25, backend):
        assert got.shape[0] == 2
    for got in _both(boxes, scores, float(np.nextafter(f32(0.25), f32(0))), backend):
        assert got.shape[0] == 1 and got[0, 4] == f32(0.6)


@pytest.mark.parametrize("n_candidates", [1000, 1001])
def test_batched_nms_dispatch_known_answer(backend, n_candidates):
    """torchvision.ops.batched_nms on CPU (ops/boxes.py): boxes.numel() <= 4000 -> ONE nms over boxes + class * (max_coordinate + 1);
    more -> one nms per class on the raw boxes.  The two differ through fp32 rounding of the shifted coordinates, which this case makes
    decisive by hand: a far box puts max_coordinate at 2^20, so class 2 is shifted by 2 * (2^20 + 1) = 2097154, where fp32 steps by 0.25.
      a = [0, 0, 1.1, 1] (score 0.9, class 2), b = [0, 0, 1, 1] (0.8, class 2): true IoU = 1 / 1.1 = 0.909.
      shifted: 2097154 + 1.1 -> 2097155.0 (nearest multiple of 0.25) = b's shifted corner: the boxes coincide, IoU = 1.
    Threshold 0.95: with 1000 candidates (numel 4000) b is suppressed, with 1001 candidates b survives.  Fillers: class-1 boxes 8 apart
    (integer coordinates stay exact under the class-1 shift 1048577), distinct scores below 0.5."""
    f32 = np.float32
    nfill = n_candidates - 3
    L, C = 3 + nfill, 3
    boxes = np.zeros((L, 4), f32)
    scores = np.zeros((L, C), f32)
    boxes[0], scores[0, 2] = [0, 0, 1.1, 1], 0.9
    boxes[1], scores[1, 2] = [0, 0, 1, 1], 0.8
    boxes[2], scores[2, 0] = [2.0 ** 20 - 1, 0, 2.0 ** 20, 1], 0.7
    for k in range(nfill):
        boxes[3 + k] = [8 * k, 100, 8 * k + 4, 104]
        scores[3 + k, 1] = 0.5 - 1e-4 * k
    thr = 0.95
    got_o = _run_oracle(boxes, scores, thr, class_agnostic=False)
    got_h = _run_hip(boxes, scores, thr, backend, class_agnostic=False)
    want_head = [[0, 0, f32(1.1), 1, f32(0.9), 2]] + ([] if n_candidates <= 1000 else [[0, 0, 1, 1, f32(0.8), 2]]) + [[f32(2.0 ** 20 - 1), 0, f32(2.0 ** 20), 1, f32(0.7), 0]]
    want = np.array(want_head + [[8 * k, 100, 8 * k + 4, 104, scores[3 + k, 1], 1] for k in range(nfill)], f32)
    assert got_o.shape == want.shape and np.array_equal(got_o, want), f"oracle: {n_candidates} candidates"
    assert got_h.shape == want.shape and np.array_equal(got_h, want), f"HIP kernel: {n_candidates} candidates"


def _hip_resize(img, h, w, backend):
    """uint8 [h0, w0, C] -> uint8 values of the device rescale stage (no padding, no standardisation)."""
    t = torch.from_numpy(img).to(backend)
    y = K.preprocess_u8([t], [(h, w, 0, 0)], h, w, torch.zeros(img.shape[2], dtype=torch.uint8, device=backend))
    return y[0, :, :, : img.shape[2]].cpu().numpy().astype(np.int64)


def test_inter_linear_u8_known_answers(backend):
    """OpenCV 8-bit INTER_LINEAR (imgproc/src/resize.cpp): source coordinate f = (d + 0.5) * scale - 0.5, weights round(f * 2048) as
    int16 (horizontal ones clamped at the borders, vertical ones not - the row index is clamped instead), horizontal pass in int32,
    vertical pass ((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2.  Worked by hand:
      row [0, 255] -> width 4: f = -0.25, 0.25, 0.75, 1.25 -> weights (2048, 0) | (1536, 512) | (512, 1536) | (2048 on pixel 1):
          S = 0 | 130560 | 391680 | 522240;  S >> 4 = 0 | 8160 | 24480 | 32640;  * 2048 >> 16 = 0 | 255 | 765 | 1020;  (+ 2) >> 2 = 0 | 64 | 191 | 255
      row [10, 20, 30] -> width 2 (scale 1.5): f = 0.25, 1.75 -> S = 10 * 1536 + 20 * 512 = 25600 | 20 * 512 + 30 * 1536 = 56320
          -> 1600 | 3520 -> 50 | 110 -> 13 | 28   (ideal 12.5 / 27.5: the +2 rounds half up)
      column [0, 255] -> height 4: vertical weights are NOT clamped: f = -0.25 -> rows (-1 -> 0, 0), weights (512, 1536) on the same row 0
          -> 0; then 64, 191; f = 1.25 -> rows (1, 2 -> 1), (1536 * 32640 >> 16) + (512 * 32640 >> 16) = 765 + 255 -> 255
      exact 2x reduction takes the INTER_AREA shortcut: [[1, 2], [3, 5]] -> (11 + 2) >> 2 = 3."""
    cases = [
        (np.array([[[0], [255]]], np.uint8), (1, 4), [[0, 64, 191, 255]]),
        (np.array([[[10], [20], [30]]], np.uint8), (1, 2), [[13, 28]]),
        (np.array([[[0]], [[255]]], np.uint8), (4, 1), [[0], [64], [191], [255]]),
        (np.array([[[1], [2]], [[3], [5]]], np.uint8), (1, 1), [[3]]),
        # both axes at once, 2x2 -> 4x4 of [[0, 255], [255, 0]]: rows of the horizontal pass S(r0) = [0, 130560, 391680, 522240] and its mirror;
        # corner (0,0): both source rows clamp to row 0 -> 0; (1,1): b = (1536, 512): (1536 * 8160 >> 16) + (512 * 24480 >> 16) = 191 + 191 -> 96
        (np.array([[[0], [255]], [[255], [0]]], np.uint8), (4, 4), [[0, 64, 191, 255], [64, 96, 159, 191], [191, 159, 96, 64], [255, 191, 64, 0]]),
    ]
    for img, (h, w), want in cases:
        want = np.array(want, np.int64)
        got_o = OI.resize_linear_u8(img, (h, w))[..., 0].astype(np.int64)
        assert np.array_equal(got_o, want), f"oracle {img[..., 0].tolist()} -> {(h, w)}: {got_o.tolist()}"
        got_h = _hip_resize(img, h, w, backend)[..., 0]
        assert np.array_equal(got_h, want), f"HIP {img[..., 0].tolist()} -> {(h, w)}: {got_h.tolist()}"

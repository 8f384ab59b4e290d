"""TEST INFRASTRUCTURE ONLY - never imported by the product package.

CPU restatement (numpy) of the reference's predict()-side image preparation, one function per reference pass:

  reverse_channels      training/processing/processing.py:230-257   image[..., ::-1]
  rescale               :510-589 -> transforms/utils.py:17-25       cv2.resize(image, (w, h), interpolation=cv2.INTER_LINEAR)
  pad                   :326-471 -> transforms/utils.py:79-158      np.pad of the uint8 image, centre / bottom-right / to-multiple coordinates
  standardize           :260-295                                    (image / max_value).astype(np.float32)      (float64 division)
  normalize             :298-323                                    (image - mean) / std                        (float32)
  permute               :205-227                                    np.ascontiguousarray(image.transpose(2, 0, 1))
  and the inverse box maps of postprocess_predictions (shift :344-350, rescale :578-589 -> transforms/utils.py:44-57,161-172).

PARITY STATUS.  Everything except `resize_linear_u8` is pinned against the reference's own classes (tests/test_predict.py runs them
through oracle/ref_shim.py when /root/reference is present, and against tests/golden/predict_processing.pt otherwise).
`resize_linear_u8` is **parity unpinned**: the arithmetic lives in OpenCV (requirements.txt: opencv-python>=4.5.1), which is neither
vendored by the reference nor installed here; the function restates OpenCV 4.x's published 8-bit INTER_LINEAR path
(modules/imgproc/src/resize.cpp: resizeGeneric_ with HResizeLinear<uchar, int, short, 2048> and the 8-bit VResizeLinear specialisation,
plus cv::resize's "exact 2x -> INTER_AREA" shortcut), and the HIP kernel is checked against this restatement only.
"""
import numpy as np


def _taps(dsize, ssize, clamp_weights):
    """source index and the two 11-bit fixed-point weights for every destination index along one axis"""
    scale = 1.0 / (float(dsize) / float(ssize))  # cv::resize: inv_scale = dsize / ssize (double); hal::resize: scale = 1 / inv_scale
    d = np.arange(dsize, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    if clamp_weights:
        lo = s < 0
        f[lo], s[lo] = 0.0, 0
        hi = s >= ssize - 1
        f[hi], s[hi] = 0.0, ssize - 1
    w0 = np.clip(np.rint((np.float32(1.0) - f) * np.float32(2048.0)), -32768, 32767).astype(np.int64)  # saturate_cast<short>: round half to even
    w1 = np.clip(np.rint(f * np.float32(2048.0)), -32768, 32767).astype(np.int64)
    return s, w0, w1


def resize_linear_u8(image: np.ndarray, target_shape) -> np.ndarray:
    """cv2.resize(image, (w, h), interpolation=cv2.INTER_LINEAR) for uint8 HWC images - restated, unpinned (module docstring)."""
    h0, w0 = image.shape[:2]
    h, w = int(target_shape[0]), int(target_shape[1])
    img = image.astype(np.int64)
    if (h, w) == (h0, w0):
        return image.copy()
    if w0 == 2 * w and h0 == 2 * h:  # exact 2x reduction: cv::resize switches INTER_LINEAR to INTER_AREA (2x2 mean, rounded)
        return ((img[0::2, 0::2] + img[0::2, 1::2] + img[1::2, 0::2] + img[1::2, 1::2] + 2) >> 2).astype(np.uint8)
    sx, a0, a1 = _taps(w, w0, True)
    sy, b0, b1 = _taps(h, h0, False)
    sx1 = np.minimum(sx + 1, w0 - 1)
    r0, r1 = np.clip(sy, 0, h0 - 1), np.clip(sy + 1, 0, h0 - 1)
    shape = (1, w) + (1,) * (img.ndim - 2)
    hpass = lambda rows: img[rows][:, sx] * a0.reshape(shape) + img[rows][:, sx1] * a1.reshape(shape)  # noqa: E731
    S0, S1 = hpass(r0), hpass(r1)
    vshape = (h, 1) + (1,) * (img.ndim - 2)
    out = (((b0.reshape(vshape) * (S0 >> 4)) >> 16) + ((b1.reshape(vshape) * (S1 >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


def center_padding(input_shape, output_shape):
    """(top, bottom, left, right): transforms/utils.py:79-94"""
    ph, pw = output_shape[0] - input_shape[0], output_shape[1] - input_shape[1]
    return ph // 2, ph - ph // 2, pw // 2, pw - pw // 2


def bottom_right_padding(input_shape, output_shape):
    """transforms/utils.py:97-106"""
    return 0, output_shape[0] - input_shape[0], 0, output_shape[1] - input_shape[1]


def auto_padding(input_shape, multiple):
    """processing.py:450-462: bottom / right padding up to the next multiple"""
    H = (input_shape[0] + multiple[0] - 1) // multiple[0] * multiple[0]
    W = (input_shape[1] + multiple[1] - 1) // multiple[1] * multiple[1]
    return 0, H - input_shape[0], 0, W - input_shape[1]


def pad(image, coords, pad_value):
    top, bottom, left, right = coords
    c = image.shape[2]
    vals = np.broadcast_to(np.asarray(pad_value, dtype=np.uint8), (c,))
    out = np.empty((image.shape[0] + top + bottom, image.shape[1] + left + right, c), dtype=image.dtype)
    out[...] = vals
    out[top:top + image.shape[0], left:left + image.shape[1]] = image
    return out


def standardize(image, max_value=255.0):
    return (image / max_value).astype(np.float32)


def normalize(image, mean, std):
    mean = np.array(mean).reshape((1, 1, -1)).astype(np.float32)
    std = np.array(std).reshape((1, 1, -1)).astype(np.float32)
    return (image - mean) / std


def longest_max_size(shape, output_shape):
    """-> (new_h, new_w, scale): processing.py:550-558"""
    h, w = shape
    s = min(output_shape[0] / h, output_shape[1] / w)
    if s != 1.0:
        return round(h * s), round(w * s), s
    return h, w, s


def shift_boxes(boxes, shift_w, shift_h):
    """transforms/utils.py:161-172"""
    b = boxes.copy()
    b[:, [0, 2]] += shift_w
    b[:, [1, 3]] += shift_h
    return b


def rescale_boxes(boxes, scale_factors):
    """transforms/utils.py:44-57: (sy, sx) factors, float32"""
    b = boxes.astype(np.float32, copy=True)
    sy, sx = scale_factors
    b[:, :4] *= np.array([[sx, sy, sx, sy]], dtype=b.dtype)
    return b

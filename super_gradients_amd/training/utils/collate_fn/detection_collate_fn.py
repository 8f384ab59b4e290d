"""Detection batch collation (reference: training/utils/collate_fn/detection_collate_fn.py:10-49) and its MI355X form.

DetectionCollateFN        the reference's host-side semantics: items (image, targets[Ni,5]) -> (images [N,C,H,W] float32,
                          targets [sum Ni, 6] with the batch index prepended).
DeviceDetectionCollateFN  the same contract with the pixel work moved to the GPU: the dataset's uint8 HWC images are
                          stacked as they are (a quarter of the fp32 bytes over PCIe), and ONE kernel (sgx_standardize_u8_hwc) produces the
                          standardized fp32 NHWC batch the first convolution reads - replacing DetectionStandardize / normalisation on the
                          host, the collate's moveaxis + float(), and the model's own NCHW->NHWC re-layout.  The returned image tensor is a
                          logical [N,C,H,W] view of that buffer, so `model(images)` is unchanged (models detect the layout: zero copies).
"""
from typing import List, Tuple, Union

import numpy as np
import torch

from .... import kernels as K


class DetectionCollateFN:
    def __init__(self):
        self.expected_item_names = ("image", "targets")

    def __call__(self, data) -> Tuple[torch.Tensor, torch.Tensor]:
        try:
            images_batch, labels_batch = list(zip(*data))
        except (ValueError, TypeError):
            raise ValueError(f"DetectionCollateFN expects items {self.expected_item_names}, got {type(data[0])}")
        return self._format_images(images_batch), self._format_targets(labels_batch)

    @staticmethod
    def _format_images(images_batch: List[Union[torch.Tensor, np.ndarray]]) -> torch.Tensor:
        stack = torch.stack([torch.as_tensor(img) for img in images_batch], 0)
        if stack.shape[3] == 3:
            stack = torch.moveaxis(stack, -1, 1).float()
        return stack

    @staticmethod
    def _format_targets(labels_batch: List[Union[torch.Tensor, np.ndarray]]) -> torch.Tensor:
        out = []
        for i, labels in enumerate(labels_batch):
            labels = torch.as_tensor(labels)
            out.append(torch.cat((labels.new_ones((labels.shape[0], 1)) * i, labels), dim=-1))
        return torch.cat(out, 0)


class DeviceDetectionCollateFN(DetectionCollateFN):
    """Items: (uint8 HWC image, targets [Ni,5]).  max_value / mean / std: the standardisation the reference recipe applies on the host
    (YOLO-NAS: DetectionStandardize(max_value=255); ImageNet-style models: / 255 then (x - mean) / std)."""

    def __init__(self, device="cuda", max_value: float = 255.0, mean=None, std=None, pad_to=None, pad_value=114, padding_mode: str = "bottom_right",
                 targets_format: str = "LABEL_CXCYWH", rescale_to=None):
        """rescale_to=(H, W): the reference's last image transform of the YOLO-NAS / YOLOX dataset recipes, DetectionPaddedRescale
        (transforms.py:945-975 -> transforms/utils.py:202-227: r = min(H / h, W / w), cv2.resize to (int(h * r), int(w * r)), bottom-right
        pad with pad_value, boxes * r), followed by DetectionStandardize, for the WHOLE ragged batch in ONE launch from the raw uint8 images
        (sgx_preprocess_u8_hwc; rescale arithmetic: the restated 8-bit INTER_LINEAR of csrc/image.hip).  Excludes pad_to.
        pad_to=(H, W): images of DIFFERENT sizes (each <= H x W) are padded on the device into one [N, C, H, W] batch - the reference's
        DetectionPadIfNeeded / DetectionPadToSize (transforms.py:846-941; padding_mode "center" or "bottom_right", pad_value as there) moved
        behind the PCIe transfer, one launch per image; boxes are shifted by the padding offsets like the transform does
        (transforms/utils.py:155-166).  targets_format: "LABEL_CXCYWH" (class, cx, cy, w, h - what the loss consumes) or "XYXY_LABEL"."""
        super().__init__()
        self.device = torch.device(device)
        self.max_value = float(max_value)
        if (mean is None) != (std is None):
            raise ValueError("mean and std go together")
        self._mean = None if mean is None else torch.as_tensor(mean, dtype=torch.float32)
        self._std = None if std is None else torch.as_tensor(std, dtype=torch.float32)
        if padding_mode not in ("center", "bottom_right"):
            raise ValueError(f"padding_mode {padding_mode!r}: 'center' or 'bottom_right'")
        if targets_format not in ("LABEL_CXCYWH", "XYXY_LABEL"):
            raise ValueError(f"targets_format {targets_format!r}: 'LABEL_CXCYWH' or 'XYXY_LABEL'")
        two = lambda v: None if v is None else ((int(v), int(v)) if isinstance(v, int) else (int(v[0]), int(v[1])))  # noqa: E731
        self.pad_to, self.rescale_to = two(pad_to), two(rescale_to)
        if self.pad_to is not None and self.rescale_to is not None:
            raise ValueError("rescale_to already pads to its size: give pad_to or rescale_to, not both")
        self.pad_value, self.padding_mode, self.targets_format = pad_value, padding_mode, targets_format

    def _padded_rescale(self, images_batch, labels_batch):
        H, W = self.rescale_to
        c = int(images_batch[0].shape[2])
        pv = [self.pad_value] * c if not hasattr(self.pad_value, "__len__") else list(self.pad_value)
        if len(pv) != c or any(int(v) != v or not 0 <= v <= 255 for v in pv):
            raise ValueError(f"pad_value {self.pad_value!r}: one uint8 value, or one per channel ({c})")
        pad = torch.tensor([int(v) for v in pv], dtype=torch.uint8).to(self.device)
        mean = None if self._mean is None else self._mean.to(self.device)
        std = None if self._std is None else self._std.to(self.device)
        dev_imgs, geometry, scaled = [], [], []
        for i, (img, labels) in enumerate(zip(images_batch, labels_batch)):
            img = torch.as_tensor(img)
            if img.dtype != torch.uint8 or img.dim() != 3:
                raise ValueError(f"DeviceDetectionCollateFN expects uint8 HWC images, got {img.dtype} {tuple(img.shape)}")
            h, w = int(img.shape[0]), int(img.shape[1])
            r = min(H / h, W / w)
            nh, nw = int(h * r), int(w * r)
            top, left = ((H - nh) // 2, (W - nw) // 2) if self.padding_mode == "center" else (0, 0)
            dev_imgs.append(img.to(self.device, non_blocking=True))
            geometry.append((nh, nw, top, left))
            t = torch.as_tensor(labels).clone().float()
            if t.numel():
                box = slice(1, 5) if self.targets_format == "LABEL_CXCYWH" else slice(0, 4)
                t[:, box] *= torch.tensor(r, dtype=torch.float32)  # _rescale_bboxes: float32 boxes times the float32 factor
                if self.targets_format == "LABEL_CXCYWH":
                    t[:, 1] += left
                    t[:, 2] += top
                else:
                    t[:, [0, 2]] += left
                    t[:, [1, 3]] += top
            scaled.append(t)
        batch = K.preprocess_u8(dev_imgs, geometry, H, W, pad, max_value=self.max_value, mean=mean, std=std)
        return K.nhwc_as_nchw_view(batch, c), self._format_targets(scaled).float().to(self.device, non_blocking=True)

    def _padded(self, images_batch, labels_batch):
        H, W = self.pad_to
        c = int(images_batch[0].shape[2])
        cp = (c + 3) // 4 * 4
        pv = torch.as_tensor([float(self.pad_value)] * c if not hasattr(self.pad_value, "__len__") else [float(v) for v in self.pad_value],
                             dtype=torch.float32)
        if pv.numel() != c:
            raise ValueError(f"A pad_value tuple ({self.pad_value} length should be {c} for an image with {c} channels")
        pv = pv.to(self.device)
        mean = None if self._mean is None else self._mean.to(self.device)
        std = None if self._std is None else self._std.to(self.device)
        batch = torch.empty(len(images_batch), H, W, cp, device=self.device, dtype=torch.float32)
        shifted = []
        for i, (img, labels) in enumerate(zip(images_batch, labels_batch)):
            img = torch.as_tensor(img)
            if img.dtype != torch.uint8 or img.dim() != 3:
                raise ValueError(f"DeviceDetectionCollateFN expects uint8 HWC images, got {img.dtype} {tuple(img.shape)}")
            h, w = int(img.shape[0]), int(img.shape[1])
            if h > H or w > W:
                raise ValueError(f"image {i} is {h}x{w}: larger than pad_to={self.pad_to} (rescale before padding, as the reference's transform chain does)")
            top, left = ((H - h) // 2, (W - w) // 2) if self.padding_mode == "center" else (0, 0)
            K.pad_standardize_u8(img.to(self.device, non_blocking=True), batch[i], top, left, pv, self.max_value, mean, std)
            t = torch.as_tensor(labels).clone().float()
            if t.numel():
                if self.targets_format == "LABEL_CXCYWH":
                    t[:, 1] += left
                    t[:, 2] += top
                else:
                    t[:, [0, 2]] += left
                    t[:, [1, 3]] += top
            shifted.append(t)
        return K.nhwc_as_nchw_view(batch, c), self._format_targets(shifted).float().to(self.device, non_blocking=True)

    def __call__(self, data) -> Tuple[torch.Tensor, torch.Tensor]:
        try:
            images_batch, labels_batch = list(zip(*data))
        except (ValueError, TypeError):
            raise ValueError(f"DeviceDetectionCollateFN expects items {self.expected_item_names}, got {type(data[0])}")
        if self.rescale_to is not None:
            return self._padded_rescale(images_batch, labels_batch)
        if self.pad_to is not None:
            return self._padded(images_batch, labels_batch)
        stack = torch.stack([torch.as_tensor(img) for img in images_batch], 0)
        if stack.dtype != torch.uint8 or stack.dim() != 4:
            raise ValueError(f"DeviceDetectionCollateFN expects uint8 HWC images (the dataset's native form), got {stack.dtype} {tuple(stack.shape)}")
        c = stack.shape[3]
        mean = None if self._mean is None else self._mean.to(self.device)
        std = None if self._std is None else self._std.to(self.device)
        y = K.standardize_u8(stack.to(self.device, non_blocking=True), self.max_value, mean, std)
        return K.nhwc_as_nchw_view(y, c), self._format_targets(labels_batch).float().to(self.device, non_blocking=True)

"""Shared helpers for the kernel parity tests (test infrastructure)."""
import os

import numpy as np
import pytest
import torch


def to_nhwc(x_nchw: torch.Tensor, device, ld_pix=None, c_off=0) -> torch.Tensor:
    """CPU NCHW tensor -> NHWC view on `device`; with ld_pix > C the view is a channel slice of a wider buffer."""
    n, c, h, w = x_nchw.shape
    if ld_pix is None:
        return x_nchw.permute(0, 2, 3, 1).contiguous().to(device)
    buf = torch.full((n, h, w, ld_pix), float("nan"), dtype=torch.float32, device=device)
    view = buf[..., c_off:c_off + c]
    view.copy_(x_nchw.permute(0, 2, 3, 1).to(device))
    return view


def empty_nhwc(n, h, w, c, device, ld_pix=None, c_off=0):
    if ld_pix is None:
        return torch.empty(n, h, w, c, device=device, dtype=torch.float32)
    buf = torch.full((n, h, w, ld_pix), float("nan"), dtype=torch.float32, device=device)
    return buf[..., c_off:c_off + c]


def to_nchw_cpu(x_nhwc: torch.Tensor) -> torch.Tensor:
    return x_nhwc.detach().cpu().permute(0, 3, 1, 2).contiguous()


def rel_err(a: torch.Tensor, b: torch.Tensor) -> float:
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def assert_close(a, b, tol=1e-5, what=""):
    assert tuple(a.shape) == tuple(b.shape), f"{what}: shape {tuple(a.shape)} vs {tuple(b.shape)}"
    assert torch.isfinite(a.detach().cpu()).all(), f"{what}: non-finite values"
    e = rel_err(a, b)
    assert e <= tol, f"{what}: max rel err {e:.3e} > {tol}"


def synthetic_targets(batch, seed=0, kmax=20, size=640, num_classes=80):
    """SURVEY 8(d) config-3 generator: per image k~U{1..kmax} boxes, class U, cx,cy~U(0.1,0.9)*size, w,h~U(16,256) clipped."""
    g = np.random.RandomState(seed)
    rows = []
    for b in range(batch):
        k = g.randint(1, kmax + 1)
        for _ in range(k):
            cx, cy = g.uniform(0.1 * size, 0.9 * size, 2)
            w, h = g.uniform(16, min(256, size * 0.4), 2)
            x1, y1, x2, y2 = max(cx - w / 2, 0), max(cy - h / 2, 0), min(cx + w / 2, size), min(cy + h / 2, size)
            rows.append([b, g.randint(0, num_classes), (x1 + x2) / 2, (y1 + y2) / 2, x2 - x1, y2 - y1])
    return torch.tensor(rows, dtype=torch.float32).reshape(-1, 6)

"""Batched torch formulation of the affine glue (reference: train.py:888-916), used by the
tests as an independent statement of the same maths: rotate / crop+resize / paste+resize
expressed as ``F.grid_sample`` calls.  Not part of the product (the trainer calls the HIP
kernels ``mvf_affine_transform_fwd`` / ``mvf_affine_restore_fwd/bwd``)."""
import math

import torch
import torch.nn.functional as F


def rotate_bilinear(img, angle_deg):
    """Rotate every image of a batch counter-clockwise by its own angle about the centre,
    bilinear, zero fill -- torchvision ``functional.rotate(img, angle, interpolation=2)``
    semantics (inverse-mapped pixel-centre grid, ``align_corners=False``), which the
    reference calls once per sample (train.py:898, 915).  torchvision is absent on both
    boxes; the oracle's rotate, which this formulation is compared with, is pinned to PIL's
    ``Image.rotate(angle, BILINEAR)`` -- the reference's own loader op -- in tests/test_pil_pins.py."""
    B, _, H, W = img.shape
    a = angle_deg.reshape(B).to(img.dtype) * (math.pi / 180.0)
    cos, sin = torch.cos(a).view(B, 1, 1), torch.sin(a).view(B, 1, 1)
    xs = torch.arange(W, device=img.device, dtype=img.dtype).view(1, 1, W) + 0.5 - W / 2.0
    ys = torch.arange(H, device=img.device, dtype=img.dtype).view(1, H, 1) + 0.5 - H / 2.0
    sx = (cos * xs - sin * ys) / (0.5 * W)
    sy = (sin * xs + cos * ys) / (0.5 * H)
    grid = torch.stack([sx, sy], -1)
    return F.grid_sample(img, grid, mode="bilinear", padding_mode="zeros", align_corners=False)


def _box_fields(box, dtype):
    b = box.to(dtype)
    return b[:, 0].view(-1, 1, 1), b[:, 1].view(-1, 1, 1), b[:, 2].view(-1, 1, 1), b[:, 3].view(-1, 1, 1)


def crop_resize_bilinear(img, box):
    """img[b, :, y0:y0+h, x0:x0+w] resized to the full [H,W] (bilinear,
    align_corners=False) for a per-sample integer box (x0,y0,w,h) -- batched form of
    train.py:899-900."""
    B, _, H, W = img.shape
    x0, y0, w, h = _box_fields(box, img.dtype)
    j = torch.arange(W, device=img.device, dtype=img.dtype).view(1, 1, W)
    i = torch.arange(H, device=img.device, dtype=img.dtype).view(1, H, 1)
    sx = x0 + torch.minimum(torch.clamp((j + 0.5) * (w / W) - 0.5, min=0.0), w - 1)
    sy = y0 + torch.minimum(torch.clamp((i + 0.5) * (h / H) - 0.5, min=0.0), h - 1)
    grid = torch.stack([(2 * sx / (W - 1) - 1).expand(B, H, W), (2 * sy / (H - 1) - 1).expand(B, H, W)], -1)
    return F.grid_sample(img, grid, mode="bilinear", padding_mode="border", align_corners=True)


def paste_resized(img, box):
    """Zeros canvas [H,W] holding ``img`` resized to (h,w) at (x0,y0) -- batched form of
    train.py:912-914."""
    B, _, H, W = img.shape
    x0, y0, w, h = _box_fields(box, img.dtype)
    X = torch.arange(W, device=img.device, dtype=img.dtype).view(1, 1, W)
    Y = torch.arange(H, device=img.device, dtype=img.dtype).view(1, H, 1)
    inside = ((X >= x0) & (X < x0 + w) & (Y >= y0) & (Y < y0 + h)).unsqueeze(1).to(img.dtype)
    sx = torch.clamp((X - x0 + 0.5) * (W / w) - 0.5, 0.0, W - 1.0)
    sy = torch.clamp((Y - y0 + 0.5) * (H / h) - 0.5, 0.0, H - 1.0)
    grid = torch.stack([(2 * sx / (W - 1) - 1).expand(B, H, W), (2 * sy / (H - 1) - 1).expand(B, H, W)], -1)
    return F.grid_sample(img, grid, mode="bilinear", padding_mode="border", align_corners=True) * inside

"""CPU: the oracle's affine_transform / affine_restore against the reference's per-sample
formulation (reference: train.py:888-922) evaluated with torch itself.

torchvision is absent on both boxes, so ``functional.rotate(img, angle, interpolation=2)`` is
restated here from its published algorithm (torchvision 0.12 ``_get_inverse_affine_matrix`` +
``_gen_affine_grid`` + ``grid_sample(align_corners=False, padding_mode="zeros")``): parity of
the rotate step is UNPINNED; slicing, paste and ``F.interpolate`` are torch's own.
"""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import affine_case
from oracle import oracle as O


def tv_rotate(img, angle):
    """torchvision.transforms.functional.rotate(img[1,C,H,W], angle, interpolation=BILINEAR)."""
    _, _, oh, ow = img.shape
    rot = math.radians(-angle)           # F.rotate inverts the sign before building the matrix
    a, b, c, d = math.cos(rot), -math.sin(rot), math.sin(rot), math.cos(rot)
    theta = torch.tensor([d, -b, 0.0, -c, a, 0.0], dtype=img.dtype).reshape(1, 2, 3)
    base = torch.empty(1, oh, ow, 3, dtype=img.dtype)
    base[..., 0].copy_(torch.linspace(-ow * 0.5 + 0.5, ow * 0.5 + 0.5 - 1, steps=ow))
    base[..., 1].copy_(torch.linspace(-oh * 0.5 + 0.5, oh * 0.5 + 0.5 - 1, steps=oh).unsqueeze_(-1))
    base[..., 2].fill_(1)
    rescaled = theta.transpose(1, 2) / torch.tensor([0.5 * ow, 0.5 * oh], dtype=img.dtype)
    grid = base.view(1, oh * ow, 3).bmm(rescaled).view(1, oh, ow, 2)
    return F.grid_sample(img, grid, mode="bilinear", padding_mode="zeros", align_corners=False)


def ref_affine_transform(img, angle, box):
    """train.py:888-902 with the .item() loop kept."""
    out = []
    H, W = img.shape[2:]
    for b in range(img.shape[0]):
        x0, y0, w, h = (int(v) for v in box[b])
        t = tv_rotate(img[b:b + 1], float(angle[b]))
        t = t[:, :, y0:y0 + h, x0:x0 + w]
        out.append(F.interpolate(t, [H, W], mode="bilinear", align_corners=False))
    return torch.cat(out, 0)


def ref_affine_restore(depth, angle, box, ratio):
    """depth_restore of train.py:909-916."""
    out = []
    H, W = depth.shape[2:]
    for b in range(depth.shape[0]):
        x0, y0, w, h = (int(v) for v in box[b])
        tmp = F.interpolate(depth[b:b + 1], [h, w], mode="bilinear", align_corners=False)
        canvas = torch.zeros((1, depth.shape[1], H, W), dtype=depth.dtype)
        canvas[:, :, y0:y0 + h, x0:x0 + w] = tmp
        canvas = tv_rotate(canvas, -float(angle[b]))
        out.append(canvas * float(ratio[b]))
    return torch.cat(out, 0)


@pytest.mark.parametrize("shape", [(2, 3, 32, 64), (3, 1, 48, 96), (1, 3, 192, 640)])
def test_affine_transform_matches_torch(shape):
    x, angle, box, _ = affine_case(11, *shape)
    want = ref_affine_transform(torch.from_numpy(x), angle, box).numpy()
    got = O.affine_transform(x, angle, box)
    # sample positions are fp32 values up to W (ulp 6e-5 px at 640): the two evaluation orders
    # of the rotation differ by ~1e-4 px, i.e. 1e-4 of the value range on i.i.d. data
    assert np.max(np.abs(got - want)) <= 2e-4


@pytest.mark.parametrize("shape", [(2, 1, 32, 64), (3, 1, 48, 96), (1, 1, 192, 640)])
def test_affine_restore_matches_torch(shape):
    x, angle, box, ratio = affine_case(12, *shape, lo=0.1, hi=100.0)
    d = torch.from_numpy(x).requires_grad_(True)
    want = ref_affine_restore(d, angle, box, ratio)
    got = O.affine_restore(x, angle, box, ratio)
    assert np.max(np.abs(got - want.detach().numpy())) <= 2e-4 * 200.0
    g = np.random.default_rng(5).standard_normal(x.shape).astype(np.float32)
    want.backward(torch.from_numpy(g))
    gd = O.affine_restore_bwd(g, angle, box, ratio)
    assert np.max(np.abs(gd - d.grad.numpy())) <= 2e-4 * np.max(np.abs(d.grad.numpy()))


def test_affine_degenerate_cases():
    """angle 0 + full box = identity; a 1x1 box spreads one value."""
    x = np.random.default_rng(3).random((2, 1, 16, 24)).astype(np.float32)
    angle = np.zeros(2, np.float32)
    box = np.array([[0, 0, 24, 16]] * 2, np.int32)
    ratio = np.ones(2, np.float32)
    assert np.max(np.abs(O.affine_transform(x, angle, box) - x)) <= 1e-6
    assert np.max(np.abs(O.affine_restore(x, angle, box, ratio) - x)) <= 1e-6
    box1 = np.array([[5, 4, 1, 1]] * 2, np.int32)
    out = O.affine_transform(x, angle, box1)
    assert np.allclose(out, x[:, :, 4:5, 5:6], atol=1e-6)

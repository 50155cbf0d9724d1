"""On-device data augmentation: the pixel work of ``MonoDataset.__getitem__ / preprocess``
(reference: datasets/mono_dataset.py:102-184, 214-256) moved off the host.

The reference decides per item, on the host, whether to flip and whether to colour-jitter
(each with probability 1/2), draws ONE torchvision ``ColorJitter`` parameter set per item
(brightness / contrast / saturation in [0.8, 1.2], hue in [-0.1, 0.1], the four adjustments
in a random order) and applies it to every frame with PIL; with ``use_affine`` it also builds
the rotated / cropped / resized views.  Here the host only draws those few numbers
(``draw_params``); the frames are flipped and jittered by ``mvf_color_jitter`` and the affine
views are produced by ``mvf_affine_transform_fwd`` (the same op ``Trainer.affine_transform``
applies to the teacher frames, train.py:888-902) -- all on the device, for the whole batch.
torchvision is on neither box: the jitter restates its published float-tensor algorithm
(the kernel is checked against the oracle's restatement; the oracle against PIL's ImageEnhance / HSV
implementation -- torchvision's PIL backend, which the reference's loader runs -- to within uint8
quantisation: tests/test_pil_pins.py).
"""
from __future__ import annotations

import numpy as np
import torch

from . import ops

FRAMES = (-1, 0, 1)


def draw_params(rng, n=1):
    """The random decisions of mono_dataset.py:214-216, 253-256 and ColorJitter.get_params for n
    items from a ``numpy.random.Generator`` -> dict of arrays (factors [n,4] = brightness,
    contrast, saturation, hue; order [n,4]; apply, flip [n])."""
    factors = np.stack([rng.uniform(0.8, 1.2, n), rng.uniform(0.8, 1.2, n), rng.uniform(0.8, 1.2, n),
                        rng.uniform(-0.1, 0.1, n)], 1).astype(np.float32)
    order = np.stack([rng.permutation(4) for _ in range(n)], 0).astype(np.int32)
    return {"aug_factors": factors, "aug_order": order,
            "aug_apply": (rng.random(n) > 0.5).astype(np.int32), "aug_flip": (rng.random(n) > 0.5).astype(np.int32)}


def augment_on_device(inputs, use_affine=True):
    """Fill ("color", f, 0) (flipped), ("color_aug", f, 0) and -- with ``use_affine`` --
    ("color_affine", f, 0), ("color_affine_aug", f, 0) from the raw frames and the per-sample
    draw, in place; returns ``inputs``.  Expects ("color", f, 0) raw device tensors and
    aug_factors / aug_order / aug_apply / aug_flip (+ angle, box when use_affine)."""
    frames = [inputs[("color", f, 0)] for f in FRAMES]
    B = frames[0].shape[0]
    dev = frames[0].device
    fac = inputs["aug_factors"].to(dev, torch.float32).reshape(B, 4)
    order = inputs["aug_order"].to(dev, torch.int32).reshape(B, 4)
    apply = inputs["aug_apply"].to(dev, torch.int32).reshape(B)
    flip = inputs["aug_flip"].to(dev, torch.int32).reshape(B)
    stacked = torch.stack(frames, 1).flatten(0, 1)                 # [B*3,3,H,W], frame-minor
    raw, aug = ops.color_jitter(stacked, fac, order, apply, flip, frames=len(FRAMES), want_raw=True)
    raw, aug = raw.view(B, len(FRAMES), *raw.shape[1:]), aug.view(B, len(FRAMES), *aug.shape[1:])
    for i, f in enumerate(FRAMES):
        inputs[("color", f, 0)] = raw[:, i]
        inputs[("color_aug", f, 0)] = aug[:, i]
    if use_affine:
        never = torch.zeros_like(flip)
        aff = ops.affine_transform(raw.flatten(0, 1), inputs["angle"].to(dev).reshape(B, 1).expand(B, len(FRAMES)).reshape(-1),
                                   inputs["box"].to(dev).reshape(B, 1, 4).expand(B, len(FRAMES), 4).reshape(-1, 4))
        _, aff_aug = ops.color_jitter(aff, fac, order, apply, never, frames=len(FRAMES), want_raw=False)
        aff, aff_aug = aff.view_as(raw), aff_aug.view_as(raw)
        for i, f in enumerate(FRAMES):
            inputs[("color_affine", f, 0)] = aff[:, i]
            inputs[("color_affine_aug", f, 0)] = aff_aug[:, i]
    return inputs

"""CPU: two restatements whose arithmetic lives in torchvision (absent on both boxes) pinned to
PIL, which is present here and is what the reference's data pipeline itself executes:

* ColorJitter -- the reference applies torchvision's ColorJitter to PIL images
  (datasets/mono_dataset.py:214-216, 253-256), i.e. torchvision's PIL backend:
  ``ImageEnhance.Brightness / Contrast / Color(img).enhance(f)`` and, for hue, an HSV round trip
  with ``h += uint8(hue * 255)`` (torchvision/transforms/functional_pil.py, torchvision 0.12 as the
  reference's README pins it).  The oracle restates the float-tensor form of the same four
  adjustments (the device kernel follows the oracle); PIL works on uint8, so the bar is its
  quantisation: <= 1.5 levels of 255 for brightness / contrast / saturation, a few levels for hue
  (PIL's HSV is 8-bit per channel and the shift is truncated to 1/255 of the circle).
* rotate -- the reference rotates the affine views and masks with ``Image.rotate(angle,
  BILINEAR)`` (datasets/mono_dataset.py:95-99, 147-149, 166-167) and the teacher frames with
  torchvision's tensor ``rotate`` (train.py:900), assuming both agree.  The oracle's rotate
  (mvf_affine_transform_fwd follows it) against PIL's on interior pixels: every difference is
  below one level of 255, i.e. PIL's truncation to uint8 alone.

Negative controls show the bars have power (flipped hue sign, flipped angle, 1-px centre error)."""
import numpy as np
import pytest

from oracle import oracle as O

PIL = pytest.importorskip("PIL")
from PIL import Image, ImageEnhance, ImageFilter  # noqa: E402


def _image(rng, H=48, W=64, t=0, noise=0.1):
    yy, xx = np.mgrid[0:H, 0:W]
    base = np.stack([0.5 + 0.4 * np.sin(xx / (5 + 3 * c + t % 4) + c + t) * np.cos(yy / (7 + c)) for c in range(3)], 0)
    base = np.clip(base + noise * rng.standard_normal(base.shape), 0, 1)
    return np.round(base * 255).astype(np.uint8)


def _pil_jitter(u8, factors, order):
    """torchvision's PIL backend, op by op (functional_pil.adjust_brightness / _contrast / _saturation / _hue)"""
    im = Image.fromarray(u8.transpose(1, 2, 0), "RGB")
    b, c, s, h = [float(v) for v in factors]
    for fn in order:
        if fn == 0:
            im = ImageEnhance.Brightness(im).enhance(b)
        elif fn == 1:
            im = ImageEnhance.Contrast(im).enhance(c)
        elif fn == 2:
            im = ImageEnhance.Color(im).enhance(s)
        else:
            hh, ss, vv = im.convert("HSV").split()
            np_h = (np.array(hh, dtype=np.uint8).astype(np.int32) + int(h * 255)).astype(np.uint8)   # uint8 wrap-around
            im = Image.merge("HSV", (Image.fromarray(np_h, "L"), ss, vv)).convert("RGB")
    return np.asarray(im).transpose(2, 0, 1).astype(np.float32)


def _oracle_jitter(u8, factors, order):
    f = np.asarray(factors, np.float32)[None]
    o = np.asarray(order, np.int32)[None]
    _, aug = O.color_jitter((u8 / 255.0).astype(np.float32)[None], f, o, np.array([1], np.int32), np.array([0], np.int32))
    return aug[0] * 255.0


def test_color_jitter_single_adjustments_vs_pil_backend():
    rng = np.random.default_rng(1)
    u8 = _image(rng)
    for op, f in [(0, 1.2), (0, 0.8), (1, 1.2), (1, 0.8), (2, 1.2), (2, 0.8)]:
        fac = [1.0, 1.0, 1.0, 0.0]
        fac[op] = f
        d = np.abs(_oracle_jitter(u8, fac, [0, 1, 2, 3]) - _pil_jitter(u8, fac, [op]))
        assert d.max() <= 1.5 and d.mean() <= 0.6, (op, f, d.max(), d.mean())
    for h in (0.0, 0.05, -0.05, 0.1, -0.1):
        fac = [1.0, 1.0, 1.0, h]
        d = np.abs(_oracle_jitter(u8, fac, [0, 1, 2, 3]) - _pil_jitter(u8, fac, [3]))
        assert d.mean() <= 1.3 and np.percentile(d, 99) <= 9.0, (h, d.mean(), np.percentile(d, 99))
    # power: the opposite hue shift is nowhere near
    d = np.abs(_oracle_jitter(u8, [1, 1, 1, -0.08], [0, 1, 2, 3]) - _pil_jitter(u8, [1, 1, 1, 0.08], [3]))
    assert d.mean() > 10.0


def test_color_jitter_random_draws_vs_pil_backend():
    """The per-item draw of mono_dataset.py (factors in [0.8, 1.2] / [-0.1, 0.1], random order)."""
    rng = np.random.default_rng(0)
    means = []
    for t in range(40):
        u8 = _image(rng, t=t)
        fac = [rng.uniform(0.8, 1.2), rng.uniform(0.8, 1.2), rng.uniform(0.8, 1.2), rng.uniform(-0.1, 0.1)]
        order = rng.permutation(4)
        d = np.abs(_oracle_jitter(u8, fac, order) - _pil_jitter(u8, fac, order))
        means.append(d.mean())
        assert d.mean() <= 2.5 and np.percentile(d, 99) <= 12.0, (t, fac, order, d.mean(), np.percentile(d, 99))
        # power: not applying the jitter at all is far away whenever the draw is not tiny
        if max(abs(fac[0] - 1), abs(fac[1] - 1), abs(fac[2] - 1), 2 * abs(fac[3])) > 0.1:
            assert np.abs(u8.astype(np.float32) - _pil_jitter(u8, fac, order)).mean() > 1.5 * d.mean()
    assert np.mean(means) <= 1.8


def test_rotate_of_the_affine_glue_vs_pil_rotate():
    rng = np.random.default_rng(2)
    H, W = 64, 96
    u8 = _image(rng, H, W, noise=0.0)
    x = (u8 / 255.0).astype(np.float32)[None]
    full = np.array([[0, 0, W, H]], np.int64)          # crop + resize of the glue are the identity for this box
    for angle in (5.0, -5.0, 2.5, -3.7, 0.8):
        out = O.affine_transform(x, np.array([[angle]], np.float32), full)[0] * 255.0
        pil = Image.fromarray(u8.transpose(1, 2, 0), "RGB").rotate(angle, resample=Image.BILINEAR, expand=False)
        ref = np.asarray(pil).transpose(2, 0, 1).astype(np.float32)
        ones = Image.new("L", (W, H), 255).rotate(angle, resample=Image.BILINEAR, expand=False)
        inner = np.asarray(ones.filter(ImageFilter.MinFilter(5))) == 255      # away from the zero fill
        inner[:2], inner[-2:], inner[:, :2], inner[:, -2:] = False, False, False, False   # (MinFilter pads with the edge)
        d = np.abs(out - ref)[:, inner]
        assert inner.sum() > 0.8 * H * W
        # PIL truncates to uint8: the difference is its quantisation alone (uniform in [0, 1))
        assert d.max() <= 1.01 and d.mean() <= 0.6, (angle, d.mean(), d.max())
        # power: the opposite angle, or a one-pixel centre error, is far outside the bar
        wrong = O.affine_transform(x, np.array([[-angle]], np.float32), full)[0] * 255.0
        if abs(angle) >= 2:
            assert np.abs(wrong - ref)[:, inner].mean() > 10.0
        shifted = O.affine_transform(np.roll(x, 1, axis=3), np.array([[angle]], np.float32), full)[0] * 255.0
        assert np.abs(shifted - ref)[:, inner].mean() > 2.5

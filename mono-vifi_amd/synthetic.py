"""Deterministic synthetic KITTI-shaped inputs for the view-synthesis hot path.

There are no datasets on either box, so every test, fixture and benchmark draws
its inputs from here.  Everything is generated with ``numpy.random.default_rng``
(PCG64) so that the GPU box regenerates exactly the tensors the golden fixtures
were captured on without the reference being present.

The shapes and value ranges follow the batch-dict contract of the reference
loader (reference: datasets/mono_dataset.py:189-279, datasets/kitti_dataset.py:23-26):
colour images fp32 in [0, 1], ``K`` = normalised KITTI intrinsics scaled by
(W, H), ``inv_K = numpy.linalg.pinv(K)`` in fp32 (so its off-diagonals carry
~1e-11 noise and must be treated as a general 3x3).
"""
from __future__ import annotations

import numpy as np

# normalised KITTI intrinsics (reference: datasets/kitti_dataset.py:23-26)
_K_NORM = np.array([[0.58, 0, 0.5, 0],
                    [0, 1.92, 0.5, 0],
                    [0, 0, 1, 0],
                    [0, 0, 0, 1]], dtype=np.float32)


def kitti_intrinsics(batch, height, width):
    """(K, inv_K), each [B,4,4] fp32, built like mono_dataset.py:243-252."""
    K = _K_NORM.copy()
    K[0, :] *= width
    K[1, :] *= height
    inv_K = np.linalg.pinv(K)
    K = np.ascontiguousarray(np.broadcast_to(K, (batch, 4, 4))).astype(np.float32)
    inv_K = np.ascontiguousarray(np.broadcast_to(inv_K, (batch, 4, 4))).astype(np.float32)
    return K, inv_K


def _box3(img):
    """3x3 box low-pass (edge replicate) so SSIM windows are not degenerate."""
    p = np.pad(img, ((0, 0), (0, 0), (1, 1), (1, 1)), mode="edge")
    acc = np.zeros_like(img, dtype=np.float32)
    for dy in range(3):
        for dx in range(3):
            acc += p[:, :, dy:dy + img.shape[2], dx:dx + img.shape[3]]
    return (acc / np.float32(9.0)).astype(np.float32)


def triplet_images(rng, batch, height, width, shift=2, noise=0.01):
    """Target frame and two neighbours: shifted crops of one canvas + 1 % noise.

    Returns (tgt, [src_m1, src_p1]), each [B,3,H,W] fp32 in [0,1].  The shifted
    copies make min-reprojection / auto-masking pick a realistic mix of
    candidates.
    """
    pad = 2 * shift
    canvas = rng.random((batch, 3, height + 2 * pad, width + 2 * pad), dtype=np.float32)
    canvas = _box3(_box3(canvas))
    # stretch contrast back to ~[0,1] after the low-pass
    canvas = np.clip((canvas - np.float32(0.5)) * np.float32(3.0) + np.float32(0.5), 0, 1).astype(np.float32)

    def crop(dx, dy):
        return np.ascontiguousarray(
            canvas[:, :, pad + dy:pad + dy + height, pad + dx:pad + dx + width])

    tgt = crop(0, 0)
    srcs = []
    for dx in (-shift, shift):
        s = crop(dx, 0) + np.float32(noise) * rng.standard_normal(
            (batch, 3, height, width), dtype=np.float32)
        srcs.append(np.clip(s, 0, 1).astype(np.float32))
    return tgt, srcs


def smooth_field(rng, batch, height, width, cells=(6, 20), lo=0.05, hi=0.95):
    """Piecewise-smooth field in [lo, hi]: a coarse random grid, bilinearly upsampled, plus a
    vertical ramp (near ground at the bottom of the image) -- the statistics of a depth
    network's sigmoid output, for which neighbouring pixels sample neighbouring source
    positions (coalesced bilinear taps)."""
    gh, gw = cells
    coarse = rng.random((batch, gh + 1, gw + 1)).astype(np.float32)
    ys = np.linspace(0, gh, height, dtype=np.float32)
    xs = np.linspace(0, gw, width, dtype=np.float32)
    y0 = np.minimum(ys.astype(np.int64), gh - 1)
    x0 = np.minimum(xs.astype(np.int64), gw - 1)
    fy = (ys - y0)[None, :, None]
    fx = (xs - x0)[None, None, :]
    c00 = coarse[:, y0][:, :, x0]
    c01 = coarse[:, y0][:, :, x0 + 1]
    c10 = coarse[:, y0 + 1][:, :, x0]
    c11 = coarse[:, y0 + 1][:, :, x0 + 1]
    f = (c00 * (1 - fy) * (1 - fx) + c01 * (1 - fy) * fx + c10 * fy * (1 - fx) + c11 * fy * fx)
    ramp = np.linspace(0.0, 1.0, height, dtype=np.float32)[None, :, None]
    f = 0.6 * f + 0.4 * ramp
    return (lo + (hi - lo) * f).astype(np.float32)[:, None]


def unit_inputs(seed, batch, height, width, num_src=2, pose_scale=0.01,
                with_mask=False, disp_lo=0.0, disp_hi=1.0, disp_mode="noise"):
    """All tensors one hot-path *unit* consumes (1 target, ``num_src`` sources).

    ``disp_mode``: "noise" = i.i.d. uniform disparity (adversarial for
    the bilinear gather: neighbouring pixels sample positions up to ~40 px apart), "smooth" =
    a piecewise-smooth field like a depth network emits.

    Keys: disp [B,1,H,W]; tgt [B,3,H,W]; src [S,B,3,H,W]; axisangle,
    translation [S,B,1,3] (PoseDecoder scale 0.01, reference: networks/posenet.py:132);
    K, inv_K [B,4,4]; noise [B,S,H,W] standard normal (the tie-break draw of
    reference train.py:1023-1024 before the 1e-5 scale); mask_rec [B,1,H,W] in
    {0,1} or None.
    """
    rng = np.random.default_rng(seed)
    tgt, srcs = triplet_images(rng, batch, height, width)
    while len(srcs) < num_src:
        srcs.append(srcs[-1][:, :, ::-1, :].copy())
    srcs = srcs[:num_src]
    disp = (np.float32(disp_lo) + np.float32(disp_hi - disp_lo)
            * rng.random((batch, 1, height, width), dtype=np.float32)).astype(np.float32)
    if disp_mode == "smooth":     # drawn AFTER the noise field so the other streams are unchanged
        disp = smooth_field(np.random.default_rng(seed + 7777), batch, height, width)
    elif disp_mode != "noise":
        raise ValueError(disp_mode)
    axisangle = (pose_scale * rng.standard_normal((num_src, batch, 1, 3))).astype(np.float32)
    translation = (pose_scale * rng.standard_normal((num_src, batch, 1, 3))).astype(np.float32)
    K, inv_K = kitti_intrinsics(batch, height, width)
    noise = rng.standard_normal((batch, num_src, height, width)).astype(np.float32)
    mask = None
    if with_mask:
        # valid-region mask of a rotated crop: ones except the four corners
        yy, xx = np.mgrid[0:height, 0:width].astype(np.float32)
        mask = np.ones((batch, 1, height, width), np.float32)
        for b in range(batch):
            ang = np.deg2rad(rng.uniform(-5.0, 5.0))
            cx, cy = (width - 1) / 2.0, (height - 1) / 2.0
            xr = np.cos(ang) * (xx - cx) - np.sin(ang) * (yy - cy)
            yr = np.sin(ang) * (xx - cx) + np.cos(ang) * (yy - cy)
            inside = (np.abs(xr) <= cx * 0.98) & (np.abs(yr) <= cy * 0.98)
            mask[b, 0] = inside.astype(np.float32)
    return {
        "disp": disp, "tgt": tgt, "src": np.stack(srcs, 0),
        "axisangle": axisangle, "translation": translation,
        "K": K, "inv_K": inv_K, "noise": noise, "mask_rec": mask,
    }


def training_batch(seed, batch, height, width):
    """A batch dict with the keys ``Trainer.process_batch`` reads
    (reference: train.py:698-886; contract in SURVEY.md section 3.4), as numpy arrays.
    """
    rng = np.random.default_rng(seed)
    tgt, srcs = triplet_images(rng, batch, height, width)
    frames = {0: tgt, -1: srcs[0], 1: srcs[1]}
    out = {}
    gain = (0.8 + 0.4 * rng.random((batch, 1, 1, 1))).astype(np.float32)
    for f, img in frames.items():
        out[("color", f, 0)] = img
        out[("color_aug", f, 0)] = np.clip(img * gain, 0, 1).astype(np.float32)
    K, inv_K = kitti_intrinsics(batch, height, width)
    for s in range(4):
        Ks, iKs = kitti_intrinsics(batch, height // (2 ** s), width // (2 ** s))
        out[("K", s)], out[("inv_K", s)] = Ks, iKs
    # affine augmentation metadata (reference: datasets/mono_dataset.py:110-149)
    ratio = rng.uniform(1.2, 2.0, size=(batch, 1)).astype(np.float32)
    angle = rng.uniform(-5.0, 5.0, size=(batch, 1)).astype(np.float32)
    box = np.zeros((batch, 4), np.int64)
    Rc = np.zeros((batch, 3, 3), np.float32)
    for b in range(batch):
        r = float(ratio[b, 0])
        h_re, w_re = int(height * r), int(width * r)
        w0 = int((w_re - width) * rng.random())
        h0 = int((h_re - height) * rng.random())
        box[b] = (round(w0 / r), round(h0 / r), round(width / r), round(height / r))
        a = np.pi / 180.0 * float(angle[b, 0])
        fs = 1.0 / r
        R = np.array([[np.cos(-a), np.sin(a), 0], [np.sin(-a), np.cos(-a), 0], [0, 0, 1]], np.float32)
        t = R @ np.array([-fs * w_re / 2, -fs * h_re / 2, fs - 1], np.float32) + \
            np.array([(w_re / 2 - w0) * fs, (h_re / 2 - h0) * fs, 0], np.float32)
        Rc_b = inv_K[b, :3, :3] @ R @ K[b, :3, :3]
        Rc_b[:, 2] += inv_K[b, :3, :3] @ t
        Rc[b] = Rc_b
    out["Rc"], out["ratio_local"], out["angle"], out["box"] = Rc, ratio, angle, box
    aff_tgt, aff_srcs = triplet_images(rng, batch, height, width)
    aff = {0: aff_tgt, -1: aff_srcs[0], 1: aff_srcs[1]}
    for f, img in aff.items():
        out[("color_affine", f, 0)] = img
        out[("color_affine_aug", f, 0)] = np.clip(img * gain, 0, 1).astype(np.float32)
    yy, xx = np.mgrid[0:height, 0:width].astype(np.float32)
    m = np.ones((batch, 1, height, width), np.float32)
    for b in range(batch):
        a = np.deg2rad(float(angle[b, 0]))
        cx, cy = (width - 1) / 2.0, (height - 1) / 2.0
        xr = np.cos(a) * (xx - cx) - np.sin(a) * (yy - cy)
        yr = np.sin(a) * (xx - cx) + np.cos(a) * (yy - cy)
        m[b, 0] = ((np.abs(xr) <= cx) & (np.abs(yr) <= cy)).astype(np.float32)
    out["valid_mask_rec"] = m
    out["valid_mask_cons"] = m.copy()
    return out

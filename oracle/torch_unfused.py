"""Un-fused PyTorch-op formulation of one hot-path unit ("the honest
stand-in for the reference CPU path").

TEST / MEASUREMENT INFRASTRUCTURE ONLY -- like everything under oracle/, this module is
imported by tests/ and by the cpu_baseline legs of bench.py, never by the product package.

The reference's Python does not travel to the GPU box, so its CPU cost there is estimated by
running the SAME chain of ATen operators the reference's code path issues for a unit --
disp_to_depth, BackprojectDepth (bmm), Project3D (bmm, divides, permute),
F.grid_sample(bilinear, border, align_corners=True), SSIM as ReflectionPad2d + five
AvgPool2d(3,1) + element-wise ops, channel means, cat / min over candidates, the
mean-normalised edge-aware smoothness -- one operator at a time with every intermediate
materialised, under autograd, on the host cores.  It restates what the functions compute
(reference: layers.py:16-25, 168-222, 231-242, 261-290; train.py:956-1051); it is checked
against the golden vectors captured from the reference in tests/test_oracle_golden.py
(parity status: pinned, tolerance 1e-6 on the loss, 1e-5 on gradients).

Parity status of this file: pinned (tests/test_oracle_golden.py::test_unfused_torch_vs_golden).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

NO_SSIM, AVG_REPROJ, NO_AUTOMASK = 1, 2, 4


def _ssim(x, y):
    """layers.py:261-290: 3x3 mean filters over reflect-padded images."""
    c1, c2 = 0.01 ** 2, 0.03 ** 2
    x = F.pad(x, (1, 1, 1, 1), mode="reflect")
    y = F.pad(y, (1, 1, 1, 1), mode="reflect")
    mu_x = F.avg_pool2d(x, 3, 1)
    mu_y = F.avg_pool2d(y, 3, 1)
    sigma_x = F.avg_pool2d(x ** 2, 3, 1) - mu_x ** 2
    sigma_y = F.avg_pool2d(y ** 2, 3, 1) - mu_y ** 2
    sigma_xy = F.avg_pool2d(x * y, 3, 1) - mu_x * mu_y
    n = (2 * mu_x * mu_y + c1) * (2 * sigma_xy + c2)
    d = (mu_x ** 2 + mu_y ** 2 + c1) * (sigma_x + sigma_y + c2)
    return torch.clamp((1 - n / d) / 2, 0, 1)


def _reprojection(pred, target, no_ssim):
    """train.py:973-985"""
    l1 = (target - pred).abs().mean(1, True)
    if no_ssim:
        return l1
    return 0.85 * _ssim(pred, target).mean(1, True) + 0.15 * l1


def _smooth(disp, img):
    """layers.py:231-242"""
    gdx = (disp[:, :, :, :-1] - disp[:, :, :, 1:]).abs()
    gdy = (disp[:, :, :-1, :] - disp[:, :, 1:, :]).abs()
    gix = (img[:, :, :, :-1] - img[:, :, :, 1:]).abs().mean(1, keepdim=True)
    giy = (img[:, :, :-1, :] - img[:, :, 1:, :]).abs().mean(1, keepdim=True)
    return (gdx * torch.exp(-gix)).mean() + (gdy * torch.exp(-giy)).mean()


def pix_coords(H, W, dtype=torch.float32):
    """[3, H*W] rows (x, y, 1), index = y*W + x (layers.py:179-190)."""
    ys, xs = torch.meshgrid(torch.arange(H, dtype=dtype), torch.arange(W, dtype=dtype), indexing="ij")
    return torch.stack([xs.reshape(-1), ys.reshape(-1), torch.ones(H * W, dtype=dtype)], 0)


def warp(disp, T, src, K, inv_K, pix, min_depth=0.1, max_depth=100.0, eps=1e-7):
    """Trainer.generate_images_pred for one source (train.py:956-971)."""
    B, _, H, W = disp.shape
    min_disp, max_disp = 1 / max_depth, 1 / min_depth
    depth = 1 / (min_disp + (max_disp - min_disp) * disp)
    cam = torch.matmul(inv_K[:, :3, :3], pix.unsqueeze(0).expand(B, -1, -1))
    cam = depth.view(B, 1, -1) * cam
    cam = torch.cat([cam, torch.ones(B, 1, H * W, dtype=cam.dtype)], 1)
    P = torch.matmul(K, T)[:, :3, :]
    c = torch.matmul(P, cam)
    uv = c[:, :2, :] / (c[:, 2:3, :] + eps)
    uv = uv.view(B, 2, H, W).permute(0, 2, 3, 1)
    uv = torch.stack([uv[..., 0] / (W - 1), uv[..., 1] / (H - 1)], -1)
    uv = (uv - 0.5) * 2
    return F.grid_sample(src, uv, mode="bilinear", padding_mode="border", align_corners=True)


def losses_base(disp, tgt, warped, srcs, noise, mask_rec, flags, smoothness=1e-3):
    """Trainer.compute_losses_base (train.py:987-1051) -> (loss, to_optimise, idxs|None)."""
    no_ssim, avg, automask = bool(flags & NO_SSIM), bool(flags & AVG_REPROJ), not flags & NO_AUTOMASK
    rp = torch.cat([_reprojection(w, tgt, no_ssim) for w in warped], 1)
    if avg:
        rp = rp.mean(1, keepdim=True)
    if automask:
        idl = torch.cat([_reprojection(s, tgt, no_ssim) for s in srcs], 1)
        if avg:
            idl = idl.mean(1, keepdim=True)
        idl = idl + noise * 0.00001
        combined = torch.cat([idl, rp], 1)
    else:
        combined = rp
    if combined.shape[1] == 1:
        to_opt, idxs = combined, None
    else:
        to_opt, idxs = torch.min(combined, dim=1)
    if mask_rec is not None:
        to_opt = to_opt * mask_rec[:, 0]
    loss = to_opt.mean()
    norm_disp = disp / (disp.mean(2, True).mean(3, True) + 1e-7)
    loss = loss + smoothness * _smooth(norm_disp, tgt)
    return loss, to_opt, idxs


def unit(disp, tgt, srcs, T, K, inv_K, noise, mask_rec, flags=0, smoothness=1e-3, want_grads=True):
    """One unit = S x generate_images_pred + compute_losses_base, forward (+ backward) with
    autograd on CPU tensors.  numpy or torch inputs; T [S,B,4,4].  Returns a dict."""
    t = lambda a: None if a is None else torch.as_tensor(a, dtype=torch.float32)  # noqa: E731
    disp, tgt, T, K, inv_K, noise, mask_rec = (t(a) for a in (disp, tgt, T, K, inv_K, noise, mask_rec))
    srcs = [t(s) for s in srcs]
    B, _, H, W = disp.shape
    if want_grads:
        disp = disp.clone().requires_grad_(True)
        T = T.clone().requires_grad_(True)
    pix = pix_coords(H, W)
    warped = [warp(disp, T[k], srcs[k], K, inv_K, pix) for k in range(len(srcs))]
    loss, to_opt, idxs = losses_base(disp, tgt, warped, srcs, noise, mask_rec, flags, smoothness)
    out = {"loss": float(loss.detach()), "to_opt": to_opt.detach().numpy(),
           "idx": None if idxs is None else idxs.numpy(), "warped": [w.detach().numpy() for w in warped]}
    if want_grads:
        loss.backward()
        out["grad_disp"] = disp.grad.numpy()
        out["grad_T"] = T.grad.numpy()
    return out

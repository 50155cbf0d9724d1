"""torch.autograd glue over the C ABI (include/mvf_hotpath.h).

PyTorch is plumbing here: it owns device memory (caching allocator), the current HIP
stream and the autograd tape; all arithmetic of the hot path happens in the gfx950
kernels.  Every Function below is the native body of one reference function (cited).
Kernels are enqueued on ``torch.cuda.current_stream()`` so they order with the
surrounding MIOpen kernels and with DDP's reducer through normal stream semantics.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _native as nat

NO_SSIM, AVG_REPROJ, NO_AUTOMASK = nat.NO_SSIM, nat.AVG_REPROJ, nat.NO_AUTOMASK


def _raw_stream():
    """The current HIP stream of the current device as an integer handle.  (`torch.cuda.current_stream()` builds a
    Python Stream object through three layers of device-index helpers: 11 us per call, four calls per unit launch --
    a twentieth of the host-bound stand-alone hot-path loop; the raw getter is a C call.)"""
    return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())


def _stream():
    return C.c_void_p(_raw_stream())


def _c(t):
    """contiguous fp32 view (the reference passes slices such as cam_points[:, :2])."""
    if t is None:
        return None
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def _ws(ref, B, H, W):
    n = nat.lib().mvf_workspace_floats(B, H, W)
    return torch.empty(n, dtype=torch.float32, device=ref.device)


def depth_consts(min_depth, max_depth):
    """(min_disp, range) as the fp32 scalars of reference layers.py:21-23."""
    min_disp = 1 / max_depth
    max_disp = 1 / min_depth
    return float(np.float32(min_disp)), float(np.float32(max_disp - min_disp))


# ------------------------------------------------------------------ a1 disp_to_depth
class DispToDepth(torch.autograd.Function):
    """reference: layers.py:16-25"""

    @staticmethod
    def forward(ctx, disp, min_depth, max_depth):
        nat.require_device(disp)
        disp = _c(disp)
        md, rg = depth_consts(min_depth, max_depth)
        scaled, depth = torch.empty_like(disp), torch.empty_like(disp)
        nat.check(nat.lib().mvf_disp_to_depth_fwd(nat.ptr(disp), nat.ptr(scaled), nat.ptr(depth),
                                                  disp.numel(), md, rg, _stream()), "disp_to_depth")
        ctx.save_for_backward(disp)
        ctx.consts = (md, rg)
        return scaled, depth

    @staticmethod
    def backward(ctx, g_scaled, g_depth):
        (disp,) = ctx.saved_tensors
        md, rg = ctx.consts
        g_scaled, g_depth = _c(g_scaled), _c(g_depth)
        g = torch.empty_like(disp)
        nat.check(nat.lib().mvf_disp_to_depth_bwd(nat.ptr(disp), nat.ptr(g_scaled), nat.ptr(g_depth),
                                                  nat.ptr(g), disp.numel(), md, rg, _stream()),
                  "disp_to_depth_bwd")
        return g, None, None


# ------------------------------------------------------------------ a2 BackprojectDepth
class Backproject(torch.autograd.Function):
    """reference: layers.py:192-197"""

    @staticmethod
    def forward(ctx, depth, inv_K, B, H, W):
        nat.require_device(depth, inv_K)
        depth, inv_K = _c(depth), _c(inv_K)
        cam = torch.empty((B, 4, H * W), dtype=torch.float32, device=depth.device)
        nat.check(nat.lib().mvf_backproject_fwd(nat.ptr(depth), nat.ptr(inv_K), nat.ptr(cam), B, H,
                                                W, _stream()), "backproject")
        ctx.save_for_backward(inv_K)
        ctx.dims = (B, H, W)
        return cam

    @staticmethod
    def backward(ctx, g_cam):
        (inv_K,) = ctx.saved_tensors
        B, H, W = ctx.dims
        g_cam = _c(g_cam)
        g_depth = torch.empty((B, 1, H, W), dtype=torch.float32, device=g_cam.device)
        nat.check(nat.lib().mvf_backproject_bwd(nat.ptr(g_cam), nat.ptr(inv_K), nat.ptr(g_depth), B,
                                                H, W, _stream()), "backproject_bwd")
        return g_depth, None, None, None, None


# ------------------------------------------------------------------ a3 Project3D
class Project(torch.autograd.Function):
    """reference: layers.py:211-222"""

    @staticmethod
    def forward(ctx, points, K, T, B, H, W, eps):
        nat.require_device(points, K, T)
        points, K, T = _c(points), _c(K), _c(T)
        pix = torch.empty((B, H, W, 2), dtype=torch.float32, device=points.device)
        nat.check(nat.lib().mvf_project_fwd(nat.ptr(points), nat.ptr(K), nat.ptr(T), nat.ptr(pix), B,
                                            H, W, eps, _stream()), "project")
        ctx.save_for_backward(points, K, T)
        ctx.dims = (B, H, W, eps)
        return pix

    @staticmethod
    def backward(ctx, g_pix):
        points, K, T = ctx.saved_tensors
        B, H, W, eps = ctx.dims
        g_pix = _c(g_pix)
        need_cam, need_T = ctx.needs_input_grad[0], ctx.needs_input_grad[2]
        g_cam = torch.empty_like(points) if need_cam else None
        g_T = torch.empty((B, 4, 4), dtype=torch.float32, device=points.device) if need_T else None
        ws = _ws(points, B, H, W)
        nat.check(nat.lib().mvf_project_bwd(nat.ptr(points), nat.ptr(K), nat.ptr(T), nat.ptr(g_pix),
                                            nat.ptr(g_cam), nat.ptr(g_T), nat.ptr(ws), B, H, W, eps,
                                            _stream()), "project_bwd")
        return g_cam, None, g_T, None, None, None, None


# ------------------------------------------------------------------ a4 grid_sample
class GridSampleBorderAC(torch.autograd.Function):
    """F.grid_sample(img, grid, padding_mode="border", align_corners=True);
    reference call site: train.py:966-969"""

    @staticmethod
    def forward(ctx, img, grid):
        nat.require_device(img, grid)
        img, grid = _c(img), _c(grid)
        B, Cc, H, W = img.shape
        if tuple(grid.shape) != (B, H, W, 2):
            raise RuntimeError(f"grid must be [B,H,W,2] matching img, got {tuple(grid.shape)}")
        out = torch.empty_like(img)
        nat.check(nat.lib().mvf_grid_sample_fwd(nat.ptr(img), nat.ptr(grid), nat.ptr(out), None, B,
                                                Cc, H, W, _stream()), "grid_sample")
        ctx.save_for_backward(img, grid)
        return out

    @staticmethod
    def backward(ctx, g_out):
        img, grid = ctx.saved_tensors
        B, Cc, H, W = img.shape
        g_out = _c(g_out)
        g_grid = torch.empty_like(grid) if ctx.needs_input_grad[1] else None
        g_img = torch.zeros_like(img) if ctx.needs_input_grad[0] else None
        nat.check(nat.lib().mvf_grid_sample_bwd(nat.ptr(img), nat.ptr(grid), nat.ptr(g_out),
                                                nat.ptr(g_grid), nat.ptr(g_img), B, Cc, H, W,
                                                _stream()), "grid_sample_bwd")
        return g_img, g_grid


def grid_sample_indices(img_shape, grid):
    """int32 [B,H,W,2] top-left taps (x0,y0): the bit-exact integers of the parity contract."""
    nat.require_device(grid)
    B, Cc, H, W = img_shape
    grid = _c(grid)
    idx = torch.empty((B, H, W, 2), dtype=torch.int32, device=grid.device)
    nat.check(nat.lib().mvf_grid_sample_fwd(None, nat.ptr(grid), None, nat.ptr(idx), B, Cc, H, W,
                                            _stream()), "grid_sample(idx)")
    return idx


# ------------------------------------------------------------------ a5 fused warp
class Warp(torch.autograd.Function):
    """Trainer.generate_images_pred for one source; reference: train.py:956-971"""

    @staticmethod
    def forward(ctx, disp, T, src, K, inv_K, min_depth, max_depth, eps):
        nat.require_device(disp, T, src, K, inv_K)
        disp, T, src, K, inv_K = _c(disp), _c(T), _c(src), _c(K), _c(inv_K)
        B, _, H, W = disp.shape
        md, rg = depth_consts(min_depth, max_depth)
        warped = torch.empty((B, 3, H, W), dtype=torch.float32, device=disp.device)
        nat.check(nat.lib().mvf_warp_fwd(nat.ptr(disp), nat.ptr(inv_K), nat.ptr(K), nat.ptr(T),
                                         nat.ptr(src), nat.ptr(warped), None, None, B, H, W, md, rg,
                                         eps, _stream()), "warp_fwd")
        ctx.save_for_backward(disp, T, src, K, inv_K)
        ctx.consts = (md, rg, eps)
        return warped

    @staticmethod
    def backward(ctx, g_warped):
        disp, T, src, K, inv_K = ctx.saved_tensors
        md, rg, eps = ctx.consts
        B, _, H, W = disp.shape
        g_warped = _c(g_warped)
        g_disp = torch.empty_like(disp)
        g_T = torch.empty_like(T)
        g_src = torch.zeros_like(src) if ctx.needs_input_grad[2] else None
        ws = _ws(disp, B, H, W)
        nat.check(nat.lib().mvf_warp_bwd(nat.ptr(disp), nat.ptr(inv_K), nat.ptr(K), nat.ptr(T),
                                         nat.ptr(src), nat.ptr(g_warped), nat.ptr(g_disp),
                                         nat.ptr(g_T), nat.ptr(g_src), nat.ptr(ws), 0, B, H, W, md,
                                         rg, eps, _stream()), "warp_bwd")
        return g_disp, g_T, g_src, None, None, None, None, None


def warp_debug(disp, T, src, K, inv_K, min_depth=0.1, max_depth=100.0, eps=1e-7):
    """(warped, pix [B,H,W,2], idx_xy int32 [B,H,W,2]) -- for stage-level parity tests."""
    nat.require_device(disp, T, src, K, inv_K)
    disp, T, src, K, inv_K = _c(disp), _c(T), _c(src), _c(K), _c(inv_K)
    B, _, H, W = disp.shape
    md, rg = depth_consts(min_depth, max_depth)
    warped = torch.empty((B, 3, H, W), dtype=torch.float32, device=disp.device)
    pix = torch.empty((B, H, W, 2), dtype=torch.float32, device=disp.device)
    idx = torch.empty((B, H, W, 2), dtype=torch.int32, device=disp.device)
    nat.check(nat.lib().mvf_warp_fwd(nat.ptr(disp), nat.ptr(inv_K), nat.ptr(K), nat.ptr(T),
                                     nat.ptr(src), nat.ptr(warped), nat.ptr(pix), nat.ptr(idx), B, H,
                                     W, md, rg, eps, _stream()), "warp_fwd")
    return warped, pix, idx


# ------------------------------------------------------------------ a6 SSIM
class SSIMFn(torch.autograd.Function):
    """reference: layers.py:277-290"""

    @staticmethod
    def forward(ctx, x, y):
        nat.require_device(x, y)
        x, y = _c(x), _c(y)
        B, Cc, H, W = x.shape
        if H < 2 or W < 2:
            raise RuntimeError("ReflectionPad2d(1) needs H, W >= 2")
        out = torch.empty_like(x)
        nat.check(nat.lib().mvf_ssim_fwd(nat.ptr(x), nat.ptr(y), nat.ptr(out), B, Cc, H, W,
                                         _stream()), "ssim")
        ctx.save_for_backward(x, y)
        return out

    @staticmethod
    def backward(ctx, g):
        x, y = ctx.saved_tensors
        B, Cc, H, W = x.shape
        g = _c(g)
        gx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        gy = torch.empty_like(y) if ctx.needs_input_grad[1] else None
        nat.check(nat.lib().mvf_ssim_bwd(nat.ptr(x), nat.ptr(y), nat.ptr(g), nat.ptr(gx), nat.ptr(gy),
                                         B, Cc, H, W, _stream()), "ssim_bwd")
        return gx, gy


# ------------------------------------------------------------------ a7 reprojection
class Reprojection(torch.autograd.Function):
    """Trainer.compute_reprojection_loss; reference: train.py:973-985 (grad w.r.t. pred)"""

    @staticmethod
    def forward(ctx, pred, target, no_ssim):
        nat.require_device(pred, target)
        pred, target = _c(pred), _c(target)
        B, Cc, H, W = pred.shape
        if Cc != 3:
            raise RuntimeError("compute_reprojection_loss expects 3-channel images")
        out = torch.empty((B, 1, H, W), dtype=torch.float32, device=pred.device)
        nat.check(nat.lib().mvf_reprojection_fwd(nat.ptr(pred), nat.ptr(target), nat.ptr(out), B, H,
                                                 W, int(no_ssim), _stream()), "reprojection")
        ctx.save_for_backward(pred, target)
        ctx.no_ssim = int(no_ssim)
        return out

    @staticmethod
    def backward(ctx, g):
        pred, target = ctx.saved_tensors
        if ctx.needs_input_grad[1]:
            raise NotImplementedError("grad w.r.t. the target image is never needed by the trainer")
        B, _, H, W = pred.shape
        g = _c(g)
        gp = torch.empty_like(pred)
        nat.check(nat.lib().mvf_reprojection_bwd(nat.ptr(pred), nat.ptr(target), nat.ptr(g),
                                                 nat.ptr(gp), B, H, W, ctx.no_ssim, _stream()),
                  "reprojection_bwd")
        return gp, None, None


# ------------------------------------------------------------------ a9 smoothness
class Smooth(torch.autograd.Function):
    """get_smooth_loss; reference: layers.py:231-242 (normalise=True adds train.py:1044-1045)"""

    @staticmethod
    def forward(ctx, disp, img, normalise):
        nat.require_device(disp, img)
        disp, img = _c(disp), _c(img)
        B, _, H, W = disp.shape
        out = torch.empty(3, dtype=torch.float32, device=disp.device)
        stats = torch.empty((B, 4), dtype=torch.float32, device=disp.device)
        ws = _ws(disp, B, H, W)
        nat.check(nat.lib().mvf_smooth_fwd(nat.ptr(disp), nat.ptr(img), nat.ptr(out), nat.ptr(stats),
                                           nat.ptr(ws), int(normalise), B, H, W, _stream()), "smooth")
        ctx.save_for_backward(disp, img, stats)
        ctx.normalise = int(normalise)
        return out[0]

    @staticmethod
    def backward(ctx, g):
        disp, img, stats = ctx.saved_tensors
        B, _, H, W = disp.shape
        g = _c(g).reshape(1)
        gd = torch.empty_like(disp)
        nat.check(nat.lib().mvf_smooth_bwd(nat.ptr(disp), nat.ptr(img), nat.ptr(stats), nat.ptr(g),
                                           1.0, nat.ptr(gd), 0, ctx.normalise, B, H, W, _stream()),
                  "smooth_bwd")
        return gd, None, None


# ------------------------------------------------------------------ a8 compute_losses_base
def _flags(no_ssim, avg_reprojection, disable_automasking):
    return (NO_SSIM if no_ssim else 0) | (AVG_REPROJ if avg_reprojection else 0) | \
        (NO_AUTOMASK if disable_automasking else 0)


class LossesBase(torch.autograd.Function):
    """Trainer.compute_losses_base on materialised warped images; reference: train.py:987-1051.
    inputs: disp, tgt, mask_rec|None, noise|None, S, flags, smoothness, *warped(S), *src(S)
    outputs: loss (0-dim), auto_mask [B,1,H,W], to_opt [B,H,W], argmin uint8 [B,H,W]"""

    @staticmethod
    def forward(ctx, disp, tgt, mask_rec, noise, S, flags, smoothness, *imgs):
        warped = [_c(t) for t in imgs[:S]]
        src = [_c(t) for t in imgs[S:2 * S]]
        nat.require_device(disp, tgt, mask_rec, noise, *warped, *src)
        disp, tgt, mask_rec, noise = _c(disp), _c(tgt), _c(mask_rec), _c(noise)
        B, _, H, W = disp.shape
        dev = disp.device
        loss = torch.empty(3, dtype=torch.float32, device=dev)
        argmin = torch.empty((B, H, W), dtype=torch.uint8, device=dev)
        auto_mask = torch.empty((B, 1, H, W), dtype=torch.float32, device=dev)
        to_opt = torch.empty((B, H, W), dtype=torch.float32, device=dev)
        stats = torch.empty((B, 4), dtype=torch.float32, device=dev)
        ws = _ws(disp, B, H, W)
        wp, wkeep = nat.ptr_array(warped)
        sp, skeep = nat.ptr_array(src) if src else (None, None)
        nat.check(nat.lib().mvf_photo_fwd(nat.ptr(disp), nat.ptr(tgt), wp, sp, nat.ptr(noise),
                                          nat.ptr(mask_rec), S, flags, smoothness, nat.ptr(loss),
                                          nat.ptr(argmin), nat.ptr(auto_mask), nat.ptr(to_opt),
                                          nat.ptr(stats), nat.ptr(ws), B, H, W, _stream()),
                  "photo_fwd")
        ctx.save_for_backward(disp, tgt, mask_rec, argmin, stats, *warped)
        ctx.cfg = (S, flags, smoothness)
        ctx.mark_non_differentiable(auto_mask, to_opt, argmin)
        ctx.set_materialize_grads(False)     # no zero tensors for the three maps in backward()
        return loss[0], auto_mask, to_opt, argmin

    @staticmethod
    def backward(ctx, g_loss, *_unused):
        disp, tgt, mask_rec, argmin, stats, *warped = ctx.saved_tensors
        S, flags, smoothness = ctx.cfg
        if g_loss is None:
            return (None,) * (7 + 2 * S)
        B, _, H, W = disp.shape
        g_loss = _c(g_loss).reshape(1)
        g_warped = [torch.empty_like(w) for w in warped]
        g_disp = torch.empty_like(disp)
        wp, wkeep = nat.ptr_array(warped)
        gp, gkeep = nat.ptr_array(g_warped)
        nat.check(nat.lib().mvf_photo_bwd(nat.ptr(disp), nat.ptr(tgt), wp, nat.ptr(argmin),
                                          nat.ptr(mask_rec), nat.ptr(stats), nat.ptr(g_loss), S,
                                          flags, smoothness, gp, nat.ptr(g_disp), B, H, W,
                                          _stream()), "photo_bwd")
        return (g_disp, None, None, None, None, None, None, *g_warped, *([None] * S))


# ------------------------------------------------------------------ fused unit
# When a gradient will be asked for (training), units with one source pair run their forward
# and backward as ONE tile kernel (mvf_units_fwdbwd): the gradients for an upstream gradient of
# 1 are produced alongside the loss and only scaled in backward().  Several mutually independent
# units of the same shape go out as one launch (`Units`).  Set UNIT_FWDBWD to False to force the
# separate forward / backward kernels (the tests compare both).
UNIT_FWDBWD = True


def unit_uses_fwdbwd(S, disp, T):
    """True when Unit.apply will take the one-kernel forward+backward route."""
    return bool(UNIT_FWDBWD and S <= 2 and torch.is_grad_enabled() and
                (disp.requires_grad or T.requires_grad))


def _img(t):
    """fp32 [B,C,H,W] whose images are contiguous planes -> (tensor, image stride in floats).
    A view that strides over the batch only (one group of a grouped network call's interleaved
    output) is passed as it lies; anything else is made contiguous."""
    if t is None:
        return None, 0
    if t.dtype != torch.float32:
        t = t.float()
    B, Cc, H, W = t.shape
    st = t.stride()
    inner_ok = st[3] == 1 and st[2] == W and (Cc == 1 or st[1] == H * W)
    if not inner_ok or (B > 1 and st[0] < Cc * H * W) or t.data_ptr() % 4:
        t = t.contiguous()
        st = t.stride()
    return t, (int(st[0]) if B > 1 else Cc * H * W)


_TICKETS = {}


def _ticket_key(dev):
    # (require_device has already held `dev` to the current device: its current stream is the launch stream)
    return (dev.index, _raw_stream())


def _tickets(dev, n):
    """Zeroed int32 counters for the in-kernel finishing folds: one persistent buffer per
    (device, stream) -- launches on a stream are ordered and each leaves the counters zero.
    A launch that FAILS may leave them non-zero (the unit kernel ran, the finishing kernel did not, or was
    aborted): `_drop_tickets` then forgets the buffer, so that the next launch on the stream starts from a
    freshly zeroed one instead of folding on a stale count."""
    key = _ticket_key(dev)
    t = _TICKETS.get(key)
    if t is None or t.numel() < n:
        t = torch.zeros(max(n, 1024), dtype=torch.int32, device=dev)
        _TICKETS[key] = t
    return t


def _drop_tickets(dev):
    _TICKETS.pop(_ticket_key(dev), None)


def _expect_shape(name, t, shape):
    if t is not None and tuple(t.shape) != tuple(shape):
        raise RuntimeError(f"{name} must be {tuple(shape)}, got {tuple(t.shape)}")


UNIT_FIELDS = 8      # tensors per unit in Units.apply before its sources


class Units(torch.autograd.Function):
    """n mutually independent hot-path units of one shape in ONE launch (reference: the three
    single-frame units train.py:747-760, the three multi-frame ones 795-810, the three affine
    ones 837-882), each = S x generate_images_pred + compute_losses_base with the warped images
    kept in LDS (train.py:956-1051).

    apply(cfg, disp_0, tgt_0, T_0, K_0, inv_K_0, mask_0, noise_0, ident_in_0, *src_0 (S), disp_1, ...)
    cfg = dict(n, S, flags, smoothness, min_depth, max_depth, eps, want_mask, want_idx,
               want_ident (bool | one bool per unit), want_sum, sum_in (bool), noise_out (list|None),
               mean_parts (list|None))
    returns (losses [n], terms [n,2] (photo, smooth), then per unit: auto_mask, argmin, idx,
             ident) -- absent outputs are empty tensors -- and, LAST, with want_sum the 0-dim sum of the n
    losses (differentiable like `losses`; the finishing kernel writes it, the backward pass takes its upstream
    gradient as one more device scalar: no reduce / expand / copy launches around a group of units).
    With cfg["sum_in"] the input tensor after the units' is a 0-dim running total (fp32, on the device) the sum starts
    from: `loss_base += group` of train.py:760 / 812 / 882 inside the finishing kernel; its gradient is the sum's.
    cfg["sinks"] (list per unit of (HeadSink, first, step) | None) + cfg["n_tokens"] trailing token inputs: a unit with a
    sink reads the disparity head's output in place (pass it DETACHED); its raw disparity gradient is deposited in the
    sink during backward and scaled by the head's adjoint kernel on load (see `HeadSink`)."""

    @staticmethod
    def forward(ctx, cfg, *flat):
        n, S = cfg["n"], cfg["S"]
        flags, smoothness = cfg["flags"], cfg["smoothness"]
        per = UNIT_FIELDS + S
        want_sum = bool(cfg.get("want_sum", False))
        n_tok = int(cfg.get("n_tokens", 0))
        if n_tok:
            flat = flat[:-n_tok]       # the tokens only carry the autograd edge to their disparity heads
        sinks = cfg.get("sinks") or [None] * n
        sum_in = None
        if cfg.get("sum_in", False):
            if not want_sum:
                raise RuntimeError("sum_in needs want_sum")
            flat, sum_in = flat[:-1], flat[-1]
            if sum_in.dtype != torch.float32 or sum_in.numel() != 1:
                raise RuntimeError("sum_in must be one fp32 value")
            nat.require_device(sum_in)
            sum_in = _c(sum_in)
        assert len(flat) == n * per and 1 <= n <= nat.MAX_UNITS
        md, rg = depth_consts(cfg["min_depth"], cfg["max_depth"])
        want_mask, want_idx, want_ident = cfg.get("want_mask", False), cfg.get("want_idx", False), \
            cfg.get("want_ident", False)
        want_ident = list(want_ident) if isinstance(want_ident, (list, tuple)) else [bool(want_ident)] * n
        noise_outs = cfg.get("noise_out") or [None] * n
        mean_parts = cfg.get("mean_parts") or [None] * n
        automask = not (flags & NO_AUTOMASK)
        d0 = flat[0]
        B, _, H, W = d0.shape
        dev = d0.device
        descs = (nat.UnitDesc * n)()
        keep = []
        loss3 = torch.empty((n, 3), dtype=torch.float32, device=dev)
        loss_sum = torch.empty((), dtype=torch.float32, device=dev) if want_sum else None
        stats = torch.empty((n, B, 4), dtype=torch.float32, device=dev)
        g_disp = torch.empty((n, B, 1, H, W), dtype=torch.float32, device=dev)
        g_T = torch.empty((n, S, B, 4, 4), dtype=torch.float32, device=dev)
        outs = []
        needs = []
        empty = torch.empty(0, device=dev)
        seeds = None
        for u in range(n):
            disp, tgt, T, K, inv_K, mask_rec, noise, ident_in = flat[u * per:u * per + UNIT_FIELDS]
            src = flat[u * per + UNIT_FIELDS:(u + 1) * per]
            if tuple(disp.shape) != (B, 1, H, W):
                raise RuntimeError("the units of one launch must share the shape [B,1,H,W]")
            nat.require_device(disp, tgt, T, K, inv_K, mask_rec, noise, ident_in, *src)
            # raw pointers and strides go to the kernel: every plane it will index is checked here (a target or
            # source of another resolution / batch size would be an out-of-bounds device read, not an error)
            _expect_shape("tgt", tgt, (B, 3, H, W))
            for k, im in enumerate(src):
                _expect_shape(f"src[{k}]", im, (B, 3, H, W))
            _expect_shape("mask_rec", mask_rec, (B, 1, H, W))
            _expect_shape("K", K, (B, 4, 4))
            _expect_shape("inv_K", inv_K, (B, 4, 4))
            if noise is not None:
                _expect_shape("noise", noise, (B, 1 if (flags & AVG_REPROJ) else S, H, W))
            disp, ds = _img(disp)
            tgt, ts = _img(tgt)
            mask_rec, ms = _img(mask_rec)
            srcs = [_img(t) for t in src]
            T, K, inv_K, noise, ident_in = _c(T), _c(K), _c(inv_K), _c(noise), _c(ident_in)
            mp = _c(mean_parts[u]) if mean_parts[u] is not None else None
            if mp is not None and tuple(mp.shape) != (B, 32):
                raise RuntimeError("disp_mean_partials must be [B,32]")
            if tuple(T.shape) != (S, B, 4, 4):
                raise RuntimeError(f"T must be [S,B,4,4] = {(S, B, 4, 4)}, got {tuple(T.shape)}")
            if ident_in is not None and tuple(ident_in.shape) != (B, H, W, 2):
                raise RuntimeError("ident_in must be [B,H,W,2]")
            argmin = torch.empty((B, H, W), dtype=torch.uint8, device=dev)
            auto_mask = torch.empty((B, 1, H, W), dtype=torch.float32, device=dev) if want_mask else None
            idx = torch.empty((S, B, H, W, 2), dtype=torch.int32, device=dev) if want_idx else None
            ident = torch.empty((B, H, W, 2), dtype=torch.float32, device=dev) \
                if (want_ident[u] and automask) else None
            seed = 0
            if noise is None and automask:
                # tie-break draw of train.py:1023-1024 generated in the kernel: one 64-bit key per
                # unit from torch's CPU generator (reproducible under torch.manual_seed; the keys of a launch in one draw)
                if seeds is None:
                    seeds = torch.randint(0, 2 ** 62, (n,), dtype=torch.int64).tolist()
                seed = int(seeds[u])
            d = descs[u]
            d.disp, d.disp_stride = disp.data_ptr(), ds
            d.tgt, d.tgt_stride = tgt.data_ptr(), ts
            for k in range(S):
                d.src[k], d.src_stride[k] = srcs[k][0].data_ptr(), srcs[k][1]
            d.T, d.K, d.inv_K = T.data_ptr(), K.data_ptr(), inv_K.data_ptr()
            if mask_rec is not None:
                d.mask_rec, d.mask_stride = mask_rec.data_ptr(), ms
            d.noise = noise.data_ptr() if noise is not None else None
            d.disp_mean_partials = mp.data_ptr() if mp is not None else None
            d.ident_in = ident_in.data_ptr() if ident_in is not None else None
            d.noise_seed = seed
            d.ident_out = ident.data_ptr() if ident is not None else None
            d.loss, d.stats = loss3[u].data_ptr(), stats[u].data_ptr()
            d.g_disp_raw, d.g_stride = g_disp[u].data_ptr(), H * W
            d.g_T_raw = g_T[u].data_ptr()
            d.argmin = argmin.data_ptr()
            d.auto_mask = auto_mask.data_ptr() if auto_mask is not None else None
            d.idx_xy = idx.data_ptr() if idx is not None else None
            d.noise_out = noise_outs[u].data_ptr() if noise_outs[u] is not None else None
            keep += [disp, tgt, mask_rec, T, K, inv_K, noise, ident_in, mp, srcs]
            outs += [auto_mask if auto_mask is not None else empty, argmin,
                     idx if idx is not None else empty, ident if ident is not None else empty]
            needs.append((ctx.needs_input_grad[1 + u * per], ctx.needs_input_grad[1 + u * per + 2]))
        if want_sum:
            descs[0].loss_sum = loss_sum.data_ptr()
            descs[0].loss_sum_in = sum_in.data_ptr() if sum_in is not None else None
            keep.append(sum_in)
        ws = torch.empty(nat.lib().mvf_units_workspace_floats(n, B, H, W), dtype=torch.float32, device=dev)
        tk = _tickets(dev, nat.lib().mvf_units_ticket_ints(n, B))
        try:
            nat.check(nat.lib().mvf_units_fwdbwd(C.cast(descs, C.c_void_p), n, S, flags, smoothness, md, rg,
                                                 cfg["eps"], nat.ptr(ws), nat.ptr(tk), B, H, W, _stream()),
                      "units_fwdbwd")
        except Exception:
            _drop_tickets(dev)          # the counters may be non-zero now: never reuse them
            raise
        ctx.save_for_backward(g_disp, g_T, stats)
        ctx.n, ctx.S, ctx.smoothness, ctx.per, ctx.needs, ctx.want_sum = n, S, smoothness, per, needs, want_sum
        ctx.has_sum_in, ctx.n_tok, ctx.sinks = sum_in is not None, n_tok, sinks
        res = (loss3[:, 0], loss3[:, 1:], *outs)
        ctx.mark_non_differentiable(*res[1:])
        if want_sum:
            res = res + (loss_sum,)
        # the engine would otherwise hand backward() a ZERO tensor for every output that received no
        # gradient -- argmin (uint8 [B,H,W]), the identity maps, ... : 16 fill launches per step
        # (profiles/r03_hotpath_kernel_stats_before_nomaterialize.csv: 747 + 581 fills in 83 steps)
        ctx.set_materialize_grads(False)
        return res

    @staticmethod
    def backward(ctx, g_losses, *rest):
        # raw gradients for an upstream gradient of 1; one pass applies the per-image constant of
        # the mean-normalised smoothness term and each unit's upstream gradient (its own + the sum's)
        g_sum = rest[-1] if ctx.want_sum else None
        tail = ((g_sum,) if ctx.has_sum_in else ()) + (None,) * ctx.n_tok          # d sum / d sum_in = 1
        if g_losses is None and g_sum is None:
            return (None,) * (1 + ctx.n * ctx.per) + tail
        g_raw, gT_raw, stats = ctx.saved_tensors
        n, S = ctx.n, ctx.S
        _, B, _, H, W = g_raw.shape
        if g_losses is not None:
            g_losses = _c(g_losses).reshape(n)
        if g_sum is not None:
            g_sum = _c(g_sum.float()).reshape(1)
        sinks = ctx.sinks
        g_disp = torch.empty_like(g_raw) if any(s is None for s in sinks) else None
        g_T = torch.empty_like(gT_raw)
        descs = (nat.UnitScaleDesc * n)()
        for u in range(n):
            d = descs[u]
            d.g_T_raw, d.stats = gT_raw[u].data_ptr(), stats[u].data_ptr()
            d.g_loss = (g_losses.data_ptr() + 4 * u) if g_losses is not None else None
            d.g_sum = g_sum.data_ptr() if g_sum is not None else None
            d.g_T = g_T[u].data_ptr()
            if sinks[u] is None:
                d.g_disp_raw, d.in_stride = g_raw[u].data_ptr(), H * W
                d.g_disp, d.out_stride = g_disp[u].data_ptr(), H * W
            else:
                # the disparity head this unit read in place takes the raw gradient and scales it on load
                sink, first, step = sinks[u]
                sink.entries.append(dict(raw=g_raw[u], stats=stats[u], smoothness=float(ctx.smoothness),
                                         g_loss=(g_losses, u) if g_losses is not None else None, g_sum=g_sum,
                                         first=int(first), step=int(step)))
        nat.check(nat.lib().mvf_units_fwdbwd_scale(C.cast(descs, C.c_void_p), n, ctx.smoothness, B, S, H, W,
                                                   _stream()), "units_fwdbwd_scale")
        grads = [None]
        for u in range(n):
            gu = [None] * ctx.per
            if ctx.needs[u][0] and sinks[u] is None:
                gu[0] = g_disp[u]
            if ctx.needs[u][1]:
                gu[2] = g_T[u]
            grads += gu
        return tuple(grads) + tail


class Unit:
    """One hot-path unit: S x generate_images_pred + compute_losses_base with the warped
    images kept in LDS (reference: train.py:956-1051).
    inputs: disp, tgt, T [S,B,4,4], K, inv_K, mask_rec|None, noise|None, cfg, *src(S)
    outputs: loss (0-dim), auto_mask [B,1,H,W] (or None), argmin uint8
    Training route (a gradient is wanted, S <= 2): `Units` with one unit; this class holds the
    route through the separate forward / backward kernels (S > 2, no gradient, UNIT_FWDBWD off)."""

    @staticmethod
    def apply(disp, tgt, T, K, inv_K, mask_rec, noise, cfg, *src):     # noqa: D102
        S = cfg[0]
        if unit_uses_fwdbwd(S, disp, T):
            flags, smoothness, min_depth, max_depth, eps, want_mask, want_idx = cfg[1:8]
            ucfg = dict(n=1, S=S, flags=flags, smoothness=smoothness, min_depth=min_depth,
                        max_depth=max_depth, eps=eps, want_mask=want_mask, want_idx=want_idx,
                        noise_out=[cfg[8]] if len(cfg) > 8 and cfg[8] is not None else None,
                        mean_parts=[cfg[9]] if len(cfg) > 9 and cfg[9] is not None else None)
            losses, terms, auto_mask, argmin, idx, _ = Units.apply(
                ucfg, disp, tgt, T, K, inv_K, mask_rec, noise, None, *src)
            return losses[0], auto_mask, argmin, idx, terms[0]
        return _UnitStaged.apply(disp, tgt, T, K, inv_K, mask_rec, noise, cfg, *src)


class _UnitStaged(torch.autograd.Function):
    """The unit through mvf_unit_fwd / mvf_unit_bwd (separate forward and backward kernels)."""

    @staticmethod
    def forward(ctx, disp, tgt, T, K, inv_K, mask_rec, noise, cfg, *src):
        S, flags, smoothness, min_depth, max_depth, eps, want_mask, want_idx = cfg[:8]
        src = [_c(t) for t in src]
        nat.require_device(disp, tgt, T, K, inv_K, mask_rec, noise, *src)
        disp, tgt, T, K, inv_K = _c(disp), _c(tgt), _c(T), _c(K), _c(inv_K)
        mask_rec, noise = _c(mask_rec), _c(noise)
        B, _, H, W = disp.shape
        dev = disp.device
        md, rg = depth_consts(min_depth, max_depth)
        loss = torch.empty(3, dtype=torch.float32, device=dev)
        argmin = torch.empty((B, H, W), dtype=torch.uint8, device=dev)
        auto_mask = torch.empty((B, 1, H, W), dtype=torch.float32, device=dev) if want_mask else None
        idx = torch.empty((S, B, H, W, 2), dtype=torch.int32, device=dev) if want_idx else None
        stats = torch.empty((B, 4), dtype=torch.float32, device=dev)
        ws = _ws(disp, B, H, W)
        sp, skeep = nat.ptr_array(src)
        if noise is None and not (flags & NO_AUTOMASK):
            raise RuntimeError("mvf_unit_fwd needs the tie-break noise tensor (only the "
                               "forward+backward kernel draws it itself)")
        nat.check(nat.lib().mvf_unit_fwd(nat.ptr(disp), nat.ptr(tgt), sp, nat.ptr(T), nat.ptr(K),
                                         nat.ptr(inv_K), nat.ptr(noise), nat.ptr(mask_rec), S, flags,
                                         smoothness, md, rg, eps, nat.ptr(loss), nat.ptr(argmin),
                                         nat.ptr(auto_mask), None, nat.ptr(stats), nat.ptr(idx),
                                         nat.ptr(ws), B, H, W, _stream()), "unit_fwd")
        ctx.save_for_backward(disp, tgt, T, K, inv_K, mask_rec, argmin, stats, *src)
        ctx.cfg = (S, flags, smoothness, md, rg, eps)
        outs = [loss[0], auto_mask if want_mask else torch.empty(0, device=dev), argmin,
                idx if want_idx else torch.empty(0, device=dev), loss[1:]]
        ctx.mark_non_differentiable(*outs[1:])
        ctx.set_materialize_grads(False)     # no zero tensors for the outputs without a gradient
        return tuple(outs)

    @staticmethod
    def backward(ctx, g_loss, *_unused):
        disp, tgt, T, K, inv_K, mask_rec, argmin, stats, *src = ctx.saved_tensors
        S, flags, smoothness, md, rg, eps = ctx.cfg
        if g_loss is None:
            return (None,) * (8 + S)
        B, _, H, W = disp.shape
        g_loss = _c(g_loss).reshape(1)
        g_disp = torch.empty_like(disp)
        g_T = torch.empty_like(T)
        ws = _ws(disp, B, H, W)
        sp, skeep = nat.ptr_array(src)
        nat.check(nat.lib().mvf_unit_bwd(nat.ptr(disp), nat.ptr(tgt), sp, nat.ptr(T), nat.ptr(K),
                                         nat.ptr(inv_K), nat.ptr(argmin), nat.ptr(mask_rec),
                                         nat.ptr(stats), nat.ptr(g_loss), S, flags, smoothness, md,
                                         rg, eps, nat.ptr(g_disp), nat.ptr(g_T), nat.ptr(ws), B, H, W,
                                         _stream()), "unit_bwd")
        return (g_disp, None, g_T, None, None, None, None, None, *([None] * S))


# ------------------------------------------------------------------ a10 pose glue
class Pose(torch.autograd.Function):
    """layers.transformation_from_parameters; reference: layers.py:28-103"""

    @staticmethod
    def forward(ctx, axisangle, translation, invert):
        nat.require_device(axisangle, translation)
        aa = _c(axisangle).reshape(-1, 3)
        tr = _c(translation).reshape(-1, 3)
        B = aa.shape[0]
        M = torch.empty((B, 4, 4), dtype=torch.float32, device=aa.device)
        nat.check(nat.lib().mvf_pose_fwd(nat.ptr(aa), nat.ptr(tr), nat.ptr(M), int(invert), B,
                                         _stream()), "pose_fwd")
        ctx.save_for_backward(aa, tr)
        ctx.meta = (int(invert), axisangle.shape, translation.shape)
        return M

    @staticmethod
    def backward(ctx, gM):
        aa, tr = ctx.saved_tensors
        invert, sa, st = ctx.meta
        gM = _c(gM)
        g_aa, g_tr = torch.empty_like(aa), torch.empty_like(tr)
        nat.check(nat.lib().mvf_pose_bwd(nat.ptr(aa), nat.ptr(tr), nat.ptr(gM), nat.ptr(g_aa),
                                         nat.ptr(g_tr), invert, aa.shape[0], _stream()), "pose_bwd")
        return g_aa.reshape(sa), g_tr.reshape(st), None


# ------------------------------------------------------------------ f1 flow warp
_LINSPACE = {}


def _linspace(n, device):
    """torch.linspace(-1, 1, n) evaluated on the CPU (the reference builds its base grid
    there, networks/IFRNet.py:9-10) and cached per device."""
    key = (n, str(device))
    t = _LINSPACE.get(key)
    if t is None:
        t = torch.linspace(-1.0, 1.0, n).to(device)
        _LINSPACE[key] = t
    return t


FLOW_WARP_BWD_GATHER = True     # False: float-atomic scatter for grad_img; tests compare both


class FlowWarp(torch.autograd.Function):
    """IFRNet.warp(img, flow); reference: networks/IFRNet.py:7-15.  grad_img is a deterministic
    gather through sorted inverse tap lists (mvf_flow_warp_bwd_gather); grad_flow is per-pixel."""

    @staticmethod
    def forward(ctx, img, flow):
        nat.require_device(img, flow)
        img, flow = _c(img), _c(flow)
        B, Cc, H, W = img.shape
        if tuple(flow.shape) != (B, 2, H, W):
            raise RuntimeError(f"flow must be [B,2,H,W] matching img, got {tuple(flow.shape)}")
        xs, ys = _linspace(W, img.device), _linspace(H, img.device)
        out = torch.empty_like(img)
        nat.check(nat.lib().mvf_flow_warp_fwd(nat.ptr(img), nat.ptr(flow), nat.ptr(xs), nat.ptr(ys),
                                              nat.ptr(out), None, B, Cc, H, W, _stream()), "flow_warp")
        ctx.save_for_backward(img, flow, xs, ys)
        return out

    @staticmethod
    def backward(ctx, g_out):
        img, flow, xs, ys = ctx.saved_tensors
        B, Cc, H, W = img.shape
        g_out = _c(g_out)
        gather = FLOW_WARP_BWD_GATHER and H >= 2 and W >= 2
        g_img = None
        if ctx.needs_input_grad[0]:
            if gather:
                g_img = torch.empty_like(img)
                iw = torch.empty(nat.lib().mvf_fusion_bwd_workspace_ints(B, H, W), dtype=torch.int32, device=img.device)
                nat.check(nat.lib().mvf_flow_warp_bwd_gather(nat.ptr(flow), nat.ptr(xs), nat.ptr(ys), nat.ptr(g_out),
                                                             nat.ptr(g_img), nat.ptr(iw), B, Cc, H, W, _stream()),
                          "flow_warp_bwd_gather")
            else:
                g_img = torch.zeros_like(img)
        g_flow = torch.empty_like(flow) if ctx.needs_input_grad[1] else None
        if g_flow is not None or (g_img is not None and not gather):
            ws = None
            if g_flow is not None:
                n = nat.lib().mvf_flow_warp_workspace_floats(B, Cc, H, W)
                ws = torch.empty(n, dtype=torch.float32, device=img.device)
            nat.check(nat.lib().mvf_flow_warp_bwd(nat.ptr(img), nat.ptr(flow), nat.ptr(xs), nat.ptr(ys),
                                                  nat.ptr(g_out), nat.ptr(None if gather else g_img),
                                                  nat.ptr(g_flow), nat.ptr(ws), B, Cc, H, W, _stream()),
                      "flow_warp_bwd")
        return g_img, g_flow


def flow_warp(img, flow):
    return FlowWarp.apply(img, flow)


def flow_warp_indices(img_shape, flow):
    """int32 [B,H,W,2] top-left taps of the flow warp (parity tests)."""
    nat.require_device(flow)
    B, Cc, H, W = img_shape
    flow = _c(flow)
    idx = torch.empty((B, H, W, 2), dtype=torch.int32, device=flow.device)
    nat.check(nat.lib().mvf_flow_warp_fwd(None, nat.ptr(flow), nat.ptr(_linspace(W, flow.device)),
                                          nat.ptr(_linspace(H, flow.device)), None, nat.ptr(idx), B, Cc,
                                          H, W, _stream()), "flow_warp(idx)")
    return idx


# ------------------------------------------------------------------ f1 fusion module
EMB_CH = 42     # Embedder: 2 * (1 + 2 * 10) channels (reference fusion_module.py:43-52)
FUSION_BWD_GATHER = True     # False: the atomic scatter (mvf_fusion_level_bwd); tests compare both
FUSION_BWD_ANCHOR = True     # deterministic route: anchor lists built once per step for all levels (round 5);
                             # False: the round-4 per-level cell lists (mvf_fusion_level_bwd_gather)


class PrepList(list):
    """The per-level side tensors of `fusion_prep`, with `.lists`: the `FusionLists` of the same call (pass
    `lists=(preps.lists, i)` to `fusion_level` and the backward pass builds the tap lists of all levels at once)."""
    lists = None


class FusionLists:
    """The inverse (anchor) tap lists of every pyramid level of one FusionModule call, built by ONE count / scan /
    fill / sort pass the first time a level's backward asks for them (they depend only on the teacher's flows)."""

    def __init__(self, preps):
        self.preps, self.lists = preps, None

    def level(self, i):
        if self.lists is None:
            preps = self.preps
            L, B, dev = len(preps), preps[0].shape[0], preps[0].device
            hs = (C.c_int32 * L)(*[int(p.shape[2]) for p in preps])
            ws = (C.c_int32 * L)(*[int(p.shape[3]) for p in preps])
            lib = nat.lib()
            lists = [torch.empty(lib.mvf_fusion_lists_level_ints(B, hs[k], ws[k]), dtype=torch.int32, device=dev)
                     for k in range(L)]
            scratch = torch.empty(lib.mvf_fusion_lists_scratch_ints(B, L, hs, ws), dtype=torch.int32, device=dev)
            xs = [_linspace(int(ws[k]), dev) for k in range(L)]
            ys = [_linspace(int(hs[k]), dev) for k in range(L)]
            arr = lambda ts: (C.c_void_p * L)(*[t.data_ptr() for t in ts])  # noqa: E731
            nat.check(lib.mvf_fusion_lists_build(arr(preps), arr(xs), arr(ys), hs, ws, L, B, arr(lists),
                                                 nat.ptr(scratch), _stream()), "fusion_lists_build")
            self.lists = lists
        return self.lists[i]


def fusion_prep(flow_n1, flow_p1, mask, sizes, litemono=False):
    """Per-level side tensors [B,9,h,w] of the FusionModule (cascaded embedding flows, resized
    warp flows, resized merge mask; reference fusion_module.py:65-103) for the pyramid
    ``sizes`` = [(h, w), ...] finest first.  Teacher outputs: no gradient."""
    nat.require_device(flow_n1, flow_p1, mask)
    flow_n1, flow_p1, mask = _c(flow_n1.detach()), _c(flow_p1.detach()), _c(mask.detach())
    B, _, Hf, Wf = flow_n1.shape
    if tuple(flow_p1.shape) != (B, 2, Hf, Wf) or tuple(mask.shape) != (B, 1, Hf, Wf):
        raise RuntimeError("fusion_prep: flows must be [B,2,H,W] and the mask [B,1,H,W]")
    preps, prev, ph, pw = [], None, 0, 0
    eh, ew = Hf, Wf
    for i, (h, w) in enumerate(sizes):
        halv = 2 if (i == 0 and litemono) else 1
        for _ in range(halv):
            eh, ew = eh // 2, ew // 2
        if (eh, ew) != (h, w):
            raise RuntimeError(f"fusion_prep: level {i} is {h}x{w} but the flow cascade gives {eh}x{ew}")
        prep = torch.empty((B, 9, h, w), dtype=torch.float32, device=flow_n1.device)
        nat.check(nat.lib().mvf_fusion_prep(nat.ptr(flow_n1), nat.ptr(flow_p1), nat.ptr(mask), nat.ptr(prev),
                                            nat.ptr(prep), B, h, w, Hf, Wf, ph, pw, halv, _stream()),
                  "fusion_prep")
        preps.append(prep)
        prev, ph, pw = prep, h, w
    out = PrepList(preps)
    out.lists = FusionLists(list(preps))    # (no reference back to `out`: nothing here forms a cycle)
    return out


class FusionLevel(torch.autograd.Function):
    """One pyramid level of FusionModule.forward up to the 1x1 convolution
    (reference: networks/fusion_module.py:105-127)."""

    @staticmethod
    def forward(ctx, feat_0, feat_n1, feat_p1, prep, plan=None):
        nat.require_device(feat_0, feat_n1, feat_p1, prep)
        feat_0, feat_n1, feat_p1, prep = _c(feat_0), _c(feat_n1), _c(feat_p1), _c(prep)
        B, Cc, h, w = feat_0.shape
        if feat_n1.shape != feat_0.shape or feat_p1.shape != feat_0.shape or tuple(prep.shape) != (B, 9, h, w):
            raise RuntimeError("fusion level: feature maps must share [B,C,h,w] and prep be [B,9,h,w]")
        xs, ys = _linspace(w, feat_0.device), _linspace(h, feat_0.device)
        out = torch.empty((B, 2 * (Cc + EMB_CH), h, w), dtype=torch.float32, device=feat_0.device)
        nat.check(nat.lib().mvf_fusion_level_fwd(nat.ptr(feat_0), nat.ptr(feat_n1), nat.ptr(feat_p1),
                                                 nat.ptr(prep), nat.ptr(xs), nat.ptr(ys), nat.ptr(out), B, Cc, h,
                                                 w, _stream()), "fusion_level_fwd")
        ctx.save_for_backward(prep, xs, ys)
        ctx.dims = (B, Cc, h, w)
        ctx.plan = plan
        return out

    @staticmethod
    def backward(ctx, g):
        prep, xs, ys = ctx.saved_tensors
        B, Cc, h, w = ctx.dims
        g = _c(g)
        g0 = g[:, :Cc] if ctx.needs_input_grad[0] else None
        if FUSION_BWD_GATHER:
            # deterministic: inverse tap lists (integer work) + a gather per channel, no float atomics
            gn = torch.empty((B, Cc, h, w), dtype=torch.float32, device=g.device) if ctx.needs_input_grad[1] else None
            gp = torch.empty((B, Cc, h, w), dtype=torch.float32, device=g.device) if ctx.needs_input_grad[2] else None
            if FUSION_BWD_ANCHOR and (gn is not None or gp is not None):
                plan, lvl = ctx.plan if ctx.plan is not None else (FusionLists([prep]), 0)
                lists = plan.level(lvl)
                nat.check(nat.lib().mvf_fusion_level_bwd_lists(nat.ptr(g), nat.ptr(lists), nat.ptr(gn), nat.ptr(gp),
                                                               B, Cc, h, w, _stream()), "fusion_level_bwd_lists")
                return g0, gn, gp, None, None
            ws = torch.empty(nat.lib().mvf_fusion_bwd_workspace_ints(B, h, w), dtype=torch.int32, device=g.device)
            nat.check(nat.lib().mvf_fusion_level_bwd_gather(nat.ptr(g), nat.ptr(prep), nat.ptr(xs), nat.ptr(ys),
                                                            nat.ptr(gn), nat.ptr(gp), nat.ptr(ws), B, Cc, h, w,
                                                            _stream()), "fusion_level_bwd_gather")
            return g0, gn, gp, None, None
        gn = torch.zeros((B, Cc, h, w), dtype=torch.float32, device=g.device) if ctx.needs_input_grad[1] else None
        gp = torch.zeros((B, Cc, h, w), dtype=torch.float32, device=g.device) if ctx.needs_input_grad[2] else None
        nat.check(nat.lib().mvf_fusion_level_bwd(nat.ptr(g), nat.ptr(prep), nat.ptr(xs), nat.ptr(ys),
                                                 nat.ptr(gn), nat.ptr(gp), B, Cc, h, w, _stream()),
                  "fusion_level_bwd")
        return g0, gn, gp, None, None


def fusion_level(feat_0, feat_n1, feat_p1, prep, lists=None):
    """`lists`: (FusionLists, level index) of the `fusion_prep` call `prep` came from (`preps.lists`), so that the
    backward pass builds the inverse tap lists of all levels in one pass; without it each level builds its own."""
    return FusionLevel.apply(feat_0, feat_n1, feat_p1, prep, lists)


# ------------------------------------------------------------------ f2 SI-log depth loss
class SILog(torch.autograd.Function):
    """Trainer.compute_SI_log_depth_loss; reference: train.py:924-941"""

    @staticmethod
    def forward(ctx, pred, target, mask, beta):
        nat.require_device(pred, target, mask)
        pred, target, mask = _c(pred), _c(target), _c(mask)
        B = pred.shape[0]
        N = pred[0].numel()
        if pred.shape[1] != 1 or target.shape != pred.shape:
            raise RuntimeError("compute_SI_log_depth_loss expects pred/target [B,1,H,W]")
        if mask is not None and mask.shape != pred.shape:
            # the kernel indexes the mask as [B,N]; the reference multiplies, so anything
            # broadcastable is legal there (train.py:930-933)
            mask = mask.expand(pred.shape).contiguous()
        dev = pred.device
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        sums = torch.empty((B, 4), dtype=torch.float32, device=dev)
        ws = torch.empty(B * 64 * 4, dtype=torch.float32, device=dev)
        nat.check(nat.lib().mvf_silog_fwd(nat.ptr(pred), nat.ptr(target), nat.ptr(mask), nat.ptr(loss),
                                          nat.ptr(sums), nat.ptr(ws), B, N, float(beta), _stream()),
                  "silog_fwd")
        ctx.save_for_backward(pred, target, mask, sums)
        ctx.beta = float(beta)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        pred, target, mask, sums = ctx.saved_tensors
        B = pred.shape[0]
        N = pred[0].numel()
        g = _c(g).reshape(1)
        gp = torch.empty_like(pred) if ctx.needs_input_grad[0] else None
        gt = torch.empty_like(target) if ctx.needs_input_grad[1] else None
        nat.check(nat.lib().mvf_silog_bwd(nat.ptr(pred), nat.ptr(target), nat.ptr(mask), nat.ptr(sums),
                                          nat.ptr(g), nat.ptr(gp), nat.ptr(gt), B, N, ctx.beta,
                                          _stream()), "silog_bwd")
        return gp, gt, None, None


def silog_loss(pred, target, mask=None, beta=0.5):
    return SILog.apply(pred, target, mask, beta)


def _img1(t):
    """[B,1,H,W] fp32 whose images are contiguous planes -> (tensor, image stride in floats): a view that strides over
    the batch only (one group of a grouped decoder call's interleaved output) is read in place."""
    t, st = _img(t)
    return t, st


class SILogMany(torch.autograd.Function):
    """Several Trainer.compute_SI_log_depth_loss evaluations (reference: train.py:924-941; process_batch adds nine of
    them into loss_dc, train.py:813-815, 868-882) as ONE forward and ONE backward launch.
    apply(beta, n, pred_0, target_0, mask_0 | None, pred_1, ...) -> (total (0-dim: the sum in job order), losses [n])."""

    @staticmethod
    def forward(ctx, beta, n, *flat):
        assert len(flat) == 3 * n and 1 <= n <= nat.MAX_SILOG_JOBS
        B, _, H, W = flat[0].shape
        N = H * W
        dev = flat[0].device
        jobs = (nat.SilogJob * n)()
        keep = []
        for j in range(n):
            pred, target, mask = flat[3 * j:3 * j + 3]
            nat.require_device(pred, target, mask)
            if pred.shape[1] != 1 or target.shape != pred.shape or tuple(pred.shape) != (B, 1, H, W):
                raise RuntimeError("compute_SI_log_depth_loss expects pred/target [B,1,H,W], one shape per launch")
            if mask is not None and mask.shape != pred.shape:
                mask = mask.expand(pred.shape).contiguous()      # the reference multiplies: anything broadcastable
            (pred, ps), (target, ts) = _img1(pred), _img1(target)
            mask, ms = _img1(mask) if mask is not None else (None, 0)
            d = jobs[j]
            d.pred, d.pred_stride, d.target, d.target_stride = pred.data_ptr(), ps, target.data_ptr(), ts
            d.mask, d.mask_stride = (mask.data_ptr(), ms) if mask is not None else (None, 0)
            keep += [pred, target, mask]
        losses = torch.empty(n, dtype=torch.float32, device=dev)
        total = torch.empty((), dtype=torch.float32, device=dev)
        sums = torch.empty((n, B, 4), dtype=torch.float32, device=dev)
        ws = torch.empty(nat.lib().mvf_silog_many_workspace_floats(n, B), dtype=torch.float32, device=dev)
        tk = _tickets(dev, n + 1)
        try:
            nat.check(nat.lib().mvf_silog_many_fwd(C.cast(jobs, C.c_void_p), n, nat.ptr(losses), nat.ptr(total),
                                                   nat.ptr(sums), nat.ptr(ws), nat.ptr(tk), B, N, float(beta),
                                                   _stream()), "silog_many_fwd")
        except Exception:
            _drop_tickets(dev)
            raise
        ctx.save_for_backward(sums, *[t for t in keep if t is not None])
        ctx.layout = [(keep[3 * j + 2] is not None) for j in range(n)]
        ctx.strides = [(jobs[j].pred_stride, jobs[j].target_stride, jobs[j].mask_stride) for j in range(n)]
        ctx.beta, ctx.n, ctx.dims = float(beta), n, (B, H, W)
        ctx.set_materialize_grads(False)
        return total, losses

    @staticmethod
    def backward(ctx, g_total, g_losses):
        n, (B, H, W) = ctx.n, ctx.dims
        if g_total is None and g_losses is None:
            return (None,) * (2 + 3 * n)
        sums, *rest = ctx.saved_tensors
        dev = sums.device
        g_total = _c(g_total.float()).reshape(1) if g_total is not None else None
        g_losses = _c(g_losses.float()).reshape(n) if g_losses is not None else None
        jobs = (nat.SilogJob * n)()
        grads = [None, None]
        it = iter(rest)
        outs = []
        for j in range(n):
            pred, target = next(it), next(it)
            mask = next(it) if ctx.layout[j] else None
            d = jobs[j]
            ps, ts, ms = ctx.strides[j]
            d.pred, d.pred_stride, d.target, d.target_stride = pred.data_ptr(), ps, target.data_ptr(), ts
            d.mask, d.mask_stride = (mask.data_ptr(), ms) if mask is not None else (None, 0)
            gp = torch.empty((B, 1, H, W), dtype=torch.float32, device=dev) if ctx.needs_input_grad[2 + 3 * j] else None
            gt = torch.empty((B, 1, H, W), dtype=torch.float32, device=dev) if ctx.needs_input_grad[3 + 3 * j] else None
            d.g_pred, d.g_target = nat.ptr(gp), nat.ptr(gt)
            grads += [gp, gt, None]
        nat.check(nat.lib().mvf_silog_many_bwd(C.cast(jobs, C.c_void_p), n, nat.ptr(sums), nat.ptr(g_total),
                                               nat.ptr(g_losses), B, H * W, ctx.beta, _stream()), "silog_many_bwd")
        return tuple(grads)


def silog_many(jobs, beta=0.5):
    """jobs: [(pred, target, mask | None), ...] of one shape [B,1,H,W] -> (sum of the losses, losses [n])."""
    flat = []
    for pred, target, mask in jobs:
        flat += [pred, target, mask]
    return SILogMany.apply(float(beta), len(jobs), *flat)


# --------------------------------------------------------------------------- f2 affine glue
def _affine_meta(angle, box, ratio, B, dev):
    """angle [B] or [B,1] degrees fp32, box [B,4] (x0,y0,w,h) int32, ratio [B] or [B,1] fp32 --
    all kept on the device (the reference reads them with .item() per sample)."""
    angle = angle.to(device=dev, dtype=torch.float32).reshape(-1).contiguous()
    box = box.to(device=dev, dtype=torch.int32).contiguous()
    if angle.numel() != B or tuple(box.shape) != (B, 4):
        raise RuntimeError("angle must hold B values and box must be [B,4]")
    if ratio is not None:
        ratio = ratio.to(device=dev, dtype=torch.float32).reshape(-1).contiguous()
        if ratio.numel() != B:
            raise RuntimeError("ratio_local must hold B values")
    return angle, box, ratio


def affine_transform(img, angle, box):
    """Trainer.affine_transform (reference: train.py:888-902): rotate by ``angle`` degrees
    (bilinear, zero fill), crop ``box`` and resize back to (H,W), for the whole batch in one
    launch.  Forward only: the trainer applies it to teacher frames."""
    nat.require_device(img)
    if img.requires_grad:
        raise RuntimeError("affine_transform has no backward (inputs are teacher frames)")
    img = _c(img)
    B, C, H, W = img.shape
    # `img` may hold several views per sample, concatenated along the batch ([V*B_meta, ...], image i uses the
    # parameters of sample i % B_meta): the two teacher frames of a step in one launch
    Bm = angle.numel()
    if Bm < 1 or B % Bm:
        raise RuntimeError("angle must hold B values (or B / V for V concatenated views per sample)")
    angle, box, _ = _affine_meta(angle, box, None, Bm, img.device)
    out = torch.empty_like(img)
    nat.check(nat.lib().mvf_affine_transform_views_fwd(nat.ptr(img), nat.ptr(angle), nat.ptr(box), nat.ptr(out),
                                                       B, Bm, C, H, W, _stream()), "affine_transform_fwd")
    return out


class AffineRestore(torch.autograd.Function):
    """depth_restore of Trainer.compute_depth_consistency_loss_affine; reference: train.py:909-916"""

    @staticmethod
    def forward(ctx, depth, angle, box, ratio):
        nat.require_device(depth)
        depth = _c(depth)
        B, C, H, W = depth.shape
        angle, box, ratio = _affine_meta(angle, box, ratio, B, depth.device)
        out = torch.empty_like(depth)
        nat.check(nat.lib().mvf_affine_restore_fwd(nat.ptr(depth), nat.ptr(angle), nat.ptr(box),
                                                   nat.ptr(ratio), nat.ptr(out), B, C, H, W, _stream()),
                  "affine_restore_fwd")
        ctx.save_for_backward(angle, box, ratio)
        return out

    @staticmethod
    def backward(ctx, g):
        angle, box, ratio = ctx.saved_tensors
        g = _c(g)
        B, C, H, W = g.shape
        ws = torch.empty_like(g)
        gd = torch.empty_like(g)
        nat.check(nat.lib().mvf_affine_restore_bwd(nat.ptr(g), nat.ptr(angle), nat.ptr(box), nat.ptr(ratio),
                                                   nat.ptr(ws), nat.ptr(gd), B, C, H, W, _stream()),
                  "affine_restore_bwd")
        return gd, None, None, None


def affine_restore(depth, angle, box, ratio):
    return AffineRestore.apply(depth, angle, box, ratio)


class AffineRestoreMany(torch.autograd.Function):
    """depth_restore (train.py:909-916) of G depth maps [B,1,H,W] that share angle / box / ratio -- the three affine
    views of a step (train.py:868-882) -- as ONE launch forward and one rotate + one resize adjoint launch backward:
    the maps are the channels of one image.  Maps that lie one plane apart in memory (consecutive groups of a grouped
    decoder call's interleaved output) are read in place.  Returns G tensors [B,1,H,W]."""

    @staticmethod
    def forward(ctx, angle, box, ratio, *depths):
        nat.require_device(*depths)
        G = len(depths)
        B, C1, H, W = depths[0].shape
        N = H * W
        if C1 != 1 or any(tuple(d.shape) != (B, 1, H, W) or d.dtype != torch.float32 for d in depths):
            raise RuntimeError("affine_restore_many expects fp32 depth maps of one shape [B,1,H,W]")
        angle, box, ratio = _affine_meta(angle, box, ratio, B, depths[0].device)
        d0 = depths[0]
        st0 = d0.stride(0) if B > 1 else G * N
        in_place = all(d.untyped_storage().data_ptr() == d0.untyped_storage().data_ptr() and
                       d.storage_offset() == d0.storage_offset() + g * N and
                       (d.stride(0) if B > 1 else st0) == st0 and d.stride(2) == W and d.stride(3) == 1
                       for g, d in enumerate(depths)) and st0 >= G * N
        if in_place:
            src, stride = d0, st0
        else:
            src, stride = torch.cat([_c(d) for d in depths], 1), G * N
        out = torch.empty((B, G, H, W), dtype=torch.float32, device=d0.device)
        nat.check(nat.lib().mvf_affine_restore_strided_fwd(src.data_ptr(), stride, nat.ptr(angle), nat.ptr(box),
                                                           nat.ptr(ratio), nat.ptr(out), B, G, H, W, _stream()),
                  "affine_restore_fwd")
        ctx.save_for_backward(angle, box, ratio)
        ctx.G = G
        ctx.set_materialize_grads(False)
        return tuple(out[:, g:g + 1] for g in range(G))

    @staticmethod
    def backward(ctx, *gs):
        angle, box, ratio = ctx.saved_tensors
        G = ctx.G
        if all(g is None for g in gs):
            return (None,) * (3 + G)
        ref = next(g for g in gs if g is not None)
        B, _, H, W = ref.shape
        g = torch.cat([_c(x) if x is not None else torch.zeros_like(ref) for x in gs], 1)
        ws = torch.empty_like(g)
        gd = torch.empty_like(g)
        nat.check(nat.lib().mvf_affine_restore_bwd(nat.ptr(g), nat.ptr(angle), nat.ptr(box), nat.ptr(ratio),
                                                   nat.ptr(ws), nat.ptr(gd), B, G, H, W, _stream()),
                  "affine_restore_bwd")
        return (None, None, None) + tuple(gd[:, k:k + 1] for k in range(G))


def affine_restore_many(depths, angle, box, ratio):
    return AffineRestoreMany.apply(angle, box, ratio, *depths)


# --------------------------------------------------------------------------- Conv3x3 glue
class ReflectPad1(torch.autograd.Function):
    """nn.ReflectionPad2d(1) of Conv3x3; reference: layers.py:121-138"""

    @staticmethod
    def forward(ctx, x):
        nat.require_device(x)
        x = _c(x)
        B, C, H, W = x.shape
        if H < 2 or W < 2:
            raise RuntimeError("ReflectionPad2d(1) needs H, W >= 2")
        out = torch.empty((B, C, H + 2, W + 2), dtype=torch.float32, device=x.device)
        nat.check(nat.lib().mvf_reflect_pad1_fwd(nat.ptr(x), nat.ptr(out), B * C, H, W, _stream()),
                  "reflect_pad1_fwd")
        return out

    @staticmethod
    def backward(ctx, g):
        g = _c(g)
        B, C, Hp, Wp = g.shape
        gi = torch.empty((B, C, Hp - 2, Wp - 2), dtype=torch.float32, device=g.device)
        nat.check(nat.lib().mvf_reflect_pad1_bwd(nat.ptr(g), nat.ptr(gi), B * C, Hp - 2, Wp - 2, _stream()),
                  "reflect_pad1_bwd")
        return gi


def reflect_pad1(x):
    return ReflectPad1.apply(x)



# --------------------------------------------------------------------------- f4 step glue
class Up2CatPad(torch.autograd.Function):
    """ReflectionPad2d(1)(cat([upsample_nearest_x2(x), skip], 1)): the padded input of a decoder
    stage's second convolution in one pass (reference: networks/monodepth2.py:84-90,
    layers.py:121-138, 225-228).  skip may be None."""

    @staticmethod
    def forward(ctx, x, skip):
        nat.require_device(x, skip)
        x, skip = _c(x), _c(skip)
        B, C1, h, w = x.shape
        C2 = 0 if skip is None else skip.shape[1]
        if skip is not None and (skip.shape[0] != B or tuple(skip.shape[2:]) != (2 * h, 2 * w)):
            raise RuntimeError(f"skip feature must be [B,C,{2 * h},{2 * w}], got {tuple(skip.shape)}")
        if h < 2 or w < 2:
            raise RuntimeError("decoder glue needs h, w >= 2")
        out = torch.empty((B, C1 + C2, 2 * h + 2, 2 * w + 2), dtype=torch.float32, device=x.device)
        nat.check(nat.lib().mvf_up2cat_pad_fwd(nat.ptr(x), nat.ptr(skip), nat.ptr(out), B, C1, C2, h, w,
                                               _stream()), "up2cat_pad_fwd")
        ctx.dims = (B, C1, C2, h, w)
        return out

    @staticmethod
    def backward(ctx, g):
        B, C1, C2, h, w = ctx.dims
        g = _c(g)
        gx = torch.empty((B, C1, h, w), dtype=torch.float32, device=g.device) if ctx.needs_input_grad[0] else None
        gs = (torch.empty((B, C2, 2 * h, 2 * w), dtype=torch.float32, device=g.device)
              if C2 and ctx.needs_input_grad[1] else None)
        nat.check(nat.lib().mvf_up2cat_pad_bwd(nat.ptr(g), nat.ptr(gx), nat.ptr(gs), B, C1, C2, h, w, _stream()),
                  "up2cat_pad_bwd")
        return gx, gs


def up2cat_pad(x, skip=None):
    return Up2CatPad.apply(x, skip)


class HeadSink:
    """What a disparity head hands to the hot-path units that read its output in place: during backward a unit
    deposits its RAW disparity gradient here (`Units.backward`) instead of returning a scaled tensor to autograd, and
    the head's adjoint kernel applies `(raw - shift_b) * g` on load (`mvf_disp_head_bwd_units`) -- no k_fb_scale pass,
    no `stack` of the group gradients.  `token` is an (empty) differentiable output of the head every such unit launch
    takes as an input: the autograd edge that makes the engine run the head's backward AFTER those units' backward."""

    def __init__(self, disp, token, entries):
        # (no reference to `disp` or to the autograd node: the node holds `entries`, this object holds the token --
        # a cycle through the node would keep the head's tensors alive until the garbage collector runs)
        self.token, self.entries = token, entries
        self.storage, self.offset, self.shape = disp.untyped_storage().data_ptr(), disp.storage_offset(), tuple(disp.shape)
        # units that have claimed this head in the current forward: the head's adjoint kernel takes at most MAX_UNITS of them;
        # a further one gets `None` from claim() and returns its (scaled) gradient through autograd as any other consumer
        # does (the backward used to raise)
        self.claimed = 0

    def claim(self, view):
        """covers(view) if this head can still take a deferred unit, else None (the caller falls back)."""
        where = self.covers(view)
        if where is None or self.claimed >= nat.MAX_UNITS:
            return None
        self.claimed += 1
        return where

    def covers(self, view):
        """(first image, image step) of `view` [B,1,H,W] inside the head's batch, or None if it is not such a view."""
        if view is None or view.dtype != torch.float32 or view.dim() != 4 or tuple(view.shape[1:]) != self.shape[1:]:
            return None
        if view.untyped_storage().data_ptr() != self.storage:
            return None
        N = self.shape[1] * self.shape[2] * self.shape[3]
        st = view.stride()
        if st[3] != 1 or st[2] != view.shape[3]:
            return None
        off = view.storage_offset() - self.offset
        step = st[0] if view.shape[0] > 1 else N
        if off < 0 or off % N or step % N or step < N:
            return None
        first, step = off // N, step // N
        if first + (view.shape[0] - 1) * step >= self.shape[0]:
            return None
        return first, step


def pad_act_ok(x_like, H, W):
    """The fused pad + bias + ELU kernels take wide shapes only (W % 4 == 0, W >= 8, H >= 4, 16-byte aligned)."""
    # (planes: the adjoint kernels put B x C in grid.y -- ADVICE r05: the forward accepted more and the backward then
    # failed mid-step; beyond the limit the separate epilogue + pad kernels run)
    planes = x_like.shape[0] * x_like.shape[1] if x_like.dim() == 4 else 1 << 30
    return bool(x_like.is_cuda and x_like.dtype == torch.float32 and x_like.is_contiguous() and planes <= 65535 and
                x_like.data_ptr() % 16 == 0 and nat.lib().mvf_pad_act_supported(int(H), int(W)))


class ReflectPad1Act(torch.autograd.Function):
    """ReflectionPad2d(1)(ELU(y + bias)) for a raw convolution output y: the ConvBlock epilogue (layers.py:106-118)
    applied inside the next Conv3x3's pad kernel (layers.py:121-138).  Backward: pad adjoint, ELU' from the padded
    tensor's interior and the bias gradient in one kernel."""

    @staticmethod
    def forward(ctx, y, bias):
        nat.require_device(y, bias)
        y, bias = _c(y), _c(bias)
        B, C, H, W = y.shape
        out = torch.empty((B, C, H + 2, W + 2), dtype=torch.float32, device=y.device)
        nat.check(nat.lib().mvf_reflect_pad1_act_fwd(nat.ptr(y), nat.ptr(bias), nat.ptr(out), B, C, H, W, _stream()),
                  "reflect_pad1_act_fwd")
        ctx.save_for_backward(out)
        return out

    @staticmethod
    def backward(ctx, g):
        (padded,) = ctx.saved_tensors
        g = _c(g)
        B, C, Hp, Wp = g.shape
        H, W = Hp - 2, Wp - 2
        gy = torch.empty((B, C, H, W), dtype=torch.float32, device=g.device)
        gb = torch.empty(C, dtype=torch.float32, device=g.device)
        ws = torch.empty(nat.lib().mvf_pad_act_workspace_floats(B, C, H, W), dtype=torch.float32, device=g.device)
        nat.check(nat.lib().mvf_reflect_pad1_act_bwd(nat.ptr(g), nat.ptr(padded), nat.ptr(gy), nat.ptr(gb), nat.ptr(ws),
                                                     B, C, H, W, _stream()), "reflect_pad1_act_bwd")
        return gy, gb


class Up2CatPadAct(torch.autograd.Function):
    """ReflectionPad2d(1)(cat([upsample_nearest_x2(ELU(y + bias)), skip], 1)) for a raw convolution output y
    (networks/monodepth2.py:84-90 after the ConvBlock of layers.py:106-118)."""

    @staticmethod
    def forward(ctx, y, bias, skip):
        nat.require_device(y, bias, skip)
        y, bias, skip = _c(y), _c(bias), _c(skip)
        B, C1, h, w = y.shape
        C2 = skip.shape[1] if skip is not None else 0
        if skip is not None and tuple(skip.shape) != (B, C2, 2 * h, 2 * w):
            raise RuntimeError("up2cat_pad: skip must be [B,C2,2h,2w]")
        out = torch.empty((B, C1 + C2, 2 * h + 2, 2 * w + 2), dtype=torch.float32, device=y.device)
        nat.check(nat.lib().mvf_up2cat_pad_act_fwd(nat.ptr(y), nat.ptr(bias), nat.ptr(skip), nat.ptr(out), B, C1, C2, h, w,
                                                   _stream()), "up2cat_pad_act_fwd")
        ctx.save_for_backward(out)
        ctx.dims = (B, C1, C2, h, w)
        return out

    @staticmethod
    def backward(ctx, g):
        (padded,) = ctx.saved_tensors
        B, C1, C2, h, w = ctx.dims
        g = _c(g)
        dev = g.device
        gy = torch.empty((B, C1, h, w), dtype=torch.float32, device=dev)
        gb = torch.empty(C1, dtype=torch.float32, device=dev)
        gs = torch.empty((B, C2, 2 * h, 2 * w), dtype=torch.float32, device=dev) if (C2 and ctx.needs_input_grad[2]) else None
        ws = torch.empty(nat.lib().mvf_up2cat_pad_act_workspace_floats(B, C1, h, w), dtype=torch.float32, device=dev)
        nat.check(nat.lib().mvf_up2cat_pad_act_bwd(nat.ptr(g), nat.ptr(padded), nat.ptr(gy), nat.ptr(gb), nat.ptr(gs),
                                                   nat.ptr(ws), B, C1, C2, h, w, _stream()), "up2cat_pad_act_bwd")
        return gy, gb, gs


def reflect_pad1_act(y, bias):
    return ReflectPad1Act.apply(y, bias)


def up2cat_pad_act(y, bias, skip=None):
    return Up2CatPadAct.apply(y, bias, skip)


class DispHead(torch.autograd.Function):
    """sigmoid (networks/monodepth2.py:93) fused with disp_to_depth (layers.py:16-25):
    logit [B,1,H,W] -> disp, depth (or None), per-image mean partials of disp [B,32] for the unit
    kernel (not differentiable), token (empty; see `HeadSink`)."""

    @staticmethod
    def forward(ctx, logit, min_depth, max_depth, want_depth, entries):
        nat.require_device(logit)
        logit = _c(logit)
        B = logit.shape[0]
        N = logit[0].numel()
        md, rg = depth_consts(min_depth, max_depth)
        disp = torch.empty_like(logit)
        depth = torch.empty_like(logit) if want_depth else None
        part = torch.empty((B, 32), dtype=torch.float32, device=logit.device)
        nat.check(nat.lib().mvf_disp_head_fwd(nat.ptr(logit), nat.ptr(disp), nat.ptr(depth), nat.ptr(part), B, N,
                                              md, rg, _stream()), "disp_head_fwd")
        ctx.save_for_backward(disp)
        ctx.consts = (md, rg)
        ctx.mark_non_differentiable(part)
        # no zero tensors for outputs without a gradient: `part` never has one, and a decoder output whose
        # depth is unused would otherwise hand backward() a full-size zero g_depth (a fill launch, and a
        # plane the adjoint kernel then reads)
        ctx.set_materialize_grads(False)
        token = logit.new_empty(0)
        ctx.entries = entries          # the list the units' backward fills (HeadSink.entries)
        return disp, (depth if want_depth else torch.empty(0, device=logit.device)), part, token

    @staticmethod
    def backward(ctx, g_disp, g_depth, _g_part, _g_token):
        (disp,) = ctx.saved_tensors
        md, rg = ctx.consts
        entries = ctx.entries
        if g_disp is None and g_depth is None and not entries:
            return None, None, None, None, None
        g_depth = _c(g_depth) if (g_depth is not None and g_depth.numel() == disp.numel()) else None
        g = torch.empty_like(disp)
        if entries:
            # the units that read this head's output in place: their raw gradients, scaled on load
            if len(entries) > nat.MAX_UNITS:
                raise RuntimeError("more than MAX_UNITS deferred units on one disparity head")
            g_disp = _c(g_disp) if g_disp is not None else None
            B, N = disp.shape[0], disp[0].numel()
            descs = (nat.HeadUnitGrad * len(entries))()
            for d, e in zip(descs, entries):
                d.g_disp_raw, d.raw_stride = e["raw"].data_ptr(), N
                d.stats = e["stats"].data_ptr()
                d.g_loss = (e["g_loss"][0].data_ptr() + 4 * e["g_loss"][1]) if e["g_loss"] is not None else None
                d.g_sum = e["g_sum"].data_ptr() if e["g_sum"] is not None else None
                d.smoothness, d.first, d.step, d.count = e["smoothness"], e["first"], e["step"], e["raw"].shape[0]
            nat.check(nat.lib().mvf_disp_head_bwd_units(
                nat.ptr(disp), nat.ptr(g_disp), nat.ptr(g_depth), nat.ptr(g), B, N, md, rg,
                C.cast(descs, C.c_void_p), len(entries), _stream()), "disp_head_bwd_units")
            del entries[:]
            return g, None, None, None, None
        g_disp = _c(g_disp) if g_disp is not None else torch.zeros_like(disp)
        nat.check(nat.lib().mvf_disp_head_bwd(nat.ptr(disp), nat.ptr(g_disp), nat.ptr(g_depth), nat.ptr(g),
                                              disp.numel(), md, rg, _stream()), "disp_head_bwd")
        return g, None, None, None, None


def disp_head(logit, min_depth=0.1, max_depth=100.0, want_depth=True, want_sink=False):
    """-> (disp, depth | None, mean_partials) and, with `want_sink`, a `HeadSink` as fourth element"""
    entries = []
    disp, depth, part, token = DispHead.apply(logit, min_depth, max_depth, bool(want_depth), entries)
    if want_sink:
        sink = HeadSink(disp, token, entries) if (disp.requires_grad and torch.is_grad_enabled()) else None
        return disp, (depth if want_depth else None), part, sink
    return disp, (depth if want_depth else None), part


class ResizeBilinear(torch.autograd.Function):
    """F.interpolate(x, mode="bilinear", align_corners=...) for feature maps (reference:
    networks/hrnet_encoder.py:275-280, networks/LiteMono.py:495, 502 via layers.py:225-228):
    one lane per output element forward, deterministic gather backward."""

    @staticmethod
    def forward(ctx, x, oh, ow, sh, sw, align):
        nat.require_device(x)
        x = _c(x)
        if x.dim() != 4:
            raise RuntimeError("resize_bilinear expects [N,C,H,W]")
        N, C, ih, iw = x.shape
        out = torch.empty((N, C, oh, ow), dtype=torch.float32, device=x.device)
        nat.check(nat.lib().mvf_resize_bilinear_fwd(nat.ptr(x), nat.ptr(out), N * C, ih, iw, oh, ow, sh, sw,
                                                    int(align), _stream()), "resize_bilinear_fwd")
        ctx.geom = (N, C, ih, iw, oh, ow, sh, sw, int(align))
        return out

    @staticmethod
    def backward(ctx, g):
        N, C, ih, iw, oh, ow, sh, sw, align = ctx.geom
        g = _c(g)
        gx = torch.empty((N, C, ih, iw), dtype=torch.float32, device=g.device)
        # rows of g folded first (N*C x ih x ow floats), then the columns: two coalesced passes instead of a
        # (2f+3) x (2f+1) candidate walk per input pixel
        ws = torch.empty(int(nat.lib().mvf_resize_bilinear_bwd_workspace_floats(N * C, ih, ow)), dtype=torch.float32,
                         device=g.device)
        nat.check(nat.lib().mvf_resize_bilinear_bwd(nat.ptr(g), nat.ptr(gx), nat.ptr(ws), N * C, ih, iw, oh, ow, sh, sw,
                                                    align, _stream()), "resize_bilinear_bwd")
        return gx, None, None, None, None, None


def resize_scales(ih, iw, oh, ow, align_corners, scale_factor=None):
    """ATen's area_pixel_compute_scale for both axes, as the fp32 values its kernels use."""
    f32 = np.float32

    def one(n_in, n_out):
        if align_corners:
            return float(f32(n_in - 1) / f32(n_out - 1)) if n_out > 1 else 0.0
        if scale_factor is not None and scale_factor > 0:
            return float(f32(1.0 / scale_factor))
        return float(f32(n_in) / f32(n_out))
    return one(ih, oh), one(iw, ow)


def resize_bilinear(x, size=None, scale_factor=None, align_corners=False):
    """Drop-in for F.interpolate(x, size= | scale_factor=, mode="bilinear", align_corners=...)."""
    ih, iw = x.shape[-2:]
    if size is not None:
        oh, ow = int(size[0]), int(size[1])
        sf = None
    else:
        oh, ow = int(np.floor(ih * float(scale_factor))), int(np.floor(iw * float(scale_factor)))
        sf = float(scale_factor)
    sh, sw = resize_scales(ih, iw, oh, ow, bool(align_corners), sf)
    return ResizeBilinear.apply(x, oh, ow, sh, sw, bool(align_corners))


class UpsampleNearest(torch.autograd.Function):
    """F.interpolate(x, scale_factor=f, mode="nearest"), integer f (reference layers.py:225-228)."""

    @staticmethod
    def forward(ctx, x, f):
        nat.require_device(x)
        x = _c(x)
        N, C, ih, iw = x.shape
        out = torch.empty((N, C, ih * f, iw * f), dtype=torch.float32, device=x.device)
        nat.check(nat.lib().mvf_upsample_nearest_fwd(nat.ptr(x), nat.ptr(out), N * C, ih, iw, f, _stream()),
                  "upsample_nearest_fwd")
        ctx.geom = (N, C, ih, iw, f)
        return out

    @staticmethod
    def backward(ctx, g):
        N, C, ih, iw, f = ctx.geom
        g = _c(g)
        gx = torch.empty((N, C, ih, iw), dtype=torch.float32, device=g.device)
        nat.check(nat.lib().mvf_upsample_nearest_bwd(nat.ptr(g), nat.ptr(gx), N * C, ih, iw, f, _stream()),
                  "upsample_nearest_bwd")
        return gx, None


def upsample_nearest(x, factor):
    return UpsampleNearest.apply(x, int(factor))


class Regroup(torch.autograd.Function):
    """Interleaved group batch -> the interleaved batches its consumers take (section 8f-3).

    ``src [B*G, ...]`` holds G independent invocations (sample n = b*G + g: networks/grouped.py);
    ``plans`` is a tuple of group-index tuples, one per consumer.  Output k is the interleaved batch
    ``[B*len(plans[k]), ...]`` of those groups -- what ``merge_groups([split_groups(src, G)[g] for g in
    plans[k]])`` builds, for every consumer in ONE launch.  The backward writes the gradient of ``src``
    once: per group the sum over the slots that read it, zeros where nobody did (reference: the
    feature pyramids of the encoder calls train.py:745-747, 830-868 feed the decoder / fusion calls
    train.py:747, 788-797, 837-868)."""

    @staticmethod
    def forward(ctx, src, G, plans):
        nat.require_device(src)
        src = _c(src)
        if src.dtype != torch.float32:
            raise RuntimeError("regroup: float32 only")
        G = int(G)
        plans = tuple(tuple(int(g) for g in p) for p in plans)
        if G <= 0 or src.shape[0] % G or not plans or any(not p for p in plans):
            raise RuntimeError(f"regroup: batch {src.shape[0]} is not {G} interleaved groups, or an empty plan")
        if any(g < 0 or g >= G for p in plans for g in p):
            raise RuntimeError(f"regroup: group index outside 0..{G - 1}")
        B = src.shape[0] // G
        chunk = src[0].numel()
        outs = [src.new_empty((B * len(p),) + tuple(src.shape[1:])) for p in plans]
        counts = (C.c_int32 * len(plans))(*[len(p) for p in plans])
        flat = [g for p in plans for g in p]
        groups = (C.c_int32 * len(flat))(*flat)
        dst, keep = nat.ptr_array(outs)
        nat.check(nat.lib().mvf_regroup_fwd(nat.ptr(src), G, B, chunk, len(plans), dst, counts, groups, _stream()),
                  "regroup_fwd")
        del keep
        ctx.G, ctx.plans, ctx.shape = G, plans, tuple(src.shape)
        ctx.set_materialize_grads(False)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gouts):
        G, plans, shape = ctx.G, ctx.plans, ctx.shape
        live = [g for g in gouts if g is not None]
        if not live:
            return None, None, None
        nat.require_device(*live)
        g_src = torch.empty(shape, dtype=torch.float32, device=live[0].device)
        B = shape[0] // G
        chunk = g_src[0].numel()

        def in_place(g):
            # a gradient whose samples are contiguous but further apart than their size (a channel slice of a
            # wider tensor: the fusion level's gradient of its centre features) is read where it is
            return g.shape[0] <= 1 or (g[0].is_contiguous() and g.stride(0) >= chunk)
        gouts = [None if g is None else (g if in_place(g) else _c(g)) for g in gouts]
        strides = (C.c_int64 * len(gouts))(*[chunk if (g is None or g.shape[0] <= 1) else g.stride(0) for g in gouts])
        counts = (C.c_int32 * len(plans))(*[len(p) for p in plans])
        flat = [g for p in plans for g in p]
        groups = (C.c_int32 * len(flat))(*flat)
        arr = (C.c_void_p * len(gouts))(*[None if g is None else g.data_ptr() for g in gouts])
        nat.check(nat.lib().mvf_regroup_bwd(C.cast(arr, C.c_void_p), strides, G, B, chunk, len(plans), counts, groups,
                                            nat.ptr(g_src), _stream()), "regroup_bwd")
        return g_src, None, None


def regroup(src, groups, plans):
    """-> tuple of interleaved batches, one per plan (see `Regroup`)."""
    return Regroup.apply(src, groups, plans)


def interleave_groups(parts):
    """``merge_groups([torch.cat(p, 1) for p in parts])`` in ONE launch: ``parts[g]`` = the tensors ``[B, C_i, ...]``
    whose channel concatenation is input g of a grouped call -> ``[B*G, sum C_i, ...]`` interleaved (sample
    b*G + g).  Forward only (the inputs of the grouped encoder / pose calls are images: reference
    train.py:724-731, 943-946)."""
    G = len(parts)
    flat = [t for p in parts for t in p]
    nat.require_device(*flat)
    if any(t.requires_grad for t in flat) and torch.is_grad_enabled():
        raise RuntimeError("interleave_groups is forward-only; inputs that need a gradient go through merge_groups")
    if any(t.dtype != torch.float32 for t in flat):
        raise RuntimeError("interleave_groups: float32 only")
    B, tail = flat[0].shape[0], tuple(flat[0].shape[2:])
    widths = [sum(t.shape[1] for t in p) for p in parts]
    if G == 0 or len(set(widths)) != 1 or any(t.shape[0] != B or tuple(t.shape[2:]) != tail for t in flat):
        raise RuntimeError("interleave_groups: every group must concatenate to the same [B, C, ...] shape")
    if len(flat) > 32:
        raise RuntimeError("interleave_groups: at most 32 parts")
    flat = [_c(t) for t in flat]
    inner = 1
    for d in tail:
        inner *= int(d)
    total = widths[0] * inner
    out = flat[0].new_empty((B * G, widths[0]) + tail)
    lens, offs, grp = [], [], []
    for g, p in enumerate(parts):
        at = 0
        for t in p:
            lens.append(t.shape[1] * inner)
            offs.append(at)
            grp.append(g)
            at += t.shape[1] * inner
    n = len(flat)
    src, keep = nat.ptr_array(flat)
    nat.check(nat.lib().mvf_interleave_fwd(src, (C.c_int64 * n)(*lens), (C.c_int64 * n)(*offs), (C.c_int32 * n)(*grp), n,
                                           nat.ptr(out), G, B, total, _stream()), "interleave_fwd")
    del keep
    return out


class MaxPool3s2(torch.autograd.Function):
    """nn.MaxPool2d(3, 2, 1) of the ResNet trunks (reference networks/monodepth2.py:39,
    networks/posenet.py:87): one byte of window-local argmax per output instead of ATen's int64
    index, backward as a deterministic gather (`mvf_maxpool3s2_fwd/bwd`)."""

    @staticmethod
    def forward(ctx, x):
        nat.require_device(x)
        x = _c(x)
        N, C, H, W = x.shape
        OH, OW = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        out = torch.empty((N, C, OH, OW), dtype=torch.float32, device=x.device)
        idx = torch.empty((N, C, OH, OW), dtype=torch.uint8, device=x.device)
        nat.check(nat.lib().mvf_maxpool3s2_fwd(nat.ptr(x), nat.ptr(out), nat.ptr(idx), N * C, H, W, _stream()),
                  "maxpool3s2_fwd")
        ctx.save_for_backward(idx)
        ctx.geom = (N, C, H, W)
        return out

    @staticmethod
    def backward(ctx, g):
        idx, = ctx.saved_tensors
        N, C, H, W = ctx.geom
        g = _c(g)
        gx = torch.empty((N, C, H, W), dtype=torch.float32, device=g.device)
        nat.check(nat.lib().mvf_maxpool3s2_bwd(nat.ptr(g), nat.ptr(idx), nat.ptr(gx), N * C, H, W, _stream()),
                  "maxpool3s2_bwd")
        return gx


def maxpool3s2(x):
    if x.dim() != 4 or x.dtype != torch.float32:
        raise RuntimeError("maxpool3s2 expects a float32 [N,C,H,W] tensor")
    return MaxPool3s2.apply(x)


class MaxPool3s2Tap(torch.autograd.Function):
    """(maxpool3s2(x), x): the pooled tensor and the input itself as a second output, for an input that is
    ALSO consumed elsewhere (the stem output of the depth encoder is pooled into layer1 and is level 0 of the
    feature pyramid: reference networks/monodepth2.py:36-41).  The backward adds the tap's gradient inside the
    pooling adjoint (`mvf_maxpool3s2_bwd_add`) instead of leaving two full-size tensors to autograd's
    accumulation pass; same two-term sum, same bits."""

    @staticmethod
    def forward(ctx, x):
        nat.require_device(x)
        x = _c(x)
        N, C, H, W = x.shape
        OH, OW = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        out = torch.empty((N, C, OH, OW), dtype=torch.float32, device=x.device)
        idx = torch.empty((N, C, OH, OW), dtype=torch.uint8, device=x.device)
        nat.check(nat.lib().mvf_maxpool3s2_fwd(nat.ptr(x), nat.ptr(out), nat.ptr(idx), N * C, H, W, _stream()),
                  "maxpool3s2_fwd")
        ctx.save_for_backward(idx)
        ctx.geom = (N, C, H, W)
        ctx.set_materialize_grads(False)
        return out, x.view_as(x)

    @staticmethod
    def backward(ctx, g, g_tap):
        if g is None:
            return g_tap
        idx, = ctx.saved_tensors
        N, C, H, W = ctx.geom
        g = _c(g)
        gx = torch.empty((N, C, H, W), dtype=torch.float32, device=g.device)
        if g_tap is None:
            nat.check(nat.lib().mvf_maxpool3s2_bwd(nat.ptr(g), nat.ptr(idx), nat.ptr(gx), N * C, H, W, _stream()),
                      "maxpool3s2_bwd")
        else:
            nat.check(nat.lib().mvf_maxpool3s2_bwd_add(nat.ptr(g), nat.ptr(idx), nat.ptr(_c(g_tap)), nat.ptr(gx), N * C, H, W,
                                                       _stream()), "maxpool3s2_bwd_add")
        return gx


def maxpool3s2_tap(x):
    """-> (pooled, x): use the returned x wherever else the input is consumed (see `MaxPool3s2Tap`)."""
    if x.dim() != 4 or x.dtype != torch.float32:
        raise RuntimeError("maxpool3s2 expects a float32 [N,C,H,W] tensor")
    return MaxPool3s2Tap.apply(x)


_ACT_CODES = {"none": 0, "elu": 1, "relu": 2, "prelu": 3}


class BiasAct(torch.autograd.Function):
    """out = act(x + bias[c] (+ res)) over [N,C,...] in one pass -- the epilogue of a biased
    convolution run without its bias (decoder ConvBlock: reference layers.py:106-118; IFRNet
    convrelu / ResBlock: networks/IFRNet.py:128-157).  Backward (act none / elu / relu): one pass
    gives g_x = g * act'(out) and the bias gradient (deterministic partial sums)."""

    @staticmethod
    def forward(ctx, x, bias, slope, res, act, inplace):
        nat.require_device(x, *[t for t in (bias, slope, res) if t is not None])
        if x.dim() < 2:
            raise RuntimeError("bias_act expects [N, C, ...]")
        xc = _c(x)
        N, C = xc.shape[0], xc.shape[1]
        HW = xc[0, 0].numel() if xc.dim() > 2 else 1
        if bias is not None and bias.numel() != C:
            raise RuntimeError(f"bias_act: bias has {bias.numel()} entries for {C} channels")
        if res is not None and res.shape != x.shape:
            raise RuntimeError("bias_act: residual shape differs from x")
        sl_n = 0
        if act == 3:
            if slope is None or slope.numel() not in (1, C):
                raise RuntimeError("bias_act: PReLU needs 1 or C slopes")
            sl_n = slope.numel()
        if inplace and xc is x:
            out = x
            ctx.mark_dirty(x)
        else:
            out = torch.empty_like(xc)
        nat.check(nat.lib().mvf_bias_act_fwd(nat.ptr(xc), nat.ptr(_c(bias) if bias is not None else None),
                                             nat.ptr(_c(slope) if slope is not None else None),
                                             nat.ptr(_c(res) if res is not None else None), nat.ptr(out), N, C, HW,
                                             act, sl_n, _stream()), "bias_act_fwd")
        ctx.act, ctx.has_bias, ctx.has_res = act, bias is not None, res is not None
        ctx.save_for_backward(out if act in (1, 2) else None)
        return out

    @staticmethod
    def backward(ctx, g):
        if ctx.act == 3:
            raise NotImplementedError("bias_act: PReLU epilogue is forward-only (frozen teacher); "
                                      "layers.conv_bias_act keeps the stock ops when a gradient is needed")
        (out,) = ctx.saved_tensors
        g = _c(g)
        N, C = g.shape[0], g.shape[1]
        HW = g[0, 0].numel() if g.dim() > 2 else 1
        gx = torch.empty_like(g) if ctx.act != 0 else g
        gb = ws = None
        if ctx.has_bias and ctx.needs_input_grad[1]:
            gb = torch.empty(C, dtype=torch.float32, device=g.device)
            ws = torch.empty(nat.lib().mvf_bias_act_workspace_floats(N, C, HW), dtype=torch.float32, device=g.device)
        if gb is not None or ctx.act != 0:
            nat.check(nat.lib().mvf_bias_act_bwd(nat.ptr(g), nat.ptr(out), nat.ptr(gx if ctx.act != 0 else None),
                                                 nat.ptr(gb), nat.ptr(ws), N, C, HW, ctx.act, _stream()),
                      "bias_act_bwd")
        return gx, gb, None, (gx if ctx.has_res else None), None, None


def bias_act(x, bias=None, act="none", slope=None, res=None, inplace=False):
    """act(x + bias[c] (+ res)); act in {"none", "elu", "relu", "prelu"}.  `inplace` overwrites x
    (safe for the fresh output of a convolution: its backward does not read its own result)."""
    return BiasAct.apply(x, bias, slope, res, _ACT_CODES[act], bool(inplace))


class SumAct(torch.autograd.Function):
    """act(((t0 + t1) + t2) + ...) in one pass (the branch sum + ReLU of an HRNet fuse layer: reference
    networks/hrnet_encoder.py:267-285 `y = y + ...; self.relu(y)`).  The gradient g * act'(out) is the same tensor for
    every term."""

    @staticmethod
    def forward(ctx, act, *terms):
        nat.require_device(*terms)
        ts = [_c(t) for t in terms]
        if any(t.shape != ts[0].shape or t.dtype != torch.float32 for t in ts):
            raise RuntimeError("sum_act: terms must be float32 tensors of one shape")
        if not 1 <= len(ts) <= 8:
            raise RuntimeError("sum_act: 1..8 terms")
        out = torch.empty_like(ts[0])
        arr, keep = nat.ptr_array(ts)
        nat.check(nat.lib().mvf_sum_act_fwd(arr, len(ts), nat.ptr(out), out.numel(), act, _stream()), "sum_act_fwd")
        del keep
        ctx.act, ctx.n = act, len(ts)
        ctx.save_for_backward(out if act == 2 else None)
        return out

    @staticmethod
    def backward(ctx, g):
        g = _c(g)
        if ctx.act == 2:
            (out,) = ctx.saved_tensors
            gx = torch.empty_like(g)
            N, C = (g.shape[0], g.shape[1]) if g.dim() >= 2 else (1, g.numel())
            HW = g.numel() // (N * C)
            nat.check(nat.lib().mvf_bias_act_bwd(nat.ptr(g), nat.ptr(out), nat.ptr(gx), None, None, N, C, HW, 2, _stream()),
                      "sum_act_bwd")
        else:
            gx = g
        return (None,) + (gx,) * ctx.n


def sum_act(terms, act="relu"):
    """act(sum of terms, left to right); act in {"none", "relu"}."""
    return SumAct.apply(_ACT_CODES[act], *terms)


def color_jitter(img, factors, order, apply, flip, frames=1, want_raw=False):
    """Flip + torchvision-style ColorJitter for a batch on the device (reference: the per-item
    host work of datasets/mono_dataset.py:214-256).  img [samples*frames,3,H,W] with the frames of
    a sample adjacent; factors [samples,4], order [samples,4] int32, apply / flip [samples] int32.
    -> (flipped frames | None, flipped + jittered frames).  No gradient (input data)."""
    nat.require_device(img, factors, order, apply, flip)
    img = _c(img.detach())
    n, ch, H, W = img.shape
    if ch != 3 or n % frames:
        raise RuntimeError("color_jitter expects [samples*frames,3,H,W]")
    S = n // frames
    factors = factors.to(torch.float32).contiguous()
    order, apply, flip = (t.to(torch.int32).contiguous() for t in (order, apply, flip))
    if tuple(factors.shape) != (S, 4) or tuple(order.shape) != (S, 4) or apply.numel() != S or flip.numel() != S:
        raise RuntimeError("color_jitter: factors / order must be [samples,4], apply / flip [samples]")
    raw = torch.empty_like(img) if want_raw else None
    aug = torch.empty_like(img)
    ws = torch.empty(nat.lib().mvf_color_jitter_workspace_floats(n), dtype=torch.float32, device=img.device)
    nat.check(nat.lib().mvf_color_jitter(nat.ptr(img), nat.ptr(factors), nat.ptr(order), nat.ptr(apply),
                                         nat.ptr(flip), nat.ptr(raw), nat.ptr(aug), nat.ptr(ws), S, frames, H, W,
                                         _stream()), "color_jitter")
    return raw, aug

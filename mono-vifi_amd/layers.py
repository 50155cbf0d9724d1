"""Drop-in mirror of the reference's ``layers.py`` API, backed by the gfx950 kernels.

Same names, argument order, shapes, dtypes and autograd behaviour as the reference
module (reference: layers.py:16-311), so ``from layers import *`` call sites keep working:

    disp_to_depth, transformation_from_parameters, get_translation_matrix,
    rot_from_axisangle, BackprojectDepth, Project3D, SSIM, get_smooth_loss, upsample,
    ConvBlock / Conv3x3 / Conv1x1 / ConvBlock1x1, compute_depth_errors

plus ``grid_sample_border_ac`` -- the exact ``F.grid_sample(..., padding_mode="border",
align_corners=True)`` call of train.py:966-969 as a standalone op.

Hot-path functions execute as hand-written HIP kernels through the C ABI
(include/mvf_hotpath.h); they require HIP tensors and have no CPU fallback.  The conv
building blocks are plain ``torch.nn`` modules (MIOpen / hipBLASLt do the contractions).
"""
from __future__ import annotations

import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops


# ----------------------------------------------------------------------------- geometry
def disp_to_depth(disp, min_depth, max_depth):
    """Sigmoid disparity -> (scaled_disp, depth); reference: layers.py:16-25."""
    return ops.DispToDepth.apply(disp, min_depth, max_depth)


def transformation_from_parameters(axisangle, translation, invert=False):
    """(axisangle [B,1,3], translation [B,1,3]) -> 4x4 pose [B,4,4]; reference:
    layers.py:28-45.  One fused kernel instead of ~60 tiny element-wise launches."""
    return ops.Pose.apply(axisangle, translation, bool(invert))


def get_translation_matrix(translation_vector):
    """reference: layers.py:48-61 (tiny host-side glue; kept in torch)."""
    T = torch.zeros(translation_vector.shape[0], 4, 4, device=translation_vector.device)
    t = translation_vector.contiguous().view(-1, 3, 1)
    T[:, 0, 0] = 1
    T[:, 1, 1] = 1
    T[:, 2, 2] = 1
    T[:, 3, 3] = 1
    T[:, :3, 3, None] = t
    return T


def rot_from_axisangle(vec):
    """Axis-angle [B,1,3] -> rotation 4x4 [B,4,4]; reference: layers.py:64-103."""
    zero = torch.zeros_like(vec)
    return ops.Pose.apply(vec, zero, False)


class BackprojectDepth(nn.Module):
    """Depth image -> homogeneous point cloud [B,4,H*W]; reference: layers.py:168-197.

    The reference bakes ``batch_size`` copies of the pixel grid into frozen parameters;
    the kernel regenerates (x, y, 1) from the lane index, so the module holds no state
    beyond the three sizes (``id_coords`` / ``ones`` / ``pix_coords`` are still exposed
    as buffers for code that reads them)."""

    def __init__(self, batch_size, height, width):
        super().__init__()
        self.batch_size = batch_size
        self.height = height
        self.width = width
        ys, xs = torch.meshgrid(torch.arange(height, dtype=torch.float32),
                                torch.arange(width, dtype=torch.float32), indexing="ij")
        self.register_buffer("id_coords", torch.stack([xs, ys], 0), persistent=False)

    @property
    def ones(self):
        return torch.ones(self.batch_size, 1, self.height * self.width,
                          device=self.id_coords.device)

    @property
    def pix_coords(self):
        flat = torch.stack([self.id_coords[0].reshape(-1), self.id_coords[1].reshape(-1)], 0)
        flat = flat.unsqueeze(0).repeat(self.batch_size, 1, 1)
        return torch.cat([flat, self.ones], 1)

    def forward(self, depth, inv_K):
        if depth.numel() != self.batch_size * self.height * self.width:
            # the reference fails in depth.view(self.batch_size, 1, -1) (layers.py:194)
            raise RuntimeError(
                f"shape '[{self.batch_size}, 1, -1]' is invalid for input of size {depth.numel()}"
                if depth.numel() % max(self.batch_size, 1) else
                f"depth has {depth.numel()} elements, expected "
                f"{self.batch_size}x1x{self.height}x{self.width}")
        return ops.Backproject.apply(depth, inv_K, self.batch_size, self.height, self.width)


class Project3D(nn.Module):
    """3-D points -> normalised sampling grid [B,H,W,2]; reference: layers.py:200-222."""

    def __init__(self, batch_size, height, width, eps=1e-7):
        super().__init__()
        self.batch_size = batch_size
        self.height = height
        self.width = width
        self.eps = eps

    def forward(self, points, K, T):
        if points.numel() != self.batch_size * 4 * self.height * self.width:
            raise RuntimeError(
                f"points has {points.numel()} elements, expected "
                f"{self.batch_size}x4x{self.height * self.width}")
        return ops.Project.apply(points, K, T, self.batch_size, self.height, self.width,
                                 float(self.eps))


def grid_sample_border_ac(img, grid):
    """``F.grid_sample(img, grid, padding_mode="border", align_corners=True)`` (bilinear);
    reference call site: train.py:966-969."""
    return ops.GridSampleBorderAC.apply(img, grid)


# ----------------------------------------------------------------------------- photometric
class SSIM(nn.Module):
    """SSIM loss map between two images, [B,C,H,W] in [0,1]; reference: layers.py:261-290."""

    def __init__(self):
        super().__init__()
        self.C1 = 0.01 ** 2
        self.C2 = 0.03 ** 2

    def forward(self, x, y):
        return ops.SSIMFn.apply(x, y)


def get_smooth_loss(disp, img):
    """Edge-aware disparity smoothness (scalar); reference: layers.py:231-242."""
    return ops.Smooth.apply(disp, img, False)


def get_smooth_loss_dyn(disp, img, mask_dyn):
    """reference: layers.py:244-258 -- the smoothness term with dynamic-object regions blanked in the image
    and their vertical disparity gradients weighted 100 x.  Part of the reference's `layers` namespace but
    called nowhere in it (not on the hot path): a few tensor expressions on the caller's device, kept so that
    `from layers import *` resolves every name the reference's does."""
    heavy = 100.0 * mask_dyn + 1.0 - mask_dyn
    still = (1.0 - mask_dyn) * img
    d_dx = (disp[..., :, :-1] - disp[..., :, 1:]).abs()
    d_dy = (disp[..., :-1, :] - disp[..., 1:, :]).abs()
    i_dx = (still[..., :, :-1] - still[..., :, 1:]).abs().mean(1, keepdim=True)
    i_dy = (still[..., :-1, :] - still[..., 1:, :]).abs().mean(1, keepdim=True)
    sx = d_dx * torch.exp(-i_dx)
    sy = d_dy * torch.exp(-i_dy) * heavy[:, :, :-1, :]
    return sx.mean() + sy.mean()


# ----------------------------------------------------------------------------- conv blocks
def upsample(x, scale_factor=2, mode="nearest"):
    """reference: layers.py:225-228.  On the HIP device (fp32) bilinear (Lite-Mono decoder) and
    integer-factor nearest (DHRNet decoder) run as element-parallel kernels with gather adjoints:
    ATen's NCHW kernels take one thread per output POSITION and loop over batch x channels."""
    if (FUSED_EPILOGUE and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.is_contiguous()
            and not torch.is_autocast_enabled()):
        if mode == "bilinear":
            return ops.resize_bilinear(x, scale_factor=scale_factor, align_corners=False)
        if mode == "nearest" and int(scale_factor) == scale_factor and scale_factor >= 1:
            return ops.upsample_nearest(x, int(scale_factor))
    return F.interpolate(x, scale_factor=scale_factor, mode=mode)


class Conv3x3(nn.Module):
    """pad (reflect or zero) + 3x3 conv; reference: layers.py:121-138"""

    def __init__(self, in_channels, out_channels, use_refl=True):
        super().__init__()
        self.use_refl = bool(use_refl)
        self.pad = nn.ReflectionPad2d(1) if use_refl else nn.ZeroPad2d(1)
        self.conv = nn.Conv2d(int(in_channels), int(out_channels), 3)

    def forward(self, x, act="none"):
        """``act`` = "elu": the ConvBlock's activation rides in the convolution's epilogue."""
        if self.use_refl and x.is_cuda and x.dtype == torch.float32:
            # fp32 on the HIP device: gather kernels (deterministic backward, no atomics);
            # other dtypes / hosts keep the stock module (the networks are plain PyTorch)
            return conv_bias_act(self.conv, ops.reflect_pad1(x), act)
        return conv_bias_act(self.conv, self.pad(x), act)


# fp32 on the HIP device: a biased convolution runs without its bias and ONE epilogue pass does
# bias + activation (+ residual) (ops.bias_act); backward one pass does the activation's adjoint
# and the bias gradient.  False = the stock op-by-op form (also what CPU tensors / autocast take).
FUSED_EPILOGUE = os.environ.get("MVF_FUSED_EPILOGUE", "1") != "0"      # developer knob for A/B timing


def conv_bias_act(conv, x, act="none", act_module=None, res=None):
    """``act(conv(x) + res)`` for an ``nn.Conv2d`` / ``nn.ConvTranspose2d`` ``conv``; ``act`` in
    {"none", "elu", "relu", "prelu"} (``act_module``: the ``nn.PReLU`` holding the slopes)."""
    slope = act_module.weight if act == "prelu" else None
    need_grad = torch.is_grad_enabled() and (x.requires_grad or conv.weight.requires_grad or
                                             (slope is not None and slope.requires_grad))
    # (NCHW-contiguous inputs only: under --channels_last the convolution's output is channels-last
    # and the epilogue kernel would first make an NCHW copy of it -- the stock ops are faster there)
    fused = (FUSED_EPILOGUE and x.is_cuda and x.dtype == torch.float32 and conv.bias is not None and
             x.is_contiguous() and
             conv.weight.dtype == torch.float32 and not torch.is_autocast_enabled() and
             getattr(conv, "padding_mode", "zeros") == "zeros" and not (act == "prelu" and need_grad))
    if fused:
        if isinstance(conv, nn.ConvTranspose2d):
            y = F.conv_transpose2d(x, conv.weight, None, conv.stride, conv.padding, conv.output_padding,
                                   conv.groups, conv.dilation)
        else:
            y = F.conv2d(x, conv.weight, None, conv.stride, conv.padding, conv.dilation, conv.groups)
        return ops.bias_act(y, conv.bias, act, slope, res, inplace=True)
    y = conv(x)
    if res is not None:
        y = y + res
    if act == "none":
        return y
    if act_module is not None:
        return act_module(y)
    return {"elu": F.elu, "relu": F.relu}[act](y)


class ConvBlock(nn.Module):
    """3x3 conv + ELU; reference: layers.py:106-118"""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.conv = Conv3x3(in_channels, out_channels)
        self.nonlin = nn.ELU()

    def forward(self, x):
        return self.conv(x, act="elu")


class Conv1x1(nn.Module):
    """reference: layers.py:141-150"""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.conv = nn.Conv2d(int(in_channels), int(out_channels), kernel_size=1, stride=1)

    def forward(self, x):
        return conv_bias_act(self.conv, x, "none")


class ConvBlock1x1(nn.Module):
    """1x1 conv + ELU; reference: layers.py:153-165"""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.conv = Conv1x1(in_channels, out_channels)
        self.nonlin = nn.ELU()

    def forward(self, x):
        return conv_bias_act(self.conv.conv, x, "elu")


def compute_depth_errors(gt, pred):
    """abs_rel, sq_rel, rmse, rmse_log, a1, a2, a3; reference: layers.py:293-311
    (evaluation-only, outside the hot path: plain tensor ops)."""
    thresh = torch.max(gt / pred, pred / gt)
    a1 = (thresh < 1.25).float().mean()
    a2 = (thresh < 1.25 ** 2).float().mean()
    a3 = (thresh < 1.25 ** 3).float().mean()
    rmse = torch.sqrt(((gt - pred) ** 2).mean())
    rmse_log = torch.sqrt(((torch.log(gt) - torch.log(pred)) ** 2).mean())
    abs_rel = torch.mean(torch.abs(gt - pred) / gt)
    sq_rel = torch.mean((gt - pred) ** 2 / gt)
    return abs_rel, sq_rel, rmse, rmse_log, a1, a2, a3

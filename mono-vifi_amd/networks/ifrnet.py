"""IFRNet video-frame-interpolation teacher (frozen, inference only in this trainer).

Architecture, call signature and state-dict keys follow reference networks/IFRNet.py
(``warp`` 7-15, encoders/decoders 160-349, ``IFRNet.forward`` 373-441) so the released
``IFRNet_{L,S}_*.pth`` teacher weights load.  The VFI *training* losses of the reference
(census / geometry / Charbonnier, IFRNet.py:18-126) belong to ``train_vfi.py`` and are out
of scope: passing ``imgt`` raises.
"""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from ..layers import conv_bias_act
from ..consts import const_tensor


def _hip_flow_warp(img, flow):
    from .. import ops
    return ops.flow_warp(img, flow).to(img.dtype)


# The warp implementation is a module attribute so that the CPU-only comparison against the
# reference's modules (tests/test_networks_vs_reference.py) can inject a torch restatement;
# the product default is the gfx950 kernel (no CPU fallback).
WARP_IMPL = _hip_flow_warp
RESIZE_KERNEL = os.environ.get("MVF_IFRNET_RESIZE", "1") != "0"      # developer knob for A/B timing


def warp(img, flow):
    """Backward-warp ``img`` by a pixel-unit flow field with the same bilinear / border /
    align_corners=True gather the depth path uses (reference: IFRNet.py:7-15); runs as the
    ``mvf_flow_warp`` kernels."""
    return WARP_IMPL(img, flow)


def _interp(x, size=None, scale_factor=None):
    """F.interpolate(..., mode="bilinear", align_corners=False); on the HIP device (fp32) the element-parallel
    kernel of ops.resize_bilinear -- ATen's NCHW kernel is position-parallel and loops over batch x channels inside a
    lane (40 us for a 35 MB flow field, eleven of them per teacher pass)."""
    if RESIZE_KERNEL and x.is_cuda and x.dtype == torch.float32 and not torch.is_autocast_enabled():
        from .. import ops
        return ops.resize_bilinear(x, size=size, scale_factor=scale_factor, align_corners=False)
    if size is not None:
        return F.interpolate(x, size=size, mode="bilinear", align_corners=False)
    return F.interpolate(x, scale_factor=scale_factor, mode="bilinear", align_corners=False)


def resize(x, scale_factor):
    return _interp(x, scale_factor=scale_factor)


class ConvPReLU(nn.Sequential):
    """conv + PReLU with the reference's ``nn.Sequential`` state-dict keys (``0.*``, ``1.weight``);
    bias and PReLU ride in one epilogue pass when no gradient is needed (frozen teacher)."""

    def forward(self, x):
        return conv_bias_act(self[0], x, "prelu", self[1])


def convrelu(cin, cout, k=3, stride=1, pad=1):
    return ConvPReLU(nn.Conv2d(cin, cout, k, stride, pad), nn.PReLU(cout))


class ResBlock(nn.Module):
    """Five 3x3 convs; convs 2 and 4 act on the last ``side`` channels only."""

    def __init__(self, ch, side):
        super().__init__()
        self.side_channels = side
        self.conv1 = convrelu(ch, ch)
        self.conv2 = convrelu(side, side)
        self.conv3 = convrelu(ch, ch)
        self.conv4 = convrelu(side, side)
        self.conv5 = nn.Conv2d(ch, ch, 3, 1, 1)
        self.prelu = nn.PReLU(ch)

    def _side(self, out, conv):
        """``cat([out[:, :-s], conv(out[:, -s:])], 1)`` (reference: networks/IFRNet.py ResBlock.forward).  With
        autograd off (the frozen teacher of a training step) ``out`` -- the fresh output of the previous
        convolution -- takes the new side channels in place: s of its C channels move instead of all of them
        twice (IFRNet-L at 36 images: 5.2 GB of cat traffic per step -> 1.2 GB)."""
        s = self.side_channels
        if not torch.is_grad_enabled() and not out.requires_grad:
            out[:, -s:] = conv(out[:, -s:])
            return out
        return torch.cat([out[:, :-s], conv(out[:, -s:])], 1)

    def forward(self, x):
        out = self._side(self.conv1(x), self.conv2)
        out = self._side(self.conv3(out), self.conv4)
        return conv_bias_act(self.conv5, out, "prelu", self.prelu, res=x)


class Encoder(nn.Module):
    def __init__(self, chans, first_kernel):
        super().__init__()
        c1, c2, c3, c4 = chans
        k, p = first_kernel, first_kernel // 2
        self.pyramid1 = nn.Sequential(convrelu(3, c1, k, 2, p), convrelu(c1, c1))
        self.pyramid2 = nn.Sequential(convrelu(c1, c2, 3, 2, 1), convrelu(c2, c2))
        self.pyramid3 = nn.Sequential(convrelu(c2, c3, 3, 2, 1), convrelu(c3, c3))
        self.pyramid4 = nn.Sequential(convrelu(c3, c4, 3, 2, 1), convrelu(c4, c4))

    def forward(self, img):
        f1 = self.pyramid1(img)
        f2 = self.pyramid2(f1)
        f3 = self.pyramid3(f2)
        f4 = self.pyramid4(f3)
        return f1, f2, f3, f4


class Decoder(nn.Module):
    """convrelu -> ResBlock -> 4x4 stride-2 transposed conv."""

    def __init__(self, cin, mid, side, cout, top=False):
        super().__init__()
        self.top = top
        self.convblock = nn.Sequential(convrelu(cin, mid), ResBlock(mid, side),
                                       nn.ConvTranspose2d(mid, cout, 4, 2, 1, bias=True))

    def _run(self, x):
        x = self.convblock[1](self.convblock[0](x))
        return conv_bias_act(self.convblock[2], x, "none")

    def forward(self, *args):
        if self.top:
            f0, f1, embt = args
            _, _, h, w = f0.shape
            return self._run(torch.cat([f0, f1, embt.repeat(1, 1, h, w)], 1))
        ft_, f0, f1, up0, up1 = args
        return self._run(torch.cat([ft_, warp(f0, up0), warp(f1, up1), up0, up1], 1))


_SCALES = {
    # encoder channels, first kernel, side channels, decoder (cin, mid, cout) top..bottom
    "large": ((64, 96, 144, 192), 7, 64, [(385, 384, 148), (436, 432, 100), (292, 288, 68), (196, 192, 8)]),
    "small": ((24, 36, 54, 72), 3, 24, [(145, 144, 58), (166, 162, 40), (112, 108, 28), (76, 72, 8)]),
}


class IFRNet(nn.Module):
    def __init__(self, scale="large"):
        super().__init__()
        chans, k, side, dec = _SCALES[scale]
        self.encoder = Encoder(chans, k)
        self.decoder4 = Decoder(dec[0][0], dec[0][1], side, dec[0][2], top=True)
        self.decoder3 = Decoder(dec[1][0], dec[1][1], side, dec[1][2])
        self.decoder2 = Decoder(dec[2][0], dec[2][1], side, dec[2][2])
        self.decoder1 = Decoder(dec[3][0], dec[3][1], side, dec[3][2])

    def forward(self, img0, img1, embt, imgt=None, scale_factor=(1.0, 0.5), onlyFlow=False):
        if imgt is not None:
            raise NotImplementedError("VFI training losses are out of scope (train_vfi.py)")
        _, _, H, W = img0.shape
        if H == 320 and W == 1024:
            scale_factor = (0.6, 0.3125)
        mean_ = torch.cat([img0, img1], 2).mean(1, keepdim=True).mean(2, keepdim=True).mean(3, keepdim=True)
        img0 = img0 - mean_
        img1 = img1 - mean_
        fh, fw = int(H * scale_factor[0]), int(W * scale_factor[1])
        f0 = self.encoder(_interp(img0, size=(fh, fw)))
        f1 = self.encoder(_interp(img1, size=(fh, fw)))

        out = self.decoder4(f0[3], f1[3], embt)
        up0, up1, ft = out[:, 0:2], out[:, 2:4], out[:, 4:]
        for dec, a, b in ((self.decoder3, f0[2], f1[2]), (self.decoder2, f0[1], f1[1]),
                          (self.decoder1, f0[0], f1[0])):
            out = dec(ft, a, b, up0, up1)
            up0 = out[:, 0:2] + 2.0 * resize(up0, 2.0)
            up1 = out[:, 2:4] + 2.0 * resize(up1, 2.0)
            ft = out[:, 4:]
        mask = torch.sigmoid(out[:, 4:5])

        sx, sy = 1.0 / scale_factor[1], 1.0 / scale_factor[0]
        scale = const_tensor((sx, sy), up0.device, up0.dtype).view(1, 2, 1, 1)
        up0 = _interp(up0, size=(H, W)) * scale
        up1 = _interp(up1, size=(H, W)) * scale
        mask = _interp(mask, size=(H, W))
        if onlyFlow:
            return up0, up1, mask
        merged = mask * warp(img0, up0) + (1 - mask) * warp(img1, up1)
        return torch.clamp(merged + mean_, 0, 1), up0, up1, mask

"""Lite-Mono depth network: CNN + cross-covariance-attention encoder, 3-scale decoder.

API / state-dict keys follow reference networks/LiteMono.py (``DepthEncoder`` 296-444,
``DepthDecoder`` 447-504): a 3-conv stem at stride 2, three stages at strides 4/8/16 made of
depth-wise dilated residual blocks ("CDC") closed by one local-global block (LGFI: channel
attention over the token axis + inverted bottleneck), average-pooled copies of the input
image concatenated before each down-sampling conv, and a decoder with bilinear upsampling.
timm is absent on both boxes: stochastic depth and truncated-normal init are implemented
here.  Variants: lite-mono, -small, -tiny, -8m (same tables as the reference)."""
import math

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from ..layers import Conv3x3, ConvBlock, upsample


class _TokenLinear(torch.autograd.Function):
    """``F.linear`` over a token tensor [B, ..., C_in] whose WEIGHT gradient is formed per image
    and then summed.  The stock gradient dW = dY^T X is one GEMM with a tiny output (C_out x C_in,
    e.g. 288 x 48) and a reduction length of B*H*W tokens (1.3 M at 1024x320): hipBLASLt runs it
    on 18 workgroups, 1.5 ms per layer and 25 ms per Lite-Mono step.  As a batched GEMM over the
    B images it fills the device; the B partial products are folded with one small sum."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return F.linear(x, weight, bias)

    @staticmethod
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        gx = gw = gb = None
        if g.dtype != weight.dtype:          # mixed precision around the call: compute in the weight's type
            g = g.to(weight.dtype)
        if x.dtype != weight.dtype:
            x = x.to(weight.dtype)
        if ctx.needs_input_grad[0]:
            gx = g @ weight
        B = x.shape[0]
        g3, x3 = g.reshape(B, -1, g.shape[-1]), x.reshape(B, -1, x.shape[-1])
        if ctx.needs_input_grad[1]:
            gw = torch.bmm(g3.transpose(1, 2), x3).sum(0)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            gb = g3.sum((0, 1))
        return gx, gw, gb


SPLIT_TOKEN_GRAD = True      # False: stock nn.Linear backward (tests compare both)


def token_linear(lin, x):
    """``lin(x)`` for an ``nn.Linear`` over [B, tokens..., C]; on the device with many tokens per
    image the weight gradient takes the per-image batched form."""
    # (not under autocast: backward() runs outside the autocast context, where a bf16 gradient
    # would meet the fp32 weight / activation in `g @ weight` and the batched GEMM)
    if (SPLIT_TOKEN_GRAD and x.is_cuda and x.dim() >= 3 and x.shape[0] > 1 and torch.is_grad_enabled()
            and not torch.is_autocast_enabled() and x.dtype == lin.weight.dtype
            and lin.weight.requires_grad and x[0].numel() // x.shape[-1] >= 1024):
        return _TokenLinear.apply(x, lin.weight, lin.bias)
    return lin(x)


class DropPath(nn.Module):
    """Stochastic depth: drops the residual branch of whole samples with probability p."""

    def __init__(self, p=0.0):
        super().__init__()
        self.p = float(p)

    def forward(self, x):
        if self.p == 0.0 or not self.training:
            return x
        keep = 1.0 - self.p
        mask = x.new_empty((x.shape[0],) + (1,) * (x.dim() - 1)).bernoulli_(keep)
        return x * (mask / keep)        # one pass over x (the scale is per sample)


class PositionalEncodingFourier(nn.Module):
    """Sine/cosine position code of the (y, x) grid projected to ``dim`` channels by a 1x1 conv."""

    def __init__(self, hidden_dim=32, dim=768, temperature=10000):
        super().__init__()
        self.token_projection = nn.Conv2d(hidden_dim * 2, dim, kernel_size=1)
        self.scale = 2 * math.pi
        self.temperature = temperature
        self.hidden_dim = hidden_dim
        self.dim = dim

    def forward(self, B, H, W):
        dev = self.token_projection.weight.device
        eps = 1e-6
        ys = torch.arange(1, H + 1, dtype=torch.float32, device=dev).view(1, H, 1).expand(B, H, W)
        xs = torch.arange(1, W + 1, dtype=torch.float32, device=dev).view(1, 1, W).expand(B, H, W)
        ys = ys / (H + eps) * self.scale
        xs = xs / (W + eps) * self.scale
        k = torch.arange(self.hidden_dim, dtype=torch.float32, device=dev)
        freq = self.temperature ** (2 * torch.div(k, 2, rounding_mode="floor") / self.hidden_dim)

        def code(v):
            a = v[..., None] / freq
            return torch.stack((a[..., 0::2].sin(), a[..., 1::2].cos()), dim=4).flatten(3)

        pos = torch.cat((code(ys), code(xs)), dim=3).permute(0, 3, 1, 2)
        return self.token_projection(pos)


class XCA(nn.Module):
    """Cross-covariance attention: softmax over the (channel x channel) Gram matrix of
    L2-normalised queries/keys, one temperature per head."""

    def __init__(self, dim, num_heads=8, qkv_bias=False, attn_drop=0.0, proj_drop=0.0):
        super().__init__()
        self.num_heads = num_heads
        self.temperature = nn.Parameter(torch.ones(num_heads, 1, 1))
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(proj_drop)

    def forward(self, x):
        B, N, C = x.shape
        qkv = token_linear(self.qkv, x).reshape(B, N, 3, self.num_heads, C // self.num_heads).permute(2, 0, 3, 4, 1)
        q, k, v = F.normalize(qkv[0], dim=-1), F.normalize(qkv[1], dim=-1), qkv[2]
        attn = self.attn_drop(((q @ k.transpose(-2, -1)) * self.temperature).softmax(dim=-1))
        x = (attn @ v).permute(0, 3, 1, 2).reshape(B, N, C)
        return self.proj_drop(token_linear(self.proj, x))


class LayerNorm(nn.Module):
    def __init__(self, normalized_shape, eps=1e-6, data_format="channels_last"):
        super().__init__()
        if data_format not in ("channels_last", "channels_first"):
            raise NotImplementedError
        self.weight = nn.Parameter(torch.ones(normalized_shape))
        self.bias = nn.Parameter(torch.zeros(normalized_shape))
        self.eps = eps
        self.data_format = data_format
        self.normalized_shape = (normalized_shape,)

    def forward(self, x):
        if self.data_format == "channels_last":
            return F.layer_norm(x, self.normalized_shape, self.weight, self.bias, self.eps)
        u = x.mean(1, keepdim=True)
        s = (x - u).pow(2).mean(1, keepdim=True)
        x = (x - u) / torch.sqrt(s + self.eps)
        return self.weight[:, None, None] * x + self.bias[:, None, None]


class BNGELU(nn.Module):
    def __init__(self, ch):
        super().__init__()
        self.bn = nn.BatchNorm2d(ch, eps=1e-5)
        self.act = nn.GELU()

    def forward(self, x):
        return self.act(self.bn(x))


class Conv(nn.Module):
    def __init__(self, cin, cout, kSize, stride, padding=0, bn_act=False):
        super().__init__()
        self.bn_act = bn_act
        self.conv = nn.Conv2d(cin, cout, kSize, stride, padding, bias=False)
        if bn_act:
            self.bn_gelu = BNGELU(cout)

    def forward(self, x):
        x = self.conv(x)
        return self.bn_gelu(x) if self.bn_act else x


class CDilated(nn.Module):
    def __init__(self, cin, cout, kSize, stride=1, d=1, groups=1):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, kSize, stride, ((kSize - 1) // 2) * d, dilation=d,
                              groups=groups, bias=False)

    def forward(self, x):
        return self.conv(x)


NCHW_MLP = True      # False: the reference's permute -> Linear -> permute form (tests compare both)


class _InvertedBottleneck(nn.Module):
    """Shared tail of both block types: (LayerNorm) -> Linear xE -> GELU -> Linear -> gamma."""

    def _mlp(self, x):
        x = token_linear(self.pwconv2, self.act(token_linear(self.pwconv1, x)))
        return x if self.gamma is None else self.gamma * x

    def _mlp_nchw(self, y):
        """The same channel MLP on a contiguous [B, C, H, W] tensor WITHOUT leaving the layout:
        out[b] = W2 gelu(W1 y[b] + b1) + b2 as batched GEMMs over the images ([E*C x C] @ [C x HW]).
        The reference permutes to channels-last, applies nn.Linear and permutes back
        (networks/LiteMono.py:160-176): per block a strided copy of the activation in, a strided
        multiply and a strided residual add out, and as many again in backward -- 28 ms per step of
        non-vectorised element-wise kernels at 1024x320.  As a by-product the weight gradients are
        batched GEMMs over the images (what `_TokenLinear` arranges for the token form)."""
        B, C, H, W = y.shape
        l1, l2 = self.pwconv1, self.pwconv2
        h = torch.bmm(l1.weight.unsqueeze(0).expand(B, -1, -1), y.view(B, C, H * W))
        if l1.bias is not None:
            h += l1.bias.view(1, -1, 1).to(h.dtype)          # (bf16 autocast: the GEMM output is bf16)
        o = torch.bmm(l2.weight.unsqueeze(0).expand(B, -1, -1), self.act(h))
        if l2.bias is not None:
            o += l2.bias.view(1, -1, 1).to(o.dtype)
        if self.gamma is not None:
            o = self.gamma.view(1, -1, 1) * o
        return o.view(B, -1, H, W)


class DilatedConv(_InvertedBottleneck):
    """Depth-wise dilated 3x3 conv + BN, then a channel MLP; residual with stochastic depth.
    (``norm`` is registered, like in the reference, but not applied in forward.)"""

    def __init__(self, dim, k, dilation=1, stride=1, drop_path=0.0, layer_scale_init_value=1e-6,
                 expan_ratio=6):
        super().__init__()
        self.ddwconv = CDilated(dim, dim, k, stride, dilation, groups=dim)
        self.bn1 = nn.BatchNorm2d(dim)
        self.norm = LayerNorm(dim, eps=1e-6)
        self.pwconv1 = nn.Linear(dim, expan_ratio * dim)
        self.act = nn.GELU()
        self.pwconv2 = nn.Linear(expan_ratio * dim, dim)
        self.gamma = nn.Parameter(layer_scale_init_value * torch.ones(dim)) \
            if layer_scale_init_value > 0 else None
        self.drop_path = DropPath(drop_path) if drop_path > 0.0 else nn.Identity()

    def forward(self, x):
        y = self.bn1(self.ddwconv(x))
        if NCHW_MLP and y.is_contiguous():
            y = self._mlp_nchw(y)
        else:
            y = self._mlp(y.permute(0, 2, 3, 1)).permute(0, 3, 1, 2)
        return x + self.drop_path(y)


class LGFI(_InvertedBottleneck):
    """Local-global feature interaction: XCA over the flattened tokens, then the MLP."""

    def __init__(self, dim, drop_path=0.0, layer_scale_init_value=1e-6, expan_ratio=6,
                 use_pos_emb=True, num_heads=6, qkv_bias=True, attn_drop=0.0, drop=0.0):
        super().__init__()
        self.dim = dim
        self.pos_embd = PositionalEncodingFourier(dim=dim) if use_pos_emb else None
        self.norm_xca = LayerNorm(dim, eps=1e-6)
        self.gamma_xca = nn.Parameter(layer_scale_init_value * torch.ones(dim)) \
            if layer_scale_init_value > 0 else None
        self.xca = XCA(dim, num_heads=num_heads, qkv_bias=qkv_bias, attn_drop=attn_drop, proj_drop=drop)
        self.norm = LayerNorm(dim, eps=1e-6)
        self.pwconv1 = nn.Linear(dim, expan_ratio * dim)
        self.act = nn.GELU()
        self.pwconv2 = nn.Linear(expan_ratio * dim, dim)
        self.gamma = nn.Parameter(layer_scale_init_value * torch.ones(dim)) \
            if layer_scale_init_value > 0 else None
        self.drop_path = DropPath(drop_path) if drop_path > 0.0 else nn.Identity()

    def forward(self, x):
        B, C, H, W = x.shape
        t = x.reshape(B, C, H * W).permute(0, 2, 1)
        if self.pos_embd is not None:
            t = t + self.pos_embd(B, H, W).reshape(B, -1, t.shape[1]).permute(0, 2, 1)
        t = t + self.gamma_xca * self.xca(self.norm_xca(t))
        y = self._mlp(self.norm(t.reshape(B, H, W, C))).permute(0, 3, 1, 2)
        return x + self.drop_path(y)


class AvgPool(nn.Module):
    def __init__(self, ratio):
        super().__init__()
        self.pool = nn.ModuleList([nn.AvgPool2d(3, stride=2, padding=1) for _ in range(ratio)])

    def forward(self, x):
        for p in self.pool:
            x = p(x)
        return x


# variant -> (channels, blocks per stage, dilations at 192-high inputs, dilations at 320x1024)
_VARIANTS = {
    "lite-mono": ([48, 80, 128], [4, 4, 10],
                  [[1, 2, 3], [1, 2, 3], [1, 2, 3, 1, 2, 3, 2, 4, 6]],
                  [[1, 2, 5], [1, 2, 5], [1, 2, 5, 1, 2, 5, 2, 4, 10]]),
    "lite-mono-small": ([48, 80, 128], [4, 4, 7],
                        [[1, 2, 3], [1, 2, 3], [1, 2, 3, 2, 4, 6]],
                        [[1, 2, 5], [1, 2, 5], [1, 2, 5, 2, 4, 10]]),
    "lite-mono-tiny": ([32, 64, 128], [4, 4, 7],
                       [[1, 2, 3], [1, 2, 3], [1, 2, 3, 2, 4, 6]],
                       [[1, 2, 5], [1, 2, 5], [1, 2, 5, 2, 4, 10]]),
    "lite-mono-8m": ([64, 128, 224], [4, 4, 10],
                     [[1, 2, 3], [1, 2, 3], [1, 2, 3, 1, 2, 3, 2, 4, 6]],
                     [[1, 2, 3], [1, 2, 3], [1, 2, 3, 1, 2, 3, 2, 4, 6]]),
}


class DepthEncoder(nn.Module):
    def __init__(self, in_chans=3, model="lite-mono", height=192, width=640, global_block=(1, 1, 1),
                 global_block_type=("LGFI", "LGFI", "LGFI"), drop_path_rate=0.2,
                 layer_scale_init_value=1e-6, expan_ratio=6, heads=(8, 8, 8),
                 use_pos_embd_xca=(True, False, False), **kwargs):
        super().__init__()
        dims, depth, dil_mr, dil_hr = _VARIANTS[model]
        self.num_ch_enc = np.array(dims)
        self.depth, self.dims = list(depth), list(dims)
        if height == 320 and width == 1024:
            self.dilation = dil_hr
        else:   # the reference defines dilations for 192x640 / 192x512 (LiteMono.py:312-315)
            self.dilation = dil_mr
        for g in global_block_type:
            assert g in ("None", "LGFI")

        d0 = dims[0]
        self.downsample_layers = nn.ModuleList([nn.Sequential(
            Conv(in_chans, d0, 3, 2, 1, bn_act=True), Conv(d0, d0, 3, 1, 1, bn_act=True),
            Conv(d0, d0, 3, 1, 1, bn_act=True))])
        self.stem2 = nn.Sequential(Conv(d0 + 3, d0, 3, 2, 1, bn_act=False))
        self.input_downsample = nn.ModuleList([AvgPool(i) for i in range(1, 5)])
        for i in range(2):
            self.downsample_layers.append(nn.Sequential(Conv(dims[i] * 2 + 3, dims[i + 1], 3, 2, 1)))

        rates = torch.linspace(0, drop_path_rate, sum(depth)).tolist()
        self.stages = nn.ModuleList()
        cur = 0
        for i in range(3):
            blocks = []
            for j in range(depth[i]):
                if j > depth[i] - global_block[i] - 1:
                    if global_block_type[i] != "LGFI":
                        raise NotImplementedError
                    blocks.append(LGFI(dims[i], rates[cur + j], layer_scale_init_value, expan_ratio,
                                       use_pos_embd_xca[i], heads[i]))
                else:
                    blocks.append(DilatedConv(dims[i], 3, self.dilation[i][j], 1, rates[cur + j],
                                              layer_scale_init_value, expan_ratio))
            self.stages.append(nn.Sequential(*blocks))
            cur += depth[i]
        self.apply(self._init_weights)

    @staticmethod
    def _init_weights(m):
        if isinstance(m, (nn.Conv2d, nn.Linear)):
            nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
        elif isinstance(m, (LayerNorm, nn.LayerNorm, nn.BatchNorm2d)):
            nn.init.constant_(m.weight, 1.0)
            nn.init.constant_(m.bias, 0)

    def forward(self, x):
        x = (x - 0.45) / 0.225
        pooled = [p(x) for p in self.input_downsample]
        x = self.stem2(torch.cat((self.downsample_layers[0](x), pooled[0]), dim=1))
        carry = [x]
        features = []
        for i in range(3):
            if i > 0:
                carry.append(pooled[i])
                x = self.downsample_layers[i](torch.cat(carry, dim=1))
                carry = [x]
            x = self.stages[i](x)
            carry.append(x)
            features.append(x)
        return features


class DepthDecoder(nn.Module):
    def __init__(self, num_ch_enc, scales=range(4), num_output_channels=1, use_skips=True):
        super().__init__()
        self.num_output_channels = num_output_channels
        self.use_skips = use_skips
        self.upsample_mode = "bilinear"
        self.scales = list(scales)
        self.num_ch_enc = num_ch_enc
        self.num_ch_dec = (np.asarray(num_ch_enc) / 2).astype("int")
        blocks, self._index = [], {}
        for i in range(2, -1, -1):
            cin = int(num_ch_enc[-1]) if i == 2 else int(self.num_ch_dec[i + 1])
            self._index[("upconv", i, 0)] = len(blocks)
            blocks.append(ConvBlock(cin, int(self.num_ch_dec[i])))
            cin = int(self.num_ch_dec[i]) + (int(num_ch_enc[i - 1]) if use_skips and i > 0 else 0)
            self._index[("upconv", i, 1)] = len(blocks)
            blocks.append(ConvBlock(cin, int(self.num_ch_dec[i])))
        for s in self.scales:
            self._index[("dispconv", s)] = len(blocks)
            blocks.append(Conv3x3(int(self.num_ch_dec[s]), num_output_channels))
        self.decoder = nn.ModuleList(blocks)
        self.sigmoid = nn.Sigmoid()
        self.apply(self._init_weights)

    @staticmethod
    def _init_weights(m):
        if isinstance(m, (nn.Conv2d, nn.Linear)):
            nn.init.trunc_normal_(m.weight, std=0.02)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)

    takes_depth_range = True

    def forward(self, input_features, min_depth=0.1, max_depth=100.0):
        self.outputs = {}
        x = input_features[-1]
        for i in range(2, -1, -1):
            x = upsample(self.decoder[self._index[("upconv", i, 0)]](x), mode="bilinear")
            if self.use_skips and i > 0:
                x = torch.cat([x, input_features[i - 1]], 1)
            x = self.decoder[self._index[("upconv", i, 1)]](x)
            if i in self.scales:
                f = upsample(self.decoder[self._index[("dispconv", i)]](x), mode="bilinear")
                if i == 0 and f.is_cuda and f.dtype == torch.float32 and self.num_output_channels == 1 \
                        and not torch.is_autocast_enabled():
                    from .. import ops
                    disp, depth, part, sink = ops.disp_head(f, min_depth, max_depth, want_sink=True)
                    self.outputs[("disp", 0)], self.outputs[("depth", 0)] = disp, depth
                    self.outputs[("disp_mean_partials", 0)] = part
                    if sink is not None:
                        self.outputs[("disp_head_sink", 0)] = sink        # ops.HeadSink (not a tensor)
                else:
                    self.outputs[("disp", i)] = self.sigmoid(f)
        return self.outputs

"""Multi-frame fusion: warp neighbour features with the VFI flow, attach a NeRF-style
sin/cos embedding of the flow, mask-merge and mix with a 1x1 conv per scale.

API / state-dict keys follow reference networks/fusion_module.py (``Embedder`` 7-37,
``FusionModule`` 40-130; parameters under ``fusion_conv.<n>.conv.conv.*``, coarsest scale
first)."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ..layers import ConvBlock1x1
from .ifrnet import warp
from ..consts import const_tensor

# On a HIP device everything up to the 1x1 convolution of a level -- the two flow warps, the
# three resizes, the 84 sin/cos embedding channels, the mask merge and the concatenations -- is
# ONE kernel per level (ops.fusion_level / csrc/mvf_fusion.hip).  False (or CPU tensors, which
# only the build-container tests against the reference's module use) = the op-by-op form below.
FUSED_LEVELS = True


class Embedder:
    """x -> [x, sin(2^0 x), cos(2^0 x), ..., sin(2^(L-1) x), cos(2^(L-1) x)] on dim 1."""

    def __init__(self, input_dims=2, num_freqs=10):
        self.freqs = [2.0 ** i for i in range(num_freqs)]
        self.out_dim = input_dims * (1 + 2 * num_freqs)

    def embed(self, x):
        parts = [x]
        for f in self.freqs:
            parts += [torch.sin(x * f), torch.cos(x * f)]
        return torch.cat(parts, 1)


class FusionModule(nn.Module):
    def __init__(self, args, num_ch_enc, embed_multires=10):
        super().__init__()
        self.embedder_obj = Embedder(2, embed_multires)
        self.embed_multires = embed_multires
        self.num_ch_enc = num_ch_enc
        self.backbone = args.backbone
        n = len(num_ch_enc)
        self._slot = {i: n - 1 - i for i in range(n)}      # coarsest scale registered first
        self.fusion_conv = nn.ModuleList([
            ConvBlock1x1(2 * (int(num_ch_enc[i]) + self.embedder_obj.out_dim), int(num_ch_enc[i]))
            for i in range(n - 1, -1, -1)])

    def get_embedding_flow(self, x):
        outs = []
        for i in range(len(self.num_ch_enc)):
            x = F.interpolate(x, scale_factor=0.5, mode="bilinear", align_corners=False) * 0.5
            if i == 0 and self.backbone == "LiteMono":
                x = F.interpolate(x, scale_factor=0.5, mode="bilinear", align_corners=False) * 0.5
            outs.append(self.embedder_obj.embed(x))
        return outs

    def warp_features(self, features, flow):
        _, _, fh, fw = flow.shape
        out = []
        for feat in features:
            _, _, H, W = feat.shape
            fl = F.interpolate(flow, size=(H, W), mode="bilinear", align_corners=False)
            scale = const_tensor((W / fw, H / fh), fl.device, fl.dtype).view(1, 2, 1, 1)
            out.append(warp(feat, fl * scale))
        return out

    def forward(self, features, flows, merge_mask):
        feats_n1, feats_0, feats_p1 = features
        flow_0_n1, flow_0_p1 = flows
        if FUSED_LEVELS and feats_0[0].is_cuda:
            from .. import ops
            sizes = [tuple(f.shape[-2:]) for f in feats_0]
            preps = ops.fusion_prep(flow_0_n1, flow_0_p1, merge_mask, sizes, self.backbone == "LiteMono")
            return [self.fusion_conv[self._slot[i]](
                ops.fusion_level(feats_0[i].float(), feats_n1[i].float(), feats_p1[i].float(), preps[i],
                                 lists=(preps.lists, i)))
                for i in range(len(feats_0))]
        w_n1 = self.warp_features(feats_n1, flow_0_n1)
        w_p1 = self.warp_features(feats_p1, flow_0_p1)
        e_0 = self.get_embedding_flow(torch.zeros_like(flow_0_n1))
        e_n1 = self.get_embedding_flow(flow_0_n1)
        e_p1 = self.get_embedding_flow(flow_0_p1)
        outs = []
        for i in range(len(feats_0)):
            f0 = torch.cat([feats_0[i], e_0[i]], 1)
            fn = torch.cat([w_n1[i], e_n1[i]], 1)
            fp = torch.cat([w_p1[i], e_p1[i]], 1)
            _, _, H, W = f0.shape
            m = F.interpolate(merge_mask, size=(H, W), mode="bilinear", align_corners=False)
            merged = m * fn + (1 - m) * fp
            outs.append(self.fusion_conv[self._slot[i]](torch.cat([f0, merged], 1)))
        return outs

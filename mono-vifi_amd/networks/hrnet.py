"""HRNet-W18 multi-resolution backbone (plain PyTorch-ROCm module).

Restated from the architecture used by the reference (networks/hrnet_encoder.py:294-498 with
the W18 entry of networks/hrnet_config.py:120-152): a two-conv stride-4 stem, a bottleneck
stage, then three multi-branch stages (1/2/3 extra resolutions, 1/4/3 exchange modules, four
basic blocks per branch) whose branches are fused by summation after 1x1-conv + bilinear
upsampling (finer target) or strided 3x3 convs (coarser target).  Module names and therefore
state-dict keys match the reference (`conv1`, `layer1.N`, `transitionK.i`, `stageK.m.branches.b.n`,
`stageK.m.fuse_layers.i.j`), so `HRNet_W18_C_*.pth.tar` ImageNet weights and the reference's
checkpoints load.  The architecture is table-driven here instead of config-object driven
(yacs is absent)."""
import torch
import torch.nn as nn
import torch.nn.functional as F

import os

from .resnet import BasicBlock, Bottleneck

FUSED_SUM = os.environ.get("MVF_HRNET_FUSED_SUM", "1") != "0"      # developer knob for A/B timing (ops.sum_act)


def _resize_ac(x, size):
    """F.interpolate(x, size, mode="bilinear", align_corners=True) (reference:
    networks/hrnet_encoder.py:275-280); on the HIP device the element-parallel kernel
    (ops.resize_bilinear) with a deterministic gather backward."""
    from .. import layers
    if layers.FUSED_EPILOGUE and x.is_cuda and x.dtype == torch.float32 and not torch.is_autocast_enabled():
        from .. import ops
        return ops.resize_bilinear(x, size=size, align_corners=True)
    return F.interpolate(x, size=size, mode="bilinear", align_corners=True)

# (number of exchange modules, channels per branch) of stages 2..4
_WIDTHS = {"hrnet18": 18, "hrnet32": 32, "hrnet48": 48, "hrnet64": 64}
_MODULES = (1, 4, 3)
_BLOCKS_PER_BRANCH = 4


def _cbr(cin, cout, k, stride, relu=True):
    layers = [nn.Conv2d(cin, cout, k, stride, k // 2, bias=False), nn.BatchNorm2d(cout)]
    if relu:
        layers.append(nn.ReLU())
    return nn.Sequential(*layers)


class ExchangeModule(nn.Module):
    """Parallel residual branches followed by all-to-all fusion (the reference's
    HighResolutionModule, hrnet_encoder.py:138-287)."""

    def __init__(self, channels):
        super().__init__()
        n = len(channels)
        self.num_branches = n
        self.branches = nn.ModuleList([
            nn.Sequential(*[BasicBlock(c, c) for _ in range(_BLOCKS_PER_BRANCH)]) for c in channels])
        fuse = []
        for i in range(n):              # target branch
            row = []
            for j in range(n):          # source branch
                if j == i:
                    row.append(None)
                elif j > i:             # coarser source: 1x1 conv, upsampled in forward
                    row.append(_cbr(channels[j], channels[i], 1, 1, relu=False))
                else:                   # finer source: (i-j) strided 3x3 convs
                    steps = []
                    for k in range(i - j):
                        last = k == i - j - 1
                        steps.append(_cbr(channels[j], channels[i] if last else channels[j], 3, 2,
                                          relu=not last))
                    row.append(nn.Sequential(*steps))
            fuse.append(nn.ModuleList(row))
        self.fuse_layers = nn.ModuleList(fuse)
        self.relu = nn.ReLU()

    def forward(self, xs):
        xs = [branch(x) for branch, x in zip(self.branches, xs)]
        out = []
        for i in range(self.num_branches):
            terms = []
            for j in range(self.num_branches):
                if j == i:
                    term = xs[j]
                elif j > i:
                    term = _resize_ac(self.fuse_layers[i][j](xs[j]), xs[i].shape[-2:])
                else:
                    term = self.fuse_layers[i][j](xs[j])
                terms.append(term)
            t0 = terms[0]
            if (FUSED_SUM and len(terms) > 1 and t0.is_cuda and t0.dtype == torch.float32 and
                    not torch.is_autocast_enabled() and all(t.shape == t0.shape and t.dtype == t0.dtype for t in terms)):
                from .. import ops
                out.append(ops.sum_act(terms, "relu"))       # ((t0 + t1) + t2) + ... and the ReLU in one pass
                continue
            acc = None
            for term in terms:
                acc = term if acc is None else acc + term
            out.append(self.relu(acc))
        return out


class HighResolutionNet(nn.Module):
    def __init__(self, arch="hrnet18"):
        super().__init__()
        w = _WIDTHS[arch]
        self.conv1 = nn.Conv2d(3, 64, 3, 2, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.conv2 = nn.Conv2d(64, 64, 3, 2, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU()
        down = nn.Sequential(nn.Conv2d(64, 256, 1, bias=False), nn.BatchNorm2d(256))
        self.layer1 = nn.Sequential(Bottleneck(64, 64, 1, down), *[Bottleneck(256, 64) for _ in range(3)])
        prev = [256]
        for s, n_mod in enumerate(_MODULES, start=2):
            chans = [w * 2 ** i for i in range(s)]
            setattr(self, f"transition{s - 1}", self._transition(prev, chans))
            setattr(self, f"stage{s}", nn.Sequential(*[ExchangeModule(chans) for _ in range(n_mod)]))
            prev = chans
        self.num_ch_enc = [64] + prev

    @staticmethod
    def _transition(prev, cur):
        layers = []
        for i, c in enumerate(cur):
            if i < len(prev):
                layers.append(_cbr(prev[i], c, 3, 1) if prev[i] != c else None)
            else:       # new, coarser branch: strided convs from the coarsest existing one
                steps = []
                for j in range(i + 1 - len(prev)):
                    steps.append(_cbr(prev[-1], c if j == i - len(prev) else prev[-1], 3, 2))
                layers.append(nn.Sequential(*steps))
        return nn.ModuleList(layers)

    def forward(self, x):
        x = self.relu(self.bn1(self.conv1(x)))
        outputs = [x]                                   # stride-2 stem feature (64 ch)
        x = self.layer1(self.relu(self.bn2(self.conv2(x))))
        ys = [x]
        for s in (2, 3, 4):
            trans = getattr(self, f"transition{s - 1}")
            xs = []
            for i, t in enumerate(trans):
                src = ys[i] if i < len(ys) else ys[-1]
                xs.append(src if t is None else t(src))
            ys = getattr(self, f"stage{s}")(xs)
        return outputs + ys


def hrnet18(pretrained=False, **_):
    """HRNet-W18; ImageNet weights cannot be fetched here (no network): random init."""
    return HighResolutionNet("hrnet18")

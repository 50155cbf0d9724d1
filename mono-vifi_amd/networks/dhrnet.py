"""DHRNet depth network: HRNet-W18 encoder + dense multi-scale fusion decoder.

API / state-dict keys follow reference networks/DHRNet.py (``DepthEncoder`` 9-24,
``DepthDecoder`` 27-146): five encoder scales [64, 18, 36, 72, 144]; the decoder runs
parallel 3x3 conv blocks per scale and, level by level, folds every coarser scale into the
finer ones through nearest upsampling + 1x1 conv blocks and summation, then two more blocks
and a single-scale sigmoid disparity head.  Blocks are registered in ``decoder`` in the
reference's order so checkpoints load."""
import numpy as np
import torch
import torch.nn as nn

from ..layers import Conv3x3, ConvBlock, ConvBlock1x1, upsample
from .hrnet import hrnet18


class DepthEncoder(nn.Module):
    def __init__(self, num_layers=18, pretrained=False):
        super().__init__()
        assert num_layers == 18
        self.encoder = hrnet18(pretrained)
        self.num_ch_enc = np.array(self.encoder.num_ch_enc)

    def forward(self, x):
        self.features = self.encoder((x - 0.45) / 0.225)
        return self.features


class DepthDecoder(nn.Module):
    def __init__(self, num_ch_enc, scales=range(4), num_output_channels=1):
        super().__init__()
        self.num_output_channels = num_output_channels
        self.scales = scales
        self.num_ch_enc = num_ch_enc
        c = [int(v) for v in num_ch_enc]
        self._index, blocks = {}, []

        def add(key, module):
            self._index[key] = len(blocks)
            blocks.append(module)

        # levels 0..2: scales still alive at level L are 1 .. 4-L
        for level, top in ((0, 4), (1, 3), (2, 2)):
            for s in range(1, top + 1):
                add(("par", level, s), ConvBlock(c[s], c[s]))
            for src in range(2, top + 1):
                for dst in range(src - 1, 0, -1):
                    add(("mix", level, src, dst), ConvBlock1x1(c[src], c[dst]))
        add(("par", 3, 0), ConvBlock(c[0], c[0]))
        add(("par", 3, 1), ConvBlock(c[1], c[1]))
        add(("mix", 3, 1, 0), ConvBlock1x1(c[1], c[0]))
        add(("par", 4, 0), ConvBlock(c[0], 32))
        add(("par", 5, 0), ConvBlock(32, 16))
        add(("dispconv", 0), Conv3x3(16, num_output_channels))
        self.decoder = nn.ModuleList(blocks)
        self.sigmoid = nn.Sigmoid()

    def _m(self, *key):
        return self.decoder[self._index[key]]

    takes_depth_range = True

    def forward(self, input_features, min_depth=0.1, max_depth=100.0):
        self.outputs = {}
        feats = {s: input_features[s] for s in range(1, 5)}
        for level, top in ((0, 4), (1, 3), (2, 2)):
            par = {s: self._m("par", level, s)(feats[s]) for s in range(1, top + 1)}
            fused = {}
            for dst in range(1, top):
                acc = par[dst]
                for src in range(dst + 1, top + 1):
                    up = upsample(par[src], 2 ** (src - dst))
                    acc = acc + self._m("mix", level, src, dst)(up)
                fused[dst] = acc
            feats = fused
        d3_0 = self._m("par", 3, 0)(input_features[0])
        d3_1 = self._m("par", 3, 1)(feats[1])
        x = d3_0 + self._m("mix", 3, 1, 0)(upsample(d3_1, 2))
        x = upsample(self._m("par", 4, 0)(x), 2)
        x = self._m("par", 5, 0)(x)
        logit = self._m("dispconv", 0)(x)
        if logit.is_cuda and logit.dtype == torch.float32 and self.num_output_channels == 1 \
                and not torch.is_autocast_enabled():
            # sigmoid + disp_to_depth + the unit kernel's mean partials in one pass (ops.disp_head)
            from .. import ops
            disp, depth, part, sink = ops.disp_head(logit, min_depth, max_depth, want_sink=True)
            self.outputs[("disp", 0)], self.outputs[("depth", 0)] = disp, depth
            self.outputs[("disp_mean_partials", 0)] = part
            if sink is not None:
                self.outputs[("disp_head_sink", 0)] = sink        # ops.HeadSink (not a tensor)
        else:
            self.outputs[("disp", 0)] = self.sigmoid(logit)
        return self.outputs

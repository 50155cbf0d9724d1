"""Monodepth2-style depth network: ResNet encoder + U-Net decoder.

API and state-dict keys follow reference networks/monodepth2.py (``DepthEncoder`` 11-45,
``DepthDecoder`` 48-96): encoder parameters live under ``encoder.*``, decoder blocks under
``decoder.<n>.conv.conv.*`` in the order upconv(4,0), upconv(4,1), ..., upconv(0,1),
dispconv(s).  Plain PyTorch-ROCm modules: the convolutions are MIOpen / hipBLASLt GEMMs
(MFMA for the genuine dense contractions); nothing here is hand-written.
"""
import os

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from ..layers import Conv3x3, ConvBlock, conv_bias_act, upsample
from .resnet import ResNetTrunk, pyramid_features

# fp32 on the HIP device: the upsample + cat + reflection pad between the two convolutions of a
# decoder stage is one kernel (ops.up2cat_pad) and sigmoid + disp_to_depth one epilogue
# (ops.disp_head); False = the stock op-by-op form (also what CPU tensors take).
FUSED_GLUE = True
# the ConvBlocks' bias + ELU applied by the pad kernel of their consumer (ops.reflect_pad1_act / up2cat_pad_act) instead of
# by an epilogue pass of their own; "0" = the round-4 form (developer knob for the same-box A/B)
FUSE_EPILOGUE_INTO_PAD = os.environ.get("MVF_PAD_EPILOGUE", "1") != "0"


class DepthEncoder(nn.Module):
    def __init__(self, num_layers, pretrained=False):
        super().__init__()
        self.num_ch_enc = np.array([64, 64, 128, 256, 512])
        if num_layers > 34:
            self.num_ch_enc[1:] *= 4
        # `pretrained` ImageNet weights cannot be fetched (no network): random init
        self.encoder = ResNetTrunk(num_layers, 1)

    def forward(self, input_image):
        self.features = pyramid_features(self.encoder, input_image)
        return self.features


class DepthDecoder(nn.Module):
    def __init__(self, num_ch_enc, scales=range(4), num_output_channels=1, use_skips=True):
        super().__init__()
        self.num_output_channels = num_output_channels
        self.use_skips = use_skips
        self.scales = list(scales)
        self.num_ch_enc = num_ch_enc
        self.num_ch_dec = np.array([16, 32, 64, 128, 256])
        blocks, self._index = [], {}
        for i in range(4, -1, -1):
            cin = int(num_ch_enc[-1]) if i == 4 else int(self.num_ch_dec[i + 1])
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

    def _blk(self, *key):
        return self.decoder[self._index[key]]

    takes_depth_range = True

    def forward(self, input_features, min_depth=0.1, max_depth=100.0):
        """-> {("disp", s)}; on the fused path additionally ("depth", 0) and ("disp_mean_partials", 0)
        (by-products of the disparity-head epilogue; the reference's keys are unchanged)."""
        self.outputs = {}
        x = input_features[-1]
        fused = FUSED_GLUE and x.is_cuda and x.dtype == torch.float32 and not torch.is_autocast_enabled()
        if fused:
            from .. import ops
        # Round 5: a ConvBlock's bias + ELU is applied by the pad kernel of its (single) consumer: `pend` = (raw
        # convolution output, bias) of the block whose epilogue is still owed.  Shapes the fused pad kernels do not
        # take (W % 4, tiny planes) and every other consumer get the materialised tensor (`settle`).
        from .. import layers as _layers
        defer = fused and FUSE_EPILOGUE_INTO_PAD and _layers.FUSED_EPILOGUE and x.is_contiguous() and \
            all(f.is_contiguous() for f in input_features)
        pend = None

        def settle(p):
            return ops.bias_act(p[0], p[1], "elu", None, None, inplace=True)

        def raw_conv(conv, xp):
            return F.conv2d(xp, conv.weight, None, conv.stride, conv.padding, conv.dilation, conv.groups)
        for i in range(4, -1, -1):
            blk0 = self._blk("upconv", i, 0)
            if defer:
                # ---- upconv(i, 0): pad (+ the previous block's epilogue) -> conv, epilogue deferred
                c0 = blk0.conv.conv
                if pend is not None and ops.pad_act_ok(pend[0], *pend[0].shape[-2:]):
                    xp = ops.reflect_pad1_act(pend[0], pend[1])
                else:
                    xp = ops.reflect_pad1(settle(pend) if pend is not None else x)
                pend = (raw_conv(c0, xp), c0.bias)
            else:
                x = blk0(x)
            skip = input_features[i - 1] if (self.use_skips and i > 0) else None
            blk = self._blk("upconv", i, 1)
            if defer:
                y, b = pend
                if y.shape[-1] >= 2 and y.shape[-2] >= 2 and ops.pad_act_ok(y, 2 * y.shape[-2], 2 * y.shape[-1]) and \
                        (skip is None or (skip.is_contiguous() and skip.data_ptr() % 16 == 0)):
                    xp = ops.up2cat_pad_act(y, b, skip)
                elif y.shape[-1] >= 2 and y.shape[-2] >= 2:
                    xp = ops.up2cat_pad(settle(pend), skip)
                else:
                    xp = None
                if xp is not None:
                    pend = (raw_conv(blk.conv.conv, xp), blk.conv.conv.bias)
                else:
                    x = upsample(settle(pend))
                    if skip is not None:
                        x = torch.cat([x, skip], 1)
                    c1 = blk.conv.conv
                    pend = (raw_conv(c1, ops.reflect_pad1(x)), c1.bias)
            elif fused and x.shape[-1] >= 2 and x.shape[-2] >= 2:
                x = conv_bias_act(blk.conv.conv, ops.up2cat_pad(x, skip), "elu")
            else:
                x = upsample(x)
                if skip is not None:
                    x = torch.cat([x, skip], 1)
                x = blk(x)
            if defer and i in self.scales:
                # the disparity convolution's pad takes the epilogue too (the block has a second consumer, the next
                # level, only for i > 0: it re-applies the epilogue on its own load)
                dc = self._blk("dispconv", i).conv
                if ops.pad_act_ok(pend[0], *pend[0].shape[-2:]):
                    xp = ops.reflect_pad1_act(pend[0], pend[1])
                else:
                    # settle works in place and the next level (i > 0) applies the epilogue to the RAW output on its own
                    # load: activate a copy, leave `pend` raw (rebinding `pend` to the clone ran it twice)
                    xp = ops.reflect_pad1(settle((pend[0].clone(), pend[1])) if i > 0 else settle(pend))
                x_logit = ops.bias_act(raw_conv(dc, xp), dc.bias, "none", None, None, inplace=True)
            if i in self.scales:
                logit = x_logit if defer else self._blk("dispconv", i)(x)
                if fused and self.num_output_channels == 1:
                    disp, depth, part, sink = ops.disp_head(logit, min_depth, max_depth, want_depth=(i == 0),
                                                            want_sink=True)
                    self.outputs[("disp", i)] = disp
                    if i == 0:
                        self.outputs[("depth", 0)] = depth
                        self.outputs[("disp_mean_partials", 0)] = part
                        if sink is not None:
                            # (not a tensor: where the hot-path units leave their raw disparity gradients, ops.HeadSink)
                            self.outputs[("disp_head_sink", 0)] = sink
                else:
                    self.outputs[("disp", i)] = self.sigmoid(logit)
        return self.outputs

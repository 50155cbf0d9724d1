"""ResNet-18/34/50 feature extractor without torchvision (absent on both boxes).

State-dict keys are torchvision's (``conv1.weight``, ``bn1.*``, ``layerN.M.conv1.weight``,
``layerN.0.downsample.0.weight`` ...), so the reference's checkpoints load
(reference: networks/monodepth2.py:19-28, networks/posenet.py:10-52 build on
``torchvision.models.ResNet``).  The unused ImageNet ``fc`` head is dropped: in the
reference it forces ``find_unused_parameters=True`` and is all-reduced as zeros
(SURVEY.md section 3.5); checkpoint loading filters keys it does not own.
"""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

RESIDUAL_EPILOGUE = os.environ.get("MVF_FUSED_RESIDUAL", "1") != "0"      # developer knob for A/B timing


def _add_relu(out, idt):
    """relu(out + identity), the tail of every residual block: on the HIP device one epilogue
    pass (ops.bias_act with a residual, in place on the batch-norm output) instead of add + clamp."""
    from .. import layers
    if (RESIDUAL_EPILOGUE and layers.FUSED_EPILOGUE and out.is_cuda and out.dtype == torch.float32 and idt.dtype == torch.float32
            and out.shape == idt.shape and not torch.is_autocast_enabled()
            and out.is_contiguous() and idt.is_contiguous()):      # (channels_last: the stock ops, no NCHW copy)
        from .. import ops
        return ops.bias_act(out, None, "relu", res=idt, inplace=True)
    return F.relu(out + idt)


def _conv3(cin, cout, stride=1):
    return nn.Conv2d(cin, cout, 3, stride, 1, bias=False)


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, cin, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = _conv3(cin, planes, stride)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = _conv3(planes, planes)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        return _add_relu(out, idt)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, cin, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = _conv3(planes, planes, stride)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        return _add_relu(out, idt)


_SPECS = {18: (BasicBlock, [2, 2, 2, 2]), 34: (BasicBlock, [3, 4, 6, 3]),
          50: (Bottleneck, [3, 4, 6, 3]), 101: (Bottleneck, [3, 4, 23, 3]),
          152: (Bottleneck, [3, 8, 36, 3])}


class ResNetTrunk(nn.Module):
    """conv1/bn1/relu/maxpool/layer1..4 with ``num_input_images*3`` input channels."""

    def __init__(self, num_layers, num_input_images=1):
        super().__init__()
        if num_layers not in _SPECS:
            raise ValueError("{} is not a valid number of resnet layers".format(num_layers))
        block, layers = _SPECS[num_layers]
        self.inplanes = 64
        self.conv1 = nn.Conv2d(num_input_images * 3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        self.layer1 = self._make(block, 64, layers[0], 1)
        self.layer2 = self._make(block, 128, layers[1], 2)
        self.layer3 = self._make(block, 256, layers[2], 2)
        self.layer4 = self._make(block, 512, layers[3], 2)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def _make(self, block, planes, n, stride):
        down = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            down = nn.Sequential(nn.Conv2d(self.inplanes, planes * block.expansion, 1, stride, bias=False),
                                 nn.BatchNorm2d(planes * block.expansion))
        blocks = [block(self.inplanes, planes, stride, down)]
        self.inplanes = planes * block.expansion
        blocks += [block(self.inplanes, planes) for _ in range(1, n)]
        return nn.Sequential(*blocks)


STEM_POOL_KERNEL = os.environ.get("MVF_STEM_POOL", "1") != "0"      # developer knob for A/B timing
STEM_POOL_TAP = os.environ.get("MVF_STEM_POOL_TAP", "1") != "0"     # same: the tap form of the stem pool


def _stem_pool(trunk, f0, tap=True):
    """-> (pooled, f0).  The trunk's 3x3 / stride-2 max pool; on the HIP device (fp32) the byte-index gather pair
    (ops.maxpool3s2) instead of ATen's int64-index kernels.  The returned f0 is what the caller puts into the
    feature pyramid (`tap`: see ops.MaxPool3s2Tap)."""
    mp = trunk.maxpool
    if (STEM_POOL_KERNEL and f0.is_cuda and f0.dtype == torch.float32 and not torch.is_autocast_enabled()
            and f0.is_contiguous() and isinstance(mp, nn.MaxPool2d) and (mp.kernel_size, mp.stride, mp.padding, mp.dilation, mp.ceil_mode)
            == (3, 2, 1, 1, False)):
        from .. import ops
        if tap and STEM_POOL_TAP and torch.is_grad_enabled() and f0.requires_grad:
            return ops.maxpool3s2_tap(f0)     # (pooled, f0): f0's two gradients meet inside the pooling adjoint
        return ops.maxpool3s2(f0), f0
    return mp(f0), f0


def pyramid_features(trunk, image):
    """The 5-level feature list both encoders return (reference: monodepth2.py:33-45,
    posenet.py:80-93): colour normalisation (x-0.45)/0.225, then stem and 4 stages."""
    x = (image - 0.45) / 0.225
    f0 = trunk.relu(trunk.bn1(trunk.conv1(x)))
    pooled, f0 = _stem_pool(trunk, f0)
    f1 = trunk.layer1(pooled)
    f2 = trunk.layer2(f1)
    f3 = trunk.layer3(f2)
    f4 = trunk.layer4(f3)
    return [f0, f1, f2, f3, f4]

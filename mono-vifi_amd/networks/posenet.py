"""PoseNet: multi-image ResNet encoder + 4-conv pose decoder.

API / state-dict keys follow reference networks/posenet.py (``ResnetEncoder`` 55-93,
``PoseDecoder`` 96-137: ``net.0`` squeeze, ``net.1-3`` pose convs).  Output scale 0.01
(posenet.py:132)."""
import numpy as np
import torch
import torch.nn as nn

from .resnet import ResNetTrunk, pyramid_features


class ResnetEncoder(nn.Module):
    def __init__(self, num_layers, pretrained=False, num_input_images=1):
        super().__init__()
        self.num_ch_enc = np.array([64, 64, 128, 256, 512])
        if num_layers > 34:
            self.num_ch_enc[1:] *= 4
        self.encoder = ResNetTrunk(num_layers, num_input_images)

    def forward(self, input_image):
        self.features = pyramid_features(self.encoder, input_image)
        return self.features


class PoseDecoder(nn.Module):
    def __init__(self, num_ch_enc, num_input_features, num_frames_to_predict_for=None, stride=1):
        super().__init__()
        self.num_ch_enc = num_ch_enc
        self.num_input_features = num_input_features
        if num_frames_to_predict_for is None:
            num_frames_to_predict_for = num_input_features - 1
        self.num_frames_to_predict_for = num_frames_to_predict_for
        self.net = nn.ModuleList([
            nn.Conv2d(int(num_ch_enc[-1]), 256, 1),
            nn.Conv2d(num_input_features * 256, 256, 3, stride, 1),
            nn.Conv2d(256, 256, 3, stride, 1),
            nn.Conv2d(256, 6 * num_frames_to_predict_for, 1)])
        self.relu = nn.ReLU()

    def forward(self, input_features):
        feats = [self.relu(self.net[0](f[-1])) for f in input_features]
        out = torch.cat(feats, 1)
        out = self.relu(self.net[1](out))
        out = self.relu(self.net[2](out))
        out = self.net[3](out)
        out = out.mean(3).mean(2)
        out = 0.01 * out.view(-1, self.num_frames_to_predict_for, 1, 6)
        return out[..., :3], out[..., 3:]

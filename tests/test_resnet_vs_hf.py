"""CPU: the ResNet trunks (networks/resnet.py, torchvision's architecture and state-dict keys --
reference networks/monodepth2.py:19-28, networks/posenet.py:10-52) against an INDEPENDENT
implementation of the same network: Hugging Face transformers' ``ResNetModel`` (present here;
it is the implementation that loads torchvision's / timm's published ResNet weights, e.g.
microsoft/resnet-18, resnet-50).  torchvision itself is on neither box, so this is the closest pin
available for the trunk arithmetic: same weights in, same feature maps out -- stem, max pool,
BasicBlock / Bottleneck (stride on the 3x3 convolution, v1.5), the 1x1 down-sampling shortcuts."""
import pytest
import torch

transformers = pytest.importorskip("transformers")


def _to_hf_key(k):
    """torchvision key -> transformers key"""
    part = {"weight": "weight", "bias": "bias", "running_mean": "running_mean", "running_var": "running_var",
            "num_batches_tracked": "num_batches_tracked"}
    t = k.split(".")
    if t[0] == "conv1":
        return "embedder.embedder.convolution." + t[1]
    if t[0] == "bn1":
        return "embedder.embedder.normalization." + part[t[1]]
    stage, blk = int(t[0][5:]) - 1, int(t[1])
    base = f"encoder.stages.{stage}.layers.{blk}."
    if t[2] == "downsample":
        return base + ("shortcut.convolution." if t[3] == "0" else "shortcut.normalization.") + t[4]
    idx = int(t[2][-1]) - 1                      # conv1/bn1 -> 0, conv2/bn2 -> 1, conv3/bn3 -> 2
    return base + f"layer.{idx}." + ("convolution." if t[2].startswith("conv") else "normalization.") + t[3]


@pytest.mark.parametrize("layers", [18, 50])
@pytest.mark.parametrize("train", [False, True])
def test_resnet_trunk_equals_transformers_resnet(layers, train):
    from transformers import ResNetConfig, ResNetModel
    from mono_vifi_amd.networks.resnet import ResNetTrunk, pyramid_features
    torch.manual_seed(layers)
    trunk = ResNetTrunk(layers)
    for m in trunk.modules():                    # non-trivial batch-norm state
        if isinstance(m, torch.nn.BatchNorm2d):
            m.weight.data.uniform_(0.5, 1.5)
            m.bias.data.uniform_(-0.2, 0.2)
            m.running_mean.uniform_(-0.1, 0.1)
            m.running_var.uniform_(0.8, 1.2)
    basic = layers in (18, 34)
    cfg = ResNetConfig(num_channels=3, embedding_size=64,
                       hidden_sizes=[64, 128, 256, 512] if basic else [256, 512, 1024, 2048],
                       depths={18: [2, 2, 2, 2], 34: [3, 4, 6, 3], 50: [3, 4, 6, 3]}[layers],
                       layer_type="basic" if basic else "bottleneck", hidden_act="relu",
                       downsample_in_first_stage=False)
    hf = ResNetModel(cfg)
    sd = {_to_hf_key(k): v for k, v in trunk.state_dict().items()}
    missing, unexpected = hf.load_state_dict(sd, strict=False)
    assert not unexpected and all(k.startswith("pooler") for k in missing), (missing, unexpected)
    trunk.train(train)
    hf.train(train)
    x = torch.rand(2, 3, 64, 96)
    with torch.no_grad():
        ours = pyramid_features(trunk, x)
        xn = (x - 0.45) / 0.225                  # the encoders' colour normalisation (monodepth2.py:34)
        f0 = hf.embedder.embedder(xn)            # conv1 + bn1 + relu (before the pool)
        hs = hf(xn, output_hidden_states=True).hidden_states
    assert len(ours) == 5 and len(hs) == 5
    assert torch.allclose(ours[0], f0, atol=1e-5, rtol=1e-5)
    for a, b in zip(ours[1:], hs[1:]):
        assert a.shape == b.shape
        assert torch.allclose(a, b, atol=2e-5 * float(b.abs().max()) + 1e-6), float((a - b).abs().max())
